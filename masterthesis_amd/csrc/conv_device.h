// Device helpers shared by the gather-GEMM kernels.
#pragma once
#include "mt_common.h"
#include "conv_params.h"

template <bool BF16>
__device__ __forceinline__ void mma_chunk(f32x4& acc, const u32x4& a, const u32x4& b) {
  if constexpr (BF16) {
    acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, a),
                                                  __builtin_bit_cast(bf16x8, b), acc, 0, 0, 0);
  } else {
#pragma unroll
    for (int s = 0; s < 4; s++)
      acc = __builtin_amdgcn_mfma_f32_16x16x4f32(__uint_as_float(a[s]), __uint_as_float(b[s]),
                                                 acc, 0, 0, 0);
  }
}

// Filter-tap offsets of the parameter block through DWORD loads: with a wave-uniform index these are scalar loads from the kernel
// arguments; a dynamically indexed `short` is fetched with a VECTOR load (global_load_sshort) whose wait stands in front of a tile's
// first copy (and, in igemm_pipe_kernel's table loop, once per pair of taps: a chain of memory latencies per tile -- round 4).
template <class P>
__device__ __forceinline__ int tap_dh(const P& p, int t) {
  const int w = reinterpret_cast<const int*>(p.dh)[t >> 1];
  return (int)(short)((t & 1) ? (w >> 16) : w);
}
template <class P>
__device__ __forceinline__ int tap_dw(const P& p, int t) {
  const int w = reinterpret_cast<const int*>(p.dw)[t >> 1];
  return (int)(short)((t & 1) ? (w >> 16) : w);
}

// bijective XCD-aware remap: blocks b, b+8, ... share an XCD (and its L2); give each XCD a
// contiguous range of tiles so neighbouring pixel tiles (shared halo rows) hit the same L2.
__device__ __forceinline__ int xcd_remap(int b, int nblk) {
  const int q = nblk >> 3, r = nblk & 7, xcd = b & 7;
  const int base = xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q;
  return base + (b >> 3);
}


// ---- the straight-line epilogue of the gather-GEMM tile kernels (round 4) ----------------------------------------------------
// Stamps in the 256x256 patch-resident kernel showed its first epilogue -- per pixel fragment a divergent "is the pixel there"
// branch, per element a bias guard and the activation switch -- at 44.7k of the tile's 138k cycles: a build WITHOUT any epilogue
// ran the K1 launch in 55.4 us instead of 75.4, and the stores themselves were worth 3.6 us of those 20.  At every control-flow
// join the compiler had put its conservative `s_waitcnt vmcnt(0)`: each 16-byte store waited for the previous one to reach memory.
// Here nothing branches between the first and the last store: absent pixels / channels carry an out-of-range buffer offset (a
// store is dropped, a load returns zero), the bias comes through a range-checked buffer resource (zero past nbias / when null),
// ReLU / LeakyReLU are max / min arithmetic, the addend tile (the skip gradient riding on a data gradient) of the next channel
// group is in flight while the current group is stored, and the waits are counted.  Same values as the branchy code, element for
// element (same operations in the same order).
// Layout (the permuted weight staging of these kernels): fragment pair (2s, 2s + 1) of lane group fg holds the 8 consecutive
// channels cob + 32 s .. + 7 of pixel fragment b's pixel (lane fr).
//   ybase / ybytes: the output (or this split-K phase's slab) and its size (< 2 GiB: the caller checks, else it keeps its general
//   code); yo[b]: byte offset of the pixel's channel 0, or EPI_OOB; vm[b]: 1 / 0 (statistics of present pixels only);
//   red_lane: &red[(wpI * WT + wcI * WC + fg * 8) * 2] of the caller's [NWP][WT][2] statistics scratch (STATS only); AUX: cache
//   policy of the bf16 stores.
constexpr unsigned EPI_OOB = 0x80000000u;
template <bool BF16, int FC, int FP, bool HAS_ADD, bool STATS, bool TANH, int AUX>
__device__ __forceinline__ void epilogue_perm(const f32x4 (&acc)[FC][FP], char* ybase, unsigned ybytes, const float* bias, int nbias,
                                              const char* addend, int act, float slope, const unsigned (&yo)[FP],
                                              const float (&vm)[FP], unsigned cob, int Co, float* red_lane, int fr) {
  constexpr int SZ = BF16 ? 2 : 4;
  const __amdgpu_buffer_rsrc_t ry = __builtin_amdgcn_make_buffer_rsrc((void*)ybase, 0, ybytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t rb = __builtin_amdgcn_make_buffer_rsrc((void*)bias, 0, bias != nullptr ? (unsigned)nbias * 4u : 0u, 0x00020000);
  const __amdgpu_buffer_rsrc_t ra = __builtin_amdgcn_make_buffer_rsrc((void*)addend, 0, HAS_ADD ? ybytes : 0u, 0x00020000);
  const float neg = act == MT_ACT_RELU ? 0.f : (act == MT_ACT_LRELU ? slope : 1.f);
  const bool relu = act == MT_ACT_RELU;
  constexpr int NA = BF16 ? 1 : 2;                   // 16-byte pieces of 8 channels
  constexpr bool AHEAD = BF16;                       // the next group's addend in flight (fp32: 64 more registers -- not there)
  u32x4 adc[FP][NA], adn[FP][NA];
  float bvc[8], bvn[8];
  auto load_bias = [&](float (&dst)[8], int sp) {
#pragma unroll
    for (int e = 0; e < 8; e++) dst[e] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rb, (cob + sp * 32 + e) * 4u, 0, 0));
  };
  auto chan_off = [&](int b, int sp) {               // byte offset of the lane's 8 channels of group sp at pixel fragment b
    return (int)(cob + sp * 32) < Co ? yo[b] + (cob + sp * 32) * SZ : EPI_OOB;
  };
  auto load_add = [&](u32x4 (&dst)[FP][NA], int sp) {
#pragma unroll
    for (int b = 0; b < FP; b++)
#pragma unroll
      for (int h = 0; h < NA; h++) dst[b][h] = __builtin_amdgcn_raw_buffer_load_b128(ra, chan_off(b, sp) + h * 16, 0, 0);
  };
  load_bias(bvc, 0);
  if constexpr (HAS_ADD && AHEAD) load_add(adc, 0);
#pragma unroll
  for (int sp = 0; sp < FC / 2; sp++) {
    if (sp + 1 < FC / 2) load_bias(bvn, sp + 1);
    if constexpr (HAS_ADD) {
      if constexpr (AHEAD) { if (sp + 1 < FC / 2) load_add(adn, sp + 1); }
      else load_add(adc, sp);
    }
    float s1[8], s2[8];
#pragma unroll
    for (int e = 0; e < 8; e++) s1[e] = s2[e] = 0.f;
#pragma unroll
    for (int b = 0; b < FP; b++) {
      float v[8];
#pragma unroll
      for (int e = 0; e < 8; e++) {
        const float z = acc[2 * sp + (e >> 2)][b][e & 3] + bvc[e];
        if constexpr (TANH) v[e] = tanhf(z);
        else v[e] = fmaxf(z, 0.f) + (relu ? 0.f : neg * fminf(z, 0.f));        // (select: relu(-inf) = 0, not 0 * -inf)
      }
      if constexpr (HAS_ADD) {
        float ad[8];
        if constexpr (BF16) {
          Elem<true>::unpack(adc[b][0], ad);
        } else {
#pragma unroll
          for (int e = 0; e < 4; e++) { ad[e] = __uint_as_float(adc[b][0][e]); ad[4 + e] = __uint_as_float(adc[b][NA - 1][e]); }
        }
#pragma unroll
        for (int e = 0; e < 8; e++) v[e] += ad[e];
      }
      if constexpr (STATS) {
#pragma unroll
        for (int e = 0; e < 8; e++) {
          const float t = v[e] * vm[b];
          s1[e] += t;
          s2[e] += t * v[e];
        }
      }
      const unsigned off = chan_off(b, sp);
      if constexpr (BF16) {
        const u32x4 o = {pack2_bf16(v[0], v[1]), pack2_bf16(v[2], v[3]), pack2_bf16(v[4], v[5]), pack2_bf16(v[6], v[7])};
        __builtin_amdgcn_raw_buffer_store_b128(o, ry, off, 0, AUX);                   // (AUX 2 = nt: streaming store)
      } else {
        const u32x4 o0 = {__float_as_uint(v[0]), __float_as_uint(v[1]), __float_as_uint(v[2]), __float_as_uint(v[3])};
        const u32x4 o1 = {__float_as_uint(v[4]), __float_as_uint(v[5]), __float_as_uint(v[6]), __float_as_uint(v[7])};
        __builtin_amdgcn_raw_buffer_store_b128(o0, ry, off, 0, 0);
        __builtin_amdgcn_raw_buffer_store_b128(o1, ry, off + 16, 0, 0);
      }
    }
    if constexpr (STATS) {
#pragma unroll
      for (int e = 0; e < 8; e++) {
        const float t1 = row16_sum(s1[e]), t2 = row16_sum(s2[e]);
        if (fr == 0) {
          red_lane[(sp * 32 + e) * 2] = t1;
          red_lane[(sp * 32 + e) * 2 + 1] = t2;
        }
      }
    }
#pragma unroll
    for (int e = 0; e < 8; e++) bvc[e] = bvn[e];
    if constexpr (HAS_ADD && AHEAD) {
#pragma unroll
      for (int b = 0; b < FP; b++)
#pragma unroll
        for (int h = 0; h < NA; h++) adc[b][h] = adn[b][h];
    }
  }
}
// ... and the split-K partial of the same layout: the fp32 accumulators go to the phase's slab (yo in units of Co * 4 bytes)
template <int FC, int FP>
__device__ __forceinline__ void epilogue_perm_raw(const f32x4 (&acc)[FC][FP], char* ybase, unsigned ybytes, const unsigned (&yo)[FP],
                                                  unsigned cob, int Co) {
  const __amdgpu_buffer_rsrc_t ry = __builtin_amdgcn_make_buffer_rsrc((void*)ybase, 0, ybytes, 0x00020000);
#pragma unroll
  for (int sp = 0; sp < FC / 2; sp++)
#pragma unroll
    for (int b = 0; b < FP; b++) {
      const unsigned off = (int)(cob + sp * 32) < Co ? yo[b] + (cob + sp * 32) * 4u : EPI_OOB;
      __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, acc[2 * sp][b]), ry, off, 0, 0);
      __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, acc[2 * sp + 1][b]), ry, off + 16, 0, 0);
    }
}
