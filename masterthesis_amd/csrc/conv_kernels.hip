// Implicit-GEMM convolution kernels for gfx950 (MI355X): forward / data-gradient /
// transposed-conv share one gather-GEMM kernel (igemm_kernel); the weight gradient is a
// second kernel that reduces over pixels (wgrad_kernel).  NHWC activations with channel
// counts padded to 8, so every 16-byte chunk of the reduction dimension lies inside one
// filter tap.  MFMA: v_mfma_f32_16x16x32_bf16 (bf16 storage) or v_mfma_f32_16x16x4_f32
// (exact fp32 parity path), fp32 accumulation in both.
//
// Replaces: nn.ReflectionPad2d + nn.Conv2d (reference blocks.py:29-35), nn.ConvTranspose2d
// (blocks.py:73) and their autograd backward.
#include "mt_common.h"
#include <stdlib.h>
#include "conv_params.h"
#include <type_traits>

// ------------------------------------------------------------------------------------------
// gather-GEMM:  Y[co][pixel] = sum_{tap, ci} Wp[co][tap][ci] * X[n, f(ho)+dh(tap), f(wo)+dw(tap), ci]
// The weight tile is the MFMA A operand (rows = output channels) and the gathered pixel
// tile the B operand, so each lane ends up with 4 consecutive output channels of one pixel
// (a packed 8/16-byte NHWC store).
// ------------------------------------------------------------------------------------------
#include "conv_device.h"

#ifdef MT_STAMPS2
// diagnostic build only (tools/diag_build.sh NAME -DMT_STAMPS2 conv_kernels.hip): s_memtime stamps of wave 0 of every
// block of the 2-stage kernel -> tools/stamp_layer.py
__device__ unsigned long long mt_stamp_buf2[8 * 16384];
#define MT_STAMP2(i) do { if (threadIdx.x == 0 && blockIdx.x < 16384) mt_stamp_buf2[blockIdx.x * 8 + (i)] = __builtin_amdgcn_s_memtime(); } while (0)
extern "C" int mt_debug_stamps2(void* dst, size_t bytes) {
  return (int)hipMemcpyFromSymbol(dst, HIP_SYMBOL(mt_stamp_buf2), bytes < sizeof(mt_stamp_buf2) ? bytes : sizeof(mt_stamp_buf2));
}
#define MT_STAMP2_END() do { MT_STAMP2(4); asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); MT_STAMP2(5); } while (0)
#else
#define MT_STAMP2(i) do {} while (0)
#define MT_STAMP2_END() do {} while (0)
#endif

// Tile geometries (WT output channels x PT pixels per block, NT threads):
//   <128,128,256>, <64,128,256>, <32,128,256>, <16,128,256>: 4 waves, wave tile up to 64x64, 2 blocks/CU
//   <256,256,512>: 8 waves (2 x 4), wave tile 128 channels x 64 pixels, 128 KiB LDS, 1 block/CU -- 25 % fewer
//   LDS fragment bytes per MFMA and the pixel tile is fetched once for all 256 output channels.
template <bool BF16, int WT, int PT, int NT>
__global__ __launch_bounds__(512) void igemm_kernel(const IgemmParams p) {
  constexpr int WC = (WT == 256) ? 128 : ((WT >= 64) ? 64 : WT);   // wave tile: output channels
  constexpr int WP = (WT >= 128) ? 64 : 32;                        // wave tile: pixels
  constexpr int NWP = PT / WP;                  // waves along the pixel dimension
  constexpr int FC = WC / 16, FP = WP / 16;
  constexpr int SZ = BF16 ? 2 : 4;
  constexpr int RPP = NT / 8;                   // tile rows staged per pass of the whole block
  constexpr int NXL = PT / RPP;                 // pixel-tile LDS-DMA instructions per wave per k-step
  constexpr int WLD = (WT + RPP - 1) / RPP;     // weight-tile LDS-DMA instructions per wave per k-step


  __shared__ u32x4 sW[2][WT * 8];
  __shared__ u32x4 sX[2][PT * 8];
  __shared__ int sTap[64];

  MT_STAMP2(0);
  const int tid = threadIdx.x;
  const int lane = tid & 63, wv = tid >> 6;
  const int fr = lane & 15, fg = lane >> 4;
  const int wcI = wv / NWP, wpI = wv % NWP;

  const int nWT = (p.CoRows + WT - 1) / WT;
  int wg = xcd_remap(blockIdx.x, gridDim.x);
  int phi = 0;
  if (p.interleave) {
    phi = wg % p.nphase;
    wg = wg / p.nphase + p.ph[phi].blk0;
  } else {
    for (int i = 1; i < p.nphase; i++) phi = (wg >= p.ph[i].blk0) ? i : phi;
  }
  const IgemmPhase& ph = p.ph[phi];
  const int ph_ntaps = ph.ntaps, ph_Ho = ph.Ho, ph_Wo = ph.Wo, ph_M = ph.M;
  const int ph_nchunks = ph_ntaps * p.cpc;
  const char* const ph_w = p.w + ph.w_off;
  wg -= ph.blk0;
  // (one tap per iteration and wave, index wave-uniform: scalar loads from the kernel arguments.  Lane-indexed they were two vector
  //  loads whose wait stood between the launch and the first copy of every tile)
  for (int t = __builtin_amdgcn_readfirstlane(tid >> 6); t < ph_ntaps && t < 64; t += NT / 64) {     // (entries >= ntaps are never read)
    const int v = (tap_dh(p, ph.tap0 + t) << 16) | (tap_dw(p, ph.tap0 + t) & 0xffff);
    if ((tid & 63) == 0) sTap[t] = v;
  }
  const int wt = wg % nWT, pt = wg / nWT;

  // ---- per-thread staging coordinates (fixed over the k loop) ----
  // All global offsets are 32-bit byte offsets from the tensor base (host checks < 4 GiB), advanced
  // incrementally: inside one filter tap a k-step is just "+128 bytes"; the reflect / bounds math runs
  // only when the tap changes.  Keeps the VALU stream short next to the MFMAs.
  // Staging is LDS-DMA (buffer_load ... lds): a wave instruction writes 64 x 16 B = 8 tile rows
  // lane-linearly, so the XOR bank swizzle is applied on the SOURCE side: lane (row r, slot pc) fetches
  // logical chunk pc ^ (r & 7) and the fragment reads apply the same XOR.
  const int r0 = tid >> 3;
  const int c = (tid & 7) ^ (r0 & 7);
  const int wvu = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int HoWo = ph_Ho * ph_Wo;
  int hb[NXL], wb[NXL];
  unsigned ib[NXL];
  unsigned rvm = 0;
#pragma unroll
  for (int i = 0; i < NXL; i++) {
    const int m = pt * PT + r0 + RPP * i;
    const bool rv = m < ph_M;
    rvm |= (rv ? 1u : 0u) << i;
    const int mm = rv ? m : 0;
    const int n = mm / HoWo;
    const int rem = mm - n * HoWo;
    const int ho = rem / ph_Wo;
    const int wo = rem - ho * ph_Wo;
    hb[i] = ho * p.is;
    wb[i] = wo * p.is;
    ib[i] = (unsigned)n * (unsigned)(p.Hi * p.Wi) * (unsigned)p.Cib;
  }
  int q = c;
  int tap = q / p.cpc;
  int cq = q - tap * p.cpc;
  const int step_t = 8 / p.cpc, step_r = 8 % p.cpc;

  unsigned wo32[WLD];
  unsigned wokm = 0;
  // 16-byte epilogue stores (as in the ping-pong kernel): within each 32-row fragment pair LDS row (a&1)*16 + r holds
  // channel (r>>2)*8 + (a&1)*4 + (r&3), so fragments (2s, 2s+1) of lane group fg end up with the 8 CONSECUTIVE channels
  // s*32 + fg*8 .. +7 of a pixel
  constexpr bool PERM = (WT % 32 == 0);
#pragma unroll
  for (int i = 0; i < WLD; i++) {
    const int rl = r0 + RPP * i;
    const int rch = PERM ? ((rl & ~31) | ((((rl & 15) >> 2) << 3) | (((rl >> 4) & 1) << 2) | (rl & 3))) : rl;
    const int row = wt * WT + rch;
    const bool ok = (rl < WT) && (row < p.CoRows);
    wokm |= (ok ? 1u : 0u) << i;
    wo32[i] = ((unsigned)(ok ? row : 0) * (unsigned)ph.wrow + (unsigned)c) * 16u;
  }
  static_assert(NXL == 4, "every geometry stages 4 pixel rows per thread");
  unsigned xo32[4];   // (literal size: hipcc 7.2 drops the host stub when this lambda-captured array is NXL-sized)
  unsigned xokm = 0;
  auto retap = [&]() {
    xokm = 0;
#pragma unroll
    for (int i = 0; i < NXL; i++) xo32[i] = 0xfffffff0u;
    if (tap < ph_ntaps) {
      const int t = sTap[tap];
      const int dh = t >> 16, dw = (int)(short)(t & 0xffff);
#pragma unroll
      for (int i = 0; i < NXL; i++) {
        int hi = hb[i] + dh, wi = wb[i] + dw;
        bool ok = (rvm >> i) & 1u;
        if (p.pad_mode == MT_PAD_REFLECT) {
          hi = hi < 0 ? -hi : hi;
          hi = hi >= p.Hi ? 2 * (p.Hi - 1) - hi : hi;
          wi = wi < 0 ? -wi : wi;
          wi = wi >= p.Wi ? 2 * (p.Wi - 1) - wi : wi;
        } else {
          ok = ok && ((unsigned)hi < (unsigned)p.Hi) && ((unsigned)wi < (unsigned)p.Wi);
          hi = ok ? hi : 0;
          wi = ok ? wi : 0;
        }
        // invalid lanes get an out-of-range offset: the buffer load returns zeros for them
        xo32[i] = ok ? ib[i] + (unsigned)(hi * p.Wi + wi) * (unsigned)p.Cib + (unsigned)cq * 16u : 0xfffffff0u;
        xokm |= (ok ? 1u : 0u) << i;
      }
    }
  };

  const __amdgpu_buffer_rsrc_t rsx = __builtin_amdgcn_make_buffer_rsrc((void*)p.x, 0, p.x_bytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t rsw = __builtin_amdgcn_make_buffer_rsrc((void*)ph_w, 0, ph.w_bytes, 0x00020000);
  typedef __attribute__((address_space(3))) void* lds_ptr;
  // The asynchronous global->LDS copies of the next k-step are issued ONE instruction at a time,
  // interleaved with the MFMA groups of the current k-step (an LDS-DMA instruction costs ~60-180 issue
  // cycles, MI355X_MICROARCH.md; a burst of 8 in front of the MFMAs leaves the matrix pipe idle).
  // (the K tail needs no check on the weight side: there the pixel operand is zero -- tap >= ntaps -- and
  // reading into the next pack row / past the end (range-checked -> 0) only multiplies finite weights by 0)
  auto issue_piece = [&](int buf, int j) {           // j in [0, WLD + NXL)
    if (j < WLD) {
      const int i = j;
      if (wvu * 8 + RPP * i < WT) {
        char* lw = reinterpret_cast<char*>(&sW[buf][0]);
        const unsigned o = ((wokm >> i) & 1u) ? wo32[i] : 0xfffffff0u;
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rsw, (lds_ptr)(lw + (wvu * 8 + RPP * i) * 128), 16, o, 0, 0, 0);
      }
      wo32[i] += 128u;
    } else if (j < WLD + NXL) {
      const int i = j - WLD;
      char* lx = reinterpret_cast<char*>(&sX[buf][0]);
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rsx, (lds_ptr)(lx + (wvu * 8 + RPP * i) * 128), 16, xo32[i], 0, 0, 0);
    }
  };
  auto issue_end = [&]() {                            // advance to the next k-step
    q += 8;
    const int otap = tap;
    tap += step_t;
    cq += step_r;
    if (cq >= p.cpc) { cq -= p.cpc; tap++; }
    if (tap != otap) {
      retap();
    } else {
#pragma unroll
      for (int i = 0; i < NXL; i++) xo32[i] += ((xokm >> i) & 1u) ? 128u : 0u;
    }
  };
  auto issue = [&](int buf) {
#pragma unroll
    for (int j = 0; j < WLD + NXL; j++) issue_piece(buf, j);
    issue_end();
  };

  f32x4 acc[FC][FP];
#pragma unroll
  for (int a = 0; a < FC; a++)
#pragma unroll
    for (int b = 0; b < FP; b++) acc[a][b] = f32x4{0.f, 0.f, 0.f, 0.f};

  const int nk = (ph_nchunks + 7) >> 3;
  __syncthreads();  // sTap visible
  retap();
  MT_STAMP2(1);
  if (nk > 0) issue(0);

  // one k-step: MFMAs on buffer `cur`; if MORE, the copies of the next k-step go out one per MFMA group,
  // all within the first half of the step so they have the second half to land before the next barrier
  auto kstep = [&](int cur, auto more_tag) {
    constexpr bool MORE = decltype(more_tag)::value;
    constexpr int NPIECE = WLD + NXL;                 // DMA instructions of one k-step (<= 8)
    // drains this wave's LDS-DMA (vmcnt(0)) and joins the block: buffer `cur` is complete, and every wave
    // has finished reading buffer `cur^1` (previous k-step), so it may be overwritten now
    __syncthreads();
    // Fragment reads: the 4-wave geometries have the registers to issue BOTH halves of the k-step up front
    // (the second half's LDS latency hides under the first half's MFMAs); the 8-wave 256x256 geometry
    // (128 accumulator VGPRs) reads each half right before its MFMAs.
    constexpr int NB = (WT == 256) ? 1 : 2;
    u32x4 wf[NB][FC], xf[NB][FP];
    auto read_frags = [&](int kc, int slotb) {
#pragma unroll
      for (int a = 0; a < FC; a++) {
        const int row = wcI * WC + a * 16 + fr;
        wf[slotb][a] = sW[cur][row * 8 + ((kc * 4 + fg) ^ (row & 7))];
      }
#pragma unroll
      for (int b = 0; b < FP; b++) {
        const int row = wpI * WP + b * 16 + fr;
        xf[slotb][b] = sX[cur][row * 8 + ((kc * 4 + fg) ^ (row & 7))];
      }
    };
    if constexpr (NB == 2) { read_frags(0, 0); read_frags(1, 1); }
#pragma unroll
    for (int kc = 0; kc < 2; kc++) {
      if constexpr (NB == 1) read_frags(kc, 0);
      const int fb = (NB == 2) ? kc : 0;
#pragma unroll
      for (int a = 0; a < FC; a++) {
        const int slot = kc * FC + a;
        if constexpr (MORE) {
          if (FC >= 4) { if (slot < NPIECE) issue_piece(cur ^ 1, slot); }
          else {
#pragma unroll
            for (int j = slot * 4; j < slot * 4 + 4; j++) if (j < NPIECE) issue_piece(cur ^ 1, j);
          }
        }
        __builtin_amdgcn_s_setprio(1);
#pragma unroll
        for (int b = 0; b < FP; b++) mma_chunk<BF16>(acc[a][b], wf[fb][a], xf[fb][b]);
        __builtin_amdgcn_s_setprio(0);
      }
    }
    if constexpr (MORE) issue_end();
  };
#ifdef MT_STAMPS2
  if (nk > 1) { kstep(0, std::true_type{}); MT_STAMP2(2); }
  for (int ks = 1; ks + 1 < nk; ks++) kstep(ks & 1, std::true_type{});
#else
  for (int ks = 0; ks + 1 < nk; ks++) kstep(ks & 1, std::true_type{});
#endif
  if (nk > 0) kstep((nk - 1) & 1, std::false_type{});
  MT_STAMP2(3);

  // ---- epilogue: bias + activation, packed NHWC store (4 consecutive channels per lane) ----
  // per pixel fragment: output address (or null when the pixel is outside the problem / output)
  char* yp[FP];
#pragma unroll
  for (int b = 0; b < FP; b++) {
    const int m = pt * PT + wpI * WP + b * 16 + fr;
    yp[b] = nullptr;
    if (m < ph_M) {
      const int n = m / HoWo;
      const int rem = m - n * HoWo;
      const int ho = rem / ph_Wo;
      const int wo = rem - ho * ph_Wo;
      const int oh = ho * p.os + ph.oh0, ow = wo * p.os + ph.ow0;
      if ((unsigned)oh < (unsigned)p.Hout && (unsigned)ow < (unsigned)p.Wout)
        yp[b] = p.y + ph.y_off + (((size_t)n * p.Hout + oh) * p.Wout + ow) * p.Co * (p.raw ? 4 : SZ);
    }
  }
  // the straight-line epilogue (conv_device.h: epilogue_perm, round 4) for the permuted layout; tanh and outputs beyond 32-bit
  // buffer offsets keep the general code below
  if constexpr (PERM) {
    const size_t ybytes = (size_t)p.N * p.Hout * p.Wout * p.Co * (p.raw ? 4 : SZ);
    if (p.act != MT_ACT_TANH && ybytes < 0x7f000000ull) {
      char* const ybase = p.y + ph.y_off;
      unsigned yo[FP];
      float vm[FP];
#pragma unroll
      for (int b = 0; b < FP; b++) {
        yo[b] = yp[b] != nullptr ? (unsigned)(yp[b] - ybase) : EPI_OOB;
        vm[b] = yp[b] != nullptr ? 1.f : 0.f;
      }
      const unsigned cob = (unsigned)(wt * WT + wcI * WC + fg * 8);
      if (p.raw) {
        epilogue_perm_raw<FC, FP>(acc, ybase, (unsigned)ybytes, yo, cob, p.Co);
        MT_STAMP2_END();
        return;
      }
      const bool stats_f = p.stats != nullptr;
      float* const red_f = reinterpret_cast<float*>(&sW[0][0]);
      if (stats_f) __syncthreads();                      // every wave is done reading the last weight tile
      float* const red_lane = red_f + (wpI * WT + wcI * WC + fg * 8) * 2;
      if (stats_f)
        epilogue_perm<BF16, FC, FP, false, true, false, 0>(acc, ybase, (unsigned)ybytes, p.bias, p.nbias, nullptr, p.act, p.slope, yo,
                                                           vm, cob, p.Co, red_lane, fr);
      else
        epilogue_perm<BF16, FC, FP, false, false, false, 0>(acc, ybase, (unsigned)ybytes, p.bias, p.nbias, nullptr, p.act, p.slope, yo,
                                                            vm, cob, p.Co, red_lane, fr);
      if (stats_f) {
        __syncthreads();
        const int m0 = pt * PT;
        if (m0 < ph_M) {
          const int n0 = m0 / HoWo;
          for (int idx = tid; idx < WT * 2; idx += NT) {
            const int col = idx >> 1;
            if (wt * WT + col < p.Co) {
              float t = 0.f;
#pragma unroll
              for (int w = 0; w < NWP; w++) t += red_f[(w * WT + col) * 2 + (idx & 1)];
              atomicAdd(p.stats + ((size_t)n0 * p.Co + wt * WT + col) * 2 + (idx & 1), t);
            }
          }
        }
      }
      MT_STAMP2_END();
      return;
    }
  }
  if (p.raw) {
    // split-K partial: the fp32 accumulators go to this phase's slab; bias / activation are applied by the finish
    if constexpr (PERM) {
#pragma unroll
      for (int sp = 0; sp < FC / 2; sp++) {
        const int co = wt * WT + wcI * WC + sp * 32 + fg * 8;
        if (co >= p.Co) continue;
#pragma unroll
        for (int b = 0; b < FP; b++)
          if (yp[b] != nullptr) {
            *reinterpret_cast<f32x4*>(yp[b] + (size_t)co * 4) = acc[2 * sp][b];
            *reinterpret_cast<f32x4*>(yp[b] + (size_t)co * 4 + 16) = acc[2 * sp + 1][b];
          }
      }
    } else {
#pragma unroll
      for (int a = 0; a < FC; a++) {
        const int co = wt * WT + wcI * WC + a * 16 + fg * 4;
        if (co >= p.Co) continue;
#pragma unroll
        for (int b = 0; b < FP; b++)
          if (yp[b] != nullptr) *reinterpret_cast<f32x4*>(yp[b] + (size_t)co * 4) = acc[a][b];
      }
    }
    return;
  }
  // fused InstanceNorm statistics (p.stats != null): all PT pixels of the block lie in ONE image (host
  // guarantees Ho*Wo % 256 == 0).  Sum over the wave's pixel fragments in registers, over the 16 pixel lanes
  // with xor-shuffles, over the block's pixel waves through LDS (the weight tile buffer is free now), then ONE
  // full-width fp32 atomic instruction per 64 (channel, moment) pairs -- 4-lane atomics issued per wave
  // and channel cost ~25 us per launch at one atomic instruction per ~50 ns per CU.
  const bool do_stats = p.stats != nullptr;
  float* red = reinterpret_cast<float*>(&sW[0][0]);     // [NWP][WT][2]
  if (do_stats) __syncthreads();                        // every wave is done reading the last weight tile
  if constexpr (PERM) {
#pragma unroll
    for (int sp = 0; sp < FC / 2; sp++) {
      const int col = wcI * WC + sp * 32 + fg * 8;
      const int co = wt * WT + col;
      if (co >= p.Co) continue;
      float bv[8], s1[8], s2[8];
#pragma unroll
      for (int e = 0; e < 8; e++) {
        bv[e] = (p.bias != nullptr && (co + e) < p.nbias) ? p.bias[co + e] : 0.f;
        s1[e] = s2[e] = 0.f;
      }
#pragma unroll
      for (int b = 0; b < FP; b++) {
        if (yp[b] == nullptr) continue;
        float v[8];
#pragma unroll
        for (int e = 0; e < 8; e++) {
          v[e] = act_apply(acc[2 * sp + (e >> 2)][b][e & 3] + bv[e], p.act, p.slope);
          s1[e] += v[e];
          s2[e] += v[e] * v[e];
        }
        if constexpr (BF16) {
          u32x4 o = {pack2_bf16(v[0], v[1]), pack2_bf16(v[2], v[3]), pack2_bf16(v[4], v[5]), pack2_bf16(v[6], v[7])};
          *reinterpret_cast<u32x4*>(yp[b] + (size_t)co * 2) = o;
        } else {
          f32x4 o0 = {v[0], v[1], v[2], v[3]}, o1 = {v[4], v[5], v[6], v[7]};
          *reinterpret_cast<f32x4*>(yp[b] + (size_t)co * 4) = o0;
          *reinterpret_cast<f32x4*>(yp[b] + (size_t)co * 4 + 16) = o1;
        }
      }
      if (do_stats) {
#pragma unroll
        for (int e = 0; e < 8; e++) {
          const float t1 = row16_sum(s1[e]), t2 = row16_sum(s2[e]);
          if (fr == 0) {
            red[((wpI * WT) + col + e) * 2] = t1;
            red[((wpI * WT) + col + e) * 2 + 1] = t2;
          }
        }
      }
    }
  } else {
#pragma unroll
  for (int a = 0; a < FC; a++) {
    const int col = wcI * WC + a * 16 + fg * 4;         // channel within the block tile
    const int co = wt * WT + col;
    if (co >= p.Co) continue;
    float bv[4];
#pragma unroll
    for (int j = 0; j < 4; j++) bv[j] = (p.bias != nullptr && (co + j) < p.nbias) ? p.bias[co + j] : 0.f;
    float s1[4] = {0.f, 0.f, 0.f, 0.f}, s2[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int b = 0; b < FP; b++) {
      if (yp[b] == nullptr) continue;
      float v[4];
#pragma unroll
      for (int j = 0; j < 4; j++) {
        v[j] = act_apply(acc[a][b][j] + bv[j], p.act, p.slope);
        s1[j] += v[j];
        s2[j] += v[j] * v[j];
      }
      if constexpr (BF16) {
        u32x2 o = {pack2_bf16(v[0], v[1]), pack2_bf16(v[2], v[3])};
        *reinterpret_cast<u32x2*>(yp[b] + (size_t)co * 2) = o;
      } else {
        f32x4 o = {v[0], v[1], v[2], v[3]};
        *reinterpret_cast<f32x4*>(yp[b] + (size_t)co * 4) = o;
      }
    }
    if (do_stats) {
#pragma unroll
      for (int j = 0; j < 4; j++) {
        const float t1 = row16_sum(s1[j]), t2 = row16_sum(s2[j]);
        if (fr == 0) {
          red[((wpI * WT) + col + j) * 2] = t1;
          red[((wpI * WT) + col + j) * 2 + 1] = t2;
        }
      }
    }
  }
  }
  if (do_stats) {
    __syncthreads();
    const int m0 = pt * PT;
    if (m0 < ph_M) {
      const int n0 = m0 / HoWo;
      for (int idx = tid; idx < WT * 2; idx += NT) {
        const int col = idx >> 1;
        if (wt * WT + col < p.Co) {
          float t = 0.f;
#pragma unroll
          for (int w = 0; w < NWP; w++) t += red[(w * WT + col) * 2 + (idx & 1)];
          atomicAdd(p.stats + ((size_t)n0 * p.Co + wt * WT + col) * 2 + (idx & 1), t);
        }
      }
    }
  }
#ifdef MT_STAMPS2
  MT_STAMP2(4);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  MT_STAMP2(5);
#endif
}

template <bool BF16>
int launch_igemm_pipe_t(IgemmParams& p, int WT, int PT, int total, hipStream_t s);   // conv_pipe_kernel.hip
int launch_igemm_persist(IgemmParams& p, int WT, int total, hipStream_t s, bool dry);          // conv_persist_kernel.hip
int launch_igemm_patch(IgemmParams& p, hipStream_t s, bool dry);                                // conv_patch_kernel.hip
template <bool BF16>
int launch_igemm_pipe_patch_t(IgemmParams& p, int total, hipStream_t s, bool dry);               // conv_pipe_patch_kernel.hip
int launch_igemm_wsreg(IgemmParams& p, hipStream_t s, bool dry);                                // conv_wsreg_kernel.hip

// dry: launch nothing; return 100 if a kernel that honours IgemmParams::y2 (persistent / patch-resident) would take the problem, 101 otherwise
template <bool BF16>
static int launch_igemm_t(IgemmParams& p, hipStream_t s, bool dry = false) {
  // tile geometry: 256x256 tiles when a single-phase problem gives (a multiple of) one block per CU
  // MT_IGEMM_FORCE (diagnostics, tools/layer_table.py): 1 = 2-stage kernel only, 2 = 4-wave ring variant wherever legal,
  // 3 = 128x512 tiles wherever legal, 4 = 256x256 tiles wherever legal, 5 = 64-channel tiles for Cout % 128 == 0
  static const int force = getenv("MT_IGEMM_FORCE") ? atoi(getenv("MT_IGEMM_FORCE")) : 0;
  // short-K / huge-M layers with 64 or 128 output channels: weights stationary in registers, the tile's input patch in LDS,
  // one barrier per tile (conv_wsreg_kernel.hip, round 4)
  if constexpr (BF16) {
    if (force == 0) {
      const int r = launch_igemm_wsreg(p, s, dry);
      if (r >= 0) return r;
    }
  }
  int PT = 128, WT = p.CoRows > 64 ? 128 : (p.CoRows > 32 ? 64 : (p.CoRows > 16 ? 32 : 16));
  // (the ping-pong kernel steps whole 4-chunk k-steps inside a tap and marks zero lanes with offsets >= 2 GiB)
  const bool pipe_ok = p.cpc % 4 == 0 && p.ph[0].ntaps <= 25 /* MT_PIPE_MAX_TAPS */ && p.x_bytes < 0x7f000000u &&
                       p.ph[0].w_bytes < 0x7f000000u;
  if (p.nphase == 1 && !p.raw && p.CoRows % 256 == 0 && pipe_ok) {
    const int n256 = cdiv(p.ph[0].M, 256) * (p.CoRows / 256);
    if ((n256 >= 192 && (n256 % 256 == 0 || n256 >= 1024)) || force == 4) { PT = 256; WT = 256; }
    if (force == 1 || force == 2 || force == 3) { PT = 128; WT = 128; }
  }
  // stride-1 gathers whose pixel tile's input patch fits LDS: the patch-resident kernel (conv_patch_kernel.hip) stages
  // every pixel once per channel slice instead of once per tap, and computes the four sub-pixel phases of a scatter
  // form from one patch
  if constexpr (BF16) {
    if (!(PT == 256 && WT == 256) && force == 0) {
      const int r = launch_igemm_patch(p, s, dry);
      if (r >= 0) return r;
    }
  }
  // 128 couts x 512 pixels ping-pong tiles for the Cout = 128 layers (any number of phases, <= 9 taps each)
  if (WT == 128 && PT == 128 && !p.raw && p.stats == nullptr && p.CoRows % 128 == 0 && p.cpc % 4 == 0 &&
      p.x_bytes < 0x7f000000u) {
    bool ok = true;
    int n512 = 0;
    for (int i = 0; i < p.nphase; i++) {
      ok = ok && p.ph[i].ntaps <= 9 && p.ph[i].w_bytes < 0x7f000000u;
      n512 += cdiv(p.ph[i].M, 512) * (p.CoRows / 128);
    }
    if (ok && ((n512 >= 224 && (n512 % 256 == 0 || n512 >= 768)) || force == 3)) { PT = 512; WT = 128; }
    if (force == 1 || force == 2) { PT = 128; WT = 128; }
  }
  // latency-bound small launches (e.g. the ring GEMM of the stride-1 data gradient, 68 blocks): 64-channel tiles
  // double the number of blocks that share the serial k loop (-0.3 ms per step)
  if (WT == 128 && PT == 128 && !p.raw && p.CoRows % 128 == 0) {
    int t128 = 0;
    for (int i = 0; i < p.nphase; i++) t128 += cdiv(p.ph[i].M, PT) * cdiv(p.CoRows, 128);
    if (t128 <= 128 || force == 5) WT = 64;
  }
  int total = 0;
  for (int i = 0; i < p.nphase; i++) {
    p.ph[i].blk0 = total;
    p.ph[i].nblk = cdiv(p.ph[i].M, PT) * cdiv(p.CoRows, WT);
    total += p.ph[i].nblk;
  }
  p.interleave = 0;
  if (p.nphase > 1) {       // (sub-pixel phases / split-K slices of one input region side by side: -0.25 ms per step)
    bool eq = true;
    for (int i = 1; i < p.nphase; i++) eq = eq && p.ph[i].nblk == p.ph[0].nblk;
    p.interleave = eq ? 1 : 0;
  }
  if (total == 0) return dry ? 101 : 0;
  // 256x256 and 128x512 tiles run the ping-pong pipelined kernel (conv_pipe_kernel.hip)
  if (PT == 256 && WT == 256) {
    // ... with the pixel operand resident in LDS as a patch where the map allows it (conv_pipe_patch_kernel.hip)
    const int r = launch_igemm_pipe_patch_t<BF16>(p, total, s, dry);
    if (r >= 0) return r;
  }
  if (p.fold || p.addend != nullptr || p.bstat_x != nullptr) {
    if (dry) return 101;
    mt_set_error("igemm: the in-operand reflection fold / epilogue addend need the patch-resident 256x256 kernel (mt_igemm_fold_ok)");
    return 1;
  }
  if (PT >= 256) return dry ? 101 : launch_igemm_pipe_t<BF16>(p, WT, PT, total, s);
  // launches that do not fill the chip (latency-bound k loops): the 4-wave ring variant
  {
    bool ok = (WT == 128 || WT == 64) && !p.raw && p.cpc % 4 == 0 && p.x_bytes < 0x7f000000u &&
              (total <= 256 || force == 2) && force != 1;
    for (int i = 0; i < p.nphase && ok; i++) ok = p.ph[i].ntaps <= 9 && p.ph[i].w_bytes < 0x7f000000u;
    if (ok) return dry ? 101 : launch_igemm_pipe_t<BF16>(p, WT, PT, total, s);
  }
  // several tiles per workgroup slot: the persistent kernel pipelines across tiles (conv_persist_kernel.hip)
  if constexpr (BF16) {
    if (force != 1) {
      const int r = launch_igemm_persist(p, WT, total, s, dry);
      if (r >= 0) return r;
    }
  }
  if (dry) return 101;
  if (WT == 128) hipLaunchKernelGGL((igemm_kernel<BF16, 128, 128, 256>), dim3(total), dim3(256), 0, s, p);
  else if (WT == 64) hipLaunchKernelGGL((igemm_kernel<BF16, 64, 128, 256>), dim3(total), dim3(256), 0, s, p);
  else if (WT == 32) hipLaunchKernelGGL((igemm_kernel<BF16, 32, 128, 256>), dim3(total), dim3(256), 0, s, p);
  else hipLaunchKernelGGL((igemm_kernel<BF16, 16, 128, 256>), dim3(total), dim3(256), 0, s, p);
  MT_LAUNCH_CHECK();
  return 0;
}

// ------------------------------------------------------------------------------------------
// Few output pixels x few output channels x a long reduction (the discriminators' 4x4 "valid" class head: 32
// pixels, 2 channels, K = 16 taps x 1024): as a tile GEMM this is ONE block walking 256 k-steps (150 us for 4
// MFLOP).  Here one block per output pixel streams its K-long gathered row against the <= 8 weight rows.
// ------------------------------------------------------------------------------------------
template <bool BF16>
__global__ __launch_bounds__(256) void thin_dot_kernel(const IgemmParams p) {
  constexpr int V = Elem<BF16>::V, SZ = Elem<BF16>::SZ;
  __shared__ float red[8][256];
  const IgemmPhase& ph = p.ph[0];
  const int m = blockIdx.x;
  const int HoWo = ph.Ho * ph.Wo;
  const int n = m / HoWo;
  const int rem = m - n * HoWo;
  const int ho = rem / ph.Wo, wo = rem - ho * ph.Wo;
  const int nchunks = ph.ntaps * p.cpc;
  const u32x4* __restrict__ wbase = reinterpret_cast<const u32x4*>(p.w + ph.w_off);
  float acc[8];
#pragma unroll
  for (int r = 0; r < 8; r++) acc[r] = 0.f;
  for (int q = threadIdx.x; q < nchunks; q += 256) {
    const int t = q / p.cpc, cq = q - t * p.cpc;
    int hi = ho * p.is + p.dh[ph.tap0 + t], wi = wo * p.is + p.dw[ph.tap0 + t];
    bool ok = true;
    if (p.pad_mode == MT_PAD_REFLECT) {
      hi = hi < 0 ? -hi : hi;
      hi = hi >= p.Hi ? 2 * (p.Hi - 1) - hi : hi;
      wi = wi < 0 ? -wi : wi;
      wi = wi >= p.Wi ? 2 * (p.Wi - 1) - wi : wi;
    } else {
      ok = ((unsigned)hi < (unsigned)p.Hi) && ((unsigned)wi < (unsigned)p.Wi);
    }
    if (!ok) continue;
    const u32x4 xv = *reinterpret_cast<const u32x4*>(p.x + ((size_t)(n * p.Hi + hi) * p.Wi + wi) * p.Cib + (size_t)cq * 16);
    float xf[V];
    Elem<BF16>::unpack(xv, xf);
#pragma unroll
    for (int r = 0; r < 8; r++) {
      if (r < p.CoRows) {
        float wf[V];
        Elem<BF16>::unpack(wbase[(size_t)r * ph.wrow + q], wf);
#pragma unroll
        for (int e = 0; e < V; e++) acc[r] += xf[e] * wf[e];
      }
    }
  }
#pragma unroll
  for (int r = 0; r < 8; r++) red[r][threadIdx.x] = wave_sum(acc[r]);
  __syncthreads();
  if (threadIdx.x < 8 && (int)threadIdx.x < p.Co) {
    const int co = threadIdx.x;
    float v = red[co][0] + red[co][64] + red[co][128] + red[co][192];
    if (p.bias != nullptr && co < p.nbias) v += p.bias[co];
    v = act_apply(v, p.act, p.slope);
    const int oh = ho * p.os + ph.oh0, ow = wo * p.os + ph.ow0;
    char* y = p.y + (((size_t)n * p.Hout + oh) * p.Wout + ow) * p.Co * SZ + (size_t)co * SZ;
    if constexpr (BF16) *reinterpret_cast<unsigned short*>(y) = f32_to_bf16_bits(v);
    else *reinterpret_cast<float*>(y) = v;
  }
}
// (round 3: 512 -> 128 chunks: the 1x1 heads of the multi-scale discriminators -- 2048 -> 1 / 2 channels on 1..16 pixels per image,
//  24 launches per step -- were ONE 16-channel tile block walking 32 k-steps, 19-24 us each)
#ifndef MT_THIN_DOT_MIN_CHUNKS
#define MT_THIN_DOT_MIN_CHUNKS 128
#endif
static bool thin_dot_ok(const IgemmParams& p) {
  return p.nphase == 1 && !p.raw && p.stats == nullptr && p.CoRows <= 8 && p.Co <= 8 && p.ph[0].M <= 1024 &&
         p.ph[0].ntaps * p.cpc >= MT_THIN_DOT_MIN_CHUNKS && p.os == 1;
}

template <bool BF16>
__global__ void splitk_finish_kernel(const f32x4* __restrict__ slabs, int nsplit, long total4, long slab4,
                                     const float* __restrict__ bias, int nbias, int Cp, void* __restrict__ y, int act,
                                     float slope) {
  for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < total4; i += (long)gridDim.x * blockDim.x) {
    f32x4 a = slabs[i];
    for (int k = 1; k < nsplit; k++) a += slabs[i + (long)k * slab4];
    const int c = (int)((i * 4) % Cp);
    float v[4];
#pragma unroll
    for (int j = 0; j < 4; j++) {
      const float b = (bias != nullptr && c + j < nbias) ? bias[c + j] : 0.f;
      v[j] = act_apply(a[j] + b, act, slope);
    }
    if constexpr (BF16) {
      u32x2 o = {pack2_bf16(v[0], v[1]), pack2_bf16(v[2], v[3])};
      reinterpret_cast<u32x2*>(y)[i] = o;
    } else {
      f32x4 o = {v[0], v[1], v[2], v[3]};
      reinterpret_cast<f32x4*>(y)[i] = o;
    }
  }
}
int mt_launch_splitk_finish(int dtype, const float* slabs, int nsplit, long total, const float* bias, int nbias, int Cp,
                            void* y, int act, float slope, hipStream_t s) {
  if (total == 0) return 0;
  const long total4 = total / 4;     // Cp is a multiple of 8
  const int blocks = (int)min((long)4096, (total4 + 255) / 256);
  if (dtype == MT_BF16)
    hipLaunchKernelGGL((splitk_finish_kernel<true>), dim3(blocks), dim3(256), 0, s, (const f32x4*)slabs, nsplit, total4,
                       total4, bias, nbias, Cp, y, act, slope);
  else
    hipLaunchKernelGGL((splitk_finish_kernel<false>), dim3(blocks), dim3(256), 0, s, (const f32x4*)slabs, nsplit, total4,
                       total4, bias, nbias, Cp, y, act, slope);
  MT_LAUNCH_CHECK();
  return 0;
}

int mt_launch_igemm(int dtype, const IgemmParams& p, hipStream_t s) {
  MT_CHECK(p.nphase >= 1 && p.nphase <= MT_MAX_PHASES, "igemm: bad phase count %d", p.nphase);
  MT_CHECK(p.cpc >= 1, "igemm: bad chunks-per-tap %d", p.cpc);
  MT_CHECK((double)p.N * p.Hi * p.Wi * p.Cib < 4294967000.0, "igemm: input tensor exceeds 4 GiB (32-bit offsets)");
  int taps = 0;
  for (int i = 0; i < p.nphase; i++) {
    MT_CHECK(p.ph[i].wrow >= p.ph[i].ntaps * p.cpc, "igemm: weight row shorter than the phase's taps");
    MT_CHECK((double)p.ph[i].w_off + (double)p.ph[i].w_bytes < 4294967000.0, "igemm: weight pack exceeds 4 GiB");
    taps += p.ph[i].ntaps;
  }
  MT_CHECK(taps <= MT_MAX_TAPS, "igemm: %d taps > %d", taps, MT_MAX_TAPS);
  IgemmParams q = p;
  q.x_bytes = (unsigned)((size_t)p.N * p.Hi * p.Wi * p.Cib);
  q.korder = 1;       // ping-pong kernels walk K channel-slice-major (see conv_pipe_kernel.hip: L2-resident re-reads)
  if (thin_dot_ok(q)) {
    if (q.ph[0].M == 0) return 0;
    if (dtype == MT_BF16) hipLaunchKernelGGL((thin_dot_kernel<true>), dim3(q.ph[0].M), dim3(256), 0, s, q);
    else hipLaunchKernelGGL((thin_dot_kernel<false>), dim3(q.ph[0].M), dim3(256), 0, s, q);
    MT_LAUNCH_CHECK();
    return 0;
  }
  MT_CHECK(q.y2 == nullptr || mt_igemm_would_persist(dtype, p), "igemm: a second destination needs the persistent kernel");
  return dtype == MT_BF16 ? launch_igemm_t<true>(q, s) : launch_igemm_t<false>(q, s);
}
bool mt_igemm_fold_ok(int dtype, const IgemmParams& p) {
  if (p.nphase != 1 || p.cpc < 1 || !p.fold || (double)p.N * p.Hi * p.Wi * p.Cib >= 4294967000.0) return false;
  IgemmParams q = p;
  q.x_bytes = (unsigned)((size_t)p.N * p.Hi * p.Wi * p.Cib);
  q.korder = 1;
  return (dtype == MT_BF16 ? launch_igemm_t<true>(q, nullptr, true) : launch_igemm_t<false>(q, nullptr, true)) == 102;
}
bool mt_igemm_would_persist(int dtype, const IgemmParams& p) {
  if (dtype != MT_BF16 || p.nphase < 1 || p.cpc < 1) return false;
  IgemmParams q = p;
  q.x_bytes = (unsigned)((size_t)p.N * p.Hi * p.Wi * p.Cib);
  q.korder = 1;
  if (thin_dot_ok(q)) return false;
  return launch_igemm_t<true>(q, nullptr, true) == 100;
}

// ------------------------------------------------------------------------------------------
// weight gradient:  out[ca][tap][cb] += sum_pixels A[pixel][ca] * B[n, f(ho)+dh, f(wo)+dw, cb]
// Both operands are pixel-major in memory, i.e. the reduction index is the slow one, so
// the bf16 path reads MFMA fragments with the hardware transposing LDS read
// ds_read_b64_tr_b16; the fp32 path needs one element per lane and uses ds_read_b32.
// Split over pixel ranges (blockIdx.y) with fp32 atomic accumulation into `out`.
// ------------------------------------------------------------------------------------------
// REFLECT / SIMPLE are compile-time so the per-instruction address update in the k loop is straight-line
// code (SIMPLE: a k-step of KP pixels wraps at most one image row and one image).
template <bool BF16, bool REFLECT, bool SIMPLE>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(2, 2))) void wgrad_kernel(const WgradParams p) {
  constexpr int KP = BF16 ? 64 : 32;       // pixels per k-step
  constexpr int V = BF16 ? 8 : 4;          // elements per 16-byte chunk
  constexpr int CPR = 128 / V;             // chunks per 128-element tile row (16 / 32)
  constexpr int RPT = 256 / CPR;           // tile rows covered by one pass of the block (16 / 8)
  constexpr int NLD = KP / RPT;            // chunks per thread per tile (4)
  constexpr int ROWB = 128 * (BF16 ? 2 : 4);  // LDS row bytes (256 / 512)

  __shared__ u32x4 sA[2][KP * CPR];
  __shared__ u32x4 sB[2][KP * CPR];

  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
  const int waI = wv >> 1, wbI = wv & 1;   // wave tile: 64 a-channels x 64 columns

  // XCD-aware decode: blocks b, b+8, ... share an XCD (own L2).  All tiles of one pixel split stream the
  // same dY / x rows, so each XCD gets a contiguous range of (split, tile) ids: the 18x re-read of the
  // operands across tiles is then served by that XCD's L2 instead of the fabric.
  const int nAT = (p.CaRows + 127) / 128;
  const int ntiles = p.ntiles;
  const int vid = xcd_remap(blockIdx.x, gridDim.x);
  const int split = vid / ntiles, tile = vid - split * ntiles;
  const int at = tile % nAT, bt = tile / nAT;
  const int mbeg = split * p.mchunk;
  const int mend = min(p.M, mbeg + p.mchunk);   // host guarantees mbeg < mend for every split

  // staging coordinates.  LDS-DMA writes a wave's 64 x 16 B lane-linearly (4 bf16 / 2 fp32 tile rows per
  // instruction), so the bank swizzle is applied to the SOURCE chunk each lane fetches:
  //   bf16: 32-byte slots XOR key(row), key = row bits {0,1,3}  (conflict-free ds_read_b64_tr_b16)
  //   fp32: 64-byte slots XOR (row & 7)
  const int pc = tid % CPR, rr = tid / CPR;
  const int wvu = __builtin_amdgcn_readfirstlane(tid >> 6);
  int cc;
  if constexpr (BF16) cc = pc ^ ((((rr & 3) | ((rr >> 1) & 4))) << 1);
  else cc = pc ^ ((rr & 7) << 2);
  // A: channel chunk
  const int a_ch0 = at * 128 + cc * V;
  const bool a_ok = a_ch0 < p.CaRows;
  // B: column chunk -> (tap, channel chunk)
  const int qb = bt * CPR + cc;
  const bool b_ok = qb < p.nchunks;
  int btap = 0, bcq = 0, bdh = 0, bdw = 0;
  if (b_ok) {
    btap = qb / p.cpc;
    bcq = qb - btap * p.cpc;
    bdh = p.dh[btap];
    bdw = p.dw[btap];
  }
  // pixel coordinates of the NLD rows this thread stages, advanced by KP every step.  Everything is kept
  // incrementally (no integer multiplies by runtime strides in the loop except the final byte scaling):
  //   nb = n*Hi*Wi, hraw = ho*is + dh, wraw = wo*is + dw
  int nb[NLD], pho[NLD], pwo[NLD], hraw[NLD], wraw[NLD];
  const int HoWo = p.Ho * p.Wo, HiWi = p.Hi * p.Wi;
#pragma unroll
  for (int i = 0; i < NLD; i++) {
    const int m = mbeg + rr + RPT * i;
    const int n = m / HoWo;
    const int rem = m - n * HoWo;
    pho[i] = rem / p.Wo;
    pwo[i] = rem - pho[i] * p.Wo;
    nb[i] = n * HiWi;
    hraw[i] = pho[i] * p.is + bdh;
    wraw[i] = pwo[i] * p.is + bdw;
  }
  const int adv_h = KP / p.Wo, adv_w = KP % p.Wo;
  const int adv_h_is = adv_h * p.is, adv_w_is = adv_w * p.is, Wo_is = p.Wo * p.is, Ho_is = p.Ho * p.is;
  const __amdgpu_buffer_rsrc_t rsa = __builtin_amdgcn_make_buffer_rsrc((void*)p.a, 0, p.a_bytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t rsb = __builtin_amdgcn_make_buffer_rsrc((void*)p.b, 0, p.b_bytes, 0x00020000);
  typedef __attribute__((address_space(3))) void* lds_ptr;
  int mrow = mbeg + rr;  // pixel index of row i=0 for the next load
  unsigned ao32 = (unsigned)mrow * (unsigned)p.Cab + (unsigned)a_ch0 * (unsigned)(16 / V);
  const unsigned a_step = (unsigned)RPT * (unsigned)p.Cab;
  const unsigned bcq16 = (unsigned)bcq * 16u;

  // asynchronous global->LDS copies of the next KP pixels into buffer `buf`, one instruction per call
  // (j = 2*row + {0: A operand, 1: B operand}) so they can be interleaved with the MFMA groups
  auto issue_piece = [&](int buf, int j) {
    const int i = j >> 1;
    const bool pv = (mrow + RPT * i) < mend;
    if ((j & 1) == 0) {
      char* la = reinterpret_cast<char*>(&sA[buf][0]);
      unsigned oa = ao32 + a_step * (unsigned)i;
      oa = (pv && a_ok) ? oa : 0xfffffff0u;
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rsa, (lds_ptr)(la + (wvu + 4 * i) * 1024), 16, oa, 0, 0, 0);
    } else {
      char* lb = reinterpret_cast<char*>(&sB[buf][0]);
      bool ok = pv && b_ok;
      int hi = hraw[i], wi = wraw[i];
      if constexpr (REFLECT) {
        // branch-free reflection: |x| then (L-1) - |(L-1) - x|
        hi = hi < 0 ? -hi : hi;
        wi = wi < 0 ? -wi : wi;
        const int th = (p.Hi - 1) - hi, tw = (p.Wi - 1) - wi;
        hi = (p.Hi - 1) - (th < 0 ? -th : th);
        wi = (p.Wi - 1) - (tw < 0 ? -tw : tw);
      } else {
        ok = ok && ((unsigned)hi < (unsigned)p.Hi) && ((unsigned)wi < (unsigned)p.Wi);
      }
      unsigned ob = (unsigned)(nb[i] + __mul24(hi, p.Wi) + wi) * (unsigned)p.Cbb + bcq16;
      ob = ok ? ob : 0xfffffff0u;
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rsb, (lds_ptr)(lb + (wvu + 4 * i) * 1024), 16, ob, 0, 0, 0);
      // advance this row's pixel by KP (select arithmetic, no divergent branches)
      if constexpr (SIMPLE) {
        pwo[i] += adv_w; wraw[i] += adv_w_is;
        const bool c1 = pwo[i] >= p.Wo;
        pwo[i] -= c1 ? p.Wo : 0; wraw[i] -= c1 ? Wo_is : 0; pho[i] += c1 ? 1 : 0; hraw[i] += c1 ? p.is : 0;
        pho[i] += adv_h; hraw[i] += adv_h_is;
        const bool c2 = pho[i] >= p.Ho;
        pho[i] -= c2 ? p.Ho : 0; hraw[i] -= c2 ? Ho_is : 0; nb[i] += c2 ? HiWi : 0;
      } else {
        pwo[i] += adv_w; pho[i] += adv_h;
        if (pwo[i] >= p.Wo) { pwo[i] -= p.Wo; pho[i]++; }
        while (pho[i] >= p.Ho) { pho[i] -= p.Ho; nb[i] += HiWi; }
        hraw[i] = pho[i] * p.is + bdh;
        wraw[i] = pwo[i] * p.is + bdw;
      }
    }
  };
  auto issue_end = [&]() {
    mrow += KP;
    ao32 += (unsigned)KP * (unsigned)p.Cab;
  };
  auto issue = [&](int buf) {
#pragma unroll
    for (int j = 0; j < 2 * NLD; j++) issue_piece(buf, j);
    issue_end();
  };
  auto swz = [&](int prow, int c16) -> int {
    if constexpr (BF16) {
      const int key = (prow & 3) | ((prow >> 1) & 4);
      return prow * CPR + (c16 ^ (key << 1));
    } else {
      return prow * CPR + (c16 ^ ((prow & 7) << 2));
    }
  };

  f32x4 acc[4][4];
#pragma unroll
  for (int a = 0; a < 4; a++)
#pragma unroll
    for (int b = 0; b < 4; b++) acc[a][b] = f32x4{0.f, 0.f, 0.f, 0.f};

  const int nk = (mend - mbeg + KP - 1) / KP;
  issue(0);

  auto kstep = [&](int cur, auto more_tag) {
    constexpr bool more = decltype(more_tag)::value;
    __syncthreads();   // buffer `cur` landed (vmcnt(0) + barrier); buffer `cur^1` is free again
    if constexpr (BF16) {
      const char* bA = reinterpret_cast<const char*>(&sA[cur][0]);
      const char* bB = reinterpret_cast<const char*>(&sB[cur][0]);
      const int g = lane >> 4, qq = (lane & 15) >> 2, pp = lane & 3;
#pragma unroll
      for (int kk = 0; kk < KP / 32; kk++) {
        bf16x8 af[4], bf[4];
        // rows 8g+qq (h=0) and 8g+4+qq (h=1) of this 32-pixel block; the swizzle key does not
        // depend on h, so both reads share one address register (+1024 bytes).
        const int prow = kk * 32 + 8 * g + qq;
        const int key = (prow & 3) | ((prow >> 1) & 4);
#pragma unroll
        for (int f = 0; f < 4; f++) {
          const char* pa = bA + prow * ROWB + (((waI * 4 + f) ^ key) * 32) + pp * 8;
          const char* pb = bB + prow * ROWB + (((wbI * 4 + f) ^ key) * 32) + pp * 8;
          const s16x4 a0 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(pa));
          const s16x4 a1 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(pa + 4 * ROWB));
          const s16x4 b0 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(pb));
          const s16x4 b1 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(pb + 4 * ROWB));
          af[f] = __builtin_bit_cast(bf16x8, __builtin_shufflevector(a0, a1, 0, 1, 2, 3, 4, 5, 6, 7));
          bf[f] = __builtin_bit_cast(bf16x8, __builtin_shufflevector(b0, b1, 0, 1, 2, 3, 4, 5, 6, 7));
        }
#pragma unroll
        for (int a = 0; a < 4; a++) {
          // one LDS-DMA instruction of the next k-step per group of 4 MFMAs
          if constexpr (more) issue_piece(cur ^ 1, kk * 4 + a);
          __builtin_amdgcn_s_setprio(1);
#pragma unroll
          for (int b = 0; b < 4; b++)
            acc[a][b] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bf[b], af[a], acc[a][b], 0, 0, 0);
          __builtin_amdgcn_s_setprio(0);
        }
      }
      if constexpr (more) issue_end();
    } else {
      if constexpr (more) issue(cur ^ 1);
      const float* fA = reinterpret_cast<const float*>(&sA[cur][0]);
      const float* fB = reinterpret_cast<const float*>(&sB[cur][0]);
      const int i16 = lane & 15, g = lane >> 4;
#pragma unroll
      for (int kk = 0; kk < KP / 4; kk++) {
        const int prow = kk * 4 + g;
        float af[4], bf[4];
#pragma unroll
        for (int f = 0; f < 4; f++) {
          const int cha = waI * 64 + f * 16 + i16;
          const int chb = wbI * 64 + f * 16 + i16;
          af[f] = fA[swz(prow, cha >> 2) * 4 + (cha & 3)];
          bf[f] = fB[swz(prow, chb >> 2) * 4 + (chb & 3)];
        }
#pragma unroll
        for (int a = 0; a < 4; a++)
#pragma unroll
          for (int b = 0; b < 4; b++)
            acc[a][b] = __builtin_amdgcn_mfma_f32_16x16x4f32(bf[b], af[a], acc[a][b], 0, 0, 0);
      }
    }
  };
  for (int ks = 0; ks + 1 < nk; ks++) kstep(ks & 1, std::true_type{});
  kstep((nk - 1) & 1, std::false_type{});

  // epilogue: the MFMAs ran with the column operand as A, so a lane holds D[col = (lane>>4)*4 + j][a-channel =
  // lane&15]: 4 CONSECUTIVE columns of one output row -> one 16-byte store per fragment into this split's own
  // fp32 slab (the splits are summed by unpack_kernel -- cheaper than 1.3 TB/s fp32 atomics at 20+ splits)
  const int fr = lane & 15, fg = lane >> 4;
  const int ncols = p.nchunks * V;     // multiple of 4
  float* slab = p.out + (size_t)split * p.CaRows * ncols;
#pragma unroll
  for (int a = 0; a < 4; a++) {
    const int ca = at * 128 + waI * 64 + a * 16 + fr;
    if (ca >= p.CaRows) continue;
#pragma unroll
    for (int b = 0; b < 4; b++) {
      const int col = bt * 128 + wbI * 64 + b * 16 + fg * 4;
      if (col >= ncols) continue;
      *reinterpret_cast<f32x4*>(slab + (size_t)ca * ncols + col) = acc[a][b];
    }
  }
}

int mt_launch_wgrad(int dtype, const WgradParams& pin, int nsplit, hipStream_t s) {
  MT_CHECK(pin.ntaps <= 64, "wgrad: %d taps > 64", pin.ntaps);
  MT_CHECK((double)pin.M * pin.Cab < 4294967000.0 && (double)pin.N * pin.Hi * pin.Wi * pin.Cbb < 4294967000.0,
           "wgrad: operand exceeds 4 GiB (32-bit offsets)");
  WgradParams p = pin;
  p.a_bytes = (unsigned)((size_t)pin.M * pin.Cab);
  p.b_bytes = (unsigned)((size_t)pin.N * pin.Hi * pin.Wi * pin.Cbb);
  const int V = dtype == MT_BF16 ? 8 : 4;
  const int ncols = p.nchunks * V;
  if (p.ntiles == -2) return mt_launch_wgrad_rows(p, nsplit, s);      // ... for the accumulator-stationary row walker
  if (p.ntiles < 0) {          // the caller's split was made for the 256x256 ping-pong kernel
    p.ntiles = (p.CaRows / 256) * (ncols / 256);
    return mt_launch_wgrad_pipe(p, nsplit, s);
  }
  p.ntiles = cdiv(p.CaRows, 128) * cdiv(ncols, 128);
  dim3 grid(p.ntiles * nsplit);
  const int KP = dtype == MT_BF16 ? 64 : 32;
  const bool refl = p.pad_mode == MT_PAD_REFLECT, simple = p.Ho * p.Wo >= KP;
#define MT_WG(B, R, S) hipLaunchKernelGGL((wgrad_kernel<B, R, S>), grid, dim3(256), 0, s, p)
  if (dtype == MT_BF16) {
    if (refl) { if (simple) MT_WG(true, true, true); else MT_WG(true, true, false); }
    else { if (simple) MT_WG(true, false, true); else MT_WG(true, false, false); }
  } else {
    if (refl) { if (simple) MT_WG(false, true, true); else MT_WG(false, true, false); }
    else { if (simple) MT_WG(false, false, true); else MT_WG(false, false, false); }
  }
#undef MT_WG
  MT_LAUNCH_CHECK();
  return 0;
}
