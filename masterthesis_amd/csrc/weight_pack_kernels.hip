// Weight layout kernels: reference OIHW / IOHW fp32 weights <-> the packed [row][tap][channel] GEMM images
// (pack, batched pack of a whole network), and the weight-gradient slabs back to the reference layout
// (unpack: sums the pixel-split slabs, optionally accumulating into param.grad).
#include "mt_common.h"
#include <string.h>
#include "conv_params.h"
#include <type_traits>

// ------------------------------------------------------------------------------------------
// weight (un)packing between the reference layouts and [row][tap][col] tiles
// ------------------------------------------------------------------------------------------
// pack[r][t][c] = (r<R && c<C) ? w[r*sr + c*sc + kh[t]*kW + kw[t]] : 0, r<Rp, c<Cp
template <bool BF16>
__global__ void pack_kernel(const float* __restrict__ w, void* __restrict__ out, PackParams p) {
  const long total = (long)p.Rp * p.ntaps * p.Cp;
  for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    const int c = (int)(i % p.Cp);
    const long rt = i / p.Cp;
    const int t = (int)(rt % p.ntaps);
    const int r = (int)(rt / p.ntaps);
    float v = 0.f;
    if (r < p.R && c < p.C) v = w[(long)r * p.sr + (long)c * p.sc + p.kh[t] * p.kW + p.kw[t]];
    if constexpr (BF16) reinterpret_cast<unsigned short*>(out)[i] = f32_to_bf16_bits(v);
    else reinterpret_cast<float*>(out)[i] = v;
  }
}
// batched variant: the table lives in device memory (built once per network, addresses are stable)
__global__ void pack_multi_kernel(const PackEntry* __restrict__ tab, int n) {
  int lo = 0, hi = n - 1;                       // entry whose block range holds blockIdx.x
  while (lo < hi) {
    const int mid = (lo + hi + 1) >> 1;
    if ((int)blockIdx.x >= tab[mid].blk0) lo = mid; else hi = mid - 1;
  }
  const PackEntry& e = tab[lo];
  const PackParams& p = e.p;
  const long total = (long)p.Rp * p.ntaps * p.Cp;
  const float* __restrict__ w = e.w;
  for (long i = (long)(blockIdx.x - e.blk0) * blockDim.x + threadIdx.x; i < total; i += (long)e.nblk * blockDim.x) {
    const int c = (int)(i % p.Cp);
    const long rt = i / p.Cp;
    const int t = (int)(rt % p.ntaps);
    const int r = (int)(rt / p.ntaps);
    float v = 0.f;
    if (r < p.R && c < p.C) v = w[(long)r * p.sr + (long)c * p.sc + p.kh[t] * p.kW + p.kw[t]];
    if (e.bf16) reinterpret_cast<unsigned short*>(e.out)[i] = f32_to_bf16_bits(v);
    else reinterpret_cast<float*>(e.out)[i] = v;
  }
}
// ---- grouped, LDS-tiled pack -------------------------------------------------------------------------------------
// The element-wise kernels above read 4 bytes out of every 36 (3x3) or 64 (4x4) of the OIHW tensor per pack image: PMC
// showed 614 MB fetched for a discriminator's 63 MB image, and a multi-scale discriminator (44.7 M weights, five images
// of its stride-2 layers) took 410 us per optimizer step.  Here a 256-thread block takes a 32 x 32 tile of (d0, d1) with
// ALL K2 <= 16 taps -- 32 contiguous runs of 32 * K2 floats -- into LDS once, and writes every image of the tensor from
// there: rows of 32 consecutive columns = 64-byte (bf16) runs, for either orientation.  The fp32 source is read once
// per optimizer step instead of once per image with 9-16x sector amplification.
// LDS image of a tile: per source tap k one 32 x 32 plane in the output element type, TWICE -- [k][d0][d1] for the images
// whose rows are d0 and [k][d1][d0] for the transposed ones -- so that 8 consecutive output columns are ONE 16-byte LDS
// read followed by ONE 16-byte store for either orientation.  Plane pitch 32 * 32 + 8 elements (conflict-free tap walks
// when the tile is written: consecutive lanes hold consecutive taps of one (d0, d1)).
template <bool BF16>
__device__ __forceinline__ void pack_group_tile(const PackGroup& g, int tile, char* smraw, unsigned char* s_t) {
  typedef typename std::conditional<BF16, unsigned short, float>::type T;
  constexpr int PL = 32 * 32 + 8;                    // elements per plane of the straight copy
  // the transposed copy: 40-element rows (80 bytes: 16-byte reads stay aligned).  With 32-element rows a wave's transposed writes
  // (lanes = 4 taps x 16 d1) met in 4 of the 64 banks -- 16-way conflicts on half of the tile's LDS writes: SQ_LDS_BANK_CONFLICT
  // was 77 % of this kernel's LDS-active cycles (round 4, tools/pmc_lds_conflicts.py); with 20-dword rows they spread over 16
  constexpr int RB = 40, PLB = 32 * RB + 8;
  constexpr int EV = BF16 ? 8 : 4;                   // elements per 16 bytes
  T* const sA = reinterpret_cast<T*>(smraw);         // [k][d0l][d1l]
  const int K2 = g.K2;
  T* const sB = sA + K2 * PL;                        // [k][d1l][d0l], row pitch RB
  const int td0 = tile / g.tiles_d1, td1 = tile - td0 * g.tiles_d1;
  const int d00 = td0 * 32, d10 = td1 * 32;
  const int run = 32 * K2;                            // floats of one d0 row of the tile (contiguous in the source)
  const float inv_k2 = 1.0f / (float)K2;
  // (small-integer divisions by multiplication: (j + 0.5) / K2 is at least 0.5 / K2 away from an integer)
  const bool vec = (((long)g.D1 * K2) & 3) == 0 && d10 + 32 <= g.D1;      // 16-byte source loads
  auto put = [&](int d0l, int j, float v) {
    const int d1l = (int)(((float)j + 0.5f) * inv_k2), k = j - d1l * K2;
    T e;
    if constexpr (BF16) e = f32_to_bf16_bits(v); else e = v;
    sA[k * PL + d0l * 32 + d1l] = e;
    sB[k * PLB + d1l * RB + d0l] = e;
  };
  if (vec && K2 == 16 && BF16) {
    // 4x4 filters (the discriminators: 9 of every 10 packed bytes), bf16: a thread keeps ONE float4 column (4 taps of one d1) and walks
    // the 32 d0 rows in adjacent PAIRS, so the transposed copy is written with 4-byte stores (two d0 of one (tap, d1)): half the LDS
    // write instructions of the element-wise scatter below.  All 16 loads of the thread are in flight first.
    const int j4 = (int)threadIdx.x & 127, hh = (int)threadIdx.x >> 7;        // float4 column, half of the row pairs
    const int d1l = j4 >> 2, k0 = (j4 & 3) * 4;
    f32x4 v[16];
#pragma unroll
    for (int i = 0; i < 16; i++) {
      const int d0l = 4 * (i >> 1) + 2 * hh + (i & 1);
      const f32x4 z = {0.f, 0.f, 0.f, 0.f};
      v[i] = (d00 + d0l < g.D0) ? reinterpret_cast<const f32x4*>(g.w + ((long)(d00 + d0l) * g.D1 + d10) * K2)[j4] : z;
    }
#pragma unroll
    for (int i = 0; i < 16; i += 2) {
      const int d0l = 4 * (i >> 1) + 2 * hh;
#pragma unroll
      for (int e = 0; e < 4; e++) {
        const unsigned short lo = f32_to_bf16_bits(v[i][e]), hi = f32_to_bf16_bits(v[i + 1][e]);
        sA[(k0 + e) * PL + d0l * 32 + d1l] = lo;
        sA[(k0 + e) * PL + (d0l + 1) * 32 + d1l] = hi;
        *reinterpret_cast<unsigned*>(sB + (k0 + e) * PLB + d1l * RB + d0l) = (unsigned)lo | ((unsigned)hi << 16);
      }
    }
  } else if (vec) {
    // the whole tile (32 rows x 8 K2 float4, <= 4096) in flight at once: up to 16 independent 16-byte loads per thread, THEN the
    // LDS scatter (round 4: one row per iteration with half the block idle was a chain of 32 memory latencies per tile --
    // 230 us for a discriminator's 44.7 M weights, 1.5 TB/s)
    const int rq = run >> 2;
    const float inv_rq = 1.0f / (float)rq;
    f32x4 v[16];
#pragma unroll
    for (int i = 0; i < 16; i++) {
      const int q = (int)threadIdx.x + i * 256;
      const int d0l = (int)(((float)q + 0.5f) * inv_rq), j4 = q - d0l * rq;
      const f32x4 z = {0.f, 0.f, 0.f, 0.f};
      v[i] = (d0l < 32 && d00 + d0l < g.D0)
                 ? reinterpret_cast<const f32x4*>(g.w + ((long)(d00 + d0l) * g.D1 + d10) * K2)[j4] : z;
    }
#pragma unroll
    for (int i = 0; i < 16; i++) {
      const int q = (int)threadIdx.x + i * 256;
      const int d0l = (int)(((float)q + 0.5f) * inv_rq), j4 = q - d0l * rq;
      if (d0l < 32) {
#pragma unroll
        for (int e = 0; e < 4; e++) put(d0l, j4 * 4 + e, v[i][e]);
      }
    }
  } else {
    for (int d0l = 0; d0l < 32; d0l++) {
      const int d0 = d00 + d0l;
      const float* src = g.w + ((long)d0 * g.D1 + d10) * K2;
      for (int j = threadIdx.x; j < run; j += 256) {
        const int d1l = (int)(((float)j + 0.5f) * inv_k2);
        put(d0l, j, (d0 < g.D0 && d10 + d1l < g.D1) ? src[j] : 0.f);
      }
    }
  }
  __syncthreads();
  for (int oi = 0; oi < g.nout; oi++) {
    const PackOut& o = g.o[oi];
    const int nt = o.ntaps;
    if ((int)threadIdx.x < nt) s_t[threadIdx.x] = o.tsrc[threadIdx.x];
    __syncthreads();
    const float inv_nt = 1.0f / (float)nt;
    const int rows_d0 = o.rows_d0, Rp = o.Rp, Cp = o.Cp;
    const int r0 = rows_d0 ? d00 : d10, c0 = rows_d0 ? d10 : d00;
    const T* const sS = rows_d0 ? sA : sB;
    // one thread = 16 bytes of consecutive columns of one (row, tap): a 16-byte store (2-byte stores are ~12x slower per
    // byte, MI355X_MICROARCH.md: that, not the read amplification, bounded the element-wise kernels too)
    constexpr int NCH = 32 / EV;
    for (int idx = threadIdx.x; idx < 32 * nt * NCH; idx += 256) {
      const int ch = idx % NCH, q = idx / NCH;
      const int rl = (int)(((float)q + 0.5f) * inv_nt), t = q - rl * nt;
      const int r = r0 + rl, c = c0 + ch * EV;
      if (r < Rp && c < Cp) {            // (Cp is a multiple of 8: a chunk is inside or outside as a whole)
        const u32x4 v = *reinterpret_cast<const u32x4*>(sS + (int)s_t[t] * (rows_d0 ? PL : PLB) + rl * (rows_d0 ? 32 : RB) + ch * EV);
        const long di = ((long)r * nt + t) * Cp + c;
        *reinterpret_cast<u32x4*>(reinterpret_cast<T*>(o.out) + di) = v;
      }
    }
    __syncthreads();
  }
}
__global__ __launch_bounds__(256) void pack_group_kernel(const PackGroup* __restrict__ groups, int ng) {
  __shared__ __attribute__((aligned(16))) char sm[16 * ((32 * 32 + 8) + (32 * 40 + 8)) * 2];   // two copies (the transposed one with 40-element rows), 16 taps, bf16 elements (fp32 images take the element-wise kernel)
  __shared__ unsigned char s_t[MT_MAX_TAPS];
  const int total_tiles = groups[ng - 1].tile0 + groups[ng - 1].ntiles;
  for (int tile = blockIdx.x; tile < total_tiles; tile += gridDim.x) {
    int lo = 0, hi = ng - 1;                        // group whose tile range holds `tile`
    while (lo < hi) {
      const int mid = (lo + hi + 1) >> 1;
      if (tile >= groups[mid].tile0) lo = mid; else hi = mid - 1;
    }
    const PackGroup& g = groups[lo];
    pack_group_tile<true>(g, tile - g.tile0, sm, s_t);
  }
}
int mt_launch_pack_groups(const PackGroup* dev_groups, int ngroups, int blocks, hipStream_t s) {
  if (ngroups <= 0 || blocks <= 0) return 0;
  hipLaunchKernelGGL(pack_group_kernel, dim3(blocks), dim3(256), 0, s, dev_groups, ngroups);
  MT_LAUNCH_CHECK();
  return 0;
}
int mt_launch_pack_multi(const PackEntry* dev_table, int n, int total_blocks, hipStream_t s) {
  if (n <= 0 || total_blocks <= 0) return 0;
  hipLaunchKernelGGL(pack_multi_kernel, dim3(total_blocks), dim3(256), 0, s, dev_table, n);
  MT_LAUNCH_CHECK();
  return 0;
}
int mt_launch_pack(int dtype, const float* w, void* out, const PackParams& p, hipStream_t s) {
  const long total = (long)p.Rp * p.ntaps * p.Cp;
  if (total == 0) return 0;
  const int blocks = (int)min((long)4096, (total + 255) / 256);
  if (dtype == MT_BF16) hipLaunchKernelGGL((pack_kernel<true>), dim3(blocks), dim3(256), 0, s, w, out, p);
  else hipLaunchKernelGGL((pack_kernel<false>), dim3(blocks), dim3(256), 0, s, w, out, p);
  MT_LAUNCH_CHECK();
  return 0;
}
// dw[r*sr + c*sc + kh[t]*kW + kw[t]] = sum_split src[split][r][t][c]   (r<R, c<C; src rows have Cp columns)
__global__ void unpack_kernel(const float* __restrict__ src, float* __restrict__ dw, PackParams p, int nsplit,
                              long slab, int accumulate) {
  // one thread = 4 consecutive packed columns (16-byte reads from every split slab)
  const int c4n = p.Cp >> 2;
  const long total = (long)p.R * p.ntaps * c4n;
  for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    const int c4 = (int)(i % c4n);
    const long rt = i / c4n;
    const int t = (int)(rt % p.ntaps);
    const int r = (int)(rt / p.ntaps);
    const f32x4* q = reinterpret_cast<const f32x4*>(src + ((long)r * p.ntaps + t) * p.Cp) + c4;
    f32x4 a = {0.f, 0.f, 0.f, 0.f};
    int k = 0;
    for (; k + 4 <= nsplit; k += 4) {             // four independent 16-byte loads in flight; added in index order
      const f32x4 v0 = q[(long)k * (slab >> 2)], v1 = q[(long)(k + 1) * (slab >> 2)];
      const f32x4 v2 = q[(long)(k + 2) * (slab >> 2)], v3 = q[(long)(k + 3) * (slab >> 2)];
      a += v0; a += v1; a += v2; a += v3;
    }
    for (; k < nsplit; k++) a += q[(long)k * (slab >> 2)];
    float* d = dw + (long)r * p.sr + p.kh[t] * p.kW + p.kw[t];
#pragma unroll
    for (int e = 0; e < 4; e++) {
      const int c = c4 * 4 + e;
      if (c < p.C) {
        float* o = d + (long)c * p.sc;
        *o = accumulate ? *o + a[e] : a[e];
      }
    }
  }
}
// Many splits of a small weight tensor (stem / to-RGB layers: up to 512 slabs of ~100 KB): one WAVE per 4
// packed columns, lanes stride over the split slabs (64 loads in flight instead of a serial chain), wave sum.
__global__ __launch_bounds__(256) void unpack_wave_kernel(const float* __restrict__ src, float* __restrict__ dw,
                                                          PackParams p, int nsplit, long slab, int accumulate) {
  const int c4n = p.Cp >> 2;
  const long total = (long)p.R * p.ntaps * c4n;
  const long i = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
  const int lane = threadIdx.x & 63;
  if (i >= total) return;
  const int c4 = (int)(i % c4n);
  const long rt = i / c4n;
  const int t = (int)(rt % p.ntaps);
  const int r = (int)(rt / p.ntaps);
  const f32x4* q = reinterpret_cast<const f32x4*>(src + ((long)r * p.ntaps + t) * p.Cp) + c4;
  f32x4 a = {0.f, 0.f, 0.f, 0.f};
  for (int k = lane; k < nsplit; k += 64) a += q[(long)k * (slab >> 2)];
#pragma unroll
  for (int e = 0; e < 4; e++) a[e] = wave_sum(a[e]);
  if (lane == 0) {
    float* d = dw + (long)r * p.sr + p.kh[t] * p.kW + p.kw[t];
#pragma unroll
    for (int e = 0; e < 4; e++) {
      const int c = c4 * 4 + e;
      if (c < p.C) {
        float* o = d + (long)c * p.sc;
        *o = accumulate ? *o + a[e] : a[e];
      }
    }
  }
}
// Fast path (sc == ntaps, natural tap order -- every weight-gradient unpack): for a fixed row r the output
// [c][kh][kw] is one contiguous run, so a block sums the split slabs for 64 columns with coalesced reads,
// transposes [t][c] -> [c][t] through LDS and writes a contiguous run.
__global__ __launch_bounds__(256) void unpack_t_kernel(const float* __restrict__ src, float* __restrict__ dw,
                                                       PackParams p, int nsplit, long slab, int accumulate) {
  __shared__ float tile[64 * MT_MAX_TAPS + 64];
  const int r = blockIdx.x, c0 = blockIdx.y * 64, nt = p.ntaps, S = nt | 1;
  const int ncl = min(64, p.Cp - c0);
  const float* base = src + ((long)r * nt) * p.Cp + c0;
  // round 3: 16-byte accesses on both sides (Cp % 8 == 0, so a 64-column block is whole float4s; the output run of a row is
  // nt * C contiguous floats and 16-byte aligned when nt * C % 4 == 0): a thread sums one float4 of every slab (eight slabs in
  // flight) -- for 16 taps that is exactly one item per thread -- and the transposed run leaves as float4 read-modify-writes
  for (int it = threadIdx.x; it < nt * 16; it += 256) {
    const int t = it >> 4, c4 = it & 15, cl = c4 * 4;
    f32x4 a = {0.f, 0.f, 0.f, 0.f};
    if (cl < ncl) {
      const f32x4* q = reinterpret_cast<const f32x4*>(base + (long)t * p.Cp + cl);
      const long sstep = slab >> 2;
      int k = 0;
      for (; k + 8 <= nsplit; k += 8) {          // eight slabs in flight, added in index order
        f32x4 v[8];
#pragma unroll
        for (int i = 0; i < 8; i++) v[i] = q[(long)(k + i) * sstep];
#pragma unroll
        for (int i = 0; i < 8; i++) a += v[i];
      }
      for (; k < nsplit; k++) a += q[(long)k * sstep];
    }
#pragma unroll
    for (int e = 0; e < 4; e++) tile[(cl + e) * S + t] = a[e];   // odd row stride: the transposing writes spread over the banks
  }
  __syncthreads();
  const int nvalid = min(64, p.C - c0);
  float* out = dw + (long)r * p.sr + (long)c0 * nt;
  const int total = nvalid * nt;
  if ((((size_t)out) & 15) == 0 && (total & 3) == 0) {
    for (int i4 = threadIdx.x; i4 < (total >> 2); i4 += 256) {
      f32x4 v;
#pragma unroll
      for (int e = 0; e < 4; e++) {
        const int idx = i4 * 4 + e, cl = idx / nt;
        v[e] = tile[cl * S + (idx - cl * nt)];
      }
      f32x4* o = reinterpret_cast<f32x4*>(out) + i4;
      *o = accumulate ? *o + v : v;
    }
  } else {
    for (int idx = threadIdx.x; idx < total; idx += 256) {
      const int cl = idx / nt;
      const float v = tile[cl * S + (idx - cl * nt)];
      out[idx] = accumulate ? out[idx] + v : v;
    }
  }
}
// ---- batched slab sums (round 4): the transposing slab sum of up to 64 weight tensors in ONE launch ------------------------------
// A backward pass of the multi-scale discriminators ends ~45 small weight gradients, each followed by its own slab sum of 3-16 us
// (91 such launches per step, 1.0 ms: mostly a launch's fixed cost).  Nothing reads a weight gradient before the optimizer step, so
// the sums of a whole backward pass go out together from the end-of-pass callback: the entries travel in the kernel arguments
// (48 bytes each; natural tap order only -- every convolution weight gradient), a block finds its entry by binary search over the
// entries' first-block numbers and then is unpack_t_kernel: sums one float4 of every slab (eight in flight, index order), transposes
// [tap][column] -> [column][tap] through LDS, leaves as contiguous float4 read-modify-writes.  Same arithmetic, same order, same
// results as the single launches.
struct UnpackEntry {
  const float* src;
  float* dw;
  long slab;          // floats per slab
  long sr;            // element stride of a row in dw
  int nsplit, R, C, Cp, nt, blk0, cblocks, pad_;
};
struct UnpackMulti {
  UnpackEntry e[MT_UNPACK_MULTI_MAX];
  int n, accumulate;
};
__global__ __launch_bounds__(256) void unpack_multi_kernel(const UnpackMulti m) {
  __shared__ float tile[64 * MT_MAX_TAPS + 64];
  int lo = 0, hi = m.n - 1;
  while (lo < hi) {
    const int mid = (lo + hi + 1) >> 1;
    if ((int)blockIdx.x >= m.e[mid].blk0) lo = mid; else hi = mid - 1;
  }
  const UnpackEntry& en = m.e[lo];
  const int b = (int)blockIdx.x - en.blk0;
  const int r = b / en.cblocks, c0 = (b - r * en.cblocks) * 64, nt = en.nt, S = nt | 1;
  const int Cp = en.Cp, nsplit = en.nsplit, accumulate = m.accumulate;
  const int ncl = min(64, Cp - c0);
  const float* base = en.src + ((long)r * nt) * Cp + c0;
  const long sstep = en.slab >> 2;
  for (int it = threadIdx.x; it < nt * 16; it += 256) {
    const int t = it >> 4, c4 = it & 15, cl = c4 * 4;
    f32x4 a = {0.f, 0.f, 0.f, 0.f};
    if (cl < ncl) {
      const f32x4* q = reinterpret_cast<const f32x4*>(base + (long)t * Cp + cl);
      int k = 0;
      for (; k + 8 <= nsplit; k += 8) {
        f32x4 v[8];
#pragma unroll
        for (int i = 0; i < 8; i++) v[i] = q[(long)(k + i) * sstep];
#pragma unroll
        for (int i = 0; i < 8; i++) a += v[i];
      }
      for (; k < nsplit; k++) a += q[(long)k * sstep];
    }
#pragma unroll
    for (int e = 0; e < 4; e++) tile[(cl + e) * S + t] = a[e];
  }
  __syncthreads();
  const int nvalid = min(64, en.C - c0);
  if (nvalid <= 0) return;
  float* out = en.dw + (long)r * en.sr + (long)c0 * nt;
  const int total = nvalid * nt;
  if ((((size_t)out) & 15) == 0 && (total & 3) == 0) {
    for (int i4 = threadIdx.x; i4 < (total >> 2); i4 += 256) {
      f32x4 v;
#pragma unroll
      for (int e = 0; e < 4; e++) {
        const int idx = i4 * 4 + e, cl = idx / nt;
        v[e] = tile[cl * S + (idx - cl * nt)];
      }
      f32x4* o = reinterpret_cast<f32x4*>(out) + i4;
      *o = accumulate ? *o + v : v;
    }
  } else {
    for (int idx = threadIdx.x; idx < total; idx += 256) {
      const int cl = idx / nt;
      const float v = tile[cl * S + (idx - cl * nt)];
      out[idx] = accumulate ? out[idx] + v : v;
    }
  }
}
// may this (natural-tap-order) unpack ride in a batched launch?
bool mt_unpack_multi_ok(const PackParams& p) {
  bool natural = (p.sc == p.ntaps) && p.ntaps >= 1 && p.ntaps <= MT_MAX_TAPS;
  for (int t = 0; t < p.ntaps && natural; t++) natural = (p.kh[t] * p.kW + p.kw[t] == t);
  return natural && p.R > 0 && p.Cp % 8 == 0;
}
// n <= MT_UNPACK_MULTI_MAX entries, every one mt_unpack_multi_ok
int mt_launch_unpack_multi(int n, const float* const* src, float* const* dw, const PackParams* ps, const int* nsplit, const long* slab,
                           int accumulate, hipStream_t s) {
  if (n <= 0) return 0;
  UnpackMulti m;
  memset(&m, 0, sizeof(m));
  int blocks = 0;
  for (int i = 0; i < n; i++) {
    UnpackEntry& e = m.e[i];
    const PackParams& p = ps[i];
    e.src = src[i]; e.dw = dw[i]; e.slab = slab[i]; e.sr = p.sr; e.nsplit = nsplit[i];
    e.R = p.R; e.C = p.C; e.Cp = p.Cp; e.nt = p.ntaps; e.cblocks = cdiv(p.Cp, 64); e.blk0 = blocks;
    blocks += p.R * e.cblocks;
  }
  m.n = n; m.accumulate = accumulate;
  hipLaunchKernelGGL(unpack_multi_kernel, dim3(blocks), dim3(256), 0, s, m);
  MT_LAUNCH_CHECK();
  return 0;
}
int mt_launch_unpack(const float* src, float* dw, const PackParams& p, int nsplit, long slab, int accumulate,
                     hipStream_t s) {
  const long total = (long)p.R * p.ntaps * (p.Cp >> 2);
  if (total == 0) return 0;
  bool natural = (p.sc == p.ntaps);
  for (int t = 0; t < p.ntaps && natural; t++) natural = (p.kh[t] * p.kW + p.kw[t] == t);
  // the transposing kernel needs enough (row, 64-column) blocks to fill the chip; tiny weight tensors with
  // many splits keep the element-parallel kernel
#ifndef MT_UNPACK_T_MAX
#define MT_UNPACK_T_MAX 16       // (round 3: 7 / 14 slabs of a grouped K1 weight gradient 12.4 -> 10.4 us; was 4)
#endif
  if (natural && nsplit <= MT_UNPACK_T_MAX && (long)p.R * cdiv(p.Cp, 64) >= 512) {
    hipLaunchKernelGGL(unpack_t_kernel, dim3(p.R, cdiv(p.Cp, 64)), dim3(256), 0, s, src, dw, p, nsplit, slab, accumulate);
  } else if (nsplit >= 32 && total <= 65536) {
    hipLaunchKernelGGL(unpack_wave_kernel, dim3((unsigned)((total + 3) / 4)), dim3(256), 0, s, src, dw, p, nsplit, slab,
                       accumulate);
  } else {
    const int blocks = (int)min((long)4096, (total + 255) / 256);
    hipLaunchKernelGGL(unpack_kernel, dim3(blocks), dim3(256), 0, s, src, dw, p, nsplit, slab, accumulate);
  }
  MT_LAUNCH_CHECK();
  return 0;
}
