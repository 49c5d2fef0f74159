// Normalisation family for NHWC tensors: InstanceNorm2d(affine=False) (+ReLU/LeakyReLU),
// AdaptiveInstanceNorm ((1+gamma)*IN(x)+beta, +ReLU, +residual) and the reference's custom
// per-sample LayerNorm with per-channel affine (+ReLU).
// Reference: functions.py:17,32-34; norm.py:5-33; blocks.py:38-42,83-87,158-167.
//
// All three share the same four HBM passes:
//   forward : mt_nc_stats (partial sum, sum^2 per (n, pixel block, c))  ->  mt_norm_finalize (tiny)  ->
//             mt_scale_shift_act  y = act(scale[n,c]*x + shift[n,c]) (+res)
//   backward: mt_nc_stats_bwd (partial sum g, sum g*x)    ->  mt_norm_bwd_finalize (tiny) ->
//             mt_norm_bwd_apply   dx = c1[n,c]*g + c2[n,c] + c3[n,c]*x,  g = dy*act'(.)
// Statistics are fp32 and REPRODUCIBLE: a block reduces its pixel range in a fixed order and writes one partial
// row [Cp][2]; the finalize kernels add the rows of an image in index order.  No atomics, nothing to zero.
#include "mt_common.h"

// Geometry shared by the statistics kernels: 256 threads = npl pixel lanes x cchunks 16-byte channel chunks,
// nparts pixel blocks per image (<= 64), about eight blocks per CU for the big maps.
static inline void stats_geometry(int N, int HW, int cchunks, int* ppb, int* nparts) {
  const int npl = 256 / cchunks;
  long want = 2048 / (N > 0 ? N : 1);
  if (want < 1) want = 1;
  if (want > 64) want = 64;
  long p = (HW + want - 1) / want;
  const long pmin = (long)npl * 4;            // at least four pixels per thread
  if (p < pmin) p = pmin;
  p = (p + npl - 1) / npl * npl;
  *ppb = (int)p;
  *nparts = cdiv(HW, p);
}
extern "C" int mt_nc_stats_parts(int dtype, int N, int HW, int Cp) {
  const int V = dtype == MT_BF16 ? 8 : 4;
  const int cchunks = Cp / V;
  if (cchunks < 1 || cchunks > 256 || HW <= 0) return 1;
  int ppb, nparts;
  stats_geometry(N, HW, cchunks, &ppb, &nparts);
  return nparts;
}

// grid = (nparts, N).  Thread (pl, cq) walks pixels p0 + pl, p0 + pl + npl, ... with U independent 16-byte loads
// (2U backward) in flight, then ONE pass through LDS combines the pixel lanes in a fixed order.
template <bool BF16, bool BWD>
__global__ __launch_bounds__(256) void nc_stats_kernel(const u32x4* __restrict__ x, const u32x4* __restrict__ dy,
                                                      const float* __restrict__ scale,
                                                      const float* __restrict__ shift, float* __restrict__ part,
                                                      int HW, int cchunks, int pix_per_block, int act,
                                                      float slope) {
  constexpr int V = Elem<BF16>::V;
  constexpr int NV = 2 * V;
  constexpr int U = 4;
  constexpr int STR = 257;                     // LDS row stride (floats): the output pass walks rows, keep them on distinct banks
  __shared__ float red[NV * STR];
  const int n = blockIdx.y;
  const int tid = threadIdx.x;
  const int cq = tid % cchunks;
  const int pl = tid / cchunks;
  const int npl = 256 / cchunks;
  const int Cp = cchunks * V;
  const long base = (long)n * HW * cchunks + cq;
  const int p0 = blockIdx.x * pix_per_block;
  const int p1 = min(HW, p0 + pix_per_block);
  float s1[V], s2[V], sc[V], sh[V];
#pragma unroll
  for (int e = 0; e < V; e++) { s1[e] = 0.f; s2[e] = 0.f; sc[e] = 1.f; sh[e] = 0.f; }
  if (pl < npl) {
    if constexpr (BWD) {
#pragma unroll
      for (int e = 0; e < V; e++) {
        sc[e] = scale[(long)n * Cp + cq * V + e];
        sh[e] = shift[(long)n * Cp + cq * V + e];
      }
    }
    auto accum = [&](const u32x4& xv, const u32x4& gv) {
      float f[V];
      Elem<BF16>::unpack(xv, f);
      if constexpr (BWD) {
        float g[V];
        Elem<BF16>::unpack(gv, g);
#pragma unroll
        for (int e = 0; e < V; e++) {
          const float gg = g[e] * act_grad_z(sc[e] * f[e] + sh[e], act, slope);
          s1[e] += gg;
          s2[e] += gg * f[e];
        }
      } else {
#pragma unroll
        for (int e = 0; e < V; e++) { s1[e] += f[e]; s2[e] += f[e] * f[e]; }
      }
    };
    int px = p0 + pl;
    for (; px + (U - 1) * npl < p1; px += U * npl) {
      u32x4 xv[U], gv[U];
#pragma unroll
      for (int u = 0; u < U; u++) {
        const long i = base + (long)(px + u * npl) * cchunks;
        xv[u] = x[i];
        if constexpr (BWD) gv[u] = dy[i]; else gv[u] = xv[u];
      }
#pragma unroll
      for (int u = 0; u < U; u++) accum(xv[u], gv[u]);
    }
    for (; px < p1; px += npl) {
      const long i = base + (long)px * cchunks;
      const u32x4 xa = x[i];
      u32x4 ga = xa;
      if constexpr (BWD) ga = dy[i];
      accum(xa, ga);
    }
  }
  // combine the pixel lanes.  Power-of-two cchunks < 64: the lanes of a wave that share cq first (xor shuffles),
  // then one LDS row per wave; otherwise one LDS row per pixel lane.
  const bool pre = (cchunks & (cchunks - 1)) == 0 && cchunks < 64;
  int rows;
  if (pre) {
#pragma unroll
    for (int e = 0; e < V; e++) {
      for (int o = cchunks; o < 64; o <<= 1) { s1[e] += __shfl_xor(s1[e], o, 64); s2[e] += __shfl_xor(s2[e], o, 64); }
    }
    rows = 4;
    if ((tid & 63) < cchunks) {
      const int slot = (tid >> 6) * cchunks + cq;
#pragma unroll
      for (int e = 0; e < V; e++) { red[e * STR + slot] = s1[e]; red[(V + e) * STR + slot] = s2[e]; }
    }
  } else {
    rows = npl;
    if (pl < npl) {
#pragma unroll
      for (int e = 0; e < V; e++) { red[e * STR + tid] = s1[e]; red[(V + e) * STR + tid] = s2[e]; }
    }
  }
  __syncthreads();
  float* dst = part + ((long)n * gridDim.x + blockIdx.x) * Cp * 2;
  for (int o = tid; o < cchunks * NV; o += 256) {      // o = (cq*V + e)*2 + k: contiguous store
    const int q = o / NV, j = o % NV, e = j >> 1, k = j & 1;
    const float* row = red + (k * V + e) * STR + q;
    float a = 0.f;
    for (int r = 0; r < rows; r++) a += row[r * cchunks];
    dst[o] = a;
  }
}

template <bool BWD>
static int launch_stats(int dtype, const void* x, const void* dy, const float* scale, const float* shift,
                        float* part, int N, int HW, int Cp, int act, float slope, hipStream_t s) {
  const int V = dtype == MT_BF16 ? 8 : 4;
  const int cchunks = Cp / V;
  MT_CHECK(Cp % 8 == 0 && cchunks >= 1 && cchunks <= 256, "nc_stats: unsupported channel count %d", Cp);
  if ((long)N * HW == 0) return 0;
  int ppb, nparts;
  stats_geometry(N, HW, cchunks, &ppb, &nparts);
  dim3 grid(nparts, N);
  if (dtype == MT_BF16)
    hipLaunchKernelGGL((nc_stats_kernel<true, BWD>), grid, dim3(256), 0, s, (const u32x4*)x, (const u32x4*)dy, scale, shift, part, HW, cchunks, ppb, act, slope);
  else
    hipLaunchKernelGGL((nc_stats_kernel<false, BWD>), grid, dim3(256), 0, s, (const u32x4*)x, (const u32x4*)dy, scale, shift, part, HW, cchunks, ppb, act, slope);
  MT_LAUNCH_CHECK();
  return 0;
}

extern "C" int mt_nc_stats(int dtype, const void* x, float* part, int N, int HW, int Cp, mt_stream_t s) {
  return launch_stats<false>(dtype, x, nullptr, nullptr, nullptr, part, N, HW, Cp, 0, 0.f, (hipStream_t)s);
}
extern "C" int mt_nc_stats_bwd(int dtype, const void* dy, const void* x, const float* scale, const float* shift,
                               float* part, int N, int HW, int Cp, int act, float slope, mt_stream_t s) {
  return launch_stats<true>(dtype, x, dy, scale, shift, part, N, HW, Cp, act, slope, (hipStream_t)s);
}

// {sum, sum of squares} of (n, c): the partial rows of image n added in index order
__device__ __forceinline__ void load_sums(const float* __restrict__ part, int n, int nparts, int Cp, int c, float& a,
                                          float& b) {
  const float2* p = (const float2*)part + ((long)n * nparts) * Cp + c;
  float sa = 0.f, sb = 0.f;
  for (int k = 0; k < nparts; k++) {
    const float2 v = p[(long)k * Cp];
    sa += v.x; sb += v.y;
  }
  a = sa; b = sb;
}

// Per-image totals of the partial rows in LDS: tot[c] = {sum, sum2} of (n, c).  FIN_T threads; with G = FIN_T / Cp
// thread groups (Cp <= FIN_T) group g adds rows g, g+G, ... and the groups are then added in index order -- a fixed
// order, independent of scheduling.  Cp <= MT_FIN_MAXC.
#define FIN_T 1024
#define MT_FIN_MAXC 2048
__device__ __forceinline__ void combine_parts(const float* __restrict__ part, int n, int nparts, int Cp,
                                              float2* __restrict__ tot, float2* __restrict__ scratch) {
  const int t = threadIdx.x;
  const float2* p = (const float2*)part + ((long)n * nparts) * Cp;
  if (nparts == 1) {
    for (int c = t; c < Cp; c += FIN_T) tot[c] = p[c];
  } else if (Cp <= FIN_T) {
    const int G = FIN_T / Cp;
    const int c = t % Cp, g = t / Cp;
    float sa = 0.f, sb = 0.f;
    if (g < G) {
#pragma unroll 4
      for (int k = g; k < nparts; k += G) {
        const float2 v = p[(long)k * Cp + c];
        sa += v.x; sb += v.y;
      }
    }
    scratch[t] = make_float2(sa, sb);
    __syncthreads();
    if (t < Cp) {
      float ta = 0.f, tb = 0.f;
      for (int j = 0; j < G; j++) { const float2 v = scratch[t + j * Cp]; ta += v.x; tb += v.y; }
      tot[t] = make_float2(ta, tb);
    }
  } else {
    for (int c = t; c < Cp; c += FIN_T) {
      float sa = 0.f, sb = 0.f;
#pragma unroll 4
      for (int k = 0; k < nparts; k++) { const float2 v = p[(long)k * Cp + c]; sa += v.x; sb += v.y; }
      tot[c] = make_float2(sa, sb);
    }
  }
  __syncthreads();
}

// one block per sample; threads stride over channels
__global__ __launch_bounds__(FIN_T) void norm_finalize_kernel(int mode, const float* __restrict__ sums, const float* __restrict__ gb,
                                     const float* __restrict__ gamma, const float* __restrict__ beta,
                                     float* __restrict__ scale, float* __restrict__ shift,
                                     float* __restrict__ mean, float* __restrict__ rstd, int HW, int C, int Cp,
                                     float eps, int nparts) {
  const int n = blockIdx.x;
  __shared__ float red[2][FIN_T / 64];
  __shared__ float bc[2];
  __shared__ float2 tot[MT_FIN_MAXC];
  __shared__ float2 scratch[FIN_T];
  combine_parts(sums, n, nparts, Cp, tot, scratch);
  float lmean = 0.f, lrstd = 0.f;
  if (mode == MT_NORM_LAYER) {
    float a = 0.f, b = 0.f;
    for (int c = threadIdx.x; c < C; c += blockDim.x) { a += tot[c].x; b += tot[c].y; }
    a = wave_sum(a); b = wave_sum(b);
    if ((threadIdx.x & 63) == 0) { red[0][threadIdx.x >> 6] = a; red[1][threadIdx.x >> 6] = b; }
    __syncthreads();
    if (threadIdx.x == 0) {
      float ta = 0.f, tb = 0.f;
      for (int k = 0; k < (int)(blockDim.x >> 6); k++) { ta += red[0][k]; tb += red[1][k]; }
      const float cnt = (float)C * (float)HW;
      const float m = ta / cnt;
      float var = tb / cnt - m * m;
      var = var > 0.f ? var : 0.f;
      bc[0] = m; bc[1] = rsqrtf(var + eps);
    }
    __syncthreads();
    lmean = bc[0]; lrstd = bc[1];
  }
  for (int c = threadIdx.x; c < Cp; c += blockDim.x) {
    const long i = (long)n * Cp + c;
    float m = 0.f, r = 0.f, sc = 0.f, sh = 0.f;
    if (c < C) {
      if (mode == MT_NORM_LAYER) {
        m = lmean; r = lrstd;
        const float g = gamma ? gamma[c] : 1.f, b = beta ? beta[c] : 0.f;
        sc = r * g; sh = b - m * r * g;
      } else {
        const float ca = tot[c].x, cb = tot[c].y;
        m = ca / (float)HW;
        float var = cb / (float)HW - m * m;
        var = var > 0.f ? var : 0.f;
        r = rsqrtf(var + eps);
        float a = 1.f, b = 0.f;
        if (mode == MT_NORM_ADAIN) { a = 1.f + gb[(long)n * 2 * C + c]; b = gb[(long)n * 2 * C + C + c]; }
        sc = r * a; sh = b - m * r * a;
      }
    }
    scale[i] = sc; shift[i] = sh; mean[i] = m; rstd[i] = r;
  }
}
extern "C" int mt_norm_finalize(int mode, const float* sums, const float* gb, const float* gamma,
                                const float* beta, float* scale, float* shift, float* mean, float* rstd, int N,
                                int HW, int C, int Cp, float eps, int nparts, mt_stream_t s) {
  MT_CHECK(mode != MT_NORM_ADAIN || gb != nullptr, "norm_finalize: adain needs gb");
  MT_CHECK(nparts >= 1 && Cp <= MT_FIN_MAXC, "norm_finalize: nparts %d, Cp %d", nparts, Cp);
  hipLaunchKernelGGL(norm_finalize_kernel, dim3(N), dim3(FIN_T), 0, (hipStream_t)s, mode, sums, gb, gamma, beta, scale, shift, mean, rstd, HW, C, Cp, eps, nparts);
  MT_LAUNCH_CHECK();
  return 0;
}

// y = act(scale*x + shift) (+res).  grid = (pixel blocks, N); a thread keeps ONE channel chunk for its
// whole pixel loop, so the per-(n,c) coefficients sit in registers and the inner loop is load-fma-store.
template <bool BF16>
__global__ __launch_bounds__(256) void scale_shift_act_kernel(const u32x4* __restrict__ x,
                                                              const float* __restrict__ scale,
                                                              const float* __restrict__ shift,
                                                              const u32x4* __restrict__ res, u32x4* __restrict__ y,
                                                              int HW, int cchunks, int pix_per_block, int act,
                                                              float slope) {
  constexpr int V = Elem<BF16>::V;
  const int n = blockIdx.y;
  const int cq = threadIdx.x % cchunks;
  const int pl = threadIdx.x / cchunks;
  const int npl = blockDim.x / cchunks;
  const int p0 = blockIdx.x * pix_per_block;
  const int p1 = min(HW, p0 + pix_per_block);
  float sc[V], sh[V];
  const long co = ((long)n * cchunks + cq) * V;
#pragma unroll
  for (int e = 0; e < V; e++) { sc[e] = scale[co + e]; sh[e] = shift[co + e]; }
  const long base = (long)n * HW * cchunks + cq;
  auto one = [&](long i, const u32x4& xv, const u32x4& rv) {
    float f[V], r[V];
    Elem<BF16>::unpack(xv, f);
    if (res) Elem<BF16>::unpack(rv, r);
#pragma unroll
    for (int e = 0; e < V; e++) {
      float v = act_apply(sc[e] * f[e] + sh[e], act, slope);
      if (res) v += r[e];
      f[e] = v;
    }
    __builtin_nontemporal_store(Elem<BF16>::pack(f), &y[i]);      // streaming: written once, read by the next kernel
  };
  constexpr int U = 4;                     // independent 16-byte loads in flight per thread (2U with a residual)
  int px = p0 + pl;
  for (; px + (U - 1) * npl < p1; px += U * npl) {
    u32x4 xv[U], rv[U];
    long idx[U];
#pragma unroll
    for (int u = 0; u < U; u++) {
      idx[u] = base + (long)(px + u * npl) * cchunks;
      xv[u] = x[idx[u]];
      rv[u] = res ? res[idx[u]] : xv[u];
    }
#pragma unroll
    for (int u = 0; u < U; u++) one(idx[u], xv[u], rv[u]);
  }
  for (; px < p1; px += npl) {
    const long i = base + (long)px * cchunks;
    const u32x4 xv = x[i];
    one(i, xv, res ? res[i] : xv);
  }
}
static inline void ew_geometry(int HW, int cchunks, int N, int* threads, int* ppb, dim3* grid) {
  *threads = (256 / cchunks) * cchunks;       // every thread owns one channel chunk
  const int npl = *threads / cchunks;
  int p = npl * 8;                            // 8 chunks per thread
  if (p < 1) p = 1;
  *ppb = p;
  *grid = dim3(cdiv(HW, p), N);
}
extern "C" int mt_scale_shift_act(int dtype, const void* x, const float* scale, const float* shift,
                                  const void* res, void* y, int N, int HW, int Cp, int act, float slope,
                                  mt_stream_t s) {
  const int V = dtype == MT_BF16 ? 8 : 4;
  const int cchunks = Cp / V;
  MT_CHECK(cchunks >= 1 && cchunks <= 256, "scale_shift_act: unsupported channel count %d", Cp);
  if ((long)N * HW == 0) return 0;
  int threads, ppb;
  dim3 grid;
  ew_geometry(HW, cchunks, N, &threads, &ppb, &grid);
  if (dtype == MT_BF16)
    hipLaunchKernelGGL((scale_shift_act_kernel<true>), grid, dim3(threads), 0, (hipStream_t)s, (const u32x4*)x, scale, shift, (const u32x4*)res, (u32x4*)y, HW, cchunks, ppb, act, slope);
  else
    hipLaunchKernelGGL((scale_shift_act_kernel<false>), grid, dim3(threads), 0, (hipStream_t)s, (const u32x4*)x, scale, shift, (const u32x4*)res, (u32x4*)y, HW, cchunks, ppb, act, slope);
  MT_LAUNCH_CHECK();
  return 0;
}

// Finalize + apply in ONE kernel for statistics that are already complete per image (sums [N][Cp][2], e.g. from the
// convolution's epilogue): every block derives the scale / shift of ITS image once (one thread per channel, through
// LDS), then streams its pixels.  InstanceNorm / AdaIN / LayerNorm + activation (+ residual) after a convolution is
// then conv(+statistics) -> this kernel.  Block (0, n) also stores scale, shift, mean, rstd for the backward pass.
// (Round 1 derived the coefficients per THREAD -- 8 channels for 64 elements of work -- and lost 1 ms per step; per
// block it is 256 channel finalizations for 16 k elements.)
template <bool BF16>
__global__ __launch_bounds__(256) void norm_apply_fused_kernel(const u32x4* __restrict__ x, const float* __restrict__ sums,
                                                               const float* __restrict__ gb, const float* __restrict__ gamma,
                                                               const float* __restrict__ beta, const u32x4* __restrict__ res,
                                                               u32x4* __restrict__ y, float* __restrict__ coef, int N, int HW,
                                                               int C, int cchunks, int pix_per_block, int mode, int act,
                                                               float slope, float eps) {
  constexpr int V = Elem<BF16>::V;
  __shared__ float s_sc[MT_FIN_MAXC], s_sh[MT_FIN_MAXC];
  __shared__ float redw[2][4];
  __shared__ float bc[2];
  const int n = blockIdx.y;
  const int Cp = cchunks * V;
  const float2* sn = (const float2*)sums + (long)n * Cp;
  float lmean = 0.f, lrstd = 0.f;
  if (mode == MT_NORM_LAYER) {
    float a = 0.f, b = 0.f;
    for (int c = threadIdx.x; c < C; c += blockDim.x) { const float2 v = sn[c]; a += v.x; b += v.y; }
    a = wave_sum(a); b = wave_sum(b);
    if ((threadIdx.x & 63) == 0) { redw[0][threadIdx.x >> 6] = a; redw[1][threadIdx.x >> 6] = b; }
    __syncthreads();
    if (threadIdx.x == 0) {
      float ta = 0.f, tb = 0.f;
      for (int w = 0; w < (int)((blockDim.x + 63) >> 6); w++) { ta += redw[0][w]; tb += redw[1][w]; }
      const float cnt = (float)C * (float)HW;
      const float m = ta / cnt;
      float var = tb / cnt - m * m;
      var = var > 0.f ? var : 0.f;
      bc[0] = m; bc[1] = rsqrtf(var + eps);
    }
    __syncthreads();
    lmean = bc[0]; lrstd = bc[1];
  }
  const bool keeper = blockIdx.x == 0;
  const long NC = (long)N * Cp;
  for (int c = threadIdx.x; c < Cp; c += blockDim.x) {
    float m = 0.f, r = 0.f, sc = 0.f, sh = 0.f;
    if (c < C) {
      if (mode == MT_NORM_LAYER) {
        m = lmean; r = lrstd;
        const float g = gamma ? gamma[c] : 1.f, b = beta ? beta[c] : 0.f;
        sc = r * g; sh = b - m * r * g;
      } else {
        const float2 v = sn[c];
        m = v.x / (float)HW;
        float var = v.y / (float)HW - m * m;
        var = var > 0.f ? var : 0.f;
        r = rsqrtf(var + eps);
        float a = 1.f, b = 0.f;
        if (mode == MT_NORM_ADAIN) { a = 1.f + gb[(long)n * 2 * C + c]; b = gb[(long)n * 2 * C + C + c]; }
        sc = r * a; sh = b - m * r * a;
      }
    }
    s_sc[c] = sc; s_sh[c] = sh;
    if (keeper) {
      const long i = (long)n * Cp + c;
      coef[i] = sc; coef[NC + i] = sh; coef[2 * NC + i] = m; coef[3 * NC + i] = r;
    }
  }
  __syncthreads();
  const int cq = threadIdx.x % cchunks;
  const int pl = threadIdx.x / cchunks;
  const int npl = blockDim.x / cchunks;
  const int p0 = blockIdx.x * pix_per_block;
  const int p1 = min(HW, p0 + pix_per_block);
  float sc[V], sh[V];
#pragma unroll
  for (int e = 0; e < V; e++) { sc[e] = s_sc[cq * V + e]; sh[e] = s_sh[cq * V + e]; }
  const long base = (long)n * HW * cchunks + cq;
  auto one = [&](long i, const u32x4& xv, const u32x4& rv) {
    float f[V], r[V];
    Elem<BF16>::unpack(xv, f);
    if (res) Elem<BF16>::unpack(rv, r);
#pragma unroll
    for (int e = 0; e < V; e++) {
      float v = act_apply(sc[e] * f[e] + sh[e], act, slope);
      if (res) v += r[e];
      f[e] = v;
    }
    __builtin_nontemporal_store(Elem<BF16>::pack(f), &y[i]);
  };
  constexpr int U = 4;
  int px = p0 + pl;
  for (; px + (U - 1) * npl < p1; px += U * npl) {
    u32x4 xv[U], rv[U];
    long idx[U];
#pragma unroll
    for (int u = 0; u < U; u++) {
      idx[u] = base + (long)(px + u * npl) * cchunks;
      xv[u] = x[idx[u]];
      rv[u] = res ? res[idx[u]] : xv[u];
    }
#pragma unroll
    for (int u = 0; u < U; u++) one(idx[u], xv[u], rv[u]);
  }
  for (; px < p1; px += npl) {
    const long i = base + (long)px * cchunks;
    const u32x4 xv = x[i];
    one(i, xv, res ? res[i] : xv);
  }
}
extern "C" int mt_norm_apply_fused(int dtype, int mode, const void* x, const float* sums, const float* gb,
                                   const float* gamma, const float* beta, const void* res, void* y, float* coef, int N,
                                   int HW, int C, int Cp, int act, float slope, float eps, mt_stream_t s) {
  const int V = dtype == MT_BF16 ? 8 : 4;
  const int cchunks = Cp / V;
  MT_CHECK(cchunks >= 1 && cchunks <= 256 && Cp <= MT_FIN_MAXC, "norm_apply_fused: unsupported channel count %d", Cp);
  MT_CHECK(mode == MT_NORM_INSTANCE || mode == MT_NORM_ADAIN || mode == MT_NORM_LAYER, "norm_apply_fused: bad mode %d", mode);
  MT_CHECK(mode != MT_NORM_ADAIN || gb != nullptr, "norm_apply_fused: adain needs gb");
  MT_CHECK(sums != nullptr && coef != nullptr, "norm_apply_fused: null sums / coef");
  if ((long)N * HW == 0) return 0;
  int threads, ppb;
  dim3 grid;
  ew_geometry(HW, cchunks, N, &threads, &ppb, &grid);
  if (dtype == MT_BF16)
    hipLaunchKernelGGL((norm_apply_fused_kernel<true>), grid, dim3(threads), 0, (hipStream_t)s, (const u32x4*)x, sums, gb, gamma, beta, (const u32x4*)res, (u32x4*)y, coef, N, HW, C, cchunks, ppb, mode, act, slope, eps);
  else
    hipLaunchKernelGGL((norm_apply_fused_kernel<false>), grid, dim3(threads), 0, (hipStream_t)s, (const u32x4*)x, sums, gb, gamma, beta, (const u32x4*)res, (u32x4*)y, coef, N, HW, C, cchunks, ppb, mode, act, slope, eps);
  MT_LAUNCH_CHECK();
  return 0;
}

// Backward coefficients.  With xh = (x-m)*r, a = per-(n,c) multiplier of xh in the forward
// (1, 1+gamma_adain or gamma_layer), S1 = sum g, S2 = sum g*x over the pixels:
//   sum g*xh = r*(S2 - m*S1)
//   INSTANCE/ADAIN: dx = r*a*(g - S1/HW - xh * sum(g*xh)/HW)
//   LAYER         : dx = r*(a*g - A/(C*HW) - xh * B/(C*HW)),  A = sum_c a_c*S1_c, B = sum_c a_c*sum(g*xh)_c
__global__ __launch_bounds__(FIN_T) void norm_bwd_finalize_kernel(int mode, const float* __restrict__ sums2,
                                         const float* __restrict__ mean, const float* __restrict__ rstd,
                                         const float* __restrict__ gb, const float* __restrict__ gamma,
                                         float* __restrict__ c1, float* __restrict__ c2, float* __restrict__ c3,
                                         float* __restrict__ dgb, int HW, int C, int Cp, int nparts) {
  const int n = blockIdx.x;
  __shared__ float red[2][FIN_T / 64];
  __shared__ float bc[2];
  __shared__ float2 tot[MT_FIN_MAXC];
  __shared__ float2 scratch[FIN_T];
  combine_parts(sums2, n, nparts, Cp, tot, scratch);
  float LA = 0.f, LB = 0.f;
  if (mode == MT_NORM_LAYER) {
    float a = 0.f, b = 0.f;
    for (int c = threadIdx.x; c < C; c += blockDim.x) {
      const long i = (long)n * Cp + c;
      const float S1 = tot[c].x, S2 = tot[c].y;
      const float gxh = rstd[i] * (S2 - mean[i] * S1);
      const float g = gamma ? gamma[c] : 1.f;
      a += g * S1; b += g * gxh;
      if (dgb) { dgb[(long)n * 2 * C + c] = gxh; dgb[(long)n * 2 * C + C + c] = S1; }   // per-image terms of dgamma, dbeta
    }
    a = wave_sum(a); b = wave_sum(b);
    if ((threadIdx.x & 63) == 0) { red[0][threadIdx.x >> 6] = a; red[1][threadIdx.x >> 6] = b; }
    __syncthreads();
    if (threadIdx.x == 0) {
      float ta = 0.f, tb = 0.f;
      for (int k = 0; k < (int)(blockDim.x >> 6); k++) { ta += red[0][k]; tb += red[1][k]; }
      bc[0] = ta; bc[1] = tb;
    }
    __syncthreads();
    LA = bc[0]; LB = bc[1];
  }
  for (int c = threadIdx.x; c < Cp; c += blockDim.x) {
    const long i = (long)n * Cp + c;
    float k1 = 0.f, k2 = 0.f, k3 = 0.f;
    if (c < C) {
      const float m = mean[i], r = rstd[i];
      const float S1 = tot[c].x, S2 = tot[c].y;
      const float gxh = r * (S2 - m * S1);
      if (mode == MT_NORM_LAYER) {
        const float g = gamma ? gamma[c] : 1.f;
        const float cnt = (float)C * (float)HW;
        // dx = r*g*gval - r*LA/cnt - r*(x-m)*r*LB/cnt
        k1 = r * g;
        k3 = -r * r * LB / cnt;
        k2 = -r * LA / cnt - k3 * m;
      } else {
        float a = 1.f;
        if (mode == MT_NORM_ADAIN) {
          a = 1.f + gb[(long)n * 2 * C + c];
          dgb[(long)n * 2 * C + c] = gxh;      // d(weight) = sum g*xh
          dgb[(long)n * 2 * C + C + c] = S1;   // d(bias)   = sum g
        }
        const float hw = (float)HW;
        k1 = r * a;
        k3 = -r * a * r * gxh / hw;
        k2 = -r * a * S1 / hw - k3 * m;
      }
    }
    c1[i] = k1; c2[i] = k2; c3[i] = k3;
  }
}
// LayerNorm affine gradients from the per-image terms the finalize kernel left in pg [N][2][C]:
// dgamma[c] = sum_n pg[n][0][c], dbeta[c] = sum_n pg[n][1][c], images added in index order (reproducible)
__global__ void ln_param_grad_kernel(const float* __restrict__ pg, float* __restrict__ dgamma,
                                     float* __restrict__ dbeta, int N, int C) {
  const int c = blockIdx.x * blockDim.x + threadIdx.x;
  if (c >= C) return;
  float dg = 0.f, db = 0.f;
  for (int n = 0; n < N; n++) { dg += pg[(long)n * 2 * C + c]; db += pg[(long)n * 2 * C + C + c]; }
  if (dgamma) dgamma[c] = dg;
  if (dbeta) dbeta[c] = db;
}
extern "C" int mt_norm_bwd_finalize(int mode, const float* sums2, const float* mean, const float* rstd,
                                    const float* gb, const float* gamma, float* c1, float* c2, float* c3,
                                    float* dgb, float* dgamma, float* dbeta, int N, int HW, int C, int Cp,
                                    int nparts, mt_stream_t st) {
  hipStream_t s = (hipStream_t)st;
  MT_CHECK(mode != MT_NORM_ADAIN || (gb != nullptr && dgb != nullptr), "norm_bwd_finalize: adain needs gb/dgb");
  MT_CHECK(mode != MT_NORM_LAYER || !(dgamma || dbeta) || dgb != nullptr,
           "norm_bwd_finalize: layer norm with affine gradients needs the [N][2][C] scratch in dgb");
  MT_CHECK(nparts >= 1 && Cp <= MT_FIN_MAXC, "norm_bwd_finalize: nparts %d, Cp %d", nparts, Cp);
  hipLaunchKernelGGL(norm_bwd_finalize_kernel, dim3(N), dim3(FIN_T), 0, s, mode, sums2, mean, rstd, gb, gamma, c1, c2, c3, dgb, HW, C, Cp, nparts);
  MT_LAUNCH_CHECK();
  if (mode == MT_NORM_LAYER && (dgamma || dbeta)) {
    hipLaunchKernelGGL(ln_param_grad_kernel, dim3(cdiv(C, 64)), dim3(64), 0, s, dgb, dgamma, dbeta, N, C);
    MT_LAUNCH_CHECK();
  }
  return 0;
}

template <bool BF16>
__global__ __launch_bounds__(256) void norm_bwd_apply_kernel(const u32x4* __restrict__ dy, const u32x4* __restrict__ x,
                                                             const float* __restrict__ scale,
                                                             const float* __restrict__ shift,
                                                             const float* __restrict__ c1, const float* __restrict__ c2,
                                                             const float* __restrict__ c3, u32x4* __restrict__ dx, int HW,
                                                             int cchunks, int pix_per_block, int act, float slope) {
  constexpr int V = Elem<BF16>::V;
  const int n = blockIdx.y;
  const int cq = threadIdx.x % cchunks;
  const int pl = threadIdx.x / cchunks;
  const int npl = blockDim.x / cchunks;
  const int p0 = blockIdx.x * pix_per_block;
  const int p1 = min(HW, p0 + pix_per_block);
  float sc[V], sh[V], k1[V], k2[V], k3[V];
  const long co = ((long)n * cchunks + cq) * V;
#pragma unroll
  for (int e = 0; e < V; e++) {
    sc[e] = scale[co + e]; sh[e] = shift[co + e];
    k1[e] = c1[co + e]; k2[e] = c2[co + e]; k3[e] = c3[co + e];
  }
  const long base = (long)n * HW * cchunks + cq;
  auto one = [&](long i, const u32x4& xv, const u32x4& gv) {
    float f[V], g[V];
    Elem<BF16>::unpack(xv, f);
    Elem<BF16>::unpack(gv, g);
#pragma unroll
    for (int e = 0; e < V; e++) {
      const float gg = g[e] * act_grad_z(sc[e] * f[e] + sh[e], act, slope);
      f[e] = k1[e] * gg + k2[e] + k3[e] * f[e];
    }
    __builtin_nontemporal_store(Elem<BF16>::pack(f), &dx[i]);
  };
  constexpr int U = 4;
  int px = p0 + pl;
  for (; px + (U - 1) * npl < p1; px += U * npl) {
    u32x4 xv[U], gv[U];
    long idx[U];
#pragma unroll
    for (int u = 0; u < U; u++) {
      idx[u] = base + (long)(px + u * npl) * cchunks;
      xv[u] = x[idx[u]];
      gv[u] = dy[idx[u]];
    }
#pragma unroll
    for (int u = 0; u < U; u++) one(idx[u], xv[u], gv[u]);
  }
  for (; px < p1; px += npl) {
    const long i = base + (long)px * cchunks;
    one(i, x[i], dy[i]);
  }
}
extern "C" int mt_norm_bwd_apply(int dtype, const void* dy, const void* x, const float* scale,
                                 const float* shift, const float* c1, const float* c2, const float* c3, void* dx,
                                 int N, int HW, int Cp, int act, float slope, mt_stream_t s) {
  const int V = dtype == MT_BF16 ? 8 : 4;
  const int cchunks = Cp / V;
  MT_CHECK(cchunks >= 1 && cchunks <= 256, "norm_bwd_apply: unsupported channel count %d", Cp);
  if ((long)N * HW == 0) return 0;
  int threads, ppb;
  dim3 grid;
  ew_geometry(HW, cchunks, N, &threads, &ppb, &grid);
  if (dtype == MT_BF16)
    hipLaunchKernelGGL((norm_bwd_apply_kernel<true>), grid, dim3(threads), 0, (hipStream_t)s, (const u32x4*)dy, (const u32x4*)x, scale, shift, c1, c2, c3, (u32x4*)dx, HW, cchunks, ppb, act, slope);
  else
    hipLaunchKernelGGL((norm_bwd_apply_kernel<false>), grid, dim3(threads), 0, (hipStream_t)s, (const u32x4*)dy, (const u32x4*)x, scale, shift, c1, c2, c3, (u32x4*)dx, HW, cchunks, ppb, act, slope);
  MT_LAUNCH_CHECK();
  return 0;
}

// ---- One-pass backward of InstanceNorm / AdaIN (round 3): statistics + coefficients + apply in ONE launch, dy and x read ONCE --
// The three-launch backward reads dy and x twice (statistics, then apply): 168 MB per K1 site where one pass needs 100 MB.  Here a
// 512-thread workgroup owns a SLICE of an image -- 8192 consecutive 16-byte chunks of the NHWC plane (K1: 256 pixels x 256
// channels), read as whole contiguous lines -- and keeps it IN REGISTERS (16 chunk pairs of x and dy = 128 VGPRs per thread,
// 256 KiB per CU, every load in flight at once).  It reduces {sum g, sum g x} of its slice in a fixed order, publishes the partial
// row, waits for the other slices of ITS IMAGE (arrive counter; the only inter-workgroup dependency, between workgroups with
// neighbouring indices: workgroups are dispatched in index order, so the ones waited for are resident or next in line -- and
// mt_norm_bwd_onepass_ok refuses problems with more slices per image than can be resident; a wait that exceeds `spin_limit`
// polls poisons the output with NaN instead of hanging AND sets the device status word that the host checks with the step's
// loss scalars: RuntimeError, never a silent NaN), adds the image's rows in index order
// (bit-reproducible), derives c1, c2, c3 exactly as norm_bwd_finalize_kernel does (slice 0 also writes the AdaIN parameter
// gradients) and streams dx out of the registers.
// Measured (tools/bench_norm.py, K1 [16, 256, 64, 64]): 31 us against 45 us for the three launches; inside the step 39 against
// 50 us.  What the numbers taught: (1) one workgroup per (image, 16 channels) with the whole plane in registers needs no counter,
// but a wave then reads 32-byte pieces at a 512-byte stride: 44 us, the L2 -> L1 path moves four times the bytes; (2) agent-scope
// fences (__threadfence) around the exchange write back and invalidate the XCD's whole L2: 93 us -- the exchange uses
// write-through atomic stores / loads and a vmcnt wait instead; (3) without the wait for the siblings (wrong results, diagnostic
// build MT_OP_EXP_NOSPIN) the kernel takes 24 us: 7 us are the skew between the slices of an image plus the detection latency
// of a far counter, and more, smaller slices cost more (4096-chunk slices: 48 us, 2048: 79 us).
// sync = [2][N] counters, zero before the first launch; the last workgroup of an image to leave resets them (self-cleaning).
#ifndef MT_OP_THREADS
#define MT_OP_THREADS 512                   // threads per workgroup: 512 x 16 pairs (2 waves per SIMD, 256 registers: no spill)
#endif
#ifndef MT_OP_PAIRS
#define MT_OP_PAIRS 16
#endif
#ifndef MT_OP_STORE_AUX
#define MT_OP_STORE_AUX 2                   // cache policy of the dx stores (2 = nt, as mt_norm_bwd_apply's streaming stores)
#endif
#ifdef MT_OP_MINW                           // (experiments: at least this many waves per SIMD, i.e. a register cap)
#define MT_OP_BOUNDS __launch_bounds__(MT_OP_THREADS, MT_OP_MINW)
#else
#define MT_OP_BOUNDS __launch_bounds__(MT_OP_THREADS)
#endif
constexpr int MT_OP_NT = MT_OP_THREADS;
constexpr int MT_OP_SLICE = MT_OP_THREADS * MT_OP_PAIRS;   // 16-byte chunks of x (and of dy) per workgroup
// Polls (~1 us each) before a waiting workgroup gives up and poisons its output with NaN (never a hang).  A normal wait is the
// skew between the slices of an image, < 50 us.  ASSUMPTION of the wait: no OTHER kernel that waits for sibling workgroups runs on
// the device at the same time (one process per GPU, launches of this kernel ordered on one stream) -- two such kernels can hold
// each other's compute units with partial sets of slices (seen with two processes sharing one GPU: every launch ran into this
// limit); masterthesis_amd/distributed.py switches the kernel off when ranks share a device.
constexpr int MT_OP_SPIN = 1 << 19;
template <int CCH, int NT, int P>
__global__ MT_OP_BOUNDS void norm_bwd_onepass_kernel(const u32x4* __restrict__ dy, const u32x4* __restrict__ x,
                                                                const float* __restrict__ scale, const float* __restrict__ shift,
                                                                const float* __restrict__ mean, const float* __restrict__ rstd,
                                                                const float* __restrict__ gb, float* __restrict__ dgb,
                                                                const float* __restrict__ gamma,
                                                                u32x4* __restrict__ dx, float* __restrict__ part,
                                                                unsigned* __restrict__ sync, unsigned* __restrict__ status,
                                                                int spin_limit, int N, int HW, int C, int S, int mode, int act,
                                                                float slope) {
  constexpr int V = 8, NW = NT / 64;
  static_assert(NT * P == MT_OP_SLICE && NT % CCH == 0 && P % 4 == 0, "slice geometry");
  constexpr int cchunks = CCH;               // 16-byte channel chunks per pixel (power of two, 8..256)
  constexpr int L = CCH < 64 ? CCH : 64;     // lanes of a wave that hold distinct chunks after the in-wave reduction
  constexpr int Cp = CCH * V;
  __shared__ float red[NW * L * 16];         // [wave][lane < L][{s1[8], s2[8]}]
  __shared__ float tot[2 * Cp];
  __shared__ float kf[3 * Cp];
  __shared__ int ok_flag;

  const int n = blockIdx.x / S, sl = blockIdx.x - n * S;
  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
  const int q = tid & (cchunks - 1);         // NT % cchunks == 0: the same channel chunk for all pairs of a thread
  const long co = (long)n * Cp + q * V;
  // slice-relative buffer resources: one 32-bit lane offset, the pair stride (16 KiB) as the scalar offset
  const long sbase = ((long)n * S + sl) * MT_OP_SLICE;
  const __amdgpu_buffer_rsrc_t rx = __builtin_amdgcn_make_buffer_rsrc((void*)(x + sbase), 0, MT_OP_SLICE * 16, 0x00020000);
  const __amdgpu_buffer_rsrc_t rg = __builtin_amdgcn_make_buffer_rsrc((void*)(dy + sbase), 0, MT_OP_SLICE * 16, 0x00020000);
  const __amdgpu_buffer_rsrc_t rd = __builtin_amdgcn_make_buffer_rsrc((void*)(dx + sbase), 0, MT_OP_SLICE * 16, 0x00020000);
  const unsigned voff = (unsigned)tid * 16u;
  u32x4 xv[P], gv[P];
#pragma unroll
  for (int j = 0; j < P; j++) {
    xv[j] = __builtin_amdgcn_raw_buffer_load_b128(rx, voff, j * (NT * 16), 0);
    gv[j] = __builtin_amdgcn_raw_buffer_load_b128(rg, voff, j * (NT * 16), 0);
  }
  __builtin_amdgcn_sched_barrier(0);         // all 2P loads are issued before anything else
  unsigned msk[P / 4];                       // act'(z) as one bit per element (z > 0): the forward scale / shift are dead after this
#pragma unroll
  for (int i = 0; i < P / 4; i++) msk[i] = 0u;
  const float neg = act == MT_ACT_RELU ? 0.f : (act == MT_ACT_LRELU ? slope : 1.f);
  // pair by pair in load order: the arithmetic of pair j runs while the later pairs are still in flight
  {
    float sc[V], sh[V], s1[V], s2[V];
#pragma unroll
    for (int e = 0; e < V; e++) { sc[e] = scale[co + e]; sh[e] = shift[co + e]; s1[e] = 0.f; s2[e] = 0.f; }
#pragma unroll
    for (int j = 0; j < P; j++) {
      // (an IR-level fence: without it the optimiser computes the compare masks of all 16 pairs first and spills; sched_barrier
      // only binds the machine scheduler)
      asm volatile("" : "+v"(xv[j]), "+v"(gv[j]), "+v"(s1[0]), "+v"(s1[1]), "+v"(s1[2]), "+v"(s1[3]), "+v"(s1[4]), "+v"(s1[5]),
                        "+v"(s1[6]), "+v"(s1[7]), "+v"(s2[0]), "+v"(s2[1]), "+v"(s2[2]), "+v"(s2[3]), "+v"(s2[4]), "+v"(s2[5]),
                        "+v"(s2[6]), "+v"(s2[7]));
      float f[V], g[V];
      Elem<true>::unpack(xv[j], f);
      Elem<true>::unpack(gv[j], g);
      unsigned mb = 0u;
#pragma unroll
      for (int e = 0; e < V; e++) {
        const bool pos = sc[e] * f[e] + sh[e] > 0.f;
        mb |= (pos ? 1u : 0u) << e;
        const float gg = g[e] * (pos ? 1.f : neg);
        s1[e] += gg;
        s2[e] += gg * f[e];
      }
      msk[j >> 2] |= mb << ((j & 3) * 8);
      __builtin_amdgcn_sched_barrier(0);
    }
#pragma unroll
    for (int e = 0; e < V; e++) {
#pragma unroll
      for (int o = cchunks; o < 64; o <<= 1) { s1[e] += __shfl_xor(s1[e], o, 64); s2[e] += __shfl_xor(s2[e], o, 64); }
    }
    if (lane < L) {
      float* r = red + (wv * L + lane) * 16;
#pragma unroll
      for (int e = 0; e < V; e++) { r[e] = s1[e]; r[V + e] = s2[e]; }
    }
  }
  __syncthreads();
  // the slice's partial row [Cp][2] = {sum g, sum g x}: waves in index order.  Chunk cq sits in lane cq % 64 of the waves
  // w = cq / 64 + k * (cchunks / 64) (cchunks <= 64: every wave)
  float* const prow = part + ((long)n * S + sl) * Cp * 2;
  {
    constexpr int wstep = cchunks <= 64 ? 1 : cchunks >> 6;
    for (int o = tid; o < Cp * 2; o += NT) {
      const int c = o >> 1, k = o & 1, cq = c >> 3, e = c & 7;
      float a = 0.f;
      for (int w = cq >> 6; w < NW; w += wstep) a += red[(w * L + (cq & 63)) * 16 + k * V + e];
      __hip_atomic_store(&prow[o], a, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
  }
  // the row went out as write-through (agent-scope) stores: waiting for their completion orders them before the arrive
  // increment.  (NOT __threadfence(): an agent-scope release / acquire pair writes back and invalidates the XCD's whole L2 --
  // measured 93 us per round of workgroups with the fences against a 20 us budget.)
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  if (tid == 0) {
    int ok = 1;
    if (S > 1) {
      atomicAdd(&sync[n], 1u);
      int spins = 0;
#ifdef MT_OP_EXP_NOSPIN
      while (false) {
#else
      while (__hip_atomic_load(&sync[n], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < (unsigned)S) {
#endif
        __builtin_amdgcn_s_sleep(4);
        if (++spins > spin_limit) { ok = 0; break; }
      }
    }
    if (!ok) {
      // host-visible give-up (hip_ops.raise_on_device_errors reads the words with the step's loss scalars): bit 0 of word 0,
      // word 1 = 1 + the first workgroup that gave up (diagnostic text of the RuntimeError)
      atomicOr(&status[0], 1u);
      atomicCAS(&status[1], 0u, (unsigned)blockIdx.x + 1u);
    }
    ok_flag = ok;
  }
  __syncthreads();
  const bool ok = ok_flag != 0;
  // the image's totals: its S rows in index order (every slice adds them the same way: identical coefficients everywhere)
  {
    const float* irow = part + (long)n * S * Cp * 2;
    for (int o = tid; o < Cp * 2; o += NT) {
      float a = 0.f;
#ifdef MT_OP_EXP_NOROWS
      for (int s0 = 0; s0 < 1; s0 += 16) {
#else
      for (int s0 = 0; s0 < S; s0 += 16) {     // sixteen rows in flight (a dependent chain of far loads would cost ~1 us each)
#endif
        float v[16];
#pragma unroll
        for (int i = 0; i < 16; i++)
          v[i] = s0 + i < S ? __hip_atomic_load(&irow[(long)(s0 + i) * Cp * 2 + o], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : 0.f;
#pragma unroll
        for (int i = 0; i < 16; i++) a += v[i];
      }
      tot[o] = ok ? a : __builtin_nanf("");
    }
  }
  __syncthreads();
  // LayerNorm (the reference's per-sample norm over C, H, W with a per-channel affine: norm.py:5-21): the image's two scalars
  // LA = sum_c gamma_c S1_c, LB = sum_c gamma_c sum(g xh)_c, reduced in a fixed order (lanes by xor shuffles, waves in index order);
  // slice 0 leaves the per-image terms of dgamma / dbeta in dgb [N][2][C] (ln_param_grad_kernel adds the images)
  float LA = 0.f, LB = 0.f;
  if (mode == MT_NORM_LAYER) {
    float a = 0.f, b = 0.f;
    for (int c = tid; c < C; c += NT) {
      const long i = (long)n * Cp + c;
      const float S1 = tot[2 * c], S2 = tot[2 * c + 1];
      const float gxh = rstd[i] * (S2 - mean[i] * S1);
      const float gm = gamma ? gamma[c] : 1.f;
      a += gm * S1; b += gm * gxh;
      if (sl == 0 && dgb != nullptr) { dgb[(long)n * 2 * C + c] = gxh; dgb[(long)n * 2 * C + C + c] = S1; }
    }
    a = wave_sum(a); b = wave_sum(b);
    if (lane == 0) { red[wv * 2] = a; red[wv * 2 + 1] = b; }       // (red is free again: the partial rows went out above)
    __syncthreads();
#pragma unroll
    for (int w = 0; w < NW; w++) { LA += red[w * 2]; LB += red[w * 2 + 1]; }
  }
  for (int c = tid; c < Cp; c += NT) {
    float k1 = 0.f, k2 = 0.f, k3 = 0.f;
    if (c < C) {
      const long i = (long)n * Cp + c;
      const float m = mean[i], r = rstd[i];
      const float S1 = tot[2 * c], S2 = tot[2 * c + 1];
      const float gxh = r * (S2 - m * S1);
      if (mode == MT_NORM_LAYER) {
        const float cnt = (float)C * (float)HW;
        k1 = r * (gamma ? gamma[c] : 1.f);
        k3 = -r * r * LB / cnt;
        k2 = -r * LA / cnt - k3 * m;
      } else {
        float a = 1.f;
        if (mode == MT_NORM_ADAIN) {
          a = 1.f + gb[(long)n * 2 * C + c];
          if (sl == 0) {
            dgb[(long)n * 2 * C + c] = gxh;      // d(weight) = sum g*xh
            dgb[(long)n * 2 * C + C + c] = S1;   // d(bias)   = sum g
          }
        }
        const float hw = (float)HW;
        k1 = r * a;
        k3 = -r * a * r * gxh / hw;
        k2 = -r * a * S1 / hw - k3 * m;
      }
    }
    kf[c] = k1; kf[Cp + c] = k2; kf[2 * Cp + c] = k3;
  }
  __syncthreads();
  // (opaque to the optimiser: it would otherwise keep the unpacked floats of the statistics phase alive for this one, 3x the
  // payload, and the 128 compare masks in SGPRs)
#pragma unroll
  for (int j = 0; j < P; j++) asm volatile("" : "+v"(xv[j]), "+v"(gv[j]));
#pragma unroll
  for (int i = 0; i < P / 4; i++) asm volatile("" : "+v"(msk[i]));
  // dx pair by pair, each stored as soon as it is complete (the stores drain under the arithmetic of the later pairs)
  float k1[V], k2[V], k3[V];
#pragma unroll
  for (int e = 0; e < V; e++) { k1[e] = kf[q * V + e]; k2[e] = kf[Cp + q * V + e]; k3[e] = kf[2 * Cp + q * V + e]; }
#pragma unroll
  for (int j = 0; j < P; j++) {
    asm volatile("" : "+v"(xv[j]), "+v"(gv[j]) : : "memory");     // pair j starts after pair j - 1 is stored
    float f[V], g[V];
    Elem<true>::unpack(xv[j], f);
    Elem<true>::unpack(gv[j], g);
    const unsigned mb = msk[j >> 2] >> ((j & 3) * 8);
#pragma unroll
    for (int e = 0; e < V; e++) {
      const float gg = g[e] * (((mb >> e) & 1u) ? 1.f : neg);
      f[e] = k1[e] * gg + k2[e] + k3[e] * f[e];
    }
    __builtin_amdgcn_raw_buffer_store_b128(Elem<true>::pack(f), rd, voff, j * (NT * 16), MT_OP_STORE_AUX);
    __builtin_amdgcn_sched_barrier(0);
  }
  if (S > 1 && tid == 0) {                     // leave (off the critical path): the last one out clears the image's counters
    const unsigned t = atomicAdd(&sync[N + n], 1u);
    if (t == (unsigned)S - 1) {
      __hip_atomic_store(&sync[n], 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      __hip_atomic_store(&sync[N + n], 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
  }
}
// Which problems the one-pass kernel takes: bf16, InstanceNorm / AdaIN, a power-of-two number of 16-byte channel chunks (64..2048
// channels), image planes that are whole 8192-chunk slices, at least half a round of workgroups, and AT MOST AS MANY SLICES PER
// IMAGE AS WORKGROUPS OF THIS KERNEL CAN BE RESIDENT (round 4; was a flat 1024): a slice waits for the other slices of its image,
// so all of them must fit the device together -- with 512 threads x 256 registers a compute unit holds one workgroup, i.e. 256 per
// device; slices beyond that could never be dispatched while the first ones wait (a guaranteed give-up per launch).
// mt_norm_bwd_onepass_capacity: occupancy x compute units of the current device (cached per device).  The caller may pass a smaller
// `max_slices` (e.g. capacity minus the compute units an overlapping collective holds); <= 0 means the capacity itself.
// *slices = workgroups per image (the part workspace is [N][slices][Cp][2] floats, sync [2][N] zeroed counters).
extern "C" int mt_norm_bwd_onepass_capacity(void) {
  static int cap[64] = {0};
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return 0;
  if (cap[dev] > 0) return cap[dev];
  int per_cu = 0, cus = 0;
  // (the instantiations differ in LDS only: take the one with the most)
  if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, (const void*)norm_bwd_onepass_kernel<256, MT_OP_NT, MT_OP_PAIRS>,
                                                   MT_OP_NT, 0) != hipSuccess) return 0;
  if (hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess) return 0;
  cap[dev] = per_cu * cus;
  return cap[dev];
}
static int onepass_geometry(int dtype, int mode, int N, int HW, int Cp, int act, long* S_out) {
  if (dtype != MT_BF16 || !(mode == MT_NORM_INSTANCE || mode == MT_NORM_ADAIN || mode == MT_NORM_LAYER) || N <= 0 || HW <= 0 || Cp < 8 || Cp % 8) return 0;
  const int cchunks = Cp / 8;
  if ((cchunks & (cchunks - 1)) != 0 || cchunks < 8 || cchunks > 256) return 0;
  if (!(act == MT_ACT_NONE || act == MT_ACT_RELU || act == MT_ACT_LRELU)) return 0;
  const long chunks = (long)HW * cchunks;
  if (chunks % MT_OP_SLICE != 0) return 0;
  const long S = chunks / MT_OP_SLICE;
  if ((long)N * S < 128 || (long)N * S > 0x7fffffffL) return 0;
  *S_out = S;
  return 1;
}
extern "C" int mt_norm_bwd_onepass_ok(int dtype, int mode, int N, int HW, int Cp, int act, int max_slices, int* slices) {
  if (slices) *slices = 0;
  long S = 0;
  if (!onepass_geometry(dtype, mode, N, HW, Cp, act, &S)) return 0;
  const int cap = mt_norm_bwd_onepass_capacity();
  const long lim = max_slices > 0 ? (max_slices < cap ? max_slices : cap) : cap;
  if (S > lim) return 0;
  if (slices) *slices = (int)S;
  return 1;
}
extern "C" int mt_norm_bwd_onepass(int dtype, int mode, const void* dy, const void* x, const float* scale, const float* shift,
                                   const float* mean, const float* rstd, const float* gb, float* dgb, const float* gamma,
                                   float* dgamma, float* dbeta, void* dx, float* part, unsigned* sync, unsigned* status,
                                   int spin_limit, int N, int HW, int C, int Cp, int act, float slope, mt_stream_t st) {
  long S = 0;
  MT_CHECK(onepass_geometry(dtype, mode, N, HW, Cp, act, &S), "norm_bwd_onepass: unsupported problem (dtype %d mode %d HW %d Cp %d act %d)",
           dtype, mode, HW, Cp, act);
  MT_CHECK(S <= mt_norm_bwd_onepass_capacity(), "norm_bwd_onepass: %ld slices per image exceed the %d workgroups that can be resident",
           S, mt_norm_bwd_onepass_capacity());
  MT_CHECK(mode != MT_NORM_ADAIN || (gb != nullptr && dgb != nullptr), "norm_bwd_onepass: adain needs gb/dgb");
  MT_CHECK(mode != MT_NORM_LAYER || ((dgamma == nullptr && dbeta == nullptr) || dgb != nullptr),
           "norm_bwd_onepass: layer norm with affine gradients needs the [N][2][C] scratch in dgb");
  MT_CHECK(part != nullptr && sync != nullptr && status != nullptr, "norm_bwd_onepass: needs the part / sync / status workspaces");
  MT_CHECK(C <= Cp && C > Cp - 8, "norm_bwd_onepass: C %d does not pad to Cp %d", C, Cp);
  if (spin_limit <= 0) spin_limit = MT_OP_SPIN;
  hipStream_t s = (hipStream_t)st;
  const int cchunks = Cp / 8;
#define MT_ONEPASS(CC)                                                                                                       \
  case CC:                                                                                                                   \
    hipLaunchKernelGGL((norm_bwd_onepass_kernel<CC, MT_OP_NT, MT_OP_PAIRS>), dim3(N * (int)S), dim3(MT_OP_NT), 0, s, (const u32x4*)dy, (const u32x4*)x, scale, \
                       shift, mean, rstd, gb, dgb, gamma, (u32x4*)dx, part, sync, status, spin_limit, N, HW, C, (int)S, mode, act, slope); \
    break
  switch (cchunks) {
    MT_ONEPASS(8); MT_ONEPASS(16); MT_ONEPASS(32); MT_ONEPASS(64); MT_ONEPASS(128); MT_ONEPASS(256);
    default: MT_CHECK(false, "norm_bwd_onepass: %d channel chunks", cchunks);
  }
#undef MT_ONEPASS
  MT_LAUNCH_CHECK();
  if (mode == MT_NORM_LAYER && (dgamma || dbeta)) {
    hipLaunchKernelGGL(ln_param_grad_kernel, dim3(cdiv(C, 64)), dim3(64), 0, s, dgb, dgamma, dbeta, N, C);
    MT_LAUNCH_CHECK();
  }
  return 0;
}

// ---- BatchNorm2d (--enc_norm / --dec_norm / --dis_norm batch; functions.py:14-15: affine, running statistics) ---------------
// Same four passes as the other norms: the per-(image, channel) sums are pooled over the batch here, the per-(n, c)
// coefficient arrays are filled with the per-channel values so the elementwise kernels are shared.
__global__ void bn_finalize_kernel(const float* __restrict__ sums, const float* __restrict__ gamma,
                                   const float* __restrict__ beta, float* __restrict__ rmean, float* __restrict__ rvar,
                                   float momentum, float eps, int training, float* __restrict__ scale,
                                   float* __restrict__ shift, float* __restrict__ mean, float* __restrict__ rstd, int N,
                                   int HW, int C, int Cp, int nparts) {
  const int c = blockIdx.x * blockDim.x + threadIdx.x;
  if (c >= Cp) return;
  float m = 0.f, r = 0.f, sc = 0.f, sh = 0.f;
  if (c < C) {
    if (training) {
      float s1 = 0.f, s2 = 0.f;
      for (int n = 0; n < N; n++) {
        float a, b;
        load_sums(sums, n, nparts, Cp, c, a, b);
        s1 += a; s2 += b;
      }
      const float cnt = (float)N * (float)HW;
      m = s1 / cnt;
      float var = s2 / cnt - m * m;
      var = var > 0.f ? var : 0.f;
      r = rsqrtf(var + eps);
      // running statistics: momentum update with the UNBIASED variance, like torch
      rmean[c] = (1.f - momentum) * rmean[c] + momentum * m;
      rvar[c] = (1.f - momentum) * rvar[c] + momentum * var * (cnt / fmaxf(cnt - 1.f, 1.f));
    } else {
      m = rmean[c];
      r = rsqrtf(rvar[c] + eps);
    }
    const float g = gamma ? gamma[c] : 1.f, b = beta ? beta[c] : 0.f;
    sc = r * g; sh = b - m * sc;
  }
  for (int n = 0; n < N; n++) {
    const long i = (long)n * Cp + c;
    scale[i] = sc; shift[i] = sh; mean[i] = m; rstd[i] = r;
  }
}
extern "C" int mt_bn_finalize(const float* sums, const float* gamma, const float* beta, float* running_mean,
                              float* running_var, float momentum, float eps, int training, float* scale, float* shift,
                              float* mean, float* rstd, int N, int HW, int C, int Cp, int nparts, mt_stream_t s) {
  MT_CHECK(running_mean != nullptr && running_var != nullptr, "bn_finalize: running statistics buffers are required");
  MT_CHECK(!training || sums != nullptr, "bn_finalize: training mode needs the batch sums");
  hipLaunchKernelGGL(bn_finalize_kernel, dim3(cdiv(Cp, 64)), dim3(64), 0, (hipStream_t)s, sums, gamma, beta, running_mean, running_var, momentum, eps, training, scale, shift, mean, rstd, N, HW, C, Cp, nparts);
  MT_LAUNCH_CHECK();
  return 0;
}
// dx = k1*g + k2 + k3*x (training: statistics depend on x; eval: running statistics are constants, k2 = k3 = 0)
__global__ void bn_bwd_finalize_kernel(const float* __restrict__ sums2, const float* __restrict__ mean,
                                       const float* __restrict__ rstd, const float* __restrict__ gamma,
                                       float* __restrict__ c1, float* __restrict__ c2, float* __restrict__ c3,
                                       float* __restrict__ dgamma, float* __restrict__ dbeta, int training, int N, int HW,
                                       int C, int Cp, int nparts) {
  const int c = blockIdx.x * blockDim.x + threadIdx.x;
  if (c >= Cp) return;
  float k1 = 0.f, k2 = 0.f, k3 = 0.f;
  if (c < C) {
    float S1 = 0.f, S2 = 0.f;
    for (int n = 0; n < N; n++) {
      float a, b;
      load_sums(sums2, n, nparts, Cp, c, a, b);
      S1 += a; S2 += b;
    }
    const float m = mean[c], r = rstd[c];
    const float gxh = r * (S2 - m * S1);
    const float g = gamma ? gamma[c] : 1.f;
    const float cnt = (float)N * (float)HW;
    k1 = r * g;
    if (training) {
      k3 = -r * g * r * gxh / cnt;
      k2 = -r * g * S1 / cnt - k3 * m;
    }
    if (dgamma) dgamma[c] = gxh;
    if (dbeta) dbeta[c] = S1;
  }
  for (int n = 0; n < N; n++) {
    const long i = (long)n * Cp + c;
    c1[i] = k1; c2[i] = k2; c3[i] = k3;
  }
}
extern "C" int mt_bn_bwd_finalize(const float* sums2, const float* mean, const float* rstd, const float* gamma, float* c1,
                                  float* c2, float* c3, float* dgamma, float* dbeta, int training, int N, int HW, int C,
                                  int Cp, int nparts, mt_stream_t s) {
  hipLaunchKernelGGL(bn_bwd_finalize_kernel, dim3(cdiv(Cp, 64)), dim3(64), 0, (hipStream_t)s, sums2, mean, rstd, gamma, c1, c2, c3, dgamma, dbeta, training, N, HW, C, Cp, nparts);
  MT_LAUNCH_CHECK();
  return 0;
}
