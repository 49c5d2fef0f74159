// Streaming helpers around the convolution GEMMs: adjoint of ReflectionPad2d (whole map / border ring only),
// bias-gradient column sums.
#include "mt_common.h"
#include "conv_params.h"

// ------------------------------------------------------------------------------------------
// adjoint of ReflectionPad2d: fold the gradient of the padded map back onto the interior.
// src: [N][H+2P][W+2P][Cp], dst: [N][H][W][Cp]
// ------------------------------------------------------------------------------------------
template <bool BF16>
__global__ void reflect_fold_kernel(const u32x4* __restrict__ src, u32x4* __restrict__ dst, int N, int H,
                                    int W, int cchunks, int P) {
  const long total = (long)N * H * W * cchunks;
  const int Hp = H + 2 * P, Wp = W + 2 * P;
  for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    const int cq = (int)(i % cchunks);
    long t = i / cchunks;
    const int w = (int)(t % W); t /= W;
    const int h = (int)(t % H);
    const int n = (int)(t / H);
    // pre-images of h in padded coordinates
    int hs[3], ws[3], nh = 0, nw = 0;
    hs[nh++] = h + P;
    if (h >= 1 && h <= P) hs[nh++] = P - h;
    if (h <= H - 2 && h >= H - 1 - P) hs[nh++] = P + 2 * (H - 1) - h;
    ws[nw++] = w + P;
    if (w >= 1 && w <= P) ws[nw++] = P - w;
    if (w <= W - 2 && w >= W - 1 - P) ws[nw++] = P + 2 * (W - 1) - w;
    float accv[Elem<BF16>::V];
#pragma unroll
    for (int e = 0; e < Elem<BF16>::V; e++) accv[e] = 0.f;
    for (int a = 0; a < nh; a++)
      for (int b = 0; b < nw; b++) {
        const u32x4 v = src[(((long)n * Hp + hs[a]) * Wp + ws[b]) * cchunks + cq];
        float f[Elem<BF16>::V];
        Elem<BF16>::unpack(v, f);
#pragma unroll
        for (int e = 0; e < Elem<BF16>::V; e++) accv[e] += f[e];
      }
    dst[i] = Elem<BF16>::pack(accv);
  }
}
int mt_launch_reflect_fold(int dtype, const void* src, void* dst, int N, int H, int W, int Cp, int P,
                           hipStream_t s) {
  const int V = dtype == MT_BF16 ? 8 : 4;
  const int cchunks = Cp / V;
  const long total = (long)N * H * W * cchunks;
  if (total == 0) return 0;
  const int blocks = (int)min((long)65535, (total + 255) / 256);
  if (dtype == MT_BF16)
    hipLaunchKernelGGL((reflect_fold_kernel<true>), dim3(blocks), dim3(256), 0, s, (const u32x4*)src, (u32x4*)dst, N, H, W, cchunks, P);
  else
    hipLaunchKernelGGL((reflect_fold_kernel<false>), dim3(blocks), dim3(256), 0, s, (const u32x4*)src, (u32x4*)dst, N, H, W, cchunks, P);
  MT_LAUNCH_CHECK();
  return 0;
}

// Border-only variant for the stride-1 path: dst already holds the interior part (pre-image (h+P, w+P)); add
// the other pre-images, which all lie in the P-wide ring of the padded map.  Only pixels within P of a border
// (excluding the outermost row / column, which reflection never hits) have any.
template <bool BF16>
__global__ void ring_fold_kernel(const u32x4* __restrict__ src, u32x4* __restrict__ dst, int N, int H, int W,
                                 int cchunks, int P, int band) {
  const int Hp = H + 2 * P, Wp = W + 2 * P;
  // band != 0 (H, W >= 2P+2): enumerate only the pixels that have a ring pre-image -- 2P full rows, then 2P
  // columns of the remaining H-2P rows; otherwise scan every pixel
  const int nrowpix = 2 * P * W;
  const int per_img = band ? nrowpix + (H - 2 * P) * 2 * P : H * W;
  const long total = (long)N * per_img * cchunks;
  for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    const int cq = (int)(i % cchunks);
    long t = i / cchunks;
    const int j = (int)(t % per_img);
    const int n = (int)(t / per_img);
    int h, w;
    if (!band) {
      h = j / W; w = j - h * W;
    } else if (j < nrowpix) {
      const int r = j / W;
      w = j - r * W;
      h = r < P ? 1 + r : H - 1 - P + (r - P);
    } else {
      const int jj = j - nrowpix;
      const int rr = jj / (2 * P), wc = jj - rr * (2 * P);
      h = rr == 0 ? 0 : (rr == H - 2 * P - 1 ? H - 1 : P + rr);
      w = wc < P ? 1 + wc : W - 1 - P + (wc - P);
    }
    int hs[3], ws[3], nh = 0, nw = 0;
    hs[nh++] = h + P;
    if (h >= 1 && h <= P) hs[nh++] = P - h;
    if (h <= H - 2 && h >= H - 1 - P) hs[nh++] = P + 2 * (H - 1) - h;
    ws[nw++] = w + P;
    if (w >= 1 && w <= P) ws[nw++] = P - w;
    if (w <= W - 2 && w >= W - 1 - P) ws[nw++] = P + 2 * (W - 1) - w;
    if (nh * nw == 1) continue;
    const long di = (((long)n * H + h) * W + w) * cchunks + cq;
    float accv[Elem<BF16>::V];
    Elem<BF16>::unpack(dst[di], accv);
    for (int a = 0; a < nh; a++)
      for (int b = 0; b < nw; b++) {
        if (a == 0 && b == 0) continue;
        const u32x4 v = src[(((long)n * Hp + hs[a]) * Wp + ws[b]) * cchunks + cq];
        float f[Elem<BF16>::V];
        Elem<BF16>::unpack(v, f);
#pragma unroll
        for (int e = 0; e < Elem<BF16>::V; e++) accv[e] += f[e];
      }
    dst[di] = Elem<BF16>::pack(accv);
  }
}
int mt_launch_ring_fold(int dtype, const void* src, void* dst, int N, int H, int W, int Cp, int P, hipStream_t s) {
  const int V = dtype == MT_BF16 ? 8 : 4;
  const int cchunks = Cp / V;
  const int band = (H >= 2 * P + 2 && W >= 2 * P + 2) ? 1 : 0;
  const long per_img = band ? (long)2 * P * W + (long)(H - 2 * P) * 2 * P : (long)H * W;
  const long total = (long)N * per_img * cchunks;
  if (total == 0) return 0;
  const int blocks = (int)min((long)65535, (total + 255) / 256);
  if (dtype == MT_BF16)
    hipLaunchKernelGGL((ring_fold_kernel<true>), dim3(blocks), dim3(256), 0, s, (const u32x4*)src, (u32x4*)dst, N, H, W, cchunks, P, band);
  else
    hipLaunchKernelGGL((ring_fold_kernel<false>), dim3(blocks), dim3(256), 0, s, (const u32x4*)src, (u32x4*)dst, N, H, W, cchunks, P, band);
  MT_LAUNCH_CHECK();
  return 0;
}

// ------------------------------------------------------------------------------------------
// bias gradient: db[c] = sum over pixels of dy[pixel][c].  Two launches, no atomics (512 blocks adding into
// the same C addresses serialise: ~100 us): blocks write fp32 partial rows, a small kernel adds them up.
// ------------------------------------------------------------------------------------------
// ACT: the same pass also applies the activation derivative (g = dy * act'(y), from the saved OUTPUT y of a convolution
// with a fused activation), writes g and sums g -- act_bwd + bias gradient in one read of dy instead of two.
template <bool BF16, bool ACT>
__global__ __launch_bounds__(256) void colsum_kernel(const u32x4* __restrict__ dy, float* __restrict__ part, long npix,
                                                     long cchunks_, long pix_per_block, const u32x4* __restrict__ y,
                                                     u32x4* __restrict__ dz, int act, float slope) {
  constexpr int V = Elem<BF16>::V;
  __shared__ float red[256];
  const int cchunks = (int)cchunks_;
  const int cq = threadIdx.x % cchunks;
  const int pl = threadIdx.x / cchunks;
  const int npl = blockDim.x / cchunks;
  const long p0 = (long)blockIdx.x * pix_per_block;
  const long p1 = min(npix, p0 + pix_per_block);
  float accv[V];
#pragma unroll
  for (int e = 0; e < V; e++) accv[e] = 0.f;
  if (pl < npl) {
    // 4 independent 16-byte loads in flight per thread (a dependent one-at-a-time loop is latency bound)
    long px = p0 + pl;
    auto masked = [&](long idx, const u32x4& gv, const u32x4& yv, float* f) {
      Elem<BF16>::unpack(gv, f);
      if constexpr (ACT) {
        float o[V];
        Elem<BF16>::unpack(yv, o);
#pragma unroll
        for (int e = 0; e < V; e++) f[e] *= act_grad_y(o[e], act, slope);
        dz[idx] = Elem<BF16>::pack(f);
      }
    };
    for (; px + 3 * (long)npl < p1; px += 4 * (long)npl) {
      const long i0 = px * cchunks + cq, i1 = (px + npl) * cchunks + cq, i2 = (px + 2 * (long)npl) * cchunks + cq,
                 i3 = (px + 3 * (long)npl) * cchunks + cq;
      const u32x4 v0 = dy[i0], v1 = dy[i1], v2 = dy[i2], v3 = dy[i3];
      u32x4 y0 = v0, y1 = v1, y2 = v2, y3 = v3;
      if constexpr (ACT) { y0 = y[i0]; y1 = y[i1]; y2 = y[i2]; y3 = y[i3]; }
      float f0[V], f1[V], f2[V], f3[V];
      masked(i0, v0, y0, f0);
      masked(i1, v1, y1, f1);
      masked(i2, v2, y2, f2);
      masked(i3, v3, y3, f3);
#pragma unroll
      for (int e = 0; e < V; e++) accv[e] += (f0[e] + f1[e]) + (f2[e] + f3[e]);
    }
    for (; px < p1; px += npl) {
      const long i0 = px * cchunks + cq;
      const u32x4 v0 = dy[i0];
      u32x4 y0 = v0;
      if constexpr (ACT) y0 = y[i0];
      float f[V];
      masked(i0, v0, y0, f);
#pragma unroll
      for (int e = 0; e < V; e++) accv[e] += f[e];
    }
  } else {
    // threads beyond the last whole pixel lane contribute zeros (their tid % cchunks still names a chunk)
  }
#pragma unroll
  for (int e = 0; e < V; e++) {
    const float a = block_sum_by_chunk(accv[e], cchunks, red);
    if (threadIdx.x < cchunks) part[(long)blockIdx.x * cchunks * V + threadIdx.x * V + e] = a;
  }
}
// 64 channels x 16 row groups per block: the partial rows are summed 16-way in parallel (one thread walking
// 512 rows is a 90 us dependent-load chain)
__global__ __launch_bounds__(1024) void colsum_final_kernel(const float* __restrict__ part, float* __restrict__ db,
                                                            int nblk, int Cp, int C, int accumulate) {
  __shared__ float red[1024];
  const int cl = threadIdx.x & 63, g = threadIdx.x >> 6;
  const int c = blockIdx.x * 64 + cl;
  float a0 = 0.f, a1 = 0.f;
  if (c < C) {
    int b = g;
#pragma unroll 4
    for (; b + 16 < nblk; b += 32) {
      a0 += part[(long)b * Cp + c];
      a1 += part[(long)(b + 16) * Cp + c];
    }
    if (b < nblk) a0 += part[(long)b * Cp + c];
  }
  red[threadIdx.x] = a0 + a1;
  __syncthreads();
  if (g == 0 && c < C) {
    float a = 0.f;
#pragma unroll
    for (int k = 0; k < 16; k++) a += red[k * 64 + cl];
    db[c] = accumulate ? db[c] + a : a;
  }
}
size_t mt_colsum_ws_bytes(int Cp) { return (size_t)512 * Cp * sizeof(float); }
static int launch_colsum(int dtype, const void* dy, float* db, long npix, int Cp, int C, int accumulate, void* ws,
                         size_t ws_bytes, hipStream_t s, const void* y, void* dz, int act, float slope);
int mt_launch_colsum(int dtype, const void* dy, float* db, long npix, int Cp, int C, int accumulate, void* ws,
                     size_t ws_bytes, hipStream_t s) {
  return launch_colsum(dtype, dy, db, npix, Cp, C, accumulate, ws, ws_bytes, s, nullptr, nullptr, 0, 0.f);
}
// dz = dy * act'(y) and dbias[c] (+)= sum over pixels of dz[pixel][c] in one pass (act_bwd + the bias gradient of a
// convolution with a fused activation: networks.py:363-371 discriminator stacks, blocks.py:109-117 style encoder)
extern "C" size_t mt_act_bwd_bias_ws_bytes(int Cp) { return mt_colsum_ws_bytes(Cp); }
extern "C" int mt_act_bwd_bias(int dtype, const void* dy, const void* y, void* dz, size_t npix, int Cp, int C, int act,
                               float slope, float* dbias, int accumulate, void* ws, size_t ws_bytes, mt_stream_t st) {
  MT_CHECK(dy != nullptr && y != nullptr && dz != nullptr && dbias != nullptr, "act_bwd_bias: null argument");
  MT_CHECK(Cp % 8 == 0 && C <= Cp, "act_bwd_bias: bad channel counts %d / %d", C, Cp);
  return launch_colsum(dtype, dy, dbias, (long)npix, Cp, C, accumulate, ws, ws_bytes, (hipStream_t)st, y, dz, act, slope);
}
static int launch_colsum(int dtype, const void* dy, float* db, long npix, int Cp, int C, int accumulate, void* ws,
                         size_t ws_bytes, hipStream_t s, const void* y, void* dz, int act, float slope) {
  const int V = dtype == MT_BF16 ? 8 : 4;
  const int cchunks = Cp / V;
  MT_CHECK(cchunks <= 256, "colsum: too many channels %d", Cp);
  MT_CHECK(ws != nullptr && ws_bytes >= mt_colsum_ws_bytes(Cp), "colsum: workspace too small");
  const int threads = 256;
  const int npl = threads / cchunks;
  // at most 512 blocks (one row of partials each), at least 16 pixels per thread
  long ppb = (npix + 511) / 512;
  if (ppb < (long)npl * 16) ppb = (long)npl * 16;
  ppb = (ppb + npl - 1) / npl * npl;
  const int blocks = (int)((npix + ppb - 1) / ppb);
  if (blocks > 0) {
    const u32x4* yq = (const u32x4*)y;
    u32x4* zq = (u32x4*)dz;
    if (y != nullptr) {
      if (dtype == MT_BF16)
        hipLaunchKernelGGL((colsum_kernel<true, true>), dim3(blocks), dim3(threads), 0, s, (const u32x4*)dy, (float*)ws, npix,
                           (long)cchunks, ppb, yq, zq, act, slope);
      else
        hipLaunchKernelGGL((colsum_kernel<false, true>), dim3(blocks), dim3(threads), 0, s, (const u32x4*)dy, (float*)ws, npix,
                           (long)cchunks, ppb, yq, zq, act, slope);
    } else {
      if (dtype == MT_BF16)
        hipLaunchKernelGGL((colsum_kernel<true, false>), dim3(blocks), dim3(threads), 0, s, (const u32x4*)dy, (float*)ws, npix,
                           (long)cchunks, ppb, yq, zq, act, slope);
      else
        hipLaunchKernelGGL((colsum_kernel<false, false>), dim3(blocks), dim3(threads), 0, s, (const u32x4*)dy, (float*)ws, npix,
                           (long)cchunks, ppb, yq, zq, act, slope);
    }
    MT_LAUNCH_CHECK();
  }
  hipLaunchKernelGGL(colsum_final_kernel, dim3(cdiv(C, 64)), dim3(1024), 0, s, (const float*)ws, db, blocks, Cp, C,
                     accumulate);
  MT_LAUNCH_CHECK();
  return 0;
}

// ---- weight gradient of a thin 1x1 head (round 3) --------------------------------------------------------------------------
// dW[co][ci] = sum_p dy[p][co] x[p][ci] for a 1x1 / stride 1 / no padding convolution with <= 8 output channels on a few hundred
// pixels (the dis / cls heads of the multi-scale discriminators: 2048 -> 1 / 2 channels): as a tile GEMM this was 16 column tiles
// x 32 pixel splits, their slabs and a slab sum (24-29 us per call, 12 calls per step).  Here one workgroup per 16-byte chunk
// of input channels streams the pixels (thread t takes pixels t, t + 256, ...), sums in a fixed order (wave shuffles, four wave
// partials through LDS) and writes -- or accumulates into -- the OIHW gradient directly: no workspace, no slab sum.
template <bool BF16, int CO>
__global__ __launch_bounds__(256) void thin_wgrad_kernel(const u32x4* __restrict__ x, const u32x4* __restrict__ dy,
                                                         float* __restrict__ dw, int M, int cchunks, int Ci, int Co, int accumulate) {
  constexpr int V = Elem<BF16>::V;
  constexpr int DYC = 8 / V;                       // 16-byte chunks of a dy pixel (8 padded output channels)
  __shared__ float red[4][CO * V];
  const int q = blockIdx.x, t = threadIdx.x;
  float acc[CO][V];
#pragma unroll
  for (int c = 0; c < CO; c++)
#pragma unroll
    for (int e = 0; e < V; e++) acc[c][e] = 0.f;
  for (int p = t; p < M; p += 256) {
    float f[V], g[8];
    Elem<BF16>::unpack(x[(long)p * cchunks + q], f);
#pragma unroll
    for (int k = 0; k < DYC; k++) Elem<BF16>::unpack(dy[(long)p * DYC + k], g + k * V);
#pragma unroll
    for (int c = 0; c < CO; c++)
#pragma unroll
      for (int e = 0; e < V; e++) acc[c][e] += g[c] * f[e];
  }
#pragma unroll
  for (int c = 0; c < CO; c++)
#pragma unroll
    for (int e = 0; e < V; e++) {
      const float v = wave_sum(acc[c][e]);
      if ((t & 63) == 0) red[t >> 6][c * V + e] = v;
    }
  __syncthreads();
  if (t < CO * V) {
    const int c = t / V, e = t - c * V;
    const int ci = q * V + e;
    if (ci < Ci && c < Co) {                       // CO is Co rounded up to 1 / 2 / 4 / 8: rows >= Co belong to the next tensor
      const float v = (red[0][t] + red[1][t]) + (red[2][t] + red[3][t]);
      float* o = dw + (long)c * Ci + ci;
      *o = accumulate ? *o + v : v;
    }
  }
}
// does the thin kernel take this problem?  (1x1, stride 1, no padding, not transposed, <= 8 output channels, <= 65536 pixels)
bool mt_thin_wgrad_ok(const mt_conv_desc* d) {
  return !d->transposed && d->kh == 1 && d->kw == 1 && d->stride == 1 && d->pad == 0 && d->out_pad == 0 && d->Co <= 8 &&
         d->Ci >= 256 && (long)d->N * d->H * d->W <= 65536;
}
int mt_launch_thin_wgrad(const mt_conv_desc* d, const void* x, const void* dy, float* dw, int accumulate, hipStream_t s) {
  const int V = d->dtype == MT_BF16 ? 8 : 4;
  const int cchunks = mt_padc(d->Ci) / V, M = d->N * d->H * d->W;
  const int CO = d->Co <= 1 ? 1 : (d->Co <= 2 ? 2 : (d->Co <= 4 ? 4 : 8));
#define MT_TW(B, C) hipLaunchKernelGGL((thin_wgrad_kernel<B, C>), dim3(cchunks), dim3(256), 0, s, (const u32x4*)x, (const u32x4*)dy, dw, M, cchunks, d->Ci, d->Co, accumulate)
  if (d->dtype == MT_BF16) {
    if (CO == 1) MT_TW(true, 1); else if (CO == 2) MT_TW(true, 2); else if (CO == 4) MT_TW(true, 4); else MT_TW(true, 8);
  } else {
    if (CO == 1) MT_TW(false, 1); else if (CO == 2) MT_TW(false, 2); else if (CO == 4) MT_TW(false, 4); else MT_TW(false, 8);
  }
#undef MT_TW
  MT_LAUNCH_CHECK();
  return 0;
}
