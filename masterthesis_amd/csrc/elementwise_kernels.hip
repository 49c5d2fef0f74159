// HBM-bound elementwise / pooling / layout kernels on NHWC tensors (16-byte vector accesses).
// Reference call sites: nn.LeakyReLU/ReLU (functions.py:32-34), GaussianNoiseLayer
// (misc.py:18-26), nn.AvgPool2d (blocks.py:113,115; networks.py:447), nn.AdaptiveAvgPool2d(1)
// (networks.py:124,376,456), one-hot class planes concat (networks.py:138-140).
#include "mt_common.h"

#define EW_GRID(total) (int)min((long)16384, ((long)(total) + 255) / 256)

// ---- activation -------------------------------------------------------------------------
template <bool BF16>
__global__ void act_fwd_kernel(const u32x4* __restrict__ x, u32x4* __restrict__ y, long nchunks, int act,
                               float slope) {
  constexpr int V = Elem<BF16>::V;
  for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < nchunks; i += (long)gridDim.x * blockDim.x) {
    float f[V];
    Elem<BF16>::unpack(x[i], f);
#pragma unroll
    for (int e = 0; e < V; e++) f[e] = act_apply(f[e], act, slope);
    y[i] = Elem<BF16>::pack(f);
  }
}
template <bool BF16>
__global__ void act_bwd_kernel(const u32x4* __restrict__ dy, const u32x4* __restrict__ y, u32x4* __restrict__ dx,
                               long nchunks, int act, float slope) {
  constexpr int V = Elem<BF16>::V;
  for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < nchunks; i += (long)gridDim.x * blockDim.x) {
    float g[V], o[V];
    Elem<BF16>::unpack(dy[i], g);
    Elem<BF16>::unpack(y[i], o);
#pragma unroll
    for (int e = 0; e < V; e++) g[e] *= act_grad_y(o[e], act, slope);
    dx[i] = Elem<BF16>::pack(g);
  }
}
template <bool BF16>
__global__ void add_kernel(const u32x4* __restrict__ a, const u32x4* __restrict__ b, u32x4* __restrict__ y,
                           long nchunks) {
  constexpr int V = Elem<BF16>::V;
  for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < nchunks; i += (long)gridDim.x * blockDim.x) {
    float f[V], g[V];
    Elem<BF16>::unpack(a[i], f);
    Elem<BF16>::unpack(b[i], g);
#pragma unroll
    for (int e = 0; e < V; e++) f[e] += g[e];
    y[i] = Elem<BF16>::pack(f);
  }
}

static inline int check_n(int dtype, size_t n, long* nchunks) {
  const int V = dtype == MT_BF16 ? 8 : 4;
  MT_CHECK(n % V == 0, "element count %zu is not a multiple of %d", n, V);
  *nchunks = (long)(n / V);
  return 0;
}

extern "C" int mt_act_fwd(int dtype, const void* x, void* y, size_t n, int act, float slope, mt_stream_t s) {
  long nc;
  if (check_n(dtype, n, &nc)) return 1;
  if (nc == 0) return 0;
  if (dtype == MT_BF16) hipLaunchKernelGGL((act_fwd_kernel<true>), dim3(EW_GRID(nc)), dim3(256), 0, (hipStream_t)s, (const u32x4*)x, (u32x4*)y, nc, act, slope);
  else hipLaunchKernelGGL((act_fwd_kernel<false>), dim3(EW_GRID(nc)), dim3(256), 0, (hipStream_t)s, (const u32x4*)x, (u32x4*)y, nc, act, slope);
  MT_LAUNCH_CHECK();
  return 0;
}
extern "C" int mt_act_bwd(int dtype, const void* dy, const void* y, void* dx, size_t n, int act, float slope,
                          mt_stream_t s) {
  long nc;
  if (check_n(dtype, n, &nc)) return 1;
  if (nc == 0) return 0;
  if (dtype == MT_BF16) hipLaunchKernelGGL((act_bwd_kernel<true>), dim3(EW_GRID(nc)), dim3(256), 0, (hipStream_t)s, (const u32x4*)dy, (const u32x4*)y, (u32x4*)dx, nc, act, slope);
  else hipLaunchKernelGGL((act_bwd_kernel<false>), dim3(EW_GRID(nc)), dim3(256), 0, (hipStream_t)s, (const u32x4*)dy, (const u32x4*)y, (u32x4*)dx, nc, act, slope);
  MT_LAUNCH_CHECK();
  return 0;
}
extern "C" int mt_add(int dtype, const void* a, const void* b, void* y, size_t n, mt_stream_t s) {
  long nc;
  if (check_n(dtype, n, &nc)) return 1;
  if (nc == 0) return 0;
  if (dtype == MT_BF16) hipLaunchKernelGGL((add_kernel<true>), dim3(EW_GRID(nc)), dim3(256), 0, (hipStream_t)s, (const u32x4*)a, (const u32x4*)b, (u32x4*)y, nc);
  else hipLaunchKernelGGL((add_kernel<false>), dim3(EW_GRID(nc)), dim3(256), 0, (hipStream_t)s, (const u32x4*)a, (const u32x4*)b, (u32x4*)y, nc);
  MT_LAUNCH_CHECK();
  return 0;
}

// ---- Gaussian noise: Philox4x32-10 counter RNG + Box-Muller ---------------------------------
__device__ __forceinline__ void philox_round(unsigned& c0, unsigned& c1, unsigned& c2, unsigned& c3, unsigned k0,
                                             unsigned k1) {
  const unsigned long long p0 = (unsigned long long)0xD2511F53u * c0;
  const unsigned long long p1 = (unsigned long long)0xCD9E8D57u * c2;
  const unsigned n0 = (unsigned)(p1 >> 32) ^ c1 ^ k0;
  const unsigned n1 = (unsigned)p1;
  const unsigned n2 = (unsigned)(p0 >> 32) ^ c3 ^ k1;
  const unsigned n3 = (unsigned)p0;
  c0 = n0; c1 = n1; c2 = n2; c3 = n3;
}
__device__ __forceinline__ void philox4(unsigned long long ctr, unsigned long long seed, unsigned out[4]) {
  unsigned c0 = (unsigned)ctr, c1 = (unsigned)(ctr >> 32), c2 = 0, c3 = 0;
  unsigned k0 = (unsigned)seed, k1 = (unsigned)(seed >> 32);
#pragma unroll
  for (int r = 0; r < 10; r++) {
    philox_round(c0, c1, c2, c3, k0, k1);
    k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
  }
  out[0] = c0; out[1] = c1; out[2] = c2; out[3] = c3;
}
__device__ __forceinline__ void box_muller(unsigned a, unsigned b, float& z0, float& z1) {
  const float u1 = ((float)a + 1.0f) * 2.3283064365386963e-10f;  // (0,1]
  const float u2 = (float)b * 2.3283064365386963e-10f;
  const float r = sqrtf(-2.0f * __logf(u1));
  float sn, cs;
  __sincosf(6.283185307179586f * u2, &sn, &cs);
  z0 = r * cs; z1 = r * sn;
}
template <bool BF16>
__global__ void noise_add_kernel(const u32x4* __restrict__ x, u32x4* __restrict__ y, long nchunks,
                                 unsigned long long seed, unsigned long long offset,
                                 const unsigned long long* __restrict__ dev_state) {
  constexpr int V = Elem<BF16>::V;
  if (dev_state) { seed = dev_state[0]; offset = dev_state[1] << 40; }    // {seed, draw counter} kept on the device
  for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < nchunks; i += (long)gridDim.x * blockDim.x) {
    float f[V];
    Elem<BF16>::unpack(x[i], f);
#pragma unroll
    for (int h = 0; h < V / 4; h++) {
      unsigned r[4];
      philox4(offset + (unsigned long long)i * (V / 4) + h, seed, r);
      float z[4];
      box_muller(r[0], r[1], z[0], z[1]);
      box_muller(r[2], r[3], z[2], z[3]);
#pragma unroll
      for (int e = 0; e < 4; e++) f[h * 4 + e] += z[e];
    }
    y[i] = Elem<BF16>::pack(f);
  }
}
static int launch_noise(int dtype, const void* x, void* y, size_t n, uint64_t seed, uint64_t offset,
                        const uint64_t* dev_state, mt_stream_t s) {
  long nc;
  if (check_n(dtype, n, &nc)) return 1;
  if (nc == 0) return 0;
  const unsigned long long* ds = (const unsigned long long*)dev_state;
  if (dtype == MT_BF16) hipLaunchKernelGGL((noise_add_kernel<true>), dim3(EW_GRID(nc)), dim3(256), 0, (hipStream_t)s, (const u32x4*)x, (u32x4*)y, nc, seed, offset, ds);
  else hipLaunchKernelGGL((noise_add_kernel<false>), dim3(EW_GRID(nc)), dim3(256), 0, (hipStream_t)s, (const u32x4*)x, (u32x4*)y, nc, seed, offset, ds);
  MT_LAUNCH_CHECK();
  return 0;
}
extern "C" int mt_gaussian_noise_add(int dtype, const void* x, void* y, size_t n, uint64_t seed, uint64_t offset,
                                     mt_stream_t s) {
  return launch_noise(dtype, x, y, n, seed, offset, nullptr, s);
}
// Device-resident generator state {seed, draw counter} (two uint64): the launch carries no host-side random value, so
// it can be captured into a hipGraph and still draw fresh noise on every replay.  Draw k uses the Philox counter block
// [k << 40, (k + 1) << 40); mt_rng_advance bumps the counter (stream ordered, after the draw).
__global__ void rng_advance_kernel(unsigned long long* st) { st[1] += 1ull; }
extern "C" int mt_rng_advance(uint64_t* dev_state, mt_stream_t s) {
  MT_CHECK(dev_state != nullptr, "rng_advance: null state");
  hipLaunchKernelGGL(rng_advance_kernel, dim3(1), dim3(1), 0, (hipStream_t)s, (unsigned long long*)dev_state);
  MT_LAUNCH_CHECK();
  return 0;
}
extern "C" int mt_gaussian_noise_add_dev(int dtype, const void* x, void* y, size_t n, const uint64_t* dev_state,
                                         mt_stream_t s) {
  MT_CHECK(dev_state != nullptr, "gaussian_noise_add_dev: null state");
  return launch_noise(dtype, x, y, n, 0, 0, dev_state, s);
}

// ---- dropout (nn.Dropout(0.5) at the end of the decoders' residual branches, --use_dropout; blocks.py:133-134,153-165,
// 192-207): y = x * mask / keep with mask ~ Bernoulli(keep).  The mask is a tensor in the activation layout (given in
// parity mode, else drawn here: Philox4x32-10, counter = element index), the same product serves the backward pass.
template <bool BF16>
__global__ void bernoulli_mask_kernel(u32x4* __restrict__ mask, long nchunks, int cchunks, int C, float keep,
                                      unsigned long long seed, unsigned long long offset,
                                      const unsigned long long* __restrict__ dev_state) {
  constexpr int V = Elem<BF16>::V;
  if (dev_state) { seed = dev_state[0]; offset = dev_state[1] << 40; }
  for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < nchunks; i += (long)gridDim.x * blockDim.x) {
    const int c0 = (int)(i % cchunks) * V;
    float f[V];
#pragma unroll
    for (int h = 0; h < V / 4; h++) {
      unsigned r[4];
      philox4(offset + (unsigned long long)i * (V / 4) + h, seed, r);
#pragma unroll
      for (int e = 0; e < 4; e++)
        f[h * 4 + e] = (c0 + h * 4 + e < C && (float)r[e] * 2.3283064365386963e-10f < keep) ? 1.f : 0.f;   // pad channels stay 0
    }
    mask[i] = Elem<BF16>::pack(f);
  }
}
static int launch_bernoulli(int dtype, void* mask, size_t npix, int C, int Cp, float keep, uint64_t seed, uint64_t offset,
                            const uint64_t* dev_state, mt_stream_t s) {
  long nc;
  if (check_n(dtype, npix * (size_t)Cp, &nc)) return 1;
  if (nc == 0) return 0;
  const int cchunks = Cp / (dtype == MT_BF16 ? 8 : 4);
  const unsigned long long* ds = (const unsigned long long*)dev_state;
  if (dtype == MT_BF16) hipLaunchKernelGGL((bernoulli_mask_kernel<true>), dim3(EW_GRID(nc)), dim3(256), 0, (hipStream_t)s, (u32x4*)mask, nc, cchunks, C, keep, seed, offset, ds);
  else hipLaunchKernelGGL((bernoulli_mask_kernel<false>), dim3(EW_GRID(nc)), dim3(256), 0, (hipStream_t)s, (u32x4*)mask, nc, cchunks, C, keep, seed, offset, ds);
  MT_LAUNCH_CHECK();
  return 0;
}
extern "C" int mt_bernoulli_mask(int dtype, void* mask, size_t npix, int C, int Cp, float keep, uint64_t seed,
                                 uint64_t offset, mt_stream_t s) {
  return launch_bernoulli(dtype, mask, npix, C, Cp, keep, seed, offset, nullptr, s);
}
extern "C" int mt_bernoulli_mask_dev(int dtype, void* mask, size_t npix, int C, int Cp, float keep,
                                     const uint64_t* dev_state, mt_stream_t s) {
  MT_CHECK(dev_state != nullptr, "bernoulli_mask_dev: null state");
  return launch_bernoulli(dtype, mask, npix, C, Cp, keep, 0, 0, dev_state, s);
}
template <bool BF16>
__global__ void mul_scale_kernel(const u32x4* __restrict__ a, const u32x4* __restrict__ b, u32x4* __restrict__ y,
                                 long nchunks, float scale) {
  constexpr int V = Elem<BF16>::V;
  for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < nchunks; i += (long)gridDim.x * blockDim.x) {
    float f[V], g[V];
    Elem<BF16>::unpack(a[i], f);
    Elem<BF16>::unpack(b[i], g);
#pragma unroll
    for (int e = 0; e < V; e++) f[e] = f[e] * g[e] * scale;
    y[i] = Elem<BF16>::pack(f);
  }
}
extern "C" int mt_mul_scale(int dtype, const void* a, const void* b, void* y, size_t n, float scale, mt_stream_t s) {
  long nc;
  if (check_n(dtype, n, &nc)) return 1;
  if (nc == 0) return 0;
  if (dtype == MT_BF16) hipLaunchKernelGGL((mul_scale_kernel<true>), dim3(EW_GRID(nc)), dim3(256), 0, (hipStream_t)s, (const u32x4*)a, (const u32x4*)b, (u32x4*)y, nc, scale);
  else hipLaunchKernelGGL((mul_scale_kernel<false>), dim3(EW_GRID(nc)), dim3(256), 0, (hipStream_t)s, (const u32x4*)a, (const u32x4*)b, (u32x4*)y, nc, scale);
  MT_LAUNCH_CHECK();
  return 0;
}

// ---- pooling ------------------------------------------------------------------------------
// AvgPool2d(2,2): H, W of the INPUT (floor semantics for odd sizes)
// `scale` = 0.25 for the average pool and its adjoint; with scale = 1 the two branches are the adjoint pair of
// nn.Upsample(scale_factor=2, mode='nearest'): BWD = replicate 2x2 (upsample forward), !BWD = sum 2x2 (its gradient)
template <bool BF16, bool BWD>
__global__ void avgpool2_kernel(const u32x4* __restrict__ src, u32x4* __restrict__ dst, int N, int H, int W,
                                int cchunks, float scale) {
  constexpr int V = Elem<BF16>::V;
  const int Ho = H / 2, Wo = W / 2;
  if constexpr (!BWD) {
    const long total = (long)N * Ho * Wo * cchunks;
    for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
      const int cq = (int)(i % cchunks);
      long t = i / cchunks;
      const int wo = (int)(t % Wo); t /= Wo;
      const int ho = (int)(t % Ho);
      const int n = (int)(t / Ho);
      float a[V];
#pragma unroll
      for (int e = 0; e < V; e++) a[e] = 0.f;
#pragma unroll
      for (int dh = 0; dh < 2; dh++)
#pragma unroll
        for (int dw = 0; dw < 2; dw++) {
          float f[V];
          Elem<BF16>::unpack(src[(((long)n * H + 2 * ho + dh) * W + 2 * wo + dw) * cchunks + cq], f);
#pragma unroll
          for (int e = 0; e < V; e++) a[e] += f[e];
        }
#pragma unroll
      for (int e = 0; e < V; e++) a[e] *= scale;
      dst[i] = Elem<BF16>::pack(a);
    }
  } else {
    // src = dy [N][Ho][Wo], dst = dx [N][H][W]
    const long total = (long)N * H * W * cchunks;
    for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
      const int cq = (int)(i % cchunks);
      long t = i / cchunks;
      const int w = (int)(t % W); t /= W;
      const int h = (int)(t % H);
      const int n = (int)(t / H);
      float a[V];
#pragma unroll
      for (int e = 0; e < V; e++) a[e] = 0.f;
      if (h / 2 < Ho && w / 2 < Wo) {
        Elem<BF16>::unpack(src[(((long)n * Ho + h / 2) * Wo + w / 2) * cchunks + cq], a);
#pragma unroll
        for (int e = 0; e < V; e++) a[e] *= scale;
      }
      dst[i] = Elem<BF16>::pack(a);
    }
  }
}
// AvgPool2d(3, stride 2, padding 1, count_include_pad=False)
template <bool BF16, bool BWD>
__global__ void avgpool3s2_kernel(const u32x4* __restrict__ src, u32x4* __restrict__ dst, int N, int H, int W,
                                  int cchunks) {
  constexpr int V = Elem<BF16>::V;
  const int Ho = (H + 2 - 3) / 2 + 1, Wo = (W + 2 - 3) / 2 + 1;
  if constexpr (!BWD) {
    const long total = (long)N * Ho * Wo * cchunks;
    for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
      const int cq = (int)(i % cchunks);
      long t = i / cchunks;
      const int wo = (int)(t % Wo); t /= Wo;
      const int ho = (int)(t % Ho);
      const int n = (int)(t / Ho);
      float a[V];
#pragma unroll
      for (int e = 0; e < V; e++) a[e] = 0.f;
      int cnt = 0;
      for (int dh = -1; dh <= 1; dh++)
        for (int dw = -1; dw <= 1; dw++) {
          const int h = 2 * ho + dh, w = 2 * wo + dw;
          if ((unsigned)h < (unsigned)H && (unsigned)w < (unsigned)W) {
            float f[V];
            Elem<BF16>::unpack(src[(((long)n * H + h) * W + w) * cchunks + cq], f);
#pragma unroll
            for (int e = 0; e < V; e++) a[e] += f[e];
            cnt++;
          }
        }
      const float inv = 1.f / (float)cnt;
#pragma unroll
      for (int e = 0; e < V; e++) a[e] *= inv;
      dst[i] = Elem<BF16>::pack(a);
    }
  } else {
    const long total = (long)N * H * W * cchunks;
    for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
      const int cq = (int)(i % cchunks);
      long t = i / cchunks;
      const int w = (int)(t % W); t /= W;
      const int h = (int)(t % H);
      const int n = (int)(t / H);
      float a[V];
#pragma unroll
      for (int e = 0; e < V; e++) a[e] = 0.f;
      // output windows containing (h, w): ho with |2*ho - h| <= 1
      for (int ho = (h) / 2; ho <= (h + 1) / 2; ho++)
        for (int wo = (w) / 2; wo <= (w + 1) / 2; wo++) {
          if (ho >= Ho || wo >= Wo) continue;
          const int h0 = max(2 * ho - 1, 0), h1 = min(2 * ho + 1, H - 1);
          const int w0 = max(2 * wo - 1, 0), w1 = min(2 * wo + 1, W - 1);
          const float inv = 1.f / (float)((h1 - h0 + 1) * (w1 - w0 + 1));
          float f[V];
          Elem<BF16>::unpack(src[(((long)n * Ho + ho) * Wo + wo) * cchunks + cq], f);
#pragma unroll
          for (int e = 0; e < V; e++) a[e] += f[e] * inv;
        }
      dst[i] = Elem<BF16>::pack(a);
    }
  }
}
#define POOL_LAUNCH(KERN, BWD, total)                                                                      \
  do {                                                                                                     \
    const int V = dtype == MT_BF16 ? 8 : 4;                                                                \
    const int cchunks = Cp / V;                                                                            \
    if ((total) == 0) return 0;                                                                            \
    if (dtype == MT_BF16)                                                                                  \
      hipLaunchKernelGGL((KERN<true, BWD>), dim3(EW_GRID((long)(total) * cchunks)), dim3(256), 0, (hipStream_t)s, (const u32x4*)src, (u32x4*)dst, N, H, W, cchunks); \
    else                                                                                                   \
      hipLaunchKernelGGL((KERN<false, BWD>), dim3(EW_GRID((long)(total) * cchunks)), dim3(256), 0, (hipStream_t)s, (const u32x4*)src, (u32x4*)dst, N, H, W, cchunks); \
    MT_LAUNCH_CHECK();                                                                                     \
    return 0;                                                                                              \
  } while (0)

#define POOL2_LAUNCH(BWD, total, scale)                                                                    \
  do {                                                                                                     \
    const int V = dtype == MT_BF16 ? 8 : 4;                                                                \
    const int cchunks = Cp / V;                                                                            \
    if ((total) == 0) return 0;                                                                            \
    if (dtype == MT_BF16)                                                                                  \
      hipLaunchKernelGGL((avgpool2_kernel<true, BWD>), dim3(EW_GRID((long)(total) * cchunks)), dim3(256), 0, (hipStream_t)s, (const u32x4*)src, (u32x4*)dst, N, H, W, cchunks, scale); \
    else                                                                                                   \
      hipLaunchKernelGGL((avgpool2_kernel<false, BWD>), dim3(EW_GRID((long)(total) * cchunks)), dim3(256), 0, (hipStream_t)s, (const u32x4*)src, (u32x4*)dst, N, H, W, cchunks, scale); \
    MT_LAUNCH_CHECK();                                                                                     \
    return 0;                                                                                              \
  } while (0)
extern "C" int mt_avgpool2_fwd(int dtype, const void* src, void* dst, int N, int H, int W, int Cp, mt_stream_t s) {
  POOL2_LAUNCH(false, (long)N * (H / 2) * (W / 2), 0.25f);
}
extern "C" int mt_avgpool2_bwd(int dtype, const void* src, void* dst, int N, int H, int W, int Cp, mt_stream_t s) {
  POOL2_LAUNCH(true, (long)N * H * W, 0.25f);
}
// nn.Upsample(scale_factor=2, mode='nearest') (blocks.py:75): H, W are the sizes of the UPSAMPLED tensor (even)
extern "C" int mt_upsample2_fwd(int dtype, const void* src, void* dst, int N, int H, int W, int Cp, mt_stream_t s) {
  MT_CHECK(H % 2 == 0 && W % 2 == 0, "upsample2: output size %dx%d must be even", H, W);
  POOL2_LAUNCH(true, (long)N * H * W, 1.0f);
}
extern "C" int mt_upsample2_bwd(int dtype, const void* src, void* dst, int N, int H, int W, int Cp, mt_stream_t s) {
  MT_CHECK(H % 2 == 0 && W % 2 == 0, "upsample2: output size %dx%d must be even", H, W);
  POOL2_LAUNCH(false, (long)N * (H / 2) * (W / 2), 1.0f);
}
extern "C" int mt_avgpool3s2_fwd(int dtype, const void* src, void* dst, int N, int H, int W, int Cp, mt_stream_t s) {
  POOL_LAUNCH(avgpool3s2_kernel, false, (long)N * ((H - 1) / 2 + 1) * ((W - 1) / 2 + 1));
}
extern "C" int mt_avgpool3s2_bwd(int dtype, const void* src, void* dst, int N, int H, int W, int Cp, mt_stream_t s) {
  POOL_LAUNCH(avgpool3s2_kernel, true, (long)N * H * W);
}

// ---- patches of a 4x4 / stride 2 / zero-padding 1 convolution as 4x4 mini-images (round 3) --------------------------------------
// col[(n, ho, wo)][a][b][c] = x[n][2 ho - 1 + a][2 wo - 1 + b][c] (0 outside): the [N Ho Wo][4][4][Cp] result is a batch of
// 4x4 images, on which the SAME weights applied as a 4x4 / stride 4 / no padding convolution give the original outputs.  The
// weight-shared scales of a multi-scale discriminator (networks.py:330-365) become ONE batch of mini-images for their deep,
// weight-streaming-bound layers: one pass over 67 MB of weights instead of three, one weight-gradient slab instead of three.
// BWD: dx[n][h][w] = sum of the (at most four) patch cells that copied it.
template <bool BF16, bool BWD>
__global__ void patch4s2_kernel(const u32x4* __restrict__ src, u32x4* __restrict__ dst, int N, int H, int W, int cchunks) {
  constexpr int V = Elem<BF16>::V;
  const int Ho = H >> 1, Wo = W >> 1;
  if constexpr (!BWD) {
    const long total = (long)N * Ho * Wo * 16 * cchunks;
    for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
      const int cq = (int)(i % cchunks);
      long t = i / cchunks;
      const int b = (int)(t & 3), a = (int)((t >> 2) & 3);
      t >>= 4;
      const int wo = (int)(t % Wo); t /= Wo;
      const int ho = (int)(t % Ho);
      const int n = (int)(t / Ho);
      const int h = 2 * ho - 1 + a, w = 2 * wo - 1 + b;
      u32x4 v = {0u, 0u, 0u, 0u};
      if ((unsigned)h < (unsigned)H && (unsigned)w < (unsigned)W) v = src[(((long)n * H + h) * W + w) * cchunks + cq];
      dst[i] = v;
    }
  } else {
    const long total = (long)N * H * W * cchunks;
    for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
      const int cq = (int)(i % cchunks);
      long t = i / cchunks;
      const int w = (int)(t % W); t /= W;
      const int h = (int)(t % H);
      const int n = (int)(t / H);
      float acc[V];
#pragma unroll
      for (int e = 0; e < V; e++) acc[e] = 0.f;
      // h = 2 ho - 1 + a: a has the parity of h + 1
#pragma unroll
      for (int ia = 0; ia < 2; ia++) {
        const int a = ((h + 1) & 1) + 2 * ia, ho = (h + 1 - a) >> 1;
        if ((unsigned)ho >= (unsigned)Ho) continue;
#pragma unroll
        for (int ib = 0; ib < 2; ib++) {
          const int b = ((w + 1) & 1) + 2 * ib, wo = (w + 1 - b) >> 1;
          if ((unsigned)wo >= (unsigned)Wo) continue;
          float f[V];
          Elem<BF16>::unpack(src[((((long)n * Ho + ho) * Wo + wo) * 16 + a * 4 + b) * cchunks + cq], f);
#pragma unroll
          for (int e = 0; e < V; e++) acc[e] += f[e];
        }
      }
      dst[i] = Elem<BF16>::pack(acc);
    }
  }
}
template <bool BWD>
static int launch_patch4s2(int dtype, const void* src, void* dst, int N, int H, int W, int Cp, hipStream_t s) {
  const int V = dtype == MT_BF16 ? 8 : 4;
  MT_CHECK(N > 0 && H >= 2 && W >= 2 && H % 2 == 0 && W % 2 == 0 && Cp % 8 == 0, "patch4s2: %d x %d x %d x %d", N, H, W, Cp);
  const long total = BWD ? (long)N * H * W * (Cp / V) : (long)N * (H / 2) * (W / 2) * 16 * (Cp / V);
  const int blocks = (int)min((long)8192, (total + 255) / 256);
  if (dtype == MT_BF16)
    hipLaunchKernelGGL((patch4s2_kernel<true, BWD>), dim3(blocks), dim3(256), 0, s, (const u32x4*)src, (u32x4*)dst, N, H, W, Cp / V);
  else
    hipLaunchKernelGGL((patch4s2_kernel<false, BWD>), dim3(blocks), dim3(256), 0, s, (const u32x4*)src, (u32x4*)dst, N, H, W, Cp / V);
  MT_LAUNCH_CHECK();
  return 0;
}
extern "C" int mt_patch4s2_fwd(int dtype, const void* x, void* col, int N, int H, int W, int Cp, mt_stream_t s) {
  return launch_patch4s2<false>(dtype, x, col, N, H, W, Cp, (hipStream_t)s);
}
extern "C" int mt_patch4s2_bwd(int dtype, const void* dcol, void* dx, int N, int H, int W, int Cp, mt_stream_t s) {
  return launch_patch4s2<true>(dtype, dcol, dx, N, H, W, Cp, (hipStream_t)s);
}

// ---- the same for up to four inputs in ONE launch (round 4): the scales of a multi-scale discriminator layer (networks.py:445-466)
// were three launches of ~6 us each way, 72 per step; the tensors are tiny next to a launch's fixed cost
struct Patch4Multi {
  const u32x4* src[4];
  u32x4* dst[4];
  int N[4], H[4], W[4];
  long end[4];        // cumulative work items (16-byte chunks) up to and including input k
  int count;
};
template <bool BF16, bool BWD>
__global__ void patch4s2_multi_kernel(const Patch4Multi m, int cchunks) {
  constexpr int V = Elem<BF16>::V;
  const long total = m.end[m.count - 1];
  for (long gi = blockIdx.x * (long)blockDim.x + threadIdx.x; gi < total; gi += (long)gridDim.x * blockDim.x) {
    int k = 0;
#pragma unroll
    for (int j = 0; j < 3; j++) k = (j + 1 < m.count && gi >= m.end[j]) ? j + 1 : k;
    const long i = gi - (k ? m.end[k - 1] : 0);
    const int H = m.H[k], W = m.W[k], Ho = H >> 1, Wo = W >> 1;
    const u32x4* __restrict__ src = m.src[k];
    u32x4* __restrict__ dst = m.dst[k];
    const int cq = (int)(i % cchunks);
    long t = i / cchunks;
    if constexpr (!BWD) {
      const int b = (int)(t & 3), a = (int)((t >> 2) & 3);
      t >>= 4;
      const int wo = (int)(t % Wo); t /= Wo;
      const int ho = (int)(t % Ho);
      const int n = (int)(t / Ho);
      const int h = 2 * ho - 1 + a, w = 2 * wo - 1 + b;
      u32x4 v = {0u, 0u, 0u, 0u};
      if ((unsigned)h < (unsigned)H && (unsigned)w < (unsigned)W) v = src[(((long)n * H + h) * W + w) * cchunks + cq];
      dst[i] = v;
    } else {
      const int w = (int)(t % W); t /= W;
      const int h = (int)(t % H);
      const int n = (int)(t / H);
      float acc[V];
#pragma unroll
      for (int e = 0; e < V; e++) acc[e] = 0.f;
#pragma unroll
      for (int ia = 0; ia < 2; ia++) {
        const int a = ((h + 1) & 1) + 2 * ia, ho = (h + 1 - a) >> 1;
        if ((unsigned)ho >= (unsigned)Ho) continue;
#pragma unroll
        for (int ib = 0; ib < 2; ib++) {
          const int b = ((w + 1) & 1) + 2 * ib, wo = (w + 1 - b) >> 1;
          if ((unsigned)wo >= (unsigned)Wo) continue;
          float f[V];
          Elem<BF16>::unpack(src[((((long)n * Ho + ho) * Wo + wo) * 16 + a * 4 + b) * cchunks + cq], f);
#pragma unroll
          for (int e = 0; e < V; e++) acc[e] += f[e];
        }
      }
      dst[i] = Elem<BF16>::pack(acc);
    }
  }
}
// count <= 4 inputs [N_k][H_k][W_k][Cp]; fwd: src = the inputs, dst = their slices of the mini-image batch; bwd: src = the slices of
// the batch's gradient, dst = the input gradients (a null dst: that input needs no gradient)
extern "C" int mt_patch4s2_multi(int dtype, int bwd, int count, const void* const* src, void* const* dst, const int* N, const int* H,
                                 const int* W, int Cp, mt_stream_t st) {
  MT_CHECK(count >= 1 && count <= 4 && Cp % 8 == 0, "patch4s2_multi: %d inputs, Cp %d", count, Cp);
  const int V = dtype == MT_BF16 ? 8 : 4;
  Patch4Multi m;
  long tot = 0;
  for (int k = 0; k < count; k++) {
    MT_CHECK(N[k] > 0 && H[k] >= 2 && W[k] >= 2 && H[k] % 2 == 0 && W[k] % 2 == 0, "patch4s2_multi: input %d is %d x %d x %d", k, N[k], H[k], W[k]);
    m.src[k] = (const u32x4*)src[k]; m.dst[k] = (u32x4*)dst[k]; m.N[k] = N[k]; m.H[k] = H[k]; m.W[k] = W[k];
    const long items = dst[k] == nullptr ? 0 : (bwd ? (long)N[k] * H[k] * W[k] : (long)N[k] * (H[k] / 2) * (W[k] / 2) * 16) * (Cp / V);
    tot += items;
    m.end[k] = tot;
  }
  for (int k = count; k < 4; k++) { m.src[k] = nullptr; m.dst[k] = nullptr; m.N[k] = m.H[k] = m.W[k] = 2; m.end[k] = tot; }
  m.count = count;
  if (tot == 0) return 0;
  const int blocks = (int)min((long)8192, (tot + 255) / 256);
  hipStream_t s = (hipStream_t)st;
  if (dtype == MT_BF16) {
    if (bwd) hipLaunchKernelGGL((patch4s2_multi_kernel<true, true>), dim3(blocks), dim3(256), 0, s, m, Cp / V);
    else hipLaunchKernelGGL((patch4s2_multi_kernel<true, false>), dim3(blocks), dim3(256), 0, s, m, Cp / V);
  } else {
    if (bwd) hipLaunchKernelGGL((patch4s2_multi_kernel<false, true>), dim3(blocks), dim3(256), 0, s, m, Cp / V);
    else hipLaunchKernelGGL((patch4s2_multi_kernel<false, false>), dim3(blocks), dim3(256), 0, s, m, Cp / V);
  }
  MT_LAUNCH_CHECK();
  return 0;
}

// AdaptiveAvgPool2d(1): one block per (n, channel slab); fp32 output [N][C]
template <bool BF16>
__global__ void gap_fwd_kernel(const u32x4* __restrict__ x, float* __restrict__ y, int HW, int cchunks, int C) {
  constexpr int V = Elem<BF16>::V;
  __shared__ float red[256];
  const int n = blockIdx.y;
  const int cq = blockIdx.x;
  float a[V];
#pragma unroll
  for (int e = 0; e < V; e++) a[e] = 0.f;
  for (int px = threadIdx.x; px < HW; px += blockDim.x) {
    float f[V];
    Elem<BF16>::unpack(x[((long)n * HW + px) * cchunks + cq], f);
#pragma unroll
    for (int e = 0; e < V; e++) a[e] += f[e];
  }
#pragma unroll
  for (int e = 0; e < V; e++) {
    float v = wave_sum(a[e]);
    __syncthreads();
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
    __syncthreads();
    if (threadIdx.x == 0) {
      float t = 0.f;
      for (int k = 0; k < (int)(blockDim.x >> 6); k++) t += red[k];
      const int ch = cq * V + e;
      if (ch < C) y[(long)n * C + ch] = t / (float)HW;
    }
  }
}
extern "C" int mt_gap_fwd(int dtype, const void* x, float* y, int N, int HW, int C, int Cp, mt_stream_t s) {
  const int V = dtype == MT_BF16 ? 8 : 4;
  dim3 grid(Cp / V, N);
  if (dtype == MT_BF16) hipLaunchKernelGGL((gap_fwd_kernel<true>), grid, dim3(256), 0, (hipStream_t)s, (const u32x4*)x, y, HW, Cp / V, C);
  else hipLaunchKernelGGL((gap_fwd_kernel<false>), grid, dim3(256), 0, (hipStream_t)s, (const u32x4*)x, y, HW, Cp / V, C);
  MT_LAUNCH_CHECK();
  return 0;
}
template <bool BF16>
__global__ void gap_bwd_kernel(const float* __restrict__ dy, u32x4* __restrict__ dx, long total, int HW,
                               int cchunks, int C) {
  constexpr int V = Elem<BF16>::V;
  const float inv = 1.f / (float)HW;
  for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    const int cq = (int)(i % cchunks);
    const int n = (int)(i / ((long)HW * cchunks));
    float f[V];
#pragma unroll
    for (int e = 0; e < V; e++) {
      const int ch = cq * V + e;
      f[e] = ch < C ? dy[(long)n * C + ch] * inv : 0.f;
    }
    dx[i] = Elem<BF16>::pack(f);
  }
}
extern "C" int mt_gap_bwd(int dtype, const float* dy, void* dx, int N, int HW, int C, int Cp, mt_stream_t s) {
  const int V = dtype == MT_BF16 ? 8 : 4;
  const long total = (long)N * HW * (Cp / V);
  if (total == 0) return 0;
  if (dtype == MT_BF16) hipLaunchKernelGGL((gap_bwd_kernel<true>), dim3(EW_GRID(total)), dim3(256), 0, (hipStream_t)s, dy, (u32x4*)dx, total, HW, Cp / V, C);
  else hipLaunchKernelGGL((gap_bwd_kernel<false>), dim3(EW_GRID(total)), dim3(256), 0, (hipStream_t)s, dy, (u32x4*)dx, total, HW, Cp / V, C);
  MT_LAUNCH_CHECK();
  return 0;
}

// ---- layout -------------------------------------------------------------------------------
template <bool SRC_BF16, bool DST_BF16>
__global__ void to_nhwc_kernel(const void* __restrict__ x, long sn, long sc, long sh, long sw,
                               void* __restrict__ y, int N, int C, int H, int W, int Cp) {
  const long total = (long)N * H * W * Cp;
  for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    const int c = (int)(i % Cp);
    long t = i / Cp;
    const int w = (int)(t % W); t /= W;
    const int h = (int)(t % H);
    const int n = (int)(t / H);
    float v = 0.f;
    if (c < C) {
      const long o = n * sn + c * sc + h * sh + w * sw;
      if constexpr (SRC_BF16) v = bf16_bits_to_f32(reinterpret_cast<const unsigned short*>(x)[o]);
      else v = reinterpret_cast<const float*>(x)[o];
    }
    if constexpr (DST_BF16) reinterpret_cast<unsigned short*>(y)[i] = f32_to_bf16_bits(v);
    else reinterpret_cast<float*>(y)[i] = v;
  }
}
extern "C" int mt_to_nhwc(int src_dtype, const void* x, int64_t sn, int64_t sc, int64_t sh, int64_t sw,
                          int dst_dtype, void* y, int N, int C, int H, int W, mt_stream_t s) {
  const int Cp = mt_padc(C);
  const long total = (long)N * H * W * Cp;
  if (total == 0) return 0;
  dim3 g(EW_GRID(total)), b(256);
  hipStream_t st = (hipStream_t)s;
  if (src_dtype == MT_BF16 && dst_dtype == MT_BF16) hipLaunchKernelGGL((to_nhwc_kernel<true, true>), g, b, 0, st, x, (long)sn, (long)sc, (long)sh, (long)sw, y, N, C, H, W, Cp);
  else if (src_dtype == MT_BF16) hipLaunchKernelGGL((to_nhwc_kernel<true, false>), g, b, 0, st, x, (long)sn, (long)sc, (long)sh, (long)sw, y, N, C, H, W, Cp);
  else if (dst_dtype == MT_BF16) hipLaunchKernelGGL((to_nhwc_kernel<false, true>), g, b, 0, st, x, (long)sn, (long)sc, (long)sh, (long)sw, y, N, C, H, W, Cp);
  else hipLaunchKernelGGL((to_nhwc_kernel<false, false>), g, b, 0, st, x, (long)sn, (long)sc, (long)sh, (long)sw, y, N, C, H, W, Cp);
  MT_LAUNCH_CHECK();
  return 0;
}
template <bool BF16>
__global__ void to_nchw_kernel(const void* __restrict__ x, float* __restrict__ y, int N, int C, int H, int W,
                               int Cp) {
  const long total = (long)N * C * H * W;
  for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    const int w = (int)(i % W);
    long t = i / W;
    const int h = (int)(t % H); t /= H;
    const int c = (int)(t % C);
    const int n = (int)(t / C);
    const long o = (((long)n * H + h) * W + w) * Cp + c;
    if constexpr (BF16) y[i] = bf16_bits_to_f32(reinterpret_cast<const unsigned short*>(x)[o]);
    else y[i] = reinterpret_cast<const float*>(x)[o];
  }
}
extern "C" int mt_to_nchw_f32(int dtype, const void* x, float* y, int N, int C, int H, int W, mt_stream_t s) {
  const long total = (long)N * C * H * W;
  if (total == 0) return 0;
  if (dtype == MT_BF16) hipLaunchKernelGGL((to_nchw_kernel<true>), dim3(EW_GRID(total)), dim3(256), 0, (hipStream_t)s, x, y, N, C, H, W, mt_padc(C));
  else hipLaunchKernelGGL((to_nchw_kernel<false>), dim3(EW_GRID(total)), dim3(256), 0, (hipStream_t)s, x, y, N, C, H, W, mt_padc(C));
  MT_LAUNCH_CHECK();
  return 0;
}

// out[n][px][0:C] = img[n][px][0:C]; out[n][px][C:C+D] = cls[n][:]; rest 0
template <bool BF16>
__global__ void cat_class_kernel(const void* __restrict__ img, const float* __restrict__ cls,
                                 void* __restrict__ out, long npix, int HW, int C, int Cip, int D, int Cop) {
  const long total = npix * Cop;
  for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    const int c = (int)(i % Cop);
    const long px = i / Cop;
    const int n = (int)(px / HW);
    if constexpr (BF16) {
      unsigned short v = 0;
      if (c < C) v = reinterpret_cast<const unsigned short*>(img)[px * Cip + c];
      else if (c < C + D) v = f32_to_bf16_bits(cls[(long)n * D + (c - C)]);
      reinterpret_cast<unsigned short*>(out)[i] = v;
    } else {
      float v = 0.f;
      if (c < C) v = reinterpret_cast<const float*>(img)[px * Cip + c];
      else if (c < C + D) v = cls[(long)n * D + (c - C)];
      reinterpret_cast<float*>(out)[i] = v;
    }
  }
}
extern "C" int mt_cat_class_planes(int dtype, const void* img, const float* cls, void* out, int N, int HW, int C,
                                   int D, mt_stream_t s) {
  const long npix = (long)N * HW;
  const int Cip = mt_padc(C), Cop = mt_padc(C + D);
  if (npix == 0) return 0;
  if (dtype == MT_BF16) hipLaunchKernelGGL((cat_class_kernel<true>), dim3(EW_GRID(npix * Cop)), dim3(256), 0, (hipStream_t)s, img, cls, out, npix, HW, C, Cip, D, Cop);
  else hipLaunchKernelGGL((cat_class_kernel<false>), dim3(EW_GRID(npix * Cop)), dim3(256), 0, (hipStream_t)s, img, cls, out, npix, HW, C, Cip, D, Cop);
  MT_LAUNCH_CHECK();
  return 0;
}
// y[px][0:C] = x[px][0:C] (x has Cx logical channels), pad of y zeroed
template <bool BF16>
__global__ void slice_channels_kernel(const void* __restrict__ x, void* __restrict__ y, long npix, int Cxp, int C,
                                      int Cyp) {
  const long total = npix * Cyp;
  for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    const int c = (int)(i % Cyp);
    const long px = i / Cyp;
    if constexpr (BF16) {
      reinterpret_cast<unsigned short*>(y)[i] = c < C ? reinterpret_cast<const unsigned short*>(x)[px * Cxp + c] : (unsigned short)0;
    } else {
      reinterpret_cast<float*>(y)[i] = c < C ? reinterpret_cast<const float*>(x)[px * Cxp + c] : 0.f;
    }
  }
}
extern "C" int mt_slice_channels(int dtype, const void* x, void* y, int N, int HW, int Cx, int C, mt_stream_t s) {
  const long npix = (long)N * HW;
  const int Cxp = mt_padc(Cx), Cyp = mt_padc(C);
  if (npix == 0) return 0;
  if (dtype == MT_BF16) hipLaunchKernelGGL((slice_channels_kernel<true>), dim3(EW_GRID(npix * Cyp)), dim3(256), 0, (hipStream_t)s, x, y, npix, Cxp, C, Cyp);
  else hipLaunchKernelGGL((slice_channels_kernel<false>), dim3(EW_GRID(npix * Cyp)), dim3(256), 0, (hipStream_t)s, x, y, npix, Cxp, C, Cyp);
  MT_LAUNCH_CHECK();
  return 0;
}
