// Scalar loss reductions and their gradients (fp32 accumulation, wavefront-shuffle reduce, block partials combined
// by the last block to finish in block-index order: reproducible, no floating-point atomics).
// Reference: GANLoss vanilla = BCEWithLogitsLoss vs constant 0/1 (loss.py:52-64),
// classification BCEWithLogitsLoss vs one-hot (adain_model.py:74), L1Loss (303-305,379-380),
// _l2_regularize (396-399), reparameterize (networks.py:130-135), KL sum (adain_model.py:313-314).
#include "mt_common.h"

#define RED_GRID(total) (int)min((long)1024, ((long)(total) + 255) / 256)

// Grid-wide sum of one value per thread into *dst, bit-reproducible: wave shuffle, the block's waves in index order,
// the block partial into a scratch slot; the block that takes the last ticket adds the slots in a fixed order and
// writes the scalar.  The scratch is a module-level device array with one SLOT PER STREAM (red_slot: the C ABI accepts
// any stream, and two reductions in flight on different streams must not share partials or the ticket):
// launches on one stream are ordered, launches on different streams use different slots.  grid <= 1024 blocks of 256
// threads; EVERY thread of every block must call it exactly once.
#define RED_SLOTS 16
__device__ float g_red_part[RED_SLOTS][1024];
__device__ unsigned g_red_ticket[RED_SLOTS];
#include <mutex>
// slot of a stream (first come, first served; the table lives as long as the library: streams are few and long-lived)
static int red_slot(hipStream_t s) {
  static std::mutex mu;
  static hipStream_t tab[RED_SLOTS];
  static int n = 0;
  std::lock_guard<std::mutex> g(mu);
  for (int i = 0; i < n; i++)
    if (tab[i] == s) return i;
  if (n < RED_SLOTS) { tab[n] = s; return n++; }
  // more streams than slots: share one by address (ordering between such streams is then the caller's business, as it was
  // for every stream before round 3)
  return (int)(((uintptr_t)s >> 6) % RED_SLOTS);
}
#define RED_SLOT_OR_FAIL(slot, s)                                                                     \
  const int slot = red_slot(s);                                                                       \
  MT_CHECK(slot >= 0 && slot < RED_SLOTS, "loss reductions: bad scratch slot %d", slot)
__device__ __forceinline__ void grid_sum_to(float* dst, float v, int slot) {
  __shared__ float sw[4];
  __shared__ int last;
  v = wave_sum(v);
  const int w = threadIdx.x >> 6;
  if ((threadIdx.x & 63) == 0) sw[w] = v;
  __syncthreads();
  if (threadIdx.x == 0) {
    const float b = (sw[0] + sw[1]) + (sw[2] + sw[3]);
    __hip_atomic_store(&g_red_part[slot][blockIdx.x], b, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __threadfence();
    const unsigned t = atomicAdd(&g_red_ticket[slot], 1u);
    last = (t == gridDim.x - 1) ? 1 : 0;
  }
  __syncthreads();
  if (last) {
    __threadfence();
    float a = 0.f;
    for (int i = threadIdx.x; i < (int)gridDim.x; i += 256)
      a += __hip_atomic_load(&g_red_part[slot][i], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    a = wave_sum(a);
    __syncthreads();
    if ((threadIdx.x & 63) == 0) sw[w] = a;
    __syncthreads();
    if (threadIdx.x == 0) {
      *dst = (sw[0] + sw[1]) + (sw[2] + sw[3]);
      g_red_ticket[slot] = 0;
    }
  }
}
// numerically stable BCE-with-logits: max(x,0) - x*t + log(1+exp(-|x|))
__device__ __forceinline__ float bce_logits(float x, float t) {
  return fmaxf(x, 0.f) - x * t + log1pf(expf(-fabsf(x)));
}
__device__ __forceinline__ float sigmoidf(float x) { return 1.f / (1.f + expf(-x)); }

static int zero_scalar(float* p, hipStream_t s) {
  if (hipMemsetAsync(p, 0, sizeof(float), s) != hipSuccess) { mt_set_error("loss: memset failed"); return 2; }
  return 0;
}

// ---- BCE vs constant target on an NHWC-padded map -------------------------------------------
template <bool BF16>
__global__ void bce_const_fwd_kernel(const void* __restrict__ x, float t, float* __restrict__ loss, long npix,
                                     int C, int Cp, float inv_count, int red_slot_id) {
  const long total = npix * C;
  float a = 0.f;
  for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    const long px = i / C;
    const int c = (int)(i % C);
    float v;
    if constexpr (BF16) v = bf16_bits_to_f32(reinterpret_cast<const unsigned short*>(x)[px * Cp + c]);
    else v = reinterpret_cast<const float*>(x)[px * Cp + c];
    a += bce_logits(v, t);
  }
  grid_sum_to(loss, a * inv_count, red_slot_id);
}
template <bool BF16>
__global__ void bce_const_bwd_kernel(const void* __restrict__ x, float t, const float* __restrict__ gscale,
                                     void* __restrict__ dx, long npix, int C, int Cp, float inv_count) {
  const long total = npix * Cp;
  const float gs = gscale[0] * inv_count;
  for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    const int c = (int)(i % Cp);
    float g = 0.f;
    if (c < C) {
      float v;
      if constexpr (BF16) v = bf16_bits_to_f32(reinterpret_cast<const unsigned short*>(x)[i]);
      else v = reinterpret_cast<const float*>(x)[i];
      g = (sigmoidf(v) - t) * gs;
    }
    if constexpr (BF16) reinterpret_cast<unsigned short*>(dx)[i] = f32_to_bf16_bits(g);
    else reinterpret_cast<float*>(dx)[i] = g;
  }
}
extern "C" int mt_bce_const_fwd(int dtype, const void* x, float t, float* loss, size_t npix, int C, int Cp,
                                mt_stream_t st) {
  hipStream_t s = (hipStream_t)st;
  RED_SLOT_OR_FAIL(red_slot_id, s);
  if (zero_scalar(loss, s)) return 2;
  const long total = (long)npix * C;
  if (total == 0) return 0;
  const float inv = 1.f / (float)total;
  if (dtype == MT_BF16) hipLaunchKernelGGL((bce_const_fwd_kernel<true>), dim3(RED_GRID(total)), dim3(256), 0, s, x, t, loss, (long)npix, C, Cp, inv, red_slot_id);
  else hipLaunchKernelGGL((bce_const_fwd_kernel<false>), dim3(RED_GRID(total)), dim3(256), 0, s, x, t, loss, (long)npix, C, Cp, inv, red_slot_id);
  MT_LAUNCH_CHECK();
  return 0;
}
extern "C" int mt_bce_const_bwd(int dtype, const void* x, float t, const float* gscale, void* dx, size_t npix,
                                int C, int Cp, mt_stream_t st) {
  hipStream_t s = (hipStream_t)st;
  const long total = (long)npix * Cp;
  if (total == 0) return 0;
  const float inv = 1.f / (float)((long)npix * C);
  if (dtype == MT_BF16) hipLaunchKernelGGL((bce_const_bwd_kernel<true>), dim3(RED_GRID(total)), dim3(256), 0, s, x, t, gscale, dx, (long)npix, C, Cp, inv);
  else hipLaunchKernelGGL((bce_const_bwd_kernel<false>), dim3(RED_GRID(total)), dim3(256), 0, s, x, t, gscale, dx, (long)npix, C, Cp, inv);
  MT_LAUNCH_CHECK();
  return 0;
}

// ---- the other GANLoss modes on an NHWC-padded map (loss.py:44-61, adain_model.py:209-210,293-295,367-369) ----
//   MT_GAN_LSGAN      mean (x - t)^2                 nn.MSELoss vs ones / zeros
//   MT_GAN_HINGE_D    mean relu(1 - x) (t = 1)  /  mean relu(1 + x) (t = 0)     discriminator hinge terms
//   MT_GAN_NEG_MEAN   -mean x (t = 1)  /  mean x (t = 0)                        generator hinge term, wgangp
__device__ __forceinline__ float gan_term(int mode, float v, float t) {
  switch (mode) {
    case MT_GAN_LSGAN: return (v - t) * (v - t);
    case MT_GAN_HINGE_D: { const float z = t > 0.5f ? 1.f - v : 1.f + v; return z > 0.f ? z : 0.f; }
    default: return t > 0.5f ? -v : v;
  }
}
__device__ __forceinline__ float gan_term_grad(int mode, float v, float t) {
  switch (mode) {
    case MT_GAN_LSGAN: return 2.f * (v - t);
    case MT_GAN_HINGE_D: { const float z = t > 0.5f ? 1.f - v : 1.f + v; return z > 0.f ? (t > 0.5f ? -1.f : 1.f) : 0.f; }
    default: return t > 0.5f ? -1.f : 1.f;
  }
}
template <bool BF16>
__global__ void gan_const_fwd_kernel(int mode, const void* __restrict__ x, float t, float* __restrict__ loss, long npix,
                                     int C, int Cp, float inv_count, int red_slot_id) {
  const long total = npix * C;
  float a = 0.f;
  for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    const long px = i / C;
    const int c = (int)(i % C);
    float v;
    if constexpr (BF16) v = bf16_bits_to_f32(reinterpret_cast<const unsigned short*>(x)[px * Cp + c]);
    else v = reinterpret_cast<const float*>(x)[px * Cp + c];
    a += gan_term(mode, v, t);
  }
  grid_sum_to(loss, a * inv_count, red_slot_id);
}
template <bool BF16>
__global__ void gan_const_bwd_kernel(int mode, const void* __restrict__ x, float t, const float* __restrict__ gscale,
                                     void* __restrict__ dx, long npix, int C, int Cp, float inv_count) {
  const long total = npix * Cp;
  const float gs = gscale[0] * inv_count;
  for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    const int c = (int)(i % Cp);
    float g = 0.f;
    if (c < C) {
      float v;
      if constexpr (BF16) v = bf16_bits_to_f32(reinterpret_cast<const unsigned short*>(x)[i]);
      else v = reinterpret_cast<const float*>(x)[i];
      g = gan_term_grad(mode, v, t) * gs;
    }
    if constexpr (BF16) reinterpret_cast<unsigned short*>(dx)[i] = f32_to_bf16_bits(g);
    else reinterpret_cast<float*>(dx)[i] = g;
  }
}
extern "C" int mt_gan_const_fwd(int dtype, int mode, const void* x, float t, float* loss, size_t npix, int C, int Cp,
                                mt_stream_t st) {
  MT_CHECK(mode == MT_GAN_LSGAN || mode == MT_GAN_HINGE_D || mode == MT_GAN_NEG_MEAN, "gan_const: bad mode %d", mode);
  hipStream_t s = (hipStream_t)st;
  RED_SLOT_OR_FAIL(red_slot_id, s);
  if (zero_scalar(loss, s)) return 2;
  const long total = (long)npix * C;
  if (total == 0) return 0;
  const float inv = 1.f / (float)total;
  if (dtype == MT_BF16) hipLaunchKernelGGL((gan_const_fwd_kernel<true>), dim3(RED_GRID(total)), dim3(256), 0, s, mode, x, t, loss, (long)npix, C, Cp, inv, red_slot_id);
  else hipLaunchKernelGGL((gan_const_fwd_kernel<false>), dim3(RED_GRID(total)), dim3(256), 0, s, mode, x, t, loss, (long)npix, C, Cp, inv, red_slot_id);
  MT_LAUNCH_CHECK();
  return 0;
}
extern "C" int mt_gan_const_bwd(int dtype, int mode, const void* x, float t, const float* gscale, void* dx, size_t npix,
                                int C, int Cp, mt_stream_t st) {
  MT_CHECK(mode == MT_GAN_LSGAN || mode == MT_GAN_HINGE_D || mode == MT_GAN_NEG_MEAN, "gan_const: bad mode %d", mode);
  hipStream_t s = (hipStream_t)st;
  const long total = (long)npix * Cp;
  if (total == 0) return 0;
  const float inv = 1.f / (float)((long)npix * C);
  if (dtype == MT_BF16) hipLaunchKernelGGL((gan_const_bwd_kernel<true>), dim3(RED_GRID(total)), dim3(256), 0, s, mode, x, t, gscale, dx, (long)npix, C, Cp, inv);
  else hipLaunchKernelGGL((gan_const_bwd_kernel<false>), dim3(RED_GRID(total)), dim3(256), 0, s, mode, x, t, gscale, dx, (long)npix, C, Cp, inv);
  MT_LAUNCH_CHECK();
  return 0;
}

// ---- BCE vs per-element target, fp32 vectors ---------------------------------------------------
__global__ void bce_target_fwd_kernel(const float* __restrict__ x, const float* __restrict__ t,
                                      float* __restrict__ loss, long n, float inv, int red_slot_id) {
  float a = 0.f;
  for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x)
    a += bce_logits(x[i], t[i]);
  grid_sum_to(loss, a * inv, red_slot_id);
}
__global__ void bce_target_bwd_kernel(const float* __restrict__ x, const float* __restrict__ t,
                                      const float* __restrict__ gscale, float* __restrict__ dx, long n,
                                      float inv) {
  const float gs = gscale[0] * inv;
  for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x)
    dx[i] = (sigmoidf(x[i]) - t[i]) * gs;
}
extern "C" int mt_bce_target_fwd(const float* x, const float* t, float* loss, size_t n, mt_stream_t st) {
  hipStream_t s = (hipStream_t)st;
  RED_SLOT_OR_FAIL(red_slot_id, s);
  if (zero_scalar(loss, s)) return 2;
  if (n == 0) return 0;
  hipLaunchKernelGGL(bce_target_fwd_kernel, dim3(RED_GRID(n)), dim3(256), 0, s, x, t, loss, (long)n, 1.f / (float)n, red_slot_id);
  MT_LAUNCH_CHECK();
  return 0;
}
extern "C" int mt_bce_target_bwd(const float* x, const float* t, const float* gscale, float* dx, size_t n,
                                 mt_stream_t st) {
  if (n == 0) return 0;
  hipLaunchKernelGGL(bce_target_bwd_kernel, dim3(RED_GRID(n)), dim3(256), 0, (hipStream_t)st, x, t, gscale, dx, (long)n, 1.f / (float)n);
  MT_LAUNCH_CHECK();
  return 0;
}

// ---- L1 / mean-square over padded tensors (pad elements are zero in both operands) ---------
template <bool BF16, int MODE>  // MODE 0: |a-b|, 1: a^2
__global__ void absdiff_fwd_kernel(const u32x4* __restrict__ a, const u32x4* __restrict__ b,
                                   float* __restrict__ loss, long nchunks, float inv, int red_slot_id) {
  constexpr int V = Elem<BF16>::V;
  float acc = 0.f;
  for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < nchunks; i += (long)gridDim.x * blockDim.x) {
    float f[V], g[V];
    Elem<BF16>::unpack(a[i], f);
    if constexpr (MODE == 0) {
      Elem<BF16>::unpack(b[i], g);
#pragma unroll
      for (int e = 0; e < V; e++) acc += fabsf(f[e] - g[e]);
    } else {
#pragma unroll
      for (int e = 0; e < V; e++) acc += f[e] * f[e];
    }
  }
  grid_sum_to(loss, acc * inv, red_slot_id);
}
template <bool BF16, int MODE>
__global__ void absdiff_bwd_kernel(const u32x4* __restrict__ a, const u32x4* __restrict__ b,
                                   const float* __restrict__ gscale, u32x4* __restrict__ da,
                                   u32x4* __restrict__ db, long nchunks, float inv) {
  constexpr int V = Elem<BF16>::V;
  const float gs = gscale[0] * inv;
  for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < nchunks; i += (long)gridDim.x * blockDim.x) {
    float f[V], g[V];
    Elem<BF16>::unpack(a[i], f);
    if constexpr (MODE == 0) {
      Elem<BF16>::unpack(b[i], g);
#pragma unroll
      for (int e = 0; e < V; e++) {
        const float d = f[e] - g[e];
        f[e] = d > 0.f ? gs : (d < 0.f ? -gs : 0.f);
        g[e] = -f[e];
      }
      if (da) da[i] = Elem<BF16>::pack(f);
      if (db) db[i] = Elem<BF16>::pack(g);
    } else {
#pragma unroll
      for (int e = 0; e < V; e++) f[e] = 2.f * f[e] * gs;
      da[i] = Elem<BF16>::pack(f);
    }
  }
}
template <int MODE>
static int absdiff_fwd(int dtype, const void* a, const void* b, float* loss, size_t n, size_t count, hipStream_t s) {
  RED_SLOT_OR_FAIL(red_slot_id, s);
  if (zero_scalar(loss, s)) return 2;
  const int V = dtype == MT_BF16 ? 8 : 4;
  MT_CHECK(n % V == 0, "loss: element count %zu not a multiple of %d", n, V);
  const long nc = (long)(n / V);
  if (nc == 0 || count == 0) return 0;
  const float inv = 1.f / (float)count;
  if (dtype == MT_BF16) hipLaunchKernelGGL((absdiff_fwd_kernel<true, MODE>), dim3(RED_GRID(nc)), dim3(256), 0, s, (const u32x4*)a, (const u32x4*)b, loss, nc, inv, red_slot_id);
  else hipLaunchKernelGGL((absdiff_fwd_kernel<false, MODE>), dim3(RED_GRID(nc)), dim3(256), 0, s, (const u32x4*)a, (const u32x4*)b, loss, nc, inv, red_slot_id);
  MT_LAUNCH_CHECK();
  return 0;
}
template <int MODE>
static int absdiff_bwd(int dtype, const void* a, const void* b, const float* gscale, void* da, void* db, size_t n,
                       size_t count, hipStream_t s) {
  const int V = dtype == MT_BF16 ? 8 : 4;
  MT_CHECK(n % V == 0, "loss: element count %zu not a multiple of %d", n, V);
  const long nc = (long)(n / V);
  if (nc == 0 || count == 0) return 0;
  const float inv = 1.f / (float)count;
  const int grid = (int)min((long)16384, (nc + 255) / 256);
  if (dtype == MT_BF16) hipLaunchKernelGGL((absdiff_bwd_kernel<true, MODE>), dim3(grid), dim3(256), 0, s, (const u32x4*)a, (const u32x4*)b, gscale, (u32x4*)da, (u32x4*)db, nc, inv);
  else hipLaunchKernelGGL((absdiff_bwd_kernel<false, MODE>), dim3(grid), dim3(256), 0, s, (const u32x4*)a, (const u32x4*)b, gscale, (u32x4*)da, (u32x4*)db, nc, inv);
  MT_LAUNCH_CHECK();
  return 0;
}
extern "C" int mt_l1_fwd(int dtype, const void* a, const void* b, float* loss, size_t n, size_t count, mt_stream_t s) {
  return absdiff_fwd<0>(dtype, a, b, loss, n, count, (hipStream_t)s);
}
extern "C" int mt_l1_bwd(int dtype, const void* a, const void* b, const float* gscale, void* da, void* db, size_t n,
                         size_t count, mt_stream_t s) {
  return absdiff_bwd<0>(dtype, a, b, gscale, da, db, n, count, (hipStream_t)s);
}
extern "C" int mt_l2mean_fwd(int dtype, const void* x, float* loss, size_t n, size_t count, mt_stream_t s) {
  return absdiff_fwd<1>(dtype, x, nullptr, loss, n, count, (hipStream_t)s);
}
extern "C" int mt_l2mean_bwd(int dtype, const void* x, const float* gscale, void* dx, size_t n, size_t count,
                             mt_stream_t s) {
  return absdiff_bwd<1>(dtype, x, nullptr, gscale, dx, nullptr, n, count, (hipStream_t)s);
}

// ---- VAE branch ---------------------------------------------------------------------------------
__global__ void reparam_fwd_kernel(const float* __restrict__ mu, const float* __restrict__ logvar,
                                   const float* __restrict__ eps, float* __restrict__ z, long n) {
  for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x)
    z[i] = eps[i] * expf(0.5f * logvar[i]) + mu[i];
}
__global__ void reparam_bwd_kernel(const float* __restrict__ logvar, const float* __restrict__ eps,
                                   const float* __restrict__ dz, float* __restrict__ dmu,
                                   float* __restrict__ dlogvar, long n) {
  for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
    dmu[i] = dz[i];
    dlogvar[i] = dz[i] * eps[i] * 0.5f * expf(0.5f * logvar[i]);
  }
}
extern "C" int mt_reparam_fwd(const float* mu, const float* logvar, const float* eps, float* z, size_t n, mt_stream_t s) {
  if (n == 0) return 0;
  hipLaunchKernelGGL(reparam_fwd_kernel, dim3(RED_GRID(n)), dim3(256), 0, (hipStream_t)s, mu, logvar, eps, z, (long)n);
  MT_LAUNCH_CHECK();
  return 0;
}
extern "C" int mt_reparam_bwd(const float* logvar, const float* eps, const float* dz, float* dmu, float* dlogvar,
                              size_t n, mt_stream_t s) {
  if (n == 0) return 0;
  hipLaunchKernelGGL(reparam_bwd_kernel, dim3(RED_GRID(n)), dim3(256), 0, (hipStream_t)s, logvar, eps, dz, dmu, dlogvar, (long)n);
  MT_LAUNCH_CHECK();
  return 0;
}
__global__ void kl_fwd_kernel(const float* __restrict__ mu, const float* __restrict__ logvar,
                              float* __restrict__ kl, long n, int red_slot_id) {
  float a = 0.f;
  for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x)
    a += 1.f + logvar[i] - mu[i] * mu[i] - expf(logvar[i]);
  grid_sum_to(kl, -0.5f * a, red_slot_id);
}
__global__ void kl_bwd_kernel(const float* __restrict__ mu, const float* __restrict__ logvar,
                              const float* __restrict__ gscale, float* __restrict__ dmu,
                              float* __restrict__ dlogvar, long n) {
  const float gs = gscale[0];
  for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
    dmu[i] = gs * mu[i];
    dlogvar[i] = gs * -0.5f * (1.f - expf(logvar[i]));
  }
}
extern "C" int mt_kl_fwd(const float* mu, const float* logvar, float* kl, size_t n, mt_stream_t st) {
  hipStream_t s = (hipStream_t)st;
  RED_SLOT_OR_FAIL(red_slot_id, s);
  if (zero_scalar(kl, s)) return 2;
  if (n == 0) return 0;
  hipLaunchKernelGGL(kl_fwd_kernel, dim3(RED_GRID(n)), dim3(256), 0, s, mu, logvar, kl, (long)n, red_slot_id);
  MT_LAUNCH_CHECK();
  return 0;
}
extern "C" int mt_kl_bwd(const float* mu, const float* logvar, const float* gscale, float* dmu, float* dlogvar,
                         size_t n, mt_stream_t s) {
  if (n == 0) return 0;
  hipLaunchKernelGGL(kl_bwd_kernel, dim3(RED_GRID(n)), dim3(256), 0, (hipStream_t)s, mu, logvar, gscale, dmu, dlogvar, (long)n);
  MT_LAUNCH_CHECK();
  return 0;
}

// ---- weighted sum of loss scalars: the whole "loss = a + 10*b + 0.01*c ..." expression in ONE launch -------------------
// (adain_model.py:193-195, 316-321, 381-389).  Term i (device scalar) belongs to group gid[i] with weight w[i]:
//   out[g] = sum_{i in g} w[i]*t[i]  (g < G: the values the model logs),  out[G] = sum_g Wb[g]*out[g]  (what is
//   differentiated),  out[G+1] = sum_g Wr[g]*out[g]  (what is reported as the total).
#define MT_LOSS_MAX_TERMS 16
#define MT_LOSS_MAX_GROUPS 8
struct LossSumArgs {
  const float* t[MT_LOSS_MAX_TERMS];
  float w[MT_LOSS_MAX_TERMS];
  int gid[MT_LOSS_MAX_TERMS];
  float Wb[MT_LOSS_MAX_GROUPS], Wr[MT_LOSS_MAX_GROUPS];
  int n, G;
};
__global__ void loss_sum_fwd_kernel(LossSumArgs a, float* __restrict__ out) {
  if (threadIdx.x != 0) return;
  float v[MT_LOSS_MAX_GROUPS];
  for (int g = 0; g < a.G; g++) v[g] = 0.f;
  for (int i = 0; i < a.n; i++) v[a.gid[i]] += a.w[i] * a.t[i][0];      // fixed order
  float tb = 0.f, tr = 0.f;
  for (int g = 0; g < a.G; g++) { out[g] = v[g]; tb += a.Wb[g] * v[g]; tr += a.Wr[g] * v[g]; }
  out[a.G] = tb;
  out[a.G + 1] = tr;
}
__global__ void loss_sum_bwd_kernel(LossSumArgs a, const float* __restrict__ g, float* __restrict__ dt) {
  const int i = threadIdx.x;
  if (i < a.n) dt[i] = g[0] * a.Wb[a.gid[i]] * a.w[i];
}
static int fill_loss_args(LossSumArgs* a, const float* const* terms, const float* w, const int* gid, int n,
                          const float* Wb, const float* Wr, int G) {
  MT_CHECK(n >= 1 && n <= MT_LOSS_MAX_TERMS && G >= 1 && G <= MT_LOSS_MAX_GROUPS, "loss_sum: %d terms, %d groups", n, G);
  a->n = n; a->G = G;
  for (int i = 0; i < n; i++) {
    MT_CHECK(gid[i] >= 0 && gid[i] < G && (terms == nullptr || terms[i] != nullptr), "loss_sum: bad term %d", i);
    a->t[i] = terms ? terms[i] : nullptr; a->w[i] = w[i]; a->gid[i] = gid[i];
  }
  for (int g = 0; g < G; g++) { a->Wb[g] = Wb[g]; a->Wr[g] = Wr ? Wr[g] : Wb[g]; }
  return 0;
}
extern "C" int mt_loss_sum_fwd(const float* const* terms, const float* w, const int* gid, int n, const float* Wb,
                               const float* Wr, int G, float* out, mt_stream_t s) {
  LossSumArgs a;
  if (fill_loss_args(&a, terms, w, gid, n, Wb, Wr, G)) return 1;
  hipLaunchKernelGGL(loss_sum_fwd_kernel, dim3(1), dim3(64), 0, (hipStream_t)s, a, out);
  MT_LAUNCH_CHECK();
  return 0;
}
extern "C" int mt_loss_sum_bwd(const float* gtotal, const float* w, const int* gid, int n, const float* Wb, int G,
                               float* dterms, mt_stream_t s) {
  LossSumArgs a;
  if (fill_loss_args(&a, nullptr, w, gid, n, Wb, nullptr, G)) return 1;
  hipLaunchKernelGGL(loss_sum_bwd_kernel, dim3(1), dim3(64), 0, (hipStream_t)s, a, gtotal, dterms);
  MT_LAUNCH_CHECK();
  return 0;
}
