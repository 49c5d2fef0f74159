// 1x1 convolutions with a thin side (<= 16 padded output channels), e.g. the decoder's to-RGB layer
// ConvTranspose2d(64, 3, k=1) + Tanh at full resolution (reference networks.py:251): 1M pixels x 64 x 3.
// These are HBM-bound (128 B read / 16 B written per pixel); the MFMA tile kernel spends its time on tile
// set-up there, so they get streaming kernels: LP = Cin/V lanes cooperate on one pixel (each owns one 16-byte
// chunk of its channels, coalesced), partial dot products are combined with DPP row reductions.
#include "mt_common.h"
#include "conv_params.h"
#include <type_traits>

template <int LP>
__device__ __forceinline__ float lp_sum(float v) {
  if constexpr (LP >= 2) v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0xB1, 0xf, 0xf, false));
  if constexpr (LP >= 4) v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x4E, 0xf, 0xf, false));
  if constexpr (LP >= 8) v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x141, 0xf, 0xf, false));
  if constexpr (LP >= 16) v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x140, 0xf, 0xf, false));
  return v;
}

// weights [COP][Cip] of type T -> fp32 in LDS
template <bool BF16>
__device__ __forceinline__ void load_w(const void* wpack, float* sw, int n) {
  for (int i = threadIdx.x; i < n; i += blockDim.x) {
    if constexpr (BF16) sw[i] = bf16_bits_to_f32(reinterpret_cast<const unsigned short*>(wpack)[i]);
    else sw[i] = reinterpret_cast<const float*>(wpack)[i];
  }
}

// y[px][co] = act(sum_ci x[px][ci] w[co][ci] + b[co])
template <bool BF16, int COP, int LP>
__global__ __launch_bounds__(256) void pw_fwd_kernel(const u32x4* __restrict__ x, const void* __restrict__ wpack,
                                                     const float* __restrict__ bias, int nbias, int nco,
                                                     void* __restrict__ y, long npix, int act, float slope) {
  constexpr int V = Elem<BF16>::V;
  constexpr int Cip = LP * V;
  __shared__ float sw[COP * Cip];
  load_w<BF16>(wpack, sw, COP * Cip);
  __syncthreads();
  const int chunk = threadIdx.x % LP;
  const long ngroups = npix * LP;
  for (long g = blockIdx.x * (long)blockDim.x + threadIdx.x; g < ((ngroups + 255) / 256) * 256;
       g += (long)gridDim.x * blockDim.x) {
    const long px = g / LP;
    const bool ok = px < npix;
    float xf[V];
    if (ok) Elem<BF16>::unpack(x[g], xf);
    else {
#pragma unroll
      for (int e = 0; e < V; e++) xf[e] = 0.f;
    }
    // only the real output rows are computed (to-RGB: 3 of the 8 padded ones; nco is uniform): the dot products,
    // DPP sums and above all the activation (tanh on one lane in LP) were what bounded this kernel, not HBM
    float out[COP];
#pragma unroll
    for (int co = 0; co < COP; co++) {
      out[co] = 0.f;
      if (co < nco) {
        float a = 0.f;
#pragma unroll
        for (int e = 0; e < V; e++) a += xf[e] * sw[co * Cip + chunk * V + e];
        out[co] = lp_sum<LP>(a);
      }
    }
    if (ok && chunk == 0) {
#pragma unroll
      for (int co = 0; co < COP; co++) {
        if (co < nco) {
          const float b = (bias != nullptr && co < nbias) ? bias[co] : 0.f;
          out[co] = act_apply(out[co] + b, act, slope);
        }
      }
      u32x4* yp = reinterpret_cast<u32x4*>(reinterpret_cast<char*>(y) + px * COP * Elem<BF16>::SZ);
#pragma unroll
      for (int q = 0; q < COP / V; q++) yp[q] = Elem<BF16>::pack(out + q * V);
    }
  }
}

// dx[px][ci] = sum_co dy[px][co] w[co][ci]
template <bool BF16, int COP, int LP>
__global__ __launch_bounds__(256) void pw_bwd_data_kernel(const u32x4* __restrict__ dy, const void* __restrict__ wpack,
                                                          u32x4* __restrict__ dx, long npix) {
  constexpr int V = Elem<BF16>::V;
  constexpr int Cip = LP * V;
  __shared__ float sw[COP * Cip];
  load_w<BF16>(wpack, sw, COP * Cip);
  __syncthreads();
  const int chunk = threadIdx.x % LP;
  const long ngroups = npix * LP;
  for (long g = blockIdx.x * (long)blockDim.x + threadIdx.x; g < ngroups; g += (long)gridDim.x * blockDim.x) {
    const long px = g / LP;
    float gy[COP];
#pragma unroll
    for (int q = 0; q < COP / V; q++) Elem<BF16>::unpack(dy[px * (COP / V) + q], gy + q * V);
    float o[V];
#pragma unroll
    for (int e = 0; e < V; e++) {
      float a = 0.f;
#pragma unroll
      for (int co = 0; co < COP; co++) a += gy[co] * sw[co * Cip + chunk * V + e];
      o[e] = a;
    }
    dx[g] = Elem<BF16>::pack(o);
  }
}

// slab[block][co][ci] = sum over the block's pixels of dy[px][co] x[px][ci]   (fp32; every block writes its whole
// slab, the caller adds the slabs in index order -- mt_launch_unpack -- so the result is reproducible: no atomics)
template <bool BF16, int COP, int LP>
__global__ __launch_bounds__(256) void pw_bwd_weight_kernel(const u32x4* __restrict__ x, const u32x4* __restrict__ dy,
                                                            float* __restrict__ out, long npix) {
  constexpr int V = Elem<BF16>::V;
  constexpr int Cip = LP * V;
  __shared__ float sacc[4][COP * Cip];
  const int chunk = threadIdx.x % LP;
  const long ngroups = npix * LP;
  float acc[COP][V];
#pragma unroll
  for (int co = 0; co < COP; co++)
#pragma unroll
    for (int e = 0; e < V; e++) acc[co][e] = 0.f;
  // four pixels per trip, all loads issued before the first use (one load -> 64 FMAs -> next load ran at 2.8 TB/s)
  constexpr int U = 4;
  const long stride = (long)gridDim.x * blockDim.x;
  for (long g0 = blockIdx.x * (long)blockDim.x + threadIdx.x; g0 < ngroups; g0 += U * stride) {
    u32x4 xv[U], gv[U][COP / V];
#pragma unroll
    for (int u = 0; u < U; u++) {
      const long g = g0 + u * stride;
      const bool ok = g < ngroups;
      const long px = (ok ? g : 0) / LP;
      xv[u] = ok ? x[g] : u32x4{0u, 0u, 0u, 0u};                      // (a zero x contributes nothing)
#pragma unroll
      for (int q = 0; q < COP / V; q++) gv[u][q] = dy[px * (COP / V) + q];
    }
#pragma unroll
    for (int u = 0; u < U; u++) {
      float xf[V], gy[COP];
      Elem<BF16>::unpack(xv[u], xf);
#pragma unroll
      for (int q = 0; q < COP / V; q++) Elem<BF16>::unpack(gv[u][q], gy + q * V);
#pragma unroll
      for (int co = 0; co < COP; co++)
#pragma unroll
        for (int e = 0; e < V; e++) acc[co][e] += gy[co] * xf[e];
    }
  }
  // lanes of a wave that own the same channel chunk (lane % LP), then the four waves in index order
  const int wv = threadIdx.x >> 6, lane = threadIdx.x & 63;
#pragma unroll
  for (int co = 0; co < COP; co++)
#pragma unroll
    for (int e = 0; e < V; e++) {
      float v = acc[co][e];
#pragma unroll
      for (int o = LP; o < 64; o <<= 1) v += __shfl_xor(v, o, 64);
      if (lane < LP) sacc[wv][co * Cip + chunk * V + e] = v;
    }
  __syncthreads();
  float* slab = out + (long)blockIdx.x * (COP * Cip);
  for (int i = threadIdx.x; i < COP * Cip; i += blockDim.x) slab[i] = (sacc[0][i] + sacc[1][i]) + (sacc[2][i] + sacc[3][i]);
}

bool mt_pointwise_small(const mt_conv_desc* d) {
  if (d->kh != 1 || d->kw != 1 || d->stride != 1 || d->pad != 0 || d->out_pad != 0) return false;
  const int V = d->dtype == MT_BF16 ? 8 : 4;
  const int Cip = mt_padc(d->Ci), Cop = mt_padc(d->Co);
  const int LP = Cip / V;
  // must mirror the instantiation table of PW_DISPATCH
  if (d->dtype == MT_BF16) return (Cop == 8 && (LP == 1 || LP == 2 || LP == 4 || LP == 8 || LP == 16)) || (Cop == 16 && LP == 8);
  return (Cop == 8 && (LP == 2 || LP == 4 || LP == 8 || LP == 16)) || (Cop == 16 && LP == 16);
}

#define PW_DISPATCH(KERN, ...)                                                                        \
  do {                                                                                                \
    const int V = d->dtype == MT_BF16 ? 8 : 4;                                                        \
    const int LP = mt_padc(d->Ci) / V, COP = mt_padc(d->Co);                                          \
    bool done = false;                                                                                \
    auto go = [&](auto bf, auto cop, auto lp) {                                                       \
      if (!done && (d->dtype == MT_BF16) == decltype(bf)::value && COP == decltype(cop)::value &&     \
          LP == decltype(lp)::value) {                                                                \
        hipLaunchKernelGGL((KERN<decltype(bf)::value, decltype(cop)::value, decltype(lp)::value>),    \
                           dim3(grid), dim3(256), 0, s, __VA_ARGS__);                                 \
        done = true;                                                                                  \
      }                                                                                               \
    };                                                                                                \
    using T = std::true_type; using F = std::false_type;                                              \
    go(T{}, std::integral_constant<int, 8>{}, std::integral_constant<int, 1>{});                      \
    go(T{}, std::integral_constant<int, 8>{}, std::integral_constant<int, 2>{});                      \
    go(T{}, std::integral_constant<int, 8>{}, std::integral_constant<int, 4>{});                      \
    go(T{}, std::integral_constant<int, 8>{}, std::integral_constant<int, 8>{});                      \
    go(T{}, std::integral_constant<int, 8>{}, std::integral_constant<int, 16>{});                     \
    go(T{}, std::integral_constant<int, 16>{}, std::integral_constant<int, 8>{});                     \
    go(F{}, std::integral_constant<int, 8>{}, std::integral_constant<int, 2>{});                      \
    go(F{}, std::integral_constant<int, 8>{}, std::integral_constant<int, 4>{});                      \
    go(F{}, std::integral_constant<int, 8>{}, std::integral_constant<int, 8>{});                      \
    go(F{}, std::integral_constant<int, 8>{}, std::integral_constant<int, 16>{});                     \
    go(F{}, std::integral_constant<int, 16>{}, std::integral_constant<int, 16>{});                    \
    if (!done) return -1;                                                                             \
  } while (0)

// each returns -1 when the (dtype, channels) combination has no instantiation (caller falls back)
int mt_pw_fwd(const mt_conv_desc* d, const void* x, const void* wpack, const float* bias, void* y, long npix,
              hipStream_t s) {
  const long groups = npix * (mt_padc(d->Ci) / (d->dtype == MT_BF16 ? 8 : 4));
  const int grid = (int)min((long)8192, (groups + 255) / 256);
  PW_DISPATCH(pw_fwd_kernel, (const u32x4*)x, wpack, bias, bias ? d->Co : 0, d->Co, y, npix, d->act, d->slope);
  MT_LAUNCH_CHECK();
  return 0;
}
int mt_pw_bwd_data(const mt_conv_desc* d, const void* dy, const void* wpack, void* dx, long npix, hipStream_t s) {
  const long groups = npix * (mt_padc(d->Ci) / (d->dtype == MT_BF16 ? 8 : 4));
  const int grid = (int)min((long)8192, (groups + 255) / 256);
  PW_DISPATCH(pw_bwd_data_kernel, (const u32x4*)dy, wpack, (u32x4*)dx, npix);
  MT_LAUNCH_CHECK();
  return 0;
}
// `out` receives `*nslabs` slabs of [Cop][Cip] floats (at most max_slabs), one per block
int mt_pw_bwd_weight(const mt_conv_desc* d, const void* x, const void* dy, float* out, long npix, int max_slabs,
                     int* nslabs, hipStream_t s) {
  const long groups = npix * (mt_padc(d->Ci) / (d->dtype == MT_BF16 ? 8 : 4));
  int grid = (int)min((long)1024, (groups + 255) / 256);
  if (grid > max_slabs) grid = max_slabs;
  if (grid < 1) return -1;
  *nslabs = grid;
  PW_DISPATCH(pw_bwd_weight_kernel, (const u32x4*)x, (const u32x4*)dy, out, npix);
  MT_LAUNCH_CHECK();
  return 0;
}
