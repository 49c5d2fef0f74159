// Spectral normalisation of a convolution weight (--dis_sn): one power iteration per forward call and the
// scaled weight W / sigma, plus the gradient of that scaling.
// Reference: functions.py:113-121 -> torch.nn.utils.spectral_norm(module, 'weight', 1, 1e-12, dim=0), applied to
// the Conv2d of every ConvBlock of the discriminators (blocks.py:33-34, networks.py:363-371, 449-451):
//     v <- normalize(W^T u);  u <- normalize(W v);  sigma = u . (W v);  weight = W / sigma
// with W the [Cout, Cin*kh*kw] view of weight_orig, u and v updated in place without gradient and then treated as
// constants, so  dL/dW = (G - <G, W/sigma> u v^T) / sigma.
//
// All passes are HBM-bound streams over W (fp32, up to 2048 x 16384): two reads for the iteration, one read + one
// write for the scaling.  Reductions are two-stage through the workspace (fixed order: results are reproducible).
#include "mt_common.h"

#define SN_ROWS_PER_SLICE 64
#define SN_MAX_SLICES 32
#define SN_DOT_BLOCKS 1024

__device__ __forceinline__ float sn_block_sum(float v, float* red) {
  // sum over the block, result in every thread (red: >= blockDim.x / 64 floats)
  v = wave_sum(v);
  const int nw = blockDim.x >> 6;
  __syncthreads();
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
  __syncthreads();
  float s = 0.f;
  for (int i = 0; i < nw; i++) s += red[i];
  return s;
}

static int sn_slices(int rows) {
  int n = cdiv(rows, SN_ROWS_PER_SLICE);
  return n > SN_MAX_SLICES ? SN_MAX_SLICES : n;
}

extern "C" size_t mt_sn_ws_bytes(int rows, int cols) {
  // [slices][cols] partial W^T u | t[rows] | SN_DOT_BLOCKS partial dots
  return ((size_t)sn_slices(rows) * cols + rows + SN_DOT_BLOCKS) * sizeof(float);
}

// part[sl][c] = sum_{r in slice sl} W[r][c] u[r]   (threads along c: coalesced rows of W)
__global__ __launch_bounds__(256) void sn_wtu_kernel(const float* __restrict__ W, const float* __restrict__ u,
                                                     float* __restrict__ part, int rows, int cols, int rps) {
  const int c = blockIdx.x * 256 + threadIdx.x;
  if (c >= cols) return;
  const int r0 = blockIdx.y * rps;
  const int r1 = min(rows, r0 + rps);
  float a = 0.f;
#pragma unroll 4
  for (int r = r0; r < r1; r++) a += W[(long)r * cols + c] * u[r];
  part[(long)blockIdx.y * cols + c] = a;
}
// v = normalize(sum_sl part[sl]) with torch's F.normalize rule x / max(|x|, eps)
__global__ __launch_bounds__(1024) void sn_norm_v_kernel(const float* __restrict__ part, float* __restrict__ v,
                                                         int cols, int nsl, float eps) {
  __shared__ float red[16];
  float ss = 0.f;
  for (int c = threadIdx.x; c < cols; c += 1024) {
    float s = 0.f;
    for (int k = 0; k < nsl; k++) s += part[(long)k * cols + c];
    v[c] = s;
    ss += s * s;
  }
  const float nrm = fmaxf(sqrtf(sn_block_sum(ss, red)), eps);
  for (int c = threadIdx.x; c < cols; c += 1024) v[c] = v[c] / nrm;
}
// t[r] = sum_c W[r][c] v[c]   (one block per row)
__global__ __launch_bounds__(256) void sn_wv_kernel(const float* __restrict__ W, const float* __restrict__ v,
                                                    float* __restrict__ t, int cols) {
  __shared__ float red[4];
  const float* row = W + (long)blockIdx.x * cols;
  float a = 0.f;
  for (int c = threadIdx.x; c < cols; c += 256) a += row[c] * v[c];
  a = sn_block_sum(a, red);
  if (threadIdx.x == 0) t[blockIdx.x] = a;
}
// update: u = normalize(t);  sigma = u . t     (update == 0: sigma = u . t with the stored u -- eval mode)
__global__ __launch_bounds__(1024) void sn_norm_u_kernel(const float* __restrict__ t, float* __restrict__ u,
                                                         float* __restrict__ sigma, int rows, float eps, int update) {
  __shared__ float red[16];
  if (update) {
    float ss = 0.f;
    for (int r = threadIdx.x; r < rows; r += 1024) ss += t[r] * t[r];
    const float nrm = fmaxf(sqrtf(sn_block_sum(ss, red)), eps);
    for (int r = threadIdx.x; r < rows; r += 1024) u[r] = t[r] / nrm;
  }
  float d = 0.f;
  for (int r = threadIdx.x; r < rows; r += 1024) d += u[r] * t[r];
  d = sn_block_sum(d, red);
  if (threadIdx.x == 0) sigma[0] = d;
}

extern "C" int mt_sn_power_iter(const float* W, float* u, float* v, float* sigma, int rows, int cols, int n_iter,
                                float eps, void* ws, size_t ws_bytes, mt_stream_t st) {
  MT_CHECK(rows > 0 && cols > 0 && n_iter >= 0, "mt_sn_power_iter: bad shape %d x %d, %d iterations", rows, cols, n_iter);
  MT_CHECK(ws_bytes >= mt_sn_ws_bytes(rows, cols), "mt_sn_power_iter: workspace of %zu bytes, need %zu", ws_bytes,
           mt_sn_ws_bytes(rows, cols));
  hipStream_t s = (hipStream_t)st;
  const int nsl = sn_slices(rows);
  const int rps = cdiv(rows, nsl);
  float* part = (float*)ws;
  float* t = part + (size_t)nsl * cols;
  for (int it = 0; it < n_iter; it++) {
    hipLaunchKernelGGL(sn_wtu_kernel, dim3(cdiv(cols, 256), nsl), dim3(256), 0, s, W, u, part, rows, cols, rps);
    hipLaunchKernelGGL(sn_norm_v_kernel, dim3(1), dim3(1024), 0, s, part, v, cols, nsl, eps);
    hipLaunchKernelGGL(sn_wv_kernel, dim3(rows), dim3(256), 0, s, W, v, t, cols);
    hipLaunchKernelGGL(sn_norm_u_kernel, dim3(1), dim3(1024), 0, s, t, u, sigma, rows, eps, 1);
  }
  if (n_iter == 0) {
    hipLaunchKernelGGL(sn_wv_kernel, dim3(rows), dim3(256), 0, s, W, v, t, cols);
    hipLaunchKernelGGL(sn_norm_u_kernel, dim3(1), dim3(1024), 0, s, t, u, sigma, rows, eps, 0);
  }
  MT_LAUNCH_CHECK();
  return 0;
}

// ---- weight = W / sigma ------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void sn_scale_kernel(const float* __restrict__ W, const float* __restrict__ sigma,
                                                       float* __restrict__ out, long n) {
  const float sg = sigma[0];
  for (long i = blockIdx.x * 256L + threadIdx.x; i < n; i += gridDim.x * 256L) out[i] = W[i] / sg;
}
extern "C" int mt_sn_scale_fwd(const float* W, const float* sigma, float* Weff, long n, mt_stream_t st) {
  if (n <= 0) return 0;
  const int nb = (int)std::min<long>(cdiv(n, 256), 4096);
  hipLaunchKernelGGL(sn_scale_kernel, dim3(nb), dim3(256), 0, (hipStream_t)st, W, sigma, Weff, n);
  MT_LAUNCH_CHECK();
  return 0;
}

// ---- dW = (G - <G, Weff> u v^T) / sigma ---------------------------------------------------------------------------
__global__ __launch_bounds__(256) void sn_dot_kernel(const float* __restrict__ G, const float* __restrict__ Weff,
                                                     float* __restrict__ part, long n) {
  __shared__ float red[4];
  float a = 0.f;
  for (long i = blockIdx.x * 256L + threadIdx.x; i < n; i += gridDim.x * 256L) a += G[i] * Weff[i];
  a = sn_block_sum(a, red);
  if (threadIdx.x == 0) part[blockIdx.x] = a;
}
__global__ __launch_bounds__(256) void sn_bwd_kernel(const float* __restrict__ G, const float* __restrict__ u,
                                                     const float* __restrict__ v, const float* __restrict__ sigma,
                                                     const float* __restrict__ part, int npart,
                                                     float* __restrict__ dW, int rows, int cols) {
  __shared__ float red[4];
  float d = 0.f;
  for (int i = threadIdx.x; i < npart; i += 256) d += part[i];
  d = sn_block_sum(d, red);
  const float sg = sigma[0];
  const long n = (long)rows * cols;
  for (long i = blockIdx.x * 256L + threadIdx.x; i < n; i += gridDim.x * 256L) {
    const int r = (int)(i / cols), c = (int)(i - (long)r * cols);
    dW[i] = (G[i] - d * u[r] * v[c]) / sg;
  }
}
extern "C" int mt_sn_scale_bwd(const float* G, const float* Weff, const float* u, const float* v, const float* sigma,
                               float* dW, int rows, int cols, void* ws, size_t ws_bytes, mt_stream_t st) {
  MT_CHECK(rows > 0 && cols > 0, "mt_sn_scale_bwd: bad shape %d x %d", rows, cols);
  MT_CHECK(ws_bytes >= mt_sn_ws_bytes(rows, cols), "mt_sn_scale_bwd: workspace of %zu bytes, need %zu", ws_bytes,
           mt_sn_ws_bytes(rows, cols));
  hipStream_t s = (hipStream_t)st;
  const long n = (long)rows * cols;
  const int nd = (int)std::min<long>(cdiv(n, 1024), SN_DOT_BLOCKS);
  float* part = (float*)ws + (size_t)sn_slices(rows) * cols + rows;
  hipLaunchKernelGGL(sn_dot_kernel, dim3(nd), dim3(256), 0, s, G, Weff, part, n);
  const int nb = (int)std::min<long>(cdiv(n, 256), 4096);
  hipLaunchKernelGGL(sn_bwd_kernel, dim3(nb), dim3(256), 0, s, G, u, v, sigma, part, nd, dW, rows, cols);
  MT_LAUNCH_CHECK();
  return 0;
}
