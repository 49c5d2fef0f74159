// Small fp32 kernels: nn.Linear on style codes (latency-bound, M <= 64 rows), the fused
// multi-tensor Adam step, and the library's error plumbing.
// Reference: nn.Linear (networks.py:127-128,256-261; norm.py:27), torch.optim.Adam with
// L2-coupled weight decay (adain_model.py:57-61,68-71).
#include "mt_common.h"
#include <string.h>

// ---- error plumbing -----------------------------------------------------------------------
static thread_local char g_err[512] = "";
void mt_set_error(const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
}
extern "C" const char* mt_last_error(void) { return g_err; }
extern "C" int mt_version(void) { return 1; }

// ---- Linear: y[n][o] = act(sum_i x[n][i] * w[o][i] + b[o]) --------------------------------
// one wave per output element; lanes stride the reduction and shuffle-reduce.
__global__ void linear_fwd_kernel(const float* __restrict__ x, const float* __restrict__ w,
                                  const float* __restrict__ b, float* __restrict__ y, int n, int in, int out,
                                  int act) {
  const int wave = (blockIdx.x * blockDim.x + threadIdx.x) >> 6;
  const int lane = threadIdx.x & 63;
  if (wave >= n * out) return;
  const int r = wave / out, o = wave % out;
  float a = 0.f;
  for (int i = lane; i < in; i += 64) a += x[(long)r * in + i] * w[(long)o * in + i];
  a = wave_sum(a);
  if (lane == 0) {
    if (b) a += b[o];
    y[wave] = act_apply(a, act, 0.01f);
  }
}
extern "C" int mt_linear_fwd(const float* x, const float* w, const float* b, float* y, int n, int in, int out,
                             int act, mt_stream_t s) {
  const long waves = (long)n * out;
  if (waves == 0) return 0;
  hipLaunchKernelGGL(linear_fwd_kernel, dim3(cdiv(waves, 4)), dim3(256), 0, (hipStream_t)s, x, w, b, y, n, in, out, act);
  MT_LAUNCH_CHECK();
  return 0;
}
// dx[n][i] = sum_o dy[n][o] w[o][i];  dw[o][i] = sum_n dy[n][o] x[n][i];  db[o] = sum_n dy[n][o]
__global__ __launch_bounds__(256) void linear_bwd_dx_kernel(const float* __restrict__ w, const float* __restrict__ dy,
                                                            float* __restrict__ dx, int n, int in, int out) {
  // block = 64 consecutive input features of one row x 4 slices of the output dimension
  __shared__ float red[256];
  const int r = blockIdx.y;
  const int i = blockIdx.x * 64 + (threadIdx.x & 63);
  const int sl = threadIdx.x >> 6;
  float a = 0.f;
  if (i < in) {
#pragma unroll 8
    for (int o = sl; o < out; o += 4) a += dy[(long)r * out + o] * w[(long)o * in + i];
  }
  red[threadIdx.x] = a;
  __syncthreads();
  if (sl == 0 && i < in) dx[(long)r * in + i] = red[threadIdx.x] + red[threadIdx.x + 64] + red[threadIdx.x + 128] + red[threadIdx.x + 192];
}
__global__ void linear_bwd_dw_kernel(const float* __restrict__ x, const float* __restrict__ dy,
                                     float* __restrict__ dw, float* __restrict__ db, int n, int in, int out,
                                     int accumulate) {
  const long idx = blockIdx.x * (long)blockDim.x + threadIdx.x;
  if (idx >= (long)out * in) return;
  const int o = (int)(idx / in), i = (int)(idx % in);
  float a = 0.f, bsum = 0.f;
  for (int r = 0; r < n; r++) {
    const float g = dy[(long)r * out + o];
    a += g * x[(long)r * in + i];
    bsum += g;
  }
  dw[idx] = accumulate ? dw[idx] + a : a;
  if (db && i == 0) db[o] = accumulate ? db[o] + bsum : bsum;
}
extern "C" int mt_linear_bwd(const float* x, const float* w, const float* dy, float* dx, float* dw, float* db,
                             int n, int in, int out, int accumulate, mt_stream_t st) {
  hipStream_t s = (hipStream_t)st;
  if (n == 0) return 0;
  if (dx) hipLaunchKernelGGL(linear_bwd_dx_kernel, dim3(cdiv(in, 64), n), dim3(256), 0, s, w, dy, dx, n, in, out);
  if (dw) hipLaunchKernelGGL(linear_bwd_dw_kernel, dim3(cdiv((long)out * in, 256)), dim3(256), 0, s, x, dy, dw, db, n, in, out, accumulate);
  MT_LAUNCH_CHECK();
  return 0;
}

// ---- grouped Linear: G layers that share their input (the four AdaIN projections of a decoder, norm.py:27 /
// blocks.py:152-164: four launches forward, eight plus three gradient sums backward) in ONE launch each way -------------
#define MT_LINEAR_MAX_GROUPS 8
struct LinearGroupArgs {
  const float* w[MT_LINEAR_MAX_GROUPS];
  const float* b[MT_LINEAR_MAX_GROUPS];
  const float* dy[MT_LINEAR_MAX_GROUPS];
  float* y[MT_LINEAR_MAX_GROUPS];
  float* dw[MT_LINEAR_MAX_GROUPS];
  float* db[MT_LINEAR_MAX_GROUPS];
  int groups;
};
__global__ void linear_group_fwd_kernel(const float* __restrict__ x, LinearGroupArgs a, int n, int in, int out) {
  const int g = blockIdx.y;
  const int wave = (blockIdx.x * blockDim.x + threadIdx.x) >> 6;
  const int lane = threadIdx.x & 63;
  if (wave >= n * out) return;
  const int r = wave / out, o = wave % out;
  const float* __restrict__ w = a.w[g];
  float acc = 0.f;
  for (int i = lane; i < in; i += 64) acc += x[(long)r * in + i] * w[(long)o * in + i];
  acc = wave_sum(acc);
  if (lane == 0) a.y[g][wave] = acc + (a.b[g] ? a.b[g][o] : 0.f);
}
// dx[n][i] = sum_g sum_o dy_g[n][o] w_g[o][i]   (groups added in index order)
__global__ __launch_bounds__(256) void linear_group_bwd_dx_kernel(LinearGroupArgs a, float* __restrict__ dx, int n, int in,
                                                                  int out) {
  __shared__ float red[256];
  const int r = blockIdx.y;
  const int i = blockIdx.x * 64 + (threadIdx.x & 63);
  const int sl = threadIdx.x >> 6;
  float acc = 0.f;
  if (i < in) {
    for (int g = 0; g < a.groups; g++) {
      const float* __restrict__ w = a.w[g];
      const float* __restrict__ dy = a.dy[g];
#pragma unroll 8
      for (int o = sl; o < out; o += 4) acc += dy[(long)r * out + o] * w[(long)o * in + i];
    }
  }
  red[threadIdx.x] = acc;
  __syncthreads();
  if (sl == 0 && i < in) dx[(long)r * in + i] = red[threadIdx.x] + red[threadIdx.x + 64] + red[threadIdx.x + 128] + red[threadIdx.x + 192];
}
__global__ void linear_group_bwd_dw_kernel(const float* __restrict__ x, LinearGroupArgs a, int n, int in, int out,
                                           int accumulate) {
  const int g = blockIdx.y;
  const long idx = blockIdx.x * (long)blockDim.x + threadIdx.x;
  if (idx >= (long)out * in) return;
  const int o = (int)(idx / in), i = (int)(idx % in);
  const float* __restrict__ dy = a.dy[g];
  float acc = 0.f, bsum = 0.f;
  for (int r = 0; r < n; r++) {
    const float gv = dy[(long)r * out + o];
    acc += gv * x[(long)r * in + i];
    bsum += gv;
  }
  float* dw = a.dw[g];
  float* db = a.db[g];
  if (dw) dw[idx] = accumulate ? dw[idx] + acc : acc;
  if (db && i == 0) db[o] = accumulate ? db[o] + bsum : bsum;
}
extern "C" int mt_linear_group_fwd(const float* x, const float* const* w, const float* const* b, float* const* y,
                                   int groups, int n, int in, int out, mt_stream_t s) {
  MT_CHECK(groups >= 1 && groups <= MT_LINEAR_MAX_GROUPS, "linear_group: %d groups", groups);
  if ((long)n * out == 0) return 0;
  LinearGroupArgs a;
  memset(&a, 0, sizeof(a));
  a.groups = groups;
  for (int g = 0; g < groups; g++) {
    MT_CHECK(w[g] != nullptr && y[g] != nullptr, "linear_group_fwd: null weight / output %d", g);
    a.w[g] = w[g]; a.b[g] = b ? b[g] : nullptr; a.y[g] = y[g];
  }
  hipLaunchKernelGGL(linear_group_fwd_kernel, dim3(cdiv((long)n * out, 4), groups), dim3(256), 0, (hipStream_t)s, x, a, n, in, out);
  MT_LAUNCH_CHECK();
  return 0;
}
extern "C" int mt_linear_group_bwd(const float* x, const float* const* w, const float* const* dy, float* dx,
                                   float* const* dw, float* const* db, int groups, int n, int in, int out,
                                   int accumulate, mt_stream_t st) {
  hipStream_t s = (hipStream_t)st;
  MT_CHECK(groups >= 1 && groups <= MT_LINEAR_MAX_GROUPS, "linear_group: %d groups", groups);
  if (n == 0) return 0;
  LinearGroupArgs a;
  memset(&a, 0, sizeof(a));
  a.groups = groups;
  bool any_dw = false;
  for (int g = 0; g < groups; g++) {
    MT_CHECK(w[g] != nullptr && dy[g] != nullptr, "linear_group_bwd: null weight / gradient %d", g);
    a.w[g] = w[g]; a.dy[g] = dy[g];
    a.dw[g] = dw ? dw[g] : nullptr; a.db[g] = db ? db[g] : nullptr;
    any_dw = any_dw || a.dw[g] || a.db[g];
  }
  if (dx) hipLaunchKernelGGL(linear_group_bwd_dx_kernel, dim3(cdiv(in, 64), n), dim3(256), 0, s, a, dx, n, in, out);
  if (any_dw) hipLaunchKernelGGL(linear_group_bwd_dw_kernel, dim3(cdiv((long)out * in, 256), groups), dim3(256), 0, s, x, a, n, in, out, accumulate);
  MT_LAUNCH_CHECK();
  return 0;
}

// ---- fused multi-tensor Adam -----------------------------------------------------------------
// grid = (chunks, tensors).  torch.optim.Adam semantics (no amsgrad, maximize=False):
//   g += wd*p;  m = b1*m + (1-b1)*g;  v = b2*v + (1-b2)*g*g;
//   p -= (lr/(1-b1^t)) * m / (sqrt(v)/sqrt(1-b2^t) + eps)
struct AdamDevState { float lr; int step; float bc1; float bc2_sqrt; };   // mirror of the float[4] / int[4] buffer
__global__ void adam_multi_kernel(void* const* __restrict__ ptrs, const int64_t* __restrict__ sizes, float lr,
                                  float b1, float b2, float eps, float wd, float bc1, float bc2_sqrt,
                                  const AdamDevState* __restrict__ dev, int zero_grads) {
  if (dev) { lr = dev->lr; bc1 = dev->bc1; bc2_sqrt = dev->bc2_sqrt; }
  const int t = blockIdx.y;
  float* __restrict__ p = (float*)ptrs[4 * t + 0];
  float* __restrict__ g = (float*)ptrs[4 * t + 1];
  float* __restrict__ m = (float*)ptrs[4 * t + 2];
  float* __restrict__ v = (float*)ptrs[4 * t + 3];
  const long n = sizes[t];
  const float step_size = lr / bc1;
  for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
    const float pi = p[i];
    const float gi = g[i] + wd * pi;
    const float mi = b1 * m[i] + (1.f - b1) * gi;
    const float vi = b2 * v[i] + (1.f - b2) * gi * gi;
    m[i] = mi;
    v[i] = vi;
    p[i] = pi - step_size * mi / (sqrtf(vi) / bc2_sqrt + eps);
    if (zero_grads) g[i] = 0.f;
  }
}
extern "C" int mt_adam_multi(void* const* ptrs, const int64_t* sizes, int count, int64_t max_size, float lr,
                             float beta1, float beta2, float eps, float wd, int step, mt_stream_t s) {
  if (count == 0) return 0;
  MT_CHECK(step >= 1, "adam: step must be >= 1");
  const float bc1 = (float)(1.0 - pow((double)beta1, (double)step));
  const float bc2 = (float)sqrt(1.0 - pow((double)beta2, (double)step));
  int chunks = (int)((max_size + 256 * 8 - 1) / (256 * 8));
  if (chunks < 1) chunks = 1;
  if (chunks > 2048) chunks = 2048;
  hipLaunchKernelGGL(adam_multi_kernel, dim3(chunks, count), dim3(256), 0, (hipStream_t)s, ptrs, sizes, lr, beta1, beta2, eps, wd, bc1, bc2, (const AdamDevState*)nullptr, 0);
  MT_LAUNCH_CHECK();
  return 0;
}
// The same update with learning rate and step count read from a 16-byte device record {float lr; int32 step; float
// 1 - beta1^step; float sqrt(1 - beta2^step)}: no host-side value changes from step to step, so the launch pair can be
// captured into a hipGraph.  One tick (step += 1, bias corrections in double precision) precedes the update.
__global__ void adam_tick_kernel(AdamDevState* st, float b1, float b2) {
  const int t = st->step + 1;
  st->step = t;
  st->bc1 = (float)(1.0 - pow((double)b1, (double)t));
  st->bc2_sqrt = (float)sqrt(1.0 - pow((double)b2, (double)t));
}
extern "C" int mt_adam_multi_dev(void* const* ptrs, const int64_t* sizes, int count, int64_t max_size, float beta1,
                                 float beta2, float eps, float wd, void* dev_state, int zero_grads, mt_stream_t s) {
  if (count == 0) return 0;
  MT_CHECK(dev_state != nullptr, "adam_multi_dev: null state");
  hipLaunchKernelGGL(adam_tick_kernel, dim3(1), dim3(1), 0, (hipStream_t)s, (AdamDevState*)dev_state, beta1, beta2);
  int chunks = (int)((max_size + 256 * 8 - 1) / (256 * 8));
  if (chunks < 1) chunks = 1;
  if (chunks > 2048) chunks = 2048;
  hipLaunchKernelGGL(adam_multi_kernel, dim3(chunks, count), dim3(256), 0, (hipStream_t)s, ptrs, sizes, 0.f, beta1, beta2, eps, wd, 1.f, 1.f, (const AdamDevState*)dev_state, zero_grads);
  MT_LAUNCH_CHECK();
  return 0;
}
