// Direct 7x7 stem convolution, forward (bf16): Conv2d(3 -> 64, k7, s1, pad 3) of the content encoder
// (reference networks.py:30, blocks.py:29-35).  gfx950 only.
//
// Why not the gather-GEMM: with Cin = 3 padded to 8 the implicit GEMM gathers 49 sixteen-byte pieces per output pixel
// -- 822 MB of LDS-DMA for a 6 MB image at N = 16, 256x256 (157 us against a 27 us HBM floor for the 134 MB output) --
// and only 3 of every 8 MFMA K lanes carry data.  Here a workgroup keeps the 22x22 input patch of a 16x16 output tile
// in LDS (4 channels = 8 bytes per pixel: every input pixel is read ONCE per tile) and orders K as (filter row,
// [filter column, channel]): for output column c and filter row dh the 32-deep MFMA k-step is the contiguous 64-byte run
// of patch row (r + dh) starting at pixel c -- columns c..c+7 x 4 channels, of which the eighth column and the fourth
// channel meet zero weights: 147 of 224 K lanes are useful, and a pixel fragment is two 8-byte LDS reads.
//   LDS: weights [7 filter rows][64 couts][32 k] (80-byte rows: conflict-free 16-byte fragment reads), built once per
//        workgroup from the generic forward pack [cout][tap][8]; two patch buffers; statistics scratch.
//   Persistent over tiles (x fastest, so neighbours share halo columns in L2); the next tile's patch is fetched into
//   registers while the current one is multiplied.
//   Epilogue as in the gather-GEMM kernels: permuted weight rows -> 8 consecutive channels per lane -> 16-byte stores;
//   optional InstanceNorm statistics (sum, sum of squares per (image, channel)) reduced per tile and added with one
//   atomic instruction per 64 (channel, moment) pairs.
#include "conv_device.h"
#include <stdlib.h>

struct StemParams {
  const char* x;        // NHWC [N][H][W][8] bf16
  const char* wpack;    // forward pack [64][49][8] bf16
  const float* bias;    // optional fp32 [nbias]
  char* y;              // NHWC [N][H][W][64] bf16
  float* stats;         // optional fp32 [N][64][2], accumulated
  int N, H, W;
  int nbias;
  int pad_mode;
  float neg_slope;      // epilogue: z > 0 ? z : z * neg_slope  (1 = none, 0 = ReLU, slope = LeakyReLU)
  int tiles_x, tiles_y, total;
  unsigned x_bytes, y_bytes;
};

constexpr int ST_TH = 16, ST_TW = 16;            // output tile
constexpr int ST_PH = ST_TH + 6, ST_PW = 24;     // patch rows; patch row pitch in pixels (22 needed + 1 over-read, padded)
constexpr int ST_WROW = 80;                      // bytes per weight row (64 + 16 of padding)

__global__ __launch_bounds__(256, 3) void stem_fwd_kernel(const StemParams p) {
  constexpr unsigned OOB = 0x80000000u;
  __shared__ u32x4 smem[(7 * 64 * ST_WROW + 2 * ST_PH * ST_PW * 8 + 4 * 64 * 2 * 4) / 16];
  char* const sWt = reinterpret_cast<char*>(smem);                                  // [7][64][80 B]
  char* const sPatch = sWt + 7 * 64 * ST_WROW;                                      // [2][22][24 px][8 B]
  float* const red = reinterpret_cast<float*>(sPatch + 2 * ST_PH * ST_PW * 8);      // [4 waves][64][2]

  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
  const int fr = lane & 15, fg = lane >> 4;

  // ---- weights: pack row co, tap (dh, dw), channels 0..3 -> sWt[dh][perm(co)][dw*4 .. +3]; column 7 stays zero ----
  for (int i = tid; i < 7 * 64 * (ST_WROW / 16); i += 256) reinterpret_cast<u32x4*>(sWt)[i] = u32x4{0u, 0u, 0u, 0u};
  __syncthreads();
  for (int i = tid; i < 64 * 49; i += 256) {
    const int co = i / 49, tap = i - co * 49;
    const int dh = tap / 7, dw = tap - dh * 7;
    const u32x2 v = *reinterpret_cast<const u32x2*>(p.wpack + ((size_t)co * 49 + tap) * 16);
    // 16-byte epilogue stores: within each 32-row group LDS row (a&1)*16 + r holds channel (r>>2)*8 + (a&1)*4 + (r&3)
    const int c32 = co & 31;
    const int row = (co & ~31) + ((c32 >> 2) & 1) * 16 + (c32 >> 3) * 4 + (c32 & 3);
    *reinterpret_cast<u32x2*>(sWt + (dh * 64 + row) * ST_WROW + dw * 8) = v;
  }

  const __amdgpu_buffer_rsrc_t rsx = __builtin_amdgcn_make_buffer_rsrc((void*)p.x, 0, p.x_bytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t rsy = __builtin_amdgcn_make_buffer_rsrc((void*)p.y, 0, p.y_bytes, 0x00020000);

  // patch pixels of this thread: idx = tid, tid + 256 (< 22 * 23)
  auto patch_offsets = [&](int tile, unsigned* off) {
    const int tx = tile % p.tiles_x;
    const int t2 = tile / p.tiles_x;
    const int ty = t2 % p.tiles_y, n = t2 / p.tiles_y;
#pragma unroll
    for (int j = 0; j < 2; j++) {
      const int idx = tid + 256 * j;
      const int pr = idx / 23, pc = idx - pr * 23;
      int r = ty * ST_TH + pr - 3, c = tx * ST_TW + pc - 3;
      bool ok = idx < ST_PH * 23;
      if (p.pad_mode == MT_PAD_REFLECT) {
        r = r < 0 ? -r : r;
        r = r >= p.H ? 2 * (p.H - 1) - r : r;
        c = c < 0 ? -c : c;
        c = c >= p.W ? 2 * (p.W - 1) - c : c;
      }
      ok = ok && (unsigned)r < (unsigned)p.H && (unsigned)c < (unsigned)p.W;   // (zero padding / beyond a ragged tile: zeros)
      off[j] = ok ? (unsigned)((n * p.H + r) * p.W + c) * 16u : OOB;
    }
  };
  auto patch_store = [&](int buf, const u32x2* v) {
#pragma unroll
    for (int j = 0; j < 2; j++) {
      const int idx = tid + 256 * j;
      if (idx < ST_PH * 23) {
        const int pr = idx / 23, pc = idx - pr * 23;
        *reinterpret_cast<u32x2*>(sPatch + ((buf * ST_PH + pr) * ST_PW + pc) * 8) = v[j];
      }
    }
  };

  // persistent schedule: contiguous tiles per XCD, consecutive tiles on consecutive workgroups
  const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3;
  const int nslot = ((int)gridDim.x - xcd + 7) >> 3;
  const int tq = p.total >> 3, tr = p.total & 7;
  const int lo = xcd * tq + (xcd < tr ? xcd : tr);
  const int cnt = tq + (xcd < tr ? 1 : 0);
  int it = slot;
  if (it >= cnt) return;

  unsigned off[2];
  u32x2 pv[2];
  patch_offsets(lo + it, off);
#pragma unroll
  for (int j = 0; j < 2; j++) pv[j] = __builtin_bit_cast(u32x2, __builtin_amdgcn_raw_buffer_load_b64(rsx, off[j], 0, 0));
  patch_store(0, pv);
  __syncthreads();

  float bv[2][8];
#pragma unroll
  for (int sp = 0; sp < 2; sp++)
#pragma unroll
    for (int e = 0; e < 8; e++) {
      const int co = sp * 32 + fg * 8 + e;
      bv[sp][e] = (p.bias != nullptr && co < p.nbias) ? p.bias[co] : 0.f;
    }

  int buf = 0;
  while (true) {
    const int tile = lo + it;
    const int it_next = it + nslot;
    const bool has_next = it_next < cnt;
    if (has_next) {
      patch_offsets(lo + it_next, off);
#pragma unroll
      for (int j = 0; j < 2; j++) pv[j] = __builtin_bit_cast(u32x2, __builtin_amdgcn_raw_buffer_load_b64(rsx, off[j], 0, 0));
    }
    // ---- 7 k-steps (filter rows) x 4 cout fragments x 4 pixel fragments (tile rows 4 wv .. 4 wv + 3, column fr) ----
    f32x4 acc[4][4];
#pragma unroll
    for (int a = 0; a < 4; a++)
#pragma unroll
      for (int b = 0; b < 4; b++) acc[a][b] = f32x4{0.f, 0.f, 0.f, 0.f};
    const char* pb = sPatch + buf * ST_PH * ST_PW * 8;
#pragma unroll
    for (int dh = 0; dh < 7; dh++) {
      u32x4 wf[4], xf[4];
#pragma unroll
      for (int a = 0; a < 4; a++)
        wf[a] = *reinterpret_cast<const u32x4*>(sWt + (dh * 64 + a * 16 + fr) * ST_WROW + fg * 16);
#pragma unroll
      for (int b = 0; b < 4; b++) {
        const char* q = pb + ((4 * wv + b + dh) * ST_PW + fr + 2 * fg) * 8;
        const u32x2 lo2 = *reinterpret_cast<const u32x2*>(q), hi2 = *reinterpret_cast<const u32x2*>(q + 8);
        xf[b] = u32x4{lo2[0], lo2[1], hi2[0], hi2[1]};
      }
#pragma unroll
      for (int a = 0; a < 4; a++)
#pragma unroll
        for (int b = 0; b < 4; b++) mma_chunk<true>(acc[a][b], wf[a], xf[b]);
    }
    // ---- epilogue ----
    const int tx = tile % p.tiles_x;
    const int t2 = tile / p.tiles_x;
    const int ty = t2 % p.tiles_y, n = t2 / p.tiles_y;
    const int ox = tx * ST_TW + fr;
    float s1[2][8], s2[2][8];
#pragma unroll
    for (int sp = 0; sp < 2; sp++)
#pragma unroll
      for (int e = 0; e < 8; e++) s1[sp][e] = s2[sp][e] = 0.f;
#pragma unroll
    for (int b = 0; b < 4; b++) {
      const int oy = ty * ST_TH + 4 * wv + b;
      const bool ok = oy < p.H && ox < p.W;
      const unsigned yo = ok ? (unsigned)((n * p.H + oy) * p.W + ox) * 128u : OOB;
#pragma unroll
      for (int sp = 0; sp < 2; sp++) {
        float v[8];
#pragma unroll
        for (int e = 0; e < 8; e++) {
          const float z = acc[2 * sp + (e >> 2)][b][e & 3] + bv[sp][e];
          v[e] = z > 0.f ? z : z * p.neg_slope;
          const float vm = ok ? v[e] : 0.f;
          s1[sp][e] += vm;
          s2[sp][e] += vm * vm;
        }
        const u32x4 o = {pack2_bf16(v[0], v[1]), pack2_bf16(v[2], v[3]), pack2_bf16(v[4], v[5]), pack2_bf16(v[6], v[7])};
        __builtin_amdgcn_raw_buffer_store_b128(o, rsy, ok ? yo + (unsigned)(sp * 32 + fg * 8) * 2u : OOB, 0, 2);
      }
    }
    if (p.stats != nullptr) {
#pragma unroll
      for (int sp = 0; sp < 2; sp++)
#pragma unroll
        for (int e = 0; e < 8; e++) {
          const float t1 = row16_sum(s1[sp][e]), t2s = row16_sum(s2[sp][e]);
          if (fr == 0) {
            red[(wv * 64 + sp * 32 + fg * 8 + e) * 2] = t1;
            red[(wv * 64 + sp * 32 + fg * 8 + e) * 2 + 1] = t2s;
          }
        }
      __syncthreads();
      if (tid < 128) {
        float t = 0.f;
#pragma unroll
        for (int w = 0; w < 4; w++) t += red[w * 128 + tid];
        atomicAdd(p.stats + (size_t)n * 128 + tid, t);
      }
    }
    if (!has_next) break;
    // the next tile's patch (in registers since the top of this iteration) -> the other buffer
    patch_store(buf ^ 1, pv);
    __syncthreads();      // (also: every wave is done with `red` and with patch buffer `buf`)
    buf ^= 1;
    it = it_next;
  }
}

static int g_stem_on = -1;
static long g_stem_launches = 0;
long mt_stem_launches() { return g_stem_launches; }
int mt_stem_enable(int on) {
  if (g_stem_on < 0) g_stem_on = getenv("MT_STEM_DIRECT") ? (atoi(getenv("MT_STEM_DIRECT")) != 0) : 1;
  const int prev = g_stem_on;
  if (on >= 0) g_stem_on = on != 0;
  return prev;
}

// -> 0 launched, 2 launch error, -1 not this kernel's shape (the caller takes the gather-GEMM)
int mt_launch_stem_fwd(const mt_conv_desc* d, const void* x, const void* pack, const float* bias, void* y, float* stats,
                       hipStream_t s) {
  mt_stem_enable(-1);
  if (!g_stem_on) return -1;
  if (d->dtype != MT_BF16 || d->transposed || d->kh != 7 || d->kw != 7 || d->stride != 1 || d->pad != 3 || d->Ci > 4 ||
      d->Co != 64 || d->act == MT_ACT_TANH || d->H < 8 || d->W < 8)
    return -1;
  const unsigned long long xb = (unsigned long long)d->N * d->H * d->W * 16ull, yb = (unsigned long long)d->N * d->H * d->W * 128ull;
  if (xb >= 0x7f000000ull || yb >= 0x7f000000ull) return -1;
  StemParams p;
  p.x = (const char*)x; p.wpack = (const char*)pack; p.bias = bias; p.nbias = bias ? d->Co : 0;
  p.y = (char*)y; p.stats = stats;
  p.N = d->N; p.H = d->H; p.W = d->W; p.pad_mode = d->pad_mode;
  p.neg_slope = d->act == MT_ACT_RELU ? 0.f : (d->act == MT_ACT_LRELU ? d->slope : 1.f);
  p.tiles_x = cdiv(d->W, ST_TW); p.tiles_y = cdiv(d->H, ST_TH);
  p.total = d->N * p.tiles_x * p.tiles_y;
  p.x_bytes = (unsigned)xb; p.y_bytes = (unsigned)yb;
  if (p.total <= 0) return 0;
  const int nb = p.total < 768 ? p.total : 768;
  hipLaunchKernelGGL(stem_fwd_kernel, dim3(nb), dim3(256), 0, s, p);
  MT_LAUNCH_CHECK();
  __atomic_fetch_add(&g_stem_launches, 1, __ATOMIC_RELAXED);
  return 0;
}
