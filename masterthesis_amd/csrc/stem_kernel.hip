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
  static const int cap = getenv("MT_STEM_FWD_BLOCKS") ? atoi(getenv("MT_STEM_FWD_BLOCKS")) : 768;
  const int nb = p.total < cap ? p.total : cap;
  hipLaunchKernelGGL(stem_fwd_kernel, dim3(nb), dim3(256), 0, s, p);
  MT_LAUNCH_CHECK();
  __atomic_fetch_add(&g_stem_launches, 1, __ATOMIC_RELAXED);
  return 0;
}

// ------------------------------------------------------------------------------------------------------------------
// Direct 7x7 stem weight gradient (bf16):  dW[co][ch][dh][dw] = sum_{n,y,x} dy[n,y,x,co] * xpad[n, y+dh-3, x+dw-3, ch].
// The gather form streams 49 sixteen-byte pieces of x per pixel (205 us at N = 16, 256x256 for 140 MB of operands);
// here a workgroup walks 8x32-pixel tiles, keeps the tile of dy ([256 px][64 co], 32 KiB) and the 14x39 input patch
// (8 bytes per pixel) in LDS and reduces over pixels with the MFMA K dimension = 32 consecutive pixels of a tile row:
//   D[j][co] += sum_px P[px][j] * dy[px][co],   j = (dh, [dw, ch]) -- per filter row dh a 32-wide run of the patch
// Both operands are pixel-major, so the fragments come from the transposing LDS read ds_read_b64_tr_b16; for the patch
// the "matrix row" of pixel x is the 64 bytes starting at that pixel (rows overlap: pitch 8 bytes).
// The 7 x 2 (filter row, run half) output fragments are split over the 4 waves (4, 4, 3, 3), each with all 4 cout
// fragments.  Every workgroup accumulates ONE partial dW over all its tiles and writes it as a slab
// [64 co][224 j] fp32; stem_wgrad_reduce_kernel adds the slabs in index order (reproducible) into the reference layout.
constexpr int SW_TH = 8, SW_TW = 32;                   // pixel tile
constexpr int SW_PH = SW_TH + 6, SW_PP = 40;           // patch rows, patch row pitch in pixels (39 read)
constexpr int SW_SLAB = 64 * 224;                      // floats per slab

struct StemWgradParams {
  const char* x;        // NHWC [N][H][W][8] bf16
  const char* dy;       // NHWC [N][H][W][64] bf16
  float* slabs;         // [gridDim.x][64][224]
  int N, H, W;
  int pad_mode;
  int tiles_x, tiles_y, total;
  unsigned x_bytes, dy_bytes;
};

__global__ __launch_bounds__(256, 2) void stem_wgrad_kernel(const StemWgradParams p) {
  constexpr unsigned OOB = 0x80000000u;
  // ONE shared array (LDS-DMA target): 2 dy tiles, then 2 patches
  __shared__ u32x4 smem[(2 * 256 * 128 + 2 * SW_PH * SW_PP * 8) / 16];
  char* const sDy = reinterpret_cast<char*>(smem);                       // [2][256 px][128 B], 32-byte slots XOR (row>>3)&3
  char* const sP = sDy + 2 * 256 * 128;                                   // [2][14][40 px][8 B]
  typedef __attribute__((address_space(3))) void* lds_ptr;
  typedef __attribute__((address_space(3))) s16x4* lds_s4;

  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
  const int wvu = __builtin_amdgcn_readfirstlane(wv);
  const int fr = lane & 15, fg = lane >> 4;
  const int g = lane >> 4, qq = (lane & 15) >> 2, pp = lane & 3;

  const __amdgpu_buffer_rsrc_t rsx = __builtin_amdgcn_make_buffer_rsrc((void*)p.x, 0, p.x_bytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t rsd = __builtin_amdgcn_make_buffer_rsrc((void*)p.dy, 0, p.dy_bytes, 0x00020000);

  auto tile_coords = [&](int tile, int& n, int& y0, int& x0) {
    const int tx = tile % p.tiles_x;
    const int t2 = tile / p.tiles_x;
    n = t2 / p.tiles_y;
    y0 = (t2 - n * p.tiles_y) * SW_TH;
    x0 = tx * SW_TW;
  };
  // dy tile -> LDS buffer `buf` by LDS-DMA: wave w, instruction i covers tile rows (pixels) 8*(w + 4 i) .. +7, a lane
  // one 16-byte chunk: LDS position (row, pos) receives logical chunk ((pos>>1) ^ key) << 1 | (pos & 1), key = (row>>3)&3
  auto issue_dy = [&](int buf, int n, int y0, int x0) {
#pragma unroll
    for (int i = 0; i < 8; i++) {
      const int blk = wvu + 4 * i;                    // 8-pixel block of the tile: pixels 8 blk .. 8 blk + 7
      const int px = blk * 8 + (lane >> 3);           // tile pixel = ty * 32 + tx
      const int pos = lane & 7;
      const int key = blk & 3;
      const int chunk = (((pos >> 1) ^ key) << 1) | (pos & 1);
      const int y = y0 + (px >> 5), xx = x0 + (px & 31);
      const bool ok = y < p.H && xx < p.W;
      const unsigned o = ok ? (unsigned)((n * p.H + y) * p.W + xx) * 128u + (unsigned)chunk * 16u : OOB;
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rsd, (lds_ptr)(sDy + buf * 256 * 128 + blk * 1024), 16, o, 0, 0, 0);
    }
  };
  // patch pixels of this thread: idx = tid + 256 j (< 14 * 39)
  auto patch_load = [&](int n, int y0, int x0, u32x2* v) {
#pragma unroll
    for (int j = 0; j < 3; j++) {
      const int idx = tid + 256 * j;
      const int pr = idx / 39, pc = idx - pr * 39;
      int r = y0 + pr - 3, c = x0 + pc - 3;
      bool ok = idx < SW_PH * 39;
      if (p.pad_mode == MT_PAD_REFLECT) {
        r = r < 0 ? -r : r;
        r = r >= p.H ? 2 * (p.H - 1) - r : r;
        c = c < 0 ? -c : c;
        c = c >= p.W ? 2 * (p.W - 1) - c : c;
      }
      ok = ok && (unsigned)r < (unsigned)p.H && (unsigned)c < (unsigned)p.W;
      const unsigned o = ok ? (unsigned)((n * p.H + r) * p.W + c) * 16u : OOB;
      v[j] = __builtin_bit_cast(u32x2, __builtin_amdgcn_raw_buffer_load_b64(rsx, o, 0, 0));
    }
  };
  auto patch_store = [&](int buf, const u32x2* v) {
#pragma unroll
    for (int j = 0; j < 3; j++) {
      const int idx = tid + 256 * j;
      if (idx < SW_PH * 39) {
        const int pr = idx / 39, pc = idx - pr * 39;
        *reinterpret_cast<u32x2*>(sP + ((buf * SW_PH + pr) * SW_PP + pc) * 8) = v[j];
      }
    }
  };

  // persistent schedule (as in the forward kernel)
  const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3;
  const int nslot = ((int)gridDim.x - xcd + 7) >> 3;
  const int tq = p.total >> 3, tr = p.total & 7;
  const int lo = xcd * tq + (xcd < tr ? xcd : tr);
  const int cnt = tq + (xcd < tr ? 1 : 0);
  int it = slot;

  f32x4 acc[4][4];        // [cout fragment][this wave's (dh, half) fragments q = wv, wv+4, wv+8, wv+12]
#pragma unroll
  for (int a = 0; a < 4; a++)
#pragma unroll
    for (int s = 0; s < 4; s++) acc[a][s] = f32x4{0.f, 0.f, 0.f, 0.f};

  if (it < cnt) {
    int n, y0, x0;
    tile_coords(lo + it, n, y0, x0);
    u32x2 pv[3];
    issue_dy(0, n, y0, x0);
    patch_load(n, y0, x0, pv);
    patch_store(0, pv);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    int buf = 0;
    while (true) {
      const int it_next = it + nslot;
      const bool has_next = it_next < cnt;
      if (has_next) {
        tile_coords(lo + it_next, n, y0, x0);
        issue_dy(buf ^ 1, n, y0, x0);
        patch_load(n, y0, x0, pv);
      }
      const char* dyb = sDy + buf * 256 * 128;
      const char* pb = sP + buf * SW_PH * SW_PP * 8;
#pragma unroll
      for (int r = 0; r < SW_TH; r++) {               // k-step = the 32 pixels of tile row r
        bf16x8 af[4], bf[4];
        const int prow = r * 32 + 8 * g + qq;         // (key of the dy swizzle = (prow >> 3) & 3 = g)
#pragma unroll
        for (int a = 0; a < 4; a++) {
          const char* pa = dyb + prow * 128 + ((a ^ g) * 32) + pp * 8;
          const s16x4 a0 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s4)(pa));
          const s16x4 a1 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s4)(pa + 4 * 128));
          af[a] = __builtin_bit_cast(bf16x8, __builtin_shufflevector(a0, a1, 0, 1, 2, 3, 4, 5, 6, 7));
        }
#pragma unroll
        for (int s = 0; s < 4; s++) {
          const int q = wv + 4 * s;                   // (dh, half) fragment of this wave; q >= 14: none
          const int dh = q >> 1, half = q & 1;
          const char* pq = pb + ((r + (dh < 7 ? dh : 0)) * SW_PP + 8 * g + qq) * 8 + half * 32 + pp * 8;
          const s16x4 b0 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s4)(pq));
          const s16x4 b1 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s4)(pq + 32));
          bf[s] = __builtin_bit_cast(bf16x8, __builtin_shufflevector(b0, b1, 0, 1, 2, 3, 4, 5, 6, 7));
        }
#pragma unroll
        for (int s = 0; s < 4; s++) {
          if (wv + 4 * s >= 14) continue;             // (wave-uniform)
#pragma unroll
          for (int a = 0; a < 4; a++)
            acc[a][s] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bf[s], af[a], acc[a][s], 0, 0, 0);
        }
      }
      if (!has_next) break;
      patch_store(buf ^ 1, pv);
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __syncthreads();       // next tile landed; everyone is done with buffer `buf`
      buf ^= 1;
      it = it_next;
    }
  }
  // slab of this workgroup (zeros if it had no tile): the MFMAs ran with the patch as the row operand, so a lane holds
  // D[j = 16 q + 4 fg + e][co = 16 a + fr]: 4 consecutive j of one cout row
  float* slab = p.slabs + (size_t)blockIdx.x * SW_SLAB;
#pragma unroll
  for (int s = 0; s < 4; s++) {
    const int q = wv + 4 * s;
    if (q >= 14) continue;
#pragma unroll
    for (int a = 0; a < 4; a++)
      *reinterpret_cast<f32x4*>(slab + (size_t)(16 * a + fr) * 224 + 16 * q + 4 * fg) = acc[a][s];
  }
}

// dw[co][ch][dh][dw] (+)= sum_b slab[b][co][dh*32 + dw*4 + ch].  One WAVE per 4 consecutive j of a cout row: lane l adds
// slabs l, l+64, ... in index order, then the fixed xor-shuffle tree of wave_sum (a serial walk over 512 slabs per
// thread took 44 us -- longer than the gradient kernel itself).  The order is fixed, so the result is reproducible.
__global__ __launch_bounds__(256) void stem_wgrad_reduce_kernel(const float* __restrict__ slabs, float* __restrict__ dw,
                                                                int nslabs, int Ci, int accumulate) {
  const int i = blockIdx.x * 4 + (threadIdx.x >> 6);        // (co, j4): 64 x 56 groups of 4 floats
  const int lane = threadIdx.x & 63;
  if (i >= 64 * 56) return;
  const int co = i / 56, j4 = i - co * 56;
  const f32x4* s = reinterpret_cast<const f32x4*>(slabs + (size_t)co * 224 + j4 * 4);
  f32x4 a = {0.f, 0.f, 0.f, 0.f};
  for (int k = lane; k < nslabs; k += 64) a += s[(size_t)k * (SW_SLAB / 4)];
#pragma unroll
  for (int e = 0; e < 4; e++) a[e] = wave_sum(a[e]);
  if (lane == 0) {
    const int j = j4 * 4;                                   // = dh*32 + dw*4 (+ ch = e)
    const int dh = j >> 5, dwi = (j & 31) >> 2;
    if (dwi < 7) {
#pragma unroll
      for (int e = 0; e < 4; e++)
        if (e < Ci) {
          float* o = dw + (((size_t)co * Ci + e) * 7 + dh) * 7 + dwi;
          *o = accumulate ? *o + a[e] : a[e];
        }
    }
  }
}

bool mt_stem_wgrad_ok(const mt_conv_desc* d) {
  mt_stem_enable(-1);
  if (!g_stem_on) return false;
  if (d->dtype != MT_BF16 || d->transposed || d->kh != 7 || d->kw != 7 || d->stride != 1 || d->pad != 3 || d->Ci > 4 ||
      d->Co != 64 || d->H < 8 || d->W < 8)
    return false;
  return (unsigned long long)d->N * d->H * d->W * 128ull < 0x7f000000ull;
}
static int stem_wgrad_blocks(const mt_conv_desc* d) {
  const int total = d->N * cdiv(d->W, SW_TW) * cdiv(d->H, SW_TH);
  static const int cap = getenv("MT_STEM_WG_BLOCKS") ? atoi(getenv("MT_STEM_WG_BLOCKS")) : 384;   // (256: 54 + 9 us, 384: 43 + 12, 512: 42 + 15 for kernel + reduce)
  return total < cap ? total : cap;
}
size_t mt_stem_wgrad_ws_bytes(const mt_conv_desc* d) { return (size_t)stem_wgrad_blocks(d) * SW_SLAB * sizeof(float); }

// slabs -> ws; *nslabs = their number
int mt_launch_stem_wgrad(const mt_conv_desc* d, const void* x, const void* dy, void* ws, int* nslabs, hipStream_t s) {
  StemWgradParams p;
  p.x = (const char*)x; p.dy = (const char*)dy; p.slabs = (float*)ws;
  p.N = d->N; p.H = d->H; p.W = d->W; p.pad_mode = d->pad_mode;
  p.tiles_x = cdiv(d->W, SW_TW); p.tiles_y = cdiv(d->H, SW_TH);
  p.total = d->N * p.tiles_x * p.tiles_y;
  p.x_bytes = (unsigned)((size_t)d->N * d->H * d->W * 16); p.dy_bytes = (unsigned)((size_t)d->N * d->H * d->W * 128);
  const int nb = stem_wgrad_blocks(d);
  *nslabs = nb;
  if (nb <= 0) return 0;
  hipLaunchKernelGGL(stem_wgrad_kernel, dim3(nb), dim3(256), 0, s, p);
  MT_LAUNCH_CHECK();
  __atomic_fetch_add(&g_stem_launches, 1, __ATOMIC_RELAXED);
  return 0;
}
int mt_launch_stem_wgrad_reduce(const mt_conv_desc* d, const void* ws, int nslabs, float* dw, int accumulate, hipStream_t s) {
  hipLaunchKernelGGL(stem_wgrad_reduce_kernel, dim3(64 * 56 / 4), dim3(256), 0, s, (const float*)ws, dw, nslabs, d->Ci,
                     accumulate);
  MT_LAUNCH_CHECK();
  return 0;
}

// ------------------------------------------------------------------------------------------------------------------
// Direct 7x7 stem data gradient (bf16): the full correlation of dy with the filter on an output grid G (the padded
// (H+6)x(W+6) grid for reflection padding -- reflect_fold_kernel adds the border back afterwards -- or the HxW image
// itself for zero padding):   dxg[py][px][ch] = sum_{dh,dw,co} dy[py+o-dh][px+o-dw][co] * w[co][ch][dh][dw],  o = 0 / 3.
// As a gather GEMM this is 49 taps x 64 couts = 3136 reduction elements for 3 output channels: dy is re-read 49 times
// through LDS-DMA (6.4 GB for 134 MB at N = 16, 256x256: 338 us, the L2->LDS peak).  Here the 14x22-pixel patch of dy
// that an 8x16 output tile needs sits in LDS (16-byte chunks XOR-swizzled by the pixel column), the four weight rows
// [ch][tap][co] too (25 KiB, straight from the data-gradient pack), and K = (tap, co) runs over the patch:
//   D[ch][pixel] += W[ch][k] * dy_patch[k][pixel]        (MFMA 16x16x32, rows 4..15 of the weight operand are zero)
// 3 of 16 MFMA rows carry data -- the kernel is bound by the LDS reads of the patch, not by the matrix pipe.
constexpr int SD_TH = 8, SD_TW = 16;
constexpr int SD_PH = SD_TH + 6, SD_PW = SD_TW + 6;     // 14 x 22 pixels of dy

struct StemDgradParams {
  const char* dy;       // NHWC [N][H][W][64] bf16
  const char* wpack;    // data-gradient pack [8 rows = ch][49 taps][64 co] bf16
  char* out;            // NHWC [N][Hg][Wg][8] bf16
  int N, H, W, Hg, Wg, off;
  int tiles_x, tiles_y, total;
  unsigned dy_bytes, out_bytes;
};

// The 64 couts are walked in two halves of 32 (one MFMA k-step per tap and half): patch half 19.3 KiB + weight half
// 12.3 KiB per workgroup -> four workgroups (16 waves) per CU.  The loop is a chain of LDS read -> MFMA pairs, so what
// it needs is waves to hide the LDS latency behind (with the whole 64 couts resident, two workgroups per CU: 175 us).
__global__ __launch_bounds__(256, 4) void stem_dgrad_kernel(const StemDgradParams p) {
  constexpr unsigned OOB = 0x80000000u;
  constexpr int WH = 4 * 49 * 64;                              // bytes of a weight half [4 ch][49][32 co]
  __shared__ u32x4 smem[(WH + SD_PH * SD_PW * 64) / 16];
  char* const sW = reinterpret_cast<char*>(smem);              // [4 ch][49][32 co] bf16 of the current half
  char* const sP = sW + WH;                                    // [14][22 px][4 chunks of 16 B, chunk ^ ((px >> 2) & 3)]
  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
  const int fr = lane & 15, fg = lane >> 4;
  const __amdgpu_buffer_rsrc_t rsd = __builtin_amdgcn_make_buffer_rsrc((void*)p.dy, 0, p.dy_bytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t rso = __builtin_amdgcn_make_buffer_rsrc((void*)p.out, 0, p.out_bytes, 0x00020000);

  const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3;
  const int nslot = ((int)gridDim.x - xcd + 7) >> 3;
  const int tq = p.total >> 3, tr = p.total & 7;
  const int lo = xcd * tq + (xcd < tr ? xcd : tr);
  const int cnt = tq + (xcd < tr ? 1 : 0);
  const bool arow = fr < 4;                                   // lanes that hold a real weight row
  const char* wbase = sW + (arow ? fr : 0) * (49 * 64) + fg * 16;

  for (int it = slot; it < cnt; it += nslot) {
    const int tile = lo + it;
    const int tx = tile % p.tiles_x;
    const int t2 = tile / p.tiles_x;
    const int n = t2 / p.tiles_y;
    const int py0 = (t2 - n * p.tiles_y) * SD_TH, px0 = tx * SD_TW;
    // this wave: tile rows 2 wv, 2 wv + 1; pixel fragment = the 16 columns of a row
    f32x4 acc[2] = {f32x4{0.f, 0.f, 0.f, 0.f}, f32x4{0.f, 0.f, 0.f, 0.f}};
#pragma unroll 1
    for (int half = 0; half < 2; half++) {
      __syncthreads();                                        // everyone is done with the previous patch / weights
      // weights of this half: [ch][tap][32 co] <- pack [ch][tap][64 co] (16-byte chunks 4 half .. 4 half + 3 of a tap)
      for (int i = tid; i < WH / 16; i += 256) {
        const int ck = i & 3, rt = i >> 2;                    // rt = ch * 49 + tap
        reinterpret_cast<u32x4*>(sW)[i] = *reinterpret_cast<const u32x4*>(p.wpack + ((size_t)rt * 8 + half * 4 + ck) * 16);
      }
      // patch half: 14 x 22 pixels x 4 chunks; dy row = py0 + off - 6 + pr, column = px0 + off - 6 + pc (zeros outside
      // dy).  All loads of a thread are issued before the first LDS store; chunk slot = chunk ^ ((pc >> 2) & 3): the 16
      // consecutive pixels of a fragment read (pitch 64 B) hit 16 different 16-byte bank groups.
      constexpr int NLD = (SD_PH * SD_PW * 4 + 255) / 256;    // 5
      u32x4 pv[NLD];
#pragma unroll
      for (int j = 0; j < NLD; j++) {
        const int i = tid + 256 * j;
        const int ck = i & 3, pix = i >> 2;
        const int pr = pix / SD_PW, pc = pix - pr * SD_PW;
        const int y = py0 + p.off - 6 + pr, x = px0 + p.off - 6 + pc;
        const bool ok = i < SD_PH * SD_PW * 4 && (unsigned)y < (unsigned)p.H && (unsigned)x < (unsigned)p.W;
        const unsigned o = ok ? (unsigned)((n * p.H + y) * p.W + x) * 128u + (unsigned)(half * 4 + ck) * 16u : OOB;
        pv[j] = __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(rsd, o, 0, 0));
      }
#pragma unroll
      for (int j = 0; j < NLD; j++) {
        const int i = tid + 256 * j;
        const int ck = i & 3, pix = i >> 2;
        const int pc = pix % SD_PW;
        if (i < SD_PH * SD_PW * 4) *reinterpret_cast<u32x4*>(sP + (pix * 4 + (ck ^ ((pc >> 2) & 3))) * 16) = pv[j];
      }
      __syncthreads();
#pragma unroll 1
      for (int dh = 0; dh < 7; dh++) {
#pragma unroll
        for (int dw = 0; dw < 7; dw++) {
          const int pc = fr + 6 - dw;
          u32x4 wf = *reinterpret_cast<const u32x4*>(wbase + (dh * 7 + dw) * 64);
          if (!arow) wf = u32x4{0u, 0u, 0u, 0u};      // (skipping the read on those lanes with a branch was slower: 170 vs 151 us)
#pragma unroll
          for (int b = 0; b < 2; b++) {
            const int pr = 2 * wv + b + 6 - dh;
            const u32x4 xf = *reinterpret_cast<const u32x4*>(sP + ((pr * SD_PW + pc) * 4 + (fg ^ ((pc >> 2) & 3))) * 16);
            mma_chunk<true>(acc[b], wf, xf);
          }
        }
      }
    }
    // lanes fg == 0 hold channels 0..3 of pixel (row, column fr); channels 4..7 of the padded layout are zero
    if (fg == 0) {
#pragma unroll
      for (int b = 0; b < 2; b++) {
        const int py = py0 + 2 * wv + b, px = px0 + fr;
        const bool ok = py < p.Hg && px < p.Wg;
        const u32x4 o = {pack2_bf16(acc[b][0], acc[b][1]), pack2_bf16(acc[b][2], acc[b][3]), 0u, 0u};
        __builtin_amdgcn_raw_buffer_store_b128(o, rso, ok ? (unsigned)((n * p.Hg + py) * p.Wg + px) * 16u : OOB, 0, 0);
      }
    }
  }
}

bool mt_stem_dgrad_ok(const mt_conv_desc* d) {
  mt_stem_enable(-1);
  if (!g_stem_on) return false;
  if (d->dtype != MT_BF16 || d->transposed || d->kh != 7 || d->kw != 7 || d->stride != 1 || d->pad != 3 || d->Ci > 4 ||
      d->Co != 64 || d->H < 8 || d->W < 8)
    return false;
  return (unsigned long long)d->N * (d->H + 6) * (d->W + 6) * 128ull < 0x7f000000ull;
}
// out: the padded grid [N][H+6][W+6][8] (reflection padding; fold it afterwards) or dx itself (zero padding)
int mt_launch_stem_dgrad(const mt_conv_desc* d, const void* dy, const void* pack_bwd, void* out, hipStream_t s) {
  StemDgradParams p;
  const bool refl = d->pad_mode == MT_PAD_REFLECT;
  p.dy = (const char*)dy; p.wpack = (const char*)pack_bwd; p.out = (char*)out;
  p.N = d->N; p.H = d->H; p.W = d->W;
  p.Hg = refl ? d->H + 6 : d->H; p.Wg = refl ? d->W + 6 : d->W; p.off = refl ? 0 : 3;
  p.tiles_x = cdiv(p.Wg, SD_TW); p.tiles_y = cdiv(p.Hg, SD_TH);
  p.total = d->N * p.tiles_x * p.tiles_y;
  p.dy_bytes = (unsigned)((size_t)d->N * d->H * d->W * 128); p.out_bytes = (unsigned)((size_t)d->N * p.Hg * p.Wg * 16);
  if (p.total <= 0) return 0;
  const int nb = p.total < 1024 ? p.total : 1024;
  hipLaunchKernelGGL(stem_dgrad_kernel, dim3(nb), dim3(256), 0, s, p);
  MT_LAUNCH_CHECK();
  __atomic_fetch_add(&g_stem_launches, 1, __ATOMIC_RELAXED);
  return 0;
}
