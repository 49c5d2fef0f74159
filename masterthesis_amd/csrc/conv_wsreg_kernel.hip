// Weight-stationary gather-GEMM for the short-K / huge-M convolutions (round 4; gfx950 only, bf16).
//
// The layers between the stem and the 256-channel bottleneck -- content-encoder down-sampling (3x3 stride 2, 64 -> 128), the
// decoder's transposed convolutions (128 -> 64), the style encoder's 3x3 layers at 64 channels, and their data gradients
// (reference networks.py:33,248; blocks.py:73,93-119) -- have K = 9 taps x 64..128 channels and hundreds of thousands of output
// pixels: by the roofline they are HBM-bound (134 MB in, 67 MB out against 39 GFLOP), and the tile GEMMs ran them at 0.18-0.28 of
// that bound because every k-step re-stages weights AND pixels through LDS-DMA (9 taps = 9 copies of every pixel) and meets at
// two barriers per 32-deep step.  Here nothing but the input patch is staged, and it is staged once:
//   * WEIGHTS LIVE IN REGISTERS.  A layer's whole weight tensor is 74-147 KB -- less than a third of one compute unit's register
//     file.  Each of the 8 waves loads the MFMA A fragments of ITS 16 or 32 output channels for all taps and channel slices once
//     per launch (144 VGPRs) and keeps them: no weight traffic, no weight staging, no k-step barrier.
//   * THE PIXEL OPERAND IS A PATCH IN LDS.  A workgroup walks tiles of TH x TW output (base-grid) pixels; the tile's input patch
//     ((TH-1) is + span rows) goes to LDS by LDS-DMA once, double-buffered across tiles, as KC planes of 64-byte rows (one plane
//     per 32-channel slice; XOR swizzle on the copies' source side and on the fragment reads: conflict-free for 16 consecutive
//     rows at any start, so a tap shift costs nothing; stride-2 gathers store even and odd patch columns in separate half
//     planes so that a fragment's 16 pixels are consecutive rows again).  A B fragment is one ds_read_b128 at
//     (pixel + tap offset); in the scatter form (transposed convolution / strided data gradient) the four sub-pixel phases are
//     computed together and an input offset shared by several phases is read once.
//   * ONE BARRIER PER TILE (2-4 k MFMA cycles), none inside: between barriers every wave runs its own stream of
//     {ds_read_b128, 2-4 MFMAs}; outputs leave as 16-byte (8-byte for 16-channel waves) buffer stores whose count per tile is a
//     constant (out-of-range pixels carry an out-of-range offset and are dropped by the hardware), so the end-of-tile wait is a
//     COUNTED vmcnt that leaves the stores in flight and only retires the next patch's copies.
// Same IgemmParams as every gather-GEMM here; launch_igemm_t asks launch_igemm_wsreg first.  MT_IGEMM_WSREG=0 /
// mt_kernel_variant_enable(4, 0) switch it off.  Parity: tests/test_wsreg_gpu.py (against the tile kernels and the fp32 CPU
// reference).
#include "conv_device.h"
#include <stdlib.h>
#include <type_traits>
#include <string.h>

// bytes of one patch slot: KC planes of 64-byte rows (KC = 2: 320 rows per plane; KC = 4: 176).  THREE slots: the patches of the
// next TWO tiles are in flight while a tile computes -- with one patch ahead the kernel ran at the DMA's latency (first
// measurement, 3x3 s2 64 -> 128: 59.6 us = 40 KB per CU per ~4 us round trip under load; the MFMA work of a tile is ~1 us)
constexpr int WS_NSLOT = 3;
template <int KC> struct WsSlot { static constexpr int B = 45056; };
constexpr int WS_MAXU = 9;                 // distinct input offsets per base pixel

struct WsGeom {
  int TH, TW;                // tile of base-grid pixels (TW a multiple of 16)
  int PW, PWh;               // patch row pitch in rows (is = 2: two half planes of PWh columns, PW = 2 PWh), padded
  int PWc;                   // patch columns that are actually read (the rest of the pitch is padding)
  int nrows;                 // patch rows per plane (PH * PW)
  int ncp;                   // copies per plane = ceil(nrows / 16)
  int dh0, dw0;              // smallest tap offsets
  int Hb, Wb;                // base grid (gather: the output grid; scatter: the largest phase grid)
  int tiles_w, tiles_hw;     // tiles per row / per image
  int ntiles;
  int qoff[WS_MAXU];         // patch-row offset of input offset u relative to the base pixel's row
};

// GEOM 0: gather form, 9 taps, one phase (tap t reads its own offset).  GEOM 1: scatter form of a 3x3 / stride 2 window: phases
// (kh % 2, kw % 2) in the order scatter_form builds them -- 4, 2, 2, 1 taps -- over the four input offsets {0,1}^2 (relative).
template <int GEOM> struct WsTaps;
template <> struct WsTaps<0> {
  static constexpr int NT = 9, NU = 9, NPH = 1;
  static constexpr int uid[9] = {0, 1, 2, 3, 4, 5, 6, 7, 8};
  static constexpr int phase[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0};
  static constexpr int tloc[9] = {0, 1, 2, 3, 4, 5, 6, 7, 8};
};
template <> struct WsTaps<1> {
  static constexpr int NT = 9, NU = 4, NPH = 4;
  static constexpr int uid[9] = {3, 2, 1, 0, 2, 0, 1, 0, 0};
  static constexpr int phase[9] = {0, 0, 0, 0, 1, 1, 2, 2, 3};
  static constexpr int tloc[9] = {0, 1, 2, 3, 0, 1, 0, 1, 0};
};

// GEOM 2: the same scatter form as the data gradient of a REFLECTION-padded stride-2 convolution builds it (e = 0 on the padded
// grid: the phases with one and two taps read the offsets at the far corner)
template <> struct WsTaps<2> {
  static constexpr int NT = 9, NU = 4, NPH = 4;
  static constexpr int uid[9] = {3, 2, 1, 0, 3, 1, 3, 2, 3};
  static constexpr int phase[9] = {0, 0, 0, 0, 1, 1, 2, 2, 3};
  static constexpr int tloc[9] = {0, 1, 2, 3, 0, 1, 0, 1, 0};
};

// CF: 16-row output-channel fragments per wave; KC: 32-channel slices of the input; FPW: pixel fragments per wave per tile
// Y2: the second destination of the persistent gather-GEMM (IgemmParams::y2: the interior of a padded gradient map straight into
// dx, the ring into the workspace): every output chunk is stored twice, the lanes of the other destination out of range
template <int GEOM, int CF, int KC, int FPW, bool STATS, bool Y2 = false>
__global__ __launch_bounds__(512) void igemm_wsreg_kernel(const IgemmParams p, const WsGeom g) {
  using TP = WsTaps<GEOM>;
  constexpr int NT = TP::NT, NU = TP::NU, NPH = TP::NPH;
  constexpr int NW = 8;
  constexpr int SLOTB = WsSlot<KC>::B;
  constexpr int PLANEB = SLOTB / KC;
  constexpr int NST = FPW * NPH * (Y2 ? 2 : 1);                    // output stores per wave per tile
  static_assert(2 * NST + 6 <= 38, "the counted wait's immediates");
  constexpr unsigned OOB = 0x80000000u;
  static_assert(NT * KC * CF * 4 <= 144, "weight fragments must fit the register budget");
  static_assert(!STATS || NPH == 1, "fused statistics: gather form only");
  __shared__ u32x4 smem[WS_NSLOT * SLOTB / 16];

  const int tid = threadIdx.x, lane = tid & 63;
  const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int fr = lane & 15, fg = lane >> 4;
  const int G = p.CoRows / (16 * CF);                              // output-channel groups; NW / G pixel groups
  const int cg = wv % G, pg = wv / G;
  const int is = p.is, os = p.os;

  // ---- this wave's weights: A fragments of its 16 * CF output channels, all taps, all slices ----
  // row permutation (CF = 2): fragment cf, row i holds channel (i >> 2) * 8 + cf * 4 + (i & 3) of the wave's 32, so that a lane
  // ends up with 8 consecutive output channels of its pixel (one 16-byte store)
  u32x4 wreg[NT][KC][CF];
#pragma unroll
  for (int t = 0; t < NT; t++) {
    const IgemmPhase& q = p.ph[TP::phase[t]];
    const u32x4* wb = reinterpret_cast<const u32x4*>(p.w + q.w_off);
#pragma unroll
    for (int cf = 0; cf < CF; cf++) {
      const int rl = CF == 2 ? ((fr >> 2) * 8 + cf * 4 + (fr & 3)) : fr;
      const int row = cg * 16 * CF + rl;
#pragma unroll
      for (int s = 0; s < KC; s++) wreg[t][s][cf] = wb[(size_t)row * q.wrow + TP::tloc[t] * p.cpc + s * 4 + fg];
    }
  }
  const float act_ns = p.act == MT_ACT_RELU ? 0.f : (p.act == MT_ACT_LRELU ? p.slope : 1.f);
  const int co0 = cg * 16 * CF + fg * 4 * CF;                      // first of this lane's 4 * CF output channels
  float bv[4 * CF];
#pragma unroll
  for (int e = 0; e < 4 * CF; e++) bv[e] = (p.bias != nullptr && co0 + e < p.nbias) ? p.bias[co0 + e] : 0.f;

  // ---- tile-invariant addressing ----
  // fragment j of this wave: base pixel (ty, tx0 + fr) of the tile; LDS byte offset of its B chunk at input offset u (plane 0)
  const int fpr = g.TW >> 4;                                       // fragments per tile row
  // (the byte offset at input offset u -- row + qoff[u], swizzled -- is formed in the loop: five vector instructions per 4+ MFMAs
  //  ride in the MFMAs' issue shadow, 9 x FPW resident addresses would spill)
  int fshift = 0;
  while ((1 << fshift) < fpr) fshift++;
  int qo[NU];
#pragma unroll
  for (int u = 0; u < NU; u++) qo[u] = __builtin_amdgcn_readfirstlane(g.qoff[u]);
  // B-fragment byte offsets (plane 0, slot 0) of this wave's FIRST fragment at every input offset: resident (NU registers).  The
  // patch pitch is padded so that is * PW is a multiple of 8 rows (host) and fragments sit 16 pixels apart, hence every other
  // fragment of the wave is a multiple of 8 rows away -- the swizzle key ((row >> 1) & 3) is the same and its offsets are
  // these plus a wave-uniform constant: one vector add per read pair instead of five instructions per offset
  int baddr0[NU];
  int rb0;
  {
    const int f0 = pg * FPW;
    const int fty0 = f0 >> fshift;
    rb0 = fty0 * is * g.PW + ((f0 - (fty0 << fshift)) << 4);
#pragma unroll
    for (int u = 0; u < NU; u++) {
      const int r = rb0 + fr + qo[u];
      baddr0[u] = r * 64 + ((fg ^ ((r >> 1) & 3)) << 4);
    }
  }
  // patch copies of this wave: copy c = wv + 8 k -> plane c / ncp, rows 16 (c % ncp) ..; this lane's row -> patch coordinates
  constexpr int MAXC = 6;
  const int ncopies = KC * g.ncp;
  const int nmine = ncopies > wv ? (ncopies - wv + NW - 1) / NW : 0;          // wave-uniform
  int cpy[MAXC];                                                   // (py << 16) | px; a row past the patch gets px = 0x7fff (never in range)
  unsigned csrc[MAXC];                                             // slice + swizzled chunk byte offset inside the pixel
  int cdst[MAXC];                                                  // LDS byte offset of the copy inside a slot (wave-uniform)
  unsigned crel[MAXC];                                             // byte offset of the lane's source chunk from the tile's first
                                                                   // patch pixel (interior tiles: no padding to resolve)
  int cib_shift = 0;
  while ((1 << cib_shift) < p.Cib) cib_shift++;                    // (Cib = 128 or 256 bytes per input pixel: host)
#pragma unroll
  for (int k = 0; k < MAXC; k++) {
    const int c = wv + NW * k;
    const int pl = c / g.ncp, rr = c - pl * g.ncp, r = rr * 16 + (lane >> 2);
    int py = r / g.PW, rem = r - py * g.PW, px = rem;
    if (is == 2) {
      const int par = rem >= g.PWh ? 1 : 0;
      px = 2 * (rem - par * g.PWh) + par;
    }
    const bool used = r < g.nrows && px < g.PWc;                   // (rows of the pitch's padding are never read: not copied)
    cpy[k] = used ? ((py << 16) | px) : 0x7fff;
    csrc[k] = (unsigned)(pl * 64 + (((lane & 3) ^ ((r >> 1) & 3)) << 4));
    cdst[k] = __builtin_amdgcn_readfirstlane(pl * PLANEB + rr * 1024);
    crel[k] = used ? (((unsigned)(py * p.Wi + px)) << cib_shift) + csrc[k] : 0u;     // (unused rows re-read the tile's first pixel)
  }
  const unsigned Hi = (unsigned)p.Hi, Wi = (unsigned)p.Wi;
  const bool reflect = p.pad_mode == MT_PAD_REFLECT;

  const __amdgpu_buffer_rsrc_t rsx = __builtin_amdgcn_make_buffer_rsrc((void*)p.x, 0, p.x_bytes, 0x00020000);
  const unsigned y_bytes = (unsigned)((size_t)p.N * p.Hout * p.Wout * p.Co * 2);
  const __amdgpu_buffer_rsrc_t rsy = __builtin_amdgcn_make_buffer_rsrc((void*)p.y, 0, y_bytes, 0x00020000);
  const unsigned y2_bytes = Y2 ? (unsigned)((size_t)p.N * p.y2H * p.y2W * p.Co * 2) : 0u;
  const __amdgpu_buffer_rsrc_t rsy2 = __builtin_amdgcn_make_buffer_rsrc((void*)(Y2 ? p.y2 : p.y), 0, y2_bytes, 0x00020000);
  typedef __attribute__((address_space(3))) void* lds_ptr;
  char* const lds0 = reinterpret_cast<char*>(&smem[0]);

  // tile coordinates, advanced one tile at a time (no divisions in the loop: the kernel is bound by instruction issue)
  struct TileC { int n, th, tw; };
  auto tile_coords = [&](int tile) {
    TileC c;
    c.n = tile / g.tiles_hw;
    const int rem = tile - c.n * g.tiles_hw;
    c.th = rem / g.tiles_w;
    c.tw = rem - c.th * g.tiles_w;
    return c;
  };
  const int tiles_h = g.tiles_hw / g.tiles_w;
  auto tile_next = [&](TileC& c) {
    c.tw++;
    if (c.tw == g.tiles_w) { c.tw = 0; c.th++; }
    if (c.th == tiles_h) { c.th = 0; c.n++; }
  };
  // the copies of one tile's patch.  Interior tiles (no padding to resolve: ~70 % of them) cost ONE instruction per copy: the lane's
  // tile-relative source offset is resident and the tile's base travels in the instruction's scalar offset; border tiles resolve
  // reflection / zero padding per row (~12 vector instructions per copy).  (First version: 65 instructions per copy, 1400 per tile
  // and wave against 72 MFMAs.)
  const int ph_rows = g.nrows / g.PW, pw_cols = g.PWc;             // patch extent in input pixels
  auto issue_patch = [&](const TileC& c, int slot) -> int {
    const int hb = c.th * g.TH * is + g.dh0, wb = c.tw * g.TW * is + g.dw0;
    const unsigned nbase = (unsigned)c.n * (Hi * Wi) << cib_shift;
    char* const sbase = lds0 + slot * SLOTB;
    const bool interior = hb >= 0 && wb >= 0 && hb + ph_rows <= (int)Hi && wb + pw_cols <= (int)Wi;
    if (interior) {
      const unsigned tbase = nbase + (((unsigned)hb * Wi + (unsigned)wb) << cib_shift);
#pragma unroll
      for (int k = 0; k < MAXC; k++)
        if (k < nmine) __builtin_amdgcn_raw_ptr_buffer_load_lds(rsx, (lds_ptr)(sbase + cdst[k]), 16, crel[k], tbase, 0, 0);
      return nmine;
    }
#pragma unroll
    for (int k = 0; k < MAXC; k++) {
      if (k < nmine) {                                             // (wave-uniform)
        int hi = hb + (cpy[k] >> 16), wi = wb + (cpy[k] & 0xffff);
        if (reflect) {
          hi = hi < 0 ? -hi : hi;
          hi = hi >= (int)Hi ? 2 * ((int)Hi - 1) - hi : hi;
          wi = wi < 0 ? -wi : wi;
          wi = wi >= (int)Wi ? 2 * ((int)Wi - 1) - wi : wi;
        }
        const bool ok = (unsigned)hi < Hi && (unsigned)wi < Wi;
        const unsigned off = ok ? nbase + ((__umul24((unsigned)hi, Wi) + (unsigned)wi) << cib_shift) + csrc[k] : OOB;
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rsx, (lds_ptr)(sbase + cdst[k]), 16, off, 0, 0, 0);
      }
    }
    return nmine;
  };
  // wait until all but the n youngest vector-memory operations of this wave have completed (n wave-uniform, an immediate per case)
  auto wait_vm = [&](int n) {
#define WS_W(k) case k: asm volatile("s_waitcnt vmcnt(%0)" ::"n"(k) : "memory"); break;
    switch (n) {
      WS_W(0) WS_W(1) WS_W(2) WS_W(3) WS_W(4) WS_W(5) WS_W(6) WS_W(7) WS_W(8) WS_W(9) WS_W(10) WS_W(11) WS_W(12) WS_W(13) WS_W(14)
      WS_W(15) WS_W(16) WS_W(17) WS_W(18) WS_W(19) WS_W(20) WS_W(21) WS_W(22) WS_W(23) WS_W(24) WS_W(25) WS_W(26) WS_W(27) WS_W(28)
      WS_W(29) WS_W(30) WS_W(31) WS_W(32) WS_W(33) WS_W(34) WS_W(35) WS_W(36) WS_W(37) WS_W(38)
      default: asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); break;
    }
#undef WS_W
  };

  // ---- persistent walk: workgroup v owns tiles [v * tpw, (v + 1) * tpw) (XCD-contiguous: neighbouring tiles share halo rows in L2)
  const int v = xcd_remap(blockIdx.x, gridDim.x);
  const int tpw = (g.ntiles + (int)gridDim.x - 1) / (int)gridDim.x;
  const int t_first = v * tpw;
  const int t_end = t_first + tpw < g.ntiles ? t_first + tpw : g.ntiles;

  float s1[STATS ? 4 * CF : 1], s2[STATS ? 4 * CF : 1];
  int stat_n = -1;
  if constexpr (STATS) {
#pragma unroll
    for (int e = 0; e < 4 * CF; e++) { s1[e] = 0.f; s2[e] = 0.f; }
  }
  auto flush_stats = [&]() {
    if constexpr (STATS) {
      if (stat_n >= 0) {
#pragma unroll
        for (int e = 0; e < 4 * CF; e++) {
          const float a = row16_sum(s1[e]), b = row16_sum(s2[e]);
          if (fr == 0 && co0 + e < p.Co) {
            atomicAdd(p.stats + ((size_t)stat_n * p.Co + co0 + e) * 2, a);
            atomicAdd(p.stats + ((size_t)stat_n * p.Co + co0 + e) * 2 + 1, b);
          }
          s1[e] = 0.f; s2[e] = 0.f;
        }
      }
    }
  };

  // ring of three slots: tile i computes from slot i % 3 while the patches of tiles i + 1 and i + 2 are in flight.
  // (first a USE of every loaded register: hipcc then waits for the weight / bias loads here, once -- with them still pending in
  //  its model it put a vmcnt(0) in front of every copy of the prologue, i.e. the first two patches went out one round trip at a
  //  time; from here on the queue holds copies and stores only)
#pragma unroll
  for (int t = 0; t < NT; t++)
#pragma unroll
    for (int s = 0; s < KC; s++)
#pragma unroll
      for (int cf = 0; cf < CF; cf++) asm volatile("" : "+v"(wreg[t][s][cf]));
#pragma unroll
  for (int e = 0; e < 4 * CF; e++) asm volatile("" : "+v"(bv[e]));
  TileC tc = tile_coords(t_first), tp = tc;                        // tile being computed / tile being prefetched
  if (t_first < t_end) issue_patch(tp, 0);
  tile_next(tp);
  const int c1 = t_first + 1 < t_end ? issue_patch(tp, 1) : 0;
  wait_vm(c1);
  __builtin_amdgcn_s_barrier();
  int slot = 0;
  for (int tile = t_first; tile < t_end; tile++) {
    __builtin_amdgcn_sched_barrier(0);
    // (slot + 2) % 3 held tile - 1: every wave's reads of it were issued and consumed before the previous barrier
    const int slot2 = slot == 0 ? 2 : slot - 1;
    tile_next(tp);
    const int c2 = tile + 2 < t_end ? issue_patch(tp, slot2) : 0;
    __builtin_amdgcn_sched_barrier(0);
    const int n = tc.n, th = tc.th, tw = tc.tw;
    if constexpr (STATS) {
      if (n != stat_n) { flush_stats(); stat_n = n; }
    }
    const char* const ldsS = lds0 + slot * SLOTB;
    // one pixel fragment
    auto frag = [&](int j) {
      const int f = pg * FPW + j;
      const int fty = f >> fshift;
      const int ftx0 = (f - (fty << fshift)) << 4;
      const int ftx = ftx0 + fr;
      const int dj = __builtin_amdgcn_readfirstlane((fty * is * g.PW + ftx0 - rb0) * 64);      // (a multiple of 512 bytes)
      f32x4 acc[NPH][CF];
#pragma unroll
      for (int q = 0; q < NPH; q++)
#pragma unroll
        for (int cf = 0; cf < CF; cf++) acc[q][cf] = f32x4{0.f, 0.f, 0.f, 0.f};
      // B fragments one input offset ahead of the MFMAs that consume them (the reads of offset u + 1 are issued before the MFMAs
      // of offset u: the compiler otherwise issues them behind and every tap waits out an LDS round trip)
      constexpr int RD = (KC == 2 && !STATS) ? 3 : 2;           // read-ahead ring (offsets in flight beside the one being consumed: RD - 1)
      u32x4 xb[RD][KC];
      auto read_b = [&](int u, u32x4* dst) {
        const int ba = baddr0[u] + dj;
#pragma unroll
        for (int s = 0; s < KC; s++) dst[s] = *reinterpret_cast<const u32x4*>(ldsS + ba + s * PLANEB);
      };
#pragma unroll
      for (int u = 0; u < RD - 1 && u < NU; u++) read_b(u, xb[u % RD]);
#pragma unroll
      for (int u = 0; u < NU; u++) {
        if (u + RD - 1 < NU) read_b(u + RD - 1, xb[(u + RD - 1) % RD]);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int t = 0; t < NT; t++) {
          if (TP::uid[t] != u) continue;
#pragma unroll
          for (int s = 0; s < KC; s++)
#pragma unroll
            for (int cf = 0; cf < CF; cf++) mma_chunk<true>(acc[TP::phase[t]][cf], wreg[t][s][cf], xb[u % RD][s]);
        }
        __builtin_amdgcn_sched_barrier(0);
      }
      // ---- outputs of this fragment: bias + activation, one packed store per phase (constant count: NST per tile) ----
      const int ho = th * g.TH + fty, wo = tw * g.TW + ftx;
#pragma unroll
      for (int q = 0; q < NPH; q++) {
        const IgemmPhase& ph = p.ph[q];
        const int oh = ho * os + ph.oh0, ow = wo * os + ph.ow0;
        const bool ok = ho < ph.Ho && wo < ph.Wo && oh < p.Hout && ow < p.Wout && n < p.N;
        float vv[4 * CF];
#pragma unroll
        for (int cf = 0; cf < CF; cf++)
#pragma unroll
          for (int e = 0; e < 4; e++) {
            // branch-free none / ReLU / LeakyReLU: max(v, 0) + ns * min(v, 0) with ns = 1 / 0 / slope (tanh: host refuses)
            const float v0 = acc[q][cf][e] + bv[cf * 4 + e];
            vv[cf * 4 + e] = fmaxf(v0, 0.f) + (act_ns == 0.f ? 0.f : act_ns * fminf(v0, 0.f));     // (select: relu(-inf) = 0)
          }
        if constexpr (STATS) {
          if (ok) {
#pragma unroll
            for (int e = 0; e < 4 * CF; e++) { s1[e] += vv[e]; s2[e] += vv[e] * vv[e]; }
          }
        }
        unsigned off = ok ? (unsigned)((((size_t)n * p.Hout + oh) * p.Wout + ow) * p.Co + co0) * 2u : OOB;
        unsigned off2 = OOB;
        if constexpr (Y2) {
          const int ih = oh - p.y2P, iw = ow - p.y2P;
          const bool inside = (unsigned)ih < (unsigned)p.y2H && (unsigned)iw < (unsigned)p.y2W;
          off2 = (ok && inside) ? (unsigned)((((size_t)n * p.y2H + ih) * p.y2W + iw) * p.Co + co0) * 2u : OOB;
          off = inside ? OOB : off;
        }
        if constexpr (CF == 2) {
          const u32x4 o = {pack2_bf16(vv[0], vv[1]), pack2_bf16(vv[2], vv[3]), pack2_bf16(vv[4], vv[5]), pack2_bf16(vv[6], vv[7])};
          __builtin_amdgcn_raw_buffer_store_b128(o, rsy, off, 0, 0);
          if constexpr (Y2) __builtin_amdgcn_raw_buffer_store_b128(o, rsy2, off2, 0, 0);
        } else {
          const u32x2 o = {pack2_bf16(vv[0], vv[1]), pack2_bf16(vv[2], vv[3])};
          __builtin_amdgcn_raw_buffer_store_b64(o, rsy, off, 0, 0);
          if constexpr (Y2) __builtin_amdgcn_raw_buffer_store_b64(o, rsy2, off2, 0, 0);
        }
      }
    };
#pragma unroll 1
    for (int j = 0; j < FPW; j++) frag(j);        // (not unrolled: the fragments' streams would interleave and spill)
    // queue of this wave, oldest first: copies(tile + 1) | stores(tile - 1) | copies(tile + 2) | stores(tile).  The patch of
    // tile + 1 has landed once all but the 2 NST + c2 youngest operations have completed; the stores stay in flight
    __builtin_amdgcn_sched_barrier(0);
    wait_vm(2 * NST + c2);
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_sched_barrier(0);
    slot = slot == 2 ? 0 : slot + 1;
    tile_next(tc);
  }
  flush_stats();
}

static long g_ws_launches = 0;
static int g_ws_on = -1;
static int ws_enabled() {
  if (g_ws_on < 0) g_ws_on = getenv("MT_IGEMM_WSREG") ? (atoi(getenv("MT_IGEMM_WSREG")) != 0) : 1;
  return g_ws_on;
}
long mt_wsreg_launches() { return g_ws_launches; }
int mt_wsreg_enable(int on) {
  const int prev = ws_enabled();
  g_ws_on = on != 0;
  return prev;
}

// -> 0 launched, 1 error, -1 not this kernel's shape; dry: 103 = would launch
#define WS_REJECT(k) do { if (getenv("MT_WSREG_DEBUG")) fprintf(stderr, "wsreg: reject %d (Co %d cpc %d nphase %d is %d os %d)\n", k, p.CoRows, p.cpc, p.nphase, p.is, p.os); return -1; } while (0)
int launch_igemm_wsreg(IgemmParams& p, hipStream_t s, bool dry) {
  if (!ws_enabled()) WS_REJECT(1);
  if (p.raw || p.addend != nullptr || p.fold || p.bstat_x != nullptr || p.act == MT_ACT_TANH) WS_REJECT(2);
  const bool y2 = p.y2 != nullptr;
  if (y2 && (double)p.N * p.y2H * p.y2W * p.Co * 2 >= 2147000000.0) WS_REJECT(3);
  if (p.x_bytes >= 0x7f000000u || (double)p.N * p.Hout * p.Wout * p.Co * 2 >= 2147000000.0) WS_REJECT(3);
  const int KC = p.cpc / 4;
  if (p.cpc % 4 != 0 || !(KC == 2 || KC == 4)) WS_REJECT(4);
  if (!(p.CoRows == 64 || p.CoRows == 128) || p.Co != p.CoRows) WS_REJECT(5);
  const int CF = (KC == 2) ? 2 : 1;                 // 9 taps x KC x CF x 4 registers = 144
  const int G = p.CoRows / (16 * CF);
  // (128 output channels x 128 input channels would put 16 channels on each of the 8 waves, every wave walking ALL pixels: one
  //  MFMA per LDS read -- measured level with the tile kernel, 30.8 vs 30.3 us: not taken)
  if (G > 4 || 8 % G != 0) WS_REJECT(6);
  const int PG = 8 / G;
  int geom = -1;
  if (p.nphase == 1 && p.ph[0].ntaps == 9 && p.os == 1 && (p.is == 1 || p.is == 2)) geom = 0;
  if (p.nphase == 4 && p.is == 1 && p.os == 2 && p.ph[0].ntaps == 4 && p.ph[1].ntaps == 2 && p.ph[2].ntaps == 2 &&
      p.ph[3].ntaps == 1 && p.stats == nullptr && p.pad_mode == MT_PAD_ZERO)
    geom = 1;
  if (geom < 0) WS_REJECT(7);
  if (p.stats != nullptr && (geom != 0 || p.act != MT_ACT_NONE)) WS_REJECT(8);
  for (int q = 0; q < p.nphase; q++)
    if (p.ph[q].y_off != 0 || p.ph[q].wrow != p.ph[q].ntaps * p.cpc) WS_REJECT(9);
  WsGeom g;
  memset(&g, 0, sizeof(g));
  int dhmin = 1 << 20, dhmax = -(1 << 20), dwmin = 1 << 20, dwmax = -(1 << 20), nt = 0;
  for (int q = 0; q < p.nphase; q++) {
    g.Hb = p.ph[q].Ho > g.Hb ? p.ph[q].Ho : g.Hb;
    g.Wb = p.ph[q].Wo > g.Wb ? p.ph[q].Wo : g.Wb;
    for (int t = 0; t < p.ph[q].ntaps; t++, nt++) {
      const int dh = p.dh[p.ph[q].tap0 + t], dw = p.dw[p.ph[q].tap0 + t];
      dhmin = dh < dhmin ? dh : dhmin; dhmax = dh > dhmax ? dh : dhmax;
      dwmin = dw < dwmin ? dw : dwmin; dwmax = dw > dwmax ? dw : dwmax;
    }
  }
  if (nt != 9 || g.Hb < 1 || g.Wb < 16) WS_REJECT(10);
  g.dh0 = dhmin; g.dw0 = dwmin;
  const int sh = dhmax - dhmin, sw = dwmax - dwmin;
  if (geom == 0 && (sh != 2 || sw != 2)) WS_REJECT(11);
  if (geom >= 1 && (sh != 1 || sw != 1)) WS_REJECT(12);
  // the taps' offsets in the order the kernel's tables assume (scatter form: one of the two known patterns)
  static const int uid1[9] = {3, 2, 1, 0, 2, 0, 1, 0, 0}, uid2[9] = {3, 2, 1, 0, 3, 1, 3, 2, 3};
  if (geom == 1) {
    bool m1 = true, m2 = true;
    int k = 0;
    for (int q = 0; q < p.nphase; q++)
      for (int t = 0; t < p.ph[q].ntaps; t++, k++) {
        const int u = (p.dh[p.ph[q].tap0 + t] - dhmin) * 2 + (p.dw[p.ph[q].tap0 + t] - dwmin);
        m1 = m1 && u == uid1[k];
        m2 = m2 && u == uid2[k];
      }
    if (!m1 && !m2) WS_REJECT(14);
    geom = m1 ? 1 : 2;
  }
  nt = 0;
  for (int q = 0; q < p.nphase; q++)
    for (int t = 0; t < p.ph[q].ntaps; t++, nt++) {
      const int rh = p.dh[p.ph[q].tap0 + t] - dhmin, rw = p.dw[p.ph[q].tap0 + t] - dwmin;
      const int u = geom == 0 ? nt : rh * 2 + rw;
      if (geom == 0 && (rh != nt / 3 || rw != nt % 3)) WS_REJECT(13);
      (void)u;
    }
  // tile: TW x TH base pixels with PG * FPW fragments; the patch must fit a slot plane
  const int max_rows = (KC == 2 ? WsSlot<2>::B : WsSlot<4>::B) / KC / 64;
  int bestTH = 0, bestTW = 0, bestF = 0;
  double best_halo = 1e30;
  for (int TW = 16; TW <= 64; TW *= 2) {
    if (TW > 16 && TW / 2 >= g.Wb) break;
    // (scatter forms keep two-fragment tiles: with the second destination the stores per tile double, and the counted wait holds
    //  2 x stores + copies in a 6-bit field; the dry run must promise the same tiling the y2 launch will use)
    for (int fpw = 2; fpw <= (geom >= 1 ? 2 : 4); fpw += 2) {
      const int nf = PG * fpw;
      if (nf % (TW / 16) != 0) continue;
      const int TH = nf / (TW / 16);
      const int PH = (TH - 1) * p.is + sh + 1, PWc = (TW - 1) * p.is + sw + 1;
      // (pitch padded so that is * PW is a multiple of 8 rows: the fragments of a wave then share one swizzle pattern)
      const int PW = p.is == 2 ? 4 * ((PWc + 3) / 4) : 8 * ((PWc + 7) / 8);
      const int rows = PH * PW;
      if (rows > max_rows || KC * ((rows + 15) / 16) > 48) continue;
      // a persistent grid of 256 workgroups wants several tiles each (and the ring two tiles ahead)
      const long nt_ = (long)((g.Wb + TW - 1) / TW) * ((g.Hb + TH - 1) / TH) * p.N;
      if (nt_ < 768) continue;
      // least halo per output pixel, then the bigger tile
      const double halo = (double)rows / (TH * TW);
      if (halo < best_halo - 1e-9 || (halo < best_halo + 1e-9 && TH * TW > bestTH * bestTW)) {
        best_halo = halo; bestTH = TH; bestTW = TW; bestF = fpw;
      }
    }
  }
  if (bestTH == 0) WS_REJECT(15);
  g.TH = bestTH; g.TW = bestTW;
  const int PH = (g.TH - 1) * p.is + sh + 1, PWc = (g.TW - 1) * p.is + sw + 1;
  g.PW = p.is == 2 ? 4 * ((PWc + 3) / 4) : 8 * ((PWc + 7) / 8);
  g.PWh = g.PW / 2;
  g.PWc = PWc;
  g.nrows = PH * g.PW;
  g.ncp = (g.nrows + 15) / 16;
  g.tiles_w = (g.Wb + g.TW - 1) / g.TW;
  g.tiles_hw = g.tiles_w * ((g.Hb + g.TH - 1) / g.TH);
  g.ntiles = g.tiles_hw * p.N;
  const int NU = geom == 0 ? 9 : 4;
  for (int u = 0; u < NU; u++) {
    const int rh = geom == 0 ? u / 3 : u / 2, rw = geom == 0 ? u % 3 : u % 2;
    g.qoff[u] = p.is == 2 ? rh * g.PW + (rw & 1) * g.PWh + (rw >> 1) : rh * g.PW + rw;
  }
  if (y2 && geom == 0) WS_REJECT(17);
  if (dry) return geom >= 1 ? 100 : 103;            // (100: honours IgemmParams::y2 -- mt_igemm_would_persist)
  const int grid = g.ntiles < 256 ? g.ntiles : 256;
  const bool st = p.stats != nullptr;
#define WS_LAUNCH(GE, C, K, F, S) hipLaunchKernelGGL((igemm_wsreg_kernel<GE, C, K, F, S>), dim3(grid), dim3(512), 0, s, p, g)
  if (geom == 0 && KC == 2) {
    if (bestF == 2) { if (st) WS_LAUNCH(0, 2, 2, 2, true); else WS_LAUNCH(0, 2, 2, 2, false); }
    else { if (st) WS_LAUNCH(0, 2, 2, 4, true); else WS_LAUNCH(0, 2, 2, 4, false); }
  } else if (geom == 0 && KC == 4) {
    if (bestF == 2) { if (st) WS_LAUNCH(0, 1, 4, 2, true); else WS_LAUNCH(0, 1, 4, 2, false); }
    else { if (st) WS_LAUNCH(0, 1, 4, 4, true); else WS_LAUNCH(0, 1, 4, 4, false); }
  } else if (geom == 1 && KC == 2) {
    if (y2) hipLaunchKernelGGL((igemm_wsreg_kernel<1, 2, 2, 2, false, true>), dim3(grid), dim3(512), 0, s, p, g); else WS_LAUNCH(1, 2, 2, 2, false);
  } else if (geom == 1) {
    if (y2) hipLaunchKernelGGL((igemm_wsreg_kernel<1, 1, 4, 2, false, true>), dim3(grid), dim3(512), 0, s, p, g); else WS_LAUNCH(1, 1, 4, 2, false);
  } else if (KC == 2) {
    if (y2) hipLaunchKernelGGL((igemm_wsreg_kernel<2, 2, 2, 2, false, true>), dim3(grid), dim3(512), 0, s, p, g); else WS_LAUNCH(2, 2, 2, 2, false);
  } else {
    if (y2) hipLaunchKernelGGL((igemm_wsreg_kernel<2, 1, 4, 2, false, true>), dim3(grid), dim3(512), 0, s, p, g); else WS_LAUNCH(2, 1, 4, 2, false);
  }
#undef WS_LAUNCH
  MT_LAUNCH_CHECK();
  __atomic_fetch_add(&g_ws_launches, 1, __ATOMIC_RELAXED);
  return 0;
}
