// Persistent software-pipelined gather-GEMM (bf16) for launches with several 128-pixel tiles per workgroup slot
// (same math and parameter block as igemm_kernel in conv_kernels.hip; chosen by launch_igemm_t there).  gfx950 only.
//
// Why: in-kernel stamps of the 2-stage kernel on the mid-size layers (tools/stamp_layer.py; 3x3 64->128 on 128x128,
// N=16: 2048 tiles, 4 per slot) show every tile spending 3.8k cycles in index set-up, 4.8k waiting for its first
// k-step, 16.5k in the k loop (MFMA-saturated with two blocks per CU) and 11.4k in the epilogue: all blocks run in
// lockstep, so all of them store their 32 KiB tile at the same moment (a 16 MiB burst against ~3 TB/s of HBM write
// bandwidth) while the matrix pipes idle.  Here a workgroup walks a contiguous run of tiles and
//   * issues the first k-step of tile t+1 during the last k-step of tile t (no exposed first-load latency),
//   * computes tile t+1's gather coordinates once, inside that k-step (float-reciprocal divisions; the filter-tap
//     table and the bias vector sit in LDS for the whole launch, so no global load is waited for between tiles),
//   * keeps tile t's packed bf16 outputs in registers and stores them two 16-byte stores per k-step during
//     tile t+1: the write traffic is spread over the k loop instead of arriving as a burst.  The stores are raw
//     buffer stores (out-of-range offset = dropped), so their number is constant and the k-step barrier waits with a
//     COUNTED s_waitcnt vmcnt(n) for the LDS-DMA copies only, never for the stores behind them.
#include "conv_device.h"
#include <type_traits>
#include <stdlib.h>

template <int N>
__device__ __forceinline__ void pk_wait_vm() {
  asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}
// wave-uniform `young` = vector-memory operations this wave issued after its last LDS-DMA copy
__device__ __forceinline__ void pk_wait_young(int young) {
  switch (young) {
    case 0: pk_wait_vm<0>(); break;
    case 1: pk_wait_vm<1>(); break;
    case 2: pk_wait_vm<2>(); break;
    case 3: pk_wait_vm<3>(); break;
    case 4: pk_wait_vm<4>(); break;
    case 5: pk_wait_vm<5>(); break;
    case 6: pk_wait_vm<6>(); break;
    case 7: pk_wait_vm<7>(); break;
    case 8: pk_wait_vm<8>(); break;
    case 9: pk_wait_vm<9>(); break;
    case 10: pk_wait_vm<10>(); break;
    case 11: pk_wait_vm<11>(); break;
    case 12: pk_wait_vm<12>(); break;
    case 13: pk_wait_vm<13>(); break;
    case 14: pk_wait_vm<14>(); break;
    case 15: pk_wait_vm<15>(); break;
    case 16: pk_wait_vm<16>(); break;
    default: pk_wait_vm<0>(); break;       // (never more than 16: waiting for everything is always correct)
  }
}
// exact floor(m / d) for 0 <= m < 2^24, 0 < d < 2^24 with inv = 1.0f / d (one correction step each way)
__device__ __forceinline__ int pk_div(int m, int d, float inv) {
  int q = (int)((float)m * inv);
  int r = m - q * d;
  q = r < 0 ? q - 1 : q;
  r = r < 0 ? r + d : r;
  q = r >= d ? q + 1 : q;
  return q;
}
__device__ __forceinline__ void pk_barrier() {
  __builtin_amdgcn_sched_barrier(0);
  __builtin_amdgcn_s_barrier();
  __builtin_amdgcn_sched_barrier(0);
}

// output channels whose bias fits the LDS copy (the 64-channel geometry keeps its LDS under a third of the CU's)
constexpr int persist_max_co(int WT) { return WT == 128 ? 2048 : 128; }

// AUX: cache policy bits of the output stores (0 = default, 2 = nt: streaming)
// DUAL: IgemmParams::y2 is set -- every output chunk is stored twice, once per destination, with the lanes that
// belong to the other one out of range (a buffer store is dropped there): the store count stays a constant
template <int WT, int BPC, int AUX, bool DUAL>
__global__ __launch_bounds__(256, BPC) void igemm_persist_kernel(const IgemmParams p, const int total, const int grp,
                                                                 const unsigned w_total, const unsigned y_total) {
  constexpr int PT = 128;
  constexpr int WC = 64, WP = WT >= 128 ? 64 : 32;
  constexpr int NWP = PT / WP;
  constexpr int FC = WC / 16, FP = WP / 16;
  constexpr int WLD = WT / 32;                 // weight-tile copies per wave per k-step
  constexpr int NPIECE = WLD + 4;              // copies per wave per k-step
  constexpr int NST = (FC / 2) * FP;           // 16-byte stores per lane per tile
  constexpr int SPK = 2;                       // held stores issued per k-step
  constexpr int SW = WT * 8, SX = PT * 8;      // u32x4 per stage
  constexpr unsigned OOB = 0x80000000u;
  constexpr int MT_PERSIST_MAX_CO = persist_max_co(WT);
  static_assert(WT == 128 || WT == 64, "tile geometries of the persistent kernel");
  constexpr int SMUL = DUAL ? 2 : 1;          // store instructions per held chunk
  static_assert(NST % SPK == 0 && NST * SMUL <= 16, "store groups / counted-wait range");

  // ONE shared array (separate __shared__ objects next to an LDS-DMA target make hipcc drain vmcnt):
  // 2 stages of weight tile, 2 stages of pixel tile, the filter-tap table of the launch, the bias vector
  __shared__ u32x4 smem[2 * SW + 2 * SX + MT_MAX_TAPS / 4 + MT_PERSIST_MAX_CO / 4 + PT + PT / 2 + (DUAL ? PT / 2 : 0)];
  u32x4* const sWb = smem;
  u32x4* const sXb = smem + 2 * SW;
  int* const sTap = reinterpret_cast<int*>(smem + 2 * SW + 2 * SX);                           // [MT_MAX_TAPS]
  float* const sBias = reinterpret_cast<float*>(smem + 2 * SW + 2 * SX + MT_MAX_TAPS / 4);   // [MT_PERSIST_MAX_CO]
  u32x2* const sRow = reinterpret_cast<u32x2*>(smem + 2 * SW + 2 * SX + MT_MAX_TAPS / 4 + MT_PERSIST_MAX_CO / 4);   // [2][PT]
  unsigned* const sOut = reinterpret_cast<unsigned*>(sRow + 2 * PT);                           // [2][PT]
  unsigned* const sOut2 = sOut + 2 * PT;                                                       // [2][PT] (DUAL: offsets into y2)

  const int tid = threadIdx.x;
  const int lane = tid & 63, wv = tid >> 6;
  const int fr = lane & 15, fg = lane >> 4;
  const int wcI = wv / NWP, wpI = wv % NWP;
  const int wvu = __builtin_amdgcn_readfirstlane(wv);
  const int r0 = tid >> 3;
  const int c = (tid & 7) ^ (r0 & 7);
  const int nWT = (p.CoRows + WT - 1) / WT;

  // ---- persistent schedule: blocks b, b+8, ... share an XCD (round-robin dispatch) and its L2; each XCD owns a
  // contiguous eighth of the tiles.  Inside it tiles go out in groups of G consecutive tiles (the sub-pixel phases /
  // weight tiles of ONE pixel tile: a block re-reads that input region G times from L2) and the XCD's blocks take
  // consecutive groups, round after round: at any moment they work on one contiguous input region.  (A contiguous
  // run of tiles per block put the 64 blocks of an XCD a power-of-two stride apart: their lines fell into the same
  // L2 sets and half of all requests missed -- rocprofv3 TCC_HIT/TCC_MISS on the 128->64 up-convolution.) ----
  const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3;
  const int nslot = ((int)gridDim.x - xcd + 7) >> 3;
  const int tq = total >> 3, tr = total & 7;
  const int lo = xcd * tq + (xcd < tr ? xcd : tr);
  const int cnt = tq + (xcd < tr ? 1 : 0);
  const int G = grp;
  // this block's tiles: (round * nslot + slot) * G + j,  j = 0 .. G-1, round = 0, 1, ...
  auto tile_of = [&](int round, int j) { return (round * nslot + slot) * G + j; };
  auto advance = [&](int& round, int& j) { j++; if (j == G) { j = 0; round++; } };
  int it = tile_of(0, 0);
  if (it >= cnt) return;
  int n_round = 0, n_j = 0;                                   // ... and the one after it
  advance(n_round, n_j);
  int it_next = tile_of(n_round, n_j);

  // ---- staging state of the tile whose copies are being issued.  Per-row coordinates live in LDS tables written
  // once per tile by one thread per pixel row (the staging rows and the output rows are the same 128 pixels):
  //   sRow[par][r] = {(ho*is << 16) | wo*is, byte offset of image n}   (second word ~0u: row outside the problem)
  //   sOut[par][r] = byte offset of the output pixel (OOB: nothing to store)
  unsigned wo_base = 0;            // weight offset of this lane's row of copy 0; copy j is s_wdelta * j further
  unsigned xo32[4], xokm = 0;
  int tap = 0, cq = 0;
  int s_ntaps = 0, s_nk = 0, s_tap0 = 0, s_par = 0, s_cob = 0;
  unsigned s_wdelta = 0;
  const int step_t = 8 / p.cpc, step_r = 8 % p.cpc;
  const float inv_cpc = 1.0f / (float)p.cpc;
  // weight-tile row of this lane in copy 0 -- 16-byte epilogue stores: within each 32-row fragment pair LDS row
  // (a&1)*16 + r holds channel (r>>2)*8 + (a&1)*4 + (r&3) (as in igemm_kernel); copy j holds rows 32 j further
  const int rch0 = (((r0 & 15) >> 2) << 3) | (((r0 >> 4) & 1) << 2) | (r0 & 3);

  const __amdgpu_buffer_rsrc_t rsx = __builtin_amdgcn_make_buffer_rsrc((void*)p.x, 0, p.x_bytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t rsw = __builtin_amdgcn_make_buffer_rsrc((void*)p.w, 0, w_total, 0x00020000);
  const __amdgpu_buffer_rsrc_t rsy = __builtin_amdgcn_make_buffer_rsrc((void*)p.y, 0, y_total, 0x00020000);
  const __amdgpu_buffer_rsrc_t rsy2 = __builtin_amdgcn_make_buffer_rsrc(
      (void*)(DUAL ? p.y2 : p.y), 0, DUAL ? (unsigned)p.N * (unsigned)(p.y2H * p.y2W) * (unsigned)(p.Co * 2) : 0u, 0x00020000);
  typedef __attribute__((address_space(3))) void* lds_ptr;

  // launch-wide tables
  for (int t = __builtin_amdgcn_readfirstlane(tid >> 6); t < MT_MAX_TAPS; t += 4) {       // (wave-uniform index: scalar loads)
    const int v = (tap_dh(p, t) << 16) | (tap_dw(p, t) & 0xffff);
    if ((tid & 63) == 0) sTap[t] = v;
  }
  for (int i = tid; i < p.Co + 8 && i < MT_PERSIST_MAX_CO; i += 256)
    sBias[i] = (p.bias != nullptr && i < p.nbias) ? p.bias[i] : 0.f;

  // tile `t_idx` of this XCD's range: phase, tile coordinates, row tables (a barrier must follow before retap())
  auto setup = [&](int t_idx, int par) {
    int wg = lo + t_idx;
    int phi = 0;
    if (p.interleave) {
      phi = wg % p.nphase;
      wg = wg / p.nphase + p.ph[phi].blk0;
    } else {
      for (int i = 1; i < p.nphase; i++) phi = (wg >= p.ph[i].blk0) ? i : phi;
    }
    const IgemmPhase& ph = p.ph[phi];
    wg -= ph.blk0;
    const int wt = wg % nWT, pt = wg / nWT;
    s_ntaps = ph.ntaps;
    s_nk = (s_ntaps * p.cpc + 7) >> 3;
    s_tap0 = ph.tap0;
    s_par = par;
    s_cob = wt * WT;
    s_wdelta = 32u * (unsigned)ph.wrow * 16u;
    if (tid < PT) {
      const int ph_Wo = ph.Wo;
      const int HoWo = ph.Ho * ph_Wo;
      const float inv_hw = 1.0f / (float)HoWo, inv_w = 1.0f / (float)ph_Wo;
      const int m = pt * PT + tid;
      u32x2 rowv = {0u, 0xffffffffu};
      unsigned yo = OOB, yo2 = OOB;
      if (m < ph.M) {
        const int n = pk_div(m, HoWo, inv_hw);
        const int rem = m - n * HoWo;
        const int ho = pk_div(rem, ph_Wo, inv_w);
        const int wo = rem - ho * ph_Wo;
        rowv[0] = ((unsigned)(ho * p.is) << 16) | (unsigned)(wo * p.is);
        rowv[1] = (unsigned)n * (unsigned)(p.Hi * p.Wi) * (unsigned)p.Cib;
        const int oh = ho * p.os + ph.oh0, ow = wo * p.os + ph.ow0;
        if ((unsigned)oh < (unsigned)p.Hout && (unsigned)ow < (unsigned)p.Wout)
          yo = ph.y_off + (unsigned)((n * p.Hout + oh) * p.Wout + ow) * (unsigned)(p.Co * 2);
        if constexpr (DUAL) {
          const int ih = oh - p.y2P, iw = ow - p.y2P;
          if (yo != OOB && (unsigned)ih < (unsigned)p.y2H && (unsigned)iw < (unsigned)p.y2W) {
            yo2 = (unsigned)((n * p.y2H + ih) * p.y2W + iw) * (unsigned)(p.Co * 2);
            yo = OOB;
          }
        }
      }
      sRow[par * PT + tid] = rowv;
      sOut[par * PT + tid] = yo;
      if constexpr (DUAL) sOut2[par * PT + tid] = yo2;
    }
    tap = pk_div(c, p.cpc, inv_cpc);
    cq = c - tap * p.cpc;
    // (rows past CoRows read whatever follows in the pack, or zeros past its end: their accumulator rows are never stored)
    wo_base = ph.w_off + ((unsigned)(wt * WT + rch0) * (unsigned)ph.wrow + (unsigned)c) * 16u;
  };

  auto retap = [&]() {
    xokm = 0;
#pragma unroll
    for (int i = 0; i < 4; i++) xo32[i] = 0xfffffff0u;
    if (tap < s_ntaps) {
      const int t = sTap[s_tap0 + tap];
      const int dh = t >> 16, dw = (int)(short)(t & 0xffff);
#pragma unroll
      for (int i = 0; i < 4; i++) {
        const u32x2 rowv = sRow[s_par * PT + r0 + 32 * i];
        int hi = (int)(rowv[0] >> 16) + dh, wi = (int)(rowv[0] & 0xffffu) + dw;
        bool ok = rowv[1] != 0xffffffffu;
        if (p.pad_mode == MT_PAD_REFLECT) {
          hi = hi < 0 ? -hi : hi;
          hi = hi >= p.Hi ? 2 * (p.Hi - 1) - hi : hi;
          wi = wi < 0 ? -wi : wi;
          wi = wi >= p.Wi ? 2 * (p.Wi - 1) - wi : wi;
        } else {
          ok = ok && ((unsigned)hi < (unsigned)p.Hi) && ((unsigned)wi < (unsigned)p.Wi);
          hi = ok ? hi : 0;
          wi = ok ? wi : 0;
        }
        // invalid lanes get an out-of-range offset: the buffer load returns zeros for them
        xo32[i] = ok ? rowv[1] + (unsigned)(hi * p.Wi + wi) * (unsigned)p.Cib + (unsigned)cq * 16u : 0xfffffff0u;
        xokm |= (ok ? 1u : 0u) << i;
      }
    }
  };
  // (no K-tail check on the weight side: there the pixel operand is zero, and reading into the next pack row only
  // multiplies finite weights by 0)
  auto issue_piece = [&](int buf, int j) {
    if (j < WLD) {
      char* lw = reinterpret_cast<char*>(sWb + buf * SW);
      // (per-lane offset, not the scalar offset operand: the hardware range check covers the per-lane part only)
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rsw, (lds_ptr)(lw + (wvu * 8 + 32 * j) * 128), 16,
                                               wo_base + s_wdelta * (unsigned)j, 0, 0, 0);
    } else {
      const int i = j - WLD;
      char* lx = reinterpret_cast<char*>(sXb + buf * SX);
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rsx, (lds_ptr)(lx + (wvu * 8 + 32 * i) * 128), 16, xo32[i], 0, 0, 0);
    }
  };
  auto issue_end = [&]() {
    wo_base += 128u;
    const int otap = tap;
    tap += step_t;
    cq += step_r;
    if (cq >= p.cpc) { cq -= p.cpc; tap++; }
    if (tap != otap) {
      retap();
    } else {
#pragma unroll
      for (int i = 0; i < 4; i++) xo32[i] += ((xokm >> i) & 1u) ? 128u : 0u;
    }
  };

  // ---- outputs of the finished tile, held in registers until the next tile's k-steps store them ----
  u32x4 held[FC / 2][FP];
  unsigned hoff[FP], hoff2[FP], hco[FC / 2];
  int held_grp = NST / SPK;          // next group of SPK stores to issue (NST / SPK = nothing held)
  auto store_one = [&](int s) {      // s is a compile-time constant after unrolling
    const int sp = s / FP, b = s % FP;
    const unsigned o = ((hoff[b] | hco[sp]) & OOB) ? OOB : hoff[b] + hco[sp];
    __builtin_amdgcn_raw_buffer_store_b128(held[sp][b], rsy, o, 0, AUX);
    if constexpr (DUAL) {
      const unsigned o2 = ((hoff2[b] | hco[sp]) & OOB) ? OOB : hoff2[b] + hco[sp];
      __builtin_amdgcn_raw_buffer_store_b128(held[sp][b], rsy2, o2, 0, AUX);
    }
  };
  auto store_group = [&](int g) {
#pragma unroll
    for (int gg = 0; gg < NST / SPK; gg++)
      if (g == gg) {
#pragma unroll
        for (int s = gg * SPK; s < gg * SPK + SPK; s++) store_one(s);
      }
  };

  f32x4 acc[FC][FP];
#pragma unroll
  for (int a = 0; a < FC; a++)
#pragma unroll
    for (int b = 0; b < FP; b++) acc[a][b] = f32x4{0.f, 0.f, 0.f, 0.f};

  const float neg_slope = p.act == MT_ACT_RELU ? 0.f : (p.act == MT_ACT_LRELU ? p.slope : 1.f);
  int young = 0;                     // vector-memory operations issued after this wave's last LDS-DMA copy

  // one k-step: MFMAs on buffer `cur`; the copies of the NEXT stage (of this tile, or stage 0 of the next tile:
  // whatever the staging state describes) go out one per MFMA group, then one group of held stores
  auto kstep = [&](int cur) {
    // the 64-channel geometry has the registers to read BOTH halves of the k-step up front (the second half's LDS
    // latency hides under the first half's MFMAs); the 128-channel one reads each half right before its MFMAs
    constexpr int NB = 2;
    u32x4 wf[NB][FC], xf[NB][FP];
    const u32x4* sWs = sWb + cur * SW;
    const u32x4* sXs = sXb + cur * SX;
    auto read_frags = [&](int kc, int slotb) {
#pragma unroll
      for (int a = 0; a < FC; a++) {
        const int row = wcI * WC + a * 16 + fr;
        wf[slotb][a] = sWs[row * 8 + ((kc * 4 + fg) ^ (row & 7))];
      }
#pragma unroll
      for (int b = 0; b < FP; b++) {
        const int row = wpI * WP + b * 16 + fr;
        xf[slotb][b] = sXs[row * 8 + ((kc * 4 + fg) ^ (row & 7))];
      }
    };
    if constexpr (NB == 2) { read_frags(0, 0); read_frags(1, 1); }
#pragma unroll
    for (int kc = 0; kc < 2; kc++) {
      if constexpr (NB == 1) read_frags(kc, 0);
      const int fb = (NB == 2) ? kc : 0;
#pragma unroll
      for (int a = 0; a < FC; a++) {
        const int slot_i = kc * FC + a;
        if (slot_i < NPIECE) issue_piece(cur ^ 1, slot_i);
        if (slot_i == (NPIECE < 2 * FC ? NPIECE : 2 * FC - 1)) {
          // (the counted wait of the next k-step relies on the stores being issued AFTER the last copy: no
          //  scheduling across this point)
          __builtin_amdgcn_sched_barrier(0);
          if (held_grp < NST / SPK) {
            store_group(held_grp);
            held_grp++;
            young += SPK * SMUL;
          }
          __builtin_amdgcn_sched_barrier(0);
        }
        __builtin_amdgcn_s_setprio(1);
#pragma unroll
        for (int b = 0; b < FP; b++) mma_chunk<true>(acc[a][b], wf[fb][a], xf[fb][b]);
        __builtin_amdgcn_s_setprio(0);
      }
    }
    issue_end();
  };

  // ---- first tile: set-up, stage 0 ----
  setup(it, 0);
  __syncthreads();       // tables visible
  retap();
  int c_cob = s_cob, c_nk = s_nk, c_par = 0;   // the tile being computed
#pragma unroll
  for (int j = 0; j < NPIECE; j++) issue_piece(0, j);
  issue_end();

  // ONE loop over the k-steps of all tiles of this block (a single k-step body keeps the register allocation simple)
  int buf = 0, ks = 0;
  bool has_next = it_next < cnt;
  while (true) {
    // this wave's copies of stage `buf` have landed (the `young` operations behind them may still be in flight);
    // after the barrier every wave's have, and every wave is done reading buffer buf^1
    pk_wait_young(young);
    young = 0;
    pk_barrier();
    if (ks + 1 == c_nk && has_next) {
      // last k-step of this tile: the staging state moves on to the next tile, whose stage 0 is issued here
      setup(it_next, c_par ^ 1);
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      pk_barrier();
      retap();
    }
    // (the very last k-step of the block issues one more set of copies from its stale staging state into the idle
    // buffer: range-checked reads that nobody consumes)
    kstep(buf);
    buf ^= 1;
    ks++;
    if (ks < c_nk) continue;
    // ---- tile finished: stores of the previous tile that found no k-step (short reductions) go out now ----
    while (held_grp < NST / SPK) {
      store_group(held_grp);
      held_grp++;
      young += SPK * SMUL;
    }
    // bias + activation, pack to bf16, keep
#pragma unroll
    for (int sp = 0; sp < FC / 2; sp++) {
      const int co = c_cob + wcI * WC + sp * 32 + fg * 8;
      hco[sp] = co < p.Co ? (unsigned)co * 2u : OOB;
      float bv[8];
      {
        const int cb = co < MT_PERSIST_MAX_CO ? co : 0;        // (co >= Co is never stored)
        const f32x4 b0 = *reinterpret_cast<const f32x4*>(sBias + cb), b1 = *reinterpret_cast<const f32x4*>(sBias + cb + 4);
#pragma unroll
        for (int e = 0; e < 4; e++) { bv[e] = b0[e]; bv[4 + e] = b1[e]; }
      }
#pragma unroll
      for (int b = 0; b < FP; b++) {
        float v[8];
#pragma unroll
        for (int e = 0; e < 8; e++) {
          const float z = acc[2 * sp + (e >> 2)][b][e & 3] + bv[e];
          v[e] = z > 0.f ? z : z * neg_slope;      // none / ReLU / LeakyReLU (tanh launches take the per-tile kernel)
        }
        held[sp][b] = u32x4{pack2_bf16(v[0], v[1]), pack2_bf16(v[2], v[3]), pack2_bf16(v[4], v[5]),
                            pack2_bf16(v[6], v[7])};
        acc[2 * sp][b] = f32x4{0.f, 0.f, 0.f, 0.f};
        acc[2 * sp + 1][b] = f32x4{0.f, 0.f, 0.f, 0.f};
      }
    }
#pragma unroll
    for (int b = 0; b < FP; b++) {
      hoff[b] = sOut[c_par * PT + wpI * WP + b * 16 + fr];
      if constexpr (DUAL) hoff2[b] = sOut2[c_par * PT + wpI * WP + b * 16 + fr];
    }
    held_grp = 0;
    if (!has_next) break;
    it = it_next;
    advance(n_round, n_j);
    it_next = tile_of(n_round, n_j);
    has_next = it_next < cnt;
    ks = 0;
    c_cob = s_cob;
    c_nk = s_nk;
    c_par ^= 1;
  }
  // the last tile's outputs
#pragma unroll
  for (int s = 0; s < NST; s++) store_one(s);
  pk_wait_vm<0>();       // the trailing copies must have landed before the wave exits
}

static long g_persist_launches = 0;
static int g_persist_on = -1;      // -1: not read from the environment yet
static int persist_enabled() {
  if (g_persist_on < 0) g_persist_on = getenv("MT_IGEMM_PERSIST") ? (atoi(getenv("MT_IGEMM_PERSIST")) != 0) : 1;
  return g_persist_on;
}
long mt_stem_launches();            // stem_kernel.hip
int mt_stem_enable(int on);
long mt_patch_launches();           // conv_patch_kernel.hip
int mt_patch_enable(int on);
long mt_pipe_patch_launches();      // conv_pipe_patch_kernel.hip
int mt_pipe_patch_enable(int on);
long mt_wsreg_launches();           // conv_wsreg_kernel.hip
int mt_wsreg_enable(int on);
long mt_wgrad_rows_launches();      // wgrad_rows_kernel.hip
int mt_wgrad_rows_enable(int on);
extern "C" long mt_kernel_variant_launches(int which) {
  return which == 0 ? g_persist_launches : (which == 1 ? mt_stem_launches() : (which == 2 ? mt_patch_launches() : (which == 3 ? mt_pipe_patch_launches() : (which == 4 ? mt_wsreg_launches() : (which == 5 ? mt_wgrad_rows_launches() : -1)))));
}
static long g_variant_epoch = 0;
extern "C" long mt_kernel_variant_epoch(void) { return g_variant_epoch; }
extern "C" int mt_kernel_variant_enable(int which, int enable) {
  g_variant_epoch++;
  if (which == 1) return mt_stem_enable(enable != 0);
  if (which == 2) return mt_patch_enable(enable);
  if (which == 3) return mt_pipe_patch_enable(enable != 0);
  if (which == 4) return mt_wsreg_enable(enable != 0);
  if (which == 5) return mt_wgrad_rows_enable(enable != 0);
  if (which != 0) return -1;
  const int prev = persist_enabled();
  g_persist_on = enable != 0;
  return prev;
}

// -> 0 launched, 1 error, -1 not applicable (the caller falls back to the per-tile kernels)
int launch_igemm_persist(IgemmParams& p, int WT, int total, hipStream_t s, bool dry) {
  if (!persist_enabled()) return -1;
  if (p.raw || p.stats != nullptr || (WT != 128 && WT != 64) || p.act == MT_ACT_TANH) return -1;
  const int cus = 256;
  const int bpc = WT == 128 ? 2 : 3;
  const int nb = cus * bpc;
  if (total < 2 * nb) return -1;      // (fewer than two tiles per block: nothing to pipeline across)
  unsigned long long w_total = 0;
  for (int i = 0; i < p.nphase; i++) {
    if (p.ph[i].ntaps < 1 || p.ph[i].M >= (1 << 24)) return -1;
    const unsigned long long e = (unsigned long long)p.ph[i].w_off + p.ph[i].w_bytes;
    w_total = e > w_total ? e : w_total;
  }
  const unsigned long long y_total = (unsigned long long)p.N * p.Hout * p.Wout * p.Co * 2ull;
  for (int i = 0; i < p.nphase; i++)
    if (p.ph[i].y_off != 0) return -1;
  if (w_total >= 0x7f000000ull || y_total >= 0x7f000000ull || p.x_bytes >= 0xf0000000u || p.cpc < 1 ||
      p.Co > persist_max_co(WT) - 8)
    return -1;
  static const int aux = getenv("MT_PK_STORE_AUX") ? atoi(getenv("MT_PK_STORE_AUX")) : 2;
  if (p.y2 != nullptr && (unsigned long long)p.N * p.y2H * p.y2W * p.Co * 2ull >= 0x7f000000ull) return -1;
  if (dry) return 100;
  // tiles of one pixel tile that are consecutive in the launch's tile order: its weight tiles, times its phases
  int grp = ((p.CoRows + WT - 1) / WT) * (p.interleave ? p.nphase : 1);
  grp = grp < 1 ? 1 : (grp > 16 ? 16 : grp);
  // ... but never so large that blocks stay without work: at least two groups per block
  while (grp > 1 && total / grp < 2 * nb) grp = (grp + 1) / 2;
  const unsigned wb = (unsigned)w_total, yb = (unsigned)y_total;
  const bool dual = p.y2 != nullptr;
#define PK_LAUNCH(W, B, A, D) hipLaunchKernelGGL((igemm_persist_kernel<W, B, A, D>), dim3(nb), dim3(256), 0, s, p, total, grp, wb, yb)
  if (WT == 128) {
    if (dual) { if (aux == 2) PK_LAUNCH(128, 2, 2, true); else PK_LAUNCH(128, 2, 0, true); }
    else { if (aux == 2) PK_LAUNCH(128, 2, 2, false); else PK_LAUNCH(128, 2, 0, false); }
  } else {
    if (dual) { if (aux == 2) PK_LAUNCH(64, 3, 2, true); else PK_LAUNCH(64, 3, 0, true); }
    else { if (aux == 2) PK_LAUNCH(64, 3, 2, false); else PK_LAUNCH(64, 3, 0, false); }
  }
#undef PK_LAUNCH
  MT_LAUNCH_CHECK();
  __atomic_fetch_add(&g_persist_launches, 1, __ATOMIC_RELAXED);
  return 0;
}
