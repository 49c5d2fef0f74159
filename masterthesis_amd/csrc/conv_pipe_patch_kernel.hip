// 256 x 256 ping-pong gather-GEMM with a PATCH-RESIDENT pixel operand (same math and parameter block as
// igemm_pipe_kernel<.., 256, 256, 512, 4, ..> in conv_pipe_kernel.hip, which it replaces for stride-1 gathers on maps whose
// width is a power of two: the 3x3 256->256 residual-block convolutions, 77 % of the step's FLOPs, forward and the
// interior of the data gradient).  gfx950 only.
//
// Why: in-kernel stamps of the ring kernel put its main loop at 1266 cycles per k-step against 1024 of pure
// MFMA issue; without any copies it runs 1065, the two weight copies alone cost +12, the two pixel copies alone +50, all
// four together +200 wherever they are placed -- the LDS-DMA instructions themselves (60-180 issue cycles each beside
// MFMAs) are the overhead, not their bytes.  The ring kernel already walks K slice-major (all taps of a 32-channel
// slice back to back) so that the tile's input patch stays in L2 between taps; here that patch -- (TH + span) x (W + span)
// pixels x 64 bytes, 25 KB for K1 -- is copied into LDS ONCE per slice (double-buffered: slice s+1 lands while slice
// s computes) and the pixel fragment of a tap is a ds_read_b128 at (pixel + tap offset).  Per k-step a wave then
// issues 2 weight copies plus at most one patch copy in 4 of 9 steps, instead of 4 copies; the gather-offset table is
// gone (the patch rows carry reflection / zero padding, resolved once per tile).
//
// FOLD (the data gradient of a reflection-padded 3x3 stride-1 convolution, pad 1, in ONE launch): dx = fold(dxp) adds the
// border ring of the padded-grid gradient back onto the pixels it was reflected from.  Written out per tap (a, b), with
// (dh, dw) = (1 - a, 1 - b) the offset at which dx[u][v] reads dy: the reflected contributions are reads at the MIRRORED
// offsets -- (-dh, dw) for the rows u = 1 (a = 0) and u = H-2 (a = 2), (dh, -dw) for the columns v = 1 (b = 0) and
// v = W-2 (b = 2), (-dh, -dw) where both hold -- multiplied with the SAME weight tap, so the fold is a sum of up to four
// patch cells in the pixel operand.  Those sums are built ONCE per slice as virtual cells behind the patch rows
// (VL[i] = P[i][0] + P[i][2] and VR per patch row; in the tiles that hold row 1 or H-2 also VT[j] = P[0][j] + P[2][j], VB
// and their corner cells; waves 0-3 do it in the memory phase of step 6 of the previous slice, when every wave's patch
// copies have landed), and in the main loop the affected lanes only read another address: no extra LDS read, no
// arithmetic beside the MFMAs.  (A first version added the mirrored cells in the memory phase -- dependent LDS reads
// and ~45 VALU per fragment-tap -- and ran 137 us against 101 us for interior + ring + fold: the memory phase must
// stay shorter than the partner group's 32 MFMAs.)  bf16: one extra rounding of the summed operand on those pixels;
// fp32: exact up to summation order.  The ring GEMM (23 us of latency), the workspace and the fold kernel go away.
//
// Everything else is igemm_pipe_kernel: 8 waves = two groups of four in ping-pong (memory phase / 32 MFMAs), 4-stage
// weight ring with a counted s_waitcnt vmcnt, permuted weight rows for 16-byte epilogue stores, fused per-(image,
// channel) statistics.  The count of copies per step varies (2 or 3), so the wait picks one of three immediates.
#include "conv_device.h"
#include <stdlib.h>
#include <type_traits>

#ifndef MT_PP_EPI_AUX
#define MT_PP_EPI_AUX 2      // cache policy of the output stores (2 = nt: streaming; A/B builds: 0)
#endif
#ifdef MT_PP_STAMPS
// diagnostic build only (tools/diag_build.sh NAME -DMT_PP_STAMPS conv_pipe_patch_kernel.hip): s_memtime stamps of wave 0 of every
// block -> tools/stamp_k1.py
__device__ unsigned long long mt_pp_stamp_buf[16 * 4096];
#define PP_STAMP(i) do { if (threadIdx.x == 0 && blockIdx.x < 4096) mt_pp_stamp_buf[blockIdx.x * 16 + (i)] = __builtin_amdgcn_s_memtime(); } while (0)
extern "C" int mt_debug_stamps(void* dst, size_t bytes) {
  return (int)hipMemcpyFromSymbol(dst, HIP_SYMBOL(mt_pp_stamp_buf), bytes < sizeof(mt_pp_stamp_buf) ? bytes : sizeof(mt_pp_stamp_buf));
}
#else
#define PP_STAMP(i) do {} while (0)
#endif

#ifndef PP_PATCH_AUX
#define PP_PATCH_AUX 0          // cache policy of the patch copies / weight copies (experiments: tools/diag_build.sh)
#endif
#ifndef PP_WEIGHT_AUX
#define PP_WEIGHT_AUX 0
#endif
constexpr int MT_PP_PCAP = 560;          // rows of a patch slot (4 x 130 for 128-wide maps, 8 x 68 for 5x5 taps on 64-wide ones)
constexpr int MT_PP_MAXTAPS = 25;

template <bool BF16>
__device__ __forceinline__ u32x4 pp_add_chunk(const u32x4& a, const u32x4& b) {
  constexpr int V = Elem<BF16>::V;
  float fa[V], fb[V];
  Elem<BF16>::unpack(a, fa);
  Elem<BF16>::unpack(b, fb);
#pragma unroll
  for (int e = 0; e < V; e++) fa[e] += fb[e];
  return Elem<BF16>::pack(fa);
}

// acc += a x b with the accumulator tied to ONE register quad.  (In the nine-step unrolled body hipcc otherwise writes every
// MFMA's result to a fresh quad -- v[2:5] = mfma(.., v[126:129]) -- and the rotating accumulators push 113 values into
// scratch, whose reloads are vector-memory operations with a vmcnt(0) in front: the LDS-DMA queue drains.)
template <bool BF16>
__device__ __forceinline__ void mma_inplace(f32x4& acc, const u32x4& a, const u32x4& b) {
  if constexpr (BF16) {
    asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+v"(acc) : "v"(a), "v"(b));
  } else {
#pragma unroll
    for (int s = 0; s < 4; s++) asm volatile("v_mfma_f32_16x16x4_f32 %0, %1, %2, %0" : "+v"(acc) : "v"(a[s]), "v"(b[s]));
  }
}

// TAPS9: nine taps (every 3x3 layer): the slice's nine steps are unrolled and the LDS address of every (tap, fragment)
// pair -- FOLD redirections included -- is computed once per tile into 36 registers.  A vector instruction in the memory
// phase waits for an issue slot beside the partner wave's prioritised MFMAs (~16 cycles each: 30 of them made the first
// FOLD version 35 % slower); with the addresses resident the memory phase is copies + LDS reads only.
template <bool BF16, bool FOLD, bool TAPS9>
__global__ __launch_bounds__(512) void igemm_pipe_patch_kernel(const IgemmParams p, const int PH, const int PW, const int dh0,
                                                               const int dw0, const int wo_shift) {
  constexpr int WT = 256, PT = 256, NT = 512, NS = 4, NW = 8;
  constexpr int WC = 128, WP = 64, NWP = PT / WP, FC = WC / 16, FP = WP / 16;
  constexpr int SZ = BF16 ? 2 : 4;
  constexpr int NWL = 2;                       // weight copies per wave per stage
  constexpr int STAGE = WT * 4;                // u32x4 per weight stage (256 rows x 64 B)
  constexpr int PCAP = MT_PP_PCAP;
  constexpr int NPW = FOLD ? 4 : (PCAP / 16 + NW - 1) / NW;       // patch copies per wave per slice (at most; FOLD: host)
  constexpr unsigned OOB = 0x80000000u;
  static_assert(NPW <= 5, "literal-sized arrays");
  static_assert(TAPS9 || !FOLD, "the in-operand fold is written for the nine-tap loop");
  static_assert((NS * STAGE + 2 * PCAP * 4) * 16 + 160 * 4 <= 160 * 1024, "LDS budget");

  // ONE shared array (a second __shared__ object next to an LDS-DMA target makes hipcc drain vmcnt)
  __shared__ u32x4 smem[NS * STAGE + 2 * PCAP * 4 + 40];
  u32x4* const sP = smem + NS * STAGE;
  // [0..31] patch row offset per tap (the generic loop; the nine-tap loop keeps its addresses in registers)
  int* const sQ = reinterpret_cast<int*>(smem + NS * STAGE + 2 * PCAP * 4);

  PP_STAMP(0);
  const int tid = threadIdx.x;
  const int lane = tid & 63, wv = tid >> 6;
  const int fr = lane & 15, fg = lane >> 4;
  const int wcI = wv / NWP, wpI = wv % NWP;
  const int wvu = __builtin_amdgcn_readfirstlane(wv);
  const int rsub = lane >> 2, csub = lane & 3;

  const IgemmPhase& ph = p.ph[0];
  const int ntaps = ph.ntaps, ph_Wo = ph.Wo, ph_M = ph.M;
  const int HoWo = ph.Ho * ph_Wo;
  const int nWT = p.CoRows / WT;
  const int wg = xcd_remap(blockIdx.x, gridDim.x);
  const int wt = wg % nWT, pt = wg / nWT;
  const int nsl = p.cpc >> 2;                  // 64-byte channel slices
  const int nk = ntaps * nsl;

  // (the nine-tap loop keeps its addresses in registers: no table, and no lane-indexed load of the tap offsets -- a vector load whose
  //  wait held wave 0, and with it every barrier of the prologue, for a memory latency at the very start of the tile)
  if constexpr (!TAPS9) {
    if (tid < 32) {
      const int dh = tid < ntaps ? (int)p.dh[tid] : 0, dw = tid < ntaps ? (int)p.dw[tid] : 0;        // (host: tap0 == 0)
      sQ[tid] = tid < ntaps ? (dh - dh0) * PW + (dw - dw0) : 0;
    }
  }

  // ---- the tile: 256 consecutive pixels of one image = TH full rows (host: 256 % Wo == 0, HoWo % 256 == 0) ----
  const int m0 = pt * PT;
  const int n0 = m0 / HoWo;
  const int ho0 = (m0 - n0 * HoWo) >> wo_shift;
  const int prows = PH * PW;
  const int npp = (prows + 15) >> 4;                                   // copies per slice
  const int nmine = npp > wvu ? (npp - wvu + NW - 1) / NW : 0;        // ... of which this wave issues
  // source offsets of this lane's patch rows (without the slice): reflection / zero padding resolved here
  unsigned xrow[5];
  {
    const unsigned nbase = (unsigned)n0 * (unsigned)(p.Hi * p.Wi) * (unsigned)p.Cib;
    const float inv_pw = 1.0f / (float)PW;
#pragma unroll
    for (int j = 0; j < NPW; j++) {
      const int r = (wvu + NW * j) * 16 + rsub;
      int py = (int)((float)r * inv_pw);
      int px = r - py * PW;
      py = px < 0 ? py - 1 : py; px = px < 0 ? px + PW : px;
      py = px >= PW ? py + 1 : py; px = px >= PW ? px - PW : px;
      int hi = ho0 + dh0 + py, wi = dw0 + px;
      bool ok = r < prows && m0 < ph_M;
      if (p.pad_mode == MT_PAD_REFLECT) {
        hi = hi < 0 ? -hi : hi;
        hi = hi >= p.Hi ? 2 * (p.Hi - 1) - hi : hi;
        wi = wi < 0 ? -wi : wi;
        wi = wi >= p.Wi ? 2 * (p.Wi - 1) - wi : wi;
      }
      ok = ok && ((unsigned)hi < (unsigned)p.Hi) && ((unsigned)wi < (unsigned)p.Wi);
      xrow[j] = ok ? nbase + (unsigned)(hi * p.Wi + wi) * (unsigned)p.Cib + (unsigned)((csub ^ ((r >> 1) & 3)) * 16) : OOB;
    }
  }
  // patch row of this lane's pixel of fragment b at tap offset 0
  int q0[FP];
#pragma unroll
  for (int b = 0; b < FP; b++) {
    const int tp = wpI * WP + b * 16 + fr;
    q0[b] = (tp >> wo_shift) * PW + (tp & (ph_Wo - 1));
  }

  // FOLD: which fragments hold the rows / columns that receive reflected contributions (wave-uniform per fragment), and
  // which lane holds the column
  // virtual cells of a patch slot (FOLD), as patch rows behind the real ones:
  //   VL[i], VR[i] (i = patch row): ONE lane of a border fragment reads such a cell in place of the patch row i PW + 3 (left:
  //   its pixel is column 1, tap dw = +1) resp. i PW + Wo - 2 (right) while the other 15 lanes read consecutive patch rows.  The
  //   cell therefore sits at a row with the SAME residue mod 8 as the row it stands in for (same banks, same swizzle key): with
  //   VL[i] = prows + i the substituted lane collided with its neighbours in every border fragment read of every step
  //   (SQ_LDS_BANK_CONFLICT 0 in the forward, 337 k in the data gradient; +10 k cycles per tile).  PW = Wo + 2 = 2 (mod 8): four
  //   consecutive i have four distinct residues -- odd ones on the left, even ones on the right -- so 8 rows hold the cells of 4
  //   patch rows for both sides.
  //   VT[j] = vT0 + j, then VT_L, VT_R;  VB[j] = vB0 + j, then VB_L, VB_R   (j = patch column; whole fragments read these)
  const int PH4 = (PH + 3) >> 2;
  const int vLb = (prows + 7) & ~7, vT0 = vLb + 8 * PH4, vB0 = vT0 + PW + 2;
  // (i PW + 3 and i PW + Wo - 2 mod 8 with PW = 2, Wo = 0 mod 8 -- host: shifts and masks only, these sit in the address set-up)
  auto vL = [&](int i) { return vLb + ((i >> 2) << 3) + ((2 * i + 3) & 7); };
  auto vR = [&](int i) { return vLb + ((i >> 2) << 3) + ((2 * i + 6) & 7); };
  const bool tileT = FOLD && ho0 <= 1 && 1 < ho0 + (PT >> wo_shift);                       // the tile holds row 1 / row H-2
  const bool tileB = FOLD && ho0 <= p.Hi - 2 && p.Hi - 2 < ho0 + (PT >> wo_shift);
  bool f_rT[FP], f_rB[FP], f_hasL[FP], f_hasR[FP], f_cL[FP], f_cR[FP];
  int f_ly[FP], f_lx[FP];
#pragma unroll
  for (int b = 0; b < FP; b++) {
    const int tpb = __builtin_amdgcn_readfirstlane(wpI) * WP + b * 16;     // (wave-uniform, and known to the compiler as such: the
                                                                           //  fragment's flags and cell rows stay on the scalar unit)
    const int u = ho0 + (tpb >> wo_shift), c0 = tpb & (ph_Wo - 1);
    f_ly[b] = tpb >> wo_shift;
    f_lx[b] = c0 + fr;
    f_rT[b] = FOLD && u == 1;
    f_rB[b] = FOLD && u == p.Hi - 2;
    f_hasL[b] = FOLD && c0 == 0;
    f_hasR[b] = FOLD && c0 + 16 == ph_Wo;
    f_cL[b] = FOLD && (c0 + fr) == 1;
    f_cR[b] = FOLD && (c0 + fr) == ph_Wo - 2;
  }

  // ---- weight staging (as in igemm_pipe_kernel) ----
  unsigned wo32[2];
#pragma unroll
  for (int i = 0; i < NWL; i++) {
    const int rs = 16 * (wvu + NW * i) + rsub;           // LDS row of the weight tile (fragment order)
    // channel held by that row: within each 32-row fragment pair, row (a&1)*16 + r <- channel (r>>2)*8 + (a&1)*4 + (r&3)
    const int rl = (rs & ~31) | ((((rs & 15) >> 2) << 3) | (((rs >> 4) & 1) << 2) | (rs & 3));
    const int row = wt * WT + rl;
    wo32[i] = row < p.CoRows ? ((unsigned)row * (unsigned)ph.wrow + (unsigned)(csub ^ ((rsub >> 1) & 3))) * 16u : OOB;
  }
  int tap_s = 0, sl_s = 0;      // wave-uniform: tap and slice of the NEXT weight stage to issue (slice-major K order)
  unsigned wk = 0;

  const __amdgpu_buffer_rsrc_t rsx = __builtin_amdgcn_make_buffer_rsrc((void*)p.x, 0, p.x_bytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t rsw = __builtin_amdgcn_make_buffer_rsrc((void*)(p.w + ph.w_off), 0, ph.w_bytes, 0x00020000);
  typedef __attribute__((address_space(3))) void* lds_ptr;
  char* const lds0 = reinterpret_cast<char*>(&smem[0]);
  char* const ldsP = reinterpret_cast<char*>(sP);

  auto issue_weights = [&](int slot) {
    // (past the end of K the offsets run past the row / the pack: range-checked zeros or finite weights that meet a
    //  stage nobody reads)
    char* base = lds0 + slot * (STAGE * 16);
#pragma unroll
    for (int j = 0; j < NWL; j++)
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rsw, (lds_ptr)(base + (wvu + NW * j) * 1024), 16, wo32[j] + wk, 0, 0, PP_WEIGHT_AUX);
    tap_s += 1;
    const bool wrap = tap_s >= ntaps;
    sl_s = wrap ? sl_s + 1 : sl_s;
    tap_s = wrap ? 0 : tap_s;
    wk = sl_s < nsl ? (unsigned)((tap_s * p.cpc + sl_s * 4) * 16) : 0x40000000u;     // past K: out of range
  };
  auto patch_piece = [&](int pslot, int j, int sl) {
    unsigned off = OOB;
#pragma unroll
    for (int jj = 0; jj < NPW; jj++) off = (j == jj) ? xrow[jj] : off;
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rsx, (lds_ptr)(ldsP + pslot * (PCAP * 64) + (wvu + NW * j) * 1024), 16,
                                             off + (unsigned)sl * 64u, 0, 0, PP_PATCH_AUX);
  };

  // FOLD: the virtual cells of patch slot `pslot` (its copies have landed and are visible), by the 256 threads of one
  // wave group: task = (cell, 16-byte chunk)
  auto cell_addr = [&](int row, int chunk) { return row * 4 + (chunk ^ ((row >> 1) & 3)); };
  auto build_virtual = [&](int pslot) {
    u32x4* const P = sP + pslot * (PCAP * 4);
    const int t = tid & 255;
    const int chunk = t & 3;
    const int iT0 = 1 - ho0 - dh0 - 1, iT2 = iT0 + 2;                  // patch rows of dy rows 0 and 2
    const int iB0 = (p.Hi - 1) - ho0 - dh0, iB2 = iB0 - 2;              // ... of dy rows H-1 and H-3
    const int jL0 = 0 - dw0, jL2 = 2 - dw0, jR0 = (ph_Wo - 1) - dw0, jR2 = (ph_Wo - 3) - dw0;    // patch columns of dy columns 0, 2, W-1, W-3
    const int ncell = 2 * PH + ((tileT || tileB) ? 2 * (PW + 2) : 0);
    for (int c = t >> 2; c < ncell; c += 64) {
      int dst, a0, a1, a2 = -1, a3 = -1;
      if (c < 2 * PH) {
        const bool right = c >= PH;
        const int i = right ? c - PH : c;
        dst = right ? vR(i) : vL(i);
        a0 = i * PW + (right ? jR0 : jL0);
        a1 = i * PW + (right ? jR2 : jL2);
      } else {
        const int cc = c - 2 * PH;
        const bool bot = cc >= PW + 2;
        const int j = bot ? cc - (PW + 2) : cc;
        if ((bot && !tileB) || (!bot && !tileT)) continue;
        const int r0 = bot ? iB0 : iT0, r2 = bot ? iB2 : iT2;
        dst = (bot ? vB0 : vT0) + j;
        if (j < PW) {
          a0 = r0 * PW + j;
          a1 = r2 * PW + j;
        } else {                                  // corner cells: columns {0, 2} (j == PW) or {W-1, W-3} (j == PW + 1) of both rows
          const int c0 = j == PW ? jL0 : jR0, c2 = j == PW ? jL2 : jR2;
          a0 = r0 * PW + c0; a1 = r2 * PW + c0; a2 = r0 * PW + c2; a3 = r2 * PW + c2;
        }
      }
      u32x4 v = pp_add_chunk<BF16>(P[cell_addr(a0, chunk)], P[cell_addr(a1, chunk)]);
      if (a2 >= 0) v = pp_add_chunk<BF16>(v, pp_add_chunk<BF16>(P[cell_addr(a2, chunk)], P[cell_addr(a3, chunk)]));
      P[cell_addr(dst, chunk)] = v;
    }
  };

  f32x4 acc[FC][FP];

  // (tap offsets of the nine-tap loop: read before any copy is in flight -- a dynamically indexed short in the kernel
  //  arguments is a vector load, and its wait would drain the LDS-DMA queue)
  int tdh[9], tdw[9];
  if constexpr (TAPS9) {
#pragma unroll
    for (int t = 0; t < 9; t++) {
      tdh[t] = (int)p.dh[t];            // (host: tap0 == 0 -- constant offsets into the kernel arguments: scalar loads)
      tdw[t] = (int)p.dw[t];
    }
  }
  // ---- prologue: patch of slice 0, weight stages 0 .. NS-2 ----
  __builtin_amdgcn_sched_barrier(0);
  PP_STAMP(1);
  for (int j = 0; j < nmine; j++) patch_piece(0, j, 0);
#pragma unroll
  for (int s = 0; s < NS - 1; s++) issue_weights(s);
  // (the accumulators are cleared and the fragment addresses computed while those copies are in flight)
  __builtin_amdgcn_sched_barrier(0);
#pragma unroll
  for (int a = 0; a < FC; a++)
#pragma unroll
    for (int b = 0; b < FP; b++) acc[a][b] = f32x4{0.f, 0.f, 0.f, 0.f};
  // TAPS9: LDS byte offset (from the start of the shared array, patch slot 0; slot 1 is a constant further, which fits
  // the DS instruction's 16-bit immediate) of this lane's chunk of fragment b at tap t
  int qa[9][FP];
  if constexpr (TAPS9) {
#pragma unroll
    for (int t = 0; t < 9; t++) {
      // (FOLD: the host admits the 3x3 window in the order of the data-gradient pack only -- dh = 1 - t / 3, dw = 1 - t % 3 -- so the
      //  border selects below fold at compile time: with run-time offsets they were ~850 vector instructions of the address set-up)
      const int dh = FOLD ? 1 - t / 3 : tdh[t], dw = FOLD ? 1 - t % 3 : tdw[t];
      const int qo = (dh - (FOLD ? -1 : dh0)) * PW + (dw - (FOLD ? -1 : dw0));
#pragma unroll
      for (int b = 0; b < FP; b++) {
        int q = q0[b] + qo;
        if constexpr (FOLD) {
          const bool rh = dh > 0 ? f_rT[b] : (dh < 0 ? f_rB[b] : false);
          const bool ch = dw > 0 ? f_hasL[b] : (dw < 0 ? f_hasR[b] : false);
          const int vrow = dh > 0 ? vT0 : vB0;
          if (rh) q = vrow + f_lx[b] + dw + 1;                                      // VT / VB cell of column v + dw
          if (ch) {
            const bool cl = dw > 0 ? f_cL[b] : f_cR[b];                             // this lane's pixel is the column
            const int ci = f_ly[b] + dh + 1;                                            // vL(ci) / vR(ci), the side chosen first
            const int alt = rh ? vrow + PW + (dw > 0 ? 0 : 1) : vLb + ((ci >> 2) << 3) + ((2 * ci + (dw > 0 ? 3 : 6)) & 7);
            q = cl ? alt : q;
          }
        }
        qa[t][b] = (NS * STAGE + q * 4 + (fg ^ ((q >> 1) & 3))) * 16;
      }
    }
  }

  PP_STAMP(6);
  asm volatile("s_waitcnt vmcnt(%0)" ::"n"((NS - 2) * NWL) : "memory");
  __syncthreads();              // sQ visible; every wave's patch rows and stage 0 have landed
  PP_STAMP(7);
  if constexpr (FOLD) {
    if (tid < 256) build_virtual(0);
    __syncthreads();
  }
  PP_STAMP(8);

  // ---- ping-pong main loop (see igemm_pipe_kernel): waves 0-3 and 4-7 alternate between a MEMORY phase (copies of
  // stage ks+NS-1 and of the next slice's patch, fragment reads of stage ks) and a COMPUTE phase (32 MFMAs) ----
  const int grp = wvu >> 2;
  if (grp) {
    __builtin_amdgcn_sched_barrier(0);
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_sched_barrier(0);
  }
  PP_STAMP(2);
  int slot = 0;                 // ring slot of stage ks
  int c_prev = NWL;             // copies issued in the previous memory phase
  if constexpr (TAPS9) {
    int sl_c = 0;               // slice being computed
    // one step: tap T (compile time) of slice sl_c, whose patch sits in slot PAR = sl_c & 1
    auto step = [&](auto T_c, auto PAR_c) {
      constexpr int T = decltype(T_c)::value, PAR = decltype(PAR_c)::value;
      int fill = slot - 1;
      fill = fill < 0 ? NS - 1 : fill;
      if constexpr (FOLD && T == 6) {
        // virtual cells of the NEXT slice's patch: its copies went out in steps 0 .. 3 of this slice and have landed, for
        // every wave, once group 0 enters step 6 (nmine <= 4 with FOLD: host).  First thing in the step, before the
        // fragments of this step occupy their registers.
#ifndef MT_PP_EXP_NOVIRT
        if (grp == 0 && sl_c + 1 < nsl) build_virtual(PAR ^ 1);
#endif
        __builtin_amdgcn_sched_barrier(0);
      }
      u32x4 wf[FC], xf[FP];
      int c_now = NWL;
      if constexpr (T < NPW) {
        // the next slice's patch: copy T in step T (read three or more steps after the last one, behind the counted wait)
        if (T < nmine && sl_c + 1 < nsl) {
          patch_piece(PAR ^ 1, T, sl_c + 1);
          c_now++;
        }
      }
      issue_weights(fill);
      __builtin_amdgcn_sched_barrier(0);
      {
        const u32x4* sWs = &smem[slot * STAGE];
#pragma unroll
        for (int a = 0; a < FC; a++) {
          const int row = wcI * WC + a * 16 + fr;
          wf[a] = sWs[row * 4 + (fg ^ ((row >> 1) & 3))];
        }
        static_assert(PCAP * 64 < 65536, "the second patch slot must be reachable through the DS offset field");
#pragma unroll
        for (int b = 0; b < FP; b++)
          xf[b] = *reinterpret_cast<const u32x4*>(reinterpret_cast<const char*>(smem) + qa[T][b] + PAR * (PCAP * 64));
      }
      {
        const int young = c_prev + c_now;      // 2 NWL .. 2 NWL + 2
        if (young == 2 * NWL) asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)" ::"n"(2 * NWL) : "memory");
        else if (young == 2 * NWL + 1) asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)" ::"n"(2 * NWL + 1) : "memory");
        else asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)" ::"n"(2 * NWL + 2) : "memory");
        c_prev = c_now;
      }
      __builtin_amdgcn_sched_barrier(0);
      __builtin_amdgcn_s_barrier();
      __builtin_amdgcn_sched_barrier(0);
      __builtin_amdgcn_s_setprio(1);
#pragma unroll
      for (int a = 0; a < FC; a++)
#pragma unroll
        for (int b = 0; b < FP; b++) mma_inplace<BF16>(acc[a][b], wf[a], xf[b]);
      __builtin_amdgcn_s_setprio(0);
      __builtin_amdgcn_sched_barrier(0);
      __builtin_amdgcn_s_barrier();
      __builtin_amdgcn_sched_barrier(0);
      slot = slot + 1 == NS ? 0 : slot + 1;
    };
#define PP_SLICE(PAR)                                                                                              \
  do {                                                                                                             \
    step(std::integral_constant<int, 0>{}, std::integral_constant<int, PAR>{});                                    \
    step(std::integral_constant<int, 1>{}, std::integral_constant<int, PAR>{});                                    \
    step(std::integral_constant<int, 2>{}, std::integral_constant<int, PAR>{});                                    \
    step(std::integral_constant<int, 3>{}, std::integral_constant<int, PAR>{});                                    \
    step(std::integral_constant<int, 4>{}, std::integral_constant<int, PAR>{});                                    \
    step(std::integral_constant<int, 5>{}, std::integral_constant<int, PAR>{});                                    \
    step(std::integral_constant<int, 6>{}, std::integral_constant<int, PAR>{});                                    \
    step(std::integral_constant<int, 7>{}, std::integral_constant<int, PAR>{});                                    \
    step(std::integral_constant<int, 8>{}, std::integral_constant<int, PAR>{});                                    \
  } while (0)
    for (; sl_c + 1 < nsl; sl_c += 2) {
      PP_SLICE(0);
      sl_c++;
      PP_SLICE(1);
      sl_c--;
    }
    if (sl_c < nsl) PP_SLICE(0);
#undef PP_SLICE
  } else {
  int tap_c = 0, sl_c = 0;      // tap / slice being computed
  int p_j = 0;                  // patch copies of slice sl_c + 1 issued so far
  for (int ks = 0; ks < nk; ks++) {
    int fill = slot - 1;
    fill = fill < 0 ? NS - 1 : fill;
    u32x4 wf[FC], xf[FP];
    // copies first: the texture path works on them while the LDS serves the fragment reads.  The next slice's patch
    // goes out one copy per step in the first steps of this slice (it is read two or more steps after the last one,
    // behind the counted wait below: ntaps >= nmine + 3, host)
    int c_now = NWL;
    if (p_j < nmine && sl_c + 1 < nsl) {
      patch_piece((sl_c + 1) & 1, p_j, sl_c + 1);
      p_j++;
      c_now++;
    }
    issue_weights(fill);
    __builtin_amdgcn_sched_barrier(0);
    {
      const u32x4* sWs = &smem[slot * STAGE];
      const u32x4* sPs = sP + (sl_c & 1) * (PCAP * 4);
      const int qo = sQ[tap_c];
#pragma unroll
      for (int a = 0; a < FC; a++) {
        const int row = wcI * WC + a * 16 + fr;
        wf[a] = sWs[row * 4 + (fg ^ ((row >> 1) & 3))];
      }
#pragma unroll
      for (int b = 0; b < FP; b++) {
        const int q = q0[b] + qo;
        xf[b] = sPs[q * 4 + (fg ^ ((q >> 1) & 3))];
      }
    }
    tap_c++;
    if (tap_c == ntaps) { tap_c = 0; sl_c++; p_j = 0; }
    // all but the copies of this and the previous memory phase have landed: stage ks+1 (and, at a slice boundary, the
    // next slice's patch) is in LDS before the barrier in front of anyone's reads
    {
      const int young = c_prev + c_now;      // 2 NWL .. 2 NWL + 2
      if (young == 2 * NWL) asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)" ::"n"(2 * NWL) : "memory");
      else if (young == 2 * NWL + 1) asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)" ::"n"(2 * NWL + 1) : "memory");
      else asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)" ::"n"(2 * NWL + 2) : "memory");
      c_prev = c_now;
    }
    __builtin_amdgcn_sched_barrier(0);
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_sched_barrier(0);
    __builtin_amdgcn_s_setprio(1);
#pragma unroll
    for (int a = 0; a < FC; a++)
#pragma unroll
      for (int b = 0; b < FP; b++) mma_chunk<BF16>(acc[a][b], wf[a], xf[b]);
    __builtin_amdgcn_s_setprio(0);
    __builtin_amdgcn_sched_barrier(0);
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_sched_barrier(0);
    slot = slot + 1 == NS ? 0 : slot + 1;
  }
  }
  if (!grp) {
    __builtin_amdgcn_sched_barrier(0);
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_sched_barrier(0);
  }
  PP_STAMP(3);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");     // the trailing copies must have landed before LDS is reused / the wave exits
  PP_STAMP(4);
#ifdef MT_PP_EXP_NOEPI
  {   // EXPERIMENT: no epilogue at all (the accumulators stay live through an impossible store)
    float t = 0.f;
#pragma unroll
    for (int a = 0; a < FC; a++)
#pragma unroll
      for (int b = 0; b < FP; b++) t += acc[a][b][0] + acc[a][b][1] + acc[a][b][2] + acc[a][b][3];
    if (t == 1.2345e-30f) p.y[threadIdx.x] = 1;
    PP_STAMP(5);
    return;
  }
#endif

  // ---- epilogue (igemm_pipe_kernel's): bias + activation, packed NHWC store, optional statistics ----
  char* yp[FP];
#pragma unroll
  for (int b = 0; b < FP; b++) {
    const int m = pt * PT + wpI * WP + b * 16 + fr;
    yp[b] = nullptr;
    if (m < ph_M) {
      const int n = m / HoWo;
      const int rem = m - n * HoWo;
      const int ho = rem >> wo_shift;
      const int wo = rem & (ph_Wo - 1);
      const int oh = ho * p.os + ph.oh0, ow = wo * p.os + ph.ow0;
      if ((unsigned)oh < (unsigned)p.Hout && (unsigned)ow < (unsigned)p.Wout)
        yp[b] = p.y + (((size_t)n * p.Hout + oh) * p.Wout + ow) * p.Co * SZ;
    }
  }
  const bool do_stats = p.stats != nullptr;
  float* red = reinterpret_cast<float*>(&smem[0]);      // [NWP][WT][2]
  if (do_stats) __syncthreads();
  // ---- the straight-line epilogue (conv_device.h: epilogue_perm; round 4) for everything but tanh, the norm-backward statistics
  // of the gradient (StatsLink, off by default) and outputs beyond the 32-bit buffer offsets, which keep the general code below
  const size_t ybytes = (size_t)p.N * p.Hout * p.Wout * p.Co * SZ;
  if (p.act != MT_ACT_TANH && p.bstat_x == nullptr && !(do_stats && p.addend != nullptr) && ybytes < 0x7f000000ull) {
    unsigned yo[FP];
    float vm[FP];
#pragma unroll
    for (int b = 0; b < FP; b++) {
      yo[b] = yp[b] != nullptr ? (unsigned)(yp[b] - p.y) : EPI_OOB;
      vm[b] = yp[b] != nullptr ? 1.f : 0.f;
    }
    const unsigned cob = (unsigned)(wt * WT + wcI * WC + fg * 8);
    float* const red_lane = red + (wpI * WT + wcI * WC + fg * 8) * 2;
    // (the forward with statistics and the data gradient with / without the skip gradient: three straight-line copies)
    if (do_stats)
      epilogue_perm<BF16, FC, FP, false, true, false, MT_PP_EPI_AUX>(acc, p.y, (unsigned)ybytes, p.bias, p.nbias, nullptr, p.act, p.slope, yo, vm, cob,
                                                      p.Co, red_lane, fr);
    else if (p.addend != nullptr)
      epilogue_perm<BF16, FC, FP, true, false, false, MT_PP_EPI_AUX>(acc, p.y, (unsigned)ybytes, p.bias, p.nbias, p.addend, p.act, p.slope, yo, vm,
                                                      cob, p.Co, red_lane, fr);
    else
      epilogue_perm<BF16, FC, FP, false, false, false, MT_PP_EPI_AUX>(acc, p.y, (unsigned)ybytes, p.bias, p.nbias, nullptr, p.act, p.slope, yo, vm,
                                                       cob, p.Co, red_lane, fr);
  } else
#pragma unroll
  for (int sp = 0; sp < FC / 2; sp++) {
    const int col = wcI * WC + sp * 32 + fg * 8;
    const int co = wt * WT + col;
    if (co >= p.Co) continue;
    float bv[8];
#pragma unroll
    for (int e = 0; e < 8; e++) bv[e] = (p.bias != nullptr && (co + e) < p.nbias) ? p.bias[co + e] : 0.f;
    float s1[8], s2[8];
#pragma unroll
    for (int e = 0; e < 8; e++) s1[e] = s2[e] = 0.f;
    const bool bstat = p.bstat_x != nullptr;
    float bsc[8], bsh[8];
    if (bstat) {
      const f32x4* q = reinterpret_cast<const f32x4*>(p.bstat_scale + (size_t)n0 * p.Co + co);
      const f32x4* r = reinterpret_cast<const f32x4*>(p.bstat_shift + (size_t)n0 * p.Co + co);
      const f32x4 q0 = q[0], q1 = q[1], r0 = r[0], r1 = r[1];
#pragma unroll
      for (int e = 0; e < 4; e++) { bsc[e] = q0[e]; bsc[4 + e] = q1[e]; bsh[e] = r0[e]; bsh[4 + e] = r1[e]; }
    }
#pragma unroll
    for (int b = 0; b < FP; b++) {
      if (yp[b] == nullptr) continue;
      float v[8];
      float ad[8];
#pragma unroll
      for (int e = 0; e < 8; e++) ad[e] = 0.f;
      if (p.addend != nullptr) {
        // (same offset as the store: the addend has the output's layout)
        const char* ap = p.addend + (yp[b] - p.y) + (size_t)co * SZ;
        if constexpr (BF16) {
          Elem<true>::unpack(*reinterpret_cast<const u32x4*>(ap), ad);
        } else {
          const f32x4 a0 = *reinterpret_cast<const f32x4*>(ap), a1 = *reinterpret_cast<const f32x4*>(ap + 16);
#pragma unroll
          for (int e = 0; e < 4; e++) { ad[e] = a0[e]; ad[4 + e] = a1[e]; }
        }
      }
#pragma unroll
      for (int e = 0; e < 8; e++) v[e] = act_apply(acc[2 * sp + (e >> 2)][b][e & 3] + bv[e], p.act, p.slope) + ad[e];
      if (bstat) {
        // statistics of the normalisation backward that consumes this gradient: g = v * act'(scale * x + shift)
        float xx[8];
        const char* xp = p.bstat_x + (yp[b] - p.y) + (size_t)co * SZ;
        if constexpr (BF16) {
          Elem<true>::unpack(*reinterpret_cast<const u32x4*>(xp), xx);
        } else {
          const f32x4 a0 = *reinterpret_cast<const f32x4*>(xp), a1 = *reinterpret_cast<const f32x4*>(xp + 16);
#pragma unroll
          for (int e = 0; e < 4; e++) { xx[e] = a0[e]; xx[4 + e] = a1[e]; }
        }
#pragma unroll
        for (int e = 0; e < 8; e++) {
          const float gg = v[e] * act_grad_z(bsc[e] * xx[e] + bsh[e], p.bstat_act, p.bstat_slope);
          s1[e] += gg;
          s2[e] += gg * xx[e];
        }
      } else {
#pragma unroll
        for (int e = 0; e < 8; e++) { s1[e] += v[e]; s2[e] += v[e] * v[e]; }
      }
      if constexpr (BF16) {
        u32x4 o = {pack2_bf16(v[0], v[1]), pack2_bf16(v[2], v[3]), pack2_bf16(v[4], v[5]), pack2_bf16(v[6], v[7])};
        __builtin_nontemporal_store(o, reinterpret_cast<u32x4*>(yp[b] + (size_t)co * 2));
      } else {
        f32x4 o0 = {v[0], v[1], v[2], v[3]}, o1 = {v[4], v[5], v[6], v[7]};
        *reinterpret_cast<f32x4*>(yp[b] + (size_t)co * 4) = o0;
        *reinterpret_cast<f32x4*>(yp[b] + (size_t)co * 4 + 16) = o1;
      }
    }
    if (do_stats) {
#pragma unroll
      for (int e = 0; e < 8; e++) {
        const float t1 = row16_sum(s1[e]), t2 = row16_sum(s2[e]);
        if (fr == 0) {
          red[((wpI * WT) + col + e) * 2] = t1;
          red[((wpI * WT) + col + e) * 2 + 1] = t2;
        }
      }
    }
  }
  if (do_stats) {
    __syncthreads();
    if (m0 < ph_M) {
      for (int idx = tid; idx < WT * 2; idx += NT) {
        const int col = idx >> 1;
        if (wt * WT + col < p.Co) {
          float t = 0.f;
#pragma unroll
          for (int w = 0; w < NWP; w++) t += red[(w * WT + col) * 2 + (idx & 1)];
          atomicAdd(p.stats + ((size_t)n0 * p.Co + wt * WT + col) * 2 + (idx & 1), t);
        }
      }
    }
  }
  PP_STAMP(5);
}

static long g_pp_launches = 0;
static int g_pp_on = -1;
static int pp_enabled() {
  if (g_pp_on < 0) g_pp_on = getenv("MT_IGEMM_PIPE_PATCH") ? (atoi(getenv("MT_IGEMM_PIPE_PATCH")) != 0) : 1;
  return g_pp_on;
}
long mt_pipe_patch_launches() { return g_pp_launches; }
int mt_pipe_patch_enable(int on) {
  const int prev = pp_enabled();
  g_pp_on = on != 0;
  return prev;
}

// -> 0 launched, 1 error, -1 not this kernel's shape (the caller launches igemm_pipe_kernel); dry: 102 = would launch
template <bool BF16>
int launch_igemm_pipe_patch_t(IgemmParams& p, int total, hipStream_t s, bool dry) {
  if (!pp_enabled()) return -1;
  const IgemmPhase& q = p.ph[0];
  if (p.nphase != 1 || p.raw || p.is != 1 || p.cpc % 4 != 0 || p.CoRows % 256 != 0) return -1;
  const int Wo = q.Wo, Ho = q.Ho;
  if (Wo < 16 || Wo > 256 || (Wo & (Wo - 1)) != 0 || (Ho * Wo) % 256 != 0 || q.M != p.N * Ho * Wo) return -1;
  if (q.ntaps > MT_PP_MAXTAPS || q.ntaps < 2 || q.y_off != 0 || q.tap0 != 0) return -1;
  int shift = 0;
  while ((1 << shift) < Wo) shift++;
  const int TH = 256 / Wo;
  int dhmin = 1 << 20, dhmax = -(1 << 20), dwmin = 1 << 20, dwmax = -(1 << 20);
  for (int t = 0; t < q.ntaps; t++) {
    const int dh = p.dh[q.tap0 + t], dw = p.dw[q.tap0 + t];
    dhmin = dh < dhmin ? dh : dhmin; dhmax = dh > dhmax ? dh : dhmax;
    dwmin = dw < dwmin ? dw : dwmin; dwmax = dw > dwmax ? dw : dwmax;
  }
  const int PH = TH + dhmax - dhmin, PW = Wo + dwmax - dwmin;
  if (PH * PW > MT_PP_PCAP) return -1;
  const int npp = (PH * PW + 15) / 16, nmine = (npp + 7) / 8;
  if (q.ntaps < nmine + 3) return -1;          // (the next slice's patch must be out three steps before the slice ends)
  if (p.x_bytes >= 0x7f000000u || q.w_bytes >= 0x3f000000u) return -1;
  if (p.fold) {
    // the in-operand fold is written for a 3x3 window at offsets -1 .. 1 on a map of the output's size (pad 1)
    for (int t = 0; t < q.ntaps && t < 9; t++)
      if (p.dh[t] != 1 - t / 3 || p.dw[t] != 1 - t % 3) return -1;      // (the kernel's address set-up has this order compiled in)
    if (q.ntaps != 9 || dhmin != -1 || dhmax != 1 || dwmin != -1 || dwmax != 1 || p.Hi < 4 || p.Wi < 4 || p.Hi != Ho || p.Wi != Wo ||
        p.pad_mode != MT_PAD_ZERO || p.os != 1 || (p.stats != nullptr && p.bstat_x == nullptr))
      return -1;
    // the virtual cells live behind the patch rows; the next slice's copies must all be out by step 3 (they are built in step 6)
    if (((PH * PW + 7) & ~7) + 8 * ((PH + 3) / 4) + 2 * (PW + 2) > MT_PP_PCAP || nmine > 4 || (PW & 7) != 2) return -1;
  }
  if (dry) return 102;
  if (p.fold)
    hipLaunchKernelGGL((igemm_pipe_patch_kernel<BF16, true, true>), dim3(total), dim3(512), 0, s, p, PH, PW, dhmin, dwmin, shift);
  else if (q.ntaps == 9 && nmine <= 5)
    hipLaunchKernelGGL((igemm_pipe_patch_kernel<BF16, false, true>), dim3(total), dim3(512), 0, s, p, PH, PW, dhmin, dwmin, shift);
  else
    hipLaunchKernelGGL((igemm_pipe_patch_kernel<BF16, false, false>), dim3(total), dim3(512), 0, s, p, PH, PW, dhmin, dwmin, shift);
  MT_LAUNCH_CHECK();
  __atomic_fetch_add(&g_pp_launches, 1, __ATOMIC_RELAXED);
  return 0;
}
template int launch_igemm_pipe_patch_t<true>(IgemmParams&, int, hipStream_t, bool);
template int launch_igemm_pipe_patch_t<false>(IgemmParams&, int, hipStream_t, bool);
