// Patch-resident gather-GEMM (bf16): the input patch of a 2-D pixel tile lives in LDS and serves ALL filter taps --
// and all sub-pixel phases of a transposed convolution / strided data gradient -- so a pixel is staged once per
// 32-channel slice instead of once per tap.  Same math and parameter block as igemm_kernel (conv_kernels.hip);
// chosen by launch_igemm_t for stride-1 gathers (IgemmParams::is == 1) with 1 or 4 phases.  gfx950 only.
//
// Why: the mid-size layers (Cin 64-256 on 32x32 .. 256x256 maps) ran at 0.06-0.25 of the
// MFMA peak, bound by what the L2 -> LDS path delivers: with 128-pixel tiles a k-step stages 24-32 KB for 1-2 MFLOP and
// the nine taps re-stage the same pixels nine times (the 3x3 64->128 layer on 128x128 moved 380-600 MB through LDS-DMA
// for 34 MB of input).  Here
//   * work item = (pixel tile of TH x TW outputs of the phase grid, tile of output channels); its K loop walks
//     32-channel SLICES, and inside a slice the distinct input offsets ("steps") of all taps of all phases;
//   * the slice's patch ((TH + span) x (TW + span) pixels x 64 B, rows XOR-swizzled like the tile rows of the other
//     kernels) is copied by LDS-DMA into one of two patch slots while the previous slice computes -- also across
//     items, so no patch latency is ever exposed; the B fragment of a step is a plain ds_read_b128 at
//     (pixel + step offset): 16 consecutive patch rows are conflict-free at any start row;
//   * only the weights go through a per-step ring (3 stages, counted s_waitcnt vmcnt): WTP x 64 B per active phase;
//   * four sub-pixel phases are computed TOGETHER from one patch: an input offset that several phases use (offset
//     (0,0) of a 3x3 stride-2 transposed convolution feeds one tap of each of the four phases) is read from LDS once
//     and multiplied with every phase's weight fragment;
//   * persistent: a workgroup walks its items (XCD-contiguous, as in conv_persist_kernel.hip), two workgroups per CU,
//     so one's epilogue stores run beside the other's MFMAs.
#include "conv_device.h"
#include <stdlib.h>
#include <string.h>

#define MT_PATCH_MAX_STEPS 16

struct PatchPlan {
  int nsteps;               // distinct input offsets over all taps of all phases
  int PH, PW, prows, npp;   // patch: PH x PW pixels, prows = PH * PW rows of 64 bytes per slice, npp = ceil(prows / 16) copies
  int dh0, dw0;             // input coordinate of patch pixel (0, 0) relative to the tile's first pixel
  int TH, TW, tw_shift;     // pixel tile of the phase grid: TH x TW, TW = 1 << tw_shift, TH * TW = pixels per workgroup
  int tiles_x, tiles_y;     // tiles per image
  int nct;                  // tiles of output channels
  int nsl;                  // 32-channel slices
  int total;                // items = N * tiles_y * tiles_x * nct
  int pks;                  // patch copies per wave per step (steps 0 .. nsteps-2 of the previous slice)
  int Ho, Wo;               // phase grid (max over the phases)
  // (dwords: a dynamically indexed short / char table in the kernel arguments is read with a VECTOR load, whose wait
  //  drains the LDS-DMA queue)
  int qoff[MT_PATCH_MAX_STEPS];              // patch row offset of a step: (dh - dh0) * PW + (dw - dw0)
  unsigned tap[MT_PATCH_MAX_STEPS];          // byte p: tap index inside phase p's weight image, 0xff = the phase has no tap there
};

template <int N>
__device__ __forceinline__ void pt_wait_vm() {
  asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}
// wave-uniform `young` = vector-memory operations this wave issued after the weight copies it is about to read
__device__ __forceinline__ void pt_wait_young(int young) {
  switch (young) {
#define PT_CASE(n) case n: pt_wait_vm<n>(); break;
    PT_CASE(0) PT_CASE(1) PT_CASE(2) PT_CASE(3) PT_CASE(4) PT_CASE(5) PT_CASE(6) PT_CASE(7) PT_CASE(8) PT_CASE(9)
    PT_CASE(10) PT_CASE(11) PT_CASE(12) PT_CASE(13) PT_CASE(14) PT_CASE(15) PT_CASE(16) PT_CASE(17) PT_CASE(18)
    PT_CASE(19) PT_CASE(20) PT_CASE(21) PT_CASE(22) PT_CASE(23) PT_CASE(24) PT_CASE(25) PT_CASE(26) PT_CASE(27)
    PT_CASE(28) PT_CASE(29) PT_CASE(30) PT_CASE(31) PT_CASE(32) PT_CASE(33) PT_CASE(34) PT_CASE(35) PT_CASE(36)
    PT_CASE(37) PT_CASE(38) PT_CASE(39) PT_CASE(40) PT_CASE(41) PT_CASE(42) PT_CASE(43) PT_CASE(44) PT_CASE(45)
    PT_CASE(46) PT_CASE(47) PT_CASE(48)
#undef PT_CASE
    default: pt_wait_vm<0>(); break;       // (waiting for everything is always correct)
  }
}
// exact floor(m / d) for 0 <= m < 2^24, 0 < d < 2^24 with inv = 1.0f / d
__device__ __forceinline__ int pt_div(int m, int d, float inv) {
  int q = (int)((float)m * inv);
  int r = m - q * d;
  q = r < 0 ? q - 1 : q;
  r = r < 0 ? r + d : r;
  q = r >= d ? q + 1 : q;
  return q;
}
__device__ __forceinline__ int pt_udiv(int m, int d, float inv) { return __builtin_amdgcn_readfirstlane(pt_div(m, d, inv)); }
__device__ __forceinline__ void pt_barrier() {
  __builtin_amdgcn_sched_barrier(0);
  __builtin_amdgcn_s_barrier();
  __builtin_amdgcn_sched_barrier(0);
}

constexpr int MT_PATCH_MAX_CO = 1024;       // output channels whose bias fits the LDS copy

// NPH: phases computed together (1: gather form, 4: the sub-pixel phases of a stride-2 scatter form)
// FCP: 16-row weight fragments per phase per wave; WCO: waves along the output channels (4 / WCO along the pixels)
// PCAP: rows of a patch slot; AUX: cache policy of the output stores (2 = nt); DUAL: IgemmParams::y2 is set
template <int NPH, int FCP, int WCO, int PCAP, int AUX, bool DUAL>
__global__ __launch_bounds__(256, 2) void igemm_patch_kernel(const IgemmParams p, const PatchPlan pl, const unsigned w_total,
                                                             const unsigned y_total) {
  constexpr int WPX = 4 / WCO, FP = 4, FC = NPH * FCP;
  constexpr int WTP = WCO * FCP * 16;          // output channels per phase per workgroup
  constexpr int SROWS = NPH * WTP;             // rows of a weight stage
  constexpr int NS = 3;
  constexpr int WPP = WTP / 64;                // weight copies per wave per active phase and step
  constexpr int NPPW = (PCAP / 16 + 3) / 4;    // patch copies per wave per slice (at most)
  constexpr unsigned OOB = 0x80000000u;
  static_assert(WTP % 64 == 0 && FCP % 2 == 0 && PCAP % 16 == 0, "geometry");
  static_assert((NS * SROWS + 2 * PCAP) * 64 + MT_PATCH_MAX_CO * 4 <= 80 * 1024, "two workgroups per CU");

  // ONE shared array (separate __shared__ objects next to an LDS-DMA target make hipcc drain vmcnt)
  __shared__ u32x4 smem[NS * SROWS * 4 + 2 * PCAP * 4 + MT_PATCH_MAX_CO / 4];
  u32x4* const sW = smem;
  u32x4* const sP = smem + NS * SROWS * 4;
  float* const sBias = reinterpret_cast<float*>(smem + NS * SROWS * 4 + 2 * PCAP * 4);

  const int tid = threadIdx.x;
  const int lane = tid & 63, wv = tid >> 6;
  const int fr = lane & 15, fg = lane >> 4;
  const int wvu = __builtin_amdgcn_readfirstlane(wv);
  const int wco = wvu / WPX, wpx = wvu % WPX;
  const int rsub = lane >> 2, csub = lane & 3;

  // ---- persistent schedule (conv_persist_kernel.hip): each XCD owns a contiguous eighth of the items; its workgroups
  // take consecutive groups of G items (the channel tiles of one pixel tile) round after round ----
  const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3;
  const int nslot = ((int)gridDim.x - xcd + 7) >> 3;
  const int tq = pl.total >> 3, tr = pl.total & 7;
  const int lo = xcd * tq + (xcd < tr ? xcd : tr);
  const int cnt = tq + (xcd < tr ? 1 : 0);
  const int G = pl.nct;
  const float inv_G = 1.0f / (float)G;
  auto item_at = [&](int i) {
    const int round = pt_udiv(i, G, inv_G);
    const int idx = (round * nslot + slot) * G + (i - round * G);
    return idx < cnt ? lo + idx : -1;
  };
  if (item_at(0) < 0) return;

  const float inv_nct = 1.0f / (float)pl.nct, inv_tx = 1.0f / (float)pl.tiles_x, inv_ty = 1.0f / (float)pl.tiles_y;
  const float inv_pw = 1.0f / (float)pl.PW;
  auto decode = [&](int item, int& n, int& ty, int& tx, int& ct) {
    int t = pt_udiv(item, pl.nct, inv_nct);
    ct = item - t * pl.nct;
    const int t2 = pt_udiv(t, pl.tiles_x, inv_tx);
    tx = t - t2 * pl.tiles_x;
    n = pt_udiv(t2, pl.tiles_y, inv_ty);
    ty = t2 - n * pl.tiles_y;
  };

  const __amdgpu_buffer_rsrc_t rsx = __builtin_amdgcn_make_buffer_rsrc((void*)p.x, 0, p.x_bytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t rsw = __builtin_amdgcn_make_buffer_rsrc((void*)p.w, 0, w_total, 0x00020000);
  const __amdgpu_buffer_rsrc_t rsy = __builtin_amdgcn_make_buffer_rsrc((void*)p.y, 0, y_total, 0x00020000);
  const __amdgpu_buffer_rsrc_t rsy2 = __builtin_amdgcn_make_buffer_rsrc(
      (void*)(DUAL ? p.y2 : p.y), 0, DUAL ? (unsigned)p.N * (unsigned)(p.y2H * p.y2W) * (unsigned)(p.Co * 2) : 0u, 0x00020000);
  typedef __attribute__((address_space(3))) void* lds_ptr;

  for (int i = tid; i < MT_PATCH_MAX_CO; i += 256) sBias[i] = (p.bias != nullptr && i < p.nbias) ? p.bias[i] : 0.f;

  // ---- patch cursor: the (item, slice) whose patch is being copied; xrow[j] = source offset of this lane's row of
  // copy j (without the slice), OOB where the operand is zero ----
  static_assert(NPPW <= 6 && NPH <= 4 && WPP <= 2, "literal-sized arrays (hipcc drops the host stub for a dependent-size lambda capture)");
  unsigned xrow[6];
  int p_i = 0, p_sl = 0, p_j = 0, p_tile = -1;
  bool p_valid = true;
  const int nmine = pl.npp > wvu ? (pl.npp - wvu + 3) >> 2 : 0;      // copies of a slice that this wave issues
  auto patch_setup = [&](int item) {
    int n, ty, tx, ct;
    decode(item, n, ty, tx, ct);
    const int h0 = ty * pl.TH + pl.dh0, w0 = tx * pl.TW + pl.dw0;
    const unsigned nbase = (unsigned)n * (unsigned)(p.Hi * p.Wi) * (unsigned)p.Cib;
#pragma unroll
    for (int j = 0; j < NPPW; j++) {
      const int r = (wvu + 4 * j) * 16 + rsub;
      const int py = pt_div(r, pl.PW, inv_pw);
      const int px = r - py * pl.PW;
      int hi = h0 + py, wi = w0 + px;
      bool ok = r < pl.prows;
      if (p.pad_mode == MT_PAD_REFLECT) {
        hi = hi < 0 ? -hi : hi;
        hi = hi >= p.Hi ? 2 * (p.Hi - 1) - hi : hi;
        wi = wi < 0 ? -wi : wi;
        wi = wi >= p.Wi ? 2 * (p.Wi - 1) - wi : wi;
        // (tiles that overhang the grid reach further than one reflection: those rows feed masked outputs only)
        ok = ok && ((unsigned)hi < (unsigned)p.Hi) && ((unsigned)wi < (unsigned)p.Wi);
      } else {
        ok = ok && ((unsigned)hi < (unsigned)p.Hi) && ((unsigned)wi < (unsigned)p.Wi);
      }
      xrow[j] = ok ? nbase + (unsigned)(hi * p.Wi + wi) * (unsigned)p.Cib + (unsigned)((csub ^ ((r >> 1) & 3)) * 16) : OOB;
    }
  };
  auto patch_piece = [&](int pslot, int j, int sl) {
    // (j is wave-uniform; the unrolled compare keeps xrow[] in registers)
    unsigned off = OOB;
#pragma unroll
    for (int jj = 0; jj < NPPW; jj++) off = (j == jj) ? xrow[jj] : off;
    char* dst = reinterpret_cast<char*>(sP + pslot * PCAP * 4) + (wvu + 4 * j) * 1024;
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rsx, (lds_ptr)dst, 16, off + (unsigned)sl * 64u, 0, 0, 0);
  };

  // ---- weight cursor: the (item, slice, step) whose weight stage is issued next ----
  unsigned wrow[4][2];
  int w_i = 0, w_sl = 0, w_st = 0;
  bool w_valid = true;
  auto weight_setup = [&](int item) {
    int n, ty, tx, ct;
    decode(item, n, ty, tx, ct);
#pragma unroll
    for (int ph = 0; ph < NPH; ph++)
#pragma unroll
      for (int jj = 0; jj < WPP; jj++) {
        const int rs = (jj * 4 + wvu) * 16 + rsub;          // LDS row inside the phase's block of WTP rows
        // 16-byte epilogue stores: within each 32-row fragment pair LDS row (a&1)*16 + r holds channel
        // (r>>2)*8 + (a&1)*4 + (r&3) (as in igemm_kernel)
        const int rl = (rs & ~31) | ((((rs & 15) >> 2) << 3) | (((rs >> 4) & 1) << 2) | (rs & 3));
        const int row = ct * WTP + rl;
        wrow[ph][jj] = row < p.CoRows
                           ? p.ph[ph].w_off + ((unsigned)row * (unsigned)p.ph[ph].wrow + (unsigned)(csub ^ ((rs >> 1) & 3))) * 16u
                           : OOB;
      }
  };
  // -> number of copies issued
  auto weight_issue = [&](int ws) {
    int n = 0;
    if (w_valid) {
      const unsigned w_taps = pl.tap[w_st];
#pragma unroll
      for (int ph = 0; ph < NPH; ph++) {
        const int ti = (int)((w_taps >> (8 * ph)) & 0xffu);
        if (ti != 0xff) {
          const unsigned k = (unsigned)((ti * p.cpc + w_sl * 4) * 16);
#pragma unroll
          for (int jj = 0; jj < WPP; jj++) {
            char* dst = reinterpret_cast<char*>(sW + (ws * SROWS + ph * WTP) * 4) + (jj * 4 + wvu) * 1024;
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rsw, (lds_ptr)dst, 16, wrow[ph][jj] + k, 0, 0, 0);
          }
          n += WPP;
        }
      }
      w_st++;
      if (w_st == pl.nsteps) {
        w_st = 0;
        w_sl++;
        if (w_sl == pl.nsl) {
          w_sl = 0;
          w_i++;
          const int item = item_at(w_i);
          w_valid = item >= 0;
          if (w_valid) weight_setup(item);
        }
      }
    }
    return n;
  };

  // ---- fragment addressing ----
  int q0[FP];                     // patch row of this lane's pixel of fragment b at step offset 0
#pragma unroll
  for (int b = 0; b < FP; b++) {
    const int tp = wpx * 64 + b * 16 + fr;
    q0[b] = (tp >> pl.tw_shift) * pl.PW + (tp & (pl.TW - 1));
  }

  f32x4 acc[FC][FP];
#pragma unroll
  for (int a = 0; a < FC; a++)
#pragma unroll
    for (int b = 0; b < FP; b++) acc[a][b] = f32x4{0.f, 0.f, 0.f, 0.f};
  const float neg_slope = p.act == MT_ACT_RELU ? 0.f : (p.act == MT_ACT_LRELU ? p.slope : 1.f);

  // ---- prologue: patch of the first slice, weight stages 0 and 1 ----
  int c_i = 0, c_sl = 0, c_st = 0;
  int c_n, c_ty, c_tx, c_ct;
  {
    const int item = item_at(0);
    decode(item, c_n, c_ty, c_tx, c_ct);
    patch_setup(item);
    p_tile = pt_udiv(item, pl.nct, inv_nct);
    for (int j = 0; j < nmine; j++) patch_piece(0, j, 0);
    weight_setup(item);
  }
  weight_issue(0);
  // younger than the weight stage a step waits for (issued two steps earlier, after that step's patch copies and
  // before its epilogue stores): the stores of that step, and everything the step in between issued
  int i1 = weight_issue(1);      // copies issued during the previous step
  int s1 = 0, s2 = 0;            // epilogue stores issued during the previous step / the one before it
  int gs = 0, gsl = 0;           // ring slot of the step being computed, parity of the slice being computed
  __syncthreads();               // bias table visible

  while (true) {
    // the weight stage of this step (issued two steps ago, before that step's other operations) has landed, and --
    // at the first step of a slice -- the slice's patch (issued before the last step of the previous slice)
    pt_wait_young(s2 + i1 + s1);
    pt_barrier();
    int issued = 0;
    if (c_st == 0) {
      // the patch cursor moves to the slice after this one
      p_sl = c_sl + 1;
      p_i = c_i;
      p_j = 0;
      p_valid = true;
      if (p_sl == pl.nsl) {
        p_sl = 0;
        p_i = c_i + 1;
        const int item = item_at(p_i);
        p_valid = item >= 0;
        if (p_valid) {
          const int tile = pt_udiv(item, pl.nct, inv_nct);
          if (tile != p_tile) { patch_setup(item); p_tile = tile; }
        }
      }
    }
    // ---- fragments of this step ----
    const int qo = pl.qoff[c_st];
    u32x4 xf[FP];
    {
      const u32x4* sPs = sP + (gsl & 1) * PCAP * 4;
#pragma unroll
      for (int b = 0; b < FP; b++) {
        const int q = q0[b] + qo;
        xf[b] = sPs[q * 4 + (fg ^ ((q >> 1) & 3))];
      }
    }
    const u32x4* sWs = sW + gs * SROWS * 4;
    const unsigned c_taps = pl.tap[c_st];
    bool tapon[NPH];
#pragma unroll
    for (int ph = 0; ph < NPH; ph++) tapon[ph] = ((c_taps >> (8 * ph)) & 0xffu) != 0xffu;
    u32x4 wf[FC];
#pragma unroll
    for (int ph = 0; ph < NPH; ph++)
      if (tapon[ph]) {
#pragma unroll
        for (int a = 0; a < FCP; a++) {
          const int row = ph * WTP + wco * (FCP * 16) + a * 16 + fr;
          wf[ph * FCP + a] = sWs[row * 4 + (fg ^ ((row >> 1) & 3))];
        }
      }
    // ---- copies: patch rows of the next slice first, the weight stage two steps ahead last ----
    if (p_valid && c_st + 1 < pl.nsteps) {
      for (int k = 0; k < pl.pks; k++)
        if (p_j < nmine) {
          patch_piece((gsl + 1) & 1, p_j, p_sl);
          p_j++;
          issued++;
        }
    }
    {
      int fill = gs - 1;
      fill = fill < 0 ? NS - 1 : fill;
      issued += weight_issue(fill);
    }
    __builtin_amdgcn_sched_barrier(0);
    // ---- MFMAs ----
    __builtin_amdgcn_s_setprio(1);
#pragma unroll
    for (int ph = 0; ph < NPH; ph++)
      if (tapon[ph]) {
#pragma unroll
        for (int a = 0; a < FCP; a++)
#pragma unroll
          for (int b = 0; b < FP; b++) mma_chunk<true>(acc[ph * FCP + a][b], wf[ph * FCP + a], xf[b]);
      }
    __builtin_amdgcn_s_setprio(0);
    gs = gs + 1 == NS ? 0 : gs + 1;
    int stores = 0;
    c_st++;
    if (c_st == pl.nsteps) {
      c_st = 0;
      c_sl++;
      gsl++;
      if (c_sl == pl.nsl) {
        // ---- item finished: bias + activation, packed NHWC stores ----
        unsigned yo[NPH][FP], yo2[NPH][FP];
#pragma unroll
        for (int ph = 0; ph < NPH; ph++)
#pragma unroll
          for (int b = 0; b < FP; b++) {
            const int tp = wpx * 64 + b * 16 + fr;
            const int ho = c_ty * pl.TH + (tp >> pl.tw_shift), wo = c_tx * pl.TW + (tp & (pl.TW - 1));
            const int oh = ho * p.os + p.ph[ph].oh0, ow = wo * p.os + p.ph[ph].ow0;
            unsigned o = OOB, o2 = OOB;
            if (ho < p.ph[ph].Ho && wo < p.ph[ph].Wo && (unsigned)oh < (unsigned)p.Hout && (unsigned)ow < (unsigned)p.Wout) {
              o = (unsigned)((c_n * p.Hout + oh) * p.Wout + ow) * (unsigned)(p.Co * 2);
              if constexpr (DUAL) {
                const int ih = oh - p.y2P, iw = ow - p.y2P;
                if ((unsigned)ih < (unsigned)p.y2H && (unsigned)iw < (unsigned)p.y2W) {
                  o2 = (unsigned)((c_n * p.y2H + ih) * p.y2W + iw) * (unsigned)(p.Co * 2);
                  o = OOB;
                }
              }
            }
            yo[ph][b] = o;
            yo2[ph][b] = o2;
          }
#pragma unroll
        for (int sp = 0; sp < FCP / 2; sp++) {
          const int co = c_ct * WTP + wco * (FCP * 16) + sp * 32 + fg * 8;
          const unsigned hco = co < p.Co ? (unsigned)co * 2u : OOB;
          float bv[8];
          {
            const int cb = co < MT_PATCH_MAX_CO - 8 ? co : 0;        // (co >= Co is never stored)
            const f32x4 b0 = *reinterpret_cast<const f32x4*>(sBias + cb), b1 = *reinterpret_cast<const f32x4*>(sBias + cb + 4);
#pragma unroll
            for (int e = 0; e < 4; e++) { bv[e] = b0[e]; bv[4 + e] = b1[e]; }
          }
#pragma unroll
          for (int ph = 0; ph < NPH; ph++)
#pragma unroll
            for (int b = 0; b < FP; b++) {
              float v[8];
#pragma unroll
              for (int e = 0; e < 8; e++) {
                const float z = acc[ph * FCP + 2 * sp + (e >> 2)][b][e & 3] + bv[e];
                v[e] = z > 0.f ? z : z * neg_slope;      // none / ReLU / LeakyReLU (tanh launches take the other kernels)
              }
              const u32x4 o = {pack2_bf16(v[0], v[1]), pack2_bf16(v[2], v[3]), pack2_bf16(v[4], v[5]), pack2_bf16(v[6], v[7])};
              const unsigned off = ((yo[ph][b] | hco) & OOB) ? OOB : yo[ph][b] + hco;
              __builtin_amdgcn_raw_buffer_store_b128(o, rsy, off, 0, AUX);
              stores++;
              if constexpr (DUAL) {
                const unsigned off2 = ((yo2[ph][b] | hco) & OOB) ? OOB : yo2[ph][b] + hco;
                __builtin_amdgcn_raw_buffer_store_b128(o, rsy2, off2, 0, AUX);
                stores++;
              }
              acc[ph * FCP + 2 * sp][b] = f32x4{0.f, 0.f, 0.f, 0.f};
              acc[ph * FCP + 2 * sp + 1][b] = f32x4{0.f, 0.f, 0.f, 0.f};
            }
        }
        c_sl = 0;
        c_i++;
        const int item = item_at(c_i);
        if (item < 0) break;
        decode(item, c_n, c_ty, c_tx, c_ct);
      }
    }
    s2 = s1;
    s1 = stores;
    i1 = issued;
  }
  pt_wait_vm<0>();       // (nothing is in flight past the last item: the cursors stop at the end of the list)
}

// ------------------------------------------------------------------------------------------------------------
static long g_patch_launches = 0;
static int g_patch_on = -1;
// 0 = off, 1 = where it measured faster inside the training step (the default), 2 = every shape it can run
// (tests; tools/bench_patch.py).  MT_IGEMM_PATCH in the environment sets the initial value.
static int patch_enabled() {
  if (g_patch_on < 0) g_patch_on = getenv("MT_IGEMM_PATCH") ? atoi(getenv("MT_IGEMM_PATCH")) : 1;
  return g_patch_on;
}
long mt_patch_launches() { return g_patch_launches; }
int mt_patch_enable(int on) {
  const int prev = patch_enabled();
  g_patch_on = on < 0 ? 0 : (on > 2 ? 2 : on);
  return prev;
}

// -> 0 launched, 1 error, -1 not applicable (the caller goes on to the other kernels), 100 (dry) would launch
int launch_igemm_patch(IgemmParams& p, hipStream_t s, bool dry) {
  if (!patch_enabled()) return -1;
  if (p.raw || p.stats != nullptr || p.act == MT_ACT_TANH || p.is != 1 || p.cpc % 4 != 0 || p.cpc < 4) return -1;
  if (p.nphase != 1 && p.nphase != 4) return -1;
  if (p.os != (p.nphase == 1 ? 1 : 2)) return -1;
  if (p.Co > MT_PATCH_MAX_CO - 8 || p.CoRows % 64 != 0) return -1;
  const int NPH = p.nphase;
  // ---- steps: the distinct input offsets of all taps ----
  PatchPlan pl;
  memset(&pl, 0, sizeof(pl));
  memset(pl.tap, 0xff, sizeof(pl.tap));
  int sdh[MT_PATCH_MAX_STEPS], sdw[MT_PATCH_MAX_STEPS];
  int Ho = 0, Wo = 0;
  unsigned long long w_total = 0;
  for (int i = 0; i < NPH; i++) {
    const IgemmPhase& q = p.ph[i];
    if (q.ntaps < 1 || q.y_off != 0 || q.M <= 0) return -1;
    if (q.Ho * q.Wo * p.N != q.M) return -1;
    Ho = q.Ho > Ho ? q.Ho : Ho;
    Wo = q.Wo > Wo ? q.Wo : Wo;
    const unsigned long long e = (unsigned long long)q.w_off + q.w_bytes;
    w_total = e > w_total ? e : w_total;
    for (int t = 0; t < q.ntaps; t++) {
      const int dh = p.dh[q.tap0 + t], dw = p.dw[q.tap0 + t];
      int st = -1;
      for (int k = 0; k < pl.nsteps; k++)
        if (sdh[k] == dh && sdw[k] == dw) st = k;
      if (st < 0) {
        if (pl.nsteps == MT_PATCH_MAX_STEPS) return -1;
        st = pl.nsteps++;
        sdh[st] = dh; sdw[st] = dw;
      }
      if (((pl.tap[st] >> (8 * i)) & 0xffu) != 0xffu || t > 127) return -1;       // (two taps of one phase at one offset: not a convolution)
      pl.tap[st] = (pl.tap[st] & ~(0xffu << (8 * i))) | ((unsigned)t << (8 * i));
    }
  }
  if (pl.nsteps < 2) return -1;
  // steps with many active phases first (the order inside a slice is free; the weight ring sees its big stages early)
  int dhmin = sdh[0], dhmax = sdh[0], dwmin = sdw[0], dwmax = sdw[0];
  for (int k = 1; k < pl.nsteps; k++) {
    dhmin = sdh[k] < dhmin ? sdh[k] : dhmin; dhmax = sdh[k] > dhmax ? sdh[k] : dhmax;
    dwmin = sdw[k] < dwmin ? sdw[k] : dwmin; dwmax = sdw[k] > dwmax ? sdw[k] : dwmax;
  }
  // ---- geometry ----
  const bool g128 = NPH == 1 && p.CoRows % 128 == 0;
  const int WTP = NPH == 4 ? 64 : (g128 ? 128 : 64);
  const int PT = NPH == 4 ? 128 : 256;
  const int PCAP = NPH == 4 ? 192 : 352;
  int TW = NPH == 4 ? 16 : 32;
  if (Wo <= 16 || (Wo % 32 != 0 && Wo % 16 == 0)) TW = 16;
  if (NPH == 1 && TW == 16 && Ho < 12) return -1;              // (small maps stay with the other kernels)
  const int TH = PT / TW;
  if (Wo < 12 || Ho < TH / 2) return -1;
  pl.TH = TH; pl.TW = TW; pl.tw_shift = TW == 32 ? 5 : 4;
  pl.PH = TH + (dhmax - dhmin); pl.PW = TW + (dwmax - dwmin);
  pl.prows = pl.PH * pl.PW;
  if (pl.prows > PCAP) return -1;
  pl.npp = (pl.prows + 15) / 16;
  pl.dh0 = dhmin; pl.dw0 = dwmin;
  for (int k = 0; k < pl.nsteps; k++) pl.qoff[k] = (sdh[k] - dhmin) * pl.PW + (sdw[k] - dwmin);
  pl.tiles_x = (Wo + TW - 1) / TW; pl.tiles_y = (Ho + TH - 1) / TH;
  pl.nct = (p.CoRows + WTP - 1) / WTP;
  pl.nsl = p.cpc / 4;
  pl.Ho = Ho; pl.Wo = Wo;
  const long total = (long)p.N * pl.tiles_x * pl.tiles_y * pl.nct;
  if (total >= (1 << 24) || total < 192) return -1;
  if (patch_enabled() < 2) {
    // In-step A/B (tools/layer_table.py, profiles/round3_*): the 128-channel single-phase geometry wins where the
    // reduction is short and the grid large (3x3 64->128 on 128x128: 87 -> 58 us); with 16-MFMA steps (64-channel
    // tiles) or four phases per step (8-32 MFMAs between barriers) the per-step synchronisation costs more than the
    // staged bytes save, and small grids leave half the wave slots empty.
    if (NPH != 1 || !g128 || pl.nsl > 2 || total < 768) return -1;
  }
  // (tiles overhanging the grid compute masked pixels: refuse shapes that waste more than a third)
  if ((double)pl.tiles_x * TW * pl.tiles_y * TH > 1.5 * (double)Ho * Wo) return -1;
  pl.total = (int)total;
  const int nppw = (pl.npp + 3) / 4;
  pl.pks = (nppw + pl.nsteps - 2) / (pl.nsteps - 1);
  const unsigned long long y_total = (unsigned long long)p.N * p.Hout * p.Wout * p.Co * 2ull;
  if (w_total >= 0x7f000000ull || y_total >= 0x7f000000ull || p.x_bytes >= 0x7f000000u) return -1;
  if (p.y2 != nullptr && (unsigned long long)p.N * p.y2H * p.y2W * p.Co * 2ull >= 0x7f000000ull) return -1;
  if (dry) return 100;
  const int nb = total < 512 ? (int)total : 512;
  const unsigned wb = (unsigned)w_total, yb = (unsigned)y_total;
  const bool dual = p.y2 != nullptr;
#define PT_LAUNCH(NPHv, FCPv, WCOv, PCAPv, D) \
  hipLaunchKernelGGL((igemm_patch_kernel<NPHv, FCPv, WCOv, PCAPv, 2, D>), dim3(nb), dim3(256), 0, s, p, pl, wb, yb)
  if (NPH == 4) { if (dual) PT_LAUNCH(4, 2, 2, 192, true); else PT_LAUNCH(4, 2, 2, 192, false); }
  else if (g128) { if (dual) PT_LAUNCH(1, 8, 1, 352, true); else PT_LAUNCH(1, 8, 1, 352, false); }
  else { if (dual) PT_LAUNCH(1, 4, 1, 352, true); else PT_LAUNCH(1, 4, 1, 352, false); }
#undef PT_LAUNCH
  MT_LAUNCH_CHECK();
  __atomic_fetch_add(&g_patch_launches, 1, __ATOMIC_RELAXED);
  return 0;
}
