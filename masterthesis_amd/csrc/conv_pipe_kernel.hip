// Ping-pong software-pipelined gather-GEMM for 256x256 tiles (same math and parameter block as igemm_kernel
// in conv_kernels.hip; chosen by launch_igemm_t there).  gfx950 only.
//
//  * 32-deep k-steps (64-byte tile rows) in an NS-stage LDS ring filled by LDS-DMA
//    (buffer_load_dwordx4 ... lds) running NS-1 stages ahead; the wait is a COUNTED
//    s_waitcnt vmcnt((NS-2)*PPS), so the copies of the newest stages stay in flight across the barriers
//    (the 2-stage kernel drains with vmcnt(0) in __syncthreads(): 51 % of its wave cycles sit there).
//    Every wave issues exactly PPS copies per stage (offsets >= 2 GiB -> zeros, also past the end of K), so
//    the count is a compile-time constant in the steady state and in the tail.
//  * 8 waves = two groups of four (one wave of each group per SIMD) that alternate, half a k-step apart,
//    between a memory phase (copies + fragment reads) and a compute phase (32 back-to-back MFMAs).
//  * all gather addressing (filter tap offsets, reflection / zero padding) is resolved once per tile into an
//    LDS table; the main loop has no branch and no address arithmetic beyond table entry + chunk offset
//    (a branchy per-tap recomputation in the loop cost 17 % of the main-loop cycles).
//  * 16-byte epilogue stores: weight rows are staged in a permuted order so that a lane ends up with 8
//    consecutive output channels of a pixel.
// In-kernel s_memtime stamps (diagnostic build, tools/diag_build.sh + tools/stamp_k1.py) on 3x3 256->256,
// 16x64x64: prologue 7.7k, main loop 91k (1266 cycles per k-step against 1024 of pure MFMA issue), epilogue
// 20k cycles (all 256 CUs store 128 KiB at once: HBM-write bound).
#include "conv_device.h"
#include <type_traits>

#ifdef MT_STAMPS
// diagnostic build only (make STAMPS=1): s_memtime stamps of wave 0 of every block -> tools/stamp_k1.py
__device__ unsigned long long mt_stamp_buf[16 * 4096];
#define MT_STAMP(i) do { if (threadIdx.x == 0 && blockIdx.x < 4096) mt_stamp_buf[blockIdx.x * 16 + (i)] = __builtin_amdgcn_s_memtime(); } while (0)
extern "C" int mt_debug_stamps(void* dst, size_t bytes) {
  return (int)hipMemcpyFromSymbol(dst, HIP_SYMBOL(mt_stamp_buf), bytes < sizeof(mt_stamp_buf) ? bytes : sizeof(mt_stamp_buf));
}
#else
#define MT_STAMP(i) do {} while (0)
#endif

template <int N>
__device__ __forceinline__ void wait_vmcnt() {
  asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}

constexpr int MT_PIPE_MAX_TAPS = 25;   // rows of the per-tile gather-offset table (+1 all-zero row)

template <bool BF16, int WT, int PT, int NT, int NS, int MAXTAPS>
__global__ __launch_bounds__(512) void igemm_pipe_kernel(const IgemmParams p) {
  constexpr int NW = NT / 64;
  // wave tile: 8 waves 128 couts x 64 pixels; 4 waves (NT = 256, latency-bound small launches) up to 64 x 64
  constexpr int WC = NW == 8 ? 128 : 64;
  constexpr int WP = (NW == 8 || WT >= 128) ? 64 : 32;
  constexpr int NWP = PT / WP;
  constexpr int FC = WC / 16, FP = WP / 16;
  constexpr int SZ = BF16 ? 2 : 4;
  constexpr int NXL = PT / 16 / NW;                 // pixel-tile copies per wave per stage
  constexpr int NWL = (WT / 16 + NW - 1) / NW;      // weight-tile copies per wave per stage (incl. dummy rows)
  constexpr int WR = NWL * NW * 16;                 // weight rows per stage (>= WT)
  constexpr int PPS = NXL + NWL;                    // copies per wave per stage
  constexpr int STAGE = (WR + PT) * 4;              // u32x4 per stage (4 chunks per 64-byte row)
  constexpr int TROWS = MAXTAPS + 1;               // rows of the gather-offset table (+1 all-zero row)
  static_assert(NW == (PT / WP) * (WT / WC), "wave grid must cover the block tile");
  static_assert(NXL * NW * 16 == PT, "the pixel tile is a whole number of copies per wave");
  static_assert(NS >= 3 && (NS - 2) * PPS <= 63, "vmcnt range");
  static_assert(FC % 2 == 0, "the epilogue stores fragment pairs");
  static_assert((NS * STAGE * 16 + TROWS * PT * 4) <= 160 * 1024, "LDS budget");

  // ONE shared array (a second __shared__ object next to an LDS-DMA target makes hipcc drain vmcnt):
  // NS stages of (weight tile | pixel tile), then the gather-offset table T[tap][pixel of the tile]
  __shared__ u32x4 smem[NS * STAGE + TROWS * PT / 4];
  unsigned* const sT = reinterpret_cast<unsigned*>(&smem[NS * STAGE]);

  MT_STAMP(0);
  const int tid = threadIdx.x;
  const int lane = tid & 63, wv = tid >> 6;
  const int fr = lane & 15, fg = lane >> 4;
  const int wcI = wv / NWP, wpI = wv % NWP;
  const int wvu = __builtin_amdgcn_readfirstlane(wv);

  const int nWT = (p.CoRows + WT - 1) / WT;
  int wg = xcd_remap(blockIdx.x, gridDim.x);
  int phi = 0;
  if (p.interleave) {
    phi = wg % p.nphase;
    wg = wg / p.nphase + p.ph[phi].blk0;
  } else {
    for (int i = 1; i < p.nphase; i++) phi = (wg >= p.ph[i].blk0) ? i : phi;
  }
  const IgemmPhase& ph = p.ph[phi];
  const int ph_ntaps = ph.ntaps, ph_Ho = ph.Ho, ph_Wo = ph.Wo, ph_M = ph.M;
  const int ph_nchunks = ph_ntaps * p.cpc;
  const char* const ph_w = p.w + ph.w_off;
  wg -= ph.blk0;
  const int wt = wg % nWT, pt = wg / nWT;
  const int HoWo = ph_Ho * ph_Wo;

  // ---- gather-offset table: T[t][r] = byte offset of the input pixel that filter tap t reads for tile
  // pixel r (reflection / zero padding resolved here, ONCE per tile), or OOB where the operand is zero: the
  // buffer load range-checks and returns 0 for offsets >= 2 GiB (the host checks the tensors are smaller).
  // Row ntaps is all-OOB (the pipeline runs NS-1 stages past the end of K).  The main loop is then free of
  // address arithmetic beyond "table entry + chunk offset" and of any branch.
  constexpr unsigned OOB = 0x80000000u;
  {
    const int r = tid & (PT - 1);
    const int m = pt * PT + r;
    const bool rv = m < ph_M;
    const int mm = rv ? m : 0;
    const int n = mm / HoWo;
    const int rem = mm - n * HoWo;
    const int ho = rem / ph_Wo;
    const int wo = rem - ho * ph_Wo;
    const unsigned ibase = (unsigned)n * (unsigned)(p.Hi * p.Wi) * (unsigned)p.Cib;
    // (t is the same for a whole wave: said so, the tap offsets are SCALAR loads from the kernel arguments.  As lane-indexed shorts
    //  they were vector loads with a full wait inside this loop -- a chain of ntaps / 2 memory latencies in front of the first copy of
    //  every tile: round 4, found next to the same pattern in the patch-resident kernel)
    for (int t = __builtin_amdgcn_readfirstlane(tid / PT); t <= ph_ntaps; t += NT / PT) {
      unsigned off = OOB;
      if (t < ph_ntaps) {
        int hi = ho * p.is + tap_dh(p, ph.tap0 + t), wi = wo * p.is + tap_dw(p, ph.tap0 + t);
        bool ok = rv;
        if (p.pad_mode == MT_PAD_REFLECT) {
          hi = hi < 0 ? -hi : hi;
          hi = hi >= p.Hi ? 2 * (p.Hi - 1) - hi : hi;
          wi = wi < 0 ? -wi : wi;
          wi = wi >= p.Wi ? 2 * (p.Wi - 1) - wi : wi;
        } else {
          ok = ok && ((unsigned)hi < (unsigned)p.Hi) && ((unsigned)wi < (unsigned)p.Wi);
        }
        if (ok) off = ibase + (unsigned)(hi * p.Wi + wi) * (unsigned)p.Cib;
      }
      sT[t * PT + r] = off;
    }
  }

  // ---- staging coordinates: a copy instruction writes 64 lanes x 16 B = 16 tile rows lane-linearly; the
  // bank swizzle (chunk ^ ((row >> 1) & 3), conflict-free for the ds_read_b128 fragment reads of 64-byte
  // rows) is applied to the SOURCE chunk.  The host guarantees cpc % 4 == 0, so a k-step (4 chunks) never
  // straddles a filter tap and the tap index is wave-uniform.
  const int rsub = lane >> 2;
  const int c = (lane & 3) ^ ((rsub >> 1) & 3);
  unsigned wo32[2], xo32[4];
  int prow[4];                  // this lane's pixel rows of the tile (one per copy instruction)
  static_assert(NWL <= 2 && NXL <= 4, "literal-sized arrays (hipcc drops the host stub for a dependent-size lambda capture)");
#pragma unroll
  for (int i = 0; i < NXL; i++) prow[i] = 16 * (wvu + NW * i) + rsub;
#pragma unroll
  for (int i = 0; i < NWL; i++) {
    const int rs = 16 * (wvu + NW * i) + rsub;           // LDS row of the weight tile (fragment order)
    // channel held by that row: within each 32-row fragment pair, row (a&1)*16 + r <- channel (r>>2)*8 + (a&1)*4 + (r&3)
    const int rl = (rs & ~31) | ((((rs & 15) >> 2) << 3) | (((rs >> 4) & 1) << 2) | (rs & 3));
    const int row = wt * WT + rl;
    wo32[i] = (rs < WT && row < p.CoRows) ? ((unsigned)row * (unsigned)ph.wrow + (unsigned)c) * 16u : OOB;
  }
  int tap_s = 0, cqb = 0;       // wave-uniform: filter tap and first chunk inside it of the NEXT stage to issue
  // K order.  Tap-major (the pack order): a tile re-reads its ~200 KB input patch once per tap, 8 k-steps apart --
  // with 32 tiles per XCD that is 6 MB against 4 MB of L2, so most re-reads go out to the fabric.  Slice-major: the
  // taps of one 32-channel slice back to back (25 KB per tile), every input byte leaves HBM/MALL once.
  const bool slice_major = p.korder != 0;
  unsigned wk = 0;              // byte offset of the next stage's chunk group inside a weight row

  const __amdgpu_buffer_rsrc_t rsx = __builtin_amdgcn_make_buffer_rsrc((void*)p.x, 0, p.x_bytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t rsw = __builtin_amdgcn_make_buffer_rsrc((void*)ph_w, 0, ph.w_bytes, 0x00020000);
  typedef __attribute__((address_space(3))) void* lds_ptr;
  char* const lds0 = reinterpret_cast<char*>(&smem[0]);

  // the PPS copy instructions of one stage into ring slot `slot`
  // (no K-tail check on the weight side: there the pixel operand is zero, and reading into the next pack row
  // or past the end -- range-checked -> 0 -- only multiplies finite weights by 0)
  auto issue_piece = [&](int slot, int j) {
    char* base = lds0 + slot * (STAGE * 16);
    if (j < NWL) {
      // (an OOB-marked row plus a k offset stays >= 2 GiB: the pack is < 2 GiB, checked by the host)
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rsw, (lds_ptr)(base + (wvu + NW * j) * 1024), 16, wo32[j] + wk, 0, 0, 0);
    } else {
      const int i = j - NWL;
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rsx, (lds_ptr)(base + WR * 64 + (wvu + NW * i) * 1024), 16, xo32[i], 0, 0,
                                               0);
    }
  };
  auto issue_stage = [&](int slot) {
#pragma unroll
    for (int j = 0; j < PPS; j++) issue_piece(slot, j);
  };
  // table entries of the stage after the one just issued (scalar tap / chunk bookkeeping, two LDS reads)
  unsigned tq[4];
  auto next_lookup = [&]() {
    if (slice_major) {
      tap_s += 1;
      const bool wrap = tap_s >= ph_ntaps;
      cqb = wrap ? cqb + 4 : cqb;
      tap_s = wrap ? 0 : tap_s;
      tap_s = cqb >= p.cpc ? ph_ntaps : tap_s;       // past the end of K: the all-OOB table row
      cqb = cqb >= p.cpc ? p.cpc : cqb;
      wk = (unsigned)((tap_s * p.cpc + cqb) * 16);
    } else {
      cqb += 4;
      const bool wrap = cqb >= p.cpc;
      cqb = wrap ? 0 : cqb;
      tap_s = wrap ? tap_s + 1 : tap_s;
      tap_s = tap_s > ph_ntaps ? ph_ntaps : tap_s;
      wk += 64u;
    }
#pragma unroll
    for (int i = 0; i < NXL; i++) tq[i] = sT[tap_s * PT + prow[i]];
  };
  auto next_offsets = [&]() {
#pragma unroll
    for (int i = 0; i < NXL; i++) xo32[i] = tq[i] + (unsigned)((cqb + c) * 16);
  };

  f32x4 acc[FC][FP];
#pragma unroll
  for (int a = 0; a < FC; a++)
#pragma unroll
    for (int b = 0; b < FP; b++) acc[a][b] = f32x4{0.f, 0.f, 0.f, 0.f};

  const int nk = ph_nchunks >> 2;
  __syncthreads();  // offset table visible
#pragma unroll
  for (int i = 0; i < NXL; i++) xo32[i] = sT[prow[i]] + (unsigned)(c * 16);
  // prologue: stages 0 .. NS-2
#pragma unroll
  for (int s = 0; s < NS - 1; s++) {
    issue_stage(s);
    next_lookup();
    next_offsets();
  }
  wait_vmcnt<(NS - 2) * PPS>();
  MT_STAMP(1);

  int slot = 0;   // ring slot of stage ks
  if constexpr (NW == 8) {
  // ---- ping-pong main loop: waves 0-3 and 4-7 (one of each per SIMD) alternate between a MEMORY phase
  // (LDS-DMA of stage ks+NS-1, fragment reads of stage ks) and a COMPUTE phase (32 back-to-back MFMAs), half
  // a k-step apart, so each SIMD's matrix pipe always has one wave feeding it while the other one waits on
  // the texture path / LDS.  Every wave executes the same number of s_barriers (2*nk + 2).
  //   slot reuse: stage ks+NS-1 overwrites the slot of stage ks-1, whose last readers (the partner group's
  //               memory phase ks-1) finished before the barrier this phase started from;
  //   landing:    at the end of memory phase ks a wave waits until all but its newest (NS-2)*PPS copies are
  //               done, i.e. its part of stage ks+1 is in LDS before the barrier in front of anyone's reads.
  const int grp = wvu >> 2;
  asm volatile("" ::: "memory");
  __builtin_amdgcn_s_barrier();          // every wave's copies of stage 0 have landed
  asm volatile("" ::: "memory");
  if (grp) {
    __builtin_amdgcn_sched_barrier(0);
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_sched_barrier(0);
  }
  for (int ks = 0; ks < nk; ks++) {
    int fill = slot - 1;
    fill = fill < 0 ? NS - 1 : fill;
    u32x4 wf[FC], xf[FP];
    // copies first: the texture path works on them while the LDS serves the fragment reads
    issue_stage(fill);
    __builtin_amdgcn_sched_barrier(0);
    {
      const u32x4* sWs = &smem[slot * STAGE];
      const u32x4* sXs = sWs + WR * 4;
#pragma unroll
      for (int a = 0; a < FC; a++) {
        const int row = wcI * WC + a * 16 + fr;
        wf[a] = sWs[row * 4 + (fg ^ ((row >> 1) & 3))];
      }
#pragma unroll
      for (int b = 0; b < FP; b++) {
        const int row = wpI * WP + b * 16 + fr;
        xf[b] = sXs[row * 4 + (fg ^ ((row >> 1) & 3))];
      }
    }
    next_lookup();
    asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)" ::"n"((NS - 2) * PPS) : "memory");
    __builtin_amdgcn_sched_barrier(0);
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_sched_barrier(0);
    __builtin_amdgcn_s_setprio(1);
#pragma unroll
    for (int a = 0; a < FC; a++)
#pragma unroll
      for (int b = 0; b < FP; b++) mma_chunk<BF16>(acc[a][b], wf[a], xf[b]);
    next_offsets();                        // 2 VALU adds in the MFMA shadow
    __builtin_amdgcn_s_setprio(0);
    __builtin_amdgcn_sched_barrier(0);
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_sched_barrier(0);
    slot = slot + 1 == NS ? 0 : slot + 1;
  }
  if (!grp) {
    __builtin_amdgcn_sched_barrier(0);
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_sched_barrier(0);
  }
  } else {
    // ---- 4-wave geometries: one raw barrier per k-step, the copies of stage ks+NS-1 issued one at a time
    // between the MFMA groups.  Used for launches that do not fill the chip (<= 1 block per CU): there the
    // 2-stage kernel exposes a full memory latency per k-step, the counted-vmcnt ring hides it.
    for (int ks = 0; ks < nk; ks++) {
      wait_vmcnt<(NS - 2) * PPS>();
      asm volatile("" ::: "memory");
      __builtin_amdgcn_s_barrier();
      asm volatile("" ::: "memory");
      int fill = slot - 1;                 // slot of stage ks-1 == slot of stage ks+NS-1
      fill = fill < 0 ? NS - 1 : fill;
      u32x4 wf[FC], xf[FP];
      {
        const u32x4* sWs = &smem[slot * STAGE];
        const u32x4* sXs = sWs + WR * 4;
#pragma unroll
        for (int a = 0; a < FC; a++) {
          const int row = wcI * WC + a * 16 + fr;
          wf[a] = sWs[row * 4 + (fg ^ ((row >> 1) & 3))];
        }
#pragma unroll
        for (int b = 0; b < FP; b++) {
          const int row = wpI * WP + b * 16 + fr;
          xf[b] = sXs[row * 4 + (fg ^ ((row >> 1) & 3))];
        }
      }
#pragma unroll
      for (int a = 0; a < FC; a++) {
#pragma unroll
        for (int j = (a * PPS) / FC; j < ((a + 1) * PPS) / FC; j++) issue_piece(fill, j);
        if (a == FC - 1) next_lookup();
        __builtin_amdgcn_s_setprio(1);
#pragma unroll
        for (int b = 0; b < FP; b++) mma_chunk<BF16>(acc[a][b], wf[a], xf[b]);
        __builtin_amdgcn_s_setprio(0);
      }
      next_offsets();
      slot = slot + 1 == NS ? 0 : slot + 1;
    }
  }
  MT_STAMP(2);
  wait_vmcnt<0>();     // the trailing (all-zero) copies must have landed before LDS is reused / the wave exits

  // ---- epilogue: bias + activation, packed NHWC store, optional statistics ----
  char* yp[FP];
#pragma unroll
  for (int b = 0; b < FP; b++) {
    const int m = pt * PT + wpI * WP + b * 16 + fr;
    yp[b] = nullptr;
    if (m < ph_M) {
      const int n = m / HoWo;
      const int rem = m - n * HoWo;
      const int ho = rem / ph_Wo;
      const int wo = rem - ho * ph_Wo;
      const int oh = ho * p.os + ph.oh0, ow = wo * p.os + ph.ow0;
      if ((unsigned)oh < (unsigned)p.Hout && (unsigned)ow < (unsigned)p.Wout)
        yp[b] = p.y + (((size_t)n * p.Hout + oh) * p.Wout + ow) * p.Co * SZ;
    }
  }
  const bool do_stats = p.stats != nullptr;
  float* red = reinterpret_cast<float*>(&smem[0]);      // [NWP][WT][2]
  if (do_stats) __syncthreads();
  // the straight-line epilogue (conv_device.h: epilogue_perm, round 4); tanh and outputs beyond 32-bit buffer offsets keep the
  // general code below
  const size_t ybytes = (size_t)p.N * p.Hout * p.Wout * p.Co * SZ;
  if (p.act != MT_ACT_TANH && ybytes < 0x7f000000ull) {
    unsigned yo[FP];
    float vm[FP];
#pragma unroll
    for (int b = 0; b < FP; b++) {
      yo[b] = yp[b] != nullptr ? (unsigned)(yp[b] - p.y) : EPI_OOB;
      vm[b] = yp[b] != nullptr ? 1.f : 0.f;
    }
    const unsigned cob = (unsigned)(wt * WT + wcI * WC + fg * 8);
    float* const red_lane = red + (wpI * WT + wcI * WC + fg * 8) * 2;
    if (do_stats)
      epilogue_perm<BF16, FC, FP, false, true, false, 2>(acc, p.y, (unsigned)ybytes, p.bias, p.nbias, nullptr, p.act, p.slope, yo, vm,
                                                         cob, p.Co, red_lane, fr);
    else
      epilogue_perm<BF16, FC, FP, false, false, false, 2>(acc, p.y, (unsigned)ybytes, p.bias, p.nbias, nullptr, p.act, p.slope, yo, vm,
                                                          cob, p.Co, red_lane, fr);
  } else
  // Weight rows were staged in permuted order (see rl_perm above): fragment pair (2s, 2s+1) of lane-group fg
  // holds the 8 CONSECUTIVE channels s*32 + fg*8 .. +7 of one pixel, so a lane stores 16 bytes (bf16) per
  // pixel fragment and the four lane-groups of a store instruction cover 64 contiguous bytes per pixel --
  // half the store instructions of the 4-channel layout (the store tail is issue-bound, MI355X_MICROARCH.md).
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
#pragma unroll
    for (int b = 0; b < FP; b++) {
      if (yp[b] == nullptr) continue;
      float v[8];
#pragma unroll
      for (int e = 0; e < 8; e++) {
        v[e] = act_apply(acc[2 * sp + (e >> 2)][b][e & 3] + bv[e], p.act, p.slope);
        s1[e] += v[e];
        s2[e] += v[e] * v[e];
      }
      if constexpr (BF16) {
        u32x4 o = {pack2_bf16(v[0], v[1]), pack2_bf16(v[2], v[3]), pack2_bf16(v[4], v[5]), pack2_bf16(v[6], v[7])};
        // streaming store: the tile is not read again by this kernel, keep the L2 for the operand re-reads (+2-4 %)
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
    const int m0 = pt * PT;
    if (m0 < ph_M) {
      const int n0 = m0 / HoWo;
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
  MT_STAMP(3);
}

// geometry choice shared with the launcher in conv_kernels.hip:
//   WT = 256: 256 couts x 256 pixels, 4-stage ring, <= 25 taps
//   WT = 128: 128 couts x 512 pixels (8 waves along the pixels), 3-stage ring, <= 9 taps -- the Cout = 128 layers
//   128 x 128 / 64 x 128 with 4 waves: the ring without ping-pong, for launches of at most one block per CU
template <bool BF16>
int launch_igemm_pipe_t(IgemmParams& p, int WT, int PT, int total, hipStream_t s) {
  if (WT == 256 && PT == 256)
    hipLaunchKernelGGL((igemm_pipe_kernel<BF16, 256, 256, 512, 4, MT_PIPE_MAX_TAPS>), dim3(total), dim3(512), 0, s, p);
  else if (WT == 128 && PT == 512)
    hipLaunchKernelGGL((igemm_pipe_kernel<BF16, 128, 512, 512, 3, 9>), dim3(total), dim3(512), 0, s, p);
  else if (WT == 128 && PT == 128)
    hipLaunchKernelGGL((igemm_pipe_kernel<BF16, 128, 128, 256, 4, 9>), dim3(total), dim3(256), 0, s, p);
  else if (WT == 64 && PT == 128)
    hipLaunchKernelGGL((igemm_pipe_kernel<BF16, 64, 128, 256, 4, 9>), dim3(total), dim3(256), 0, s, p);
  else { mt_set_error("igemm_pipe: no instantiation for %d x %d", WT, PT); return 1; }
  MT_LAUNCH_CHECK();
  return 0;
}
template int launch_igemm_pipe_t<true>(IgemmParams&, int, int, int, hipStream_t);
template int launch_igemm_pipe_t<false>(IgemmParams&, int, int, int, hipStream_t);
