// Software-pipelined gather-GEMM (same math and parameter block as igemm_kernel in conv_kernels.hip).
//
// 32-deep k-steps (64-byte tile rows), an NS-stage LDS ring filled by LDS-DMA
// (buffer_load_dwordx4 ... lds) that runs NS-1 stages ahead, ONE raw s_barrier per k-step and a COUNTED
// s_waitcnt vmcnt((NS-2)*PPS): the copies of the newest NS-2 stages stay in flight across the barrier
// (a __syncthreads() would drain them with vmcnt(0) -- with the 2-stage kernel 51 % of the wave cycles
// sit in that wait).  Every wave issues exactly PPS copies per stage (out-of-range offsets -> zeros, also
// past the end of K), so the count is a compile-time constant in the steady state and in the tail.
//
//   iteration ks:  s_waitcnt vmcnt((NS-2)*PPS)  -> this wave's copies of stage ks have landed
//                  s_barrier                    -> everybody's have; everybody finished reading ks-1
//                  ds_read fragments of stage ks; MFMAs, interleaved with the copies of stage ks+NS-1
//                  (which overwrite the ring slot of stage ks-1)
#include "conv_device.h"
#include <type_traits>

#ifdef MT_STAMPS
// diagnostic build only (make STAMPS=1): s_memtime stamps of wave 0 of every block -> tools/stamp_k1.py
__device__ unsigned long long mt_stamp_buf[8 * 4096];
#define MT_STAMP(i) do { if (threadIdx.x == 0 && blockIdx.x < 4096) mt_stamp_buf[blockIdx.x * 8 + (i)] = __builtin_amdgcn_s_memtime(); } while (0)
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

template <bool BF16, int WT, int PT, int NT, int NS>
__global__ __launch_bounds__(512) void igemm_pipe_kernel(const IgemmParams p) {
  constexpr int NW = NT / 64;
  constexpr int WC = (WT == 256) ? 128 : ((WT >= 64) ? 64 : WT);   // wave tile: output channels
  constexpr int WP = (WT >= 128) ? 64 : 32;                        // wave tile: pixels
  constexpr int NWP = PT / WP;
  constexpr int FC = WC / 16, FP = WP / 16;
  constexpr int SZ = BF16 ? 2 : 4;
  constexpr int NXL = PT / 16 / NW;                 // pixel-tile copies per wave per stage
  constexpr int NWL = ((WT + 15) / 16 + NW - 1) / NW;   // weight-tile copies per wave per stage (incl. dummies)
  constexpr int WR = NWL * NW * 16;                 // weight rows allocated per stage (>= WT)
  constexpr int PPS = NXL + NWL;                    // copies per wave per stage
  constexpr int STAGE = (WR + PT) * 4;              // u32x4 per stage (4 chunks per 64-byte row)
  static_assert((NT / 64) == (PT / WP) * (WT / WC), "wave grid must cover the block tile");
  static_assert(NXL * NW * 16 == PT, "pixel tile must be a whole number of copy instructions per wave");
  static_assert(NS >= 3 && (NS - 2) * PPS <= 63, "vmcnt range");
  static_assert(FC % 2 == 0, "the epilogue stores fragment pairs");

  // ONE shared array (a second __shared__ object next to an LDS-DMA target makes hipcc drain vmcnt)
  __shared__ u32x4 smem[NS * STAGE + 16];
  int* sTap = reinterpret_cast<int*>(&smem[NS * STAGE]);

  MT_STAMP(0);
  const int tid = threadIdx.x;
  const int lane = tid & 63, wv = tid >> 6;
  const int fr = lane & 15, fg = lane >> 4;
  const int wcI = wv / NWP, wpI = wv % NWP;
  const int wvu = __builtin_amdgcn_readfirstlane(wv);

  const int nWT = (p.CoRows + WT - 1) / WT;
  int wg = xcd_remap(blockIdx.x, gridDim.x);
  int phi = 0;
  for (int i = 1; i < p.nphase; i++) phi = (wg >= p.ph[i].blk0) ? i : phi;
  const IgemmPhase& ph = p.ph[phi];
  const int ph_ntaps = ph.ntaps, ph_Ho = ph.Ho, ph_Wo = ph.Wo, ph_M = ph.M;
  const int ph_nchunks = ph_ntaps * p.cpc;
  const char* const ph_w = p.w + ph.w_off;
  wg -= ph.blk0;
  if (tid < 64) {
    const int t = ph.tap0 + tid;
    sTap[tid] = tid < ph_ntaps ? (((int)p.dh[t] << 16) | ((int)p.dw[t] & 0xffff)) : 0;
  }
  const int wt = wg % nWT, pt = wg / nWT;

  // ---- staging coordinates: a copy instruction writes 64 lanes x 16 B = 16 tile rows lane-linearly; the
  // bank swizzle (chunk ^ ((row >> 1) & 3), conflict-free for the ds_read_b128 fragment reads of 64-byte
  // rows) is applied to the SOURCE chunk
  const int rsub = lane >> 2;
  const int c = (lane & 3) ^ ((rsub >> 1) & 3);
  const int HoWo = ph_Ho * ph_Wo;
  int hb[NXL], wb[NXL];
  unsigned ib[NXL];
  unsigned rvm = 0;
#pragma unroll
  for (int i = 0; i < NXL; i++) {
    const int m = pt * PT + 16 * (wvu + NW * i) + rsub;
    const bool rv = m < ph_M;
    rvm |= (rv ? 1u : 0u) << i;
    const int mm = rv ? m : 0;
    const int n = mm / HoWo;
    const int rem = mm - n * HoWo;
    const int ho = rem / ph_Wo;
    const int wo = rem - ho * ph_Wo;
    hb[i] = ho * p.is;
    wb[i] = wo * p.is;
    ib[i] = (unsigned)n * (unsigned)(p.Hi * p.Wi) * (unsigned)p.Cib;
  }
  int q = c;
  int tap = q / p.cpc;
  int cq = q - tap * p.cpc;
  const int step_t = 4 / p.cpc, step_r = 4 % p.cpc;

  unsigned wo32[NWL];
  unsigned wokm = 0;
#pragma unroll
  for (int i = 0; i < NWL; i++) {
    const int rs = 16 * (wvu + NW * i) + rsub;           // LDS row of the weight tile (fragment order)
    // channel held by that row: within each 32-row fragment pair, row (a&1)*16 + r <- channel (r>>2)*8 + (a&1)*4 + (r&3)
    const int rl = (rs & ~31) | ((((rs & 15) >> 2) << 3) | (((rs >> 4) & 1) << 2) | (rs & 3));
    const int row = wt * WT + rl;
    const bool ok = (rs < WT) && (row < p.CoRows);
    wokm |= (ok ? 1u : 0u) << i;
    wo32[i] = ((unsigned)(ok ? row : 0) * (unsigned)ph_nchunks + (unsigned)c) * 16u;
  }
  unsigned xo32[2];
  static_assert(NXL <= 2, "xo32 is literal-sized (hipcc drops the host stub for a dependent-size lambda capture)");
  unsigned xokm = 0;
  auto retap = [&]() {
    xokm = 0;
#pragma unroll
    for (int i = 0; i < NXL; i++) xo32[i] = 0xfffffff0u;
    if (tap < ph_ntaps) {
      const int t = sTap[tap];
      const int dh = t >> 16, dw = (int)(short)(t & 0xffff);
#pragma unroll
      for (int i = 0; i < NXL; i++) {
        int hi = hb[i] + dh, wi = wb[i] + dw;
        bool ok = (rvm >> i) & 1u;
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
        xo32[i] = ok ? ib[i] + (unsigned)(hi * p.Wi + wi) * (unsigned)p.Cib + (unsigned)cq * 16u : 0xfffffff0u;
        xokm |= (ok ? 1u : 0u) << i;
      }
    }
  };

  const __amdgpu_buffer_rsrc_t rsx = __builtin_amdgcn_make_buffer_rsrc((void*)p.x, 0, p.x_bytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t rsw =
      __builtin_amdgcn_make_buffer_rsrc((void*)ph_w, 0, (unsigned)p.CoRows * (unsigned)ph_nchunks * 16u, 0x00020000);
  typedef __attribute__((address_space(3))) void* lds_ptr;
  char* const lds0 = reinterpret_cast<char*>(&smem[0]);

  // one copy instruction of the stage being filled into ring slot `slot` (j in [0, PPS))
  auto issue_piece = [&](int slot, int j) {
    char* base = lds0 + slot * (STAGE * 16);
    if (j < NWL) {
      const int i = j;
      const unsigned o = ((wokm >> i) & 1u) ? wo32[i] : 0xfffffff0u;
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rsw, (lds_ptr)(base + (wvu + NW * i) * 1024), 16, o, 0, 0, 0);
      wo32[i] += 64u;
    } else {
      const int i = j - NWL;
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rsx, (lds_ptr)(base + WR * 64 + (wvu + NW * i) * 1024), 16, xo32[i], 0, 0,
                                               0);
    }
  };
  auto issue_end = [&]() {   // advance the gather state by one k-step (4 chunks)
    q += 4;
    const int otap = tap;
    tap += step_t;
    cq += step_r;
    if (cq >= p.cpc) { cq -= p.cpc; tap++; }
    if (tap != otap) {
      retap();
    } else {
#pragma unroll
      for (int i = 0; i < NXL; i++) xo32[i] += ((xokm >> i) & 1u) ? 64u : 0u;
    }
  };

  f32x4 acc[FC][FP];
#pragma unroll
  for (int a = 0; a < FC; a++)
#pragma unroll
    for (int b = 0; b < FP; b++) acc[a][b] = f32x4{0.f, 0.f, 0.f, 0.f};

  const int nk = (ph_nchunks + 3) >> 2;
  __syncthreads();  // tap table visible
  retap();
  // prologue: stages 0 .. NS-2
#pragma unroll
  for (int s = 0; s < NS - 1; s++) {
#pragma unroll
    for (int j = 0; j < PPS; j++) issue_piece(s, j);
    issue_end();
  }
  if constexpr (NW == 8) wait_vmcnt<(NS - 2) * PPS>();
  MT_STAMP(1);

  auto read_frags = [&](int slot_, u32x4* wf, u32x4* xf) {
    const u32x4* sWs = &smem[slot_ * STAGE];
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
  };
  int slot = 0;   // ring slot of stage ks
  if constexpr (NW == 8) {
    // ---- ping-pong: waves 0-3 and 4-7 (one of each per SIMD) alternate between a MEMORY phase (fragment
    // reads of stage ks, LDS-DMA of stage ks+NS-1) and a COMPUTE phase (32 back-to-back MFMAs), half a k-step
    // apart, so each SIMD's matrix pipe always has one wave feeding it while the other waits on LDS / the
    // texture path.  Every wave executes the same number of s_barriers (2*nk + 1).
    const int grp = wvu >> 2;
    asm volatile("" ::: "memory");
    __builtin_amdgcn_s_barrier();          // every wave's prologue copies of stage 0 have landed
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
      read_frags(slot, wf, xf);
#pragma unroll
      for (int j = 0; j < PPS; j++) issue_piece(fill, j);
      issue_end();
      // this wave's copies of stage ks+1 have landed; its fragments are in registers
      asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)" ::"n"((NS - 2) * PPS) : "memory");
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
    if (!grp) {
      __builtin_amdgcn_sched_barrier(0);
      __builtin_amdgcn_s_barrier();
      __builtin_amdgcn_sched_barrier(0);
    }
  } else {
    for (int ks = 0; ks < nk; ks++) {
      wait_vmcnt<(NS - 2) * PPS>();
      asm volatile("" ::: "memory");
      __builtin_amdgcn_s_barrier();
      asm volatile("" ::: "memory");
      int fill = slot - 1;                 // slot of stage ks-1 == slot of stage ks+NS-1
      fill = fill < 0 ? NS - 1 : fill;
      u32x4 wf[FC], xf[FP];
      read_frags(slot, wf, xf);
#pragma unroll
      for (int a = 0; a < FC; a++) {
        // the PPS copies of stage ks+NS-1 are spread over the first MFMA groups
        if (FC >= PPS) { if (a < PPS) issue_piece(fill, a); }
        else {
#pragma unroll
          for (int j = (a * PPS) / FC; j < ((a + 1) * PPS) / FC; j++) issue_piece(fill, j);
        }
        __builtin_amdgcn_s_setprio(1);
#pragma unroll
        for (int b = 0; b < FP; b++) mma_chunk<BF16>(acc[a][b], wf[a], xf[b]);
        __builtin_amdgcn_s_setprio(0);
      }
      issue_end();
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
        *reinterpret_cast<u32x4*>(yp[b] + (size_t)co * 2) = o;
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

// geometry choice shared with the launcher in conv_kernels.hip
template <bool BF16>
int launch_igemm_pipe_t(IgemmParams& p, int WT, int PT, int total, hipStream_t s) {
  if (WT == 256) hipLaunchKernelGGL((igemm_pipe_kernel<BF16, 256, 256, 512, 4>), dim3(total), dim3(512), 0, s, p);
  else { mt_set_error("igemm_pipe: no instantiation for WT=%d", WT); return 1; }
  (void)PT;
  MT_LAUNCH_CHECK();
  return 0;
}
template int launch_igemm_pipe_t<true>(IgemmParams&, int, int, int, hipStream_t);
template int launch_igemm_pipe_t<false>(IgemmParams&, int, int, int, hipStream_t);
