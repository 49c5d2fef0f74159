// Ping-pong weight-gradient kernel for 256 x 256 output tiles (bf16; same parameter block and slab layout as
// wgrad_kernel in conv_kernels.hip, chosen by mt_launch_wgrad).  gfx950 only.
//
//   dW[a][tap][c] = sum over pixels m of  A[m][a] * B[gather(m, tap)][c]        (A = dY, B = x for Conv2d)
//
// Every GEMM kernel of this library ends up bound by what the LDS-DMA path delivers (~10 TB/s over the chip, see
// DESIGN.md); the 128 x 128 tile of wgrad_kernel moves twice the bytes per MAC of a 256 x 256 tile.  This kernel
// is the structure of igemm_pipe_kernel (conv_pipe_kernel.hip) with the reduction over PIXELS:
//  * tile = 256 a-channels x 256 columns (= one filter tap x 256 b-channels), 8 waves, wave tile 128 x 64;
//  * 32-pixel k-steps: both operands are staged pixel-major (512-byte rows) in a 4-stage LDS ring by LDS-DMA with a
//    counted s_waitcnt vmcnt; fragments are read with the transposing ds_read_b64_tr_b16 (32-byte slots XOR-ed
//    with row bits {0,1,3}: conflict-free, swizzle applied on the source side of the lane-linear LDS-DMA);
//  * two wave groups alternate memory / compute phases half a k-step apart (ping-pong);
//  * the gathered operand's per-pixel source offsets (tap shift, reflection / zero padding) are resolved once per
//    block into an LDS table, so the main loop has no address arithmetic beyond table entry + chunk offset.
// Each (tile, pixel split) block writes its own fp32 slab; unpack_kernel sums the slabs.
#include "conv_device.h"

constexpr int MT_WGP_MAXM = 7168;   // entries of the per-block pixel-offset table (28 KiB)

template <int N>
__device__ __forceinline__ void wgp_wait_vmcnt() {
  asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}

// Transposing LDS read as inline asm: hipcc models the ds_read_tr16 builtin as aliasing the pending LDS-DMA writes
// and puts an s_waitcnt vmcnt(0) in front of it (the whole ring drained every k-step: +40 us on K1).  The asm form
// is invisible to that pass; ordering is ours: the reads of a stage come after the barrier that follows its
// counted vmcnt wait, and they are retired by the explicit lgkmcnt(0) before the MFMAs.
template <int OFF>
__device__ __forceinline__ s16x4 lds_tr16(unsigned addr) {
  s16x4 r;
  asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=v"(r) : "v"(addr), "n"(OFF) : "memory");
  return r;
}

// TB = bits of a table entry.  32: byte offsets (one problem per launch).  16 / 8 (grouped launches, whose splits are longer than
// the 32-bit table holds): signed PIXEL deltas (gathered pixel - own pixel; stride 1 and coinciding pixel grids, so |delta| <=
// pad * (W + 1): 8 bits serve maps up to 126 pixels wide at pad 1), the byte offset is formed in the memory phase (one
// multiply-add and a select per copy); the most negative value marks "no pixel".
template <bool REFLECT, int NS, int TB>
__global__ __launch_bounds__(512) void wgrad_pipe_kernel(const WgradParams p) {
  constexpr bool COMPACT = TB != 32;
  constexpr unsigned NOPIX = TB == 16 ? 0x8000u : 0x80u;           // table code of "no pixel"
  constexpr unsigned NOPIX_SX = TB == 16 ? 0xffff8000u : 0xffffff80u;   // ... sign-extended
  constexpr int KP = 32;                    // pixels per k-step = K of one v_mfma_f32_16x16x32_bf16
  constexpr int ROWB = 512;                 // LDS row: 256 bf16 of one pixel
  constexpr int TILE = KP * ROWB / 16;      // u32x4 per operand tile (16 KiB)
  constexpr int STAGE = 2 * TILE;           // A tile | B tile
  constexpr int PPS = 4;                    // copies per wave per stage: 2 (A) + 2 (B), 1 KiB = 2 pixel rows each
  constexpr unsigned OOB = 0x80000000u;     // offsets >= 2 GiB are out of range of both buffers -> zeros
  static_assert(NS * STAGE * 16 + MT_WGP_MAXM * 4 <= 160 * 1024, "LDS budget");

  // ONE shared array (a second __shared__ object next to an LDS-DMA target makes hipcc drain vmcnt)
  __shared__ u32x4 smem[NS * STAGE + MT_WGP_MAXM / 4];
  unsigned* const sT = reinterpret_cast<unsigned*>(&smem[NS * STAGE]);

  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
  const int wvu = __builtin_amdgcn_readfirstlane(wv);
  const int waI = wv >> 2, wbI = wv & 3;    // wave tile: 128 a-channels x 64 columns; waI is also the ping-pong group

  const int nAT = p.CaRows >> 8;
  const int vid = xcd_remap(blockIdx.x, gridDim.x);
  // (split, problem of the group, tile): all tiles of all problems of a pixel split on one XCD
  const int ngrp = p.ngroup > 1 ? p.ngroup : 1;
  const int ntt = p.ntiles * ngrp;
  const int split = vid / ntt, rest = vid - split * ntt;
  const int gi = rest / p.ntiles, tile = rest - gi * p.ntiles;
  const char* const pa = p.ngroup > 1 ? p.ga[gi] : p.a;
  const char* const pb = p.ngroup > 1 ? p.gb[gi] : p.b;
  float* const pout = p.ngroup > 1 ? p.gout[gi] : p.out;
  const int at = tile % nAT, bt = tile / nAT;
  const int mbeg = split * p.mchunk;
  const int mend = min(p.M, mbeg + p.mchunk);   // host guarantees mbeg < mend
  const int nk = (mend - mbeg + KP - 1) / KP;

  // column tile -> filter tap and 256-channel block of the gathered operand
  const int cb_per_tap = (p.cpc * 8) >> 8;
  const int btap = bt / cb_per_tap;
  const unsigned bch_bytes = (unsigned)(bt - btap * cb_per_tap) * 512u;
  {
    const int bdh = p.dh[btap], bdw = p.dw[btap];
    const int HoWo = p.Ho * p.Wo;
    const int nent = (nk + NS) * KP;            // the pipeline looks NS stages past the end (all zero rows)
    for (int i = tid; i < nent; i += 512) {
      const int m = mbeg + i;
      unsigned off = OOB;
      if (m < mend) {
        const int n = m / HoWo;
        const int rem = m - n * HoWo;
        const int ho = rem / p.Wo;
        const int wo = rem - ho * p.Wo;
        int hi = ho * p.is + bdh, wi = wo * p.is + bdw;
        bool ok = true;
        if constexpr (REFLECT) {
          hi = hi < 0 ? -hi : hi;
          hi = hi >= p.Hi ? 2 * (p.Hi - 1) - hi : hi;
          wi = wi < 0 ? -wi : wi;
          wi = wi >= p.Wi ? 2 * (p.Wi - 1) - wi : wi;
        } else {
          ok = ((unsigned)hi < (unsigned)p.Hi) && ((unsigned)wi < (unsigned)p.Wi);
        }
        if constexpr (COMPACT) {
          if (ok) off = (unsigned)(((n * p.Hi + hi) * p.Wi + wi) - m) & (TB == 16 ? 0xffffu : 0xffu);
          else off = NOPIX;
        } else {
          if (ok) off = (unsigned)((n * p.Hi + hi) * p.Wi + wi) * (unsigned)p.Cbb + bch_bytes;
        }
      } else if constexpr (COMPACT) {
        off = NOPIX;
      }
      if constexpr (TB == 16) reinterpret_cast<unsigned short*>(sT)[i] = (unsigned short)off;
      else if constexpr (TB == 8) reinterpret_cast<unsigned char*>(sT)[i] = (unsigned char)off;
      else sT[i] = off;
    }
  }

  // ---- staging coordinates: a copy instruction writes 64 lanes x 16 B = two 512-byte pixel rows lane-linearly;
  // the 32-byte-slot swizzle (slot ^= key(row), key = row bits {0,1,3}) is applied to the SOURCE chunk
  const int rsub = lane >> 5, pcs = lane & 31;
  int srow[2];
  unsigned ao32[2], xo32[2], cc16[2];
  int mrow[2];
#pragma unroll
  for (int i = 0; i < 2; i++) {
    srow[i] = 2 * (wvu + 8 * i) + rsub;                       // pixel row of the stage tile (0..31)
    const int key = (srow[i] & 3) | ((srow[i] >> 1) & 4);
    const int cc = pcs ^ (key << 1);                           // 16-byte chunk of the pixel this lane fetches
    cc16[i] = (unsigned)cc * 16u;
    mrow[i] = mbeg + srow[i];
    ao32[i] = (unsigned)mrow[i] * (unsigned)p.Cab + (unsigned)at * 512u + cc16[i];
  }
  const __amdgpu_buffer_rsrc_t rsa = __builtin_amdgcn_make_buffer_rsrc((void*)pa, 0, p.a_bytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t rsb = __builtin_amdgcn_make_buffer_rsrc((void*)pb, 0, p.b_bytes, 0x00020000);
  typedef __attribute__((address_space(3))) void* lds_ptr;
  char* const lds0 = reinterpret_cast<char*>(&smem[0]);
  const unsigned a_step = (unsigned)KP * (unsigned)p.Cab;

  auto issue_stage = [&](int slot) {
    char* base = lds0 + slot * (STAGE * 16);
#pragma unroll
    for (int i = 0; i < 2; i++) {
      const unsigned oa = mrow[i] < mend ? ao32[i] : OOB;
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rsa, (lds_ptr)(base + (wvu + 8 * i) * 1024), 16, oa, 0, 0, 0);
      ao32[i] += a_step;
      mrow[i] += KP;
    }
#pragma unroll
    for (int i = 0; i < 2; i++)
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rsb, (lds_ptr)(base + TILE * 16 + (wvu + 8 * i) * 1024), 16, xo32[i], 0, 0,
                                               0);
  };
  int tpos = 0;                 // table row block of the NEXT stage to issue
  unsigned tq[2];
  const short* const sT16 = reinterpret_cast<const short*>(sT);
  const unsigned cbb = (unsigned)p.Cbb;      // (a local: a lambda that touches p by reference sends the whole parameter block to scratch)
  auto lookup = [&](int i) -> unsigned {
    if constexpr (TB == 16) return (unsigned)(int)sT16[tpos + srow[i]];     // sign-extended pixel delta
    else if constexpr (TB == 8) return (unsigned)(int)reinterpret_cast<const signed char*>(sT)[tpos + srow[i]];
    else return sT[tpos + srow[i]];
  };
  auto offset_of = [&](int i, unsigned t) -> unsigned {
    if constexpr (COMPACT) {
      const unsigned o = (unsigned)(mbeg + tpos + srow[i] + (int)t) * cbb + bch_bytes + cc16[i];
      return t == NOPIX_SX ? OOB : o;
    } else {
      return t + cc16[i];
    }
  };
  auto next_lookup = [&]() {
    tpos += KP;
#pragma unroll
    for (int i = 0; i < 2; i++) tq[i] = lookup(i);
  };
  auto next_offsets = [&]() {      // (always called with tpos still at the stage next_lookup read)
#pragma unroll
    for (int i = 0; i < 2; i++) xo32[i] = offset_of(i, tq[i]);
  };

  f32x4 acc[8][4];
#pragma unroll
  for (int a = 0; a < 8; a++)
#pragma unroll
    for (int b = 0; b < 4; b++) acc[a][b] = f32x4{0.f, 0.f, 0.f, 0.f};

  __syncthreads();   // offset table visible
#pragma unroll
  for (int i = 0; i < 2; i++) xo32[i] = offset_of(i, lookup(i));
#pragma unroll
  for (int s = 0; s < NS - 1; s++) {
    issue_stage(s);
    next_lookup();
    next_offsets();
  }
  wgp_wait_vmcnt<(NS - 2) * PPS>();

  // ---- ping-pong main loop (see conv_pipe_kernel.hip for the ordering argument) ----
  const int g = lane >> 4, qq = (lane & 15) >> 2, pp = lane & 3;
  const int prow = 8 * g + qq;                                   // rows prow and prow+4 of the 32-pixel tile
  const int rkey = (prow & 3) | ((prow >> 1) & 4);
  // per-lane fragment offsets inside a stage (the XOR makes them non-affine in f: kept in registers)
  typedef __attribute__((address_space(3))) char* lds_char_ptr;
  const unsigned lds_base = (unsigned)(size_t)(lds_char_ptr)lds0;
  unsigned aoff[8], boff[4];
#pragma unroll
  for (int f = 0; f < 8; f++) aoff[f] = (unsigned)(prow * ROWB + pp * 8 + (((waI * 8 + f) ^ rkey) * 32));
#pragma unroll
  for (int f = 0; f < 4; f++) boff[f] = (unsigned)(TILE * 16 + prow * ROWB + pp * 8 + (((wbI * 4 + f) ^ rkey) * 32));
  const int grp = wvu >> 2;
  int slot = 0;
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
    issue_stage(fill);
    __builtin_amdgcn_sched_barrier(0);
    bf16x8 af[8], bf[4];
    {
      const unsigned sb = lds_base + (unsigned)slot * (STAGE * 16);
#pragma unroll
      for (int f = 0; f < 8; f++) {
        const unsigned pa = sb + aoff[f];
        const s16x4 a0 = lds_tr16<0>(pa);
        const s16x4 a1 = lds_tr16<4 * ROWB>(pa);
        af[f] = __builtin_bit_cast(bf16x8, __builtin_shufflevector(a0, a1, 0, 1, 2, 3, 4, 5, 6, 7));
      }
#pragma unroll
      for (int f = 0; f < 4; f++) {
        const unsigned pb = sb + boff[f];
        const s16x4 b0 = lds_tr16<0>(pb);
        const s16x4 b1 = lds_tr16<4 * ROWB>(pb);
        bf[f] = __builtin_bit_cast(bf16x8, __builtin_shufflevector(b0, b1, 0, 1, 2, 3, 4, 5, 6, 7));
      }
    }
    next_lookup();
    asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)" ::"n"((NS - 2) * PPS) : "memory");
    __builtin_amdgcn_sched_barrier(0);
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_sched_barrier(0);
    __builtin_amdgcn_s_setprio(1);
    // column operand as MFMA A: a lane ends with 4 consecutive columns of one a-channel (16-byte slab stores)
#pragma unroll
    for (int a = 0; a < 8; a++)
#pragma unroll
      for (int b = 0; b < 4; b++)
        acc[a][b] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bf[b], af[a], acc[a][b], 0, 0, 0);
    next_offsets();
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
  wgp_wait_vmcnt<0>();   // the trailing (all-zero) copies must have landed before the wave exits

  // ---- epilogue: this split's fp32 slab [CaRows][ncols] ----
  const int fr = lane & 15, fg = lane >> 4;
  const int ncols = p.nchunks * 8;
  float* slab = pout + (size_t)split * p.CaRows * ncols;
#pragma unroll
  for (int a = 0; a < 8; a++) {
    const int ca = at * 256 + waI * 128 + a * 16 + fr;
#pragma unroll
    for (int b = 0; b < 4; b++) {
      const int col = bt * 256 + wbI * 64 + b * 16 + fg * 4;
      *reinterpret_cast<f32x4*>(slab + (size_t)ca * ncols + col) = acc[a][b];
    }
  }
}

// Does the 256 x 256 ping-pong kernel cover this problem?  (shared by the launcher and the split heuristic)
bool mt_wgrad_pipe_ok(int dtype, int CaRows, int cpc, long a_bytes, long b_bytes) {
  return dtype == MT_BF16 && CaRows % 256 == 0 && (cpc * 8) % 256 == 0 && a_bytes < 0x7f000000L && b_bytes < 0x7f000000L;
}
int mt_wgrad_pipe_max_chunk() { return MT_WGP_MAXM - 6 * 32; }
int mt_wgrad_pipe_max_chunk_compact() { return 2 * MT_WGP_MAXM - 6 * 32; }
int mt_wgrad_pipe_max_chunk_compact8() { return 4 * MT_WGP_MAXM - 6 * 32; }
bool mt_wgrad_pipe_compact_ok(const WgradParams& p) {
  return p.is == 1 && p.Hi == p.Ho && p.Wi == p.Wo && (long)p.Wi * 4 + 4 < 32000;
}
// largest |gathered pixel - own pixel| of the problem's taps (zero / reflection padding never reach further)
static int wgrad_max_delta(const WgradParams& p) {
  int m = 0;
  for (int t = 0; t < p.ntaps; t++) {
    const int d = abs((int)p.dh[t]) * p.Wi + abs((int)p.dw[t]);
    m = d > m ? d : m;
  }
  return m;
}
bool mt_wgrad_pipe_compact8_ok(const WgradParams& p) { return mt_wgrad_pipe_compact_ok(p) && wgrad_max_delta(p) <= 126; }

int mt_launch_wgrad_pipe(const WgradParams& p, int nsplit, hipStream_t s) {
  const int ngrp = p.ngroup > 1 ? p.ngroup : 1;
  MT_CHECK(ngrp <= MT_WGRAD_MAX_GROUP, "wgrad_pipe: group of %d", ngrp);
  int tb = 32;
  if (p.mchunk > mt_wgrad_pipe_max_chunk()) tb = 16;
  if (p.mchunk > mt_wgrad_pipe_max_chunk_compact()) tb = 8;
  if (tb == 16)
    MT_CHECK(mt_wgrad_pipe_compact_ok(p), "wgrad_pipe: pixel chunk %d exceeds the offset table", p.mchunk);
  if (tb == 8)
    MT_CHECK(mt_wgrad_pipe_compact8_ok(p) && p.mchunk <= mt_wgrad_pipe_max_chunk_compact8(),
             "wgrad_pipe: pixel chunk %d exceeds the offset table", p.mchunk);
  dim3 grid(p.ntiles * nsplit * ngrp);
  const bool refl = p.pad_mode == MT_PAD_REFLECT;
#define MT_WGP(R, B) hipLaunchKernelGGL((wgrad_pipe_kernel<R, 4, B>), grid, dim3(512), 0, s, p)
  if (tb == 8) { if (refl) MT_WGP(true, 8); else MT_WGP(false, 8); }
  else if (tb == 16) { if (refl) MT_WGP(true, 16); else MT_WGP(false, 16); }
  else { if (refl) MT_WGP(true, 32); else MT_WGP(false, 32); }
#undef MT_WGP
  MT_LAUNCH_CHECK();
  return 0;
}
