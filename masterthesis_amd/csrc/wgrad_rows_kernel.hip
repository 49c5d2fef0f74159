// Accumulator-stationary weight gradient for the 3x3 layers at 64..256 channels (round 4; gfx950 only, bf16).
//
//   dW[a][tap][c] = sum over pixels m of  A[m][a] * B[gather(m, tap)][c]        (A = dY, B = x for Conv2d; swapped for ConvTranspose2d)
//
// The weight gradients of the layers between the stem and the 256-channel bottleneck (content-encoder down-sampling, the decoder's
// transposed convolutions, the style encoder's 3x3 layers: reference networks.py:33,248; blocks.py:73,93-119) have a SMALL output
// (64..256 x 9 x 64..128 values) and a reduction over 65 k .. 1 M pixels.  The 128 x 128 tile kernel (wgrad_kernel) ran them at
// 0.12-0.30 of the matrix peak: every k-step stages a pixel tile of dY and one TAP's gathered tile of x through LDS -- nine copies
// of every input pixel -- and meets at two barriers per 64 pixels.  Here the output is what stays put:
//   * A WORKGROUP OWNS A 64 x 9 x 64 BLOCK OF dW IN REGISTERS: nine waves, one per filter tap, each with the 64 x 64 fp32
//     accumulator of its tap (64 VGPRs).  Channel counts above 64 are further workgroups (blockIdx -> (pixel split, a-block, c-block)).
//   * THE REDUCTION WALKS DOWN A 32-PIXEL-WIDE COLUMN STRIP of one image, one output row (= one 32-deep MFMA k-step) per tick.  A
//     tick brings in ONE new row segment of dY (32 pixels) and `is` new row segments of x (32 is + 2 pixels, halo included) by LDS-DMA;
//     the three input rows a tick needs sit in a ring of row segments, so every input byte reaches LDS once per workgroup instead of
//     nine times, and the tap shift is an LDS address.
//   * LDS LAYOUT FOR THE TRANSPOSING READS: a segment is stored as one 32-byte-per-pixel PLANE per 16-channel MFMA fragment
//     (stride 2: per fragment an even and an odd pixel plane, so that the 32 gathered pixels of a k-step are consecutive plane
//     entries for every tap), entry index XOR-ed with ((e >> 3) & 1) << 2 on the source side of the lane-linear LDS-DMA: the eight
//     rows a 32-lane group of ds_read_b64_tr_b16 touches -- entries e0 .. e0 + 3 and e0 + 8 .. e0 + 11 for ANY e0 -- then cover the 64
//     banks exactly once (no conflicts at any tap shift).
//   * ONE BARRIER PER TICK; the copies of the next WR_D ticks are in flight behind a counted vmcnt.
// Each (split, a-block, c-block) workgroup writes its part of the split's fp32 slab [CaRows][taps][Cb] -- the same slab layout as
// wgrad_kernel / wgrad_pipe_kernel, summed by the same unpack kernels.  Chosen by wgrad_split (conv_api.hip) through
// mt_wgrad_rows_plan; MT_WGRAD_ROWS=0 / mt_kernel_variant_enable(5, 0) switch it off.  Parity: tests/test_wgrad_rows_gpu.py.
#include "conv_device.h"
#include <stdlib.h>

constexpr int WR_D = 4;                     // ticks of copies in flight
constexpr int WR_NA = WR_D + 1;             // ring slots of the dense operand (4 KiB each)
constexpr int WR_APLANE = 1024;             // 32 entries x 32 B
constexpr int WR_ASLOT = 4 * WR_APLANE;
constexpr int WR_BPLANE = 1280;             // 40 entries x 32 B (34 / 33 used)
template <int S> struct WrGeom {
  static constexpr int NB = S * WR_D + 3;                     // ring slots of gathered row segments
  static constexpr int BSLOT = (S == 1 ? 4 : 8) * WR_BPLANE;  // bytes of one row segment: [parity][fragment][entry][32 B]
  static constexpr int CB = BSLOT / 1024;                     // copies per row segment (5 / 10)
  static constexpr int NCOPY = 4 + S * CB;                    // copies per tick (9 / 24)
  static constexpr int CPW = (NCOPY + 8) / 9;                 // ... per wave (1 / 3; the spare ones are dummies)
  static constexpr int WARM = S == 1 ? 2 : 1;                 // load-only ticks at the top of a strip
  static constexpr int A_OFF = 0;
  static constexpr int B_OFF = WR_NA * WR_ASLOT;
  static constexpr int DUMMY_OFF = B_OFF + NB * BSLOT;
  static constexpr int LDS_BYTES = DUMMY_OFF + 1024;
};

template <int OFF>
__device__ __forceinline__ s16x4 wr_tr16(unsigned addr) {     // (inline asm: see wgrad_pipe_kernel.hip)
  s16x4 r;
  asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=v"(r) : "v"(addr), "n"(OFF) : "memory");
  return r;
}
__device__ __forceinline__ int wr_swz(int e) { return e ^ (((e >> 3) & 1) << 2); }

template <int S, bool REFLECT>
__global__ __launch_bounds__(576) void wgrad_rows_kernel(const WgradParams p) {
  using G = WrGeom<S>;
  constexpr unsigned OOB = 0x80000000u;
  static_assert(G::LDS_BYTES <= 160 * 1024, "LDS budget");
  __shared__ u32x4 smem[G::LDS_BYTES / 16];          // ONE shared array (LDS-DMA target: see wgrad_pipe_kernel.hip)
  typedef __attribute__((address_space(3))) void* lds_ptr;
  typedef __attribute__((address_space(3))) char* lds_char_ptr;
  char* const lds0 = reinterpret_cast<char*>(&smem[0]);
  const unsigned lds_base = (unsigned)(size_t)(lds_char_ptr)lds0;

  const int tid = threadIdx.x, lane = tid & 63;
  const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);       // wave = filter tap
  const int dh = wv / 3 - 1, dwo = wv % 3;                         // tap row offset, tap column offset + 1

  // ---- which block of dW, which strip, which rows ----
  const int nca = p.CaRows >> 6, ncb = (p.cpc * 8) >> 6, nblk = nca * ncb;
  const int vid = xcd_remap(blockIdx.x, gridDim.x);
  const int split = vid / nblk, rest = vid - split * nblk;
  const int cab = rest % nca, cbb = rest / nca;
  const int sps = p.rows_sps, wblocks = p.Wo >> 5;
  const int strip = split / sps, part = split - strip * sps;
  const int n = strip / wblocks, w0 = (strip - n * wblocks) << 5;
  const int h0 = part * p.rows_rs;
  const int h1 = min(p.Ho, h0 + p.rows_rs);
  const int T = h1 - h0 + G::WARM;                                 // ticks (host: h0 < Ho)

  // ---- copy duties of this wave: copy q = wv + 9 e of the tick's NCOPY (q < 4: plane q of the dense segment; else row (q-4) / CB,
  // KiB (q-4) % CB of a gathered row segment; q >= NCOPY: dummy) ----
  unsigned loff[G::CPW];          // per-lane source offset inside the row (bytes), or OOB
  int ckind[G::CPW];              // 0 dense, 1 gathered, 2 dummy (wave-uniform)
  int crow[G::CPW];               // gathered: which of the tick's S rows
  unsigned cdst[G::CPW];          // destination offset inside the ring slot
#pragma unroll
  for (int e = 0; e < G::CPW; e++) {
    const int q = wv + 9 * e;
    if (q < 4) {
      const int pos = lane >> 1, half = lane & 1, k = wr_swz(pos);
      ckind[e] = 0; crow[e] = 0; cdst[e] = (unsigned)q * WR_APLANE;
      loff[e] = (unsigned)k * (unsigned)p.Cab + (unsigned)cab * 128u + (unsigned)q * 32u + (unsigned)half * 16u;
    } else if (q < G::NCOPY) {
      const int qb = q - 4, r = qb / G::CB, i = qb - r * G::CB;
      const int o = i * 1024 + lane * 16;
      const int par = S == 2 ? o / (4 * WR_BPLANE) : 0;
      const int f = (o - par * 4 * WR_BPLANE) / WR_BPLANE;
      const int rem = o - par * 4 * WR_BPLANE - f * WR_BPLANE;
      const int hx = wr_swz(rem >> 5), half = (rem >> 4) & 1;
      const int c = S == 2 ? 2 * hx + par : hx;                    // segment-relative input pixel (0 = the left halo pixel)
      bool ok = c < (S == 2 ? 65 : 34);
      int wi = S * w0 - 1 + c;
      if constexpr (REFLECT) {
        wi = wi < 0 ? -wi : wi;
        wi = wi >= p.Wi ? 2 * (p.Wi - 1) - wi : wi;
        ok = ok && wi >= 0 && wi < p.Wi;
      } else {
        ok = ok && (unsigned)wi < (unsigned)p.Wi;
      }
      ckind[e] = 1; crow[e] = r; cdst[e] = (unsigned)i * 1024u;
      loff[e] = ok ? (unsigned)wi * (unsigned)p.Cbb + (unsigned)cbb * 128u + (unsigned)f * 32u + (unsigned)half * 16u : OOB;
    } else {
      ckind[e] = 2; crow[e] = 0; cdst[e] = 0; loff[e] = OOB;
    }
  }
  const __amdgpu_buffer_rsrc_t rsa = __builtin_amdgcn_make_buffer_rsrc((void*)p.a, 0, p.a_bytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t rsb = __builtin_amdgcn_make_buffer_rsrc((void*)p.b, 0, p.b_bytes, 0x00020000);
  const unsigned cab_b = (unsigned)p.Cab, cbb_b = (unsigned)p.Cbb;
  const int Ho = p.Ho, Wo = p.Wo, Hi = p.Hi, Wi = p.Wi;

  int l_tick = 0, l_aslot = 0, l_bslot = 0;          // the next tick to load and the ring slots it fills
  auto issue_tick = [&]() {
    const bool live = l_tick < T;
#pragma unroll
    for (int e = 0; e < G::CPW; e++) {
      if (ckind[e] == 0) {
        const int h = h0 + l_tick - G::WARM;
        const bool ok = live && l_tick >= G::WARM;                 // (h < h1 follows from l_tick < T)
        const unsigned base = (unsigned)((n * Ho + h) * Wo + w0) * cab_b;
        const unsigned vo = ok ? base + loff[e] : OOB;
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rsa, (lds_ptr)(lds0 + G::A_OFF + l_aslot * WR_ASLOT + cdst[e]), 16, vo, 0, 0, 0);
      } else if (ckind[e] == 1) {
        int hi = S == 1 ? h0 - 1 + l_tick : 2 * (h0 + l_tick - 1) + crow[e];
        bool ok = live;
        if constexpr (REFLECT) {
          hi = hi < 0 ? -hi : hi;
          hi = hi >= Hi ? 2 * (Hi - 1) - hi : hi;
          ok = ok && hi >= 0 && hi < Hi;
        } else {
          ok = ok && (unsigned)hi < (unsigned)Hi;
        }
        const unsigned base = (unsigned)((n * Hi + hi) * Wi) * cbb_b;
        const unsigned vo = (ok && loff[e] != OOB) ? base + loff[e] : OOB;
        int bs = l_bslot + crow[e];
        bs = bs >= G::NB ? bs - G::NB : bs;
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rsb, (lds_ptr)(lds0 + G::B_OFF + bs * G::BSLOT + cdst[e]), 16, vo, 0, 0, 0);
      } else {
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rsa, (lds_ptr)(lds0 + G::DUMMY_OFF), 16, OOB, 0, 0, 0);
      }
    }
    l_tick++;
    l_aslot = l_aslot + 1 == WR_NA ? 0 : l_aslot + 1;
    l_bslot += S;
    l_bslot = l_bslot >= G::NB ? l_bslot - G::NB : l_bslot;
  };

  f32x4 acc[4][4];
#pragma unroll
  for (int a = 0; a < 4; a++)
#pragma unroll
    for (int b = 0; b < 4; b++) acc[a][b] = f32x4{0.f, 0.f, 0.f, 0.f};

#pragma unroll
  for (int s = 0; s < WR_D; s++) issue_tick();

  // ---- fragment addresses: lane (g, qq, pp) supplies [entry 8 g + qq (+ 4)][channels 4 pp .. 4 pp + 3] of a fragment plane ----
  const int g = lane >> 4, qq = (lane & 15) >> 2, pp = lane & 3;
  const int kk = 8 * g + qq;
  const unsigned aoff0 = (unsigned)(wr_swz(kk) * 32 + pp * 8), aoff1 = (unsigned)(wr_swz(kk + 4) * 32 + pp * 8);
  unsigned boff0, boff1;
  if constexpr (S == 1) {
    boff0 = (unsigned)(wr_swz(kk + dwo) * 32 + pp * 8);
    boff1 = (unsigned)(wr_swz(kk + dwo + 4) * 32 + pp * 8);
  } else {
    const int par = dwo & 1, hx = kk + (dwo >> 1);
    boff0 = (unsigned)(par * 4 * WR_BPLANE + wr_swz(hx) * 32 + pp * 8);
    boff1 = (unsigned)(par * 4 * WR_BPLANE + wr_swz(hx + 4) * 32 + pp * 8);
  }
  // ring slot of this tap's input row at tick 0: row sequence index  S j + (S == 1 ? dh - 1 : dh)
  int c_aslot = 0;
  int c_bslot = (S == 1 ? dh - 1 : dh) + G::NB;
  c_bslot = c_bslot >= G::NB ? c_bslot - G::NB : c_bslot;

  for (int j = 0; j < T; j++) {
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"((WR_D - 1) * G::CPW) : "memory");
    __builtin_amdgcn_sched_barrier(0);
    __builtin_amdgcn_s_barrier();                  // every wave's copies of tick j have landed; tick j - 1 has been read
    __builtin_amdgcn_sched_barrier(0);
    issue_tick();
    __builtin_amdgcn_sched_barrier(0);
    if (j >= G::WARM) {
      const unsigned sa = lds_base + G::A_OFF + (unsigned)c_aslot * WR_ASLOT;
      const unsigned sb = lds_base + G::B_OFF + (unsigned)c_bslot * G::BSLOT;
      bf16x8 af[4], bf[4];
#define WR_FRAG(dst, base, o0, o1, OFF)                                                                     \
      {                                                                                                     \
        const s16x4 lo = wr_tr16<OFF>(base + o0), hi = wr_tr16<OFF>(base + o1);                             \
        dst = __builtin_bit_cast(bf16x8, __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7));          \
      }
      WR_FRAG(af[0], sa, aoff0, aoff1, 0 * WR_APLANE);
      WR_FRAG(af[1], sa, aoff0, aoff1, 1 * WR_APLANE);
      WR_FRAG(af[2], sa, aoff0, aoff1, 2 * WR_APLANE);
      WR_FRAG(af[3], sa, aoff0, aoff1, 3 * WR_APLANE);
      WR_FRAG(bf[0], sb, boff0, boff1, 0 * WR_BPLANE);
      WR_FRAG(bf[1], sb, boff0, boff1, 1 * WR_BPLANE);
      WR_FRAG(bf[2], sb, boff0, boff1, 2 * WR_BPLANE);
      WR_FRAG(bf[3], sb, boff0, boff1, 3 * WR_BPLANE);
#undef WR_FRAG
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      __builtin_amdgcn_sched_barrier(0);
      // gathered operand as MFMA A: a lane ends with 4 consecutive columns of one a-channel (16-byte slab stores)
#pragma unroll
      for (int a = 0; a < 4; a++)
#pragma unroll
        for (int b = 0; b < 4; b++)
          acc[a][b] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bf[b], af[a], acc[a][b], 0, 0, 0);
      __builtin_amdgcn_sched_barrier(0);
    }
    c_aslot = c_aslot + 1 == WR_NA ? 0 : c_aslot + 1;
    c_bslot += S;
    c_bslot = c_bslot >= G::NB ? c_bslot - G::NB : c_bslot;
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // the trailing (all-zero) copies must have landed before the wave exits

  // ---- epilogue: this workgroup's 64 x 9 x 64 block of the split's fp32 slab [CaRows][taps][Cb] ----
  const int fr = lane & 15, fg = lane >> 4;
  const int cbt = p.cpc * 8, ncols = p.nchunks * 8;
  float* const slab = p.out + (size_t)split * p.CaRows * ncols;
#pragma unroll
  for (int a = 0; a < 4; a++) {
    const int ca = cab * 64 + a * 16 + fr;
#pragma unroll
    for (int b = 0; b < 4; b++) {
      const int col = wv * cbt + cbb * 64 + b * 16 + fg * 4;
      *reinterpret_cast<f32x4*>(slab + (size_t)ca * ncols + col) = acc[a][b];
    }
  }
}

static long g_wr_launches = 0;
static int g_wr_on = -1;
static int wr_enabled() {
  if (g_wr_on < 0) g_wr_on = getenv("MT_WGRAD_ROWS") ? (atoi(getenv("MT_WGRAD_ROWS")) != 0) : 1;
  return g_wr_on;
}
long mt_wgrad_rows_launches() { return g_wr_launches; }
int mt_wgrad_rows_enable(int on) {
  const int prev = wr_enabled();
  g_wr_on = on != 0;
  return prev;
}

// Does the accumulator-stationary kernel take this problem, and with which pixel split?  (geometry only: p.a / p.b / p.out unused)
//   *nsplit = slabs; *rs = output rows per split; the kernel derives the splits per strip as ceil(Ho / rs).
bool mt_wgrad_rows_plan(int dtype, const WgradParams& p, int* nsplit, int* rs) {
  if (!wr_enabled() || dtype != MT_BF16) return false;
  if (p.ntaps != 9 || (p.is != 1 && p.is != 2)) return false;
  for (int t = 0; t < 9; t++)
    if (p.dh[t] != t / 3 - 1 || p.dw[t] != t % 3 - 1) return false;
  const int cb = p.cpc * 8;
  if (p.CaRows % 64 || cb % 64 || p.Cab != p.CaRows * 2 || p.Cbb != cb * 2) return false;
  if (p.Wo % 32 || p.Hi != p.is * p.Ho || p.Wi != p.is * p.Wo || p.Hi < 2 || p.Wi < 2) return false;
  if ((double)p.M * p.Cab >= 2147483000.0 || (double)p.N * p.Hi * p.Wi * p.Cbb >= 2147483000.0) return false;
  const int blocks = (p.CaRows / 64) * (cb / 64);
  if (blocks > 64) return false;
  const long strips = (long)p.N * (p.Wo / 32);
  const long ticks = strips * p.Ho;
  if (ticks * blocks < 256L * 16) return false;       // less than 16 ticks per compute unit: the tile kernels' splits serve it
  const int target = 256 / blocks;                     // workgroups of one round, per channel block
  long sps = target / strips;
  if (sps < 1) sps = 1;
  if (sps > p.Ho / 8) sps = p.Ho / 8 > 0 ? p.Ho / 8 : 1;
  const int rows = (int)((p.Ho + sps - 1) / sps);
  sps = (p.Ho + rows - 1) / rows;
  if (strips * sps > 4096) return false;
  *nsplit = (int)(strips * sps);
  *rs = rows;
  return true;
}

int mt_launch_wgrad_rows(const WgradParams& pin, int nsplit, hipStream_t s) {
  WgradParams p = pin;
  MT_CHECK(p.rows_rs > 0, "wgrad_rows: no row split");
  p.rows_sps = (p.Ho + p.rows_rs - 1) / p.rows_rs;
  const long strips = (long)p.N * (p.Wo / 32);
  MT_CHECK((long)nsplit == strips * p.rows_sps, "wgrad_rows: %d splits for %ld strips x %d", nsplit, strips, p.rows_sps);
  const int blocks = (p.CaRows / 64) * (p.cpc * 8 / 64);
  dim3 grid((unsigned)(nsplit * blocks));
  const bool refl = p.pad_mode == MT_PAD_REFLECT;
#define MT_WR(S, R) hipLaunchKernelGGL((wgrad_rows_kernel<S, R>), grid, dim3(576), 0, s, p)
  if (p.is == 1) { if (refl) MT_WR(1, true); else MT_WR(1, false); }
  else { if (refl) MT_WR(2, true); else MT_WR(2, false); }
#undef MT_WR
  MT_LAUNCH_CHECK();
  g_wr_launches++;
  return 0;
}
