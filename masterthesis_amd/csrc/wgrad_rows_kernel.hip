// Accumulator-stationary weight gradient for the 3x3 layers at 64..256 channels (round 4; gfx950 only, bf16).
//
//   dW[a][tap][c] = sum over pixels m of  A[m][a] * B[gather(m, tap)][c]        (A = dY, B = x for Conv2d; swapped for ConvTranspose2d)
//
// The weight gradients of the layers between the stem and the 256-channel bottleneck (content-encoder down-sampling, the decoder's
// transposed convolutions, the style encoder's 3x3 layers: reference networks.py:33,248; blocks.py:73,93-119) have a SMALL output
// (64..256 x 9 x 64..128 values) and a reduction over 65 k .. 1 M pixels.  The 128 x 128 tile kernel (wgrad_kernel) ran them at
// 0.12-0.30 of the matrix peak: every k-step stages a pixel tile of dY and one TAP's gathered tile of x through LDS -- nine copies
// of every input pixel -- and meets at two barriers per 64 pixels.  Here the output is what stays put:
//   * A WORKGROUP OWNS A 64 x 9 x 64 BLOCK OF dW IN REGISTERS: eight waves, wave t with the 64 x 64 fp32 accumulator of filter tap
//     t (64 VGPRs) plus two of the sixteen 16 x 16 tiles of tap 8 (18 MFMAs per wave and k-step: two waves per SIMD, all four SIMDs
//     level; a first form with nine waves had three on one SIMD).  Channel counts above 64 are further workgroups
//     (blockIdx -> (problem, pixel split, a-block, c-block)).
//   * THE REDUCTION WALKS DOWN A 32-PIXEL-WIDE COLUMN STRIP of one image, one output row (= one 32-deep MFMA k-step) per tick.  A
//     tick brings in ONE new row segment of dY (32 pixels) and `is` new row segments of x (32 is + 2 pixels, halo included) by LDS-DMA;
//     the three input rows a tick needs sit in a ring of row segments, so every input byte reaches LDS once per workgroup instead of
//     nine times, and the tap shift is an LDS address.
//   * LDS LAYOUT FOR THE TRANSPOSING READS: a segment is stored pixel-major, 128 bytes (the block's 64 channels) per entry (stride
//     2: an even and an odd pixel plane, so that the 32 gathered pixels of a k-step are consecutive entries for every tap), the
//     32-byte slot of a 16-channel MFMA fragment XOR-ed with key(e) = bit 1 | bit 3 << 1 of the entry index on the source side of
//     the lane-linear LDS-DMA: the eight rows a 32-lane group of ds_read_b64_tr_b16 touches -- entries e0 .. e0 + 3 and e0 + 8 ..
//     e0 + 11 for ANY e0 -- then cover the 64 banks exactly once (no conflicts at any tap shift), and a copy instruction fetches
//     eight WHOLE 128-byte pixel rows (a first layout with one 32-byte-per-pixel plane per fragment read 32 quarter rows per
//     instruction and ran at the texture addresser's pace).
//   * ONE BARRIER PER TICK; the copies of the next D ticks (8 at stride 1, 5 at stride 2: what LDS holds) are in flight behind a
//     counted vmcnt; tick j's fragments are read (two register sets) while the MFMAs of tick j - 1 run; the copy issue is
//     branch-free.  Nothing in a tick is conditional: a tick without an output row finds zeros in the dense operand's ring slot.
//   * SEVERAL LAYERS PER LAUNCH: the kernel argument is a table of problems (WrMulti); a workgroup finds its problem from the block
//     index and walks a run of that problem's k-steps (k-step = (strip, output row), strip-major, runs may cross strips).  A layer
//     alone fills 256 compute units only by cutting its reduction ~256 ways (38 MB of slabs per layer); the eligible layers of a
//     whole backward pass share the chip instead (mt_conv_bwd_weight_rows_multi: ~25 slabs per layer for ten layers).
// Each (split, a-block, c-block) workgroup writes its part of the split's fp32 slab [CaRows][taps][Cb] -- the same slab layout as
// wgrad_kernel / wgrad_pipe_kernel, summed by the same unpack kernels.  Single problems come through wgrad_split (conv_api.hip);
// MT_WGRAD_ROWS=0 / mt_kernel_variant_enable(5, 0) switch the kernel off.  Parity: tests/test_wgrad_rows_gpu.py.
// Measured (profiles/round4_wgrad_rows_*): 1.2-1.4x the tile kernel per launch, -0.75 ms of kernel time per --ms_dis step with the
// shared launches.  Knock-out builds (-DMT_WR_EXP_NOMFMA / NOREAD / NOCOPY / NOSLAB, tools/wgrad_rows_knockouts.sh): a tick costs
// 0.57 us at stride 1 with everything, 0.39 without the MFMAs, 0.40 without the LDS reads, 0.56 without any memory traffic; LDS is
// conflict-free (SQ_LDS_BANK_CONFLICT 0) and 38 % busy, the matrix pipes hold 576 of the tick's ~850 cycles: what is left is
// instruction ISSUE: 3.5 scalar + 3.1 vector + 1.2 LDS instructions per MFMA (SQ_INSTS_*; 144 instructions per wave and tick for 18
// MFMAs: slot counters, the copies' row arithmetic, one address add per read) -- a wave issues one instruction per four cycles, so its
// stream needs ~36 cycles per MFMA against 32 of the shared matrix pipe.  Next: two output rows per tick.  The ping-pong arrangement of wgrad_pipe_kernel (wave groups half a
// tick apart, -DMT_WR_PINGPONG=1) was built and measured SLOWER (0.78 us per tick): a tick's memory phase is longer than its MFMAs.
#include "conv_device.h"
#include <stdlib.h>
#include <string.h>
#include <type_traits>

constexpr int WR_ASLOT = 4096;              // 32 entries x 128 B
constexpr int WR_BPLANE = 5120;             // 40 entries x 128 B (34 / 33 used)
template <int S> struct WrGeom {
// (1: the ping-pong form of the tick loop -- built, parity-green, and SLOWER: 0.78 us per stride-1 tick against 0.58; its memory phase
// -- 22 reads, the copies, the waits -- is ~1.7x its 18 MFMAs, so the groups wait for each other's memory phases)
#ifndef MT_WR_PINGPONG
#define MT_WR_PINGPONG 0
#endif
#ifndef MT_WR_D1
#define MT_WR_D1 8
#endif
#ifndef MT_WR_D2
#define MT_WR_D2 5
#endif
  static constexpr int D = S == 1 ? MT_WR_D1 : MT_WR_D2;      // ticks of copies in flight (what the latency of the copies needs, what LDS holds)
  static constexpr int NA = D + 1;                            // ring slots of the dense operand (4 KiB each)
  static constexpr int NB = S * D + 3;                     // ring slots of gathered row segments
  static constexpr int BSLOT = S * WR_BPLANE;                 // bytes of one row segment: [parity][entry][128 B]
  static constexpr int CB = BSLOT / 1024;                     // copies per row segment (5 / 10)
  static constexpr int NCOPY = 4 + S * CB;                    // copies per tick (9 / 24)
  static constexpr int CPW = (NCOPY + 7) / 8;                 // ... per wave at most (stride 1: wave 0 has two, the others one; stride 2: three)
  static constexpr int WARM = S == 1 ? 2 : 1;                 // load-only ticks at the top of a strip
  static constexpr int A_OFF = 0;
  static constexpr int B_OFF = NA * WR_ASLOT;
  static constexpr int LDS_BYTES = B_OFF + NB * BSLOT;
};

// one weight-gradient problem of a launch (several layers' problems share a launch: mt_launch_wgrad_rows_multi)
struct WrProb {
  const char* a;            // dense operand, pixel-major [N Ho Wo][Cab bytes]
  const char* b;            // gathered operand, NHWC [N][Hi][Wi][Cbb bytes], Hi = S Ho, Wi = S Wo
  float* out;               // fp32 slabs [nsplit][64 nca][9][64 ncb]
  unsigned a_bytes, b_bytes;
  int Ho, Wo, Hi, Wi;
  int Cab, Cbb;
  int nca, ncb;             // 64-channel blocks of the two operands
  int rows;                 // k-steps = strips x Ho (strip-major)
  int rps;                  // k-steps per split
  int blk0, nblk;           // first block of the launch (a multiple of 8), blocks = nsplit x nca x ncb
  int reflect;
};
struct WrMulti {
  int n;
  WrProb p[MT_WR_MAXP];
};

template <int OFF>
__device__ __forceinline__ s16x4 wr_tr16(unsigned addr) {     // (inline asm: see wgrad_pipe_kernel.hip)
  s16x4 r;
  asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=v"(r) : "v"(addr), "n"(OFF) : "memory");
  return r;
}
__device__ __forceinline__ int wr_key(int e) { return ((e >> 1) & 1) | (((e >> 3) & 1) << 1); }
// byte offset of fragment f's 32-byte slot of entry e
__device__ __forceinline__ unsigned wr_slot(int e, int f) { return (unsigned)(e * 128 + ((f ^ wr_key(e)) << 5)); }

template <int S>
__global__ __launch_bounds__(512) void wgrad_rows_kernel(const WrMulti m) {
  using G = WrGeom<S>;
  constexpr unsigned OOB = 0x80000000u;
  static_assert(G::LDS_BYTES <= 160 * 1024, "LDS budget");
  __shared__ u32x4 smem[G::LDS_BYTES / 16];          // ONE shared array (LDS-DMA target: see wgrad_pipe_kernel.hip)
  typedef __attribute__((address_space(3))) void* lds_ptr;
  typedef __attribute__((address_space(3))) char* lds_char_ptr;
  char* const lds0 = reinterpret_cast<char*>(&smem[0]);
  const unsigned lds_base = (unsigned)(size_t)(lds_char_ptr)lds0;

  const int tid = threadIdx.x, lane = tid & 63;
  const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);       // wave = filter tap 0 .. 7 (+ two of the sixteen tiles of tap 8)
  const int dh = wv / 3 - 1, dwo = wv % 3;                         // tap row offset, tap column offset + 1
  [[maybe_unused]] const int grp = wv >> 2;                         // ping-pong group (waves w and w + 4 share a SIMD)
  const int a9 = wv >> 1, fb9 = 2 * (wv & 1);                      // tap 8 (dh = 1, dwo = 2): a-fragment a9 x c-fragments fb9, fb9 + 1

  // ---- which problem of the launch (blocks of a problem are consecutive, blk0 a multiple of 8), which block of its dW, which
  // range of its k-steps (k-step = (strip, output row), strip-major) ----
  int pi = 0;
  for (int i = 1; i < m.n; i++) pi = (int)blockIdx.x >= m.p[i].blk0 ? i : pi;
  const char* const pa = m.p[pi].a;
  const char* const pb = m.p[pi].b;
  float* const pout = m.p[pi].out;
  const unsigned a_bytes = m.p[pi].a_bytes, b_bytes = m.p[pi].b_bytes;
  const int Ho = m.p[pi].Ho, Wo = m.p[pi].Wo, Hi = m.p[pi].Hi, Wi = m.p[pi].Wi;
  const unsigned cab_b = (unsigned)m.p[pi].Cab, cbb_b = (unsigned)m.p[pi].Cbb;
  const int nca = m.p[pi].nca, ncb = m.p[pi].ncb, rows = m.p[pi].rows, rps = m.p[pi].rps;
  const int nblk = m.p[pi].nblk;
  const bool refl = m.p[pi].reflect != 0;
  const int vid = xcd_remap((int)blockIdx.x - m.p[pi].blk0, (nblk + 7) & ~7);
  if (vid >= nblk) return;
  const int nb = nca * ncb;
  const int split = vid / nb, rest = vid - split * nb;
  const int cab = rest % nca, cbb = rest / nca;
  const int wblocks = Wo >> 5;
  const int gbeg = split * rps;
  const int gend = min(rows, gbeg + rps);                          // (host: gbeg < rows)


  // ---- fragment addresses: lane (g, qq, pp) supplies [entry 8 g + qq (+ 4)][channels 4 pp .. 4 pp + 3] of a fragment plane ----
  const int fgq = lane >> 4, qq = (lane & 15) >> 2, pp = lane & 3;
  const int kk = 8 * fgq + qq;
  unsigned aoff[2][4], boff[2][4], aoff9[2], boff9[2][2];
  {
    const int par = S == 2 ? (dwo & 1) : 0;
    const int eb = S == 2 ? kk + (dwo >> 1) : kk + dwo;
    const int eb9 = S == 2 ? kk + 1 : kk + 2;                      // (tap 8: column offset + 1 = 2 -> even plane, entry + 1)
#pragma unroll
    for (int h = 0; h < 2; h++) {
#pragma unroll
      for (int f = 0; f < 4; f++) {
        aoff[h][f] = wr_slot(kk + 4 * h, f) + (unsigned)(pp * 8);
        boff[h][f] = (unsigned)(par * WR_BPLANE) + wr_slot(eb + 4 * h, f) + (unsigned)(pp * 8);
      }
      aoff9[h] = wr_slot(kk + 4 * h, a9) + (unsigned)(pp * 8);
      boff9[h][0] = wr_slot(eb9 + 4 * h, fb9) + (unsigned)(pp * 8);
      boff9[h][1] = wr_slot(eb9 + 4 * h, fb9 + 1) + (unsigned)(pp * 8);
    }
  }

  f32x4 acc[4][4];
#pragma unroll
  for (int a = 0; a < 4; a++)
#pragma unroll
    for (int b = 0; b < 4; b++) acc[a][b] = f32x4{0.f, 0.f, 0.f, 0.f};
  f32x4 acc9[2] = {f32x4{0.f, 0.f, 0.f, 0.f}, f32x4{0.f, 0.f, 0.f, 0.f}};

  // copy duties of this wave: copy q = wv + 8 e of a tick's NCOPY (q < 4: KiB q of the dense segment; else row (q-4) / CB,
  // KiB (q-4) % CB of a gathered row segment; q >= NCOPY: none): kind / row / destination are wave-uniform
  int ckind[G::CPW], crow[G::CPW];
  unsigned cdst[G::CPW];
#pragma unroll
  for (int e = 0; e < G::CPW; e++) {
    const int q = wv + 8 * e;
    if (q < 4) { ckind[e] = 0; crow[e] = 0; cdst[e] = (unsigned)q * 1024u; }
    else if (q < G::NCOPY) { const int qb = q - 4; ckind[e] = 1; crow[e] = qb / G::CB; cdst[e] = (unsigned)(qb - crow[e] * G::CB) * 1024u; }
    else { ckind[e] = 2; crow[e] = 0; cdst[e] = 0; }
  }
  const bool two_copies = G::NCOPY - 8 * (G::CPW - 1) > wv;        // this wave issues CPW copies per tick (else CPW - 1)
  // (three named resources, not an array: an array of the opaque resource type silently drops the kernel's host-side stub)
  const __amdgpu_buffer_rsrc_t rsc0 = __builtin_amdgcn_make_buffer_rsrc((void*)(ckind[0] == 0 ? pa : pb), 0, ckind[0] == 0 ? a_bytes : b_bytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t rsc1 = __builtin_amdgcn_make_buffer_rsrc((void*)(ckind[1] == 0 ? pa : pb), 0, ckind[1] == 0 ? a_bytes : b_bytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t rsc2 = __builtin_amdgcn_make_buffer_rsrc((void*)(ckind[G::CPW - 1] == 0 ? pa : pb), 0, ckind[G::CPW - 1] == 0 ? a_bytes : b_bytes, 0x00020000);
  // LDS starts as zeros (a ring slot that no copy has reached yet is read by the unconditional ticks), both fragment sets too
  for (int i = tid; i < G::LDS_BYTES / 16; i += 512) smem[i] = u32x4{0u, 0u, 0u, 0u};
#if !MT_WR_PINGPONG
  bf16x8 fP[11], fQ[11];           // 0..3 a-fragments, 4..7 c-fragments of the wave's tap, 8 a-fragment + 9, 10 c-fragments of tap 8
#pragma unroll
  for (int i = 0; i < 11; i++) { fP[i] = __builtin_bit_cast(bf16x8, u32x4{0u, 0u, 0u, 0u}); fQ[i] = fP[i]; }
#endif
  __syncthreads();

  for (int gpos = gbeg; gpos < gend;) {
    // ---- one run of consecutive output rows h0 .. h1 - 1 of one 32-pixel column strip ----
    const int strip = gpos / Ho, h0 = gpos - strip * Ho;
    const int h1 = min(Ho, h0 + (gend - gpos));
    const int n = strip / wblocks, w0 = (strip - n * wblocks) << 5;
    const int T = h1 - h0 + G::WARM;                               // ticks
    gpos += h1 - h0;

    unsigned loff[G::CPW];          // per-lane source offset inside the row (bytes), or OOB
#pragma unroll
    for (int e = 0; e < G::CPW; e++) {
      const int q = wv + 8 * e;
      const int sub = lane >> 3, slot = (lane & 7) >> 1, half = lane & 1;     // a copy = 8 entries x 128 B
      if (q < 4) {
        const int k = 8 * q + sub, f = slot ^ wr_key(k);
        loff[e] = (unsigned)k * cab_b + (unsigned)cab * 128u + (unsigned)f * 32u + (unsigned)half * 16u;
      } else if (q < G::NCOPY) {
        const int qb = q - 4, i = qb - (qb / G::CB) * G::CB;
        const int par = S == 2 ? i / 5 : 0;
        const int hx = 8 * (i - par * 5) + sub, f = slot ^ wr_key(hx);
        const int c = S == 2 ? 2 * hx + par : hx;                  // segment-relative input pixel (0 = the left halo pixel)
        bool ok = c < (S == 2 ? 65 : 34);
        int wi = S * w0 - 1 + c;
        if (refl) {
          wi = wi < 0 ? -wi : wi;
          wi = wi >= Wi ? 2 * (Wi - 1) - wi : wi;
        }
        ok = ok && (unsigned)wi < (unsigned)Wi;
        loff[e] = ok ? (unsigned)wi * cbb_b + (unsigned)cbb * 128u + (unsigned)f * 32u + (unsigned)half * 16u : OOB;
      } else {
        loff[e] = OOB;
      }
    }

    // loader state: the next tick to load, the ring slots it fills (BYTE offsets into LDS), the dense operand's row base and the
    // gathered operand's first row of that tick -- advanced by additions (the tick is bound by instruction issue: every scalar counts)
    int l_tick = 0;
    unsigned l_a = G::A_OFF, l_b = G::B_OFF;
    const unsigned rowa_step = (unsigned)Wo * cab_b, rowb_step = (unsigned)Wi * cbb_b;
    unsigned rowa = (unsigned)((n * Ho + h0 - G::WARM) * Wo + w0) * cab_b;     // (meaningless while l_tick < WARM: masked)
    const unsigned nb_base = (unsigned)(n * Hi) * rowb_step;
    int hib = S == 1 ? h0 - 1 : 2 * (h0 - 1);
    const unsigned t_live = (unsigned)(T - G::WARM);
    auto issue_copy = [&](auto ec) {
      constexpr int e = decltype(ec)::value;
      unsigned base, okm, dst;
      if (ckind[e] == 0) {
        okm = (unsigned)(l_tick - G::WARM) < t_live ? 0u : OOB;      // WARM <= l_tick < T  (h < h1 follows)
        base = rowa;
        dst = l_a + cdst[e];
      } else {
        const int hi0 = hib + crow[e];
        int hr = hi0 < 0 ? -hi0 : hi0;
        hr = hr >= Hi ? 2 * (Hi - 1) - hr : hr;
        const int hi = refl ? hr : hi0;
        okm = (l_tick < T && (unsigned)hi < (unsigned)Hi) ? 0u : OOB;
        base = nb_base + (unsigned)hi * rowb_step;
        unsigned bs = l_b + (unsigned)crow[e] * G::BSLOT;
        bs = bs >= (unsigned)(G::B_OFF + G::NB * G::BSLOT) ? bs - (unsigned)(G::NB * G::BSLOT) : bs;
        dst = bs + cdst[e];
      }
      // out of range if the row is (scalar mask) or the lane's pixel is (loff = OOB: the sum keeps the top bit, base < 2 GiB)
#ifdef MT_WR_EXP_NOCOPY
      const unsigned vo = OOB | (base & 1u) | (okm >> 30);
#else
      const unsigned vo = (base + loff[e]) | okm;
#endif
      if constexpr (e == 0) __builtin_amdgcn_raw_ptr_buffer_load_lds(rsc0, (lds_ptr)(lds0 + dst), 16, vo, 0, 0, 0);
      else if constexpr (e == 1 && G::CPW > 2) __builtin_amdgcn_raw_ptr_buffer_load_lds(rsc1, (lds_ptr)(lds0 + dst), 16, vo, 0, 0, 0);
      else __builtin_amdgcn_raw_ptr_buffer_load_lds(rsc2, (lds_ptr)(lds0 + dst), 16, vo, 0, 0, 0);
    };
    auto advance_load = [&]() {
      l_tick++;
      l_a = l_a + WR_ASLOT == (unsigned)(G::A_OFF + G::NA * WR_ASLOT) ? (unsigned)G::A_OFF : l_a + WR_ASLOT;
      l_b += S * G::BSLOT;
      l_b = l_b >= (unsigned)(G::B_OFF + G::NB * G::BSLOT) ? l_b - (unsigned)(G::NB * G::BSLOT) : l_b;
      rowa += rowa_step;
      hib += S;
    };
#pragma unroll
    for (int s = 0; s < G::D; s++) {
      issue_copy(std::integral_constant<int, 0>{});
      if constexpr (G::CPW > 2) issue_copy(std::integral_constant<int, 1>{});
      if (two_copies) issue_copy(std::integral_constant<int, G::CPW - 1>{});
      advance_load();
    }

    // ring slots of this tap's input row (and of tap 8's: dh = 1) and of the dense segment at tick 0, as LDS byte addresses: row
    // sequence index  S j + (S == 1 ? dh - 1 : dh)
    const unsigned a_end = lds_base + G::A_OFF + G::NA * WR_ASLOT, b_end = lds_base + G::B_OFF + G::NB * G::BSLOT;
    unsigned c_a = lds_base + G::A_OFF;
    int cb0 = (S == 1 ? dh - 1 : dh) + G::NB;
    cb0 = cb0 >= G::NB ? cb0 - G::NB : cb0;
    unsigned c_b = lds_base + G::B_OFF + (unsigned)cb0 * G::BSLOT;
    unsigned c_b9 = lds_base + G::B_OFF + (S == 1 ? 0u : (unsigned)G::BSLOT);
    auto advance_compute = [&]() {
      c_a = c_a + WR_ASLOT == a_end ? lds_base + G::A_OFF : c_a + WR_ASLOT;
      c_b += S * G::BSLOT;
      c_b = c_b >= b_end ? c_b - (unsigned)(G::NB * G::BSLOT) : c_b;
      c_b9 += S * G::BSLOT;
      c_b9 = c_b9 >= b_end ? c_b9 - (unsigned)(G::NB * G::BSLOT) : c_b9;
    };

#if MT_WR_PINGPONG
    // PING-PONG over the two waves of a SIMD (waves w and w + 4: group = wv >> 2; the arrangement of wgrad_pipe_kernel): a tick is a
    // memory phase (22 transposing reads of tick j, the copies of tick j + D, the waits) and a compute phase (18 MFMAs), two barriers
    // per tick, and group 1 runs ONE BARRIER behind group 0 -- while one wave of a SIMD waits, copies and reads, the other multiplies.
    // With everything in one phase the two waves of a SIMD read together and multiplied together: knock-outs put the matrix pipes at
    // 0.69 busy, the rest being the barrier, the copy issue and the read latency with nothing to multiply.
    // Ordering: a wave ends its memory phase with vmcnt((D-1) c) behind the copies of tick j + D, i.e. its copies up to tick j + 1
    // have landed when it passes the next barrier -- group 0 reads tick j behind barrier 2 j (group 1 passed it at the end of its
    // memory phase j - 1: tick j landed), group 1 behind barrier 2 j + 1 (group 0 passed it at the end of its memory phase j).  The
    // dense ring slot that the copies of tick j + D refill was read for tick j - 1: retired by both groups before barrier 2 j.
    bf16x8 fr_[11];
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(0) : "memory");       // (run start: the first D ticks; short runs are rare)
    __builtin_amdgcn_sched_barrier(0);
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_sched_barrier(0);
    if (grp) {
      __builtin_amdgcn_sched_barrier(0);
      __builtin_amdgcn_s_barrier();
      __builtin_amdgcn_sched_barrier(0);
    }
    for (int j = 0; j < T; j++) {
      const unsigned sa = c_a, sb = c_b, sb9 = c_b9;
#define WR_FRAG(dst, base, o0, o1)                                                                          \
      {                                                                                                     \
        const s16x4 lo = wr_tr16<0>(base + o0), hi = wr_tr16<0>(base + o1);                                 \
        dst = __builtin_bit_cast(bf16x8, __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7));          \
      }
#pragma unroll
      for (int f = 0; f < 4; f++) WR_FRAG(fr_[f], sa, aoff[0][f], aoff[1][f]);
      WR_FRAG(fr_[4], sb, boff[0][0], boff[1][0]);
      WR_FRAG(fr_[5], sb, boff[0][1], boff[1][1]);
      WR_FRAG(fr_[6], sb, boff[0][2], boff[1][2]);
      __builtin_amdgcn_sched_barrier(0);
      issue_copy(std::integral_constant<int, 0>{});                  // (15 reads may be outstanding: the copies' arithmetic fills the gap)
      __builtin_amdgcn_sched_barrier(0);
      WR_FRAG(fr_[7], sb, boff[0][3], boff[1][3]);
      WR_FRAG(fr_[8], sa, aoff9[0], aoff9[1]);
      __builtin_amdgcn_sched_barrier(0);
      if constexpr (G::CPW > 2) issue_copy(std::integral_constant<int, 1>{});
      __builtin_amdgcn_sched_barrier(0);
      WR_FRAG(fr_[9], sb9, boff9[0][0], boff9[1][0]);
      WR_FRAG(fr_[10], sb9, boff9[0][1], boff9[1][1]);
#undef WR_FRAG
      __builtin_amdgcn_sched_barrier(0);
      if (two_copies) issue_copy(std::integral_constant<int, G::CPW - 1>{});
      advance_load();
      advance_compute();
      __builtin_amdgcn_sched_barrier(0);
      if (two_copies) asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)" ::"n"((G::D - 1) * G::CPW) : "memory");
      else asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)" ::"n"((G::D - 1) * (G::CPW - 1)) : "memory");
      __builtin_amdgcn_sched_barrier(0);
      __builtin_amdgcn_s_barrier();
      __builtin_amdgcn_sched_barrier(0);
      __builtin_amdgcn_s_setprio(1);
      // gathered operand as MFMA A: a lane ends with 4 consecutive columns of one a-channel (16-byte slab stores)
#pragma unroll
      for (int a = 0; a < 4; a++)
#pragma unroll
        for (int b = 0; b < 4; b++)
          acc[a][b] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fr_[4 + b], fr_[a], acc[a][b], 0, 0, 0);
      acc9[0] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fr_[9], fr_[8], acc9[0], 0, 0, 0);
      acc9[1] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fr_[10], fr_[8], acc9[1], 0, 0, 0);
      __builtin_amdgcn_s_setprio(0);
      __builtin_amdgcn_sched_barrier(0);
      __builtin_amdgcn_s_barrier();
      __builtin_amdgcn_sched_barrier(0);
    }
    if (!grp) {
      __builtin_amdgcn_sched_barrier(0);
      __builtin_amdgcn_s_barrier();
      __builtin_amdgcn_sched_barrier(0);
    }
#else
    // One tick.  Knock-outs (round 4) put the first form of this loop -- barrier, copies, 22 transposing reads, wait, 18 MFMAs, one
    // after the other -- at 0.21 us of barrier + copy issue, 0.19 us of LDS reads and 0.19 us of MFMAs per tick with NO overlap
    // between them (a wave can have 15 LDS reads outstanding, so the 16th blocks the issue of everything behind it, and the copy
    // issue was a chain of wave-uniform branches).  Here tick j's fragments (set rd) are read WHILE the MFMAs of tick j - 1 (set mm)
    // run, eight reads ahead and then four reads per three MFMAs, and the branch-free copy issue of tick j + D follows in the shadow of
    // the last MFMAs.  Nothing is conditional: the dense operand's ring slot of a tick without an output row (warm-up, past the end)
    // holds zeros (out-of-range copies write zeros; LDS is cleared at kernel start), so its MFMAs add 0.
    auto tick = [&](bf16x8 (&rd)[11], const bf16x8 (&mm)[11]) {
      // (stride 2: every wave has CPW copies per tick -- no branch; stride 1: wave 0 alone has a second one)
      if (G::NCOPY == 8 * G::CPW || two_copies) asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)" ::"n"((G::D - 1) * G::CPW) : "memory");
      else asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)" ::"n"((G::D - 1) * (G::CPW - 1)) : "memory");
      __builtin_amdgcn_sched_barrier(0);
      __builtin_amdgcn_s_barrier();
      __builtin_amdgcn_sched_barrier(0);
      const unsigned sa = c_a, sb = c_b, sb9 = c_b9;
#define WR_FRAG(dst, base, o0, o1)                                                                          \
      {                                                                                                     \
        const s16x4 lo = wr_tr16<0>(base + o0), hi = wr_tr16<0>(base + o1);                                 \
        dst = __builtin_bit_cast(bf16x8, __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7));          \
      }
#define WR_MMA(a, b) acc[a][b] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(mm[4 + b], mm[a], acc[a][b], 0, 0, 0)
#ifdef MT_WR_EXP_NOREAD
#undef WR_FRAG
#define WR_FRAG(dst, base, o0, o1) asm volatile("" : "+v"(dst))
#endif
#ifdef MT_WR_EXP_NOMFMA
#undef WR_MMA
#define WR_MMA(a, b) asm volatile("" : "+v"(acc[a][b]))
#endif
      WR_FRAG(rd[0], sa, aoff[0][0], aoff[1][0]);
      WR_FRAG(rd[1], sa, aoff[0][1], aoff[1][1]);
      WR_FRAG(rd[4], sb, boff[0][0], boff[1][0]);
      WR_FRAG(rd[5], sb, boff[0][1], boff[1][1]);
      __builtin_amdgcn_sched_barrier(0);
      WR_MMA(0, 0); WR_MMA(0, 1); WR_MMA(0, 2);
      __builtin_amdgcn_sched_barrier(0);
      WR_FRAG(rd[2], sa, aoff[0][2], aoff[1][2]);
      WR_FRAG(rd[6], sb, boff[0][2], boff[1][2]);
      __builtin_amdgcn_sched_barrier(0);
      WR_MMA(0, 3); WR_MMA(1, 0); WR_MMA(1, 1);
      __builtin_amdgcn_sched_barrier(0);
      WR_FRAG(rd[3], sa, aoff[0][3], aoff[1][3]);
      WR_FRAG(rd[7], sb, boff[0][3], boff[1][3]);
      __builtin_amdgcn_sched_barrier(0);
      // (the copies' address arithmetic rides between the MFMAs from here on: every MFMA holds the matrix pipe for 16 cycles)
      WR_MMA(1, 2);
      issue_copy(std::integral_constant<int, 0>{});
      WR_MMA(1, 3); WR_MMA(2, 0);
      __builtin_amdgcn_sched_barrier(0);
      WR_FRAG(rd[8], sa, aoff9[0], aoff9[1]);
      WR_FRAG(rd[9], sb9, boff9[0][0], boff9[1][0]);
      __builtin_amdgcn_sched_barrier(0);
      WR_MMA(2, 1);
      if constexpr (G::CPW > 2) issue_copy(std::integral_constant<int, 1>{});
      WR_MMA(2, 2); WR_MMA(2, 3);
      __builtin_amdgcn_sched_barrier(0);
      WR_FRAG(rd[10], sb9, boff9[0][1], boff9[1][1]);
      __builtin_amdgcn_sched_barrier(0);
      WR_MMA(3, 0);
      if constexpr (G::CPW > 2) issue_copy(std::integral_constant<int, G::CPW - 1>{});
      WR_MMA(3, 1); WR_MMA(3, 2); WR_MMA(3, 3);
#ifndef MT_WR_EXP_NOMFMA
      acc9[0] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(mm[9], mm[8], acc9[0], 0, 0, 0);
      acc9[1] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(mm[10], mm[8], acc9[1], 0, 0, 0);
#endif
#undef WR_FRAG
#undef WR_MMA
      advance_compute();
      __builtin_amdgcn_sched_barrier(0);
      if constexpr (G::CPW == 2) {                 // (stride 1: wave 0 alone has a second copy)
        if (two_copies) issue_copy(std::integral_constant<int, 1>{});
      }
      advance_load();
      __builtin_amdgcn_sched_barrier(0);
    };
    for (int j = 0; j <= T; j += 2) {
      tick(fP, fQ);
      tick(fQ, fP);
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#endif
    // the trailing (all-zero) copies must have landed, and every wave must be done with the ring, before the next run refills it
    // (or the wave exits)
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_sched_barrier(0);
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_sched_barrier(0);
  }

  // ---- epilogue: this workgroup's 64 x 9 x 64 block of the split's fp32 slab [CaRows][taps][Cb] ----
#ifdef MT_WR_EXP_NOSLAB
  if (acc[0][0][0] != 12345.678f) return;
#endif
  const int fr = lane & 15, fg = lane >> 4;
  const int cbt = ncb * 64, ncols = 9 * cbt;
  float* const slab = pout + (size_t)split * (size_t)(nca * 64) * ncols;
#pragma unroll
  for (int a = 0; a < 4; a++) {
    const int ca = cab * 64 + a * 16 + fr;
#pragma unroll
    for (int b = 0; b < 4; b++) {
      const int col = wv * cbt + cbb * 64 + b * 16 + fg * 4;
      *reinterpret_cast<f32x4*>(slab + (size_t)ca * ncols + col) = acc[a][b];
    }
  }
#pragma unroll
  for (int i = 0; i < 2; i++) {
    const int ca = cab * 64 + a9 * 16 + fr;
    const int col = 8 * cbt + cbb * 64 + (fb9 + i) * 16 + fg * 4;
    *reinterpret_cast<f32x4*>(slab + (size_t)ca * ncols + col) = acc9[i];
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

// Does the row walker take this problem?  (geometry only: p.a / p.b / p.out unused)
bool mt_wgrad_rows_ok(int dtype, const WgradParams& p) {
  if (!wr_enabled() || dtype != MT_BF16) return false;
  if (p.ntaps != 9 || (p.is != 1 && p.is != 2)) return false;
  for (int t = 0; t < 9; t++)
    if (p.dh[t] != t / 3 - 1 || p.dw[t] != t % 3 - 1) return false;
  const int cb = p.cpc * 8;
  if (p.CaRows % 64 || cb % 64 || p.Cab != p.CaRows * 2 || p.Cbb != cb * 2) return false;
  if (p.Wo % 32 || p.Hi != p.is * p.Ho || p.Wi != p.is * p.Wo || p.Hi < 2 || p.Wi < 2) return false;
  if ((double)p.M * p.Cab >= 2147483000.0 || (double)p.N * p.Hi * p.Wi * p.Cbb >= 2147483000.0) return false;
  if ((p.CaRows / 64) * (cb / 64) > 64) return false;
  if ((long)p.N * (p.Wo / 32) * p.Ho > 0x3fffffffL) return false;
  return true;
}
static inline long wr_rows(const WgradParams& p) { return (long)p.N * (p.Wo / 32) * p.Ho; }
static inline int wr_blocks(const WgradParams& p) { return (p.CaRows / 64) * (p.cpc * 8 / 64); }

// Pixel splits of n problems that share launches (one launch per stride class): the 256 compute units are dealt out in proportion
// to the problems' k-steps x channel blocks, every workgroup of a launch walks about the same number of ticks, and a problem's slab
// count shrinks with the company it has (10 layers in one launch: ~25 slabs each instead of 256).  nsplit[i] slabs, rps[i] k-steps
// per split.  All problems must have passed mt_wgrad_rows_ok.
void mt_wgrad_rows_plan_multi(int n, const WgradParams* ps, int* nsplit, int* rps) {
  for (int S = 1; S <= 2; S++) {
    double total = 0;
    int cnt = 0;
    for (int i = 0; i < n; i++)
      if (ps[i].is == S) { total += (double)wr_rows(ps[i]) * wr_blocks(ps[i]); cnt++; }
    if (!cnt) continue;
    // (launches of more than MT_WR_MAXP problems are cut into chunks by the launcher; the budget is per launch)
    const int launches = (cnt + MT_WR_MAXP - 1) / MT_WR_MAXP;
    int budget = 256 * launches;
    for (int round = 0; round < 16; round++) {
      long used = 0;
      for (int i = 0; i < n; i++) {
        if (ps[i].is != S) continue;
        const long rows = wr_rows(ps[i]);
        const int nb = wr_blocks(ps[i]);
        long w = (long)(budget * ((double)rows * nb / total));
        long ns = w / nb;
        if (ns < 1) ns = 1;
        if (ns > rows / 8) ns = rows / 8 > 0 ? rows / 8 : 1;
        const long r = (rows + ns - 1) / ns;
        ns = (rows + r - 1) / r;
        nsplit[i] = (int)ns;
        rps[i] = (int)r;
        used += ((ns * nb + 7) & ~7L);
      }
      if (used <= 256L * launches) break;
      budget -= (int)(used - 256L * launches) + 1;
      if (budget < cnt) break;
    }
  }
}

// single problem: is it worth a launch of its own, and with which split?  *rs = k-steps per split
bool mt_wgrad_rows_plan(int dtype, const WgradParams& p, int* nsplit, int* rs) {
  if (!mt_wgrad_rows_ok(dtype, p)) return false;
  if (wr_rows(p) * wr_blocks(p) < 256L * 16) return false;      // less than 16 ticks per compute unit: the tile kernels' splits serve it
  mt_wgrad_rows_plan_multi(1, &p, nsplit, rs);
  return true;
}

// n problems (ps[i].a / .b / .out set; out = the problem's nsplit[i] slabs) in one launch per stride class and MT_WR_MAXP problems
int mt_launch_wgrad_rows_multi(int n, const WgradParams* ps, const int* nsplit, const int* rps, hipStream_t s) {
  for (int S = 1; S <= 2; S++) {
    int i = 0;
    while (i < n) {
      WrMulti m;
      memset(&m, 0, sizeof(m));
      int blocks = 0;
      for (; i < n && m.n < MT_WR_MAXP; i++) {
        const WgradParams& p = ps[i];
        if (p.is != S) continue;
        MT_CHECK(nsplit[i] > 0 && rps[i] > 0 && (long)nsplit[i] * rps[i] >= wr_rows(p) && (long)(nsplit[i] - 1) * rps[i] < wr_rows(p),
                 "wgrad_rows: split %d x %d does not cover %ld k-steps", nsplit[i], rps[i], wr_rows(p));
        WrProb& q = m.p[m.n++];
        q.a = p.a; q.b = p.b; q.out = p.out;
        q.a_bytes = (unsigned)((size_t)p.M * p.Cab);
        q.b_bytes = (unsigned)((size_t)p.N * p.Hi * p.Wi * p.Cbb);
        q.Ho = p.Ho; q.Wo = p.Wo; q.Hi = p.Hi; q.Wi = p.Wi; q.Cab = p.Cab; q.Cbb = p.Cbb;
        q.nca = p.CaRows / 64; q.ncb = p.cpc * 8 / 64;
        q.rows = (int)wr_rows(p); q.rps = rps[i];
        q.blk0 = blocks; q.nblk = nsplit[i] * q.nca * q.ncb;
        q.reflect = p.pad_mode == MT_PAD_REFLECT;
        blocks += (q.nblk + 7) & ~7;
      }
      if (m.n == 0) break;
      if (S == 1) hipLaunchKernelGGL((wgrad_rows_kernel<1>), dim3((unsigned)blocks), dim3(512), 0, s, m);
      else hipLaunchKernelGGL((wgrad_rows_kernel<2>), dim3((unsigned)blocks), dim3(512), 0, s, m);
      MT_LAUNCH_CHECK();
      g_wr_launches++;
    }
  }
  return 0;
}

int mt_launch_wgrad_rows(const WgradParams& p, int nsplit, hipStream_t s) {
  MT_CHECK(p.rows_rs > 0, "wgrad_rows: no split");
  const int rps = p.rows_rs;
  return mt_launch_wgrad_rows_multi(1, &p, &nsplit, &rps, s);
}
