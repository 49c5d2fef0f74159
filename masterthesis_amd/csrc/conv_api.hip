// Host side of the convolution family: turns an mt_conv_desc into gather-GEMM launches.
//  - Conv2d forward: one gather-GEMM, reflection folded into the gather (no padded copy).
//  - Conv2d data gradient / ConvTranspose2d forward: stride^2 sub-pixel phases, each a
//    stride-1 gather-GEMM over the taps of matching parity writing a strided output grid;
//    for reflect padding the phases write the gradient of the padded map into the
//    workspace and reflect_fold adds the border back (adjoint of ReflectionPad2d).
//  - weight gradients: wgrad kernel into an fp32 [row][tap][col] workspace, then unpack to
//    the reference layout (OIHW / IOHW).
// Reference call sites: blocks.py:29-35 (pad+Conv2d), blocks.py:73 (ConvTranspose2d).
#include "mt_common.h"
#include "conv_params.h"
#include <string.h>
#include <stdlib.h>

static inline int esz(int dtype) { return dtype == MT_BF16 ? 2 : 4; }
static inline int vec(int dtype) { return 16 / esz(dtype); }
static inline int posmod(int a, int m) { return ((a % m) + m) % m; }

static int check_desc(const mt_conv_desc* d) {
  MT_CHECK(d != nullptr, "conv: null descriptor");
  MT_CHECK(d->dtype == MT_F32 || d->dtype == MT_BF16, "conv: bad dtype %d", d->dtype);
  MT_CHECK(d->N > 0 && d->H > 0 && d->W > 0 && d->Ci > 0 && d->Co > 0, "conv: bad shape");
  MT_CHECK(d->kh > 0 && d->kw > 0 && d->kh * d->kw <= MT_MAX_TAPS, "conv: %dx%d filter unsupported", d->kh, d->kw);
  MT_CHECK(d->stride >= 1 && d->stride <= 4, "conv: stride %d unsupported", d->stride);
  MT_CHECK(d->pad >= 0, "conv: negative padding");
  if (!d->transposed && d->pad_mode == MT_PAD_REFLECT)
    MT_CHECK(d->pad < d->H && d->pad < d->W, "conv: reflect pad %d >= input size", d->pad);
  if (d->transposed) MT_CHECK(d->pad_mode == MT_PAD_ZERO, "convT: reflect padding unsupported");
  int Ho, Wo;
  if (d->transposed) {
    Ho = (d->H - 1) * d->stride - 2 * d->pad + d->kh + d->out_pad;
    Wo = (d->W - 1) * d->stride - 2 * d->pad + d->kw + d->out_pad;
  } else {
    Ho = (d->H + 2 * d->pad - d->kh) / d->stride + 1;
    Wo = (d->W + 2 * d->pad - d->kw) / d->stride + 1;
    MT_CHECK(d->H + 2 * d->pad >= d->kh && d->W + 2 * d->pad >= d->kw, "conv: filter larger than padded input");
  }
  MT_CHECK(Ho > 0 && Wo > 0, "conv: empty output");
  return 0;
}

extern "C" int mt_conv_out_hw(const mt_conv_desc* d, int* Ho, int* Wo) {
  if (check_desc(d)) return 1;
  if (d->transposed) {
    *Ho = (d->H - 1) * d->stride - 2 * d->pad + d->kh + d->out_pad;
    *Wo = (d->W - 1) * d->stride - 2 * d->pad + d->kw + d->out_pad;
  } else {
    *Ho = (d->H + 2 * d->pad - d->kh) / d->stride + 1;
    *Wo = (d->W + 2 * d->pad - d->kw) / d->stride + 1;
  }
  return 0;
}

// Does `which` use the phase-decomposed ("scatter form") pack?
static inline bool phased(const mt_conv_desc* d, int which) {
  return d->transposed ? (which == MT_PACK_FWD) : (which == MT_PACK_BWD_DATA);
}

// ---- stride-1 reflect-padded Conv2d data gradient ("interior + ring") -------------------------------------
// dx = fold(dxp), dxp = full correlation of dy on the padded (H+2P)x(W+2P) grid.  Computing all of dxp wastes a
// ragged grid (K1: 1090 tiles of 128x128 = 2.13 rounds of the chip); instead
//   (1) the interior pre-image dxp[h+P][w+P] is written straight into dx (an HxW grid: K1 = 256 tiles of the
//       256x256 ping-pong kernel, exactly one round),
//   (2) the four P-wide border strips of dxp go to the workspace -- only the filter taps that can reach dy
//       from there (P*kw resp. kh*P of the kh*kw taps) -- and
//   (3) ring_fold adds them onto the border band of dx.
// The data-gradient weight pack therefore carries the tap runs [all | top | bottom | left | right] in ONE image
// (rows of T taps); a launch phase addresses its run through IgemmPhase::{w_off, wrow}.
struct RingTaps {
  int total;               // taps in the pack image
  int start[5], count[5];  // runs: 0 all, 1 top, 2 bottom, 3 left, 4 right
  short kh[MT_MAX_TAPS], kw[MT_MAX_TAPS];
};
static bool ring_taps(const mt_conv_desc* d, RingTaps* r) {
  if (d->transposed || d->pad_mode != MT_PAD_REFLECT || d->pad == 0 || d->stride != 1 || mt_pointwise_small(d))
    return false;
  const int P = d->pad;
  int n = 0;
  for (int run = 0; run < 5; run++) {
    r->start[run] = n;
    for (int a = 0; a < d->kh; a++)
      for (int b = 0; b < d->kw; b++) {
        const bool use = run == 0 || (run == 1 && a <= P - 1) || (run == 2 && a >= d->kh - P) ||
                         (run == 3 && b <= P - 1) || (run == 4 && b >= d->kw - P);
        if (!use) continue;
        if (n >= MT_MAX_TAPS) return false;   // (7x7 stem: the padded-grid path stays)
        r->kh[n] = (short)a; r->kw[n] = (short)b; n++;
      }
    r->count[run] = n - r->start[run];
  }
  r->total = n;
  return true;
}

extern "C" size_t mt_conv_pack_bytes(const mt_conv_desc* d, int which) {
  RingTaps r;
  const int taps = (which == MT_PACK_BWD_DATA && ring_taps(d, &r)) ? r.total : d->kh * d->kw;
  return (size_t)mt_padc(d->Ci) * mt_padc(d->Co) * taps * esz(d->dtype);
}

// taps of phase (ph, pw) in (kh, kw) order
static int phase_taps(const mt_conv_desc* d, int ph, int pw, short* kh, short* kw) {
  int n = 0;
  for (int a = 0; a < d->kh; a++)
    for (int b = 0; b < d->kw; b++)
      if (a % d->stride == ph && b % d->stride == pw) { kh[n] = (short)a; kw[n] = (short)b; n++; }
  return n;
}

// The pack images of one (descriptor, which): 1 for the gather form and the ring image, stride^2 for the scatter form.
// emit(PackParams, byte offset into `pack`) is called per image.
template <class F>
static int pack_images(const mt_conv_desc* d, int which, F emit) {
  const int K2 = d->kh * d->kw;
  PackParams p;
  memset(&p, 0, sizeof(p));
  p.kW = d->kw;
  // which tensor dimension is the GEMM row (output channel of the launch) / column
  long s_ci, s_co;  // element strides of ci / co in the reference layout
  if (d->transposed) { s_ci = (long)d->Co * K2; s_co = K2; }   // [Ci][Co][kh][kw]
  else { s_co = (long)d->Ci * K2; s_ci = K2; }                 // [Co][Ci][kh][kw]
  // thin 1x1 convolutions (pointwise_kernels.hip) use the [co][ci] image for forward AND data gradient
  const bool rows_are_co = mt_pointwise_small(d) || (which == MT_PACK_FWD);
  if (rows_are_co) { p.R = d->Co; p.C = d->Ci; p.sr = s_co; p.sc = s_ci; }
  else { p.R = d->Ci; p.C = d->Co; p.sr = s_ci; p.sc = s_co; }
  p.Rp = mt_padc(p.R);
  p.Cp = mt_padc(p.C);
  RingTaps rt;
  if (which == MT_PACK_BWD_DATA && ring_taps(d, &rt)) {
    p.ntaps = rt.total;
    memcpy(p.kh, rt.kh, sizeof(short) * rt.total);
    memcpy(p.kw, rt.kw, sizeof(short) * rt.total);
    return emit(p, (size_t)0);
  }
  if (!phased(d, which)) {
    p.ntaps = K2;
    for (int a = 0; a < d->kh; a++)
      for (int b = 0; b < d->kw; b++) { p.kh[a * d->kw + b] = (short)a; p.kw[a * d->kw + b] = (short)b; }
    return emit(p, (size_t)0);
  }
  size_t off = 0;
  for (int ph = 0; ph < d->stride; ph++)
    for (int pw = 0; pw < d->stride; pw++) {
      p.ntaps = phase_taps(d, ph, pw, p.kh, p.kw);
      if (p.ntaps == 0) continue;
      if (emit(p, off)) return 2;
      off += (size_t)p.Rp * p.ntaps * p.Cp * esz(d->dtype);
    }
  return 0;
}

extern "C" int mt_conv_pack(const mt_conv_desc* d, int which, const float* w, void* pack, mt_stream_t st) {
  if (check_desc(d)) return 1;
  hipStream_t s = (hipStream_t)st;
  return pack_images(d, which, [&](const PackParams& p, size_t off) {
    return mt_launch_pack(d->dtype, w, (char*)pack + off, p, s);
  });
}

// ---- batched pack: all weight images of a network in one launch ----------------------------------------------
// mt_conv_pack_multi_build fills a HOST table (the caller copies it to the device once; weight and pack addresses
// are stable across steps), mt_conv_pack_multi_run launches over the device copy.  Table = [PackGroup x n] [PackEntry x ...]:
// images of big k <= 4 weights go through the grouped LDS-tiled kernel (one read of the tensor for all of its images),
// the rest (7x7 stem, thin 1x1 layers, small tensors) through the element-wise kernel.  The two counts returned to the
// caller carry both kinds: n_entries = entries | groups << 16, total_blocks = entry blocks | n << 16.
static size_t pack_groups_bytes(int n) { return (((size_t)n * sizeof(PackGroup)) + 255) & ~(size_t)255; }
extern "C" size_t mt_conv_pack_multi_table_bytes(int n) {
  return pack_groups_bytes(n) + (size_t)n * MT_MAX_PHASES * sizeof(PackEntry);
}
extern "C" int mt_conv_pack_multi_build(int n, const mt_conv_desc* descs, const int* which, const float* const* w,
                                        void* const* packs, void* host_table, int* n_entries, int* total_blocks) {
  PackGroup* grp = (PackGroup*)host_table;
  PackEntry* tab = (PackEntry*)((char*)host_table + pack_groups_bytes(n));
  int ne = 0, blocks = 0, ng = 0;
  for (int i = 0; i < n; i++) {
    const mt_conv_desc* d = &descs[i];
    if (check_desc(d)) return 1;
    const int K2 = d->kh * d->kw;
    const int D0 = d->transposed ? d->Ci : d->Co, D1 = d->transposed ? d->Co : d->Ci;
    const int rc = pack_images(d, which[i], [&](const PackParams& p, size_t off) {
      const long total = (long)p.Rp * p.ntaps * p.Cp;
      if (total == 0) return 0;
      // grouped path: the image is one of the two orientations of the contiguous [D0][D1][K2] tensor
      const bool r0 = p.sr == (long)D1 * K2 && p.sc == K2 && p.R == D0 && p.C == D1;
      const bool r1 = p.sr == K2 && p.sc == (long)D1 * K2 && p.R == D1 && p.C == D0;
      if ((r0 || r1) && d->dtype == MT_BF16 && K2 <= 16 && K2 >= 4 && p.kW == d->kw && (long)D0 * D1 >= 8192) {
        int gi = -1;
        for (int k = 0; k < ng; k++)
          if (grp[k].w == w[i] && grp[k].D0 == D0 && grp[k].D1 == D1 && grp[k].K2 == K2 &&
              grp[k].bf16 == (d->dtype == MT_BF16) && grp[k].nout < MT_PACK_GROUP_OUTS)
            gi = k;
        if (gi < 0) {
          gi = ng++;
          memset(&grp[gi], 0, sizeof(PackGroup));
          grp[gi].w = w[i]; grp[gi].D0 = D0; grp[gi].D1 = D1; grp[gi].K2 = K2; grp[gi].bf16 = d->dtype == MT_BF16;
        }
        PackOut& o = grp[gi].o[grp[gi].nout++];
        o.out = (char*)packs[i] + off; o.rows_d0 = r0 ? 1 : 0; o.Rp = p.Rp; o.Cp = p.Cp; o.ntaps = p.ntaps;
        for (int t = 0; t < p.ntaps; t++) o.tsrc[t] = (unsigned char)(p.kh[t] * p.kW + p.kw[t]);
        return 0;
      }
      PackEntry& e = tab[ne++];
      e.p = p; e.w = w[i]; e.out = (char*)packs[i] + off; e.bf16 = d->dtype == MT_BF16;
      e.blk0 = blocks;
      e.nblk = (int)(total + 255) / 256 < 512 ? (int)((total + 255) / 256) : 512;
      blocks += e.nblk;
      return 0;
    });
    if (rc) return rc;
  }
  int tiles = 0;
  for (int k = 0; k < ng; k++) {
    // (tiles cover the padded extents of every image: rows / columns past the logical size are written as zeros)
    int e0 = grp[k].D0, e1 = grp[k].D1;
    for (int j = 0; j < grp[k].nout; j++) {
      const PackOut& o = grp[k].o[j];
      e0 = (o.rows_d0 ? o.Rp : o.Cp) > e0 ? (o.rows_d0 ? o.Rp : o.Cp) : e0;
      e1 = (o.rows_d0 ? o.Cp : o.Rp) > e1 ? (o.rows_d0 ? o.Cp : o.Rp) : e1;
    }
    grp[k].tiles_d1 = (e1 + 31) / 32;
    grp[k].ntiles = ((e0 + 31) / 32) * grp[k].tiles_d1;
    grp[k].tile0 = tiles;
    tiles += grp[k].ntiles;
  }
  MT_CHECK(ne < 65536 && ng < 32768 && blocks < 65536 && n < 65536, "pack_multi: table too large");
  // (the element-wise blocks are capped at 512 per entry so that their sum fits 16 bits for any realistic network)
  *n_entries = ne | (ng << 16);
  *total_blocks = blocks | (n << 16);       // n locates the entries behind the groups' slots in the table
  return 0;
}
extern "C" int mt_conv_pack_multi_run(const void* dev_table, int n_entries, int total_blocks, mt_stream_t st) {
  const int ne = n_entries & 0xffff, ng = (n_entries >> 16) & 0x7fff;
  const int blocks = total_blocks & 0xffff, n = (total_blocks >> 16) & 0xffff;
  const PackGroup* grp = (const PackGroup*)dev_table;
  const PackEntry* tab = (const PackEntry*)((const char*)dev_table + pack_groups_bytes(n));
  // (the group kernel reads its tile count from the table on the device; 2048 blocks walk the tiles)
  if (mt_launch_pack_groups(grp, ng, 2048, (hipStream_t)st)) return 2;
  return mt_launch_pack_multi(tab, ne, blocks, (hipStream_t)st);
}

// scatter-form launches shared by Conv2d bwd_data and ConvTranspose2d fwd.
//   out[o] = sum_k in[(o + e - k)/stride] * W[k]   over k with (o + e - k) % stride == 0
// in: [N][Hin][Win][Cin_p], out: [N][Hd][Wd][Cout_p]
// split-K plan of a scatter-form launch (see gather_splitk): number of tap splits per sub-pixel phase (1 = none)
static int scatter_splitk(const mt_conv_desc* d, int Hd, int Wd, int Cin_p, int Cout_p, int e) {
  const int st = d->stride, V = vec(d->dtype);
  const int WT = Cout_p > 64 ? 128 : (Cout_p > 32 ? 64 : (Cout_p > 16 ? 32 : 16));
  int tiles = 0, maxtaps = 0, np = 0;
  for (int ph = 0; ph < st; ph++)
    for (int pw = 0; pw < st; pw++) {
      short kh[MT_MAX_TAPS], kw[MT_MAX_TAPS];
      const int nt = phase_taps(d, ph, pw, kh, kw);
      const int o0h = posmod(ph - e, st), o0w = posmod(pw - e, st);
      const int Hg = o0h < Hd ? (Hd - o0h + st - 1) / st : 0;
      const int Wg = o0w < Wd ? (Wd - o0w + st - 1) / st : 0;
      const int M = d->N * Hg * Wg;
      if (M == 0) continue;
      np++;
      tiles += cdiv(M, 128) * cdiv(Cout_p, WT);
      maxtaps = nt > maxtaps ? nt : maxtaps;
    }
  if (np == 0 || tiles > 128 || maxtaps < 2) return 1;
  const int nk = cdiv((long)maxtaps * (Cin_p / V), 8);
  if (nk < 32) return 1;
  int ks = 512 / tiles;
  if (ks > maxtaps) ks = maxtaps;
  if (ks > MT_MAX_PHASES / np) ks = MT_MAX_PHASES / np;
  while (ks > 1 && nk / ks < 8) ks--;
  return ks < 2 ? 1 : ks;
}

// split_ws: optional fp32 workspace of scatter_splitk(...) * N*Hd*Wd*Cout_p floats for the split-K partial slabs
// interior / iP: optional second destination for the pixels [iP, Hd-iP) x [iP, Wd-iP) of the (Hd, Wd) grid (the
// un-padded gradient map of a reflection-padded convolution): used -- return value 100 -- only when the persistent
// gather-GEMM takes the launch; then `out` receives the border ring only and the caller folds just that.
static int scatter_form(const mt_conv_desc* d, const void* in, int Hin, int Win, int Cin_p, const void* pack,
                        const float* bias, int nbias, void* out, int Hd, int Wd, int Cout_p, int e, int act,
                        hipStream_t s, void* split_ws = nullptr, size_t split_ws_bytes = 0, void* interior = nullptr,
                        int iP = 0) {
  const int sz = esz(d->dtype), V = vec(d->dtype), st = d->stride;
  MT_CHECK(st * st <= MT_MAX_PHASES, "conv: stride %d has too many phases", st);
  IgemmParams p;
  memset(&p, 0, sizeof(p));
  p.x = (const char*)in; p.w = (const char*)pack; p.bias = bias; p.nbias = nbias; p.y = (char*)out;
  p.N = d->N; p.Hi = Hin; p.Wi = Win; p.Cib = Cin_p * sz;
  p.Co = Cout_p; p.CoRows = Cout_p;
  p.Hout = Hd; p.Wout = Wd; p.os = st; p.is = 1;
  p.cpc = Cin_p / V;
  p.pad_mode = MT_PAD_ZERO; p.act = act; p.slope = d->slope;
  size_t w_off = 0;
  int tap0 = 0;
  for (int ph = 0; ph < st; ph++)
    for (int pw = 0; pw < st; pw++) {
      short kh[MT_MAX_TAPS], kw[MT_MAX_TAPS];
      const int nt = phase_taps(d, ph, pw, kh, kw);
      const int o0h = posmod(ph - e, st), o0w = posmod(pw - e, st);
      const int Hg = o0h < Hd ? (Hd - o0h + st - 1) / st : 0;
      const int Wg = o0w < Wd ? (Wd - o0w + st - 1) / st : 0;
      IgemmPhase& q = p.ph[p.nphase];
      q.w_off = (unsigned)w_off; q.ntaps = nt; q.tap0 = tap0;
      q.wrow = nt * p.cpc; q.w_bytes = (unsigned)((size_t)Cout_p * nt * Cin_p * sz);
      q.Ho = Hg; q.Wo = Wg; q.M = d->N * Hg * Wg; q.oh0 = o0h; q.ow0 = o0w;
      for (int t = 0; t < nt; t++) {
        p.dh[tap0 + t] = (short)((o0h + e - kh[t]) / st);
        p.dw[tap0 + t] = (short)((o0w + e - kw[t]) / st);
      }
      tap0 += nt;
      w_off += (size_t)Cout_p * nt * Cin_p * sz;
      if (q.M > 0) p.nphase++;   // an empty phase (no output pixels) is dropped; a phase without taps writes zeros
    }
  if (p.nphase == 0) return 0;
  const int ks = split_ws != nullptr ? scatter_splitk(d, Hd, Wd, Cin_p, Cout_p, e) : 1;
  const size_t slab = (size_t)d->N * Hd * Wd * Cout_p * sizeof(float);
  if (ks > 1 && split_ws_bytes >= slab * ks && slab * ks < 0xf0000000ull && p.nphase * ks <= MT_MAX_PHASES) {
    // every sub-pixel phase is split over its taps; split i of every phase writes slab i (a split without taps
    // writes zeros there, so the finish kernel can sum all slabs everywhere)
    IgemmPhase orig[MT_MAX_PHASES];
    const int np = p.nphase;
    for (int i = 0; i < np; i++) orig[i] = p.ph[i];
    p.nphase = 0;
    for (int j = 0; j < np; j++) {
      const int base = orig[j].ntaps / ks, rem = orig[j].ntaps % ks;
      int t0 = 0;
      for (int i = 0; i < ks; i++) {
        IgemmPhase& f = p.ph[p.nphase++];
        f = orig[j];
        f.ntaps = base + (i < rem ? 1 : 0);
        f.tap0 = orig[j].tap0 + t0;
        const size_t sub = (size_t)t0 * Cin_p * sz;
        f.w_off = orig[j].w_off + (unsigned)sub; f.w_bytes = orig[j].w_bytes - (unsigned)sub;
        f.y_off = (unsigned)(slab * i);
        t0 += f.ntaps;
      }
    }
    p.y = (char*)split_ws; p.raw = 1; p.bias = nullptr; p.nbias = 0; p.act = MT_ACT_NONE;
    if (mt_launch_igemm(d->dtype, p, s)) return 2;
    return mt_launch_splitk_finish(d->dtype, (const float*)split_ws, ks, (long)d->N * Hd * Wd * Cout_p, bias, nbias,
                                   Cout_p, out, act, d->slope, s);
  }
  if (interior != nullptr && iP > 0 && Hd > 2 * iP && Wd > 2 * iP && mt_igemm_would_persist(d->dtype, p)) {
    p.y2 = (char*)interior; p.y2P = iP; p.y2H = Hd - 2 * iP; p.y2W = Wd - 2 * iP;
    return mt_launch_igemm(d->dtype, p, s) ? 2 : 100;
  }
  return mt_launch_igemm(d->dtype, p, s);
}

// gather-form launch shared by Conv2d fwd and ConvTranspose2d bwd_data.
//   out[o] = sum_k in[o*stride - pad + k] * W[k]
// split-K for few-pixel / long-K problems (the discriminators' deep layers: 16-64 tiles walking 144-256 k-steps
// each): the filter taps are divided over up to 16 launch phases that write fp32 partial slabs, a finish kernel
// sums them and applies bias + activation.  Returns the number of splits (1 = no split).
static int gather_splitk(int M, int Cout_p, int Cin_p, int ntaps, int V) {
  const int WT = Cout_p > 64 ? 128 : (Cout_p > 32 ? 64 : (Cout_p > 16 ? 32 : 16));
  const int tiles = cdiv(M, 128) * cdiv(Cout_p, WT);
  const int nk = cdiv((long)ntaps * (Cin_p / V), 8);
#ifndef MT_SPLITK_MAX_TILES
#define MT_SPLITK_MAX_TILES 256      // (round 3: 128 -> 256, the merged mini-image batches of the discriminators: step -0.13 ms)
#endif
  if (tiles > MT_SPLITK_MAX_TILES || nk < 48 || ntaps < 2) return 1;
  int ks = 512 / tiles;
  if (ks > ntaps) ks = ntaps;
  if (ks > MT_MAX_PHASES) ks = MT_MAX_PHASES;
  while (ks > 1 && nk / ks < 8) ks--;      // keep at least 8 k-steps per split
  return ks < 2 ? 1 : ks;
}

static int gather_form(const mt_conv_desc* d, const void* in, int Hin, int Win, int Cin_p, const void* pack,
                       const float* bias, int nbias, void* out, int Hg, int Wg, int Cout_p, int pad_mode, int act,
                       hipStream_t s, float* stats = nullptr, void* ws = nullptr, size_t ws_bytes = 0) {
  const int sz = esz(d->dtype), V = vec(d->dtype);
  IgemmParams p;
  memset(&p, 0, sizeof(p));
  p.x = (const char*)in; p.w = (const char*)pack; p.bias = bias; p.nbias = nbias; p.y = (char*)out;
  p.stats = stats;
  p.N = d->N; p.Hi = Hin; p.Wi = Win; p.Cib = Cin_p * sz;
  p.Co = Cout_p; p.CoRows = Cout_p;
  p.Hout = Hg; p.Wout = Wg; p.os = 1; p.is = d->stride;
  p.cpc = Cin_p / V;
  p.pad_mode = pad_mode; p.act = act; p.slope = d->slope;
  p.nphase = 1;
  IgemmPhase& q = p.ph[0];
  q.w_off = 0; q.ntaps = d->kh * d->kw; q.tap0 = 0; q.Ho = Hg; q.Wo = Wg; q.M = d->N * Hg * Wg; q.oh0 = 0; q.ow0 = 0;
  q.wrow = q.ntaps * p.cpc; q.w_bytes = (unsigned)((size_t)Cout_p * q.ntaps * Cin_p * sz);
  for (int a = 0; a < d->kh; a++)
    for (int b = 0; b < d->kw; b++) {
      p.dh[a * d->kw + b] = (short)(a - d->pad);
      p.dw[a * d->kw + b] = (short)(b - d->pad);
    }
  const int ks = (stats == nullptr && ws != nullptr) ? gather_splitk(q.M, Cout_p, Cin_p, q.ntaps, V) : 1;
  const size_t slab = (size_t)q.M * Cout_p * sizeof(float);
  if (ks > 1 && ws_bytes >= slab * ks && slab * ks < 0xf0000000ull) {
    const IgemmPhase full = q;
    const int base = full.ntaps / ks, rem = full.ntaps % ks;
    int t0 = 0;
    for (int i = 0; i < ks; i++) {
      IgemmPhase& f = p.ph[i];
      f = full;
      f.tap0 = t0; f.ntaps = base + (i < rem ? 1 : 0);
      f.w_off = (unsigned)((size_t)t0 * Cin_p * sz); f.w_bytes = full.w_bytes - f.w_off;
      f.y_off = (unsigned)(slab * i);
      t0 += f.ntaps;
    }
    p.nphase = ks;
    p.y = (char*)ws; p.raw = 1; p.bias = nullptr; p.nbias = 0; p.act = MT_ACT_NONE;
    if (mt_launch_igemm(d->dtype, p, s)) return 2;
    return mt_launch_splitk_finish(d->dtype, (const float*)ws, ks, (long)q.M * Cout_p, bias, nbias, Cout_p, out, act,
                                   d->slope, s);
  }
  return mt_launch_igemm(d->dtype, p, s);
}

extern "C" size_t mt_conv_fwd_ws_bytes(const mt_conv_desc* d) {
  if (d->transposed || mt_pointwise_small(d)) return 0;
  int Ho, Wo;
  if (mt_conv_out_hw(d, &Ho, &Wo)) return 0;
  const int Cip = mt_padc(d->Ci), Cop = mt_padc(d->Co);
  const int M = d->N * Ho * Wo;
  const int ks = gather_splitk(M, Cop, Cip, d->kh * d->kw, vec(d->dtype));
  return ks > 1 ? (size_t)ks * M * Cop * sizeof(float) : 0;
}

extern "C" int mt_conv_fwd_ex(const mt_conv_desc* d, const void* x, const void* pack_fwd, const float* bias, void* y,
                              void* ws, size_t ws_bytes, mt_stream_t st) {
  if (check_desc(d)) return 1;
  hipStream_t s = (hipStream_t)st;
  int Ho, Wo;
  mt_conv_out_hw(d, &Ho, &Wo);
  const int Cip = mt_padc(d->Ci), Cop = mt_padc(d->Co);
  if (mt_pointwise_small(d)) return mt_pw_fwd(d, x, pack_fwd, bias, y, (long)d->N * d->H * d->W, s);
  if (!d->transposed) {
    const int r = mt_launch_stem_fwd(d, x, pack_fwd, bias, y, nullptr, s);     // the 7x7 stem: direct kernel
    if (r >= 0) return r;
  }
  if (!d->transposed)
    return gather_form(d, x, d->H, d->W, Cip, pack_fwd, bias, d->Co, y, Ho, Wo, Cop, d->pad_mode, d->act, s, nullptr, ws,
                       ws_bytes);
  return scatter_form(d, x, d->H, d->W, Cip, pack_fwd, bias, d->Co, y, Ho, Wo, Cop, d->pad, d->act, s);
}

extern "C" int mt_conv_fwd(const mt_conv_desc* d, const void* x, const void* pack_fwd, const float* bias, void* y,
                           mt_stream_t st) {
  return mt_conv_fwd_ex(d, x, pack_fwd, bias, y, nullptr, 0, st);
}

static bool stats_fusable(const mt_conv_desc* d) {
  int Ho, Wo;
  mt_conv_out_hw(d, &Ho, &Wo);
  return !d->transposed && d->act == MT_ACT_NONE && ((Ho * Wo) % 256 == 0);   // a block's pixel tile stays inside one image
}
extern "C" int mt_conv_fwd_stats_fused(const mt_conv_desc* d) {
  if (check_desc(d)) return 0;
  return stats_fusable(d) ? 1 : 0;
}

// Forward + per-(image, channel) {sum, sum^2} of the output (the InstanceNorm / AdaIN / LayerNorm
// statistics pass), fused into the GEMM epilogue; only for shapes where a wave's pixels cannot straddle two
// images (mt_conv_fwd_stats_fused).  stats: fp32 [N][Cop][2], MUST BE ZERO on entry (the
// epilogue accumulates with atomics).
extern "C" int mt_conv_fwd_stats(const mt_conv_desc* d, const void* x, const void* pack_fwd, const float* bias,
                                 void* y, float* stats, mt_stream_t st) {
  if (check_desc(d)) return 1;
  MT_CHECK(stats != nullptr, "conv_fwd_stats: null stats");
  MT_CHECK(d->act == MT_ACT_NONE, "conv_fwd_stats: statistics are taken of the conv output, activation must be none");
  hipStream_t s = (hipStream_t)st;
  int Ho, Wo;
  mt_conv_out_hw(d, &Ho, &Wo);
  const int Cip = mt_padc(d->Ci), Cop = mt_padc(d->Co);
  MT_CHECK(stats_fusable(d), "conv_fwd_stats: this shape has no fused statistics epilogue (mt_conv_fwd_stats_fused() == 0): "
                             "run mt_conv_fwd followed by mt_nc_stats");
  {
    const int r = mt_launch_stem_fwd(d, x, pack_fwd, bias, y, stats, s);
    if (r >= 0) return r;
  }
  return gather_form(d, x, d->H, d->W, Cip, pack_fwd, bias, d->Co, y, Ho, Wo, Cop, d->pad_mode, d->act, s, stats);
}

// workspace of the data gradient: [padded gradient map (reflect padding only)] [split-K partial slabs (if any)]
static size_t bwd_data_padded_bytes(const mt_conv_desc* d) {
  if (d->transposed || d->pad_mode != MT_PAD_REFLECT || d->pad == 0) return 0;
  const size_t b = (size_t)d->N * (d->H + 2 * d->pad) * (d->W + 2 * d->pad) * mt_padc(d->Ci) * esz(d->dtype);
  return (b + 255) & ~(size_t)255;
}
static size_t bwd_data_split_bytes(const mt_conv_desc* d) {
  if (d->transposed || mt_pointwise_small(d)) return 0;
  RingTaps rt;
  if (ring_taps(d, &rt)) return 0;
  const int P = (d->pad_mode == MT_PAD_REFLECT) ? d->pad : 0;
  const int Hd = d->H + 2 * P, Wd = d->W + 2 * P, Cip = mt_padc(d->Ci), Cop = mt_padc(d->Co);
  const int ks = scatter_splitk(d, Hd, Wd, Cop, Cip, P ? 0 : d->pad);
  return ks > 1 ? (size_t)ks * d->N * Hd * Wd * Cip * sizeof(float) : 0;
}
extern "C" size_t mt_conv_bwd_data_ws_bytes(const mt_conv_desc* d) {
  return bwd_data_padded_bytes(d) + bwd_data_split_bytes(d);
}

// *added: set when `addend` went into the kernel's epilogue (else the caller adds it afterwards)
static int conv_bwd_data_impl(const mt_conv_desc* d, const void* dy, const void* pack_bwd, void* dx, const void* addend,
                              bool* added, void* ws, size_t ws_bytes, mt_stream_t st, const mt_bwd_stats* bs = nullptr,
                              bool* stats_done = nullptr) {
  *added = false;
  if (stats_done) *stats_done = false;
  if (check_desc(d)) return 1;
  hipStream_t s = (hipStream_t)st;
  int Ho, Wo;
  mt_conv_out_hw(d, &Ho, &Wo);
  const int Cip = mt_padc(d->Ci), Cop = mt_padc(d->Co);
  if (mt_pointwise_small(d)) return mt_pw_bwd_data(d, dy, pack_bwd, dx, (long)d->N * d->H * d->W, s);
  if (d->transposed)
    return gather_form(d, dy, Ho, Wo, Cop, pack_bwd, nullptr, 0, dx, d->H, d->W, Cip, MT_PAD_ZERO, MT_ACT_NONE, s);
  const int P = (d->pad_mode == MT_PAD_REFLECT) ? d->pad : 0;
  const size_t padded_b = bwd_data_padded_bytes(d), split_b = bwd_data_split_bytes(d);
  const bool have_ws = ws != nullptr && ws_bytes >= padded_b + split_b;
  void* split_ws = (have_ws && split_b) ? (char*)ws + padded_b : nullptr;
  if (mt_stem_dgrad_ok(d)) {                 // the 7x7 stem: direct kernel (stem_kernel.hip)
    if (P == 0) return mt_launch_stem_dgrad(d, dy, pack_bwd, dx, s) ? 2 : 0;
    MT_CHECK(ws != nullptr && ws_bytes >= padded_b, "conv_bwd_data: workspace too small");
    if (mt_launch_stem_dgrad(d, dy, pack_bwd, ws, s)) return 2;
    return mt_launch_reflect_fold(d->dtype, ws, dx, d->N, d->H, d->W, Cip, P, s);
  }
  if (P == 0)
    return scatter_form(d, dy, Ho, Wo, Cop, pack_bwd, nullptr, 0, dx, d->H, d->W, Cip, d->pad, MT_ACT_NONE, s, split_ws,
                        split_b);
  MT_CHECK(ws != nullptr && ws_bytes >= padded_b, "conv_bwd_data: workspace too small");
  RingTaps rt;
  if (ring_taps(d, &rt)) {
    const int sz = esz(d->dtype), V = vec(d->dtype);
    IgemmParams p;
    memset(&p, 0, sizeof(p));
    p.x = (const char*)dy; p.w = (const char*)pack_bwd; p.N = d->N; p.Hi = Ho; p.Wi = Wo; p.Cib = Cop * sz;
    p.Co = Cip; p.CoRows = Cip; p.os = 1; p.is = 1; p.cpc = Cop / V;
    p.pad_mode = MT_PAD_ZERO; p.act = MT_ACT_NONE; p.slope = d->slope;
    const size_t image = (size_t)Cip * rt.total * Cop * sz;
    auto phase = [&](int run, int oh0, int ow0, int Hs, int Ws) {
      if (rt.count[run] == 0 || Hs <= 0 || Ws <= 0) return;
      IgemmPhase& q = p.ph[p.nphase];
      int tap0 = 0;
      for (int i = 0; i < p.nphase; i++) tap0 += p.ph[i].ntaps;
      q.ntaps = rt.count[run]; q.tap0 = tap0;
      q.w_off = (unsigned)((size_t)rt.start[run] * Cop * sz); q.w_bytes = (unsigned)(image - q.w_off);
      q.wrow = rt.total * p.cpc;
      q.Ho = Hs; q.Wo = Ws; q.M = d->N * Hs * Ws; q.oh0 = oh0; q.ow0 = ow0;
      // padded-grid pixel (ho + oh0, wo + ow0) reads dy[(ho + oh0) - a][(wo + ow0) - b]
      for (int t = 0; t < q.ntaps; t++) {
        p.dh[tap0 + t] = (short)(oh0 - rt.kh[rt.start[run] + t]);
        p.dw[tap0 + t] = (short)(ow0 - rt.kw[rt.start[run] + t]);
      }
      p.nphase++;
    };
    // (1) interior: output pixel (h, w) is padded pixel (h + P, w + P); reuse `phase` with the offset folded
    //     into the tap table and no output offset
    p.y = (char*)dx; p.Hout = d->H; p.Wout = d->W;
    phase(0, 0, 0, d->H, d->W);
    for (int t = 0; t < p.ph[0].ntaps; t++) { p.dh[t] = (short)(p.dh[t] + P); p.dw[t] = (short)(p.dw[t] + P); }
    // 3x3, pad 1 on the patch-resident 256x256 kernel: the reflected ring is added inside the pixel operand -- one
    // launch, no ring GEMM, no workspace, no fold (conv_pipe_patch_kernel.hip)
    static const int fold_on = getenv("MT_IGEMM_FOLD") ? atoi(getenv("MT_IGEMM_FOLD")) : 1;
    if (fold_on && P == 1 && d->kh == 3 && d->kw == 3) {
      p.fold = 1;
      if (mt_igemm_fold_ok(d->dtype, p)) {
        p.addend = (const char*)addend;
        *added = addend != nullptr;
        if (bs != nullptr && bs->x != nullptr && bs->sums != nullptr) {
          p.stats = bs->sums; p.bstat_x = (const char*)bs->x; p.bstat_scale = bs->scale; p.bstat_shift = bs->shift;
          p.bstat_act = bs->act; p.bstat_slope = bs->slope;
          if (stats_done) *stats_done = true;
        }
        return mt_launch_igemm(d->dtype, p, s) ? 2 : 0;
      }
      p.fold = 0;
    }
    if (mt_launch_igemm(d->dtype, p, s)) return 2;
    // (2) the four strips of the ring, into the padded workspace
    p.nphase = 0;
    p.y = (char*)ws; p.Hout = d->H + 2 * P; p.Wout = d->W + 2 * P;
    phase(1, 0, 0, P, d->W + 2 * P);
    phase(2, d->H + P, 0, P, d->W + 2 * P);
    phase(3, P, 0, d->H, P);
    phase(4, P, d->W + P, d->H, P);
    if (p.nphase > 0 && mt_launch_igemm(d->dtype, p, s)) return 2;
    // (3) fold the ring onto the border band
    return mt_launch_ring_fold(d->dtype, ws, dx, d->N, d->H, d->W, Cip, P, s);
  }
  // the padded gradient map: whole into the workspace + full fold, or -- on the persistent gather-GEMM -- its interior
  // straight into dx, only the P-wide ring into the workspace, and the band-only fold (saves a read and a write of dx)
  const int rc = scatter_form(d, dy, Ho, Wo, Cop, pack_bwd, nullptr, 0, ws, d->H + 2 * P, d->W + 2 * P, Cip, 0, MT_ACT_NONE, s,
                              split_ws, split_b, dx, P);
  if (rc == 100) return mt_launch_ring_fold(d->dtype, ws, dx, d->N, d->H, d->W, Cip, P, s);
  if (rc) return 2;
  return mt_launch_reflect_fold(d->dtype, ws, dx, d->N, d->H, d->W, Cip, P, s);
}

extern "C" int mt_conv_bwd_data(const mt_conv_desc* d, const void* dy, const void* pack_bwd, void* dx, void* ws,
                                size_t ws_bytes, mt_stream_t st) {
  bool added;
  return conv_bwd_data_impl(d, dy, pack_bwd, dx, nullptr, &added, ws, ws_bytes, st);
}
// dx = data gradient + addend (a tensor of dx's shape and type, e.g. the skip-connection gradient of a residual block whose
// first convolution this is): inside the GEMM epilogue where the kernel supports it (the patch-resident 256x256 kernel: one
// extra read of the addend instead of a separate read-read-write pass), as a separate in-place add otherwise.
extern "C" int mt_conv_bwd_data_add(const mt_conv_desc* d, const void* dy, const void* pack_bwd, void* dx, const void* addend,
                                    void* ws, size_t ws_bytes, mt_stream_t st) {
  return mt_conv_bwd_data_ex(d, dy, pack_bwd, dx, addend, nullptr, nullptr, ws, ws_bytes, st);
}
// ... and, where the kernel supports it, the statistics of the normalisation backward that consumes dx (mt_bwd_stats):
// *stats_done = 1 when bs->sums was filled (else the caller runs mt_nc_stats_bwd as usual).  addend may be NULL.
extern "C" int mt_conv_bwd_data_ex(const mt_conv_desc* d, const void* dy, const void* pack_bwd, void* dx, const void* addend,
                                   const mt_bwd_stats* bs, int* stats_done, void* ws, size_t ws_bytes, mt_stream_t st) {
  bool added, sdone;
  if (stats_done) *stats_done = 0;
  const int rc = conv_bwd_data_impl(d, dy, pack_bwd, dx, addend, &added, ws, ws_bytes, st, bs, &sdone);
  if (rc) return rc;
  if (stats_done) *stats_done = sdone ? 1 : 0;
  if (addend == nullptr || added) return 0;
  const size_t n = (size_t)d->N * d->H * d->W * mt_padc(d->Ci);
  return mt_add(d->dtype, dx, addend, dx, n, st);
}

// pixel-split of the weight-gradient reduction: enough (tile, split) blocks to fill 256 CUs x 2
// *pipe: the 256x256 ping-pong kernel takes the problem (then the split fills one 8-wave block per CU)
static void wgrad_params(const mt_conv_desc* d, const void* x, const void* dy, WgradParams* pp);
static int wgrad_pixels(const mt_conv_desc* d);
// *pipe: 1 the 256x256 ping-pong kernel takes the problem (then the split fills one 8-wave block per CU), 2 the
// accumulator-stationary kernel of the 3x3 layers does (*mchunk = output ROWS per split then), 0 the 128x128 tile kernel
static void wgrad_split(const mt_conv_desc* d, int M, int* nsplit, int* mchunk, int* pipe = nullptr) {
  const int V = vec(d->dtype);
  const int Cip = mt_padc(d->Ci), Cop = mt_padc(d->Co), K2 = d->kh * d->kw;
  const int rows = d->transposed ? Cip : Cop, cols = (d->transposed ? Cop : Cip) * K2;
  if (pipe) *pipe = 0;
  bool pipe_shape = false;
  {
    int Ho, Wo;
    mt_conv_out_hw(d, &Ho, &Wo);
    const long ab = (long)M * rows * esz(d->dtype);
    const long bb = d->transposed ? (long)d->N * Ho * Wo * Cop * esz(d->dtype) : (long)d->N * d->H * d->W * Cip * esz(d->dtype);
    const int cb = d->transposed ? Cop : Cip;
    if (mt_wgrad_pipe_ok(d->dtype, rows, cb / V, ab, bb)) {
      pipe_shape = true;
      const int tiles = (rows / 256) * (cols / 256);
      // one round of 256 blocks; two or four rounds when a split would not fit the per-block pixel-offset table
      for (int rounds = 1; rounds <= 4; rounds *= 2) {
        int ns = rounds * 256 / tiles;
        if (ns < 1 && tiles <= 1024 && M <= mt_wgrad_pipe_max_chunk()) ns = 1;     // few pixels, many tiles: whole rounds of tiles
        if (ns < 1 || M / ns < 512) break;
        const int mc = cdiv(cdiv(M, ns), 32) * 32;
        if (mc <= mt_wgrad_pipe_max_chunk()) {
          *mchunk = mc;
          *nsplit = cdiv(M, mc);
          if (pipe) *pipe = 1;
          return;
        }
      }
    }
  }
  if (!pipe_shape && d->kh == 3 && d->kw == 3 && d->pad == 1 && M == wgrad_pixels(d)) {
    WgradParams p;
    wgrad_params(d, nullptr, nullptr, &p);
    int ns = 0, rs = 0;
    if (mt_wgrad_rows_plan(d->dtype, p, &ns, &rs)) {
      *nsplit = ns;
      *mchunk = rs;
      if (pipe) *pipe = 2;
      return;
    }
  }
  const int tiles = cdiv(rows, 128) * cdiv(cols, 128);
  // 256 CUs x 2 resident blocks = 512 slots: fill exactly one round (a 1.3-round grid costs two rounds)
  int ns = 512 / tiles;
  const int maxsplit = cdiv(M, 256);
  if (ns > maxsplit) ns = maxsplit;
  if (ns < 1) ns = 1;
  *mchunk = cdiv(cdiv(M, ns), 64) * 64;
  *nsplit = cdiv(M, *mchunk);
}
static int wgrad_pixels(const mt_conv_desc* d) {
  int Ho, Wo;
  mt_conv_out_hw(d, &Ho, &Wo);
  return d->transposed ? d->N * d->H * d->W : d->N * Ho * Wo;
}

extern "C" size_t mt_conv_bwd_weight_ws_bytes(const mt_conv_desc* d) {
  if (mt_stem_wgrad_ok(d)) {                 // direct 7x7 stem kernel: one slab per workgroup
    const size_t a = mt_stem_wgrad_ws_bytes(d), b = mt_colsum_ws_bytes(mt_padc(d->Co));
    return a > b ? a : b;
  }
  int ns, mc;
  wgrad_split(d, wgrad_pixels(d), &ns, &mc);
  // split slabs of the weight gradient; the bias-gradient partials use the same bytes BEFORE them (stream order)
  const size_t slabs = (size_t)ns * mt_padc(d->Ci) * mt_padc(d->Co) * d->kh * d->kw * sizeof(float);
  const size_t bias = mt_colsum_ws_bytes(mt_padc(d->Co));
  return slabs > bias ? slabs : bias;
}

// geometry of the weight-gradient GEMM of one convolution (everything but the pixel split and the slab pointer)
static void wgrad_params(const mt_conv_desc* d, const void* x, const void* dy, WgradParams* pp) {
  WgradParams& p = *pp;
  memset(&p, 0, sizeof(p));
  int Ho, Wo;
  mt_conv_out_hw(d, &Ho, &Wo);
  const int sz = esz(d->dtype), V = vec(d->dtype);
  const int Cip = mt_padc(d->Ci), Cop = mt_padc(d->Co), K2 = d->kh * d->kw;
  p.N = d->N; p.is = d->stride; p.ntaps = K2;
  for (int a = 0; a < d->kh; a++)
    for (int b = 0; b < d->kw; b++) {
      const int t = a * d->kw + b;
      p.dh[t] = (short)(a - d->pad); p.dw[t] = (short)(b - d->pad);
    }
  if (!d->transposed) {
    // dW[co][kh][kw][ci] = sum dy[n,ho,wo,co] * x[n, pad(ho*s-p+kh), pad(wo*s-p+kw), ci]
    p.a = (const char*)dy; p.Cab = Cop * sz; p.CaRows = Cop; p.Ho = Ho; p.Wo = Wo;
    p.b = (const char*)x; p.Hi = d->H; p.Wi = d->W; p.Cbb = Cip * sz; p.cpc = Cip / V;
    p.pad_mode = d->pad_mode;
  } else {
    // dW[ci][co][kh][kw] = sum x[n,hi,wi,ci] * dy[n, hi*s-p+kh, wi*s-p+kw, co]  (zero outside)
    p.a = (const char*)x; p.Cab = Cip * sz; p.CaRows = Cip; p.Ho = d->H; p.Wo = d->W;
    p.b = (const char*)dy; p.Hi = Ho; p.Wi = Wo; p.Cbb = Cop * sz; p.cpc = Cop / V;
    p.pad_mode = MT_PAD_ZERO;
  }
  p.M = d->N * p.Ho * p.Wo;
  p.nchunks = p.ntaps * p.cpc;
}

// The weight gradient in two halves, so that a caller may run the second one (a pure streaming reduction of the
// split slabs) on another stream beside the next layer's GEMMs:
//   partial: [bias gradient, if asked for] then the split GEMM -> fp32 slabs in ws; *nslabs = how many
//   finish:  dw (+)= sum of the slabs, reference layout
// mt_conv_bwd_weight = partial + finish on one stream.
static int bwd_weight_unpack_params(const mt_conv_desc* d, PackParams* u) {
  memset(u, 0, sizeof(*u));
  const int Cip = mt_padc(d->Ci), Cop = mt_padc(d->Co), K2 = d->kh * d->kw;
  if (mt_pointwise_small(d)) {
    u->kW = 1; u->ntaps = 1; u->R = d->Co; u->C = d->Ci; u->Cp = Cip;
    if (d->transposed) { u->sr = 1; u->sc = d->Co; } else { u->sr = d->Ci; u->sc = 1; }
    return 0;
  }
  u->kW = d->kw; u->ntaps = K2;
  for (int a = 0; a < d->kh; a++)
    for (int b = 0; b < d->kw; b++) { u->kh[a * d->kw + b] = (short)a; u->kw[a * d->kw + b] = (short)b; }
  if (!d->transposed) { u->R = d->Co; u->C = d->Ci; u->Cp = Cip; u->sr = (long)d->Ci * K2; u->sc = K2; }
  else { u->R = d->Ci; u->C = d->Co; u->Cp = Cop; u->sr = (long)d->Co * K2; u->sc = K2; }
  return 0;
}

extern "C" int mt_conv_bwd_weight_partial(const mt_conv_desc* d, const void* x, const void* dy, float* dbias, void* ws,
                                          size_t ws_bytes, int accumulate, int want_dw, int* nslabs, mt_stream_t st) {
  if (check_desc(d)) return 1;
  hipStream_t s = (hipStream_t)st;
  MT_CHECK(ws != nullptr && ws_bytes >= mt_conv_bwd_weight_ws_bytes(d), "conv_bwd_weight: workspace too small");
  MT_CHECK(nslabs != nullptr, "conv_bwd_weight_partial: nslabs is NULL");
  *nslabs = 0;
  int Ho, Wo;
  mt_conv_out_hw(d, &Ho, &Wo);
  const int Cip = mt_padc(d->Ci), Cop = mt_padc(d->Co);
  // the bias gradient first: its partials use the workspace before the slabs do (stream order)
  if (dbias != nullptr) {
    const long npix = mt_pointwise_small(d) ? (long)d->N * d->H * d->W : (long)d->N * Ho * Wo;
    if (mt_launch_colsum(d->dtype, dy, dbias, npix, Cop, d->Co, accumulate, ws, ws_bytes, s)) return 2;
  }
  if (!want_dw) return 0;
  if (mt_stem_wgrad_ok(d)) return mt_launch_stem_wgrad(d, x, dy, ws, nslabs, s) ? 2 : 0;
  if (mt_pointwise_small(d)) {
    // streaming outer-product reduction: one fp32 slab [Cop][Cip] per block (no atomics, nothing to zero); the
    // unpack adds the slabs in index order
    const long slab = (long)Cop * Cip;
    const int max_slabs = (int)min((size_t)1024, ws_bytes / (slab * sizeof(float)));
    if (mt_pw_bwd_weight(d, x, dy, (float*)ws, (long)d->N * d->H * d->W, max_slabs, nslabs, s)) {
      mt_set_error("conv_bwd_weight: thin 1x1 weight gradient could not be launched");
      return 2;
    }
    return 0;
  }
  WgradParams p;
  wgrad_params(d, x, dy, &p);
  p.out = (float*)ws;
  int nsplit;
  int pipe;
  wgrad_split(d, p.M, &nsplit, &p.mchunk, &pipe);
  p.ntiles = -pipe;               // tells mt_launch_wgrad which kernel the split was made for (-1 ping-pong, -2 row walker)
  if (pipe == 2) { p.rows_rs = p.mchunk; p.mchunk = 0; }
  if (mt_launch_wgrad(d->dtype, p, nsplit, s)) return 2;
  *nslabs = nsplit;
  return 0;
}

extern "C" int mt_conv_bwd_weight_finish(const mt_conv_desc* d, const void* ws, int nslabs, float* dw, int accumulate,
                                         mt_stream_t st) {
  if (check_desc(d)) return 1;
  MT_CHECK(ws != nullptr && dw != nullptr && nslabs > 0, "conv_bwd_weight_finish: nothing to reduce");
  if (mt_stem_wgrad_ok(d)) return mt_launch_stem_wgrad_reduce(d, ws, nslabs, dw, accumulate, (hipStream_t)st) ? 2 : 0;
  PackParams u;
  bwd_weight_unpack_params(d, &u);
  const int Cip = mt_padc(d->Ci), Cop = mt_padc(d->Co), K2 = d->kh * d->kw;
  const long slab = mt_pointwise_small(d) ? (long)Cop * Cip : (long)Cop * Cip * K2;
  return mt_launch_unpack((const float*)ws, dw, u, nslabs, slab, accumulate, (hipStream_t)st) ? 2 : 0;
}

// Several mt_conv_bwd_weight_finish calls of one backward pass in as few launches as possible (round 4): the slab sums whose
// unpack has the natural tap order ride in batched launches of up to 64 entries, the rest (7x7 stem, thin 1x1) keep their own.
extern "C" int mt_conv_bwd_weight_finish_multi(int n, const mt_conv_desc* descs, const void* const* ws, const int* nslabs,
                                               float* const* dw, int accumulate, mt_stream_t st) {
  MT_CHECK(n >= 0 && (n == 0 || (descs && ws && nslabs && dw)), "conv_bwd_weight_finish_multi: null argument");
  hipStream_t s = (hipStream_t)st;
  const float* src[MT_UNPACK_MULTI_MAX];
  float* dst[MT_UNPACK_MULTI_MAX];
  PackParams ps[MT_UNPACK_MULTI_MAX];
  int ns[MT_UNPACK_MULTI_MAX];
  long slabs[MT_UNPACK_MULTI_MAX];
  int k = 0;
  for (int i = 0; i < n; i++) {
    const mt_conv_desc* d = &descs[i];
    if (check_desc(d)) return 1;
    MT_CHECK(ws[i] != nullptr && dw[i] != nullptr && nslabs[i] > 0, "conv_bwd_weight_finish_multi: entry %d has nothing to reduce", i);
    PackParams u;
    if (!mt_stem_wgrad_ok(d) && !mt_pointwise_small(d)) {
      bwd_weight_unpack_params(d, &u);
      if (mt_unpack_multi_ok(u)) {
        src[k] = (const float*)ws[i]; dst[k] = dw[i]; ps[k] = u; ns[k] = nslabs[i];
        slabs[k] = (long)mt_padc(d->Co) * mt_padc(d->Ci) * d->kh * d->kw;
        if (++k == MT_UNPACK_MULTI_MAX) {
          if (mt_launch_unpack_multi(k, src, dst, ps, ns, slabs, accumulate, s)) return 2;
          k = 0;
        }
        continue;
      }
    }
    const int rc = mt_conv_bwd_weight_finish(d, ws[i], nslabs[i], dw[i], accumulate, st);
    if (rc) return rc;
  }
  if (k > 0 && mt_launch_unpack_multi(k, src, dst, ps, ns, slabs, accumulate, s)) return 2;
  return 0;
}

// bytes of ONE slab of mt_conv_bwd_weight_partial when its slabs have the generic [rows][taps][channels] form, which depends on
// the weight's shape only -- so the slabs of several uses of one weight (different N, H, W) may sit behind each other in one
// workspace and be summed by ONE mt_conv_bwd_weight_finish; 0 for the layers with their own slab forms (7x7 stem, thin 1x1)
bool mt_thin_wgrad_ok(const mt_conv_desc* d);                       // conv_aux_kernels.hip
int mt_launch_thin_wgrad(const mt_conv_desc* d, const void* x, const void* dy, float* dw, int accumulate, hipStream_t s);
extern "C" size_t mt_conv_bwd_weight_slab_bytes(const mt_conv_desc* d) {
  if (check_desc(d) || mt_stem_wgrad_ok(d) || mt_pointwise_small(d) || mt_thin_wgrad_ok(d)) return 0;
  return (size_t)mt_padc(d->Ci) * mt_padc(d->Co) * d->kh * d->kw * sizeof(float);
}

extern "C" int mt_conv_bwd_weight(const mt_conv_desc* d, const void* x, const void* dy, float* dw, float* dbias,
                                  void* ws, size_t ws_bytes, int accumulate, mt_stream_t st) {
  if (check_desc(d)) return 1;
  if (dw != nullptr && mt_thin_wgrad_ok(d) && !mt_pointwise_small(d) && !mt_stem_wgrad_ok(d)) {
    // thin 1x1 head: streaming kernel straight into dw (the bias gradient, if asked for, keeps its own pass)
    if (dbias != nullptr) {
      MT_CHECK(ws != nullptr && ws_bytes >= mt_colsum_ws_bytes(mt_padc(d->Co)), "conv_bwd_weight: workspace too small");
      if (mt_launch_colsum(d->dtype, dy, dbias, (long)d->N * d->H * d->W, mt_padc(d->Co), d->Co, accumulate, ws, ws_bytes,
                           (hipStream_t)st)) return 2;
    }
    return mt_launch_thin_wgrad(d, x, dy, dw, accumulate, (hipStream_t)st) ? 2 : 0;
  }
  int nslabs = 0;
  const int rc = mt_conv_bwd_weight_partial(d, x, dy, dbias, ws, ws_bytes, accumulate, dw != nullptr, &nslabs, st);
  if (rc) return rc;
  return dw != nullptr ? mt_conv_bwd_weight_finish(d, ws, nslabs, dw, accumulate, st) : 0;
}

// ---- weight gradients of several 3x3 layers in shared launches of the row walker (round 4; wgrad_rows_kernel.hip) ------------
// A layer of the encoders / the decoder alone cannot fill 256 compute units without cutting its pixel reduction ~256 ways: 38 MB
// of fp32 slabs written and read back for a 150-600 KB gradient, and a prologue / epilogue per 30-60 k-steps.  Weight gradients
// are leaves of the backward pass, so the caller parks them and hands ALL of a pass's eligible layers over at once: one launch
// per stride class walks every problem with a share of the chip in proportion to its work (~25 slabs per layer for ten layers),
// and one batched slab sum adds them into the gradients.  Entries that accumulate into the SAME dw must be adjacent: their
// slabs are summed by one entry of the batched sum.
#define MT_ROWS_MULTI_MAX 64
extern "C" int mt_conv_bwd_weight_rows_ok(const mt_conv_desc* d) {
  if (d == nullptr || d->kh != 3 || d->kw != 3 || d->pad != 1 || d->stride < 1 || d->stride > 2) return 0;
  if (mt_stem_wgrad_ok(d) || mt_pointwise_small(d)) return 0;
  WgradParams p;
  wgrad_params(d, nullptr, nullptr, &p);
  // the 256-multiples keep the ping-pong kernel (each operand byte is read once per 256-wide tile there, four times here) -- except
  // on small maps, where its splits are too short to pay for their slabs (3x3 256 -> 256 on 32 x 32: 63 us alone, 0.12 of peak)
  if (mt_wgrad_pipe_ok(d->dtype, p.CaRows, p.cpc, (long)p.M * p.Cab, (long)p.N * p.Hi * p.Wi * p.Cbb) && p.Ho * p.Wo > 1024) return 0;
  return mt_wgrad_rows_ok(d->dtype, p) ? 1 : 0;
}
static int rows_multi_plan(int n, const mt_conv_desc* descs, WgradParams* ps, int* nsplit, int* rps, size_t* offs, size_t* total) {
  MT_CHECK(n >= 1 && n <= MT_ROWS_MULTI_MAX && descs != nullptr, "conv_bwd_weight_rows_multi: %d problems (1..%d)", n, MT_ROWS_MULTI_MAX);
  for (int i = 0; i < n; i++) {
    if (check_desc(&descs[i])) return 1;
    MT_CHECK(mt_conv_bwd_weight_rows_ok(&descs[i]), "conv_bwd_weight_rows_multi: problem %d is not a row-walker shape", i);
    wgrad_params(&descs[i], nullptr, nullptr, &ps[i]);
  }
  mt_wgrad_rows_plan_multi(n, ps, nsplit, rps);
  size_t off = 0;
  for (int i = 0; i < n; i++) {
    offs[i] = off;
    off += (size_t)nsplit[i] * mt_padc(descs[i].Ci) * mt_padc(descs[i].Co) * 9 * sizeof(float);
  }
  *total = off;
  return 0;
}
extern "C" size_t mt_conv_bwd_weight_rows_multi_ws_bytes(int n, const mt_conv_desc* descs) {
  WgradParams ps[MT_ROWS_MULTI_MAX];
  int nsplit[MT_ROWS_MULTI_MAX], rps[MT_ROWS_MULTI_MAX];
  size_t offs[MT_ROWS_MULTI_MAX], total = 0;
  if (rows_multi_plan(n, descs, ps, nsplit, rps, offs, &total)) return 0;
  return total;
}
extern "C" int mt_conv_bwd_weight_rows_multi(int n, const mt_conv_desc* descs, const void* const* x, const void* const* dy,
                                             float* const* dw, void* ws, size_t ws_bytes, int accumulate, mt_stream_t st) {
  WgradParams ps[MT_ROWS_MULTI_MAX];
  int nsplit[MT_ROWS_MULTI_MAX], rps[MT_ROWS_MULTI_MAX];
  size_t offs[MT_ROWS_MULTI_MAX], total = 0;
  if (rows_multi_plan(n, descs, ps, nsplit, rps, offs, &total)) return 1;
  MT_CHECK(x && dy && dw && ws != nullptr && ws_bytes >= total, "conv_bwd_weight_rows_multi: workspace too small / null argument");
  hipStream_t s = (hipStream_t)st;
  for (int i = 0; i < n; i++) {
    MT_CHECK(x[i] && dy[i] && dw[i], "conv_bwd_weight_rows_multi: problem %d has a null operand", i);
    WgradParams g;
    wgrad_params(&descs[i], x[i], dy[i], &g);
    ps[i].a = g.a; ps[i].b = g.b;
    ps[i].out = (float*)((char*)ws + offs[i]);
  }
  if (mt_launch_wgrad_rows_multi(n, ps, nsplit, rps, s)) return 2;
  // one slab sum per gradient tensor: adjacent problems of one dw (uses of one weight: same slab form) are one entry
  const float* src[MT_UNPACK_MULTI_MAX];
  float* dst[MT_UNPACK_MULTI_MAX];
  PackParams us[MT_UNPACK_MULTI_MAX];
  int ns[MT_UNPACK_MULTI_MAX];
  long slabs[MT_UNPACK_MULTI_MAX];
  int k = 0;
  for (int i = 0; i < n;) {
    int j = i, tot = 0;
    const long slab = (long)mt_padc(descs[i].Ci) * mt_padc(descs[i].Co) * 9;
    for (; j < n && dw[j] == dw[i]; j++) {
      MT_CHECK((long)mt_padc(descs[j].Ci) * mt_padc(descs[j].Co) * 9 == slab && descs[j].transposed == descs[i].transposed &&
               descs[j].Ci == descs[i].Ci && descs[j].Co == descs[i].Co,
               "conv_bwd_weight_rows_multi: problems %d and %d share a gradient but not a weight shape", i, j);
      tot += nsplit[j];
    }
    PackParams u;
    bwd_weight_unpack_params(&descs[i], &u);
    if (!mt_unpack_multi_ok(u)) {
      if (mt_launch_unpack((const float*)((char*)ws + offs[i]), dw[i], u, tot, slab, accumulate, s)) return 2;
    } else {
      src[k] = (const float*)((char*)ws + offs[i]); dst[k] = dw[i]; us[k] = u; ns[k] = tot; slabs[k] = slab;
      if (++k == MT_UNPACK_MULTI_MAX) {
        if (mt_launch_unpack_multi(k, src, dst, us, ns, slabs, accumulate, s)) return 2;
        k = 0;
      }
    }
    i = j;
  }
  if (k > 0 && mt_launch_unpack_multi(k, src, dst, us, ns, slabs, accumulate, s)) return 2;
  return 0;
}

// ---- grouped weight gradient (round 3) ----------------------------------------------------------------------------------
// The 256x256 ping-pong weight gradient splits the pixel reduction so that (tiles x splits) fills the 256 CUs: the dominant
// layer (9 tiles) runs 28 splits, i.e. it writes 28 fp32 slabs of the whole gradient (66 MB) and the slab sum reads them
// back -- 36 of its 93 us at N = 16.  Weight gradients are leaves of the backward pass (nothing waits for them before the
// optimizer step), so G problems OF THE SAME GEOMETRY can share one launch: G x 9 tiles x 28 / G splits, a quarter of the slab
// bytes per problem at G = 4.  The longer splits need the 16-bit pixel-delta table of wgrad_pipe_kernel<COMPACT>.
static bool wgrad_group_split(const mt_conv_desc* d, int G, int* nsplit, int* mchunk) {
  if (G < 2 || G > MT_WGRAD_MAX_GROUP || mt_stem_wgrad_ok(d) || mt_pointwise_small(d)) return false;
  WgradParams p;
  wgrad_params(d, nullptr, nullptr, &p);
  const int V = vec(d->dtype), sz = esz(d->dtype);
  const long ab = (long)p.M * p.Cab, bb = (long)p.N * p.Hi * p.Wi * p.Cbb;
  if (!mt_wgrad_pipe_ok(d->dtype, p.CaRows, p.cpc, ab, bb) || !mt_wgrad_pipe_compact_ok(p)) return false;
  (void)sz;
  const int tiles = (p.CaRows / 256) * (p.nchunks * V / 256);
  const int ns = 256 / (tiles * G);
  if (ns < 1 || p.M / ns < 512 || ns * tiles * G < 200) return false;
  const int mc = cdiv(cdiv(p.M, ns), 32) * 32;
  const int mcmax = mt_wgrad_pipe_compact8_ok(p) ? mt_wgrad_pipe_max_chunk_compact8() : mt_wgrad_pipe_max_chunk_compact();
  if (mc > mcmax) return false;
  *mchunk = mc;
  *nsplit = cdiv(p.M, mc);
  return true;
}
extern "C" int mt_conv_bwd_weight_group_max(const mt_conv_desc* d) {
  if (check_desc(d)) return 1;
  // the largest group that still fills the chip (G = 7 of the dominant layer: 63 tiles x 4 splits = 252 blocks; G = 8 would
  // leave 216), else the largest that fits at all
  int ns, mc, best = 1;
  for (int G = MT_WGRAD_MAX_GROUP; G >= 2; G--) {
    if (!wgrad_group_split(d, G, &ns, &mc)) continue;
    if (best == 1) best = G;
    WgradParams p;
    wgrad_params(d, nullptr, nullptr, &p);
    const int tiles = (p.CaRows / 256) * (p.nchunks * vec(d->dtype) / 256);
    if (ns * tiles * G >= 240) return G;
  }
  return best;
}
extern "C" size_t mt_conv_bwd_weight_group_ws_bytes(const mt_conv_desc* d, int G) {
  int ns, mc;
  if (check_desc(d) || !wgrad_group_split(d, G, &ns, &mc)) return 0;
  return (size_t)G * ns * mt_padc(d->Ci) * mt_padc(d->Co) * d->kh * d->kw * sizeof(float);
}
extern "C" int mt_conv_bwd_weight_group(const mt_conv_desc* d, int G, const void* const* x, const void* const* dy,
                                        float* const* dw, void* ws, size_t ws_bytes, int accumulate, mt_stream_t st) {
  if (check_desc(d)) return 1;
  hipStream_t s = (hipStream_t)st;
  int nsplit, mchunk;
  MT_CHECK(wgrad_group_split(d, G, &nsplit, &mchunk), "conv_bwd_weight_group: %d problems of this geometry cannot share a launch", G);
  MT_CHECK(x != nullptr && dy != nullptr && dw != nullptr, "conv_bwd_weight_group: NULL pointer table");
  const size_t need = mt_conv_bwd_weight_group_ws_bytes(d, G);
  MT_CHECK(ws != nullptr && ws_bytes >= need, "conv_bwd_weight_group: workspace too small");
  WgradParams p;
  wgrad_params(d, x[0], dy[0], &p);
  const long slab = (long)mt_padc(d->Ci) * mt_padc(d->Co) * d->kh * d->kw;
  p.ngroup = G;
  for (int g = 0; g < G; g++) {
    MT_CHECK(x[g] != nullptr && dy[g] != nullptr && dw[g] != nullptr, "conv_bwd_weight_group: NULL operand %d", g);
    p.ga[g] = (const char*)(d->transposed ? x[g] : dy[g]);
    p.gb[g] = (const char*)(d->transposed ? dy[g] : x[g]);
    p.gout[g] = (float*)ws + (size_t)g * nsplit * slab;
  }
  p.out = p.gout[0];
  p.mchunk = mchunk;
  p.ntiles = -1;
  if (mt_launch_wgrad(d->dtype, p, nsplit, s)) return 2;
  PackParams u;
  bwd_weight_unpack_params(d, &u);
  for (int g = 0; g < G; g++)
    if (mt_launch_unpack(p.gout[g], dw[g], u, nsplit, slab, accumulate, s)) return 2;
  return 0;
}
