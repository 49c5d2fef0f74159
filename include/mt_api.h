/*
 * mt_api.h -- C ABI of libmt_hip.so, the MI355X (gfx950) native op library that sits
 * under the GAN+VAE training step of AdaINModel / BaseModel.
 *
 * The reference (kartikkadur/MasterThesis) has no FFI of its own: every arithmetic op on
 * its hot path is an implicit torch.nn call.  Each entry
 * point below replaces one of those call sites; the reference site is cited per group.
 *
 * Conventions
 *  - Activations are NHWC, channels padded to a multiple of 8 ("Cp"); pad channels hold
 *    zeros and every kernel preserves that.  dtype is MT_F32 or MT_BF16 (storage type of
 *    activations; all accumulation, statistics and losses are fp32).
 *  - Parameters stay in the reference's layouts (OIHW fp32 for Conv2d, IOHW for
 *    ConvTranspose2d, [out,in] for Linear) so checkpoints interchange; the library packs
 *    them into MFMA-friendly tiles with mt_conv_pack().
 *  - Ownership: the caller (PyTorch) allocates every buffer and workspace; the library
 *    borrows pointers for the duration of a call and never allocates or frees.
 *  - Every call takes the HIP stream explicitly; host side re-entrant, no host-side global state apart from the
 *    lazily loaded RCCL entry points of mt_comm_* and one small table: the scalar loss reductions (mt_bce_*_fwd,
 *    mt_gan_const_fwd, mt_l1_fwd, mt_l2mean_fwd, mt_kl_fwd) combine their block partials through a module-level device
 *    scratch (fixed-order, reproducible sums without float atomics) that has ONE SLOT PER STREAM (the first 16 distinct
 *    streams of a process get their own, assigned on first use; later ones share by address): calls on one stream are
 *    ordered by the stream, calls on different streams use different slots, so any stream may be passed.
 *  - Return value: 0 on success, non-zero on error; mt_last_error() gives a thread-local
 *    message.  No C++ exception crosses the ABI.
 */
#ifndef MT_API_H
#define MT_API_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef void* mt_stream_t; /* hipStream_t */

enum { MT_F32 = 0, MT_BF16 = 1 };
enum { MT_PAD_ZERO = 0, MT_PAD_REFLECT = 1 };
enum { MT_ACT_NONE = 0, MT_ACT_RELU = 1, MT_ACT_LRELU = 2, MT_ACT_TANH = 3 };
/* which packed copy of a weight tensor */
enum { MT_PACK_FWD = 0, MT_PACK_BWD_DATA = 1 };
enum { MT_NORM_INSTANCE = 0, MT_NORM_ADAIN = 1, MT_NORM_LAYER = 2, MT_NORM_BATCH = 3 };

/* Convolution problem: nn.Conv2d (blocks.py:33-35) preceded by nn.ReflectionPad2d
 * (functions.py:51) when pad_mode == MT_PAD_REFLECT, or nn.ConvTranspose2d (blocks.py:73)
 * when transposed != 0.  H, W, Ci describe the op's INPUT, Co its output (logical channel
 * counts; buffers use the padded counts). */
typedef struct mt_conv_desc {
  int dtype;
  int transposed; /* 0: Conv2d, 1: ConvTranspose2d */
  int N, H, W;
  int Ci, Co;
  int kh, kw;
  int stride;
  int pad;
  int pad_mode;   /* Conv2d only */
  int out_pad;    /* ConvTranspose2d only */
  int act;        /* fused epilogue activation of the forward */
  float slope;    /* LeakyReLU slope */
} mt_conv_desc;

const char* mt_last_error(void);
int mt_version(void);
/* launches routed so far to a kernel variant the dispatcher chooses by problem size (tests assert that a shape class
 * really exercised the variant it is meant to cover): which = 0 persistent gather-GEMM (conv_persist_kernel.hip),
 * 1 direct 7x7 stem forward (stem_kernel.hip; same values up to fp32 summation order),
 * 2 patch-resident gather-GEMM (conv_patch_kernel.hip: stride-1 gathers with 1 or 4 phases whose pixel tile's input
 *   patch stays in LDS; same values up to fp32 summation order -- its K loop walks channel slices first),
 * 3 the 256x256 ping-pong gather-GEMM with a patch-resident pixel operand (conv_pipe_patch_kernel.hip; bit-identical to
 *   the ring kernel it replaces: same fragments, same accumulation order),
 * 4 the weight-stationary gather-GEMM of the short-K layers (conv_wsreg_kernel.hip),
 * 5 the accumulator-stationary weight gradient of the 3x3 layers (wgrad_rows_kernel.hip; same values up to fp32 summation order) */
long mt_kernel_variant_launches(int which);
/* switch such a variant off / on again (tests compare it bit for bit with the kernel it replaces; both are
 * results-identical by construction); returns the previous setting.  MT_IGEMM_PERSIST=0 in the environment disables
 * variant 0 from the start, MT_IGEMM_PATCH=0 variant 2.  Variant 2 has three settings: 0 off, 1 the shapes where it
 * measured faster inside the training step (default), 2 every shape it can run.  MT_IGEMM_PIPE_PATCH=0 disables variant 3. */
int mt_kernel_variant_enable(int which, int enable);
/* changes whenever mt_kernel_variant_enable was called: workspace sizes and kernel choices of a descriptor may be cached by a
 * caller as long as this value stands */
long mt_kernel_variant_epoch(void);
static inline int mt_padc(int c) { return (c + 7) & ~7; }

/* ---- convolution family (K1-K8, K12, K17): blocks.py:10-91, networks.py ------------- */
int mt_conv_out_hw(const mt_conv_desc* d, int* Ho, int* Wo);
size_t mt_conv_pack_bytes(const mt_conv_desc* d, int which);
/* w: reference-layout fp32 weights.  pack: mt_conv_pack_bytes() bytes. */
int mt_conv_pack(const mt_conv_desc* d, int which, const float* w, void* pack, mt_stream_t s);
/* Batched pack of n (descriptor, which, weights, pack buffer) tuples in ONE launch (a network's weights after its
 * optimizer step).  _build fills a HOST table of mt_conv_pack_multi_table_bytes(n) bytes; the caller copies it to
 * device memory once (the addresses in it must stay valid) and calls _run with that copy whenever the weights
 * changed.  Same images as mt_conv_pack, byte for byte.  *n_entries / *total_blocks are opaque values to hand back to
 * _run (they carry the counts of both kernels behind the call: tensors with k*k <= 16 taps are read ONCE per call through
 * LDS tiles for all of their images, the rest element-wise).  _run issues at most two launches on `s`. */
size_t mt_conv_pack_multi_table_bytes(int n);
int mt_conv_pack_multi_build(int n, const mt_conv_desc* descs, const int* which, const float* const* w,
                             void* const* packs, void* host_table, int* n_entries, int* total_blocks);
int mt_conv_pack_multi_run(const void* dev_table, int n_entries, int total_blocks, mt_stream_t s);
/* y = act(conv(x) + bias).  bias may be NULL (length Co, fp32). */
int mt_conv_fwd(const mt_conv_desc* d, const void* x, const void* pack_fwd, const float* bias,
                void* y, mt_stream_t s);
/* The same with an optional workspace of mt_conv_fwd_ws_bytes(d) bytes (0 for most shapes): few-pixel / long-K
 * layers (the discriminators' deep 3x3 / 4x4 stride-2 layers) are then split over the filter taps into fp32
 * partial slabs (split-K) and summed by a finish kernel -- same result up to fp32 summation order. */
size_t mt_conv_fwd_ws_bytes(const mt_conv_desc* d);
int mt_conv_fwd_ex(const mt_conv_desc* d, const void* x, const void* pack_fwd, const float* bias,
                   void* y, void* ws, size_t ws_bytes, mt_stream_t s);
/* Forward with the normalisation statistics of the output fused into the GEMM epilogue:
 * stats [N][Cp][2] = {sum, sum of squares} over H*W per (image, channel) -- one "partial row" per image in the
 * sense of mt_nc_stats (nparts = 1), so the InstanceNorm/AdaIN/LayerNorm that follows (blocks.py:38-42,158-164)
 * skips its statistics pass.  Only for shapes with mt_conv_fwd_stats_fused(d) != 0 (plain convolution, no
 * activation, Ho*Wo a multiple of 256); the caller passes stats ZERO-FILLED (the epilogue accumulates into it with
 * fp32 atomics: the one statistics path whose summation order varies from run to run -- callers that need
 * bit-reproducible runs use mt_conv_fwd + mt_nc_stats instead). */
int mt_conv_fwd_stats_fused(const mt_conv_desc* d);
int mt_conv_fwd_stats(const mt_conv_desc* d, const void* x, const void* pack_fwd, const float* bias,
                      void* y, float* stats, mt_stream_t s);
size_t mt_conv_bwd_data_ws_bytes(const mt_conv_desc* d);
/* dx = conv_bwd_data(dy).  dy is the gradient w.r.t. the pre-activation output. */
int mt_conv_bwd_data(const mt_conv_desc* d, const void* dy, const void* pack_bwd, void* dx,
                     void* ws, size_t ws_bytes, mt_stream_t s);
/* dx = data gradient + addend (a tensor of dx's shape and type; the skip-connection gradient of a residual block riding on
 * its first convolution's data gradient: replaces autograd's separate accumulation pass).  Added inside the GEMM epilogue
 * where the kernel supports it, by a separate in-place add otherwise; same workspace as mt_conv_bwd_data. */
int mt_conv_bwd_data_add(const mt_conv_desc* d, const void* dy, const void* pack_bwd, void* dx, const void* addend,
                         void* ws, size_t ws_bytes, mt_stream_t s);
/* ... and the statistics of the normalisation BACKWARD that consumes dx, taken in the same epilogue (the sums mt_nc_stats_bwd
 * would compute in a pass of its own over dx and x):  with g = dx * act'(scale[n][c] * x + shift[n][c]),
 * sums[n][c] += {sum g, sum g * x} over the pixels.  x: the norm's input (dx's shape and type); scale / shift: the fp32 [N][Cp]
 * coefficients of its forward; sums: fp32 [N][Cp][2], MUST BE ZERO on entry (float atomics: not in deterministic runs).
 * *stats_done = 1 if the kernel that ran produced them (the patch-resident 256x256 kernel), 0 if the caller still has to
 * run mt_nc_stats_bwd.  addend and bs may each be NULL. */
typedef struct mt_bwd_stats {
  const void* x;
  const float* scale;
  const float* shift;
  float* sums;
  int act;
  float slope;
} mt_bwd_stats;
int mt_conv_bwd_data_ex(const mt_conv_desc* d, const void* dy, const void* pack_bwd, void* dx, const void* addend,
                        const mt_bwd_stats* bs, int* stats_done, void* ws, size_t ws_bytes, mt_stream_t s);
size_t mt_conv_bwd_weight_ws_bytes(const mt_conv_desc* d);
/* dw (reference layout, fp32), dbias (fp32 [Co]; may be NULL).  accumulate == 0: both are overwritten;
 * accumulate != 0: the gradients are ADDED to their current contents (lets the caller point dw/dbias at
 * the parameter's .grad and skip a separate accumulation pass). */
int mt_conv_bwd_weight(const mt_conv_desc* d, const void* x, const void* dy, float* dw,
                       float* dbias, void* ws, size_t ws_bytes, int accumulate, mt_stream_t s);
/* Grouped weight gradient: G (2..8) convolutions OF THE SAME DESCRIPTOR in one launch (reference: the weight gradients
 * autograd computes one by one for nn.Conv2d, blocks.py:131-132,150-151 -- they are leaves of the backward pass, so their
 * order is free).  The 256x256 weight-gradient kernel splits the pixel reduction to fill the chip and writes one fp32 slab
 * of the whole gradient per split (28 for the dominant layer); G problems share the chip with 28 / G splits each.
 * mt_conv_bwd_weight_group_max: the largest G this descriptor supports (1 = use mt_conv_bwd_weight); x, dy, dw: tables of
 * G device pointers; ws >= mt_conv_bwd_weight_group_ws_bytes(d, G); no bias gradient (take it from mt_conv_bwd_weight_partial
 * or mt_act_bwd_bias).  Results equal mt_conv_bwd_weight's up to the fp32 summation order over the pixels. */
int mt_conv_bwd_weight_group_max(const mt_conv_desc* d);
size_t mt_conv_bwd_weight_group_ws_bytes(const mt_conv_desc* d, int G);
int mt_conv_bwd_weight_group(const mt_conv_desc* d, int G, const void* const* x, const void* const* dy, float* const* dw,
                             void* ws, size_t ws_bytes, int accumulate, mt_stream_t s);
/* The same in two halves (mt_conv_bwd_weight = partial + finish on one stream): `partial` runs the bias gradient (if
 * dbias != NULL) and the split GEMM into fp32 slabs in ws and reports their number; `finish` is the streaming sum of
 * the slabs into dw.  A caller may issue `finish` on ANOTHER stream (after an event recorded behind `partial`) so that
 * it runs beside the next layer's GEMMs; ws must stay untouched until it has run. */
int mt_conv_bwd_weight_partial(const mt_conv_desc* d, const void* x, const void* dy, float* dbias, void* ws,
                               size_t ws_bytes, int accumulate, int want_dw, int* nslabs, mt_stream_t s);
int mt_conv_bwd_weight_finish(const mt_conv_desc* d, const void* ws, int nslabs, float* dw, int accumulate,
                              mt_stream_t s);
/* n mt_conv_bwd_weight_finish calls in as few launches as possible (the slab sums of a whole backward pass from its end: nothing
 * reads a weight gradient before the optimizer step): entries with the generic slab form ride in batched launches of up to 64, the
 * others run their own finish.  Same sums in the same order as n single calls. */
int mt_conv_bwd_weight_finish_multi(int n, const mt_conv_desc* descs, const void* const* ws, const int* nslabs, float* const* dw,
                                    int accumulate, mt_stream_t s);
/* Size of one slab of mt_conv_bwd_weight_partial if the slabs have the generic form (it depends on the weight's shape only), else
 * 0.  Several uses of ONE weight in a backward pass (the weight-shared scales of MultiScaleDiscriminator, networks.py:330-365; an
 * encoder applied twice) may then write their slabs behind each other into one workspace -- partial(d_i, ..., ws + used slabs) --
 * and ONE mt_conv_bwd_weight_finish(d_any, ws, total slabs, dw, accumulate) sums them: one pass over the gradient instead of one
 * per use. */
size_t mt_conv_bwd_weight_slab_bytes(const mt_conv_desc* d);
/* Weight gradients of SEVERAL 3x3 / pad 1 layers (stride 1 or 2, transposed or not, 64-wide channel blocks, maps whole 32-pixel
 * strips wide: mt_conv_bwd_weight_rows_ok) in shared launches of the accumulator-stationary kernel (wgrad_rows_kernel.hip; replaces
 * the per-layer weight-gradient GEMMs autograd would run for nn.Conv2d / nn.ConvTranspose2d of blocks.py:10-91, networks.py:33,248):
 * a layer alone fills the chip only by cutting its pixel reduction ~256 ways; the eligible layers of a whole backward pass share
 * the compute units instead (one launch per stride class, one batched slab sum).  dw[i] (+)= the weight gradient of problem i in
 * the reference layout; problems that accumulate into the SAME dw must be adjacent.  n <= 64.  bf16 storage only. */
int mt_conv_bwd_weight_rows_ok(const mt_conv_desc* d);
size_t mt_conv_bwd_weight_rows_multi_ws_bytes(int n, const mt_conv_desc* descs);
int mt_conv_bwd_weight_rows_multi(int n, const mt_conv_desc* descs, const void* const* x, const void* const* dy, float* const* dw,
                                  void* ws, size_t ws_bytes, int accumulate, mt_stream_t s);

/* ---- nn.Linear fp32 (K17): networks.py:127-128,256-261, norm.py:27 ------------------ */
int mt_linear_fwd(const float* x, const float* w, const float* b, float* y, int n, int in,
                  int out, int act, mt_stream_t s);
/* dx, dw, db may each be NULL.  accumulate != 0: dw and db are ADDED to (gradient accumulation straight into
 * param.grad), otherwise overwritten. */
int mt_linear_bwd(const float* x, const float* w, const float* dy, float* dx, float* dw,
                  float* db, int n, int in, int out, int accumulate, mt_stream_t s);

/* G <= 8 nn.Linear layers of equal shape that share their input x [n][in] (the four AdaIN projections of a decoder, all
 * applied to the same style code: norm.py:27, blocks.py:152-164) in one launch: y_g = x w_g^T + b_g.  w, b, y: HOST arrays
 * of G device pointers (b may be NULL, or hold NULLs).  Backward: dx = sum_g dy_g w_g (groups added in index order; may
 * be NULL), dw_g / db_g as mt_linear_bwd (entries may be NULL), accumulate as there. */
int mt_linear_group_fwd(const float* x, const float* const* w, const float* const* b, float* const* y, int groups, int n,
                        int in, int out, mt_stream_t s);
int mt_linear_group_bwd(const float* x, const float* const* w, const float* const* dy, float* dx, float* const* dw,
                        float* const* db, int groups, int n, int in, int out, int accumulate, mt_stream_t s);

/* ---- normalisation family (K9, K10, K11): functions.py:17, norm.py:5-33 -------------- */
/* Statistics pass: part[n][k][c] = {sum x, sum x^2} over the k-th pixel block of image n (fp32,
 * [N][nparts][Cp][2], nparts = mt_nc_stats_parts(dtype, N, HW, Cp) <= 64; every element is written, nothing to
 * zero).  Fixed summation order inside a block and in the finalize kernels, which add the nparts rows of an
 * image in index order: results are bit-reproducible. */
int mt_nc_stats_parts(int dtype, int N, int HW, int Cp);
int mt_nc_stats(int dtype, const void* x, float* part, int N, int HW, int Cp, mt_stream_t s);
/* scale/shift [N][Cp] for y = act(scale*x + shift); mean/rstd saved for backward.
 * mode INSTANCE: gamma=beta=NULL.  ADAIN: gb = fc(s) [N][2*C] (weight = 1+gb[:, :C], bias =
 * gb[:, C:]).  LAYER: per-sample statistics over (C,H,W); gamma,beta [C] (norm.py:16-21). */
int mt_norm_finalize(int mode, const float* sums, const float* gb, const float* gamma,
                     const float* beta, float* scale, float* shift, float* mean, float* rstd,
                     int N, int HW, int C, int Cp, float eps, int nparts, mt_stream_t s);
/* y = act(scale[n][c]*x + shift[n][c]) (+ res).  res may be NULL. */
int mt_scale_shift_act(int dtype, const void* x, const float* scale, const float* shift,
                       const void* res, void* y, int N, int HW, int Cp, int act, float slope,
                       mt_stream_t s);
/* mt_norm_finalize + mt_scale_shift_act in ONE launch for complete per-image statistics (sums [N][Cp][2]: the stats of
 * mt_conv_fwd_stats, or an mt_nc_stats result with nparts == 1): every block derives the coefficients of its image once
 * and streams its pixels.  mode INSTANCE / ADAIN / LAYER; coef [4][N][Cp] receives scale, shift, mean, rstd (what the
 * backward calls take).  "conv -> norm -> activation (+ residual)" is then two launches: conv(+statistics), this. */
int mt_norm_apply_fused(int dtype, int mode, const void* x, const float* sums, const float* gb, const float* gamma,
                        const float* beta, const void* res, void* y, float* coef, int N, int HW, int C, int Cp, int act,
                        float slope, float eps, mt_stream_t s);
/* g = dy * act'(scale*x+shift); part2[n][k][c] = {sum g, sum g*x} over pixel block k (layout and nparts as
 * mt_nc_stats). */
int mt_nc_stats_bwd(int dtype, const void* dy, const void* x, const float* scale,
                    const float* shift, float* part2, int N, int HW, int Cp, int act,
                    float slope, mt_stream_t s);
/* coefficients for dx = c1*g + c2 + c3*x; plus parameter gradients:
 * ADAIN: dgb [N][2C]; LAYER: dgamma, dbeta [C], with dgb a caller-provided [N][2][C] scratch (per-image terms,
 * added over the images in index order). */
int mt_norm_bwd_finalize(int mode, const float* sums2, const float* mean, const float* rstd,
                         const float* gb, const float* gamma, float* c1, float* c2, float* c3,
                         float* dgb, float* dgamma, float* dbeta, int N, int HW, int C, int Cp,
                         int nparts, mt_stream_t s);
int mt_norm_bwd_apply(int dtype, const void* dy, const void* x, const float* scale,
                      const float* shift, const float* c1, const float* c2, const float* c3,
                      void* dx, int N, int HW, int Cp, int act, float slope, mt_stream_t s);
/* The three backward calls above in ONE launch and ONE pass over dy and x (InstanceNorm / AdaIN / LayerNorm, bf16; functions.py:17,
 * norm.py:5-33 backward; LAYER (round 4): gamma [C] or null, dgb = the [N][2][C] per-image scratch of mt_norm_bwd_finalize,
 * dgamma / dbeta [C] or null -- the images' terms are added in index order by a second, tiny launch): a workgroup keeps a slice of an image (8192 sixteen-byte chunks of x and of dy) in registers, the
 * slices of an image exchange their partial {sum g, sum g x} rows through `part` ([N][slices][Cp][2] floats, need not be
 * initialised) and meet at per-image counters in `sync` ([2][N] unsigned, ZERO before the first launch; the kernel leaves them
 * zero), every slice adds the rows in index order (reproducible), derives c1, c2, c3 as mt_norm_bwd_finalize does (dgb [N][2C]
 * for ADAIN) and writes dx.  scale / shift / mean / rstd: the coef rows of mt_norm_apply_fused / mt_norm_finalize.
 * mt_norm_bwd_onepass_ok says whether a problem qualifies (power-of-two channel chunks, planes that are whole slices, act none /
 * relu / lrelu, at least 128 workgroups, and no more slices per image than min(max_slices, mt_norm_bwd_onepass_capacity()) --
 * every slice of an image waits for the others, so all of them must be resident together; max_slices <= 0: the capacity) and
 * returns the slice count; otherwise use the three calls.  mt_norm_bwd_onepass_capacity = occupancy of the kernel x compute units
 * of the current device.
 * Contract of the wait: launches that share a `sync` buffer must be ordered (one stream), and no other kernel that waits for
 * sibling workgroups may run on the device at the same time (one process per GPU; hip_ops orders launches from different
 * streams behind each other).  The wait is bounded by `spin_limit` polls of ~1 us (<= 0: 2^19): a workgroup that gives up
 * poisons dx / dgb with NaN AND sets bit 0 of status[0] (status[1] = 1 + its workgroup index; `status`: two zeroed unsigned
 * words owned by the caller, who reads them at its next device->host copy and treats a set bit as a failed step). */
int mt_norm_bwd_onepass_capacity(void);
int mt_norm_bwd_onepass_ok(int dtype, int mode, int N, int HW, int Cp, int act, int max_slices, int* slices);
int mt_norm_bwd_onepass(int dtype, int mode, const void* dy, const void* x, const float* scale, const float* shift,
                        const float* mean, const float* rstd, const float* gb, float* dgb, const float* gamma, float* dgamma,
                        float* dbeta, void* dx, float* part, unsigned* sync, unsigned* status, int spin_limit, int N, int HW,
                        int C, int Cp, int act, float slope, mt_stream_t s);
/* BatchNorm2d(affine, running statistics; functions.py:14-15) on the shared passes: bn_finalize pools the per-(n, c) sums
 * of mt_nc_stats / mt_conv_fwd_stats over the batch (training: batch statistics + momentum update of the running
 * buffers with the unbiased variance; eval: the running buffers) and fills the [N][Cp] coefficient arrays for
 * mt_scale_shift_act; bn_bwd_finalize does the same for mt_norm_bwd_apply from mt_nc_stats_bwd's sums
 * (dgamma, dbeta: [C]). */
int mt_bn_finalize(const float* sums, const float* gamma, const float* beta, float* running_mean, float* running_var,
                   float momentum, float eps, int training, float* scale, float* shift, float* mean, float* rstd, int N,
                   int HW, int C, int Cp, int nparts, mt_stream_t s);
int mt_bn_bwd_finalize(const float* sums2, const float* mean, const float* rstd, const float* gamma, float* c1, float* c2,
                       float* c3, float* dgamma, float* dbeta, int training, int N, int HW, int C, int Cp, int nparts,
                       mt_stream_t s);

/* ---- elementwise / pooling / layout (K13, K14, K15, K19) ----------------------------- */
int mt_act_fwd(int dtype, const void* x, void* y, size_t n, int act, float slope, mt_stream_t s);
/* dx = dy * act'(.) evaluated from the activation OUTPUT y. */
int mt_act_bwd(int dtype, const void* dy, const void* y, void* dx, size_t n, int act,
               float slope, mt_stream_t s);
/* act_bwd and the bias gradient in one pass over dy (convolutions with a fused activation): dz = dy * act'(y),
 * dbias[c] (+)= sum_pixels dz[pixel][c] (fp32 [C]; block partials in ws, added in block order -- reproducible).
 * ws: mt_act_bwd_bias_ws_bytes(Cp) bytes. */
size_t mt_act_bwd_bias_ws_bytes(int Cp);
int mt_act_bwd_bias(int dtype, const void* dy, const void* y, void* dz, size_t npix, int Cp, int C, int act,
                    float slope, float* dbias, int accumulate, void* ws, size_t ws_bytes, mt_stream_t s);
int mt_add(int dtype, const void* a, const void* b, void* y, size_t n, mt_stream_t s);
/* y = x + noise (misc.py:22-26).  noise given (parity mode) ... */
/* ... or generated on device: Philox4x32-10 + Box-Muller, N(0,1), counter = element index. */
int mt_gaussian_noise_add(int dtype, const void* x, void* y, size_t n, uint64_t seed,
                          uint64_t offset, mt_stream_t s);
/* The same draw from a DEVICE-resident generator state {uint64 seed, uint64 draw counter}: nothing random travels in
 * the launch arguments, so the call can sit in a captured hipGraph and still produce fresh noise on every replay (draw
 * k uses Philox counters [k << 40, (k+1) << 40)).  mt_rng_advance(state) increments the draw counter in stream order;
 * call it after each draw. */
int mt_gaussian_noise_add_dev(int dtype, const void* x, void* y, size_t n, const uint64_t* dev_state, mt_stream_t s);
int mt_rng_advance(uint64_t* dev_state, mt_stream_t s);
/* nn.Dropout(0.5) (--use_dropout; blocks.py:133-134, 153-165, 192-207): mask ~ Bernoulli(keep) in the activation layout
 * [npix][Cp] (pad channels 0; Philox4x32-10, counter = element index), and y = a * b * scale (forward: x * mask / keep,
 * backward: dy * mask / keep). */
int mt_bernoulli_mask(int dtype, void* mask, size_t npix, int C, int Cp, float keep, uint64_t seed, uint64_t offset,
                      mt_stream_t s);
int mt_bernoulli_mask_dev(int dtype, void* mask, size_t npix, int C, int Cp, float keep, const uint64_t* dev_state,
                          mt_stream_t s);
int mt_mul_scale(int dtype, const void* a, const void* b, void* y, size_t n, float scale, mt_stream_t s);
int mt_avgpool2_fwd(int dtype, const void* x, void* y, int N, int H, int W, int Cp, mt_stream_t s);
int mt_avgpool2_bwd(int dtype, const void* dy, void* dx, int N, int H, int W, int Cp, mt_stream_t s);
/* nn.Upsample(scale_factor=2, mode='nearest') (blocks.py:75, --up_type nearest).  H, W: size of the UPSAMPLED map.
 * fwd: src [N][H/2][W/2][Cp] -> dst [N][H][W][Cp]; bwd: src = d(upsampled) [N][H][W][Cp] -> dst [N][H/2][W/2][Cp]. */
int mt_upsample2_fwd(int dtype, const void* src, void* dst, int N, int H, int W, int Cp, mt_stream_t s);
int mt_upsample2_bwd(int dtype, const void* src, void* dst, int N, int H, int W, int Cp, mt_stream_t s);
/* Spectral normalisation of a conv weight (--dis_sn; functions.py:113-121 -> torch.nn.utils.spectral_norm with
 * n_power_iterations=1, eps=1e-12, dim=0; blocks.py:33-34).  W: fp32 [rows = Cout][cols = Cin*kh*kw].
 * power_iter: n_iter times { v <- normalize(W^T u); u <- normalize(W v) } in place, then sigma[0] = u . (W v);
 *             n_iter = 0 (eval mode) only evaluates sigma with the stored u, v.
 * scale_fwd : Weff = W / sigma.     scale_bwd: dW = (G - <G, Weff> u v^T) / sigma   (u, v constants).
 * ws: mt_sn_ws_bytes(rows, cols) bytes. */
size_t mt_sn_ws_bytes(int rows, int cols);
int mt_sn_power_iter(const float* W, float* u, float* v, float* sigma, int rows, int cols, int n_iter, float eps,
                     void* ws, size_t ws_bytes, mt_stream_t s);
int mt_sn_scale_fwd(const float* W, const float* sigma, float* Weff, long n, mt_stream_t s);
int mt_sn_scale_bwd(const float* G, const float* Weff, const float* u, const float* v, const float* sigma, float* dW,
                    int rows, int cols, void* ws, size_t ws_bytes, mt_stream_t s);
/* AvgPool2d(3, stride 2, pad 1, count_include_pad=False) (networks.py:447) */
int mt_avgpool3s2_fwd(int dtype, const void* x, void* y, int N, int H, int W, int Cp, mt_stream_t s);
int mt_avgpool3s2_bwd(int dtype, const void* dy, void* dx, int N, int H, int W, int Cp, mt_stream_t s);
/* Patches of a 4x4 / stride 2 / zero-padding 1 convolution (nn.Conv2d(.., 4, 2, 1), the layers of MultiScaleDiscriminator,
 * networks.py:330-365) as a batch of 4x4 mini-images: col [N*(H/2)*(W/2)][4][4][Cp] with col[(n,ho,wo)][a][b] = x[n][2ho-1+a]
 * [2wo-1+b] (zeros outside).  The same weights applied as a 4x4 / stride 4 / padding 0 convolution to the mini-images give the
 * original outputs, so inputs of DIFFERENT sizes that share a weight can run as one batch.  bwd: dx = the adjoint (each input
 * pixel sums the at most four patch cells that copied it).  H, W even. */
int mt_patch4s2_fwd(int dtype, const void* x, void* col, int N, int H, int W, int Cp, mt_stream_t s);
int mt_patch4s2_bwd(int dtype, const void* dcol, void* dx, int N, int H, int W, int Cp, mt_stream_t s);
/* the same for up to four inputs of equal channel count in ONE launch (the weight-shared scales of a multi-scale discriminator
 * layer): bwd = 0: src[k] = input k, dst[k] = its slice of the mini-image batch; bwd = 1: src[k] = the slice of the batch's
 * gradient, dst[k] = the gradient of input k (null: not needed). */
int mt_patch4s2_multi(int dtype, int bwd, int count, const void* const* src, void* const* dst, const int* N, const int* H,
                      const int* W, int Cp, mt_stream_t s);
/* AdaptiveAvgPool2d(1): y fp32 [N][C] (logical channels). */
int mt_gap_fwd(int dtype, const void* x, float* y, int N, int HW, int C, int Cp, mt_stream_t s);
int mt_gap_bwd(int dtype, const float* dy, void* dx, int N, int HW, int C, int Cp, mt_stream_t s);
/* generic strided (element strides sn,sc,sh,sw) source of dtype src_dtype -> canonical NHWC
 * padded tensor of dtype dst_dtype, pad channels zeroed. */
int mt_to_nhwc(int src_dtype, const void* x, int64_t sn, int64_t sc, int64_t sh, int64_t sw,
               int dst_dtype, void* y, int N, int C, int H, int W, mt_stream_t s);
/* canonical NHWC padded -> dense NCHW fp32 (logical channels). */
int mt_to_nchw_f32(int dtype, const void* x, float* y, int N, int C, int H, int W, mt_stream_t s);
/* Style-encoder input (networks.py:138-140): out[...,0:C]=img, out[...,C:C+D]=onehot[n]. */
int mt_cat_class_planes(int dtype, const void* img, const float* cls, void* out, int N, int HW,
                        int C, int D, mt_stream_t s);
/* backward of the above for the image part: dimg[..., 0:C] = dout[..., 0:C]. */
int mt_slice_channels(int dtype, const void* x, void* y, int N, int HW, int Cx, int C,
                      mt_stream_t s);

/* ---- losses (K16, K18): loss.py:52-64, adain_model.py:74,303-314,379-381,396-399 ------ */
/* mean BCE-with-logits of x (NHWC padded, logical C) against a constant target t in {0,1}.
 * loss: fp32 scalar (overwritten). */
int mt_bce_const_fwd(int dtype, const void* x, float t, float* loss, size_t npix, int C, int Cp,
                     mt_stream_t s);
/* dx = gscale[0] * (sigmoid(x) - t) / (npix*C) on logical channels, 0 on pad. */
int mt_bce_const_bwd(int dtype, const void* x, float t, const float* gscale, void* dx,
                     size_t npix, int C, int Cp, mt_stream_t s);
/* The other GANLoss modes, same conventions as mt_bce_const_* (loss.py:44-61; adain_model.py:209-210, 293-295):
 * LSGAN mean (x-t)^2; HINGE_D mean relu(1-x) for t=1 / mean relu(1+x) for t=0; NEG_MEAN -mean x for t=1 / mean x. */
#define MT_GAN_LSGAN 1
#define MT_GAN_HINGE_D 2
#define MT_GAN_NEG_MEAN 3
int mt_gan_const_fwd(int dtype, int mode, const void* x, float t, float* loss, size_t npix, int C, int Cp,
                     mt_stream_t s);
int mt_gan_const_bwd(int dtype, int mode, const void* x, float t, const float* gscale, void* dx, size_t npix,
                     int C, int Cp, mt_stream_t s);
/* mean BCE-with-logits of fp32 x[n] against fp32 targets t[n]. */
int mt_bce_target_fwd(const float* x, const float* t, float* loss, size_t n, mt_stream_t s);
int mt_bce_target_bwd(const float* x, const float* t, const float* gscale, float* dx, size_t n,
                      mt_stream_t s);
/* mean |a-b| over `count` logical elements (n = padded element count). */
int mt_l1_fwd(int dtype, const void* a, const void* b, float* loss, size_t n, size_t count,
              mt_stream_t s);
/* da = gscale * sign(a-b)/count ; db = -da (either may be NULL). */
int mt_l1_bwd(int dtype, const void* a, const void* b, const float* gscale, void* da, void* db,
              size_t n, size_t count, mt_stream_t s);
/* mean x^2 over count logical elements (adain_model.py:396-399). */
int mt_l2mean_fwd(int dtype, const void* x, float* loss, size_t n, size_t count, mt_stream_t s);
int mt_l2mean_bwd(int dtype, const void* x, const float* gscale, void* dx, size_t n, size_t count,
                  mt_stream_t s);
/* z = eps*exp(0.5*logvar)+mu (networks.py:130-135). */
int mt_reparam_fwd(const float* mu, const float* logvar, const float* eps, float* z, size_t n,
                   mt_stream_t s);
int mt_reparam_bwd(const float* logvar, const float* eps, const float* dz, float* dmu,
                   float* dlogvar, size_t n, mt_stream_t s);
/* kl = -0.5 * sum(1 + logvar - mu^2 - exp(logvar)) (adain_model.py:313-314; a SUM). */
int mt_kl_fwd(const float* mu, const float* logvar, float* kl, size_t n, mt_stream_t s);
int mt_kl_bwd(const float* mu, const float* logvar, const float* gscale, float* dmu,
              float* dlogvar, size_t n, mt_stream_t s);

/* The model's loss expression in one launch (adain_model.py:193-195, 316-321, 381-389).  terms: HOST array of n device
 * scalars; term i has weight w[i] and belongs to group gid[i] < G.  out (device, G + 2 floats): out[g] = sum of the
 * weighted terms of group g (the values the model logs), out[G] = sum_g Wb[g]*out[g] (the loss that is differentiated),
 * out[G+1] = sum_g Wr[g]*out[g] (the reported total; Wr NULL = Wb).  n <= 16, G <= 8; fixed summation order.
 * mt_loss_sum_bwd: dterms[i] = gtotal * Wb[gid[i]] * w[i]. */
int mt_loss_sum_fwd(const float* const* terms, const float* w, const int* gid, int n, const float* Wb, const float* Wr,
                    int G, float* out, mt_stream_t s);
int mt_loss_sum_bwd(const float* gtotal, const float* w, const int* gid, int n, const float* Wb, int G, float* dterms,
                    mt_stream_t s);

/* ---- optimizer (K20): torch.optim.Adam, adain_model.py:57-61 -------------------------- */
/* One fused launch over `count` tensors.  ptrs: device array of 4*count pointers laid out
 * [p0,g0,m0,v0,p1,...]; sizes: device array of element counts; L2-coupled weight decay. */
int mt_adam_multi(void* const* ptrs, const int64_t* sizes, int count, int64_t max_size, float lr,
                  float beta1, float beta2, float eps, float wd, int step, mt_stream_t s);
/* The same update with the per-step scalars on the device: dev_state = 16 bytes {float lr; int32 step; float bc1; float
 * bc2_sqrt}.  The call first ticks the record (step += 1, bias corrections recomputed in double precision), then updates
 * with it -- no launch argument changes between steps, so the pair is hipGraph-capturable.  The caller initialises
 * lr and step (bc* are outputs) and rewrites lr when the schedule changes it.  zero_grads != 0: the gradient buffers
 * are cleared in the same pass (they were just read), which replaces the separate zero_grad() memset. */
int mt_adam_multi_dev(void* const* ptrs, const int64_t* sizes, int count, int64_t max_size, float beta1, float beta2,
                      float eps, float wd, void* dev_state, int zero_grads, mt_stream_t s);

/* ---- gradient exchange over RCCL / xGMI (C1): replaces nn.DataParallel, functions.py:98-101 ---------------- */
/* One process per GPU.  Rank 0 creates a 128-byte id (mt_comm_unique_id) and hands it to the other ranks through any
 * side channel (the host code broadcasts it over torch.distributed); every rank then calls mt_comm_init(.., device).
 * mt_comm_allreduce_async: buf (fp32 [count], in place) <- AVERAGE over the ranks, enqueued on the communicator's own
 * side stream behind everything `producer` has been given so far; returns a handle >= 0 (negative on error).
 * mt_comm_wait: `consumer` waits on the device for that handle (no host blocking).  Up to 64 handles may be in flight.
 * RCCL is dlopen()ed on first use: the library has no link-time dependency on it. */
typedef struct mt_comm mt_comm;
int mt_comm_unique_id(void* id128);
int mt_comm_init(mt_comm** comm, int rank, int world, const void* id128, int device);
int mt_comm_allreduce_async(mt_comm* comm, float* buf, size_t count, mt_stream_t producer);
int mt_comm_wait(mt_comm* comm, int handle, mt_stream_t consumer);
int mt_comm_rank(const mt_comm* comm);
int mt_comm_world(const mt_comm* comm);
int mt_comm_destroy(mt_comm* comm);

#ifdef __cplusplus
}
#endif
#endif /* MT_API_H */
