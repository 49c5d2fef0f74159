// Kernel parameter blocks shared between the host-side problem builders (conv_api.hip) and
// the device kernels (conv_kernels.hip).
#pragma once
#include <stdint.h>
#include <hip/hip_runtime.h>

#define MT_MAX_TAPS 64

#define MT_MAX_PHASES 16

// One sub-problem of a launch.  Gather-form convolutions have one; the scatter form (stride-s data
// gradient / transposed conv) has s*s sub-pixel phases that differ in their tap subset, weight slice and
// output offset.  All phases run in ONE launch (blocks [blk0, blk0+nblk) belong to a phase), so the
// dispatcher balances them instead of paying a ragged last round of blocks per phase.
struct IgemmPhase {
  unsigned w_off;     // byte offset of this phase's first weight chunk (row 0) from IgemmParams::w
  unsigned w_bytes;   // bytes of the weight image that are addressable from there (buffer range check)
  int wrow;           // 16-byte chunks per row of the weight image (>= ntaps*cpc: a phase may use a run of
                      // consecutive taps of a wider image)
  int ntaps, tap0;    // taps [tap0, tap0+ntaps) of the dh/dw tables
  int Ho, Wo, M;      // GEMM pixel grid of the phase, M = N*Ho*Wo
  int oh0, ow0;       // output pixel = (ho*os + oh0, wo*os + ow0)
  int blk0, nblk;     // filled by the launcher for the chosen tile geometry
  unsigned y_off;     // byte offset of this phase's output tensor from IgemmParams::y (split-K partial slabs)
};

struct IgemmParams {
  const char* x;      // gathered operand, NHWC [N][Hi][Wi][Cib bytes]
  const char* w;      // packed weights [CoRows][nchunks * 16 bytes]
  const float* bias;  // optional fp32 [nbias]
  char* y;            // output NHWC [N][Hout][Wout][Co]
  float* stats;       // optional fp32 [N][Co][2]: += {sum, sum of squares} of the outputs per (image, channel)
                      // (InstanceNorm statistics fused into the epilogue; needs Ho*Wo % 256 == 0, single phase)
  int N, Hi, Wi;
  int Cib;            // bytes per input pixel (padded channels * element size)
  unsigned x_bytes;   // size of the input tensor / weight pack in bytes (hardware range check of the
  unsigned w_bytes;   // buffer loads: out-of-range lanes read zeros -> zero padding costs no branch)
  int Co;             // padded output channels (elements per output pixel)
  int CoRows;         // rows of the weight pack (== Co)
  int nbias;
  int Hout, Wout;     // output tensor spatial size
  int os;             // output pixel stride (see IgemmPhase); out-of-range pixels are skipped
  int is;             // input base coordinate = ho*is (+ dh[tap])
  int cpc;            // 16-byte chunks per tap
  int nphase;
  IgemmPhase ph[MT_MAX_PHASES];
  int pad_mode;
  int act;
  float slope;
  int interleave;     // multi-phase launches whose phases have equal block counts: block b -> (phase b % nphase, tile b / nphase),
                      // so the sub-pixel phases of one input region run side by side and share it in L2
  int korder;         // ping-pong kernel: 1 = walk K channel-slice-major (all taps of a 32-channel slice, then the next slice)
  char* y2;           // optional second destination (persistent gather-GEMM only, mt_igemm_would_persist): output pixels
  int y2P, y2H, y2W;  // inside [y2P, y2P+y2H) x [y2P, y2P+y2W) of the (Hout, Wout) grid go to y2 -- an [N][y2H][y2W][Co] tensor,
                      // at (oh - y2P, ow - y2P) -- instead of y: the interior of a padded gradient map straight into dx
  const char* addend; // optional tensor of the output's shape and type, added to the result in the epilogue (the residual
                      // block's skip gradient riding on conv1's data gradient; conv_pipe_patch_kernel.hip only: mt_igemm_fold_ok)
  // optional statistics of the normalisation BACKWARD whose gradient this launch produces (conv_pipe_patch_kernel.hip only):
  // with g = out * act'(bstat_scale * x + bstat_shift), stats[n][c] += {sum g, sum g * x} over the pixels (x = bstat_x: the
  // norm's input, same shape / type as the output; scale, shift: fp32 [N][Co]); `stats` is then that output, zero on entry
  const char* bstat_x;
  const float* bstat_scale;
  const float* bstat_shift;
  int bstat_act;
  float bstat_slope;
  int fold;           // stride-1 3x3 gather over dy at offsets -1 .. 1 (the interior of a reflection-padded data gradient): also
                      // add the reflected ring inside the pixel operand (conv_pipe_patch_kernel.hip; mt_igemm_fold_ok)
  int raw;            // split-K: write the fp32 accumulators as they are (no bias / activation, fp32 elements
                      // whatever the storage type); mt_launch_splitk_finish sums the slabs
  short dh[MT_MAX_TAPS];
  short dw[MT_MAX_TAPS];
};

#define MT_WGRAD_MAX_GROUP 8
struct WgradParams {
  const char* a;   // pixel-major [M][Cab bytes]: its channels become output rows
  const char* b;   // gathered NHWC [N][Hi][Wi][Cbb bytes]: (tap, channel) become output columns
  float* out;      // fp32 [nsplit][CaRows][nchunks * V]: one slab per pixel split, summed by unpack_kernel
  int N, Hi, Wi;
  int Cab, Cbb;    // bytes per pixel
  unsigned a_bytes, b_bytes;  // tensor sizes in bytes (buffer-load range checks)
  int CaRows;      // padded channel count of a
  int Ho, Wo, M;   // pixel grid of a
  int is;
  int ntaps, cpc, nchunks;
  int pad_mode;
  int mchunk;      // pixels per split
  int ntiles;      // output tiles (128x128) per split; grid = ntiles * nsplit blocks
  int rows_rs;     // wgrad_rows_kernel only: k-steps (strip, output row) per split
  short dh[MT_MAX_TAPS];
  short dw[MT_MAX_TAPS];
  // grouped launch (wgrad_pipe_kernel only): ngroup > 1 problems of this geometry in one grid, blocks = ngroup * ntiles * nsplit;
  // problem g reads ga[g] / gb[g] and writes its nsplit slabs at gout[g] (a, b, out unused)
  int ngroup;
  const char* ga[MT_WGRAD_MAX_GROUP];
  const char* gb[MT_WGRAD_MAX_GROUP];
  float* gout[MT_WGRAD_MAX_GROUP];
};

struct PackParams {
  int R, C;      // logical rows / cols
  int Rp, Cp;    // padded rows / cols of the packed image
  long sr, sc;   // element strides of (row, col) in the reference-layout weight tensor
  int kW;        // filter width (tap element offset = kh*kW + kw)
  int ntaps;
  short kh[MT_MAX_TAPS];
  short kw[MT_MAX_TAPS];
};

int mt_launch_igemm(int dtype, const IgemmParams& p, hipStream_t s);
// would mt_launch_igemm run this problem (IgemmParams::fold set) on the kernel that folds inside the operand?
bool mt_igemm_fold_ok(int dtype, const IgemmParams& p);
// would mt_launch_igemm run this problem on the persistent kernel (the only one that honours IgemmParams::y2)?
bool mt_igemm_would_persist(int dtype, const IgemmParams& p);
// y[i] = act(sum_s slabs[s][i] + bias[i % Cp])  (i over `total` NHWC elements, fp32 slabs, output in `dtype`)
int mt_launch_splitk_finish(int dtype, const float* slabs, int nsplit, long total, const float* bias, int nbias,
                            int Cp, void* y, int act, float slope, hipStream_t s);
int mt_launch_wgrad(int dtype, const WgradParams& p, int nsplit, hipStream_t s);
// 256x256 ping-pong variant (wgrad_pipe_kernel.hip)
bool mt_wgrad_pipe_ok(int dtype, int CaRows, int cpc, long a_bytes, long b_bytes);
int mt_wgrad_pipe_max_chunk();
// ... with 16-bit pixel deltas in the per-block table (stride-1 problems whose two pixel grids coincide): twice the pixels per split
bool mt_wgrad_pipe_compact_ok(const WgradParams& p);
int mt_wgrad_pipe_max_chunk_compact();
// ... and with 8-bit deltas (additionally: every tap within 126 pixels of its own pixel): four times the pixels per split
bool mt_wgrad_pipe_compact8_ok(const WgradParams& p);
int mt_wgrad_pipe_max_chunk_compact8();
int mt_launch_wgrad_pipe(const WgradParams& p, int nsplit, hipStream_t s);
// accumulator-stationary variant for 3x3 / pad 1 layers (wgrad_rows_kernel.hip).  ok: geometry check; plan: "is a launch of its
// own worth it, with how many slabs and k-steps per split"; *_multi: several problems (ps[i].a / .b / .out set) sharing launches
#define MT_WR_MAXP 24
bool mt_wgrad_rows_ok(int dtype, const WgradParams& p);
bool mt_wgrad_rows_plan(int dtype, const WgradParams& p, int* nsplit, int* rs);
void mt_wgrad_rows_plan_multi(int n, const WgradParams* ps, int* nsplit, int* rps);
int mt_launch_wgrad_rows_multi(int n, const WgradParams* ps, const int* nsplit, const int* rps, hipStream_t s);
int mt_launch_wgrad_rows(const WgradParams& p, int nsplit, hipStream_t s);
int mt_launch_pack(int dtype, const float* w, void* out, const PackParams& p, hipStream_t s);
// one entry of a batched weight pack (mt_conv_pack_multi_*): a whole network's weights in ONE launch
struct PackEntry {
  PackParams p;
  const float* w;
  void* out;
  int bf16;
  int blk0, nblk;   // blocks [blk0, blk0 + nblk) of the launch belong to this entry
};
int mt_launch_pack_multi(const PackEntry* dev_table, int n, int total_blocks, hipStream_t s);
// grouped variant (weight_pack_kernels.hip): ONE coalesced read of a weight tensor [D0][D1][kh][kw] per 32 x 32 tile of
// (d0, d1), all of the tensor's pack images written from the LDS copy in 64-byte runs
#define MT_PACK_GROUP_OUTS 20      // (a 4x4 / stride 4 data gradient has 16 one-tap phase images)
struct PackOut {
  void* out;
  int rows_d0;      // image rows are d0 (else d1)
  int Rp, Cp, ntaps;
  unsigned char tsrc[MT_MAX_TAPS];   // source tap (kh * kW + kw) of image tap t
};
struct PackGroup {
  const float* w;
  int D0, D1, K2, bf16, nout;
  int tile0, ntiles, tiles_d1;       // tiles [tile0, tile0 + ntiles) of the launch belong to this group
  PackOut o[MT_PACK_GROUP_OUTS];
};
int mt_launch_pack_groups(const PackGroup* dev_groups, int ngroups, int blocks, hipStream_t s);
int mt_launch_unpack(const float* src, float* dw, const PackParams& p, int nsplit, long slab, int accumulate,
                     hipStream_t s);
#define MT_UNPACK_MULTI_MAX 64
// batched form (weight_pack_kernels.hip): up to 64 natural-tap-order slab sums in one launch
bool mt_unpack_multi_ok(const PackParams& p);
int mt_launch_unpack_multi(int n, const float* const* src, float* const* dw, const PackParams* ps, const int* nsplit, const long* slab,
                           int accumulate, hipStream_t s);
int mt_launch_reflect_fold(int dtype, const void* src, void* dst, int N, int H, int W, int Cp, int P,
                           hipStream_t s);
int mt_launch_ring_fold(int dtype, const void* src, void* dst, int N, int H, int W, int Cp, int P, hipStream_t s);
size_t mt_colsum_ws_bytes(int Cp);
int mt_launch_colsum(int dtype, const void* dy, float* db, long npix, int Cp, int C, int accumulate, void* ws,
                     size_t ws_bytes, hipStream_t s);

// direct 7x7 stem forward (stem_kernel.hip): 0 launched, 2 error, -1 not its shape
int mt_launch_stem_fwd(const mt_conv_desc* d, const void* x, const void* pack, const float* bias, void* y, float* stats,
                       hipStream_t s);

// ... and its weight gradient: slabs [nslabs][64][224] into ws, then the reduce into the reference layout
bool mt_stem_wgrad_ok(const mt_conv_desc* d);
size_t mt_stem_wgrad_ws_bytes(const mt_conv_desc* d);
int mt_launch_stem_wgrad(const mt_conv_desc* d, const void* x, const void* dy, void* ws, int* nslabs, hipStream_t s);
int mt_launch_stem_wgrad_reduce(const mt_conv_desc* d, const void* ws, int nslabs, float* dw, int accumulate, hipStream_t s);

// ... and its data gradient on the padded grid (reflection padding: fold afterwards) or straight into dx (zero padding)
bool mt_stem_dgrad_ok(const mt_conv_desc* d);
int mt_launch_stem_dgrad(const mt_conv_desc* d, const void* dy, const void* pack_bwd, void* out, hipStream_t s);

// thin 1x1 convolutions (pointwise_kernels.hip); the launchers return -1 if no instantiation matches
bool mt_pointwise_small(const mt_conv_desc* d);
int mt_pw_fwd(const mt_conv_desc* d, const void* x, const void* wpack, const float* bias, void* y, long npix, hipStream_t s);
int mt_pw_bwd_data(const mt_conv_desc* d, const void* dy, const void* wpack, void* dx, long npix, hipStream_t s);
int mt_pw_bwd_weight(const mt_conv_desc* d, const void* x, const void* dy, float* out, long npix, int max_slabs,
                     int* nslabs, hipStream_t s);
