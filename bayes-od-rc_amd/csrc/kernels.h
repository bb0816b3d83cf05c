// Internal launch interface between engine.hip and the kernel translation units.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "plan_tables.h"

// ------------------------------------------------------------------------------------------------
// Implicit-GEMM convolution (conv_igemm.hip)
//
// D[cout][pixel] = sum_k W[cout][k] * X[pixel][k],  k = (tap, cin).   NHWC bf16 activations live
// in spatially zero-padded planes so the im2col gather never needs bounds checks; output "rows"
// (pixels) are described by a row table, which lets one launch cover all pyramid levels, MC
// samples and images at once.
// ------------------------------------------------------------------------------------------------
// struct RowEnt (32 B per output pixel) and XR_EXT_ROWS: plan_tables.h (plain C++, shared with the host sanitizer build)

struct ConvGroup {         // element type of in / w / res / out_relu: bf16 (default) or fp32 (fp32 precision mode)
    const void* in;        // activations
    const void* w;         // [Cout_pad][taps][Cin]
    const float* bias;     // fp32 [Cout_pad]
    void* out;             // activation type, or fp32 with CONV_OUT_F32
    const void* res;       // residual or nullptr
    void* out_relu;        // optional second output relu(out) (P6 -> P7 input) or nullptr
    int32_t in_coff;       // channel offset inside an input pixel
    int32_t layer_id;      // dropout stream id (head*4 + layer)
    // Fused head output conv (1x1, 256 -> cout2) applied to this group's finished output tile while it
    // is still in LDS; the tile itself is then not stored (nothing else reads the last tower layer).
    const void* w2;        // bf16 [cout2 rounded up to 32][256] or nullptr
    const float* bias2;    // fp32, padded like w2
    float* out2;           // fp32 [rows][out2_cstride]; row index = RowEnt.pad0
    int32_t cout2;         // real output channels (A*C, A*4, A*10)
    int32_t out2_cstride;
    // MC aggregation fused behind the 1x1 (inference_utils.py:31-60,220-244): when agg_kind != 0 the per-sample head outputs of
    // a tile never leave the CU -- the tile holds ALL agg_n samples of its pixels (row = pixel_slot * agg_n + sample, see the
    // "aggregated" row tables of engine.hip), the fp32 outputs go through LDS and one thread per (pixel, anchor) reduces over
    // the samples: AGG_CLS sum_n softmax(logits) [B,A,C]; AGG_BOX Welford mean + M2 of the decoded boxes [B,A,16] (mean 4, lower
    // triangle of M2 10, 2 pad); AGG_COV sum_n of the covariance parameters [B,A,10].  out2 is then unused.
    int32_t agg_kind;
    int32_t agg_n;         // MC samples per pixel in a tile
    int32_t agg_P;         // pixels per image (RowEnt.pad0 = (image * agg_n + sample) * agg_P + pixel)
    int32_t agg_C;         // classes (AGG_CLS) -- 4 or 8
    float* agg_out;
    const float* anchors;  // [A,4] (v,u,h,w): box decode of AGG_BOX
    // Bottleneck chain (ResNet stages 2-3, bf16 inference; 64x128 / 128x128 tiles whose cout tile holds ALL couts of this 3x3 conv):
    // the finished tile of this group's conv -- a block's `2b` -- stays in LDS and feeds the block's 1x1 expansion `2c`
    // (ch_w2: [ch_c2][cout] bf16, + bias + shortcut ch_res + ReLU -> plane ch_out) and, on that result, the NEXT block's 1x1
    // reduction `2a` (ch_w3: [cout][ch_c2] bf16, + bias + ReLU -> plane ch_out3; optional).  All planes share this conv's output
    // geometry (pixel index = RowEnt.out_off); the 2b output itself is not stored.  feature_extractor.py:195-213,283-309.
    const void* ch_w2; const float* ch_b2; const void* ch_res; void* ch_out; int32_t ch_c2;
    const void* ch_w3; const float* ch_b3; void* ch_out3;
    // Pointwise kernel, dual form (conv_pointwise.hip, NEXT = 2): ch_w3 / ch_b3 / ch_out3 describe a second 1x1 convolution (64 couts,
    // + bias + ReLU) of the SAME input tile -- a ConvBlock's `2a` riding on its projection shortcut `branch1` -- instead of one of
    // the finished output tile.
    int32_t ch_dual;
    // f16mx precision (ConvArgs.mx != 0): 1 = this group's output rows leave in the tower format "hx" (header of conv_igemm.hip: per 64
    // channels a 128-byte H chunk of 64 f16 hi and a 128-byte X chunk of four 32-byte slots -- 16 channels as 32 e2m3 (fp6) elements
    // {hi6, lo6'}, lo' = lo * 2^11, 24 bytes + the block's E8M0 scale at byte 28) for the next tower layer; 0 = (hi, lo) bf16 pairs (what
    // the fused 1x1 + aggregation and every other consumer read); 2 = "h4" rows (f16mx4: e2m1 cross terms, scale bytes in chunk 6)
    int32_t out_hx;
};
enum : int32_t { AGG_NONE = 0, AGG_CLS = 1, AGG_BOX = 2, AGG_COV = 3 };

enum : int32_t { CONV_RELU = 1, CONV_DROPOUT = 2, CONV_OUT_F32 = 4,
                 CONV_ACCUM = 8,       // with CONV_OUT_F32: out += result (input gradients of 1x1 layers accumulate in place)
                 CONV_NT_OUT = 16 };   // bf16 outputs stored non-temporally (set by launch_conv_igemm, BOD_NT_STORES)

struct ConvArgs {
    ConvGroup g[3];
    const RowEnt* rows;
    int32_t M;             // number of output pixels (rows)
    int32_t taps, KW;      // KH*KW, KW
    int32_t cin;           // channels reduced per tap (multiple of 64)
    int32_t in_cstride;    // channels per input pixel
    int32_t cout_pad;      // multiple of the cout tile
    int32_t cout_valid;    // real output channels
    int32_t out_cstride;   // channels per output pixel
    int32_t res_cstride;
    int32_t flags;
    int32_t fan_count;     // >1: write fan_count dropout variants (sample n at out_off + n*fan_stride)
    int32_t fan_stride;
    uint32_t seed_lo, seed_hi;
    uint32_t drop_threshold;
    float drop_scale;
    uint32_t image_base;   // global id of image 0 of the batch
    int32_t groups;
    int32_t variant;       // 0 = production kernel; >0 = ablation / experimental builds (tests/tools)
    // Activation row reuse (3x3 stride-1 convs, 256x256 tiles): `rows` is then organised in tiles of 256
    // slots (invalid slots have out_off = -1) and `ext` lists, per tile, the XR_EXT_ROWS "extended" input
    // pixels {pixel index for ky = 0, plane pitch}: every run of x-adjacent output pixels contributes its
    // pixels plus one on either side, so the three kx taps read the SAME staged rows at offsets 0/1/2
    // (RowEnt.pad1 = a pixel's extended-row index) and activations are staged once per (chunk, ky).
    const int2* ext;
    int32_t xreuse;        // 0 off; 2 on: compact-state loop, 32-bit byte offsets against the tile's first extended row
                           // (ext[tile * 320] must be the tile's smallest pixel index); 1: first-generation loop, 64-bit pointers
    // Split-K (small-M layers: P6, the stage-4/5 layers at batch 1): ksplit > 1 splits the input-channel chunks over
    // ksplit workgroups per tile (blockIdx.z = group * ksplit + split); every split writes its raw fp32 accumulators to
    // partial[(z * M + m) * cout_pad + co] and launch_conv_igemm runs the reduce kernel (sum, bias, residual, ReLU,
    // bf16 store through the row table) behind it.  cin / 64 must be divisible by ksplit.
    int32_t ksplit;
    float* partial;
    // Optional device copy of {seed_lo, seed_hi, image_base}: when set it overrides the three by-value fields, so a launch
    // recorded in a hipGraph (training step) can be replayed with a new dropout seed / image id (nullptr in inference).
    const uint32_t* dyn_rng;
    uint32_t sample_base;  // added to every MC sample index before it enters the dropout counter (sample sharding)
    // bf16x3 precision mode: activations / weights / residuals are (hi, lo) bf16 pairs, 32 hi then 32 lo per 64-slot group
    // (conv_igemm.hip); cin, in_cstride, res_cstride, in_coff and -- for non-fp32 outputs -- out_cstride count SLOTS (2 per
    // channel), cout_pad / cout_valid stay in channels.
    int32_t split;
    // Output plane geometry of a plane -> plane convolution whose row table is ordered (image, y, x): rows per image and pixels per
    // row (0 = unknown).  The sliding-window 3x3 kernel (conv_pointwise.hip) walks column strips with it.
    int32_t plane_h, plane_w;
    // Compute units the launch may fill (0 = unknown: the device's count).  The engine sets it per launch -- the whole chip, or the
    // CU partition of the stream the launch goes to (bod_config.pipeline_overlap) -- and the planner compares workgroup counts with it.
    int32_t n_cu;
    // Fan-out launch (first tower layer): the first workgroup of CU slot k of every XCD starts (k & 3) * stagger_ticks 100-MHz ticks
    // late, so that the ten-fold store bursts of the CUs' epilogues interleave with other CUs' main loops instead of all hitting HBM
    // at once (conv_igemm.hip).  0 = off.
    int32_t stagger_ticks;
    // Fan-out launch, order of the (pixel tile, head) work items inside an XCD's range (round 6, BOD_FAN_CHUNK=T, A/B): 0 = interleaved
    // (tile 0 heads 0 1 2, tile 1 ...: the three heads of a tile share its input rows through L2); T > 0 = chunks of T tiles, head-major
    // inside a chunk (T tiles of head 0, the same T tiles of head 1, ...: one head's weights at a time in L2)
    int32_t fan_chunk;
    // f16mx precision, head towers on the row-reuse loop (with `split`: same slot counts, same 1 KiB pixel rows): 1 = activations and
    // weights in the hx format, one f16 product + half a block-scaled e2m3 (fp6) product (the two cross terms) per multiplication; 2 = (hi, lo)
    // bf16 pairs in (the bf16x3 loop), epilogue able to write hx rows (first tower layer); 3 = activations and weights in the h4 format
    // (f16mx4: the cross terms as e2m1 products of twice the channels; the weight buffer carries a compact copy of its scale bytes behind
    // the 256 rows).  0 = off.
    int32_t mx;
    int32_t mx_loader;     // f16mx loop: which waves issue the weight pieces (conv_igemm.hip; 0 all, 1 lower four, 2 upper four)
};

// hipFuncSetAttribute is per device: remember which devices of this process already have the attribute
struct PerDeviceOnce {
    bool done[64] = {false};
    bool* slot() { int d = 0; if (hipGetDevice(&d) != hipSuccess || d < 0 || d >= 64) d = 0; return &done[d]; }
};

// Compute units a launch may fill: ConvArgs.n_cu when the engine set it (the whole chip, or a CU-partitioned stream's share), else the
// current device's count (hipDeviceProp_t::multiProcessorCount, cached per device).  The planner's rules compare workgroup counts with it.
int device_cu_count();
inline int launch_cus(const ConvArgs& a) { return a.n_cu > 0 ? a.n_cu : device_cu_count(); }
hipError_t launch_conv_igemm(const ConvArgs& a, hipStream_t s);
bool conv_igemm_uses_full_cout_tile(const ConvArgs& a);   // true => 256-wide cout tile => 1x1 fusion possible
bool conv_igemm_uses_big_tile(const ConvArgs& a);         // true => the 256x256 tile (any cout); false => 128-pixel tiles
void conv_igemm_phase_cycles(unsigned long long* out16, bool reset);   // instrumented build (variant 90)
hipError_t launch_conv_igemm_f32(const ConvArgs& a, hipStream_t s);      // conv_igemm_f32.hip (fp32 planes / weights)
// f16mx precision: (hi, lo) bf16 pair rows [npix][2C] -> hx rows [npix][4C bytes] (conv_igemm.hip; C a multiple of 64)
hipError_t launch_pairs_to_hx(const void* in, void* out, long npix, int C, hipStream_t s, int fmt = 1);          // fmt 1: hx rows, 2: h4 rows (f16mx4)

// ------------------------------------------------------------------------------------------------
// Stem + pooling + small elementwise (aux_kernels.hip)
// ------------------------------------------------------------------------------------------------
// 7x7 s2 VALID conv (fp32 image, fp32 folded weights [7][7][3][64]) + bias + ReLU -> bf16 [B,oh,ow,64]
hipError_t launch_stem_conv(const float* img, const float* w, const float* bias, void* out, int out_f32,
                            int B, int H, int W, int oh, int ow, hipStream_t s);
// ZeroPadding2D((1,2)) + MaxPool 3x3 s2 VALID on [B,ih,iw,64] -> padded-plane output.  mode 0: bf16 -> bf16; 1: fp32 -> fp32;
// 2: fp32 -> (hi, lo) bf16 pairs (bf16x3 precision)
// stem + zero-pad + max-pool in one kernel (bf16 inference, stem rows of <= 256 pixels; aux_kernels.hip)
bool stem_pool_fused_applies(const float* img, int B, int W, int ow, int n_cu);
// split = 1: the (hi, lo) precisions -- three bf16 products, fp32 pooling, pooled pixels stored as pairs (aux_kernels.hip)
hipError_t launch_stem_pool_fused(const float* img, const float* w, const float* bias, void* pooled, int split, int B, int H, int W, int oh,
                                  int ow, int ph, int pw, int pool_pitch, int pool_plane, hipStream_t s);
hipError_t launch_stem_pool(const void* in, void* out, int mode, int B, int ih, int iw, int oh, int ow,
                            int out_pitch, int out_plane, hipStream_t s);

// ------------------------------------------------------------------------------------------------
// Bayesian post-processing (post_kernels.hip)
// ------------------------------------------------------------------------------------------------
struct PostCfg {
    int32_t B, N, A, C, draws;
    int32_t use_full_covar, has_covar, dirichlet, gaussian_iso, ranking_method;
    float iso_var;
    float kitti_sh, kitti_sw;       // 0 => off
    uint32_t seed_lo, seed_hi, image_base;
    int32_t aggregated;             // 1: the MC statistics come from the conv epilogue (PostBuffers.agg_*), not from raw [B,N,A,.]
};

struct PostBuffers {
    // inputs
    const float* cls;       // [B,N,A,C]
    const float* box;       // [B,N,A,4]
    const float* cov;       // [B,N,A,10]
    const float* anchors;   // [A,4]
    // PostCfg.aggregated: per-anchor MC statistics written by the last tower layers' epilogues (ConvGroup.agg_kind)
    const float* agg_cls;   // [B,A,C]   sum over samples of softmax(logits)
    const float* agg_box;   // [B,A,16]  Welford mean[4], lower triangle of M2[10], 2 pad
    const float* agg_cov;   // [B,A,10]  sum over samples of the covariance parameters
    // dense per-anchor scratch
    uint8_t* keep;          // [B,A]
    float* d_counts;        // [B,A,C]   sampled counts (likelihood)
    // compacted outputs, capacity A per image
    int32_t* block_counts;  // [B, nblocks]
    int32_t* num_kept;      // [B]
    float* counts;          // [B,A,C]   Dirichlet posterior counts
    float* score;           // [B,A,C]
    float* means;           // [B,A,4]
    float* covs;            // [B,A,16]
    float* ranking;         // [B,A]
    float* corners;         // [B,A,4]
    int32_t* anchor_index;  // [B,A]
};

hipError_t launch_posterior(const PostCfg& c, const PostBuffers& b, hipStream_t s);
hipError_t launch_joint_entropy_rank(const PostCfg& c, const PostBuffers& b, hipStream_t s);
hipError_t launch_validation_post(const PostCfg& c, const PostBuffers& b, hipStream_t s);   // validation_utils.py:10-77

struct NmsArgs {
    int32_t B, A;                 // per-image capacity A
    const int32_t* num_kept;      // [B]
    const float* corners;         // [B,A,4]
    const float* ranking;         // [B,A]
    float* work_scores;           // [B, A rounded up to 512] scratch (used when M exceeds the LDS capacity)
    int32_t* work_begin;          // same shape
    int32_t* selected;            // [B,max_out]
    int32_t* num_selected;        // [B]
    int32_t max_out;
    float iou_thr, sigma;
    int32_t variant;
};
hipError_t launch_nms(const NmsArgs& a, hipStream_t s);

struct ClusterArgs {
    int32_t B, A, C, max_out;
    const int32_t* num_kept;
    const int32_t* selected;      // [B,max_out]
    const int32_t* num_selected;  // [B]
    const float* corners;         // [B,A,4]
    const float* counts;          // [B,A,C]
    const float* means;           // [B,A,4]
    const float* covs;            // [B,A,16]
    float thr;
    // optional caller-supplied affinity (bayes_od_clustering's `affinity_matrix`, inference_utils.py:316): for image
    // `affinity_img`, row k = affinity_matrix[:, centre_k] ([max_out][A] floats); nullptr = IoU of the means on the fly
    const float* affinity; int32_t affinity_img;
    // outputs [B,max_out,...]
    float* out_scores; float* out_means; float* out_covs; float* out_counts;
};
hipError_t launch_cluster_fuse(const ClusterArgs& a, hipStream_t s);
hipError_t launch_iou_matrix(const float* corners, int M, float* out, hipStream_t s);
// detection records [B][K][1 + 4 + 16 + 2C] of the multi-GPU gather (zero rows beyond num[b])
hipError_t launch_pack_records(const int32_t* num, const float* scores, const float* means, const float* covs, const float* counts,
                               float* rec, int B, int K, int C, hipStream_t s);

struct PreprocArgs {
    const uint8_t* src;        // [B, sh, sw, 3] uint8 RGB
    float* dst;                // [B, H, W, 3] fp32 BGR, mean-subtracted
    int32_t B, sh, sw;         // source size
    int32_t rh, rw;            // size after the (optional) aspect-preserving bilinear resize (= sh, sw without)
    int32_t H, W;              // network input size
    int32_t crop_y, crop_x, pad_y, pad_x, vis_h, vis_w;   // tf.image.resize_with_crop_or_pad geometry
    int32_t resize;
    float scale_y, scale_x;    // sh/rh, sw/rw as float32
    float mean[3];             // RGB order
};
hipError_t launch_preprocess(const PreprocArgs& a, hipStream_t s);

// ------------------------------------------------------------------------------------------------
// Loss forward (loss_kernels.hip)
// ------------------------------------------------------------------------------------------------
struct LossArgs {
    int32_t B, A, C;
    int32_t do_cls;          // focal classification term
    int32_t reg_kind;        // 0 none, 1 'regression', 2 'regression_var', 3 'regression_covar'
    float label_smoothing;
    const float* cls; const float* cls_t;       // [B,A,C]
    const float* box; const float* box_t;       // [B,A,4]
    const float* cov;                           // [B,A,10] fill_triangular parameters
    const float* anchors;                       // [A,4]
    const uint8_t* pos; const uint8_t* neg;     // [B,A]
};
hipError_t launch_loss(const LossArgs& a, float* partial, int nblocks, hipStream_t s);
hipError_t launch_loss_reduce(const float* partial, int nblocks, float* sums4, hipStream_t s);
hipError_t launch_loss_backward(const LossArgs& a, const float* sums4, float w_cls, float w_reg, float* dcls, float* dbox, float* dcov,
                                hipStream_t s);

// ------------------------------------------------------------------------------------------------
// Streaming pointwise (1x1) convolution for reductions of <= 256 channels (conv_pointwise.hip); launch_conv_igemm routes eligible
// launches there (BOD_POINTWISE=0: off)
bool conv_pointwise_eligible(const ConvArgs& a);
bool conv_pointwise_can_fuse_next(const ConvArgs& a);      // plan time: may a 64 -> 256 expansion carry the next block's 2a (ch_w3)?
bool conv_pointwise_can_fuse_dual(const ConvArgs& a);      // plan time: may a 64 -> 256 projection shortcut carry its own block's 2a (ch_w3 + ch_dual)?
hipError_t launch_conv_pointwise(const ConvArgs& a, hipStream_t s);
// Sliding-window 3x3 stride-1 SAME convolution, 64 -> 64 channels (ResNet stage 2's `2b`; conv_pointwise.hip, BOD_SLIDE3X3=0: off)
bool conv_slide3x3_eligible(const ConvArgs& a);
hipError_t launch_conv_slide3x3(const ConvArgs& a, hipStream_t s);

// Training-step building blocks (train_kernels.hip)
// ------------------------------------------------------------------------------------------------
hipError_t launch_gather_transpose(const void* in, const RowEnt* rows, void* out, int M, int Kpad, int C, int cstride,
                                   int taps, int KW, bool append_ones_row, hipStream_t s, bool f32 = false);
hipError_t launch_fill_row_bf16(void* row, int n_set, int n_total, float value, hipStream_t s);

struct FoldArgs {
    const float* kernel;        // master [taps][cin][cout]
    const float* bias;          // master [cout] or nullptr
    const float *gamma, *beta, *mean, *var;   // BatchNorm (nullptr: no BN)
    float eps;
    int32_t taps, cin, cout, cout_pad;
    uint16_t* w_fwd;            // [cout_pad][taps][cin] bf16 (or nullptr)
    float* w_fwd32;             // stem: fp32 [taps*cin][cout] (or nullptr)
    uint16_t* w_bwd;            // [(tap, ci)][cout_pad] bf16 (or nullptr)
    uint16_t* w_flip;           // [cin][taps, flipped][cout_pad] bf16: the input gradient of a stride-1 SAME layer as a convolution (or nullptr)
    float* b_fwd;               // folded bias [cout_pad]
    int32_t f32;                // fp32 training handle: the three packings above hold float (same layouts)
};
struct ActBwdArgs {
    const float* dout;          // gradient of the layer output, laid out like the output buffer
    const uint16_t* out_bf16;   // stored output (mask source) or nullptr (no activation: fp32 raw head outputs)
    const RowEnt* rows;
    float* dres;                // gradient buffer of the residual input or nullptr
    uint16_t* dz;               // dense [M][cout_pad] bf16
    uint16_t* dzp;              // the same values in the output plane's own (zero-bordered) layout, or nullptr
    uint16_t* dzt;              // the same values transposed, [cout_pad][Kpad] (columns M..Kpad-1 zero), or nullptr
    int32_t M, cout, cout_pad, out_cstride, res_cstride, Kpad;
    float scale;                // dropout keep scale (1 without dropout)
    int32_t f32;                // fp32 training handle: out_bf16 / dz / dzp hold float, dzt is not written
};
struct UnfoldArgs {
    const float* dwp;           // [(taps*cin) + 1][cout]: folded-weight gradient, last row = folded-bias gradient
    const float* kernel; const float* bias;
    const float *gamma, *mean, *var;
    float eps;
    int32_t taps, cin, cout;
    float *d_kernel, *d_bias, *d_gamma, *d_beta;
    float* dot;                 // [cout] zero-initialised scratch for the gamma dot products (left zeroed)
    float l2;                   // Keras l2(rate) on the kernel: d_kernel += 2 rate K, *l2_loss += rate sum K^2 (0 = none)
    float* l2_loss;
};
hipError_t launch_fold_pack(const FoldArgs& a, hipStream_t s);
hipError_t launch_fold_pack_all(const FoldArgs* device_array, int count, long max_elems, hipStream_t s);
// *wrote_transpose tells the caller whether a.dzt was produced by the same launch (tile form) or still needs a transpose pass
hipError_t launch_act_backward_gather(const ActBwdArgs& a, hipStream_t s, bool* wrote_transpose = nullptr);
hipError_t launch_relu_merge(const float* dout_relu, const void* out, float* dout, long n, hipStream_t s, bool f32 = false);
hipError_t launch_col2im(const float* dxcol, const RowEnt* rows, float* din, int M, int taps, int KW, int cin, int in_cstride, hipStream_t s);
hipError_t launch_stem_pool_backward(const void* stem_out, const float* dpool, void* dz, int B, int ih, int iw, int oh, int ow, int pool_pitch,
                                     int pool_plane, hipStream_t s, bool f32 = false);
hipError_t launch_unfold_grad(const UnfoldArgs& a, hipStream_t s);
hipError_t launch_l2_grad(const float* w, float* g, long n, float rate, float* loss_acc, hipStream_t s);
hipError_t launch_sumsq(const float* g, long n, float* acc, float* partial1024, hipStream_t s);
hipError_t launch_adam(float* w, const float* g, float* m, float* v, long n, const float* sumsq, float clip, const float* lr_t, float beta1, float beta2,
                       float eps, hipStream_t s);
hipError_t launch_f32_to_bf16(const float* in, void* out, long n, hipStream_t s);
hipError_t launch_make_dgrad_rows(const RowEnt* fwd, RowEnt* out, int M, hipStream_t s);
