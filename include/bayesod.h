/*
 * bayesod.h -- C ABI of libbayesod_hip.so, the MI355X (gfx950) BayesOD inference hot path.
 *
 * The reference (asharakeh/bayes-od-rc) is pure Python/TensorFlow and has no FFI or plugin
 * interface; its boundary for this path is the Python call surface.
 * Each entry point below names the reference interface it stands behind (paths relative to
 * the reference root).  Plain pointers and sizes only; no torch / HIP types.  All host
 * buffers are caller-owned; the library owns every device allocation.  A handle is not
 * thread-safe: one handle per GPU / host thread.  Every call returns BOD_OK or an error code;
 * bod_last_error() gives the message.  Empty results (M = 0 / K = 0) are not errors
 * (src/retina_net/experiments/run_inference.py:147-161).
 */
#ifndef BAYESOD_H
#define BAYESOD_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct bod_context* bod_handle;

typedef enum {
    BOD_OK = 0,
    BOD_ERR_INVALID_ARG = 1,   /* maps to ValueError in the Python mirror            */
    BOD_ERR_HIP = 2,           /* HIP runtime failure (message holds hipGetErrorString) */
    BOD_ERR_OOM = 3,
    BOD_ERR_NOT_READY = 4,     /* weights / anchors not loaded, or stage order violated */
    BOD_ERR_NO_DEVICE = 5
} bod_status;

enum { BOD_RANK_SCORE = 0, BOD_RANK_JOINT_ENTROPY = 1 };
enum { BOD_NMS_VARIANT_A = 0, BOD_NMS_VARIANT_B = 1 };      /* SURVEY.md App. A.8 */
enum { BOD_HEAD_CLS = 0, BOD_HEAD_REG = 1, BOD_HEAD_COV = 2 };
enum { BOD_PRECISION_BF16 = 0, BOD_PRECISION_FP32 = 1, BOD_PRECISION_BF16X3 = 2, BOD_PRECISION_F16MX = 3, BOD_PRECISION_F16MX4 = 4 };

/* Mirrors model_config / testing_config of src/retina_net/configs/retinanet_bdd_covar.yaml
 * (:61-143) plus the geometry the reference derives at run time. */
typedef struct {
    int32_t device;              /* HIP device ordinal (--gpu_device, run_inference.py:267)           */
    int32_t image_h, image_w;    /* network input size                                                */
    int32_t batch;               /* images per call (reference: 1, run_inference.py:68)               */
    int32_t mc_samples;          /* model_config.mc_dropout_samples (yaml :66)                        */
    int32_t num_classes;         /* header.num_classes + 1 background (multitask_headers.py:86-88)    */
    int32_t anchors_per_location;/* len(scales)*len(aspect_ratios) (config_utils.py:81-86)            */
    int32_t min_level, max_level;/* anchor_generator.layers (yaml :53) -> 3..7                         */
    float   dropout_rate;        /* header.dropout_rate (yaml :82)                                    */
    int32_t use_full_covar;      /* testing_config.use_full_covar (yaml :125)                         */
    int32_t dirichlet_non_informative; /* bayes_od_config.dirichlet_prior.type == 'non_informative'   */
    int32_t gaussian_isotropic;  /* bayes_od_config.gaussian_prior.type == 'isotropic'                */
    float   isotropic_variance;  /* yaml :141                                                         */
    int32_t ranking_method;      /* BOD_RANK_*  (yaml :133)                                           */
    int32_t nms_max_output_size; /* nms_config.max_output_size (yaml :128)                            */
    float   nms_iou_threshold;   /* nms_config.iou_threshold, also the clustering affinity threshold
                                    (run_inference.py:148-149)                                        */
    float   nms_soft_sigma;      /* nms_config.soft_nms_sigma                                         */
    int32_t nms_variant;         /* BOD_NMS_VARIANT_*                                                 */
    int32_t num_categorical_draws; /* Categorical.sample(30) (inference_utils.py:42)                  */
    int32_t has_covar_head;      /* 'regression_covar' in output_names (retinanet_model.py:50)        */
    float   kitti_scale_h, kitti_scale_w; /* orig/net size; 0 => dataset != 'kitti'
                                    (inference_utils.py:147-167)                                      */
    int32_t precision;           /* BOD_PRECISION_BF16 (default, throughput path: bf16 storage + bf16 MFMA),
                                    BOD_PRECISION_FP32 (fp32 storage + exact-fp32 MFMA: the reference's fp32
                                    arithmetic end to end; ~1/10 of the speed) or BOD_PRECISION_BF16X3 (every value a
                                    (hi, lo) bf16 pair, products hi*hi + hi*lo + lo*hi on the bf16 MFMA with fp32
                                    accumulation: within 1e-3 of the float64 reference END TO END at about a third
                                    of the bf16 rate -- the parity mode of the throughput path) or
                                    BOD_PRECISION_F16MX (the parity mode with the head towers -- 80 % of its time --
                                    on one f16 product + half a block-scaled e2m3 product per multiplication instead of
                                    three bf16 products: x = f16 hi + lo, hi*hi exact, the two cross terms on the MX
                                    pipe; everything else as in BF16X3.  END TO END against the fp32 CPU forward with the same
                                    dropout masks: raw head outputs <= 2e-4 of |ref| + rms(ref); final detections, apart from
                                    at most one discrete flip per frame (a categorical draw at a CDF edge, a cluster member at
                                    the affinity threshold): box means <= 2e-5, scores <= 1e-7, covariance entries <= 1e-3 of
                                    |entry| + rms(matrix) -- bench.py's `meets_1e-3` states it per run) or
                                    BOD_PRECISION_F16MX4 (F16MX with the cross terms as block-scaled e2m1 (fp4) products of
                                    twice the channels: three quarters of the tower bytes and K-tiles, ~4x F16MX's
                                    rounding error: raw outputs 6e-4, boxes and scores inside 1e-3, fused covariance
                                    entries up to 4e-3 -- an opt-in mode BETWEEN bf16 and F16MX, not a parity mode)    */
    int32_t mc_sample_base;      /* index of this handle's first MC sample in the dropout RNG streams (default 0).
                                    A handle with mc_samples = n and base = r*n computes samples r*n .. r*n+n-1 of
                                    a larger ensemble bit-identically: the MC-sample-sharded multi-GPU mode
                                    (second mode).  May change on a live handle.        */
    int32_t mc_ensemble_size;    /* total MC samples of the ensemble this handle contributes to; 0 = mc_samples.
                                    MC dropout is on iff max(mc_ensemble_size, mc_samples) > 1
                                    (retinanet_model.py:74-77 decides on the ensemble size), so a rank holding a
                                    single sample of a sharded ensemble still applies its dropout masks.       */
    int32_t training;            /* 1: the handle also runs training steps (bod_train_step):
                                    every layer keeps its activation, dropout is on with mc_samples = 1
                                    (retinanet_model.py:113-147, training branch), fp32 master weights, gradients
                                    and Adam moments live on the device.  precision = BOD_PRECISION_BF16 (the product) or
                                    BOD_PRECISION_FP32 (the gradient-verification mode: same executor on fp32 tensors and
                                    the exact-fp32 MFMA kernel; meets float64 autograd element-wise).           */
    int32_t backbone_depth;      /* 0 / 50: ResNet-50, the reference's only backbone (feature_extractor.py:6-9).  101: stage 4 with
                                    1 ConvBlock + 22 IdentityBlocks (layer names res4a .. res4w) -- BASELINE config 5's
                                    "ResNet-101", which has no counterpart in the reference.             */
    int32_t pipeline_overlap;    /* 1: bod_infer_async overlaps the memory-bound front of batch i+1 (stem, backbone, FPN) with the
                                    MFMA-bound back of batch i (fan-out layer, towers, posterior) on two CU-partitioned streams
                                    (hipExtStreamCreateWithCUMask: the front owns the last 4 CU slots of every XCD, 32 of 256 CUs,
                                    the back the other 224; the pyramid is double-buffered).  EXPERIMENTAL: measured 8.5 % SLOWER
                                    than one stream (the chip is power-bound: DESIGN.md 8.3), and it runs kernels of the library
                                    beside each other by design, which until round 6 was not reproducible bit for bit (DESIGN.md
                                    8.4: packed fp32 results of a wave are corrupted in lanes 48-63 beside a wave that interleaves
                                    VALU work with MFMAs -- a gfx950 erratum; the library is built without packed fp32
                                    instructions since, and no difference has been seen after that).  bod_create refuses the
                                    mode unless BOD_OVERLAP_EXPERIMENTAL=1 is set in the environment.  Every other entry point
                                    keeps the whole chip.  Inference handles only.  0 (default): one stream: no kernel of the
                                    library runs beside another.                                                  */
    int32_t reserved[2];
} bod_config;

/* Sizes the caller needs to allocate host buffers. */
typedef struct {
    int32_t num_pixels;          /* P: sum over levels of h*w            */
    int32_t num_anchors;         /* A = P * anchors_per_location         */
    int32_t level_h[8], level_w[8];
    int32_t num_levels;
    int32_t max_detections;      /* nms_max_output_size                  */
    int64_t device_bytes;        /* total HBM the handle holds           */
} bod_sizes;

const char* bod_version(void);
const char* bod_last_error(bod_handle h);         /* h may be NULL: last create() failure */

/* RetinaNetModel(model_config)  (src/retina_net/models/retinanet_model.py:19-65) */
bod_status bod_create(const bod_config* cfg, bod_handle* out);
bod_status bod_destroy(bod_handle h);
bod_status bod_query_sizes(bod_handle h, bod_sizes* out);
/* Change the testing_config-derived fields (use_full_covar, priors, ranking, nms_*, kitti scale,
 * num_categorical_draws) of a live handle; geometry / batch / mc_samples / heads must be unchanged. */
bod_status bod_update_config(bod_handle h, const bod_config* cfg);

/* ckpt.restore(...) (run_inference.py:120): one call per Keras variable, fp32 host data.
 * kind: 0 conv kernel HWIO [kh,kw,cin,cout]; 1 conv bias [cout]; 2..5 BN gamma/beta/mean/var.
 * name: Keras layer name ('conv1', 'bn_conv1', 'res2a_branch2a', 'C5_reduced', 'P3',
 * 'pyramid_classification_0', 'pyramid_classification', 'pyramid_regression', 'pyramid_cov' ...;
 * feature_extractor.py:154-155,231-232; feature_decoder.py:28-132; multitask_headers.py:30-314). */
bod_status bod_load_weight(bod_handle h, const char* name, int32_t kind,
                           const int64_t* shape, int32_t ndim, const float* data);
/* Folds frozen BN (feature_extractor.py:108 ... training=False), packs bf16 OHWI, uploads. */
bod_status bod_finalize_weights(bod_handle h);

/* sample_dict['anchors'] (src/core/constants.py:50; bdd_dataset_handler.py:183-186): [A,4] (v,u,h,w) */
bod_status bod_set_anchors(bod_handle h, const float* anchors_vuhw, int32_t num_anchors);

/* model(image, train_val_test='testing')  (retinanet_model.py:67-112).
 * images: [batch,H,W,3] fp32 normalised BGR (sample_dict['image_normalized']), host pointer or,
 * if images_on_device != 0, a device pointer already resident in HBM.
 * seed / first_image_id key the Philox dropout + categorical streams (DESIGN.md RNG contract).
 * On a handle created with training = 1 this is model(image, train_val_test='training') (retinanet_model.py:113-147):
 * one sample with dropout ON, batch-norm frozen, the current (trained) master weights. */
bod_status bod_forward(bod_handle h, const float* images, int32_t images_on_device,
                       uint64_t seed, uint32_t first_image_id);
/* prediction_dict tensors (src/core/constants.py:61-63), copied to host:
 * cls [batch,N,A,C], box [batch,N,A,4], covar params [batch,N,A,10] (pre fill_triangular).
 * Any pointer may be NULL. */
bod_status bod_get_raw(bod_handle h, float* cls, float* box, float* covar_params);
/* Replace the head outputs with caller-supplied ones (stage-level parity tests). */
bod_status bod_set_raw(bod_handle h, const float* cls, const float* box, const float* covar_params);
/* Backbone / FPN taps for parity tests: which = 0..4 -> p3..p7 ([batch,h,w,256] fp32 out). */
bod_status bod_get_pyramid(bod_handle h, int32_t level_index, float* out);

/* bayes_od_inference after the model call (src/retina_net/experiments/inference_utils.py:25-202):
 * decode, softmax, mean over MC, categorical sampling, filter, per-anchor mean / 4x4 covariance,
 * aleatoric L D L^T, mixing, Dirichlet + Gaussian prior fusion, ranking; compacted in anchor order. */
bod_status bod_posterior(bod_handle h, uint64_t seed, uint32_t first_image_id);
/* validation_utils.post_process_predictions (:10-77), the deterministic validation path: softmax of MC sample 0
 * of the raw class outputs, anchors whose arg-max class is background dropped, candidates ranked by their top
 * score.  Fills the same compacted buffers as bod_posterior (score = counts = softmax row, means = decoded
 * box, covs = 0), so bod_nms / bod_get_posterior / bod_get_nms follow as usual. */
bod_status bod_validation_post(bod_handle h);
/* Per image results of bod_posterior. num_kept[batch]; the arrays are [M,...] for image_index.
 * counts/score [M,C], means [M,4], covs [M,16], ranking [M], anchor_index [M]. NULLs skipped. */
bod_status bod_get_num_kept(bod_handle h, int32_t* num_kept);
bod_status bod_get_posterior(bod_handle h, int32_t image_index, float* counts, float* score,
                             float* means, float* covs, float* ranking, int32_t* anchor_index);
/* Inject a posterior (stage-level parity of NMS / clustering): arrays as above, M rows. */
bod_status bod_set_posterior(bod_handle h, int32_t image_index, int32_t m, const float* counts,
                             const float* means, const float* covs, const float* ranking);

/* tf.image.non_max_suppression_with_scores on vuhw_to_vuvu(means) (inference_utils.py:204-212). */
bod_status bod_nms(bod_handle h);
bod_status bod_get_nms(bod_handle h, int32_t image_index, int32_t* indices, int32_t* num_selected);
/* Inject cluster centres (the `cluster_centers` argument of bayes_od_clustering, :289). */
bod_status bod_set_nms(bod_handle h, int32_t image_index, const int32_t* indices, int32_t num_selected);
/* box_utils.bbox_iou_vuvu(corners, corners) (inference_utils.py:214-215): [M,M] fp32 to host. */
bod_status bod_get_iou_matrix(bod_handle h, int32_t image_index, float* iou);

/* bayes_od_clustering (inference_utils.py:285-364) on the device, affinity = the IoU above,
 * threshold = nms_iou_threshold; covariances x70. */
bod_status bod_cluster_fuse(bod_handle h);
/* The `affinity_matrix` argument of bayes_od_clustering (inference_utils.py:290,316) when the caller's affinity is
 * not the IoU of the posterior means: centre_columns [k][m] holds, for each of the image's k cluster centres (in
 * bod_set_nms / bod_get_nms order), the column affinity_matrix[:, centre] over the image's m boxes.  The NEXT
 * bod_cluster_fuse tests `centre_columns[k][i] > nms_iou_threshold` for that image instead of evaluating the IoU on
 * the fly, then the columns are dropped (one-shot).  k and m must equal the image's centre and box counts. */
bod_status bod_set_affinity(bod_handle h, int32_t image_index, const float* centre_columns, int32_t k, int32_t m);
/* Final detections of one image: K <= max_detections rows.
 * scores [K,C], means [K,4] (v,u,h,w), covs [K,16], counts [K,C]. */
bod_status bod_get_detections(bod_handle h, int32_t image_index, int32_t* num_detections,
                              float* scores, float* means, float* covs, float* counts);

/* All images of the batch in one call (one device->host copy per array, one sync):
 * num_detections [batch]; scores/counts [batch,max_detections,C]; means [batch,max_detections,4];
 * covs [batch,max_detections,16]; rows >= num_detections[b] are unspecified. NULLs skipped. */
bod_status bod_get_detections_batch(bod_handle h, int32_t* num_detections, float* scores, float* means,
                                    float* covs, float* counts);
/* Device addresses of the same five arrays of record slot 0/1 (order: num, scores, means, covs,
 * counts) for zero-copy hand-off to a collective library (the RCCL gather of SURVEY.md section
 * 8e).  Valid until bod_destroy; contents are defined once the batch that filled the slot has
 * completed (bod_synchronize, or bod_collect with NULL destinations). Synchronous calls use slot 0
 * until the first bod_infer_async. */
bod_status bod_device_detections(bod_handle h, int32_t slot, void** ptrs5);
/* Device addresses of the raw head outputs of bod_forward (order: cls [B,N,A,C], box [B,N,A,4],
 * cov [B,N,A,10] or NULL), fp32 -- the tensors RetinaNetModel.call returns (retinanet_model.py:99-112) --
 * for zero-copy exchange between handles / GPUs (MC-sample sharding gathers every rank's [1,n,A,.] slices
 * into one handle's buffers).  mark_ready != 0 declares the buffers filled by the caller (like bod_set_raw),
 * so bod_posterior may run on them; the caller orders its writes before that call (bod_synchronize or
 * its own stream/event ordering). */
bod_status bod_device_raw(bod_handle h, void** ptrs3, int32_t mark_ready);

/* The whole per-image body of run_inference.test_model's loop (:137-149) for `batch` images:
 * forward -> posterior -> nms -> cluster_fuse, one stream, no host round trip. */
bod_status bod_infer(bod_handle h, const float* images, int32_t images_on_device,
                     uint64_t seed, uint32_t first_image_id);

/* Pipelined form for sustained throughput: enqueue the whole pass and return at once, so that the host
 * enqueues batch i+1 while batch i runs and its records travel to pinned memory.  (Until round 5 the
 * soft-NMS + cluster-fuse of a batch ran on a side stream underneath the next batch's convolutions; they
 * follow the posterior on the main stream now -- DESIGN.md 8.4.)  Detection records are double-buffered ("slots").  *slot receives the
 * ticket to pass to bod_collect, which waits for that batch and copies its padded records
 * (layout as bod_get_detections_batch).  At most two batches may be in flight. */
bod_status bod_infer_async(bod_handle h, const float* images, int32_t images_on_device,
                           uint64_t seed, uint32_t first_image_id, int32_t* slot);
bod_status bod_collect(bod_handle h, int32_t slot, int32_t* num_detections, float* scores,
                       float* means, float* covs, float* counts);

/* Device buffer of [batch,H,W,3] fp32 owned by the handle (fill with bod_upload_images, then
 * pass to bod_forward/bod_infer with images_on_device=1). */
bod_status bod_upload_images(bod_handle h, const float* host_images);
/* Same destination, filled from decoded uint8 RGB frames [batch, src_h, src_w, 3]: the dataset handlers'
 * preprocessing runs on the device -- float conversion, mean subtraction (rgb_means[3], constants.py:12),
 * RGB->BGR (bdd_dataset_handler.py:128-139) and, with aspect_resize != 0, KITTI's
 * tf.image.resize(BILINEAR, preserve_aspect_ratio=True) + resize_with_crop_or_pad to the network size
 * (kitti_dataset_handler.py:120-148).  A quarter of the PCIe bytes of bod_upload_images. */
bod_status bod_upload_frames_u8(bod_handle h, const uint8_t* rgb, int32_t src_h, int32_t src_w,
                                const float* rgb_means, int32_t aspect_resize);
/* Pipelined form of the same upload: the host->device copy of `rgb` (pinned host memory for a truly asynchronous copy)
 * and the device preprocessing run on the handle's COPY stream into image buffer 0 or 1 and return at once, so the PCIe
 * transfer of batch i+1 overlaps the convolutions of batch i (the reference gets the same overlap from tf.data's
 * prefetch, run_inference.py:71-72).  Pass bod_device_images_buffer(h, buffer) with images_on_device = 1 to bod_forward /
 * bod_infer / bod_infer_async: the forward waits for the upload, and the next upload into the same buffer waits until the
 * stem of that forward has consumed the frames.  `rgb` must stay valid until that forward has been enqueued AND the copy
 * has completed (bod_synchronize, or a later bod_collect of that batch).  Buffer 0 is bod_device_images(). */
bod_status bod_upload_frames_u8_async(bod_handle h, const uint8_t* rgb, int32_t src_h, int32_t src_w,
                                      const float* rgb_means, int32_t aspect_resize, int32_t buffer);
const float* bod_device_images_buffer(bod_handle h, int32_t buffer);
const float* bod_device_images(bod_handle h);
bod_status bod_synchronize(bod_handle h);

/* Stage entry point for parity tests of the hot kernel: ONE convolution of the forward pass
 * (keras Conv2D semantics) through the same implicit-GEMM MFMA kernel and
 * epilogue the pipeline uses.  x [B,H,W,Cin] and w [KH,KW,Cin,Cout] (HWIO) are fp32 host arrays
 * and are rounded to bf16 exactly as stored activations / packed weights are; bias fp32 (may be
 * NULL); residual [B,OH,OW,Cout] optional (added before ReLU, like the bottleneck shortcut,
 * feature_extractor.py:210-212).  dropout_rate > 0 applies the head-tower epilogue
 * (multitask_headers.py:102-116): batch item b plays MC sample b, pixel index = y*OW+x.
 * out [B,OH,OW,Cout] fp32; round_output_bf16 != 0 rounds it like a stored activation.
 * precision = BOD_PRECISION_FP32 runs the fp32 twin of the kernel on unrounded fp32 operands.
 * precision = BOD_PRECISION_F16MX / BOD_PRECISION_F16MX4 (3x3, stride 1, SAME, 256 -> 256 only: a head-tower layer) runs that tower kernel on the
 * row-reuse loop; round_output_bf16 then selects the data path under test: 0 = hx rows in, (hi, lo) pairs out (a head's last
 * layer); 1 = hx rows in, hx rows out (a middle layer; `out` is the decoded hx row: f16 hi + e2m3 lo); 2 = (hi, lo) pairs in,
 * hx rows out (the first layer).
 * Errors are reported through bod_last_error(NULL). */
bod_status bod_stage_conv(int32_t device, const float* x, int32_t B, int32_t H, int32_t W, int32_t Cin,
                          const float* w, const float* bias, int32_t KH, int32_t KW, int32_t Cout,
                          int32_t stride, int32_t same_padding, int32_t relu, const float* residual,
                          float dropout_rate, uint64_t seed, int32_t layer_id, uint32_t image_id,
                          int32_t round_output_bf16, int32_t precision, float* out);

/* Stage entry point of the training step's hot kernel (run_training.py:208-247
 * tape.gradient): the WEIGHT GRADIENT of one Conv2D, dw[KH,KW,Cin,Cout] = dL/dW and db[Cout] = dL/db given the
 * layer input x [B,H,W,Cin] and the output gradient dy [B,OH,OW,Cout] (keras Conv2D geometry as in
 * bod_stage_conv; x and dy are rounded to bf16 like stored activations, accumulation is fp32).  Runs as a
 * pixel-reduction GEMM on the forward implicit-GEMM MFMA kernel: im2col^T of x through the forward row table,
 * dy transposed, reduction split over `ksplit` workgroups per tile (0 = chosen automatically).  The input
 * gradient needs no entry point of its own: it is bod_stage_conv on dy with the spatially flipped, cin/cout
 * swapped weights (tests/test_gpu_train_blocks.py).  Errors via bod_last_error(NULL). */
bod_status bod_stage_conv_wgrad(int32_t device, const float* x, int32_t B, int32_t H, int32_t W, int32_t Cin,
                                const float* dy, int32_t KH, int32_t KW, int32_t Cout, int32_t stride,
                                int32_t same_padding, int32_t ksplit, float* dw, float* db);

/* One training step, run_training.train_single_step (:208-247), on a handle created with training = 1:
 * forward in training mode (dropout on, batch-norm frozen), total loss = w_cls * focal + w_reg * regression
 * (reg_kind as in bod_loss_forward) + Keras l2(l2_rate) on the header tower kernels and cov_out, backward through
 * the whole network, tf.clip_by_global_norm(5.0) and keras Adam(epsilon = 1e-2) with `learning_rate` (the piecewise
 * schedule is the caller's, :48-61).  Targets as the dataset handler produces them: cls_targets [B,A,C],
 * box_targets [B,A,4], positive / negative anchor masks [B,A] (bytes).  apply_update = 0 stops after the
 * gradients (tests).  out6 = {total_loss, cls_loss, reg_loss, covariance_loss, regularization_loss, global
 * gradient norm before clipping}. */
bod_status bod_train_step(bod_handle h, const float* images, int32_t images_on_device, const float* cls_targets,
                          const float* box_targets, const uint8_t* positive_mask, const uint8_t* negative_mask,
                          uint64_t seed, uint32_t first_image_id, int32_t reg_kind, float label_smoothing,
                          float w_cls, float w_reg, float l2_rate, float learning_rate, int32_t apply_update,
                          double* out6);
/* Data-parallel training (one process per GPU): after bod_train_step(..., apply_update = 0) the gradients of ALL
 * trainable tensors lie in one contiguous fp32 device array -- bod_train_gradients returns its address and length
 * (the handle's stream is synchronised first) -- so the ranks need ONE all-reduce (RCCL) over it, then
 * bod_train_apply runs the clip + Adam update on whatever the array then holds (the mean gradient). */
bod_status bod_train_gradients(bod_handle h, void** device_ptr, int64_t* count);
bod_status bod_train_apply(bod_handle h, float learning_rate, double* grad_norm);
/* Read a trainable tensor of a training handle back: layer = Keras layer name (conv or batch-norm), kind 0 kernel
 * (HWIO) / 1 bias / 2 gamma / 3 beta, what 0 value / 1 gradient of the last step / 2, 3 Adam moments. */
bod_status bod_train_get(bod_handle h, const char* layer, int32_t kind, int32_t what, float* out, int64_t n);
/* Checkpoint / resume of the optimizer (tf.train.Checkpoint(step, optimizer, net) + ckpt.restore(manager.latest_checkpoint),
 * run_training.py:68-82): bod_train_set writes an Adam moment (what = 2 first, 3 second) of a trainable tensor back -- the
 * values themselves are restored through bod_load_weight before bod_finalize_weights -- and bod_train_step_count reads
 * (get != NULL) and / or sets (set >= 0) the number of updates applied so far, which enters Adam's bias correction. */
bod_status bod_train_set(bod_handle h, const char* layer, int32_t kind, int32_t what, const float* data, int64_t n);
bod_status bod_train_step_count(bod_handle h, int64_t* get, int64_t set);

/* model.get_loss(sample_dict, prediction_dict) forward (retinanet_model.py:151-328, core/losses.py:30-61;
 * BASELINE config 5's loss, forward only).  Host arrays: cls/cls_targets [B,A,C], box/box_targets [B,A,4],
 * covar_params [B,A,10] (pre fill_triangular; may be NULL unless reg_kind >= 2), anchors [A,4],
 * positive/negative masks [B,A] (bytes).  reg_kind: 0 none, 1 'regression', 2 'regression_var',
 * 3 'regression_covar'.  out4 = {sum of masked focal terms, sum of positive regression terms,
 * sum of positive 0.5*sum(log D) terms, number of positives}; the caller applies the
 * /max(num_pos,1) normalisation and the yaml loss weights. Errors via bod_last_error(NULL). */
bod_status bod_loss_forward(int32_t device, int32_t B, int32_t A, int32_t C, const float* cls,
                            const float* cls_targets, const float* box, const float* box_targets,
                            const float* covar_params, const float* anchors, const uint8_t* positive_mask,
                            const uint8_t* negative_mask, int32_t do_classification, int32_t reg_kind,
                            float label_smoothing, double* out4);

/* Gradient of  total = w_cls * S_cls / max(n_pos,1) + w_reg * (S_cmp + S_reg) / max(n_pos,1)  (the reference's
 * total_loss before the L2 term, retinanet_model.py:183-323) with respect to the raw head outputs: dcls [B,A,C],
 * dbox [B,A,4], dcov [B,A,10] (NULL = not wanted).  Arguments as bod_loss_forward; out4 receives the same sums.
 * First piece of the training step's backward pass. */
bod_status bod_loss_backward(int32_t device, int32_t B, int32_t A, int32_t C, const float* cls,
                             const float* cls_targets, const float* box, const float* box_targets,
                             const float* covar_params, const float* anchors, const uint8_t* positive_mask,
                             const uint8_t* negative_mask, int32_t do_cls, int32_t reg_kind, float label_smoothing,
                             float w_cls, float w_reg, double* out4, float* dcls, float* dbox, float* dcov);

/* Kernel micro-benchmark (tests/tools): re-launches the `layer`-th head-tower launch of a bod_infer step
 * (0 = de-duplicated fan-out layer, 1 = tower layer 1; on plans with the fused MC aggregation 2 = layer 2 of the
 * heads that continue, 3 = layer 2 of the head that ends there (aggregating), 4 = layer 3; otherwise 2, 3 = layers
 * 2, 3) `iters` times on the handle's own buffers and returns the mean duration; `variant` selects an ablation
 * build of the kernel (0 = production). */
bod_status bod_bench_head_conv(bod_handle h, int32_t layer, int32_t variant, int32_t iters,
                               double* mean_ms, double* flops_per_launch);

/* Measurement hooks (bench.py): HIP-event timing of the dominant kernel on the handle's stream. */
bod_status bod_profile_begin(bod_handle h);
/* Which head 3x3 launches the hook times: 0 = all of them (default), 1 = only the launches of the row-reuse tower kernel
 * (tower layers 1..3 in bf16 mode: ONE kernel symbol, the one a rocprofv3 --kernel-trace summary lists first), 2 = the
 * others (the N-way fan-out launch of layer 0).  Stays in force until changed. */
bod_status bod_profile_select(bod_handle h, int32_t which);
/* total ms in head 3x3 conv launches since begin, number of launches, FLOPs (2*MACs) issued */
bod_status bod_profile_end(bod_handle h, double* head_conv_ms, int64_t* head_conv_launches,
                           double* head_conv_flops, double* posterior_ms, int64_t* posterior_launches);

/* ---- the path's ONE multi-GPU exchange for callers of the C ABI (no PyTorch in the loop) ----
 * Images shard across processes (one per GPU); per step every rank contributes the detection records of its batch and `root`
 * receives all of them.  A record row is W = bod_record_width() = 1 + 4 + 16 + 2C floats: [valid, mean (v,u,h,w), covariance
 * row-major, score[C], counts[C]]; a rank's block is [batch][max_detections][W], rows beyond an image's detection count are zero.
 *
 * bod_gather_detections packs the records of `slot` (the ticket of bod_infer_async; pass -1 after a synchronous bod_infer) on the
 * device and issues ONE RCCL gather -- ncclGather(send, recv, batch*K*W, ncclFloat32, root, comm, stream).  For a ticket
 * (slot >= 0) both run on the stream that finished the slot's records -- the handle's MAIN stream -- behind its cluster-and-fuse
 * kernels and record copies, and the slot's event is re-recorded behind them: bod_collect(slot) and the next bod_infer_async that
 * reuses the slot wait for the send.  If the next bod_infer_async has already been enqueued, pack and gather run BEHIND its kernels
 * (one step of latency): since round 6 no kernel of this library runs beside another one by default (until then the
 * two ran on a side stream underneath the next batch's convolutions; BOD_SIDE_STREAM=1 restores that placement for reproduction
 * runs).  For slot == -1 both run on the handle's MAIN stream, behind the
 * synchronous bod_infer's own kernels: every later call on the handle (the next bod_infer, bod_synchronize, ...) is ordered behind
 * the pack and the gather by stream order.
 * `nccl_comm` is the caller's ncclComm_t (created with ncclCommInitRank on this handle's device; librccl.so is opened at run time,
 * the library does not link it); `world` / `rank` are the communicator's size and this process' rank.  nccl_comm == NULL is the
 * single-process form (world must be 1: the block is "gathered" by a device copy).
 * On `root`, `gathered_host` (may be NULL) receives [world][batch][K][W] floats, in rank order, after the copy has been waited
 * for (an event behind it, not the whole stream); *gathered_device (may be NULL) is set to the device copy, valid until the next gather: its contents are complete once
 * bod_collect(slot) has returned (slot >= 0) or after bod_synchronize (slot == -1) -- unless gathered_host was given, in which case
 * the call itself has waited.  Other ranks pass NULL for both (their call returns once the send is enqueued).  A pending slot is
 * NOT released: bod_collect still may. */
int32_t bod_record_width(bod_handle h);
bod_status bod_gather_detections(bod_handle h, int32_t slot, void* nccl_comm, int32_t world, int32_t rank, int32_t root,
                                 float* gathered_host, float** gathered_device);

/* What the handle's plan looks like (tests / bench report it; nothing on the hot path reads it).  info8[0] = 1 when the MC
 * statistics are reduced inside the last tower layers' tiles (no [B,N,A,.] tensors on the bod_infer path), [1] = 1 when the 1x1
 * head output convs are fused into the last tower layers' epilogues, [2] = 1 when the per-sample tower layers run on the
 * activation-row-reuse kernel, [3] = 1 when the fan-out layer does, [4] = number of ops of the forward plan, [5] = number of
 * backbone / FPN 3x3 layers planned on the row-reuse kernel, [6] = 1 / 2 when the head towers run the f16mx / f16mx4 arithmetic
 * (BOD_PRECISION_F16MX / BOD_PRECISION_F16MX4), [7] = 0. */
bod_status bod_plan_info(bod_handle h, int32_t* info8);

#ifdef __cplusplus
}
#endif
#endif /* BAYESOD_H */
