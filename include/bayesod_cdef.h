/* GENERATED from include/bayesod.h by bayes_od_rc_amd.build.write_cdef(): the same declarations without
 * comments / preprocessor lines, for ffi.cdef(open('include/bayesod_cdef.h').read()).  Do not edit. */
typedef struct bod_context* bod_handle;
typedef enum {
    BOD_OK = 0,
    BOD_ERR_INVALID_ARG = 1,
    BOD_ERR_HIP = 2,
    BOD_ERR_OOM = 3,
    BOD_ERR_NOT_READY = 4,
    BOD_ERR_NO_DEVICE = 5
} bod_status;
enum { BOD_RANK_SCORE = 0, BOD_RANK_JOINT_ENTROPY = 1 };
enum { BOD_NMS_VARIANT_A = 0, BOD_NMS_VARIANT_B = 1 };
enum { BOD_HEAD_CLS = 0, BOD_HEAD_REG = 1, BOD_HEAD_COV = 2 };
enum { BOD_PRECISION_BF16 = 0, BOD_PRECISION_FP32 = 1, BOD_PRECISION_BF16X3 = 2, BOD_PRECISION_F16MX = 3, BOD_PRECISION_F16MX4 = 4 };
typedef struct {
    int32_t device;
    int32_t image_h, image_w;
    int32_t batch;
    int32_t mc_samples;
    int32_t num_classes;
    int32_t anchors_per_location;
    int32_t min_level, max_level;
    float   dropout_rate;
    int32_t use_full_covar;
    int32_t dirichlet_non_informative;
    int32_t gaussian_isotropic;
    float   isotropic_variance;
    int32_t ranking_method;
    int32_t nms_max_output_size;
    float   nms_iou_threshold;
    float   nms_soft_sigma;
    int32_t nms_variant;
    int32_t num_categorical_draws;
    int32_t has_covar_head;
    float   kitti_scale_h, kitti_scale_w;
    int32_t precision;
    int32_t mc_sample_base;
    int32_t mc_ensemble_size;
    int32_t training;
    int32_t backbone_depth;
    int32_t pipeline_overlap;
    int32_t reserved[2];
} bod_config;
typedef struct {
    int32_t num_pixels;
    int32_t num_anchors;
    int32_t level_h[8], level_w[8];
    int32_t num_levels;
    int32_t max_detections;
    int64_t device_bytes;
} bod_sizes;
const char* bod_version(void);
const char* bod_last_error(bod_handle h);
bod_status bod_create(const bod_config* cfg, bod_handle* out);
bod_status bod_destroy(bod_handle h);
bod_status bod_query_sizes(bod_handle h, bod_sizes* out);
bod_status bod_update_config(bod_handle h, const bod_config* cfg);
bod_status bod_load_weight(bod_handle h, const char* name, int32_t kind,
                           const int64_t* shape, int32_t ndim, const float* data);
bod_status bod_finalize_weights(bod_handle h);
bod_status bod_set_anchors(bod_handle h, const float* anchors_vuhw, int32_t num_anchors);
bod_status bod_forward(bod_handle h, const float* images, int32_t images_on_device,
                       uint64_t seed, uint32_t first_image_id);
bod_status bod_get_raw(bod_handle h, float* cls, float* box, float* covar_params);
bod_status bod_set_raw(bod_handle h, const float* cls, const float* box, const float* covar_params);
bod_status bod_get_pyramid(bod_handle h, int32_t level_index, float* out);
bod_status bod_posterior(bod_handle h, uint64_t seed, uint32_t first_image_id);
bod_status bod_validation_post(bod_handle h);
bod_status bod_get_num_kept(bod_handle h, int32_t* num_kept);
bod_status bod_get_posterior(bod_handle h, int32_t image_index, float* counts, float* score,
                             float* means, float* covs, float* ranking, int32_t* anchor_index);
bod_status bod_set_posterior(bod_handle h, int32_t image_index, int32_t m, const float* counts,
                             const float* means, const float* covs, const float* ranking);
bod_status bod_nms(bod_handle h);
bod_status bod_get_nms(bod_handle h, int32_t image_index, int32_t* indices, int32_t* num_selected);
bod_status bod_set_nms(bod_handle h, int32_t image_index, const int32_t* indices, int32_t num_selected);
bod_status bod_get_iou_matrix(bod_handle h, int32_t image_index, float* iou);
bod_status bod_cluster_fuse(bod_handle h);
bod_status bod_set_affinity(bod_handle h, int32_t image_index, const float* centre_columns, int32_t k, int32_t m);
bod_status bod_get_detections(bod_handle h, int32_t image_index, int32_t* num_detections,
                              float* scores, float* means, float* covs, float* counts);
bod_status bod_get_detections_batch(bod_handle h, int32_t* num_detections, float* scores, float* means,
                                    float* covs, float* counts);
bod_status bod_device_detections(bod_handle h, int32_t slot, void** ptrs5);
bod_status bod_device_raw(bod_handle h, void** ptrs3, int32_t mark_ready);
bod_status bod_infer(bod_handle h, const float* images, int32_t images_on_device,
                     uint64_t seed, uint32_t first_image_id);
bod_status bod_infer_async(bod_handle h, const float* images, int32_t images_on_device,
                           uint64_t seed, uint32_t first_image_id, int32_t* slot);
bod_status bod_collect(bod_handle h, int32_t slot, int32_t* num_detections, float* scores,
                       float* means, float* covs, float* counts);
bod_status bod_upload_images(bod_handle h, const float* host_images);
bod_status bod_upload_frames_u8(bod_handle h, const uint8_t* rgb, int32_t src_h, int32_t src_w,
                                const float* rgb_means, int32_t aspect_resize);
bod_status bod_upload_frames_u8_async(bod_handle h, const uint8_t* rgb, int32_t src_h, int32_t src_w,
                                      const float* rgb_means, int32_t aspect_resize, int32_t buffer);
const float* bod_device_images_buffer(bod_handle h, int32_t buffer);
const float* bod_device_images(bod_handle h);
bod_status bod_synchronize(bod_handle h);
bod_status bod_stage_conv(int32_t device, const float* x, int32_t B, int32_t H, int32_t W, int32_t Cin,
                          const float* w, const float* bias, int32_t KH, int32_t KW, int32_t Cout,
                          int32_t stride, int32_t same_padding, int32_t relu, const float* residual,
                          float dropout_rate, uint64_t seed, int32_t layer_id, uint32_t image_id,
                          int32_t round_output_bf16, int32_t precision, float* out);
bod_status bod_stage_conv_wgrad(int32_t device, const float* x, int32_t B, int32_t H, int32_t W, int32_t Cin,
                                const float* dy, int32_t KH, int32_t KW, int32_t Cout, int32_t stride,
                                int32_t same_padding, int32_t ksplit, float* dw, float* db);
bod_status bod_train_step(bod_handle h, const float* images, int32_t images_on_device, const float* cls_targets,
                          const float* box_targets, const uint8_t* positive_mask, const uint8_t* negative_mask,
                          uint64_t seed, uint32_t first_image_id, int32_t reg_kind, float label_smoothing,
                          float w_cls, float w_reg, float l2_rate, float learning_rate, int32_t apply_update,
                          double* out6);
bod_status bod_train_gradients(bod_handle h, void** device_ptr, int64_t* count);
bod_status bod_train_apply(bod_handle h, float learning_rate, double* grad_norm);
bod_status bod_train_get(bod_handle h, const char* layer, int32_t kind, int32_t what, float* out, int64_t n);
bod_status bod_train_set(bod_handle h, const char* layer, int32_t kind, int32_t what, const float* data, int64_t n);
bod_status bod_train_step_count(bod_handle h, int64_t* get, int64_t set);
bod_status bod_loss_forward(int32_t device, int32_t B, int32_t A, int32_t C, const float* cls,
                            const float* cls_targets, const float* box, const float* box_targets,
                            const float* covar_params, const float* anchors, const uint8_t* positive_mask,
                            const uint8_t* negative_mask, int32_t do_classification, int32_t reg_kind,
                            float label_smoothing, double* out4);
bod_status bod_loss_backward(int32_t device, int32_t B, int32_t A, int32_t C, const float* cls,
                             const float* cls_targets, const float* box, const float* box_targets,
                             const float* covar_params, const float* anchors, const uint8_t* positive_mask,
                             const uint8_t* negative_mask, int32_t do_cls, int32_t reg_kind, float label_smoothing,
                             float w_cls, float w_reg, double* out4, float* dcls, float* dbox, float* dcov);
bod_status bod_bench_head_conv(bod_handle h, int32_t layer, int32_t variant, int32_t iters,
                               double* mean_ms, double* flops_per_launch);
bod_status bod_profile_begin(bod_handle h);
bod_status bod_profile_select(bod_handle h, int32_t which);
bod_status bod_profile_end(bod_handle h, double* head_conv_ms, int64_t* head_conv_launches,
                           double* head_conv_flops, double* posterior_ms, int64_t* posterior_launches);
int32_t bod_record_width(bod_handle h);
bod_status bod_gather_detections(bod_handle h, int32_t slot, void* nccl_comm, int32_t world, int32_t rank, int32_t root,
                                 float* gathered_host, float** gathered_device);
bod_status bod_plan_info(bod_handle h, int32_t* info8);
