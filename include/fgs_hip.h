/*
 * fgs_hip.h -- C ABI of libfgs_hip.so, the MI355X (gfx950) implementation of the FasterGS rasterizer hot path.
 *
 * This is the drop-in boundary: each entry point replaces one function of the reference's pybind module
 * `FasterGSCudaBackend._C` (reference: FasterGSCudaBackend/FasterGSCudaBackend/torch_bindings/bindings.cpp:12-21).
 * Plain pointers and sizes only -- no torch types. All pointers are DEVICE pointers unless marked [host].
 * Every function returns FGS_OK (0) or a negative fgs_status and never throws; fgs_last_error() gives the text.
 * All work is enqueued on `stream` (a hipStream_t passed as void*); the reference uses the legacy default stream
 * (rasterization/src/forward.cu:64 etc.), an explicit stream is the MI355X-side change.
 *
 * Tensor layouts are the reference's (torch_bindings/rasterization.py:113-132):
 *   means[N,3] scales[N,3] (log) rotations[N,4] (w first, unnormalised) opacities[N,1] (logit)
 *   sh_coefficients_0[N,1,3] sh_coefficients_rest[N,K-1,3], fp32, contiguous.
 */
#ifndef FGS_HIP_H
#define FGS_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* 3 (round 6): the private layout of the primitive blob grew in round 5 (footprint rows, tile counts, wave / block sums, big-footprint list) and the
 * product library stopped exporting fgs_debug_set_option / fgs_debug_set_backward_variant (libfgs_hip_dev.so has them): a caller that sized or
 * decoded blobs by the version-2 layout, or bound the two debug symbols, must notice. Signatures of the version-2 entry points are unchanged. */
#define FGS_ABI_VERSION 3

typedef enum fgs_status {
    FGS_OK = 0,
    FGS_ERR_INVALID_ARGUMENT = -1,
    FGS_ERR_ALLOC = -2,      /* the resize callback returned NULL */
    FGS_ERR_HIP = -3,        /* a HIP runtime call or kernel launch failed */
    FGS_ERR_INTERNAL = -4
} fgs_status;

/* The 13 fields of RasterizerSettings (torch_bindings/rasterization.py:8-38) plus total_sh_bases_rest, which the
 * reference derives from sh_coefficients_rest.size(1) (rasterization/src/rasterization_api.cu:44). */
typedef struct fgs_settings {
    const float* w2c;           /* >= 12 floats: rows 0..2 of the row-major world-to-camera matrix */
    const float* cam_position;  /* 3 floats */
    const float* bg_color;      /* 3 floats */
    int32_t active_sh_bases;
    int32_t total_sh_bases_rest;
    int32_t width;
    int32_t height;
    float focal_x, focal_y, center_x, center_y, near_plane, far_plane;
    int32_t proper_antialiasing;
} fgs_settings;

/* Scratch ownership follows the reference: the caller owns four byte buffers and hands the library a callback that
 * (re)sizes buffer `which` to `bytes` and returns its device pointer -- the C form of resize_function_wrapper
 * (utils/torch_utils.h:6-12) as used at rasterization/src/forward.cu:44,58,179,236. Layout inside is private. */
enum { FGS_BUF_PRIMITIVE = 0, FGS_BUF_TILE = 1, FGS_BUF_INSTANCE = 2, FGS_BUF_BUCKET = 3, FGS_BUF_COUNT = 4 };
typedef void* (*fgs_resize_fn)(void* user, int32_t which, size_t bytes);

/* What _C.forward returns next to the image and what _C.backward takes back (rasterization_api.h:8-59):
 * (n_instances, n_buckets, selector). Opaque to callers; n_visible is extra (reported by bench.py). */
typedef struct fgs_forward_state {
    int32_t n_visible;
    int32_t n_instances;
    int32_t n_buckets;   /* capacity of the bucket buffer (upper bound of the device-side count) */
    int32_t selector;    /* which half of the instance double buffer holds the tile-sorted list */
} fgs_forward_state;

int32_t fgs_abi_version(void);
const char* fgs_last_error(void);   /* [host] thread-local, valid until the next call on this thread */
const char* fgs_build_info(void);   /* [host] e.g. "gfx950 wave64 tile16x12 bucket64" */

/* replaces _C.forward  (rasterization_api.cu:13-91 -> rasterization/src/forward.cu:11-259). image: [3,H,W]. */
int32_t fgs_forward(const float* means, const float* scales, const float* rotations, const float* opacities,
                    const float* sh_coefficients_0, const float* sh_coefficients_rest, int32_t n_primitives,
                    const fgs_settings* settings, float* image,
                    fgs_resize_fn resize, void* resize_user, fgs_forward_state* state_out, void* stream);

/* Bytes of zero-initialisable scratch fgs_backward needs (replaces the two helper tensors grad_mean2d_helper /
 * grad_conic_helper of rasterization_api.cu:133-134 plus per-pixel staging). */
size_t fgs_backward_scratch_bytes(int32_t n_primitives, int32_t width, int32_t height);

/* replaces _C.backward (rasterization_api.cu:94-178 -> rasterization/src/backward.cu:8-125).
 * The six grad_* outputs need NOT be zero-filled by the caller: every element is written (zeros for primitives
 * that were not visible), which replaces the reference's 8 torch::zeros fills (rasterization_api.cu:127-134).
 * densification_info: [2,N] accumulated in place (kernels_backward.cuh:194-201) or NULL. */
int32_t fgs_backward(const float* grad_image, const float* image,
                     const float* means, const float* scales, const float* rotations, const float* opacities,
                     const float* sh_coefficients_rest,
                     void* primitive_buffers, void* tile_buffers, void* instance_buffers, void* bucket_buffers,
                     float* grad_means, float* grad_scales, float* grad_rotations, float* grad_opacities,
                     float* grad_sh_coefficients_0, float* grad_sh_coefficients_rest,
                     float* densification_info, void* scratch,
                     int32_t n_primitives, const fgs_settings* settings, const fgs_forward_state* state, void* stream);
/* fgs_backward plus one byte per block of 64 consecutive Gaussians: live_blocks[b] = 1 if any Gaussian 64 b .. 64 b + 63 was visible, 0 if
 * all gradients of the block are zero (they are written all the same: the gradient tensors are dense and valid). [ceil(n_primitives / 64)]
 * bytes, or NULL (= fgs_backward). One third of the Gaussians is invisible in a view and 85 % of them sit in such blocks (Morton order):
 * fgs_adam_step_multi_live does not read the zeros back. */
int32_t fgs_backward_live(const float* grad_image, const float* image,
                          const float* means, const float* scales, const float* rotations, const float* opacities,
                          const float* sh_coefficients_rest,
                          void* primitive_buffers, void* tile_buffers, void* instance_buffers, void* bucket_buffers,
                          float* grad_means, float* grad_scales, float* grad_rotations, float* grad_opacities,
                          float* grad_sh_coefficients_0, float* grad_sh_coefficients_rest,
                          float* densification_info, void* scratch,
                          int32_t n_primitives, const fgs_settings* settings, const fgs_forward_state* state, uint8_t* live_blocks, void* stream);

/* fgs_forward WITHOUT its host synchronisation (the reference blocks three times per forward pass, forward.cu:100,102,234; fgs_forward
 * once): nothing is read back. The instance-stage buffers and launches are sized by `instance_capacity` -- the caller's bound, e.g. 1.25 x
 * the largest count fgs_forward_counts() has reported, scaled with the primitive count -- and every kernel reads the exact counts on the
 * device. state_out: n_visible = n_primitives and n_instances = instance_capacity (bounds; pass them back to fgs_backward unchanged).
 * If the real instance count exceeds the capacity the excess instances are DROPPED (incomplete image) and a flag is raised that
 * fgs_forward_counts() reports: the caller repeats the pass with fgs_forward or a larger capacity. All work is enqueued on `stream`;
 * together with fgs_backward / fgs_adam_step_multi a whole iteration is free of host waits (and capturable, resize callback aside). */
int32_t fgs_forward_async(const float* means, const float* scales, const float* rotations, const float* opacities,
                          const float* sh_coefficients_0, const float* sh_coefficients_rest, int32_t n_primitives,
                          const fgs_settings* settings, float* image, int32_t instance_capacity, fgs_resize_fn resize, void* resize_user,
                          fgs_forward_state* state_out, void* stream);
/* Enqueues the copy of (n_visible, n_instances, capacity_exceeded) of the forward pass that filled `primitive_buffers` into
 * host_out[3] (pinned memory recommended); valid once `stream` has reached this point. No synchronisation. */
int32_t fgs_forward_counts(const void* primitive_buffers, int32_t n_primitives, int32_t* host_out, void* stream);

/* replaces _C.inference (rasterization_api.cu:181-247 -> rasterization/src/inference.cu:11-226).
 * image: [3,H,W] if to_chw else [H,W,3]. Only FGS_BUF_PRIMITIVE/TILE/INSTANCE are requested. */
int32_t fgs_inference(const float* means, const float* scales, const float* rotations, const float* opacities,
                      const float* sh_coefficients_0, const float* sh_coefficients_rest, int32_t n_primitives,
                      const fgs_settings* settings, float* image, int32_t to_chw, int32_t clamp_output,
                      fgs_resize_fn resize, void* resize_user, fgs_forward_state* state_out, void* stream);

/* replaces _C.pruning_scores (rasterization_api.cu:250-309 -> rasterization/src/pruning_scores.cu):
 * accumulates the Speedy-Splat importance score of every primitive for one view into scores[N]. */
int32_t fgs_pruning_scores(float* scores, const float* means, const float* scales, const float* rotations, const float* opacities,
                           const float* sh_coefficients_0, const float* sh_coefficients_rest, int32_t n_primitives,
                           const fgs_settings* settings, fgs_resize_fn resize, void* resize_user, fgs_forward_state* state_out, void* stream);

/* replaces _C.adam_step (adam/src/adam.cu:36-71): in-place Adam on one tensor, bias corrections in double on the host. */
int32_t fgs_adam_step(const float* grad, float* param, float* exp_avg, float* exp_avg_sq, int64_t n_elements,
                      int32_t step, double lr, double beta1, double beta2, double eps, void* stream);

/* All parameter groups of FusedAdam.step() (torch_bindings/adam.py:11-36) in ONE launch. Arrays are [host], length n_groups <= 8. */
int32_t fgs_adam_step_multi(int32_t n_groups, const float* const* grads, float* const* params, float* const* exp_avgs,
                            float* const* exp_avg_sqs, const int64_t* n_elements, const int32_t* steps, const double* lrs,
                            double beta1, double beta2, double eps, void* stream);
/* The same with the caller's PROMISE that gradient rows of dead blocks are zero: live_blocks as written by fgs_backward_live for exactly these
 * gradient tensors (device, [ceil(N / 64)] bytes), floats_per_gaussian[k] = row length of group k ([host]). Gradients of dead blocks are not
 * read -- except one sentinel float per dead block and tensor (the block's first element): if it is not +-0 the block is read after all, so a
 * whole-tensor edit behind the caller's back does not go unnoticed; parameters and moments of every Gaussian are updated as always (the result
 * is bit-identical to fgs_adam_step_multi). NULL = no promise. */
int32_t fgs_adam_step_multi_live(int32_t n_groups, const float* const* grads, float* const* params, float* const* exp_avgs,
                                 float* const* exp_avg_sqs, const int64_t* n_elements, const int32_t* steps, const double* lrs,
                                 double beta1, double beta2, double eps, const uint8_t* live_blocks, const int32_t* floats_per_gaussian,
                                 void* stream);

/* Fused backward + Adam (the reference's FasterGSFused branch, README.md:37; not in /root/reference -- defined here as
 * "equal to fgs_backward followed by FusedAdam.step() on all six groups"). Gradients are never
 * materialised. params/exp_avg/exp_avg_sq are arrays [host] of 6 device pointers in the order
 * means, sh_coefficients_0, sh_coefficients_rest, opacities, scales, rotations (Model.py:238-245); lrs likewise. */
int32_t fgs_backward_adam_fused(const float* grad_image, const float* image,
                                float* const* params, float* const* exp_avgs, float* const* exp_avg_sqs,
                                void* primitive_buffers, void* tile_buffers, void* instance_buffers, void* bucket_buffers,
                                float* densification_info, void* scratch,
                                int32_t n_primitives, const fgs_settings* settings, const fgs_forward_state* state,
                                int32_t step, const double* lrs, double beta1, double beta2, double eps, void* stream);

/* ---- Gaussian-sharded multi-GPU path ------------------------------------------------------------------------------
 * The reference is single-GPU (Renderer.py:58-61, no collective anywhere: SURVEY.md D4 / 8e); these four entry points
 * are the MI355X-side addition that lets G ranks each OWN N/G Gaussians and still render whole views: the pipeline of
 * fgs_forward / fgs_backward cut at the two places where per-Gaussian data is small.
 *
 *   owner of a shard, all views of the step at once: fgs_shard_preprocess    K1 -> 56-byte projected records of the visible
 *   -- all-to-all (records of view v go to the rank rendering v) --
 *   renderer of a view:                            fgs_forward_from_records  K2..K10 over the concatenated records
 *                                                  fgs_backward_to_records   K11 -> 36-byte accumulator record per record
 *   -- all-to-all back (segment s returns to the owner of shard s) --
 *   owner, all views at once:                      fgs_shard_backward        K12 on the shard, summing over views
 *
 * A splat record is the private 48-byte projected primitive + its depth key + its tile count; opaque to callers, only
 * its size is ABI. Records of a (shard, view) keep the order K1 compacted them in; accumulator records come back in
 * the same order. */
#define FGS_SPLAT_RECORD_BYTES 56
#define FGS_ACC_RECORD_BYTES 36

/* K1 over the shard for the n_views cameras of the step in one launch per 8 views. settings: [host] array of n_views
 * (same image size and SH layout). records_out: n_views x n_primitives records, view-major (view v starts at record
 * v * n_primitives; its first n_visible[v] are filled). counts_out: n_views x 2 uint32 device words, (n_visible,
 * n_instances) per view -- read them back for all views at once. Requests FGS_BUF_PRIMITIVE only; keep that buffer until
 * fgs_shard_backward of the same step. No host synchronisation. */
int32_t fgs_shard_preprocess(const float* means, const float* scales, const float* rotations, const float* opacities,
                             const float* sh_coefficients_0, const float* sh_coefficients_rest, int32_t n_primitives,
                             int32_t n_views, const fgs_settings* settings, void* records_out, uint32_t* counts_out,
                             fgs_resize_fn resize, void* resize_user, void* stream);

/* fgs_forward with K1 replaced by n_records received records (any concatenation order). n_instances MUST be the sum of
 * the producers' instance counts (their counts_out[1]); it sizes the instance buffer. No host synchronisation. */
int32_t fgs_forward_from_records(const void* records, int32_t n_records, int32_t n_instances, const fgs_settings* settings,
                                 float* image, fgs_resize_fn resize, void* resize_user, fgs_forward_state* state_out, void* stream);

/* The same for records that are the concatenation of n_shards segments (shard_counts[s] records of shard 0, 1, ...: [host] array): the renderer
 * places them interleaved -- rank r of shard s becomes primitive sum_s' min(count[s'], r + [s' < s]) -- which restores the Morton neighbourhood of
 * owners that hold every n_shards-th Gaussian (K11: 0.51 -> 0.43 ms at 3 M Gaussians, 8 shards). Results are those of fgs_forward_from_records up to
 * the summation order of K11's atomics. More than 8 segments, or NULL: as received. Pair with fgs_backward_to_shard_records and the SAME counts. */
int32_t fgs_forward_from_shard_records(const void* records, int32_t n_records, int32_t n_instances, const int32_t* shard_counts, int32_t n_shards,
                                       const fgs_settings* settings, float* image, fgs_resize_fn resize, void* resize_user,
                                       fgs_forward_state* state_out, void* stream);

/* K11 over the buffers of fgs_forward_from_records; acc_records_out[n_records] (36 bytes each, record order).
 * scratch: fgs_backward_scratch_bytes(n_records, width, height). */
int32_t fgs_backward_to_records(const float* grad_image, const float* image,
                                void* primitive_buffers, void* tile_buffers, void* instance_buffers, void* bucket_buffers,
                                void* scratch, float* acc_records_out, int32_t n_records,
                                const fgs_settings* settings, const fgs_forward_state* state, void* stream);

/* ... over the buffers of fgs_forward_from_shard_records (same shard_counts): the accumulator records come out in the order the records came in. */
int32_t fgs_backward_to_shard_records(const float* grad_image, const float* image,
                                      void* primitive_buffers, void* tile_buffers, void* instance_buffers, void* bucket_buffers,
                                      void* scratch, float* acc_records_out, int32_t n_records, const int32_t* shard_counts, int32_t n_shards,
                                      const fgs_settings* settings, const fgs_forward_state* state, void* stream);

/* K12 on the shard for all views of the step: acc_records = the accumulator records returned for the records of
 * fgs_shard_preprocess, concatenated in view order (n_visible[v] records for view v, [host] array; same order as they
 * were sent), primitive_buffers = that call's FGS_BUF_PRIMITIVE, settings = the same array. The gradients of each
 * Gaussian are summed over the views and every gradient element is written once (zeros where no view sees it);
 * densification_info [2,N] or NULL is accumulated per visible view as in fgs_backward.
 * scratch: fgs_shard_backward_scratch_bytes(n_primitives, n_views). */
size_t fgs_shard_backward_scratch_bytes(int32_t n_primitives, int32_t n_views);
int32_t fgs_shard_backward(const float* acc_records, const int32_t* n_visible, const void* primitive_buffers,
                           const float* means, const float* scales, const float* rotations, const float* opacities,
                           const float* sh_coefficients_rest,
                           float* grad_means, float* grad_scales, float* grad_rotations, float* grad_opacities,
                           float* grad_sh_coefficients_0, float* grad_sh_coefficients_rest,
                           float* densification_info, void* scratch, int32_t n_primitives, int32_t n_views,
                           const fgs_settings* settings, void* stream);

/* fgs_shard_backward fused with Adam on the shard (as fgs_backward_adam_fused is for the single-GPU path): the gradients,
 * summed over the n_views <= 8 views in registers / LDS, are never materialised. params / exp_avgs / exp_avg_sqs / lrs:
 * [host] arrays of 6 in optimizer-group order (means, sh_coefficients_0, sh_coefficients_rest, opacities, scales, rotations). */
int32_t fgs_shard_backward_adam_fused(const float* acc_records, const int32_t* n_visible, const void* primitive_buffers,
                                      float* const* params, float* const* exp_avgs, float* const* exp_avg_sqs,
                                      float* densification_info, void* scratch, int32_t n_primitives, int32_t n_views,
                                      const fgs_settings* settings, int32_t step, const double* lrs, double beta1, double beta2, double eps,
                                      void* stream);

/* Test/bench introspection: byte offsets of the named sub-arrays inside a scratch buffer, so tests can compare every
 * intermediate with the oracle. Returns the number of entries written (<= max_entries); names are static strings. */
typedef struct fgs_blob_entry { const char* name; size_t offset; size_t bytes; } fgs_blob_entry;
int32_t fgs_blob_layout(int32_t which, int32_t n_primitives, int32_t width, int32_t height, int32_t n_instances,
                        int32_t n_buckets, fgs_blob_entry* entries, int32_t max_entries);

/* replaces _C.update_3d_filter (filter3d/src/filter3d.cu:40-83). visibility_mask: one byte per point (torch.bool). */
int32_t fgs_update_3d_filter(const float* positions, const float* w2c, float* filter_3d, uint8_t* visibility_mask, int32_t n_points,
                             int32_t width, int32_t height, float focal_x, float focal_y, float center_x, float center_y,
                             float near_plane, float clipping_tolerance, float distance2filter, void* stream);
/* replaces _C.relocation_adjustment (densification/src/densification_api.cu:9-31, kernels_mcmc.cuh:28-59). The binomial table of
 * Eq. (9) (the reference's __constant__ array, kernels_mcmc.cuh:10-26) is produced on the host by fgs_relocation_table
 * (2500 floats) and passed in as a device buffer. */
int32_t fgs_relocation_table(float* table_host_2500);
int32_t fgs_relocation_adjustment(const float* old_opacities, const float* old_scales, const int64_t* n_samples_per_primitive,
                                  const float* table_device, float* new_opacities, float* new_scales, int32_t n_primitives, void* stream);
/* replaces _C.add_noise (densification_api.cu:33-58, kernels_mcmc.cuh:69-127); random_samples ~ N(0,1) [N,3] from the caller. */
int32_t fgs_add_noise(const float* raw_scales, const float* raw_rotations, const float* raw_opacities, const float* random_samples,
                      float* means, int32_t n_primitives, float current_lr, void* stream);

/* "Next" row: maintenance of the Gaussian set between iterations, on the device and INCLUDING the Adam moments
 * (the reference runs Model.py:275-366, 459-463 as torch mask / index / cat chains plus NeRFICG's extend / prune / sort_param_groups).
 *
 * Adaptive density control (Model.py:312-366) in two calls, because the caller owns the new tensors and must size them:
 *   fgs_adc_plan   classifies every Gaussian (clone if small, split if large, above the mean-gradient threshold; prune by opacity,
 *                  degenerate rotation and -- optionally -- size; the verdict for new Gaussians is taken on the values they will have)
 *                  and scans the plan. counts_out [host, 4]: surviving old Gaussians, surviving clones, surviving children PER COPY
 *                  (two copies), Gaussians that are split. New size = counts[0] + counts[1] + 2 * counts[2]. Synchronises the stream.
 *   fgs_adc_apply  writes the new parameter tensors and Adam moments (arrays of 6 in optimizer-group order means, sh_coefficients_0,
 *                  sh_coefficients_rest, opacities, scales, rotations; exp_avgs / exp_avg_sqs and their outputs may all be NULL): survivors
 *                  in their order with their moments, then clones, then first children, then second children, all with zero moments.
 *                  Children: means + R(q) (exp(s) * noise) with noise [2 * counts[3], 3] ~ N(0,1) from the caller (row = copy * counts[3]
 *                  + rank among the split Gaussians, the layout of the reference's randn_like), scales log(0.625 exp(s)).
 * scratch: fgs_adc_scratch_bytes(n) bytes, the same buffer for both calls. */
size_t fgs_adc_scratch_bytes(int32_t n_primitives);
int32_t fgs_adc_plan(const float* densification_info, const float* scales, const float* rotations, const float* opacities, int32_t n_primitives,
                     float grad_threshold, float min_opacity, int32_t prune_large_gaussians, float percent_dense, float extent,
                     void* scratch, int32_t* counts_out, void* stream);
int32_t fgs_adc_apply(const float* const* params, const float* const* exp_avgs, const float* const* exp_avg_sqs,
                      float* const* out_params, float* const* out_exp_avgs, float* const* out_exp_avg_sqs,
                      const float* noise, const void* scratch, int32_t n_primitives, int32_t total_sh_bases_rest, void* stream);
/* out[k][r, :] = in[k][index[r], :] for k < n_tensors <= 18 float tensors of row widths widths[k] in ONE launch: prune (index = the
 * survivors, Model.py:275-291) and sort (index = the ordering, :293-306) of the six parameters and their twelve moment tensors. */
int32_t fgs_gather_rows(int32_t n_tensors, const float* const* in, float* const* out, const int32_t* widths, const int64_t* index,
                        int32_t n_rows, void* stream);
/* Morton order of the means (Model.py:459-463): order_out[r] = index of the r-th point along a 30-bit Z-curve over the box lo..hi
 * ([device] 3 floats each; 10 bits per axis, x most significant), equal keys in index order (stable). */
size_t fgs_morton_order_temp_bytes(int32_t n_points);
int32_t fgs_morton_order(const float* means, const float* lo, const float* hi, int64_t* order_out, int32_t n_points, void* temp, size_t temp_bytes,
                         void* stream);

/* "Next" row: the loss between forward and backward of every iteration,
 *   loss = lambda_l1 * mean|image - target| + lambda_dssim * (1 - SSIM(image, target))        (Loss.py:15-16, Trainer.py:52-53)
 * replacing torch.nn.functional.l1_loss + NeRFICG's fused_dssim (Optim/Losses/DSSIM.py, not vendored). Writes three device
 * floats out[0] = mean|x-y|, out[1] = mean SSIM, out[2] = the loss (no host sync) and, if grad_image != NULL,
 * dloss/dimage [3,H,W]. scratch: fgs_l1_dssim_scratch_bytes(width, height) bytes, 8-byte aligned (it holds float2 partial sums; a
 * misaligned pointer is refused with FGS_ERR_INVALID_ARGUMENT). */
size_t fgs_l1_dssim_scratch_bytes(int32_t width, int32_t height);
int32_t fgs_l1_dssim_loss(const float* image, const float* target, int32_t width, int32_t height, float lambda_l1, float lambda_dssim,
                          float* out3, float* grad_image, void* scratch, void* stream);
/* The gradient alone, later: dloss/dimage * upstream from the derivative maps that fgs_l1_dssim_loss (with grad_image == NULL or not) left in
 * `scratch` for the same image / target. `upstream` is a DEVICE float (the dL/dloss an autograd engine hands to the loss node) or NULL for 1:
 * the scalar is folded into the kernel, so a framework does not spend a pass over the image on `grad * upstream`. This is the shape of the
 * reference's loss node (fused_dssim: forward saves maps, backward launches one kernel). */
int32_t fgs_l1_dssim_backward(const float* image, const float* target, int32_t width, int32_t height, float lambda_l1, float lambda_dssim,
                              const float* upstream, float* grad_image, const void* scratch, void* stream);

/* Optional per-stage timing. While enabled (1), every pipeline stage is bracketed by hipEvents recorded on the caller's stream
 * (each event costs a few microseconds of GPU idle time: ~0.2 ms per training iteration with all stages on); enable = 2 + k brackets
 * only stage k (its index in the table fgs_profile_read returns), which leaves a timed loop practically undisturbed;
 * fgs_profile_read() waits for them, returns accumulated milliseconds + launch counts per stage since the last read and
 * clears the records. Not thread-safe; intended for bench.py (roofline) and tests. */
typedef struct fgs_stage_time { const char* name; double total_ms; int64_t calls; } fgs_stage_time;
int32_t fgs_profile_enable(int32_t enable);
int32_t fgs_profile_read(fgs_stage_time* out, int32_t max_entries);

/* Device self-test of the wave64 primitives (DPP shift/rotate direction, ballot prefix, readlane); writes 256 words that
 * tests/test_gpu_parity.py checks against the expected pattern. which == FGS_BUF_COUNT in fgs_blob_layout describes the
 * backward scratch buffer. */
int32_t fgs_debug_wave_selftest(uint32_t* out_device_256, void* stream);

/* Test hook for the binning stage's radix sort (csrc/radix_sort.hip): stable sort of n (key, uint32 value) pairs on key bits
 * [0, end_bit); key_bytes 2 or 4. Returns 0 / 1 = the buffer pair that holds the result, or a negative fgs_status. */
size_t fgs_debug_radix_sort_temp_bytes(int32_t n, int32_t end_bit);
int32_t fgs_debug_radix_sort(void* keys0, void* keys1, uint32_t* vals0, uint32_t* vals1, int32_t n, int32_t key_bytes, int32_t end_bit,
                             void* temp, size_t temp_bytes, void* stream);
/* The same hook for the depth sort as the forward pass runs it (K2): keys are the bit patterns of float depths in [near_plane, far_plane]
 * (everything else is culled in preprocess, kernels_forward.cuh:67), sorted as key - bits(near_plane) in as few 9-bit passes as the range
 * needs (fgs_debug_set_option(9, m) selects the variants). temp as for fgs_debug_radix_sort with end_bit 32. */
int32_t fgs_debug_depth_sort(uint32_t* keys0, uint32_t* keys1, uint32_t* vals0, uint32_t* vals1, int32_t n, float near_plane, float far_plane,
                             void* temp, size_t temp_bytes, void* stream);
/* ---- libfgs_hip_dev.so only (built with -DFGS_DEV_SWITCHES: `make -C faster-gaussian-splatting_amd/csrc dev`). The product library has ONE formulation
 * of every kernel and no process-wide switches (every switch below is a compile-time constant there, csrc/fgs_kernels.h: FGS_SWITCH); the A/B tools
 * under tools/ and the variant tests load the dev library. ---- */
#ifdef FGS_DEV_SWITCHES
/* Selects the blend-backward formulation: 3 (default) = live-bucket list + compacted pixels + two-value pipeline state, 2 / 0 =
 * round-1 systolic form (dL/dC from global memory / LDS), 1 = strip (lane = pixel, DPP reductions), 4 = lane = pixel walk with the
 * per-Gaussian sums reduced on the matrix cores (round 4; measured against 3 in profiles/r04_k11m_closeout.txt). A/B switch for tests and
 * bench: process-wide, unsynchronised; all variants must give the same gradients. */
int32_t fgs_debug_set_backward_variant(int32_t variant);
/* Tuning switches for A/B measurements inside one process (process-wide, unsynchronised: bench / test processes only):
 * key 0 = blend-backward variant, 1 = Adam float4 pieces per thread (1, 2, 4), 2 = Adam non-temporal accesses, 3 = fused
 * backward+Adam as one kernel (1, default) or round 1's two (0), 5 = K1 tile counting: 0 flattened (default) or n sequential
 * candidates per lane, 7 = K11 timing experiments (results WRONG: 1 no atomics, 2 no step loop; variant 4: 4 no
 * matrix instructions / write-out, 8 no pair arithmetic),
 * 8 = Adam walks the arenas from the end (1, default) or the start (0), 9 = depth sort: bit 0 key - bits(near) in 9-bit passes, bit 1
 * 2048-item workgroups (1 default; 0 = round 1: 4 x 8 bits, 4096 items; bit 1 measured slower), 10 = forward-blend tile -> workgroup
 * mapping: 252 (default) = one vertical strip of tile columns per XCD, walked row by row from the top (251: from the bottom), 0 = one
 * contiguous band of tile rows per XCD (rounds 1-2), g = 1..64 = groups of g rows dealt to the XCDs in turn, 255 = the bands walked
 * bottom-up, 254 = blocks of tiles weighed and dealt to the XCDs on the device (plan_tiles_kernel), 253 = the bands read through the plan
 * table (blend_forward.hip has the measurements), 11 = 1: rocPRIM scan for the per-tile bucket offsets (and no block plan), 12 = 1: the
 * block plan without sorting (XCD x = block column x), 13 = upper bound of the grid of blend-backward variant 4.
 * Apart from key 7, results never depend on them. */
int32_t fgs_debug_set_option(int32_t key, int32_t value);

#endif /* FGS_DEV_SWITCHES */

#ifdef __cplusplus
}
#endif
#endif /* FGS_HIP_H */
