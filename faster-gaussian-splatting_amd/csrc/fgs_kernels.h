// Launch interface between the C-ABI host layer (api.hip) and the gfx950 kernels. One launcher per pipeline stage;
// stage ids K0..K13 refer to SURVEY.md section 2.1.
#pragma once
#include <hip/hip_runtime.h>
#include <atomic>
#include <cmath>
#include <cstddef>
#include <cstdint>
#include "fgs_config.h"
#include "fgs_math.h"

namespace fgs {

struct PreprocessArgs {                 // K1
    const float* means; const float* scales; const float* rotations; const float* opacities;
    const float* sh0; const float* sh_rest;
    PrimRec* rec; uint32_t* n_touched;
    uint32_t* depth_keys; uint32_t* prim_idx;     // compacted (unsorted) visible list
    uint4* foot;                                   // footprint row of every entry of that list (fgs_math.h), or nullptr (sharded owner: no sort behind K1)
    uint32_t* counters;                            // [0] n_visible, [1] n_instances (one packed 64-bit word), [2] K5 work list, [3] huge list, [4] hot slots
    uint32_t* huge_list;                           // indices of footprints > kHugeFootprint candidate tiles (counted by a second kernel)
    uint32_t* hot_list;                            // [kMaxHot] primitive index of hot-accumulator slot s (counters[4] = slots handed out)
    uint2* ranges; uint32_t n_tiles;               // cleared by the kernel (K0)
    float* acc;                                    // training: K11's accumulator records [N][9]; K1 clears those of the Gaussians it finds visible (else nullptr)
    uint32_t n;
    int seq_tiles;                                 // candidate tiles each lane tests itself before the wave cooperates (1..32)
    int count_appended;                            // sharded path: counters[2] counts the huge-footprint entries appended to the list
    CameraArgs cam;
};
hipError_t launch_preprocess(bool inference, const PreprocessArgs& a, hipStream_t s);
// sharded path: the same Gaussians projected for up to kMaxBatchViews cameras in ONE launch (grid.y = view); a shard is too
// small to fill the chip per view (3 M / 8 Gaussians = 733 workgroups) and 8 back-to-back launches measured 2.2x the time
struct PreprocessBatch { int n_views; PreprocessArgs v[kMaxBatchViews]; };
hipError_t launch_preprocess_batch(const PreprocessBatch& b, hipStream_t s);

// K2-K4: depth sort of the visible list + exclusive scan of per-primitive tile counts in depth order
// depth keys lie in [bits(near), bits(far)] (kf:67): the depth sort orders key - base in `bits` bits (radix_sort.hip)
struct DepthKeyRange { uint32_t base; int bits; };
DepthKeyRange depth_key_range(float near_plane, float far_plane);
size_t depth_sort_temp_bytes(uint32_t n);
// `n_visible` = the visible count, or with n_visible_ptr != nullptr a bound of the count stored there on the device (the sort reads it there: it
// can be enqueued before the host knows the count). foot[0] = the footprint rows in compaction order (K1), foot[1] receives them in depth order,
// tile_counts their tile counts in depth order; vals[selector] = the primitives in depth order.
hipError_t run_depth_sort(void* temp, size_t temp_bytes, uint32_t* keys[2], uint32_t* vals[2], int& selector, uint32_t n_visible,
                          const uint32_t* n_visible_ptr, DepthKeyRange range, uint4* foot[2], uint32_t* tile_counts,
                          uint32_t* big_list, uint32_t* big_count, hipStream_t s);
// K3 + K4: sums of the depth-ordered tile counts per 64-Gaussian wave segment (wave_sums[ceil(n / 64)]) and per 4096-Gaussian block
// (block_sums[ceil(n / 4096)]). K5 derives every Gaussian's first output slot from them.
// n_visible_ptr != nullptr: n_visible is a bound (the primitive count) and the exact count is read on the device
hipError_t launch_tile_count_sums(const uint32_t* tile_counts, uint32_t* wave_sums, uint32_t* block_sums,
                                  uint32_t n_visible, const uint32_t* n_visible_ptr, hipStream_t s);

// K5-K7: instance creation, tile sort, per-tile ranges. key_bytes is 2 (<= 65536 tiles) or 4.
size_t tile_sort_temp_bytes(uint32_t n_instances, int key_bytes, int end_bit);
// The *_ptr forms serve the host-synchronisation-free forward pass: counts are bounds / capacities, the exact ones are read on the device
// (n_visible from counters[0]; the instance count clamped to `capacity` is written to counters[5], an overflow flag to counters[6]).
hipError_t launch_create_instances(int key_bytes, const uint4* foot_sorted, const uint32_t* wave_sums,
                                   const uint32_t* block_sums, uint32_t* offsets,
                                   const PrimRec* rec, void* inst_keys, uint32_t* inst_prims, uint32_t grid_w, uint32_t n_visible,
                                   const uint32_t* n_visible_ptr, uint32_t capacity, uint32_t* counters,
                                   const uint32_t* big_list, const uint32_t* big_count, hipStream_t s);
hipError_t run_tile_sort(void* temp, size_t temp_bytes, int key_bytes, void* keys[2], uint32_t* vals[2], int& selector,
                         uint32_t n_instances, const uint32_t* n_instances_ptr, int end_bit, hipStream_t s);
hipError_t launch_extract_ranges(int key_bytes, const void* sorted_keys, uint2* ranges, uint32_t n_instances, const uint32_t* n_instances_ptr, hipStream_t s);

// A/B switches. In the product library every one of them is a compile-time constant (its adopted value): nothing process-wide can change what a
// launch does. libfgs_hip_dev.so (-DFGS_DEV_SWITCHES) makes them process-wide atomics behind fgs_debug_set_option for the A/B tools; atomics make a
// concurrent set / launch well defined (a launch sees the old or the new value, never a torn one).
#ifdef FGS_DEV_SWITCHES
#define FGS_SWITCH(name, value) inline std::atomic<int> name{value}
#else
#define FGS_SWITCH(name, value) constexpr int name = (value)
#endif
constexpr unsigned kPlannedBlocks = 254u;
constexpr unsigned kColumnsTopDown = 252u, kColumnsBottomUp = 251u;   // one vertical strip of the image per XCD, walked row by row
constexpr unsigned kBandsThroughPlan = 253u;         // A/B only: the round-1 bands, but with the plan's dependent load on every workgroup's path
FGS_SWITCH(g_tile_row_group, static_cast<int>(kColumnsTopDown));   // blend_forward.hip: tile -> workgroup mapping (254: device-side block plan; 0: round-1 bands; 1..64 row groups)
#ifdef FGS_DEV_SWITCHES
FGS_SWITCH(g_k11_chain_waves, 4096);                 // blend_backward.hip, option 14: waves of the chained K11 exhibit (variant 5); 4096 = 16 resident waves x 256 CUs
#endif
FGS_SWITCH(g_plan_experiment, 0);                    // binning.hip, option 12: 1 = blocks unsorted and dealt statically (A/B of the deal itself)
// radix_sort.hip: stable LSD radix sort of (key, uint32) pairs sized for these two sorts
FGS_SWITCH(g_depth_sort_mode, 1);                    // option 9 -- bit 0: key range / 9-bit digits, bit 1: 2048-item workgroups (radix_sort.hip)
size_t own_sort_temp_bytes(uint32_t n, int end_bit);
// Side table carried out of the depth sort's LAST scatter pass (radix_sort.hip): the sort's values start as the input positions (the first pass
// makes them up), the last pass gathers rows_in[value] and writes the row, its first word as the sorted value, and the row's tile count in sorted order.
struct SortPayload { const uint4* rows_in; uint4* rows_out; uint32_t* count_out; int iota_values;
                     uint32_t* big_list; uint32_t* big_count; };      // depth-order positions of the rows whose boxes exceed kBigInstanceFootprint candidates
hipError_t own_depth_sort(void* temp, size_t temp_bytes, uint32_t* keys[2], uint32_t* vals[2], int& selector, uint32_t n, const uint32_t* n_ptr,
                          DepthKeyRange range, hipStream_t s, const SortPayload* payload = nullptr);
hipError_t own_sort_pairs_u32(void* temp, size_t temp_bytes, uint32_t* keys[2], uint32_t* vals[2], int& selector, uint32_t n, int end_bit, hipStream_t s);
hipError_t own_sort_pairs_u16(void* temp, size_t temp_bytes, uint16_t* keys[2], uint32_t* vals[2], int& selector, uint32_t n, int end_bit, hipStream_t s);
// the item count lives on the device (*n_ptr <= capacity): lets the depth sort start before the host has read the counters back
hipError_t own_sort_pairs_u32_device_count(void* temp, size_t temp_bytes, uint32_t* keys[2], uint32_t* vals[2], int& selector, uint32_t capacity,
                                           const uint32_t* n_ptr, int end_bit, hipStream_t s);
hipError_t own_sort_pairs_u16_device_count(void* temp, size_t temp_bytes, uint16_t* keys[2], uint32_t* vals[2], int& selector, uint32_t capacity,
                                           const uint32_t* n_ptr, int end_bit, hipStream_t s);

// K8+K9: inclusive scan of ceil(len/kBucket) per tile, and the tile -> workgroup plan of K10 (one single-workgroup kernel)
hipError_t launch_plan_tiles(const uint2* ranges, uint32_t* bucket_offsets, uint32_t* tile_plan, uint32_t n_tiles, uint32_t grid_w, uint32_t grid_h,
                             hipStream_t s);
#ifdef FGS_DEV_SWITCHES
size_t bucket_scan_temp_bytes(uint32_t n_tiles);      // the library scan, kept for A/B (fgs_debug_set_option(11, 1))
#endif
hipError_t run_bucket_scan(void* temp, size_t temp_bytes, const uint2* ranges, uint32_t* bucket_offsets, uint32_t n_tiles, hipStream_t s);

struct BlendArgs {                      // K10 / inference blend
    const uint2* ranges; const uint32_t* bucket_offsets; const uint32_t* inst_prims; const PrimRec* rec;
    const float* bg; float* image;
    float* final_T; uint32_t* n_processed; uint32_t* max_n_processed;     // tile-major [T][192]
    uint32_t* bucket_tile; float4* ckpt;                                   // [B], [B][192]
    uint32_t width, height, grid_w, n_tiles;
    uint32_t row_group;                      // tile -> workgroup mapping (blend_forward.hip: tile_of_workgroup); set by the launchers
    const uint32_t* tile_plan;               // [kPlanWords] written by plan_tiles_kernel (row_group == kPlannedBlocks)
    uint32_t grid_h;
    int to_chw, clamp_output;
    float* scores;                        // pruning-score mode: accumulated per primitive [N]
};
hipError_t launch_blend(bool training, const BlendArgs& a, hipStream_t s);
hipError_t launch_pruning_scores(const BlendArgs& a, hipStream_t s);   // kernels_pruning_scores.cuh:348-505

struct BlendBackwardArgs {              // K11 (+ per-pixel staging pass)
    const uint2* ranges; const uint32_t* bucket_offsets; const uint32_t* inst_prims; const PrimRec* rec;
    const float* bg; const float* grad_image; const float* image;
    const float* final_T; const uint32_t* n_processed; const uint32_t* max_n_processed;
    const uint32_t* bucket_tile; const float4* ckpt;
    float4* pixrec;                       // [T][192][2] staged per-pixel constants
    float* acc;                           // records [N][9]: d/d(mean2d.x, mean2d.y, conic a,b,c, opacity, colour r,g,b) of each Gaussian, nine consecutive floats
    float* acc_hot;                       // [kHotReplicas][kMaxHot][9]: private records of the hot Gaussians; follows acc in the scratch blob (K11 addresses both as float offsets from acc)
    const uint32_t* hot_list; const uint32_t* hot_count;   // slot -> primitive, number of slots handed out (may exceed kMaxHot)
    uint2* work_list; uint32_t* live_count;   // variant 3: (tile, bucket in tile) of every live bucket and their number
    uint32_t* live_offsets;                   // [T] first list slot of each tile (planning pass -> stage_pixels_kernel)
    uint32_t n, width, height, grid_w, n_tiles, n_buckets_cap;
    // what the staging pass sets to zero: the hot replicas (clear_hot_f4 16-byte pieces from acc_hot), or -- `clear_everything`, or *dirty_flag != 0 --
    // records and replicas (clear_all_f4 pieces from acc). K1 cleared the records of the visible Gaussians during the forward pass (api.hip).
    uint32_t clear_all_f4, clear_hot_f4; int clear_everything;
    uint32_t* dirty_flag;                 // counters[7]: set by the last kernel of a backward pass over these buffers
    int proper_aa;
    int ablate;                           // dev build only: timing experiments (fgs_debug_set_option key 7)
    int variant;                          // K11 formulation, read once per backward pass (blend_backward_variant(); 3 in the product build)
};
hipError_t launch_stage_pixels(const BlendBackwardArgs& a, hipStream_t s);      // per-pixel staging pass
hipError_t launch_blend_backward(const BlendBackwardArgs& a, hipStream_t s);    // K11 proper

struct AdamHyper { float step_size, beta1, beta2, eps, bc2_sqrt_rcp; };

struct BackwardView {                   // what K12 needs per camera view (one on the single-GPU path, up to kMaxBatchViews on the sharded path)
    CameraArgs cam;
    const uint32_t* n_touched;            // [N] tile count of the view, 0 = invisible
    const float* acc;                     // single view: records [N][9] (K11's accumulators). Sharded path (several views per launch):
    const uint32_t* slot;                 //   the returned 9-float accumulator RECORDS, that of visible primitive i being record slot[i]
    float* view_dir;                      // [N][3] scratch: unit view direction of visible primitives, consumed by the SH-rest pass
};
struct PreprocessBackwardArgs {         // K12, optionally fused with K13 for the 14 non-SH-rest floats
    const float* means; const float* scales; const float* rotations; const float* opacities; const float* sh_rest;
    float* grad_means; float* grad_scales; float* grad_rotations; float* grad_opacities; float* grad_sh0;
    float* densification_info;
    uint32_t n;
    int accumulate;                       // unfused only: add into grad_* for visible primitives instead of writing every element
    int n_views; BackwardView view[kMaxBatchViews];   // gradients are summed over the views in registers
    // fused mode (grad_* unused, one view): parameters and moments updated in place, order means, sh0, opacities, scales, rotations
    float* p[5]; float* m[5]; float* v[5]; AdamHyper h[5];
    // single-kernel unfused form only: [ceil(n / 64)] bytes or nullptr -- 1 if any Gaussian of the block 64 b .. 64 b + 63 is visible, 0 if the
    // block's gradients are all zero (they are still written); lets the optimizer skip the read of those zeros (launch_adam)
    uint8_t* live_blocks;
    int vector_ok;                        // set by the launchers: every per-Gaussian tensor of the call is 16-byte aligned (coalesced 16-byte phase A)
};
hipError_t launch_preprocess_backward(bool fused_adam, const PreprocessBackwardArgs& a, hipStream_t s);

struct ShRestView { const float* view_dir; const uint32_t* n_touched; const float* acc; const uint32_t* slot; };   // acc / slot as in BackwardView
struct ShRestArgs {                     // the 45/59 of the per-Gaussian payload, streamed flat and fully coalesced
    int n_views; ShRestView view[kMaxBatchViews];    // n_views > 1 or a slot table = the sharded path (records found through the slot table), else K11's own records [N][9]
    float* grad_sh_rest;                  // unfused: [N][K-1][3] written for every primitive
    float* p; float* m; float* v; AdamHyper h;   // fused (one view)
    uint32_t n; uint32_t total_sh_rest; uint32_t active_sh_bases;
    int accumulate;                       // unfused only: grad_sh_rest += (sharded path, view batches after the first)
};
hipError_t launch_sh_rest_backward(bool fused_adam, const ShRestArgs& a, hipStream_t s);
// single-view fused backward + Adam over all 59 floats of every Gaussian in one kernel (a: fused-mode arguments, sh: the SH-rest group)
hipError_t launch_fused_backward_adam(const PreprocessBackwardArgs& a, const ShRestArgs& sh, hipStream_t s);
// single-view unfused K12 in one kernel: all six gradients written once (sh: grad_sh_rest + SH layout)
hipError_t launch_backward_gradients(const PreprocessBackwardArgs& a, const ShRestArgs& sh, hipStream_t s);

struct AdamGroup { const float* grad; float* param; float* exp_avg; float* exp_avg_sq; int64_t n; AdamHyper h; uint32_t first_block;
                   uint32_t row_len; };      // floats per Gaussian in this tensor (live_blocks only; 0: read every gradient)
// live_blocks: [ceil(N / 64)] bytes or nullptr; 0 = the caller guarantees that the gradient rows of Gaussians 64 b .. 64 b + 63 are zero:
// they are not read (one third of the Gaussians is invisible per view, 85 % of them in such blocks: tools/dead_blocks.py)
struct AdamArgs { AdamGroup g[8]; int n_groups; uint32_t total_blocks; int reverse; const uint8_t* live_blocks; };
hipError_t launch_adam(const AdamArgs& a, hipStream_t s);   // K13, all groups in one launch

struct LossArgs {                       // fused L1 + DSSIM loss and its image gradient (loss.hip)
    const float* image; const float* target;   // [3,H,W]
    float* sums;                                 // [0] sum |x-y|, [1] sum SSIM   (zeroed by the host before the launch)
    float* grad;                                 // [3,H,W] or nullptr
    float* d_mu; float* d_m11; float* d_m12;     // scratch derivative maps, [3,H,W] each
    float* partials;                             // scratch: 2 floats per workgroup of the forward kernel
    int width, height;
    float lambda_l1, lambda_dssim;
    const float* upstream;                       // device scalar dL/dloss the gradient is multiplied with, or nullptr (= 1)
};
size_t l1_dssim_partials(int width, int height);   // number of floats in LossArgs::partials
hipError_t launch_l1_dssim(const LossArgs& a, hipStream_t s);
hipError_t launch_l1_dssim_backward(const LossArgs& a, hipStream_t s);   // gradient only, from the maps a forward launch left in d_mu / d_m11 / d_m12

// shard_exchange.hip: record (un)packing either side of the two exchanges of the Gaussian-sharded multi-GPU path
struct PackRecordsView { const PrimRec* rec; const uint32_t* n_touched; const uint32_t* depth_keys; const uint32_t* prim_idx;
                         const uint32_t* counters; uint32_t* slot; uint32_t* out; uint32_t* counts_out; };
struct PackRecordsBatch { int n_views; uint32_t capacity; PackRecordsView v[kMaxBatchViews]; };
hipError_t launch_pack_splat_records(const PackRecordsBatch& b, hipStream_t s);
// how the renderer's record concatenation is made of the shards' segments (n_shards <= 1: the records keep their order)
struct ShardOrder { int n_shards; uint32_t count[kMaxBatchViews]; };
hipError_t launch_unpack_splat_records(const uint32_t* records, uint32_t n, PrimRec* rec, uint32_t* n_touched, uint32_t* depth_keys,
                                       uint32_t* prim_idx, uint4* foot, uint2* ranges, uint32_t n_tiles, uint32_t* hot_list, uint32_t* hot_count, const ShardOrder& order,
                                       hipStream_t s);
hipError_t launch_pack_acc(const float* acc, uint32_t n, float* out, const ShardOrder& order, hipStream_t s);

// aux_ops.hip: the reference's remaining exported operators
hipError_t launch_update_3d_filter(const float* positions, const float* w2c, float* filter_3d, uint8_t* visibility_mask, int n,
                                   float left, float right, float top, float bottom, float near_plane, float distance2filter, hipStream_t s);
void relocation_coefficients(float* out /*[50*50] host*/);
hipError_t launch_relocation(const float* old_opacities, const float* old_scales, const int64_t* n_samples, const float* table_device,
                             float* new_opacities, float* new_scales, unsigned n, hipStream_t s);
hipError_t launch_add_noise(const float* raw_scales, const float* raw_rotations, const float* raw_opacities, const float* random_samples,
                            float* means, unsigned n, float current_lr, hipStream_t s);

// densify.hip: maintenance of the Gaussian set (adaptive density control, prune / sort gathers, Morton order) with the Adam moments
struct AdcPlanArgs {
    const float* densification_info;      // [2, N]
    const float* scales; const float* rotations; const float* opacities;
    uint32_t n;
    float grad_threshold, min_opacity_logit, log_small, log_large; int prune_large;
    uint32_t* plan; uint4* offsets; uint32_t* totals;      // scratch: [N], [N], [4] = survivors, clones, children per copy, split
    void* scan_temp; size_t scan_temp_bytes;
};
size_t adc_scan_temp_bytes(uint32_t n);
hipError_t launch_adc_plan(const AdcPlanArgs& a, hipStream_t s);
struct AdcScatterArgs {
    const float* in_p; const float* in_m; const float* in_v;     // moments may be NULL (no optimizer state yet)
    float* out_p; float* out_m; float* out_v;
    const float* scales; const float* rotations; const float* noise;   // KIND 1 (means) only; noise [2 * n_split, 3]
    const uint32_t* plan; const uint4* offsets; const uint32_t* totals;
    uint32_t n, width;
};
hipError_t launch_adc_scatter(int kind, const AdcScatterArgs& a, hipStream_t s);
constexpr int kGatherTensors = 18;
struct GatherTensor { const float* in; float* out; uint32_t width; uint32_t first_block; };
struct GatherArgs { GatherTensor t[kGatherTensors]; int n_tensors; uint32_t n_rows; const int64_t* index; };
hipError_t launch_gather_rows(const GatherArgs& a, hipStream_t s);
size_t morton_temp_bytes(uint32_t n);
hipError_t run_morton_order(const float* means, const float* lo, const float* hi, int64_t* order_out, uint32_t n, void* temp, size_t temp_bytes, hipStream_t s);

FGS_SWITCH(g_adam_reverse, 1);                                               // option 8: reversed workgroup order (preprocess_backward.hip)
FGS_SWITCH(g_adam_nontemporal, 1);                                           // option 2: non-temporal loads / stores
FGS_SWITCH(g_adam_unroll, 1);                                                // option 1: 1, 2 or 4 float4 pieces per thread
#ifdef FGS_DEV_SWITCHES
extern std::atomic<int> g_backward_ablate;
extern std::atomic<int> g_k11m_max_blocks;
extern std::atomic<int> g_backward_variant;                                  // 3 compact (default), 0 / 2 systolic, 1 strip, 4 lane = pixel (blend_backward.hip)
#endif
int blend_backward_variant();                                                // the K11 formulation of this pass (always 3 in the product build)
hipError_t launch_wave_selftest(uint32_t* out /*[4*64]*/, hipStream_t s);

}  // namespace fgs
