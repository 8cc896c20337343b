// C-ABI entry points of libfgs_hip.so (declared in include/fgs_hip.h) and the host-side orchestration of the pipeline.
// Replaces the reference's C++ wrappers + host code: rasterization_api.cu:13-247, rasterization/src/forward.cu:11-259,
// backward.cu:8-125, inference.cu:11-226, adam/src/adam.cu:36-71.
//
// Host-side differences that are deliberate (MI355X-first), not omissions:
//  * every launch goes to the caller's hipStream_t (the reference uses the legacy default stream + a static side stream);
//  * ONE device->host read per forward (n_visible, n_instances through pinned memory) instead of three blocking copies:
//    the bucket buffer is sized by the bound B <= I/64 + min(T, I) and kernels read the exact bucket count on the device;
//  * no zero-fill of the 59-float gradients (the backward kernels write every element), only the 9-float atomic
//    accumulators are cleared.
#include <fgs_hip.h>
#include "fgs_kernels.h"
#include <cstdarg>
#include <cstdio>
#include <cmath>
#include <cstring>
#include <vector>

using namespace fgs;

namespace {

thread_local char g_error[512] = "";

int fail(int code, const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_error, sizeof(g_error), fmt, ap);
    va_end(ap);
    return code;
}
#define FGS_HIP(expr)                                                                                       \
    do {                                                                                                    \
        hipError_t e_ = (expr);                                                                             \
        if (e_ != hipSuccess) return fail(FGS_ERR_HIP, "%s failed: %s", #expr, hipGetErrorString(e_));       \
    } while (0)

// ---- optional per-stage timing with HIP events recorded on the caller's stream (fgs_profile_enable / fgs_profile_read) ----
enum Stage { ST_PREPROCESS, ST_DEPTH_SORT, ST_OFFSETS_SCAN, ST_CREATE_INSTANCES, ST_TILE_SORT, ST_RANGES, ST_BUCKET_SCAN,
             ST_BLEND_FORWARD, ST_STAGE_PIXELS, ST_BLEND_BACKWARD, ST_PREPROCESS_BACKWARD, ST_SH_REST_BACKWARD, ST_ADAM, ST_LOSS, ST_RECORDS, ST_FUSED_BACKWARD_ADAM, ST_COUNT };
const char* const kStageNames[ST_COUNT] = {"preprocess", "depth_sort", "offsets_scan", "create_instances", "tile_sort", "extract_ranges",
                                           "bucket_scan", "blend_forward", "stage_pixels", "blend_backward", "preprocess_backward",
                                           "sh_rest_backward", "adam", "l1_dssim_loss", "shard_records", "fused_backward_adam"};
struct StageRecord { int stage; hipEvent_t start, stop; };
struct Profiler {
    bool enabled = false;
    int only = -1;                     // >= 0: record this stage only (two events per launch of ONE stage perturb a timed loop far less than 30)
    std::vector<StageRecord> records;
    std::vector<hipEvent_t> pool;
    hipEvent_t get() {
        if (!pool.empty()) { hipEvent_t e = pool.back(); pool.pop_back(); return e; }
        hipEvent_t e = nullptr;
        if (hipEventCreate(&e) != hipSuccess) return nullptr;
        return e;
    }
};
Profiler g_prof;
struct StageScope {                    // records start now and stop at scope exit, both on `stream`
    hipStream_t stream; int idx = -1;
    StageScope(int stage, hipStream_t s) : stream(s) {
        if (!g_prof.enabled || (g_prof.only >= 0 && g_prof.only != stage)) return;
        StageRecord r{stage, g_prof.get(), g_prof.get()};
        if (!r.start || !r.stop) return;
        (void)hipEventRecord(r.start, stream);
        g_prof.records.push_back(r);
        idx = static_cast<int>(g_prof.records.size()) - 1;
    }
    ~StageScope() { if (idx >= 0) (void)hipEventRecord(g_prof.records[idx].stop, stream); }
};

// bu:10-18
int extract_end_bit(uint32_t n) {
    if (n == 0) return 1;             // the reference's bit-twiddling version yields 1 for a single tile
    int bits = 0;
    while (n != 0) { ++bits; n >>= 1; }
    return bits;
}

struct Carver {                       // 256-byte aligned bump allocation inside a caller-owned byte buffer (cf. bu:30-36)
    char* base; size_t off = 0;
    fgs_blob_entry* entries; int max_entries; int n = 0;
    explicit Carver(void* b, fgs_blob_entry* e = nullptr, int m = 0) : base(static_cast<char*>(b)), entries(e), max_entries(m) {}
    template <typename T> T* take(const char* name, size_t count) {
        off = (off + 255) & ~static_cast<size_t>(255);
        T* p = base ? reinterpret_cast<T*>(base + off) : nullptr;
        if (entries && n < max_entries) { entries[n].name = name; entries[n].offset = off; entries[n].bytes = count * sizeof(T); }
        ++n;
        off += count * sizeof(T);
        return p;
    }
    size_t total() const { return (off + 255) & ~static_cast<size_t>(255); }
};

constexpr size_t kCounterWords = 8;      // PreprocessArgs::counters
struct Geometry { uint32_t grid_w, grid_h, n_tiles; int end_bit, key_bytes; };
Geometry geometry_of(int width, int height) {
    Geometry g;
    g.grid_w = (width + kTileW - 1) / kTileW;
    g.grid_h = (height + kTileH - 1) / kTileH;
    g.n_tiles = g.grid_w * g.grid_h;
    g.end_bit = extract_end_bit(g.n_tiles - 1);          // fwd:42
    g.key_bytes = g.end_bit <= 16 ? 2 : 4;                // fwd:152-153
    return g;
}

struct PrimitiveBuffers {             // cf. bu:45-94
    PrimRec* rec; uint32_t* n_touched; uint32_t* keys[2]; uint32_t* prims[2]; uint32_t* offsets; uint32_t* counters; uint32_t* hot_list;
    uint4* foot[2]; uint32_t* tile_counts;      // footprint rows in compaction / depth order, tile counts in depth order (fgs_math.h, radix_sort.hip)
    uint32_t* wave_sums; uint32_t* block_sums;  // their sums per 64-Gaussian wave segment / per 4096-Gaussian block (binning.hip)
    uint32_t* big_list;                         // depth-order positions of the footprints of more than kBigInstanceFootprint candidate tiles (counters[2] of them)
    char* temp; size_t temp_bytes;
    // K11's accumulator records [N][9] and the hot Gaussians' replicas behind them (K11 addresses both as 32-bit float offsets from `acc`). Round 6:
    // they live HERE, at the end of the forward pass's primitive blob, not in the backward scratch -- K1 clears the record of every Gaussian it
    // finds visible on the side of its own (latency-bound) work, so the backward pass starts without a 117 MB clear on its critical path. Passes
    // that are never differentiated (inference, pruning scores, the sharded owner's K1) carve the blob without them.
    float* acc; float* acc_hot;
    static constexpr size_t kHotFloats = (size_t)kHotReplicas * 9 * kMaxHot;
    static PrimitiveBuffers carve(Carver& c, uint32_t n, bool with_acc = true) {
        PrimitiveBuffers b;
        b.rec = c.take<PrimRec>("rec", n);
        b.n_touched = c.take<uint32_t>("n_touched", n);
        b.keys[0] = c.take<uint32_t>("depth_keys0", n); b.keys[1] = c.take<uint32_t>("depth_keys1", n);
        b.prims[0] = c.take<uint32_t>("prim_idx0", n); b.prims[1] = c.take<uint32_t>("prim_idx1", n);
        b.offsets = c.take<uint32_t>("offsets", n);
        b.counters = c.take<uint32_t>("counters", kCounterWords);
        b.hot_list = c.take<uint32_t>("hot_list", kMaxHot);
        b.foot[0] = c.take<uint4>("foot0", n); b.foot[1] = c.take<uint4>("foot1", n);
        b.tile_counts = c.take<uint32_t>("tile_counts", n);
        b.wave_sums = c.take<uint32_t>("wave_sums", (static_cast<size_t>(n) + 63) / 64 + 64);
        b.block_sums = c.take<uint32_t>("block_sums", (static_cast<size_t>(n) + 4095) / 4096 + 1);
        b.big_list = c.take<uint32_t>("big_list", n);
        b.temp_bytes = depth_sort_temp_bytes(n);
        b.temp = c.take<char>("sort_temp", b.temp_bytes);
        b.acc = with_acc ? c.take<float>("acc", (size_t)n * 9) : nullptr;
        b.acc_hot = with_acc ? c.take<float>("acc_hot", kHotFloats) : nullptr;
        return b;
    }
};
struct TileBuffers {                  // cf. bu:126-152; final_T / n_processed are tile-major here
    uint2* ranges; uint32_t* bucket_offsets; uint32_t* max_n_processed; float* final_T; uint32_t* n_processed;
    uint32_t* tile_plan;              // K10's tile -> workgroup plan (plan_tiles_kernel)
    uint32_t* live_count;             // backward: number of live buckets (K11 planning pass)
    uint32_t* live_offsets;           // backward: first slot of each tile in the live-bucket list
    char* temp; size_t temp_bytes;
    static TileBuffers carve(Carver& c, uint32_t t, bool training) {
        TileBuffers b{};
        b.ranges = c.take<uint2>("ranges", t);
        b.bucket_offsets = c.take<uint32_t>("bucket_offsets", t);         // inference too: the plan's block weights are differences of this scan
        b.tile_plan = c.take<uint32_t>("tile_plan", kPlanWords);
        if (training) {
            b.max_n_processed = c.take<uint32_t>("max_n_processed", t);
            b.final_T = c.take<float>("final_T", (size_t)t * kTilePixels);
            b.n_processed = c.take<uint32_t>("n_processed", (size_t)t * kTilePixels);
#ifdef FGS_DEV_SWITCHES
            b.temp_bytes = bucket_scan_temp_bytes(t);      // the library scan, an A/B option of the dev build
#else
            b.temp_bytes = 0;
#endif
            b.temp = c.take<char>("scan_temp", b.temp_bytes);
            b.live_count = c.take<uint32_t>("live_count", 4);
            b.live_offsets = c.take<uint32_t>("live_offsets", t);
        }
        return b;
    }
};
struct InstanceBuffers {              // cf. bu:96-124
    void* keys[2]; uint32_t* prims[2]; char* temp; size_t temp_bytes;
    static InstanceBuffers carve(Carver& c, uint32_t n, int key_bytes, int end_bit) {
        InstanceBuffers b;
        b.keys[0] = c.take<char>("keys0", (size_t)n * key_bytes); b.keys[1] = c.take<char>("keys1", (size_t)n * key_bytes);
        b.prims[0] = c.take<uint32_t>("prims0", n); b.prims[1] = c.take<uint32_t>("prims1", n);
        b.temp_bytes = tile_sort_temp_bytes(n, key_bytes, end_bit);
        b.temp = c.take<char>("sort_temp", b.temp_bytes);
        return b;
    }
};
struct BucketBuffers {                // cf. bu:154-163
    uint32_t* tile_index; float4* ckpt; uint2* work_list;
    static BucketBuffers carve(Carver& c, uint32_t n) {
        BucketBuffers b;
        b.tile_index = c.take<uint32_t>("tile_index", n);
        b.ckpt = c.take<float4>("ckpt", (size_t)n * kTilePixels);
        b.work_list = c.take<uint2>("work_list", n);        // backward: the live (tile, bucket) pairs
        return b;
    }
};
struct BackwardScratch {             // (K11's accumulator records moved into the primitive blob in round 6: PrimitiveBuffers::acc)
    float* view_dir; float4* pixrec;
    static BackwardScratch carve(Carver& c, uint32_t n, uint32_t t) {
        BackwardScratch b;
        b.view_dir = c.take<float>("view_dir", (size_t)n * 3);
        b.pixrec = c.take<float4>("pixrec", (size_t)t * kTilePixels * 2);
        return b;
    }
};

FGS_SWITCH(g_seq_tiles, kSeqTiles);                // fgs_debug_set_option key 5 (dev build; a constant in the product, like every switch: fgs_kernels.h)
#ifdef FGS_DEV_SWITCHES
std::atomic<int> g_library_bucket_scan{0};         // fgs_debug_set_option key 11: 1 = rocPRIM scan for K8+K9 and no tile plan (round-2 form, A/B)
#endif
FGS_SWITCH(g_fused_single_kernel, 1);               // fgs_debug_set_option key 3: K12 / fused K12+K13 of the single-GPU path as one kernel (1) or as round 1's two (0)

uint32_t bucket_capacity(uint32_t n_instances, uint32_t n_tiles) {   // sum_t ceil(len_t/64) <= I/64 + #non-empty tiles
    return n_instances / kBucket + (n_instances < n_tiles ? n_instances : n_tiles);
}

CameraArgs camera_of(const fgs_settings& s, const Geometry& g) {
    CameraArgs c;
    c.w2c = s.w2c; c.cam_pos = s.cam_position;
    c.width = static_cast<float>(s.width); c.height = static_cast<float>(s.height);   // fwd:82-83
    c.fx = s.focal_x; c.fy = s.focal_y; c.cx = s.center_x; c.cy = s.center_y;
    c.near_plane = s.near_plane; c.far_plane = s.far_plane; c.proper_aa = s.proper_antialiasing ? 1 : 0;
    c.active_sh_bases = s.active_sh_bases; c.total_sh_rest = s.total_sh_bases_rest;
    c.grid_w = g.grid_w; c.grid_h = g.grid_h;
    return c;
}

BackwardView backward_view(const fgs_settings& s, const Geometry& g, const uint32_t* n_touched, const uint32_t* slot, const float* acc, float* view_dir) {
    BackwardView v;
    v.cam = camera_of(s, g); v.n_touched = n_touched; v.slot = slot; v.acc = acc; v.view_dir = view_dir;
    return v;
}
ShRestView sh_rest_view(const BackwardView& b) {
    ShRestView v;
    v.view_dir = b.view_dir; v.n_touched = b.n_touched; v.slot = b.slot; v.acc = b.acc;
    return v;
}

int check_settings(const fgs_settings* s) {
    if (!s) return fail(FGS_ERR_INVALID_ARGUMENT, "settings is NULL");
    if (!s->w2c || !s->cam_position || !s->bg_color) return fail(FGS_ERR_INVALID_ARGUMENT, "w2c / cam_position / bg_color must be device pointers");
    if (s->width <= 0 || s->height <= 0) return fail(FGS_ERR_INVALID_ARGUMENT, "image size %dx%d", s->width, s->height);
    if (s->active_sh_bases < 1 || s->active_sh_bases > 16) return fail(FGS_ERR_INVALID_ARGUMENT, "active_sh_bases %d", s->active_sh_bases);
    if (s->total_sh_bases_rest > 15) return fail(FGS_ERR_INVALID_ARGUMENT, "sh_coefficients_rest has %d bases (SH degree 3 = 15 is the maximum)", s->total_sh_bases_rest);
    if (s->total_sh_bases_rest < 0 || (s->active_sh_bases > 1 && s->total_sh_bases_rest < s->active_sh_bases - 1))
        return fail(FGS_ERR_INVALID_ARGUMENT, "sh_coefficients_rest has %d bases, active_sh_bases %d", s->total_sh_bases_rest, s->active_sh_bases);
    return FGS_OK;
}

// The one D2H read of a forward pass goes through 16 bytes of pinned host memory and an event; both belong to the device that
// was current when they were created, so they are kept per (host thread, device) -- a thread driving two GPUs gets two sets.
constexpr int kMaxDevices = 64;
struct CounterReadback { uint32_t* host = nullptr; hipEvent_t ready = nullptr; };
CounterReadback* counter_readback() {
    thread_local CounterReadback slots[kMaxDevices];
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= kMaxDevices) return nullptr;
    CounterReadback& c = slots[dev];
    if (!c.host && hipHostMalloc(reinterpret_cast<void**>(&c.host), 16, hipHostMallocDefault) != hipSuccess) c.host = nullptr;
    if (!c.ready && hipEventCreateWithFlags(&c.ready, hipEventDisableTiming) != hipSuccess) c.ready = nullptr;
    return (c.host && c.ready) ? &c : nullptr;
}

AdamHyper adam_hyper(int step, double lr, double beta1, double beta2, double eps) {   // adam.cu:52-54
    const double bc1_rcp = 1.0 / (1.0 - std::pow(beta1, step));
    const double bc2_sqrt_rcp = 1.0 / std::sqrt(1.0 - std::pow(beta2, step));
    AdamHyper h;
    h.step_size = static_cast<float>(lr * bc1_rcp);
    h.beta1 = static_cast<float>(beta1); h.beta2 = static_cast<float>(beta2); h.eps = static_cast<float>(eps);
    h.bc2_sqrt_rcp = static_cast<float>(bc2_sqrt_rcp);
    return h;
}

// shared by fgs_forward (training) and fgs_inference
enum ForwardMode { MODE_TRAINING, MODE_INFERENCE, MODE_SCORES };

int forward_tail(ForwardMode mode, const PrimitiveBuffers& pb_in, const TileBuffers& tb, const Geometry& geo, uint32_t n_visible,
                 uint32_t n_instances, int depth_sel, const fgs_settings* settings, float* image, int to_chw, int clamp_output,
                 fgs_resize_fn resize, void* user, fgs_forward_state* state_out, hipStream_t stream, float* scores, bool device_counts = false);

int run_forward(ForwardMode mode, const float* means, const float* scales, const float* rotations, const float* opacities,
                const float* sh0, const float* sh_rest, int32_t n_primitives, const fgs_settings* settings, float* image,
                int to_chw, int clamp_output, fgs_resize_fn resize, void* user, fgs_forward_state* state_out, void* stream_,
                float* scores = nullptr, int32_t instance_capacity = 0) {
    const bool training = mode == MODE_TRAINING;
    if (int rc = check_settings(settings)) return rc;
    if (n_primitives < 0 || (!image && mode != MODE_SCORES) || (!scores && mode == MODE_SCORES) || !resize || !state_out) return fail(FGS_ERR_INVALID_ARGUMENT, "bad argument (n_primitives=%d)", n_primitives);
    if (n_primitives > 0 && (!means || !scales || !rotations || !opacities || !sh0 || (settings->total_sh_bases_rest > 0 && !sh_rest)))
        return fail(FGS_ERR_INVALID_ARGUMENT, "NULL parameter tensor");
    hipStream_t stream = static_cast<hipStream_t>(stream_);
    const uint32_t n = static_cast<uint32_t>(n_primitives);
    const Geometry geo = geometry_of(settings->width, settings->height);

    // tile buffers + K0 (fwd:44-55)
    Carver tile_size(nullptr);
    TileBuffers::carve(tile_size, geo.n_tiles, training);
    void* tile_blob = resize(user, FGS_BUF_TILE, tile_size.total());
    if (!tile_blob && tile_size.total() > 0) return fail(FGS_ERR_ALLOC, "resize(tile, %zu) returned NULL", tile_size.total());
    Carver tile_c(tile_blob);
    TileBuffers tb = TileBuffers::carve(tile_c, geo.n_tiles, training);

    // primitive buffers + K1 (fwd:58-98)
    Carver prim_size(nullptr);
    PrimitiveBuffers::carve(prim_size, n, training);
    void* prim_blob = resize(user, FGS_BUF_PRIMITIVE, prim_size.total());
    if (!prim_blob && prim_size.total() > 0) return fail(FGS_ERR_ALLOC, "resize(primitive, %zu) returned NULL", prim_size.total());
    Carver prim_c(prim_blob);
    PrimitiveBuffers pb = PrimitiveBuffers::carve(prim_c, n, training);
    FGS_HIP(hipMemsetAsync(pb.counters, 0, kCounterWords * sizeof(uint32_t), stream));      // incl. counters[7]: "the accumulator records are dirty"
    PreprocessArgs pa{};
    pa.acc = pb.acc;                                   // training: K1 clears the accumulator record of every visible Gaussian
    pa.means = means; pa.scales = scales; pa.rotations = rotations; pa.opacities = opacities; pa.sh0 = sh0; pa.sh_rest = sh_rest;
    pa.rec = pb.rec; pa.n_touched = pb.n_touched; pa.depth_keys = pb.keys[0]; pa.prim_idx = pb.prims[0]; pa.counters = pb.counters; pa.huge_list = pb.offsets;   // `offsets` is free until the K4 scan writes it
    pa.hot_list = pb.hot_list; pa.foot = pb.foot[0];
    pa.n = n; pa.cam = camera_of(*settings, geo); pa.ranges = tb.ranges; pa.n_tiles = geo.n_tiles; pa.seq_tiles = g_seq_tiles;
    if (n == 0) FGS_HIP(hipMemsetAsync(tb.ranges, 0, sizeof(uint2) * geo.n_tiles, stream));   // no preprocess launch to clear them
    { StageScope t(ST_PREPROCESS, stream); FGS_HIP(launch_preprocess(!training, pa, stream)); }

    if (instance_capacity > 0) {
        // Host-synchronisation-free form (fgs_forward_async): nothing is read back. Every launch behind K1 is sized by a bound -- the
        // primitive count for the visible list, the caller's capacity for the instance stages -- and reads the exact count on the device.
        int depth_sel = 0;
        if (n > 0) {
            StageScope t(ST_DEPTH_SORT, stream);
            FGS_HIP(run_depth_sort(pb.temp, pb.temp_bytes, pb.keys, pb.prims, depth_sel, n, pb.counters, depth_key_range(settings->near_plane, settings->far_plane), pb.foot, pb.tile_counts, pb.big_list, pb.counters + 2, stream));
        }
        return forward_tail(mode, pb, tb, geo, n, static_cast<uint32_t>(instance_capacity), depth_sel, settings, image, to_chw, clamp_output, resize, user,
                            state_out, stream, scores, true);
    }

    // the one host read of the pass: V and I (fwd:99-102). The depth sort does not need them on the host (radix_sort.hip reads
    // the count on the device), so it is enqueued BEHIND the copy and runs while the host waits for the two words.
    CounterReadback* rb = counter_readback();
    if (!rb) return fail(FGS_ERR_HIP, "pinned memory / event for the counter read-back unavailable on the current device");
    uint32_t* host = rb->host;
    hipEvent_t ready = rb->ready;
    FGS_HIP(hipMemcpyAsync(host, pb.counters, 2 * sizeof(uint32_t), hipMemcpyDeviceToHost, stream));
    FGS_HIP(hipEventRecord(ready, stream));
    int depth_sel = -1;
    if (n > 0) {
        StageScope t(ST_DEPTH_SORT, stream);
        FGS_HIP(run_depth_sort(pb.temp, pb.temp_bytes, pb.keys, pb.prims, depth_sel, n, pb.counters, depth_key_range(settings->near_plane, settings->far_plane), pb.foot, pb.tile_counts, pb.big_list, pb.counters + 2, stream));
    }
    FGS_HIP(hipEventSynchronize(ready));
    const uint32_t n_visible = host[0], n_instances = host[1];

    return forward_tail(mode, pb, tb, geo, n_visible, n_instances, depth_sel, settings, image, to_chw, clamp_output, resize, user, state_out, stream, scores);
}

// K2..K10 over a filled primitive buffer (rec, n_touched, depth keys + indices of the n_visible visible entries; depth_sel >= 0:
// already depth-sorted, the sorted half is depth_sel)
int forward_tail(ForwardMode mode, const PrimitiveBuffers& pb_in, const TileBuffers& tb, const Geometry& geo, uint32_t n_visible,
                 uint32_t n_instances, int depth_sel, const fgs_settings* settings, float* image, int to_chw, int clamp_output,
                 fgs_resize_fn resize, void* user, fgs_forward_state* state_out, hipStream_t stream, float* scores, bool device_counts) {
    const bool training = mode == MODE_TRAINING;
    PrimitiveBuffers pb = pb_in;
    // device_counts: n_visible / n_instances are BOUNDS (primitive count / caller's instance capacity); the exact counts stay on the device:
    // counters[0] = visible, counters[5] = min(instances, capacity) (written by K5), counters[6] = the capacity was exceeded
    const uint32_t* const visible_ptr = device_counts ? pb.counters : nullptr;
    const uint32_t* const instances_ptr = device_counts ? pb.counters + 5 : nullptr;
    // K2-K4 (fwd:104-127)
    if (depth_sel < 0) { StageScope t(ST_DEPTH_SORT, stream); FGS_HIP(run_depth_sort(pb.temp, pb.temp_bytes, pb.keys, pb.prims, depth_sel, n_visible, visible_ptr, depth_key_range(settings->near_plane, settings->far_plane), pb.foot, pb.tile_counts, pb.big_list, pb.counters + 2, stream)); }
    { StageScope t(ST_OFFSETS_SCAN, stream); FGS_HIP(launch_tile_count_sums(pb.tile_counts, pb.wave_sums, pb.block_sums, n_visible, visible_ptr, stream)); }

    // K5-K7 (fwd:179-216)
    Carver inst_size(nullptr);
    InstanceBuffers::carve(inst_size, n_instances, geo.key_bytes, geo.end_bit);
    void* inst_blob = resize(user, FGS_BUF_INSTANCE, inst_size.total());
    if (!inst_blob && inst_size.total() > 0) return fail(FGS_ERR_ALLOC, "resize(instance, %zu) returned NULL", inst_size.total());
    Carver inst_c(inst_blob);
    InstanceBuffers ib = InstanceBuffers::carve(inst_c, n_instances, geo.key_bytes, geo.end_bit);
    { StageScope t(ST_CREATE_INSTANCES, stream); FGS_HIP(launch_create_instances(geo.key_bytes, pb.foot[1], pb.wave_sums, pb.block_sums, pb.offsets, pb.rec, ib.keys[0], ib.prims[0], geo.grid_w, n_visible,
                                                                                 visible_ptr, device_counts ? n_instances : 0xffffffffu, pb.counters,
                                                                                 pb.big_list, pb.counters + 2, stream)); }
    int tile_sel = 0;
    { StageScope t(ST_TILE_SORT, stream); FGS_HIP(run_tile_sort(ib.temp, ib.temp_bytes, geo.key_bytes, ib.keys, ib.prims, tile_sel, n_instances, instances_ptr, geo.end_bit, stream)); }
    // the key double buffer flips together with the value double buffer
    { StageScope t(ST_RANGES, stream); FGS_HIP(launch_extract_ranges(geo.key_bytes, ib.keys[tile_sel], tb.ranges, n_instances, instances_ptr, stream)); }

    BlendArgs ba{};
    ba.ranges = tb.ranges; ba.inst_prims = ib.prims[tile_sel]; ba.rec = pb.rec; ba.bg = settings->bg_color; ba.image = image;
    ba.width = settings->width; ba.height = settings->height; ba.grid_w = geo.grid_w; ba.n_tiles = geo.n_tiles;
    ba.to_chw = to_chw; ba.clamp_output = clamp_output;
    uint32_t n_buckets_cap = 0;
    // The tile -> workgroup mapping is read ONCE per pass and travels in BlendArgs, so that planning and launch see the same value (it is a
    // process-wide A/B switch another thread may flip). K8+K9 (fwd:218-231) and K10's optional block plan are one single-workgroup kernel.
    const uint32_t row_group = static_cast<uint32_t>(static_cast<int>(fgs::g_tile_row_group));
    const bool need_plan = row_group == kPlannedBlocks || row_group == kBandsThroughPlan;     // A/B mappings that read a device-side table
    const bool need_scan = training || need_plan;                                             // per-tile bucket offsets: the training blend's checkpoints
    ba.row_group = row_group;
    if (need_scan) {
        StageScope t(ST_BUCKET_SCAN, stream);
#ifdef FGS_DEV_SWITCHES
        if (g_library_bucket_scan && training) {
            FGS_HIP(run_bucket_scan(tb.temp, tb.temp_bytes, tb.ranges, tb.bucket_offsets, geo.n_tiles, stream));      // (A/B: rocPRIM scan, no plan)
        } else
#endif
        {
            FGS_HIP(launch_plan_tiles(tb.ranges, tb.bucket_offsets, need_plan ? tb.tile_plan : nullptr, geo.n_tiles, geo.grid_w, geo.grid_h, stream));
            ba.tile_plan = need_plan ? tb.tile_plan : nullptr;
        }
    }
    ba.grid_h = geo.grid_h;
    if (training) {
        // the bucket buffer sized by its bound (no read-back of n_buckets, fwd:234)
        n_buckets_cap = bucket_capacity(n_instances, geo.n_tiles);
        Carver bucket_size(nullptr);
        BucketBuffers::carve(bucket_size, n_buckets_cap);
        void* bucket_blob = resize(user, FGS_BUF_BUCKET, bucket_size.total());
        if (!bucket_blob && bucket_size.total() > 0) return fail(FGS_ERR_ALLOC, "resize(bucket, %zu) returned NULL", bucket_size.total());
        Carver bucket_c(bucket_blob);
        BucketBuffers bb = BucketBuffers::carve(bucket_c, n_buckets_cap);
        ba.bucket_offsets = tb.bucket_offsets; ba.final_T = tb.final_T; ba.n_processed = tb.n_processed;
        ba.max_n_processed = tb.max_n_processed; ba.bucket_tile = bb.tile_index; ba.ckpt = bb.ckpt;
    }
    if (mode == MODE_SCORES) { ba.scores = scores; StageScope t(ST_BLEND_FORWARD, stream); FGS_HIP(launch_pruning_scores(ba, stream)); }
    else { StageScope t(ST_BLEND_FORWARD, stream); FGS_HIP(launch_blend(training, ba, stream)); }   // K10 (fwd:239)

    state_out->n_visible = static_cast<int32_t>(n_visible);
    state_out->n_instances = static_cast<int32_t>(n_instances);
    state_out->n_buckets = static_cast<int32_t>(n_buckets_cap);
    state_out->selector = tile_sel;
    return FGS_OK;
}

struct BackwardPlan {
    Geometry geo; PrimitiveBuffers pb; TileBuffers tb; InstanceBuffers ib; BucketBuffers bb; BackwardScratch sc;
};

int plan_backward(BackwardPlan& P, void* prim_blob, void* tile_blob, void* inst_blob, void* bucket_blob, void* scratch,
                  int32_t n_primitives, const fgs_settings* settings, const fgs_forward_state* state) {
    if (int rc = check_settings(settings)) return rc;
    if (!state || n_primitives < 0) return fail(FGS_ERR_INVALID_ARGUMENT, "bad state / n_primitives");
    if (static_cast<uint64_t>(n_primitives) * kAccRecordWords + PrimitiveBuffers::kHotFloats > 0xfffffff0ull)      // K11 addresses the accumulator records by 32-bit float offsets
        return fail(FGS_ERR_INVALID_ARGUMENT, "n_primitives %d: more than 477 M Gaussians per backward pass are not supported", n_primitives);
    if (!prim_blob || !tile_blob || !scratch || (state->n_instances > 0 && !inst_blob) || (state->n_buckets > 0 && !bucket_blob))
        return fail(FGS_ERR_INVALID_ARGUMENT, "NULL scratch buffer");
    P.geo = geometry_of(settings->width, settings->height);
    Carver pc(prim_blob), tc(tile_blob), ic(inst_blob), bc(bucket_blob), sc(scratch);   // same carve order as the forward pass (bwd:46-52)
    P.pb = PrimitiveBuffers::carve(pc, static_cast<uint32_t>(n_primitives));
    P.tb = TileBuffers::carve(tc, P.geo.n_tiles, true);
    P.ib = InstanceBuffers::carve(ic, static_cast<uint32_t>(state->n_instances), P.geo.key_bytes, P.geo.end_bit);
    P.bb = BucketBuffers::carve(bc, static_cast<uint32_t>(state->n_buckets));
    P.sc = BackwardScratch::carve(sc, static_cast<uint32_t>(n_primitives), P.geo.n_tiles);
    return FGS_OK;
}

int run_blend_backward(const BackwardPlan& P, const float* grad_image, const float* image, int32_t n_primitives,
                       const fgs_settings* settings, const fgs_forward_state* state, hipStream_t stream, bool cleared_by_preprocess = true) {
    BlendBackwardArgs a{};
    // replaces api:127-134. K11 adds into 9-float records that must start at zero. The records of the visible Gaussians were cleared by K1 during the
    // forward pass (PrimitiveBuffers::acc); what is left for the staging kernel is the hot replicas (9 MB) -- or everything, when no K1 of this
    // library filled the blob (the sharded renderer: records arrive from the owners) or when these buffers already went through a backward pass
    // (a retained graph differentiated twice): counters[7], set by the last kernel of a backward pass, read on the device.
    static_assert(PrimitiveBuffers::kHotFloats % 4 == 0, "the cleared regions are whole numbers of 16-byte pieces");
    const size_t all_bytes = n_primitives > 0 ? static_cast<size_t>(reinterpret_cast<char*>(P.pb.acc_hot + PrimitiveBuffers::kHotFloats) - reinterpret_cast<char*>(P.pb.acc)) : 0;
    a.clear_all_f4 = static_cast<uint32_t>(all_bytes / 16);          // n <= 477 M (plan_backward): < 2^32 pieces
    a.clear_hot_f4 = n_primitives > 0 ? static_cast<uint32_t>(PrimitiveBuffers::kHotFloats / 4) : 0u;
    a.clear_everything = cleared_by_preprocess ? 0 : 1;
    a.dirty_flag = P.pb.counters + 7;
    a.ranges = P.tb.ranges; a.bucket_offsets = P.tb.bucket_offsets; a.inst_prims = P.ib.prims[state->selector]; a.rec = P.pb.rec;
    a.bg = settings->bg_color; a.grad_image = grad_image; a.image = image;
    a.final_T = P.tb.final_T; a.n_processed = P.tb.n_processed; a.max_n_processed = P.tb.max_n_processed;
    a.bucket_tile = P.bb.tile_index; a.ckpt = P.bb.ckpt; a.pixrec = P.sc.pixrec; a.acc = P.pb.acc;
    a.work_list = P.bb.work_list; a.live_count = P.tb.live_count; a.live_offsets = P.tb.live_offsets;
    a.acc_hot = P.pb.acc_hot; a.hot_list = P.pb.hot_list; a.hot_count = P.pb.counters + 4;
    a.n = static_cast<uint32_t>(n_primitives); a.width = settings->width; a.height = settings->height;
    a.grid_w = P.geo.grid_w; a.n_tiles = P.geo.n_tiles; a.n_buckets_cap = static_cast<uint32_t>(state->n_buckets);
    a.proper_aa = settings->proper_antialiasing ? 1 : 0;
    a.variant = blend_backward_variant();           // once per pass: the planning pass and the kernel see the same formulation
    { StageScope t(ST_STAGE_PIXELS, stream); FGS_HIP(launch_stage_pixels(a, stream)); }
    { StageScope t(ST_BLEND_BACKWARD, stream); FGS_HIP(launch_blend_backward(a, stream)); }     // K11 (bwd:56)
    return FGS_OK;
}

}  // namespace

extern "C" {
#pragma GCC visibility push(default)

int32_t fgs_abi_version(void) { return FGS_ABI_VERSION; }
const char* fgs_last_error(void) { return g_error; }
#ifdef FGS_DEV_SWITCHES
const char* fgs_build_info(void) { return "libfgs_hip_dev gfx950 wave64 tile16x12 bucket64 radix-sort-v1 +dev-switches"; }
#else
const char* fgs_build_info(void) { return "libfgs_hip gfx950 wave64 tile16x12 bucket64 radix-sort-v1"; }
#endif

int32_t fgs_forward(const float* means, const float* scales, const float* rotations, const float* opacities,
                    const float* sh_coefficients_0, const float* sh_coefficients_rest, int32_t n_primitives,
                    const fgs_settings* settings, float* image, fgs_resize_fn resize, void* resize_user,
                    fgs_forward_state* state_out, void* stream) {
    return run_forward(MODE_TRAINING, means, scales, rotations, opacities, sh_coefficients_0, sh_coefficients_rest, n_primitives, settings,
                       image, 1, 0, resize, resize_user, state_out, stream);
}

int32_t fgs_forward_async(const float* means, const float* scales, const float* rotations, const float* opacities,
                          const float* sh_coefficients_0, const float* sh_coefficients_rest, int32_t n_primitives,
                          const fgs_settings* settings, float* image, int32_t instance_capacity, fgs_resize_fn resize, void* resize_user,
                          fgs_forward_state* state_out, void* stream) {
    if (instance_capacity <= 0) return fail(FGS_ERR_INVALID_ARGUMENT, "instance_capacity %d", instance_capacity);
    return run_forward(MODE_TRAINING, means, scales, rotations, opacities, sh_coefficients_0, sh_coefficients_rest, n_primitives, settings,
                       image, 1, 0, resize, resize_user, state_out, stream, nullptr, instance_capacity);
}

int32_t fgs_forward_counts(const void* primitive_buffers, int32_t n_primitives, int32_t* host_out, void* stream_) {
    if (!primitive_buffers || n_primitives < 0 || !host_out) return fail(FGS_ERR_INVALID_ARGUMENT, "bad argument");
    Carver c(const_cast<void*>(primitive_buffers));
    const PrimitiveBuffers pb = PrimitiveBuffers::carve(c, static_cast<uint32_t>(n_primitives));
    hipStream_t stream = static_cast<hipStream_t>(stream_);
    // counters: [0] visible, [1] instances; [6] overflow flag of fgs_forward_async. Two small copies, no synchronisation here.
    FGS_HIP(hipMemcpyAsync(host_out, pb.counters, 2 * sizeof(uint32_t), hipMemcpyDeviceToHost, stream));
    FGS_HIP(hipMemcpyAsync(host_out + 2, pb.counters + 6, sizeof(uint32_t), hipMemcpyDeviceToHost, stream));
    return FGS_OK;
}

int32_t fgs_inference(const float* means, const float* scales, const float* rotations, const float* opacities,
                      const float* sh_coefficients_0, const float* sh_coefficients_rest, int32_t n_primitives,
                      const fgs_settings* settings, float* image, int32_t to_chw, int32_t clamp_output,
                      fgs_resize_fn resize, void* resize_user, fgs_forward_state* state_out, void* stream) {
    return run_forward(MODE_INFERENCE, means, scales, rotations, opacities, sh_coefficients_0, sh_coefficients_rest, n_primitives, settings,
                       image, to_chw, clamp_output, resize, resize_user, state_out, stream);
}

int32_t fgs_pruning_scores(float* scores, const float* means, const float* scales, const float* rotations, const float* opacities,
                           const float* sh_coefficients_0, const float* sh_coefficients_rest, int32_t n_primitives,
                           const fgs_settings* settings, fgs_resize_fn resize, void* resize_user, fgs_forward_state* state_out, void* stream) {
    return run_forward(MODE_SCORES, means, scales, rotations, opacities, sh_coefficients_0, sh_coefficients_rest, n_primitives, settings,
                       nullptr, 1, 0, resize, resize_user, state_out, stream, scores);
}

size_t fgs_backward_scratch_bytes(int32_t n_primitives, int32_t width, int32_t height) {
    if (n_primitives < 0 || width <= 0 || height <= 0) return 0;
    Carver c(nullptr);
    BackwardScratch::carve(c, static_cast<uint32_t>(n_primitives), geometry_of(width, height).n_tiles);
    return c.total();
}

int32_t fgs_backward_live(const float* grad_image, const float* image,
                          const float* means, const float* scales, const float* rotations, const float* opacities,
                          const float* sh_coefficients_rest,
                          void* primitive_buffers, void* tile_buffers, void* instance_buffers, void* bucket_buffers,
                          float* grad_means, float* grad_scales, float* grad_rotations, float* grad_opacities,
                          float* grad_sh_coefficients_0, float* grad_sh_coefficients_rest,
                          float* densification_info, void* scratch,
                          int32_t n_primitives, const fgs_settings* settings, const fgs_forward_state* state, uint8_t* live_blocks, void* stream_) {
    BackwardPlan P;
    if (int rc = plan_backward(P, primitive_buffers, tile_buffers, instance_buffers, bucket_buffers, scratch, n_primitives, settings, state)) return rc;
    if (!grad_image || !image) return fail(FGS_ERR_INVALID_ARGUMENT, "NULL image / grad_image");
    if (n_primitives == 0) return FGS_OK;
    if (!means || !scales || !rotations || !opacities || !grad_means || !grad_scales || !grad_rotations || !grad_opacities || !grad_sh_coefficients_0)
        return fail(FGS_ERR_INVALID_ARGUMENT, "NULL parameter / gradient tensor");
    hipStream_t stream = static_cast<hipStream_t>(stream_);
    if (int rc = run_blend_backward(P, grad_image, image, n_primitives, settings, state, stream)) return rc;

    PreprocessBackwardArgs a{};
    a.means = means; a.scales = scales; a.rotations = rotations; a.opacities = opacities; a.sh_rest = sh_coefficients_rest;
    a.n_views = 1;
    a.view[0] = backward_view(*settings, P.geo, P.pb.n_touched, nullptr, P.pb.acc, P.sc.view_dir);
    a.grad_means = grad_means; a.grad_scales = grad_scales; a.grad_rotations = grad_rotations; a.grad_opacities = grad_opacities;
    a.grad_sh0 = grad_sh_coefficients_0; a.densification_info = densification_info;
    a.n = static_cast<uint32_t>(n_primitives);
    if (settings->total_sh_bases_rest > 0 && !grad_sh_coefficients_rest) return fail(FGS_ERR_INVALID_ARGUMENT, "NULL grad_sh_coefficients_rest");
    ShRestArgs sh{};
    sh.n_views = 1; sh.view[0] = sh_rest_view(a.view[0]); sh.grad_sh_rest = grad_sh_coefficients_rest;
    sh.n = a.n; sh.total_sh_rest = settings->total_sh_bases_rest; sh.active_sh_bases = settings->active_sh_bases;
    if (g_fused_single_kernel) {           // K12 (bwd:94) as one kernel
        StageScope t(ST_PREPROCESS_BACKWARD, stream);
        a.live_blocks = live_blocks;
        FGS_HIP(launch_backward_gradients(a, sh, stream));
        return FGS_OK;
    }
    if (live_blocks != nullptr) FGS_HIP(hipMemsetAsync(live_blocks, 1, (static_cast<size_t>(n_primitives) + 63) / 64, stream));   // A/B form: no flags, every block "live"
    { StageScope t(ST_PREPROCESS_BACKWARD, stream); FGS_HIP(launch_preprocess_backward(false, a, stream)); }   // round-1 form: geometry kernel + SH-rest kernel
    if (settings->total_sh_bases_rest > 0) { StageScope t(ST_SH_REST_BACKWARD, stream); FGS_HIP(launch_sh_rest_backward(false, sh, stream)); }
    return FGS_OK;
}

int32_t fgs_backward(const float* grad_image, const float* image,
                     const float* means, const float* scales, const float* rotations, const float* opacities,
                     const float* sh_coefficients_rest,
                     void* primitive_buffers, void* tile_buffers, void* instance_buffers, void* bucket_buffers,
                     float* grad_means, float* grad_scales, float* grad_rotations, float* grad_opacities,
                     float* grad_sh_coefficients_0, float* grad_sh_coefficients_rest,
                     float* densification_info, void* scratch,
                     int32_t n_primitives, const fgs_settings* settings, const fgs_forward_state* state, void* stream) {
    return fgs_backward_live(grad_image, image, means, scales, rotations, opacities, sh_coefficients_rest, primitive_buffers, tile_buffers,
                             instance_buffers, bucket_buffers, grad_means, grad_scales, grad_rotations, grad_opacities, grad_sh_coefficients_0,
                             grad_sh_coefficients_rest, densification_info, scratch, n_primitives, settings, state, nullptr, stream);
}

int32_t fgs_backward_adam_fused(const float* grad_image, const float* image,
                                float* const* params, float* const* exp_avgs, float* const* exp_avg_sqs,
                                void* primitive_buffers, void* tile_buffers, void* instance_buffers, void* bucket_buffers,
                                float* densification_info, void* scratch,
                                int32_t n_primitives, const fgs_settings* settings, const fgs_forward_state* state,
                                int32_t step, const double* lrs, double beta1, double beta2, double eps, void* stream_) {
    BackwardPlan P;
    if (int rc = plan_backward(P, primitive_buffers, tile_buffers, instance_buffers, bucket_buffers, scratch, n_primitives, settings, state)) return rc;
    if (!grad_image || !image || !params || !exp_avgs || !exp_avg_sqs || !lrs || step < 1) return fail(FGS_ERR_INVALID_ARGUMENT, "bad argument");
    if (n_primitives == 0) return FGS_OK;
    for (int k = 0; k < 6; ++k)
        if (!params[k] || !exp_avgs[k] || !exp_avg_sqs[k]) {
            if (k == 2 && settings->total_sh_bases_rest == 0) continue;
            return fail(FGS_ERR_INVALID_ARGUMENT, "NULL tensor in group %d", k);
        }
    hipStream_t stream = static_cast<hipStream_t>(stream_);
    if (int rc = run_blend_backward(P, grad_image, image, n_primitives, settings, state, stream)) return rc;

    // API group order (Model.py:238-245): 0 means, 1 sh0, 2 sh_rest, 3 opacities, 4 scales, 5 rotations
    PreprocessBackwardArgs a{};
    a.means = params[0]; a.scales = params[4]; a.rotations = params[5]; a.opacities = params[3]; a.sh_rest = params[2];
    a.densification_info = densification_info;
    a.n_views = 1;
    a.view[0] = backward_view(*settings, P.geo, P.pb.n_touched, nullptr, P.pb.acc, P.sc.view_dir);
    a.n = static_cast<uint32_t>(n_primitives);
    const int map[5] = {0, 1, 3, 4, 5};     // kernel group order: means, sh0, opacities, scales, rotations
    for (int k = 0; k < 5; ++k) {
        a.p[k] = params[map[k]]; a.m[k] = exp_avgs[map[k]]; a.v[k] = exp_avg_sqs[map[k]];
        a.h[k] = adam_hyper(step, lrs[map[k]], beta1, beta2, eps);
    }
    ShRestArgs sh{};
    sh.n_views = 1; sh.view[0] = sh_rest_view(a.view[0]);
    sh.p = params[2]; sh.m = exp_avgs[2]; sh.v = exp_avg_sqs[2]; sh.h = adam_hyper(step, lrs[2], beta1, beta2, eps);
    sh.n = a.n; sh.total_sh_rest = settings->total_sh_bases_rest; sh.active_sh_bases = settings->active_sh_bases;
    if (g_fused_single_kernel) {
        // one kernel for all 59 floats: a wave gathers its Gaussians' sh_rest once, keeps the view direction in registers
        StageScope t(ST_FUSED_BACKWARD_ADAM, stream);
        FGS_HIP(launch_fused_backward_adam(a, sh, stream));
        return FGS_OK;
    }
    // Two-kernel form (round 1, kept for A/B): the geometry kernel reads sh_rest (pre-update) and leaves the view direction for
    // the SH-rest pass, which then updates sh_rest in place; means are updated by the geometry kernel after it has taken the direction.
    { StageScope t(ST_PREPROCESS_BACKWARD, stream); FGS_HIP(launch_preprocess_backward(true, a, stream)); }
    if (settings->total_sh_bases_rest > 0) { StageScope t(ST_SH_REST_BACKWARD, stream); FGS_HIP(launch_sh_rest_backward(true, sh, stream)); }
    return FGS_OK;
}

// ---- Gaussian-sharded multi-GPU path (shard_exchange.hip; no reference counterpart, the reference is single-GPU) ----

int32_t fgs_shard_preprocess(const float* means, const float* scales, const float* rotations, const float* opacities,
                             const float* sh_coefficients_0, const float* sh_coefficients_rest, int32_t n_primitives,
                             int32_t n_views, const fgs_settings* settings, void* records_out, uint32_t* counts_out,
                             fgs_resize_fn resize, void* resize_user, void* stream_) {
    if (n_views < 1 || !settings) return fail(FGS_ERR_INVALID_ARGUMENT, "n_views %d / settings", n_views);
    for (int v = 0; v < n_views; ++v) {
        if (int rc = check_settings(settings + v)) return rc;
        if (settings[v].width != settings[0].width || settings[v].height != settings[0].height || settings[v].total_sh_bases_rest != settings[0].total_sh_bases_rest)
            return fail(FGS_ERR_INVALID_ARGUMENT, "all views of a step must share the image size and SH layout");
    }
    if (n_primitives < 0 || !counts_out || !resize || (n_primitives > 0 && !records_out)) return fail(FGS_ERR_INVALID_ARGUMENT, "bad argument (n_primitives=%d)", n_primitives);
    if (n_primitives > 0 && (!means || !scales || !rotations || !opacities || !sh_coefficients_0 || (settings->total_sh_bases_rest > 0 && !sh_coefficients_rest)))
        return fail(FGS_ERR_INVALID_ARGUMENT, "NULL parameter tensor");
    hipStream_t stream = static_cast<hipStream_t>(stream_);
    const uint32_t n = static_cast<uint32_t>(n_primitives);
    const Geometry geo = geometry_of(settings->width, settings->height);
    Carver one(nullptr);
    PrimitiveBuffers::carve(one, n, false);
    const size_t per_view = one.total();
    char* prim_blob = static_cast<char*>(resize(resize_user, FGS_BUF_PRIMITIVE, per_view * n_views));
    if (!prim_blob && per_view > 0) return fail(FGS_ERR_ALLOC, "resize(primitive, %zu) returned NULL", per_view * n_views);
    for (int v0 = 0; v0 < n_views; v0 += kMaxBatchViews) {
        PreprocessBatch pb{};
        PackRecordsBatch rb{};
        pb.n_views = rb.n_views = n_views - v0 < kMaxBatchViews ? n_views - v0 : kMaxBatchViews;
        rb.capacity = n;
        for (int k = 0; k < pb.n_views; ++k) {
            const int v = v0 + k;
            Carver c(prim_blob + per_view * v);
            const PrimitiveBuffers b = PrimitiveBuffers::carve(c, n, false);
            FGS_HIP(hipMemsetAsync(b.counters, 0, kCounterWords * sizeof(uint32_t), stream));
            PreprocessArgs& pa = pb.v[k];
            pa.means = means; pa.scales = scales; pa.rotations = rotations; pa.opacities = opacities; pa.sh0 = sh_coefficients_0; pa.sh_rest = sh_coefficients_rest;
            pa.rec = b.rec; pa.n_touched = b.n_touched; pa.depth_keys = b.keys[0]; pa.prim_idx = b.prims[0]; pa.counters = b.counters; pa.huge_list = b.offsets; pa.hot_list = b.hot_list; pa.foot = nullptr;
            pa.count_appended = 1; pa.seq_tiles = g_seq_tiles;
            pa.n = n; pa.cam = camera_of(settings[v], geo); pa.ranges = nullptr; pa.n_tiles = 0;   // the tile ranges belong to the renderer of the view
            // slot table for fgs_shard_backward: the second depth-key buffer is free on this path (no sort on the owner)
            rb.v[k] = PackRecordsView{b.rec, b.n_touched, b.keys[0], b.prims[0], b.counters, b.keys[1],
                                      static_cast<uint32_t*>(records_out) + (size_t)v * n * kSplatRecordWords, counts_out + 2 * v};
        }
        if (n == 0) { FGS_HIP(hipMemsetAsync(counts_out + 2 * v0, 0, 2 * sizeof(uint32_t) * pb.n_views, stream)); continue; }
        { StageScope t(ST_PREPROCESS, stream); FGS_HIP(launch_preprocess_batch(pb, stream)); }
        { StageScope t(ST_RECORDS, stream); FGS_HIP(launch_pack_splat_records(rb, stream)); }
    }
    return FGS_OK;
}

// records of the shards, concatenated -> ShardOrder (nullptr / fewer than two segments / more than kMaxBatchViews: the order as received)
static int shard_order_of(ShardOrder& order, const int32_t* shard_counts, int32_t n_shards, int32_t n_records) {
    order = ShardOrder{};
    if (!shard_counts || n_shards <= 1) return FGS_OK;
    int64_t total = 0;
    for (int32_t k = 0; k < n_shards; ++k) {
        if (shard_counts[k] < 0) return fail(FGS_ERR_INVALID_ARGUMENT, "shard_counts[%d] = %d", k, shard_counts[k]);
        total += shard_counts[k];
    }
    if (total != n_records) return fail(FGS_ERR_INVALID_ARGUMENT, "shard_counts sum to %lld, n_records = %d", static_cast<long long>(total), n_records);
    if (n_shards > kMaxBatchViews) return FGS_OK;
    order.n_shards = n_shards;
    for (int32_t k = 0; k < n_shards; ++k) order.count[k] = static_cast<uint32_t>(shard_counts[k]);
    return FGS_OK;
}

int32_t fgs_forward_from_records(const void* records, int32_t n_records, int32_t n_instances, const fgs_settings* settings, float* image,
                                 fgs_resize_fn resize, void* resize_user, fgs_forward_state* state_out, void* stream_) {
    return fgs_forward_from_shard_records(records, n_records, n_instances, nullptr, 0, settings, image, resize, resize_user, state_out, stream_);
}

int32_t fgs_forward_from_shard_records(const void* records, int32_t n_records, int32_t n_instances, const int32_t* shard_counts, int32_t n_shards,
                                       const fgs_settings* settings, float* image, fgs_resize_fn resize, void* resize_user,
                                       fgs_forward_state* state_out, void* stream_) {
    if (int rc = check_settings(settings)) return rc;
    ShardOrder order;
    if (int rc = shard_order_of(order, shard_counts, n_shards, n_records)) return rc;
    if (n_records < 0 || n_instances < 0 || !image || !resize || !state_out || (n_records > 0 && !records))
        return fail(FGS_ERR_INVALID_ARGUMENT, "bad argument (n_records=%d, n_instances=%d)", n_records, n_instances);
    hipStream_t stream = static_cast<hipStream_t>(stream_);
    const uint32_t n = static_cast<uint32_t>(n_records);
    const Geometry geo = geometry_of(settings->width, settings->height);
    Carver tile_size(nullptr);
    TileBuffers::carve(tile_size, geo.n_tiles, true);
    void* tile_blob = resize(resize_user, FGS_BUF_TILE, tile_size.total());
    if (!tile_blob && tile_size.total() > 0) return fail(FGS_ERR_ALLOC, "resize(tile, %zu) returned NULL", tile_size.total());
    Carver tile_c(tile_blob);
    TileBuffers tb = TileBuffers::carve(tile_c, geo.n_tiles, true);
    Carver prim_size(nullptr);
    PrimitiveBuffers::carve(prim_size, n);
    void* prim_blob = resize(resize_user, FGS_BUF_PRIMITIVE, prim_size.total());
    if (!prim_blob && prim_size.total() > 0) return fail(FGS_ERR_ALLOC, "resize(primitive, %zu) returned NULL", prim_size.total());
    Carver prim_c(prim_blob);
    PrimitiveBuffers pb = PrimitiveBuffers::carve(prim_c, n);
    FGS_HIP(hipMemsetAsync(pb.counters, 0, kCounterWords * sizeof(uint32_t), stream));
    { StageScope t(ST_RECORDS, stream);
      FGS_HIP(launch_unpack_splat_records(static_cast<const uint32_t*>(records), n, pb.rec, pb.n_touched, pb.keys[0], pb.prims[0], pb.foot[0], tb.ranges, geo.n_tiles, pb.hot_list, pb.counters + 4, order, stream)); }
    return forward_tail(MODE_TRAINING, pb, tb, geo, n, static_cast<uint32_t>(n_instances), -1, settings, image, 1, 0, resize, resize_user, state_out, stream, nullptr);
}

int32_t fgs_backward_to_records(const float* grad_image, const float* image,
                                void* primitive_buffers, void* tile_buffers, void* instance_buffers, void* bucket_buffers,
                                void* scratch, float* acc_records_out, int32_t n_records,
                                const fgs_settings* settings, const fgs_forward_state* state, void* stream_) {
    return fgs_backward_to_shard_records(grad_image, image, primitive_buffers, tile_buffers, instance_buffers, bucket_buffers, scratch, acc_records_out,
                                         n_records, nullptr, 0, settings, state, stream_);
}

int32_t fgs_backward_to_shard_records(const float* grad_image, const float* image,
                                      void* primitive_buffers, void* tile_buffers, void* instance_buffers, void* bucket_buffers,
                                      void* scratch, float* acc_records_out, int32_t n_records, const int32_t* shard_counts, int32_t n_shards,
                                      const fgs_settings* settings, const fgs_forward_state* state, void* stream_) {
    ShardOrder order;
    if (int rc = shard_order_of(order, shard_counts, n_shards, n_records)) return rc;
    BackwardPlan P;
    if (int rc = plan_backward(P, primitive_buffers, tile_buffers, instance_buffers, bucket_buffers, scratch, n_records, settings, state)) return rc;
    if (!grad_image || !image) return fail(FGS_ERR_INVALID_ARGUMENT, "NULL image / grad_image");
    if (n_records == 0) return FGS_OK;
    if (!acc_records_out) return fail(FGS_ERR_INVALID_ARGUMENT, "NULL acc_records_out");
    hipStream_t stream = static_cast<hipStream_t>(stream_);
    if (int rc = run_blend_backward(P, grad_image, image, n_records, settings, state, stream, false)) return rc;      // no K1 of this library wrote this blob
    { StageScope t(ST_RECORDS, stream); FGS_HIP(launch_pack_acc(P.pb.acc, static_cast<uint32_t>(n_records), acc_records_out, order, stream)); }
    return FGS_OK;
}

size_t fgs_shard_backward_scratch_bytes(int32_t n_primitives, int32_t n_views) {
    if (n_primitives < 0 || n_views < 1) return 0;
    return ((size_t)n_primitives * 3 * sizeof(float) + 255) / 256 * 256 * (size_t)n_views + 256;     // one view-direction array per view
}

struct ShardAdam { float* const* params; float* const* exp_avgs; float* const* exp_avg_sqs; int step; const double* lrs; double beta1, beta2, eps; };

static int run_shard_backward(const float* acc_records, const int32_t* n_visible, const void* primitive_buffers,
                              const float* means, const float* scales, const float* rotations, const float* opacities,
                              const float* sh_coefficients_rest,
                              float* grad_means, float* grad_scales, float* grad_rotations, float* grad_opacities,
                              float* grad_sh_coefficients_0, float* grad_sh_coefficients_rest,
                              float* densification_info, void* scratch, int32_t n_primitives, int32_t n_views,
                              const fgs_settings* settings, const ShardAdam* adam, void* stream_) {
    if (n_views < 1 || !settings || !n_visible) return fail(FGS_ERR_INVALID_ARGUMENT, "n_views %d / settings / n_visible", n_views);
    int64_t total_visible = 0;
    for (int v = 0; v < n_views; ++v) {
        if (int rc = check_settings(settings + v)) return rc;
        if (n_visible[v] < 0 || n_visible[v] > n_primitives) return fail(FGS_ERR_INVALID_ARGUMENT, "view %d: n_visible %d of %d primitives", v, n_visible[v], n_primitives);
        total_visible += n_visible[v];
    }
    if (n_primitives < 0) return fail(FGS_ERR_INVALID_ARGUMENT, "n_primitives %d", n_primitives);
    if (n_primitives == 0) return FGS_OK;
    if (!primitive_buffers || !scratch || (total_visible > 0 && !acc_records)) return fail(FGS_ERR_INVALID_ARGUMENT, "NULL buffer");
    if (!means || !scales || !rotations || !opacities) return fail(FGS_ERR_INVALID_ARGUMENT, "NULL parameter tensor");
    if (!adam && (!grad_means || !grad_scales || !grad_rotations || !grad_opacities || !grad_sh_coefficients_0 ||
                  (settings->total_sh_bases_rest > 0 && !grad_sh_coefficients_rest)))
        return fail(FGS_ERR_INVALID_ARGUMENT, "NULL gradient tensor");
    if (adam && n_views > kMaxBatchViews) return fail(FGS_ERR_INVALID_ARGUMENT, "the fused form sums at most %d views in registers (got %d)", kMaxBatchViews, n_views);
    hipStream_t stream = static_cast<hipStream_t>(stream_);
    const uint32_t n = static_cast<uint32_t>(n_primitives);
    const Geometry geo = geometry_of(settings->width, settings->height);
    Carver one(nullptr);
    PrimitiveBuffers::carve(one, n, false);
    const size_t per_view = one.total();
    const size_t dir_stride = ((size_t)n * 3 * sizeof(float) + 255) / 256 * 256;
    char* const dir_base = reinterpret_cast<char*>((reinterpret_cast<uintptr_t>(scratch) + 255) & ~static_cast<uintptr_t>(255));
    size_t first_record = 0;
    for (int v0 = 0; v0 < n_views; v0 += kMaxBatchViews) {
        PreprocessBackwardArgs a{};
        ShRestArgs sh{};
        a.means = means; a.scales = scales; a.rotations = rotations; a.opacities = opacities; a.sh_rest = sh_coefficients_rest;
        a.grad_means = grad_means; a.grad_scales = grad_scales; a.grad_rotations = grad_rotations; a.grad_opacities = grad_opacities;
        a.grad_sh0 = grad_sh_coefficients_0; a.densification_info = densification_info;
        a.n = n; a.accumulate = v0 > 0 ? 1 : 0;           // gradients of a batch of views are summed in registers; later batches add
        a.n_views = sh.n_views = n_views - v0 < kMaxBatchViews ? n_views - v0 : kMaxBatchViews;
        for (int k = 0; k < a.n_views; ++k) {
            const int v = v0 + k;
            Carver c(const_cast<char*>(static_cast<const char*>(primitive_buffers)) + per_view * v);
            const PrimitiveBuffers b = PrimitiveBuffers::carve(c, n, false);
            // accumulator records are read in place through the slot table K1 left behind: no scatter pass, no dense copy
            a.view[k] = backward_view(settings[v], geo, b.n_touched, b.keys[1], acc_records + first_record * kAccRecordWords,
                                      reinterpret_cast<float*>(dir_base + dir_stride * v));
            sh.view[k] = sh_rest_view(a.view[k]);
            first_record += static_cast<size_t>(n_visible[v]);
        }
        if (adam) {       // API group order (Model.py:238-245): 0 means, 1 sh0, 2 sh_rest, 3 opacities, 4 scales, 5 rotations
            const int map[5] = {0, 1, 3, 4, 5};     // kernel group order: means, sh0, opacities, scales, rotations
            for (int k = 0; k < 5; ++k) {
                a.p[k] = adam->params[map[k]]; a.m[k] = adam->exp_avgs[map[k]]; a.v[k] = adam->exp_avg_sqs[map[k]];
                a.h[k] = adam_hyper(adam->step, adam->lrs[map[k]], adam->beta1, adam->beta2, adam->eps);
            }
            sh.p = adam->params[2]; sh.m = adam->exp_avgs[2]; sh.v = adam->exp_avg_sqs[2];
            sh.h = adam_hyper(adam->step, adam->lrs[2], adam->beta1, adam->beta2, adam->eps);
        }
        // fused: the geometry kernel reads sh_rest (pre-update) and leaves the view directions, then the SH-rest pass updates it
        { StageScope t(ST_PREPROCESS_BACKWARD, stream); FGS_HIP(launch_preprocess_backward(adam != nullptr, a, stream)); }
        if (settings->total_sh_bases_rest > 0) {
            sh.grad_sh_rest = grad_sh_coefficients_rest;
            sh.n = n; sh.total_sh_rest = settings->total_sh_bases_rest; sh.active_sh_bases = settings->active_sh_bases; sh.accumulate = a.accumulate;
            { StageScope t(ST_SH_REST_BACKWARD, stream); FGS_HIP(launch_sh_rest_backward(adam != nullptr, sh, stream)); }
        }
    }
    return FGS_OK;
}

int32_t fgs_shard_backward(const float* acc_records, const int32_t* n_visible, const void* primitive_buffers,
                           const float* means, const float* scales, const float* rotations, const float* opacities,
                           const float* sh_coefficients_rest,
                           float* grad_means, float* grad_scales, float* grad_rotations, float* grad_opacities,
                           float* grad_sh_coefficients_0, float* grad_sh_coefficients_rest,
                           float* densification_info, void* scratch, int32_t n_primitives, int32_t n_views,
                           const fgs_settings* settings, void* stream) {
    return run_shard_backward(acc_records, n_visible, primitive_buffers, means, scales, rotations, opacities, sh_coefficients_rest, grad_means,
                              grad_scales, grad_rotations, grad_opacities, grad_sh_coefficients_0, grad_sh_coefficients_rest, densification_info,
                              scratch, n_primitives, n_views, settings, nullptr, stream);
}

int32_t fgs_shard_backward_adam_fused(const float* acc_records, const int32_t* n_visible, const void* primitive_buffers,
                                      float* const* params, float* const* exp_avgs, float* const* exp_avg_sqs,
                                      float* densification_info, void* scratch, int32_t n_primitives, int32_t n_views,
                                      const fgs_settings* settings, int32_t step, const double* lrs, double beta1, double beta2, double eps,
                                      void* stream) {
    if (!params || !exp_avgs || !exp_avg_sqs || !lrs || step < 1 || !settings) return fail(FGS_ERR_INVALID_ARGUMENT, "bad argument");
    for (int k = 0; k < 6; ++k)
        if (n_primitives > 0 && (!params[k] || !exp_avgs[k] || !exp_avg_sqs[k])) {
            if (k == 2 && settings->total_sh_bases_rest == 0) continue;
            return fail(FGS_ERR_INVALID_ARGUMENT, "NULL tensor in group %d", k);
        }
    const ShardAdam adam{params, exp_avgs, exp_avg_sqs, step, lrs, beta1, beta2, eps};
    return run_shard_backward(acc_records, n_visible, primitive_buffers, params[0], params[4], params[5], params[3], params[2], nullptr, nullptr,
                              nullptr, nullptr, nullptr, nullptr, densification_info, scratch, n_primitives, n_views, settings, &adam, stream);
}

int32_t fgs_adam_step_multi_live(int32_t n_groups, const float* const* grads, float* const* params, float* const* exp_avgs,
                                 float* const* exp_avg_sqs, const int64_t* n_elements, const int32_t* steps, const double* lrs,
                                 double beta1, double beta2, double eps, const uint8_t* live_blocks, const int32_t* floats_per_gaussian,
                                 void* stream) {
    if (n_groups < 0 || n_groups > 8) return fail(FGS_ERR_INVALID_ARGUMENT, "n_groups %d (max 8)", n_groups);
    if (live_blocks != nullptr && floats_per_gaussian == nullptr) return fail(FGS_ERR_INVALID_ARGUMENT, "live_blocks without floats_per_gaussian");
    AdamArgs a{};
    a.live_blocks = live_blocks;
    uint32_t blocks = 0;
    for (int k = 0; k < n_groups; ++k) {
        if (n_elements[k] < 0 || steps[k] < 1) return fail(FGS_ERR_INVALID_ARGUMENT, "group %d: n_elements / step", k);
        if (n_elements[k] == 0) continue;
        if (!grads[k] || !params[k] || !exp_avgs[k] || !exp_avg_sqs[k]) return fail(FGS_ERR_INVALID_ARGUMENT, "group %d: NULL tensor", k);
        AdamGroup& g = a.g[a.n_groups++];
        g.grad = grads[k]; g.param = params[k]; g.exp_avg = exp_avgs[k]; g.exp_avg_sq = exp_avg_sqs[k]; g.n = n_elements[k];
        g.h = adam_hyper(steps[k], lrs[k], beta1, beta2, eps);
        g.row_len = 0;
        if (live_blocks != nullptr) {
            if (floats_per_gaussian[k] < 1 || n_elements[k] % floats_per_gaussian[k] != 0) return fail(FGS_ERR_INVALID_ARGUMENT, "group %d: floats_per_gaussian", k);
            if (n_elements[k] < (int64_t{1} << 32)) g.row_len = static_cast<uint32_t>(floats_per_gaussian[k]);   // 32-bit index arithmetic in the kernel
        }
        g.first_block = blocks;
        blocks += static_cast<uint32_t>((n_elements[k] + 1023) / 1024);
    }
    a.total_blocks = blocks;
    { StageScope t(ST_ADAM, static_cast<hipStream_t>(stream)); FGS_HIP(launch_adam(a, static_cast<hipStream_t>(stream))); }
    return FGS_OK;
}

int32_t fgs_adam_step_multi(int32_t n_groups, const float* const* grads, float* const* params, float* const* exp_avgs,
                            float* const* exp_avg_sqs, const int64_t* n_elements, const int32_t* steps, const double* lrs,
                            double beta1, double beta2, double eps, void* stream) {
    return fgs_adam_step_multi_live(n_groups, grads, params, exp_avgs, exp_avg_sqs, n_elements, steps, lrs, beta1, beta2, eps, nullptr, nullptr, stream);
}

int32_t fgs_adam_step(const float* grad, float* param, float* exp_avg, float* exp_avg_sq, int64_t n_elements,
                      int32_t step, double lr, double beta1, double beta2, double eps, void* stream) {
    return fgs_adam_step_multi(1, &grad, &param, &exp_avg, &exp_avg_sq, &n_elements, &step, &lr, beta1, beta2, eps, stream);
}

int32_t fgs_blob_layout(int32_t which, int32_t n_primitives, int32_t width, int32_t height, int32_t n_instances,
                        int32_t n_buckets, fgs_blob_entry* entries, int32_t max_entries) {
    if (width <= 0 || height <= 0 || n_primitives < 0 || n_instances < 0 || n_buckets < 0) return fail(FGS_ERR_INVALID_ARGUMENT, "bad sizes");
    const Geometry geo = geometry_of(width, height);
    Carver c(nullptr, entries, max_entries);
    switch (which) {
        case FGS_BUF_PRIMITIVE: PrimitiveBuffers::carve(c, n_primitives); break;
        case FGS_BUF_TILE: TileBuffers::carve(c, geo.n_tiles, true); break;
        case FGS_BUF_INSTANCE: InstanceBuffers::carve(c, n_instances, geo.key_bytes, geo.end_bit); break;
        case FGS_BUF_BUCKET: BucketBuffers::carve(c, n_buckets); break;
        case FGS_BUF_COUNT: BackwardScratch::carve(c, n_primitives, geo.n_tiles); break;   // the backward scratch buffer
        default: return fail(FGS_ERR_INVALID_ARGUMENT, "unknown buffer %d", which);
    }
    return c.n < max_entries ? c.n : max_entries;
}

int32_t fgs_update_3d_filter(const float* positions, const float* w2c, float* filter_3d, uint8_t* visibility_mask, int32_t n_points,
                             int32_t width, int32_t height, float focal_x, float focal_y, float center_x, float center_y,
                             float near_plane, float clipping_tolerance, float distance2filter, void* stream) {
    if (n_points < 0 || (n_points > 0 && (!positions || !w2c || !filter_3d || !visibility_mask))) return fail(FGS_ERR_INVALID_ARGUMENT, "bad argument");
    // host-side frustum bounds exactly as filter3d.cu:55-66
    const float bounds_factor = clipping_tolerance + 0.5f;
    const float width_f = static_cast<float>(width), height_f = static_cast<float>(height);
    const float max_x = bounds_factor * width_f, max_y = bounds_factor * height_f;
    const float off_x = center_x - 0.5f * width_f, off_y = center_y - 0.5f * height_f;
    const float left = (-max_x - off_x) / focal_x, right = (max_x - off_x) / focal_x;
    const float top = (-max_y - off_y) / focal_y, bottom = (max_y - off_y) / focal_y;
    FGS_HIP(launch_update_3d_filter(positions, w2c, filter_3d, visibility_mask, n_points, left, right, top, bottom, near_plane,
                                    distance2filter, static_cast<hipStream_t>(stream)));
    return FGS_OK;
}

int32_t fgs_relocation_table(float* table_host_2500) {
    if (!table_host_2500) return fail(FGS_ERR_INVALID_ARGUMENT, "NULL table");
    relocation_coefficients(table_host_2500);
    return FGS_OK;
}

int32_t fgs_relocation_adjustment(const float* old_opacities, const float* old_scales, const int64_t* n_samples_per_primitive,
                                  const float* table_device, float* new_opacities, float* new_scales, int32_t n_primitives, void* stream) {
    if (n_primitives < 0 || (n_primitives > 0 && (!old_opacities || !old_scales || !n_samples_per_primitive || !table_device || !new_opacities || !new_scales)))
        return fail(FGS_ERR_INVALID_ARGUMENT, "bad argument");
    FGS_HIP(launch_relocation(old_opacities, old_scales, n_samples_per_primitive, table_device, new_opacities, new_scales,
                              static_cast<unsigned>(n_primitives), static_cast<hipStream_t>(stream)));
    return FGS_OK;
}

int32_t fgs_add_noise(const float* raw_scales, const float* raw_rotations, const float* raw_opacities, const float* random_samples,
                      float* means, int32_t n_primitives, float current_lr, void* stream) {
    if (n_primitives < 0 || (n_primitives > 0 && (!raw_scales || !raw_rotations || !raw_opacities || !random_samples || !means)))
        return fail(FGS_ERR_INVALID_ARGUMENT, "bad argument");
    FGS_HIP(launch_add_noise(raw_scales, raw_rotations, raw_opacities, random_samples, means, static_cast<unsigned>(n_primitives), current_lr,
                             static_cast<hipStream_t>(stream)));
    return FGS_OK;
}

// ---- maintenance of the Gaussian set on the device (densify.hip; Model.py:275-366, 459-463) ----
namespace {
struct AdcScratch {
    uint32_t* plan; uint4* offsets; uint32_t* totals; char* scan_temp; size_t scan_temp_bytes;
    static AdcScratch carve(Carver& c, uint32_t n) {
        AdcScratch b;
        b.plan = c.take<uint32_t>("plan", n);
        b.offsets = c.take<uint4>("offsets", n);
        b.totals = c.take<uint32_t>("totals", 4);
        b.scan_temp_bytes = adc_scan_temp_bytes(n);
        b.scan_temp = c.take<char>("scan_temp", b.scan_temp_bytes);
        return b;
    }
};
}  // namespace

size_t fgs_adc_scratch_bytes(int32_t n_primitives) {
    if (n_primitives < 0) return 0;
    Carver c(nullptr);
    AdcScratch::carve(c, static_cast<uint32_t>(n_primitives));
    return c.total();
}

int32_t fgs_adc_plan(const float* densification_info, const float* scales, const float* rotations, const float* opacities, int32_t n_primitives,
                     float grad_threshold, float min_opacity, int32_t prune_large_gaussians, float percent_dense, float extent,
                     void* scratch, int32_t* counts_out, void* stream_) {
    if (n_primitives < 0 || !counts_out || !scratch) return fail(FGS_ERR_INVALID_ARGUMENT, "bad argument");
    if (n_primitives > 0 && (!densification_info || !scales || !rotations || !opacities)) return fail(FGS_ERR_INVALID_ARGUMENT, "NULL tensor");
    if (!(min_opacity > 0.0f && min_opacity < 1.0f) || !(percent_dense * extent > 0.0f)) return fail(FGS_ERR_INVALID_ARGUMENT, "min_opacity / percent_dense * extent out of range");
    hipStream_t stream = static_cast<hipStream_t>(stream_);
    Carver c(scratch);
    const AdcScratch sc = AdcScratch::carve(c, static_cast<uint32_t>(n_primitives));
    AdcPlanArgs a{};
    a.densification_info = densification_info; a.scales = scales; a.rotations = rotations; a.opacities = opacities;
    a.n = static_cast<uint32_t>(n_primitives);
    a.grad_threshold = grad_threshold;
    a.min_opacity_logit = static_cast<float>(std::log(static_cast<double>(min_opacity) / (1.0 - static_cast<double>(min_opacity))));   // Model.py:360
    a.log_small = static_cast<float>(std::log(static_cast<double>(percent_dense) * static_cast<double>(extent)));                         // :315
    a.log_large = static_cast<float>(std::log(0.1 * static_cast<double>(extent)));                                                         // :363
    a.prune_large = prune_large_gaussians ? 1 : 0;
    a.plan = sc.plan; a.offsets = sc.offsets; a.totals = sc.totals; a.scan_temp = sc.scan_temp; a.scan_temp_bytes = sc.scan_temp_bytes;
    FGS_HIP(launch_adc_plan(a, stream));
    uint32_t host[4] = {0, 0, 0, 0};
    FGS_HIP(hipMemcpyAsync(host, sc.totals, sizeof(host), hipMemcpyDeviceToHost, stream));     // the caller sizes the new tensors from these
    FGS_HIP(hipStreamSynchronize(stream));
    for (int k = 0; k < 4; ++k) counts_out[k] = static_cast<int32_t>(host[k]);
    return FGS_OK;
}

int32_t fgs_adc_apply(const float* const* params, const float* const* exp_avgs, const float* const* exp_avg_sqs,
                      float* const* out_params, float* const* out_exp_avgs, float* const* out_exp_avg_sqs,
                      const float* noise, const void* scratch, int32_t n_primitives, int32_t total_sh_bases_rest, void* stream_) {
    if (n_primitives < 0 || !params || !out_params || !scratch || total_sh_bases_rest < 0) return fail(FGS_ERR_INVALID_ARGUMENT, "bad argument");
    if ((exp_avgs == nullptr) != (exp_avg_sqs == nullptr) || (exp_avgs && (!out_exp_avgs || !out_exp_avg_sqs))) return fail(FGS_ERR_INVALID_ARGUMENT, "moments: all four arrays or none");
    if (n_primitives == 0) return FGS_OK;
    hipStream_t stream = static_cast<hipStream_t>(stream_);
    Carver c(const_cast<void*>(scratch));
    const AdcScratch sc = AdcScratch::carve(c, static_cast<uint32_t>(n_primitives));
    // optimizer-group order (Model.py:238-245): means, sh0, sh_rest, opacities, scales, rotations
    const uint32_t width[6] = {3u, 3u, 3u * static_cast<uint32_t>(total_sh_bases_rest), 1u, 3u, 4u};
    const int kind[6] = {1, 0, 0, 0, 2, 0};
    for (int k = 0; k < 6; ++k) {
        if (width[k] == 0) continue;
        if (!params[k] || !out_params[k]) return fail(FGS_ERR_INVALID_ARGUMENT, "NULL tensor in group %d", k);
        AdcScatterArgs a{};
        a.in_p = params[k]; a.out_p = out_params[k];
        if (exp_avgs && exp_avgs[k]) {
            if (!exp_avg_sqs[k] || !out_exp_avgs[k] || !out_exp_avg_sqs[k]) return fail(FGS_ERR_INVALID_ARGUMENT, "NULL moment tensor in group %d", k);
            a.in_m = exp_avgs[k]; a.in_v = exp_avg_sqs[k]; a.out_m = out_exp_avgs[k]; a.out_v = out_exp_avg_sqs[k];
        }
        a.scales = params[4]; a.rotations = params[5]; a.noise = noise;
        a.plan = sc.plan; a.offsets = sc.offsets; a.totals = sc.totals;
        a.n = static_cast<uint32_t>(n_primitives); a.width = width[k];
        FGS_HIP(launch_adc_scatter(kind[k], a, stream));
    }
    return FGS_OK;
}

int32_t fgs_gather_rows(int32_t n_tensors, const float* const* in, float* const* out, const int32_t* widths, const int64_t* index,
                        int32_t n_rows, void* stream) {
    if (n_tensors < 0 || n_tensors > kGatherTensors || n_rows < 0) return fail(FGS_ERR_INVALID_ARGUMENT, "n_tensors %d (max %d) / n_rows %d", n_tensors, kGatherTensors, n_rows);
    if (n_rows == 0 || n_tensors == 0) return FGS_OK;
    if (!in || !out || !widths || !index) return fail(FGS_ERR_INVALID_ARGUMENT, "NULL argument");
    GatherArgs a{};
    for (int k = 0; k < n_tensors; ++k) {
        if (widths[k] < 0) return fail(FGS_ERR_INVALID_ARGUMENT, "tensor %d: width %d", k, widths[k]);
        if (widths[k] == 0) continue;
        if (!in[k] || !out[k]) return fail(FGS_ERR_INVALID_ARGUMENT, "tensor %d: NULL", k);
        GatherTensor& t = a.t[a.n_tensors++];
        t.in = in[k]; t.out = out[k]; t.width = static_cast<uint32_t>(widths[k]);
    }
    a.n_rows = static_cast<uint32_t>(n_rows); a.index = index;
    FGS_HIP(launch_gather_rows(a, static_cast<hipStream_t>(stream)));
    return FGS_OK;
}

size_t fgs_morton_order_temp_bytes(int32_t n_points) { return n_points < 0 ? 0 : morton_temp_bytes(static_cast<uint32_t>(n_points)); }

int32_t fgs_morton_order(const float* means, const float* lo, const float* hi, int64_t* order_out, int32_t n_points, void* temp, size_t temp_bytes,
                         void* stream) {
    if (n_points < 0) return fail(FGS_ERR_INVALID_ARGUMENT, "n_points %d", n_points);
    if (n_points == 0) return FGS_OK;
    if (!means || !lo || !hi || !order_out || !temp || temp_bytes < morton_temp_bytes(static_cast<uint32_t>(n_points))) return fail(FGS_ERR_INVALID_ARGUMENT, "NULL / short buffer");
    FGS_HIP(run_morton_order(means, lo, hi, order_out, static_cast<uint32_t>(n_points), temp, temp_bytes, static_cast<hipStream_t>(stream)));
    return FGS_OK;
}

size_t fgs_l1_dssim_scratch_bytes(int32_t width, int32_t height) {
    if (width <= 0 || height <= 0) return 0;
    // three derivative maps, then the per-workgroup partial sums on an 8-byte boundary (read as float2: 9 W H is odd for odd x odd images)
    return sizeof(float) * (((9 * static_cast<size_t>(width) * static_cast<size_t>(height) + 1) & ~static_cast<size_t>(1)) + l1_dssim_partials(width, height));
}

int32_t fgs_l1_dssim_loss(const float* image, const float* target, int32_t width, int32_t height, float lambda_l1, float lambda_dssim,
                          float* sums, float* grad_image, void* scratch, void* stream_) {
    if (!image || !target || !sums || !scratch || width <= 0 || height <= 0) return fail(FGS_ERR_INVALID_ARGUMENT, "bad argument");
    hipStream_t stream = static_cast<hipStream_t>(stream_);
    const size_t plane3 = 3 * static_cast<size_t>(width) * static_cast<size_t>(height);
    LossArgs a{};
    a.image = image; a.target = target; a.sums = sums; a.grad = grad_image;
    a.d_mu = static_cast<float*>(scratch); a.d_m11 = a.d_mu + plane3; a.d_m12 = a.d_m11 + plane3;
    a.partials = a.d_mu + ((3 * plane3 + 1) & ~static_cast<size_t>(1));
    if ((reinterpret_cast<uintptr_t>(a.partials) & 7u) != 0) return fail(FGS_ERR_INVALID_ARGUMENT, "scratch must be 8-byte aligned");
    a.width = width; a.height = height; a.lambda_l1 = lambda_l1; a.lambda_dssim = lambda_dssim;
    { StageScope t(ST_LOSS, stream); FGS_HIP(launch_l1_dssim(a, stream)); }
    return FGS_OK;
}

int32_t fgs_l1_dssim_backward(const float* image, const float* target, int32_t width, int32_t height, float lambda_l1, float lambda_dssim,
                              const float* upstream, float* grad_image, const void* scratch, void* stream_) {
    if (!image || !target || !grad_image || !scratch || width <= 0 || height <= 0) return fail(FGS_ERR_INVALID_ARGUMENT, "bad argument");
    hipStream_t stream = static_cast<hipStream_t>(stream_);
    const size_t plane3 = 3 * static_cast<size_t>(width) * static_cast<size_t>(height);
    LossArgs a{};
    a.image = image; a.target = target; a.grad = grad_image; a.upstream = upstream;
    a.d_mu = static_cast<float*>(const_cast<void*>(scratch)); a.d_m11 = a.d_mu + plane3; a.d_m12 = a.d_m11 + plane3;
    a.width = width; a.height = height; a.lambda_l1 = lambda_l1; a.lambda_dssim = lambda_dssim;
    { StageScope t(ST_LOSS, stream); FGS_HIP(launch_l1_dssim_backward(a, stream)); }
    return FGS_OK;
}

int32_t fgs_profile_enable(int32_t enable) {
    g_prof.enabled = enable != 0;
    g_prof.only = enable >= 2 ? enable - 2 : -1;          // 1: every stage; 2 + k: stage k only (index into fgs_profile_read's table)
    if (g_prof.only >= ST_COUNT) return fail(FGS_ERR_INVALID_ARGUMENT, "stage index %d", g_prof.only);
    return FGS_OK;
}

int32_t fgs_profile_read(fgs_stage_time* out, int32_t max_entries) {
    if (!out || max_entries < 0) return fail(FGS_ERR_INVALID_ARGUMENT, "bad output array");
    double ms[ST_COUNT] = {0}; int64_t calls[ST_COUNT] = {0};
    for (StageRecord& r : g_prof.records) {
        float t = 0.0f;
        if (hipEventSynchronize(r.stop) == hipSuccess && hipEventElapsedTime(&t, r.start, r.stop) == hipSuccess) { ms[r.stage] += t; ++calls[r.stage]; }
        g_prof.pool.push_back(r.start); g_prof.pool.push_back(r.stop);
    }
    g_prof.records.clear();
    int n = 0;
    for (int k = 0; k < ST_COUNT && n < max_entries; ++k) { out[n].name = kStageNames[k]; out[n].total_ms = ms[k]; out[n].calls = calls[k]; ++n; }
    return n;
}

#ifdef FGS_DEV_SWITCHES      // the A/B switchboard exists in libfgs_hip_dev.so only (tools/, the variant tests); the product library has no process-wide knobs
int32_t fgs_debug_set_backward_variant(int32_t variant) {
    if (variant < 0 || variant > 5) return fail(FGS_ERR_INVALID_ARGUMENT, "variant must be 0 (systolic), 1 (strip), 2 (systolic, global dL/dC), 3 (live list + compacted pixels), 4 (lane = pixel, matrix-core reduction) or 5 (3 with the items of a wave chained through the lanes)");
    fgs::g_backward_variant = variant;
    return FGS_OK;
}

int32_t fgs_debug_set_option(int32_t key, int32_t value) {
    switch (key) {
        case 0: return fgs_debug_set_backward_variant(value);
        case 1: if (value != 1 && value != 2 && value != 4) return fail(FGS_ERR_INVALID_ARGUMENT, "adam unroll must be 1, 2 or 4");
                fgs::g_adam_unroll = value; return FGS_OK;
        case 2: fgs::g_adam_nontemporal = value ? 1 : 0; return FGS_OK;
        case 3: g_fused_single_kernel = value ? 1 : 0; return FGS_OK;
        case 7: fgs::g_backward_ablate = value & 15; return FGS_OK;
        case 13: if (value < 1) return fail(FGS_ERR_INVALID_ARGUMENT, "K11 variant 4 needs at least one workgroup"); fgs::g_k11m_max_blocks = value; return FGS_OK;
        case 8: fgs::g_adam_reverse = value ? 1 : 0; return FGS_OK;
        case 9: fgs::g_depth_sort_mode = value & 3; return FGS_OK;
        case 10: if (value < 0 || (value > 64 && (value < 251 || value > 255))) return fail(FGS_ERR_INVALID_ARGUMENT, "tile mapping must be 252 (one strip of tile columns per XCD, default), 254 (device-side block plan), 0 (bands), 255 (bands, bottom first) or 1..64 (row groups)");
                 fgs::g_tile_row_group = value; return FGS_OK;
        case 11: g_library_bucket_scan = value ? 1 : 0; return FGS_OK;
        case 12: fgs::g_plan_experiment = value & 3; return FGS_OK;
        case 14: if (value < 1) return fail(FGS_ERR_INVALID_ARGUMENT, "the chained K11 needs at least one wave"); fgs::g_k11_chain_waves = value; return FGS_OK;
        case 5: if (value < 0 || value > 32) return fail(FGS_ERR_INVALID_ARGUMENT, "seq_tiles must be 0 (flattened counting) or 1..32");
                g_seq_tiles = value; return FGS_OK;
        default: return fail(FGS_ERR_INVALID_ARGUMENT, "unknown option %d", key);
    }
}

#endif  // FGS_DEV_SWITCHES

size_t fgs_debug_radix_sort_temp_bytes(int32_t n, int32_t end_bit) {
    return n < 0 ? 0 : own_sort_temp_bytes(static_cast<uint32_t>(n), end_bit);
}

int32_t fgs_debug_radix_sort(void* keys0, void* keys1, uint32_t* vals0, uint32_t* vals1, int32_t n, int32_t key_bytes, int32_t end_bit,
                             void* temp, size_t temp_bytes, void* stream) {
    if (n < 0 || (key_bytes != 2 && key_bytes != 4) || end_bit < 1 || end_bit > 8 * key_bytes)
        return fail(FGS_ERR_INVALID_ARGUMENT, "bad sort arguments");
    if (n > 0 && (!keys0 || !keys1 || !vals0 || !vals1 || !temp)) return fail(FGS_ERR_INVALID_ARGUMENT, "NULL buffer");
    int selector = 0;
    uint32_t* vals[2] = {vals0, vals1};
    if (key_bytes == 2) {
        uint16_t* k[2] = {static_cast<uint16_t*>(keys0), static_cast<uint16_t*>(keys1)};
        FGS_HIP(own_sort_pairs_u16(temp, temp_bytes, k, vals, selector, static_cast<uint32_t>(n), end_bit, static_cast<hipStream_t>(stream)));
    } else {
        uint32_t* k[2] = {static_cast<uint32_t*>(keys0), static_cast<uint32_t*>(keys1)};
        FGS_HIP(own_sort_pairs_u32(temp, temp_bytes, k, vals, selector, static_cast<uint32_t>(n), end_bit, static_cast<hipStream_t>(stream)));
    }
    return selector;                     // 0 / 1: which buffer pair holds the sorted result
}

int32_t fgs_debug_depth_sort(uint32_t* keys0, uint32_t* keys1, uint32_t* vals0, uint32_t* vals1, int32_t n, float near_plane, float far_plane,
                             void* temp, size_t temp_bytes, void* stream) {
    if (n < 0) return fail(FGS_ERR_INVALID_ARGUMENT, "bad sort arguments");
    if (n > 0 && (!keys0 || !keys1 || !vals0 || !vals1 || !temp)) return fail(FGS_ERR_INVALID_ARGUMENT, "NULL buffer");
    int selector = 0;
    uint32_t* k[2] = {keys0, keys1};
    uint32_t* vals[2] = {vals0, vals1};
    FGS_HIP(own_depth_sort(temp, temp_bytes, k, vals, selector, static_cast<uint32_t>(n), nullptr, depth_key_range(near_plane, far_plane),
                           static_cast<hipStream_t>(stream)));
    return selector;
}

int32_t fgs_debug_wave_selftest(uint32_t* out_device_256, void* stream) {
    if (!out_device_256) return fail(FGS_ERR_INVALID_ARGUMENT, "NULL output");
    FGS_HIP(launch_wave_selftest(out_device_256, static_cast<hipStream_t>(stream)));
    return FGS_OK;
}

#pragma GCC visibility pop
}  // extern "C"
