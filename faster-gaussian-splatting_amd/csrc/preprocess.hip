// K1: per-Gaussian preprocess for gfx950 -- cull, 3D covariance, EWA projection, conic, screen bounds, EXACT tile
// count, SH colour, compaction of the visible list. Semantics: reference kernels_forward.cuh:14-209 (training) and
// kernels_inference.cuh:14-207 (colour clamped at store, n_touched not cleared).
//
// CDNA4 shape: 256-thread workgroups (4 wave64; fgs_config.h: the compaction barrier couples fewer waves than at 512). The exact tile count for large footprints is done by the whole
// wave, 64 candidate tiles per step, with the owning lane's parameters broadcast through v_readlane (SGPRs, no LDS);
// compaction uses ONE packed 64-bit atomic per workgroup (ballot + mbcnt prefix inside, LDS across waves) instead of two
// atomics per visible Gaussian: measured on MI355X, per-wave atomics on the two counters alone cost 0.6 ms at 3 M Gaussians.
// Built with -ffp-contract=off: screen bounds / tile counts / depth keys are bit-reproducible against the oracle.
#include "fgs_kernels.h"
#include <fgs_wave.h>

namespace fgs {

// __float2int_rd / __float2int_ru of the reference (kf:171-174) saturate and map NaN to 0; the callers clamp the result
// to [0, padded extent], so clamping the float to +-1e9 first gives the same bounds without relying on fptosi overflow.
__device__ __forceinline__ int float_to_int_floor(float x) { return static_cast<int>(floorf(fminf(fmaxf(x, -1.0e9f), 1.0e9f))); }
__device__ __forceinline__ int float_to_int_ceil(float x) { return static_cast<int>(ceilf(fminf(fmaxf(x, -1.0e9f), 1.0e9f))); }

// Debug-only phase timer (tools/k1_phase_timer.sh builds a separate library with -DFGS_K1_PHASE_TIMER; the product build has none of it):
// every wave keeps the cycles it spent between two marks in (scalar) registers and stores them ONCE, to its own slot of g_k1_phase, at the
// end -- a first version with one atomic per mark onto eight shared words slowed the kernel 9x and measured only itself. A wave's memory
// waits land in the phase that first uses the data, i.e. where the wave stalls. Result at S2 (profiles/archive/r02_k1_phases.txt): loads +
// projection 20 %, flattened tile count 26 %, SH colour + record 23 %, and 24 % in the last phase -- waves waiting at the workgroup
// barrier of the compaction for slower siblings, not the counter's round trip: requesting the counter before the colour phase so that
// the round trip overlaps with it made the kernel 6.5 % SLOWER (profiles/archive/r02_ab_k1_early_counter.txt) and was reverted.
#ifdef FGS_K1_PHASE_TIMER
constexpr unsigned kK1TimerWaves = 1u << 17;
__device__ unsigned long long g_k1_phase[kK1TimerWaves * 8];
#define FGS_K1_MARK(i) do { const unsigned long long now_ = __builtin_readcyclecounter(); t_phase_[i] += now_ - t_prev_; t_prev_ = now_; } while (0)
#define FGS_K1_START unsigned long long t_phase_[8] = {}; unsigned long long t_prev_ = __builtin_readcyclecounter()
#define FGS_K1_FLUSH do { const unsigned w_ = (blockIdx.x * kPreprocessBlock + threadIdx.x) >> 6; \
                          if (lane_id() < 8u && w_ < kK1TimerWaves) g_k1_phase[w_ * 8u + lane_id()] += t_phase_[0] * (lane_id() == 0) + t_phase_[1] * (lane_id() == 1) + \
                              t_phase_[2] * (lane_id() == 2) + t_phase_[3] * (lane_id() == 3) + t_phase_[4] * (lane_id() == 4) + t_phase_[5] * (lane_id() == 5); } while (0)
#else
#define FGS_K1_MARK(i) do { } while (0)
#define FGS_K1_START do { } while (0)
#define FGS_K1_FLUSH do { } while (0)
#endif

template <bool INFERENCE>
__device__ __forceinline__ void preprocess_body(const PreprocessArgs& a) {
    FGS_K1_START;
    const Camera cam = load_camera(a.cam);
    const unsigned gid = blockIdx.x * kPreprocessBlock + threadIdx.x;
    const unsigned lane = lane_id(), wave = threadIdx.x >> 6;
    bool active = gid < a.n;
    const unsigned idx = active ? gid : a.n - 1;
    // K0 (fwd:54): clear the per-tile ranges here instead of a separate memset launch (tiles without instances stay (0,0))
    for (unsigned t = gid; t < a.n_tiles; t += gridDim.x * kPreprocessBlock) a.ranges[t] = make_uint2(0u, 0u);

    float m[3];
    m[0] = a.means[3 * (size_t)idx]; m[1] = a.means[3 * (size_t)idx + 1]; m[2] = a.means[3 * (size_t)idx + 2];
    // (round 6, measured: opacity / scales / rotation requested here, together with the mean -- one round trip less on paper -- 0.194 vs 0.196 ms:
    // nothing, profiles/r06_ab_k1_hoisted_loads.txt)
    const float depth = view_depth(cam, m[0], m[1], m[2]);
    if (depth < cam.near_plane || depth > cam.far_plane) active = false;              // kf:67
    FGS_K1_MARK(0);                                                                    // camera + mean loaded, depth cull

    bool visible = false;
    unsigned cnt = 0;
    uint32_t foot_box = kFootprintEscape, foot_lo = 0, foot_hi = 0, foot_n_max = 0;        // the Gaussian's footprint row (fgs_math.h)
    if (wave_ballot(active) != 0) {                                                    // wave-uniform: skip culled waves (kf:70)
        float opacity = sigmoid_f(a.opacities[idx]);
        if (opacity < kMinAlphaThreshold) active = false;                              // kf:75

        float s[3], q[4];
#pragma unroll
        for (int i = 0; i < 3; ++i) s[i] = a.scales[3 * (size_t)idx + i];
#pragma unroll
        for (int i = 0; i < 4; ++i) q[i] = a.rotations[4 * (size_t)idx + i];
        Projection P;
        project_gaussian(cam, m, s, q, P);
        if (P.norm_sq < 1e-8f) active = false;                                         // kf:83

        float cov_x = P.a_raw, cov_y = P.b, cov_z = P.c_raw;
        const float det_raw = cov_x * cov_z - cov_y * cov_y;
        const float ks = cam.proper_aa ? kDilationProperAA : kDilation;
        cov_x += ks; cov_z += ks;
        const float det = cov_x * cov_z - cov_y * cov_y;
        if (det < kMinCov2dDeterminant) active = false;                                // kf:144
        const float ca = cov_z / det, cb = -cov_y / det, cc = cov_x / det;
        if (cam.proper_aa) {
            opacity *= sqrtf(fmaxf(det_raw / det, 0.0f));
            if (opacity < kMinAlphaThreshold) active = false;                          // kf:153
        }
        const float m2x = P.x * cam.fx + cam.cx, m2y = P.y * cam.fy + cam.cy;         // kf:157-160

        const float power_threshold = logf(opacity * kMinAlphaThresholdRcp);           // kf:163
        const float cutoff = 2.0f * power_threshold;
        const float ext_x = fmaxf(sqrtf(cov_x * cutoff) - 0.5f, 0.0f);
        const float ext_y = fmaxf(sqrtf(cov_z * cutoff) - 0.5f, 0.0f);
        const int padded_w = static_cast<int>(cam.grid_w * kTileW), padded_h = static_cast<int>(cam.grid_h * kTileH);
        const unsigned x_min = static_cast<unsigned>(min(padded_w, max(0, float_to_int_floor(m2x - ext_x))));
        const unsigned x_max = static_cast<unsigned>(min(padded_w, max(0, float_to_int_ceil(m2x + ext_x))));
        const unsigned y_min = static_cast<unsigned>(min(padded_h, max(0, float_to_int_floor(m2y - ext_y))));
        const unsigned y_max = static_cast<unsigned>(min(padded_h, max(0, float_to_int_ceil(m2y + ext_y))));
        const uint32_t bx = x_min | (x_max << 16), by = y_min | (y_max << 16);
        unsigned tx0, tx1, ty0, ty1;
        tile_rect(bx, by, tx0, tx1, ty0, ty1);
        const unsigned tbw = tx1 - tx0;
        const unsigned n_max = tbw * (ty1 - ty0);
        if (n_max == 0) active = false;                                                // kf:178
        FGS_K1_MARK(1);                                                                // opacity / scale / rotation loads, projection, bounds

        if (wave_ballot(active) != 0) {                                                // kf:181
            // ---- exact tile count (kernel_utils.cuh:117-180), wave64 version ----
            // Footprints above kHugeFootprint candidate tiles are counted by preprocess_huge_kernel (a workgroup each):
            // Morton order keeps the Gaussians nearest to the camera in the same wave, and counting their tens of
            // thousands of candidates here would stall that wave and the workgroup barrier below for ~0.3 ms.
            const bool huge = active && n_max > kHugeFootprint;
            if (huge) active = false;
            const TileTest tt = make_tile_test(m2x - 0.5f, m2y - 0.5f, ca, cb, cc, power_threshold);
            uint64_t hit_mask = 0;          // bit t = candidate tile t (row-major in the tile bounding box) is overlapped; t < 64
            unsigned first_shared = 0;      // candidates below this index were handled by one of the first two schemes
            if (a.seq_tiles > 0) {
                // (A/B reference, fgs_debug_set_option(5, n)) the reference's scheme: every lane tests the first n candidates of its
                // own Gaussian (cfg:54: 4), the wave cooperates on the rest. At a mean footprint of 9 candidates a wave runs all n
                // rounds with half of its lanes idle.
                first_shared = static_cast<unsigned>(a.seq_tiles);
                if (active) {
                    const unsigned n_seq = n_max < first_shared ? n_max : first_shared;
                    for (unsigned t = 0; t < n_seq; ++t)
                        if (tile_contributes(tt, tx0 + t % tbw, ty0 + t / tbw)) { ++cnt; hit_mask |= 1ull << t; }
                }
            } else {
                // Flattened: the (Gaussian, candidate) pairs of all footprints of <= 64 candidates are laid end to end (prefix sum of
                // the counts) and the wave takes 64 PAIRS per round, every lane busy: lane l of round r handles pair 64 r + l, finds its
                // owner with a 6-step search over the prefix array in the wave's LDS slice, and each owner picks its own bits out of
                // the round's ballot. ~9 rounds per wave at S2 instead of 16 half-empty ones.
                __shared__ uint32_t s_pre[kPreprocessBlock / kWave][kWave];
                __shared__ float4 s_g0[kPreprocessBlock / kWave][kWave], s_g1[kPreprocessBlock / kWave][kWave];   // sx sy ca cb | cc pt rcp_x rcp_y
                __shared__ uint2 s_box[kPreprocessBlock / kWave][kWave];                                          // tx0 | ty0 << 16, width | ceil(2^16 / width) << 8
                first_shared = kWave;
                const bool flat = active && n_max <= static_cast<unsigned>(kWave);
                const uint32_t n_flat = flat ? n_max : 0u;
                const uint32_t my_pre = wave_exclusive_sum(n_flat);
                const uint32_t total = wave_read(my_pre + n_flat, kWave - 1);
                s_pre[wave][lane] = my_pre;
                s_g0[wave][lane] = make_float4(tt.sx, tt.sy, tt.ca, tt.cb);
                s_g1[wave][lane] = make_float4(tt.cc, tt.pt, tt.rcp_x, tt.rcp_y);
                s_box[wave][lane] = make_uint2(tx0 | (ty0 << 16), tbw | (((65536u + tbw - 1u) / (tbw ? tbw : 1u)) << 8));
                wave_lds_fence();
                for (uint32_t r0 = 0; r0 < total; r0 += kWave) {                       // wave-uniform
                    const uint32_t p = r0 + lane;
                    unsigned o = 0;                                                    // largest lane with s_pre <= p (ties: the owning lane is last)
#pragma unroll
                    for (unsigned step = 32; step >= 1; step >>= 1) {
                        const unsigned mid = o + step;
                        if (mid < static_cast<unsigned>(kWave) && s_pre[wave][mid] <= p) o = mid;
                    }
                    const uint32_t t = p - s_pre[wave][o];
                    const float4 g0 = s_g0[wave][o], g1 = s_g1[wave][o];
                    const uint2 box = s_box[wave][o];
                    const unsigned w = box.y & 0xffu, row = (t * (box.y >> 8)) >> 16, col = t - row * w;      // t / w, t % w for t < 64, w <= 64
                    TileTest ot;
                    ot.sx = g0.x; ot.sy = g0.y; ot.ca = g0.z; ot.cb = g0.w; ot.cc = g1.x; ot.pt = g1.y; ot.rcp_x = g1.z; ot.rcp_y = g1.w;
                    ot.den_x = (static_cast<float>(kTileW - 1) * ot.ca) * static_cast<float>(kTileW - 1);
                    ot.den_y = (static_cast<float>(kTileH - 1) * ot.cc) * static_cast<float>(kTileH - 1);
                    const bool hit = p < total && tile_contributes(ot, (box.x & 0xffffu) + col, (box.x >> 16) + row);
                    const uint64_t hits = wave_ballot(hit);
                    // owner side: my pairs are [my_pre, my_pre + n_flat); their part inside this round is lanes [lo, hi)
                    const int rel = static_cast<int>(my_pre) - static_cast<int>(r0);
                    const int lo = min(max(rel, 0), kWave), hi = min(max(rel + static_cast<int>(n_flat), 0), kWave);
                    if (hi > lo) {
                        const int len = hi - lo;
                        const uint64_t mine = (hits >> lo) & (len >= 64 ? ~0ull : ((1ull << len) - 1ull));
                        cnt += static_cast<unsigned>(__popcll(static_cast<unsigned long long>(mine)));
                        const int first_t = lo - rel;                                  // candidate index of the pair at lane lo (< 64: flat footprints only)
                        hit_mask |= mine << first_t;
                    }
                }
            }
            FGS_K1_MARK(2);                                                            // flattened exact tile count
            uint64_t pending = wave_ballot(active && n_max > first_shared);
            while (pending != 0) {                              // wave-uniform loop over lanes with large footprints
                const int src = __ffsll(static_cast<unsigned long long>(pending)) - 1;
                pending &= pending - 1;
                const unsigned o_tx0 = wave_read(tx0, src), o_ty0 = wave_read(ty0, src);
                const unsigned o_tbw = wave_read(tbw, src), o_cnt = wave_read(n_max, src);
                TileTest ot;
                ot.sx = wave_read(tt.sx, src); ot.sy = wave_read(tt.sy, src);
                ot.ca = wave_read(tt.ca, src); ot.cb = wave_read(tt.cb, src); ot.cc = wave_read(tt.cc, src);
                ot.pt = wave_read(tt.pt, src);
                ot.den_x = wave_read(tt.den_x, src); ot.den_y = wave_read(tt.den_y, src);
                ot.rcp_x = wave_read(tt.rcp_x, src); ot.rcp_y = wave_read(tt.rcp_y, src);
                const unsigned first = a.seq_tiles > 0 ? first_shared : 0u;
                unsigned found = 0;
                for (unsigned base = first; base < o_cnt; base += kWave) {
                    const unsigned t = base + lane;
                    const bool hit = t < o_cnt && tile_contributes(ot, o_tx0 + t % o_tbw, o_ty0 + t / o_tbw);
                    const uint64_t hits = wave_ballot(hit);
                    found += static_cast<unsigned>(__popcll(static_cast<unsigned long long>(hits)));
                    if (base == first && first < 64u && lane == static_cast<unsigned>(src)) hit_mask |= hits << first;
                }
                if (lane == static_cast<unsigned>(src)) cnt += found;
            }

            FGS_K1_MARK(3);                                                            // footprints of > 64 candidates
            visible = active && cnt > 0;                                               // kf:190
            foot_n_max = n_max;
            if (footprint_box_fits(tx0, ty0, tbw, n_max)) {                            // every candidate of the box was tested above: the bitmap is exact
                foot_box = footprint_box(tx0, ty0, tbw, ty1 - ty0);
                foot_lo = static_cast<uint32_t>(hit_mask); foot_hi = static_cast<uint32_t>(hit_mask >> 32);
            }
            // hot-accumulator slots for K11 (fgs_config.h): one counter atomic per wave that holds such a footprint
            uint32_t slot_word = n_max <= 32u ? static_cast<uint32_t>(hit_mask) : 0u;
            const bool hot = (visible || huge) && n_max > kHotFootprint;
            const uint64_t hot_mask = wave_ballot(hot);
            if (hot_mask != 0) {
                const int leader = __ffsll(static_cast<unsigned long long>(hot_mask)) - 1;
                unsigned hot_base = 0;
                if (lane == static_cast<unsigned>(leader)) hot_base = atomicAdd(&a.counters[4], static_cast<unsigned>(__popcll(static_cast<unsigned long long>(hot_mask))));
                hot_base = wave_read(hot_base, leader);
                const unsigned slot = hot_base + lanes_below(hot_mask);
                if (hot && slot < kMaxHot) { a.hot_list[slot] = idx; slot_word = slot + 1u; }
            }
            if (visible || huge) {
                float col[3];
#if defined(FGS_K1_SH_PROBE)
                // TIMING PROBE, wrong colours (tools/build_variant.sh k1probe preprocess.hip -DFGS_K1_SH_PROBE): the lane's 45 coefficients are taken from the
                // wave's 11.25 KB block with perfectly coalesced 16-byte loads (lane l takes float4 i*64 + l of the block) -- what staging the block through LDS
                // could reach at most, without the LDS traffic. Measured: 0.198-0.201 -> 0.188-0.189 ms (profiles/r06_ab_k1_sh_probe.txt): the staging was not built.
                float kk[48];
                {
                    const size_t wave_first = (size_t)(idx & ~63u) * 45u;
                    const size_t total = (size_t)a.n * 45u;
                    const float* blk = a.sh_rest + wave_first;
#pragma unroll
                    for (int i = 0; i < 12; ++i) {
                        const size_t e = ((size_t)i * 64u + lane) * 4u;
                        const bool in = i < 11 ? (wave_first + e + 3u < total) : (lane < 16u && wave_first + e + 3u < total);
                        const float4 v = in ? *reinterpret_cast<const float4*>(blk + e) : make_float4(0.f, 0.f, 0.f, 0.f);
                        kk[4 * i] = v.x; kk[4 * i + 1] = v.y; kk[4 * i + 2] = v.z; kk[4 * i + 3] = v.w;
                    }
                }
                const float* k = kk;
#else
                const float* k = a.sh_rest + (size_t)idx * cam.total_sh_rest * 3;
#endif
                sh_to_color(a.sh0 + 3 * (size_t)idx, k, m[0] - cam.pos[0], m[1] - cam.pos[1], m[2] - cam.pos[2],
                            (unsigned)cam.active_sh_bases, col);
                if (INFERENCE) { col[0] = fmaxf(col[0], 0.0f); col[1] = fmaxf(col[1], 0.0f); col[2] = fmaxf(col[2], 0.0f); }  // ki:200
                float4* dst = reinterpret_cast<float4*>(a.rec + idx);
                dst[0] = make_float4(m2x, m2y, ca, cb);
                dst[1] = make_float4(cc, opacity, col[0], col[1]);
                // footprints of <= 32 candidate tiles hand their exact-overlap bitmap to the instance generator; hot ones their slot
                dst[2] = make_float4(col[2], __uint_as_float(bx), __uint_as_float(by), __uint_as_float(slot_word));
                // training: K11's accumulator record of this Gaussian starts at zero (replaces api:127-134; the only records K11 adds into and K12
                // reads are those of visible Gaussians). Nine stores that nothing waits for, in a kernel whose HBM traffic is a third of the rate.
                if (!INFERENCE && a.acc != nullptr) {
                    float* const z = a.acc + (size_t)idx * kAccRecordWords;
#pragma unroll
                    for (unsigned k = 0; k < kAccRecordWords; ++k) z[k] = 0.0f;      // (as non-temporal stores: no gain, profiles/r06_ab_k1_acc_nt.txt)
                }
            }
            const uint64_t huge_mask = wave_ballot(huge);
            if (huge_mask != 0) {
                const int leader = __ffsll(static_cast<unsigned long long>(huge_mask)) - 1;
                unsigned hbase = 0;
                if (lane == static_cast<unsigned>(leader)) hbase = atomicAdd(&a.counters[3], static_cast<unsigned>(__popcll(static_cast<unsigned long long>(huge_mask))));
                hbase = wave_read(hbase, leader);
                if (huge) a.huge_list[hbase + lanes_below(huge_mask)] = idx;
            }
        }
    }
    FGS_K1_MARK(4);                                             // hot slots, SH colour, record write (waves that were not culled whole)
    if (gid < a.n) a.n_touched[idx] = visible ? cnt : 0u;      // kf:59,193; also the scan input of K4 and the skip test of K12

    // ---- compaction (kf:204-208): ONE 64-bit atomic per workgroup (256 threads). Both counters share one word
    // (low = n_visible, high = n_instances): a same-address atomic retires at ~88/us on this chip, so one per Gaussian
    // (reference) or even one per wave would serialise the whole kernel behind the counter. ----
    __shared__ unsigned s_vis[kPreprocessBlock / kWave], s_inst[kPreprocessBlock / kWave];
    __shared__ unsigned s_base;
    const uint64_t vis_mask = wave_ballot(visible);
    const unsigned wave_instances = wave_sum(visible ? cnt : 0u);
    if (lane == 0) { s_vis[wave] = static_cast<unsigned>(__popcll(static_cast<unsigned long long>(vis_mask))); s_inst[wave] = wave_instances; }
    __syncthreads();
    if (threadIdx.x == 0) {
        unsigned v = 0, inst = 0;
#pragma unroll
        for (int w = 0; w < kPreprocessBlock / kWave; ++w) { v += s_vis[w]; inst += s_inst[w]; }
        unsigned base = 0;
        if (v != 0) {
            const unsigned long long packed = (static_cast<unsigned long long>(inst) << 32) | v;
            base = static_cast<unsigned>(atomicAdd(reinterpret_cast<unsigned long long*>(a.counters), packed));
        }
        s_base = base;
    }
    __syncthreads();
    if (visible) {
        unsigned off = s_base + lanes_below(vis_mask);
        for (unsigned w = 0; w < wave; ++w) off += s_vis[w];
        a.depth_keys[off] = __float_as_uint(depth);
        // the visible list's primitive index travels in the footprint row (the depth sort makes up its own values and takes the primitive from the
        // row in its last pass); the bare index list is written for the sharded owner only, whose record packing reads it
        if (a.foot != nullptr) a.foot[off] = foot_box == kFootprintEscape ? make_uint4(idx, kFootprintEscape, cnt, foot_n_max) : make_uint4(idx, foot_box, foot_lo, foot_hi);
        else a.prim_idx[off] = idx;
    }
    FGS_K1_MARK(5);                                             // tile-count store, workgroup barrier + compaction atomic, key / index store
    FGS_K1_FLUSH;
}

// The register budget is capped for FGS_K1_WAVES waves per SIMD. History: with packed fp32 instructions and 512-thread workgroups the uncapped
// kernel took 116 VGPRs (two workgroups per CU) and was latency-bound: 0.216 ms uncapped, 0.196 at 6 waves (80 VGPRs, 43 spilled), 0.215 at 5,
// 0.222 at 8 (profiles/archive/r03_ab_k1_occupancy.txt). Built without packed fp32 (Makefile) it needs 63 VGPRs and no scratch at any cap, and its
// time does not depend on the instruction count; MORE than 6 waves per SIMD measures slower (0.226 vs 0.213 ms, r03_ab_nopk.txt: the lanes'
// 180-byte-stride coefficient reads thrash the 32 KB L1 sooner), so the cap stays. 0 = no cap.
// Round 5 (the kernel now also writes the footprint rows; 67 registers, no scratch): 5 / 6 / 7 / 8 waves 0.198 / 0.195 / 0.191 / 0.192 ms, 512-thread
// workgroups 0.200, 128-thread ones 0.257 (profiles/r05_ab_k1_shapes.txt): 7 it is.
#ifndef FGS_K1_WAVES
#define FGS_K1_WAVES 7
#endif
#if FGS_K1_WAVES > 0
#define FGS_K1_BOUNDS __launch_bounds__(kPreprocessBlock) __attribute__((amdgpu_waves_per_eu(FGS_K1_WAVES, FGS_K1_WAVES)))
#else
#define FGS_K1_BOUNDS __launch_bounds__(kPreprocessBlock)
#endif
template <bool INFERENCE>
__global__ void FGS_K1_BOUNDS preprocess_kernel(const PreprocessArgs a) { preprocess_body<INFERENCE>(a); }
__global__ void __launch_bounds__(kPreprocessBlock) preprocess_batch_kernel(const PreprocessBatch b) { preprocess_body<false>(b.v[blockIdx.y]); }

// Exact tile count + compaction for the few screen-filling footprints: one kHugeBlock-thread workgroup per Gaussian, kHugeBlock candidate
// tiles per step (kernel_utils.cuh:117-180 with the whole workgroup cooperating instead of one warp). 1024 threads (round 5; 256 before): the
// kernel's span is the step chain of its largest footprint (10 800 candidates at 1080p: 11 steps instead of 43).
constexpr unsigned kHugeBlock = 1024;
__device__ __forceinline__ void preprocess_huge_body(const PreprocessArgs& a) {
    __shared__ unsigned s_cnt[kHugeBlock / kWave];
    const Camera cam = load_camera(a.cam);
    const unsigned lane = lane_id(), wv = threadIdx.x >> 6;
    const unsigned n_huge = a.counters[3];
    for (unsigned h = blockIdx.x; h < n_huge; h += gridDim.x) {            // workgroup-uniform
        const uint32_t idx = a.huge_list[h];
        const float4* r = reinterpret_cast<const float4*>(a.rec + idx);
        const float4 r0 = r[0], r1 = r[1], r2 = r[2];
        const TileTest tt = make_tile_test(r0.x - 0.5f, r0.y - 0.5f, r0.z, r0.w, r1.x, logf(r1.y * kMinAlphaThresholdRcp));
        unsigned tx0, tx1, ty0, ty1;
        tile_rect(__float_as_uint(r2.y), __float_as_uint(r2.z), tx0, tx1, ty0, ty1);
        const unsigned tbw = tx1 - tx0, count = tbw * (ty1 - ty0);
        unsigned mine = 0;
        for (unsigned t = threadIdx.x; t < count; t += kHugeBlock)
            mine += tile_contributes(tt, tx0 + t % tbw, ty0 + t / tbw) ? 1u : 0u;
        const unsigned wsum = wave_sum(mine);
        if (lane == 0) s_cnt[wv] = wsum;
        __syncthreads();
        if (threadIdx.x == 0) {
            unsigned cnt = 0;
            for (unsigned k = 0; k < kHugeBlock / kWave; ++k) cnt += s_cnt[k];
            a.n_touched[idx] = cnt;
            if (cnt != 0) {
                const unsigned long long packed = (static_cast<unsigned long long>(cnt) << 32) | 1ull;
                const unsigned off = static_cast<unsigned>(atomicAdd(reinterpret_cast<unsigned long long*>(a.counters), packed));
                const float depth = view_depth(cam, a.means[3 * (size_t)idx], a.means[3 * (size_t)idx + 1], a.means[3 * (size_t)idx + 2]);
                a.depth_keys[off] = __float_as_uint(depth);
                if (a.foot != nullptr) a.foot[off] = make_uint4(idx, kFootprintEscape, cnt, count);
                else a.prim_idx[off] = idx;
                if (a.count_appended) atomicAdd(&a.counters[2], 1u);     // sharded path: how many entries this kernel appended
            }
        }
        __syncthreads();
    }
}
__global__ void __launch_bounds__(kHugeBlock) preprocess_huge_kernel(const PreprocessArgs a) { preprocess_huge_body(a); }
__global__ void __launch_bounds__(kHugeBlock) preprocess_huge_batch_kernel(const PreprocessBatch b) { preprocess_huge_body(b.v[blockIdx.y]); }

// (Round 6, measured and withdrawn: K1's counter block cleared and published by one-wave kernels of our own instead of hipMemsetAsync /
// hipMemcpyAsync -- the trace shows a ~6 us gap behind each of the runtime's two operations -- left the forward-only frame unchanged, 0.674-0.682 vs
// 0.673-0.675 ms in alternating processes on one box: the gaps are the host's wake-up behind the event, not the operations. profiles/r06_ab_counter_kernels.txt)
hipError_t launch_preprocess(bool inference, const PreprocessArgs& a, hipStream_t s) {
    if (a.n == 0) return hipSuccess;
    const dim3 grid((a.n + kPreprocessBlock - 1) / kPreprocessBlock), block(kPreprocessBlock);
    if (inference) hipLaunchKernelGGL(preprocess_kernel<true>, grid, block, 0, s, a);
    else hipLaunchKernelGGL(preprocess_kernel<false>, grid, block, 0, s, a);
    hipLaunchKernelGGL(preprocess_huge_kernel, dim3(a.n < 512u ? a.n : 512u), dim3(kHugeBlock), 0, s, a);
    return hipGetLastError();
}

hipError_t launch_preprocess_batch(const PreprocessBatch& b, hipStream_t s) {
    const uint32_t n = b.v[0].n;
    if (n == 0 || b.n_views <= 0) return hipSuccess;
    const dim3 grid((n + kPreprocessBlock - 1) / kPreprocessBlock, static_cast<unsigned>(b.n_views)), block(kPreprocessBlock);
    hipLaunchKernelGGL(preprocess_batch_kernel, grid, block, 0, s, b);
    hipLaunchKernelGGL(preprocess_huge_batch_kernel, dim3(n < 256u ? n : 256u, static_cast<unsigned>(b.n_views)), dim3(kHugeBlock), 0, s, b);
    return hipGetLastError();
}

}  // namespace fgs

#ifdef FGS_K1_PHASE_TIMER
// debug build only: sum (and optionally clear) the per-wave, per-phase cycle counts of preprocess_body
extern "C" __attribute__((visibility("default"))) int fgs_debug_k1_phases(unsigned long long* out8, int reset) {
    static unsigned long long host[fgs::kK1TimerWaves * 8];
    if (hipMemcpyFromSymbol(host, HIP_SYMBOL(fgs::g_k1_phase), sizeof(host)) != hipSuccess) return -1;
    for (int i = 0; i < 8; ++i) out8[i] = 0;
    for (unsigned w = 0; w < fgs::kK1TimerWaves; ++w) for (int i = 0; i < 8; ++i) out8[i] += host[w * 8u + i];
    if (reset) {
        void* dev = nullptr;
        if (hipGetSymbolAddress(&dev, HIP_SYMBOL(fgs::g_k1_phase)) != hipSuccess || hipMemset(dev, 0, sizeof(host)) != hipSuccess) return -1;
    }
    return 0;
}
// the raw per-wave table ([wave][8] cycles, waves in launch order): for the distribution of a phase over the waves (is the kernel's time a tail?)
extern "C" __attribute__((visibility("default"))) int fgs_debug_k1_phase_waves(unsigned long long* out, unsigned n_waves) {
    if (n_waves > fgs::kK1TimerWaves) n_waves = fgs::kK1TimerWaves;
    return hipMemcpyFromSymbol(out, HIP_SYMBOL(fgs::g_k1_phase), sizeof(unsigned long long) * 8u * n_waves) == hipSuccess ? 0 : -1;
}
#endif
