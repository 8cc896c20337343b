// K1: per-Gaussian preprocess for gfx950 -- cull, 3D covariance, EWA projection, conic, screen bounds, EXACT tile
// count, SH colour, compaction of the visible list. Semantics: reference kernels_forward.cuh:14-209 (training) and
// kernels_inference.cuh:14-207 (colour clamped at store, n_touched not cleared).
//
// CDNA4 shape: 256-thread workgroups (4 wave64). The exact tile count for large footprints is done by the whole
// wave, 64 candidate tiles per step, with the owning lane's parameters broadcast through v_readlane (SGPRs, no LDS);
// compaction uses ONE atomic per wave (64-bit ballot + mbcnt prefix) instead of one per visible Gaussian.
// Built with -ffp-contract=off: screen bounds / tile counts / depth keys are bit-reproducible against the oracle.
#include "fgs_kernels.h"
#include <fgs_wave.h>

namespace fgs {

// __float2int_rd / __float2int_ru of the reference (kf:171-174) saturate and map NaN to 0; the callers clamp the result
// to [0, padded extent], so clamping the float to +-1e9 first gives the same bounds without relying on fptosi overflow.
__device__ __forceinline__ int float_to_int_floor(float x) { return static_cast<int>(floorf(fminf(fmaxf(x, -1.0e9f), 1.0e9f))); }
__device__ __forceinline__ int float_to_int_ceil(float x) { return static_cast<int>(ceilf(fminf(fmaxf(x, -1.0e9f), 1.0e9f))); }

template <bool INFERENCE>
__global__ void __launch_bounds__(kPreprocessBlock) preprocess_kernel(const PreprocessArgs a) {
    const Camera cam = load_camera(a.cam);
    const unsigned gid = blockIdx.x * kPreprocessBlock + threadIdx.x;
    const unsigned lane = lane_id();
    bool active = gid < a.n;
    const unsigned idx = active ? gid : a.n - 1;

    if (!INFERENCE && active) a.n_touched[idx] = 0;       // kf:59 (K12 keys its skip test on this)

    float m[3];
    m[0] = a.means[3 * (size_t)idx]; m[1] = a.means[3 * (size_t)idx + 1]; m[2] = a.means[3 * (size_t)idx + 2];
    const float depth = view_depth(cam, m[0], m[1], m[2]);
    if (depth < cam.near_plane || depth > cam.far_plane) active = false;              // kf:67
    if (wave_ballot(active) == 0) return;

    float opacity = sigmoid_f(a.opacities[idx]);
    if (opacity < kMinAlphaThreshold) active = false;                                  // kf:75

    float s[3], q[4];
#pragma unroll
    for (int i = 0; i < 3; ++i) s[i] = a.scales[3 * (size_t)idx + i];
#pragma unroll
    for (int i = 0; i < 4; ++i) q[i] = a.rotations[4 * (size_t)idx + i];
    Projection P;
    project_gaussian(cam, m, s, q, P);
    if (P.norm_sq < 1e-8f) active = false;                                             // kf:83

    float cov_x = P.a_raw, cov_y = P.b, cov_z = P.c_raw;
    const float det_raw = cov_x * cov_z - cov_y * cov_y;
    const float ks = cam.proper_aa ? kDilationProperAA : kDilation;
    cov_x += ks; cov_z += ks;
    const float det = cov_x * cov_z - cov_y * cov_y;
    if (det < kMinCov2dDeterminant) active = false;                                    // kf:144
    const float ca = cov_z / det, cb = -cov_y / det, cc = cov_x / det;
    if (cam.proper_aa) {
        opacity *= sqrtf(fmaxf(det_raw / det, 0.0f));
        if (opacity < kMinAlphaThreshold) active = false;                              // kf:153
    }
    const float m2x = P.x * cam.fx + cam.cx, m2y = P.y * cam.fy + cam.cy;             // kf:157-160

    const float power_threshold = logf(opacity * kMinAlphaThresholdRcp);               // kf:163
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
    if (n_max == 0) active = false;                                                    // kf:178
    if (wave_ballot(active) == 0) return;

    // ---- exact tile count (kernel_utils.cuh:117-180), wave64 version ----
    const float sx = m2x - 0.5f, sy = m2y - 0.5f;
    unsigned cnt = 0;
    if (active) {
        const unsigned n_seq = n_max < (unsigned)kSeqTiles ? n_max : (unsigned)kSeqTiles;
        for (unsigned t = 0; t < n_seq; ++t)
            cnt += tile_contributes(sx, sy, ca, cb, cc, tx0 + t % tbw, ty0 + t / tbw, power_threshold) ? 1u : 0u;
    }
    uint64_t pending = wave_ballot(active && n_max > (unsigned)kSeqTiles);
    while (pending != 0) {                                  // wave-uniform loop over lanes with large footprints
        const int src = __ffsll(static_cast<unsigned long long>(pending)) - 1;
        pending &= pending - 1;
        const unsigned o_tx0 = wave_read(tx0, src), o_ty0 = wave_read(ty0, src);
        const unsigned o_tbw = wave_read(tbw, src), o_cnt = wave_read(n_max, src);
        const float o_sx = wave_read(sx, src), o_sy = wave_read(sy, src);
        const float o_ca = wave_read(ca, src), o_cb = wave_read(cb, src), o_cc = wave_read(cc, src);
        const float o_pt = wave_read(power_threshold, src);
        unsigned found = 0;
        for (unsigned base = kSeqTiles; base < o_cnt; base += kWave) {
            const unsigned t = base + lane;
            const bool hit = t < o_cnt && tile_contributes(o_sx, o_sy, o_ca, o_cb, o_cc, o_tx0 + t % o_tbw, o_ty0 + t / o_tbw, o_pt);
            found += static_cast<unsigned>(__popcll(static_cast<unsigned long long>(wave_ballot(hit))));
        }
        if (lane == static_cast<unsigned>(src)) cnt += found;
    }

    const bool visible = active && cnt > 0;                                            // kf:190
    const uint64_t vis_mask = wave_ballot(visible);
    if (vis_mask == 0) return;

    if (visible) {
        float col[3];
        const float* k = a.sh_rest + (size_t)idx * cam.total_sh_rest * 3;
        sh_to_color(a.sh0 + 3 * (size_t)idx, k, m[0] - cam.pos[0], m[1] - cam.pos[1], m[2] - cam.pos[2],
                    (unsigned)cam.active_sh_bases, col);
        if (INFERENCE) { col[0] = fmaxf(col[0], 0.0f); col[1] = fmaxf(col[1], 0.0f); col[2] = fmaxf(col[2], 0.0f); }  // ki:200
        float4* dst = reinterpret_cast<float4*>(a.rec + idx);
        dst[0] = make_float4(m2x, m2y, ca, cb);
        dst[1] = make_float4(cc, opacity, col[0], col[1]);
        dst[2] = make_float4(col[2], __uint_as_float(bx), __uint_as_float(by), __uint_as_float(cnt));
        if (!INFERENCE) a.n_touched[idx] = cnt;
    }

    // ---- compaction: one atomic per wave (kf:204-208 uses one per Gaussian) ----
    const unsigned wave_instances = wave_sum(visible ? cnt : 0u);
    const int leader = __ffsll(static_cast<unsigned long long>(vis_mask)) - 1;
    unsigned base = 0;
    if (lane == static_cast<unsigned>(leader)) {
        base = atomicAdd(&a.counters[0], static_cast<unsigned>(__popcll(static_cast<unsigned long long>(vis_mask))));
        atomicAdd(&a.counters[1], wave_instances);
    }
    base = wave_read(base, leader);
    if (visible) {
        const unsigned off = base + lanes_below(vis_mask);
        a.depth_keys[off] = __float_as_uint(depth);
        a.prim_idx[off] = idx;
    }
}

hipError_t launch_preprocess(bool inference, const PreprocessArgs& a, hipStream_t s) {
    if (a.n == 0) return hipSuccess;
    const dim3 grid((a.n + kPreprocessBlock - 1) / kPreprocessBlock), block(kPreprocessBlock);
    if (inference) hipLaunchKernelGGL(preprocess_kernel<true>, grid, block, 0, s, a);
    else hipLaunchKernelGGL(preprocess_kernel<false>, grid, block, 0, s, a);
    return hipGetLastError();
}

}  // namespace fgs
