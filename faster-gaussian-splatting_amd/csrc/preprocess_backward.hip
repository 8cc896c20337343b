// K12 (+ optional K13 fusion): per-Gaussian backward of projection / EWA / covariance / quaternion / SH, for gfx950.
// Semantics: reference kernels_backward.cuh:15-257 + sh_utils.cuh:71-155; the fused mode equals
// "backward -> FusedAdam.step()" of the reference (torch_bindings/adam.py:11-36, adam/src/adam.cu:10-34).
//
// CDNA4 shape -- the per-Gaussian payload is 59 floats of which 45 are sh_coefficients_rest, so the work is split by
// access pattern instead of by Gaussian:
//  * geometry kernel: one lane per Gaussian for the 14 "small" floats (means, scales, rotations, opacity, sh0): 12-16 B
//    per lane per tensor, contiguous across the wave. It also leaves the unit view direction of visible Gaussians in a
//    12-byte scratch slot.
//  * SH-rest kernel: the [N,15,3] tensors (parameter, both Adam moments, gradient) are streamed flat in 16-byte pieces by
//    a grid-stride loop; the gradient basis_k(dir) * dL/dcolour is recomputed per (Gaussian, basis) pair from the 12-byte
//    direction + 12-byte colour gradient (L1/L2 hits shared by neighbouring lanes) instead of being gathered with a
//    180-byte per-lane stride as in the reference.
//  * every element of every gradient is written (zeros for invisible Gaussians), which replaces the reference's eight
//    torch::zeros fills (rasterization_api.cu:127-134, 256 B per Gaussian).
//  * fused mode never materialises the 59-float gradient: Adam is applied in registers. Invisible Gaussians still
//    decay their moments and move by momentum (reference: dense zero grads, adam.py:16).
#include "fgs_kernels.h"
#include <fgs_wave.h>

namespace fgs {

__device__ __forceinline__ void adam_update(float& p, float& m, float& v, const float g, const AdamHyper& h) {   // adam.cu:22-33
    const float gsq = g * g;
    const float m1 = fmaf(h.beta1, m - g, g);
    const float m2 = fmaf(h.beta2, v - gsq, gsq);
    const float denom = sqrtf(m2) * h.bc2_sqrt_rcp + h.eps;
    p -= h.step_size * m1 / denom;
    m = m1;
    v = m2;
}

// fused mode: the 14 small floats of a Gaussian (kernel group order means 3, sh0 3, opacity 1, scales 3, rotations 4) and
// their two Adam moments are requested at kernel entry, so the 42 loads are in flight while the gradient is computed.
constexpr int kGroupWidth[5] = {3, 3, 1, 3, 4};
constexpr int kGroupOffset[5] = {0, 3, 6, 7, 10};

// MULTI: the sharded path's owner sums the gradient of its Gaussians over the views of the step in registers (one launch,
// every gradient element written once) instead of one launch and one read-modify-write of the gradients per view.
template <bool MULTI> __device__ __forceinline__ void sum_into(float& dst, const float v) { dst = MULTI ? dst + v : v; }

// Gradient of the 14 small floats of Gaussian i (group order means 3, sh0 3, opacity 1, scales 3, rotations 4), summed over the
// views of the launch; zeros if no view sees it. st_p: the parameters already in registers (fused mode). KEEP_DIR: the unit view
// direction and the colour gradient of the (single) view stay in registers for the caller instead of going through the
// view_dir scratch array.
template <bool FUSED, bool MULTI, bool KEEP_DIR>
__device__ __forceinline__ bool gaussian_backward(const PreprocessBackwardArgs& a, const unsigned i, const float (&st_p)[14],
                                                  float (&grad)[14], float (&dir)[3], float (&gcol_out)[3]) {
    const size_t n = a.n;
    float g_mean[3] = {0.0f, 0.0f, 0.0f}, g_scale[3] = {0.0f, 0.0f, 0.0f}, g_rot[4] = {0.0f, 0.0f, 0.0f, 0.0f};
    float g_op[1] = {0.0f}, g_sh0[3] = {0.0f, 0.0f, 0.0f};
    bool visible = false;
    float m[3] = {0.0f, 0.0f, 0.0f}, s[3] = {0.0f, 0.0f, 0.0f}, q[4] = {0.0f, 0.0f, 0.0f, 0.0f};
    const int n_views = MULTI ? a.n_views : 1;
    const Camera cam0 = load_camera(a.view[0].cam);     // single view: requested before the visibility test, as are the moments
    for (int vw = 0; vw < n_views; ++vw) {
        const BackwardView& V = a.view[MULTI ? vw : 0];
        if (V.n_touched[i] == 0) continue;                                             // kb:45
        if (!visible) {                                                                // parameters: once, whatever the number of views
#pragma unroll
            for (int k = 0; k < 3; ++k) {
                m[k] = FUSED ? st_p[0 + k] : a.means[3 * (size_t)i + k];
                s[k] = FUSED ? st_p[7 + k] : a.scales[3 * (size_t)i + k];
            }
#pragma unroll
            for (int k = 0; k < 4; ++k) q[k] = FUSED ? st_p[10 + k] : a.rotations[4 * (size_t)i + k];
        }
        visible = true;
        const Camera cam = (MULTI && vw > 0) ? load_camera(V.cam) : cam0;
        // K11's record of this primitive, 9 contiguous floats: single view [N][9]; sharded path: the record that came back, found through the slot table
        const float* const accp = V.acc + (size_t)(MULTI ? V.slot[i] : i) * kAccRecordWords;
        constexpr size_t es = 1;
        const float gcol[3] = {accp[6 * es], accp[7 * es], accp[8 * es]};
        if (KEEP_DIR) { gcol_out[0] = gcol[0]; gcol_out[1] = gcol[1]; gcol_out[2] = gcol[2]; }

        // ---- SH backward w.r.t. sh0 and the view direction (sh_utils.cuh:84-153) ----
#pragma unroll
        for (int c = 0; c < 3; ++c) sum_into<MULTI>(g_sh0[c], kC0 * gcol[c]);
        float dpos[3] = {0.0f, 0.0f, 0.0f};
        const unsigned active = static_cast<unsigned>(cam.active_sh_bases);
        if (active > 1) {
            const float xr = m[0] - cam.pos[0], yr = m[1] - cam.pos[1], zr = m[2] - cam.pos[2];
            const float inv = 1.0f / sqrtf(xr * xr + yr * yr + zr * zr);
            const float x = xr * inv, y = yr * inv, z = zr * inv;
            if (KEEP_DIR) { dir[0] = x; dir[1] = y; dir[2] = z; }
            else { V.view_dir[3 * (size_t)i] = x; V.view_dir[3 * (size_t)i + 1] = y; V.view_dir[3 * (size_t)i + 2] = z; }
            // (round 6, measured and withdrawn: the wave's contiguous 11.25 KB coefficient block taken with coalesced 16-byte loads into the LDS slice the
            // products go to afterwards, every lane reading its 45 words from there -- K12 0.282 vs 0.251 ms: the load -> LDS -> read chain in front of
            // the arithmetic costs more at 3 waves per SIMD than the lane-strided loads do, profiles/r06_ab_k12_staged_coeffs.txt)
            const float* k = a.sh_rest + (size_t)i * cam.total_sh_rest * 3;
            float gdx[3], gdy[3], gdz[3];
#pragma unroll
            for (int c = 0; c < 3; ++c) { gdx[c] = -kC1 * k[6 + c]; gdy[c] = -kC1 * k[0 + c]; gdz[c] = kC1 * k[3 + c]; }
            if (active > 4) {
                const float xx = x * x, yy = y * y, zz = z * z, xy = x * y, xz = x * z, yz = y * z;
#pragma unroll
                for (int c = 0; c < 3; ++c) {
                    gdx[c] = gdx[c] + kC2a * y * k[9 + c] - kC2a * z * k[18 + c] + kC2a * x * k[21 + c];
                    gdy[c] = gdy[c] + kC2a * x * k[9 + c] - kC2a * z * k[12 + c] - kC2a * y * k[21 + c];
                    gdz[c] = gdz[c] - kC2a * y * k[12 + c] + kC2e * z * k[15 + c] - kC2a * x * k[18 + c];
                }
                if (active > 9) {
#pragma unroll
                    for (int c = 0; c < 3; ++c) {
                        gdx[c] = gdx[c] - kC3i * xy * k[24 + c] + kC3c * yz * k[27 + c] + (kC3d - kC3e * zz) * k[36 + c]
                                 + kC3c * xz * k[39 + c] + kC3b * (yy - xx) * k[42 + c];
                        gdy[c] = gdy[c] + kC3b * (yy - xx) * k[24 + c] + kC3c * xz * k[27 + c] + (kC3d - kC3e * zz) * k[30 + c]
                                 - kC3c * yz * k[39 + c] + kC3i * xy * k[42 + c];
                        gdz[c] = gdz[c] + kC3c * xy * k[27 + c] - kC3j * yz * k[30 + c] + (kC3k * zz - kC3g) * k[33 + c]
                                 - kC3j * xz * k[36 + c] + kC3h * (xx - yy) * k[39 + c];
                    }
                }
            }
            const float gd0 = gdx[0] * gcol[0] + gdx[1] * gcol[1] + gdx[2] * gcol[2];
            const float gd1 = gdy[0] * gcol[0] + gdy[1] * gcol[1] + gdy[2] * gcol[2];
            const float gd2 = gdz[0] * gcol[0] + gdz[1] * gcol[1] + gdz[2] * gcol[2];
            const float xxr = xr * xr, yyr = yr * yr, zzr = zr * zr, xyr = xr * yr, xzr = xr * zr, yzr = yr * zr;
            const float nsq = xxr + yyr + zzr;
            const float sc = 1.0f / sqrtf(nsq * nsq * nsq);
            dpos[0] = ((yyr + zzr) * gd0 - xyr * gd1 - xzr * gd2) * sc;
            dpos[1] = (-xyr * gd0 + (xxr + zzr) * gd1 - yzr * gd2) * sc;
            dpos[2] = (-xzr * gd0 - yzr * gd1 + (xxr + yyr) * gd2) * sc;
        }

        // ---- EWA backward (kb:57-255) ----
        Projection P;
        project_gaussian(cam, m, s, q, P);
        const float ks = cam.proper_aa ? kDilationProperAA : kDilation;
        const float ea = P.a_raw + ks, eb = P.b, ec = P.c_raw + ks;
        const float aa = ea * ea, bb = eb * eb, cc = ec * ec, ac = ea * ec, ab = ea * eb, bc = eb * ec;
        const float det = ac - bb;
        const float det_rcp_sq = 1.0f / (det * det);
        const float gcx = accp[2 * es], gcy = accp[3 * es], gcz = accp[4 * es];
        const float dcov_x = det_rcp_sq * (2.0f * bc * gcy - cc * gcx - bb * gcz);     // kb:130-134
        const float dcov_y = det_rcp_sq * (bc * gcx - (ac + bb) * gcy + ab * gcz);
        const float dcov_z = det_rcp_sq * (2.0f * ab * gcy - bb * gcx - aa * gcz);
        float d_opacity = accp[5 * es];
        if (cam.proper_aa) {                                                           // kb:137-145 (cov2d branch off, cfg:12)
            const float opacity = sigmoid_f(FUSED ? st_p[6] : a.opacities[i]);
            const float det_raw = P.a_raw * P.c_raw - bb;
            d_opacity = d_opacity * sqrtf(fmaxf(det_raw / det, 0.0f)) * opacity * (1.0f - opacity);
        }
        sum_into<MULTI>(g_op[0], d_opacity);
        const float* j1 = P.jw1; const float* j2 = P.jw2;
        float d3[6];                                                                   // kb:163-170
        d3[0] = j1[0] * j1[0] * dcov_x + 2.0f * j1[0] * j2[0] * dcov_y + j2[0] * j2[0] * dcov_z;
        d3[1] = j1[0] * j1[1] * dcov_x + (j1[0] * j2[1] + j1[1] * j2[0]) * dcov_y + j2[0] * j2[1] * dcov_z;
        d3[2] = j1[0] * j1[2] * dcov_x + (j1[0] * j2[2] + j1[2] * j2[0]) * dcov_y + j2[0] * j2[2] * dcov_z;
        d3[3] = j1[1] * j1[1] * dcov_x + 2.0f * j1[1] * j2[1] * dcov_y + j2[1] * j2[1] * dcov_z;
        d3[4] = j1[1] * j1[2] * dcov_x + (j1[1] * j2[2] + j1[2] * j2[1]) * dcov_y + j2[1] * j2[2] * dcov_z;
        d3[5] = j1[2] * j1[2] * dcov_x + 2.0f * j1[2] * j2[2] * dcov_y + j2[2] * j2[2] * dcov_z;
        float djw1[3], djw2[3];                                                        // kb:173-182
#pragma unroll
        for (int k = 0; k < 3; ++k) {
            djw1[k] = 2.0f * (P.jwc1[k] * dcov_x + P.jwc2[k] * dcov_y);
            djw2[k] = 2.0f * (P.jwc1[k] * dcov_y + P.jwc2[k] * dcov_z);
        }
        const float dj11 = cam.r1[0] * djw1[0] + cam.r1[1] * djw1[1] + cam.r1[2] * djw1[2];
        const float dj22 = cam.r2[0] * djw2[0] + cam.r2[1] * djw2[1] + cam.r2[2] * djw2[2];
        const float dj13 = cam.r3[0] * djw1[0] + cam.r3[1] * djw1[1] + cam.r3[2] * djw1[2];
        const float dj23 = cam.r3[0] * djw2[0] + cam.r3[1] * djw2[1] + cam.r3[2] * djw2[2];
        const float gm2x = accp[0], gm2y = accp[es];
        if (a.densification_info != nullptr) {                                         // kb:194-201
            a.densification_info[i] += 1.0f;
            const float nx = 0.5f * (gm2x * cam.width), ny = 0.5f * (gm2y * cam.height);
            a.densification_info[n + i] += sqrtf(nx * nx + ny * ny);
        }
        float dcam[3];                                                                 // kb:204-217
        dcam[0] = P.j11 * gm2x;
        dcam[1] = P.j22 * gm2y;
        dcam[2] = -P.j11 * P.x * gm2x - P.j22 * P.y * gm2y;
        const bool valid_x = P.x >= P.clip_l && P.x <= P.clip_r;
        const bool valid_y = P.y >= P.clip_t && P.y <= P.clip_b;
        if (valid_x) dcam[0] -= P.j11 * dj13 / P.depth;
        if (valid_y) dcam[1] -= P.j22 * dj23 / P.depth;
        const float fxm = valid_x ? 2.0f : 1.0f, fym = valid_y ? 2.0f : 1.0f;
        dcam[2] += (P.j11 * (fxm * P.x_clipped * dj13 - dj11) + P.j22 * (fym * P.y_clipped * dj23 - dj22)) / P.depth;
#pragma unroll
        for (int k = 0; k < 3; ++k)                                                    // kb:220-228
            sum_into<MULTI>(g_mean[k], (cam.r1[k] * dcam[0] + cam.r2[k] * dcam[1] + cam.r3[k] * dcam[2]) + dpos[k]);
        const float* R = P.R; const float* G = P.RSS;
#pragma unroll
        for (int k = 0; k < 3; ++k) {                                                  // kb:231-240
            const float dvar = R[k] * R[k] * d3[0] + R[3 + k] * R[3 + k] * d3[3] + R[6 + k] * R[6 + k] * d3[5]
                               + 2.0f * (R[k] * R[3 + k] * d3[1] + R[k] * R[6 + k] * d3[2] + R[3 + k] * R[6 + k] * d3[4]);
            sum_into<MULTI>(g_scale[k], 2.0f * P.var[k] * dvar);
        }
        float dR[9];                                                                   // kb:243-253
#pragma unroll
        for (int k = 0; k < 3; ++k) {
            dR[0 + k] = 2.0f * (G[0 + k] * d3[0] + G[3 + k] * d3[1] + G[6 + k] * d3[2]);
            dR[3 + k] = 2.0f * (G[0 + k] * d3[1] + G[3 + k] * d3[3] + G[6 + k] * d3[4]);
            dR[6 + k] = 2.0f * (G[0 + k] * d3[2] + G[3 + k] * d3[4] + G[6 + k] * d3[5]);
        }
        float d_rot[4];
        quat_to_rotation_backward(q[0], q[1], q[2], q[3], dR, d_rot);
#pragma unroll
        for (int k = 0; k < 4; ++k) sum_into<MULTI>(g_rot[k], d_rot[k]);
    }
    // group order: 0 means, 1 sh0, 2 opacities, 3 scales, 4 rotations
#pragma unroll
    for (int k = 0; k < 3; ++k) { grad[0 + k] = g_mean[k]; grad[3 + k] = g_sh0[k]; grad[7 + k] = g_scale[k]; }
    grad[6] = g_op[0];
#pragma unroll
    for (int k = 0; k < 4; ++k) grad[10 + k] = g_rot[k];
    return visible;
}

template <bool FUSED, bool MULTI>
__global__ void __launch_bounds__(kPreprocessBackwardBlock) preprocess_backward_kernel(const PreprocessBackwardArgs a) {
    const unsigned i = blockIdx.x * kPreprocessBackwardBlock + threadIdx.x;
    if (i >= a.n) return;
    float st_p[14], st_m[14], st_v[14];
    if (FUSED) {
#pragma unroll
        for (int grp = 0; grp < 5; ++grp)
#pragma unroll
            for (int k = 0; k < kGroupWidth[grp]; ++k) {
                const size_t e = (size_t)i * kGroupWidth[grp] + k;
                st_p[kGroupOffset[grp] + k] = a.p[grp][e]; st_m[kGroupOffset[grp] + k] = a.m[grp][e]; st_v[kGroupOffset[grp] + k] = a.v[grp][e];
            }
    }
    float grad[14], dir[3], gcol[3];
    const bool visible = gaussian_backward<FUSED, MULTI, false>(a, i, st_p, grad, dir, gcol);

    // Every element is written (zeros if invisible).
    float* const outs[5] = {a.grad_means, a.grad_sh0, a.grad_opacities, a.grad_scales, a.grad_rotations};
#pragma unroll
    for (int grp = 0; grp < 5; ++grp)
#pragma unroll
        for (int k = 0; k < kGroupWidth[grp]; ++k) {
            const size_t e = (size_t)i * kGroupWidth[grp] + k;
            const int o = kGroupOffset[grp] + k;
            if (!FUSED) {
                if (!MULTI || !a.accumulate) outs[grp][e] = grad[o];       // view batches after the first add (sharded path only)
                else if (visible) outs[grp][e] += grad[o];
            } else {
                adam_update(st_p[o], st_m[o], st_v[o], grad[o], a.h[grp]);
                a.p[grp][e] = st_p[o]; a.m[grp][e] = st_m[o]; a.v[grp][e] = st_v[o];
            }
        }
}

// ---- the wave's SH-rest gradient block, kept FACTORED in LDS ---------------------------------------------------------
// The gradient of sh_coefficients_rest[g][k][c] is basis_k(dir_g) * dL/dcolour_c(g) (sh_utils.cuh:90-111): 18 numbers per Gaussian
// (3 colour gradients + 15 basis values) instead of 45 products: 4.5 KB of LDS per wave instead of 11.25 KB, and the product is formed when
// a 16-byte piece of the [N, R, 3] block is assembled -- the same single multiplication, so the result is bit-identical. A float4 piece
// starting at element e0 of the wave's block covers exactly the (Gaussian, basis) pairs e0/3 and e0/3 + 1. Used by the fused kernel, where
// the slice is also the staging area of phase A (one 14 x 64 array at a time); the unfused K12 keeps the block of products (measured: the
// piece assembly costs it 0.003 ms and it gains nothing from the LDS, profiles/archive/r03_ab_fused_factored.txt).
constexpr uint32_t kShFactorStride = 18;                 // floats per Gaussian: colour gradient 3, basis values 15
constexpr uint32_t kShFactorFloats = kWave * kShFactorStride;

__device__ __forceinline__ void put_sh_factors(float* const slice, const uint32_t lane, const bool visible, const float (&dir)[3],
                                               const float (&gcol)[3], const int active_sh_bases) {
    float B[15];
#pragma unroll
    for (int k = 0; k < 15; ++k) B[k] = 0.0f;                         // degrees above the active one keep a zero gradient
    if (visible && active_sh_bases > 1) sh_basis(dir[0], dir[1], dir[2], active_sh_bases, B);
    float* const mine = slice + lane * kShFactorStride;
    mine[0] = gcol[0]; mine[1] = gcol[1]; mine[2] = gcol[2];           // zeros for an invisible Gaussian
#pragma unroll
    for (int k = 0; k < 15; ++k) mine[3 + k] = B[k];
}

template <int RT>
__device__ __forceinline__ float sh_rest_gradient_at(const float* const slice, const uint32_t e, const uint32_t R) {
    const uint32_t pair = e / 3u, c = e - 3u * pair;
    const uint32_t g = RT > 0 ? pair / static_cast<uint32_t>(RT) : pair / R, k = pair - g * R;
    const float* const f = slice + g * kShFactorStride;
    return f[3u + k] * f[c];
}

template <int RT>
__device__ __forceinline__ float4 sh_rest_gradient_piece(const float* const slice, const uint32_t e0, const uint32_t R) {
    const uint32_t p0 = e0 / 3u, r0 = e0 - 3u * p0;                    // first pair of the piece, channel its first element starts at
    const uint32_t g0 = RT > 0 ? p0 / static_cast<uint32_t>(RT) : p0 / R, k0 = p0 - g0 * R;
    const bool wrap = k0 + 1u == R;                                    // the second pair belongs to the next Gaussian
    const float* const f0 = slice + g0 * kShFactorStride;
    const float* const f1 = wrap ? f0 + kShFactorStride : f0;
    const float b0 = f0[3u + k0], b1 = f1[wrap ? 3u : 4u + k0];
    const float v0 = b0 * f0[0], v1 = b0 * f0[1], v2 = b0 * f0[2], v3 = b1 * f1[0], v4 = b1 * f1[1], v5 = b1 * f1[2];
    float4 g;                                                          // elements r0 .. r0 + 3 of (v0 .. v5)
    g.x = r0 == 0u ? v0 : r0 == 1u ? v1 : v2;
    g.y = r0 == 0u ? v1 : r0 == 1u ? v2 : v3;
    g.z = r0 == 0u ? v2 : r0 == 1u ? v3 : v4;
    g.w = r0 == 0u ? v3 : r0 == 1u ? v4 : v5;
    return g;
}

// ---- single-GPU fused backward + Adam (BASELINE.json configs[3]): ONE kernel for all 59 floats of a Gaussian -----------
// A wave owns 64 consecutive Gaussians. Phase A, one lane per Gaussian: the 14 small floats and their moments are requested,
// the gradient of the Gaussian is formed (gaussian_backward, which also gathers the lane's 45 SH-rest coefficients for the
// view-direction term -- the only read of that tensor from HBM), Adam is applied to the 14 floats, and the 45 SH-rest
// gradient floats basis_k(dir) * dL/dcolour (sh_utils.cuh:90-111) go to the wave's private LDS slice, which then holds the
// gradient of the wave's CONTIGUOUS 64 x R x 3 block of the [N, R, 3] tensors. Phase B streams that block of parameter /
// exp_avg / exp_avg_sq as non-temporal 16-byte accesses (the parameter block was just gathered: L2 hits). Compared with the
// two-kernel form of round 1 the SH-rest parameters are read from HBM once instead of twice, the view direction never
// leaves registers, and the 15 basis values are evaluated once per Gaussian instead of once per (Gaussian, basis) pair.
// Bytes per Gaussian: 59 x 24 (state in / out) + 36 + 4 (accumulators, tile count) [+ 16 densification] = 1456.
#ifndef FGS_FUSED_UNROLL
#define FGS_FUSED_UNROLL 3
#endif
constexpr int kFusedUnroll = FGS_FUSED_UNROLL;          // 16-byte pieces of each of the three streams in flight per lane
#ifndef FGS_FUSED_WAVES
#define FGS_FUSED_WAVES 3
#endif
template <int RT>
__global__ void __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(FGS_FUSED_WAVES, FGS_FUSED_WAVES)))
fused_backward_adam_kernel(const PreprocessBackwardArgs a, const ShRestArgs sh) {
    __shared__ __attribute__((aligned(16))) float s_grad[256 / kWave][kShFactorFloats];   // >= the 14 x 64 floats phase A stages per array
    const uint32_t R = RT > 0 ? static_cast<uint32_t>(RT) : sh.total_sh_rest;
    const uint32_t lane = lane_id(), wv = threadIdx.x >> 6;
    const uint32_t first = blockIdx.x * 256u + wv * kWave;            // first Gaussian of this wave
    if (first >= a.n) return;                                         // wave-uniform
    const uint32_t i = first + lane;
    const bool in_range = i < a.n;
    float* const slice = s_grad[wv];                                  // the wave's LDS slice: phase A staging now, the SH-rest gradient block later
    const uint32_t n_here = a.n - first < kWave ? a.n - first : kWave;   // Gaussians of this wave
    const bool whole = a.vector_ok != 0 && n_here == kWave;           // 16-byte accesses need 16-byte aligned tensors (checked at launch)

    // ---- phase A: the 14 small floats of parameters and both moments. A wave's 64 x w floats of a group are contiguous: they come in (and
    // go out) as ONE coalesced 16-byte access per lane and pass through LDS, instead of w scalar accesses per lane at a stride of 4 w bytes --
    // 42 loads + 42 stores per lane before, more memory instructions than phase B issues for three times the data. ----
    // The parameters come first (the gradient needs them); the moments are requested after the gradient is formed, so 28 registers are not
    // held across gaussian_backward, and every array passes through the same 3.5 KB of the slice. Round 3, one box: 0.850 -> 0.812 ms
    // (profiles/archive/r03_ab_fused_factored.txt; 3 or 4 waves per SIMD measure the same, 5 spills; requesting phase B's first pieces before the
    // gradient, or double-buffering phase B, measured slower at every depth tried).
    constexpr int kLanes[5] = {48, 48, 16, 48, 64};                   // 64 w / 4 float4 pieces
    float st_p[14], st_m[14], st_v[14];
    const uint32_t ic = in_range ? i : a.n - 1u;                      // partial wave: out-of-range lanes shadow the last Gaussian (loads only)
    if (whole) {
        float4 in[5];
#pragma unroll
        for (int grp = 0; grp < 5; ++grp)
            if (lane < static_cast<uint32_t>(kLanes[grp])) in[grp] = load_float4_nt(a.p[grp] + (size_t)first * kGroupWidth[grp] + 4u * lane);
#pragma unroll
        for (int grp = 0; grp < 5; ++grp)
            if (lane < static_cast<uint32_t>(kLanes[grp])) *reinterpret_cast<float4*>(slice + kGroupOffset[grp] * kWave + 4u * lane) = in[grp];
        wave_lds_fence();
#pragma unroll
        for (int grp = 0; grp < 5; ++grp)
#pragma unroll
            for (int k = 0; k < kGroupWidth[grp]; ++k) st_p[kGroupOffset[grp] + k] = slice[kGroupOffset[grp] * kWave + lane * kGroupWidth[grp] + k];
        wave_lds_fence();                                             // the moments overwrite the slice
    } else {
#pragma unroll
        for (int grp = 0; grp < 5; ++grp)
#pragma unroll
            for (int k = 0; k < kGroupWidth[grp]; ++k) st_p[kGroupOffset[grp] + k] = a.p[grp][(size_t)ic * kGroupWidth[grp] + k];
    }
    float grad[14], dir[3] = {0.0f, 0.0f, 0.0f}, gcol[3] = {0.0f, 0.0f, 0.0f};
    bool visible = false;
#pragma unroll
    for (int k = 0; k < 14; ++k) grad[k] = 0.0f;
    if (in_range) visible = gaussian_backward<true, false, true>(a, i, st_p, grad, dir, gcol);
    if (whole) {
        float4 in[2][5];
#pragma unroll
        for (int arr = 0; arr < 2; ++arr)
#pragma unroll
            for (int grp = 0; grp < 5; ++grp)
                if (lane < static_cast<uint32_t>(kLanes[grp]))
                    in[arr][grp] = load_float4_nt((arr == 0 ? a.m[grp] : a.v[grp]) + (size_t)first * kGroupWidth[grp] + 4u * lane);
#pragma unroll
        for (int arr = 0; arr < 2; ++arr) {
#pragma unroll
            for (int grp = 0; grp < 5; ++grp)
                if (lane < static_cast<uint32_t>(kLanes[grp])) *reinterpret_cast<float4*>(slice + kGroupOffset[grp] * kWave + 4u * lane) = in[arr][grp];
            wave_lds_fence();
#pragma unroll
            for (int grp = 0; grp < 5; ++grp)
#pragma unroll
                for (int k = 0; k < kGroupWidth[grp]; ++k) {
                    const float x = slice[kGroupOffset[grp] * kWave + lane * kGroupWidth[grp] + k];
                    if (arr == 0) st_m[kGroupOffset[grp] + k] = x; else st_v[kGroupOffset[grp] + k] = x;
                }
            wave_lds_fence();
        }
    } else {
#pragma unroll
        for (int grp = 0; grp < 5; ++grp)
#pragma unroll
            for (int k = 0; k < kGroupWidth[grp]; ++k) {
                const size_t e = (size_t)ic * kGroupWidth[grp] + k;
                st_m[kGroupOffset[grp] + k] = a.m[grp][e]; st_v[kGroupOffset[grp] + k] = a.v[grp][e];
            }
    }
    if (in_range) {
#pragma unroll
        for (int grp = 0; grp < 5; ++grp)
#pragma unroll
            for (int k = 0; k < kGroupWidth[grp]; ++k) {
                const int o = kGroupOffset[grp] + k;
                adam_update(st_p[o], st_m[o], st_v[o], grad[o], a.h[grp]);
                if (!whole) {
                    const size_t e = (size_t)i * kGroupWidth[grp] + k;
                    a.p[grp][e] = st_p[o]; a.m[grp][e] = st_m[o]; a.v[grp][e] = st_v[o];
                }
            }
    }
    if (whole) {
#pragma unroll
        for (int arr = 0; arr < 3; ++arr) {
#pragma unroll
            for (int grp = 0; grp < 5; ++grp)
#pragma unroll
                for (int k = 0; k < kGroupWidth[grp]; ++k) {
                    const int o = kGroupOffset[grp] + k;
                    slice[kGroupOffset[grp] * kWave + lane * kGroupWidth[grp] + k] = arr == 0 ? st_p[o] : arr == 1 ? st_m[o] : st_v[o];
                }
            wave_lds_fence();
#pragma unroll
            for (int grp = 0; grp < 5; ++grp) {
                float* const dst = (arr == 0 ? a.p[grp] : arr == 1 ? a.m[grp] : a.v[grp]) + (size_t)first * kGroupWidth[grp];
                if (lane < static_cast<uint32_t>(kLanes[grp]))
                    store_float4_nt(dst + 4u * lane, *reinterpret_cast<const float4*>(slice + kGroupOffset[grp] * kWave + 4u * lane));
            }
            wave_lds_fence();                                         // the slice is rewritten by the next array, then by the SH-rest factors
        }
    }
    if (R == 0) return;

    // ---- the wave's SH-rest gradient factors -> LDS (skipped when no lane of the wave is visible: the block is zero) ----
    const bool any_visible = wave_ballot(visible) != 0;
    if (any_visible) {
        put_sh_factors(slice, lane, visible, dir, gcol, sh.active_sh_bases);
        wave_lds_fence();
    }

    // ---- phase B: Adam over the wave's contiguous block, 3 streams x kFusedUnroll pieces in flight per lane ----
    const uint32_t count = (a.n - first < kWave ? a.n - first : kWave) * R * 3u;      // floats this wave owns
    const size_t base = (size_t)first * R * 3u;                                        // 64 * R * 12 bytes per wave: 16-byte aligned
    float* const P = sh.p + base; float* const M = sh.m + base; float* const V = sh.v + base;
    for (uint32_t e0 = 4u * lane; e0 < count; e0 += 4u * kWave * kFusedUnroll) {
        float4 p4[kFusedUnroll], m4[kFusedUnroll], v4[kFusedUnroll];
        bool full[kFusedUnroll];
#pragma unroll
        for (int u = 0; u < kFusedUnroll; ++u) {
            const uint32_t e = e0 + 4u * kWave * static_cast<uint32_t>(u);
            full[u] = e + 4u <= count && a.vector_ok != 0;                             // 16-byte pieces need 16-byte aligned tensors (checked at launch)
            if (full[u]) { p4[u] = load_float4_nt(P + e); m4[u] = load_float4_nt(M + e); v4[u] = load_float4_nt(V + e); }
        }
#pragma unroll
        for (int u = 0; u < kFusedUnroll; ++u) {
            const uint32_t e = e0 + 4u * kWave * static_cast<uint32_t>(u);
            if (full[u]) {
                const float4 g = any_visible ? sh_rest_gradient_piece<RT>(slice, e, R) : make_float4(0.0f, 0.0f, 0.0f, 0.0f);
                adam_update(p4[u].x, m4[u].x, v4[u].x, g.x, sh.h); adam_update(p4[u].y, m4[u].y, v4[u].y, g.y, sh.h);
                adam_update(p4[u].z, m4[u].z, v4[u].z, g.z, sh.h); adam_update(p4[u].w, m4[u].w, v4[u].w, g.w, sh.h);
                store_float4_nt(P + e, p4[u]); store_float4_nt(M + e, m4[u]); store_float4_nt(V + e, v4[u]);
            } else if (e < count) {                                                    // ragged tail of the last wave (< 4 floats), or unaligned tensors
                for (uint32_t j = e; j < count && j < e + 4u; ++j) {
                    float pp = P[j], mm = M[j], vv = V[j];
                    adam_update(pp, mm, vv, any_visible ? sh_rest_gradient_at<RT>(slice, j, R) : 0.0f, sh.h);
                    P[j] = pp; M[j] = mm; V[j] = vv;
                }
            }
        }
    }
}

// Unfused single-view K12 as ONE kernel (the form fgs_backward uses): the same wave layout as fused_backward_adam_kernel, with the
// gradients written instead of applied. Against round 1's two kernels it drops the view-direction round trip through HBM and the second
// read of tile counts / colour gradients, and keeps the fully coalesced 16-byte stores of the [N, R, 3] gradient (every element of every
// gradient is written exactly once, zeros for invisible Gaussians: no zero-fill, rasterization_api.cu:127-134).
#ifndef FGS_K12_NT_STORES
#define FGS_K12_NT_STORES 1      // round 6: the 540 MB SH-rest gradient leaves as non-temporal stores (K12 0.248 -> 0.242 ms and the Adam kernel behind it
                                 // 0.771 -> 0.758 in alternating processes, profiles/r06_ab_k12_nt_stores.txt); 2: the 14 small floats as well (A/B)
#endif
template <int RT>
__global__ void __launch_bounds__(256) backward_gradients_kernel(const PreprocessBackwardArgs a, const ShRestArgs sh) {
    __shared__ __attribute__((aligned(16))) float s_grad[256 / kWave][kWave * 15 * 3];
    const uint32_t R = RT > 0 ? static_cast<uint32_t>(RT) : sh.total_sh_rest;
    const uint32_t lane = lane_id(), wv = threadIdx.x >> 6;
    const uint32_t first = blockIdx.x * 256u + wv * kWave;
    if (first >= a.n) return;                                         // wave-uniform
    const uint32_t i = first + lane;
    const bool in_range = i < a.n;
    float grad[14], dir[3] = {0.0f, 0.0f, 0.0f}, gcol[3] = {0.0f, 0.0f, 0.0f};
    const float unused[14] = {};
    bool visible = false;
    if (in_range) {
        visible = gaussian_backward<false, false, true>(a, i, unused, grad, dir, gcol);
        // scalar stores at a stride of 4 w bytes: the write path combines them. Staging the wave's 64 x w block in LDS and storing it as one
        // coalesced 16-byte access per lane -- what pays in the fused kernel, where the same floats are also LOADED three times -- measured
        // 0.261 vs 0.252 ms here (profiles/archive/r02_ab_k12_coalesced_stores.txt)
        float* const outs[5] = {a.grad_means, a.grad_sh0, a.grad_opacities, a.grad_scales, a.grad_rotations};
#pragma unroll
        for (int grp = 0; grp < 5; ++grp)
#pragma unroll
            for (int k = 0; k < kGroupWidth[grp]; ++k) {
#if FGS_K12_NT_STORES >= 2
                __builtin_nontemporal_store(grad[kGroupOffset[grp] + k], outs[grp] + (size_t)i * kGroupWidth[grp] + k);
#else
                outs[grp][(size_t)i * kGroupWidth[grp] + k] = grad[kGroupOffset[grp] + k];
#endif
            }
    }
    float* const slice = s_grad[wv];
    const bool any_visible = wave_ballot(visible) != 0;
    if (a.live_blocks != nullptr && lane == 0) a.live_blocks[first >> 6] = any_visible ? 1 : 0;   // first is a multiple of 64
    if (R == 0) return;
    if (any_visible) {
        float B[15];
#pragma unroll
        for (int k = 0; k < 15; ++k) B[k] = 0.0f;
        if (visible && sh.active_sh_bases > 1) sh_basis(dir[0], dir[1], dir[2], sh.active_sh_bases, B);
        float* const mine = slice + lane * R * 3u;
#pragma unroll
        for (int k = 0; k < 15; ++k)
            if (static_cast<uint32_t>(k) < R) { mine[3 * k] = B[k] * gcol[0]; mine[3 * k + 1] = B[k] * gcol[1]; mine[3 * k + 2] = B[k] * gcol[2]; }
        wave_lds_fence();
    }
    const uint32_t count = (a.n - first < kWave ? a.n - first : kWave) * R * 3u;
    float* const out = sh.grad_sh_rest + (size_t)first * R * 3u;
    for (uint32_t e = 4u * lane; e < count; e += 4u * kWave) {
        if (e + 4u <= count && a.vector_ok) {                          // 16-byte stores need a 16-byte aligned gradient tensor (checked at launch)
            const float4 g = any_visible ? *reinterpret_cast<const float4*>(slice + e) : make_float4(0.0f, 0.0f, 0.0f, 0.0f);
#if FGS_K12_NT_STORES
            store_float4_nt(out + e, g);
#else
            *reinterpret_cast<float4*>(out + e) = g;
#endif
        } else {
            for (uint32_t j = e; j < count && j < e + 4u; ++j) out[j] = any_visible ? slice[j] : 0.0f;
        }
    }
}

static bool aligned16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15u) == 0; }

hipError_t launch_backward_gradients(const PreprocessBackwardArgs& a_in, const ShRestArgs& sh, hipStream_t s) {
    if (a_in.n == 0) return hipSuccess;
    if (sh.total_sh_rest > 15) return hipErrorInvalidValue;
    PreprocessBackwardArgs a = a_in;
    a.vector_ok = aligned16(sh.grad_sh_rest) ? 1 : 0;                  // a wave's block starts 64 * R * 12 bytes into the tensor: aligned iff the tensor is
    const dim3 grid((a.n + 255u) / 256u), block(256);
    if (sh.total_sh_rest == 15) hipLaunchKernelGGL(backward_gradients_kernel<15>, grid, block, 0, s, a, sh);
    else hipLaunchKernelGGL(backward_gradients_kernel<0>, grid, block, 0, s, a, sh);
    return hipGetLastError();
}

hipError_t launch_fused_backward_adam(const PreprocessBackwardArgs& a_in, const ShRestArgs& sh, hipStream_t s) {
    if (a_in.n == 0) return hipSuccess;
    if (sh.total_sh_rest > 15) return hipErrorInvalidValue;
    PreprocessBackwardArgs a = a_in;
    a.vector_ok = 1;
    for (int g = 0; g < 5; ++g) a.vector_ok = a.vector_ok && aligned16(a.p[g]) && aligned16(a.m[g]) && aligned16(a.v[g]);
    a.vector_ok = a.vector_ok && aligned16(sh.p) && aligned16(sh.m) && aligned16(sh.v);          // phase B: a wave's block starts 64 * R * 12 bytes in
    const dim3 grid((a.n + 255u) / 256u), block(256);
    if (sh.total_sh_rest == 15) hipLaunchKernelGGL(fused_backward_adam_kernel<15>, grid, block, 0, s, a, sh);
    else hipLaunchKernelGGL(fused_backward_adam_kernel<0>, grid, block, 0, s, a, sh);
    return hipGetLastError();
}

hipError_t launch_preprocess_backward(bool fused_adam, const PreprocessBackwardArgs& a, hipStream_t s) {
    if (a.n == 0) return hipSuccess;
    const dim3 grid((a.n + kPreprocessBackwardBlock - 1) / kPreprocessBackwardBlock), block(kPreprocessBackwardBlock);
    if (fused_adam && a.view[0].slot != nullptr) hipLaunchKernelGGL((preprocess_backward_kernel<true, true>), grid, block, 0, s, a);
    else if (fused_adam) hipLaunchKernelGGL((preprocess_backward_kernel<true, false>), grid, block, 0, s, a);
    else if (a.view[0].slot == nullptr) hipLaunchKernelGGL((preprocess_backward_kernel<false, false>), grid, block, 0, s, a);
    else hipLaunchKernelGGL((preprocess_backward_kernel<false, true>), grid, block, 0, s, a);
    return hipGetLastError();
}

// ---- SH-rest pass: the [N, R, 3] tensors are streamed FLAT in 16-byte pieces (grid-stride, a few thousand workgroups) ----
// A float4 piece starting at element e0 covers exactly the (Gaussian, basis) pairs p0 = e0/3 and p0+1; for each of the two
// the lane rebuilds basis_k(view_dir) * dL/dcolour (sh_utils.cuh:90-111) from 28 bytes that 15 neighbouring lanes share.
struct PairGrad { float b; float c[3]; };

template <int RT, bool RECORDS>
__device__ __forceinline__ PairGrad pair_gradient(const ShRestArgs& a, const ShRestView& V, const uint32_t pair_in, const uint32_t n_pairs) {
    // All seven loads are issued unconditionally (one memory round trip instead of a dependent chain); the result is
    // SELECTED to zero for invisible Gaussians / inactive degrees because their view_dir slot is uninitialised scratch.
    const uint32_t pair = pair_in < n_pairs ? pair_in : n_pairs - 1u;
    const uint32_t R = RT > 0 ? static_cast<uint32_t>(RT) : a.total_sh_rest;
    const uint32_t gi = pair / R, k = pair - gi * R;
    const uint32_t touched = V.n_touched[gi];
    const float x = V.view_dir[3 * (size_t)gi], y = V.view_dir[3 * (size_t)gi + 1], z = V.view_dir[3 * (size_t)gi + 2];
    float c0 = 0.0f, c1 = 0.0f, c2 = 0.0f;
    if (!RECORDS) {                      // K11's own records [N][9]: always in bounds, requested together with everything else
        const float* const accp = V.acc + (size_t)gi * kAccRecordWords;
        c0 = accp[6]; c1 = accp[7]; c2 = accp[8];
    } else if (touched != 0) {           // sharded path: a record exists only for visible primitives (slot[] of the others is scratch)
        const float* const accp = V.acc + (size_t)V.slot[gi] * kAccRecordWords;
        c0 = accp[6]; c1 = accp[7]; c2 = accp[8];
    }
    // coefficient k belongs to degree 1 (k<3), 2 (k<8), 3 (k<15); it receives a gradient only if that degree is active
    const bool degree_on = (k < 3 && a.active_sh_bases > 1) || (k >= 3 && k < 8 && a.active_sh_bases > 4) ||
                           (k >= 8 && k < 15 && a.active_sh_bases > 9);
    const bool on = pair_in < n_pairs && degree_on && touched != 0;
    float B[15];
#pragma unroll
    for (int j = 0; j < 15; ++j) B[j] = 0.0f;
    sh_basis(x, y, z, a.active_sh_bases, B);
    float bk = 0.0f;
#pragma unroll
    for (int j = 0; j < 15; ++j) bk = (k == static_cast<uint32_t>(j)) ? B[j] : bk;   // select, no dynamic register indexing
    PairGrad r;
    r.b = on ? bk : 0.0f;
    r.c[0] = on ? c0 : 0.0f; r.c[1] = on ? c1 : 0.0f; r.c[2] = on ? c2 : 0.0f;
    return r;
}

// Unfused form (the gradient tensor is materialised). One lane per GAUSSIAN evaluates the 15 basis values once and forms its
// 45 gradient floats; the wave's 64 x 45 block is contiguous in the [N, R, 3] tensor, so it is transposed through the wave's
// private LDS slice and leaves as fully coalesced 16-byte stores. (v1: one lane per (Gaussian, basis) pair evaluated all 15
// basis functions to keep one -- 33 VALU instructions per output float, half VALU-bound at 0.184 ms for a 540 MB write.)
// MULTI (sharded path): the gradient is summed over the views of the launch, colour gradients come from the accumulator records.
// lane `lane` of a wave whose first Gaussian is `first`: the 3 * R gradient floats of Gaussian first + lane, summed over the
// views of the launch, written to the wave's LDS slice at [lane * R * 3 ...] (the slice then holds the wave's contiguous
// block of the [N, R, 3] tensor).
template <int RT, bool MULTI>
__device__ __forceinline__ void sh_rest_block_to_lds(const ShRestArgs& a, const uint32_t first, const uint32_t lane, float* slice) {
    constexpr int kMaxRest = 15;
    const uint32_t R = RT > 0 ? static_cast<uint32_t>(RT) : a.total_sh_rest;
    const uint32_t gi = first + lane;
    const bool in_range = gi < a.n;
    float g[kMaxRest][3];
#pragma unroll
    for (int k = 0; k < kMaxRest; ++k) { g[k][0] = 0.0f; g[k][1] = 0.0f; g[k][2] = 0.0f; }
    const int n_views = MULTI ? a.n_views : 1;
    for (int vw = 0; vw < n_views; ++vw) {
        const ShRestView& V = a.view[MULTI ? vw : 0];
        if (!in_range || V.n_touched[gi] == 0) continue;
        const float x = V.view_dir[3 * (size_t)gi], y = V.view_dir[3 * (size_t)gi + 1], z = V.view_dir[3 * (size_t)gi + 2];
        float c[3];
        const float* const r = V.acc + (size_t)(MULTI ? V.slot[gi] : gi) * kAccRecordWords;       // K11's record (sharded path: through the slot table)
        c[0] = r[6]; c[1] = r[7]; c[2] = r[8];
        float B[kMaxRest];
#pragma unroll
        for (int k = 0; k < kMaxRest; ++k) B[k] = 0.0f;                 // degrees above the active one keep a zero gradient
        if (a.active_sh_bases > 1) sh_basis(x, y, z, a.active_sh_bases, B);
#pragma unroll
        for (int k = 0; k < kMaxRest; ++k) {
            if (!MULTI) { g[k][0] = B[k] * c[0]; g[k][1] = B[k] * c[1]; g[k][2] = B[k] * c[2]; }
            else { g[k][0] += B[k] * c[0]; g[k][1] += B[k] * c[1]; g[k][2] += B[k] * c[2]; }
        }
    }
    float* mine = slice + lane * R * 3u;
#pragma unroll
    for (int k = 0; k < kMaxRest; ++k)
        if (static_cast<uint32_t>(k) < R) { mine[3 * k] = g[k][0]; mine[3 * k + 1] = g[k][1]; mine[3 * k + 2] = g[k][2]; }
    wave_lds_fence();
}

template <int RT, bool MULTI>
__global__ void __launch_bounds__(256) sh_rest_gradient_kernel(const ShRestArgs a) {
    __shared__ __attribute__((aligned(16))) float s_out[256 / kWave][kWave * 15 * 3];
    const uint32_t R = RT > 0 ? static_cast<uint32_t>(RT) : a.total_sh_rest;
    const uint32_t lane = lane_id(), wv = threadIdx.x >> 6;
    const uint32_t first = (blockIdx.x * 256u + wv * kWave);         // first Gaussian of this wave
    if (first >= a.n) return;                                         // wave-uniform
    sh_rest_block_to_lds<RT, MULTI>(a, first, lane, s_out[wv]);
    const uint32_t count = (a.n - first < kWave ? a.n - first : kWave) * R * 3u;      // floats this wave owns
    float* out = a.grad_sh_rest + (size_t)first * R * 3u;                             // 64 * R * 12 bytes per wave: 16-byte aligned
    const bool add = MULTI && a.accumulate;                                            // view batches after the first
    for (uint32_t e = 4u * lane; e < count; e += 4u * kWave) {
        if (e + 4u <= count) {
            float4 v = *reinterpret_cast<const float4*>(&s_out[wv][e]);
            if (add) { const float4 o = *reinterpret_cast<const float4*>(out + e); v.x += o.x; v.y += o.y; v.z += o.z; v.w += o.w; }
            *reinterpret_cast<float4*>(out + e) = v;
        } else {
            for (uint32_t j = e; j < count; ++j) out[j] = add ? out[j] + s_out[wv][j] : s_out[wv][j];
        }
    }
}

// Sharded path, fused with Adam: the wave's gradient block (summed over the views) stays in LDS and feeds the update of the
// wave's contiguous 64 x R x 3 slice of parameter / exp_avg / exp_avg_sq, streamed as non-temporal 16-byte accesses.
template <int RT>
__global__ void __launch_bounds__(256) sh_rest_adam_views_kernel(const ShRestArgs a) {
    __shared__ __attribute__((aligned(16))) float s_out[256 / kWave][kWave * 15 * 3];
    const uint32_t R = RT > 0 ? static_cast<uint32_t>(RT) : a.total_sh_rest;
    const uint32_t lane = lane_id(), wv = threadIdx.x >> 6;
    const uint32_t first = (blockIdx.x * 256u + wv * kWave);
    if (first >= a.n) return;
    sh_rest_block_to_lds<RT, true>(a, first, lane, s_out[wv]);
    const uint32_t count = (a.n - first < kWave ? a.n - first : kWave) * R * 3u;
    const size_t base = (size_t)first * R * 3u;
    for (uint32_t e = 4u * lane; e < count; e += 4u * kWave) {
        if (e + 4u <= count) {
            float4 p4 = load_float4_nt(a.p + base + e), m4 = load_float4_nt(a.m + base + e), v4 = load_float4_nt(a.v + base + e);
            const float4 g = *reinterpret_cast<const float4*>(&s_out[wv][e]);
            adam_update(p4.x, m4.x, v4.x, g.x, a.h); adam_update(p4.y, m4.y, v4.y, g.y, a.h);
            adam_update(p4.z, m4.z, v4.z, g.z, a.h); adam_update(p4.w, m4.w, v4.w, g.w, a.h);
            store_float4_nt(a.p + base + e, p4); store_float4_nt(a.m + base + e, m4); store_float4_nt(a.v + base + e, v4);
        } else {
            for (uint32_t j = e; j < count; ++j) {
                float pp = a.p[base + j], mm = a.m[base + j], vv = a.v[base + j];
                adam_update(pp, mm, vv, s_out[wv][j], a.h);
                a.p[base + j] = pp; a.m[base + j] = mm; a.v[base + j] = vv;
            }
        }
    }
}

template <bool FUSED, int RT>
__global__ void __launch_bounds__(256) sh_rest_backward_kernel(const ShRestArgs a) {
    const uint32_t R = RT > 0 ? static_cast<uint32_t>(RT) : a.total_sh_rest;
    const uint32_t n_pairs = a.n * R;
    const uint32_t n_elems = n_pairs * 3u;                 // host guarantees < 2^32
    const uint32_t n_vec = (n_elems + 3u) / 4u;
    for (uint32_t v = blockIdx.x * 256u + threadIdx.x; v < n_vec; v += gridDim.x * 256u) {
        const uint32_t e0 = 4u * v;
        const uint32_t p0 = e0 / 3u, c0 = e0 - 3u * p0;
        const bool full = e0 + 4u <= n_elems;
        float4 p4 = make_float4(0.0f, 0.0f, 0.0f, 0.0f), m4 = p4, v4 = p4;
        if (FUSED && full) {                     // request the 48 bytes of state before the gradient is rebuilt
            p4 = load_float4_nt(a.p + e0);                 // streamed once per step: non-temporal like the Adam kernel
            m4 = load_float4_nt(a.m + e0);
            v4 = load_float4_nt(a.v + e0);
        }
        const PairGrad q0 = pair_gradient<RT, false>(a, a.view[0], p0, n_pairs), q1 = pair_gradient<RT, false>(a, a.view[0], p0 + 1u, n_pairs);
        float g[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const uint32_t cj = c0 + static_cast<uint32_t>(j);               // 0..5: first pair while < 3
            const bool in_first = cj < 3u;                                   // scalar selects only: a pointer/reference select
            const uint32_t c = in_first ? cj : cj - 3u;                      // between q0 and q1 sends both structs to scratch
            const float qb = in_first ? q0.b : q1.b;
            const float qc0 = in_first ? q0.c[0] : q1.c[0], qc1 = in_first ? q0.c[1] : q1.c[1], qc2 = in_first ? q0.c[2] : q1.c[2];
            g[j] = qb * (c == 0u ? qc0 : (c == 1u ? qc1 : qc2));
        }
        if (full) {
            if (!FUSED) {
                *reinterpret_cast<float4*>(a.grad_sh_rest + e0) = make_float4(g[0], g[1], g[2], g[3]);
            } else {
                adam_update(p4.x, m4.x, v4.x, g[0], a.h); adam_update(p4.y, m4.y, v4.y, g[1], a.h);
                adam_update(p4.z, m4.z, v4.z, g[2], a.h); adam_update(p4.w, m4.w, v4.w, g[3], a.h);
                store_float4_nt(a.p + e0, p4);
                store_float4_nt(a.m + e0, m4);
                store_float4_nt(a.v + e0, v4);
            }
        } else {
            for (uint32_t j = 0; e0 + j < n_elems; ++j) {
                if (!FUSED) a.grad_sh_rest[e0 + j] = g[j];
                else {
                    float pp = a.p[e0 + j], mm = a.m[e0 + j], vv = a.v[e0 + j];
                    adam_update(pp, mm, vv, g[j], a.h);
                    a.p[e0 + j] = pp; a.m[e0 + j] = mm; a.v[e0 + j] = vv;
                }
            }
        }
    }
}

hipError_t launch_sh_rest_backward(bool fused_adam, const ShRestArgs& a, hipStream_t s) {
    const uint64_t n_elems = (uint64_t)a.n * a.total_sh_rest * 3u;
    if (n_elems == 0) return hipSuccess;
    if (n_elems >= (1ull << 32)) return hipErrorInvalidValue;        // 32-bit element indices: N * (K-1) * 3 < 2^32
    const uint64_t n_vec = (n_elems + 3) / 4;
    const unsigned blocks = static_cast<unsigned>(n_vec / 256 + 1 < 8192 ? n_vec / 256 + 1 : 8192);   // grid-stride: 32 workgroups per CU
    const dim3 grid(blocks), block(256);
    if (!fused_adam) {
        if (a.total_sh_rest > 15) return hipErrorInvalidValue;
        const dim3 ggrid((a.n + 255u) / 256u);
        const bool views = a.view[0].slot != nullptr;           // sharded path: records + sum over views
        if (a.total_sh_rest == 15) {
            if (views) hipLaunchKernelGGL((sh_rest_gradient_kernel<15, true>), ggrid, block, 0, s, a);
            else hipLaunchKernelGGL((sh_rest_gradient_kernel<15, false>), ggrid, block, 0, s, a);
        } else {
            if (views) hipLaunchKernelGGL((sh_rest_gradient_kernel<0, true>), ggrid, block, 0, s, a);
            else hipLaunchKernelGGL((sh_rest_gradient_kernel<0, false>), ggrid, block, 0, s, a);
        }
    } else if (a.view[0].slot != nullptr) {                      // sharded path fused with Adam
        if (a.total_sh_rest > 15) return hipErrorInvalidValue;
        const dim3 ggrid((a.n + 255u) / 256u);
        if (a.total_sh_rest == 15) hipLaunchKernelGGL(sh_rest_adam_views_kernel<15>, ggrid, block, 0, s, a);
        else hipLaunchKernelGGL(sh_rest_adam_views_kernel<0>, ggrid, block, 0, s, a);
    } else if (a.total_sh_rest == 15) {
        hipLaunchKernelGGL((sh_rest_backward_kernel<true, 15>), grid, block, 0, s, a);
    } else {
        hipLaunchKernelGGL((sh_rest_backward_kernel<true, 0>), grid, block, 0, s, a);
    }
    return hipGetLastError();
}

// ---- K13: Adam for all parameter groups in one launch (adam.cu:10-34), float4-vectorised, U pieces per thread ----------
// g_adam_unroll = 1: 16-byte pieces per thread (fgs_debug_set_option(1, u)); measured on MI355X: 1, 2 and 4 are within 2 %
// g_adam_reverse = 1 (fgs_debug_set_option(8, 0|1) in the dev build): reversed workgroup order (measured 0.837 vs 0.855 ms at S2, tools/ab_adam_order.py)
// g_adam_nontemporal = 1 (fgs_debug_set_option(2, 0|1) in the dev build): non-temporal loads / stores (state is streamed once per step: +2.3 % measured)
// (round 6, measured and withdrawn: the updated PARAMETERS alone as ordinary stores, on the idea that the next forward pass reads them first -- K1 0.227 vs 0.201 ms,
// Adam 0.797 vs 0.782: dirty lines in eight L2s are the last thing the next kernel's reads want to meet, profiles/r06_ab_adam_param_nt.txt)

template <bool NT> __device__ __forceinline__ float4 load4(const float* p) { return NT ? load_float4_nt(p) : *reinterpret_cast<const float4*>(p); }
template <bool NT> __device__ __forceinline__ void store4(float* p, const float4 v) {
    if (NT) store_float4_nt(p, v);
    else *reinterpret_cast<float4*>(p) = v;
}

template <int U, bool NT>
__global__ void __launch_bounds__(256) adam_kernel(const AdamArgs a) {
    // a.reverse: workgroups walk the arenas from the end -- the gradient elements the backward pass wrote LAST are the ones most likely to
    // still sit in the 256 MB memory-side cache
    const uint32_t blk = a.reverse ? a.total_blocks - 1u - blockIdx.x : blockIdx.x;
    int gidx = 0;
#pragma unroll
    for (int j = 1; j < 8; ++j) if (j < a.n_groups && blk >= a.g[j].first_block) gidx = j;
    const AdamGroup& G = a.g[gidx];
    const int64_t block_base = (int64_t)(blk - G.first_block) * (256 * 4 * U);
    float4 g4[U], p4[U], m4[U], v4[U];
    bool full[U];
#pragma unroll
    for (int u = 0; u < U; ++u) {                       // all loads of the thread are issued before any arithmetic
        const int64_t base = block_base + ((int64_t)u * 256 + threadIdx.x) * 4;
        full[u] = base + 4 <= G.n;
        if (full[u]) {
            // the float4 covers the rows base / L .. (base + 3) / L of the [N, L] gradient; they lie in at most two consecutive blocks of 64
            // Gaussians. Both dead: every element is a zero the backward pass wrote -- skip the read (the kernel is HBM-bound; the index
            // arithmetic is free)
            bool need = true;
            if (a.live_blocks != nullptr && G.row_len != 0u) {
                const uint32_t r0 = static_cast<uint32_t>(base) / G.row_len, r1 = static_cast<uint32_t>(base + 3) / G.row_len;
                const uint32_t b0 = r0 >> 6, b1 = r1 >> 6;
                need = (a.live_blocks[b0] | a.live_blocks[b1]) != 0;
                if (!need) {
                    // Belt and braces: the promise rests on the caller's proof that nobody touched the gradients since the backward pass,
                    // and a write that bypasses the framework's bookkeeping (`.grad.data.add_(...)`, a raw-pointer kernel) cannot be seen by
                    // it. One SENTINEL float per dead block and tensor (the first element of the block: the same cached 4 bytes for every
                    // float4 of the block) is read anyway; anything but +-0 there -- a whole-tensor edit such as hand-written weight decay,
                    // NaN / Inf -- and the block's gradients are read after all.
                    const float s0 = G.grad[(size_t)b0 * 64u * G.row_len], s1 = G.grad[(size_t)b1 * 64u * G.row_len];
                    need = !(s0 == 0.0f) || !(s1 == 0.0f);
                }
            }
            g4[u] = need ? load4<NT>(G.grad + base) : make_float4(0.0f, 0.0f, 0.0f, 0.0f);
            p4[u] = load4<NT>(G.param + base);
            m4[u] = load4<NT>(G.exp_avg + base);
            v4[u] = load4<NT>(G.exp_avg_sq + base);
        }
    }
#pragma unroll
    for (int u = 0; u < U; ++u) {
        const int64_t base = block_base + ((int64_t)u * 256 + threadIdx.x) * 4;
        if (full[u]) {
            adam_update(p4[u].x, m4[u].x, v4[u].x, g4[u].x, G.h); adam_update(p4[u].y, m4[u].y, v4[u].y, g4[u].y, G.h);
            adam_update(p4[u].z, m4[u].z, v4[u].z, g4[u].z, G.h); adam_update(p4[u].w, m4[u].w, v4[u].w, g4[u].w, G.h);
            store4<NT>(G.param + base, p4[u]);
            store4<NT>(G.exp_avg + base, m4[u]);
            store4<NT>(G.exp_avg_sq + base, v4[u]);
        } else {
            for (int64_t e = base; e < G.n && e < base + 4; ++e) {
                float pp = G.param[e], mm = G.exp_avg[e], vv = G.exp_avg_sq[e];
                adam_update(pp, mm, vv, G.grad[e], G.h);
                G.param[e] = pp; G.exp_avg[e] = mm; G.exp_avg_sq[e] = vv;
            }
        }
    }
}

hipError_t launch_adam(const AdamArgs& a_in, hipStream_t s) {
    AdamArgs a = a_in;
    const int unroll_opt = g_adam_unroll, nontemporal = g_adam_nontemporal;
    const int u = nontemporal ? 1 : (unroll_opt == 2 || unroll_opt == 4 ? unroll_opt : 1);
    uint32_t blocks = 0;                                  // first_block / total_blocks depend on the elements per workgroup
    for (int k = 0; k < a.n_groups; ++k) { a.g[k].first_block = blocks; blocks += static_cast<uint32_t>((a.g[k].n + 1024 * u - 1) / (1024 * u)); }
    a.total_blocks = blocks;
    a.reverse = g_adam_reverse;
    if (blocks == 0) return hipSuccess;
    if (nontemporal) hipLaunchKernelGGL((adam_kernel<1, true>), dim3(blocks), dim3(256), 0, s, a);
    else if (u == 1) hipLaunchKernelGGL((adam_kernel<1, false>), dim3(blocks), dim3(256), 0, s, a);
    else if (u == 2) hipLaunchKernelGGL((adam_kernel<2, false>), dim3(blocks), dim3(256), 0, s, a);
    else hipLaunchKernelGGL((adam_kernel<4, false>), dim3(blocks), dim3(256), 0, s, a);
    return hipGetLastError();
}

}  // namespace fgs
