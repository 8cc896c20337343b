/*
 * fgs_oracle.c -- CPU restatement of the FasterGS rasterizer hot path (TEST INFRASTRUCTURE ONLY).
 *
 * PARITY UNPINNED: the reference ships no tests, golden vectors or fixtures for this path
 *  and its CUDA implementation cannot be compiled or run in this environment
 * (no nvcc, no NVIDIA device). This file restates the reference arithmetic from its sources; it is
 * pinned only by (a) an independent fp64 torch.autograd compositor (oracle/torch_check.py) and finite
 * differences of it, (b) hand-derived closed-form cases (one / two Gaussians on the optical axis: image
 * and analytic gradients) and the orthonormality of the SH basis, (c) structural invariants, (d) an fp64 render
 * BY DEFINITION (torch_check.py: brute_force_forward -- every Gaussian at every pixel under the per-pair alpha / transmittance
 * rules alone) for the discrete half: bounds, tile test, sub-tile test, sorts, lists, culls
 * (all in tests/test_oracle.py). None of these is an output of the reference itself. See DESIGN.md "Oracle".
 * (Round 6: the reference's Python glue -- torch_bindings/*.py -- IS executed in the build container and pinned by committed fixtures,
 * tests/golden/ref_glue_*; that pins the operator surface above the kernels, not the arithmetic restated in this file, which stays unpinned.)
 *
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load this library.
 * The product path (faster-gaussian-splatting_amd/) never links, imports or calls it.
 *
 * Reference path shorthand used in the citations below (all under /root/reference/):
 *   cfg = FasterGSCudaBackend/FasterGSCudaBackend/rasterization/include/rasterization_config.h
 *   ku  = .../rasterization/include/kernel_utils.cuh
 *   sh  = .../rasterization/include/sh_utils.cuh
 *   kf  = .../rasterization/include/kernels_forward.cuh
 *   ki  = .../rasterization/include/kernels_inference.cuh
 *   kb  = .../rasterization/include/kernels_backward.cuh
 *   fwd = .../rasterization/src/forward.cu      bwd = .../rasterization/src/backward.cu
 *   bu  = .../rasterization/include/buffer_utils.h
 *   adam= FasterGSCudaBackend/FasterGSCudaBackend/adam/src/adam.cu
 *
 * Arithmetic policy: fp32, no FMA contraction (build with -ffp-contract=off), expressions kept in the
 * reference's association order. rsqrtf is restated as 1/sqrtf, __saturatef as a NaN->0 clamp.
 * Nondeterministic pieces of the reference (atomic compaction order kf:204-208, float atomicAdd order
 * kb:460-469) are made deterministic: compaction by ascending primitive index, accumulation in bucket order.
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#include <limits.h>

#ifdef _OPENMP
#include <omp.h>
#endif

typedef unsigned int uint;

/*
 * ORC_F64 (libfgs_oracle64.so, built from this same file): every `float` below becomes `double` -- the same formulas in the same order
 * with the reference's fp32 CONSTANTS (literals keep their f suffix: 0.3f, 1.0f / 255.0f, the SH constants are the float values)
 * evaluated in double precision. It is the tests' estimate of the TRUE value of every float the fp32 pipelines produce, used to
 * separate implementation error from the rounding noise any fp32 evaluation of an ill-conditioned sum carries (tests/helpers.py:
 * three-way element-wise check; measured: the fp32 restatement itself misses an element-wise 1e-4 bar against this build on
 * 1-4 % of the gradient entries of a deep scene). Only the continuous half is meaningful in this build: the discrete structure
 * (visible set, bounds, instance lists, ranges) is taken from the fp32 run (orc_preprocess_follow), bit-punning helpers
 * (depth keys) are not called.
 */
#ifdef ORC_F64
#define float double
#define expf exp
#define logf log
#define sqrtf sqrt
#define floorf floor
#define ceilf ceil
#define fmaxf fmax
#define fminf fmin
#define fabsf fabs
#define fmaf fma
#define copysignf copysign
#endif

/* ---- cfg:8-60 --------------------------------------------------------------------------------------- */
#define DILATION 0.3f
#define DILATION_PROPER_AA 0.1f
#define MIN_COV2D_DET 1e-6f
#define ONE_MINUS_ALPHA_EPS 1e-6f
#define TRANSMITTANCE_THRESHOLD 1e-4f
#define MIN_ALPHA_THRESHOLD_RCP 255.0f
#define MIN_ALPHA_THRESHOLD (1.0f / 255.0f)
#define N_SEQUENTIAL_THRESHOLD 4
#define TILE_W 16
#define TILE_H 12
#define BLOCK_BLEND (TILE_W * TILE_H)
#define SUBTILE_W 8
#define SUBTILE_H 4

/* settings block shared with the python wrapper (ctypes.Structure of the same layout) */
typedef struct {
    float w2c[12];      /* rows 0..2 of the world-to-camera matrix, row major (kf:21, 65, 99-102) */
    float cam_pos[3];
    float bg[3];
    int active_sh_bases;
    int total_sh_rest;  /* sh_coefficients_rest.size(1), api:44 */
    int width, height;
    float fx, fy, cx, cy, near_plane, far_plane;
    int proper_aa;
} orc_settings;

int orc_num_threads(void) {
#ifdef _OPENMP
    return omp_get_max_threads();
#else
    return 1;
#endif
}

/* bu:10-18 */
int orc_extract_end_bit(uint n) {
    int leading_zeros = 0;
    if ((n & 0xffff0000u) == 0) { leading_zeros += 16; n <<= 16; }
    if ((n & 0xff000000u) == 0) { leading_zeros += 8; n <<= 8; }
    if ((n & 0xf0000000u) == 0) { leading_zeros += 4; n <<= 4; }
    if ((n & 0xc0000000u) == 0) { leading_zeros += 2; n <<= 2; }
    if ((n & 0x80000000u) == 0) { leading_zeros += 1; }
    return 32 - leading_zeros;
}

static inline float satf(float x) { /* __saturatef: NaN -> 0 */
    if (x != x) return 0.0f;
    return x < 0.0f ? 0.0f : (x > 1.0f ? 1.0f : x);
}
static inline float clampf(float f, float a, float b) { return fmaxf(a, fminf(f, b)); } /* helper_math.h:1250 */
static inline float lerpf(float a, float b, float t) { return a + t * (b - a); }          /* helper_math.h:1227 */
static inline float sigmoidf(float x) { return 1.0f / (1.0f + expf(-x)); }                /* ku:11-13 */
static inline int float2int_rd(float x) {
    if (x != x) return 0;
    float f = floorf(x);
    if (f <= -2147483648.0f) return INT_MIN;
    if (f >= 2147483648.0f) return INT_MAX;
    return (int)f;
}
static inline int float2int_ru(float x) {
    if (x != x) return 0;
    float f = ceilf(x);
    if (f <= -2147483648.0f) return INT_MIN;
    if (f >= 2147483648.0f) return INT_MAX;
    return (int)f;
}
static inline int imin(int a, int b) { return a < b ? a : b; }
static inline int imax(int a, int b) { return a > b ? a : b; }
static inline uint div_round_up_u(uint a, uint b) { return (a + b - 1) / b; }

/* ku:15-30 : rotation matrix from an unnormalised quaternion (r,x,y,z), 1/|q|^2 folded in */
static inline void quat_to_rot(const float q[4], float R[9], float* norm_sq) {
    const float r = q[0], x = q[1], y = q[2], z = q[3];
    const float xx = x * x, yy = y * y, zz = z * z;
    const float xy = x * y, xz = x * z, yz = y * z;
    const float rx = r * x, ry = r * y, rz = r * z;
    *norm_sq = r * r + xx + yy + zz;
    const float n = 1.0f / *norm_sq;
    R[0] = 1.0f - 2.0f * (yy + zz) * n; R[1] = 2.0f * (xy - rz) * n;        R[2] = 2.0f * (xz + ry) * n;
    R[3] = 2.0f * (xy + rz) * n;        R[4] = 1.0f - 2.0f * (xx + zz) * n; R[5] = 2.0f * (yz - rx) * n;
    R[6] = 2.0f * (xz - ry) * n;        R[7] = 2.0f * (yz + rx) * n;        R[8] = 1.0f - 2.0f * (xx + yy) * n;
}

/* ku:32-59 */
static inline void quat_to_rot_backward(const float q[4], const float dR[9], float out[4]) {
    const float r = q[0], x = q[1], y = q[2], z = q[3];
    const float xx = x * x, yy = y * y, zz = z * z;
    const float xy = x * y, xz = x * z, yz = y * z;
    const float rx = r * x, ry = r * y, rz = r * z;
    const float norm_sq = r * r + xx + yy + zz;
    const float norm_sq_rcp = 1.0f / norm_sq;
    const float m11 = dR[0], m12 = dR[1], m13 = dR[2], m21 = dR[3], m22 = dR[4], m23 = dR[5], m31 = dR[6], m32 = dR[7], m33 = dR[8];
    const float dxx = m22 + m33, dyy = m11 + m33, dzz = m11 + m22;
    const float drz = m21 - m12, dxy = m21 + m12;
    const float dry = m13 - m31, dxz = m13 + m31;
    const float drx = m32 - m23, dyz = m32 + m23;
    const float two = 2.0f * norm_sq_rcp;
    const float h = two * (xy * dxy + xz * dxz + yz * dyz + rx * drx + ry * dry + rz * drz - xx * dxx - yy * dyy - zz * dzz);
    out[0] = two * (x * drx + y * dry + z * drz - r * h);
    out[1] = two * (r * drx - 2.0f * x * dxx + y * dxy + z * dxz - x * h);
    out[2] = two * (r * dry + x * dxy - 2.0f * y * dyy + z * dyz - y * h);
    out[3] = two * (r * drz + x * dxz + y * dyz - 2.0f * z * dzz - z * h);
}

/* ku:72-114 : exact tile/Gaussian overlap test (StopThePop); mean is already shifted by -0.5 */
static inline int will_primitive_contribute(float mx, float my, float ca, float cb, float cc,
                                            uint tile_x, uint tile_y, float power_threshold) {
    const float rect_min_x = (float)(tile_x * TILE_W), rect_min_y = (float)(tile_y * TILE_H);
    const float rect_max_x = (float)((tile_x + 1) * TILE_W - 1), rect_max_y = (float)((tile_y + 1) * TILE_H - 1);
    const float x_min_diff = rect_min_x - mx;
    const float x_left = (float)(x_min_diff >= 0.0f);
    const float not_in_x_range = x_left + (float)(mx > rect_max_x);
    const float y_min_diff = rect_min_y - my;
    const float y_above = (float)(y_min_diff >= 0.0f);
    const float not_in_y_range = y_above + (float)(my > rect_max_y);
    if (not_in_y_range + not_in_x_range == 0.0f) return 1;
    const float corner_x = lerpf(rect_max_x, rect_min_x, x_left);
    const float corner_y = lerpf(rect_max_y, rect_min_y, y_above);
    const float diff_x = mx - corner_x, diff_y = my - corner_y;
    const float dx = copysignf((float)(TILE_W - 1), x_min_diff);
    const float dy = copysignf((float)(TILE_H - 1), y_min_diff);
    const float tx = not_in_y_range * satf((dx * ca * diff_x + dx * cb * diff_y) / (dx * ca * dx));
    const float ty = not_in_x_range * satf((dy * cb * diff_x + dy * cc * diff_y) / (dy * cc * dy));
    const float px = corner_x + tx * dx, py = corner_y + ty * dy;
    const float ex = mx - px, ey = my - py;
    const float max_power = 0.5f * (ca * ex * ex + cc * ey * ey) + cb * ex * ey;
    return max_power <= power_threshold;
}

/* ku:61-68 */
static inline void tile_bounds_of(const uint16_t sb[4], uint tb[4]) {
    tb[0] = sb[0] / TILE_W;
    tb[1] = div_round_up_u(sb[1], TILE_W);
    tb[2] = sb[2] / TILE_H;
    tb[3] = div_round_up_u(sb[3], TILE_H);
}

/* sh:4-30 */
static const float C0 = 0.28209479177387814f, C1 = 0.48860251190291987f;
static const float C2a = 1.0925484305920792f, C2b = 0.94617469575755997f, C2c = 0.31539156525251999f,
                   C2d = 0.54627421529603959f, C2e = 1.8923493915151202f;
static const float C3a = 0.59004358992664352f, C3b = 1.7701307697799304f, C3c = 2.8906114426405538f,
                   C3d = 0.45704579946446572f, C3e = 2.2852289973223288f, C3f = 1.865881662950577f,
                   C3g = 1.1195289977703462f, C3h = 1.4453057213202769f, C3i = 3.5402615395598609f,
                   C3j = 4.5704579946446566f, C3k = 5.597644988851731f;

/* sh:32-69 */
static void sh_to_color(const float* sh0, const float* sh_rest, const float pos[3], const float cam[3],
                        uint idx, uint active, uint total_rest, float out[3]) {
    for (int c = 0; c < 3; c++) out[c] = 0.5f + C0 * sh0[3 * idx + c];
    if (active > 1) {
        const float* k = sh_rest + (size_t)idx * total_rest * 3;
        const float vx = pos[0] - cam[0], vy = pos[1] - cam[1], vz = pos[2] - cam[2];
        const float inv = 1.0f / sqrtf(vx * vx + vy * vy + vz * vz); /* normalize = v * rsqrtf(dot) */
        const float x = vx * inv, y = vy * inv, z = vz * inv;
        for (int c = 0; c < 3; c++)
            out[c] = out[c] - C1 * y * k[0 + c] + C1 * z * k[3 + c] - C1 * x * k[6 + c];
        if (active > 4) {
            const float xx = x * x, yy = y * y, zz = z * z, xy = x * y, xz = x * z, yz = y * z;
            for (int c = 0; c < 3; c++)
                out[c] = out[c] + C2a * xy * k[9 + c] - C2a * yz * k[12 + c] + (C2b * zz - C2c) * k[15 + c]
                         - C2a * xz * k[18 + c] + C2d * (xx - yy) * k[21 + c];
            if (active > 9) {
                for (int c = 0; c < 3; c++)
                    out[c] = out[c] + y * (C3a * yy - C3b * xx) * k[24 + c] + C3c * xy * z * k[27 + c]
                             + y * (C3d - C3e * zz) * k[30 + c] + z * (C3f * zz - C3g) * k[33 + c]
                             + x * (C3d - C3e * zz) * k[36 + c] + C3h * z * (xx - yy) * k[39 + c]
                             + x * (C3b * yy - C3a * xx) * k[42 + c];
            }
        }
    }
}

/* the part of kf:61-160 / kb:57-114 shared by forward and backward preprocessing */
typedef struct {
    float depth, x, y;
    float var[3], R[9], RSS[9], cov3d[6];
    float norm_sq;
    float clip_l, clip_r, clip_t, clip_b, x_clipped, y_clipped;
    float j11, j13, j22, j23;
    float jw1[3], jw2[3], jwc1[3], jwc2[3];
    float a_raw, b, c_raw;
} proj_t;

static void project(const float m[3], const float s[3], const float q[4], const orc_settings* S, proj_t* P) {
    const float* r1 = S->w2c; const float* r2 = S->w2c + 4; const float* r3 = S->w2c + 8;
    P->depth = r3[0] * m[0] + r3[1] * m[1] + r3[2] * m[2] + r3[3];
    for (int i = 0; i < 3; i++) P->var[i] = expf(2.0f * s[i]);
    quat_to_rot(q, P->R, &P->norm_sq);
    const float* R = P->R;
    for (int r = 0; r < 3; r++) for (int c = 0; c < 3; c++) P->RSS[3 * r + c] = R[3 * r + c] * P->var[c];
    const float* G = P->RSS;
    P->cov3d[0] = G[0] * R[0] + G[1] * R[1] + G[2] * R[2];
    P->cov3d[1] = G[0] * R[3] + G[1] * R[4] + G[2] * R[5];
    P->cov3d[2] = G[0] * R[6] + G[1] * R[7] + G[2] * R[8];
    P->cov3d[3] = G[3] * R[3] + G[4] * R[4] + G[5] * R[5];
    P->cov3d[4] = G[3] * R[6] + G[4] * R[7] + G[5] * R[8];
    P->cov3d[5] = G[6] * R[6] + G[7] * R[7] + G[8] * R[8];
    P->x = (r1[0] * m[0] + r1[1] * m[1] + r1[2] * m[2] + r1[3]) / P->depth;
    P->y = (r2[0] * m[0] + r2[1] * m[1] + r2[2] * m[2] + r2[3]) / P->depth;
    const float width = (float)S->width, height = (float)S->height;
    P->clip_l = (-0.15f * width - S->cx) / S->fx;
    P->clip_r = (1.15f * width - S->cx) / S->fx;
    P->clip_t = (-0.15f * height - S->cy) / S->fy;
    P->clip_b = (1.15f * height - S->cy) / S->fy;
    P->x_clipped = clampf(P->x, P->clip_l, P->clip_r);
    P->y_clipped = clampf(P->y, P->clip_t, P->clip_b);
    P->j11 = S->fx / P->depth; P->j13 = -P->j11 * P->x_clipped;
    P->j22 = S->fy / P->depth; P->j23 = -P->j22 * P->y_clipped;
    for (int i = 0; i < 3; i++) {
        P->jw1[i] = P->j11 * r1[i] + P->j13 * r3[i];
        P->jw2[i] = P->j22 * r2[i] + P->j23 * r3[i];
    }
    const float* C = P->cov3d; /* m11 m12 m13 m22 m23 m33 */
    P->jwc1[0] = P->jw1[0] * C[0] + P->jw1[1] * C[1] + P->jw1[2] * C[2];
    P->jwc1[1] = P->jw1[0] * C[1] + P->jw1[1] * C[3] + P->jw1[2] * C[4];
    P->jwc1[2] = P->jw1[0] * C[2] + P->jw1[1] * C[4] + P->jw1[2] * C[5];
    P->jwc2[0] = P->jw2[0] * C[0] + P->jw2[1] * C[1] + P->jw2[2] * C[2];
    P->jwc2[1] = P->jw2[0] * C[1] + P->jw2[1] * C[3] + P->jw2[2] * C[4];
    P->jwc2[2] = P->jw2[0] * C[2] + P->jw2[1] * C[4] + P->jw2[2] * C[5];
    P->a_raw = P->jwc1[0] * P->jw1[0] + P->jwc1[1] * P->jw1[1] + P->jwc1[2] * P->jw1[2];
    P->b     = P->jwc1[0] * P->jw2[0] + P->jwc1[1] * P->jw2[1] + P->jwc1[2] * P->jw2[2];
    P->c_raw = P->jwc2[0] * P->jw2[0] + P->jwc2[1] * P->jw2[1] + P->jwc2[2] * P->jw2[2];
}

/*
 * K1  kf:14-209 (training) / ki:14-207 (inference: colour clamped at store ki:200, n_touched not zeroed ki:59).
 * Outputs are indexed by primitive; n_touched[i] == 0 marks an invisible primitive. The compacted
 * (depth_key, primitive_idx) list is written in ascending primitive order (deterministic restatement of the
 * atomicAdd compaction kf:204-208). Returns n_visible; *n_instances_out = sum of n_touched (kf:208).
 */
int orc_preprocess(int N, const float* means, const float* scales, const float* rotations, const float* opacities,
                   const float* sh0, const float* sh_rest, const orc_settings* S, int inference,
                   uint* n_touched, uint16_t* screen_bounds, float* mean2d, float* conic_opacity, float* color,
                   uint* depth_keys, uint* prim_indices, uint* n_instances_out) {
    const uint grid_w = div_round_up_u((uint)S->width, TILE_W), grid_h = div_round_up_u((uint)S->height, TILE_H);
#pragma omp parallel for schedule(static)
    for (int i = 0; i < N; i++) {
        n_touched[i] = 0; /* kf:59 */
        const float* m = means + 3 * (size_t)i;
        const float* r3 = S->w2c + 8;
        const float depth0 = r3[0] * m[0] + r3[1] * m[1] + r3[2] * m[2] + r3[3];
        if (depth0 < S->near_plane || depth0 > S->far_plane) continue;           /* kf:67 */
        float opacity = sigmoidf(opacities[i]);
        if (opacity < MIN_ALPHA_THRESHOLD) continue;                              /* kf:75 */
        proj_t P;
        project(m, scales + 3 * (size_t)i, rotations + 4 * (size_t)i, S, &P);
        if (P.norm_sq < 1e-8f) continue;                                          /* kf:83 */
        float cov_x = P.a_raw, cov_y = P.b, cov_z = P.c_raw;
        const float det_raw = cov_x * cov_z - cov_y * cov_y;
        const float ks = S->proper_aa ? DILATION_PROPER_AA : DILATION;
        cov_x += ks; cov_z += ks;
        const float det = cov_x * cov_z - cov_y * cov_y;
        if (det < MIN_COV2D_DET) continue;                                        /* kf:144 */
        const float ca = cov_z / det, cb = -cov_y / det, cc = cov_x / det;
        if (S->proper_aa) {
            opacity *= sqrtf(fmaxf(det_raw / det, 0.0f));
            if (opacity < MIN_ALPHA_THRESHOLD) continue;                          /* kf:153 */
        }
        const float m2x = P.x * S->fx + S->cx, m2y = P.y * S->fy + S->cy;        /* kf:157-160 */
        const float power_threshold = logf(opacity * MIN_ALPHA_THRESHOLD_RCP);    /* kf:163 */
        const float cutoff = 2.0f * power_threshold;
        const float ext_x = fmaxf(sqrtf(cov_x * cutoff) - 0.5f, 0.0f);
        const float ext_y = fmaxf(sqrtf(cov_z * cutoff) - 0.5f, 0.0f);
        const int padded_w = (int)(grid_w * TILE_W), padded_h = (int)(grid_h * TILE_H);
        uint16_t sb[4];
        sb[0] = (uint16_t)imin(padded_w, imax(0, float2int_rd(m2x - ext_x)));
        sb[1] = (uint16_t)imin(padded_w, imax(0, float2int_ru(m2x + ext_x)));
        sb[2] = (uint16_t)imin(padded_h, imax(0, float2int_rd(m2y - ext_y)));
        sb[3] = (uint16_t)imin(padded_h, imax(0, float2int_ru(m2y + ext_y)));
        uint tb[4];
        tile_bounds_of(sb, tb);
        const uint tbw = tb[1] - tb[0];
        const uint n_max = tbw * (tb[3] - tb[2]);
        if (n_max == 0) continue;                                                 /* kf:178 */
        /* ku:117-180: the sequential/cooperative split only changes who does the work; the count is the same */
        uint cnt = 0;
        const float sx = m2x - 0.5f, sy = m2y - 0.5f;
        for (uint t = 0; t < n_max; t++)
            cnt += (uint)will_primitive_contribute(sx, sy, ca, cb, cc, tb[0] + t % tbw, tb[2] + t / tbw, power_threshold);
        if (cnt == 0) continue;                                                   /* kf:190 */
        n_touched[i] = cnt;
        memcpy(screen_bounds + 4 * (size_t)i, sb, sizeof(sb));
        mean2d[2 * (size_t)i] = m2x; mean2d[2 * (size_t)i + 1] = m2y;
        conic_opacity[4 * (size_t)i] = ca; conic_opacity[4 * (size_t)i + 1] = cb;
        conic_opacity[4 * (size_t)i + 2] = cc; conic_opacity[4 * (size_t)i + 3] = opacity;
        float col[3];
        sh_to_color(sh0, sh_rest, m, S->cam_pos, (uint)i, (uint)S->active_sh_bases, (uint)S->total_sh_rest, col);
        for (int c = 0; c < 3; c++) color[3 * (size_t)i + c] = inference ? fmaxf(col[c], 0.0f) : col[c];
    }
    int V = 0; uint I = 0;
    for (int i = 0; i < N; i++) {
        if (n_touched[i] == 0) continue;
        const float* m = means + 3 * (size_t)i; const float* r3 = S->w2c + 8;
        const float depth = r3[0] * m[0] + r3[1] * m[1] + r3[2] * m[2] + r3[3];
        uint key; memcpy(&key, &depth, 4);                                        /* __float_as_uint, kf:205 */
        depth_keys[V] = key; prim_indices[V] = (uint)i; V++;
        I += n_touched[i];
    }
    *n_instances_out = I;
    return V;
}

/*
 * TEST SUPPORT (ORC_F64 build): the float outputs of K1 for exactly the primitives another run (the fp32 one) found visible --
 * no cull is re-decided here, so both precisions work on the same visible set, bounds and instance lists.
 */
void orc_preprocess_follow(int N, const float* means, const float* scales, const float* rotations, const float* opacities,
                           const float* sh0, const float* sh_rest, const orc_settings* S, const uint* n_touched,
                           float* mean2d, float* conic_opacity, float* color) {
#pragma omp parallel for schedule(static)
    for (int i = 0; i < N; i++) {
        if (n_touched[i] == 0) continue;
        const float* m = means + 3 * (size_t)i;
        float opacity = sigmoidf(opacities[i]);
        proj_t P;
        project(m, scales + 3 * (size_t)i, rotations + 4 * (size_t)i, S, &P);
        float cov_x = P.a_raw, cov_y = P.b, cov_z = P.c_raw;
        const float det_raw = cov_x * cov_z - cov_y * cov_y;
        const float ks = S->proper_aa ? DILATION_PROPER_AA : DILATION;
        cov_x += ks; cov_z += ks;
        const float det = cov_x * cov_z - cov_y * cov_y;
        if (S->proper_aa) opacity *= sqrtf(fmaxf(det_raw / det, 0.0f));
        mean2d[2 * (size_t)i] = P.x * S->fx + S->cx; mean2d[2 * (size_t)i + 1] = P.y * S->fy + S->cy;
        conic_opacity[4 * (size_t)i] = cov_z / det; conic_opacity[4 * (size_t)i + 1] = -cov_y / det;
        conic_opacity[4 * (size_t)i + 2] = cov_x / det; conic_opacity[4 * (size_t)i + 3] = opacity;
        float col[3];
        sh_to_color(sh0, sh_rest, m, S->cam_pos, (uint)i, (uint)S->active_sh_bases, (uint)S->total_sh_rest, col);
        for (int c = 0; c < 3; c++) color[3 * (size_t)i + c] = col[c];
    }
}

/* stable LSD radix sort of (key,value) pairs on bits [0,end_bit) -- semantics of cub::DeviceRadixSort::SortPairs
 * as used at fwd:104-110 (32 bits) and fwd:195-202 (end_bit bits of the tile key). */
void orc_sort_pairs(int n, uint* keys, uint* vals, int end_bit) {
    if (n <= 0) return;
    uint* k2 = (uint*)malloc(sizeof(uint) * (size_t)n);
    uint* v2 = (uint*)malloc(sizeof(uint) * (size_t)n);
    uint *ka = keys, *va = vals, *kb = k2, *vb = v2;
    for (int shift = 0; shift < end_bit; shift += 8) {
        const int bits = end_bit - shift < 8 ? end_bit - shift : 8;
        const uint mask = (1u << bits) - 1u;
        size_t hist[257]; memset(hist, 0, sizeof(hist));
        for (int i = 0; i < n; i++) hist[((ka[i] >> shift) & mask) + 1]++;
        for (int d = 0; d < 256; d++) hist[d + 1] += hist[d];
        for (int i = 0; i < n; i++) { const size_t p = hist[(ka[i] >> shift) & mask]++; kb[p] = ka[i]; vb[p] = va[i]; }
        uint* t = ka; ka = kb; kb = t; t = va; va = vb; vb = t;
    }
    if (ka != keys) { memcpy(keys, ka, sizeof(uint) * (size_t)n); memcpy(vals, va, sizeof(uint) * (size_t)n); }
    free(k2); free(v2);
}

/* K3+K4+K5  kf:211-221, fwd:121-127, kf:225-328. Instances of one primitive are emitted in row-major order over
 * its tile bounding box (the order both the sequential and the cooperative branch of the reference produce). */
void orc_create_instances(int V, const uint* sorted_prim, const uint* n_touched, const uint16_t* screen_bounds,
                          const float* mean2d, const float* conic_opacity, int grid_w,
                          uint* offsets, uint* inst_keys, uint* inst_prims) {
    uint acc = 0;
    for (int i = 0; i < V; i++) { offsets[i] = acc; acc += n_touched[sorted_prim[i]]; }
#pragma omp parallel for schedule(dynamic, 256)
    for (int i = 0; i < V; i++) {
        const uint p = sorted_prim[i];
        uint tb[4];
        tile_bounds_of(screen_bounds + 4 * (size_t)p, tb);
        const uint tbw = tb[1] - tb[0];
        const uint count = (tb[3] - tb[2]) * tbw;
        const float sx = mean2d[2 * (size_t)p] - 0.5f, sy = mean2d[2 * (size_t)p + 1] - 0.5f;
        const float ca = conic_opacity[4 * (size_t)p], cb = conic_opacity[4 * (size_t)p + 1], cc = conic_opacity[4 * (size_t)p + 2];
        const float pt = logf(conic_opacity[4 * (size_t)p + 3] * MIN_ALPHA_THRESHOLD_RCP);   /* kf:267 */
        uint w = offsets[i];
        for (uint t = 0; t < count; t++) {
            const uint tx = tb[0] + t % tbw, ty = tb[2] + t / tbw;
            if (will_primitive_contribute(sx, sy, ca, cb, cc, tx, ty, pt)) {
                inst_keys[w] = ty * (uint)grid_w + tx; inst_prims[w] = p; w++;
            }
        }
    }
}

/* K7 kf:331-348 (ranges pre-zeroed: fwd:54), K8 kf:350-360, K9 fwd:225-231. bucket_size is 32 in the reference. */
uint orc_ranges_and_buckets(int I, const uint* inst_keys, int T, int bucket_size,
                            uint* ranges /*[T][2]*/, uint* n_buckets /*[T]*/, uint* bucket_offsets /*[T] inclusive*/) {
    memset(ranges, 0, sizeof(uint) * 2 * (size_t)T);
    for (int i = 0; i < I; i++) {
        const uint t = inst_keys[i];
        if (i == 0) ranges[2 * t] = 0;
        else {
            const uint pt = inst_keys[i - 1];
            if (t != pt) { ranges[2 * pt + 1] = (uint)i; ranges[2 * t] = (uint)i; }
        }
        if (i == I - 1) ranges[2 * t + 1] = (uint)I;
    }
    uint acc = 0;
    for (int t = 0; t < T; t++) {
        n_buckets[t] = div_round_up_u(ranges[2 * t + 1] - ranges[2 * t], (uint)bucket_size);
        acc += n_buckets[t]; bucket_offsets[t] = acc;
    }
    return acc;
}

/*
 * K10 kf:362-498 (training) and ki:348-463 (inference: no checkpoints / T_final / n_processed; optional clamp+HWC).
 * One iteration of the outer loop = one tile; the inner restatement is per pixel, with the reference's 8x4
 * sub-tile bounding-box cull (kf:445-451) applied per pixel's own sub-tile. mode: 0 training, 1 inference.
 */
void orc_blend_forward(int mode, int to_chw, int clamp_output, int bucket_size,
                       const uint* ranges, const uint* bucket_offsets, const uint* inst_prims,
                       const uint16_t* screen_bounds, const float* mean2d, const float* conic_opacity, const float* color,
                       const orc_settings* S,
                       float* image, float* final_T /*[P] image-linear*/, uint* n_processed /*[P]*/, uint* max_n_processed /*[T]*/,
                       uint* bucket_tile_index, float* bucket_ckpt /*[B][192][4]*/) {
    const int W = S->width, H = S->height;
    const int grid_w = (W + TILE_W - 1) / TILE_W, grid_h = (H + TILE_H - 1) / TILE_H;
    const int T = grid_w * grid_h;
    const size_t n_pixels = (size_t)W * H;
#pragma omp parallel for schedule(dynamic, 4)
    for (int tile = 0; tile < T; tile++) {
        const int tyi = tile / grid_w, txi = tile % grid_w;
        const uint r0 = ranges[2 * tile], r1 = ranges[2 * tile + 1];
        const int n_total = (int)(r1 - r0);
        uint bucket_base = 0;
        if (mode == 0) {
            const int nb = (n_total + bucket_size - 1) / bucket_size;
            bucket_base = tile == 0 ? 0 : bucket_offsets[tile - 1];
            for (int b = 0; b < nb; b++) bucket_tile_index[bucket_base + b] = (uint)tile;        /* kf:407-411 */
        }
        uint tile_max = 0;
        for (int local = 0; local < BLOCK_BLEND; local++) {
            const int px = txi * TILE_W + local % TILE_W, py = tyi * TILE_H + local / TILE_W;
            if (px >= W || py >= H) continue;                                                      /* done = !inside */
            /* sub-tile of this pixel, kf:389-398 */
            const int sx0 = txi * TILE_W + ((local % TILE_W) / SUBTILE_W) * SUBTILE_W;
            const int sy0 = tyi * TILE_H + ((local / TILE_W) / SUBTILE_H) * SUBTILE_H;
            const int sx1 = sx0 + SUBTILE_W, sy1 = sy0 + SUBTILE_H;
            const float pxf = (float)px + 0.5f, pyf = (float)py + 0.5f;
            float cr = 0.0f, cg = 0.0f, cbl = 0.0f, Tr = 1.0f;
            uint n_used = 0; int done = 0;
            for (int j = 0; j < n_total && !done; j++) {
                if (mode == 0 && j % bucket_size == 0) {                                           /* kf:436-442 */
                    float* ck = bucket_ckpt + ((size_t)(bucket_base + j / bucket_size) * BLOCK_BLEND + local) * 4;
                    ck[0] = cr; ck[1] = cg; ck[2] = cbl; ck[3] = Tr;
                }
                const uint p = inst_prims[r0 + j];
                const uint16_t* sb = screen_bounds + 4 * (size_t)p;
                if (!(sb[0] < sx1 && sx0 < sb[1] && sb[2] < sy1 && sy0 < sb[3])) continue;         /* kf:447-449 */
                const float* co = conic_opacity + 4 * (size_t)p;
                const float dx = mean2d[2 * (size_t)p] - pxf, dy = mean2d[2 * (size_t)p + 1] - pyf;
                const float expo = -0.5f * (co[0] * dx * dx + co[2] * dy * dy) - co[1] * dx * dy;
                const float g = expf(fminf(expo, 0.0f));
                const float alpha = co[3] * g;
                if (alpha < MIN_ALPHA_THRESHOLD) continue;
                const float* col = color + 3 * (size_t)p;
                const float w = Tr * alpha;
                if (mode == 0) { cr += w * fmaxf(col[0], 0.0f); cg += w * fmaxf(col[1], 0.0f); cbl += w * fmaxf(col[2], 0.0f); } /* kf:430 */
                else { cr += w * col[0]; cg += w * col[1]; cbl += w * col[2]; }
                Tr *= 1.0f - alpha;
                n_used = (uint)j + 1;
                if (Tr < TRANSMITTANCE_THRESHOLD) done = 1;
            }
            cr += Tr * S->bg[0]; cg += Tr * S->bg[1]; cbl += Tr * S->bg[2];
            const size_t pix = (size_t)W * py + px;
            if (mode == 0) {
                image[pix] = cr; image[n_pixels + pix] = cg; image[2 * n_pixels + pix] = cbl;
                final_T[pix] = Tr; n_processed[pix] = n_used;
                if (n_used > tile_max) tile_max = n_used;
            } else {
                if (clamp_output) { cr = satf(cr); cg = satf(cg); cbl = satf(cbl); }
                if (to_chw) { image[pix] = cr; image[n_pixels + pix] = cg; image[2 * n_pixels + pix] = cbl; }
                else { image[3 * pix] = cr; image[3 * pix + 1] = cg; image[3 * pix + 2] = cbl; }
            }
        }
        if (mode == 0) max_n_processed[tile] = tile_max;
    }
}

/*
 * K11 kb:260-471. One bucket = bucket_size consecutive instances of one tile; per-"lane" accumulators sum over the
 * tile's pixels in local-index order (the order the lane pipeline visits them), then are added to the per-primitive
 * accumulators in bucket order (deterministic restatement of the atomicAdds kb:460-469).
 * grad_conic is planar [3][N] (api:134), grad_mean2d is [N][2], grad_sh0 receives dL/dcolor (read back by K12, sh:85).
 */
void orc_blend_backward(int N, int n_buckets_total, int bucket_size,
                        const uint* ranges, const uint* bucket_offsets, const uint* inst_prims,
                        const float* mean2d, const float* conic_opacity, const float* color,
                        const orc_settings* S, const float* grad_image, const float* image,
                        const float* final_T, const uint* max_n_processed, const uint* n_processed,
                        const uint* bucket_tile_index, const float* bucket_ckpt,
                        float* grad_mean2d, float* grad_conic, float* grad_opacity, float* grad_sh0) {
    const int W = S->width, H = S->height;
    const int grid_w = (W + TILE_W - 1) / TILE_W;
    const size_t n_pixels = (size_t)W * H;
    float* acc = (float*)calloc((size_t)n_buckets_total * bucket_size * 9, sizeof(float));
#pragma omp parallel for schedule(dynamic, 16)
    for (int b = 0; b < n_buckets_total; b++) {
        const uint tile = bucket_tile_index[b];
        const uint r0 = ranges[2 * tile], r1 = ranges[2 * tile + 1];
        const int tile_n = (int)(r1 - r0);
        const uint first = tile == 0 ? 0 : bucket_offsets[tile - 1];
        const int tb = b - (int)first;
        if ((uint)(tb * bucket_size) >= max_n_processed[tile]) continue;                            /* kb:295 */
        const int tx = (int)(tile % (uint)grid_w), ty = (int)(tile / (uint)grid_w);
        for (int local = 0; local < BLOCK_BLEND; local++) {
            const int px = tx * TILE_W + local % TILE_W, py = ty * TILE_H + local / TILE_W;
            if (px >= W || py >= H) continue;
            const size_t pix = (size_t)W * py + px;
            const float* ck = bucket_ckpt + ((size_t)b * BLOCK_BLEND + local) * 4;
            const float fT = final_T[pix];
            float gpx[3], after[3];
            for (int c = 0; c < 3; c++) {
                gpx[c] = grad_image[c * n_pixels + pix];
                after[c] = image[c * n_pixels + pix] - fT * S->bg[c] - ck[c];                      /* kb:371-373 */
            }
            float Tr = ck[3];
            const float galpha_common = fT * -(gpx[0] * S->bg[0] + gpx[1] * S->bg[1] + gpx[2] * S->bg[2]); /* kb:375-377 */
            const uint last = n_processed[pix];
            const float pxf = (float)px + 0.5f, pyf = (float)py + 0.5f;
            for (int l = 0; l < bucket_size; l++) {
                const int tp = tb * bucket_size + l;
                if (tp >= tile_n) break;
                if ((uint)tp >= last) break;                                                        /* kb:412 */
                const uint p = inst_prims[r0 + tp];
                const float* co = conic_opacity + 4 * (size_t)p;
                const float dx = mean2d[2 * (size_t)p] - pxf, dy = mean2d[2 * (size_t)p + 1] - pyf;
                const float expo = -0.5f * (co[0] * dx * dx + co[2] * dy * dy) - co[1] * dx * dy;
                const float g = expf(fminf(expo, 0.0f));
                const float alpha = co[3] * g;
                if (alpha < MIN_ALPHA_THRESHOLD) continue;
                const float* cu = color + 3 * (size_t)p;
                float* A = acc + ((size_t)b * bucket_size + l) * 9;
                const float w = Tr * alpha;
                float colc[3];
                for (int c = 0; c < 3; c++) {
                    colc[c] = fmaxf(cu[c], 0.0f);
                    A[6 + c] += w * gpx[c] * (cu[c] >= 0.0f ? 1.0f : 0.0f);                         /* kb:426-427 */
                    after[c] -= w * colc[c];                                                         /* kb:429 */
                }
                const float oma = 1.0f - alpha;
                const float oma_rcp = 1.0f / fmaxf(oma, ONE_MINUS_ALPHA_EPS);
                const float dLda_color = (Tr * colc[0] - after[0] * oma_rcp) * gpx[0] + (Tr * colc[1] - after[1] * oma_rcp) * gpx[1]
                                         + (Tr * colc[2] - after[2] * oma_rcp) * gpx[2];
                const float dLda = dLda_color + galpha_common * oma_rcp;
                A[5] += g * dLda;                                                                    /* kb:438-439 */
                const float h = -alpha * dLda;
                A[2] += 0.5f * h * (dx * dx); A[3] += 0.5f * h * (dx * dy); A[4] += 0.5f * h * (dy * dy); /* kb:443-448 */
                A[0] += h * (co[0] * dx + co[1] * dy); A[1] += h * (co[1] * dx + co[2] * dy);       /* kb:449-453 */
                Tr *= oma;
            }
        }
    }
    for (int b = 0; b < n_buckets_total; b++) {
        const uint tile = bucket_tile_index[b];
        const uint r0 = ranges[2 * tile], r1 = ranges[2 * tile + 1];
        const uint first = tile == 0 ? 0 : bucket_offsets[tile - 1];
        const int tb = b - (int)first;
        if ((uint)(tb * bucket_size) >= max_n_processed[tile]) continue;
        for (int l = 0; l < bucket_size; l++) {
            const int tp = tb * bucket_size + l;
            if (tp >= (int)(r1 - r0)) break;
            const uint p = inst_prims[r0 + tp];
            const float* A = acc + ((size_t)b * bucket_size + l) * 9;
            const float op = conic_opacity[4 * (size_t)p + 3];
            grad_mean2d[2 * (size_t)p] += A[0]; grad_mean2d[2 * (size_t)p + 1] += A[1];
            grad_conic[p] += A[2]; grad_conic[(size_t)N + p] += A[3]; grad_conic[2 * (size_t)N + p] += A[4];
            grad_opacity[p] += S->proper_aa ? A[5] : op * (1.0f - op) * A[5];                       /* kb:465 */
            grad_sh0[3 * (size_t)p] += A[6]; grad_sh0[3 * (size_t)p + 1] += A[7]; grad_sh0[3 * (size_t)p + 2] += A[8];
        }
    }
    free(acc);
}

/*
 * TEST SUPPORT (no reference counterpart): which outputs sit next to one of the blend's two hard thresholds?
 * The blend passes decide `alpha >= 1/255` (kf:467, kb:421) and `T < 1e-4` (kf:477) on floats; an implementation whose exp /
 * FMA contraction differs from this restatement by a few ULP can land on the other side for a (pixel, Gaussian) pair whose
 * value lies within `eps` (relative) of the threshold, which moves that pixel by up to ~1/255 and that Gaussian's gradient by
 * one pixel's contribution. Parity tests exclude exactly those entries (and count them) and hold everything else to 1e-4.
 *   risk_pixel[P] : the pixel has a pair (any Gaussian in front of its last contributor, kb:412) within eps of the alpha
 *                   threshold, or its transmittance passes within eps_T of the termination threshold
 *   risk_prim[N]  : the Gaussian is the partner in such an alpha pair
 *   near_prim[N]  : the Gaussian blends into a risky pixel (second-order: its gradient sees that pixel's changed T / colour)
 */
void orc_threshold_risk(const uint* ranges, const uint* inst_prims, const uint16_t* screen_bounds, const float* mean2d,
                        const float* conic_opacity, const orc_settings* S, const uint* n_processed, float eps, float eps_T,
                        uint8_t* risk_pixel, uint8_t* risk_prim, uint8_t* near_prim) {
    const int W = S->width, H = S->height;
    const int grid_w = (W + TILE_W - 1) / TILE_W, grid_h = (H + TILE_H - 1) / TILE_H;
    const int T = grid_w * grid_h;
#pragma omp parallel for schedule(dynamic, 4)
    for (int tile = 0; tile < T; tile++) {
        const int tyi = tile / grid_w, txi = tile % grid_w;
        const uint r0 = ranges[2 * tile];
        for (int local = 0; local < BLOCK_BLEND; local++) {
            const int px = txi * TILE_W + local % TILE_W, py = tyi * TILE_H + local / TILE_W;
            if (px >= W || py >= H) continue;
            const size_t pix = (size_t)W * py + px;
            const int sx0 = txi * TILE_W + ((local % TILE_W) / SUBTILE_W) * SUBTILE_W;
            const int sy0 = tyi * TILE_H + ((local / TILE_W) / SUBTILE_H) * SUBTILE_H;
            const int sx1 = sx0 + SUBTILE_W, sy1 = sy0 + SUBTILE_H;
            const float pxf = (float)px + 0.5f, pyf = (float)py + 0.5f;
            /* The walk covers what the blend kernel walks: the tile's list until the transmittance test ends it -- NOT only the entries in front of the
             * pixel's last contributor (n_processed): a pair that fails the alpha test by a hair BEHIND the last contributor is a flip like any other
             * (the device then blends one entry more than this restatement; found by a 4000-seed fuzz sweep in round 6, seed 3316). */
            const int n_total = (int)(ranges[2 * tile + 1] - r0);
            (void)n_processed;
            int n_scan = n_total;
            float Tr = 1.0f;
            int risky = 0;
            for (int pass = 0; pass < 2; pass++) {
                if (pass == 1 && !risky) break;
                for (int j = 0; j < n_scan; j++) {
                    const uint p = inst_prims[r0 + j];
                    const float* co = conic_opacity + 4 * (size_t)p;
                    const float dx = mean2d[2 * (size_t)p] - pxf, dy = mean2d[2 * (size_t)p + 1] - pyf;
                    const float expo = -0.5f * (co[0] * dx * dx + co[2] * dy * dy) - co[1] * dx * dy;
                    const float alpha = co[3] * expf(fminf(expo, 0.0f));
                    /* The exponent is a sum of three terms that may cancel: another association / an FMA (the HIP blend kernels are built with
                     * contraction) moves it by a few ulp OF THE TERMS, and alpha by that much relatively. The margin grows accordingly
                     * (found by a wide fuzz sweep, round 3: a pair 2e-5 from the cut with terms of 40 flipped outside the fixed margin). */
                    const float slack = eps + 4.0f * 1.1920929e-7f * (0.5f * (fabsf(co[0] * dx * dx) + fabsf(co[2] * dy * dy)) + fabsf(co[1] * dx * dy));
                    if (pass == 1) { if (alpha >= MIN_ALPHA_THRESHOLD * (1.0f - slack)) near_prim[p] = 1; continue; }
                    if (fabsf(alpha * MIN_ALPHA_THRESHOLD_RCP - 1.0f) < slack) { risky = 1; risk_prim[p] = 1; }
                    const uint16_t* sb = screen_bounds + 4 * (size_t)p;
                    if (!(sb[0] < sx1 && sx0 < sb[1] && sb[2] < sy1 && sy0 < sb[3])) continue;
                    if (alpha < MIN_ALPHA_THRESHOLD) continue;
                    Tr *= 1.0f - alpha;
                    if (fabsf(Tr / TRANSMITTANCE_THRESHOLD - 1.0f) < eps_T) risky = 1;
                    if (Tr < TRANSMITTANCE_THRESHOLD) { n_scan = j + 1; break; }                    /* kf:477: the pixel is done */
                }
                if (risky) risk_pixel[pix] = 1;
            }
        }
    }
}

/* sh:71-155 ; returns dcolor/dposition contribution, overwrites grad_sh0 with C0*g and writes grad_sh_rest */
static void sh_to_color_backward(const float* sh_rest, float* grad_sh0, float* grad_sh_rest, const float pos[3],
                                 const float cam[3], uint idx, uint active, uint total_rest, float dpos[3]) {
    const size_t base = (size_t)idx * total_rest * 3;
    const float* k = sh_rest + base; float* gk = grad_sh_rest + base;
    float g[3];
    for (int c = 0; c < 3; c++) { g[c] = grad_sh0[3 * (size_t)idx + c]; grad_sh0[3 * (size_t)idx + c] = C0 * g[c]; }
    dpos[0] = dpos[1] = dpos[2] = 0.0f;
    if (active <= 1) return;
    const float xr = pos[0] - cam[0], yr = pos[1] - cam[1], zr = pos[2] - cam[2];
    const float inv = 1.0f / sqrtf(xr * xr + yr * yr + zr * zr);
    const float x = xr * inv, y = yr * inv, z = zr * inv;
    float gdx[3], gdy[3], gdz[3];
    for (int c = 0; c < 3; c++) {
        gk[0 + c] = -C1 * y * g[c]; gk[3 + c] = C1 * z * g[c]; gk[6 + c] = -C1 * x * g[c];
        gdx[c] = -C1 * k[6 + c]; gdy[c] = -C1 * k[0 + c]; gdz[c] = C1 * k[3 + c];
    }
    if (active > 4) {
        const float xx = x * x, yy = y * y, zz = z * z, xy = x * y, xz = x * z, yz = y * z;
        for (int c = 0; c < 3; c++) {
            gk[9 + c] = C2a * xy * g[c]; gk[12 + c] = -C2a * yz * g[c]; gk[15 + c] = (C2b * zz - C2c) * g[c];
            gk[18 + c] = -C2a * xz * g[c]; gk[21 + c] = C2d * (xx - yy) * g[c];
            gdx[c] = gdx[c] + C2a * y * k[9 + c] - C2a * z * k[18 + c] + C2a * x * k[21 + c];
            gdy[c] = gdy[c] + C2a * x * k[9 + c] - C2a * z * k[12 + c] - C2a * y * k[21 + c];
            gdz[c] = gdz[c] - C2a * y * k[12 + c] + C2e * z * k[15 + c] - C2a * x * k[18 + c];
        }
        if (active > 9) {
            for (int c = 0; c < 3; c++) {
                gk[24 + c] = y * (C3a * yy - C3b * xx) * g[c]; gk[27 + c] = C3c * xy * z * g[c];
                gk[30 + c] = y * (C3d - C3e * zz) * g[c]; gk[33 + c] = z * (C3f * zz - C3g) * g[c];
                gk[36 + c] = x * (C3d - C3e * zz) * g[c]; gk[39 + c] = C3h * z * (xx - yy) * g[c];
                gk[42 + c] = x * (C3b * yy - C3a * xx) * g[c];
                gdx[c] = gdx[c] - C3i * xy * k[24 + c] + C3c * yz * k[27 + c] + (C3d - C3e * zz) * k[36 + c]
                         + C3c * xz * k[39 + c] + C3b * (yy - xx) * k[42 + c];
                gdy[c] = gdy[c] + C3b * (yy - xx) * k[24 + c] + C3c * xz * k[27 + c] + (C3d - C3e * zz) * k[30 + c]
                         - C3c * yz * k[39 + c] + C3i * xy * k[42 + c];
                gdz[c] = gdz[c] + C3c * xy * k[27 + c] - C3j * yz * k[30 + c] + (C3k * zz - C3g) * k[33 + c]
                         - C3j * xz * k[36 + c] + C3h * (xx - yy) * k[39 + c];
            }
        }
    }
    const float gd0 = gdx[0] * g[0] + gdx[1] * g[1] + gdx[2] * g[2];
    const float gd1 = gdy[0] * g[0] + gdy[1] * g[1] + gdy[2] * g[2];
    const float gd2 = gdz[0] * g[0] + gdz[1] * g[1] + gdz[2] * g[2];
    const float xxr = xr * xr, yyr = yr * yr, zzr = zr * zr, xyr = xr * yr, xzr = xr * zr, yzr = yr * zr;
    const float nsq = xxr + yyr + zzr;
    const float s = 1.0f / sqrtf(nsq * nsq * nsq); /* rsqrtf */
    dpos[0] = ((yyr + zzr) * gd0 - xyr * gd1 - xzr * gd2) * s;
    dpos[1] = (-xyr * gd0 + (xxr + zzr) * gd1 - yzr * gd2) * s;
    dpos[2] = (-xzr * gd0 - yzr * gd1 + (xxr + yyr) * gd2) * s;
}

/* K12 kb:15-257. All grad_* outputs must be zero-initialised by the caller except the four accumulators written by
 * K11 (api:127-134). densification_info may be NULL (api:136). */
void orc_preprocess_backward(int N, const float* means, const float* scales, const float* rotations, const float* opacities,
                             const float* sh_rest, const orc_settings* S, const uint* n_touched,
                             const float* grad_mean2d, const float* grad_conic,
                             float* grad_means, float* grad_scales, float* grad_rotations, float* grad_opacities,
                             float* grad_sh0, float* grad_sh_rest, float* densification_info) {
    const float width = (float)S->width, height = (float)S->height;
#pragma omp parallel for schedule(static)
    for (int i = 0; i < N; i++) {
        if (n_touched[i] == 0) continue;                                                             /* kb:45 */
        const float* m = means + 3 * (size_t)i;
        float dpos_color[3];
        sh_to_color_backward(sh_rest, grad_sh0, grad_sh_rest, m, S->cam_pos, (uint)i, (uint)S->active_sh_bases,
                             (uint)S->total_sh_rest, dpos_color);
        proj_t P;
        project(m, scales + 3 * (size_t)i, rotations + 4 * (size_t)i, S, &P);
        const float* r1 = S->w2c; const float* r2 = S->w2c + 4; const float* r3 = S->w2c + 8;
        const float ks = S->proper_aa ? DILATION_PROPER_AA : DILATION;
        const float a = P.a_raw + ks, b = P.b, c = P.c_raw + ks;
        const float aa = a * a, bb = b * b, cc = c * c, ac = a * c, ab = a * b, bc = b * c;
        const float det = ac - bb;
        const float det_rcp_sq = 1.0f / (det * det);
        const float gcx = grad_conic[i], gcy = grad_conic[(size_t)N + i], gcz = grad_conic[2 * (size_t)N + i];
        float dcov_x = det_rcp_sq * (2.0f * bc * gcy - cc * gcx - bb * gcz);                         /* kb:130-134 */
        float dcov_y = det_rcp_sq * (bc * gcx - (ac + bb) * gcy + ab * gcz);
        float dcov_z = det_rcp_sq * (2.0f * ab * gcy - bb * gcx - aa * gcz);
        if (S->proper_aa) {                                                                          /* kb:137-160 (cov2d branch compiled out, cfg:12) */
            const float opacity = sigmoidf(opacities[i]);
            const float gconv = grad_opacities[i];
            const float det_raw = P.a_raw * P.c_raw - bb;
            const float conv = sqrtf(fmaxf(det_raw / det, 0.0f));
            grad_opacities[i] = gconv * conv * opacity * (1.0f - opacity);
        }
        const float* j1 = P.jw1; const float* j2 = P.jw2;
        float dcov3[6];                                                                               /* kb:163-170 */
        dcov3[0] = j1[0] * j1[0] * dcov_x + 2.0f * j1[0] * j2[0] * dcov_y + j2[0] * j2[0] * dcov_z;
        dcov3[1] = j1[0] * j1[1] * dcov_x + (j1[0] * j2[1] + j1[1] * j2[0]) * dcov_y + j2[0] * j2[1] * dcov_z;
        dcov3[2] = j1[0] * j1[2] * dcov_x + (j1[0] * j2[2] + j1[2] * j2[0]) * dcov_y + j2[0] * j2[2] * dcov_z;
        dcov3[3] = j1[1] * j1[1] * dcov_x + 2.0f * j1[1] * j2[1] * dcov_y + j2[1] * j2[1] * dcov_z;
        dcov3[4] = j1[1] * j1[2] * dcov_x + (j1[1] * j2[2] + j1[2] * j2[1]) * dcov_y + j2[1] * j2[2] * dcov_z;
        dcov3[5] = j1[2] * j1[2] * dcov_x + 2.0f * j1[2] * j2[2] * dcov_y + j2[2] * j2[2] * dcov_z;
        float djw1[3], djw2[3];                                                                       /* kb:173-182 */
        for (int k = 0; k < 3; k++) {
            djw1[k] = 2.0f * (P.jwc1[k] * dcov_x + P.jwc2[k] * dcov_y);
            djw2[k] = 2.0f * (P.jwc1[k] * dcov_y + P.jwc2[k] * dcov_z);
        }
        const float dj11 = r1[0] * djw1[0] + r1[1] * djw1[1] + r1[2] * djw1[2];
        const float dj22 = r2[0] * djw2[0] + r2[1] * djw2[1] + r2[2] * djw2[2];
        const float dj13 = r3[0] * djw1[0] + r3[1] * djw1[1] + r3[2] * djw1[2];
        const float dj23 = r3[0] * djw2[0] + r3[1] * djw2[1] + r3[2] * djw2[2];
        const float gm2x = grad_mean2d[2 * (size_t)i], gm2y = grad_mean2d[2 * (size_t)i + 1];
        if (densification_info) {                                                                     /* kb:194-201 */
            densification_info[i] += 1.0f;
            const float nx = 0.5f * (gm2x * width), ny = 0.5f * (gm2y * height);
            densification_info[(size_t)N + i] += sqrtf(nx * nx + ny * ny);
        }
        float dcam[3];                                                                                /* kb:204-217 */
        dcam[0] = P.j11 * gm2x;
        dcam[1] = P.j22 * gm2y;
        dcam[2] = -P.j11 * P.x * gm2x - P.j22 * P.y * gm2y;
        const int valid_x = P.x >= P.clip_l && P.x <= P.clip_r;
        const int valid_y = P.y >= P.clip_t && P.y <= P.clip_b;
        if (valid_x) dcam[0] -= P.j11 * dj13 / P.depth;
        if (valid_y) dcam[1] -= P.j22 * dj23 / P.depth;
        const float fxm = 1.0f + (float)valid_x, fym = 1.0f + (float)valid_y;
        dcam[2] += (P.j11 * (fxm * P.x_clipped * dj13 - dj11) + P.j22 * (fym * P.y_clipped * dj23 - dj22)) / P.depth;
        for (int k = 0; k < 3; k++)                                                                   /* kb:220-228 */
            grad_means[3 * (size_t)i + k] = (r1[k] * dcam[0] + r2[k] * dcam[1] + r3[k] * dcam[2]) + dpos_color[k];
        const float* R = P.R; const float* G = P.RSS;
        for (int k = 0; k < 3; k++) {                                                                 /* kb:231-240 */
            const float dvar = R[k] * R[k] * dcov3[0] + R[3 + k] * R[3 + k] * dcov3[3] + R[6 + k] * R[6 + k] * dcov3[5]
                               + 2.0f * (R[k] * R[3 + k] * dcov3[1] + R[k] * R[6 + k] * dcov3[2] + R[3 + k] * R[6 + k] * dcov3[4]);
            grad_scales[3 * (size_t)i + k] = 2.0f * P.var[k] * dvar;
        }
        float dR[9];                                                                                  /* kb:243-253 */
        for (int k = 0; k < 3; k++) {
            dR[0 + k] = 2.0f * (G[0 + k] * dcov3[0] + G[3 + k] * dcov3[1] + G[6 + k] * dcov3[2]);
            dR[3 + k] = 2.0f * (G[0 + k] * dcov3[1] + G[3 + k] * dcov3[3] + G[6 + k] * dcov3[4]);
            dR[6 + k] = 2.0f * (G[0 + k] * dcov3[2] + G[3 + k] * dcov3[4] + G[6 + k] * dcov3[5]);
        }
        quat_to_rot_backward(rotations + 4 * (size_t)i, dR, grad_rotations + 4 * (size_t)i);
    }
}

/* K13 adam:10-34 (device) + adam:52-54 (host-side bias corrections in double) */
void orc_adam_step(const float* grad, float* param, float* exp_avg, float* exp_avg_sq, long long n,
                   int step, double lr, double beta1, double beta2, double eps) {
    const double bc1_rcp = 1.0 / (1.0 - pow(beta1, step));
    const double bc2_sqrt_rcp = 1.0 / sqrt(1.0 - pow(beta2, step));
    const float step_size = (float)(lr * bc1_rcp);
    const float b1 = (float)beta1, b2 = (float)beta2, e = (float)eps, bc2 = (float)bc2_sqrt_rcp;
#pragma omp parallel for schedule(static)
    for (long long i = 0; i < n; i++) {
        const float g = grad[i];
        const float gsq = g * g;
        const float m1 = fmaf(b1, exp_avg[i] - g, g);
        const float m2 = fmaf(b2, exp_avg_sq[i] - gsq, gsq);
        const float denom = sqrtf(m2) * bc2 + e;
        param[i] -= step_size * m1 / denom;
        exp_avg[i] = m1;
        exp_avg_sq[i] = m2;
    }
}

/* ---- "next" row: loss = lambda_l1 * L1 + lambda_dssim * (1 - SSIM)  (Loss.py:15-16, Trainer.py:52-53) ----
 * The reference calls `fused_dssim` from the NeRFICG framework (Optim/Losses/DSSIM.py), an UN-VENDORED, UN-PINNED dependency
 * that is absent from /root/reference (README.md:72-79 only says "clone NeRFICG"); its source cannot be read here. This
 * restates the published algorithm it wraps: SSIM of Wang et al. 2004 exactly as used by 3D Gaussian Splatting (Kerbl et
 * al. 2023, utils/loss_utils.py) and the fused-ssim kernels: 11x11 Gaussian window (sigma 1.5, outer product of the
 * normalised 1-D window), zero padding ("same" convolution), per channel, C1 = 0.01^2, C2 = 0.03^2, mean over all
 * channels and pixels; DSSIM = 1 - SSIM. PARITY UNPINNED (no fixture of the original exists); pinned against an
 * independent torch conv2d implementation + autograd in tests/test_loss.py.
 * image/target/grad: [3,H,W]. Returns the loss; grad (if not NULL) receives dloss/dimage. */
static void gauss_window(float g[11]) {
    double w[11], s = 0.0;
    for (int i = 0; i < 11; i++) { w[i] = exp(-((i - 5) * (i - 5)) / (2.0 * 1.5 * 1.5)); s += w[i]; }
    for (int i = 0; i < 11; i++) g[i] = (float)(w[i] / s);
}

float orc_l1_dssim(const float* image, const float* target, int H, int W, float lambda_l1, float lambda_dssim,
                   float* grad, float* out_l1, float* out_ssim) {
    float g[11];
    gauss_window(g);
    const float C1 = 0.01f * 0.01f, C2 = 0.03f * 0.03f;
    const size_t P = (size_t)H * W, n_total = 3 * P;
    float* d_mu = (float*)calloc(n_total, sizeof(float));
    float* d_m11 = (float*)calloc(n_total, sizeof(float));
    float* d_m12 = (float*)calloc(n_total, sizeof(float));
    double ssim_sum = 0.0, l1_sum = 0.0;
#pragma omp parallel for reduction(+ : ssim_sum, l1_sum) schedule(static)
    for (long long e = 0; e < (long long)n_total; e++) {
        const int c = (int)(e / (long long)P), y = (int)((e % (long long)P) / W), x = (int)(e % W);
        const float* X = image + (size_t)c * P; const float* Y = target + (size_t)c * P;
        float mu1 = 0, mu2 = 0, m11 = 0, m22 = 0, m12 = 0;
        for (int dy = -5; dy <= 5; dy++) {
            const int yy = y + dy;
            if (yy < 0 || yy >= H) continue;
            for (int dx = -5; dx <= 5; dx++) {
                const int xx = x + dx;
                if (xx < 0 || xx >= W) continue;
                const float w = g[dy + 5] * g[dx + 5];
                const float a = X[(size_t)yy * W + xx], b = Y[(size_t)yy * W + xx];
                mu1 += w * a; mu2 += w * b; m11 += w * a * a; m22 += w * b * b; m12 += w * a * b;
            }
        }
        const float s11 = m11 - mu1 * mu1, s22 = m22 - mu2 * mu2, s12 = m12 - mu1 * mu2;
        const float A1 = 2.0f * mu1 * mu2 + C1, A2 = 2.0f * s12 + C2, B1 = mu1 * mu1 + mu2 * mu2 + C1, B2 = s11 + s22 + C2;
        const float S = (A1 * A2) / (B1 * B2);
        ssim_sum += S;
        l1_sum += fabsf(X[(size_t)y * W + x] - Y[(size_t)y * W + x]);
        /* partial derivatives of S w.r.t. the filtered quantities mu1, E[x^2], E[xy] (mu2, E[y^2] do not depend on x) */
        const float den = B1 * B2;
        d_mu[e] = ((2.0f * mu2 * A2 - 2.0f * mu2 * A1) * den - A1 * A2 * (2.0f * mu1 * B2 - 2.0f * mu1 * B1)) / (den * den);
        d_m11[e] = -(A1 * A2) / (B1 * B2 * B2);
        d_m12[e] = 2.0f * A1 / den;
    }
    const float ssim = (float)(ssim_sum / (double)n_total), l1 = (float)(l1_sum / (double)n_total);
    if (out_l1) *out_l1 = l1;
    if (out_ssim) *out_ssim = ssim;
    if (grad) {
        const float ks = -lambda_dssim / (float)n_total, kl = lambda_l1 / (float)n_total;
#pragma omp parallel for schedule(static)
        for (long long e = 0; e < (long long)n_total; e++) {
            const int c = (int)(e / (long long)P), y = (int)((e % (long long)P) / W), x = (int)(e % W);
            const size_t base = (size_t)c * P;
            float f_mu = 0, f_11 = 0, f_12 = 0;       /* adjoint of the zero-padded convolution = the same symmetric filter */
            for (int dy = -5; dy <= 5; dy++) {
                const int yy = y + dy;
                if (yy < 0 || yy >= H) continue;
                for (int dx = -5; dx <= 5; dx++) {
                    const int xx = x + dx;
                    if (xx < 0 || xx >= W) continue;
                    const float w = g[dy + 5] * g[dx + 5];
                    const size_t q = base + (size_t)yy * W + xx;
                    f_mu += w * d_mu[q]; f_11 += w * d_m11[q]; f_12 += w * d_m12[q];
                }
            }
            const float a = image[e], b = target[e];
            const float diff = a - b;
            const float sgn = diff > 0.0f ? 1.0f : (diff < 0.0f ? -1.0f : 0.0f);
            grad[e] = ks * (f_mu + 2.0f * a * f_11 + b * f_12) + kl * sgn;
        }
    }
    free(d_mu); free(d_m11); free(d_m12);
    return lambda_l1 * l1 + lambda_dssim * (1.0f - ssim);
}


/* ---- "next" row: Speedy-Splat pruning scores, kernels_pruning_scores.cuh:348-505 ----
 * Same pipeline as inference (colour clamped at store, kp = ki up to the blend); per pixel the tile list is blended twice:
 * pass 1 gives the final colour / transmittance, pass 2 re-walks it with dL/dC = (1,1,1) and adds (opacity * dL/dalpha)^2 of
 * every blended (pixel, Gaussian) pair to scores[primitive] (kp:470-494). Accumulates into `scores` (Renderer.py:141-156). */
void orc_pruning_scores(const uint* ranges, const uint* inst_prims, const uint16_t* screen_bounds, const float* mean2d,
                        const float* conic_opacity, const float* color, const orc_settings* S, int N, float* scores) {
    const int W = S->width, H = S->height;
    const int grid_w = (W + TILE_W - 1) / TILE_W, grid_h = (H + TILE_H - 1) / TILE_H;
    const int T = grid_w * grid_h;
    double* acc = (double*)calloc((size_t)N, sizeof(double));
#pragma omp parallel for schedule(dynamic, 4)
    for (int tile = 0; tile < T; tile++) {
        const int tyi = tile / grid_w, txi = tile % grid_w;
        const uint r0 = ranges[2 * tile], r1 = ranges[2 * tile + 1];
        const int n_total = (int)(r1 - r0);
        for (int local = 0; local < BLOCK_BLEND; local++) {
            const int px = txi * TILE_W + local % TILE_W, py = tyi * TILE_H + local / TILE_W;
            if (px >= W || py >= H) continue;
            const int sx0 = txi * TILE_W + ((local % TILE_W) / SUBTILE_W) * SUBTILE_W, sx1 = sx0 + SUBTILE_W;
            const int sy0 = tyi * TILE_H + ((local / TILE_W) / SUBTILE_H) * SUBTILE_H, sy1 = sy0 + SUBTILE_H;
            const float pxf = (float)px + 0.5f, pyf = (float)py + 0.5f;
            float after[3] = {0.0f, 0.0f, 0.0f}, Tr = 1.0f;
            for (int pass = 0; pass < 2; pass++) {
                float galpha = 0.0f;
                if (pass == 1) { galpha = Tr * -(S->bg[0] + S->bg[1] + S->bg[2]); Tr = 1.0f; }         /* kp:441-444 */
                int done = 0;
                for (int j = 0; j < n_total && !done; j++) {
                    const uint p = inst_prims[r0 + j];
                    const uint16_t* sb = screen_bounds + 4 * (size_t)p;
                    if (!(sb[0] < sx1 && sx0 < sb[1] && sb[2] < sy1 && sy0 < sb[3])) continue;
                    const float* co = conic_opacity + 4 * (size_t)p;
                    const float dx = mean2d[2 * (size_t)p] - pxf, dy = mean2d[2 * (size_t)p + 1] - pyf;
                    const float expo = -0.5f * (co[0] * dx * dx + co[2] * dy * dy) - co[1] * dx * dy;
                    const float alpha = co[3] * expf(fminf(expo, 0.0f));
                    if (alpha < MIN_ALPHA_THRESHOLD) continue;
                    const float* col = color + 3 * (size_t)p;
                    const float w = Tr * alpha;
                    if (pass == 0) { after[0] += w * col[0]; after[1] += w * col[1]; after[2] += w * col[2]; }
                    else {
                        after[0] -= w * col[0]; after[1] -= w * col[1]; after[2] -= w * col[2];         /* kp:477 */
                        const float oma = 1.0f - alpha;
                        const float rcp = 1.0f / fmaxf(oma, ONE_MINUS_ALPHA_EPS);
                        const float dLda = ((Tr * col[0] - after[0] * rcp) + (Tr * col[1] - after[1] * rcp) + (Tr * col[2] - after[2] * rcp))
                                           + galpha * rcp;                                               /* kp:480-484 */
                        const float dLdg = co[3] * dLda;
#pragma omp atomic
                        acc[p] += (double)(dLdg * dLdg);                                                 /* kp:487-490 */
                    }
                    Tr *= 1.0f - alpha;
                    if (Tr < TRANSMITTANCE_THRESHOLD) done = 1;
                }
            }
        }
    }
    for (int i = 0; i < N; i++) scores[i] += (float)acc[i];
    free(acc);
}
