// Per-Gaussian device math shared by the preprocess forward/backward kernels and the instance generator.
// Algorithms follow the reference (cited per function); code is written for gfx950 scalar/vector registers, not
// translated from the CUDA helper types. TUs that include this header for key/bound generation are compiled with
// -ffp-contract=off so integer outputs (screen bounds, tile counts, depth keys) are reproducible bit-for-bit against
// the CPU oracle.
#pragma once
#include <hip/hip_runtime.h>
#include <cstdint>
#include "fgs_config.h"

namespace fgs {

struct Camera {           // per-thread view of the camera; the matrix rows are wave-uniform (scalar loads -> SGPRs)
    float r1[4], r2[4], r3[4];
    float pos[3];
    float width, height, fx, fy, cx, cy, near_plane, far_plane;
    int proper_aa;
    int active_sh_bases, total_sh_rest;
    unsigned grid_w, grid_h;
};

// Kernel-argument form: w2c / cam_position stay DEVICE pointers exactly as the reference passes them
// (rasterization_api.cu:68-69), so building the arguments never needs a device->host copy.
struct CameraArgs {
    const float* w2c;       // >= 12 floats, row-major rows 0..2
    const float* cam_pos;   // 3 floats
    float width, height, fx, fy, cx, cy, near_plane, far_plane;
    int proper_aa;
    int active_sh_bases, total_sh_rest;
    unsigned grid_w, grid_h;
};

__device__ __forceinline__ Camera load_camera(const CameraArgs& a) {
    Camera c;
    const float* __restrict__ w = a.w2c;
#pragma unroll
    for (int i = 0; i < 4; ++i) { c.r1[i] = w[i]; c.r2[i] = w[4 + i]; c.r3[i] = w[8 + i]; }
    c.pos[0] = a.cam_pos[0]; c.pos[1] = a.cam_pos[1]; c.pos[2] = a.cam_pos[2];
    c.width = a.width; c.height = a.height; c.fx = a.fx; c.fy = a.fy; c.cx = a.cx; c.cy = a.cy;
    c.near_plane = a.near_plane; c.far_plane = a.far_plane; c.proper_aa = a.proper_aa;
    c.active_sh_bases = a.active_sh_bases; c.total_sh_rest = a.total_sh_rest; c.grid_w = a.grid_w; c.grid_h = a.grid_h;
    return c;
}

// Per-primitive record consumed by instance creation and both blend kernels: ONE 48-byte line instead of the
// reference's four separate arrays (buffer_utils.h:52-56), so a gathered read touches one cache line.
struct alignas(16) PrimRec {
    float mx, my, ca, cb;          // mean2d.xy, conic.x (a), conic.y (b)
    float cc, opacity, r, g;       // conic.z (c), opacity, colour.rg
    float b;                       // colour.b
    uint32_t bx, by;               // screen bounds: x_min | x_max<<16, y_min | y_max<<16   (ushort4 of kf:168-175)
    uint32_t hit_mask;             // <= 32 candidate tiles: exact-overlap bitmap over them (row-major in the tile bounding box);
                                   // > kHotFootprint candidates: hot-accumulator slot + 1 (0 = none); otherwise 0
};
static_assert(sizeof(PrimRec) == 48, "PrimRec must be 48 bytes");

__device__ __forceinline__ float sigmoid_f(float x) { return 1.0f / (1.0f + expf(-x)); }   // kernel_utils.cuh:11-13
__device__ __forceinline__ float saturate_f(float x) { return fminf(fmaxf(x, 0.0f), 1.0f); } // NaN -> 0 like __saturatef

// kernel_utils.cuh:15-30
__device__ __forceinline__ void quat_to_rotation(float r, float x, float y, float z, float (&R)[9], float& norm_sq) {
    const float xx = x * x, yy = y * y, zz = z * z;
    const float xy = x * y, xz = x * z, yz = y * z;
    const float rx = r * x, ry = r * y, rz = r * z;
    norm_sq = r * r + xx + yy + zz;
    const float n = 1.0f / norm_sq;
    R[0] = 1.0f - 2.0f * (yy + zz) * n; R[1] = 2.0f * (xy - rz) * n;        R[2] = 2.0f * (xz + ry) * n;
    R[3] = 2.0f * (xy + rz) * n;        R[4] = 1.0f - 2.0f * (xx + zz) * n; R[5] = 2.0f * (yz - rx) * n;
    R[6] = 2.0f * (xz - ry) * n;        R[7] = 2.0f * (yz + rx) * n;        R[8] = 1.0f - 2.0f * (xx + yy) * n;
}

// kernel_utils.cuh:32-59
__device__ __forceinline__ void quat_to_rotation_backward(float r, float x, float y, float z, const float (&d)[9], float (&out)[4]) {
    const float xx = x * x, yy = y * y, zz = z * z;
    const float xy = x * y, xz = x * z, yz = y * z;
    const float rx = r * x, ry = r * y, rz = r * z;
    const float n = 1.0f / (r * r + xx + yy + zz);
    const float dxx = d[4] + d[8], dyy = d[0] + d[8], dzz = d[0] + d[4];
    const float drz = d[3] - d[1], dxy = d[3] + d[1];
    const float dry = d[2] - d[6], dxz = d[2] + d[6];
    const float drx = d[7] - d[5], dyz = d[7] + d[5];
    const float two = 2.0f * n;
    const float h = two * (xy * dxy + xz * dxz + yz * dyz + rx * drx + ry * dry + rz * drz - xx * dxx - yy * dyy - zz * dzz);
    out[0] = two * (x * drx + y * dry + z * drz - r * h);
    out[1] = two * (r * drx - 2.0f * x * dxx + y * dxy + z * dxz - x * h);
    out[2] = two * (r * dry + x * dxy - 2.0f * y * dyy + z * dyz - y * h);
    out[3] = two * (r * drz + x * dxz + y * dyz - 2.0f * z * dzz - z * h);
}

// Exact tile/Gaussian overlap (StopThePop), kernel_utils.cuh:72-114: the closest point of the tile rectangle (pixel centres) to the
// Gaussian centre along the conic; the tile contributes iff the power there stays within the threshold. (sx, sy) is the mean shifted by -0.5.
// Per-Gaussian invariants are hoisted: the reference's two IEEE divisions have denominators (dx ca) dx and (dy cc) dy with
// dx = +-15, dy = +-11, i.e. (15 ca) 15 and (11 cc) 11 whatever the signs (a sign flip is exact), so they depend on the Gaussian only.
// With r = RN(1 / den) taken once per Gaussian, each quotient is formed by two FMA-residual corrections of q = num * r (Markstein's
// theorem: q' = RN(q + r (num - den q)) is the correctly rounded quotient when r is the correctly rounded reciprocal and q is within
// an ulp), which reproduces the division bit for bit in the normal range at 5 full-rate instructions instead of the ~11 instruction
// v_div_scale / v_rcp / v_div_fmas / v_div_fixup sequence. Explicit fmaf: this header is compiled with -ffp-contract=off.
struct TileTest {
    float sx, sy, ca, cb, cc, pt;      // mean shifted by -0.5, conic, power threshold
    float den_x, den_y, rcp_x, rcp_y;
};
__device__ __forceinline__ TileTest make_tile_test(float sx, float sy, float ca, float cb, float cc, float pt) {
    TileTest t;
    t.sx = sx; t.sy = sy; t.ca = ca; t.cb = cb; t.cc = cc; t.pt = pt;
    const float wx = static_cast<float>(kTileW - 1), wy = static_cast<float>(kTileH - 1);
    t.den_x = (wx * ca) * wx; t.den_y = (wy * cc) * wy;
    t.rcp_x = 1.0f / t.den_x; t.rcp_y = 1.0f / t.den_y;
    return t;
}
__device__ __forceinline__ float exact_quotient(float num, float den, float rcp) {
    float q = num * rcp;
    q = fmaf(fmaf(-den, q, num), rcp, q);
    q = fmaf(fmaf(-den, q, num), rcp, q);
    return q;
}
__device__ __forceinline__ bool tile_contributes(const TileTest& g, unsigned tile_x, unsigned tile_y) {
    const float min_x = static_cast<float>(tile_x * kTileW), min_y = static_cast<float>(tile_y * kTileH);
    const float max_x = static_cast<float>((tile_x + 1) * kTileW - 1), max_y = static_cast<float>((tile_y + 1) * kTileH - 1);
    const float x_min_diff = min_x - g.sx, y_min_diff = min_y - g.sy;
    const float x_left = x_min_diff >= 0.0f ? 1.0f : 0.0f;
    const float y_above = y_min_diff >= 0.0f ? 1.0f : 0.0f;
    const float out_x = x_left + (g.sx > max_x ? 1.0f : 0.0f);
    const float out_y = y_above + (g.sy > max_y ? 1.0f : 0.0f);
    if (out_y + out_x == 0.0f) return true;
    const float corner_x = max_x + x_left * (min_x - max_x);
    const float corner_y = max_y + y_above * (min_y - max_y);
    const float diff_x = g.sx - corner_x, diff_y = g.sy - corner_y;
    const float dx = copysignf(static_cast<float>(kTileW - 1), x_min_diff);
    const float dy = copysignf(static_cast<float>(kTileH - 1), y_min_diff);
    const float tx = out_y * saturate_f(exact_quotient(dx * g.ca * diff_x + dx * g.cb * diff_y, g.den_x, g.rcp_x));
    const float ty = out_x * saturate_f(exact_quotient(dy * g.cb * diff_x + dy * g.cc * diff_y, g.den_y, g.rcp_y));
    const float ex = g.sx - (corner_x + tx * dx), ey = g.sy - (corner_y + ty * dy);
    const float max_power = 0.5f * (g.ca * ex * ex + g.cc * ey * ey) + g.cb * ex * ey;
    return max_power <= g.pt;
}

// SH basis constants, sh_utils.cuh:8-28
constexpr float kC0 = 0.28209479177387814f, kC1 = 0.48860251190291987f;
constexpr float kC2a = 1.0925484305920792f, kC2b = 0.94617469575755997f, kC2c = 0.31539156525251999f,
                kC2d = 0.54627421529603959f, kC2e = 1.8923493915151202f;
constexpr float kC3a = 0.59004358992664352f, kC3b = 1.7701307697799304f, kC3c = 2.8906114426405538f,
                kC3d = 0.45704579946446572f, kC3e = 2.2852289973223288f, kC3f = 1.865881662950577f,
                kC3g = 1.1195289977703462f, kC3h = 1.4453057213202769f, kC3i = 3.5402615395598609f,
                kC3j = 4.5704579946446566f, kC3k = 5.597644988851731f;

// The 15 non-constant SH basis values for direction (x,y,z), in coefficient order (sh_utils.cuh:47-65).
// Entries beyond the active degree are left untouched.
__device__ __forceinline__ void sh_basis(float x, float y, float z, unsigned active, float (&B)[15]) {
    B[0] = -kC1 * y; B[1] = kC1 * z; B[2] = -kC1 * x;
    if (active > 4) {
        const float xx = x * x, yy = y * y, zz = z * z, xy = x * y, xz = x * z, yz = y * z;
        B[3] = kC2a * xy; B[4] = -kC2a * yz; B[5] = kC2b * zz - kC2c; B[6] = -kC2a * xz; B[7] = kC2d * (xx - yy);
        if (active > 9) {
            B[8] = y * (kC3a * yy - kC3b * xx); B[9] = kC3c * xy * z; B[10] = y * (kC3d - kC3e * zz);
            B[11] = z * (kC3f * zz - kC3g); B[12] = x * (kC3d - kC3e * zz); B[13] = kC3h * z * (xx - yy);
            B[14] = x * (kC3b * yy - kC3a * xx);
        }
    }
}

// sh_utils.cuh:32-69. `k` points at this primitive's first rest coefficient (3 floats per basis).
__device__ __forceinline__ void sh_to_color(const float* __restrict__ sh0, const float* __restrict__ k,
                                            float vx, float vy, float vz, unsigned active, float (&out)[3]) {
#pragma unroll
    for (int c = 0; c < 3; ++c) out[c] = 0.5f + kC0 * sh0[c];
    if (active > 1) {
        const float inv = 1.0f / sqrtf(vx * vx + vy * vy + vz * vz);
        const float x = vx * inv, y = vy * inv, z = vz * inv;
#pragma unroll
        for (int c = 0; c < 3; ++c) out[c] = out[c] - kC1 * y * k[0 + c] + kC1 * z * k[3 + c] - kC1 * x * k[6 + c];
        if (active > 4) {
            const float xx = x * x, yy = y * y, zz = z * z, xy = x * y, xz = x * z, yz = y * z;
#pragma unroll
            for (int c = 0; c < 3; ++c)
                out[c] = out[c] + kC2a * xy * k[9 + c] - kC2a * yz * k[12 + c] + (kC2b * zz - kC2c) * k[15 + c]
                         - kC2a * xz * k[18 + c] + kC2d * (xx - yy) * k[21 + c];
            if (active > 9) {
#pragma unroll
                for (int c = 0; c < 3; ++c)
                    out[c] = out[c] + y * (kC3a * yy - kC3b * xx) * k[24 + c] + kC3c * xy * z * k[27 + c]
                             + y * (kC3d - kC3e * zz) * k[30 + c] + z * (kC3f * zz - kC3g) * k[33 + c]
                             + x * (kC3d - kC3e * zz) * k[36 + c] + kC3h * z * (xx - yy) * k[39 + c]
                             + x * (kC3b * yy - kC3a * xx) * k[42 + c];
            }
        }
    }
}

// Projection + EWA terms needed by both preprocess directions (kernels_forward.cuh:61-139 / kernels_backward.cuh:57-114).
struct Projection {
    float depth, x, y;
    float var[3], R[9], RSS[9], cov3d[6];
    float norm_sq;
    float clip_l, clip_r, clip_t, clip_b, x_clipped, y_clipped;
    float j11, j13, j22, j23;
    float jw1[3], jw2[3], jwc1[3], jwc2[3];
    float a_raw, b, c_raw;
};

__device__ __forceinline__ float view_depth(const Camera& cam, float mx, float my, float mz) {
    return cam.r3[0] * mx + cam.r3[1] * my + cam.r3[2] * mz + cam.r3[3];
}

__device__ __forceinline__ void project_gaussian(const Camera& cam, const float (&m)[3], const float (&s)[3], const float (&q)[4],
                                                 Projection& P) {
    P.depth = view_depth(cam, m[0], m[1], m[2]);
#pragma unroll
    for (int i = 0; i < 3; ++i) P.var[i] = expf(2.0f * s[i]);
    quat_to_rotation(q[0], q[1], q[2], q[3], P.R, P.norm_sq);
#pragma unroll
    for (int r = 0; r < 3; ++r)
#pragma unroll
        for (int c = 0; c < 3; ++c) P.RSS[3 * r + c] = P.R[3 * r + c] * P.var[c];
    const float* G = P.RSS; const float* R = P.R;
    P.cov3d[0] = G[0] * R[0] + G[1] * R[1] + G[2] * R[2];
    P.cov3d[1] = G[0] * R[3] + G[1] * R[4] + G[2] * R[5];
    P.cov3d[2] = G[0] * R[6] + G[1] * R[7] + G[2] * R[8];
    P.cov3d[3] = G[3] * R[3] + G[4] * R[4] + G[5] * R[5];
    P.cov3d[4] = G[3] * R[6] + G[4] * R[7] + G[5] * R[8];
    P.cov3d[5] = G[6] * R[6] + G[7] * R[7] + G[8] * R[8];
    P.x = (cam.r1[0] * m[0] + cam.r1[1] * m[1] + cam.r1[2] * m[2] + cam.r1[3]) / P.depth;
    P.y = (cam.r2[0] * m[0] + cam.r2[1] * m[1] + cam.r2[2] * m[2] + cam.r2[3]) / P.depth;
    P.clip_l = (-0.15f * cam.width - cam.cx) / cam.fx;
    P.clip_r = (1.15f * cam.width - cam.cx) / cam.fx;
    P.clip_t = (-0.15f * cam.height - cam.cy) / cam.fy;
    P.clip_b = (1.15f * cam.height - cam.cy) / cam.fy;
    P.x_clipped = fmaxf(P.clip_l, fminf(P.x, P.clip_r));
    P.y_clipped = fmaxf(P.clip_t, fminf(P.y, P.clip_b));
    P.j11 = cam.fx / P.depth; P.j13 = -P.j11 * P.x_clipped;
    P.j22 = cam.fy / P.depth; P.j23 = -P.j22 * P.y_clipped;
#pragma unroll
    for (int i = 0; i < 3; ++i) {
        P.jw1[i] = P.j11 * cam.r1[i] + P.j13 * cam.r3[i];
        P.jw2[i] = P.j22 * cam.r2[i] + P.j23 * cam.r3[i];
    }
    const float* C = P.cov3d;
    P.jwc1[0] = P.jw1[0] * C[0] + P.jw1[1] * C[1] + P.jw1[2] * C[2];
    P.jwc1[1] = P.jw1[0] * C[1] + P.jw1[1] * C[3] + P.jw1[2] * C[4];
    P.jwc1[2] = P.jw1[0] * C[2] + P.jw1[1] * C[4] + P.jw1[2] * C[5];
    P.jwc2[0] = P.jw2[0] * C[0] + P.jw2[1] * C[1] + P.jw2[2] * C[2];
    P.jwc2[1] = P.jw2[0] * C[1] + P.jw2[1] * C[3] + P.jw2[2] * C[4];
    P.jwc2[2] = P.jw2[0] * C[2] + P.jw2[1] * C[4] + P.jw2[2] * C[5];
    P.a_raw = P.jwc1[0] * P.jw1[0] + P.jwc1[1] * P.jw1[1] + P.jwc1[2] * P.jw1[2];
    P.b     = P.jwc1[0] * P.jw2[0] + P.jwc1[1] * P.jw2[1] + P.jwc1[2] * P.jw2[2];
    P.c_raw = P.jwc2[0] * P.jw2[0] + P.jwc2[1] * P.jw2[1] + P.jwc2[2] * P.jw2[2];
}

// tile rectangle of a screen-bounds pair (kernel_utils.cuh:61-68)
__device__ __forceinline__ void tile_rect(uint32_t bx, uint32_t by, unsigned& tx0, unsigned& tx1, unsigned& ty0, unsigned& ty1) {
    const unsigned x_min = bx & 0xffffu, x_max = bx >> 16, y_min = by & 0xffffu, y_max = by >> 16;
    tx0 = x_min / kTileW; tx1 = (x_max + kTileW - 1) / kTileW;
    ty0 = y_min / kTileH; ty1 = (y_max + kTileH - 1) / kTileH;
}

// Footprint row: 16 bytes per VISIBLE Gaussian, written by preprocess in compaction order and carried into depth order by the depth sort's last
// scatter pass (radix_sort.hip), so that the offsets scan and the instance kernel stream what they need instead of gathering a record line per
// Gaussian:  x = primitive index,  y = tile box (tx0 | ty0 << 10 | (width - 1) << 20 | (height - 1) << 26),  z, w = exact-overlap bitmap of the
// box's <= 64 candidate tiles, row-major. Boxes of more than 64 candidates, or beyond 1024 tiles in x or y, are ESCAPE rows: y = kFootprintEscape,
// z = the tile count, w = the number of candidate tiles of the box -- the instance kernel re-tests those from the record (boxes above
// kBigInstanceFootprint candidates by a workgroup each: the depth sort's last pass lists them).
constexpr uint32_t kFootprintEscape = 0xffffffffu;          // width - 1 = height - 1 = 63 is no box of <= 64 candidates
constexpr unsigned kFootprintBitmapTiles = 64;
__device__ __forceinline__ bool footprint_box_fits(unsigned tx0, unsigned ty0, unsigned tbw, unsigned n_max) {
    return n_max >= 1u && n_max <= kFootprintBitmapTiles && tx0 < 1024u && ty0 < 1024u && tbw >= 1u;
}
__device__ __forceinline__ uint32_t footprint_box(unsigned tx0, unsigned ty0, unsigned tbw, unsigned tbh) {
    return tx0 | (ty0 << 10) | ((tbw - 1u) << 20) | ((tbh - 1u) << 26);
}
__host__ __device__ __forceinline__ uint32_t footprint_tile_count(const uint4& row) {
    if (row.y == kFootprintEscape) return row.z;
#if defined(__HIP_DEVICE_COMPILE__)
    return static_cast<uint32_t>(__popc(row.z) + __popc(row.w));
#else
    return static_cast<uint32_t>(__builtin_popcount(row.z) + __builtin_popcount(row.w));
#endif
}

}  // namespace fgs
