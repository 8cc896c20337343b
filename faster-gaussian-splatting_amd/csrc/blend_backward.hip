// K11: blend backward for gfx950 -- one wave64 per bucket of 64 depth-consecutive Gaussians of one tile; the tile's 192
// pixels stream through the wave's lanes as a systolic pipeline (lane l holds Gaussian l, pixel state moves one lane per
// step), gradients accumulate in registers and leave through 9 atomics per lane at the end.
// Semantics: reference kernels_backward.cuh:260-471 (per-32 buckets, 9 shfl_up + an LDS reload per step).
//
// CDNA4 shape:
//  * bucket = wavefront = 64 Gaussians: 255 steps serve 64 lanes (reference: 223 steps serve 32), and the forward
//    pass writes half as many checkpoints.
//  * the per-step lane shift is DPP wave_shr:1 and the next pixel is injected into lane 0 by the same instruction
//    (its `old` operand), fed from a register ring rotated with DPP wave_rol:1 -- no LDS, no barrier, no readlane.
//  * per-pixel constants (dL/dC, C_final - T_final*bg, T_final*-(dL/dC . bg), last contributor) are staged ONCE per
//    backward pass into a tile-major 32-byte record, so each bucket reads 48 B per pixel with three coalesced 16-byte
//    loads instead of gathering 9 scalars from image-linear arrays per bucket.
#include "fgs_kernels.h"
#include <fgs_wave.h>

namespace fgs {

// staging pass: kb:349-380 hoisted out of the per-bucket loop
__global__ void __launch_bounds__(kTilePixels) stage_pixels_kernel(const BlendBackwardArgs a) {
    const unsigned tile = blockIdx.x, local = threadIdx.x;
    const unsigned tile_x = tile % a.grid_w, tile_y = tile / a.grid_w;
    const unsigned px = tile_x * kTileW + (local % kTileW), py = tile_y * kTileH + (local / kTileW);
    float4 g = make_float4(0.0f, 0.0f, 0.0f, 0.0f), c = make_float4(0.0f, 0.0f, 0.0f, __uint_as_float(0u));
    if (px < a.width && py < a.height) {
        const size_t pix = (size_t)a.width * py + px, n_pixels = (size_t)a.width * a.height;
        const float fT = a.final_T[(size_t)tile * kTilePixels + local];
        const float b0 = a.bg[0], b1 = a.bg[1], b2 = a.bg[2];
        g.x = a.grad_image[pix]; g.y = a.grad_image[n_pixels + pix]; g.z = a.grad_image[2 * n_pixels + pix];
        g.w = fT * -(g.x * b0 + g.y * b1 + g.z * b2);                                  // kb:375-377
        c.x = a.image[pix] - fT * b0; c.y = a.image[n_pixels + pix] - fT * b1; c.z = a.image[2 * n_pixels + pix] - fT * b2;
        c.w = __uint_as_float(a.n_processed[(size_t)tile * kTilePixels + local]);
    }
    a.pixrec[((size_t)tile * kTilePixels + local) * 2] = g;
    a.pixrec[((size_t)tile * kTilePixels + local) * 2 + 1] = c;
}

constexpr int kNV = 9;   // after.rgb, T, dL/dC.rgb, alpha-common, last contributor

__global__ void __launch_bounds__(kBackwardWavesPerBlock * kWave) blend_backward_kernel(const BlendBackwardArgs a) {
    const unsigned lane = threadIdx.x & 63u;
    const unsigned bucket = blockIdx.x * kBackwardWavesPerBlock + (threadIdx.x >> 6);
    const unsigned n_buckets = a.bucket_offsets[a.n_tiles - 1];          // device-side count: no host sync for the grid size
    if (bucket >= n_buckets) return;                                       // wave-uniform
    const unsigned tile = a.bucket_tile[bucket];
    const uint2 range = a.ranges[tile];
    const unsigned tile_n = range.y - range.x;
    const unsigned first = tile == 0 ? 0u : a.bucket_offsets[tile - 1];
    const unsigned tb = bucket - first;
    if (tb * kBucket >= a.max_n_processed[tile]) return;                   // kb:295

    const unsigned tp = tb * kBucket + lane;
    const bool valid_prim = tp < tile_n;
    uint32_t prim = 0;
    float mx = 0.0f, my = 0.0f, ca = 0.0f, cb = 0.0f, cc = 0.0f, op = 0.0f;
    float col0 = 0.0f, col1 = 0.0f, col2 = 0.0f, f0 = 0.0f, f1 = 0.0f, f2 = 0.0f;
    if (valid_prim) {
        prim = a.inst_prims[range.x + tp];
        const float4* r = reinterpret_cast<const float4*>(a.rec + prim);
        const float4 r0 = r[0], r1 = r[1];
        const float raw2 = reinterpret_cast<const float*>(r + 2)[0];
        mx = r0.x; my = r0.y; ca = r0.z; cb = r0.w; cc = r1.x; op = r1.y;
        col0 = fmaxf(r1.z, 0.0f); col1 = fmaxf(r1.w, 0.0f); col2 = fmaxf(raw2, 0.0f);
        f0 = r1.z >= 0.0f ? 1.0f : 0.0f; f1 = r1.w >= 0.0f ? 1.0f : 0.0f; f2 = raw2 >= 0.0f ? 1.0f : 0.0f;   // kb:313-318
    }
    const float x0 = static_cast<float>((tile % a.grid_w) * kTileW) + 0.5f;
    const float y0 = static_cast<float>((tile / a.grid_w) * kTileH) + 0.5f;
    const float4* __restrict__ pix = a.pixrec + (size_t)tile * kTilePixels * 2;
    const float4* __restrict__ ck = a.ckpt + (size_t)bucket * kTilePixels;

    float d_mx = 0.0f, d_my = 0.0f, d_ca = 0.0f, d_cb = 0.0f, d_cc = 0.0f, d_op = 0.0f, d_c0 = 0.0f, d_c1 = 0.0f, d_c2 = 0.0f;
    float state[kNV], feed[kNV];
#pragma unroll
    for (int k = 0; k < kNV; ++k) { state[k] = 0.0f; feed[k] = 0.0f; }

    for (int chunk = 0; chunk < 4; ++chunk) {
        if (chunk < 3) {                                                   // lane l stages pixel 64*chunk + l
            const unsigned p = static_cast<unsigned>(chunk) * kWave + lane;
            const float4 g = pix[2 * p], c = pix[2 * p + 1], k = ck[p];
            feed[0] = c.x - k.x; feed[1] = c.y - k.y; feed[2] = c.z - k.z; feed[3] = k.w;      // kb:371-374
            feed[4] = g.x; feed[5] = g.y; feed[6] = g.z; feed[7] = g.w; feed[8] = c.w;
        }
        const int steps = chunk < 3 ? kWave : kWave - 1;
        for (int s = 0; s < steps; ++s) {
            pipeline_advance<kNV>(state, feed);                            // kb:383-410 in two DPP ops per value
            const int idx = chunk * kWave + s - static_cast<int>(lane);    // pixel handled by this lane in this step
            const unsigned last = __float_as_uint(state[8]);
            if (!valid_prim || idx < 0 || idx >= kTilePixels || tp >= last) continue;          // kb:412
            const float dx = mx - (x0 + static_cast<float>(idx & (kTileW - 1)));
            const float dy = my - (y0 + static_cast<float>(idx >> 4));
            const float power = -0.5f * (ca * dx * dx + cc * dy * dy) - cb * dx * dy;
            const float gauss = __expf(fminf(power, 0.0f));
            const float alpha = op * gauss;
            if (alpha < kMinAlphaThreshold) continue;
            const float T = state[3];
            const float w = T * alpha;
            d_c0 += w * state[4] * f0; d_c1 += w * state[5] * f1; d_c2 += w * state[6] * f2;  // kb:426-427
            state[0] -= w * col0; state[1] -= w * col1; state[2] -= w * col2;                  // kb:429
            const float oma = 1.0f - alpha;
            const float oma_rcp = __frcp_rn(fmaxf(oma, kOneMinusAlphaEps));
            const float dl_dalpha = (T * col0 - state[0] * oma_rcp) * state[4] + (T * col1 - state[1] * oma_rcp) * state[5]
                                    + (T * col2 - state[2] * oma_rcp) * state[6] + state[7] * oma_rcp;   // kb:434-436
            d_op += gauss * dl_dalpha;
            const float h = -alpha * dl_dalpha;
            const float hh = 0.5f * h;
            d_ca += hh * (dx * dx); d_cb += hh * (dx * dy); d_cc += hh * (dy * dy);            // kb:443-448
            d_mx += h * (ca * dx + cb * dy); d_my += h * (cb * dx + cc * dy);                  // kb:449-453
            state[3] = T * oma;
        }
    }

    if (valid_prim) {                                                      // kb:459-470
        const size_t n = a.n;
        unsafeAtomicAdd(a.acc + prim, d_mx);
        unsafeAtomicAdd(a.acc + n + prim, d_my);
        unsafeAtomicAdd(a.acc + 2 * n + prim, d_ca);
        unsafeAtomicAdd(a.acc + 3 * n + prim, d_cb);
        unsafeAtomicAdd(a.acc + 4 * n + prim, d_cc);
        unsafeAtomicAdd(a.acc + 5 * n + prim, a.proper_aa ? d_op : op * (1.0f - op) * d_op);
        unsafeAtomicAdd(a.acc + 6 * n + prim, d_c0);
        unsafeAtomicAdd(a.acc + 7 * n + prim, d_c1);
        unsafeAtomicAdd(a.acc + 8 * n + prim, d_c2);
    }
}

hipError_t launch_stage_pixels(const BlendBackwardArgs& a, hipStream_t s) {
    hipLaunchKernelGGL(stage_pixels_kernel, dim3(a.n_tiles), dim3(kTilePixels), 0, s, a);
    return hipGetLastError();
}

hipError_t launch_blend_backward(const BlendBackwardArgs& a, hipStream_t s) {
    if (a.n_buckets_cap == 0) return hipSuccess;
    const dim3 grid((a.n_buckets_cap + kBackwardWavesPerBlock - 1) / kBackwardWavesPerBlock), block(kBackwardWavesPerBlock * kWave);
    hipLaunchKernelGGL(blend_backward_kernel, grid, block, 0, s, a);
    return hipGetLastError();
}

}  // namespace fgs
