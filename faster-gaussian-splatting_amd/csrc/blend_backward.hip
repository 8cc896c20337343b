// K11: blend backward for gfx950 -- one wave64 per bucket of 64 depth-consecutive Gaussians of one tile; the tile's 192
// pixels stream through the wave's lanes as a systolic pipeline (lane l holds Gaussian l, pixel state moves one lane per
// step), gradients accumulate in registers and leave through 9 atomics per lane at the end.
// Semantics: reference kernels_backward.cuh:260-471 (per-32 buckets, 9 shfl_up + an LDS reload per step).
//
// CDNA4 shape:
//  * bucket = wavefront = 64 Gaussians: 255 steps serve 64 lanes (reference: 223 steps serve 32), and the forward
//    pass writes half as many checkpoints.
//  * only the 4 values a Gaussian changes (remaining colour, transmittance) travel through the lanes, one in-place DPP
//    wave_shr:1 each; the next pixel enters at lane 0 from a wave-uniform LDS read, and the 5 per-pixel constants are
//    read by lane l at LDS slot (step - l) -- conflict-free consecutive slots -- instead of being shifted. The wave's
//    6.9 KB LDS slice is private: no workgroup barrier anywhere. (v1 shifted all 9 values through a register ring:
//    ~40 mov/DPP per step; this form needs 8 + 3 LDS reads.)
//  * per-pixel constants (dL/dC, C_final - T_final*bg, T_final*-(dL/dC . bg), last contributor) are staged ONCE per
//    backward pass into a tile-major 32-byte record, so each bucket reads 48 B per pixel with three coalesced 16-byte
//    loads instead of gathering 9 scalars from image-linear arrays per bucket.
#include <type_traits>

#include "fgs_kernels.h"
#include <fgs_wave.h>
#include "fgs_tile_scan.h"

#ifndef FGS_CKPT_NT
#define FGS_CKPT_NT 1      // round 6: ... and are read as non-temporal loads (training iteration 2.218 -> 2.187 ms, layered scene 3.728 -> 3.706, three alternating pairs: profiles/r06_ab_ckpt_nt.txt); 0: A/B
#endif
namespace fgs {

// staging pass: kb:349-380 hoisted out of the per-bucket loop
__global__ void __launch_bounds__(kTilePixels) stage_pixels_kernel(const BlendBackwardArgs a) {
    // (Round 6, measured and withdrawn: XCD x taking a contiguous band of tiles, so that the two tiles sharing a 128-byte line of the six image planes
    // run under one L2 -- what took 6 % off the loss kernels -- leaves this kernel at 0.040 ms: profiles/r06_ab_stage_pixels_xcd.txt.)
    const unsigned tile = blockIdx.x;
    const unsigned local = threadIdx.x;
    if (tile < a.n_tiles) {
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
        // the tile's entries of the live-bucket list (variant 3): the planning pass has scanned the per-tile counts, the entries are written here,
        // by 12 k workgroups instead of one
        if (a.live_offsets != nullptr) {
            const unsigned nl = (a.max_n_processed[tile] + kBucket - 1) / kBucket;              // kb:295
            const unsigned base = a.live_offsets[tile];
            for (unsigned k = local; k < nl; k += kTilePixels) a.work_list[base + k] = make_uint2(tile, k);
        }
    }
    // K11 adds into records that start at zero (replaces api:127-134). The records of the visible Gaussians were cleared by K1 in the forward pass;
    // left for this kernel are the hot replicas -- or records and replicas alike when no K1 filled the blob or a backward pass already ran over it
    // (BlendBackwardArgs). Every workgroup clears an equal share, 16 bytes per store. (Round 5: a hipMemsetAsync of 117 MB in front of this kernel,
    // 16 us on the critical path of every backward pass; doing all of it here costs the same 16 us -- HBM write rate, profiles/r06_ab_acc_clear.txt.)
    {
        const bool everything = a.clear_everything != 0 || *a.dirty_flag != 0u;
        const unsigned n_f4 = everything ? a.clear_all_f4 : a.clear_hot_f4;
        float4* const z = reinterpret_cast<float4*>(everything ? a.acc : a.acc_hot);
        const unsigned per_group = (n_f4 + gridDim.x - 1u) / gridDim.x;
        const unsigned first = tile * per_group, last = min(first + per_group, n_f4);
        for (unsigned k = first + local; k < last; k += kTilePixels) z[k] = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
    }
}

#ifdef FGS_DEV_SWITCHES   // A/B exhibits (variants 0 / 2 systolic, 1 strip): built into libfgs_hip_dev.so only, see docs/history.md
template <bool GLOBAL_GRAD>
__global__ void __launch_bounds__(kBackwardWavesPerBlock * kWave) blend_backward_kernel(const BlendBackwardArgs a) {
    const unsigned lane = threadIdx.x & 63u, wv = threadIdx.x >> 6;
    const unsigned bucket = blockIdx.x * kBackwardWavesPerBlock + wv;
    const unsigned n_buckets = a.bucket_offsets[a.n_tiles - 1];          // device-side count: no host sync for the grid size
    if (bucket >= n_buckets) return;                                       // wave-uniform (no block-level barrier below)
    const unsigned tile = a.bucket_tile[bucket];
    const uint2 range = a.ranges[tile];
    const unsigned tile_n = range.y - range.x;
    const unsigned first = tile == 0 ? 0u : a.bucket_offsets[tile - 1];
    const unsigned tb = bucket - first;
    if (tb * kBucket >= a.max_n_processed[tile]) return;                   // kb:295

    // ---- stage this bucket's 192 pixels in the wave's private LDS slice (kb:349-380): 36 B per pixel ----
    __shared__ float4 s_init[kBackwardWavesPerBlock][kTilePixels];   // C_final - T_final*bg - C_ckpt (rgb), T_ckpt: injected at lane 0
    // dL/dC rgb, T_final * -(dL/dC . bg): read by lane l at pixel i-l. GLOBAL_GRAD reads it from the tile-major staging record
    // (L1/L2-resident, shared by the tile's buckets) instead, which frees 3 KB of LDS per wave -> 32 instead of 21 waves per CU
    __shared__ float4 s_grad[kBackwardWavesPerBlock][GLOBAL_GRAD ? 1 : kTilePixels];
    // last contributor, padded by one wave width on both sides: slots outside the tile read 0, so `tp < last` is the only
    // per-step validity test (no range compare, no index clamp)
    __shared__ uint32_t s_last[kBackwardWavesPerBlock][kWave + kTilePixels + kWave];
    {
        const float4* __restrict__ pix = a.pixrec + (size_t)tile * kTilePixels * 2;
        const float4* __restrict__ ck = a.ckpt + (size_t)bucket * kTilePixels;
#pragma unroll
        for (int c = 0; c < kTilePixels / kWave; ++c) {
            const unsigned p = static_cast<unsigned>(c) * kWave + lane;
            const float4 g = pix[2 * p], cst = pix[2 * p + 1], k = ck[p];
            // A pixel that finished before this bucket never wrote its checkpoint (kf:436): that slot is uninitialised memory.
            // Such pixels are gated out of every update below, but the step body is branch-free (0 * NaN != 0), so they enter
            // the pipeline with a clean zero state instead of whatever the allocator left there.
            const bool live = __float_as_uint(cst.w) > tb * kBucket;
            s_init[wv][p] = live ? make_float4(cst.x - k.x, cst.y - k.y, cst.z - k.z, k.w) : make_float4(0.0f, 0.0f, 0.0f, 0.0f);   // kb:371-374
            if (!GLOBAL_GRAD) s_grad[wv][p] = g;
            s_last[wv][kWave + p] = __float_as_uint(cst.w);
        }
        s_last[wv][lane] = 0u;
        s_last[wv][kWave + kTilePixels + lane] = 0u;
    }

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
    wave_lds_fence();

    float d_mx = 0.0f, d_my = 0.0f, d_ca = 0.0f, d_cb = 0.0f, d_cc = 0.0f, d_op = 0.0f, d_c0 = 0.0f, d_c1 = 0.0f, d_c2 = 0.0f;
    float s0 = 0.0f, s1 = 0.0f, s2 = 0.0f, sT = 0.0f;                      // the pixel state travelling through the lanes
    const bool lane0 = lane == 0;
    uint64_t ever = 0;                   // lanes whose Gaussian passed the alpha test at least once (scalar registers)

    // software-pipelined LDS reads: the values of step i+1 are requested while step i computes. Lanes beyond the bucket's
    // Gaussians carry opacity 0 and fall out at the alpha test, so `tp < last` is the only per-step validity test.
    float4 init_next = s_init[wv][0];
    const uint32_t* my_last = &s_last[wv][kWave - lane];   // slot of pixel (i - lane) at step i is my_last[i]
    uint32_t last_next = my_last[0];
    const float4* __restrict__ gpix = a.pixrec + (size_t)tile * kTilePixels * 2;
    float4 g_next = GLOBAL_GRAD ? gpix[2 * min(max(0 - static_cast<int>(lane), 0), kTilePixels - 1)] : make_float4(0.0f, 0.0f, 0.0f, 0.0f);
    // The step body is branch-free (contributions are gated by selects, not by `continue`) and unrolled (the compiler turns
    // `unroll 2` of the 255 constant-trip steps into 85-step straight-line blocks): the only loop-carried dependence is the
    // 4-value pixel state, so the scheduler overlaps step i+1's exponent / alpha with step i's gradient arithmetic. Measured
    // in-process on MI355X (tools/ab_backward.py, S2): unrolled 0.69 ms, rolled 0.73 ms, dL/dC in LDS 0.73 ms, strip variant 0.85 ms.
#pragma unroll 2
    for (int i = 0; i < kTilePixels + kWave - 1; ++i) {
        // shift the 4 mutable values one lane up (kb:383-393) and inject pixel i at lane 0 (kb:401-410)
        s0 = wave_shift_up1(s0); s1 = wave_shift_up1(s1); s2 = wave_shift_up1(s2); sT = wave_shift_up1(sT);
        const float4 init = init_next;
        const uint32_t last = last_next;
        const float4 g_cur = g_next;
        init_next = s_init[wv][i + 1 < kTilePixels ? i + 1 : kTilePixels - 1];   // wave-uniform address: LDS broadcast
        last_next = my_last[i + 1];
        const int idx = i - static_cast<int>(lane);                        // pixel handled by this lane in this step
        if (GLOBAL_GRAD) g_next = gpix[2 * min(max(idx + 1, 0), kTilePixels - 1)];   // prefetched one step ahead
        s0 = lane0 ? init.x : s0; s1 = lane0 ? init.y : s1; s2 = lane0 ? init.z : s2; sT = lane0 ? init.w : sT;
        const float dx = mx - (x0 + static_cast<float>(idx & (kTileW - 1)));
        const float dy = my - (y0 + static_cast<float>(idx >> 4));
        const float power = -0.5f * (ca * dx * dx + cc * dy * dy) - cb * dx * dy;
        const float gauss_raw = __expf(fminf(power, 0.0f));
        const float alpha_raw = op * gauss_raw;
        const bool contrib = tp < last && alpha_raw >= kMinAlphaThreshold;                     // kb:412,419-421
        const uint64_t contrib_mask = wave_ballot(contrib);
        if (contrib_mask == 0) continue;                                                       // wave-uniform: nothing to do this step
        ever |= contrib_mask;
        const float alpha = contrib ? alpha_raw : 0.0f, gauss = contrib ? gauss_raw : 0.0f;
        const float4 g = GLOBAL_GRAD ? g_cur : s_grad[wv][min(max(idx, 0), kTilePixels - 1)];
        const float T = sT;
        const float w = T * alpha;
        d_c0 += w * g.x * f0; d_c1 += w * g.y * f1; d_c2 += w * g.z * f2;                      // kb:426-427
        s0 -= w * col0; s1 -= w * col1; s2 -= w * col2;                                        // kb:429
        const float oma = 1.0f - alpha;
        const float oma_rcp = fast_rcp(fmaxf(oma, kOneMinusAlphaEps));
        const float dl_dalpha = (T * col0 - s0 * oma_rcp) * g.x + (T * col1 - s1 * oma_rcp) * g.y
                                + (T * col2 - s2 * oma_rcp) * g.z + g.w * oma_rcp;             // kb:434-436
        d_op += gauss * dl_dalpha;
        const float h = -alpha * dl_dalpha;
        const float hh = 0.5f * h;
        d_ca += hh * (dx * dx); d_cb += hh * (dx * dy); d_cc += hh * (dy * dy);                // kb:443-448
        d_mx += h * (ca * dx + cb * dy); d_my += h * (cb * dx + cc * dy);                      // kb:449-453
        sT = T * oma;
    }

    // A Gaussian that never passed the alpha test in this tile has nine zero sums: adding them is a no-op, and the kernel's
    // tail is bound by atomic throughput on contended lines (near-camera Gaussians cover thousands of tiles).
    const bool silent = ((ever >> lane) & 1ull) == 0;
    if (valid_prim && !silent) {                                           // kb:459-470
        float* const rec = a.acc + (size_t)prim * kAccRecordWords;         // the Gaussian's record of nine consecutive floats
        unsafeAtomicAdd(rec, d_mx);
        unsafeAtomicAdd(rec + 1, d_my);
        unsafeAtomicAdd(rec + 2, d_ca);
        unsafeAtomicAdd(rec + 3, d_cb);
        unsafeAtomicAdd(rec + 4, d_cc);
        unsafeAtomicAdd(rec + 5, a.proper_aa ? d_op : op * (1.0f - op) * d_op);
        unsafeAtomicAdd(rec + 6, d_c0);
        unsafeAtomicAdd(rec + 7, d_c1);
        unsafeAtomicAdd(rec + 8, d_c2);
    }
}

// ---- variant 2: pixel-per-lane ("strip") formulation ---------------------------------------------------------------
// One 192-thread workgroup per bucket; wave w owns the 16x4 strip w of the tile, lane = pixel (state in registers, like the
// forward pass). The bucket's 64 Gaussians are staged in LDS; each wave culls them against its strip with one ballot and
// walks the survivors in depth order. Per surviving Gaussian the 9 per-pixel partial gradients are summed across the wave
// with 6 DPP-fused adds each (wave_sum_to_lane63) and lane 63 stores them into the wave's LDS slice; at the end 64 lanes
// add the three slices and issue the 9 global atomics (kb:459-470). Compared with the systolic form it touches only
// (Gaussian, strip) pairs whose bounding boxes overlap -- about 1.4 of 3 strips per Gaussian -- instead of all 192 pixels.
__global__ void __launch_bounds__(kTilePixels) blend_backward_strip_kernel(const BlendBackwardArgs a) {
    const unsigned bucket = blockIdx.x;
    const unsigned n_buckets = a.bucket_offsets[a.n_tiles - 1];
    if (bucket >= n_buckets) return;                                       // workgroup-uniform
    const unsigned tile = a.bucket_tile[bucket];
    const uint2 range = a.ranges[tile];
    const unsigned tile_n = range.y - range.x;
    const unsigned first = tile == 0 ? 0u : a.bucket_offsets[tile - 1];
    const unsigned tb = bucket - first;
    if (tb * kBucket >= a.max_n_processed[tile]) return;                   // kb:295, workgroup-uniform

    __shared__ float4 s_a[kBucket];                  // mean.x mean.y conic.a conic.b
    __shared__ float4 s_b[kBucket];                  // conic.c opacity colour.r colour.g (clamped)
    __shared__ float4 s_c[kBucket];                  // colour.b (clamped), clamp mask bits, bounds x, bounds y
    __shared__ float s_acc[kTilePixels / kWave][kBucket][12];   // per-wave slice: 9 sums per Gaussian (padded to 48 B)
    __shared__ uint32_t s_prim[kBucket];

    const unsigned tid = threadIdx.x, wave = tid >> 6, lane = tid & 63u;
    const unsigned n_here = min(static_cast<unsigned>(kBucket), tile_n - tb * kBucket);
    if (tid < kBucket) {
        float4 ga = make_float4(0.0f, 0.0f, 0.0f, 0.0f), gb = ga, gc = ga;
        uint32_t prim = 0;
        if (tid < n_here) {
            prim = a.inst_prims[range.x + tb * kBucket + tid];
            const float4* r = reinterpret_cast<const float4*>(a.rec + prim);
            const float4 r0 = r[0], r1 = r[1], r2 = r[2];
            const unsigned fm = (r1.z >= 0.0f ? 1u : 0u) | (r1.w >= 0.0f ? 2u : 0u) | (r2.x >= 0.0f ? 4u : 0u);   // kb:313-318
            ga = r0;
            gb = make_float4(r1.x, r1.y, fmaxf(r1.z, 0.0f), fmaxf(r1.w, 0.0f));
            gc = make_float4(fmaxf(r2.x, 0.0f), __uint_as_float(fm), r2.y, r2.z);
        }
        s_a[tid] = ga; s_b[tid] = gb; s_c[tid] = gc; s_prim[tid] = prim;
    }
    for (unsigned e = tid; e < (kTilePixels / kWave) * kBucket * 12; e += kTilePixels) (&s_acc[0][0][0])[e] = 0.0f;

    // this lane's pixel (same mapping as the forward pass) and its per-pixel constants / checkpointed state
    const unsigned tile_x = tile % a.grid_w, tile_y = tile / a.grid_w;
    const unsigned lx = (lane >> 5) * kSubtileW + (lane & 7u), ly = wave * kSubtileH + ((lane >> 3) & 3u);
    const unsigned local = ly * kTileW + lx;
    const float pxf = static_cast<float>(tile_x * kTileW + lx) + 0.5f, pyf = static_cast<float>(tile_y * kTileH + ly) + 0.5f;
    const float4 g = a.pixrec[((size_t)tile * kTilePixels + local) * 2];
    const float4 cst = a.pixrec[((size_t)tile * kTilePixels + local) * 2 + 1];
    const float4 ck = a.ckpt[(size_t)bucket * kTilePixels + local];
    float s0 = cst.x - ck.x, s1 = cst.y - ck.y, s2 = cst.z - ck.z, sT = ck.w;                  // kb:371-374
    const unsigned last = __float_as_uint(cst.w);                                               // 0 outside the image
    const unsigned strip_y0 = tile_y * kTileH + wave * kSubtileH, strip_y1 = strip_y0 + kSubtileH;
    const unsigned strip_x0 = tile_x * kTileW, strip_x1 = strip_x0 + kTileW;
    __syncthreads();

    bool overlaps = false;
    if (lane < n_here) {
        const uint32_t bx = __float_as_uint(s_c[lane].z), by = __float_as_uint(s_c[lane].w);
        overlaps = (bx & 0xffffu) < strip_x1 && strip_x0 < (bx >> 16) && (by & 0xffffu) < strip_y1 && strip_y0 < (by >> 16);
    }
    uint64_t pending = wave_ballot(overlaps);
    // Gaussians at or beyond every pixel's last contributor cannot contribute (kb:412): trim with the wave maximum
    const unsigned wave_last = wave_max(last);
    while (pending != 0) {                                                 // wave-uniform, depth order
        const int j = __ffsll(static_cast<unsigned long long>(pending)) - 1;
        pending &= pending - 1;
        const unsigned tp = tb * kBucket + static_cast<unsigned>(j);
        if (tp >= wave_last) break;
        const float4 ga = s_a[j], gb = s_b[j], gc = s_c[j];
        const float dx = ga.x - pxf, dy = ga.y - pyf;
        const float power = -0.5f * (ga.z * dx * dx + gb.x * dy * dy) - ga.w * dx * dy;
        const float gauss = __expf(fminf(power, 0.0f));
        const float alpha = gb.y * gauss;
        const bool contrib = tp < last && alpha >= kMinAlphaThreshold;
        if (wave_ballot(contrib) == 0) continue;
        float p_c0 = 0.0f, p_c1 = 0.0f, p_c2 = 0.0f, p_op = 0.0f, p_ca = 0.0f, p_cb = 0.0f, p_cc = 0.0f, p_mx = 0.0f, p_my = 0.0f;
        if (contrib) {
            const unsigned fm = __float_as_uint(gc.y);
            const float T = sT, w = T * alpha;
            p_c0 = (fm & 1u) ? w * g.x : 0.0f; p_c1 = (fm & 2u) ? w * g.y : 0.0f; p_c2 = (fm & 4u) ? w * g.z : 0.0f;   // kb:426-427
            s0 -= w * gb.z; s1 -= w * gb.w; s2 -= w * gc.x;                                                            // kb:429
            const float oma = 1.0f - alpha;
            const float oma_rcp = fast_rcp(fmaxf(oma, kOneMinusAlphaEps));
            const float dl_dalpha = (T * gb.z - s0 * oma_rcp) * g.x + (T * gb.w - s1 * oma_rcp) * g.y
                                    + (T * gc.x - s2 * oma_rcp) * g.z + g.w * oma_rcp;                                 // kb:434-436
            p_op = gauss * dl_dalpha;
            const float h = -alpha * dl_dalpha, hh = 0.5f * h;
            p_ca = hh * (dx * dx); p_cb = hh * (dx * dy); p_cc = hh * (dy * dy);                                      // kb:443-448
            p_mx = h * (ga.z * dx + ga.w * dy); p_my = h * (ga.w * dx + gb.x * dy);                                   // kb:449-453
            sT = T * oma;
        }
        p_mx = wave_sum_to_lane63(p_mx); p_my = wave_sum_to_lane63(p_my);
        p_ca = wave_sum_to_lane63(p_ca); p_cb = wave_sum_to_lane63(p_cb); p_cc = wave_sum_to_lane63(p_cc);
        p_op = wave_sum_to_lane63(p_op);
        p_c0 = wave_sum_to_lane63(p_c0); p_c1 = wave_sum_to_lane63(p_c1); p_c2 = wave_sum_to_lane63(p_c2);
        if (lane == 63u) {
            float4* dst = reinterpret_cast<float4*>(&s_acc[wave][j][0]);
            dst[0] = make_float4(p_mx, p_my, p_ca, p_cb);
            dst[1] = make_float4(p_cc, p_op, p_c0, p_c1);
            s_acc[wave][j][8] = p_c2;
        }
    }
    __syncthreads();

    if (tid < n_here) {                                                    // kb:459-470
        float t[9];
#pragma unroll
        for (int k = 0; k < 9; ++k) t[k] = s_acc[0][tid][k] + s_acc[1][tid][k] + s_acc[2][tid][k];
        const uint32_t prim = s_prim[tid];
        const float op = s_b[tid].y;
        float* const rec = a.acc + (size_t)prim * kAccRecordWords;         // the Gaussian's record of nine consecutive floats
        unsafeAtomicAdd(rec, t[0]);
        unsafeAtomicAdd(rec + 1, t[1]);
        unsafeAtomicAdd(rec + 2, t[2]);
        unsafeAtomicAdd(rec + 3, t[3]);
        unsafeAtomicAdd(rec + 4, t[4]);
        unsafeAtomicAdd(rec + 5, a.proper_aa ? t[5] : op * (1.0f - op) * t[5]);
        unsafeAtomicAdd(rec + 6, t[6]);
        unsafeAtomicAdd(rec + 7, t[7]);
        unsafeAtomicAdd(rec + 8, t[8]);
    }
}

#endif  // FGS_DEV_SWITCHES

#ifdef FGS_DEV_SWITCHES
#define FGS_ABLATE(a) ((a).ablate)      // timing experiments of the dev build (fgs_debug_set_option key 7)
#else
#define FGS_ABLATE(a) 0
#endif
// ---- variant 3 (default): work list of live buckets + compacted live pixels + two-value pipeline state -----------------
// Three observations about the systolic form above (rocprofv3 PMC, round 1: VALU-issue bound, ~70 instructions per step):
//  (1) 91 % of the launched buckets lie behind their tile's max_n_processed (kb:295) and exit after four dependent loads.
//      A one-workgroup planning pass turns (ranges, max_n_processed) into a dense list of LIVE (tile, bucket) pairs and its
//      length; the blend kernel walks that list (grid-stride), so no wave is ever launched for a dead bucket.
//  (2) inside a live bucket only pixels whose last contributor lies at or behind the bucket's first Gaussian can receive
//      anything (kb:412). The order in which pixels travel through the lanes is irrelevant, so staging COMPACTS the live
//      pixels (ballot + prefix count) and the pipeline runs n_live_pixels + 63 steps instead of 255. Per pixel the LDS holds
//      24 bytes: dL/dC (3 floats) + one packed word (x, y inside the tile and min(last - first Gaussian, 64) as three bytes,
//      each converted by a single v_cvt_f32_ubyteN) read by lane l at slot step - l, and the injected state (8 bytes).
//  (3) dL/dalpha only needs the remaining colour behind a Gaussian projected on the pixel's dL/dC (kb:434-436):
//          dL/dalpha = T (c . g) - (S - g_w) / (1 - alpha),   S = (C_final - T_final bg - C_front) . g
//      so the pixel state that shifts through the lanes is TWO scalars (T, S - g_w) instead of four, S is updated with one
//      FMA, and the per-Gaussian sums that are linear in per-lane constants are factored out of the loop: the colour-clamp
//      gate multiplies sum(w g) once at the end (kb:426-427), dL/dopacity = -2 sum(hh) / opacity because G = alpha / opacity
//      (kb:438), and dL/dmean2d = 2 [a b; b c] (sum(hh dx), sum(hh dy)) (kb:449-453) with hh = -alpha/2 dL/dalpha.
//      About 50 VALU instructions per step remain.
__global__ void __launch_bounds__(kTileScanThreads) plan_blend_backward_kernel(const BlendBackwardArgs a) {
    // One workgroup: exclusive scan of the live-bucket count of every tile (fgs_tile_scan.h: a thread sums 16 consecutive tiles, one DPP wave
    // scan, one barrier per 16 Ki tiles; the first version -- one barrier per 1024-tile chunk -- took 15 us at 1080p, all of it latency). live_offsets[tile] = first list slot of the tile; the entries themselves are written by stage_pixels_kernel
    // (a single workgroup writing 115 k entries took 58 us on the layered scene).
    __shared__ TileScanShared s_scan;
    const unsigned tid = threadIdx.x;
    uint32_t base = 0;                                        // live buckets in front of the current pass (uniform)
    int parity = 0;
    for (unsigned t0 = 0; t0 < a.n_tiles; t0 += kTileScanThreads * kTileScanPerThread, parity ^= 1) {
        uint32_t nl[kTileScanPerThread], ex[kTileScanPerThread];
        const unsigned first = t0 + tid * kTileScanPerThread;
        const bool whole = first + kTileScanPerThread <= a.n_tiles;
        if (whole) {
            const uint4* q = reinterpret_cast<const uint4*>(a.max_n_processed + first);        // 64 contiguous bytes
#pragma unroll
            for (int k = 0; k < kTileScanPerThread / 4; ++k) {
                const uint4 m = q[k];
                nl[4 * k] = (m.x + kBucket - 1) / kBucket; nl[4 * k + 1] = (m.y + kBucket - 1) / kBucket;   // live buckets of the tile (kb:295)
                nl[4 * k + 2] = (m.z + kBucket - 1) / kBucket; nl[4 * k + 3] = (m.w + kBucket - 1) / kBucket;
            }
        } else {
#pragma unroll
            for (int k = 0; k < kTileScanPerThread; ++k) nl[k] = first + k < a.n_tiles ? (a.max_n_processed[first + k] + kBucket - 1) / kBucket : 0u;
        }
        const uint32_t total = tile_scan_pass(nl, ex, s_scan, base, parity);
        if (whole) {
            uint4* o = reinterpret_cast<uint4*>(a.live_offsets + first);
#pragma unroll
            for (int k = 0; k < kTileScanPerThread / 4; ++k) o[k] = make_uint4(ex[4 * k], ex[4 * k + 1], ex[4 * k + 2], ex[4 * k + 3]);
        } else {
#pragma unroll
            for (int k = 0; k < kTileScanPerThread; ++k) if (first + k < a.n_tiles) a.live_offsets[first + k] = ex[k];
        }
        base += total;
    }
    if (tid == 0) *a.live_count = base;
}

// Debug-only timeline (tools/k11_timeline.sh builds a separate library with -DFGS_K11_TIMELINE; the product build has none of it): per work
// item its start / end timestamp, the number of pipeline steps and the hardware id of the wave -- concurrency over time, per-item durations,
// load per XCD / CU.
#ifdef FGS_K11_TIMELINE
constexpr unsigned kK11TimelineItems = 1u << 18;
__device__ unsigned long long g_k11_timeline[kK11TimelineItems * 4];
#endif
// Debug-only pair statistics (tools/pair_stats.sh builds a separate library with -DFGS_PAIR_STATS; the product build has none of it): how many
// of the (pixel, Gaussian) lane-steps K11 issues pass the alpha test -- [0] work items, [1] pipeline steps, [2] steps whose contribution block ran
// (at least one lane passed), [3] lane-steps with a real pixel in front of its last contributor, [4] lane-steps that passed the alpha test.
#ifdef FGS_PAIR_STATS
__device__ unsigned long long g_k11_pair_stats[8];
#endif
#ifndef FGS_K11_WAVES_PER_GROUP
#define FGS_K11_WAVES_PER_GROUP 1
#endif
constexpr unsigned kCompactWaves = FGS_K11_WAVES_PER_GROUP;      // independent waves (one work item each, no barrier) per workgroup
__global__ void __launch_bounds__(kWave * kCompactWaves) blend_backward_compact_kernel(const BlendBackwardArgs a) {
    const unsigned lane = threadIdx.x & 63u, wave_in_group = threadIdx.x >> 6;
    // Per live pixel: (dL/dC rgb, rel_last as a float) in s_pix and the pixel centre (x, y) in s_xy; slot n_px = dead sentinel (rel 0).
    // Until here x | y << 8 | rel << 16 were packed in the fourth float: three v_cvt_f32_ubyte (4.3 cycles each, tools/valu_rate.hip) and two
    // adds of the tile origin per (pixel, Gaussian) step -- 14 % of the loop's instructions. The centre is x0 + small integer either way
    // (exact), so dx, dy and every result are bit-identical. TWO dense arrays, not one 32-byte slot: lane l reads slot (step - l), and with a
    // 32-byte lane stride the 16-byte read conflicts every 8 lanes and the 8-byte read four-fold -- SQ_LDS_BANK_CONFLICT 2.3 M -> 42.6 M
    // cycles per launch at S2, LDS busy 34 M -> 86 M, which ate the whole gain (profiles/archive/r02_pmc_k11_lds.txt). 6.7 KB of LDS per wave
    // instead of 5.1: no effect on this kernel up to 8.2 KB (profiles/archive/r02_k11_occupancy.txt).
    // Round 3: the two arrays are RINGS of kRing = 256 slots (live pixels in [0, n_px), sentinels with rel 0 behind them): lane l reads slot
    // (step - l) mod 256, which is a sentinel both before the lane's first pixel arrives (negative -> 193..255) and after its last one has
    // passed (n_px .. n_px + 62 <= 254). The per-step index arithmetic was five vector instructions (step - lane, + look-ahead, unsigned min
    // against n_px, two shifts for the two strides); on the ring it is three: the byte offset into the 8-byte ring walks by 8 and wraps
    // (add, and), the 16-byte ring's offset is one shift-add of it. ONE shared block per wave with the 8-byte ring at offset 0, so that its
    // address IS the ring offset (no base to add): [xy ring 2 KB | inj 1.5 KB | pix ring 4 KB] = 7 680 bytes = six LDS allocation granules
    // (1 280 bytes on this part: with 8 KB the timeline showed 17 waves per CU in flight, with 6.7 KB before the ring 19).
    constexpr unsigned kRing = 256;
    constexpr unsigned kXyBytes = kRing * 8u, kInjBytes = kTilePixels * 8u, kPixBytes = kRing * 16u, kPixBase = kXyBytes + kInjBytes;
    static_assert(kPixBase % 16u == 0, "the 16-byte ring is 16-byte aligned");
    __shared__ __attribute__((aligned(16))) char s_block[kCompactWaves][kXyBytes + kInjBytes + kPixBytes];
    char* const s_base = s_block[wave_in_group];
    float2* const s_xy = reinterpret_cast<float2*>(s_base);
    float4* const s_pix = reinterpret_cast<float4*>(s_base + kPixBase);
    // T_ckpt, S - g_w of the live pixels: enters the pipeline at lane 0, which walks this array one slot per step -- and on into the bytes
    // behind it for the 63 (+ look-ahead) steps after the last pixel: what it reads there is never used, because every lane that the value
    // reaches sees a sentinel pixel (rel 0) in that step and contributes nothing, and the state registers are cleared per work item. Lanes
    // 1..63 read a zero in every step (slot 255 of the xy ring: always a sentinel), which makes "shift up by one lane, inject at lane 0" ONE
    // DPP-fused add per value (shifted-in zero at lane 0 + the lane's own read) instead of a DPP move plus a select.
    float2* const s_inj = reinterpret_cast<float2*>(s_base + kXyBytes);
    const unsigned n_live = *a.live_count;
    const float lane_f = static_cast<float>(lane);
    const bool lane0 = lane == 0;
    // (round 6, measured and withdrawn: XCD x walking a contiguous eighth of the live list, so that the buckets of neighbouring tiles share an L2 --
    // K11 0.313 -> 0.319 ms at S2, layered scene 3.73 -> 3.80 ms: the kernel is bound by vector issue, and the bands unbalance the XCDs. profiles/r06_ab_k11_xcd_bands.txt)
    for (unsigned item = blockIdx.x * kCompactWaves + wave_in_group; item < n_live; item += gridDim.x * kCompactWaves) {            // wave-uniform
#ifdef FGS_K11_TIMELINE
        const unsigned long long t_start_ = __builtin_amdgcn_s_memrealtime();       // 100 MHz, the same clock on every CU (the cycle counter is not)
        const unsigned long long c_start_ = __builtin_readcyclecounter();
#endif
        const uint2 work = a.work_list[item];
        const unsigned tile = work.x, tb = work.y;
        const uint2 range = a.ranges[tile];
        const unsigned tile_n = range.y - range.x;
        const unsigned bucket = (tile == 0 ? 0u : a.bucket_offsets[tile - 1]) + tb;
        const unsigned first_gaussian = tb * kBucket;

        const float x0 = static_cast<float>((tile % a.grid_w) * kTileW) + 0.5f;
        const float y0 = static_cast<float>((tile / a.grid_w) * kTileH) + 0.5f;
        // ---- stage the live pixels, compacted (kb:349-380) ----
        unsigned n_px = 0;
#ifdef FGS_PAIR_STATS
        uint64_t st_live[kTilePixels / kWave] = {};
#endif
        {
            const float4* __restrict__ pix = a.pixrec + (size_t)tile * kTilePixels * 2;
            const float4* __restrict__ ck = a.ckpt + (size_t)bucket * kTilePixels;
            float4 g[kTilePixels / kWave], cst[kTilePixels / kWave], k[kTilePixels / kWave];
#pragma unroll
            for (int c = 0; c < kTilePixels / kWave; ++c) {                        // all nine loads in flight together
                const unsigned p = static_cast<unsigned>(c) * kWave + lane;
                g[c] = pix[2 * p]; cst[c] = pix[2 * p + 1];
#if FGS_CKPT_NT
                k[c] = load_float4_nt(reinterpret_cast<const float*>(ck + p));
#else
                k[c] = ck[p];
#endif
            }
#pragma unroll
            for (int c = 0; c < kTilePixels / kWave; ++c) {
                const unsigned p = static_cast<unsigned>(c) * kWave + lane;
                const unsigned last = __float_as_uint(cst[c].w);
                // a pixel that finished before this bucket never wrote its checkpoint (kf:436) and receives nothing here
                const bool live = last > first_gaussian;
                const uint64_t m = wave_ballot(live);
#ifdef FGS_PAIR_STATS
                st_live[c] = m;
#endif
                if (live) {
                    const unsigned slot = n_px + lanes_below(m);
                    const unsigned rel = min(last - first_gaussian, static_cast<unsigned>(kBucket));
                    s_pix[slot] = make_float4(g[c].x, g[c].y, g[c].z, static_cast<float>(rel));
                    s_xy[slot] = make_float2(x0 + static_cast<float>(p & (kTileW - 1)), y0 + static_cast<float>(p / kTileW));
                    const float S = (cst[c].x - k[c].x) * g[c].x + (cst[c].y - k[c].y) * g[c].y + (cst[c].z - k[c].z) * g[c].z;   // kb:371-374
                    s_inj[slot] = make_float2(k[c].w, S - g[c].w);
                }
                n_px += static_cast<unsigned>(__popcll(m));
            }
            for (unsigned sl = n_px + lane; sl < kRing; sl += kWave) {          // sentinels (rel_last 0: never contributes): 1 to 4 rounds
                s_pix[sl] = make_float4(0.0f, 0.0f, 0.0f, 0.0f); s_xy[sl] = make_float2(0.0f, 0.0f);
            }

        }

        const unsigned tp = first_gaussian + lane;
        const bool valid_prim = tp < tile_n;
        uint32_t prim = 0;
        float mx = 0.0f, my = 0.0f, ca = 0.0f, cb = 0.0f, cc = 0.0f, op = 0.0f;
        float col0 = 0.0f, col1 = 0.0f, col2 = 0.0f, f0 = 0.0f, f1 = 0.0f, f2 = 0.0f;
        unsigned footprint = 0;               // candidate tiles of this lane's Gaussian
        uint32_t hot_slot_word = 0;
        if (valid_prim) {
            prim = a.inst_prims[range.x + tp];
            const float4* r = reinterpret_cast<const float4*>(a.rec + prim);
            const float4 r0 = r[0], r1 = r[1];
            const float4 r2 = r[2];
            const float raw2 = r2.x;
            mx = r0.x; my = r0.y; ca = r0.z; cb = r0.w; cc = r1.x; op = r1.y;
            col0 = fmaxf(r1.z, 0.0f); col1 = fmaxf(r1.w, 0.0f); col2 = fmaxf(raw2, 0.0f);
            f0 = r1.z >= 0.0f ? 1.0f : 0.0f; f1 = r1.w >= 0.0f ? 1.0f : 0.0f; f2 = raw2 >= 0.0f ? 1.0f : 0.0f;   // kb:313-318
            unsigned tx0, tx1, ty0, ty1;
            tile_rect(__float_as_uint(r2.y), __float_as_uint(r2.z), tx0, tx1, ty0, ty1);
            footprint = (tx1 - tx0) * (ty1 - ty0);
            hot_slot_word = __float_as_uint(r2.w);
        }
#ifdef FGS_PAIR_STATS
        unsigned st_trim = 0;                                                    // [5]: live pixels outside the union of the bucket's screen bounds
        {
            unsigned bx = 0xffffu, by = 0xffffu;                                  // x_min | x_max << 16: an empty box for lanes without a Gaussian
            if (valid_prim) { const float4 r2q = reinterpret_cast<const float4*>(a.rec + prim)[2]; bx = __float_as_uint(r2q.y); by = __float_as_uint(r2q.z); }
            const unsigned ux0 = 0xffffu - wave_max(0xffffu - (bx & 0xffffu)), ux1 = wave_max(bx >> 16);
            const unsigned uy0 = 0xffffu - wave_max(0xffffu - (by & 0xffffu)), uy1 = wave_max(by >> 16);
            const unsigned tx_px = (tile % a.grid_w) * kTileW, ty_px = (tile / a.grid_w) * kTileH;
#pragma unroll
            for (int c = 0; c < kTilePixels / kWave; ++c) {
                const unsigned p = static_cast<unsigned>(c) * kWave + lane;
                const unsigned px_ = tx_px + (p & (kTileW - 1)), py_ = ty_px + p / kTileW;
                const bool inside = px_ >= ux0 && px_ < ux1 && py_ >= uy0 && py_ < uy1;
                st_trim += static_cast<unsigned>(__popcll(st_live[c] & wave_ballot(!inside)));
            }
        }
#endif
        wave_lds_fence();
        // alpha is recomputed with the FORWARD kernel's expression, operation for operation (kf:455-466 / kb:415-418; blend_forward.hip): the
        // backward pass replays the forward pass's alpha bit for bit, so both passes agree on every alpha >= 1/255 decision and the
        // transmittance rebuilt here is the one the forward pass used. (Round 3 first folded log2(e) and the -1/2 into per-lane constants --
        // five instructions and a bare v_exp_f32 instead of eight: 1 % of this kernel -- at the price of an alpha that differed from the
        // forward pass's by a rounding; a wide fuzz sweep in the simulator then showed a (pixel, Gaussian) pair blended by one pass and
        // skipped by the other.)
        float a_c0 = 0.0f, a_c1 = 0.0f, a_c2 = 0.0f;                 // sum w g_c               (kb:426-427 without the clamp gate)
        float a_h = 0.0f, a_x = 0.0f, a_y = 0.0f;                     // sum hh, sum hh dx, sum hh dy
        float a_xx = 0.0f, a_xy = 0.0f, a_yy = 0.0f;                  // sum hh dx^2, hh dx dy, hh dy^2   (kb:443-448)
        float sT = 0.0f, sS = 0.0f;                                   // the pixel state travelling through the lanes

        // One pipeline step. `inj` / `px`: what this lane read for THIS step (software pipelined: the reads of the following step
        // are issued first). Contributions are made under the lane mask of `contrib` (EXEC), not by selects: a v_cndmask costs
        // several times an FMA on this chip (tools/valu_rate.hip), and the empty-mask branch of the `if` is the wave-uniform skip.
        unsigned inj_at = lane0 ? kXyBytes : (kRing - 1u) * 8u;       // byte offsets: lane 0 the slot of the step, other lanes the zero slot
        const unsigned inj_step = lane0 ? 8u : 0u;
        // byte offset into the 8-byte ring of the slot this lane reads for the step whose reads are issued next: (step - lane) mod 256
        unsigned ring_at = ((0u - lane) & (kRing - 1u)) * 8u;
        struct PixRead { float4 g; float2 xy; };
        auto read_inj = [&]() { const float2 v = *reinterpret_cast<const float2*>(s_base + inj_at); inj_at += inj_step; return v; };
        auto read_pix = [&]() {
            PixRead r;
            r.xy = *reinterpret_cast<const float2*>(s_base + ring_at);
            r.g = *reinterpret_cast<const float4*>(s_base + (2u * ring_at + kPixBase));
            ring_at = (ring_at + 8u) & (kXyBytes - 1u);
            return r;
        };
#ifdef FGS_PAIR_STATS
        unsigned st_steps = 0, st_body = 0, st_elig = 0, st_pass = 0;
#endif
        auto step = [&](const float2 inj, const PixRead pr) {
            sT = wave_shift_up1_zero(sT) + inj.x;                                               // kb:383-410
            sS = wave_shift_up1_zero(sS) + inj.y;
            const float4 px = pr.g;
            const float rel = px.w;
            const float dx = mx - pr.xy.x, dy = my - pr.xy.y;
            const float power = -0.5f * (ca * dx * dx + cc * dy * dy) - cb * dx * dy;
            const float alpha = op * __expf(fminf(power, 0.0f));
#ifdef FGS_PAIR_STATS
            {
                const uint64_t me_ = wave_ballot(lane_f < rel), mp_ = wave_ballot(lane_f < rel && alpha >= kMinAlphaThreshold);
                st_steps += 1u; st_body += mp_ != 0 ? 1u : 0u;
                st_elig += static_cast<unsigned>(__popcll(me_)); st_pass += static_cast<unsigned>(__popcll(mp_));
            }
#endif
            if (lane_f < rel && alpha >= kMinAlphaThreshold) {                                  // kb:412,419-421
                const float T = sT;
                const float w = T * alpha;
                a_c0 += w * px.x; a_c1 += w * px.y; a_c2 += w * px.z;
                const float cg = col0 * px.x + col1 * px.y + col2 * px.z;
                sS -= w * cg;                                                                    // kb:429 projected on dL/dC
                const float oma = 1.0f - alpha;
                const float oma_rcp = fast_rcp(fmaxf(oma, kOneMinusAlphaEps));
                const float dl_dalpha = T * cg - sS * oma_rcp;                                   // kb:434-436
                const float hh = (-0.5f * alpha) * dl_dalpha;
                const float t = hh * dx, u = hh * dy;
                a_h += hh; a_x += t; a_y += u;
                a_xx += t * dx; a_xy += t * dy; a_yy += u * dy;
                sT = T * oma;
            }
        };
        // two steps per trip with the read registers ping-ponged (no copies); an odd step count runs one extra step in which every
        // lane sees the sentinel
        float2 inj_a = read_inj(), inj_b;
        PixRead pix_a = read_pix(), pix_b;
        const int n_steps = (FGS_ABLATE(a) & 2) ? 0 : static_cast<int>(n_px) + kWave - 1;
        for (int i = 0; i < n_steps; i += 2) {
            inj_b = read_inj(); pix_b = read_pix();
            step(inj_a, pix_a);
            inj_a = read_inj(); pix_a = read_pix();
            step(inj_b, pix_b);
        }

        // A Gaussian whose nine sums are all zero has nothing to add (it never passed the alpha test, or only at pixels with a zero
        // image gradient).
        const bool silent = a_h == 0.0f && a_c0 == 0.0f && a_c1 == 0.0f && a_c2 == 0.0f && a_x == 0.0f && a_y == 0.0f
                            && a_xx == 0.0f && a_xy == 0.0f && a_yy == 0.0f;
        if (!(FGS_ABLATE(a) & 1)) {                                                              // kb:459-470
            // The nine sums of a Gaussian leave as its RECORD of nine consecutive floats, SEVEN Gaussians per atomic instruction (lane l of
            // instruction k adds word 63 k + l of the bucket's 64 x 9 block, transposed through the LDS the rings no longer need). Round 4: the
            // memory pipeline merges the lanes of one atomic instruction that fall into one 128-byte line and is bound by LINE requests, about
            // 20 000 per microsecond whatever they carry (tools/atomic_rate.hip: 64 scattered floats 20.8 k atomics / us, 64 consecutive ones 278 k).
            // With one plane per sum (rounds 1-3) an item cost 9 instructions x as many lines as its 64 primitives are scattered over: hidden
            // behind the arithmetic on a Morton-ordered synthetic scene, HALF of this kernel's time on a model trained under the MCMC policy
            // (1.56 -> 0.81 ms, profiles/r04_k11_record_atomics.txt). Now: ~1.3 lines per Gaussian, 10 instructions per item.
            // dL/dopacity = sum G dL/dalpha with G = alpha / opacity; through the sigmoid unless proper antialiasing (kb:462-466)
            const float v5 = a.proper_aa ? -2.0f * a_h / op : -2.0f * a_h * (1.0f - op);
            const float v0 = 2.0f * (ca * a_x + cb * a_y), v1 = 2.0f * (cb * a_x + cc * a_y);
            const uint32_t hot_word = footprint > kHotFootprint ? hot_slot_word : 0u;
            const bool adds = valid_prim && !silent;
            // float offset of the record from a.acc: the Gaussian's own record, or -- a hot Gaussian (fgs_config.h) -- the record of its slot in the
            // tile's replica (acc_hot follows acc in the scratch blob: [kHotReplicas][kMaxHot][9])
            const uint32_t rec_off = hot_word != 0u ? static_cast<uint32_t>(a.acc_hot - a.acc) + ((tile % kHotReplicas) * kMaxHot + (hot_word - 1u)) * kAccRecordWords
                                                    : prim * kAccRecordWords;
            float* const s_t = reinterpret_cast<float*>(s_base + kPixBase);                      // [64][9] in the dead 16-byte ring
            uint32_t* const s_off = reinterpret_cast<uint32_t*>(s_base);                         // record offset of each lane's Gaussian, or "nothing to add"
            constexpr uint32_t kNoRecord = 0xffffffffu;
            wave_lds_fence();                                                                     // the loop's last ring reads are done
            float* const mine = s_t + lane * kAccRecordWords;
            mine[0] = v0; mine[1] = v1; mine[2] = a_xx; mine[3] = a_xy; mine[4] = a_yy; mine[5] = v5;
            mine[6] = a_c0 * f0; mine[7] = a_c1 * f1; mine[8] = a_c2 * f2;
            s_off[lane] = adds ? rec_off : kNoRecord;
            wave_lds_fence();
            const unsigned sub = lane / kAccRecordWords, comp = lane - sub * kAccRecordWords;     // lane 63 idles: 7 records of 9 words per instruction
#pragma unroll
            for (unsigned k = 0; k < (kBucket + 6u) / 7u; ++k) {
                const unsigned gsn = 7u * k + sub;
                if (lane < 63u && gsn < static_cast<unsigned>(kBucket)) {
                    const uint32_t rec = s_off[gsn];
                    if (rec != kNoRecord) unsafeAtomicAdd(a.acc + (size_t)rec + comp, s_t[63u * k + lane]);
                }
            }
        }
#ifdef FGS_PAIR_STATS
        if (lane == 0) {
            atomicAdd(&g_k11_pair_stats[0], 1ull); atomicAdd(&g_k11_pair_stats[1], static_cast<unsigned long long>(st_steps));
            atomicAdd(&g_k11_pair_stats[2], static_cast<unsigned long long>(st_body)); atomicAdd(&g_k11_pair_stats[3], static_cast<unsigned long long>(st_elig));
            atomicAdd(&g_k11_pair_stats[4], static_cast<unsigned long long>(st_pass)); atomicAdd(&g_k11_pair_stats[5], static_cast<unsigned long long>(st_trim));
        }
#endif
#ifdef FGS_K11_TIMELINE
        if (lane == 0 && item < kK11TimelineItems) {
            g_k11_timeline[item * 4u] = t_start_;
            g_k11_timeline[item * 4u + 1u] = __builtin_amdgcn_s_memrealtime();
            g_k11_timeline[item * 4u + 2u] = static_cast<unsigned long long>(n_steps) | ((__builtin_readcyclecounter() - c_start_) << 16);
            g_k11_timeline[item * 4u + 3u] = (static_cast<unsigned long long>(__builtin_amdgcn_s_getreg((4 << 0) | (0 << 6) | (31 << 11))) << 32)   // HW_ID, all 32 bits
                                            | static_cast<unsigned long long>(__builtin_amdgcn_s_getreg((20 << 0) | (0 << 6) | (31 << 11)));            // XCC_ID (gfx94x+)
        }
#endif
        wave_lds_fence();                                  // the next item restages the LDS slices
    }
}

#ifdef FGS_K11_TIMELINE
}  // namespace fgs
extern "C" __attribute__((visibility("default"))) int fgs_debug_k11_timeline(unsigned long long* out, unsigned n_items, int reset) {
    if (n_items > fgs::kK11TimelineItems) n_items = fgs::kK11TimelineItems;
    if (out != nullptr && hipMemcpyFromSymbol(out, HIP_SYMBOL(fgs::g_k11_timeline), sizeof(unsigned long long) * 4 * n_items) != hipSuccess) return -1;
    if (reset) {
        void* dev = nullptr;
        if (hipGetSymbolAddress(&dev, HIP_SYMBOL(fgs::g_k11_timeline)) != hipSuccess
            || hipMemset(dev, 0, sizeof(unsigned long long) * 4 * fgs::kK11TimelineItems) != hipSuccess) return -1;
    }
    return 0;
}
namespace fgs {
#endif


#ifdef FGS_DEV_SWITCHES   // A/B exhibit (variant 5, round 6): libfgs_hip_dev.so only -- parity-green, and SLOWER than the kernel above: see the end of this comment
// ---- K11, chained (round 6): the items of a wave follow each other through the lanes without draining ------------------------------------------
// The kernel above runs one item -- (tile, 64-Gaussian bucket) -- at a time: n_px live pixels stream through the 64 lanes in n_px + 63 steps, and
// 63 of the ~240 steps are fill / drain (27 %; removing the drain alone measured -18 % at S2 and -22 % on the layered scene,
// profiles/r06_k11_chain_ceiling.txt). Here a wave owns a CHAIN of items of the live list (every G-th, G = waves launched) and their pixels form ONE stream of positions:
// item i occupies positions [S_i, S_i + n_i), followed by sentinel positions (rel = 0: never contribute) up to S_{i+1} = S_i + L_i with
// L_i = max(round_up_8(n_i + 8), 64). Lane l handles position s - l at step s, so while the tail of item i still travels through the upper lanes the
// lower lanes already work on item i + 1. A lane changes Gaussians when the boundary passes it -- in GROUPS of eight lanes: the >= 8 sentinels in
// front of every boundary mean that at step S_{i+1} + 8 g - 1 all eight lanes of group g look at sentinels, so the group flushes its nine sums of
// item i (72 floats through LDS, two atomic instructions) and takes the parameters of item i + 1 from a shadow register set between two steps;
// L_i >= 64 keeps the eight group switches of one boundary apart from those of the next. Cost per item: n_i + 8..15 steps + eight switches of
// ~60 instructions + the wave's single fill / drain spread over its chain, against n_i + 63.
// LDS must not grow (12.5 KB per wave instead of 7.7 costs this kernel 10 %, same file), so the rings keep their 256 slots = the 64 positions in
// flight + at most 192 staged ahead: the stream is staged in UNITS of one 64-pixel third of an item (its raw records prefetched into registers a
// unit ahead) whenever fewer than 16 positions lie in front of lane 0, into the slots the tail has left.
// MEASURED (profiles/r06_ab_k11_chained.txt, r06_k11_variants_pmc.txt): correct on the simulator and on the MI355X (all parity suites), and never faster than
// the kernel above. A first version defined its prefetch registers inside the loop that runs the steps: the register allocator copies such registers at
// the loop's back edge and the copy waits for ALL outstanding loads (`s_waitcnt vmcnt(0)` in front of every block of eight steps): S2 0.354 vs 0.310 ms,
// layered scene 1.74 vs 1.47. This version issues every load at the outer level of a loop nest (104 registers, no wait left in the step loop): it executes
// 13 % fewer vector instructions than the kernel above and still takes 0.344 / 1.66 ms with 4096 waves -- on average 12 waves per CU are resident instead
// of 19 (16 fit; static chains end at different times), the bookkeeping per block of eight steps adds 60 % scalar instructions, the group switches their
// share. With 16 384 waves (shorter chains) it reaches the kernel above on the layered scene (1.46 ms) and stays behind it at S2 (0.342 vs 0.317).
// Kept as an exhibit of the dev library (variant 5) with its tests; the product's K11 stays the kernel above.
// Which items: wave w chains the items w, w + G, w + 2 G, ... of the live list, G = waves launched = g_k11_chain_waves (fgs_kernels.h: 4096 = 16 resident
// waves x 256 CUs at 110 registers and 8.5 KB of LDS per wave, so every wave of the launch runs from the first cycle): no queue, no atomics, +-1 item of
// imbalance. A first version gave each wave 8 CONSECUTIVE items: 2 770 waves at S2, two thirds of the chip, 0.46 ms instead of 0.32
// (profiles/r06_ab_k11_chained.txt).
constexpr unsigned kChDesc = 32;                            // descriptor window: lane q mod 32 holds item q of the wave's chain, refilled 16 at a time
constexpr unsigned kChRing = 256, kChXyBytes = kChRing * 8u, kChInjBase = kChXyBytes, kChPixBase = 2u * kChXyBytes, kChFlushBase = kChPixBase + kChRing * 16u;
constexpr unsigned kChOffBase = kChFlushBase + 8u * kAccRecordWords * 4u, kChZeroBase = kChOffBase + 8u * 4u, kChBytes = kChZeroBase + 16u;
#ifndef FGS_K11_CHAIN_WAVES_PER_SIMD
#define FGS_K11_CHAIN_WAVES_PER_SIMD 4      // register budget 128: without the cap the compiler takes 143 (3 waves per SIMD, 12 per CU: K11 loses 10 % there)
#endif
__global__ void __launch_bounds__(kWave) __attribute__((amdgpu_waves_per_eu(FGS_K11_CHAIN_WAVES_PER_SIMD, FGS_K11_CHAIN_WAVES_PER_SIMD)))
blend_backward_chained_kernel(const BlendBackwardArgs a) {
    __shared__ __attribute__((aligned(16))) char s_base[kChBytes];
    const unsigned lane = lane_id();
    const float lane_f = static_cast<float>(lane);
    const unsigned n_live = *a.live_count;
    const unsigned first_item = blockIdx.x, stride = gridDim.x;
    if (first_item >= n_live) return;                                                     // wave-uniform
    const unsigned n_items = wave_uniform((n_live - first_item + stride - 1u) / stride);  // this wave's chain: items first_item + q * stride, q < n_items
    float2* const s_xy = reinterpret_cast<float2*>(s_base);
    float2* const s_inj = reinterpret_cast<float2*>(s_base + kChInjBase);
    float4* const s_pix = reinterpret_cast<float4*>(s_base + kChPixBase);
    float* const s_flush = reinterpret_cast<float*>(s_base + kChFlushBase);
    uint32_t* const s_off = reinterpret_cast<uint32_t*>(s_base + kChOffBase);
    if (lane == 0) *reinterpret_cast<float2*>(s_base + kChZeroBase) = make_float2(0.0f, 0.0f);
    // positions -64 .. -1 (what the lanes above lane 0 look at until the stream reaches them): sentinels
    s_pix[kChRing - kWave + lane] = make_float4(0.0f, 0.0f, 0.0f, 0.0f); s_xy[kChRing - kWave + lane] = make_float2(0.0f, 0.0f);

    // ---- descriptors: lane j holds item j (tile, bucket in tile, list start, list length, checkpoint row) and, later, its first position ----
    unsigned d_tile = 0, d_tb = 0, d_rx = 0, d_n = 0, d_bucket = 0, d_start = 0;
    auto load_descriptors = [&](const unsigned q0) {                                       // items q0 .. q0 + 15 of the chain (q0 a multiple of 16) -> lanes (q0 & 31) ..
        const unsigned base = q0 & (kChDesc - 1u), q = q0 + (lane - base);
        if (lane >= base && lane < base + 16u && q < n_items) {
            const uint2 work = a.work_list[first_item + q * stride];
            d_tile = work.x; d_tb = work.y;
            const uint2 range = a.ranges[d_tile];
            d_rx = range.x; d_n = range.y - range.x;
            d_bucket = (d_tile == 0 ? 0u : a.bucket_offsets[d_tile - 1]) + d_tb;
        }
    };
    load_descriptors(0u);
    load_descriptors(16u);
    auto slot_of = [&](const unsigned q) { return static_cast<int>(wave_uniform(q & (kChDesc - 1u))); };

    // ---- the lane's Gaussian: ACTIVE set (what the steps use) and SHADOW set (the next item's, requested early) ----
    uint32_t prim = 0, hot_word = 0, rep_tile = 0;
    bool have = false;
    float mx = 0.0f, my = 0.0f, ca = 0.0f, cb = 0.0f, cc = 0.0f, op = 0.0f;           // op = 0: alpha = 0, nothing passes the test (no item yet / after the last)
    float col0 = 0.0f, col1 = 0.0f, col2 = 0.0f, f0 = 0.0f, f1 = 0.0f, f2 = 0.0f;
    // The SHADOW set: the next item's Gaussian, eleven registers per lane. All loads of this kernel are issued at the OUTER level of the loop nest below:
    // a register that a load defines inside the loop that also runs the steps is copied at that loop's back edge, and the copy makes the compiler
    // wait for every outstanding load in front of every block of steps (the first version of this kernel: profiles/r06_ab_k11_chained.txt).
    struct Shadow { uint32_t prim, tile, hot; bool valid; float mx, my, ca, cb, cc, op, c0, c1, c2; };
    Shadow sh{0u, 0u, 0u, false, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f};
    uint32_t pf_prim = 0;
    bool pf_valid = false;
    auto prefetch_prim = [&](const unsigned j) {                                         // the primitive index of item j's Gaussian for this lane
        pf_valid = false; pf_prim = 0;
        if (j < n_items) {
            const int jj = slot_of(j);
            const unsigned tp = wave_read(d_tb, jj) * kBucket + lane, list_n = wave_read(d_n, jj), list_first = wave_read(d_rx, jj);   // (convergent: all lanes)
            pf_valid = tp < list_n;
            if (pf_valid) pf_prim = a.inst_prims[list_first + tp];
        }
    };
    auto load_shadow = [&](const unsigned j) {                                           // item j's record (j == n_items: the empty item behind the last)
        sh.valid = pf_valid; sh.prim = pf_prim;
        sh.tile = j < n_items ? wave_read(d_tile, slot_of(j)) : 0u;
        float4 r0 = make_float4(0.0f, 0.0f, 0.0f, 0.0f), r1 = r0, r2 = r0;
        if (sh.valid) { const float4* r = reinterpret_cast<const float4*>(a.rec + sh.prim); r0 = r[0]; r1 = r[1]; r2 = r[2]; }
        sh.mx = r0.x; sh.my = r0.y; sh.ca = r0.z; sh.cb = r0.w; sh.cc = r1.x; sh.op = r1.y; sh.c0 = r1.z; sh.c1 = r1.w; sh.c2 = r2.x;
        unsigned tx0, tx1, ty0, ty1;
        tile_rect(__float_as_uint(r2.y), __float_as_uint(r2.z), tx0, tx1, ty0, ty1);
        sh.hot = (sh.valid && (tx1 - tx0) * (ty1 - ty0) > kHotFootprint) ? __float_as_uint(r2.w) : 0u;
        prefetch_prim(j + 1u);
    };

    float a_c0 = 0.0f, a_c1 = 0.0f, a_c2 = 0.0f, a_h = 0.0f, a_x = 0.0f, a_y = 0.0f, a_xx = 0.0f, a_xy = 0.0f, a_yy = 0.0f;
    float sT = 0.0f, sS = 0.0f;

    // group g (lanes 8 g .. 8 g + 7) leaves its current item (its nine sums go out, kb:459-470) and takes the shadow set
    constexpr uint32_t kNoRecord = 0xffffffffu;
    auto switch_group = [&](const unsigned g, const bool flush) {
        const bool mine = (lane >> 3) == g;
        if (flush) {
            const bool silent = a_h == 0.0f && a_c0 == 0.0f && a_c1 == 0.0f && a_c2 == 0.0f && a_x == 0.0f && a_y == 0.0f
                                && a_xx == 0.0f && a_xy == 0.0f && a_yy == 0.0f;
            const float v5 = a.proper_aa ? -2.0f * a_h / op : -2.0f * a_h * (1.0f - op);
            const float v0 = 2.0f * (ca * a_x + cb * a_y), v1 = 2.0f * (cb * a_x + cc * a_y);
            const uint32_t rec_off = hot_word != 0u ? static_cast<uint32_t>(a.acc_hot - a.acc) + ((rep_tile % kHotReplicas) * kMaxHot + (hot_word - 1u)) * kAccRecordWords
                                                    : prim * kAccRecordWords;
            if (mine) {
                float* const out = s_flush + (lane & 7u) * kAccRecordWords;
                out[0] = v0; out[1] = v1; out[2] = a_xx; out[3] = a_xy; out[4] = a_yy; out[5] = v5;
                out[6] = a_c0 * f0; out[7] = a_c1 * f1; out[8] = a_c2 * f2;
                s_off[lane & 7u] = (have && !silent) ? rec_off : kNoRecord;
            }
            wave_lds_fence();
            if (lane < 63u) {                                                              // seven records of nine words
                const uint32_t rec = s_off[lane / kAccRecordWords];
                if (rec != kNoRecord) unsafeAtomicAdd(a.acc + (size_t)rec + (lane % kAccRecordWords), s_flush[lane]);
            }
            if (lane < kAccRecordWords) {                                                  // the eighth
                const uint32_t rec = s_off[7];
                if (rec != kNoRecord) unsafeAtomicAdd(a.acc + (size_t)rec + lane, s_flush[63u + lane]);
            }
            wave_lds_fence();
        }
        if (mine) {
            prim = sh.prim; have = sh.valid; rep_tile = sh.tile; hot_word = sh.hot;
            mx = sh.mx; my = sh.my; ca = sh.ca; cb = sh.cb; cc = sh.cc; op = sh.op;
            col0 = fmaxf(sh.c0, 0.0f); col1 = fmaxf(sh.c1, 0.0f); col2 = fmaxf(sh.c2, 0.0f);
            f0 = sh.c0 >= 0.0f ? 1.0f : 0.0f; f1 = sh.c1 >= 0.0f ? 1.0f : 0.0f; f2 = sh.c2 >= 0.0f ? 1.0f : 0.0f;   // kb:313-318
            a_c0 = a_c1 = a_c2 = a_h = a_x = a_y = a_xx = a_xy = a_yy = 0.0f;
        }
    };

    // ---- staging cursor: the next UNIT = third `st_chunk` of item `st_item`, its raw records in (rg, rc, rk) ----
    unsigned st_item = 0, st_chunk = 0, st_pos = 0, st_n = 0;       // uniform; st_pos = first position not staged yet, st_n = live pixels of the item so far
    bool st_done = false;
    float4 rg = make_float4(0.0f, 0.0f, 0.0f, 0.0f), rc = rg, rk = rg;
    auto fetch_unit = [&]() {                                                             // requests the cursor's unit (nothing waits for it here)
        if (st_item >= n_items) return;
        const unsigned tile = wave_read(d_tile, slot_of(st_item)), bucket = wave_read(d_bucket, slot_of(st_item));
        const unsigned p = st_chunk * kWave + lane;
        const float4* __restrict__ pix = a.pixrec + (size_t)tile * kTilePixels * 2;
        rg = pix[2 * p]; rc = pix[2 * p + 1];
#if FGS_CKPT_NT
        rk = load_float4_nt(reinterpret_cast<const float*>(a.ckpt + (size_t)bucket * kTilePixels + p));
#else
        rk = a.ckpt[(size_t)bucket * kTilePixels + p];
#endif
    };
    auto put_sentinels = [&](const unsigned count) {                                      // `count` <= 64 positions from st_pos on that never contribute
        if (lane < count) {
            const unsigned slot = (st_pos + lane) & (kChRing - 1u);
            s_pix[slot] = make_float4(0.0f, 0.0f, 0.0f, 0.0f); s_xy[slot] = make_float2(0.0f, 0.0f); s_inj[slot] = make_float2(0.0f, 0.0f);
        }
        st_pos += count;
    };
    auto stage_unit = [&]() {
        if (st_item >= n_items) { put_sentinels(kWave); st_done = true; return; }         // behind the last item: what the lanes see while it drains
        const unsigned tile = wave_read(d_tile, slot_of(st_item)), first_gaussian = wave_read(d_tb, slot_of(st_item)) * kBucket;
        const float x0 = static_cast<float>((tile % a.grid_w) * kTileW) + 0.5f, y0 = static_cast<float>((tile / a.grid_w) * kTileH) + 0.5f;
        const unsigned p = st_chunk * kWave + lane;
        const unsigned last = __float_as_uint(rc.w);
        const bool live = last > first_gaussian;                                           // kb:349-380, as in the kernel above
        const uint64_t m = wave_ballot(live);
        if (live) {
            const unsigned slot = (st_pos + lanes_below(m)) & (kChRing - 1u);
            const unsigned rel = min(last - first_gaussian, static_cast<unsigned>(kBucket));
            s_pix[slot] = make_float4(rg.x, rg.y, rg.z, static_cast<float>(rel));
            s_xy[slot] = make_float2(x0 + static_cast<float>(p & (kTileW - 1)), y0 + static_cast<float>(p / kTileW));
            const float S = (rc.x - rk.x) * rg.x + (rc.y - rk.y) * rg.y + (rc.z - rk.z) * rg.z;                           // kb:371-374
            s_inj[slot] = make_float2(rk.w, S - rg.w);
        }
        const unsigned added = static_cast<unsigned>(__popcll(m));
        st_pos += added; st_n += added;
        if (++st_chunk == kTilePixels / kWave) {                                           // the item is complete: pad it, the next one starts behind the pads
            const unsigned padded = (st_n + 8u + 7u) & ~7u;
            const unsigned length = padded < static_cast<unsigned>(kWave) ? static_cast<unsigned>(kWave) : padded;
            put_sentinels(length - st_n);                                                  // 8 .. 63 of them
            ++st_item; st_chunk = 0; st_n = 0;
            if (lane == (st_item & (kChDesc - 1u))) d_start = st_pos;                      // (item n_items: where the stream ends)
        }
        fetch_unit();
    };

    // ---- the pipeline ----
    unsigned ring_at = ((0u - lane) & (kChRing - 1u)) * 8u;                                // byte offset of position (step - lane) in the 8-byte rings
    const unsigned inj_mul = lane == 0 ? 1u : 0u, inj_add = lane == 0 ? kChInjBase : kChZeroBase;   // lane 0 injects the position's state, the others add zero
    struct PixRead { float4 g; float2 xy; };
    auto read_inj = [&]() { return *reinterpret_cast<const float2*>(s_base + (ring_at * inj_mul + inj_add)); };
    auto read_pix = [&]() {
        PixRead r;
        r.xy = *reinterpret_cast<const float2*>(s_base + ring_at);
        r.g = *reinterpret_cast<const float4*>(s_base + (2u * ring_at + kChPixBase));
        ring_at = (ring_at + 8u) & (kChXyBytes - 1u);
        return r;
    };
    auto step = [&](const float2 inj, const PixRead pr) {                                 // exactly the step of the kernel above
        sT = wave_shift_up1_zero(sT) + inj.x;                                               // kb:383-410
        sS = wave_shift_up1_zero(sS) + inj.y;
        const float4 px = pr.g;
        const float rel = px.w;
        const float dx = mx - pr.xy.x, dy = my - pr.xy.y;
        const float power = -0.5f * (ca * dx * dx + cc * dy * dy) - cb * dx * dy;
        const float alpha = op * __expf(fminf(power, 0.0f));
        if (lane_f < rel && alpha >= kMinAlphaThreshold) {                                  // kb:412,419-421
            const float T = sT;
            const float w = T * alpha;
            a_c0 += w * px.x; a_c1 += w * px.y; a_c2 += w * px.z;
            const float cg = col0 * px.x + col1 * px.y + col2 * px.z;
            sS -= w * cg;                                                                    // kb:429 projected on dL/dC
            const float oma = 1.0f - alpha;
            const float oma_rcp = fast_rcp(fmaxf(oma, kOneMinusAlphaEps));
            const float dl_dalpha = T * cg - sS * oma_rcp;                                   // kb:434-436
            const float hh = (-0.5f * alpha) * dl_dalpha;
            const float t = hh * dx, u = hh * dy;
            a_h += hh; a_x += t; a_y += u;
            a_xx += t * dx; a_xy += t * dy; a_yy += u * dy;
            sT = T * oma;
        }
    };

    prefetch_prim(0u);
    fetch_unit();
    unsigned next_shadow = 0;                                                              // items < next_shadow have their records in a shadow set
    unsigned sw_item = 0, sw_g = 0;                                                        // the next group switch: group sw_g enters item sw_item
    unsigned s0 = 0;                                                                       // the next step (a multiple of 8)
    unsigned desc_due = 0;                                                                 // first item of the half window to refill (0: none)
    float2 inj_a, inj_b;
    PixRead pix_a, pix_b;
    bool primed = false;
    for (;;) {                                                                             // OUTER level: everything that loads
        while (!st_done && st_pos < s0 + 16u) stage_unit();                                // lane 0 never runs into positions that are not there yet
        wave_lds_fence();
        // the shadow set is free once every group has entered the item it holds: it then takes the next one, a whole item's length before its first use
        if (desc_due != 0u) { load_descriptors(desc_due); desc_due = 0u; }
        if (next_shadow <= n_items && next_shadow <= sw_item) { load_shadow(next_shadow); ++next_shadow; }
        if (!primed) { inj_a = read_inj(); pix_a = read_pix(); primed = true; }            // the reads of step 0
        bool finished = false;
        for (;;) {                                                                         // INNER level: group switches and steps, no load is issued here
            if (sw_item <= st_item) {                                                      // (d_start of item j is written when item j - 1 completes; item 0: 0)
                const unsigned start = wave_read(d_start, slot_of(sw_item));
                if (s0 == start + 8u * sw_g) {
                    if (sw_item >= next_shadow) break;                                     // its record is not requested yet: outer level
                    switch_group(sw_g, sw_item > 0u);
                    if (++sw_g == 8u) {
                        sw_g = 0; ++sw_item;
                        // every group has entered item sw_item - 1, so item sw_item - 2 is staged to its end (an item's start is known only then) and
                        // flushed: nothing refers to the descriptors of items <= sw_item - 2 any more, while staging may still be busy with the tail of
                        // item sw_item - 1 and runs at most three items ahead (256 ring slots, items of >= 64 positions). One item into the other half
                        // of the window, the half the chain has left takes the sixteen items after the next sixteen
                        if (sw_item > n_items) { finished = true; break; }                  // every group has left the last item
                        if ((sw_item & 15u) == 1u && sw_item > 1u) { desc_due = sw_item + 15u; break; }      // (a load: outer level)
                    }
                }
            }
#pragma unroll
            for (int i = 0; i < 4; ++i) {                                                  // eight steps, the reads of a step issued one step ahead
                inj_b = read_inj(); pix_b = read_pix();
                step(inj_a, pix_a);
                inj_a = read_inj(); pix_a = read_pix();
                step(inj_b, pix_b);
            }
            s0 += 8u;
            if (!st_done && st_pos < s0 + 16u) break;                                      // the stream runs low: outer level
            if (next_shadow <= n_items && next_shadow <= sw_item) break;                   // the shadow set fell free: outer level
        }
        if (finished) break;
    }
}

#endif  // FGS_DEV_SWITCHES (variant 5)

#ifdef FGS_DEV_SWITCHES   // A/B exhibit (variant 4): libfgs_hip_dev.so only
// ---- variant 4: lane = PIXEL, reduction over the pixels on the matrix cores ----------------------------------------------------------------
// Measured in round 4 (tools/pair_stats.sh, profiles/r04_k11_pair_efficiency.txt): of the (pixel, Gaussian) lane-steps the systolic kernel above
// issues, 33-41 % pass the alpha test (S2, the layered scene, a trained export alike); a quarter of its steps is pipeline fill, and it evaluates
// every Gaussian of a bucket against all 192 pixels of the tile. The forward kernel's formulation -- lane = pixel, each wave walks only the
// Gaussians whose bounding box reaches its 16x4 strip -- visits 0.65-0.72 as many lane-steps, needs no fill, no ring of per-pixel constants (they
// sit in registers) and no shifted state (T and S belong to the lane). What it needs instead is a sum over the 64 pixels of a strip for each of
// the nine per-Gaussian gradients, which as DPP reductions costs 54 cross-lane adds per (Gaussian, strip) pair (the strip variant 1: slower).
// Here that sum is a matrix product. All nine sums are linear in two per-pair values, w = T alpha and hh = -alpha/2 dL/dalpha:
//     dL/dcolour_c = sum_p w g_c(p),      sum_p hh { 1, x', y', x'^2, x'y', y'^2 }   (x', y' = pixel centre relative to the TILE centre: exact
// small half-integers), from which the sums over dx = Dx - x', dy = Dy - y' (Dx, Dy = mean2d relative to the tile centre) follow per Gaussian:
//     sum hh dx = Dx Sh - Sx,  sum hh dx^2 = Dx (Dx Sh - 2 Sx) + Sxx,  sum hh dx dy = Dx Dy Sh - Dx Sy - Dy Sx + Sxy, ...
// So each wave keeps, for up to 8 walked Gaussians, w and hh of its 64 pixels in 16 rows of a private LDS buffer (the transposition: lanes are
// pixels when the rows are written and (k, column) pairs of the matrix instruction when they are read), and one pass of 16
// v_mfma_f32_16x16x4_f32 (fp32 in, fp32 accumulate: bitwise an fmaf chain) multiplies the 9 x 64 feature matrix of the strip
// [g_r, g_g, g_b, 1, x', y', x'^2, x'y', y'^2] with those rows: columns 0..7 = w of the 8 slots, 8..15 = hh. The matrix pipe runs beside the
// vector pipe; the vector instructions per pair are the forward walk's plus the gradient arithmetic. The results of the tile's three strips
// add up in LDS accumulators of the bucket's 64 Gaussians; then lane = Gaussian converts and issues the nine global atomics as the other variants do.
// alpha is the forward kernel's expression operation for operation (same dx, dy: pixel centre = px + 0.5), T and S restart from the bucket's
// checkpoint exactly as in the systolic form.
constexpr unsigned kPixSlots = 8;                        // walked Gaussians per matrix pass
constexpr unsigned kPixRows = 2 * kPixSlots;             // w rows, then hh rows
constexpr unsigned kPixStride = 68;                      // floats per row: 64 pixels + 4, so that the 16-byte reads of 16 columns hit 64 distinct banks
#ifndef FGS_K11M_MAX_BLOCKS
#define FGS_K11M_MAX_BLOCKS 65536
#endif
// Debug-only phase timer (tools/k11m_phases.sh, -DFGS_K11M_PHASES; the product build has none of it): shader cycles per wave summed over the launch --
// [0] items, [1] staging of the records, [2] per strip: pixel loads, feature operand, cull + order table, [3] the walk without its matrix passes,
// [4] the matrix passes, [5] tail (conversion, atomics), [6] pairs walked, [7] matrix passes.
#ifdef FGS_K11M_PHASES
__device__ unsigned long long g_k11m_phases[8];
#define FGS_PH(i) ph_[i] += __builtin_readcyclecounter() - pt_, pt_ = __builtin_readcyclecounter()
#else
#define FGS_PH(i)
#endif
__global__ void __launch_bounds__(kWave) blend_backward_pixel_kernel(const BlendBackwardArgs a) {
    // ONE wave per work item (tile, bucket), as in the systolic form: it walks the tile's three 16x4 strips one after the other, so there is no
    // workgroup barrier (the first version gave each strip its own wave: the two waves with the shorter lists waited at the barrier for the
    // third, and eight 3-wave workgroups per CU did not cover the four dependent loads at the head of every item).
    __shared__ float4 s_rec[3 * kBucket];                                      // mean.xy conic.ab | conic.c opacity r g (clamped) | b (clamped) bounds_x bounds_y flags
    __shared__ float s_acc[9 * kBucket];                                       // planes Sh Sx Sy Sxx Sxy Syy c0 c1 c2 of the bucket's Gaussians
    __shared__ __attribute__((aligned(16))) float s_v[kPixRows * kPixStride];
    __shared__ uint8_t s_order[kBucket + kPixSlots + 4];                       // bucket-relative index of the i-th Gaussian the current strip walks
    const unsigned lane = threadIdx.x, half = lane >> 5;
    float* const v_mine = s_v;
    const unsigned pos = (lane & 3u) * 16u + (lane >> 2);                      // pixel p = 4 s + q of matrix k-step s sits at q * 16 + s of its row
    const unsigned col = lane & 15u, q = lane >> 4;                            // this lane's column / k index in the matrix instruction
    const unsigned lx = half * kSubtileW + (lane & 7u), ly_in_strip = (lane >> 3) & 3u;
    const unsigned n_live = *a.live_count;
#ifdef FGS_K11M_PHASES
    unsigned long long ph_[8] = {0, 0, 0, 0, 0, 0, 0, 0}, pt_ = __builtin_readcyclecounter();
#endif
    // The head of an item is a chain of dependent loads (list entry -> tile range / bucket base -> primitive index -> record), and each strip
    // had two more (pixel record -> checkpoint). With four waves per SIMD that latency was two thirds of a wave's time (tools/k11m_phases.sh). So:
    // the scalar part of the chain is fetched one item AHEAD (the wave keeps the next item's list entry, range and bucket base in scalar
    // registers), and everything per pixel -- the three strips' records and checkpoints -- is requested in one go at the top of the item, next
    // to the primitive indices: two exposed round trips per item instead of ten.
    unsigned item = blockIdx.x;
    uint2 work = make_uint2(0u, 0u), range = make_uint2(0u, 0u);
    unsigned bucket_base = 0;
    if (item < n_live) {
        work = a.work_list[item];
        range = a.ranges[work.x];
        bucket_base = work.x == 0 ? 0u : a.bucket_offsets[work.x - 1];
    }
    for (; item < n_live; item += gridDim.x) {                                 // wave-uniform
#ifdef FGS_K11M_PHASES
        ph_[0] += 1; pt_ = __builtin_readcyclecounter();
#endif
        const unsigned tile = work.x, tb = work.y;
        const unsigned tile_n = range.y - range.x;
        const unsigned bucket = bucket_base + tb;
        const unsigned first_gaussian = tb * kBucket;
        const unsigned n_here = min(static_cast<unsigned>(kBucket), tile_n - first_gaussian);
        const unsigned tile_x = tile % a.grid_w, tile_y = tile / a.grid_w;
        const unsigned range_x = range.x;

        // ---- requests of this item: primitive index, the three strips' pixel records and checkpoints ----
        uint32_t prim = 0, hot_slot_word = 0;
        if (lane < n_here) prim = a.inst_prims[range_x + first_gaussian + lane];
        float4 cst_[3], g_[3], ck_[3];
#pragma unroll
        for (unsigned st = 0; st < 3u; ++st) {
            const unsigned local = (st * kSubtileH + ly_in_strip) * kTileW + lx;
            cst_[st] = a.pixrec[((size_t)tile * kTilePixels + local) * 2 + 1];
            g_[st] = a.pixrec[((size_t)tile * kTilePixels + local) * 2];
            ck_[st] = a.ckpt[(size_t)bucket * kTilePixels + local];
        }
        // ---- the scalar head of the NEXT item ----
        {
            const unsigned next = item + gridDim.x;
            if (next < n_live) {
                work = a.work_list[next];
                range = a.ranges[work.x];
                bucket_base = work.x == 0 ? 0u : a.bucket_offsets[work.x - 1];
            }
        }
        {                                                                      // the bucket's records (kb:297-319), lane = Gaussian
            float4 r0 = make_float4(0.0f, 0.0f, 0.0f, 0.0f), r1 = r0, r2 = r0;
            uint32_t flags = 0;
            if (lane < n_here) {
                const float4* r = reinterpret_cast<const float4*>(a.rec + prim);
                r0 = r[0]; r1 = r[1]; r2 = r[2];
                flags = (r1.z >= 0.0f ? 1u : 0u) | (r1.w >= 0.0f ? 2u : 0u) | (r2.x >= 0.0f ? 4u : 0u);      // kb:313-318
                hot_slot_word = __float_as_uint(r2.w);
            }
            s_rec[lane] = r0;
            s_rec[kBucket + lane] = make_float4(r1.x, r1.y, fmaxf(r1.z, 0.0f), fmaxf(r1.w, 0.0f));
            s_rec[2 * kBucket + lane] = make_float4(fmaxf(r2.x, 0.0f), r2.y, r2.z, __uint_as_float(flags));
#pragma unroll
            for (unsigned e = 0; e < 9u; ++e) s_acc[e * kBucket + lane] = 0.0f;
        }
        // ---- the pixels' state at the bucket's checkpoint (kb:349-380): six values per strip stay in registers ----
        float gx_[3], gy_[3], gz_[3], T_[3], S_[3];
        unsigned rel_[3];
#pragma unroll
        for (unsigned st = 0; st < 3u; ++st) {
            const unsigned last = __float_as_uint(cst_[st].w);                // 0 outside the image
            // a pixel that finished before this bucket never wrote its checkpoint (kf:436) and receives nothing here
            const bool live = last > first_gaussian;
            rel_[st] = live ? last - first_gaussian : 0u;                      // Gaussians of this bucket in front of the pixel's last contributor
            gx_[st] = g_[st].x; gy_[st] = g_[st].y; gz_[st] = g_[st].z;
            T_[st] = live ? ck_[st].w : 0.0f;
            S_[st] = live ? ((cst_[st].x - ck_[st].x) * g_[st].x + (cst_[st].y - ck_[st].y) * g_[st].y + (cst_[st].z - ck_[st].z) * g_[st].z) - g_[st].w : 0.0f;   // kb:371-377 projected on dL/dC
        }
        wave_lds_fence();
        FGS_PH(1);

#pragma unroll 1
        for (unsigned strip = 0; strip < static_cast<unsigned>(kTilePixels / kWave); ++strip) {
            const unsigned ly = strip * kSubtileH + ly_in_strip;
            const float pxf = static_cast<float>(tile_x * kTileW + lx) + 0.5f, pyf = static_cast<float>(tile_y * kTileH + ly) + 0.5f;
            // (the strip loop stays rolled -- the walk below is long -- so the strip's six values are picked by selects, not by indexing register arrays)
            const unsigned rel = strip == 0u ? rel_[0] : strip == 1u ? rel_[1] : rel_[2];
            // Gaussians at or behind every pixel's last contributor take nothing (kb:412). (The maximum is wave-uniform; the compiler only knows that of
            // a value read through v_readfirstlane, and a list it believes divergent turns the walk below into an EXEC-masked vector loop.)
            const unsigned rel_max = wave_uniform(wave_max(rel));
            if (rel_max == 0u) continue;                                       // no live pixel in this strip
            const float4 g = make_float4(strip == 0u ? gx_[0] : strip == 1u ? gx_[1] : gx_[2], strip == 0u ? gy_[0] : strip == 1u ? gy_[1] : gy_[2],
                                         strip == 0u ? gz_[0] : strip == 1u ? gz_[1] : gz_[2], 0.0f);
            float T = strip == 0u ? T_[0] : strip == 1u ? T_[1] : T_[2], sS = strip == 0u ? S_[0] : strip == 1u ? S_[1] : S_[2];

            // ---- the strip's feature matrix as matrix operand A: row c of [g_r g_g g_b 1 x' y' x'^2 x'y' y'^2], 16 k-steps ----
            float A[16];
            {
                const float xr = static_cast<float>(lx) - 7.5f, yr = static_cast<float>(ly) - 5.5f;
                v_mine[0 * kPixStride + pos] = g.x; v_mine[1 * kPixStride + pos] = g.y; v_mine[2 * kPixStride + pos] = g.z;
                v_mine[3 * kPixStride + pos] = 1.0f; v_mine[4 * kPixStride + pos] = xr; v_mine[5 * kPixStride + pos] = yr;
                v_mine[6 * kPixStride + pos] = xr * xr; v_mine[7 * kPixStride + pos] = xr * yr; v_mine[8 * kPixStride + pos] = yr * yr;
                wave_lds_fence();
                const float4* ap = reinterpret_cast<const float4*>(v_mine + min(col, 8u) * kPixStride + q * 16u);
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const float4 t = ap[i];
                    const bool used = col < 9u;
                    A[4 * i] = used ? t.x : 0.0f; A[4 * i + 1] = used ? t.y : 0.0f; A[4 * i + 2] = used ? t.z : 0.0f; A[4 * i + 3] = used ? t.w : 0.0f;
                }
                wave_lds_fence();                                              // the rows are reused for w / hh below
            }

            // ---- cull the bucket against this strip's two 8x4 sub-tiles (kf:445-451) ----
            const unsigned sub_y0 = tile_y * kTileH + strip * kSubtileH, sub_y1 = sub_y0 + kSubtileH;
            const unsigned subl_x0 = tile_x * kTileW, subl_x1 = subl_x0 + kSubtileW, subr_x1 = subl_x1 + kSubtileW;
            bool in_l = false, in_r = false;
            if (lane < n_here) {
                const float4 gc = s_rec[2 * kBucket + lane];
                const uint32_t bx = __float_as_uint(gc.y), by = __float_as_uint(gc.z);
                const unsigned x_min = bx & 0xffffu, x_max = bx >> 16, y_min = by & 0xffffu, y_max = by >> 16;
                const bool in_y = y_min < sub_y1 && sub_y0 < y_max;
                in_l = in_y && x_min < subl_x1 && subl_x0 < x_max;
                in_r = in_y && x_min < subr_x1 && subl_x1 < x_max;
            }
            const uint64_t mask_l = wave_ballot(in_l), mask_r = wave_ballot(in_r);
            const uint64_t pending = (mask_l | mask_r) & (rel_max >= 64u ? ~0ull : ((1ull << rel_max) - 1ull));
            // slot -> Gaussian of the matrix batches: the walk visits the set bits of `pending` in order, so its i-th pair is the i-th set bit
            // (every lane stores -- the lanes outside the list into a spare element -- so that no divergent branch sits between the list and the walk)
            s_order[((pending >> lane) & 1ull) ? lanes_below(pending) : kBucket + kPixSlots] = static_cast<uint8_t>(lane);
            FGS_PH(2);

            // ---- the walk: eight pairs into the rows, then one matrix pass. (Two other schedules were built and measured slower on the layered scene,
            // profiles/r04_k11m_closeout.txt: the matrix instructions of batch n issued between the pairs of batch n + 1 -- 1.69 ms against 1.58 --
            // and groups of four pairs as straight-line code for instruction-level parallelism -- 132 registers, three waves per SIMD, 1.83 ms.)
            unsigned n_slots = 0, n_flushed = 0;                               // filled rows of the current matrix batch, pairs of earlier batches (wave-uniform)
            auto flush = [&](const unsigned n) {
                FGS_PH(3);
                wave_lds_fence();
                const float4* bp = reinterpret_cast<const float4*>(v_mine + col * kPixStride + q * 16u);
                const float4 b0 = bp[0], b1 = bp[1], b2 = bp[2], b3 = bp[3];
                const unsigned slot = col & 7u;
                const unsigned gi = s_order[n_flushed + slot];
                fgs_acc4 d0 = {0.0f, 0.0f, 0.0f, 0.0f}, d1 = {0.0f, 0.0f, 0.0f, 0.0f};      // two chains: a dependent matrix instruction waits 40 cycles, an independent one 32
                if (!(a.ablate & 4)) {
                    wave_mfma_16x16x4(A[0], b0.x, d0); wave_mfma_16x16x4(A[1], b0.y, d1); wave_mfma_16x16x4(A[2], b0.z, d0); wave_mfma_16x16x4(A[3], b0.w, d1);
                    wave_mfma_16x16x4(A[4], b1.x, d0); wave_mfma_16x16x4(A[5], b1.y, d1); wave_mfma_16x16x4(A[6], b1.z, d0); wave_mfma_16x16x4(A[7], b1.w, d1);
                    wave_mfma_16x16x4(A[8], b2.x, d0); wave_mfma_16x16x4(A[9], b2.y, d1); wave_mfma_16x16x4(A[10], b2.z, d0); wave_mfma_16x16x4(A[11], b2.w, d1);
                    wave_mfma_16x16x4(A[12], b3.x, d0); wave_mfma_16x16x4(A[13], b3.y, d1); wave_mfma_16x16x4(A[14], b3.z, d0); wave_mfma_16x16x4(A[15], b3.w, d1);
                }
                // this lane holds D[row 4 q + r][col]: rows 0..2 = colour sums (columns 0..7, the w rows), rows 3..8 = moment sums (columns 8..15, the hh rows).
                // One wave owns the accumulators, and a Gaussian sits in one slot of one pass per strip: plain read-modify-write, in program order.
                const bool is_w = col < 8u;
                if (slot < n && !(a.ablate & 4)) {
                    if (q == 0u) {
                        if (is_w) {
                            s_acc[6 * kBucket + gi] += d0[0] + d1[0]; s_acc[7 * kBucket + gi] += d0[1] + d1[1]; s_acc[8 * kBucket + gi] += d0[2] + d1[2];
                        } else s_acc[gi] += d0[3] + d1[3];
                    } else if (!is_w) {
                        if (q == 1u) {
                            s_acc[1 * kBucket + gi] += d0[0] + d1[0]; s_acc[2 * kBucket + gi] += d0[1] + d1[1];
                            s_acc[3 * kBucket + gi] += d0[2] + d1[2]; s_acc[4 * kBucket + gi] += d0[3] + d1[3];
                        } else if (q == 2u) s_acc[5 * kBucket + gi] += d0[0] + d1[0];
                    }
                }
                n_flushed += n;
                wave_lds_fence();                                              // the next batch overwrites the rows
#ifdef FGS_K11M_PHASES
                ph_[7] += 1; ph_[6] += n;
#endif
                FGS_PH(4);
            };

            float* v_row = v_mine + pos;                                       // this lane's element of the row of the current slot
            uint64_t pend = wave_uniform(pending);                             // (re-stated uniform: see rel_max)
            while (pend != 0ull) {                                             // wave-uniform scalar loop
                const unsigned j = static_cast<unsigned>(__ffsll(static_cast<unsigned long long>(pend))) - 1u;
                pend &= pend - 1ull;
                if (!(a.ablate & 8)) {
                    const float4* const entry = s_rec + j;
                    const float4 ga = entry[0], gb = entry[kBucket];
                    const float colb = entry[2 * kBucket].x;
                    const float dx = ga.x - pxf, dy = ga.y - pyf;
                    const float power = -0.5f * (ga.z * dx * dx + gb.x * dy * dy) - ga.w * dx * dy;
                    const float gauss = __expf(fminf(power, 0.0f));
                    const float alpha_raw = gb.y * gauss;
                    // Contributes (kb:412,419-421; kf:445-467): alpha >= 1/255, the Gaussian in front of this pixel's last contributor, and its box on this
                    // lane's 8x4 sub-tile -- the last one is the same for the 32 lanes of a half, so it is a scalar mask. Branch-free: a pair that does not
                    // contribute runs the same instructions with alpha = 0, which leaves T and S as they are and gives w = hh = 0 -- the contribution block
                    // would run anyway in 92-96 % of the pairs (profiles/r04_k11_pair_efficiency.txt: some lane passes), and there is no EXEC bookkeeping
                    // and no zero-fill of the two values that go to the matrix rows.
                    const uint64_t not_mine = (((mask_l >> j) & 1ull) ? 0ull : 0x00000000ffffffffull) | (((mask_r >> j) & 1ull) ? 0ull : 0xffffffff00000000ull);
                    const uint64_t pass = wave_ballot(alpha_raw >= kMinAlphaThreshold && j < rel) & ~not_mine;
                    const float alpha = lane_select(pass, 0.0f, alpha_raw);
                    const float w = T * alpha;
                    const float cg = gb.z * g.x + gb.w * g.y + colb * g.z;
                    sS -= w * cg;                                               // kb:429 projected on dL/dC
                    const float oma = 1.0f - alpha;
                    const float oma_rcp = fast_rcp(fmaxf(oma, kOneMinusAlphaEps));
                    const float dl_dalpha = T * cg - sS * oma_rcp;              // kb:434-436
                    const float hh = (-0.5f * alpha) * dl_dalpha;
                    T *= oma;
                    v_row[0] = w;
                    v_row[kPixSlots * kPixStride] = hh;
                }
                v_row += kPixStride;
                if (++n_slots == kPixSlots) { flush(kPixSlots); n_slots = 0; v_row = v_mine + pos; }
            }
            if (n_slots != 0) flush(n_slots);
            FGS_PH(3);
        }
        wave_lds_fence();

        // ---- per Gaussian (lane = Gaussian again): moments about the tile centre -> the nine gradients, added to the Gaussian's record (kb:459-470) ----
        if (lane < n_here) {
            const float Sh = s_acc[lane], Sx = s_acc[kBucket + lane], Sy = s_acc[2 * kBucket + lane];
            const float Sxx = s_acc[3 * kBucket + lane], Sxy = s_acc[4 * kBucket + lane], Syy = s_acc[5 * kBucket + lane];
            const float c0 = s_acc[6 * kBucket + lane], c1 = s_acc[7 * kBucket + lane], c2 = s_acc[8 * kBucket + lane];
            const bool silent = Sh == 0.0f && Sx == 0.0f && Sy == 0.0f && Sxx == 0.0f && Sxy == 0.0f && Syy == 0.0f && c0 == 0.0f && c1 == 0.0f && c2 == 0.0f;
            if (!silent && !(a.ablate & 1)) {
                const float4 ga = s_rec[lane], gb = s_rec[kBucket + lane], gc = s_rec[2 * kBucket + lane];
                const float ca = ga.z, cb = ga.w, cc = gb.x, op = gb.y;
                const float Dx = ga.x - (static_cast<float>(tile_x * kTileW) + 8.0f), Dy = ga.y - (static_cast<float>(tile_y * kTileH) + 6.0f);
                const float a_x = Dx * Sh - Sx, a_y = Dy * Sh - Sy;
                const float a_xx = Dx * (Dx * Sh - 2.0f * Sx) + Sxx, a_yy = Dy * (Dy * Sh - 2.0f * Sy) + Syy;
                const float a_xy = Dx * (Dy * Sh - Sy) - Dy * Sx + Sxy;
                const unsigned flags = __float_as_uint(gc.w);
                unsigned tx0, tx1, ty0, ty1;
                tile_rect(__float_as_uint(gc.y), __float_as_uint(gc.z), tx0, tx1, ty0, ty1);
                const unsigned footprint = (tx1 - tx0) * (ty1 - ty0);
                const uint32_t hot_word = footprint > kHotFootprint ? hot_slot_word : 0u;
                // the Gaussian's own record of nine consecutive floats, or (hot) the record of its slot in the tile's replica
                float* dst = hot_word != 0u ? a.acc_hot + ((size_t)(tile % kHotReplicas) * kMaxHot + (hot_word - 1u)) * kAccRecordWords : a.acc + (size_t)prim * kAccRecordWords;
                unsafeAtomicAdd(dst, 2.0f * (ca * a_x + cb * a_y));
                unsafeAtomicAdd(dst + 1, 2.0f * (cb * a_x + cc * a_y));
                unsafeAtomicAdd(dst + 2, a_xx);
                unsafeAtomicAdd(dst + 3, a_xy);
                unsafeAtomicAdd(dst + 4, a_yy);
                unsafeAtomicAdd(dst + 5, a.proper_aa ? -2.0f * Sh / op : -2.0f * Sh * (1.0f - op));
                unsafeAtomicAdd(dst + 6, (flags & 1u) ? c0 : 0.0f);
                unsafeAtomicAdd(dst + 7, (flags & 2u) ? c1 : 0.0f);
                unsafeAtomicAdd(dst + 8, (flags & 4u) ? c2 : 0.0f);
            }
        }
        wave_lds_fence();                                                      // the next item restages the records and clears the accumulators
        FGS_PH(5);
    }
#ifdef FGS_K11M_PHASES
    if (lane == 0) for (int i = 0; i < 8; ++i) if (ph_[i] != 0) atomicAdd(&g_k11m_phases[i], ph_[i]);
#endif
}
#ifdef FGS_K11M_PHASES
}  // namespace fgs
extern "C" __attribute__((visibility("default"))) int fgs_debug_k11m_phases(unsigned long long* out, int reset) {
    if (out != nullptr && hipMemcpyFromSymbol(out, HIP_SYMBOL(fgs::g_k11m_phases), sizeof(unsigned long long) * 8) != hipSuccess) return -1;
    if (reset) {
        void* dev = nullptr;
        if (hipGetSymbolAddress(&dev, HIP_SYMBOL(fgs::g_k11m_phases)) != hipSuccess || hipMemset(dev, 0, sizeof(unsigned long long) * 8) != hipSuccess) return -1;
    }
    return 0;
}
namespace fgs {
#endif

#endif  // FGS_DEV_SWITCHES

#ifdef FGS_PAIR_STATS
}  // namespace fgs
extern "C" __attribute__((visibility("default"))) int fgs_debug_k11_pair_stats(unsigned long long* out, int reset) {
    if (out != nullptr && hipMemcpyFromSymbol(out, HIP_SYMBOL(fgs::g_k11_pair_stats), sizeof(unsigned long long) * 8) != hipSuccess) return -1;
    if (reset) {
        void* dev = nullptr;
        if (hipGetSymbolAddress(&dev, HIP_SYMBOL(fgs::g_k11_pair_stats)) != hipSuccess || hipMemset(dev, 0, sizeof(unsigned long long) * 8) != hipSuccess) return -1;
    }
    return 0;
}
namespace fgs {
#endif

// after K11: the hot Gaussians' replicas are summed and added to their records (one thread per (slot, sum))
__global__ void __launch_bounds__(256) fold_hot_accumulators_kernel(const BlendBackwardArgs a) {
    const unsigned n_hot = min(*a.hot_count, kMaxHot);
    const unsigned e = blockIdx.x * 256u + threadIdx.x;
    // this is the last kernel of a backward pass: the accumulator records now hold sums -- a second backward pass over the same buffers (a retained
    // graph) must clear them itself (stage_pixels_kernel reads the flag)
    if (e == 0u) *a.dirty_flag = 1u;
    const unsigned slot = e / kAccRecordWords, k = e % kAccRecordWords;          // consecutive threads: the nine sums of a slot, then the next slot
    if (slot >= n_hot) return;
    float sum = 0.0f;
#pragma unroll
    for (unsigned r = 0; r < kHotReplicas; ++r) sum += a.acc_hot[((size_t)r * kMaxHot + slot) * kAccRecordWords + k];
    if (sum != 0.0f) a.acc[(size_t)a.hot_list[slot] * kAccRecordWords + k] += sum;      // one slot per primitive: no other writer at this point
}
#ifdef FGS_DEV_SWITCHES
__global__ void mark_accumulators_dirty_kernel(uint32_t* flag) { *flag = 1u; }      // the A/B variants that do not end in the fold kernel
#endif

#ifdef FGS_DEV_SWITCHES
std::atomic<int> g_k11m_max_blocks{FGS_K11M_MAX_BLOCKS};   // variant 4: upper bound of its grid (fgs_debug_set_option(13, n)); items beyond it are walked grid-stride
std::atomic<int> g_backward_ablate{0};    // fgs_debug_set_option(7, bits): timing experiments only -- 1: no atomics, 2: no step loop (results are wrong)
std::atomic<int> g_backward_variant{3};   // 3 (default): work list + compacted pixels + two-value state; 2: systolic over all buckets / all 192 pixels, dL/dC from
                              // global memory (round 1: 0.70 ms at S2); 0: same with dL/dC in LDS (0.74); 1: strip (lane = pixel, 0.85 ms);
                              // fgs_debug_set_backward_variant()
int blend_backward_variant() { return g_backward_variant.load(); }
#else
int blend_backward_variant() { return 3; }     // the product build has one formulation
#endif

// a.variant: read ONCE per backward pass by the caller (api.hip: run_blend_backward) -- the planning pass and the kernel must agree on it
hipError_t launch_stage_pixels(const BlendBackwardArgs& a_in, hipStream_t s) {
    BlendBackwardArgs a = a_in;
    if (a.variant >= 3 && a.n_buckets_cap != 0) hipLaunchKernelGGL(plan_blend_backward_kernel, dim3(1), dim3(kTileScanThreads), 0, s, a);
    else a.live_offsets = nullptr;                            // the other variants walk all buckets: no list
    hipLaunchKernelGGL(stage_pixels_kernel, dim3(a.n_tiles), dim3(kTilePixels), 0, s, a);
    return hipGetLastError();
}

hipError_t launch_blend_backward(const BlendBackwardArgs& a_in, hipStream_t s) {
    if (a_in.n_buckets_cap == 0) return hipSuccess;
    BlendBackwardArgs a = a_in;
#ifdef FGS_DEV_SWITCHES
    a.ablate = g_backward_ablate;
    if (a.variant == 4) {
        const unsigned cap_blocks = static_cast<unsigned>(g_k11m_max_blocks.load());
        const unsigned blocks = a.n_buckets_cap < cap_blocks ? a.n_buckets_cap : cap_blocks;
        hipLaunchKernelGGL(blend_backward_pixel_kernel, dim3(blocks), dim3(kWave), 0, s, a);
        hipLaunchKernelGGL(fold_hot_accumulators_kernel, dim3(9u * kMaxHot / 256u), dim3(256), 0, s, a);
        return hipGetLastError();
    }
    if (a.variant == 5) {        // chained (round 6): option 14 = its number of waves (tests shorten it so that chains get long)
        const unsigned chain_waves = static_cast<unsigned>(static_cast<int>(g_k11_chain_waves));
        hipLaunchKernelGGL(blend_backward_chained_kernel, dim3(a.n_buckets_cap < chain_waves ? a.n_buckets_cap : chain_waves), dim3(kWave), 0, s, a);
        hipLaunchKernelGGL(fold_hot_accumulators_kernel, dim3(9u * kMaxHot / 256u), dim3(256), 0, s, a);
        return hipGetLastError();
    }
    if (a.variant == 1) {
        hipLaunchKernelGGL(blend_backward_strip_kernel, dim3(a.n_buckets_cap), dim3(kTilePixels), 0, s, a);
        hipLaunchKernelGGL(mark_accumulators_dirty_kernel, dim3(1), dim3(1), 0, s, a.dirty_flag);
        return hipGetLastError();
    }
    if (a.variant != 3) {
        const dim3 grid((a.n_buckets_cap + kBackwardWavesPerBlock - 1) / kBackwardWavesPerBlock), block(kBackwardWavesPerBlock * kWave);
        if (a.variant == 2) hipLaunchKernelGGL(blend_backward_kernel<true>, grid, block, 0, s, a);
        else hipLaunchKernelGGL(blend_backward_kernel<false>, grid, block, 0, s, a);
        hipLaunchKernelGGL(mark_accumulators_dirty_kernel, dim3(1), dim3(1), 0, s, a.dirty_flag);
        return hipGetLastError();
    }
#endif
    // grid-stride over the live list: at most 64 Ki single-wave workgroups, so a scene with few live buckets does not pay
    // for the launch of a quarter of a million empty ones
    const unsigned blocks = a.n_buckets_cap < kBackwardMaxBlocks ? a.n_buckets_cap : kBackwardMaxBlocks;
    hipLaunchKernelGGL(blend_backward_compact_kernel, dim3((blocks + kCompactWaves - 1) / kCompactWaves), dim3(kWave * kCompactWaves), 0, s, a);
    hipLaunchKernelGGL(fold_hot_accumulators_kernel, dim3(9u * kMaxHot / 256u), dim3(256), 0, s, a);
    return hipGetLastError();
}

}  // namespace fgs
