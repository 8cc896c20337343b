// K10: per-tile front-to-back alpha blending for gfx950 (training variant with per-bucket checkpoints, and the
// forward-only inference variant). Semantics: reference kernels_forward.cuh:362-498 / kernels_inference.cuh:348-463.
//
// CDNA4 shape (not a translation of the 6-warp CUDA block):
//  * a 16x12 tile is 3 wave64; each wave owns a 16x4 strip = the two 8x4 sub-tiles of the reference side by side
//    (lanes 0-31 left, 32-63 right). The reference's per-sub-tile bounding-box cull is kept exactly: each lane tests one
//    of 64 staged Gaussians against BOTH sub-tiles, two 64-bit ballots give the per-half masks, the wave walks the union
//    with scalar bit ops and each half predicates on its own mask.
//  * Gaussians are staged 192 at a time through LDS from ONE 48-byte record per primitive (3 x 16-B loads from one line).
//  * checkpoints are written every kBucket=64 Gaussians (one backward wavefront) -- half the reference's checkpoint
//    traffic -- 1 KiB contiguous per wave; T_final / n_processed are tile-major so backward reads them coalesced.
//  * workgroup -> tile mapping keeps contiguous image bands on one XCD (workgroup b runs on XCD b % 8), so the
//    records gathered by neighbouring tiles stay in that XCD's 4 MiB L2.
#include <atomic>

#include "fgs_kernels.h"
#include <fgs_wave.h>

#ifndef FGS_CKPT_NT
#define FGS_CKPT_NT 1      // round 6: K10's checkpoints -- written once, read once by K11 a millisecond later -- leave as non-temporal stores (training iteration 2.218 -> 2.187 ms, layered scene 3.728 -> 3.706, three alternating pairs: profiles/r06_ab_ckpt_nt.txt); 0: A/B
#endif
namespace fgs {

// Which tile does workgroup `block` blend? The hardware deals workgroups to the 8 XCDs round-robin (XCD = block % 8), and a Gaussian's
// records are re-read by every tile it overlaps -- from the XCD's own L2 if the neighbouring tiles run there. Round 1 gave every XCD one
// contiguous band of tile rows. A per-tile timeline (tools/k10_timeline.sh, profiles/archive/r02_k10_timeline_before.txt) showed what that costs: the
// top band of the image is nearly empty (XCD 0 had 30 ms of summed tile time against 44-47 ms for the others at S2, 47 against 210-220 ms on the
// layered scene, and idled for a third / two thirds of the kernel), and the heaviest rows -- the bottom of the image, nearest to the camera --
// came LAST in every band. Alternative mapping (row_group >= 1): groups of `row_group` consecutive tile rows are dealt to the XCDs in turn over
// the whole image and every XCD walks its rows from the bottom of the image upwards (heaviest first). Measured (tools/ab_tile_rows.py,
// profiles/archive/r02_ab_tile_rows.txt; training / inference blend): S2 bands 0.163 / 0.157 ms, g = 1 0.177 / 0.171, g = 2 0.189 / 0.182 -- at two
// blended buckets per tile the kernel lives on the L2 locality of vertical neighbours; layered scene (11 buckets per tile) bands 0.725 / 0.700,
// g = 1 0.663 / 0.646, g = 2 0.654 / 0.633 -- there balance wins 10 %. Bands walked bottom-up (255): no difference. Which of the two a scene
// wants depends on how deep its tiles blend, which the host does not know at launch: the DEFAULT stays the bands (the benchmark workload),
// fgs_debug_set_option(10, g) selects the other. Returns n_tiles for padding workgroups.
// Round 3, measured on one box (tools/ab_tile_plan.py, profiles/archive/r03_ab_tile_plan.txt; training blend S2 / layered scene, ms):
//   bands (round 1/2 default)                          0.168 / 0.765
//   single rows interleaved                            0.179 / 0.657
//   8 x 10 blocks weighed on the device by their bucket counts, sorted, dealt heaviest-first to the least-loaded XCD
//   (row_group == kPlannedBlocks, plan_tiles_kernel)   0.181 / 0.646   -- balance, but the scattered block order costs S2 what rows cost
//   the same blocks in natural order, XCD x = block column x          0.164 / 0.673
//   COLUMNS (kColumnsTopDown, the default now): XCD x owns the vertical strip of tile columns [x w, (x + 1) w), w = ceil(grid_w / 8),
//   and walks it row by row from the top                              0.164 / 0.669   (bottom-up: 0.177 / 0.695)
// The work gradient of a rendered view is vertical (sky on top, near ground at the bottom), so a vertical strip per XCD is balanced by
// construction and as compact as a band: -2 % at S2 and -12 % on the layered scene against the bands, closed form, no device data. The
// device-side plan stays as an A/B option (it wins 3 % more on the layered scene and loses 10 % at S2).
constexpr unsigned kBandsBottomFirst = 255u;     // row_group value: the round-1 bands, each walked from its last tile to its first
__device__ __forceinline__ unsigned tile_of_workgroup(const unsigned block, const unsigned grid_w, const unsigned n_tiles, const unsigned row_group,
                                                      const uint32_t* __restrict__ plan = nullptr, const unsigned grid_h = 0u) {
    if (row_group == kPlannedBlocks) {
        const unsigned bw = (grid_w + kPlanBlocksX - 1) / kPlanBlocksX, bh = (grid_h + kPlanBlocksY - 1) / kPlanBlocksY;   // = plan[0], plan[1]
        const unsigned per_block = bw * bh;
        const unsigned xcd = block % kXcds, q = block / kXcds;
        const unsigned slot = q / per_block, local = q - slot * per_block;
        if (slot >= kPlanBlocksPerXcd) return n_tiles;
        const unsigned b = plan[kPlanHeader + xcd * kPlanBlocksPerXcd + slot];                 // wave-uniform: a scalar load
        const unsigned ly = local / bw, lx = local - ly * bw;
        const unsigned tx = (b % kPlanBlocksX) * bw + lx, ty = (b / kPlanBlocksX) * bh + ly;
        return (tx < grid_w && ty < grid_h) ? ty * grid_w + tx : n_tiles;
    }
    if (row_group == kColumnsTopDown || row_group == kColumnsBottomUp) {
        // every XCD owns one vertical strip of the image, ceil(grid_w / 8) tiles wide, and walks it row by row: compact (the strip's rows
        // follow each other in time, so a Gaussian's record is still in this XCD's L2 when the row below needs it), and balanced by
        // construction against the dominant work gradient of a rendered scene -- the vertical one (sky / far background on top, near
        // ground at the bottom): every XCD gets every image row
        const unsigned bw = (grid_w + kXcds - 1) / kXcds;
        const unsigned xcd = block % kXcds, q = block / kXcds;
        const unsigned r = q / bw, c = q - r * bw;
        const unsigned tx = xcd * bw + c, ty = row_group == kColumnsTopDown ? r : grid_h - 1u - r;
        return (tx < grid_w && r < grid_h) ? ty * grid_w + tx : n_tiles;
    }
    if (row_group == kBandsThroughPlan) {                                                   // A/B: what does the plan's load alone cost?
        const unsigned per_xcd = (n_tiles + kXcds - 1) / kXcds;
        const unsigned tile = (block % kXcds) * per_xcd + block / kXcds + (plan[kPlanHeader + (block % kXcds) * kPlanBlocksPerXcd] >> 30);
        return tile < n_tiles ? tile : n_tiles;
    }
    if (row_group == 0u || row_group == kBandsBottomFirst) {                                // one contiguous band per XCD, top-down or bottom-up
        const unsigned per_xcd = (n_tiles + kXcds - 1) / kXcds;
        const unsigned idx = block / kXcds;
        const unsigned tile = (block % kXcds) * per_xcd + (row_group == 0u ? idx : per_xcd - 1u - idx);
        return tile < n_tiles ? tile : n_tiles;
    }
    const unsigned n_rows = n_tiles / grid_w;
    const unsigned xcd = block % kXcds, j = block / kXcds;
    const unsigned k = j / grid_w, col = j - k * grid_w;                                   // k-th row this XCD walks
    const unsigned cycles = (n_rows + kXcds * row_group - 1) / (kXcds * row_group);        // groups per XCD
    if (k >= cycles * row_group) return n_tiles;
    const unsigned kk = cycles * row_group - 1u - k;                                       // bottom of the image first
    const unsigned row = (kk / row_group) * (kXcds * row_group) + xcd * row_group + kk % row_group;
    return row < n_rows ? row * grid_w + col : n_tiles;
}
// Round 5 tried the strips as per-XCD QUEUES with stealing (a workgroup pops the next tile of the strip of the XCD it runs on, hardware XCC_ID, and takes
// from the fullest other strip once its own is empty) for object-centric scenes, whose outer strips are nearly empty (bench.py's surface scene: 5.8 ms of
// summed tile time on XCD 0 against 108 ms on XCD 3). Measured (tools/ab_k10_mapping.py at the commit that had it): the one returning atomic per workgroup
// costs S2 0.145 -> 0.176 ms (10 800 pops onto eight words), a device-scope snapshot of the queue heads in front of it 0.73 ms; and the scene that
// motivated it gains nothing from balance alone -- the device-side block plan, which balances it, measures 0.366 against 0.387 ms -- because its span is
// the serial walk of single tiles with lists of thousands (389 us for one tile of 3 860 walked entries). Removed.
static unsigned blend_grid(const BlendArgs& a) {
    if (a.row_group == kColumnsTopDown || a.row_group == kColumnsBottomUp) return kXcds * ((a.grid_w + kXcds - 1) / kXcds) * a.grid_h;
    if (a.row_group == kBandsThroughPlan) return ((a.n_tiles + kXcds - 1) / kXcds) * kXcds;
    if (a.row_group == kPlannedBlocks)
        return kPlanBlocks * ((a.grid_w + kPlanBlocksX - 1) / kPlanBlocksX) * ((a.grid_h + kPlanBlocksY - 1) / kPlanBlocksY);
    if (a.row_group == 0u || a.row_group == kBandsBottomFirst) return ((a.n_tiles + kXcds - 1) / kXcds) * kXcds;
    const unsigned grid_h = a.n_tiles / a.grid_w;
    const unsigned cycles = (grid_h + kXcds * a.row_group - 1) / (kXcds * a.row_group);
    return kXcds * cycles * a.row_group * a.grid_w;
}

// Debug-only timeline (tools/k10_timeline.sh builds a separate library with -DFGS_K10_TIMELINE; the product build has none of it): per tile
// its start / end on the chip-wide 100 MHz counter, the length of its list and how far it was walked.
#ifdef FGS_K10_TIMELINE
constexpr unsigned kK10TimelineTiles = 1u << 17;
__device__ unsigned long long g_k10_timeline[kK10TimelineTiles * 4];
#endif

// Debug-only pair statistics (tools/pair_stats.sh, -DFGS_PAIR_STATS; the product build has none of it): [0] tiles, [1] instances staged, [2] (Gaussian,
// 16x4 strip) pairs walked (the union of the two sub-tile masks), [3] lanes of walked pairs whose own 8x4 sub-tile is hit and whose pixel is not finished,
// [4] lanes that blended (alpha test passed), [5] (Gaussian, strip) slots offered to the cull = 64-Gaussian chunks x 64 seen by a wave that still had a live pixel.
#ifdef FGS_PAIR_STATS
__device__ unsigned long long g_k10_pair_stats[8];
#endif
template <bool TRAINING>
__global__ void __launch_bounds__(kBlendBlock) blend_kernel(const BlendArgs a) {
    const unsigned tile = tile_of_workgroup(blockIdx.x, a.grid_w, a.n_tiles, a.row_group, a.tile_plan, a.grid_h);
    if (tile >= a.n_tiles) return;
#ifdef FGS_K10_TIMELINE
    const unsigned long long t_start_ = __builtin_amdgcn_s_memrealtime();
#endif
    const unsigned tile_x = tile % a.grid_w, tile_y = tile / a.grid_w;
    const unsigned tid = threadIdx.x, wave = tid >> 6, lane = tid & 63u, half = lane >> 5;
    const unsigned lx = half * kSubtileW + (lane & 7u), ly = wave * kSubtileH + ((lane >> 3) & 3u);
    const unsigned px = tile_x * kTileW + lx, py = tile_y * kTileH + ly;
    const bool inside = px < a.width && py < a.height;
    const unsigned local = ly * kTileW + lx;
    const float pxf = static_cast<float>(px) + 0.5f, pyf = static_cast<float>(py) + 0.5f;
    // sub-tile rectangles of this wave (kf:389-398)
    const unsigned sub_y0 = tile_y * kTileH + wave * kSubtileH, sub_y1 = sub_y0 + kSubtileH;
    const unsigned subl_x0 = tile_x * kTileW, subl_x1 = subl_x0 + kSubtileW, subr_x1 = subl_x1 + kSubtileW;

    const uint2 range = a.ranges[tile];
    const unsigned n_total = range.y - range.x;
    unsigned bucket_base = 0;
    if (TRAINING) {
        bucket_base = tile == 0 ? 0u : a.bucket_offsets[tile - 1];
        const unsigned nb = (n_total + kBucket - 1) / kBucket;
        for (unsigned b = tid; b < nb; b += kBlendBlock) a.bucket_tile[bucket_base + b] = tile;   // kf:407-411
    }

    __shared__ float4 s_rec[3 * kBlendBlock];                        // one array: the walk addresses all three rows off one register
    float4* const s_a = s_rec;                                        // mean.x mean.y conic.a conic.b
    float4* const s_b = s_rec + kBlendBlock;                          // conic.c opacity r g
    float4* const s_c = s_rec + 2 * kBlendBlock;                      // b bounds_x bounds_y -
    __shared__ unsigned s_max[kBlendBlock / kWave];

#ifdef FGS_PAIR_STATS
    unsigned st_staged = 0, st_pairs = 0, st_mine = 0, st_pass = 0, st_offered = 0;
#endif
    float cr = 0.0f, cg = 0.0f, cb = 0.0f, T = 1.0f;
    float gate = inside ? kMinAlphaThreshold : __builtin_inff();       // the alpha a pair has to reach to be blended (see the walk below)
    unsigned n_used = 0;
    // "done" (kf:424,477) is not carried as a flag: a pixel is finished exactly when its transmittance has dropped below the threshold
    // (T only ever decreases and the flag is set right after the update that takes it there), or when it lies outside the image. A
    // divergent flag carried through the per-Gaussian loop lives in a scalar register pair that has to be merged after every
    // conditional update (three scalar instructions per merge), and this loop is bound by the ONE scalar unit of the CU.
#define FGS_PIXEL_DONE (!inside || T < kTransmittanceThreshold)

    // Deep tiles fetch ahead: from its second batch on (a tile that needs one has proved to be deep; at S2 a tile walks ~100 of its ~1 500 entries and
    // never gets here) the records of the NEXT batch are loaded into registers while this batch is walked, and the primitive indices of the one after
    // that, so that the two dependent global round trips of the staging (index -> record, ~3 us per batch with the SIMD to itself) leave the serial
    // path of a long list -- the tail of K10 on object-centric / trained scenes (tools/k10_timeline.py).
    float4 n0 = make_float4(0.f, 0.f, 0.f, 0.f), n1 = n0, n2 = n0;      // records of the next batch (have_next)
    uint32_t prim_ahead = 0;                                           // primitive of the batch after next (have_ahead)
    bool have_next = false, have_ahead = false;                        // workgroup-uniform
    for (unsigned batch_start = 0; batch_start < n_total; batch_start += kBlendBlock) {
        if (__syncthreads_and(FGS_PIXEL_DONE ? 1 : 0)) break;                          // kf:424
        const unsigned batch = min(static_cast<unsigned>(kBlendBlock), n_total - batch_start);
        if (tid < batch) {
            float4 r0, r1, r2;
            if (have_next) { r0 = n0; r1 = n1; r2 = n2; }
            else {
                const uint32_t prim = a.inst_prims[range.x + batch_start + tid];
                const float4* r = reinterpret_cast<const float4*>(a.rec + prim);
                r0 = r[0]; r1 = r[1]; r2 = r[2];
            }
            s_a[tid] = r0;
            if (TRAINING) {                                                            // kf:430 (inference clamps at store, ki:200)
                s_b[tid] = make_float4(r1.x, r1.y, fmaxf(r1.z, 0.0f), fmaxf(r1.w, 0.0f));
                s_c[tid] = make_float4(fmaxf(r2.x, 0.0f), r2.y, r2.z, 0.0f);
            } else {
                s_b[tid] = r1;
                s_c[tid] = r2;
            }
        }
        __syncthreads();
        {
            const unsigned next_start = batch_start + kBlendBlock, after_start = next_start + kBlendBlock;
            const bool had_ahead = have_ahead;
            have_next = batch_start >= static_cast<unsigned>(kBlendBlock) && next_start < n_total;
            have_ahead = have_next && after_start < n_total;
            if (have_next && next_start + tid < n_total) {
                const uint32_t prim = had_ahead ? prim_ahead : a.inst_prims[range.x + next_start + tid];
                const float4* r = reinterpret_cast<const float4*>(a.rec + prim);
                n0 = r[0]; n1 = r[1]; n2 = r[2];
            }
            if (have_ahead && after_start + tid < n_total) prim_ahead = a.inst_prims[range.x + after_start + tid];
        }
        for (unsigned chunk = 0; chunk < batch; chunk += kBucket) {
            const bool done = FGS_PIXEL_DONE;
            if (TRAINING && !done)                                                     // kf:436-442, every 64 instead of 32
#if FGS_CKPT_NT
                store_float4_nt(reinterpret_cast<float*>(a.ckpt + (size_t)(bucket_base + (batch_start + chunk) / kBucket) * kTilePixels + local), make_float4(cr, cg, cb, T));
#else
                a.ckpt[(size_t)(bucket_base + (batch_start + chunk) / kBucket) * kTilePixels + local] = make_float4(cr, cg, cb, T);
#endif
            bool in_l = false, in_r = false;
            const unsigned j = chunk + lane;
            if (j < batch) {                                                           // kf:445-450
                const uint32_t bx = __float_as_uint(s_c[j].y), by = __float_as_uint(s_c[j].z);
                const unsigned x_min = bx & 0xffffu, x_max = bx >> 16, y_min = by & 0xffffu, y_max = by >> 16;
                const bool in_y = y_min < sub_y1 && sub_y0 < y_max;
                in_l = in_y && x_min < subl_x1 && subl_x0 < x_max;
                in_r = in_y && x_min < subr_x1 && subl_x1 < x_max;
            }
            const uint64_t mask_l = wave_ballot(in_l), mask_r = wave_ballot(in_r);
            const uint64_t mine = inside ? (half ? mask_r : mask_l) : 0ull;            // pixels outside the image never blend
            uint64_t pending = mask_l | mask_r;
            if (wave_ballot(!done) == 0) pending = 0;
#ifdef FGS_PAIR_STATS
            if (wave == 0) st_staged += min(static_cast<unsigned>(kBucket), batch - chunk);
            if (wave_ballot(!done) != 0) st_offered += min(static_cast<unsigned>(kBucket), batch - chunk);
#endif
            // The walk over the set bits saturates the scalar unit (ONE per CU for four SIMDs; rocprofv3 on the layered scene, round 2:
            // SQ_INSTS_SALU = SQ_INSTS_VALU = 457 M per launch), so every test of a (pixel, Gaussian) pair is ONE vector compare:
            //  * the 64-bit list is walked as two bit-reversed 32-bit words from the top (count-leading-zeros / clear on single registers);
            //  * "this Gaussian overlaps my sub-tile" (kf:445-451): the lane's complemented, reversed word shifted left by the same count
            //    has its sign bit set exactly when it does NOT, and that bit is OR-ed into alpha -- a negative alpha fails the alpha test;
            //  * "pixel not finished" (kf:424,477) is folded into the threshold: `gate` is 1/255 (kf:467) while the pixel is alive and
            //    +inf from the update that takes T below the threshold (or outside the image), refreshed where T changes.
            // Three compares, two scalar ANDs and a 64-bit vector shift before; the arithmetic is untouched (bit-identical images).
#pragma unroll
            for (unsigned word = 0; word < 2u; ++word) {
                uint32_t pend = __brev(static_cast<uint32_t>(word ? pending >> 32 : pending));
                const uint32_t not_mine = __brev(~static_cast<uint32_t>(word ? mine >> 32 : mine));
                const unsigned j0 = chunk + 32u * word;
                const unsigned row0 = in_vector_register(j0 * 16u);                     // byte offset of entry j0, kept out of the scalar unit
                // Two list entries per trip, their six LDS reads issued together in front of the first use: left to itself the compiler reads the
                // colours inside the blend branch -- a second LDS round trip per blended entry on the serial path of a long list. The two alphas are
                // independent chains, only the conditional blends are serial. Measured (tools/ab_k10_mapping.py, one box): S2 0.144 -> 0.140 ms,
                // layered 0.600 -> 0.590, bench.py's surface scene (lists of thousands walked by a few tiles at the end of the kernel) 0.387 -> 0.357;
                // three or four entries per trip are slower everywhere (0.156 / 0.162 ms at S2: the control flow, not the LDS, is what a trip pays).
                while (pend != 0) {                                                    // wave-uniform scalar loop
                    const unsigned k = static_cast<unsigned>(__clz(static_cast<int>(pend)));
                    pend &= ~(0x80000000u >> k);
                    const bool second = pend != 0u;                                    // wave-uniform
                    const unsigned k2 = second ? static_cast<unsigned>(__clz(static_cast<int>(pend))) : k;
                    pend &= ~(0x80000000u >> k2);
                    const float4* const entry = reinterpret_cast<const float4*>(reinterpret_cast<const char*>(s_rec) + (row0 + (k << 4)));
                    const float4* const entry2 = reinterpret_cast<const float4*>(reinterpret_cast<const char*>(s_rec) + (row0 + (k2 << 4)));
                    float4 ga = entry[0], gb = entry[kBlendBlock];
                    float4 ha = entry2[0], hb = entry2[kBlendBlock];
                    float blue = entry[2 * kBlendBlock].x, blue2 = entry2[2 * kBlendBlock].x;
                    asm volatile("" : "+v"(gb.z), "+v"(gb.w), "+v"(blue), "+v"(hb.z), "+v"(hb.w), "+v"(blue2), "+v"(ha.x), "+v"(hb.x));
                    const float dx = ga.x - pxf, dy = ga.y - pyf;
                    const float ex = ha.x - pxf, ey = ha.y - pyf;
                    const float power = -0.5f * (ga.z * dx * dx + gb.x * dy * dy) - ga.w * dx * dy;
                    const float power2 = -0.5f * (ha.z * ex * ex + hb.x * ey * ey) - ha.w * ex * ey;
                    const float gauss = __expf(fminf(power, 0.0f));
                    const float gauss2 = __expf(fminf(power2, 0.0f));
                    const float alpha = gb.y * gauss;
                    const float alpha2 = hb.y * gauss2;
                    const float tested = __uint_as_float(((not_mine << k) & 0x80000000u) | __float_as_uint(alpha));
                    const float tested2 = __uint_as_float(((not_mine << k2) & 0x80000000u) | __float_as_uint(alpha2));
#ifdef FGS_PAIR_STATS
                    st_pairs += second ? 2u : 1u;
                    st_mine += static_cast<unsigned>(__popcll(wave_ballot(((not_mine << k) & 0x80000000u) == 0u && gate < 1.0f)));
                    st_pass += static_cast<unsigned>(__popcll(wave_ballot(tested >= gate)));
#endif
                    if (tested >= gate) {
                        const float w = T * alpha;
                        cr += w * gb.z; cg += w * gb.w; cb += w * blue;
                        T *= 1.0f - alpha;
                        gate = T < kTransmittanceThreshold ? __builtin_inff() : gate;
                        n_used = batch_start + j0 + k + 1;                             // kf:474
                    }
                    if (second) {
#ifdef FGS_PAIR_STATS
                        st_mine += static_cast<unsigned>(__popcll(wave_ballot(((not_mine << k2) & 0x80000000u) == 0u && gate < 1.0f)));
                        st_pass += static_cast<unsigned>(__popcll(wave_ballot(tested2 >= gate)));
#endif
                        if (tested2 >= gate) {
                            const float w = T * alpha2;
                            cr += w * hb.z; cg += w * hb.w; cb += w * blue2;
                            T *= 1.0f - alpha2;
                            gate = T < kTransmittanceThreshold ? __builtin_inff() : gate;
                            n_used = batch_start + j0 + k2 + 1;
                        }
                    }
                }
            }
        }
    }
#undef FGS_PIXEL_DONE

    if (inside) {
        cr += T * a.bg[0]; cg += T * a.bg[1]; cb += T * a.bg[2];                      // kf:483
        const size_t pix = (size_t)a.width * py + px;
        const size_t n_pixels = (size_t)a.width * a.height;
        if (TRAINING) {
            a.image[pix] = cr; a.image[n_pixels + pix] = cg; a.image[2 * n_pixels + pix] = cb;
        } else {
            if (a.clamp_output) { cr = saturate_f(cr); cg = saturate_f(cg); cb = saturate_f(cb); }   // ki:445-449
            if (a.to_chw) { a.image[pix] = cr; a.image[n_pixels + pix] = cg; a.image[2 * n_pixels + pix] = cb; }
            else { a.image[3 * pix] = cr; a.image[3 * pix + 1] = cg; a.image[3 * pix + 2] = cb; }
        }
    }
    if (TRAINING) {
        // tile-major (the reference indexes these image-linear, kf:485-491; they are private to the backend)
        a.final_T[(size_t)tile * kTilePixels + local] = T;
        a.n_processed[(size_t)tile * kTilePixels + local] = n_used;                    // 0 for pixels outside the image
        const unsigned wmax = wave_max(n_used);
        if (lane == 0) s_max[wave] = wmax;
        __syncthreads();
        if (tid == 0) a.max_n_processed[tile] = max(s_max[0], max(s_max[1], s_max[2]));   // kf:493-497
    }
#ifdef FGS_PAIR_STATS
    if (TRAINING && lane == 0) {
        if (wave == 0) { atomicAdd(&g_k10_pair_stats[0], 1ull); atomicAdd(&g_k10_pair_stats[1], static_cast<unsigned long long>(st_staged)); }
        atomicAdd(&g_k10_pair_stats[2], static_cast<unsigned long long>(st_pairs)); atomicAdd(&g_k10_pair_stats[3], static_cast<unsigned long long>(st_mine));
        atomicAdd(&g_k10_pair_stats[4], static_cast<unsigned long long>(st_pass)); atomicAdd(&g_k10_pair_stats[5], static_cast<unsigned long long>(st_offered));
    }
#endif
#ifdef FGS_K10_TIMELINE
    if (tid == 0 && tile < kK10TimelineTiles) {
        g_k10_timeline[tile * 4u] = t_start_;
        g_k10_timeline[tile * 4u + 1u] = __builtin_amdgcn_s_memrealtime();
        g_k10_timeline[tile * 4u + 2u] = (static_cast<unsigned long long>(n_total) << 32) | blockIdx.x;
        g_k10_timeline[tile * 4u + 3u] = static_cast<unsigned long long>(__builtin_amdgcn_s_getreg((20 << 0) | (0 << 6) | (31 << 11)));   // XCC_ID
    }
#endif
}

#ifdef FGS_K10_TIMELINE
}  // namespace fgs
extern "C" __attribute__((visibility("default"))) int fgs_debug_k10_timeline(unsigned long long* out, unsigned n_tiles, int reset) {
    if (n_tiles > fgs::kK10TimelineTiles) n_tiles = fgs::kK10TimelineTiles;
    if (out != nullptr && hipMemcpyFromSymbol(out, HIP_SYMBOL(fgs::g_k10_timeline), sizeof(unsigned long long) * 4 * n_tiles) != hipSuccess) return -1;
    if (reset) {
        void* dev = nullptr;
        if (hipGetSymbolAddress(&dev, HIP_SYMBOL(fgs::g_k10_timeline)) != hipSuccess
            || hipMemset(dev, 0, sizeof(unsigned long long) * 4 * fgs::kK10TimelineTiles) != hipSuccess) return -1;
    }
    return 0;
}
namespace fgs {
#endif

#ifdef FGS_PAIR_STATS
}  // namespace fgs
extern "C" __attribute__((visibility("default"))) int fgs_debug_k10_pair_stats(unsigned long long* out, int reset) {
    if (out != nullptr && hipMemcpyFromSymbol(out, HIP_SYMBOL(fgs::g_k10_pair_stats), sizeof(unsigned long long) * 8) != hipSuccess) return -1;
    if (reset) {
        void* dev = nullptr;
        if (hipGetSymbolAddress(&dev, HIP_SYMBOL(fgs::g_k10_pair_stats)) != hipSuccess || hipMemset(dev, 0, sizeof(unsigned long long) * 8) != hipSuccess) return -1;
    }
    return 0;
}
namespace fgs {
#endif

// Speedy-Splat pruning scores (kernels_pruning_scores.cuh:348-505): the tile list is blended twice -- pass 1
// for the final colour / transmittance, pass 2 re-walks it with dL/dC = 1 and adds (opacity * dL/dalpha)^2 of every blended
// (pixel, Gaussian) pair to scores[primitive]. Same wave64 strip / two-sub-tile cull as blend_kernel. The reference issues
// one atomicAdd per (pixel, Gaussian) pair (kp:490); here the 64 lanes of a wave evaluate the SAME Gaussian, so their
// scores are summed with 6 DPP adds and leave through one atomic per (Gaussian, wave).
__global__ void __launch_bounds__(kBlendBlock) pruning_scores_kernel(const BlendArgs a) {
    const unsigned tile = tile_of_workgroup(blockIdx.x, a.grid_w, a.n_tiles, a.row_group, a.tile_plan, a.grid_h);
    if (tile >= a.n_tiles) return;
    const unsigned tile_x = tile % a.grid_w, tile_y = tile / a.grid_w;
    const unsigned tid = threadIdx.x, wave = tid >> 6, lane = tid & 63u, half = lane >> 5;
    const unsigned lx = half * kSubtileW + (lane & 7u), ly = wave * kSubtileH + ((lane >> 3) & 3u);
    const unsigned px = tile_x * kTileW + lx, py = tile_y * kTileH + ly;
    const bool inside = px < a.width && py < a.height;
    const float pxf = static_cast<float>(px) + 0.5f, pyf = static_cast<float>(py) + 0.5f;
    const unsigned sub_y0 = tile_y * kTileH + wave * kSubtileH, sub_y1 = sub_y0 + kSubtileH;
    const unsigned subl_x0 = tile_x * kTileW, subl_x1 = subl_x0 + kSubtileW, subr_x1 = subl_x1 + kSubtileW;
    const uint2 range = a.ranges[tile];
    const unsigned n_total = range.y - range.x;

    __shared__ float4 s_a[kBlendBlock], s_b[kBlendBlock], s_c[kBlendBlock];
    __shared__ uint32_t s_prim[kBlendBlock];
    float c0 = 0.0f, c1 = 0.0f, c2 = 0.0f, T = 1.0f, galpha = 0.0f;

    for (int pass = 0; pass < 2; ++pass) {
        if (pass == 1) { galpha = T * -(a.bg[0] + a.bg[1] + a.bg[2]); T = 1.0f; }                 // kp:441-444
        // "done" derived from T as in blend_kernel (the loop is bound by the scalar unit; a carried divergent flag costs scalar merges)
#define FGS_PIXEL_DONE (!inside || T < kTransmittanceThreshold)
        for (unsigned batch_start = 0; batch_start < n_total; batch_start += kBlendBlock) {
            if (__syncthreads_and(FGS_PIXEL_DONE ? 1 : 0)) break;
            const unsigned batch = min(static_cast<unsigned>(kBlendBlock), n_total - batch_start);
            if (tid < batch) {
                const uint32_t prim = a.inst_prims[range.x + batch_start + tid];
                const float4* r = reinterpret_cast<const float4*>(a.rec + prim);
                s_a[tid] = r[0]; s_b[tid] = r[1]; s_c[tid] = r[2]; s_prim[tid] = prim;
            }
            __syncthreads();
            for (unsigned chunk = 0; chunk < batch; chunk += kWave) {
                bool in_l = false, in_r = false;
                const unsigned j = chunk + lane;
                if (j < batch) {
                    const uint32_t bx = __float_as_uint(s_c[j].y), by = __float_as_uint(s_c[j].z);
                    const unsigned x_min = bx & 0xffffu, x_max = bx >> 16, y_min = by & 0xffffu, y_max = by >> 16;
                    const bool in_y = y_min < sub_y1 && sub_y0 < y_max;
                    in_l = in_y && x_min < subl_x1 && subl_x0 < x_max;
                    in_r = in_y && x_min < subr_x1 && subl_x1 < x_max;
                }
                const uint64_t mask_l = wave_ballot(in_l), mask_r = wave_ballot(in_r);
                const uint64_t mine = half ? mask_r : mask_l;
                uint64_t pending = mask_l | mask_r;
                if (wave_ballot(!FGS_PIXEL_DONE) == 0) pending = 0;
                while (pending != 0) {
                    const int k = __ffsll(static_cast<unsigned long long>(pending)) - 1;
                    pending &= pending - 1;
                    const unsigned jj = chunk + static_cast<unsigned>(k);
                    const float4 ga = s_a[jj], gb = s_b[jj];
                    const float col2 = s_c[jj].x;
                    const float dx = ga.x - pxf, dy = ga.y - pyf;
                    const float power = -0.5f * (ga.z * dx * dx + gb.x * dy * dy) - ga.w * dx * dy;
                    const float alpha = gb.y * __expf(fminf(power, 0.0f));
                    const bool contrib = inside && ((mine >> k) & 1ull) != 0 && T >= kTransmittanceThreshold && alpha >= kMinAlphaThreshold;
                    float score = 0.0f;
                    if (contrib) {
                        const float w = T * alpha;
                        if (pass == 0) { c0 += w * gb.z; c1 += w * gb.w; c2 += w * col2; }
                        else {
                            c0 -= w * gb.z; c1 -= w * gb.w; c2 -= w * col2;                          // kp:477
                            const float rcp = fast_rcp(fmaxf(1.0f - alpha, kOneMinusAlphaEps));
                            const float dl_dalpha = ((T * gb.z - c0 * rcp) + (T * gb.w - c1 * rcp) + (T * col2 - c2 * rcp)) + galpha * rcp;
                            const float dl_dg = gb.y * dl_dalpha;
                            score = dl_dg * dl_dg;                                                   // kp:487-488
                        }
                        T *= 1.0f - alpha;
                    }
                    if (pass == 1 && wave_ballot(contrib) != 0) {                                   // wave-uniform
                        const float total = wave_sum_to_lane63(score);
                        if (lane == 63u) unsafeAtomicAdd(a.scores + s_prim[jj], total);             // kp:490, one per (Gaussian, wave)
                    }
                }
            }
        }
    }
#undef FGS_PIXEL_DONE
}

hipError_t launch_pruning_scores(const BlendArgs& a_in, hipStream_t s) {
    BlendArgs a = a_in;                          // row_group: set by the caller (api.hip reads the switch once per pass)
    if (a.tile_plan == nullptr && (a.row_group == kPlannedBlocks || a.row_group == kBandsThroughPlan)) a.row_group = 0u;
    hipLaunchKernelGGL(pruning_scores_kernel, dim3(blend_grid(a)), dim3(kBlendBlock), 0, s, a);
    return hipGetLastError();
}

hipError_t launch_blend(bool training, const BlendArgs& a_in, hipStream_t s) {
    BlendArgs a = a_in;                          // row_group: set by the caller (api.hip reads the switch once per pass)
    if (a.tile_plan == nullptr && (a.row_group == kPlannedBlocks || a.row_group == kBandsThroughPlan)) a.row_group = 0u;          // no plan was made: the bands
    const dim3 grid(blend_grid(a)), block(kBlendBlock);
    if (training) hipLaunchKernelGGL(blend_kernel<true>, grid, block, 0, s, a);
    else hipLaunchKernelGGL(blend_kernel<false>, grid, block, 0, s, a);
    return hipGetLastError();
}

}  // namespace fgs
