// Gaussian-sharded multi-GPU path: the three small kernels either side of the two xGMI exchanges.
//
// The reference is single-GPU (Renderer.py:58-61). With parameters replicated, a data-parallel step has to move the
// whole 236-B/Gaussian gradient and parameter set over xGMI every iteration (708 MB at 3 M Gaussians).
// Here every rank OWNS N/G Gaussians (parameters + Adam moments), projects them for all G views of the step (K1), and
// ships only what the renderer of a view needs: one 56-byte projected record per VISIBLE Gaussian. The renderer sends
// back the 36-byte pixel-space gradient accumulators, and the owner finishes K12 + Adam on its shard. Wire volume per
// step drops from 2 x 236 B x N to (56 + 36) B x V.
//
//   owner:    K1 on the shard  -> pack_splat_records  ==all-to-all==>  unpack_splat_records -> K2..K10   :renderer
//   owner:    K12 (reads the records in place through K1's slot table), K13  <==all-to-all==  pack_acc <- K11  :renderer
#include "fgs_kernels.h"
#include <fgs_wave.h>

namespace fgs {

static_assert(kSplatRecordWords == 14 && sizeof(PrimRec) == 48, "a splat record is the 48-byte PrimRec + depth key + tile count");

// One lane per visible Gaussian j of a (shard, view); grid.y = view; `counts_out` = (V, I). Record order = the order K1
// compacted the visible list in, with one exception: K1 appends the huge-footprint Gaussians (thousands of tiles each -- the
// heavy hitters of K11's atomics) as ONE run at the end, and neighbours in record order share the 128-byte lines of the
// accumulators on the renderer (measured with one plane per sum, rounds 2-3: K11 0.71 -> 0.97 ms on views with ~200 of them). They trade places with
// evenly spaced regular records instead. slot[i] = record index of visible primitive i, for fgs_shard_backward.
__global__ void __launch_bounds__(256) pack_splat_records_kernel(const PackRecordsBatch b) {
    const PackRecordsView& w = b.v[blockIdx.y];
    const uint32_t n_visible = w.counters[0];
    const uint32_t j = blockIdx.x * 256u + threadIdx.x;
    if (j == 0) { w.counts_out[0] = n_visible; w.counts_out[1] = w.counters[1]; }
    if (j >= n_visible) return;
    const uint32_t n_heavy = w.counters[2], n_regular = n_visible - n_heavy;
    const uint32_t stride = n_heavy != 0 ? n_regular / n_heavy : 0u;
    uint32_t dst = j;
    if (stride >= 64u) {
        if (j >= n_regular) dst = (j - n_regular) * stride + stride / 2u;                                  // heavy #a -> a spread slot
        else if (j % stride == stride / 2u && j / stride < n_heavy) dst = n_regular + j / stride;           // its former tenant -> the end
    }
    const uint32_t i = w.prim_idx[j];
    w.slot[i] = dst;
    const uint4* r = reinterpret_cast<const uint4*>(w.rec + i);
    const uint4 a = r[0], c = r[1], d = r[2];
    uint2* o = reinterpret_cast<uint2*>(w.out + (size_t)kSplatRecordWords * dst);    // 56-byte records: 8-byte aligned
    o[0] = make_uint2(a.x, a.y); o[1] = make_uint2(a.z, a.w);
    o[2] = make_uint2(c.x, c.y); o[3] = make_uint2(c.z, c.w);
    o[4] = make_uint2(d.x, d.y); o[5] = make_uint2(d.z, d.w);
    o[6] = make_uint2(w.depth_keys[j], w.n_touched[i]);
}

// Where does record j of the concatenation (all of shard 0, then shard 1, ...) go on the renderer? Owners hold the Gaussians r, r + G, r + 2 G ... of a
// Morton-ordered scene, so the shard-major concatenation puts Morton neighbours -- Gaussians of the same tiles -- V / G records apart, and K11 on it
// measured 0.508 ms against 0.433 in memory order (S2, G = 8, one GPU, profiles/r04_ab_sharded_order.txt). The renderer therefore INTERLEAVES the
// shards again, inside the two passes it runs anyway (this unpack, and pack_acc on the way back): record with rank r in shard s goes to primitive
// slot sum_s' min(count[s'], r + [s' < s]) -- r G + s when all shards sent the same number, and a bijection onto 0..n-1 for any counts.
__device__ __forceinline__ uint32_t renderer_slot(const uint32_t j, const ShardOrder& o) {
    if (o.n_shards <= 1) return j;
    uint32_t s = 0, first = 0, end = 0;                                      // shard of record j = number of segments that end at or before j
#pragma unroll
    for (int k = 0; k < kMaxBatchViews; ++k) {
        if (k < o.n_shards) {
            end += o.count[k];
            if (j >= end) { s = static_cast<uint32_t>(k) + 1u; first = end; }
        }
    }
    const uint32_t r = j - first;
    uint32_t dst = 0;
#pragma unroll
    for (int k = 0; k < kMaxBatchViews; ++k)
        if (k < o.n_shards) dst += min(o.count[k], r + (static_cast<uint32_t>(k) < s ? 1u : 0u));
    return dst;
}

// The inverse: which record of the concatenation becomes primitive slot d? Ranks below r fill f(r) = sum_s' min(count[s'], r) slots (monotone in r):
// r = the largest rank with f(r) <= d (binary search, 8 terms per probe), and the shard is the (d - f(r))-th of those that have a record of rank r.
// Both passes are written per SLOT (thread = slot: coalesced accesses to the renderer's arrays, and lanes 8 apart touch consecutive records of one
// shard's segment): per record they measured 0.090 ms for the two passes against 0.040 as received -- more than K11 gained.
__device__ __forceinline__ uint32_t record_of_slot(const uint32_t d, const ShardOrder& o) {
    if (o.n_shards <= 1) return d;
    uint32_t lo = 0, hi = 0;
#pragma unroll
    for (int k = 0; k < kMaxBatchViews; ++k) if (k < o.n_shards) hi = max(hi, o.count[k]);
    while (lo < hi) {                                                          // largest r in [0, max count] with f(r) <= d
        const uint32_t mid = lo + (hi - lo + 1u) / 2u;
        uint32_t f = 0;
#pragma unroll
        for (int k = 0; k < kMaxBatchViews; ++k) if (k < o.n_shards) f += min(o.count[k], mid);
        if (f <= d) lo = mid; else hi = mid - 1u;
    }
    const uint32_t r = lo;
    uint32_t f = 0;
#pragma unroll
    for (int k = 0; k < kMaxBatchViews; ++k) if (k < o.n_shards) f += min(o.count[k], r);
    uint32_t left = d - f, first = 0, j = 0;
    bool found = false;
#pragma unroll
    for (int k = 0; k < kMaxBatchViews; ++k) {
        if (k < o.n_shards) {
            if (!found && o.count[k] > r) {
                if (left == 0u) { j = first + r; found = true; }
                else --left;
            }
            first += o.count[k];
        }
    }
    return j;
}

// Renderer side: the concatenated records of all shards become primitives 0..n-1 of a pipeline that starts at K2.
// Also K0 (clears the per-tile ranges, which K1 does on the single-GPU path).
__global__ void __launch_bounds__(256) unpack_splat_records_kernel(const uint32_t* __restrict__ records, uint32_t n, PrimRec* __restrict__ rec,
                                                                   uint32_t* __restrict__ n_touched, uint32_t* __restrict__ depth_keys,
                                                                   uint32_t* __restrict__ prim_idx, uint4* __restrict__ foot, uint2* __restrict__ ranges, uint32_t n_tiles,
                                                                   uint32_t* __restrict__ hot_list, uint32_t* __restrict__ hot_count, const ShardOrder order) {
    const uint32_t dst = blockIdx.x * 256u + threadIdx.x;                     // thread = primitive slot; its record is gathered
    for (uint32_t t = dst; t < n_tiles; t += gridDim.x * 256u) ranges[t] = make_uint2(0u, 0u);
    const bool in_range = dst < n;
    uint2 w0{}, w1{}, w2{}, w3{}, w4{}, w5{}, w6{};
    if (in_range) {
        const uint32_t j = record_of_slot(dst, order);
        const uint2* m = reinterpret_cast<const uint2*>(records + (size_t)kSplatRecordWords * j);
        w0 = m[0]; w1 = m[1]; w2 = m[2]; w3 = m[3]; w4 = m[4]; w5 = m[5]; w6 = m[6];
    }
    // hot-accumulator slots are local to a pipeline: the owners' slot numbers mean nothing here, the renderer hands out its own
    unsigned tx0, tx1, ty0, ty1;
    tile_rect(w4.y, w5.x, tx0, tx1, ty0, ty1);
    const unsigned n_max = (tx1 - tx0) * (ty1 - ty0);
    const bool hot = in_range && n_max > kHotFootprint;
    uint32_t slot_word = n_max <= 32u ? w5.y : 0u;
    const uint64_t hot_mask = wave_ballot(hot);
    if (hot_mask != 0) {
        const unsigned lane = lane_id();
        const int leader = __ffsll(static_cast<unsigned long long>(hot_mask)) - 1;
        unsigned base = 0;
        if (lane == static_cast<unsigned>(leader)) base = atomicAdd(hot_count, static_cast<unsigned>(__popcll(static_cast<unsigned long long>(hot_mask))));
        base = wave_read(base, leader);
        const unsigned slot = base + lanes_below(hot_mask);
        if (hot && slot < kMaxHot) { hot_list[slot] = dst; slot_word = slot + 1u; }
    }
    if (!in_range) return;
    uint4* r = reinterpret_cast<uint4*>(rec + dst);
    r[0] = make_uint4(w0.x, w0.y, w1.x, w1.y);
    r[1] = make_uint4(w2.x, w2.y, w3.x, w3.y);
    r[2] = make_uint4(w4.x, w4.y, w5.x, slot_word);
    depth_keys[dst] = w6.x; prim_idx[dst] = dst; n_touched[dst] = w6.y;   // the visible list in slot order: equal depth keys keep that order through the stable sort
    // footprint row (fgs_math.h): a record carries the 32-bit overlap bitmap of boxes of <= 32 candidates; larger boxes are re-tested by the instance kernel
    const bool bitmap = n_max <= 32u && footprint_box_fits(tx0, ty0, tx1 - tx0, n_max);
    foot[dst] = bitmap ? make_uint4(dst, footprint_box(tx0, ty0, tx1 - tx0, ty1 - ty0), w5.y, 0u) : make_uint4(dst, kFootprintEscape, w6.y, n_max);
}

// K11's accumulator records [n][9] (by primitive slot) -> the same 36-byte records in the order of the concatenation (record j), ready to be cut
// into per-shard segments
__global__ void __launch_bounds__(256) pack_acc_kernel(const float* __restrict__ acc, uint32_t n, float* __restrict__ out, const ShardOrder order) {
    const uint32_t src = blockIdx.x * 256u + threadIdx.x;                     // thread = primitive slot: the wave reads 64 consecutive records, each is scattered
    if (src >= n) return;
    const uint32_t j = record_of_slot(src, order);
    float v[kAccRecordWords];
#pragma unroll
    for (int k = 0; k < kAccRecordWords; ++k) v[k] = acc[(size_t)src * kAccRecordWords + k];
#pragma unroll
    for (int k = 0; k < kAccRecordWords; ++k) out[(size_t)kAccRecordWords * j + k] = v[k];
}

hipError_t launch_pack_splat_records(const PackRecordsBatch& b, hipStream_t s) {
    if (b.n_views <= 0) return hipSuccess;
    const dim3 grid(b.capacity == 0 ? 1u : (b.capacity + 255u) / 256u, static_cast<unsigned>(b.n_views)), block(256);
    hipLaunchKernelGGL(pack_splat_records_kernel, grid, block, 0, s, b);
    return hipGetLastError();
}

hipError_t launch_unpack_splat_records(const uint32_t* records, uint32_t n, PrimRec* rec, uint32_t* n_touched, uint32_t* depth_keys,
                                       uint32_t* prim_idx, uint4* foot, uint2* ranges, uint32_t n_tiles, uint32_t* hot_list, uint32_t* hot_count, const ShardOrder& order,
                                       hipStream_t s) {
    const dim3 grid(n == 0 ? 1u : (n + 255u) / 256u), block(256);
    hipLaunchKernelGGL(unpack_splat_records_kernel, grid, block, 0, s, records, n, rec, n_touched, depth_keys, prim_idx, foot, ranges, n_tiles, hot_list, hot_count, order);
    return hipGetLastError();
}

hipError_t launch_pack_acc(const float* acc, uint32_t n, float* out, const ShardOrder& order, hipStream_t s) {
    if (n == 0) return hipSuccess;
    hipLaunchKernelGGL(pack_acc_kernel, dim3((n + 255u) / 256u), dim3(256), 0, s, acc, n, out, order);
    return hipGetLastError();
}

}  // namespace fgs
