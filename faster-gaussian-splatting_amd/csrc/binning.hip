// K2-K9: tile binning for gfx950.
//   depth sort (32-bit keys) -> exclusive scan of tile counts in depth order -> instance creation (exact overlap)
//   -> stable tile sort on end_bit bits -> per-tile [start,end) ranges -> inclusive scan of per-tile bucket counts.
// Semantics: reference rasterization/src/forward.cu:104-231 + kernels_forward.cuh:211-360 (two-stage "Splatshop" sort,
// no 64-bit tile|depth key). The two sorts are radix_sort.hip (rocPRIM's onesweep stays selectable for A/B runs), the scans
// use rocPRIM (the native AMD device primitives); the gather of K3
// (apply_depth_ordering_cu) and the bucket-count kernel K8 are folded into the scans as transform iterators, so two
// kernel launches and two V/T-sized round trips through HBM disappear.
// Built with -ffp-contract=off (the exact-overlap test must agree bit-for-bit with the one in preprocess.hip).
#include "fgs_kernels.h"
#include <fgs_wave.h>
#include "fgs_tile_scan.h"
#include <cstring>
#include <rocprim/device/device_radix_sort.hpp>
#include <rocprim/device/device_scan.hpp>
#include <rocprim/iterator/counting_iterator.hpp>
#include <rocprim/iterator/transform_iterator.hpp>
#include <rocprim/types/double_buffer.hpp>

namespace fgs {

// ---- K2-K4 -------------------------------------------------------------------------------------------------
struct TouchedFromRec {                  // offsets input: n_touched of the i-th primitive in depth order (kf:211-221)
    const uint32_t* n_touched;           // compact 4-byte array (L2-resident gather), not the 48-byte records
    __host__ __device__ uint32_t operator()(uint32_t prim) const { return n_touched[prim]; }
};

size_t depth_sort_temp_bytes(uint32_t n) {
    size_t sort_bytes = 0, scan_bytes = 0;
    rocprim::double_buffer<uint32_t> k(nullptr, nullptr), v(nullptr, nullptr);
    (void)rocprim::radix_sort_pairs(nullptr, sort_bytes, k, v, n, 0u, 32u);
    const size_t own = own_sort_temp_bytes(n, 32);                         // radix_sort.hip; either implementation fits
    sort_bytes = own > sort_bytes ? own : sort_bytes;
    auto in = rocprim::make_transform_iterator(static_cast<const uint32_t*>(nullptr), TouchedFromRec{nullptr});
    (void)rocprim::exclusive_scan(nullptr, scan_bytes, in, static_cast<uint32_t*>(nullptr), 0u, n, rocprim::plus<uint32_t>());
    return sort_bytes > scan_bytes ? sort_bytes : scan_bytes;
}

hipError_t run_depth_sort(void* temp, size_t temp_bytes, uint32_t* keys[2], uint32_t* vals[2], int& selector,
                          uint32_t n_visible, DepthKeyRange range, hipStream_t s) {
    selector = 0;
    if (n_visible == 0) return hipSuccess;
    if (g_sort_implementation & 2) return own_depth_sort(temp, temp_bytes, keys, vals, selector, n_visible, nullptr, range, s);
    rocprim::double_buffer<uint32_t> k(keys[0], keys[1]), v(vals[0], vals[1]);
    hipError_t e = rocprim::radix_sort_pairs(temp, temp_bytes, k, v, n_visible, 0u, 32u, s);
    if (e != hipSuccess) return e;
    selector = (v.current() == vals[0]) ? 0 : 1;
    return hipSuccess;
}

bool depth_sort_takes_device_count() { return (g_sort_implementation & 2) != 0; }
hipError_t run_depth_sort_device_count(void* temp, size_t temp_bytes, uint32_t* keys[2], uint32_t* vals[2], int& selector,
                                       uint32_t capacity, const uint32_t* n_visible_ptr, DepthKeyRange range, hipStream_t s) {
    return own_depth_sort(temp, temp_bytes, keys, vals, selector, capacity, n_visible_ptr, range, s);
}

// the same input when the host does not know the visible count: entries at and beyond *count contribute nothing
struct TouchedGuarded {
    const uint32_t* sorted_prims; const uint32_t* n_touched; const uint32_t* count;
    __host__ __device__ uint32_t operator()(uint32_t i) const { return i < *count ? n_touched[sorted_prims[i]] : 0u; }
};

hipError_t run_offsets_scan(void* temp, size_t temp_bytes, const uint32_t* sorted_prims, const uint32_t* n_touched, uint32_t* offsets,
                            uint32_t n_visible, const uint32_t* n_visible_ptr, hipStream_t s) {
    if (n_visible == 0) return hipSuccess;
    if (n_visible_ptr != nullptr) {       // n_visible is a bound (the primitive count), the exact count lives on the device
        auto in = rocprim::make_transform_iterator(rocprim::counting_iterator<uint32_t>(0u), TouchedGuarded{sorted_prims, n_touched, n_visible_ptr});
        return rocprim::exclusive_scan(temp, temp_bytes, in, offsets, 0u, n_visible, rocprim::plus<uint32_t>(), s);
    }
    auto in = rocprim::make_transform_iterator(sorted_prims, TouchedFromRec{n_touched});
    return rocprim::exclusive_scan(temp, temp_bytes, in, offsets, 0u, n_visible, rocprim::plus<uint32_t>(), s);
}

// ---- K5 ----------------------------------------------------------------------------------------------------
// Emits (tile key, primitive) for every exactly-overlapped tile of every depth-sorted visible primitive, in row-major
// order over its tile bounding box (kf:225-328). CDNA4 shape: a wave owns 64 consecutive primitives, whose outputs form
// ONE contiguous range [offset(first), offset(last)+n). Small footprints (<= 32 candidate tiles -- the common case) carry
// their exact-overlap bitmap from preprocess, so nothing is re-tested: the wave walks its output range 64 slots at a
// time, each lane finds the owning primitive by a 6-step search over the wave's offsets in LDS and decodes the r-th set
// bit of its bitmap -- every store instruction covers 64 consecutive slots. Larger footprints are re-tested by the
// whole wave, 64 candidate tiles per step, with ballot-prefix write slots (also consecutive).
__device__ __forceinline__ unsigned nth_set_bit(uint32_t m, unsigned r) {   // position of the r-th (0-based) set bit of m
    unsigned pos = 0;
#pragma unroll
    for (unsigned w = 16; w >= 1; w >>= 1) {
        const unsigned c = static_cast<unsigned>(__popc(m & ((1u << w) - 1u)));
        if (r >= c) { r -= c; m >>= w; pos += w; }
    }
    return pos;
}

template <typename KeyT>
__global__ void __launch_bounds__(kInstanceBlock) create_instances_kernel(
    const uint32_t* __restrict__ sorted_prims, const uint32_t* __restrict__ offsets, const uint32_t* __restrict__ n_touched,
    const PrimRec* __restrict__ rec, KeyT* __restrict__ inst_keys, uint32_t* __restrict__ inst_prims, const uint32_t grid_w,
    const uint32_t n_visible_value, const uint32_t* __restrict__ n_visible_ptr, const uint32_t capacity, uint32_t* __restrict__ counters,
    uint32_t* __restrict__ big_list, uint32_t* __restrict__ big_count) {
    // The visible count by value, or (host-synchronisation-free forward) through n_visible_ptr with the grid sized by a bound. `capacity` =
    // size of the instance arrays: exact in the first case; in the second a caller-side estimate -- stores beyond it are dropped, and
    // the clamped instance count / an overflow flag are left in counters[5] / counters[6] for the sort and for the caller to check.
    const uint32_t n_visible = n_visible_ptr != nullptr ? *n_visible_ptr : n_visible_value;
    if (n_visible_ptr != nullptr && blockIdx.x == 0 && threadIdx.x == 0) {
        const uint32_t n_instances = counters[1];
        counters[5] = n_instances < capacity ? n_instances : capacity;
        counters[6] = n_instances > capacity ? 1u : 0u;
    }
    constexpr int kWaves = kInstanceBlock / kWave;
    __shared__ uint32_t s_off[kWaves][kWave];         // global write offset of each primitive
    __shared__ uint32_t s_loc[kWaves][kWave];         // wave-local slot of each bitmap primitive's first instance
    __shared__ uint32_t s_mask[kWaves][kWave];        // overlap bitmap (0: large footprint or padding lane)
    __shared__ uint32_t s_prim[kWaves][kWave];
    __shared__ uint32_t s_org[kWaves][kWave];         // tile-box origin: tx0 | ty0 << 16
    __shared__ uint32_t s_div[kWaves][kWave];         // box width | ceil(2^16 / width) << 8   (width <= 32)

    const unsigned gid = blockIdx.x * kInstanceBlock + threadIdx.x;
    const unsigned lane = lane_id(), wv = threadIdx.x >> 6;
    const bool active = gid < n_visible;
    if (wave_ballot(active) == 0) return;              // wave-uniform; waves are independent (no workgroup barrier)
    const unsigned i = active ? gid : n_visible - 1;
    const uint32_t prim = sorted_prims[i];
    const float4 r2 = reinterpret_cast<const float4*>(rec + prim)[2];     // colour.b, bounds x, bounds y, overlap bitmap
    unsigned tx0, tx1, ty0, ty1;
    tile_rect(__float_as_uint(r2.y), __float_as_uint(r2.z), tx0, tx1, ty0, ty1);
    const unsigned tbw = tx1 - tx0;
    const uint32_t mask = (active && tbw * (ty1 - ty0) <= 32u) ? __float_as_uint(r2.w) : 0u;   // larger footprints keep a hot-accumulator slot there
    const uint32_t my_off = offsets[i];
    // No read of n_touched[prim]: that second random gather (a 128-byte line per primitive for 4 bytes, like the record's) made this kernel
    // fetch 528 MB per launch at S2 for ~100 MB of input (rocprofv3 FETCH_SIZE, round 2). A bitmap footprint's count is the number of set
    // bits (preprocess wrote exactly the overlapped tiles), and every entry of the visible list has at least one tile (preprocess appends
    // only cnt > 0: visible = active && cnt > 0, kf:190), which is all the other two paths need to know.

    // Bitmap primitives of this wave get wave-local output slots: prefix of their counts. Walking these (not the global
    // range) keeps the loop proportional to what is written here -- a wave holding screen-filling Gaussians would otherwise
    // step over tens of thousands of slots that belong to the other two paths.
    const uint32_t n_small = static_cast<uint32_t>(__popc(mask));
    const uint32_t local = wave_exclusive_sum(n_small);
    const uint32_t total_small = wave_sum(n_small);
    s_off[wv][lane] = my_off;
    s_loc[wv][lane] = local;      // non-decreasing; a lane without bitmap shares its value with the next bitmap lane, which wins ties
    s_mask[wv][lane] = mask;
    s_prim[wv][lane] = prim;
    s_org[wv][lane] = tx0 | (ty0 << 16);
    s_div[wv][lane] = tbw | (((65536u + tbw - 1u) / (tbw ? tbw : 1u)) << 8);
    wave_lds_fence();

    // ---- small footprints: walk the wave's local output slots, 64 per step ----
    for (uint32_t sl = lane; sl < total_small; sl += kWave) {
        unsigned lo = 0;                               // largest lane index with s_loc[lo] <= sl; ties resolve to the bitmap lane
#pragma unroll
        for (unsigned step = 32; step >= 1; step >>= 1) {
            const unsigned mid = lo + step;
            if (mid < 64u && s_loc[wv][mid] <= sl) lo = mid;
        }
        const uint32_t m = s_mask[wv][lo];
        const unsigned rank = sl - s_loc[wv][lo];
        const unsigned pos = nth_set_bit(m, rank);
        const uint32_t dv = s_div[wv][lo], org = s_org[wv][lo];
        const unsigned w = dv & 0xffu, row = (pos * (dv >> 8)) >> 16, col = pos - row * w;
        const unsigned tx = (org & 0xffffu) + col, ty = (org >> 16) + row;
        const uint32_t o = s_off[wv][lo] + rank;
        if (o < capacity) {
            inst_keys[o] = static_cast<KeyT>(ty * grid_w + tx);
            inst_prims[o] = s_prim[wv][lo];
        }
    }

    // ---- medium footprints (33 .. kBigInstanceFootprint candidate tiles): re-tested by this wave, 64 candidates per step, with
    // ballot-prefix write slots (kf:283-326) ----
    const unsigned count = tbw * (ty1 - ty0);
    const bool recompute = active && mask == 0u;
    uint64_t pending = wave_ballot(recompute && count <= kBigInstanceFootprint);
    if (pending != 0) {
        const float4* rr = reinterpret_cast<const float4*>(rec + prim);
        const float4 r0 = rr[0], r1 = rr[1];
        const TileTest tt = make_tile_test(r0.x - 0.5f, r0.y - 0.5f, r0.z, r0.w, r1.x, logf(r1.y * kMinAlphaThresholdRcp));   // kf:267
        while (pending != 0) {
            const int src = __ffsll(static_cast<unsigned long long>(pending)) - 1;
            pending &= pending - 1;
            const unsigned o_tx0 = wave_read(tx0, src), o_ty0 = wave_read(ty0, src);
            const unsigned o_tbw = wave_read(tbw, src), o_cnt = wave_read(count, src);
            TileTest ot;
            ot.sx = wave_read(tt.sx, src); ot.sy = wave_read(tt.sy, src);
            ot.ca = wave_read(tt.ca, src); ot.cb = wave_read(tt.cb, src); ot.cc = wave_read(tt.cc, src); ot.pt = wave_read(tt.pt, src);
            ot.den_x = wave_read(tt.den_x, src); ot.den_y = wave_read(tt.den_y, src);
            ot.rcp_x = wave_read(tt.rcp_x, src); ot.rcp_y = wave_read(tt.rcp_y, src);
            const unsigned o_prim = wave_read(prim, src);
            unsigned o_w = wave_read(my_off, src);
            for (unsigned base = 0; base < o_cnt; base += kWave) {
                const unsigned t = base + lane;
                const unsigned tx = o_tx0 + t % o_tbw, ty = o_ty0 + t / o_tbw;
                const bool hit = t < o_cnt && tile_contributes(ot, tx, ty);
                const uint64_t hits = wave_ballot(hit);
                if (hit) {
                    const unsigned slot = o_w + lanes_below(hits);
                    if (slot < capacity) {
                        inst_keys[slot] = static_cast<KeyT>(ty * grid_w + tx);
                        inst_prims[slot] = o_prim;
                    }
                }
                o_w += static_cast<unsigned>(__popcll(static_cast<unsigned long long>(hits)));
            }
        }
    }

    // ---- huge footprints go to a work list: depth order puts the nearest (largest) Gaussians next to each other, so
    // finishing them here would serialise tens of thousands of candidate tiles in a handful of waves (measured: +0.2 ms on
    // views with screen-filling Gaussians). A second kernel gives each of them a whole workgroup. ----
    const bool is_big = recompute && count > kBigInstanceFootprint;
    const uint64_t big_mask = wave_ballot(is_big);
    if (big_mask != 0) {
        const int leader = __ffsll(static_cast<unsigned long long>(big_mask)) - 1;
        unsigned base = 0;
        if (lane == static_cast<unsigned>(leader)) base = atomicAdd(big_count, static_cast<unsigned>(__popcll(static_cast<unsigned long long>(big_mask))));
        base = wave_read(base, leader);
        if (is_big) big_list[base + lanes_below(big_mask)] = i;
    }
}

// One workgroup per large footprint: 256 candidate tiles per step, exact test (kf:283-326), write slots from a
// ballot prefix inside each wave and an LDS prefix across the 4 waves -- consecutive slots, stable row-major order.
template <typename KeyT>
__global__ void __launch_bounds__(kInstanceBlock) create_instances_big_kernel(
    const uint32_t* __restrict__ sorted_prims, const uint32_t* __restrict__ offsets, const PrimRec* __restrict__ rec,
    const uint32_t* __restrict__ big_list, const uint32_t* __restrict__ big_count,
    KeyT* __restrict__ inst_keys, uint32_t* __restrict__ inst_prims, const uint32_t grid_w, const uint32_t capacity) {
    constexpr int kWaves = kInstanceBlock / kWave;
    __shared__ unsigned s_hits[2][kWaves];
    const unsigned lane = lane_id(), wv = threadIdx.x >> 6;
    const unsigned n_big = *big_count;
    for (unsigned b = blockIdx.x; b < n_big; b += gridDim.x) {            // workgroup-uniform loop
        const uint32_t i = big_list[b];
        const uint32_t prim = sorted_prims[i];
        const float4* r = reinterpret_cast<const float4*>(rec + prim);
        const float4 r0 = r[0], r1 = r[1], r2 = r[2];
        const TileTest tt = make_tile_test(r0.x - 0.5f, r0.y - 0.5f, r0.z, r0.w, r1.x, logf(r1.y * kMinAlphaThresholdRcp));   // kf:267
        unsigned tx0, tx1, ty0, ty1;
        tile_rect(__float_as_uint(r2.y), __float_as_uint(r2.z), tx0, tx1, ty0, ty1);
        const unsigned tbw = tx1 - tx0;
        const unsigned count = tbw * (ty1 - ty0);
        unsigned w = offsets[i];
        unsigned parity = 0;
        for (unsigned base = 0; base < count; base += kInstanceBlock, parity ^= 1u) {
            const unsigned t = base + threadIdx.x;
            const unsigned tx = tx0 + t % tbw, ty = ty0 + t / tbw;
            const bool hit = t < count && tile_contributes(tt, tx, ty);
            const uint64_t hits = wave_ballot(hit);
            if (lane == 0) s_hits[parity][wv] = static_cast<unsigned>(__popcll(static_cast<unsigned long long>(hits)));
            __syncthreads();                                                       // double-buffered counts: one barrier per step
            unsigned before = 0, total = 0;
#pragma unroll
            for (int k = 0; k < kWaves; ++k) { const unsigned c = s_hits[parity][k]; total += c; if (k < static_cast<int>(wv)) before += c; }
            if (hit) {
                const unsigned slot = w + before + lanes_below(hits);
                if (slot < capacity) {
                    inst_keys[slot] = static_cast<KeyT>(ty * grid_w + tx);
                    inst_prims[slot] = prim;
                }
            }
            w += total;
        }
        __syncthreads();                                                           // s_hits is reused by the next footprint
    }
}

hipError_t launch_create_instances(int key_bytes, const uint32_t* sorted_prims, const uint32_t* offsets, const uint32_t* n_touched,
                                   const PrimRec* rec, void* inst_keys, uint32_t* inst_prims, uint32_t grid_w, uint32_t n_visible,
                                   const uint32_t* n_visible_ptr, uint32_t capacity, uint32_t* counters,
                                   uint32_t* big_list, uint32_t* big_count, hipStream_t s) {
    if (n_visible == 0) return hipSuccess;
    const dim3 grid((n_visible + kInstanceBlock - 1) / kInstanceBlock), block(kInstanceBlock);
    const dim3 big_grid(n_visible < 1024u ? n_visible : 1024u);       // grid-stride over the (short) device-side work list
    if (key_bytes == 2) {
        hipLaunchKernelGGL(create_instances_kernel<uint16_t>, grid, block, 0, s, sorted_prims, offsets, n_touched, rec,
                           static_cast<uint16_t*>(inst_keys), inst_prims, grid_w, n_visible, n_visible_ptr, capacity, counters, big_list, big_count);
        hipLaunchKernelGGL(create_instances_big_kernel<uint16_t>, big_grid, block, 0, s, sorted_prims, offsets, rec, big_list, big_count,
                           static_cast<uint16_t*>(inst_keys), inst_prims, grid_w, capacity);
    } else {
        hipLaunchKernelGGL(create_instances_kernel<uint32_t>, grid, block, 0, s, sorted_prims, offsets, n_touched, rec,
                           static_cast<uint32_t*>(inst_keys), inst_prims, grid_w, n_visible, n_visible_ptr, capacity, counters, big_list, big_count);
        hipLaunchKernelGGL(create_instances_big_kernel<uint32_t>, big_grid, block, 0, s, sorted_prims, offsets, rec, big_list, big_count,
                           static_cast<uint32_t*>(inst_keys), inst_prims, grid_w, capacity);
    }
    return hipGetLastError();
}

// ---- K6 ----------------------------------------------------------------------------------------------------
template <typename KeyT>
static size_t tile_sort_temp_bytes_t(uint32_t n, int end_bit) {
    size_t bytes = 0;
    rocprim::double_buffer<KeyT> k(nullptr, nullptr);
    rocprim::double_buffer<uint32_t> v(nullptr, nullptr);
    (void)rocprim::radix_sort_pairs(nullptr, bytes, k, v, n, 0u, static_cast<unsigned>(end_bit));
    const size_t own = own_sort_temp_bytes(n, end_bit);
    return own > bytes ? own : bytes;
}
size_t tile_sort_temp_bytes(uint32_t n_instances, int key_bytes, int end_bit) {
    return key_bytes == 2 ? tile_sort_temp_bytes_t<uint16_t>(n_instances, end_bit) : tile_sort_temp_bytes_t<uint32_t>(n_instances, end_bit);
}

template <typename KeyT>
static hipError_t run_tile_sort_t(void* temp, size_t temp_bytes, void* keys[2], uint32_t* vals[2], int& selector, uint32_t n,
                                  int end_bit, hipStream_t s) {
    rocprim::double_buffer<KeyT> k(static_cast<KeyT*>(keys[0]), static_cast<KeyT*>(keys[1]));
    rocprim::double_buffer<uint32_t> v(vals[0], vals[1]);
    hipError_t e = rocprim::radix_sort_pairs(temp, temp_bytes, k, v, n, 0u, static_cast<unsigned>(end_bit), s);
    if (e != hipSuccess) return e;
    selector = (v.current() == vals[0]) ? 0 : 1;      // fwd:205 records which half holds the sorted list
    return hipSuccess;
}
bool tile_sort_takes_device_count() { return (g_sort_implementation & 1) != 0; }
hipError_t run_tile_sort(void* temp, size_t temp_bytes, int key_bytes, void* keys[2], uint32_t* vals[2], int& selector,
                         uint32_t n_instances, const uint32_t* n_instances_ptr, int end_bit, hipStream_t s) {
    selector = 0;
    if (n_instances == 0) return hipSuccess;
    if (g_sort_implementation & 1) {       // n_instances_ptr != nullptr: n_instances is the capacity, the count lives on the device
        if (key_bytes == 2) {
            uint16_t* k16[2] = {static_cast<uint16_t*>(keys[0]), static_cast<uint16_t*>(keys[1])};
            return own_sort_pairs_u16_device_count(temp, temp_bytes, k16, vals, selector, n_instances, n_instances_ptr, end_bit, s);
        }
        uint32_t* k32[2] = {static_cast<uint32_t*>(keys[0]), static_cast<uint32_t*>(keys[1])};
        return own_sort_pairs_u32_device_count(temp, temp_bytes, k32, vals, selector, n_instances, n_instances_ptr, end_bit, s);
    }
    if (n_instances_ptr != nullptr) return hipErrorInvalidValue;
    return key_bytes == 2 ? run_tile_sort_t<uint16_t>(temp, temp_bytes, keys, vals, selector, n_instances, end_bit, s)
                          : run_tile_sort_t<uint32_t>(temp, temp_bytes, keys, vals, selector, n_instances, end_bit, s);
}

// ---- K7 (kf:331-348); ranges are pre-zeroed by the host (fwd:54) ----------------------------------------------
template <typename KeyT>
__global__ void __launch_bounds__(256) extract_ranges_kernel(const KeyT* __restrict__ keys, uint2* __restrict__ ranges, const uint32_t n_value,
                                                             const uint32_t* __restrict__ n_ptr) {
    // One 16-byte load (8 / 4 keys) per thread plus the key in front of them: with one key per thread (round 1) the kernel took 17 us for
    // 32 MB -- 250 k waves of two 2-byte loads each; range boundaries are rare (12 k among 16 M keys).
    constexpr unsigned kPer = 16 / sizeof(KeyT);
    const uint32_t n = n_ptr != nullptr ? *n_ptr : n_value;
    const uint32_t first = (blockIdx.x * 256u + threadIdx.x) * kPer;
    if (first >= n) return;
    KeyT k[kPer];
    if (first + kPer <= n) {
        const uint4 q = reinterpret_cast<const uint4*>(keys)[first / kPer];          // the key array starts on a 256-byte boundary (carve)
        const uint32_t w[4] = {q.x, q.y, q.z, q.w};
#pragma unroll
        for (unsigned j = 0; j < kPer; ++j)
            k[j] = sizeof(KeyT) == 4 ? static_cast<KeyT>(w[j]) : static_cast<KeyT>((w[j / 2] >> (16u * (j & 1u))) & 0xffffu);
    } else {
#pragma unroll
        for (unsigned j = 0; j < kPer; ++j) k[j] = first + j < n ? keys[first + j] : static_cast<KeyT>(0);
    }
    KeyT prev = first == 0 ? k[0] : keys[first - 1];
    if (first == 0) ranges[k[0]].x = 0;
#pragma unroll
    for (unsigned j = 0; j < kPer; ++j) {
        const uint32_t i = first + j;
        if (i < n) {
            if (k[j] != prev) { ranges[prev].y = i; ranges[k[j]].x = i; }
            if (i == n - 1) ranges[k[j]].y = n;
            prev = k[j];
        }
    }
}
hipError_t launch_extract_ranges(int key_bytes, const void* sorted_keys, uint2* ranges, uint32_t n_instances, const uint32_t* n_instances_ptr, hipStream_t s) {
    if (n_instances == 0) return hipSuccess;
    const uint32_t per_block = 256u * (16u / static_cast<uint32_t>(key_bytes));          // 16 bytes of keys per thread
    const dim3 grid((n_instances + per_block - 1) / per_block), block(256);
    if (key_bytes == 2) hipLaunchKernelGGL(extract_ranges_kernel<uint16_t>, grid, block, 0, s, static_cast<const uint16_t*>(sorted_keys), ranges, n_instances, n_instances_ptr);
    else hipLaunchKernelGGL(extract_ranges_kernel<uint32_t>, grid, block, 0, s, static_cast<const uint32_t*>(sorted_keys), ranges, n_instances, n_instances_ptr);
    return hipGetLastError();
}

// ---- K8+K9 (kf:350-360 + fwd:225-231) + the tile -> workgroup plan of K10 ---------------------------------------------
// ONE single-workgroup kernel (fgs_tile_scan.h) replaces the library scan (rocPRIM look-back, 15 us for 12 k tiles):
//  (1) bucket_offsets[t] = inclusive scan of ceil(len_t / 64)                                          (kf:350-360, fwd:225-231)
//  (2) the plan that K10 reads to decide which tile a workgroup blends (blend_forward.hip: tile_of_workgroup).
// Why a plan: the hardware deals workgroups to the 8 XCDs round-robin (XCD = workgroup % 8), every XCD has its own L2, and a
// Gaussian's record is re-read by every tile it overlaps -- so an XCD should own compact pieces of the image. Round 1/2 gave every
// XCD one contiguous band of tile rows: good locality, but the bands differ in work (at S2 the top band has 30 ms of summed tile time
// against 44-47 ms for the others; on a layered scene 47 against 210-220: XCD 0 idles for two thirds of the kernel) and the heaviest
// rows came last in every band (profiles/r02_k10_timeline_before.txt). Interleaving single rows balances but gives up vertical
// locality (+10 % layered, -9..16 % S2). The plan keeps both: the image is cut into 8 x 10 rectangular blocks of tiles (15 x 9 tiles
// at 1080p: every XCD gets exactly 10 blocks, i.e. the same number of workgroups, which the round-robin deal requires); a block's
// weight is its number of 64-Gaussian buckets (+ 1 per tile) -- known here, on the device, from the scan itself: no host read; the
// blocks are sorted by weight and dealt in 10 rounds of 8, heaviest block of a round to the XCD with the least work so far; an XCD
// walks its blocks in the order received = heaviest first, so the kernel's tail consists of the lightest blocks.
__global__ void __launch_bounds__(kTileScanThreads) plan_tiles_kernel(const uint2* __restrict__ ranges, uint32_t* __restrict__ bucket_offsets,
                                                                      uint32_t* __restrict__ tile_plan, const uint32_t n_tiles,
                                                                      const uint32_t grid_w, const uint32_t grid_h, const int experiment) {
    __shared__ TileScanShared s_scan;
    __shared__ uint32_t s_weight[kPlanBlocks], s_sorted[kPlanBlocks];
    const uint32_t tid = threadIdx.x, lane = tid & 63u;
    if (tid < kPlanBlocks) s_weight[tid] = 0u;
    uint32_t base = 0;
    int parity = 0;
    for (uint32_t t0 = 0; t0 < n_tiles; t0 += kTileScanThreads * kTileScanPerThread, parity ^= 1) {   // one pass at 1080p (12 240 tiles)
        uint32_t nb[kTileScanPerThread], ex[kTileScanPerThread];
        const uint32_t first = t0 + tid * kTileScanPerThread;
        if (first + kTileScanPerThread <= n_tiles) {                                         // 128 contiguous bytes: eight 16-byte loads
            const uint4* q = reinterpret_cast<const uint4*>(ranges + first);
#pragma unroll
            for (int k = 0; k < kTileScanPerThread / 2; ++k) {
                const uint4 r = q[k];
                nb[2 * k] = (r.y - r.x + kBucket - 1) / kBucket;                              // kf:350-360
                nb[2 * k + 1] = (r.w - r.z + kBucket - 1) / kBucket;
            }
        } else {
#pragma unroll
            for (int k = 0; k < kTileScanPerThread; ++k) {
                uint2 r = make_uint2(0u, 0u);
                if (first + k < n_tiles) r = ranges[first + k];
                nb[k] = (r.y - r.x + kBucket - 1) / kBucket;
            }
        }
        const uint32_t total = tile_scan_pass(nb, ex, s_scan, base, parity);
        if (first + kTileScanPerThread <= n_tiles) {
            uint4* o = reinterpret_cast<uint4*>(bucket_offsets + first);                      // inclusive (fwd:225-231)
#pragma unroll
            for (int k = 0; k < kTileScanPerThread / 4; ++k)
                o[k] = make_uint4(ex[4 * k] + nb[4 * k], ex[4 * k + 1] + nb[4 * k + 1], ex[4 * k + 2] + nb[4 * k + 2], ex[4 * k + 3] + nb[4 * k + 3]);
        } else {
#pragma unroll
            for (int k = 0; k < kTileScanPerThread; ++k) if (first + k < n_tiles) bucket_offsets[first + k] = ex[k] + nb[k];
        }
        base += total;
    }
    if (tile_plan == nullptr) return;                                                        // the default mapping of K10 is closed-form: no plan
    __syncthreads();                                                                         // bucket_offsets visible to the workgroup
    // block weights: one (block, tile row) pair per work item -- a difference of two scan values
    const uint32_t bw = (grid_w + kPlanBlocksX - 1) / kPlanBlocksX, bh = (grid_h + kPlanBlocksY - 1) / kPlanBlocksY;
    for (uint32_t i = tid; i < kPlanBlocks * bh; i += kTileScanThreads) {
        const uint32_t b = i / bh, r = i - b * bh;
        const uint32_t bx = b % kPlanBlocksX, by = b / kPlanBlocksX;
        const uint32_t ty = by * bh + r, x0 = bx * bw, x1 = min(x0 + bw, grid_w);
        if (ty < grid_h && x0 < x1) {
            const uint32_t last = ty * grid_w + x1 - 1u, first = ty * grid_w + x0;
            const uint32_t w = bucket_offsets[last] - (first != 0u ? bucket_offsets[first - 1u] : 0u) + (x1 - x0);
            atomicAdd(&s_weight[b], w);
        }
    }
    __syncthreads();
    // sort the blocks by weight (descending, ties by index): rank by counting -- 80 broadcast reads per thread
    if (tid < kPlanBlocks) {
        const uint32_t w = s_weight[tid];
        uint32_t rank = 0;
        for (uint32_t o = 0; o < kPlanBlocks; ++o) {
            const uint32_t wo = s_weight[o];
            rank += (wo > w || (wo == w && o < tid)) ? 1u : 0u;
        }
        s_sorted[(experiment & 1) ? tid : rank] = tid;                                       // experiment bit 0: no sort (blocks in natural order)
    }
    __syncthreads();
    if (tid < kWave) {                                                                       // wave 0: the deal, lanes 0..7 = the XCDs
        uint32_t load = 0;
        for (uint32_t round = 0; round < kPlanBlocksPerXcd; ++round) {
            uint32_t rank = 0;                                                               // my position among the XCDs by work so far
#pragma unroll
            for (int x = 0; x < kXcds; ++x) {
                const uint32_t lx = wave_read(load, x);
                rank += (lx < load || (lx == load && static_cast<uint32_t>(x) < lane)) ? 1u : 0u;
            }
            if (experiment & 1) rank = lane;                                                 // ... dealt statically: XCD x owns block column x
            if (lane < kXcds) {
                const uint32_t b = s_sorted[round * kXcds + rank];                           // least work so far <- heaviest block of the round
                load += s_weight[b];
                tile_plan[kPlanHeader + lane * kPlanBlocksPerXcd + round] = b;
            }
        }
        if (lane == 0) { tile_plan[0] = bw; tile_plan[1] = bh; tile_plan[2] = bw * bh; tile_plan[3] = kPlanBlocksPerXcd; }
    }
}

std::atomic<int> g_plan_experiment{0};     // fgs_debug_set_option(12, bits): 1 = blocks unsorted and dealt statically (A/B of the deal itself)
hipError_t launch_plan_tiles(const uint2* ranges, uint32_t* bucket_offsets, uint32_t* tile_plan, uint32_t n_tiles, uint32_t grid_w, uint32_t grid_h,
                             hipStream_t s) {
    hipLaunchKernelGGL(plan_tiles_kernel, dim3(1), dim3(kTileScanThreads), 0, s, ranges, bucket_offsets, tile_plan, n_tiles, grid_w, grid_h,
                       g_plan_experiment.load());
    return hipGetLastError();
}

// the library scan (rocPRIM): kept for A/B runs (fgs_debug_set_option(11, 1))
struct BucketsOfRange {
    __host__ __device__ uint32_t operator()(const uint2& r) const { return (r.y - r.x + kBucket - 1) / kBucket; }
};
size_t bucket_scan_temp_bytes(uint32_t n_tiles) {
    size_t bytes = 0;
    auto in = rocprim::make_transform_iterator(static_cast<const uint2*>(nullptr), BucketsOfRange{});
    (void)rocprim::inclusive_scan(nullptr, bytes, in, static_cast<uint32_t*>(nullptr), n_tiles, rocprim::plus<uint32_t>());
    return bytes;
}
hipError_t run_bucket_scan(void* temp, size_t temp_bytes, const uint2* ranges, uint32_t* bucket_offsets, uint32_t n_tiles, hipStream_t s) {
    auto in = rocprim::make_transform_iterator(ranges, BucketsOfRange{});
    return rocprim::inclusive_scan(temp, temp_bytes, in, bucket_offsets, n_tiles, rocprim::plus<uint32_t>(), s);
}

}  // namespace fgs
