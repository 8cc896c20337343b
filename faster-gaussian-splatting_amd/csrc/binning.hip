// K2-K9: tile binning for gfx950.
//   depth sort (32-bit keys) -> exclusive scan of tile counts in depth order -> instance creation (exact overlap)
//   -> stable tile sort on end_bit bits -> per-tile [start,end) ranges -> inclusive scan of per-tile bucket counts.
// Semantics: reference rasterization/src/forward.cu:104-231 + kernels_forward.cuh:211-360 (two-stage "Splatshop" sort,
// no 64-bit tile|depth key). Sorts and scans use rocPRIM (the native AMD device primitives); the gather of K3
// (apply_depth_ordering_cu) and the bucket-count kernel K8 are folded into the scans as transform iterators, so two
// kernel launches and two V/T-sized round trips through HBM disappear.
// Built with -ffp-contract=off (the exact-overlap test must agree bit-for-bit with the one in preprocess.hip).
#include "fgs_kernels.h"
#include <fgs_wave.h>
#include <cstring>
#include <rocprim/device/device_radix_sort.hpp>
#include <rocprim/device/device_scan.hpp>
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
    auto in = rocprim::make_transform_iterator(static_cast<const uint32_t*>(nullptr), TouchedFromRec{nullptr});
    (void)rocprim::exclusive_scan(nullptr, scan_bytes, in, static_cast<uint32_t*>(nullptr), 0u, n, rocprim::plus<uint32_t>());
    return sort_bytes > scan_bytes ? sort_bytes : scan_bytes;
}

hipError_t run_depth_sort(void* temp, size_t temp_bytes, uint32_t* keys[2], uint32_t* vals[2], int& selector,
                          uint32_t n_visible, hipStream_t s) {
    selector = 0;
    if (n_visible == 0) return hipSuccess;
    rocprim::double_buffer<uint32_t> k(keys[0], keys[1]), v(vals[0], vals[1]);
    hipError_t e = rocprim::radix_sort_pairs(temp, temp_bytes, k, v, n_visible, 0u, 32u, s);
    if (e != hipSuccess) return e;
    selector = (v.current() == vals[0]) ? 0 : 1;
    return hipSuccess;
}

hipError_t run_offsets_scan(void* temp, size_t temp_bytes, const uint32_t* sorted_prims, const uint32_t* n_touched, uint32_t* offsets,
                            uint32_t n_visible, hipStream_t s) {
    if (n_visible == 0) return hipSuccess;
    auto in = rocprim::make_transform_iterator(sorted_prims, TouchedFromRec{n_touched});
    return rocprim::exclusive_scan(temp, temp_bytes, in, offsets, 0u, n_visible, rocprim::plus<uint32_t>(), s);
}

// ---- K5 ----------------------------------------------------------------------------------------------------
// One lane per depth-sorted visible primitive; emits (tile key, primitive) for every exactly-overlapped tile in
// row-major order over the tile bounding box (kf:225-328). Footprints larger than kSeqTiles candidates are finished by
// the whole wave, 64 candidates per step; write slots come from a 64-bit ballot prefix (mbcnt).
template <typename KeyT>
__global__ void __launch_bounds__(kInstanceBlock) create_instances_kernel(
    const uint32_t* __restrict__ sorted_prims, const uint32_t* __restrict__ offsets, const PrimRec* __restrict__ rec,
    KeyT* __restrict__ inst_keys, uint32_t* __restrict__ inst_prims, const uint32_t grid_w, const uint32_t n_visible) {
    const unsigned gid = blockIdx.x * kInstanceBlock + threadIdx.x;
    const unsigned lane = lane_id();
    const bool active = gid < n_visible;
    if (wave_ballot(active) == 0) return;
    const unsigned i = active ? gid : n_visible - 1;
    const uint32_t prim = sorted_prims[i];
    const float4* r = reinterpret_cast<const float4*>(rec + prim);
    const float4 r0 = r[0], r1 = r[1], r2 = r[2];
    const float sx = r0.x - 0.5f, sy = r0.y - 0.5f;
    const float ca = r0.z, cb = r0.w, cc = r1.x;
    const float pt = logf(r1.y * kMinAlphaThresholdRcp);                          // kf:267
    unsigned tx0, tx1, ty0, ty1;
    tile_rect(__float_as_uint(r2.y), __float_as_uint(r2.z), tx0, tx1, ty0, ty1);
    const unsigned tbw = tx1 - tx0;
    const unsigned count = tbw * (ty1 - ty0);
    unsigned w = offsets[i];

    if (active) {
        const unsigned n_seq = count < (unsigned)kSeqTiles ? count : (unsigned)kSeqTiles;
        for (unsigned t = 0; t < n_seq; ++t) {
            const unsigned tx = tx0 + t % tbw, ty = ty0 + t / tbw;
            if (tile_contributes(sx, sy, ca, cb, cc, tx, ty, pt)) {
                inst_keys[w] = static_cast<KeyT>(ty * grid_w + tx);
                inst_prims[w] = prim;
                ++w;
            }
        }
    }
    uint64_t pending = wave_ballot(active && count > (unsigned)kSeqTiles);
    while (pending != 0) {
        const int src = __ffsll(static_cast<unsigned long long>(pending)) - 1;
        pending &= pending - 1;
        const unsigned o_tx0 = wave_read(tx0, src), o_ty0 = wave_read(ty0, src);
        const unsigned o_tbw = wave_read(tbw, src), o_cnt = wave_read(count, src);
        const float o_sx = wave_read(sx, src), o_sy = wave_read(sy, src);
        const float o_ca = wave_read(ca, src), o_cb = wave_read(cb, src), o_cc = wave_read(cc, src);
        const float o_pt = wave_read(pt, src);
        const unsigned o_prim = wave_read(prim, src);
        unsigned o_w = wave_read(w, src);
        for (unsigned base = kSeqTiles; base < o_cnt; base += kWave) {
            const unsigned t = base + lane;
            const unsigned tx = o_tx0 + t % o_tbw, ty = o_ty0 + t / o_tbw;
            const bool hit = t < o_cnt && tile_contributes(o_sx, o_sy, o_ca, o_cb, o_cc, tx, ty, o_pt);
            const uint64_t hits = wave_ballot(hit);
            if (hit) {
                const unsigned slot = o_w + lanes_below(hits);
                inst_keys[slot] = static_cast<KeyT>(ty * grid_w + tx);
                inst_prims[slot] = o_prim;
            }
            o_w += static_cast<unsigned>(__popcll(static_cast<unsigned long long>(hits)));
        }
    }
}

hipError_t launch_create_instances(int key_bytes, const uint32_t* sorted_prims, const uint32_t* offsets, const PrimRec* rec,
                                   void* inst_keys, uint32_t* inst_prims, uint32_t grid_w, uint32_t n_visible, hipStream_t s) {
    if (n_visible == 0) return hipSuccess;
    const dim3 grid((n_visible + kInstanceBlock - 1) / kInstanceBlock), block(kInstanceBlock);
    if (key_bytes == 2)
        hipLaunchKernelGGL(create_instances_kernel<uint16_t>, grid, block, 0, s, sorted_prims, offsets, rec,
                           static_cast<uint16_t*>(inst_keys), inst_prims, grid_w, n_visible);
    else
        hipLaunchKernelGGL(create_instances_kernel<uint32_t>, grid, block, 0, s, sorted_prims, offsets, rec,
                           static_cast<uint32_t*>(inst_keys), inst_prims, grid_w, n_visible);
    return hipGetLastError();
}

// ---- K6 ----------------------------------------------------------------------------------------------------
template <typename KeyT>
static size_t tile_sort_temp_bytes_t(uint32_t n, int end_bit) {
    size_t bytes = 0;
    rocprim::double_buffer<KeyT> k(nullptr, nullptr);
    rocprim::double_buffer<uint32_t> v(nullptr, nullptr);
    (void)rocprim::radix_sort_pairs(nullptr, bytes, k, v, n, 0u, static_cast<unsigned>(end_bit));
    return bytes;
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
hipError_t run_tile_sort(void* temp, size_t temp_bytes, int key_bytes, void* keys[2], uint32_t* vals[2], int& selector,
                         uint32_t n_instances, int end_bit, hipStream_t s) {
    selector = 0;
    if (n_instances == 0) return hipSuccess;
    return key_bytes == 2 ? run_tile_sort_t<uint16_t>(temp, temp_bytes, keys, vals, selector, n_instances, end_bit, s)
                          : run_tile_sort_t<uint32_t>(temp, temp_bytes, keys, vals, selector, n_instances, end_bit, s);
}

// ---- K7 (kf:331-348); ranges are pre-zeroed by the host (fwd:54) ----------------------------------------------
template <typename KeyT>
__global__ void __launch_bounds__(256) extract_ranges_kernel(const KeyT* __restrict__ keys, uint2* __restrict__ ranges, const uint32_t n) {
    const unsigned i = blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const KeyT t = keys[i];
    if (i == 0) ranges[t].x = 0;
    else {
        const KeyT prev = keys[i - 1];
        if (t != prev) { ranges[prev].y = i; ranges[t].x = i; }
    }
    if (i == n - 1) ranges[t].y = n;
}
hipError_t launch_extract_ranges(int key_bytes, const void* sorted_keys, uint2* ranges, uint32_t n_instances, hipStream_t s) {
    if (n_instances == 0) return hipSuccess;
    const dim3 grid((n_instances + 255) / 256), block(256);
    if (key_bytes == 2) hipLaunchKernelGGL(extract_ranges_kernel<uint16_t>, grid, block, 0, s, static_cast<const uint16_t*>(sorted_keys), ranges, n_instances);
    else hipLaunchKernelGGL(extract_ranges_kernel<uint32_t>, grid, block, 0, s, static_cast<const uint32_t*>(sorted_keys), ranges, n_instances);
    return hipGetLastError();
}

// ---- K8+K9 (kf:350-360 + fwd:225-231) as one scan with a transform iterator -----------------------------------
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
