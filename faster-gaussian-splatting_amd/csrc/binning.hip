// K2-K9: tile binning for gfx950.
//   depth sort (32-bit keys) -> exclusive scan of tile counts in depth order -> instance creation (exact overlap)
//   -> stable tile sort on end_bit bits -> per-tile [start,end) ranges -> inclusive scan of per-tile bucket counts.
// Semantics: reference rasterization/src/forward.cu:104-231 + kernels_forward.cuh:211-360 (two-stage "Splatshop" sort,
// no 64-bit tile|depth key). Both sorts are radix_sort.hip; the offsets are a reduction + per-wave sums (below), the bucket scan a single-workgroup kernel.
// Round 5: the depth sort's last scatter pass carries a 16-byte FOOTPRINT ROW per visible Gaussian (tile box + exact-overlap
// bitmap, fgs_math.h) into depth order and writes its tile count beside it, so the scan (K3 + K4: apply_depth_ordering_cu + ExclusiveSum)
// and the instance kernel (K5) read streams: until round 4 both gathered a 128-byte line per Gaussian for 4 / 16 useful bytes
// (rocprofv3: K5 fetched 3.2x its algorithmic bytes).
// Built with -ffp-contract=off (the exact-overlap test must agree bit-for-bit with the one in preprocess.hip).
#include "fgs_kernels.h"
#include <fgs_wave.h>
#include "fgs_tile_scan.h"
#include <cstring>
#ifdef FGS_DEV_SWITCHES
#include <rocprim/device/device_scan.hpp>            // the library scan of the per-tile bucket counts: an A/B option of the dev build only
#include <rocprim/iterator/transform_iterator.hpp>
#endif

namespace fgs {

// ---- K2-K4 -------------------------------------------------------------------------------------------------
size_t depth_sort_temp_bytes(uint32_t n) { return own_sort_temp_bytes(n, 32); }

hipError_t run_depth_sort(void* temp, size_t temp_bytes, uint32_t* keys[2], uint32_t* vals[2], int& selector, uint32_t n_visible,
                          const uint32_t* n_visible_ptr, DepthKeyRange range, uint4* foot[2], uint32_t* tile_counts,
                          uint32_t* big_list, uint32_t* big_count, hipStream_t s) {
    selector = 0;
    if (n_visible == 0) return hipSuccess;
    const SortPayload payload{foot[0], foot[1], tile_counts, 0, big_list, big_count};
    return own_depth_sort(temp, temp_bytes, keys, vals, selector, n_visible, n_visible_ptr, range, s, &payload);
}

// K3 + K4 (apply_depth_ordering_cu + ExclusiveSum, kf:211-221, fwd:104-127) without a device-wide scan. The offsets K5 needs are the exclusive
// prefix sums of the tile counts in depth order. Until round 4 this was rocPRIM's look-back scan: two launches, 0.020 ms for an 8 MB stream (0.032 with
// the gather it had then). A wave of K5 only needs the offset of ITS first Gaussian -- inside the wave a DPP scan does the rest -- so ONE small kernel
// reduces the counts to a sum per 64-Gaussian wave segment and per 4096-Gaussian block, and a K5 wave adds up the block sums in front of its block
// (500 words at S2: eight L2-resident loads per lane) and the <= 63 segment sums in front of it inside its block. No prefix over the block sums is
// formed: a "last workgroup" doing it behind a ticket needs a device-scope release fence per workgroup, and on this chip that fence writes back the
// XCD's L2 -- measured 0.016 ms for the kernel against 0.020 for the library scan it replaced (profiles/r05_ab_scan_sums.txt). K5 writes the
// per-Gaussian offsets on its way (bu:60; the parity tests compare them with the reference's scan).
constexpr int kSumBlock = 4096, kSumThreads = 256, kSumPerThread = kSumBlock / kSumThreads;      // 16 consecutive counts per thread = a quarter wave segment
__global__ void __launch_bounds__(kSumThreads) tile_count_sums_kernel(const uint32_t* __restrict__ tile_counts, const uint32_t n_value,
                                                                      const uint32_t* __restrict__ n_ptr, uint32_t* __restrict__ wave_sums,
                                                                      uint32_t* __restrict__ block_sums) {
    __shared__ uint32_t s_part[kSumThreads / kWave];
    const uint32_t n = n_ptr != nullptr ? *n_ptr : n_value;
    if (blockIdx.x * static_cast<uint32_t>(kSumBlock) >= n) return;         // workgroup-uniform (grid sized by a bound in the synchronisation-free forward)
    const uint32_t first = blockIdx.x * kSumBlock + threadIdx.x * kSumPerThread;
    uint32_t sum = 0;
    if (first + kSumPerThread <= n) {
        const uint4* q = reinterpret_cast<const uint4*>(tile_counts + first);
#pragma unroll
        for (int k = 0; k < kSumPerThread / 4; ++k) { const uint4 v = q[k]; sum += v.x + v.y + v.z + v.w; }
    } else {
#pragma unroll
        for (int k = 0; k < kSumPerThread; ++k) sum += first + k < n ? tile_counts[first + k] : 0u;
    }
    // a 64-Gaussian wave segment of K5 = 4 consecutive threads here
    const uint32_t incl = wave_inclusive_sum(sum);
    const unsigned lane = lane_id();
    const uint32_t before = wave_shuffle(incl, (lane & ~3u) - 1u);                                   // inclusive sum of the lane in front of this group of four (all lanes shuffle)
    if ((lane & 3u) == 3u) wave_sums[(first - 3u * kSumPerThread) / kWave] = incl - (lane >= 4u ? before : 0u);
    if (lane == kWave - 1) s_part[threadIdx.x >> 6] = incl;
    __syncthreads();
    if (threadIdx.x == 0) block_sums[blockIdx.x] = s_part[0] + s_part[1] + s_part[2] + s_part[3];
}

hipError_t launch_tile_count_sums(const uint32_t* tile_counts, uint32_t* wave_sums, uint32_t* block_sums,
                                  uint32_t n_visible, const uint32_t* n_visible_ptr, hipStream_t s) {
    if (n_visible == 0) return hipSuccess;
    hipLaunchKernelGGL(tile_count_sums_kernel, dim3((n_visible + kSumBlock - 1) / kSumBlock), dim3(kSumThreads), 0, s, tile_counts, n_visible, n_visible_ptr,
                       wave_sums, block_sums);
    return hipGetLastError();
}

// ---- K5, big footprints ------------------------------------------------------------------------------------------
// Big footprints (more than kBigInstanceFootprint candidate tiles: 0.07 % of the visible Gaussians at S2, 3 % of the instances) get a workgroup
// each: kInstanceBlock candidate tiles per step, exact test (kf:283-326), write slots from a ballot prefix inside each wave and an LDS prefix
// across the waves -- consecutive slots, stable row-major order. Depth order puts the nearest (largest) Gaussians next to each other, so
// finishing them in the main walk would serialise tens of thousands of candidate tiles in a handful of waves (+0.2 ms on views with
// screen-filling Gaussians). Until round 4 they had a launch of their own behind the main kernel (10-20 us of a 0.70 ms frame, set by the step
// chain of the largest item); now the depth sort's last pass lists them and the first kBigBlocks workgroups of the SAME launch work the list off,
// each item finding its own output offset from the wave-segment / block sums.
constexpr unsigned kBigBlocks = 512;      // leading workgroups of the launch: 128 / 256 / 512 / 1024 measured 0.058 / 0.045 / 0.043 / 0.044 ms (profiles/r05_ab_k5_block.txt)
template <typename KeyT>
__device__ __forceinline__ void expand_big_footprints(const uint4* __restrict__ foot, const uint32_t* __restrict__ wave_sums,
                                                      const uint32_t* __restrict__ block_sums, const PrimRec* __restrict__ rec,
                                                      KeyT* __restrict__ inst_keys, uint32_t* __restrict__ inst_prims, const uint32_t grid_w,
                                                      const uint32_t capacity, const uint32_t* __restrict__ big_list, const unsigned n_big) {
    constexpr int kWaves = kInstanceBlock / kWave;
    __shared__ unsigned s_hits[2][kWaves];
    __shared__ unsigned s_off[kWaves];
    const unsigned lane = lane_id(), wv = threadIdx.x >> 6;
    for (unsigned b = blockIdx.x; b < n_big; b += kBigBlocks) {             // workgroup-uniform loop
        const uint32_t i = big_list[b];
        // first output slot of entry i (as in the main walk): blocks in front of its block, wave segments in front of its segment inside the
        // block, entries in front of it inside its segment
        const unsigned segment = i >> 6, block_of = segment >> 6, in_block = segment & 63u, in_segment = i & 63u;
        uint32_t part = 0;
        for (unsigned j = threadIdx.x; j < block_of; j += kInstanceBlock) part += block_sums[j];
        if (threadIdx.x < in_block) part += wave_sums[(segment & ~63u) + threadIdx.x];
        if (threadIdx.x >= 64u && threadIdx.x - 64u < in_segment) part += footprint_tile_count(foot[(segment << 6) + threadIdx.x - 64u]);
        const uint32_t wsum = wave_sum(part);
        if (lane == 0) s_off[wv] = wsum;
        const uint4 row = foot[i];
        const uint32_t prim = row.x;
        const float4* r = reinterpret_cast<const float4*>(rec + prim);
        const float4 r0 = r[0], r1 = r[1], r2 = r[2];
        const TileTest tt = make_tile_test(r0.x - 0.5f, r0.y - 0.5f, r0.z, r0.w, r1.x, logf(r1.y * kMinAlphaThresholdRcp));   // kf:267
        unsigned tx0, tx1, ty0, ty1;
        tile_rect(__float_as_uint(r2.y), __float_as_uint(r2.z), tx0, tx1, ty0, ty1);
        const unsigned tbw = tx1 - tx0;
        const unsigned count = tbw * (ty1 - ty0);
        __syncthreads();
        unsigned w = 0;
#pragma unroll
        for (int k = 0; k < kWaves; ++k) w += s_off[k];
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
        __syncthreads();                                                           // s_hits / s_off are reused by the next footprint
    }
}

// ---- K5 ----------------------------------------------------------------------------------------------------
// Emits (tile key, primitive) for every exactly-overlapped tile of every depth-sorted visible primitive, in row-major
// order over its tile bounding box (kf:225-328). CDNA4 shape: a wave owns 64 consecutive primitives, whose outputs form
// ONE contiguous range [offset(first), offset(last)+n), and reads their footprint rows (fgs_math.h) as a stream. Boxes of
// <= 64 candidate tiles (99 % of the visible Gaussians and 85 % of the instances at S2) carry their exact-overlap bitmap from
// preprocess, so nothing is re-tested and no record is touched. Round 5: the wave walks the CANDIDATE tiles of its bitmap
// footprints laid end to end, 64 per step (rounds 1-4 walked the output slots: a 6-step owner search in LDS plus a 5-step
// "r-th set bit" per slot, ~80 vector instructions per step). Every footprint sets one head bit at its first candidate
// slot; a lane finds its owner as (heads before the step) + (head bits at or below its lane) -- one broadcast LDS read
// and an mbcnt --, tests its candidate's bit of the owner's bitmap and writes to (owner's first output slot) + (set bits
// below): consecutive lanes store ascending addresses with the unset candidates left out. Escape rows (larger boxes) are
// re-tested from the record by the whole wave, 64 candidate tiles per step, with ballot-prefix write slots (also consecutive).
template <typename KeyT>
__global__ void __launch_bounds__(kInstanceBlock) create_instances_kernel(
    const uint4* __restrict__ foot, const uint32_t* __restrict__ wave_sums, const uint32_t* __restrict__ block_sums, uint32_t* __restrict__ offsets,
    const PrimRec* __restrict__ rec, KeyT* __restrict__ inst_keys, uint32_t* __restrict__ inst_prims, const uint32_t grid_w,
    const uint32_t n_visible_value, const uint32_t* __restrict__ n_visible_ptr, const uint32_t capacity, uint32_t* __restrict__ counters,
    const uint32_t* __restrict__ big_list, const uint32_t* __restrict__ big_count) {
    // The first kBigBlocks workgroups of the grid expand the big footprints (listed by the depth sort's last pass) while the others walk the
    // visible list: the long poles start first and run beside the main walk instead of in a launch of their own behind it.
    if (blockIdx.x < kBigBlocks) {
        expand_big_footprints<KeyT>(foot, wave_sums, block_sums, rec, inst_keys, inst_prims, grid_w, capacity, big_list, *big_count);
        return;
    }
    const unsigned main_block = blockIdx.x - kBigBlocks;
    // The visible count by value, or (host-synchronisation-free forward) through n_visible_ptr with the grid sized by a bound. `capacity` =
    // size of the instance arrays: exact in the first case; in the second a caller-side estimate -- stores beyond it are dropped, and
    // the clamped instance count / an overflow flag are left in counters[5] / counters[6] for the sort and for the caller to check.
    const uint32_t n_visible = n_visible_ptr != nullptr ? *n_visible_ptr : n_visible_value;
    if (n_visible_ptr != nullptr && main_block == 0 && threadIdx.x == 0) {
        const uint32_t n_instances = counters[1];
        counters[5] = n_instances < capacity ? n_instances : capacity;
        counters[6] = n_instances > capacity ? 1u : 0u;
    }
    constexpr int kWaves = kInstanceBlock / kWave;
    // per bitmap footprint of the wave, compacted (index = its rank among the wave's bitmap footprints):
    __shared__ uint4 s_pack[kWaves][kWave];           // first candidate slot, overlap bitmap (2 words), first output slot
    __shared__ uint4 s_geo[kWaves][kWave];            // primitive, box origin tx0 | ty0 << 16, box width, ceil(2^16 / width)
    __shared__ uint32_t s_head[kWaves][2 * kWave];    // the wave's 64 x 64 candidate slots: bit = a footprint starts here

    const unsigned gid = main_block * kInstanceBlock + threadIdx.x;
    const unsigned lane = lane_id(), wv = threadIdx.x >> 6;
    const bool active = gid < n_visible;
    if (wave_ballot(active) == 0) return;              // wave-uniform; waves are independent (no workgroup barrier)
    const unsigned i = active ? gid : n_visible - 1;
    const uint4 row = foot[i];
    const uint32_t prim = row.x;
    // this Gaussian's first output slot (tile_count_sums_kernel above): the instances of the 4096-Gaussian blocks in front of the wave's block, of
    // the wave segments in front of it inside the block, and of the lanes in front of it inside the wave
    const unsigned segment = gid >> 6, block_of_wave = segment >> 6, in_block = segment & 63u;
    uint32_t before_wave = lane < in_block ? wave_sums[(segment & ~63u) + lane] : 0u;
    for (unsigned j = lane; j < block_of_wave; j += kWave) before_wave += block_sums[j];
    const uint32_t n_tiles_mine = active ? footprint_tile_count(row) : 0u;
    const uint32_t my_off = wave_sum(before_wave) + wave_exclusive_sum(n_tiles_mine);
    if (active) offsets[i] = my_off;                   // bu:60 -- kept as an output of the stage: the parity tests compare it with the reference's scan
    const bool small = active && row.y != kFootprintEscape;

    // ---- bitmap footprints: their candidates end to end, 64 per step ----
    const unsigned tbw_small = ((row.y >> 20) & 63u) + 1u;
    const uint32_t n_cand = small ? tbw_small * (((row.y >> 26) & 63u) + 1u) : 0u;
    const uint32_t cand_end = wave_inclusive_sum(n_cand);
    const uint32_t cand_start = cand_end - n_cand;
    const uint32_t total_cand = wave_read(cand_end, kWave - 1);
    const unsigned slot = lanes_below(wave_ballot(small));
    s_head[wv][lane] = 0u; s_head[wv][kWave + lane] = 0u;
    wave_lds_fence();
    if (small) {
        s_pack[wv][slot] = make_uint4(cand_start, row.z, row.w, my_off);
        s_geo[wv][slot] = make_uint4(prim, (row.y & 1023u) | (((row.y >> 10) & 1023u) << 16), tbw_small, (65536u + tbw_small - 1u) / tbw_small);
        atomicOr(&s_head[wv][cand_start >> 5], 1u << (cand_start & 31u));                                  // distinct bits: every footprint has >= 1 candidate
    }
    wave_lds_fence();
    unsigned heads_before = 0;                         // wave-uniform
    for (uint32_t base = 0; base < total_cand; base += kWave) {
        const uint64_t heads = wave_uniform(static_cast<uint64_t>(s_head[wv][base >> 5]) | (static_cast<uint64_t>(s_head[wv][(base >> 5) + 1u]) << 32));
        const uint32_t p = base + lane;
        // owner = the last footprint starting at or before this candidate slot; slot 0 of the wave is a head, so the index is never negative
        const unsigned owner = heads_before + lanes_below(heads) + static_cast<unsigned>((heads >> lane) & 1ull) - 1u;
        heads_before += static_cast<unsigned>(__popcll(static_cast<unsigned long long>(heads)));
        const uint4 pk = s_pack[wv][owner];
        const unsigned t = (p - pk.x) & 63u;           // candidate index inside the owner's box (< 64 for every lane below total_cand)
        const uint64_t bitmap = static_cast<uint64_t>(pk.y) | (static_cast<uint64_t>(pk.z) << 32);
        if (p < total_cand && ((bitmap >> t) & 1ull) != 0ull) {
            const uint4 geo = s_geo[wv][owner];
            const unsigned row_t = (t * geo.w) >> 16, col_t = t - row_t * geo.z;                         // t / width, t % width (t < 64, width <= 64)
            const uint32_t o = pk.w + static_cast<unsigned>(__popcll(static_cast<unsigned long long>(bitmap & ((1ull << t) - 1ull))));
            if (o < capacity) {
                inst_keys[o] = static_cast<KeyT>(((geo.y >> 16) + row_t) * grid_w + (geo.y & 0xffffu) + col_t);
                inst_prims[o] = geo.x;
            }
        }
    }

    // ---- escape rows (boxes of 65 .. kBigInstanceFootprint candidate tiles; larger ones belong to the leading workgroups): re-tested by this wave
    // from the record, 64 candidates per step, with ballot-prefix write slots (kf:283-326) ----
    const bool recompute = active && !small && row.w <= kBigInstanceFootprint;
    unsigned tx0 = 0, ty0 = 0, tbw = 1, count = 0;
    float4 r0{}, r1{};
    if (recompute) {                                   // the only lanes that touch the record
        const float4* rr = reinterpret_cast<const float4*>(rec + prim);
        r0 = rr[0]; r1 = rr[1];
        const float4 r2 = rr[2];                       // colour.b, bounds x, bounds y, hot-accumulator slot
        unsigned tx1, ty1;
        tile_rect(__float_as_uint(r2.y), __float_as_uint(r2.z), tx0, tx1, ty0, ty1);
        tbw = tx1 - tx0;
        count = tbw * (ty1 - ty0);
    }
    uint64_t pending = wave_ballot(recompute);
    if (pending != 0) {
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

}

hipError_t launch_create_instances(int key_bytes, const uint4* foot_sorted, const uint32_t* wave_sums,
                                   const uint32_t* block_sums, uint32_t* offsets,
                                   const PrimRec* rec, void* inst_keys, uint32_t* inst_prims, uint32_t grid_w, uint32_t n_visible,
                                   const uint32_t* n_visible_ptr, uint32_t capacity, uint32_t* counters,
                                   const uint32_t* big_list, const uint32_t* big_count, hipStream_t s) {
    if (n_visible == 0) return hipSuccess;
    const dim3 grid(kBigBlocks + (n_visible + kInstanceBlock - 1) / kInstanceBlock), block(kInstanceBlock);
    if (key_bytes == 2)
        hipLaunchKernelGGL(create_instances_kernel<uint16_t>, grid, block, 0, s, foot_sorted, wave_sums, block_sums, offsets, rec,
                           static_cast<uint16_t*>(inst_keys), inst_prims, grid_w, n_visible, n_visible_ptr, capacity, counters, big_list, big_count);
    else
        hipLaunchKernelGGL(create_instances_kernel<uint32_t>, grid, block, 0, s, foot_sorted, wave_sums, block_sums, offsets, rec,
                           static_cast<uint32_t*>(inst_keys), inst_prims, grid_w, n_visible, n_visible_ptr, capacity, counters, big_list, big_count);
    return hipGetLastError();
}

// ---- K6 ----------------------------------------------------------------------------------------------------
size_t tile_sort_temp_bytes(uint32_t n_instances, int /*key_bytes*/, int end_bit) { return own_sort_temp_bytes(n_instances, end_bit); }

// n_instances_ptr != nullptr: n_instances is the capacity of the arrays, the count lives on the device. selector: which half of the
// double buffers holds the sorted list (fwd:205).
hipError_t run_tile_sort(void* temp, size_t temp_bytes, int key_bytes, void* keys[2], uint32_t* vals[2], int& selector,
                         uint32_t n_instances, const uint32_t* n_instances_ptr, int end_bit, hipStream_t s) {
    selector = 0;
    if (n_instances == 0) return hipSuccess;
    if (key_bytes == 2) {
        uint16_t* k16[2] = {static_cast<uint16_t*>(keys[0]), static_cast<uint16_t*>(keys[1])};
        return own_sort_pairs_u16_device_count(temp, temp_bytes, k16, vals, selector, n_instances, n_instances_ptr, end_bit, s);
    }
    uint32_t* k32[2] = {static_cast<uint32_t*>(keys[0]), static_cast<uint32_t*>(keys[1])};
    return own_sort_pairs_u32_device_count(temp, temp_bytes, k32, vals, selector, n_instances, n_instances_ptr, end_bit, s);
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
// rows came last in every band (profiles/archive/r02_k10_timeline_before.txt). Interleaving single rows balances but gives up vertical
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

hipError_t launch_plan_tiles(const uint2* ranges, uint32_t* bucket_offsets, uint32_t* tile_plan, uint32_t n_tiles, uint32_t grid_w, uint32_t grid_h,
                             hipStream_t s) {
    hipLaunchKernelGGL(plan_tiles_kernel, dim3(1), dim3(kTileScanThreads), 0, s, ranges, bucket_offsets, tile_plan, n_tiles, grid_w, grid_h,
                       static_cast<int>(g_plan_experiment));
    return hipGetLastError();
}

#ifdef FGS_DEV_SWITCHES
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

#endif  // FGS_DEV_SWITCHES

}  // namespace fgs
