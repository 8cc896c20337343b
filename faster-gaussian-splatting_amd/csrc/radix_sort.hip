// Stable LSD radix sort of (key, uint32 value) pairs for the two sorts of the binning stage (K2: 32-bit depth keys of the
// visible Gaussians; K6: tile keys of the instances), written for their sizes on gfx950.
//
// Why not rocPRIM: its onesweep sort is tuned for large inputs -- 16 384 items per 1024-thread workgroup (80 B of scratch per
// lane), one histogram fill plus two fills per 8-bit pass. The depth sort has ~2 M items: ~125 workgroups for 256 CUs behind a
// decoupled look-back chain, 14 launches, 0.167 ms for 32 MB of traffic; the tile sort moves 16 M items at 1.7 TB/s (0.266 ms).
// Here a pass is three launches over 8192-item workgroups (4096 until round 4) -- per-workgroup digit histogram -> one workgroup per digit scans its
// row of the [digit][workgroup] table -> stable scatter -- with ceil(end_bit / 8) passes of evenly split digit widths
// (depth keys 4 x 8 bits, tile keys at 1080p 2 x 7 bits); 8 / 16 / 24 items per thread measured 0.193 / 0.181 / 0.188 ms for the tile
// sort. Measured on MI355X (tools/ab_sort.py, S2): depth sort 0.135 ms,
// tile sort 0.173 ms with rocPRIM's scan between the kernels; wider (11-bit) digits were slower: without the LDS reorder below,
// 2 M x 2 scattered 4-byte stores per pass cost 55-72 us.
//
// Stability (equal keys keep their input order -- the tile sort relies on it to keep each tile's list in depth order,
// fwd:195-202) comes from ranking inside a workgroup in input order: a wave walks its 1024-item segment 64 consecutive items
// at a time; lanes holding the same digit find each other with one ballot per digit bit (the wave64 form of match_any),
// take the digit's running count from the wave's private LDS counters plus their position among the matching lanes, and the
// lowest matching lane advances the counter. Counts of the four waves are prefix-summed per digit afterwards, the items are
// reordered through LDS by digit, and leave so that consecutive lanes store to consecutive addresses of a digit's run.
#include <cstring>

#include "fgs_kernels.h"
#include <fgs_wave.h>

namespace fgs {

namespace sortimpl {

constexpr int kSortThreads = 256, kSortWaves = kSortThreads / kWave;
constexpr int kMaxBits = 9, kMaxBins = 1 << kMaxBits;                                  // up to two digits per thread in the block-wide scans
constexpr int kScanPerThread = 16;                                                     // row scan: table entries per thread and round
// Items per thread (IPT) and threads per workgroup (TH) are template parameters: 16 x 512 = 8192-item workgroups for both sorts since round 5 (16 x 256 before; what follows was measured then). The 16 M-item tile sort is throughput-bound
// (8 / 16 / 24 measured 0.193 / 0.181 / 0.188 ms). The 2 M-item depth sort runs < 2 workgroups per CU and looked latency-bound by a
// workgroup's chain (load -> IPT ranking rounds -> reorder -> store), but halving the chain (IPT 8) measured 10 % SLOWER (0.119 vs 0.108 ms):
// twice the workgroups pay their fixed costs twice and the table doubles. The instantiation stays as an A/B switch (g_depth_sort_mode bit 1).
template <int IPT, int TH> struct SortShape {
    static constexpr int kBlockItems = TH * IPT;
    static constexpr int kWaveItems = kBlockItems / (TH / kWave);                      // IPT rounds of 64 consecutive items
};

// `base` is subtracted first (0 for tile keys): depth keys are bit patterns of depths in [near, far], and key - bits(near) keeps their
// order in fewer bits (DepthKeyRange)
template <typename KeyT>
__device__ __forceinline__ uint32_t digit_of(KeyT key, uint32_t base, int shift, uint32_t mask) { return ((static_cast<uint32_t>(key) - base) >> shift) & mask; }

// exclusive prefix of one value per thread over a workgroup of WAVES waves; `total` = sum of all
template <int WAVES = kSortWaves>
__device__ __forceinline__ uint32_t block_exclusive_scan(uint32_t v, uint32_t* s_part /*[WAVES]*/, uint32_t& total) {
    const uint32_t lane = threadIdx.x & 63u, wv = threadIdx.x >> 6;
    const uint32_t excl = wave_exclusive_sum(v);
    __syncthreads();                                                 // s_part may still be read from a previous call
    if (lane == kWave - 1) s_part[wv] = excl + v;
    __syncthreads();
    uint32_t base = 0, sum = 0;
#pragma unroll
    for (uint32_t w = 0; w < static_cast<uint32_t>(WAVES); ++w) { const uint32_t p = s_part[w]; base += w < wv ? p : 0u; sum += p; }
    total = sum;
    return base + excl;
}

// per-workgroup digit histogram, written digit-major: hist[digit * n_blocks + block]
// The item count comes by value or -- when the host does not know it yet -- through `n_ptr` (grid sized by a capacity, workgroups
// beyond the count contribute zero rows and scatter nothing).
template <typename KeyT, int IPT, int TH>
__global__ void __launch_bounds__(TH) radix_histogram_kernel(const KeyT* __restrict__ keys, const uint32_t n_value, const uint32_t* __restrict__ n_ptr,
                                                                       const uint32_t key_base, const int shift, const int bits, uint32_t* __restrict__ hist,
                                                                       const uint32_t n_blocks) {
    constexpr int kBlockItems = SortShape<IPT, TH>::kBlockItems;
    __shared__ uint32_t s_hist[kMaxBins];
    const uint32_t n = n_ptr != nullptr ? *n_ptr : n_value;
    const uint32_t bins = 1u << bits, mask = bins - 1u;
    for (uint32_t d = threadIdx.x; d < bins; d += TH) s_hist[d] = 0u;
    __syncthreads();
    const uint32_t base = blockIdx.x * kBlockItems;
    constexpr int kPerLoad = 16 / sizeof(KeyT);                                        // keys per 16-byte load
    static_assert(IPT % kPerLoad == 0, "whole 16-byte loads per thread");
    if (base + kBlockItems <= n) {                                                     // full workgroup: 16-byte loads (order is irrelevant here)
#pragma unroll
        for (int i = 0; i < IPT / kPerLoad; ++i) {
            const uint4 q = reinterpret_cast<const uint4*>(keys + base)[i * TH + threadIdx.x];
            const uint32_t w[4] = {q.x, q.y, q.z, q.w};
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                if (sizeof(KeyT) == 4) atomicAdd(&s_hist[((w[j] - key_base) >> shift) & mask], 1u);
                else { atomicAdd(&s_hist[(((w[j] & 0xffffu) - key_base) >> shift) & mask], 1u); atomicAdd(&s_hist[(((w[j] >> 16) - key_base) >> shift) & mask], 1u); }
            }
        }
    } else {
#pragma unroll
        for (int i = 0; i < IPT; ++i) {
            const uint32_t idx = base + i * TH + threadIdx.x;
            if (idx < n) atomicAdd(&s_hist[digit_of(keys[idx], key_base, shift, mask)], 1u);
        }
    }
    __syncthreads();
    for (uint32_t d = threadIdx.x; d < bins; d += TH) hist[(size_t)d * n_blocks + blockIdx.x] = s_hist[d];
}

// One workgroup per digit: exclusive scan of that digit's row of the table (over the workgroups of the sort), in place, and the
// row total. The scatter kernel adds the exclusive scan of the digit totals itself.
__global__ void __launch_bounds__(kSortThreads) radix_row_scan_kernel(uint32_t* __restrict__ table, uint32_t* __restrict__ totals, const uint32_t n_blocks) {
    __shared__ uint32_t s_part[kSortWaves];
    uint32_t* row = table + (size_t)blockIdx.x * n_blocks;
    uint32_t carry = 0;
    for (uint32_t c0 = 0; c0 < n_blocks; c0 += kSortThreads * kScanPerThread) {        // workgroup-uniform trip count
        uint32_t v[kScanPerThread], sum = 0;
        const uint32_t first = c0 + threadIdx.x * kScanPerThread;
#pragma unroll
        for (int i = 0; i < kScanPerThread; ++i) { v[i] = first + i < n_blocks ? row[first + i] : 0u; sum += v[i]; }
        uint32_t chunk_total;
        uint32_t run = carry + block_exclusive_scan(sum, s_part, chunk_total);
#pragma unroll
        for (int i = 0; i < kScanPerThread; ++i) { if (first + i < n_blocks) row[first + i] = run; run += v[i]; }
        carry += chunk_total;
    }
    if (threadIdx.x == 0) totals[blockIdx.x] = carry;
}

// table[d * n_blocks + blk] (after the row scan) + exclusive scan of totals[] over d = where this workgroup's first item with
// digit d goes.
// BITS (the digit width) is a template parameter so that the match loop is straight-line code: as a run-time loop it cost
// 8 VALU + 4 SALU + a branch per bit and round. A thread owns DPT = max(1, 2^BITS / 256) ADJACENT digits in the per-digit steps.
template <typename KeyT, int BITS, int IPT, int TH>
__global__ void __launch_bounds__(TH) radix_scatter_kernel(const KeyT* __restrict__ keys_in, const uint32_t* __restrict__ vals_in,
                                                                     KeyT* __restrict__ keys_out, uint32_t* __restrict__ vals_out,
                                                                     const uint32_t n_value, const uint32_t* __restrict__ n_ptr, const uint32_t key_base,
                                                                     const int shift, const uint32_t* __restrict__ table,
                                                                     const uint32_t* __restrict__ totals, const uint32_t n_blocks, const SortPayload pl) {
    constexpr int kBlockItems = SortShape<IPT, TH>::kBlockItems, kWaveItems = SortShape<IPT, TH>::kWaveItems;
    constexpr int kWaves = TH / kWave;
    constexpr uint32_t kBins = 1u << BITS, mask = kBins - 1u;
    constexpr int DPT = kBins > static_cast<uint32_t>(TH) ? static_cast<int>(kBins) / TH : 1;
    const uint32_t n = n_ptr != nullptr ? *n_ptr : n_value;
    if (blockIdx.x * kBlockItems >= n) return;                      // workgroup-uniform (capacity-sized grid)
    __shared__ uint32_t s_cnt[kWaves][kBins];                 // per wave and digit: running count, later start inside the digit's run
    __shared__ uint32_t s_first[kBins];                           // first workgroup-local position of each digit
    __shared__ uint32_t s_dst[kBins];                             // global position of this workgroup's first item of each digit
    __shared__ uint32_t s_part[kWaves];
    __shared__ KeyT s_key[kBlockItems];
    __shared__ uint32_t s_val[kBlockItems];
    constexpr uint32_t kBigPerBlock = 256;              // payload pass only: big footprints found by this workgroup (see the end of the kernel)
    __shared__ uint32_t s_big[kBigPerBlock];
    __shared__ uint32_t s_n_big, s_big_base;
    // gfx950 only: the 8192-item / 512-thread shape stages ~85 KB per workgroup -- fine in CDNA4's 160 KB of LDS per CU (one workgroup per CU, which
    // the measured times accept), impossible on a 64 KB target. The Makefile's ARCH is overridable; this is where such a build has to stop.
    static_assert(sizeof(s_cnt) + sizeof(s_first) + sizeof(s_dst) + sizeof(s_part) + sizeof(s_key) + sizeof(s_val) + sizeof(s_big) + 8 <= 160 * 1024,
                  "radix_scatter_kernel: workgroup LDS exceeds the 160 KB of a gfx950 CU");
#if defined(__HIP_DEVICE_COMPILE__) && !defined(__gfx950__)
#error "radix_sort.hip sizes its workgroups for the 160 KB LDS of gfx950 (MI355X); other targets need the 4096-item / 256-thread shape"
#endif
    if (threadIdx.x == 0) s_n_big = 0u;                 // several workgroup barriers lie between this and the first append
    const uint32_t lane = lane_id(), wv = threadIdx.x >> 6;
    const uint32_t d0 = threadIdx.x * DPT;                          // this thread's first digit
    // global base of every digit: exclusive scan of the digit totals (requested now, used after the ranking)
    uint32_t digit_total[DPT], row_offset[DPT];
#pragma unroll
    for (int j = 0; j < DPT; ++j) {
        const uint32_t d = d0 + j;
        const bool own = d < kBins;
        if (own) {
#pragma unroll
            for (int w = 0; w < kWaves; ++w) s_cnt[w][d] = 0u;
        }
        digit_total[j] = own ? totals[d] : 0u;
        row_offset[j] = own ? table[(size_t)d * n_blocks + blockIdx.x] : 0u;
    }
    __syncthreads();

    const uint32_t seg = blockIdx.x * kBlockItems + wv * kWaveItems;                   // this wave's consecutive items
    KeyT key[IPT];
    uint32_t val[IPT], rank[IPT];
#pragma unroll
    for (int r = 0; r < IPT; ++r) {
        const uint32_t idx = seg + r * kWave + lane;
        const bool valid = idx < n;
        key[r] = valid ? keys_in[idx] : static_cast<KeyT>(key_base);
        val[r] = pl.iota_values ? idx : (valid ? vals_in[idx] : 0u);
    }
#pragma unroll
    for (int r = 0; r < IPT; ++r) {                                                    // input order: round by round, lane by lane
        const bool valid = seg + r * kWave + lane < n;
        const uint32_t d = digit_of(key[r], key_base, shift, mask);
        uint64_t peers = wave_ballot(valid);                                           // lanes of this round holding the same digit
#pragma unroll
        for (int b = 0; b < BITS; ++b) {
            const bool mine = (d & (1u << b)) != 0u;
            const uint64_t set = wave_ballot(mine), clear = ~set;
            peers &= mine ? set : clear;
        }
        const uint32_t before = s_cnt[wv][d];                                          // every lane reads before the leaders write
        rank[r] = before + lanes_below(peers);
        wave_lds_fence();
        if (valid && lanes_below(peers) == 0u) s_cnt[wv][d] = before + static_cast<uint32_t>(__popcll(static_cast<unsigned long long>(peers)));
        wave_lds_fence();
    }
    __syncthreads();
    // per digit: counts of the four waves -> start of each wave's items inside the digit's run; the digit's count in this workgroup ->
    // its first local position (exclusive scan over the digits); its global destination
    uint32_t count[DPT], count_sum = 0, total_sum = 0;
#pragma unroll
    for (int j = 0; j < DPT; ++j) {
        const uint32_t d = d0 + j;
        count[j] = 0;
        if (d < kBins) {
#pragma unroll
            for (int w = 0; w < kWaves; ++w) { const uint32_t c = s_cnt[w][d]; s_cnt[w][d] = count[j]; count[j] += c; }
        }
        count_sum += count[j];
        total_sum += digit_total[j];
    }
    uint32_t unused;
    uint32_t first_local = block_exclusive_scan<kWaves>(count_sum, s_part, unused);
    uint32_t digit_base = block_exclusive_scan<kWaves>(total_sum, s_part, unused);
#pragma unroll
    for (int j = 0; j < DPT; ++j) {
        const uint32_t d = d0 + j;
        if (d < kBins) { s_first[d] = first_local; s_dst[d] = digit_base + row_offset[j]; }
        first_local += count[j];
        digit_base += digit_total[j];
    }
    __syncthreads();
    // items to their workgroup-local sorted position, then out in that order: consecutive lanes -> consecutive addresses per run
#pragma unroll
    for (int r = 0; r < IPT; ++r) {
        if (seg + r * kWave + lane >= n) continue;
        const uint32_t d = digit_of(key[r], key_base, shift, mask);
        const uint32_t pos = s_first[d] + s_cnt[wv][d] + rank[r];
        s_key[pos] = key[r];
        s_val[pos] = val[r];
    }
    __syncthreads();
    const uint32_t block_first = blockIdx.x * kBlockItems;
    const uint32_t n_here = n - block_first < static_cast<uint32_t>(kBlockItems) ? n - block_first : static_cast<uint32_t>(kBlockItems);
    if (pl.rows_in == nullptr) {
        for (uint32_t pos = threadIdx.x; pos < n_here; pos += TH) {
            const KeyT k = s_key[pos];
            const uint32_t d = digit_of(k, key_base, shift, mask);
            const uint32_t dst = s_dst[d] + (pos - s_first[d]);
            keys_out[dst] = k;
            vals_out[dst] = s_val[pos];
        }
        return;
    }
    // Last pass of the depth sort: the values are row indices into a 16-byte side table (preprocess.hip: one footprint row per visible Gaussian, in
    // compaction order). The row is gathered HERE and leaves in sorted order, so that the offsets scan and the instance kernel behind the sort
    // stream it -- their own per-Gaussian random gathers (a 128-byte line for 4 / 16 useful bytes each) were 3x their algorithmic traffic. Four
    // gathers per thread are in flight at a time; a row's first word is the primitive index (= the sorted value), its tile count goes to count_out.
    constexpr int kBatch = 4 < IPT ? 4 : IPT;          // 8 / 16 in flight measured the same (profiles/r05_ab_sort_gather_batch.txt): HBM random access, not latency
    static_assert(IPT % kBatch == 0, "whole gather batches");
#pragma unroll 1
    for (int b = 0; b < IPT / kBatch; ++b) {
        uint4 row[kBatch];
        uint32_t dst[kBatch];
        KeyT key_b[kBatch];
#pragma unroll
        for (int j = 0; j < kBatch; ++j) {
            const uint32_t pos = (b * kBatch + j) * TH + threadIdx.x;
            const bool in = pos < n_here;
            const uint32_t p = in ? pos : 0u;
            key_b[j] = s_key[p];
            const uint32_t d = digit_of(key_b[j], key_base, shift, mask);
            dst[j] = in ? s_dst[d] + (p - s_first[d]) : 0xffffffffu;
            row[j] = pl.rows_in[in ? s_val[p] : 0u];
        }
#pragma unroll
        for (int j = 0; j < kBatch; ++j) {
            if (dst[j] == 0xffffffffu) continue;
            keys_out[dst[j]] = key_b[j];
            vals_out[dst[j]] = row[j].x;
            pl.rows_out[dst[j]] = row[j];
            pl.count_out[dst[j]] = footprint_tile_count(row[j]);
            // boxes the instance kernel gives a workgroup each: listed by their depth-order position (a few hundred to a few thousand per view),
            // collected per workgroup so that the list's counter sees one atomic per workgroup (a same-address atomic retires at ~88 / us)
            if (row[j].y == kFootprintEscape && row[j].w > kBigInstanceFootprint) {
                const uint32_t k = atomicAdd(&s_n_big, 1u);
                if (k < kBigPerBlock) s_big[k] = dst[j];
                else pl.big_list[atomicAdd(pl.big_count, 1u)] = dst[j];
            }
        }
    }
    __syncthreads();
    const uint32_t n_big = s_n_big < kBigPerBlock ? s_n_big : kBigPerBlock;
    if (n_big == 0u) return;                            // workgroup-uniform
    if (threadIdx.x == 0) s_big_base = atomicAdd(pl.big_count, n_big);
    __syncthreads();
    for (uint32_t k = threadIdx.x; k < n_big; k += TH) pl.big_list[s_big_base + k] = s_big[k];
}

struct SortPlan { int n_passes; int bits[8]; uint32_t n_blocks; size_t table_bytes, totals_bytes; };

SortPlan plan_sort(uint32_t n, int end_bit, int max_bits, int block_items_) {
    SortPlan p{};
    p.n_passes = (end_bit + max_bits - 1) / max_bits;
    if (p.n_passes < 1) p.n_passes = 1;
    int left = end_bit;
    for (int i = 0; i < p.n_passes; ++i) { p.bits[i] = (left + (p.n_passes - i) - 1) / (p.n_passes - i); left -= p.bits[i]; }   // even split
    const uint32_t block_items = static_cast<uint32_t>(block_items_);
    p.n_blocks = (n + block_items - 1) / block_items;
    p.table_bytes = ((size_t)kMaxBins * p.n_blocks * sizeof(uint32_t) + 255) / 256 * 256;
    p.totals_bytes = kMaxBins * sizeof(uint32_t);
    return p;
}

template <typename KeyT, int IPT, int TH>
void launch_scatter(int bits, dim3 grid, dim3 block, hipStream_t s, const KeyT* keys_in, const uint32_t* vals_in, KeyT* keys_out, uint32_t* vals_out,
                    uint32_t n, const uint32_t* n_ptr, uint32_t key_base, int shift, const uint32_t* table, const uint32_t* totals, uint32_t n_blocks,
                    const SortPayload& pl) {
#define FGS_SCATTER(B) case B: hipLaunchKernelGGL((radix_scatter_kernel<KeyT, B, IPT, TH>), grid, block, 0, s, keys_in, vals_in, keys_out, vals_out, n, n_ptr, key_base, shift, table, totals, n_blocks, pl); break;
    switch (bits) { FGS_SCATTER(1) FGS_SCATTER(2) FGS_SCATTER(3) FGS_SCATTER(4) FGS_SCATTER(5) FGS_SCATTER(6) FGS_SCATTER(7) FGS_SCATTER(8) default: FGS_SCATTER(9) }
#undef FGS_SCATTER
}

// `n` = item count, or with n_ptr != nullptr an upper bound of the count stored at n_ptr on the device. Keys are sorted by
// (key - key_base) & (2^end_bit - 1): the caller guarantees key >= key_base.
template <typename KeyT, int IPT, int TH = kSortThreads>
hipError_t sort_pairs(void* temp, size_t temp_bytes, KeyT* keys[2], uint32_t* vals[2], int& selector, uint32_t n, const uint32_t* n_ptr,
                      uint32_t key_base, int end_bit, int max_bits, hipStream_t s, const SortPayload* payload = nullptr) {
    selector = 0;
    if (n == 0) return hipSuccess;
    const SortPlan p = plan_sort(n, end_bit, max_bits, IPT * TH);
    if (temp_bytes < p.table_bytes + p.totals_bytes) return hipErrorInvalidValue;
    uint32_t* table = static_cast<uint32_t*>(temp);
    uint32_t* totals = reinterpret_cast<uint32_t*>(static_cast<char*>(temp) + p.table_bytes);
    const dim3 grid(p.n_blocks), block(TH);
    int shift = 0;
    for (int i = 0; i < p.n_passes; ++i) {
        const int bits = p.bits[i];
        hipLaunchKernelGGL((radix_histogram_kernel<KeyT, IPT, TH>), grid, block, 0, s, keys[selector], n, n_ptr, key_base, shift, bits, table, p.n_blocks);
        hipLaunchKernelGGL(radix_row_scan_kernel, dim3(1u << bits), dim3(kSortThreads), 0, s, table, totals, p.n_blocks);
        SortPayload pl{};                                   // with a payload: the values are the input positions (first pass) and become the rows' primitives (last pass)
        if (payload != nullptr) {
            pl.iota_values = i == 0 ? 1 : 0;
            if (i == p.n_passes - 1) { pl.rows_in = payload->rows_in; pl.rows_out = payload->rows_out; pl.count_out = payload->count_out;
                                       pl.big_list = payload->big_list; pl.big_count = payload->big_count; }
        }
        launch_scatter<KeyT, IPT, TH>(bits, grid, block, s, keys[selector], vals[selector], keys[selector ^ 1], vals[selector ^ 1], n, n_ptr, key_base, shift,
                                  table, totals, p.n_blocks, pl);
        selector ^= 1;
        shift += bits;
    }
    return hipGetLastError();
}

}  // namespace sortimpl
using namespace sortimpl;

// g_depth_sort_mode (fgs_kernels.h; fgs_debug_set_option(9, m) in the dev build) -- bit 0: sort key - bits(near) in ceil(bits / 9) passes (near 0.2,
// far 1e4: 27 bits = 3 passes instead of 4); bit 1: 2048-item workgroups (8 items per thread); 0 = round 1 (4 x 8 bits, 4096 items).
// tools/ab_depth_sort.py, S2 (2 M keys), one process: mode 0 0.108 ms, 1 0.096, 2 0.119, 3 0.117

// Workgroup shape of both sorts (the kernels are templates on it): **8192 items over 512 threads** (round 5; 4096 over 256 before). Twice the items per
// workgroup double the length of a digit's run in the scatter (depth sort, 512 digits: 8 -> 16 items = 32 -> 64-byte stores; tile sort, 128 digits:
// 32 -> 64 items) and halve the [digit][workgroup] table, its row scans and the histogram workgroups; twice the waves keep the ranking rounds per wave
// at 16. Measured on one box (profiles/r05_ab_sort_blocks.txt, S2): depth sort 0.128 -> 0.115 ms, tile sort 0.164 -> 0.148. Other shapes: depth sort
// 4096 items over 512 / 1024 threads 0.125 / 0.125, 8192 over 1024 0.117, 12288 over 1024 0.124; tile sort 4096 over 512 0.158, 8192 over 1024 0.197,
// 12288 over 512 (24 items per thread: registers) 0.217; 16384 items do not fit the LDS with 32-bit keys.
constexpr int kTileSortThreads = 512, kTileSortItems = 8192 / kTileSortThreads, kDepthSortItems = 8, kGenericMaxBits = 8;
constexpr int kDepthSortThreads = 512, kDepthSortIpt = 8192 / kDepthSortThreads;

size_t own_sort_temp_bytes(uint32_t n, int end_bit) {                                  // fits every configuration above (smallest workgroups, full table)
    const SortPlan p = plan_sort(n, end_bit, kGenericMaxBits, kDepthSortItems * kSortThreads);
    return p.table_bytes + p.totals_bytes;
}

hipError_t own_sort_pairs_u32(void* temp, size_t temp_bytes, uint32_t* keys[2], uint32_t* vals[2], int& selector, uint32_t n, int end_bit, hipStream_t s) {
    return sort_pairs<uint32_t, kTileSortItems, kTileSortThreads>(temp, temp_bytes, keys, vals, selector, n, nullptr, 0u, end_bit, kGenericMaxBits, s);
}
hipError_t own_sort_pairs_u16(void* temp, size_t temp_bytes, uint16_t* keys[2], uint32_t* vals[2], int& selector, uint32_t n, int end_bit, hipStream_t s) {
    return sort_pairs<uint16_t, kTileSortItems, kTileSortThreads>(temp, temp_bytes, keys, vals, selector, n, nullptr, 0u, end_bit, kGenericMaxBits, s);
}
hipError_t own_sort_pairs_u32_device_count(void* temp, size_t temp_bytes, uint32_t* keys[2], uint32_t* vals[2], int& selector, uint32_t capacity,
                                           const uint32_t* n_ptr, int end_bit, hipStream_t s) {
    return sort_pairs<uint32_t, kTileSortItems, kTileSortThreads>(temp, temp_bytes, keys, vals, selector, capacity, n_ptr, 0u, end_bit, kGenericMaxBits, s);
}
hipError_t own_sort_pairs_u16_device_count(void* temp, size_t temp_bytes, uint16_t* keys[2], uint32_t* vals[2], int& selector, uint32_t capacity,
                                           const uint32_t* n_ptr, int end_bit, hipStream_t s) {
    return sort_pairs<uint16_t, kTileSortItems, kTileSortThreads>(temp, temp_bytes, keys, vals, selector, capacity, n_ptr, 0u, end_bit, kGenericMaxBits, s);
}

// Depth keys are the bit patterns of positive depths that passed the near / far cull (kf:67), i.e. values in [bits(near), bits(far)]:
// sorting key - bits(near) gives the same order in fewer bits.
DepthKeyRange depth_key_range(float near_plane, float far_plane) {
    DepthKeyRange r{0u, 32};
    if (!(near_plane >= 0.0f) || !(far_plane >= near_plane)) return r;                 // negative / NaN planes: no assumption, all 32 bits
    uint32_t lo, hi;
    std::memcpy(&lo, &near_plane, 4); std::memcpy(&hi, &far_plane, 4);
    const uint32_t span = hi - lo;
    int bits = 1;
    while (bits < 32 && (span >> bits) != 0u) ++bits;
    r.base = lo; r.bits = bits;
    return r;
}

// `n` = visible count, or with n_ptr != nullptr a bound of the count stored at n_ptr on the device
hipError_t own_depth_sort(void* temp, size_t temp_bytes, uint32_t* keys[2], uint32_t* vals[2], int& selector, uint32_t n, const uint32_t* n_ptr,
                          DepthKeyRange range, hipStream_t s, const SortPayload* payload) {
    const int mode = g_depth_sort_mode;
    const uint32_t base = (mode & 1) ? range.base : 0u;
    const int end_bit = (mode & 1) ? range.bits : 32, max_bits = (mode & 1) ? kMaxBits : kGenericMaxBits;
    if ((mode & 2) && payload == nullptr) return sort_pairs<uint32_t, kDepthSortItems>(temp, temp_bytes, keys, vals, selector, n, n_ptr, base, end_bit, max_bits, s);
    return sort_pairs<uint32_t, kDepthSortIpt, kDepthSortThreads>(temp, temp_bytes, keys, vals, selector, n, n_ptr, base, end_bit, max_bits, s, payload);
}

}  // namespace fgs
