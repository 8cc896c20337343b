// Stable LSD radix sort of (key, uint32 value) pairs for the two sorts of the binning stage (K2: 32-bit depth keys of the
// visible Gaussians; K6: tile keys of the instances), written for their sizes on gfx950.
//
// Why not rocPRIM: its onesweep sort is tuned for large inputs -- 16 384 items per 1024-thread workgroup (80 B of scratch per
// lane), one histogram fill plus two fills per 8-bit pass. The depth sort has ~2 M items: ~125 workgroups for 256 CUs behind a
// decoupled look-back chain, 14 launches, 0.167 ms for 32 MB of traffic; the tile sort moves 16 M items at 1.7 TB/s (0.266 ms).
// Here a pass is three launches over 4096-item workgroups -- per-workgroup digit histogram -> one workgroup per digit scans its
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
#include "fgs_kernels.h"
#include <fgs_wave.h>

namespace fgs {

namespace sortimpl {

constexpr int kSortThreads = 256, kSortItemsPerThread = 16, kSortWaves = kSortThreads / kWave;
constexpr int kSortBlockItems = kSortThreads * kSortItemsPerThread;                    // 4096
constexpr int kSortWaveItems = kSortBlockItems / kSortWaves;                           // 1024 = 16 rounds of 64
constexpr int kMaxBits = 8, kMaxBins = 1 << kMaxBits;                                  // one thread per digit in the block-wide scans

template <typename KeyT>
__device__ __forceinline__ uint32_t digit_of(KeyT key, int shift, uint32_t mask) { return (static_cast<uint32_t>(key) >> shift) & mask; }

// exclusive prefix of one value per thread over the 256-thread workgroup; `total` = sum of all
__device__ __forceinline__ uint32_t block_exclusive_scan(uint32_t v, uint32_t* s_part /*[kSortWaves]*/, uint32_t& total) {
    const uint32_t lane = threadIdx.x & 63u, wv = threadIdx.x >> 6;
    const uint32_t excl = wave_exclusive_sum(v);
    __syncthreads();                                                 // s_part may still be read from a previous call
    if (lane == kWave - 1) s_part[wv] = excl + v;
    __syncthreads();
    uint32_t base = 0, sum = 0;
#pragma unroll
    for (uint32_t w = 0; w < kSortWaves; ++w) { const uint32_t p = s_part[w]; base += w < wv ? p : 0u; sum += p; }
    total = sum;
    return base + excl;
}

// per-workgroup digit histogram, written digit-major: hist[digit * n_blocks + block]
// The item count comes by value or -- when the host does not know it yet -- through `n_ptr` (grid sized by a capacity, workgroups
// beyond the count contribute zero rows and scatter nothing).
template <typename KeyT>
__global__ void __launch_bounds__(kSortThreads) radix_histogram_kernel(const KeyT* __restrict__ keys, const uint32_t n_value, const uint32_t* __restrict__ n_ptr,
                                                                       const int shift, const int bits, uint32_t* __restrict__ hist, const uint32_t n_blocks) {
    __shared__ uint32_t s_hist[kMaxBins];
    const uint32_t n = n_ptr != nullptr ? *n_ptr : n_value;
    const uint32_t bins = 1u << bits, mask = bins - 1u;
    if (threadIdx.x < bins) s_hist[threadIdx.x] = 0u;
    __syncthreads();
    const uint32_t base = blockIdx.x * kSortBlockItems;
    constexpr int kPerLoad = 16 / sizeof(KeyT);                                        // keys per 16-byte load
    if (base + kSortBlockItems <= n) {                                                 // full workgroup: 16-byte loads (order is irrelevant here)
#pragma unroll
        for (int i = 0; i < kSortItemsPerThread / kPerLoad; ++i) {
            const uint4 q = reinterpret_cast<const uint4*>(keys + base)[i * kSortThreads + threadIdx.x];
            const uint32_t w[4] = {q.x, q.y, q.z, q.w};
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                if (sizeof(KeyT) == 4) atomicAdd(&s_hist[(w[j] >> shift) & mask], 1u);
                else { atomicAdd(&s_hist[((w[j] & 0xffffu) >> shift) & mask], 1u); atomicAdd(&s_hist[((w[j] >> 16) >> shift) & mask], 1u); }
            }
        }
    } else {
#pragma unroll
        for (int i = 0; i < kSortItemsPerThread; ++i) {
            const uint32_t idx = base + i * kSortThreads + threadIdx.x;
            if (idx < n) atomicAdd(&s_hist[digit_of(keys[idx], shift, mask)], 1u);
        }
    }
    __syncthreads();
    if (threadIdx.x < bins) hist[(size_t)threadIdx.x * n_blocks + blockIdx.x] = s_hist[threadIdx.x];
}

// One workgroup per digit: exclusive scan of that digit's row of the table (over the workgroups of the sort), in place, and the
// row total. The scatter kernel adds the exclusive scan of the <= 256 totals itself.
__global__ void __launch_bounds__(kSortThreads) radix_row_scan_kernel(uint32_t* __restrict__ table, uint32_t* __restrict__ totals, const uint32_t n_blocks) {
    __shared__ uint32_t s_part[kSortWaves];
    uint32_t* row = table + (size_t)blockIdx.x * n_blocks;
    uint32_t carry = 0;
    for (uint32_t c0 = 0; c0 < n_blocks; c0 += kSortBlockItems) {                      // workgroup-uniform trip count
        uint32_t v[kSortItemsPerThread], sum = 0;
        const uint32_t first = c0 + threadIdx.x * kSortItemsPerThread;
#pragma unroll
        for (int i = 0; i < kSortItemsPerThread; ++i) { v[i] = first + i < n_blocks ? row[first + i] : 0u; sum += v[i]; }
        uint32_t chunk_total;
        uint32_t run = carry + block_exclusive_scan(sum, s_part, chunk_total);
#pragma unroll
        for (int i = 0; i < kSortItemsPerThread; ++i) { if (first + i < n_blocks) row[first + i] = run; run += v[i]; }
        carry += chunk_total;
    }
    if (threadIdx.x == 0) totals[blockIdx.x] = carry;
}

// table[d * n_blocks + blk] (after the row scan) + exclusive scan of totals[] over d = where this workgroup's first item with
// digit d goes.
// BITS (the digit width) is a template parameter so that the match loop is straight-line code: as a run-time loop it cost
// 8 VALU + 4 SALU + a branch per bit and round.
template <typename KeyT, int BITS>
__global__ void __launch_bounds__(kSortThreads) radix_scatter_kernel(const KeyT* __restrict__ keys_in, const uint32_t* __restrict__ vals_in,
                                                                     KeyT* __restrict__ keys_out, uint32_t* __restrict__ vals_out,
                                                                     const uint32_t n_value, const uint32_t* __restrict__ n_ptr, const int shift,
                                                                     const uint32_t* __restrict__ table,
                                                                     const uint32_t* __restrict__ totals, const uint32_t n_blocks) {
    constexpr int bits = BITS;
    const uint32_t n = n_ptr != nullptr ? *n_ptr : n_value;
    if (blockIdx.x * kSortBlockItems >= n) return;                  // workgroup-uniform (capacity-sized grid)
    __shared__ uint32_t s_cnt[kSortWaves][kMaxBins];              // per wave and digit: running count, later start inside the digit's run
    __shared__ uint32_t s_first[kMaxBins];                        // first workgroup-local position of each digit
    __shared__ uint32_t s_dst[kMaxBins];                          // global position of this workgroup's first item of each digit
    __shared__ uint32_t s_part[kSortWaves];
    __shared__ KeyT s_key[kSortBlockItems];
    __shared__ uint32_t s_val[kSortBlockItems];
    const uint32_t bins = 1u << bits, mask = bins - 1u;
    const uint32_t lane = lane_id(), wv = threadIdx.x >> 6;
    if (threadIdx.x < bins) {
#pragma unroll
        for (int w = 0; w < kSortWaves; ++w) s_cnt[w][threadIdx.x] = 0u;
    }
    // global base of every digit: exclusive scan of the digit totals (requested now, used after the ranking)
    const uint32_t digit_total = threadIdx.x < bins ? totals[threadIdx.x] : 0u;
    const uint32_t row_offset = threadIdx.x < bins ? table[(size_t)threadIdx.x * n_blocks + blockIdx.x] : 0u;
    __syncthreads();

    const uint32_t seg = blockIdx.x * kSortBlockItems + wv * kSortWaveItems;          // this wave's 1024 consecutive items
    KeyT key[kSortItemsPerThread];
    uint32_t val[kSortItemsPerThread], rank[kSortItemsPerThread];
#pragma unroll
    for (int r = 0; r < kSortItemsPerThread; ++r) {
        const uint32_t idx = seg + r * kWave + lane;
        const bool valid = idx < n;
        key[r] = valid ? keys_in[idx] : static_cast<KeyT>(0);
        val[r] = valid ? vals_in[idx] : 0u;
    }
#pragma unroll
    for (int r = 0; r < kSortItemsPerThread; ++r) {                                    // input order: round by round, lane by lane
        const bool valid = seg + r * kWave + lane < n;
        const uint32_t d = digit_of(key[r], shift, mask);
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
    // per digit (one thread each): counts of the four waves -> start of each wave's items inside the digit's run; the digit's
    // count in this workgroup -> its first local position (exclusive scan over the digits); its global destination
    uint32_t count = 0;
    if (threadIdx.x < bins) {
#pragma unroll
        for (int w = 0; w < kSortWaves; ++w) { const uint32_t c = s_cnt[w][threadIdx.x]; s_cnt[w][threadIdx.x] = count; count += c; }
    }
    uint32_t unused;
    const uint32_t first_local = block_exclusive_scan(count, s_part, unused);
    const uint32_t digit_base = block_exclusive_scan(digit_total, s_part, unused);
    if (threadIdx.x < bins) { s_first[threadIdx.x] = first_local; s_dst[threadIdx.x] = digit_base + row_offset; }
    __syncthreads();
    // items to their workgroup-local sorted position, then out in that order: consecutive lanes -> consecutive addresses per run
#pragma unroll
    for (int r = 0; r < kSortItemsPerThread; ++r) {
        if (seg + r * kWave + lane >= n) continue;
        const uint32_t d = digit_of(key[r], shift, mask);
        const uint32_t pos = s_first[d] + s_cnt[wv][d] + rank[r];
        s_key[pos] = key[r];
        s_val[pos] = val[r];
    }
    __syncthreads();
    const uint32_t block_first = blockIdx.x * kSortBlockItems;
    const uint32_t n_here = n - block_first < static_cast<uint32_t>(kSortBlockItems) ? n - block_first : static_cast<uint32_t>(kSortBlockItems);
    for (uint32_t pos = threadIdx.x; pos < n_here; pos += kSortThreads) {
        const KeyT k = s_key[pos];
        const uint32_t d = digit_of(k, shift, mask);
        const uint32_t dst = s_dst[d] + (pos - s_first[d]);
        keys_out[dst] = k;
        vals_out[dst] = s_val[pos];
    }
}

struct SortPlan { int n_passes; int bits[8]; uint32_t n_blocks; size_t table_bytes, totals_bytes; };

SortPlan plan_sort(uint32_t n, int end_bit) {
    SortPlan p{};
    p.n_passes = (end_bit + kMaxBits - 1) / kMaxBits;
    if (p.n_passes < 1) p.n_passes = 1;
    int left = end_bit;
    for (int i = 0; i < p.n_passes; ++i) { p.bits[i] = (left + (p.n_passes - i) - 1) / (p.n_passes - i); left -= p.bits[i]; }   // even split
    p.n_blocks = (n + kSortBlockItems - 1) / kSortBlockItems;
    p.table_bytes = ((size_t)kMaxBins * p.n_blocks * sizeof(uint32_t) + 255) / 256 * 256;
    p.totals_bytes = 1024;
    return p;
}

template <typename KeyT>
void launch_scatter(int bits, dim3 grid, dim3 block, hipStream_t s, const KeyT* keys_in, const uint32_t* vals_in, KeyT* keys_out, uint32_t* vals_out,
                    uint32_t n, const uint32_t* n_ptr, int shift, const uint32_t* table, const uint32_t* totals, uint32_t n_blocks) {
#define FGS_SCATTER(B) case B: hipLaunchKernelGGL((radix_scatter_kernel<KeyT, B>), grid, block, 0, s, keys_in, vals_in, keys_out, vals_out, n, n_ptr, shift, table, totals, n_blocks); break;
    switch (bits) { FGS_SCATTER(1) FGS_SCATTER(2) FGS_SCATTER(3) FGS_SCATTER(4) FGS_SCATTER(5) FGS_SCATTER(6) FGS_SCATTER(7) default: FGS_SCATTER(8) }
#undef FGS_SCATTER
}

// `n` = item count, or with n_ptr != nullptr an upper bound of the count stored at n_ptr on the device
template <typename KeyT>
hipError_t sort_pairs(void* temp, size_t temp_bytes, KeyT* keys[2], uint32_t* vals[2], int& selector, uint32_t n, const uint32_t* n_ptr,
                      int end_bit, hipStream_t s) {
    selector = 0;
    if (n == 0) return hipSuccess;
    const SortPlan p = plan_sort(n, end_bit);
    if (temp_bytes < p.table_bytes + p.totals_bytes) return hipErrorInvalidValue;
    uint32_t* table = static_cast<uint32_t*>(temp);
    uint32_t* totals = reinterpret_cast<uint32_t*>(static_cast<char*>(temp) + p.table_bytes);
    const dim3 grid(p.n_blocks), block(kSortThreads);
    int shift = 0;
    for (int i = 0; i < p.n_passes; ++i) {
        const int bits = p.bits[i];
        hipLaunchKernelGGL(radix_histogram_kernel<KeyT>, grid, block, 0, s, keys[selector], n, n_ptr, shift, bits, table, p.n_blocks);
        hipLaunchKernelGGL(radix_row_scan_kernel, dim3(1u << bits), block, 0, s, table, totals, p.n_blocks);
        launch_scatter<KeyT>(bits, grid, block, s, keys[selector], vals[selector], keys[selector ^ 1], vals[selector ^ 1], n, n_ptr, shift, table,
                             totals, p.n_blocks);
        selector ^= 1;
        shift += bits;
    }
    return hipGetLastError();
}

}  // namespace sortimpl
using namespace sortimpl;

std::atomic<int> g_sort_implementation{3};          // bit 0: tile sort here, bit 1: depth sort here; cleared bit = rocPRIM onesweep (fgs_debug_set_option key 6)

size_t own_sort_temp_bytes(uint32_t n, int end_bit) {
    const SortPlan p = plan_sort(n, end_bit);
    return p.table_bytes + p.totals_bytes;
}

hipError_t own_sort_pairs_u32(void* temp, size_t temp_bytes, uint32_t* keys[2], uint32_t* vals[2], int& selector, uint32_t n, int end_bit, hipStream_t s) {
    return sort_pairs<uint32_t>(temp, temp_bytes, keys, vals, selector, n, nullptr, end_bit, s);
}
hipError_t own_sort_pairs_u16(void* temp, size_t temp_bytes, uint16_t* keys[2], uint32_t* vals[2], int& selector, uint32_t n, int end_bit, hipStream_t s) {
    return sort_pairs<uint16_t>(temp, temp_bytes, keys, vals, selector, n, nullptr, end_bit, s);
}
hipError_t own_sort_pairs_u32_device_count(void* temp, size_t temp_bytes, uint32_t* keys[2], uint32_t* vals[2], int& selector, uint32_t capacity,
                                           const uint32_t* n_ptr, int end_bit, hipStream_t s) {
    return sort_pairs<uint32_t>(temp, temp_bytes, keys, vals, selector, capacity, n_ptr, end_bit, s);
}
hipError_t own_sort_pairs_u16_device_count(void* temp, size_t temp_bytes, uint16_t* keys[2], uint32_t* vals[2], int& selector, uint32_t capacity,
                                           const uint32_t* n_ptr, int end_bit, hipStream_t s) {
    return sort_pairs<uint16_t>(temp, temp_bytes, keys, vals, selector, capacity, n_ptr, end_bit, s);
}

}  // namespace fgs
