// Maintenance of the Gaussian set on the device: adaptive density control, pruning, re-ordering and the Morton
// order, INCLUDING the Adam-state surgery, as three kinds of pass over the 59 + 2 x 59 floats of every Gaussian:
//   classify (one lane per Gaussian) -> 4-way exclusive scan (own kernels) -> scatter of parameters and both moments        adaptive_density_control
//   gather of parameters and both moments through an index list                                              prune / sort
//   30-bit Morton key (one lane per Gaussian) + the stable radix sort of radix_sort.hip                       apply_morton_ordering
// Semantics: reference Model.py:312-366 (clone small / split large above the gradient threshold, then prune), :275-306 (prune, sort),
// :459-463 (Morton order; the encoder itself lives in the un-vendored NeRFICG CudaUtils, the 10-bit-per-axis curve of
// harness/scenes.py is used). The reference runs these as ~60 torch mask / index / cat calls on the six tensors plus NeRFICG's
// extend / prune / sort_param_groups on the optimizer state; at 3 M Gaussians that chain allocates and copies every tensor several times.
// Here every tensor is read once and written once: ~3 x 236 B x N each way.
//
// Output order of adaptive density control (identical to the reference's cat / boolean-index result): the surviving old Gaussians in
// their order, then the surviving clones in the order of their originals, then the surviving first children of the split Gaussians,
// then the surviving second children. New Gaussians start with zero Adam moments (extend_param_groups), survivors keep theirs.
// Built with -ffp-contract=off (Makefile): keys and thresholds reproduce the numpy restatement of Model.py that the tests compare with, bit for bit.
#include "fgs_kernels.h"
#include <fgs_wave.h>
#include "fgs_tile_scan.h"

namespace fgs {

// per-Gaussian plan word: bit 0 keep the old one, bit 1 keep its clone, bit 2 keep its two children, bit 3 it is split at all
__global__ void __launch_bounds__(256) adc_classify_kernel(const AdcPlanArgs a) {
    const uint32_t i = blockIdx.x * 256u + threadIdx.x;
    if (i >= a.n) return;
    const float count = a.densification_info[i], grad_sum = a.densification_info[(size_t)a.n + i];
    const bool densify = grad_sum >= a.grad_threshold * fmaxf(count, 1.0f);                                   // Model.py:314
    const float s0 = a.scales[3 * (size_t)i], s1 = a.scales[3 * (size_t)i + 1], s2 = a.scales[3 * (size_t)i + 2];
    const float s_max = fmaxf(s0, fmaxf(s1, s2));
    const bool is_small = s_max <= a.log_small;                                                                // :315
    const bool duplicate = densify && is_small, split = densify && !is_small;                                  // :318, :328
    const float q0 = a.rotations[4 * (size_t)i], q1 = a.rotations[4 * (size_t)i + 1], q2 = a.rotations[4 * (size_t)i + 2], q3 = a.rotations[4 * (size_t)i + 3];
    const bool dead = a.opacities[i] < a.min_opacity_logit || (q0 * q0 + q1 * q1 + q2 * q2 + q3 * q3) < 1e-8f;  // :360-361
    const bool too_large = a.prune_large && s_max > a.log_large;                                               // :362-363
    // the children's scales are log(0.625 exp(s)) (:333): the same expression the scatter pass stores, so the test sees the stored value
    const float child_max = fmaxf(logf(expf(s0) * 0.625f), fmaxf(logf(expf(s1) * 0.625f), logf(expf(s2) * 0.625f)));
    const bool child_too_large = a.prune_large && child_max > a.log_large;
    uint32_t word = 0;
    if (!split && !dead && !too_large) word |= 1u;                // :359: a split Gaussian is replaced by its children
    if (duplicate && !dead && !too_large) word |= 2u;             // a clone is a copy: same verdict as its original
    if (split && !dead && !child_too_large) word |= 4u;
    if (split) word |= 8u;
    a.plan[i] = word;
}

// Exclusive scan of the four plan bits over all Gaussians, offsets[i] = (survivor, clone, child, split) ranks in front of Gaussian i, as three
// small kernels of this repository's own (until round 5 one rocPRIM look-back scan over uint4 counters -- the last library call of the package):
//   adc_block_sums_kernel   a 256-thread workgroup owns kAdcBlock = 4096 consecutive Gaussians, a thread 16 consecutive plan words (four 16-byte
//                           loads); the four counts travel PACKED, two 16-bit fields per 32-bit word (a workgroup's count is <= 4096), through one
//                           DPP wave scan per word -> block_sums[b]
//   adc_scan_blocks_kernel  ONE 1024-thread workgroup: exclusive scan of the block sums (fgs_tile_scan.h, 16 Ki blocks = 67 M Gaussians per pass),
//                           in place, and the four totals
//   adc_offsets_kernel      the same local scan again, now written out: block base + ranks inside the block
// 24 bytes per Gaussian in all; runs every 100 iterations (Model.py:312).
constexpr uint32_t kAdcPerThread = 16, kAdcBlock = 256u * kAdcPerThread;

__device__ __forceinline__ uint32_t plan_lo(const uint32_t w) { return (w & 1u) | (((w >> 1) & 1u) << 16); }          // survivors | clones << 16
__device__ __forceinline__ uint32_t plan_hi(const uint32_t w) { return ((w >> 2) & 1u) | (((w >> 3) & 1u) << 16); }   // children | split << 16

// the 16 plan words of this thread (0 beyond the end), their packed exclusive prefixes inside the thread, and the thread's packed totals
__device__ __forceinline__ void adc_thread_items(const uint32_t* __restrict__ plan, const uint32_t n, const uint32_t first, uint32_t (&w)[kAdcPerThread],
                                                 uint32_t (&ex_lo)[kAdcPerThread], uint32_t (&ex_hi)[kAdcPerThread], uint32_t& lo, uint32_t& hi) {
    if (first + kAdcPerThread <= n) {
#pragma unroll
        for (uint32_t q = 0; q < kAdcPerThread / 4u; ++q) {
            const uint4 v = *reinterpret_cast<const uint4*>(plan + first + 4u * q);          // first is a multiple of 16: 64-byte aligned
            w[4 * q] = v.x; w[4 * q + 1] = v.y; w[4 * q + 2] = v.z; w[4 * q + 3] = v.w;
        }
    } else {
#pragma unroll
        for (uint32_t k = 0; k < kAdcPerThread; ++k) w[k] = first + k < n ? plan[first + k] : 0u;
    }
    lo = 0u; hi = 0u;
#pragma unroll
    for (uint32_t k = 0; k < kAdcPerThread; ++k) { ex_lo[k] = lo; ex_hi[k] = hi; lo += plan_lo(w[k]); hi += plan_hi(w[k]); }
}

// packed exclusive prefix of (lo, hi) over the 256 threads of the workgroup; returns the workgroup's packed totals through tot_lo / tot_hi
__device__ __forceinline__ void adc_workgroup_scan(const uint32_t lo, const uint32_t hi, uint32_t& before_lo, uint32_t& before_hi, uint32_t& tot_lo,
                                                   uint32_t& tot_hi, uint32_t (&s_tot)[2][4]) {
    const uint32_t lane = threadIdx.x & 63u, wv = threadIdx.x >> 6;
    const uint32_t inc_lo = wave_inclusive_sum(lo), inc_hi = wave_inclusive_sum(hi);
    if (lane == 63u) { s_tot[0][wv] = inc_lo; s_tot[1][wv] = inc_hi; }
    __syncthreads();
    before_lo = inc_lo - lo; before_hi = inc_hi - hi; tot_lo = 0u; tot_hi = 0u;
#pragma unroll
    for (uint32_t k = 0; k < 4u; ++k) {
        const uint32_t a = s_tot[0][k], b = s_tot[1][k];
        before_lo += k < wv ? a : 0u; before_hi += k < wv ? b : 0u;
        tot_lo += a; tot_hi += b;
    }
}

__global__ void __launch_bounds__(256) adc_block_sums_kernel(const uint32_t* __restrict__ plan, uint4* __restrict__ block_sums, const uint32_t n) {
    __shared__ uint32_t s_tot[2][4];
    uint32_t w[kAdcPerThread], ex_lo[kAdcPerThread], ex_hi[kAdcPerThread], lo, hi, before_lo, before_hi, tot_lo, tot_hi;
    adc_thread_items(plan, n, blockIdx.x * kAdcBlock + threadIdx.x * kAdcPerThread, w, ex_lo, ex_hi, lo, hi);
    adc_workgroup_scan(lo, hi, before_lo, before_hi, tot_lo, tot_hi, s_tot);
    if (threadIdx.x == 0) block_sums[blockIdx.x] = make_uint4(tot_lo & 0xffffu, tot_lo >> 16, tot_hi & 0xffffu, tot_hi >> 16);
}

__global__ void __launch_bounds__(kTileScanThreads) adc_scan_blocks_kernel(uint4* __restrict__ block_sums, uint32_t* __restrict__ totals, const uint32_t n_blocks) {
    __shared__ TileScanShared s;
    uint32_t base[4] = {0u, 0u, 0u, 0u};
    int parity = 0;
    for (uint32_t first = 0; first < n_blocks; first += kTileScanThreads * kTileScanPerThread) {
        const uint32_t mine = first + threadIdx.x * kTileScanPerThread;
        uint32_t v[4][kTileScanPerThread], ex[4][kTileScanPerThread];
#pragma unroll
        for (int k = 0; k < kTileScanPerThread; ++k) {
            const uint4 b = mine + k < n_blocks ? block_sums[mine + k] : make_uint4(0u, 0u, 0u, 0u);
            v[0][k] = b.x; v[1][k] = b.y; v[2][k] = b.z; v[3][k] = b.w;
        }
#pragma unroll
        for (int c = 0; c < 4; ++c) { base[c] += tile_scan_pass(v[c], ex[c], s, base[c], parity); parity ^= 1; }
#pragma unroll
        for (int k = 0; k < kTileScanPerThread; ++k)
            if (mine + k < n_blocks) block_sums[mine + k] = make_uint4(ex[0][k], ex[1][k], ex[2][k], ex[3][k]);
    }
    if (threadIdx.x == 0) { totals[0] = base[0]; totals[1] = base[1]; totals[2] = base[2]; totals[3] = base[3]; }
}

__global__ void __launch_bounds__(256) adc_offsets_kernel(const uint32_t* __restrict__ plan, const uint4* __restrict__ block_base, uint4* __restrict__ offsets,
                                                          const uint32_t n) {
    __shared__ uint32_t s_tot[2][4];
    uint32_t w[kAdcPerThread], ex_lo[kAdcPerThread], ex_hi[kAdcPerThread], lo, hi, before_lo, before_hi, tot_lo, tot_hi;
    const uint32_t first = blockIdx.x * kAdcBlock + threadIdx.x * kAdcPerThread;
    adc_thread_items(plan, n, first, w, ex_lo, ex_hi, lo, hi);
    adc_workgroup_scan(lo, hi, before_lo, before_hi, tot_lo, tot_hi, s_tot);
    const uint4 base = block_base[blockIdx.x];
#pragma unroll
    for (uint32_t k = 0; k < kAdcPerThread; ++k) {
        if (first + k >= n) break;
        const uint32_t l = before_lo + ex_lo[k], h = before_hi + ex_hi[k];
        offsets[first + k] = make_uint4(base.x + (l & 0xffffu), base.y + (l >> 16), base.z + (h & 0xffffu), base.w + (h >> 16));
    }
}

// Scatter of ONE parameter group (row width W floats) and its two moments. One thread per source float: coalesced reads, writes in runs.
// KIND 0: plain copy; 1: means (children move by R(q) (exp(s) * noise), Model.py:331-332); 2: scales (children get log(0.625 exp(s)), :333).
template <int KIND>
__global__ void __launch_bounds__(256) adc_scatter_kernel(const AdcScatterArgs a) {
    const uint64_t e = (uint64_t)blockIdx.x * 256u + threadIdx.x;
    if (e >= (uint64_t)a.n * a.width) return;
    const uint32_t i = static_cast<uint32_t>(e / a.width), j = static_cast<uint32_t>(e - (uint64_t)i * a.width);
    const uint32_t w = a.plan[i];
    if ((w & 7u) == 0u) return;
    const uint4 off = a.offsets[i];
    const uint32_t n_old = a.totals[0], n_clone = a.totals[1], n_child = a.totals[2], n_split = a.totals[3];
    const float p = a.in_p[e];
    const bool has_state = a.in_m != nullptr;
    if (w & 1u) {
        const size_t d = (size_t)off.x * a.width + j;
        a.out_p[d] = p;
        if (has_state) { a.out_m[d] = a.in_m[e]; a.out_v[d] = a.in_v[e]; }
    }
    if (w & 2u) {
        const size_t d = (size_t)(n_old + off.y) * a.width + j;
        a.out_p[d] = p;
        if (has_state) { a.out_m[d] = 0.0f; a.out_v[d] = 0.0f; }
    }
    if (w & 4u) {
#pragma unroll
        for (uint32_t c = 0; c < 2u; ++c) {
            float v = p;
            if (KIND == 1) {
                // offsets = R(q / |q|) (exp(s) * noise): row j of the rotation matrix (Model.py:331; noise row = copy * n_split + split rank)
                const float* q = a.rotations + 4 * (size_t)i;
                const float* s = a.scales + 3 * (size_t)i;
                const float* z = a.noise + 3 * ((size_t)c * n_split + off.w);
                const float nrm = sqrtf(q[0] * q[0] + q[1] * q[1] + q[2] * q[2] + q[3] * q[3]);
                const float r = q[0] / nrm, x = q[1] / nrm, y = q[2] / nrm, zz = q[3] / nrm;
                const float t0 = expf(s[0]) * z[0], t1 = expf(s[1]) * z[1], t2 = expf(s[2]) * z[2];
                float R0, R1, R2;
                if (j == 0u) { R0 = 1.0f - 2.0f * (y * y + zz * zz); R1 = 2.0f * (x * y - r * zz); R2 = 2.0f * (x * zz + r * y); }
                else if (j == 1u) { R0 = 2.0f * (x * y + r * zz); R1 = 1.0f - 2.0f * (x * x + zz * zz); R2 = 2.0f * (y * zz - r * x); }
                else { R0 = 2.0f * (x * zz - r * y); R1 = 2.0f * (y * zz + r * x); R2 = 1.0f - 2.0f * (x * x + y * y); }
                v = p + ((R0 * t0 + R1 * t1) + R2 * t2);
            }
            if (KIND == 2) v = logf(expf(p) * 0.625f);
            const size_t d = (size_t)(n_old + n_clone + c * n_child + off.z) * a.width + j;
            a.out_p[d] = v;
            if (has_state) { a.out_m[d] = 0.0f; a.out_v[d] = 0.0f; }
        }
    }
}

// out[r, :] = in[index[r], :] for up to kGatherTensors tensors of different row widths in ONE launch (prune = gather through the list of
// survivors, sort = gather through the ordering; Model.py:275-306). One thread per output float; blockIdx ranges select the tensor.
__global__ void __launch_bounds__(256) gather_rows_kernel(const GatherArgs a) {
    int t = 0;
#pragma unroll
    for (int k = 1; k < kGatherTensors; ++k) if (k < a.n_tensors && blockIdx.x >= a.t[k].first_block) t = k;
    const GatherTensor& T = a.t[t];
    const uint64_t e = (uint64_t)(blockIdx.x - T.first_block) * 256u + threadIdx.x;
    if (e >= (uint64_t)a.n_rows * T.width) return;
    const uint32_t r = static_cast<uint32_t>(e / T.width), j = static_cast<uint32_t>(e - (uint64_t)r * T.width);
    const int64_t src = a.index[r];
    T.out[e] = T.in[(size_t)src * T.width + j];
}

// 30-bit Morton key of every mean (10 bits per axis over the bounding box lo..hi, x most significant), value = index
__device__ __forceinline__ uint32_t spread_bits_10(uint32_t v) {
    v = (v | (v << 16)) & 0x030000FFu;
    v = (v | (v << 8)) & 0x0300F00Fu;
    v = (v | (v << 4)) & 0x030C30C3u;
    v = (v | (v << 2)) & 0x09249249u;
    return v;
}
__global__ void __launch_bounds__(256) morton_keys_kernel(const float* __restrict__ means, const float* __restrict__ lo, const float* __restrict__ hi,
                                                          uint32_t* __restrict__ keys, uint32_t* __restrict__ vals, const uint32_t n) {
    const uint32_t i = blockIdx.x * 256u + threadIdx.x;
    if (i >= n) return;
    uint32_t q[3];
#pragma unroll
    for (int c = 0; c < 3; ++c) {
        const float span = fmaxf(hi[c] - lo[c], 1e-12f);
        const float t = (means[3 * (size_t)i + c] - lo[c]) / span * 1023.0f;
        const int v = static_cast<int>(t);                       // truncation, as torch's .to(int64)
        q[c] = static_cast<uint32_t>(min(max(v, 0), 1023));
    }
    keys[i] = (spread_bits_10(q[0]) << 2) | (spread_bits_10(q[1]) << 1) | spread_bits_10(q[2]);
    vals[i] = i;
}
__global__ void __launch_bounds__(256) widen_indices_kernel(const uint32_t* __restrict__ in, int64_t* __restrict__ out, const uint32_t n) {
    const uint32_t i = blockIdx.x * 256u + threadIdx.x;
    if (i < n) out[i] = static_cast<int64_t>(in[i]);
}

size_t adc_scan_temp_bytes(uint32_t n) { return (size_t)((n + kAdcBlock - 1u) / kAdcBlock) * sizeof(uint4); }      // the block sums

hipError_t launch_adc_plan(const AdcPlanArgs& a, hipStream_t s) {
    if (a.n == 0) return hipMemsetAsync(a.totals, 0, 4 * sizeof(uint32_t), s);
    const uint32_t n_blocks = (a.n + kAdcBlock - 1u) / kAdcBlock;
    uint4* const block_sums = static_cast<uint4*>(a.scan_temp);
    hipLaunchKernelGGL(adc_classify_kernel, dim3((a.n + 255u) / 256u), dim3(256), 0, s, a);
    hipLaunchKernelGGL(adc_block_sums_kernel, dim3(n_blocks), dim3(256), 0, s, a.plan, block_sums, a.n);
    hipLaunchKernelGGL(adc_scan_blocks_kernel, dim3(1), dim3(kTileScanThreads), 0, s, block_sums, a.totals, n_blocks);
    hipLaunchKernelGGL(adc_offsets_kernel, dim3(n_blocks), dim3(256), 0, s, a.plan, block_sums, a.offsets, a.n);
    return hipGetLastError();
}

hipError_t launch_adc_scatter(int kind, const AdcScatterArgs& a, hipStream_t s) {
    const uint64_t total = (uint64_t)a.n * a.width;
    if (total == 0) return hipSuccess;
    const dim3 grid(static_cast<unsigned>((total + 255u) / 256u)), block(256);
    if (kind == 1) hipLaunchKernelGGL(adc_scatter_kernel<1>, grid, block, 0, s, a);
    else if (kind == 2) hipLaunchKernelGGL(adc_scatter_kernel<2>, grid, block, 0, s, a);
    else hipLaunchKernelGGL(adc_scatter_kernel<0>, grid, block, 0, s, a);
    return hipGetLastError();
}

hipError_t launch_gather_rows(const GatherArgs& a_in, hipStream_t s) {
    GatherArgs a = a_in;
    uint32_t blocks = 0;
    for (int k = 0; k < a.n_tensors; ++k) { a.t[k].first_block = blocks; blocks += static_cast<uint32_t>(((uint64_t)a.n_rows * a.t[k].width + 255u) / 256u); }
    if (blocks == 0) return hipSuccess;
    hipLaunchKernelGGL(gather_rows_kernel, dim3(blocks), dim3(256), 0, s, a);
    return hipGetLastError();
}

size_t morton_temp_bytes(uint32_t n) { return 4 * (size_t)((n + 63u) / 64u * 64u) * sizeof(uint32_t) + own_sort_temp_bytes(n, 30) + 256; }

hipError_t run_morton_order(const float* means, const float* lo, const float* hi, int64_t* order_out, uint32_t n, void* temp, size_t temp_bytes, hipStream_t s) {
    if (n == 0) return hipSuccess;
    const size_t stride = (size_t)((n + 63u) / 64u * 64u);
    uint32_t* base = static_cast<uint32_t*>(temp);
    uint32_t* keys[2] = {base, base + stride};
    uint32_t* vals[2] = {base + 2 * stride, base + 3 * stride};
    char* sort_temp = reinterpret_cast<char*>(base + 4 * stride);
    const size_t sort_bytes = temp_bytes - 4 * stride * sizeof(uint32_t);
    hipLaunchKernelGGL(morton_keys_kernel, dim3((n + 255u) / 256u), dim3(256), 0, s, means, lo, hi, keys[0], vals[0], n);
    int sel = 0;
    const hipError_t e = own_sort_pairs_u32(sort_temp, sort_bytes, keys, vals, sel, n, 30, s);
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL(widen_indices_kernel, dim3((n + 255u) / 256u), dim3(256), 0, s, vals[sel], order_out, n);
    return hipGetLastError();
}

}  // namespace fgs
