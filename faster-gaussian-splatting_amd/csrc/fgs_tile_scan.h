// Prefix sums over the per-tile arrays (12 240 tiles at 1080p) by ONE 1024-thread workgroup -- the two planning passes of the pipeline
// (binning.hip: bucket scan after K7; blend_backward.hip: plan_blend_backward_kernel before K11).
//
// Why a workgroup and not a device-wide scan: at this size the library scan (rocPRIM look-back, two launches) costs 15 us and two earlier
// own versions -- one barrier per 1024-tile chunk; sixteen strided wave scans through ds_bpermute -- 15-19 us: all latency. Here a thread
// owns kTileScanPerThread CONSECUTIVE tiles (16-byte loads, all issued at once), sums them serially, ONE DPP wave scan (6 instructions)
// ranks the threads, and the sixteen wave totals meet behind a single barrier.
#pragma once
#include <fgs_wave.h>

namespace fgs {

constexpr int kTileScanThreads = 1024;
constexpr int kTileScanPerThread = 16;                            // 16 Ki tiles per pass
constexpr int kTileScanWaves = kTileScanThreads / kWave;          // 16

struct TileScanShared { uint32_t wave_total[2][kTileScanWaves]; };   // double-buffered: one barrier per pass

// v[k] = value of tile first + tid * kTileScanPerThread + k (0 beyond the end). On return ex[k] = `base` + the sum of the values of all
// tiles in front of that tile; returns the sum of all values of the pass. Every thread of the 1024-thread workgroup must call it;
// `parity` alternates between consecutive passes of a kernel.
__device__ __forceinline__ uint32_t tile_scan_pass(const uint32_t (&v)[kTileScanPerThread], uint32_t (&ex)[kTileScanPerThread], TileScanShared& s,
                                                   const uint32_t base, const int parity) {
    const uint32_t tid = threadIdx.x, lane = tid & 63u, wv = tid >> 6;
    uint32_t run = 0;
#pragma unroll
    for (int k = 0; k < kTileScanPerThread; ++k) { ex[k] = run; run += v[k]; }
    const uint32_t incl = wave_inclusive_sum(run);
    if (lane == 63u) s.wave_total[parity][wv] = incl;
    __syncthreads();
    uint32_t before = base + incl - run, total = 0;
#pragma unroll
    for (uint32_t w = 0; w < kTileScanWaves; ++w) {               // 16 broadcast reads
        const uint32_t t = s.wave_total[parity][w];
        before += w < wv ? t : 0u;
        total += t;
    }
#pragma unroll
    for (int k = 0; k < kTileScanPerThread; ++k) ex[k] += before;
    return total;
}

}  // namespace fgs
