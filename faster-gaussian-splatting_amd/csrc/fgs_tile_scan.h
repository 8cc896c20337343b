// Prefix sums over the per-tile arrays (12 240 tiles at 1080p) by ONE 1024-thread workgroup -- the two planning passes of the pipeline
// (binning.hip: plan_tiles_kernel after K7; blend_backward.hip: plan_blend_backward_kernel before K11).
//
// Why a workgroup and not a device-wide scan: at this size the library scan (rocPRIM look-back, two launches) costs 15 us and a first own
// version -- one barrier per 1024-tile chunk, twelve in a row -- 15 us as well; both are pure latency. Here a thread takes the tiles
// c * 1024 + tid of all sixteen chunks at once (coalesced loads, all in flight together), the sixteen wave scans are independent
// instruction streams, and the 16 x 16 (chunk, wave) totals are combined by four waves behind ONE barrier pair: three barriers per 16 Ki
// tiles whatever their number.
#pragma once
#include <fgs_wave.h>

namespace fgs {

constexpr int kTileScanThreads = 1024;
constexpr int kTileScanChunks = 16;                               // 16 Ki tiles per pass
constexpr int kTileScanWaves = kTileScanThreads / kWave;          // 16

struct TileScanShared {
    uint32_t cell[kTileScanChunks * kTileScanWaves];              // total of (chunk, wave), then its exclusive prefix
    uint32_t part[4];
};

// v[c] = value of tile first + c * 1024 + tid (0 beyond the end). On return ex[c] = `base` + the sum of the values of all tiles in front of
// that tile; the function returns the sum of all values of the pass. Every thread of the 1024-thread workgroup must call it.
__device__ __forceinline__ uint32_t tile_scan_pass(const uint32_t (&v)[kTileScanChunks], uint32_t (&ex)[kTileScanChunks], TileScanShared& s,
                                                   const uint32_t base) {
    const uint32_t tid = threadIdx.x, lane = tid & 63u, wv = tid >> 6;
#pragma unroll
    for (int c = 0; c < kTileScanChunks; ++c) {
        ex[c] = wave_exclusive_sum(v[c]);
        if (lane == 63u) s.cell[c * kTileScanWaves + wv] = ex[c] + v[c];
    }
    __syncthreads();
    uint32_t x = 0, e = 0;
    if (tid < kTileScanChunks * kTileScanWaves) {                 // 256 cells in tile order: chunk-major, wave-minor
        x = s.cell[tid];
        e = wave_exclusive_sum(x);
        if (lane == 63u) s.part[wv] = e + x;
    }
    __syncthreads();
    const uint32_t p0 = s.part[0], p1 = s.part[1], p2 = s.part[2], p3 = s.part[3];
    if (tid < kTileScanChunks * kTileScanWaves) s.cell[tid] = e + (wv > 0 ? p0 : 0u) + (wv > 1 ? p1 : 0u) + (wv > 2 ? p2 : 0u);
    __syncthreads();
#pragma unroll
    for (int c = 0; c < kTileScanChunks; ++c) ex[c] += base + s.cell[c * kTileScanWaves + wv];
    __syncthreads();                                              // s is rewritten by the next pass
    return p0 + p1 + p2 + p3;
}

}  // namespace fgs
