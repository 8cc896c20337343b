// Compile-time constants of the gfx950 rasterizer.
// Rendering constants restate the reference (rasterization/include/rasterization_config.h:8-60); the work-shape
// constants (wave width, bucket size, block sizes) are chosen for CDNA4 wave64, not translated from warp32.
#pragma once
#include <cstdint>

namespace fgs {

// --- rendering constants (must equal the reference, cfg:10-22) ---
constexpr float kDilation = 0.3f;
constexpr float kDilationProperAA = 0.1f;
constexpr float kMinCov2dDeterminant = 1e-6f;
constexpr float kOneMinusAlphaEps = 1e-6f;
constexpr float kTransmittanceThreshold = 1e-4f;
constexpr float kMinAlphaThresholdRcp = 255.0f;
constexpr float kMinAlphaThreshold = 1.0f / kMinAlphaThresholdRcp;

// --- tiling (cfg:55-60): tile ids / sort keys are only comparable with the reference at 16x12 ---
constexpr int kTileW = 16;
constexpr int kTileH = 12;
constexpr int kTilePixels = kTileW * kTileH;  // 192 = 3 wavefronts
constexpr int kSubtileW = 8;                  // the reference culls per 8x4 sub-tile (kernels_forward.cuh:445-451);
constexpr int kSubtileH = 4;                  // one wave64 covers a 16x4 strip = two such sub-tiles

// --- CDNA4 work shapes ---
constexpr int kWave = 64;
constexpr int kBucket = 64;            // Gaussians per backward bucket = one wavefront (reference: 32 = one warp)
#ifndef FGS_HUGE_FOOTPRINT
#define FGS_HUGE_FOOTPRINT 1024
#endif
constexpr unsigned kHugeFootprint = FGS_HUGE_FOOTPRINT;   // candidate tiles above which a footprint gets its own workgroup (K1; K5: kBigInstanceFootprint)
#ifndef FGS_K5_BIG_FOOTPRINT
#define FGS_K5_BIG_FOOTPRINT 256
#endif
constexpr unsigned kBigInstanceFootprint = FGS_K5_BIG_FOOTPRINT;   // K5: candidate tiles above which a footprint is expanded by one of the launch's leading workgroups instead of by its wave
constexpr int kSeqTiles = 0;           // 0 (default): K1 counts small footprints in flattened (Gaussian, candidate) order (preprocess.hip); n > 0: A/B reference,
// the reference's scheme with n sequential candidates per lane --          // candidate tiles each lane tests itself before the wave cooperates (reference: 4, cfg:54; measured 4: 0.355 ms, 8: 0.300, 12: 0.252, 16: 0.248, 24: 0.256, 32: 0.269 on S2)
// One packed counter atomic per workgroup (see preprocess.hip). Round 3: 256 instead of 512 threads -- a quarter of K1's wave time was spent
// parked at the compaction barrier waiting for slower siblings, and four waves wait less for each other than eight: 0.214 -> 0.199 ms at S2
// (profiles/archive/r03_ab_k1_lookback.txt). The atomics double to 11.7 k per launch at 3 M Gaussians (a same-address atomic retires at ~88 / us:
// 0.13 ms of the counter's time, still below the kernel's); 128 threads would put the counter in front.
#ifndef FGS_PREPROCESS_BLOCK
#define FGS_PREPROCESS_BLOCK 256
#endif
constexpr int kPreprocessBlock = FGS_PREPROCESS_BLOCK;
constexpr int kPreprocessBackwardBlock = 256;
constexpr int kInstanceBlock = 256;            // K5: 128 / 256 / 512 threads measured 0.056 / 0.043 / 0.052 ms (profiles/r05_ab_k5_block.txt)
constexpr int kBlendBlock = kTilePixels;       // 3 waves
constexpr int kBackwardWavesPerBlock = 1;      // 1 bucket per 64-thread workgroup: inactive buckets free their slot at once
// "Hot" Gaussians: footprints above kHotFootprint candidate tiles (0.1 % of the visible ones at S2) are front-most in hundreds to
// thousands of tiles, so K11 adds into their nine accumulators from as many waves -- and a 128-byte line of device memory retires
// only ~1 atomic per ns (tools/atomic_rate.hip). They get kHotReplicas private records each (replica = tile mod kHotReplicas,
// [replica][slot][9], written by the same record flush as everybody else's), folded into their own records after K11.
constexpr unsigned kHotFootprint = 256;
constexpr unsigned kMaxHot = 16384;            // more hot Gaussians than this: the rest accumulate directly (correct, only slower)
constexpr unsigned kHotReplicas = 16;
constexpr unsigned kBackwardMaxBlocks = 1u << 16;   // K11 walks the live-bucket list with at most this many single-wave workgroups
constexpr int kSplatRecordWords = 14;          // sharded path: PrimRec (12 words) + depth key + tile count = FGS_SPLAT_RECORD_BYTES / 4
constexpr int kMaxBatchViews = 8;              // sharded path: views handled by one K1 / K12 launch (grid.y / in-kernel loop)
constexpr int kAccRecordWords = 9;             // sharded path: the 9 pixel-space accumulators = FGS_ACC_RECORD_BYTES / 4
constexpr int kXcds = 8;                       // tile -> workgroup mapping keeps compact pieces of the image on one XCD's L2
// K10's tile plan (binning.hip: plan_tiles_kernel): the tile grid is cut into kPlanBlocksX x kPlanBlocksY rectangular blocks, dealt to the
// XCDs by weight; tile_plan = [block width, block height, tiles per block, blocks per XCD, then the block ids of XCD 0, XCD 1, ...]
constexpr unsigned kPlanBlocksX = 8, kPlanBlocksY = 10, kPlanBlocks = kPlanBlocksX * kPlanBlocksY;
constexpr unsigned kPlanBlocksPerXcd = kPlanBlocks / kXcds;
constexpr unsigned kPlanHeader = 4, kPlanWords = kPlanHeader + kPlanBlocks;

}  // namespace fgs
