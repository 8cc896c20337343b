// Device self-test of the wave64 primitives in fgs_wave.h (DPP wave_shr/wave_rol semantics, ballot prefix, readlane).
// Exposed through fgs_debug_wave_selftest(); the GPU test-suite runs it first so a wrong lane-shift direction is
// reported as such and not as a gradient mismatch.
#include "fgs_kernels.h"
#include <fgs_wave.h>

namespace fgs {

__global__ void __launch_bounds__(kWave) wave_selftest_kernel(uint32_t* out) {
    const unsigned lane = lane_id();
    float state[1] = {static_cast<float>(lane + 100u)};
    float feed[1] = {static_cast<float>(lane + 1000u)};
    pipeline_advance<1>(state, feed);
    out[lane] = static_cast<uint32_t>(state[0]);             // expect lane 0: 1000, lane l: 99 + l
    out[64 + lane] = static_cast<uint32_t>(feed[0]);         // expect 1000 + (l + 1) % 64
    const uint64_t m = wave_ballot(lane % 3u == 0u);
    out[128 + lane] = lanes_below(m);                        // expect ceil(l / 3)
    const int src = __ffsll(static_cast<unsigned long long>(m >> 7)) - 1 + 7;   // first multiple of 3 at or above 7 -> 9
    const unsigned shifted = static_cast<unsigned>(wave_shift_up1(static_cast<float>(lane + 10u)));   // lane 0: 10, lane l: 9 + l
    const unsigned shift_ok = shifted == (lane == 0 ? 10u : 9u + lane) ? 0u : 1000000u;
    const float red = wave_sum_to_lane63(static_cast<float>(lane) * 0.5f + 1.0f);                   // 0.5 * 2016 + 64 = 1072
    const unsigned red_ok = (lane != 63u || red == 1072.0f) ? 0u : 4000000u;
    const unsigned scan_ok = wave_exclusive_sum(lane + 1u) == lane * (lane + 1u) / 2u ? 0u : 2000000u;
    // round 4: the fp32 matrix instruction's operand / result layout (A[i][k] = i + 1; B[k][j] = k + 1 in column 3 only -> D[i][3] = 10 (i + 1): rows and
    // columns cannot be confused), the scalar-mask select and the lane shuffle
    fgs_acc4 d = {0.0f, 0.0f, 0.0f, 0.0f};
    wave_mfma_16x16x4(static_cast<float>((lane & 15u) + 1u), (lane & 15u) == 3u ? static_cast<float>((lane >> 4) + 1u) : 0.0f, d);
    bool mfma_good = true;
    for (int r = 0; r < 4; ++r) mfma_good = mfma_good && d[r] == ((lane & 15u) == 3u ? 10.0f * static_cast<float>(4u * (lane >> 4) + r + 1u) : 0.0f);
    const unsigned mfma_ok = wave_ballot(!mfma_good) == 0ull ? 0u : 8000000u;
    const unsigned sel_ok = lane_select(0x00000000ffff0000ull, 1.0f, 2.0f) == ((lane >= 16u && lane < 32u) ? 2.0f : 1.0f) ? 0u : 16000000u;
    const unsigned shf_ok = wave_shuffle(lane * 3u, 63u - lane) == (63u - lane) * 3u ? 0u : 32000000u;
    out[192 + lane] = wave_read(lane * 7u, src) + wave_sum(lane) + wave_max(lane ^ 5u) + shift_ok + scan_ok + red_ok + mfma_ok + sel_ok + shf_ok;   // 63 + 2016 + 63
}

hipError_t launch_wave_selftest(uint32_t* out, hipStream_t s) {
    hipLaunchKernelGGL(wave_selftest_kernel, dim3(1), dim3(kWave), 0, s, out);
    return hipGetLastError();
}

}  // namespace fgs
