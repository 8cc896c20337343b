// Micro-benchmark (tools): throughput of non-returning float atomics (global_atomic_add_f32, agent scope) from the whole chip onto
// (a) ONE address, (b) the 32 addresses of ONE 128-byte line, (c) K hot lines, (d) K hot addresses in K different lines,
// (e) addresses spread over a large array, (f, round 4) per WAVE INSTRUCTION: 64 consecutive floats, 4 x 16 consecutive, 4 x 9 of 16, or L distinct
// lines with 64 / L lanes each -- does the memory pipeline merge the lanes of one instruction that fall into one line? Answers how K11's per-Gaussian accumulators must be laid out when a few
// "hot" Gaussians receive thousands of adds.  hipcc --offload-arch=gfx950 -O2 tools/atomic_rate.hip -o /tmp/atomic_rate
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

// every lane issues `per_lane` atomics; target index = f(global lane id, j)
template <int MODE>
__global__ void __launch_bounds__(256) atomic_kernel(float* buf, unsigned n_targets, int per_lane) {
    const unsigned gid = blockIdx.x * 256u + threadIdx.x;
    unsigned h = gid * 2654435761u;
    for (int j = 0; j < per_lane; ++j) {
        unsigned idx;
        if (MODE == 0) idx = 0;                                              // one address
        else if (MODE == 1) idx = (gid + j) & 31u;                           // one line, 32 addresses
        else if (MODE == 2) idx = ((h >> 8) % n_targets) * 32u + ((gid + j) & 31u);   // n_targets hot lines, all 32 addresses of each
        else if (MODE == 3) idx = ((h >> 8) % n_targets) * 32u;              // n_targets hot addresses, one per line
        else if (MODE == 4) idx = (h >> 4) % n_targets;                      // spread over n_targets floats
        else {
            // per-instruction patterns: every wave draws its own bases from its wave id (uniform over the lanes) and j
            const unsigned wave = gid >> 6, lane = gid & 63u;
            unsigned hw = (wave * 2654435761u) ^ (static_cast<unsigned>(j) * 40503u);
            hw = hw * 1664525u + 1013904223u;
            if (MODE == 5) idx = ((hw >> 4) % (n_targets / 64u)) * 64u + lane;                          // 64 consecutive floats: 2 lines per instruction
            else if (MODE == 6 || MODE == 7) {                                                           // 4 groups of 16 consecutive floats (MODE 7: 9 of the 16 lanes active)
                const unsigned grp = lane >> 4;
                const unsigned hg = (hw + grp * 0x9e3779b9u) * 1664525u + 1013904223u;
                idx = ((hg >> 4) % (n_targets / 16u)) * 16u + (lane & 15u);
                if (MODE == 7 && (lane & 15u) >= 9u) continue;
            } else {                                                                                     // MODE >= 8: L = MODE distinct lines per instruction, 64 / L lanes in each
                const unsigned L = static_cast<unsigned>(MODE);
                const unsigned which = lane % L;
                const unsigned hl = (hw + which * 0x9e3779b9u) * 1664525u + 1013904223u;
                idx = ((hl >> 4) % (n_targets / 32u)) * 32u + (lane / L);
            }
        }
        unsafeAtomicAdd(buf + idx, 1.0f);
        h = h * 1664525u + 1013904223u;
    }
}

template <int MODE>
void run(const char* name, float* buf, unsigned n_targets, int blocks, int per_lane) {
    hipLaunchKernelGGL(atomic_kernel<MODE>, dim3(blocks), dim3(256), 0, 0, buf, n_targets, per_lane);
    (void)hipDeviceSynchronize();
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    (void)hipEventRecord(e0, 0);
    hipLaunchKernelGGL(atomic_kernel<MODE>, dim3(blocks), dim3(256), 0, 0, buf, n_targets, per_lane);
    (void)hipEventRecord(e1, 0); (void)hipEventSynchronize(e1);
    float ms = 0; (void)hipEventElapsedTime(&ms, e0, e1);
    const double total = (double)blocks * 256 * per_lane;
    printf("%-44s targets %8u: %8.3f ms for %.1f M atomics = %8.1f atomics/us\n", name, n_targets, ms, total / 1e6, total / (ms * 1e3));
}

int main() {
    float* buf; (void)hipMalloc(&buf, 256u << 20); (void)hipMemset(buf, 0, 256u << 20);
    const int blocks = 2048, per = 16;      // 8.4 M atomics
    run<0>("one address", buf, 1, blocks, per);
    run<1>("one 128-B line (32 addresses)", buf, 1, blocks, per);
    for (unsigned k : {16u, 256u, 4096u}) run<2>("K hot lines (32 addresses each)", buf, k, blocks, per);
    for (unsigned k : {16u, 256u, 4096u}) run<3>("K hot addresses (one per line)", buf, k, blocks, per);
    for (unsigned k : {1u << 16, 1u << 20, 1u << 24}) run<4>("spread over K floats", buf, k, blocks, per);
    const unsigned big = 1u << 24;
    run<5>("per instruction: 64 consecutive floats", buf, big, blocks, per);
    run<6>("per instruction: 4 x 16 consecutive floats", buf, big, blocks, per);
    run<7>("per instruction: 4 x 9 of 16 (x 9/16 atomics)", buf, big, blocks, per);
    run<8>("per instruction: 8 lines x 8 lanes", buf, big, blocks, per);
    run<16>("per instruction: 16 lines x 4 lanes", buf, big, blocks, per);
    run<32>("per instruction: 32 lines x 2 lanes", buf, big, blocks, per);
    run<64>("per instruction: 64 lines x 1 lane", buf, big, blocks, per);
    return 0;
}
