// Micro-benchmark (tools): throughput of non-returning float atomics (global_atomic_add_f32, agent scope) from the whole chip onto
// (a) ONE address, (b) the 32 addresses of ONE 128-byte line, (c) K hot lines, (d) K hot addresses in K different lines,
// (e) addresses spread over a large array. Answers how K11's per-Gaussian accumulators must be laid out when a few
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
        else idx = (h >> 4) % n_targets;                                     // spread over n_targets floats
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
    return 0;
}
