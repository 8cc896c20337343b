// Does an LDS read cost less pipe time when only one lane is active? 256 CUs x 16 single-wave workgroups; every wave issues N x 16 LDS reads from
// inline assembly (no address arithmetic, no consumer: the LDS pipe is the only thing busy), with EXEC = all lanes or EXEC = lane 0.
// usage: hipcc --offload-arch=gfx950 -O2 tools/lds_exec_rate.hip -o /tmp/lds_exec && /tmp/lds_exec
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f4 __attribute__((ext_vector_type(4)));
typedef float f2 __attribute__((ext_vector_type(2)));
template <int BYTES>
__global__ void __launch_bounds__(64) k(float* out, int n, int one_lane) {
    __shared__ float4 s[64 * 17];
    for (int i = threadIdx.x; i < 64 * 17; i += 64) s[i] = make_float4(i, 1, 2, 3);
    __syncthreads();
    const unsigned addr = static_cast<unsigned>(reinterpret_cast<uintptr_t>(s)) + threadIdx.x * BYTES;   // LDS byte address: lane-consecutive
    float keep = 0.0f;
    if (!one_lane || threadIdx.x == 0) {
        for (int it = 0; it < n; ++it) {
            if (BYTES == 16) {
                f4 v0, v1, v2, v3;
                asm volatile("ds_read_b128 %0, %4\n ds_read_b128 %1, %4 offset:1024\n ds_read_b128 %2, %4 offset:2048\n ds_read_b128 %3, %4 offset:3072\n"
                             "ds_read_b128 %0, %4 offset:4096\n ds_read_b128 %1, %4 offset:5120\n ds_read_b128 %2, %4 offset:6144\n ds_read_b128 %3, %4 offset:7168\n"
                             "ds_read_b128 %0, %4 offset:8192\n ds_read_b128 %1, %4 offset:9216\n ds_read_b128 %2, %4 offset:10240\n ds_read_b128 %3, %4 offset:11264\n"
                             "ds_read_b128 %0, %4 offset:12288\n ds_read_b128 %1, %4 offset:13312\n ds_read_b128 %2, %4 offset:14336\n ds_read_b128 %3, %4 offset:15360\n"
                             "s_waitcnt lgkmcnt(0)" : "=&v"(v0), "=&v"(v1), "=&v"(v2), "=&v"(v3) : "v"(addr) : "memory");
                keep += v0.x;
            } else {
                f2 v0, v1, v2, v3;
                asm volatile("ds_read_b64 %0, %4\n ds_read_b64 %1, %4 offset:512\n ds_read_b64 %2, %4 offset:1024\n ds_read_b64 %3, %4 offset:1536\n"
                             "ds_read_b64 %0, %4 offset:2048\n ds_read_b64 %1, %4 offset:2560\n ds_read_b64 %2, %4 offset:3072\n ds_read_b64 %3, %4 offset:3584\n"
                             "ds_read_b64 %0, %4 offset:4096\n ds_read_b64 %1, %4 offset:4608\n ds_read_b64 %2, %4 offset:5120\n ds_read_b64 %3, %4 offset:5632\n"
                             "ds_read_b64 %0, %4 offset:6144\n ds_read_b64 %1, %4 offset:6656\n ds_read_b64 %2, %4 offset:7168\n ds_read_b64 %3, %4 offset:7680\n"
                             "s_waitcnt lgkmcnt(0)" : "=&v"(v0), "=&v"(v1), "=&v"(v2), "=&v"(v3) : "v"(addr) : "memory");
                keep += v0.x;
            }
        }
    }
    if (keep == 12345.678f) out[threadIdx.x] = keep;
}
int main() {
    float* d; (void)hipMalloc(&d, 4096);
    hipEvent_t a, b; (void)hipEventCreate(&a); (void)hipEventCreate(&b);
    for (int wide = 0; wide < 2; ++wide) for (int one = 0; one < 2; ++one) {
        const int n = 4000, grid = 256 * 16;
        for (int rep = 0; rep < 2; ++rep) {
            (void)hipEventRecord(a);
            if (wide) hipLaunchKernelGGL(k<16>, dim3(grid), dim3(64), 0, 0, d, n, one); else hipLaunchKernelGGL(k<8>, dim3(grid), dim3(64), 0, 0, d, n, one);
            (void)hipEventRecord(b); (void)hipEventSynchronize(b);
        }
        float ms; (void)hipEventElapsedTime(&ms, a, b);
        const double reads_per_cu = 16.0 * n * 16;
        const double cycles = ms * 1e-3 * 2.1e9 / reads_per_cu;
        printf("ds_read_b%-3d %s: %7.3f ms  %5.2f cycles per wave-read and CU at 2.1 GHz  (%5.1f bytes / cycle / CU if all 64 lanes count)\n", wide ? 128 : 64,
               one ? "EXEC = lane 0   " : "EXEC = all lanes", ms, cycles, 64.0 * (wide ? 16 : 8) / cycles);
    }
    return 0;
}
