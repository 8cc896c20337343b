// Micro-benchmark (MI355X): what does synchronisation BETWEEN workgroups cost on this chip? Decides how the two radix sorts of the
// binning stage should be structured: three launches per pass (now), a persistent kernel with grid barriers, or a
// single-pass sort with a decoupled look-back chain.
//   1. kernel boundary : N trivial kernels back to back on one stream                          -> us per launch
//   2. grid barrier    : G co-resident workgroups, B barriers (agent-scope counter + spin)      -> us per barrier
//   3. flag chain      : workgroup i waits for flag[i-1], then publishes flag[i] (the worst case of a look-back: every hop serial)
//                        -> us per hop; and the same with every workgroup first doing ~10 us of work (hops overlap with work)
// Every spin is bounded (kSpinLimit polls): a bug sets an error word instead of hanging the GPU.
// build + run:  hipcc --offload-arch=gfx950 -O2 tools/sync_cost.hip -o /tmp/sync_cost && /tmp/sync_cost
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { std::printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); return 1; } } while (0)

constexpr unsigned kSpinLimit = 1u << 18;      // ~0.3 s of polling at most

// POLL = 0: every poll is an acquire load (an L2 invalidate per poll on a part with one L2 per XCD); POLL = 1: relaxed polls, ONE acquire
// fence after the wait -- the usual form of a spin-wait
template <int POLL>
__device__ __forceinline__ uint32_t load_agent(const uint32_t* p) {
    return POLL == 0 ? __hip_atomic_load(p, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT) : __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
template <int POLL>
__device__ __forceinline__ void after_wait() { if (POLL == 1) __atomic_thread_fence(__ATOMIC_ACQUIRE); }
__device__ __forceinline__ void store_agent(uint32_t* p, uint32_t v) { __hip_atomic_store(p, v, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT); }

__global__ void trivial_kernel(uint32_t* sink) { if (threadIdx.x == 0 && blockIdx.x == 0xffffffffu) *sink = 1; }

// sense-free counting barrier: barrier b is passed when counter >= (b + 1) * gridDim.x
template <int POLL>
__global__ void __launch_bounds__(256) grid_barrier_kernel(uint32_t* counter, uint32_t* error, const unsigned n_barriers) {
    for (unsigned b = 0; b < n_barriers; ++b) {
        __syncthreads();
        if (threadIdx.x == 0) {
            __hip_atomic_fetch_add(counter, 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
            const uint32_t target = (b + 1u) * gridDim.x;
            unsigned polls = 0;
            while (load_agent<POLL>(counter) < target) { if (++polls > kSpinLimit) { atomicExch(error, 1u); break; } __builtin_amdgcn_s_sleep(1); }
            after_wait<POLL>();
        }
        __syncthreads();
    }
}

// ticket order = scheduling order, as a decoupled look-back needs it (a workgroup only ever waits for tickets handed out before its own)
template <int POLL>
__global__ void __launch_bounds__(256) flag_chain_kernel(uint32_t* ticket, uint32_t* flags, uint32_t* error, const unsigned work_iters, float* sink) {
    __shared__ uint32_t s_id;
    if (threadIdx.x == 0) s_id = __hip_atomic_fetch_add(ticket, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __syncthreads();
    const uint32_t id = s_id;
    float acc = static_cast<float>(threadIdx.x);
    for (unsigned i = 0; i < work_iters; ++i) acc = __builtin_fmaf(acc, 1.0000001f, 0.5f);      // dependent FMAs: ~4 cycles each
    if (acc == 12345.678f) *sink = acc;
    if (threadIdx.x == 0) {
        uint32_t prefix = 0;
        if (id > 0) {
            unsigned polls = 0;
            uint32_t v;
            while ((v = load_agent<POLL>(flags + id - 1)) == 0u) { if (++polls > kSpinLimit) { atomicExch(error, 2u); break; } __builtin_amdgcn_s_sleep(1); }
            after_wait<POLL>();
            prefix = v;
        }
        store_agent(flags + id, prefix + 1u);
    }
}

int main() {
    hipStream_t s;
    CHECK(hipStreamCreate(&s));
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    uint32_t* d;                         // [0] counter / ticket, [1] error, [2] sink, [16..] flags
    const unsigned kMaxGroups = 8192;
    CHECK(hipMalloc(&d, (16 + kMaxGroups) * sizeof(uint32_t)));
    float ms = 0.0f;
    uint32_t h[2];

    // 1. kernel boundary
    for (unsigned grid : {1u, 512u}) {
        const int n = 200;
        for (int rep = 0; rep < 2; ++rep) {
            CHECK(hipEventRecord(e0, s));
            for (int i = 0; i < n; ++i) hipLaunchKernelGGL(trivial_kernel, dim3(grid), dim3(256), 0, s, d + 2);
            CHECK(hipEventRecord(e1, s));
            CHECK(hipEventSynchronize(e1));
            CHECK(hipEventElapsedTime(&ms, e0, e1));
        }
        std::printf("kernel boundary   grid %4u x 256 threads: %6.2f us per launch (200 back to back)\n", grid, ms * 1000.0f / n);
    }

    // 2. grid barrier (co-resident: <= 2 workgroups of 256 threads per CU here)
    for (int poll : {0, 1}) for (unsigned grid : {256u, 512u}) {
        for (unsigned nb : {1u, 33u}) {
            for (int rep = 0; rep < 2; ++rep) {
                CHECK(hipMemsetAsync(d, 0, 16 * sizeof(uint32_t), s));
                CHECK(hipEventRecord(e0, s));
                if (poll == 0) hipLaunchKernelGGL(grid_barrier_kernel<0>, dim3(grid), dim3(256), 0, s, d, d + 1, nb);
                else hipLaunchKernelGGL(grid_barrier_kernel<1>, dim3(grid), dim3(256), 0, s, d, d + 1, nb);
                CHECK(hipEventRecord(e1, s));
                CHECK(hipEventSynchronize(e1));
                CHECK(hipEventElapsedTime(&ms, e0, e1));
            }
            CHECK(hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost));
            std::printf("grid barrier      %s polls, grid %4u, %2u barriers: kernel %7.2f us   (error word %u)\n", poll ? "relaxed" : "acquire", grid, nb, ms * 1000.0f, h[1]);
        }
    }

    // 3. flag chain
    for (int poll : {0, 1}) for (unsigned grid : {512u, 4096u}) {
        for (unsigned work : {0u, 6000u}) {
            for (int rep = 0; rep < 2; ++rep) {
                CHECK(hipMemsetAsync(d, 0, (16 + kMaxGroups) * sizeof(uint32_t), s));
                CHECK(hipEventRecord(e0, s));
                if (poll == 0) hipLaunchKernelGGL(flag_chain_kernel<0>, dim3(grid), dim3(256), 0, s, d, d + 16, d + 1, work, reinterpret_cast<float*>(d + 2));
                else hipLaunchKernelGGL(flag_chain_kernel<1>, dim3(grid), dim3(256), 0, s, d, d + 16, d + 1, work, reinterpret_cast<float*>(d + 2));
                CHECK(hipEventRecord(e1, s));
                CHECK(hipEventSynchronize(e1));
                CHECK(hipEventElapsedTime(&ms, e0, e1));
            }
            std::vector<uint32_t> flags(grid);
            CHECK(hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost));
            CHECK(hipMemcpy(flags.data(), d + 16, grid * sizeof(uint32_t), hipMemcpyDeviceToHost));
            std::printf("flag chain        %s polls, %4u workgroups, %4u dependent FMAs of work each: kernel %8.2f us = %6.3f us per hop   (last flag %u, error word %u)\n",
                        poll ? "relaxed" : "acquire", grid, work, ms * 1000.0f, ms * 1000.0f / grid, flags[grid - 1], h[1]);
        }
    }
    CHECK(hipFree(d));
    return 0;
}
