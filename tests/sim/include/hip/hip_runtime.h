// TEST-ONLY stand-in for <hip/hip_runtime.h>: lets g++ compile the product's .hip sources into a CPU library in which
// every workgroup runs as cooperatively scheduled fibers (one per work-item). Purpose: exercise the real kernel
// sources + the C-ABI host orchestration against the oracle in the GPU-less build container (tests/test_sim_parity.py).
// It is never built by, shipped with, or loaded from the product package; see tests/sim/README.md.
#pragma once
#include <algorithm>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <functional>
#include <vector>

#define __global__
#define __device__
#define __host__
#define __forceinline__ inline
#define __launch_bounds__(...)
#define __shared__ static

typedef int hipError_t;
enum { hipSuccess = 0, hipErrorInvalidValue = 1, hipMemcpyDeviceToHost = 2, hipHostMallocDefault = 0 };
typedef void* hipStream_t;

struct uint2 { unsigned x, y; };
struct uint4 { unsigned x, y, z, w; };
struct float2 { float x, y; };
struct float4 { float x, y, z, w; };
struct dim3 { unsigned x, y, z; dim3(unsigned x_ = 1, unsigned y_ = 1, unsigned z_ = 1) : x(x_), y(y_), z(z_) {} };
inline float4 make_float4(float x, float y, float z, float w) { return float4{x, y, z, w}; }
inline uint2 make_uint2(unsigned x, unsigned y) { return uint2{x, y}; }
inline float2 make_float2(float x, float y) { return float2{x, y}; }
inline uint4 make_uint4(unsigned x, unsigned y, unsigned z, unsigned w) { return uint4{x, y, z, w}; }

// Fiber switch: the six callee-saved registers and the stack pointer (System V x86-64), nothing else -- glibc's swapcontext / getcontext make a
// signal-mask system call per switch, and a forward pass of 200 Gaussians runs ~0.5 M fibers through here (the fixed-size grids of the sort's
// row scans alone are 3 x 512 workgroups of 256 work-items): 1.6 s per pass with ucontext. A file-local symbol per translation unit.
#if !defined(__x86_64__)
#error "the CPU simulation's fiber switch is written for x86-64"
#endif
asm(R"(
    .text
    .type fgs_sim_switch_local,@function
fgs_sim_switch_local:
    pushq %rbp
    pushq %rbx
    pushq %r12
    pushq %r13
    pushq %r14
    pushq %r15
    movq %rsp, (%rdi)
    movq (%rsi), %rsp
    popq %r15
    popq %r14
    popq %r13
    popq %r12
    popq %rbx
    popq %rbp
    ret
    .size fgs_sim_switch_local, .-fgs_sim_switch_local
)");
extern "C" void fgs_sim_switch(void** save_sp, void** load_sp) asm("fgs_sim_switch_local");

namespace sim {

constexpr int kSlots = 24;
struct Lane {
    void* sp = nullptr;          // saved stack pointer while the fiber is not running
    bool alive = false;
    unsigned gen_wave = 0, gen_block = 0;
    uint64_t slot[2][kSlots];      // exchange slots of WAVE-scope collectives (parity = wave generation)
    int block_slot[2];             // of __syncthreads_and (parity = block generation): its own storage -- a lane that has passed a workgroup barrier may
                                   // enter a wave collective (or the other way round) before a slower lane has read what it published for the first
};
struct Block {
    std::vector<Lane> lanes;
    std::vector<char> stacks;
    int n = 0, cur = 0;
    unsigned block_idx = 0, block_idy = 0, block_idz = 0, block_dim = 0, grid_dim = 0, grid_dimy = 1, grid_dimz = 1;
    void* main_sp = nullptr;
    std::function<void()> body;
    unsigned long progress = 0;
};
inline Block& blk() { static Block b; return b; }
inline Lane& me() { return blk().lanes[blk().cur]; }
inline unsigned tid() { return static_cast<unsigned>(blk().cur); }
inline void yield() { Block& b = blk(); fgs_sim_switch(&b.lanes[b.cur].sp, &b.main_sp); }

inline void trampoline() {
    Block& b = blk();
    b.body();
    b.lanes[b.cur].alive = false;
    ++b.progress;
    fgs_sim_switch(&b.lanes[b.cur].sp, &b.main_sp);      // never resumed
    abort();
}

inline void run_block(unsigned block_idx, unsigned block_dim, unsigned grid_dim, const std::function<void()>& body) {
    Block& b = blk();
    constexpr size_t kStack = 128 * 1024;
    if (b.lanes.size() < block_dim) { b.lanes.resize(block_dim); b.stacks.resize(kStack * block_dim); }
    b.n = static_cast<int>(block_dim); b.block_idx = block_idx; b.block_dim = block_dim; b.grid_dim = grid_dim; b.body = body;
    for (int i = 0; i < b.n; ++i) {
        Lane& L = b.lanes[i];
        L.alive = true; L.gen_wave = 0; L.gen_block = 0;
        // a fresh fiber: six zeroed callee-saved registers and the entry point as the return address of the first switch; the entry point
        // must see the stack as after a CALL (rsp = 8 mod 16), and a zero return address ends any unwinder's walk
        uintptr_t top = reinterpret_cast<uintptr_t>(b.stacks.data() + kStack * (i + 1));
        top &= ~static_cast<uintptr_t>(15);
        void** sp = reinterpret_cast<void**>(top);
        *--sp = nullptr;                                            // fake return address of trampoline: rsp = 8 mod 16 at its first instruction
        *--sp = reinterpret_cast<void*>(&trampoline);
        for (int r = 0; r < 6; ++r) *--sp = nullptr;
        L.sp = sp;
    }
    int alive = b.n;
    unsigned long stale_rounds = 0;
    while (alive > 0) {
        const unsigned long before = b.progress;
        alive = 0;
        for (int i = 0; i < b.n; ++i) {
            if (!b.lanes[i].alive) continue;
            b.cur = i;
            fgs_sim_switch(&b.main_sp, &b.lanes[i].sp);
            if (b.lanes[i].alive) ++alive;
        }
        if (b.progress == before && alive > 0) {
            if (++stale_rounds > 4) {
                fprintf(stderr, "[sim] deadlock: a collective is waiting for lanes that never arrive (block %u); per wave: alive lanes, min..max wave / block generation\n", block_idx);
                for (int w0 = 0; w0 < b.n; w0 += 64) {
                    unsigned lo_w = ~0u, hi_w = 0, lo_b = ~0u, hi_b = 0; int live = 0;
                    for (int i = w0; i < std::min(w0 + 64, b.n); ++i) {
                        if (!b.lanes[i].alive) continue;
                        ++live; lo_w = std::min(lo_w, b.lanes[i].gen_wave); hi_w = std::max(hi_w, b.lanes[i].gen_wave);
                        lo_b = std::min(lo_b, b.lanes[i].gen_block); hi_b = std::max(hi_b, b.lanes[i].gen_block);
                    }
                    fprintf(stderr, "[sim]   wave %d: %d alive, wave generation %u..%u, block generation %u..%u\n", w0 / 64, live, lo_w, hi_w, lo_b, hi_b);
                }
                abort();
            }
        } else stale_rounds = 0;
    }
}

// Wait until every ALIVE lane of the scope (my wave, or the whole block) reached my generation; returns that generation.
// A lane took part in collective `g` iff its generation counter is >= g (it may have exited since -- its slots stay valid).
inline unsigned sync_scope(bool wave) {
    Block& b = blk();
    Lane& L = me();
    const unsigned my_gen = wave ? ++L.gen_wave : ++L.gen_block;
    ++b.progress;
    const int first = wave ? (b.cur / 64) * 64 : 0;
    const int last = wave ? std::min(first + 64, b.n) : b.n;
    for (;;) {
        bool all = true;
        for (int i = first; i < last; ++i) {
            const Lane& o = b.lanes[i];
            if (o.alive && (wave ? o.gen_wave : o.gen_block) < my_gen) { all = false; break; }
        }
        if (all) break;
        yield();
    }
    return my_gen;
}
inline int next_parity(bool wave) { Lane& L = me(); return static_cast<int>(((wave ? L.gen_wave : L.gen_block) + 1u) & 1u); }

struct Idx3 { unsigned x, y, z; };
template <class F> inline void launch(dim3 grid, dim3 block, F f) {
    if (getenv("FGS_SIM_TRACE")) fprintf(stderr, "[sim] launch grid=%ux%ux%u block=%u\n", grid.x, grid.y, grid.z, block.x);
    for (unsigned bz = 0; bz < grid.z; ++bz)
        for (unsigned by = 0; by < grid.y; ++by)
            for (unsigned bi = 0; bi < grid.x; ++bi) {
                blk().block_idy = by; blk().block_idz = bz; blk().grid_dimy = grid.y; blk().grid_dimz = grid.z;
                run_block(bi, block.x, grid.x, f);
            }
}

}  // namespace sim

#define threadIdx (sim::Idx3{sim::tid(), 0u, 0u})
#define blockIdx (sim::Idx3{sim::blk().block_idx, sim::blk().block_idy, sim::blk().block_idz})
#define blockDim (sim::Idx3{sim::blk().block_dim, 1u, 1u})
#define gridDim (sim::Idx3{sim::blk().grid_dim, sim::blk().grid_dimy, sim::blk().grid_dimz})

template <class K, class... A>
inline void hipLaunchKernelGGL(K kernel, dim3 grid, dim3 block, size_t, hipStream_t, A... args) {
    sim::launch(grid, block, [=]() { kernel(args...); });
}

inline void __syncthreads() { sim::sync_scope(false); }
inline void __threadfence() {}
inline int __syncthreads_and(int p) {
    const int par = sim::next_parity(false);
    sim::me().block_slot[par] = p ? 1 : 0;
    const unsigned g = sim::sync_scope(false);
    sim::Block& b = sim::blk();
    int r = 1;
    for (int i = 0; i < b.n; ++i) if (b.lanes[i].gen_block >= g && !b.lanes[i].block_slot[par]) r = 0;
    return r;
}

using std::min;
using std::max;
inline float __expf(float x) { return expf(x); }
inline float __frcp_rn(float x) { return 1.0f / x; }
inline unsigned __float_as_uint(float f) { unsigned u; memcpy(&u, &f, 4); return u; }
inline float __uint_as_float(unsigned u) { float f; memcpy(&f, &u, 4); return f; }
inline int __float_as_int(float f) { int u; memcpy(&u, &f, 4); return u; }
inline float __int_as_float(int u) { float f; memcpy(&f, &u, 4); return f; }
inline int __popc(unsigned v) { return __builtin_popcount(v); }
inline int __clzll(unsigned long long v) { return v ? __builtin_clzll(v) : 64; }
inline int __popcll(unsigned long long v) { return __builtin_popcountll(v); }
inline int __ffsll(unsigned long long v) { return __builtin_ffsll(static_cast<long long>(v)); }
inline int __clz(int v) { return v ? __builtin_clz(static_cast<unsigned>(v)) : 32; }
inline unsigned __brev(unsigned v) { unsigned r = 0; for (int i = 0; i < 32; ++i) r |= ((v >> i) & 1u) << (31 - i); return r; }
inline unsigned atomicAdd(unsigned* p, unsigned v) { const unsigned o = *p; *p = o + v; return o; }
inline unsigned atomicOr(unsigned* p, unsigned v) { const unsigned o = *p; *p = o | v; return o; }
inline unsigned long long atomicAdd(unsigned long long* p, unsigned long long v) { const unsigned long long o = *p; *p = o + v; return o; }
inline float atomicAdd(float* p, float v) { const float o = *p; *p = o + v; return o; }
inline float unsafeAtomicAdd(float* p, float v) { const float o = *p; *p = o + v; return o; }

typedef struct SimEvent_* hipEvent_t;
inline hipError_t hipEventCreate(hipEvent_t* e) { *e = reinterpret_cast<hipEvent_t>(malloc(1)); return hipSuccess; }
enum { hipEventDisableTiming = 2 };
inline hipError_t hipEventCreateWithFlags(hipEvent_t* e, unsigned) { return hipEventCreate(e); }
inline hipError_t hipEventRecord(hipEvent_t, hipStream_t) { return hipSuccess; }
inline hipError_t hipEventSynchronize(hipEvent_t) { return hipSuccess; }
inline hipError_t hipEventElapsedTime(float* ms, hipEvent_t, hipEvent_t) { *ms = 0.0f; return hipSuccess; }
inline hipError_t hipGetLastError() { return hipSuccess; }
inline hipError_t hipGetDevice(int* d) { *d = 0; return hipSuccess; }
inline const char* hipGetErrorString(hipError_t) { return "sim"; }
inline hipError_t hipMemsetAsync(void* p, int v, size_t n, hipStream_t) { if (n) memset(p, v, n); return hipSuccess; }
inline hipError_t hipMemcpyAsync(void* d, const void* s, size_t n, int, hipStream_t) { memcpy(d, s, n); return hipSuccess; }
inline hipError_t hipStreamSynchronize(hipStream_t) { return hipSuccess; }
inline hipError_t hipHostMalloc(void** p, size_t n, unsigned) { *p = malloc(n); return *p ? hipSuccess : 1; }
