// Micro-benchmark (tools, not product): issue cost of the VALU instruction classes the blend kernels are made of, on gfx950.
// Each wave runs REPS x 32 independent copies of one instruction (inline asm, so the compiler cannot fold / pack / hoist
// them) and reads the shader clock (s_memtime) around the loop; with W waves per SIMD the cost per wave-instruction is
// cycles / (W x instructions). Prints cycles per wave64 instruction for W = 1, 2, 4, 8.
//   hipcc --offload-arch=gfx950 -O2 tools/valu_rate.hip -o /tmp/valu_rate && /tmp/valu_rate
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define REP8(X) X X X X X X X X
#define REP32(X) REP8(X) REP8(X) REP8(X) REP8(X)

template <int KIND>
__global__ void __launch_bounds__(256) rate_kernel(unsigned long long* cycles, float* sink, int reps) {
    float a0 = threadIdx.x * 1e-3f, a1 = a0 + 1.0f, a2 = a0 + 2.0f, a3 = a0 + 3.0f, b = 1.0001f, c = 0.5f;
    float p0 = a0, p1 = a1, p2 = a2, p3 = a3;     // second halves of packed pairs live in adjacent registers via constraints
    unsigned u = threadIdx.x * 2654435761u;
    const unsigned long long t0 = __builtin_readcyclecounter();
    for (int r = 0; r < reps; ++r) {
        if (KIND == 0) { REP8(asm volatile("v_fma_f32 %0, %0, %4, %5\n v_fma_f32 %1, %1, %4, %5\n v_fma_f32 %2, %2, %4, %5\n v_fma_f32 %3, %3, %4, %5" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(b), "v"(c));) }
        if (KIND == 1) { REP8(asm volatile("v_mul_f32 %0, %0, %4\n v_mul_f32 %1, %1, %4\n v_mul_f32 %2, %2, %4\n v_mul_f32 %3, %3, %4" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(b));) }
        if (KIND == 2) { REP8(asm volatile("v_exp_f32 %0, %0\n v_exp_f32 %1, %1\n v_exp_f32 %2, %2\n v_exp_f32 %3, %3" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3));) }
        if (KIND == 3) { REP8(asm volatile("v_rcp_f32 %0, %0\n v_rcp_f32 %1, %1\n v_rcp_f32 %2, %2\n v_rcp_f32 %3, %3" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3));) }
        if (KIND == 4) { REP8(asm volatile("v_cndmask_b32 %0, %0, %4, vcc\n v_cndmask_b32 %1, %1, %4, vcc\n v_cndmask_b32 %2, %2, %4, vcc\n v_cndmask_b32 %3, %3, %4, vcc" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(b) : "vcc");) }
        if (KIND == 5) { REP8(asm volatile("v_mov_b32_dpp %0, %0 wave_shr:1 row_mask:0xf bank_mask:0xf\n v_mov_b32_dpp %1, %1 wave_shr:1 row_mask:0xf bank_mask:0xf\n v_mov_b32_dpp %2, %2 wave_shr:1 row_mask:0xf bank_mask:0xf\n v_mov_b32_dpp %3, %3 wave_shr:1 row_mask:0xf bank_mask:0xf" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3));) }
        if (KIND == 6) { REP8(asm volatile("v_cvt_f32_ubyte0 %0, %4\n v_cvt_f32_ubyte1 %1, %4\n v_cvt_f32_ubyte2 %2, %4\n v_cvt_f32_ubyte0 %3, %4" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(u));) }
        if (KIND == 7) {   // packed: 2 floats per lane per instruction
            typedef float f2 __attribute__((ext_vector_type(2)));
            f2 q0 = {a0, p0}, q1 = {a1, p1}, q2 = {a2, p2}, q3 = {a3, p3}, bb = {b, b}, cc = {c, c};
            REP8(asm volatile("v_pk_fma_f32 %0, %0, %4, %5\n v_pk_fma_f32 %1, %1, %4, %5\n v_pk_fma_f32 %2, %2, %4, %5\n v_pk_fma_f32 %3, %3, %4, %5" : "+v"(q0), "+v"(q1), "+v"(q2), "+v"(q3) : "v"(bb), "v"(cc));)
            a0 = q0.x + q0.y; a1 = q1.x + q1.y; a2 = q2.x + q2.y; a3 = q3.x + q3.y;
        }
        if (KIND == 8) { REP8(asm volatile("v_add_f32 %0, %0, %4\n v_add_f32 %1, %1, %4\n v_add_f32 %2, %2, %4\n v_add_f32 %3, %3, %4" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(b));) }
        if (KIND == 10) { REP8(asm volatile("v_cndmask_b32_e64 %0, %0, %4, s[20:21]\n v_cndmask_b32_e64 %1, %1, %4, s[20:21]\n v_cndmask_b32_e64 %2, %2, %4, s[20:21]\n v_cndmask_b32_e64 %3, %3, %4, s[20:21]" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(b) : "s20", "s21");) }
        if (KIND == 11) { REP8(asm volatile("v_cmp_lt_f32 vcc, %0, %4\n v_cmp_lt_f32 vcc, %1, %4\n v_cmp_lt_f32 vcc, %2, %4\n v_cmp_lt_f32 vcc, %3, %4" : : "v"(a0), "v"(a1), "v"(a2), "v"(a3), "v"(b) : "vcc");) }
        if (KIND == 12) { REP8(asm volatile("v_add_f32_dpp %0, %0, %4 wave_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:0\n v_add_f32_dpp %1, %1, %4 wave_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:0\n v_add_f32_dpp %2, %2, %4 wave_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:0\n v_add_f32_dpp %3, %3, %4 wave_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:0" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(b));) }
        if (KIND == 13) { REP8(asm volatile("v_mov_b32 %0, %4\n v_mov_b32 %1, %4\n v_mov_b32 %2, %4\n v_mov_b32 %3, %4" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(b));) }
        if (KIND == 14) { REP8(asm volatile("v_med3_i32 %0, %0, %4, %5\n v_med3_i32 %1, %1, %4, %5\n v_med3_i32 %2, %2, %4, %5\n v_med3_i32 %3, %3, %4, %5" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(b), "v"(c));) }
        if (KIND == 15) { REP8(asm volatile("v_max_f32 %0, %0, %4\n v_min_f32 %1, %1, %4\n v_max_f32 %2, %2, %4\n v_min_f32 %3, %3, %4" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(b));) }
        if (KIND == 16) { REP8(asm volatile("v_lshlrev_b32 %0, 4, %0\n v_and_b32 %1, 255, %1\n v_add_u32 %2, %2, %4\n v_sub_u32 %3, %3, %4" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(u));) }
        if (KIND == 17) {   // compare + exec-masked block of 2 FMAs + restore (what `if (lane-varying) {..}` costs around a short body)
            REP8(asm volatile("v_cmp_lt_f32 vcc, %0, %4\n s_and_saveexec_b64 s[20:21], vcc\n v_fma_f32 %1, %1, %4, %5\n v_fma_f32 %2, %2, %4, %5\n s_or_b64 exec, exec, s[20:21]" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(b), "v"(c) : "vcc", "s20", "s21");) }
        if (KIND == 18) { REP8(asm volatile("v_fmac_f32 %0, %4, %5\n v_fmac_f32 %1, %4, %5\n v_fmac_f32 %2, %4, %5\n v_fmac_f32 %3, %4, %5" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(b), "v"(c));) }
        if (KIND == 19) { REP8(asm volatile("v_cmp_lt_f32 vcc, %0, %4\n v_cndmask_b32 %1, %1, %4, vcc\n v_cmp_lt_f32 s[20:21], %2, %4\n v_cndmask_b32_e64 %3, %3, %4, s[20:21]" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(b) : "vcc", "s20", "s21");) }
        if (KIND == 20) { REP8(asm volatile("v_readfirstlane_b32 s20, %0\n v_readfirstlane_b32 s21, %1\n v_readfirstlane_b32 s22, %2\n v_readfirstlane_b32 s23, %3" : : "v"(a0), "v"(a1), "v"(a2), "v"(a3) : "s20", "s21", "s22", "s23");) }
        if (KIND == 9) { REP8(asm volatile("v_fma_f32 %0, %0, %4, %5\n v_fma_f32 %0, %0, %4, %5\n v_fma_f32 %0, %0, %4, %5\n v_fma_f32 %0, %0, %4, %5" : "+v"(a0) : "v"(a1), "v"(a2), "v"(a3), "v"(b), "v"(c));) }   // dependent chain
    }
    const unsigned long long t1 = __builtin_readcyclecounter();
    if ((threadIdx.x & 63) == 0) cycles[blockIdx.x * 4 + (threadIdx.x >> 6)] = t1 - t0;
    sink[blockIdx.x * 256 + threadIdx.x] = a0 + a1 + a2 + a3 + p0 + p1 + p2 + p3;
}

template <int KIND>
void run(const char* name) {
    const int reps = 2000, per_rep = 32;
    unsigned long long* d_c; float* d_s;
    hipMalloc(&d_c, 256 * 8 * 4 * sizeof(unsigned long long));
    hipMalloc(&d_s, 256 * 8 * 256 * sizeof(float));
    printf("%-28s", name);
    for (int w : {1, 2, 4, 8}) {                       // waves per SIMD: 256-thread blocks put one wave on each of the 4 SIMDs
        const int blocks = 256 * w;
        hipLaunchKernelGGL(rate_kernel<KIND>, dim3(blocks), dim3(256), 0, 0, d_c, d_s, reps);
        hipDeviceSynchronize();
        hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
        hipEventRecord(e0, 0);
        hipLaunchKernelGGL(rate_kernel<KIND>, dim3(blocks), dim3(256), 0, 0, d_c, d_s, reps);
        hipEventRecord(e1, 0); hipEventSynchronize(e1);
        float ms = 0; hipEventElapsedTime(&ms, e0, e1);
        std::vector<unsigned long long> h(blocks * 4);
        hipMemcpy(h.data(), d_c, h.size() * sizeof(unsigned long long), hipMemcpyDeviceToHost);
        double mean = 0; for (auto v : h) mean += (double)v; mean /= h.size();
        // readcyclecounter on gfx9 = s_memtime (a constant 100 MHz-class reference on some parts): also derive from wall time
        const double inst_per_simd = (double)reps * per_rep * w;
        printf("  W=%d: %6.2f cyc(memtime) %6.2f ns*2.4GHz", w, mean / inst_per_simd, ms * 1e6 * 2.4 / inst_per_simd);
    }
    printf("\n");
    hipFree(d_c); hipFree(d_s);
}

int main() {
    printf("cycles per wave64 instruction per SIMD (lower = faster); second figure assumes 2.4 GHz wall clock\n");
    run<0>("v_fma_f32 (4 indep)");
    run<9>("v_fma_f32 (dependent)");
    run<1>("v_mul_f32");
    run<8>("v_add_f32");
    run<7>("v_pk_fma_f32");
    run<2>("v_exp_f32");
    run<3>("v_rcp_f32");
    run<4>("v_cndmask_b32");
    run<5>("v_mov_b32_dpp wave_shr:1");
    run<6>("v_cvt_f32_ubyteN");
    run<10>("v_cndmask_b32_e64 sgpr mask");
    run<19>("v_cmp + v_cndmask pairs");
    run<11>("v_cmp_lt_f32 -> vcc");
    run<12>("v_add_f32_dpp wave_shr:1");
    run<13>("v_mov_b32");
    run<14>("v_med3_i32");
    run<15>("v_max/min_f32");
    run<16>("int lshl/and/add/sub");
    run<18>("v_fmac_f32");
    run<20>("v_readfirstlane_b32");
    run<17>("cmp+saveexec+2fma+restore (5)");
    return 0;
}
