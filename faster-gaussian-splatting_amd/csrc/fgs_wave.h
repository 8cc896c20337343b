// Wave64 primitives for gfx950 (CDNA4). Everything cross-lane in the kernels goes through these few functions:
// 64-bit ballots, v_readlane broadcasts (scalar, no LDS), DPP wave shifts/rotates for the backward pixel pipeline.
#pragma once
#include <hip/hip_runtime.h>
#include <cstdint>

#if defined(__HIP_DEVICE_COMPILE__) && !defined(__gfx950__)
#error "fgs_wave.h is written for gfx950 (CDNA4, wave64): the DPP scans below use row_bcast:15 / row_bcast:31, which exist on gfx9 / CDNA only"
#endif

namespace fgs {

// PRECONDITION of every scan / reduction / shift below (wave_inclusive_sum, wave_exclusive_sum, wave_sum, wave_max, wave_sum_to_lane63,
// wave_shift_up1*, pipeline_advance, wave_mfma_16x16x4): FULL EXEC -- all 64 lanes active, in particular lane 63, with no gaps. DPP reads of an
// inactive lane return the `old` / zero operand and v_readlane(63) of an inactive lane is undefined. Every call site is convergent code (no call
// under a divergent branch); the on-device self-test (fgs_debug_wave_selftest, tests/test_gpu_parity.py) exercises them with non-trivial data.
__device__ __forceinline__ unsigned lane_id() {
    return __builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, 0u));
}

// (the builtin, not HIP's __ballot: that one materialises the predicate as 0 / 1 in a vector register and compares it again -- two vector
// instructions per ballot in loops that are bound by vector issue)
__device__ __forceinline__ uint64_t wave_ballot(bool p) { return __builtin_amdgcn_ballot_w64(p); }

// number of set bits of m strictly below this lane
__device__ __forceinline__ unsigned lanes_below(uint64_t m) {
    return __builtin_amdgcn_mbcnt_hi(static_cast<unsigned>(m >> 32), __builtin_amdgcn_mbcnt_lo(static_cast<unsigned>(m), 0u));
}

// broadcast from a wave-uniform source lane (v_readlane_b32 -> SGPR)
__device__ __forceinline__ unsigned wave_read(unsigned v, int src_lane) {
    return static_cast<unsigned>(__builtin_amdgcn_readlane(static_cast<int>(v), src_lane));
}
__device__ __forceinline__ float wave_read(float v, int src_lane) {
    return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), src_lane));
}

// Inclusive prefix sum over the wave in 6 DPP-fused adds: a Hillis-Steele scan inside each 16-lane row (row_shr:1/2/4/8, lanes without a
// source read 0), then row_bcast:15 carries a row's total into rows 1 and 3 and row_bcast:31 the total of rows 0-1 into rows 2 and 3 (the
// gfx9 wave-scan idiom). Round 1/2 used __shfl_up (ds_bpermute through the LDS crossbar + a select per step: 18 instructions and six
// LDS round trips); the single-workgroup planning kernels (fgs_tile_scan.h) spent most of their 15-19 us in those.
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ unsigned dpp_read_u(unsigned x) {
    return static_cast<unsigned>(__builtin_amdgcn_update_dpp(0, static_cast<int>(x), CTRL, ROW_MASK, 0xf, true));
}
__device__ __forceinline__ unsigned wave_inclusive_sum(unsigned v) {
    v += dpp_read_u<0x111 /*row_shr:1*/, 0xf>(v);
    v += dpp_read_u<0x112 /*row_shr:2*/, 0xf>(v);
    v += dpp_read_u<0x114 /*row_shr:4*/, 0xf>(v);
    v += dpp_read_u<0x118 /*row_shr:8*/, 0xf>(v);
    v += dpp_read_u<0x142 /*row_bcast:15*/, 0xa>(v);
    v += dpp_read_u<0x143 /*row_bcast:31*/, 0xc>(v);
    return v;
}
__device__ __forceinline__ unsigned wave_exclusive_sum(unsigned v) { return wave_inclusive_sum(v) - v; }
__device__ __forceinline__ unsigned wave_sum(unsigned v) {
    return static_cast<unsigned>(__builtin_amdgcn_readlane(static_cast<int>(wave_inclusive_sum(v)), 63));
}
// maximum over the wave in every lane: the inclusive-scan idiom of wave_inclusive_sum with max (0 fills lanes without a source: the identity of an
// unsigned maximum), then a broadcast of lane 63. (Until round 4: six ds_bpermute round trips through the LDS crossbar.)
__device__ __forceinline__ unsigned wave_max(unsigned v) {
    v = max(v, dpp_read_u<0x111 /*row_shr:1*/, 0xf>(v));
    v = max(v, dpp_read_u<0x112 /*row_shr:2*/, 0xf>(v));
    v = max(v, dpp_read_u<0x114 /*row_shr:4*/, 0xf>(v));
    v = max(v, dpp_read_u<0x118 /*row_shr:8*/, 0xf>(v));
    v = max(v, dpp_read_u<0x142 /*row_bcast:15*/, 0xa>(v));
    v = max(v, dpp_read_u<0x143 /*row_bcast:31*/, 0xc>(v));
    return static_cast<unsigned>(__builtin_amdgcn_readlane(static_cast<int>(v), 63));
}

// Sum over the 64 lanes, result valid in lane 63 only: 4 row_shr scan steps inside each 16-lane row, then row_bcast:15 /
// row_bcast:31 carry the row totals upward (6 DPP-fused adds; the gfx9 wave-reduction idiom).
template <int CTRL, int ROW_MASK, bool ZERO_FILL>
__device__ __forceinline__ float dpp_read(float x) {
    return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(x), CTRL, ROW_MASK, 0xf, ZERO_FILL));
}
__device__ __forceinline__ float wave_sum_to_lane63(float v) {
    v += dpp_read<0x111 /*row_shr:1*/, 0xf, true>(v);
    v += dpp_read<0x112 /*row_shr:2*/, 0xf, true>(v);
    v += dpp_read<0x114 /*row_shr:4*/, 0xf, true>(v);
    v += dpp_read<0x118 /*row_shr:8*/, 0xf, true>(v);
    v += dpp_read<0x142 /*row_bcast:15*/, 0xa, false>(v);
    v += dpp_read<0x143 /*row_bcast:31*/, 0xc, false>(v);
    return v;
}

// lane l takes lane l-1's value, lane 0 keeps its own (DPP wave_shr:1, in place: one instruction, no copy)
__device__ __forceinline__ float wave_shift_up1(float v) {
    return __int_as_float(__builtin_amdgcn_update_dpp(__float_as_int(v), __float_as_int(v), 0x138 /*wave_shr:1*/, 0xf, 0xf, false));
}

// lane l takes lane l-1's value, lane 0 takes 0 (DPP wave_shr:1 bound_ctrl:0). Written so that the DPP combiner can fold the
// move into the consuming VALU instruction (v_add_f32_dpp): shift + inject costs one instruction.
__device__ __forceinline__ float wave_shift_up1_zero(float v) {
    return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x138 /*wave_shr:1*/, 0xf, 0xf, true));
}

// LDS written by some lanes of this wave and read by others: DS operations of one wave execute in order, this only stops
// the compiler from moving the reads above the writes.
__device__ __forceinline__ void wave_lds_fence() {
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    __builtin_amdgcn_wave_barrier();
}

// 16-byte non-temporal global accesses for data streamed exactly once per kernel (Adam state)
typedef float fgs_f32x4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ float4 load_float4_nt(const float* p) {
    const fgs_f32x4 v = __builtin_nontemporal_load(reinterpret_cast<const fgs_f32x4*>(p));
    return make_float4(v.x, v.y, v.z, v.w);
}
__device__ __forceinline__ void store_float4_nt(float* p, const float4 v) {
    fgs_f32x4 t; t.x = v.x; t.y = v.y; t.z = v.z; t.w = v.w;
    __builtin_nontemporal_store(t, reinterpret_cast<fgs_f32x4*>(p));
}

// a load of data another XCD's workgroup wrote before a device-scope fence + atomic (the L2s of the eight XCDs are not coherent with each other)
__device__ __forceinline__ uint32_t load_device_scope(const uint32_t* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }

__device__ __forceinline__ float fast_rcp(float x) { return __builtin_amdgcn_rcpf(x); }   // v_rcp_f32, 1 ulp
// A value the compiler must keep in a vector register (and cannot fold back into scalar address arithmetic): moves the per-pair LDS address
// of the blend walk from two scalar instructions + a copy to one vector shift-add, in a loop bound by the scalar unit.
__device__ __forceinline__ unsigned in_vector_register(unsigned x) { asm volatile("" : "+v"(x)); return x; }
__device__ __forceinline__ float fast_exp2(float x) { return __builtin_amdgcn_exp2f(x); } // v_exp_f32, 1 ulp (__expf = this after a multiply by log2 e)

// Matrix core, fp32 in / fp32 accumulate (v_mfma_f32_16x16x4_f32, 32 cycles on the SIMD's matrix pipe, beside the vector pipe):
// D[i][j] += sum_k A[i][k] B[k][j] for a 16 x 4 A and a 4 x 16 B. Lane l supplies A[l & 15][l >> 4] and B[l >> 4][l & 15] and holds
// D[4 (l >> 4) + r][l & 15] in acc[r], r = 0..3. Bitwise a k-ordered fmaf chain (no wider internal accumulation). Full EXEC required.
typedef float fgs_acc4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ void wave_mfma_16x16x4(const float a, const float b, fgs_acc4& acc) {
    acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc, 0, 0, 0);
}
// tells the compiler that v is the same in every lane (v_readfirstlane_b32 -> a scalar register): loops and branches on it stay scalar
__device__ __forceinline__ unsigned wave_uniform(const unsigned v) { return static_cast<unsigned>(__builtin_amdgcn_readfirstlane(static_cast<int>(v))); }
__device__ __forceinline__ uint64_t wave_uniform(const uint64_t v) {
    return (static_cast<uint64_t>(wave_uniform(static_cast<unsigned>(v >> 32))) << 32) | wave_uniform(static_cast<unsigned>(v));
}
// per lane: bit `lane` of the wave-uniform mask set ? b : a -- ONE v_cndmask_b32 with the mask as its scalar condition (what a ballot produces and
// scalar instructions combine); written out because the compiler would shift the 64-bit mask by the lane id instead
__device__ __forceinline__ float lane_select(const uint64_t mask, const float a, const float b) {
    float r;
    asm("v_cndmask_b32_e64 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "s"(mask));
    return r;
}
// v (wave-uniform) into lane `lane` (wave-uniform) of `old`, the other lanes keep theirs (this compiler has no v_writelane builtin: a compare + select)
__device__ __forceinline__ unsigned wave_write_lane(const unsigned old, const unsigned v, const unsigned lane) {
    return lane_id() == lane ? v : old;
}
// every lane reads `v` of an arbitrary lane (ds_bpermute_b32: the LDS crossbar, no memory)
__device__ __forceinline__ unsigned wave_shuffle(const unsigned v, const unsigned src_lane) {
    return static_cast<unsigned>(__builtin_amdgcn_ds_bpermute(static_cast<int>(src_lane << 2), static_cast<int>(v)));
}

// One step of the backward pixel pipeline for NV per-pixel values:
//   feed[]  rotates down by one lane (lane l takes lane l+1, lane 63 takes lane 0)      -- DPP wave_rol:1
//   state[] shifts up by one lane (lane l takes lane l-1) and lane 0 takes its own feed[] -- DPP wave_shr:1 with
//           `old` = feed, so the inject costs no extra instruction.
// Call order inside a step: state first (uses feed before it rotates), then feed.
template <int NV>
__device__ __forceinline__ void pipeline_advance(float (&state)[NV], float (&feed)[NV]) {
#pragma unroll
    for (int k = 0; k < NV; ++k) {
        state[k] = __int_as_float(__builtin_amdgcn_update_dpp(__float_as_int(feed[k]), __float_as_int(state[k]),
                                                              0x138 /*wave_shr:1*/, 0xf, 0xf, false));
        feed[k] = __int_as_float(__builtin_amdgcn_update_dpp(__float_as_int(feed[k]), __float_as_int(feed[k]),
                                                             0x134 /*wave_rol:1*/, 0xf, 0xf, false));
    }
}

}  // namespace fgs
