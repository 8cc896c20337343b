// TEST-ONLY shadow of csrc/fgs_wave.h for the CPU simulation build: same interface, collectives implemented on the fiber
// scheduler in hip/hip_runtime.h instead of gfx950 cross-lane instructions. Semantics mirror the hardware ones
// (64-bit ballots of active lanes, wave_shr:1 / wave_rol:1 data movement) as documented in the product header.
#pragma once
#include <hip/hip_runtime.h>
#include <cstdint>

namespace fgs {

inline unsigned lane_id() { return sim::tid() & 63u; }

inline uint64_t wave_ballot(bool p) {
    const int par = sim::next_parity(true);
    sim::me().slot[par][0] = p ? 1 : 0;
    const unsigned g = sim::sync_scope(true);
    sim::Block& b = sim::blk();
    const int first = (b.cur / 64) * 64;
    uint64_t m = 0;
    for (int i = first; i < std::min(first + 64, b.n); ++i)
        if (b.lanes[i].gen_wave >= g && b.lanes[i].slot[par][0]) m |= 1ull << (i - first);
    return m;
}
inline unsigned lanes_below(uint64_t m) { return static_cast<unsigned>(__builtin_popcountll(m & ((1ull << lane_id()) - 1ull))); }

inline uint64_t sim_exchange(uint64_t v, int src_lane) {
    const int par = sim::next_parity(true);
    sim::me().slot[par][0] = v;
    const unsigned g = sim::sync_scope(true);
    sim::Block& b = sim::blk();
    const int first = (b.cur / 64) * 64;
    const sim::Lane& s = b.lanes[first + src_lane];
    return s.gen_wave >= g ? s.slot[par][0] : 0;
}
inline unsigned wave_read(unsigned v, int src_lane) { return static_cast<unsigned>(sim_exchange(v, src_lane)); }
inline float wave_read(float v, int src_lane) { return __uint_as_float(static_cast<unsigned>(sim_exchange(__float_as_uint(v), src_lane))); }

template <class Op> inline unsigned sim_reduce(unsigned v, Op op) {
    const int par = sim::next_parity(true);
    sim::me().slot[par][0] = v;
    const unsigned g = sim::sync_scope(true);
    sim::Block& b = sim::blk();
    const int first = (b.cur / 64) * 64;
    bool have = false; unsigned r = 0;
    for (int i = first; i < std::min(first + 64, b.n); ++i) {
        if (b.lanes[i].gen_wave < g) continue;
        const unsigned x = static_cast<unsigned>(b.lanes[i].slot[par][0]);
        r = have ? op(r, x) : x; have = true;
    }
    return r;
}
inline unsigned wave_exclusive_sum(unsigned v) {
    const int par = sim::next_parity(true);
    sim::me().slot[par][0] = v;
    const unsigned g = sim::sync_scope(true);
    sim::Block& b = sim::blk();
    const int first = (b.cur / 64) * 64;
    unsigned r = 0;
    for (int i = first; i < b.cur; ++i) if (b.lanes[i].gen_wave >= g) r += static_cast<unsigned>(b.lanes[i].slot[par][0]);
    return r;
}
inline unsigned wave_inclusive_sum(unsigned v) { return wave_exclusive_sum(v) + v; }
inline unsigned wave_sum(unsigned v) { return sim_reduce(v, [](unsigned a, unsigned b) { return a + b; }); }
inline unsigned wave_max(unsigned v) { return sim_reduce(v, [](unsigned a, unsigned b) { return a > b ? a : b; }); }

inline float wave_sum_to_lane63(float v) {   // valid in lane 63 only on hardware; the stand-in returns the total everywhere else too
    const int par = sim::next_parity(true);
    sim::me().slot[par][0] = __float_as_uint(v);
    const unsigned g = sim::sync_scope(true);
    sim::Block& b = sim::blk();
    const int first = (b.cur / 64) * 64;
    float r = 0.0f;
    for (int i = first; i < std::min(first + 64, b.n); ++i) if (b.lanes[i].gen_wave >= g) r += __uint_as_float(static_cast<unsigned>(b.lanes[i].slot[par][0]));
    return (b.cur - first) == 63 ? r : -12345.0f;   // poison the other lanes so misuse shows up in the parity tests
}
inline float wave_shift_up1(float v) {
    const unsigned lane = lane_id();
    const float up = __uint_as_float(static_cast<unsigned>(sim_exchange(__float_as_uint(v), static_cast<int>((lane + 63u) % 64u))));
    return lane == 0 ? v : up;
}
inline float wave_shift_up1_zero(float v) {
    const unsigned lane = lane_id();
    const float up = __uint_as_float(static_cast<unsigned>(sim_exchange(__float_as_uint(v), static_cast<int>((lane + 63u) % 64u))));
    return lane == 0 ? 0.0f : up;
}
inline void wave_lds_fence() { sim::sync_scope(true); }
inline uint32_t load_device_scope(const uint32_t* p) { return *p; }
inline float fast_rcp(float x) { return 1.0f / x; }
inline float fast_exp2(float x) { return exp2f(x); }
inline unsigned in_vector_register(unsigned x) { return x; }
inline float4 load_float4_nt(const float* p) { return *reinterpret_cast<const float4*>(p); }
inline void store_float4_nt(float* p, const float4 v) { *reinterpret_cast<float4*>(p) = v; }

// v_mfma_f32_16x16x4_f32 on the fiber scheduler: a k-ordered fmaf chain, which is what the hardware computes bit for bit
// (cdna_hip_programming.md, "FP32-input MFMA"). Lane l supplies A[l & 15][l >> 4], B[l >> 4][l & 15], holds D[4 (l >> 4) + r][l & 15].
struct fgs_acc4 { float v[4]; float& operator[](int i) { return v[i]; } const float& operator[](int i) const { return v[i]; } };
inline void wave_mfma_16x16x4(const float a, const float b, fgs_acc4& acc) {
    const int par = sim::next_parity(true);
    sim::Lane& L = sim::me();
    L.slot[par][0] = __float_as_uint(a); L.slot[par][1] = __float_as_uint(b);
    const unsigned g = sim::sync_scope(true);
    sim::Block& blk = sim::blk();
    const int first = (blk.cur / 64) * 64, lane = blk.cur - first;
    const int col = lane & 15, rowg = lane >> 4;
    for (int r = 0; r < 4; ++r) {
        const int row = 4 * rowg + r;
        float d = acc[r];
        for (int k = 0; k < 4; ++k) {
            const sim::Lane& la = blk.lanes[first + row + 16 * k];      // A[row][k]
            const sim::Lane& lb = blk.lanes[first + 16 * k + col];      // B[k][col]
            if (la.gen_wave < g || lb.gen_wave < g) { fprintf(stderr, "[sim] wave_mfma_16x16x4 needs all 64 lanes\n"); abort(); }
            d = fmaf(__uint_as_float(static_cast<unsigned>(la.slot[par][0])), __uint_as_float(static_cast<unsigned>(lb.slot[par][1])), d);
        }
        acc[r] = d;
    }
}
inline unsigned wave_uniform(const unsigned v) { return v; }
inline uint64_t wave_uniform(const uint64_t v) { return v; }
inline float lane_select(const uint64_t mask, const float a, const float b) { return ((mask >> lane_id()) & 1ull) ? b : a; }
inline unsigned wave_write_lane(const unsigned old, const unsigned v, const unsigned lane) { return lane_id() == lane ? v : old; }
inline unsigned wave_shuffle(const unsigned v, const unsigned src_lane) { return static_cast<unsigned>(sim_exchange(v, static_cast<int>(src_lane & 63u))); }

template <int NV>
inline void pipeline_advance(float (&state)[NV], float (&feed)[NV]) {
    static_assert(2 * NV <= sim::kSlots, "slot overflow");
    const int par = sim::next_parity(true);
    sim::Lane& L = sim::me();
    for (int k = 0; k < NV; ++k) { L.slot[par][k] = __float_as_uint(state[k]); L.slot[par][NV + k] = __float_as_uint(feed[k]); }
    sim::sync_scope(true);
    sim::Block& b = sim::blk();
    const int first = (b.cur / 64) * 64;
    const int lane = b.cur - first;
    const sim::Lane& up = b.lanes[first + (lane + 63) % 64];     // lane - 1
    const sim::Lane& down = b.lanes[first + (lane + 1) % 64];    // lane + 1
    for (int k = 0; k < NV; ++k) {
        state[k] = lane == 0 ? feed[k] : __uint_as_float(static_cast<unsigned>(up.slot[par][k]));       // wave_shr:1, old = feed
        feed[k] = __uint_as_float(static_cast<unsigned>(down.slot[par][NV + k]));                       // wave_rol:1
    }
}

}  // namespace fgs
