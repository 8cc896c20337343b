// TEST-ONLY CPU stand-in for rocprim::exclusive_scan / inclusive_scan
#pragma once
#include <hip/hip_runtime.h>
#include <rocprim/types/double_buffer.hpp>
namespace rocprim {
template <class In, class Out, class Init, class Op>
inline hipError_t exclusive_scan(void* temp, size_t& bytes, In in, Out out, Init init, size_t n, Op op, hipStream_t = nullptr, bool = false) {
    if (temp == nullptr) { bytes = 64; return hipSuccess; }
    auto acc = init;
    for (size_t i = 0; i < n; ++i) { const auto v = in[i]; out[i] = acc; acc = op(acc, v); }
    return hipSuccess;
}
template <class In, class Out, class Op>
inline hipError_t inclusive_scan(void* temp, size_t& bytes, In in, Out out, size_t n, Op op, hipStream_t = nullptr, bool = false) {
    if (temp == nullptr) { bytes = 64; return hipSuccess; }
    if (n == 0) return hipSuccess;
    auto acc = in[0]; out[0] = acc;
    for (size_t i = 1; i < n; ++i) { acc = op(acc, in[i]); out[i] = acc; }
    return hipSuccess;
}
}  // namespace rocprim
