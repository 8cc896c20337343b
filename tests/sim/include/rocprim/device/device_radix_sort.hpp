// TEST-ONLY CPU stand-in for rocprim::radix_sort_pairs: stable sort on bits [begin_bit, end_bit), result left in the
// buffer that was `alternate` (so callers' selector handling is exercised).
#pragma once
#include <hip/hip_runtime.h>
#include <numeric>
#include <rocprim/types/double_buffer.hpp>
namespace rocprim {
template <class Key, class Value, class Size>
inline hipError_t radix_sort_pairs(void* temp, size_t& bytes, double_buffer<Key>& keys, double_buffer<Value>& values, Size size,
                                   unsigned begin_bit = 0, unsigned end_bit = 8 * sizeof(Key), hipStream_t = nullptr, bool = false) {
    if (temp == nullptr) { bytes = 64; return hipSuccess; }
    const size_t n = static_cast<size_t>(size);
    std::vector<size_t> perm(n);
    std::iota(perm.begin(), perm.end(), size_t{0});
    const Key* k = keys.current();
    const unsigned long long mask = (end_bit - begin_bit) >= 64 ? ~0ull : ((1ull << (end_bit - begin_bit)) - 1ull);
    std::stable_sort(perm.begin(), perm.end(), [&](size_t a, size_t b) {
        return ((static_cast<unsigned long long>(k[a]) >> begin_bit) & mask) < ((static_cast<unsigned long long>(k[b]) >> begin_bit) & mask);
    });
    for (size_t i = 0; i < n; ++i) { keys.alternate()[i] = keys.current()[perm[i]]; values.alternate()[i] = values.current()[perm[i]]; }
    keys.swap(); values.swap();
    return hipSuccess;
}
}  // namespace rocprim
