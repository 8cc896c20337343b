// TEST-ONLY CPU stand-in for rocprim::make_transform_iterator
#pragma once
#include <cstddef>
namespace rocprim {
template <class It, class F> struct transform_iterator {
    It it; F f;
    auto operator[](size_t i) const { return f(it[i]); }
};
template <class It, class F> inline transform_iterator<It, F> make_transform_iterator(It it, F f) { return {it, f}; }
}  // namespace rocprim
