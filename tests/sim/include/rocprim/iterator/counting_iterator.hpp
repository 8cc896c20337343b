// TEST-ONLY CPU stand-in for rocprim::counting_iterator
#pragma once
#include <cstddef>
namespace rocprim {
template <class T> struct counting_iterator {
    T first;
    explicit counting_iterator(T f) : first(f) {}
    T operator[](size_t i) const { return static_cast<T>(first + static_cast<T>(i)); }
};
}  // namespace rocprim
