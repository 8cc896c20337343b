// TEST-ONLY CPU stand-in (see tests/sim/README.md)
#pragma once
namespace rocprim {
template <class T> class double_buffer {
    T* b[2]; int sel = 0;
public:
    double_buffer(T* c, T* a) { b[0] = c; b[1] = a; }
    T* current() const { return b[sel]; }
    T* alternate() const { return b[sel ^ 1]; }
    void swap() { sel ^= 1; }
};
template <class T> struct plus { T operator()(const T& a, const T& b) const { return a + b; } };
}  // namespace rocprim
