// lq_fastdiv.hpp -- division by a launch-invariant 32-bit divisor (plain C++: also compiled on the host by
// tests/tools/check_fastdiv.cpp, which compares it with `/` -- the group index of the flat streaming kernels rests on it)
#ifndef LQ_FASTDIV_HPP_
#define LQ_FASTDIV_HPP_
#include <stdint.h>

#if defined(__HIPCC__)
#define LQ_HD __host__ __device__ __forceinline__
#else
#define LQ_HD static inline
#endif

namespace lq {

// Granlund & Montgomery, "Division by invariant integers using multiplication", fig. 4.1: exact for every 32-bit dividend,
// 5 VALU instead of the ~25 of a 32-bit udiv.
struct FastDiv {
    uint32_t d, m, sh1, sh2;
};
static inline FastDiv make_fastdiv(uint32_t d) {      // d >= 1
    FastDiv f;
    uint32_t l = 0;
    while (l < 32 && ((uint64_t)1 << l) < d) ++l;
    f.d = d;
    f.m = (uint32_t)((((uint64_t)1 << 32) * (((uint64_t)1 << l) - d)) / d + 1);
    f.sh1 = l < 1 ? l : 1;
    f.sh2 = l > 1 ? l - 1 : 0;
    return f;
}
LQ_HD uint32_t fd_mulhi(uint32_t a, uint32_t b) {
#if defined(__HIP_DEVICE_COMPILE__)
    return __umulhi(a, b);
#else
    return (uint32_t)(((uint64_t)a * (uint64_t)b) >> 32);
#endif
}
LQ_HD uint32_t fd_div(const FastDiv& f, uint32_t n) {
    const uint32_t t = fd_mulhi(f.m, n);
    return (t + ((n - t) >> f.sh1)) >> f.sh2;
}
LQ_HD uint32_t fd_mod(const FastDiv& f, uint32_t n) { return n - fd_div(f, n) * f.d; }

}  // namespace lq

#endif
