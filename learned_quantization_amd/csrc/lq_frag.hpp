// lq_frag.hpp -- fragment arithmetic of the batch's column partials (plain C++: also compiled on the host by
// tests/tools/check_frag.cpp, which compares it with a brute-force enumeration; layout described in lq_batch_cols.hpp)
#ifndef LQ_FRAG_HPP_
#define LQ_FRAG_HPP_
#include <stdint.h>

#include "lq_fastdiv.hpp"      // LQ_HD

namespace lq {

struct FragGeom {
    uint32_t finner;      // columns per group in the partial layout (inner, or 1: a fragment per column)
    uint32_t lq;          // lcm(finner, 256) / 256
    uint32_t F;           // fragments per row block
    uint32_t gpb;         // groups per finalize block (frag_gpb)
};

LQ_HD uint32_t frag_gcd(uint32_t a, uint32_t b) {
    while (b) {
        const uint32_t t = a % b;
        a = b;
        b = t;
    }
    return a;
}
// index of the fragment that contains column x (x a tile start or a group start)
LQ_HD uint32_t frag_index(uint32_t x, uint32_t finner, uint32_t lq) { return x / finner + (x >> 8) - (x >> 8) / lq; }
// groups per 256-thread finalize block: its 64 lanes hold the fragments of `gpb` consecutive groups, cut groups included
LQ_HD uint32_t frag_gpb(uint32_t finner) {
    if (256u % finner == 0u) return 64u;                       // no tile boundary ever cuts a group
    uint32_t g = 64u;
    while (g > 1u && g + (g * finner) / 256u + 1u > 64u) --g;
    return g;
}
static inline FragGeom make_frag_geom(int64_t G, int64_t inner, int64_t C, bool grouped) {
    FragGeom f;
    f.finner = grouped ? (uint32_t)inner : 1u;
    f.lq = f.finner / frag_gcd(f.finner, 256u);
    uint32_t cuts = 0;
    for (int64_t x = 256; x < C; x += 256) cuts += (x % f.finner) ? 1u : 0u;
    f.F = (uint32_t)(C / f.finner) + cuts;
    f.gpb = frag_gpb(f.finner);
    (void)G;
    return f;
}

}  // namespace lq

#endif
