// check_frag.cpp -- csrc/lq_frag.hpp (fragment index arithmetic of the batch's column partials, lq_batch_cols.hpp) against a
// brute-force enumeration: host build of the very header the kernels include.  Prints "<cases> cases, <n> mismatches".
#include <stdio.h>
#include <stdlib.h>

#include <vector>

#include "lq_frag.hpp"

int main() {
    unsigned long long cases = 0, bad = 0;
    const uint32_t Gs[] = {1, 2, 3, 7, 28, 29, 57, 64, 100, 128, 255, 256, 257, 512, 1000, 2048};
    for (uint32_t finner = 1; finner <= 64; ++finner) {
        for (uint32_t G : Gs) {
            const uint32_t C = G * finner;
            if (C % 4) continue;                       // float4 tiles only
            const lq::FragGeom fg = lq::make_frag_geom(G, finner, C, true);
            // brute force: walk the columns, a new fragment starts at every group start and every tile start
            std::vector<uint32_t> frag_of_col(C);
            uint32_t f = 0;
            for (uint32_t c = 0; c < C; ++c) {
                if (c && (c % finner == 0 || c % 256 == 0)) ++f;
                frag_of_col[c] = f;
            }
            ++cases;
            if (fg.F != f + 1) ++bad;
            if (fg.finner != finner || fg.lq * 256u % finner != 0) ++bad;
            for (uint32_t x = 0; x < C; x += 256) {    // tile starts
                ++cases;
                if (lq::frag_index(x, finner, fg.lq) != frag_of_col[x]) ++bad;
            }
            for (uint32_t g = 0; g < G; ++g) {         // group starts, fragments per group (the finalize's `n`), adjacency
                const uint32_t xs = g * finner;
                const uint32_t n = 1u + (((xs + finner - 1u) >> 8) - (xs >> 8));
                ++cases;
                if (lq::frag_index(xs, finner, fg.lq) != frag_of_col[xs]) ++bad;
                if (frag_of_col[xs + finner - 1] - frag_of_col[xs] + 1 != n || n > 2) ++bad;
            }
            for (uint32_t g0 = 0; g0 < G; g0 += fg.gpb) {      // a finalize block's fragments fit its 64 lanes
                const uint32_t gend = g0 + fg.gpb < G ? g0 + fg.gpb : G;
                const uint32_t c0 = frag_of_col[g0 * finner], c1 = frag_of_col[gend * finner - 1] + 1;
                ++cases;
                if (c1 - c0 > 64 || fg.gpb < 1 || fg.gpb > 64) ++bad;
            }
        }
    }
    // the per-column layout (finner forced to 1): fragment == column
    {
        const lq::FragGeom fg = lq::make_frag_geom(512, 9, 4608, false);
        ++cases;
        if (fg.finner != 1 || fg.lq != 1 || fg.F != 4608 || lq::frag_index(1024, 1, 1) != 1024) ++bad;
    }
    printf("%llu cases, %llu mismatches\n", cases, bad);
    return bad ? 1 : 0;
}
