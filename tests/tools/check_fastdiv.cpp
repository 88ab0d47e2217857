// check_fastdiv.cpp -- csrc/lq_fastdiv.hpp against the CPU's `/` and `%` (host build of the very header the kernels include).
// usage: check_fastdiv [random divisors] [seed]; prints "<pairs> pairs, <n> mismatches", exit code 1 on any mismatch.
#include <stdio.h>
#include <stdlib.h>

#include "lq_fastdiv.hpp"

static uint64_t rng_state = 88172645463325252ull;
static uint32_t rnd() {
    rng_state ^= rng_state << 13;
    rng_state ^= rng_state >> 7;
    rng_state ^= rng_state << 17;
    return (uint32_t)(rng_state >> 16);
}

int main(int argc, char** argv) {
    const int n_rand = argc > 1 ? atoi(argv[1]) : 2000;
    if (argc > 2) rng_state ^= (uint64_t)atoll(argv[2]) * 0x9E3779B97F4A7C15ull;
    unsigned long long pairs = 0, bad = 0;
    const uint32_t fixed[] = {1u, 2u, 3u, 4u, 5u, 7u, 9u, 10u, 17u, 30u, 32u, 49u, 63u, 64u, 100u, 196u, 1000u, 1001u, 1023u, 1024u, 1025u,
                              4099u, 4100u, 50176u, 50177u, 65535u, 65536u, 65537u, 12845056u, 0x7fffffffu, 0x80000000u, 0x80000001u,
                              0xfffffffeu, 0xffffffffu};
    const int n_fixed = (int)(sizeof(fixed) / sizeof(fixed[0]));
    for (int k = 0; k < n_fixed + n_rand; ++k) {
        uint32_t d = k < n_fixed ? fixed[k] : (rnd() >> (rnd() % 32));
        if (d == 0) d = 1;
        const lq::FastDiv f = lq::make_fastdiv(d);
        auto check = [&](uint32_t n) {
            ++pairs;
            if (lq::fd_div(f, n) != n / d || lq::fd_mod(f, n) != n % d) ++bad;
        };
        // the edges of every quotient step near 0, near d^2 and near 2^32, then random dividends; fixed divisors get a dense sweep
        const uint32_t edges[] = {0u, 1u, d - 1u, d, d + 1u, 2u * d - 1u, 2u * d, 0x7fffffffu, 0x80000000u, 0xfffffffeu, 0xffffffffu};
        for (uint32_t e : edges) check(e);
        const uint32_t qmax = 0xffffffffu / d;
        for (int j = 0; j < 64; ++j) {
            const uint32_t q = qmax ? rnd() % (qmax + 1u > qmax ? qmax + 1u : qmax) : 0u;
            const uint64_t n0 = (uint64_t)q * d;
            if (n0 <= 0xffffffffull) check((uint32_t)n0);
            if (n0 >= 1 && n0 - 1 <= 0xffffffffull) check((uint32_t)(n0 - 1));
            if (n0 + d - 1 <= 0xffffffffull) check((uint32_t)(n0 + d - 1));
        }
        const int n_dense = k < n_fixed ? 2000000 : 2000;
        for (int j = 0; j < n_dense; ++j) check(rnd());
    }
    printf("%llu pairs, %llu mismatches\n", pairs, bad);
    return bad != 0;
}
