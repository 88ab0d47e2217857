/* Exhaustive-mantissa check of the "uniform divisor" fast division used by the HIP kernels
 * (learned_quantization_amd/csrc/lq_kernels.hip: div_by_uniform):
 *     r  = RN(1/s)                      once per block
 *     q0 = RN(x*r); e0 = fma(-s,q0,x); q1 = fma(e0,r,q0); e1 = fma(-s,q1,x); t = fma(e1,r,q1)
 * must equal the IEEE correctly rounded x/s for every x in the exponent window the kernel uses the
 * fast path for, and every s in its window except all-ones mantissas (Markstein's condition).
 * For each tested s, all 2^23 mantissas of x are tried at three exponents and both signs
 * (scaling x by a power of two scales every intermediate exactly inside the window).
 * usage: check_fast_div <n_random_s> <seed>   -> prints mismatches, exit code 1 if any. */
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

static inline float u2f(uint32_t u) { float f; memcpy(&f, &u, 4); return f; }
static inline uint32_t f2u(float f) { uint32_t u; memcpy(&u, &f, 4); return u; }

static inline float fast_div(float x, float s, float r) {
    float q0 = x * r;
    float e0 = fmaf(-s, q0, x);
    float q1 = fmaf(e0, r, q0);
    float e1 = fmaf(-s, q1, x);
    return fmaf(e1, r, q1);
}

static uint64_t rng_state;
static uint32_t rnd(void) {
    rng_state = rng_state * 6364136223846793005ull + 1442695040888963407ull;
    return (uint32_t)(rng_state >> 32);
}

int main(int argc, char** argv) {
    int n_random = argc > 1 ? atoi(argv[1]) : 16;
    rng_state = argc > 2 ? strtoull(argv[2], 0, 10) : 1;
    /* fixed set: the scales the reference and the benchmarks actually use + nasty mantissas */
    float fixed[] = {1.1920929e-05f, 0.5f, 1.0f, 2.0f, 1e-3f, 1e-4f, 0.0123f, 3.0f, 0.1f, 0.7f,
                     1.0000001f, 1.9999998f /* mantissa 0x7ffffe */, 1.5f, 1.3333334f, 5.9604645e-08f, 7.0f};
    int n_fixed = (int)(sizeof(fixed) / sizeof(fixed[0]));
    uint64_t bad = 0, total = 0;
    for (int k = 0; k < n_fixed + n_random; ++k) {
        float s;
        if (k < n_fixed) s = fixed[k];
        else {
            uint32_t man = rnd() & 0x7fffffu;
            if (man == 0x7fffffu) man = 0x7ffffeu;           /* excluded by the kernel's window */
            uint32_t ex = 127u - 40u + (rnd() % 81u);         /* 2^-40 .. 2^40 */
            s = u2f((ex << 23) | man);
        }
        volatile float rv = 1.0f / s;
        float r = rv;
        static const uint32_t exps[3] = {127u, 127u - 80u, 127u + 80u};
        for (int e = 0; e < 3; ++e) {
            for (uint32_t man = 0; man < (1u << 23); ++man) {
                float x = u2f((exps[e] << 23) | man);
                volatile float ref = x / s;
                float got = fast_div(x, s, r);
                float gotn = fast_div(-x, s, r);
                ++total;
                if (f2u(got) != f2u(ref) || f2u(gotn) != (f2u(ref) ^ 0x80000000u)) {
                    if (bad < 10) fprintf(stderr, "MISMATCH s=%a x=%a ref=%a got=%a\n", s, x, ref, got);
                    ++bad;
                }
            }
        }
    }
    printf("checked %llu (x,s) pairs over %d divisors: %llu mismatches\n", (unsigned long long)total,
           n_fixed + n_random, (unsigned long long)bad);
    return bad ? 1 : 0;
}
