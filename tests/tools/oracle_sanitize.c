/* Runs the scalar C oracle under AddressSanitizer + UBSan on ragged shapes (CPU only; the GPU pool has no GPU ASan).
 * Build: gcc -O1 -g -fsanitize=address,undefined -fno-sanitize-recover=all -o oracle_sanitize oracle_sanitize.c ../../oracle/lq_oracle.c -lm */
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>

void lqo_fq_forward(const float*, const float*, float*, float*, int64_t, int64_t, int64_t);
void lqo_nq_scale_grad(const float*, const float*, const float*, float, float*, float*, float*, int64_t*, int64_t, int64_t, int64_t);
float lqo_maxbin_term(const float*, const float*, int64_t, int64_t, int64_t);
float lqo_difference_term(const float*, const float*, int64_t, int64_t, int64_t);
float lqo_inverse_term(const float*, int64_t);

int main(void) {
    static const int64_t descs[][3] = {{1, 1, 1}, {1, 1, 7}, {3, 5, 1}, {2, 3, 4}, {1, 17, 3}, {9, 4, 2}, {5, 1, 13}};
    for (unsigned k = 0; k < sizeof(descs) / sizeof(descs[0]); ++k) {
        const int64_t outer = descs[k][0], G = descs[k][1], inner = descs[k][2], n = outer * G * inner;
        float* P = malloc(n * sizeof(float)); float* dy = malloc(n * sizeof(float)); float* out = malloc(n * sizeof(float));
        float* q = malloc(n * sizeof(float)); float* s = malloc(G * sizeof(float)); float* ds = malloc(G * sizeof(float));
        float* mq = malloc(G * sizeof(float)); float* mean = malloc(G * sizeof(float)); int64_t* below = malloc(G * sizeof(int64_t));
        for (int64_t i = 0; i < n; ++i) { P[i] = (float)((i * 37 % 101) - 50) * 0.013f; dy[i] = (float)((i * 11 % 13) - 6) * 1e-4f; }
        if (n > 2) { P[1] = 0.0f; dy[2] = NAN; }
        for (int64_t g = 0; g < G; ++g) s[g] = 0.01f + 0.003f * (float)g;
        lqo_fq_forward(P, s, out, q, outer, G, inner);
        lqo_fq_forward(P, s, out, NULL, outer, G, inner);
        lqo_nq_scale_grad(P, s, dy, 1e-3f, ds, mq, mean, below, outer, G, inner);
        lqo_nq_scale_grad(P, s, dy, 0.0f, ds, NULL, NULL, NULL, outer, G, inner);
        volatile float sink = lqo_maxbin_term(P, s, outer, G, inner) + lqo_difference_term(P, s, outer, G, inner) + lqo_inverse_term(s, G);
        (void)sink;
        free(P); free(dy); free(out); free(q); free(s); free(ds); free(mq); free(mean); free(below);
    }
    printf("oracle_sanitize: ok\n");
    return 0;
}
