/* Scalar C restatement of the learned-quantization hot path -- TEST INFRASTRUCTURE ONLY.
 *
 * Third, independent implementation used to cross-check oracle/lq_oracle.py
 * (NumPy, axis reductions) and oracle/lq_oracle_f64.py (thesis math).  Plain
 * float32 arithmetic, one element at a time, through the C-ABI group descriptor
 *     element i of a contiguous tensor uses scale[(i / inner) % G],  numel = outer*G*inner.
 * Build: oracle/Makefile  (gcc -O2 -ffp-contract=off -fno-fast-math; x86-64 SSE2
 * float math is IEEE-754 single precision, division correctly rounded).
 *
 * Follows, line by line:
 *   forward            /root/reference/MNIST/nested_quantization_layer/custom_components/custom_layers.py:55-60
 *   ratio stage        custom_layers.py:63-64
 *   group reductions   custom_layers.py:67-73, 91-99
 *   vote + mean        custom_layers.py:77-88, 103-114
 *   result             custom_layers.py:116-118
 *   penalties          /root/reference/CIFAR-10/custom_loss_terms/custom_components/custom_loss_functions.py:75-116,161-195,240-275
 *
 * reduce_mean is restated as (double-accumulated sum, rounded to float) / count:
 * TensorFlow's float32 summation order is unspecified, so the most accurate sum
 * is the neutral choice; consumers compare with rtol 1e-5.
 * Parity status: "parity unpinned" by upstream (the reference has no tests); see
 * the header of oracle/lq_oracle.py for what pins the oracle instead.
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#define LQO_EPS_F32 1.1920928955078125e-07f

#ifdef __cplusplus
extern "C" {
#endif

int lqo_version(void) { return 1; }

/* a1: t = P / s ; q = floor(t) ; out = q * s */
void lqo_fq_forward(const float* P, const float* s, float* out, float* q,
                    int64_t outer, int64_t G, int64_t inner) {
    int64_t n = outer * G * inner;
    for (int64_t i = 0; i < n; ++i) {
        float sg = s[(i / inner) % G];
        volatile float t = P[i] / sg;
        float r = floorf(t);
        if (q) q[i] = r;
        if (out) out[i] = r * sg;
    }
}

/* a2-a5: ds[g] = mean_g(sg) * max_g|q| ; optionally exposes the per-group pieces. */
void lqo_nq_scale_grad(const float* P, const float* s, const float* dy, float lam,
                       float* ds, float* maxq_out, float* mean_out, int64_t* below_out,
                       int64_t outer, int64_t G, int64_t inner) {
    int64_t n = outer * G * inner;
    float* maxq = (float*)calloc((size_t)G, sizeof(float));
    double* sum_below = (double*)calloc((size_t)G, sizeof(double));
    int64_t* n_below = (int64_t*)calloc((size_t)G, sizeof(int64_t));
    int* has_nan_max = (int*)calloc((size_t)G, sizeof(int));
    for (int64_t i = 0; i < n; ++i) {
        int64_t g = (i / inner) % G;
        float sg = s[g];
        volatile float t = P[i] / sg;
        float r = floorf(t);
        volatile float o = r * sg;
        float nz = (o == 0.0f) ? LQO_EPS_F32 : o;            /* :63 */
        volatile float ratio = fabsf(dy[i]) / fabsf(nz);      /* :64 */
        float aq = fabsf(r);
        if (aq != aq) has_nan_max[g] = 1;
        if (aq > maxq[g]) maxq[g] = aq;                       /* :68 / :94 */
        if (!(ratio >= lam)) {                                /* :70 / :97 ; NaN counts as "not above" */
            n_below[g] += 1;
            volatile float d = lam - ratio;
            sum_below[g] += (double)(-1.0f * fabsf(tanhf(d)));   /* :84 / :110 */
        }
    }
    for (int64_t g = 0; g < G; ++g) {
        int64_t cnt = outer * inner;
        float mean;
        if (n_below[g] == 0) {
            mean = -1.0f * fabsf(tanhf(lam));                 /* :79 / :105: every element gets the constant */
        } else {
            mean = (float)sum_below[g] / (float)cnt;          /* :87 / :113 */
        }
        float m = has_nan_max[g] ? NAN : maxq[g];
        if (ds) ds[g] = mean * m;                             /* :116 */
        if (maxq_out) maxq_out[g] = m;
        if (mean_out) mean_out[g] = mean;
        if (below_out) below_out[g] = n_below[g];
    }
    free(maxq); free(sum_below); free(n_below); free(has_nan_max);
}

/* a10 tensor term: mean over groups of max(|P| / s)  (custom_loss_functions.py:90-94,110) */
float lqo_maxbin_term(const float* P, const float* s, int64_t outer, int64_t G, int64_t inner) {
    int64_t n = outer * G * inner;
    float* mb = (float*)malloc((size_t)G * sizeof(float));
    for (int64_t g = 0; g < G; ++g) mb[g] = -INFINITY;
    for (int64_t i = 0; i < n; ++i) {
        int64_t g = (i / inner) % G;
        volatile float t = fabsf(P[i]) / s[g];
        if (t > mb[g]) mb[g] = t;
    }
    double acc = 0.0;
    for (int64_t g = 0; g < G; ++g) acc += (double)mb[g];
    free(mb);
    return (float)acc / (float)G;
}

/* a11 tensor term: mean |P - P/s|  (custom_loss_functions.py:172-176) */
float lqo_difference_term(const float* P, const float* s, int64_t outer, int64_t G, int64_t inner) {
    int64_t n = outer * G * inner;
    double acc = 0.0;
    for (int64_t i = 0; i < n; ++i) {
        volatile float pq = P[i] / s[(i / inner) % G];
        volatile float u = P[i] - pq;
        acc += (double)fabsf(u);
    }
    return (float)acc / (float)n;
}

/* a12 tensor term: mean 1/where(s==0, eps, s)  (custom_loss_functions.py:252-256) */
float lqo_inverse_term(const float* s, int64_t G) {
    double acc = 0.0;
    for (int64_t g = 0; g < G; ++g) {
        float v = (s[g] == 0.0f) ? LQO_EPS_F32 : s[g];
        volatile float inv = 1.0f / v;
        acc += (double)inv;
    }
    return (float)acc / (float)G;
}

#ifdef __cplusplus
}
#endif
