// Standalone user of the C ABI (no Python, no torch): what a non-Python host would write.
// Build: hipcc --offload-arch=gfx950 -I include -o abi_smoke tests/tools/abi_smoke.cpp -L learned_quantization_amd/csrc -llq_hip
// Checks lq_fq_forward / lq_fq_scale_grad / lq_fq_fwd_bwd_fused on a 64x3x32x32 tensor against a host
// restatement of custom_layers.py:55-118 (same arithmetic as oracle/lq_oracle.c) and the error paths.
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <vector>

#include "lq_hip.h"

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s line %d\n", hipGetErrorString(e), __LINE__); return 2; } } while (0)
#define LQ(x) do { int rc = (x); if (rc != LQ_OK) { printf("lq error %s: %s (line %d)\n", lq_status_string(rc), lq_last_error(), __LINE__); return 3; } } while (0)

int main() {
    const int64_t outer = 64, G = 3, inner = 32 * 32, n = outer * G * inner;
    const float lam = 1e-3f;
    std::vector<float> P(n), dy(n), s = {0.5f, 1.0f, 2.0f}, out(n), ds(G), ds2(G), out2(n);
    srand(42);
    for (int64_t i = 0; i < n; ++i) {
        P[i] = 255.0f * (float)rand() / (float)RAND_MAX;
        dy[i] = 1e-3f * ((float)rand() / (float)RAND_MAX - 0.5f);
    }
    float *dP, *dDy, *dS, *dOut, *dDs;
    void* ws;
    const size_t ws_bytes = lq_workspace_bytes(outer, G, inner);
    CK(hipMalloc(&dP, n * 4)); CK(hipMalloc(&dDy, n * 4)); CK(hipMalloc(&dS, G * 4)); CK(hipMalloc(&dOut, n * 4));
    CK(hipMalloc(&dDs, G * 4)); CK(hipMalloc(&ws, ws_bytes));
    CK(hipMemcpy(dP, P.data(), n * 4, hipMemcpyHostToDevice)); CK(hipMemcpy(dDy, dy.data(), n * 4, hipMemcpyHostToDevice));
    CK(hipMemcpy(dS, s.data(), G * 4, hipMemcpyHostToDevice));
    hipStream_t st; CK(hipStreamCreate(&st));

    if (lq_version() != LQ_ABI_VERSION) { printf("ABI version mismatch\n"); return 4; }
    LQ(lq_fq_forward(dP, dS, dOut, nullptr, LQ_Q_NONE, outer, G, inner, st));
    LQ(lq_fq_scale_grad(dP, dS, dDy, lam, dDs, nullptr, ws, ws_bytes, outer, G, inner, st));
    CK(hipStreamSynchronize(st));
    CK(hipMemcpy(out.data(), dOut, n * 4, hipMemcpyDeviceToHost)); CK(hipMemcpy(ds.data(), dDs, G * 4, hipMemcpyDeviceToHost));
    LQ(lq_fq_fwd_bwd_fused(dP, dS, dDy, lam, dOut, dDs, ws, ws_bytes, outer, G, inner, st));
    CK(hipStreamSynchronize(st));
    CK(hipMemcpy(out2.data(), dOut, n * 4, hipMemcpyDeviceToHost)); CK(hipMemcpy(ds2.data(), dDs, G * 4, hipMemcpyDeviceToHost));

    // host restatement
    std::vector<double> sum(G, 0.0); std::vector<float> maxq(G, 0.f); std::vector<long long> below(G, 0);
    long long bad = 0;
    for (int64_t i = 0; i < n; ++i) {
        const int64_t g = (i / inner) % G;
        volatile float t = P[i] / s[g];
        const float q = floorf(t);
        volatile float o = q * s[g];
        if (memcmp((const void*)&o, &out[i], 4) != 0 || memcmp((const void*)&o, &out2[i], 4) != 0) ++bad;
        const float nz = (o == 0.0f) ? 1.1920928955078125e-07f : o;
        volatile float ratio = fabsf(dy[i]) / fabsf(nz);
        if (fabsf(q) > maxq[g]) maxq[g] = fabsf(q);
        if (!(ratio >= lam)) { below[g]++; volatile float d = lam - ratio; sum[g] += -(double)fabsf(tanhf(d)); }
    }
    for (int64_t g = 0; g < G; ++g) {
        const float mean = below[g] == 0 ? -fabsf(tanhf(lam)) : (float)(sum[g] / (double)(outer * inner));
        const float want = mean * maxq[g];
        if (fabsf(ds[g] - want) > 1e-5f * fabsf(want) || ds[g] != ds2[g]) { printf("ds[%lld] = %.9g / %.9g, want %.9g\n", (long long)g, ds[g], ds2[g], want); ++bad; }
    }
    // error paths: no launch may happen
    if (lq_fq_forward(nullptr, dS, dOut, nullptr, 0, outer, G, inner, st) != LQ_EINVAL) ++bad;
    if (lq_fq_scale_grad(dP, dS, dDy, lam, dDs, nullptr, ws, 16, outer, G, inner, st) != LQ_EWORKSPACE) ++bad;
    if (strlen(lq_last_error()) == 0) ++bad;
    printf("abi_smoke: %lld mismatches (n = %lld, ws = %zu bytes)\n", bad, (long long)n, ws_bytes);
    return bad ? 1 : 0;
}
