// lq_aux_kernels.hpp -- scale-sized vector kernels (K5c, K6), integer-view statistics, device self-test
#ifndef LQ_AUX_KERNELS_HPP_
#define LQ_AUX_KERNELS_HPP_
#include "lq_traverse.hpp"

namespace lq {

// ------------------------------------------------------------------------------------------
//  Small vector kernels (scale-sized data).
// ------------------------------------------------------------------------------------------
// mode 0: out = mean(v[0..n))      mode 1: out = mean(1 / where(v==0, eps, v))
template <int MODE>
__global__ __launch_bounds__(kBlock) void k_vec_mean(const float* v, int64_t n, float* out) {
    double acc = 0.0;
    for (int64_t i = threadIdx.x; i < n; i += kBlock) {
        float x = v[i];
        if (MODE == 1) {
            float nz = (x == 0.0f) ? kEpsF32 : x;     // custom_loss_functions.py:252
            x = 1.0f / nz;                            // :255
        }
        acc += (double)x;
    }
    for (int off = 32; off > 0; off >>= 1) acc += __shfl_xor(acc, off, 64);
    __shared__ double lds[kWavesPerBlock];
    if ((threadIdx.x & 63) == 0) lds[threadIdx.x >> 6] = acc;
    __syncthreads();
    if (threadIdx.x == 0) {
        double t = lds[0];
        for (int w = 1; w < kWavesPerBlock; ++w) t += lds[w];
        out[0] = (float)(t / (double)n);
    }
}

__global__ void k_maxbin_ds(const float* s, const float* mb, const float* c_dev, float c_scale, float* ds, int64_t G) {
    const int64_t g = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (g >= G) return;
    const float up = c_dev[0] * c_scale;
    // sum over ties of -(g_i) * t / s with g_i = up/(G*ties): = -(up/G) * mb / s
    ds[g] = -((up / (float)G) * mb[g]) / s[g];
}

__global__ void k_inverse_bwd(const float* s, const float* c_dev, float c_scale, float* ds, int64_t G) {
    const int64_t g = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (g >= G) return;
    const float up = c_dev[0] * c_scale;
    const float sg = s[g];
    // d/ds mean(1/s) = -1/(G s^2); tf.where routes no gradient into s where s == 0
    ds[g] = (sg == 0.0f) ? 0.0f : -((up / (float)G) / sg) / sg;
}

// K6: Adam + MinValueConstraint for one scale tensor.  The 1-based step comes from device memory (hipGraph-capturable: nothing
// about the step is baked into the launch) or, for eager steps, from `step_host` -- the same device arithmetic either way.
// Keras mode forms beta^t in fp32 like tf.pow; torch mode in fp64.
__global__ void k_adam_dev(float* s, const float* ds, float* m, float* v, int64_t n, float lr, float b1, float b2, double lr_d,
                           double b1_d, double b2_d, float f0, float f1, float eps, const int64_t* step_dev, int64_t step_host,
                           float min_value, int mode) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const int64_t step = step_dev ? step_dev[0] : step_host;      // one arithmetic for both forms: graphed == eager bit for bit
    const float g = ds[i];
    float mi = m[i], vi = v[i], w = s[i];
    mi = mi + (g - mi) * f0;
    vi = vi + (g * g - vi) * f1;
    if (mode == LQ_ADAM_KERAS) {
        const float b1p = powf(b1, (float)step), b2p = powf(b2, (float)step);
        const float alpha = lr * sqrtf(1.0f - b2p) / (1.0f - b1p);
        w = w - (mi * alpha) / (sqrtf(vi) + eps);
    } else {
        const double bc1 = 1.0 - pow(b1_d, (double)step), bc2 = 1.0 - pow(b2_d, (double)step);
        const float step_size = (float)(lr_d / bc1);
        const float denom = sqrtf(vi) / (float)sqrt(bc2) + eps;
        w = w - step_size * (mi / denom);
    }
    w = (w < min_value) ? min_value : w;
    m[i] = mi;
    v[i] = vi;
    s[i] = w;
}

__global__ void k_min_project(float* w, int64_t n, float min_value) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const float x = w[i];
    w[i] = (x < min_value) ? min_value : x;   // tf.maximum(w, min_value); NaN propagates
}

// result[a, b] = max over the middle axis of |floor(P/s)| for a tensor viewed (pre, n_axis, post)
// (custom_callbacks.py:98-99: np.max(np.abs(floor(k/s)), axis=1)).  The tensor is walked as a flat stream -- lane l of a
// wave reads element c0 + l: every load is one coalesced 256-byte access whatever the reduced axis is -- in blocks of
// kAbsChunk consecutive elements.  |q| >= 0, so the maximum is taken on the uint32 bit pattern with INTEGER atomics: exact
// and independent of arrival order.  A block first folds its chunk into an LDS table (the outputs a chunk can touch span
// at most (chunk / (n_axis*post) + 2) * post entries); a wave whose 64 elements all belong to one output (post == 1 with
// a long axis: Dense (in, out) reduced over `out`) folds them with a shuffle reduction and issues one atomic.  Entries that
// stayed 0 are not flushed (the caller's result buffer is zeroed by the same ABI call).  IT: 32-bit indices below 2^32
// elements (a 64-bit div/mod per element used to dominate this kernel).
constexpr int kAbsChunk = kBlock * 16;
constexpr int kAbsTab = 4096;

template <typename IT>
__global__ __launch_bounds__(kBlock) void k_q_absmax_axis(const float* __restrict__ P, const float* __restrict__ s, uint32_t* result,
                                                          IT n, IT n_axis, IT post, IT G, IT inner, int use_lds, int span) {
    __shared__ uint32_t tab[kAbsTab];
    const IT c0 = (IT)blockIdx.x * (IT)kAbsChunk;
    const IT slice = n_axis * post;
    const IT o_base = (c0 / slice) * post;
    if (use_lds) {
        for (int t = threadIdx.x; t < span; t += kBlock) tab[t] = 0u;
        __syncthreads();
    }
    float x[16];
#pragma unroll
    for (int u = 0; u < 16; ++u) {                      // all loads first: 16 independent coalesced accesses in flight
        const IT i = c0 + (IT)(u * kBlock) + (IT)threadIdx.x;
        x[u] = P[i < n ? i : n - 1];
    }
#pragma unroll
    for (int u = 0; u < 16; ++u) {
        const IT i = c0 + (IT)(u * kBlock) + (IT)threadIdx.x;
        const bool valid = i < n;
        const IT ic = valid ? i : n - 1;
        const float q = floorf(x[u] / s[(ic / inner) % G]);             // custom_layers.py:56-59 (IEEE division, floor)
        uint32_t bits = valid ? __float_as_uint(fabsf(q)) : 0u;         // NaN compares above every finite value: it propagates like np.max
        const IT a = ic / slice;
        const IT o = a * post + (ic - a * slice) % post;
        const IT o0 = __shfl(o, 0, 64);
        if (__all(o == o0)) {                                           // the whole wave feeds one output
            for (int off = 32; off > 0; off >>= 1) {
                const uint32_t other = __shfl_xor(bits, off, 64);
                bits = other > bits ? other : bits;
            }
            if ((threadIdx.x & 63) == 0 && bits) {
                if (use_lds) atomicMax(&tab[(int)(o - o_base)], bits);
                else atomicMax(&result[o], bits);
            }
        } else if (bits) {
            if (use_lds) atomicMax(&tab[(int)(o - o_base)], bits);
            else atomicMax(&result[o], bits);
        }
    }
    if (use_lds) {
        __syncthreads();
        for (int t = threadIdx.x; t < span; t += kBlock) {
            const uint32_t v = tab[t];
            if (v) atomicMax(&result[o_base + (IT)t], v);
        }
    }
}

// ------------------------------------------------------------------------------------------
//  Integer-view statistics for the tracking callbacks (custom_callbacks.py:85-96, 131-207): range and
//  histogram of q = floor(P/s).  Integer atomics only -> exact and independent of arrival order.
// ------------------------------------------------------------------------------------------
__device__ __forceinline__ bool q_as_int(float x, float sg, int32_t& qi) {
    const float q = floorf(x / sg);
    if (!(fabsf(q) < 2147483520.0f)) return false;   // NaN / Inf / beyond int32: not counted
    qi = (int32_t)q;
    return true;
}

// IT = uint32_t below 2^31 elements: (i / inner) % G in 32-bit arithmetic (the 64-bit division expansion dominated the loop)
template <typename IT>
__global__ __launch_bounds__(kBlock) void k_q_minmax(const float* __restrict__ P, const float* __restrict__ s, int32_t* minmax,
                                                     int64_t n64, int64_t G64, int64_t inner64) {
    const IT n = (IT)n64, G = (IT)G64, inner = (IT)inner64;
    int32_t lo = INT32_MAX, hi = INT32_MIN;
    // four independent (element, scale) loads in flight per thread; indices past the end are clamped and not counted
    const IT stride = (IT)gridDim.x * kBlock;
    for (IT i = (IT)blockIdx.x * kBlock + threadIdx.x; i < n; i += 4 * stride) {
        float x[4], sg[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const IT j = (n - i > (IT)u * stride) ? i + (IT)u * stride : i;
            x[u] = P[j];
            sg[u] = s[(j / inner) % G];
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            int32_t qi;
            if (n - i > (IT)u * stride && q_as_int(x[u], sg[u], qi)) {
                lo = qi < lo ? qi : lo;
                hi = qi > hi ? qi : hi;
            }
        }
    }
    for (int off = 32; off > 0; off >>= 1) {
        const int32_t l2 = __shfl_xor(lo, off, 64), h2 = __shfl_xor(hi, off, 64);
        lo = l2 < lo ? l2 : lo;
        hi = h2 > hi ? h2 : hi;
    }
    // one pair of atomics per BLOCK: every wave hitting the same two words serialises at ~12 ns per atomic
    // (16 K waves used to cost 100-400 us here)
    __shared__ int32_t wlo[kBlock / 64], whi[kBlock / 64];
    if ((threadIdx.x & 63) == 0) {
        wlo[threadIdx.x >> 6] = lo;
        whi[threadIdx.x >> 6] = hi;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        for (int w = 1; w < kBlock / 64; ++w) {
            lo = wlo[w] < lo ? wlo[w] : lo;
            hi = whi[w] > hi ? whi[w] : hi;
        }
        // the range only widens: a block whose extreme does not beat the value it can currently see has nothing to add
        // (a stale read only costs a redundant atomic), so most blocks skip the serialised same-address update
        volatile int32_t* cur = minmax;
        if (lo != INT32_MAX && lo < cur[0]) atomicMin(&minmax[0], lo);
        if (hi != INT32_MIN && hi > cur[1]) atomicMax(&minmax[1], hi);
    }
}

constexpr int kHistLds = 4096;   // bins privatised in LDS per block (the trained models use a few dozen integers)

template <typename IT>
__global__ __launch_bounds__(kBlock) void k_q_histogram(const float* __restrict__ P, const float* __restrict__ s, int32_t qmin,
                                                        int64_t nbins, uint32_t* bins, int64_t n64, int64_t G64, int64_t inner64) {
    const IT n = (IT)n64, G = (IT)G64, inner = (IT)inner64;
    __shared__ uint32_t lh[kHistLds];
    const bool priv = nbins <= kHistLds;
    if (priv) {
        for (int b = threadIdx.x; b < (int)nbins; b += kBlock) lh[b] = 0u;
        __syncthreads();
    }
    const IT stride = (IT)gridDim.x * kBlock;
    for (IT i = (IT)blockIdx.x * kBlock + threadIdx.x; i < n; i += 4 * stride) {
        float x[4], sg[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const IT j = (n - i > (IT)u * stride) ? i + (IT)u * stride : i;
            x[u] = P[j];
            sg[u] = s[(j / inner) % G];
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            int32_t qi;
            if (!(n - i > (IT)u * stride) || !q_as_int(x[u], sg[u], qi)) continue;
            const int64_t b = (int64_t)qi - (int64_t)qmin;
            if (b < 0 || b >= nbins) continue;
            if (priv) atomicAdd(&lh[b], 1u);
            else atomicAdd(&bins[b], 1u);
        }
    }
    if (priv) {
        __syncthreads();
        for (int b = threadIdx.x; b < (int)nbins; b += kBlock) {
            const uint32_t c = lh[b];
            if (c) atomicAdd(&bins[b], c);
        }
    }
}

// ------------------------------------------------------------------------------------------
//  Device self-test of window_div against the IEEE division (random in-window operand pairs).
// ------------------------------------------------------------------------------------------
__global__ __launch_bounds__(kBlock) void k_selftest_ratio_div(uint64_t seed, uint32_t per_thread, unsigned long long* mismatches) {
    uint64_t x = seed ^ (0x9E3779B97F4A7C15ull * ((uint64_t)blockIdx.x * kBlock + threadIdx.x + 1));
    unsigned long long bad = 0;
    for (uint32_t k = 0; k < per_thread; ++k) {
        x ^= x << 13;
        x ^= x >> 7;
        x ^= x << 17;
        // exponents 87..167 (2^-40 .. 2^40), random mantissas; the top of the window is clamped to exactly 2^40
        uint32_t ea = 87u + (uint32_t)((x >> 8) % 81u), eb = 87u + (uint32_t)((x >> 24) % 81u);
        uint32_t ma = (uint32_t)(x >> 40) & 0x7fffffu, mb = (uint32_t)(x * 0x2545F4914F6CDD1Dull >> 41) & 0x7fffffu;
        if (ea == 167u) ma = 0u;
        if (eb == 167u) mb = 0u;
        const float a = __uint_as_float((ea << 23) | ma), b = __uint_as_float((eb << 23) | mb);
        const float want = a / b;
        const float got = window_div(a, b);
        bad += (__float_as_uint(want) != __float_as_uint(got)) ? 1ull : 0ull;
    }
    if (bad) atomicAdd(mismatches, bad);
}

// Same for the uniform-divisor division of the streaming kernels: x / s with r = RN(1/s), 2^-40 <= s <= 2^40
// (mantissa not all ones), 2^-80 <= |x| < 2^81, random signs.
__global__ __launch_bounds__(kBlock) void k_selftest_uniform_div(uint64_t seed, uint32_t per_thread, unsigned long long* mismatches) {
    uint64_t x = seed ^ (0xD1B54A32D192ED03ull * ((uint64_t)blockIdx.x * kBlock + threadIdx.x + 1));
    unsigned long long bad = 0;
    for (uint32_t k = 0; k < per_thread; ++k) {
        x ^= x << 13;
        x ^= x >> 7;
        x ^= x << 17;
        const uint32_t es = 87u + (uint32_t)((x >> 8) % 81u), ex = 47u + (uint32_t)((x >> 24) % 161u);
        uint32_t ms = (uint32_t)(x >> 40) & 0x7fffffu;
        const uint32_t mx = (uint32_t)(x * 0x2545F4914F6CDD1Dull >> 41) & 0x7fffffu;
        if (ms == 0x7fffffu) ms = 0x7ffffeu;
        if (es == 167u) ms = 0u;
        Ctx c;
        c.s = __uint_as_float((es << 23) | ms);
        div_ctx(c);
        const float xv = __uint_as_float((uint32_t)((x >> 63) << 31) | (ex << 23) | mx);
        const float want = xv / c.s;
        const float got = c.fast ? fast_div(xv, c.s, c.r) : want;
        bad += (__float_as_uint(want) != __float_as_uint(got)) ? 1ull : 0ull;
        bad += c.fast ? 0ull : 1ull;      // every generated divisor must be inside the fast window
    }
    if (bad) atomicAdd(mismatches, bad);
}

}  // namespace lq

#endif
