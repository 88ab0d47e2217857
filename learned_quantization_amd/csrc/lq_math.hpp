// lq_math.hpp -- exact fp32 arithmetic of the path: uniform-divisor division, in-window ratio division, |tanh|, vote
#ifndef LQ_MATH_HPP_
#define LQ_MATH_HPP_
#include "lq_common.hpp"

namespace lq {

// ------------------------------------------------------------------------------------------
//  x / s for a divisor that is uniform over the block: correctly rounded, ~8 VALU instead of the
//  ~15-instruction v_div_scale / v_rcp / v_div_fmas / v_div_fixup sequence.
//    r = RN(1/s);  q0 = RN(x r);  e0 = x - s q0 (exact, fma);  q1 = RN(q0 + e0 r);
//    e1 = x - s q1 (exact);  t = RN(q1 + e1 r)
//  By Markstein's theorem the last step rounds correctly when r is the correctly rounded
//  reciprocal and q1 is within 1 ulp, for every s whose mantissa is not all ones.  The window
//  (2^-40 <= s <= 2^40, 2^-80 <= |x| < 2^81) keeps every intermediate normal so that the
//  residuals are exact; anything outside (zeros, denormals, Inf, NaN, huge, s <= 0) takes the IEEE
//  `/`.  tests/tools/check_fast_div.c checks the sequence against `/` for all 2^23 mantissas of x
//  per divisor; tests/test_gpu_parity.py checks the kernels bit for bit against the oracle.
// ------------------------------------------------------------------------------------------
__device__ __forceinline__ void div_ctx(Ctx& c) {
    const uint32_t sb = __float_as_uint(c.s);
    const uint32_t ex = (sb >> 23) & 0xffu;
    const bool ok = (sb >> 31) == 0u && ex >= 127u - 40u && ex <= 127u + 40u && (sb & 0x7fffffu) != 0x7fffffu;
    c.r = 1.0f / c.s;
    c.fast = ok ? 1 : 0;
}

__device__ __forceinline__ float div_by_uniform(float x, const Ctx& c) {
    if (c.fast) {   // block/row-uniform
        const uint32_t ex = (__float_as_uint(x) >> 23) & 0xffu;
        if (__builtin_expect((ex - 47u) <= 160u, 1)) {
            const float q0 = x * c.r;
            const float e0 = __builtin_fmaf(-c.s, q0, x);
            const float q1 = __builtin_fmaf(e0, c.r, q0);
            const float e1 = __builtin_fmaf(-c.s, q1, x);
            return __builtin_fmaf(e1, c.r, q1);
        }
        if (x == 0.0f) return x;   // (+-0) / s = +-0 for s > 0
    }
    return x / c.s;                // IEEE RN fp32 division (hipcc default: correctly rounded)
}

// |tanh(d)| for d = lambda - ratio, 0 < d <= lambda (or NaN).
//   tmode 0 (lambda < 4e-4): tanh(d) == d to fp32 precision (d^2/3 < 2^-24) -- every published
//           threshold (lambda <= 1e-8) is here;
//   tmode 1 (lambda <= 0.25): odd minimax polynomial, < 1 ulp on [0, 0.25];
//   tmode 2: ocml tanhf.
template <int TM>
__device__ __forceinline__ float abs_tanh_t(float d) {
    const float a = fabsf(d);
    if (TM == 0) return a;
    if (TM == 1) {
        const float z = a * a;
        float p = 2.0800685256e-02f;
        p = __builtin_fmaf(p, z, -5.3927052600e-02f);
        p = __builtin_fmaf(p, z, 1.3333282305e-01f);
        p = __builtin_fmaf(p, z, -3.3333333236e-01f);
        return __builtin_fmaf(a * z, p, a);
    }
    return a < 4.0e-4f ? a : tanhf(a);
}

__device__ __forceinline__ float abs_tanh(float d, int tmode) {
    if (tmode == 0) return abs_tanh_t<0>(d);
    if (tmode == 1) return abs_tanh_t<1>(d);
    return abs_tanh_t<2>(d);
}

__device__ __forceinline__ void fq_core(float x, const Ctx& c, float& q, float& o) {
    const float t = div_by_uniform(x, c);   // custom_layers.py:56-58
    q = floorf(t);                           // :59
    o = q * c.s;                             // :60
}

__device__ __forceinline__ void nq_accumulate(float q, float o, float dy, float lam, int tmode, Acc& acc) {
    const float nz = (o == 0.0f) ? kEpsF32 : o;        // :63
    const float a = fabsf(dy), b = fabsf(nz);
    acc.a = __float_as_uint(fmaxf(__uint_as_float(acc.a), fabsf(q)));   // :68 / :94
    const float ratio = a / b;                         // :64
    if (!(ratio >= lam)) {                             // :70 / :97 (NaN counts as "not above")
        acc.b += 1u;
        acc.c -= (double)abs_tanh(lam - ratio, tmode); // :84 / :110 (f64 sum: exact for tmode 0, see Acc)
    }
}

// ---- float4 forms: branch-light.  One (rarely taken) branch for the division window, one for
// "does any of the 4 elements need the exact ratio", everything else straight-line VALU.
__device__ __forceinline__ float fast_div(float x, float s, float r) {
    const float q0 = x * r;
    const float e0 = __builtin_fmaf(-s, q0, x);
    const float q1 = __builtin_fmaf(e0, r, q0);
    const float e1 = __builtin_fmaf(-s, q1, x);
    return __builtin_fmaf(e1, r, q1);
}

__device__ __forceinline__ void fq_core4(const float4& x, const Ctx& c, float4& q, float4& o) {
    // window 2^-80 <= |x| < 2^81 for all four, via min3/max3 with |.| source modifiers.  fminf/fmaxf
    // ignore a NaN operand, which is harmless: a NaN x gives a NaN quotient on the fast path too; an
    // all-NaN group fails the comparison and takes the IEEE path.  Inf and 0 fail the window.
    const float amax = fmaxf(fmaxf(fabsf(x.x), fabsf(x.y)), fmaxf(fabsf(x.z), fabsf(x.w)));
    const float amin = fminf(fminf(fabsf(x.x), fabsf(x.y)), fminf(fabsf(x.z), fabsf(x.w)));
    float4 t;
    if (__builtin_expect((c.fast != 0) & (amin >= 8.271806125530277e-25f) & (amax < 2.4178516392292583e+24f), 1)) {
        t.x = fast_div(x.x, c.s, c.r);
        t.y = fast_div(x.y, c.s, c.r);
        t.z = fast_div(x.z, c.s, c.r);
        t.w = fast_div(x.w, c.s, c.r);
    } else {   // zeros, denormals, Inf, huge, or a divisor outside the window: IEEE division
        t.x = x.x / c.s;
        t.y = x.y / c.s;
        t.z = x.z / c.s;
        t.w = x.w / c.s;
    }
    q.x = floorf(t.x); q.y = floorf(t.y); q.z = floorf(t.z); q.w = floorf(t.w);
    o.x = q.x * c.s; o.y = q.y * c.s; o.z = q.z * c.s; o.w = q.w * c.s;
}

__device__ __forceinline__ void vote_ctx(Ctx& c, float lam) {
    c.lam_hi = lam * 1.000001f;
    const float bmin = fminf(fabsf(c.s), kEpsF32);      // b = |q*s| >= |s| when q != 0, else eps (:63)
    c.sure_ok = (lam == 0.0f || c.lam_hi * bmin >= 1.0e-30f) ? 1 : 0;
}

// exact vote of one element (IEEE ratio): custom_layers.py:64, :70/:97, :84/:110
// ------------------------------------------------------------------------------------------
//  a / b for per-element operands, bit-identical to the IEEE `/` inside a window.  hipcc expands `/` to
//     d' = v_div_scale(b); n' = v_div_scale(a); r0 = v_rcp(d'); e = fma(-d',r0,1); r = fma(e,r0,r0);
//     q0 = n'*r; e1 = fma(-d',q0,n'); q1 = fma(e1,r,q0); e2 = fma(-d',q1,n'); q = v_div_fmas(e2,r,q1); v_div_fixup
//  For 2^-40 <= a,b <= 2^40 the two v_div_scale are identities (exponent difference < 96, no denormals),
//  v_div_fmas is a plain fma and v_div_fixup changes nothing (no NaN/Inf/0 operands), so the SAME
//  rcp + fma chain without them gives the same bits -- and, being plain fma/mul, is packed two-wide
//  (v_pk_fma_f32) across the elements of a float4.  Verified on the device against `/` by
//  lq_selftest_ratio_division (tests/test_gpu_parity.py) on 2^33 random in-window pairs.
// ------------------------------------------------------------------------------------------
__device__ __forceinline__ float window_div(float a, float b) {
    const float r0 = __builtin_amdgcn_rcpf(b);
    const float e = __builtin_fmaf(-b, r0, 1.0f);
    const float r = __builtin_fmaf(e, r0, r0);
    const float q0 = a * r;
    const float e1 = __builtin_fmaf(-b, q0, a);
    const float q1 = __builtin_fmaf(e1, r, q0);
    const float e2 = __builtin_fmaf(-b, q1, a);
    return __builtin_fmaf(e2, r, q1);
}
constexpr float kWinLo = 9.094947017729282e-13f;   // 2^-40
constexpr float kWinHi = 1.099511627776e+12f;      // 2^40

// exact vote of one element given its IEEE ratio: custom_layers.py:70/:97, :84/:110
template <int TM>
__device__ __forceinline__ void vote_tally(float ratio, float lam, Acc& acc) {
    const bool below = !(ratio >= lam);                  // NaN counts as "not above"
    acc.b += below ? 1u : 0u;
    const float t = abs_tanh_t<TM>(lam - ratio);
    acc.c -= (double)(below ? t : 0.0f);                 // f64 sum: exact for TM == 0, see Acc
}

template <int TM>
__device__ __forceinline__ void vote_cast(float a, float b, float lam, Acc& acc) {
    vote_tally<TM>(a / b, lam, acc);                     // :64 (IEEE)
}

template <int TM>
__device__ __forceinline__ void vote_cast4(const float4& dy, float b0, float b1, float b2, float b3, float lam, Acc& acc) {
    const float a0 = fabsf(dy.x), a1 = fabsf(dy.y), a2 = fabsf(dy.z), a3 = fabsf(dy.w);
    const float lo = fminf(fminf(fminf(a0, a1), fminf(a2, a3)), fminf(fminf(b0, b1), fminf(b2, b3)));
    const float hi = fmaxf(fmaxf(fmaxf(a0, a1), fmaxf(a2, a3)), fmaxf(fmaxf(b0, b1), fmaxf(b2, b3)));
    // fminf/fmaxf skip a NaN operand: test every operand for NaN through one sum (NaN propagates through +)
    const float nan_probe = (a0 + a1) + (a2 + a3) + ((b0 + b1) + (b2 + b3));
    float r0, r1, r2, r3;
    if ((lo >= kWinLo) & (hi <= kWinHi) & (nan_probe == nan_probe)) {
        r0 = window_div(a0, b0);
        r1 = window_div(a1, b1);
        r2 = window_div(a2, b2);
        r3 = window_div(a3, b3);
    } else {
        r0 = a0 / b0;
        r1 = a1 / b1;
        r2 = a2 / b2;
        r3 = a3 / b3;
    }
    vote_tally<TM>(r0, lam, acc);
    vote_tally<TM>(r1, lam, acc);
    vote_tally<TM>(r2, lam, acc);
    vote_tally<TM>(r3, lam, acc);
}

__device__ __forceinline__ void nq_accumulate4(const float4& q, const float4& o, const float4& dy, const Ctx& c, float lam,
                                               int tmode, Acc& acc) {
    // max|q| as a float max with |.| modifiers (q is integer-valued: exact).  A NaN q is ignored here, but
    // then out is NaN -> ratio NaN -> the vote sum and ds are NaN anyway.
    const float mq = fmaxf(fmaxf(fabsf(q.x), fabsf(q.y)), fmaxf(fabsf(q.z), fabsf(q.w)));
    acc.a = __float_as_uint(fmaxf(__uint_as_float(acc.a), mq));
    const float b0 = (o.x == 0.0f) ? kEpsF32 : fabsf(o.x);   // :63
    const float b1 = (o.y == 0.0f) ? kEpsF32 : fabsf(o.y);
    const float b2 = (o.z == 0.0f) ? kEpsF32 : fabsf(o.z);
    const float b3 = (o.w == 0.0f) ? kEpsF32 : fabsf(o.w);
    // ratio >= lambda is certain when |dy| >= RN(lam_hi*b): a/b >= lambda(1+8e-7) > lambda and RN is monotonic.
    // Such elements contribute nothing to the vote (:82/:108); only if some lane has an uncertain element
    // are the four IEEE divisions done (a sure element then simply evaluates to "not below").
    const bool all_sure = (c.sure_ok != 0) & (fabsf(dy.x) >= c.lam_hi * b0) & (fabsf(dy.y) >= c.lam_hi * b1) &
                          (fabsf(dy.z) >= c.lam_hi * b2) & (fabsf(dy.w) >= c.lam_hi * b3);
    if (!all_sure) {
        if (tmode == 0) vote_cast4<0>(dy, b0, b1, b2, b3, lam, acc);        // kernel-uniform
        else if (tmode == 1) vote_cast4<1>(dy, b0, b1, b2, b3, lam, acc);
        else vote_cast4<2>(dy, b0, b1, b2, b3, lam, acc);
    }
}

// ---- float4 whose four elements belong to four different groups (column traversals): the same branch-light
// forms with one context and one accumulator per element.
__device__ __forceinline__ void fq_core4c(const float4& x, const Ctx* c, float4& q, float4& o) {
    const float amax = fmaxf(fmaxf(fabsf(x.x), fabsf(x.y)), fmaxf(fabsf(x.z), fabsf(x.w)));
    const float amin = fminf(fminf(fabsf(x.x), fabsf(x.y)), fminf(fabsf(x.z), fabsf(x.w)));
    const int fast = c[0].fast & c[1].fast & c[2].fast & c[3].fast;
    float4 t;
    if (__builtin_expect((fast != 0) & (amin >= 8.271806125530277e-25f) & (amax < 2.4178516392292583e+24f), 1)) {
        t.x = fast_div(x.x, c[0].s, c[0].r);
        t.y = fast_div(x.y, c[1].s, c[1].r);
        t.z = fast_div(x.z, c[2].s, c[2].r);
        t.w = fast_div(x.w, c[3].s, c[3].r);
    } else {
        t.x = x.x / c[0].s;
        t.y = x.y / c[1].s;
        t.z = x.z / c[2].s;
        t.w = x.w / c[3].s;
    }
    q.x = floorf(t.x); q.y = floorf(t.y); q.z = floorf(t.z); q.w = floorf(t.w);
    o.x = q.x * c[0].s; o.y = q.y * c[1].s; o.z = q.z * c[2].s; o.w = q.w * c[3].s;
}

template <int TM>
__device__ __forceinline__ void vote_cast4c(const float4& dy, float b0, float b1, float b2, float b3, float lam, Acc* acc) {
    const float a0 = fabsf(dy.x), a1 = fabsf(dy.y), a2 = fabsf(dy.z), a3 = fabsf(dy.w);
    const float lo = fminf(fminf(fminf(a0, a1), fminf(a2, a3)), fminf(fminf(b0, b1), fminf(b2, b3)));
    const float hi = fmaxf(fmaxf(fmaxf(a0, a1), fmaxf(a2, a3)), fmaxf(fmaxf(b0, b1), fmaxf(b2, b3)));
    const float nan_probe = (a0 + a1) + (a2 + a3) + ((b0 + b1) + (b2 + b3));
    float r0, r1, r2, r3;
    if ((lo >= kWinLo) & (hi <= kWinHi) & (nan_probe == nan_probe)) {
        r0 = window_div(a0, b0);
        r1 = window_div(a1, b1);
        r2 = window_div(a2, b2);
        r3 = window_div(a3, b3);
    } else {
        r0 = a0 / b0;
        r1 = a1 / b1;
        r2 = a2 / b2;
        r3 = a3 / b3;
    }
    vote_tally<TM>(r0, lam, acc[0]);
    vote_tally<TM>(r1, lam, acc[1]);
    vote_tally<TM>(r2, lam, acc[2]);
    vote_tally<TM>(r3, lam, acc[3]);
}

__device__ __forceinline__ void nq_accumulate4c(const float4& q, const float4& o, const float4& dy, const Ctx* c, float lam,
                                                int tmode, Acc* acc) {
    acc[0].a = __float_as_uint(fmaxf(__uint_as_float(acc[0].a), fabsf(q.x)));
    acc[1].a = __float_as_uint(fmaxf(__uint_as_float(acc[1].a), fabsf(q.y)));
    acc[2].a = __float_as_uint(fmaxf(__uint_as_float(acc[2].a), fabsf(q.z)));
    acc[3].a = __float_as_uint(fmaxf(__uint_as_float(acc[3].a), fabsf(q.w)));
    const float b0 = (o.x == 0.0f) ? kEpsF32 : fabsf(o.x);   // :63
    const float b1 = (o.y == 0.0f) ? kEpsF32 : fabsf(o.y);
    const float b2 = (o.z == 0.0f) ? kEpsF32 : fabsf(o.z);
    const float b3 = (o.w == 0.0f) ? kEpsF32 : fabsf(o.w);
    const float lh = c[0].lam_hi;                             // lambda-only: the same in all four contexts
    const bool all_sure = ((c[0].sure_ok & c[1].sure_ok & c[2].sure_ok & c[3].sure_ok) != 0) & (fabsf(dy.x) >= lh * b0) &
                          (fabsf(dy.y) >= lh * b1) & (fabsf(dy.z) >= lh * b2) & (fabsf(dy.w) >= lh * b3);
    if (!all_sure) {
        if (tmode == 0) vote_cast4c<0>(dy, b0, b1, b2, b3, lam, acc);
        else if (tmode == 1) vote_cast4c<1>(dy, b0, b1, b2, b3, lam, acc);
        else vote_cast4c<2>(dy, b0, b1, b2, b3, lam, acc);
    }
}

// ---- float4 whose four elements belong to at most TWO adjacent groups, A then B: elements 0 .. nA-1 are A's (nA = 1..4).  The
// column traversals of lq_batch_cols.hpp use it where `inner` >= 2: two contexts and two accumulators per lane instead of four
// (12 VGPRs less than the per-column form), the element's group picked by a lane mask.  Per-group totals are those of the
// per-column form: max and counts are order-free, the vote sums exact for tmode 0 (lq_common.hpp Acc).
struct Ctx2 {
    float sA, rA, sB, rB;
    float lam_hi;
    int ok;       // bit 0: both divisors inside the fast-division window; bit 1: both `sure_ok`
};

__device__ __forceinline__ void fq_core4g(const float4& x, const Ctx2& c, bool m1, bool m2, bool m3, float4& q, float4& o) {
    const float s0 = c.sA, s1 = m1 ? c.sB : c.sA, s2 = m2 ? c.sB : c.sA, s3 = m3 ? c.sB : c.sA;
    const float amax = fmaxf(fmaxf(fabsf(x.x), fabsf(x.y)), fmaxf(fabsf(x.z), fabsf(x.w)));
    const float amin = fminf(fminf(fabsf(x.x), fabsf(x.y)), fminf(fabsf(x.z), fabsf(x.w)));
    float4 t;
    if (__builtin_expect(((c.ok & 1) != 0) & (amin >= 8.271806125530277e-25f) & (amax < 2.4178516392292583e+24f), 1)) {
        const float r1 = m1 ? c.rB : c.rA, r2 = m2 ? c.rB : c.rA, r3 = m3 ? c.rB : c.rA;
        t.x = fast_div(x.x, s0, c.rA);
        t.y = fast_div(x.y, s1, r1);
        t.z = fast_div(x.z, s2, r2);
        t.w = fast_div(x.w, s3, r3);
    } else {
        t.x = x.x / s0;
        t.y = x.y / s1;
        t.z = x.z / s2;
        t.w = x.w / s3;
    }
    q.x = floorf(t.x); q.y = floorf(t.y); q.z = floorf(t.z); q.w = floorf(t.w);
    o.x = q.x * s0; o.y = q.y * s1; o.z = q.z * s2; o.w = q.w * s3;
}

template <int TM>
__device__ __forceinline__ void vote_tally_g(float ratio, float lam, bool m, Acc& A, Acc& B) {
    const bool below = !(ratio >= lam);                  // NaN counts as "not above"
    const float t = below ? abs_tanh_t<TM>(lam - ratio) : 0.0f;
    A.b += (below & !m) ? 1u : 0u;
    B.b += (below & m) ? 1u : 0u;
    A.c -= (double)(m ? 0.0f : t);
    B.c -= (double)(m ? t : 0.0f);
}

template <int TM>
__device__ __forceinline__ void vote_cast4g(const float4& dy, float b0, float b1, float b2, float b3, float lam, bool m1, bool m2, bool m3,
                                            Acc& A, Acc& B) {
    const float a0 = fabsf(dy.x), a1 = fabsf(dy.y), a2 = fabsf(dy.z), a3 = fabsf(dy.w);
    const float lo = fminf(fminf(fminf(a0, a1), fminf(a2, a3)), fminf(fminf(b0, b1), fminf(b2, b3)));
    const float hi = fmaxf(fmaxf(fmaxf(a0, a1), fmaxf(a2, a3)), fmaxf(fmaxf(b0, b1), fmaxf(b2, b3)));
    const float nan_probe = (a0 + a1) + (a2 + a3) + ((b0 + b1) + (b2 + b3));
    float r0, r1, r2, r3;
    if ((lo >= kWinLo) & (hi <= kWinHi) & (nan_probe == nan_probe)) {
        r0 = window_div(a0, b0);
        r1 = window_div(a1, b1);
        r2 = window_div(a2, b2);
        r3 = window_div(a3, b3);
    } else {
        r0 = a0 / b0;
        r1 = a1 / b1;
        r2 = a2 / b2;
        r3 = a3 / b3;
    }
    vote_tally_g<TM>(r0, lam, false, A, B);
    vote_tally_g<TM>(r1, lam, m1, A, B);
    vote_tally_g<TM>(r2, lam, m2, A, B);
    vote_tally_g<TM>(r3, lam, m3, A, B);
}

__device__ __forceinline__ void nq_accumulate4g(const float4& q, const float4& o, const float4& dy, const Ctx2& c, float lam, int tmode,
                                                bool m1, bool m2, bool m3, Acc& A, Acc& B) {
    const float q0 = fabsf(q.x), q1 = fabsf(q.y), q2 = fabsf(q.z), q3 = fabsf(q.w);
    const float mA = fmaxf(fmaxf(q0, m1 ? 0.0f : q1), fmaxf(m2 ? 0.0f : q2, m3 ? 0.0f : q3));
    const float mB = fmaxf(m1 ? q1 : 0.0f, fmaxf(m2 ? q2 : 0.0f, m3 ? q3 : 0.0f));
    A.a = __float_as_uint(fmaxf(__uint_as_float(A.a), mA));
    B.a = __float_as_uint(fmaxf(__uint_as_float(B.a), mB));
    const float b0 = (o.x == 0.0f) ? kEpsF32 : fabsf(o.x);   // :63
    const float b1 = (o.y == 0.0f) ? kEpsF32 : fabsf(o.y);
    const float b2 = (o.z == 0.0f) ? kEpsF32 : fabsf(o.z);
    const float b3 = (o.w == 0.0f) ? kEpsF32 : fabsf(o.w);
    const float lh = c.lam_hi;
    const bool all_sure = ((c.ok & 2) != 0) & (fabsf(dy.x) >= lh * b0) & (fabsf(dy.y) >= lh * b1) & (fabsf(dy.z) >= lh * b2) &
                          (fabsf(dy.w) >= lh * b3);
    if (!all_sure) {
        if (tmode == 0) vote_cast4g<0>(dy, b0, b1, b2, b3, lam, m1, m2, m3, A, B);
        else if (tmode == 1) vote_cast4g<1>(dy, b0, b1, b2, b3, lam, m1, m2, m3, A, B);
        else vote_cast4g<2>(dy, b0, b1, b2, b3, lam, m1, m2, m3, A, B);
    }
}

template <int Q>
__device__ __forceinline__ void store_q_scalar(void* qp, int64_t i, float q) {
    if (Q == LQ_Q_F32) {
        reinterpret_cast<float*>(qp)[i] = q;
    } else if (Q == LQ_Q_I32) {
        reinterpret_cast<int32_t*>(qp)[i] = (q != q) ? 0 : (q >= 2147483648.0f ? INT32_MAX : (q <= -2147483648.0f ? INT32_MIN : (int32_t)q));
    } else if (Q == LQ_Q_I8) {
        // two's-complement wrap of the (finite) integer value: what a C cast chain float->int64->int8 gives
        long long w = (q != q || fabsf(q) > 9.0e18f) ? 0ll : (long long)q;
        reinterpret_cast<int8_t*>(qp)[i] = (int8_t)(uint8_t)(w & 0xff);
    }
}

__device__ __forceinline__ void store_q(void* qp, int q_dtype, int64_t i, float q) {
    switch (q_dtype) {
        case LQ_Q_F32: store_q_scalar<LQ_Q_F32>(qp, i, q); break;
        case LQ_Q_I32: store_q_scalar<LQ_Q_I32>(qp, i, q); break;
        case LQ_Q_I8: store_q_scalar<LQ_Q_I8>(qp, i, q); break;
        default: break;
    }
}

}  // namespace lq

#endif
