// lq_ops.hpp -- per-operation traits (what one element contributes), consumed by every traversal
#ifndef LQ_OPS_HPP_
#define LQ_OPS_HPP_
#include "lq_math.hpp"

namespace lq {

// ------------------------------------------------------------------------------------------
//  Op traits.  elem() consumes one element (x, dy) of group context c at flat index i, returns
//  the value of the primary dense output (stored, vectorised, by the traversal when kStore)
//  and folds into acc when kReduce.
// ------------------------------------------------------------------------------------------
template <int OP>
struct OpT;

struct OpBase {
    static constexpr bool kStdMerge = true;   // merge is (max, add, add): DPP reduction applies
    static constexpr bool kVec4 = false;      // op provides elem4()
    static constexpr bool kVec4c = false;     // op provides elem4c(): float4 with one context/accumulator per element
    static constexpr bool kDy = false;
    static constexpr bool kStore = false;
    static constexpr bool kReduce = false;
    template <typename A>
    __device__ static __forceinline__ A init() {
        A a;
        a.a = 0u;
        a.b = 0;
        a.c = 0;
        return a;
    }
    template <typename A, typename B>
    __device__ static __forceinline__ void merge(A& x, const B& y) {
        x.a = y.a > x.a ? y.a : x.a;
        x.b += y.b;
        x.c += y.c;
    }
    __device__ static __forceinline__ Ctx ctx_of(const Params& p, float s) {      // context of a scale that is already in a register
        Ctx c;
        c.s = s;
        div_ctx(c);
        c.k0 = 0.f;
        c.k1 = 0.f;
        vote_ctx(c, p.lam);
        return c;
    }
    __device__ static __forceinline__ Ctx ctx(const Params& p, int64_t g) { return ctx_of(p, p.s[g]); }
};

// PERM = true: the variant that can also emit the OIHW companion of an HWIO conv kernel ELEMENT BY ELEMENT -- the fallback
// for the conv kernels lq_conv_tile.hpp does not take (more than 9 taps: a 7x7 stem; co % 4 != 0; exotic descriptors).  The
// companion costs two integer divisions and a 4-byte scattered store per element, so these instantiations provide the
// scalar element path only (kVec4 = false: the traversals still load float4, the op walks its elements).
template <bool PERM>
struct FwdOp : OpBase {
    static constexpr bool kStore = true;
    __device__ static __forceinline__ void side4(const Params& p, int64_t i, const float4& q, const float4& o) {
        if (p.q) {
            store_q(p.q, p.q_dtype, i + 0, q.x);
            store_q(p.q, p.q_dtype, i + 1, q.y);
            store_q(p.q, p.q_dtype, i + 2, q.z);
            store_q(p.q, p.q_dtype, i + 3, q.w);
        }
    }
    __device__ static __noinline__ void scatter1(float* __restrict__ out_perm, uint32_t hw, uint32_t ci, uint32_t co, uint32_t i, float o) {
        const uint32_t t = i / co, oc = i - t * co;
        const uint32_t h = t / ci, c = t - h * ci;
        out_perm[(oc * ci + c) * hw + h] = o;
    }
    __device__ static __forceinline__ float elem(const Params& p, const Ctx& c, int64_t i, float x, float, Acc&) {
        float q, o;
        fq_core(x, c, q, o);
        if (p.q) store_q(p.q, p.q_dtype, i, q);
        if constexpr (PERM) {
            if (p.out_perm) scatter1(p.out_perm, p.perm_hw, p.perm_ci, p.perm_co, (uint32_t)i, o);
        }
        return o;
    }
    static constexpr bool kVec4 = !PERM;
    __device__ static __forceinline__ float4 elem4(const Params& p, const Ctx& c, int64_t i, const float4& x, const float4&, Acc&) {
        float4 q, o;
        fq_core4(x, c, q, o);
        side4(p, i, q, o);
        return o;
    }
    static constexpr bool kVec4c = !PERM;
    __device__ static __forceinline__ float4 elem4c(const Params& p, const Ctx* c, int64_t i, const float4& x, const float4&, Acc*) {
        float4 q, o;
        fq_core4c(x, c, q, o);
        side4(p, i, q, o);
        return o;
    }
};
template <>
struct OpT<OP_FWD> : FwdOp<false> {};
template <>
struct OpT<OP_FWD_PERM> : FwdOp<true> {};

template <>
struct OpT<OP_QONLY> : OpBase {
    __device__ static __forceinline__ float elem(const Params& p, const Ctx& c, int64_t i, float x, float, Acc&) {
        float q, o;
        fq_core(x, c, q, o);
        store_q(p.q, p.q_dtype, i, q);
        return 0.f;
    }
};

template <bool PERM>
struct BwdOp : OpBase {
    static constexpr bool kDy = true;
    static constexpr bool kReduce = true;
    // upstream gradient of an HWIO conv kernel that arrives in OIHW order (MIOpen's weight gradient): gathered here, and
    // handed back in HWIO order as dP (dP == dy, custom_layers.py:118) -- weight-sized tensors, L2-resident
    // out of line: the fallback serves a 7x7 stem or an odd head, its divisions need not sit in every traversal's registers
    __device__ static __noinline__ float gather1(const float* __restrict__ dy_perm, float* __restrict__ dp_out, uint32_t hw, uint32_t ci,
                                                 uint32_t co, uint32_t i) {
        const uint32_t t = i / co, o = i - t * co;
        const uint32_t h = t / ci, c = t - h * ci;
        const float d = dy_perm[(o * ci + c) * hw + h];
        dp_out[i] = d;
        return d;
    }
    __device__ static __forceinline__ float elem(const Params& p, const Ctx& c, int64_t i, float x, float dy, Acc& acc) {
        float q, o;
        fq_core(x, c, q, o);
        if constexpr (PERM) {
            if (p.dy_perm) dy = gather1(p.dy_perm, p.dp_out, p.perm_hw, p.perm_ci, p.perm_co, (uint32_t)i);
        }
        nq_accumulate(q, o, dy, p.lam, p.tmode, acc);
        return 0.f;
    }
    static constexpr bool kVec4 = !PERM;       // PERM: scalar element path only (see FwdOp)
    __device__ static __forceinline__ float4 elem4(const Params& p, const Ctx& c, int64_t, const float4& x, const float4& dy, Acc& acc) {
        float4 q, o;
        fq_core4(x, c, q, o);
        nq_accumulate4(q, o, dy, c, p.lam, p.tmode, acc);
        return o;
    }
    static constexpr bool kVec4c = !PERM;
    __device__ static __forceinline__ float4 elem4c(const Params& p, const Ctx* c, int64_t, const float4& x, const float4& dy, Acc* acc) {
        float4 q, o;
        fq_core4c(x, c, q, o);
        nq_accumulate4c(q, o, dy, c, p.lam, p.tmode, acc);
        return o;
    }
};
template <>
struct OpT<OP_BWD> : BwdOp<false> {};
template <>
struct OpT<OP_BWD_PERM> : BwdOp<true> {};

template <>
struct OpT<OP_FUSED> : OpBase {
    static constexpr bool kDy = true;
    static constexpr bool kStore = true;
    static constexpr bool kReduce = true;
    __device__ static __forceinline__ float elem(const Params& p, const Ctx& c, int64_t, float x, float dy, Acc& acc) {
        float q, o;
        fq_core(x, c, q, o);
        nq_accumulate(q, o, dy, p.lam, p.tmode, acc);
        return o;
    }
    static constexpr bool kVec4 = true;
    __device__ static __forceinline__ float4 elem4(const Params& p, const Ctx& c, int64_t, const float4& x, const float4& dy, Acc& acc) {
        float4 q, o;
        fq_core4(x, c, q, o);
        nq_accumulate4(q, o, dy, c, p.lam, p.tmode, acc);
        return o;
    }
    static constexpr bool kVec4c = true;
    __device__ static __forceinline__ float4 elem4c(const Params& p, const Ctx* c, int64_t, const float4& x, const float4& dy, Acc* acc) {
        float4 q, o;
        fq_core4c(x, c, q, o);
        nq_accumulate4c(q, o, dy, c, p.lam, p.tmode, acc);
        return o;
    }
};

// K5a forward: a = bits(max |P|/s), b = number of elements attaining it.
template <>
struct OpT<OP_MAXBIN_FWD> : OpBase {
    static constexpr bool kReduce = true;
    static constexpr bool kStdMerge = false;
    template <typename A, typename B>
    __device__ static __forceinline__ void merge(A& x, const B& y) {
        if (y.a > x.a) {
            x.a = y.a;
            x.b = y.b;
        } else if (y.a == x.a) {
            x.b += y.b;
        }
    }
    __device__ static __forceinline__ float elem(const Params&, const Ctx& c, int64_t, float x, float, Acc& acc) {
        float t = fabsf(x) / c.s;                       // custom_loss_functions.py:92
        uint32_t tb = __float_as_uint(fabsf(t));        // s > 0 in practice; |.| keeps the bit trick valid if not
        if (tb > acc.a) {
            acc.a = tb;
            acc.b = 1u;
        } else if (tb == acc.a) {
            acc.b += 1u;
        }
        return 0.f;
    }
};

// K5a backward: dP_i = (|P_i|/s == mb) ? sign(P_i) * coef / s : 0, coef = c / (G * ties)
template <>
struct OpT<OP_MAXBIN_BWD> : OpBase {
    static constexpr bool kStore = true;
    __device__ static __forceinline__ Ctx ctx(const Params& p, int64_t g) {
        Ctx c;
        c.s = p.s[g];
        c.r = 0.f;
        c.fast = 0;
        c.lam_hi = 0.f;
        c.sure_ok = 0;
        c.k0 = p.mb[g];
        float up = (p.c_dev ? p.c_dev[0] : 1.0f) * p.c_scale;
        c.k1 = (up / (float)p.G) / (float)p.ties[g];
        return c;
    }
    __device__ static __forceinline__ float elem(const Params& p, const Ctx& c, int64_t i, float x, float, Acc&) {
        float t = fabsf(fabsf(x) / c.s);
        float sgn = (x > 0.f) ? 1.f : ((x < 0.f) ? -1.f : 0.f);
        const float g = (t == c.k0) ? (c.k1 / c.s) * sgn : 0.f;
        return p.accum ? p.out[i] + g : g;
    }
};

// K5b forward: c = sum |P - P/s|
template <>
struct OpT<OP_DIFF_FWD> : OpBase {
    static constexpr bool kReduce = true;
    __device__ static __forceinline__ float elem(const Params&, const Ctx& c, int64_t, float x, float, Acc& acc) {
        float pq = x / c.s;                             // custom_loss_functions.py:172
        acc.c += (double)fabsf(x - pq);                 // :175
        return 0.f;
    }
};

// K5b backward: gi = sign(u) * c/N ; dP = gi - gi/s ; ds[g] = sum gi * (P/s) / s
template <>
struct OpT<OP_DIFF_BWD> : OpBase {
    static constexpr bool kStore = true;
    static constexpr bool kReduce = true;
    __device__ static __forceinline__ Ctx ctx(const Params& p, int64_t g) {
        Ctx c;
        c.s = p.s[g];
        c.r = 0.f;
        c.fast = 0;
        c.lam_hi = 0.f;
        c.sure_ok = 0;
        double n = (double)p.outer * (double)p.G * (double)p.inner;
        c.k0 = ((p.c_dev ? p.c_dev[0] : 1.0f) * p.c_scale) / (float)n;
        c.k1 = 0.f;
        return c;
    }
    __device__ static __forceinline__ float elem(const Params& p, const Ctx& c, int64_t i, float x, float, Acc& acc) {
        float pq = x / c.s;
        float u = x - pq;
        float sgn = (u > 0.f) ? 1.f : ((u < 0.f) ? -1.f : 0.f);
        float gi = sgn * c.k0;
        acc.c += (double)((gi * pq) / c.s);
        const float g = gi - gi / c.s;
        return p.accum ? p.out[i] + g : g;
    }
};

}  // namespace lq

#endif
