// lq_reduce.hpp -- wave / block reductions: shuffle butterfly for custom merges, DPP for the standard accumulator
#ifndef LQ_REDUCE_HPP_
#define LQ_REDUCE_HPP_
#include "lq_ops.hpp"

namespace lq {

// ------------------------------------------------------------------------------------------
//  Reductions: wave butterfly (64 lanes) -> LDS across the block's waves.
// ------------------------------------------------------------------------------------------
template <class O, class A>
__device__ __forceinline__ void wave_reduce(A& acc, int width = 64) {
    for (int off = width >> 1; off > 0; off >>= 1) {
        A o;
        o.a = __shfl_xor(acc.a, off, 64);
        o.b = __shfl_xor(acc.b, off, 64);
        o.c = __shfl_xor(acc.c, off, 64);
        O::merge(acc, o);
    }
}

// Result valid in thread 0.  BS = block size (multiple of 64).
template <class O, class A, int BS>
__device__ __forceinline__ void block_reduce(A& acc) {
    __shared__ A lds[BS / 64];
    wave_reduce<O>(acc);
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
    if (lane == 0) lds[wid] = acc;
    __syncthreads();
    if (threadIdx.x == 0) {
        A r = lds[0];
#pragma unroll
        for (int w = 1; w < BS / 64; ++w) O::merge(r, lds[w]);
        acc = r;
    }
}

__device__ __forceinline__ void write_partial(const Params& p, int64_t idx, const Acc& acc) {
    p.pa[idx] = acc.a;
    p.pb[idx] = acc.b;
    p.pc[idx] = acc.c;
}

template <int OP>
__device__ __forceinline__ void emit_direct(const Params& p, int64_t g, const Acc& acc);     // lq_traverse.hpp (needs FinT)

// What every traversal calls: the partial triple of work unit `idx`, or -- when the host saw that each group has exactly one
// partial (then idx is the group index) -- the finished outputs.
template <int OP>
__device__ __forceinline__ void write_partial_t(const Params& p, int64_t idx, const Acc& acc) {
    if constexpr (OP == OP_BWD || OP == OP_FUSED || OP == OP_BWD_PERM) {
        if (p.direct) {
            emit_direct<OP>(p, idx, acc);
            return;
        }
    }
    write_partial(p, idx, acc);
}

// ------------------------------------------------------------------------------------------
//  DPP wave reduction for the standard accumulator (a: max, b: add, c: add) -- VALU only, no LDS
//  crossbar traffic: quad_perm x2, row_half_mirror, row_mirror give every lane of a 16-lane row
//  the row total; row_bcast15 / row_bcast31 carry it across rows; lane 63 ends with the total.
//  The combination order is fixed, so float sums are run-to-run bit-stable.
// ------------------------------------------------------------------------------------------
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ uint32_t dpp_u32(uint32_t identity, uint32_t v) {
    return (uint32_t)__builtin_amdgcn_update_dpp((int)identity, (int)v, CTRL, ROW_MASK, 0xf, false);
}
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ double dpp_f64(double v) {
    const unsigned long long u = __double_as_longlong(v);
    const uint32_t lo = dpp_u32<CTRL, ROW_MASK>(0u, (uint32_t)u);
    const uint32_t hi = dpp_u32<CTRL, ROW_MASK>(0u, (uint32_t)(u >> 32));
    return __longlong_as_double(((unsigned long long)hi << 32) | lo);
}
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ void dpp_step(Acc& acc) {
    const uint32_t a = dpp_u32<CTRL, ROW_MASK>(0u, acc.a);
    const uint32_t b = dpp_u32<CTRL, ROW_MASK>(0u, acc.b);
    const double c = dpp_f64<CTRL, ROW_MASK>(acc.c);
    acc.a = a > acc.a ? a : acc.a;
    acc.b += b;
    acc.c += c;
}
__device__ __forceinline__ void dpp_row_reduce(Acc& acc) {   // every lane of each 16-lane row <- row total
    dpp_step<0xB1, 0xf>(acc);    // quad_perm [1,0,3,2]
    dpp_step<0x4E, 0xf>(acc);    // quad_perm [2,3,0,1]
    dpp_step<0x141, 0xf>(acc);   // row_half_mirror
    dpp_step<0x140, 0xf>(acc);   // row_mirror
}
__device__ __forceinline__ void dpp_wave_reduce(Acc& acc) {  // lane 63 <- wave total
    dpp_row_reduce(acc);
    dpp_step<0x142, 0xa>(acc);   // row_bcast15 into rows 1 and 3
    dpp_step<0x143, 0xc>(acc);   // row_bcast31 into rows 2 and 3
}

// Block reduction for the standard accumulator; result valid in thread 0.  BS multiple of 64, <= 1024.
template <int BS>
__device__ __forceinline__ void block_reduce_dpp(Acc& acc) {
    constexpr int NW = BS / 64;
    __shared__ uint32_t sa[NW], sb[NW];
    __shared__ double sc[NW];
    dpp_wave_reduce(acc);
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
    if (lane == 63) {
        sa[wid] = acc.a;
        sb[wid] = acc.b;
        sc[wid] = acc.c;
    }
    __syncthreads();
    if (wid == 0) {
        Acc r;
        r.a = lane < NW ? sa[lane] : 0u;
        r.b = lane < NW ? sb[lane] : 0u;
        r.c = lane < NW ? sc[lane] : 0.0;
        dpp_row_reduce(r);       // NW <= 16: one row holds every wave's partial
        acc = r;
    }
}

}  // namespace lq

#endif
