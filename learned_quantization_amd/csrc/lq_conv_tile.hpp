// lq_conv_tile.hpp -- conv kernels (HWIO parameter, OIHW consumer) as LDS tiles (round 3)
//
// The reference keeps conv kernels in HWIO (custom_layers.py:321) and MIOpen consumes OIHW.  Round 2 emitted the OIHW
// companion of K1 (and gathered K2's upstream gradient) element by element: 4-byte accesses 4*ci*hw bytes apart.
// rocprofv3 on the ResNet-18-like weight set (profiles/r03/baseline_*): forward 64.8 us with 183 MB written for 89 MB of
// output (every 128-byte line of the companion reaches memory in pieces), gathered backward 58.7 us -- 2.5-3x the plain
// traversals.  Here a block owns a TILE
//
//      c in [c0, c0 + 32 m)   x   o in [o0, o0 + 32)   x   every h = (kh, kw)            32 m hw <= 288 elements per o
//
// of the kernel.  In HWIO, element (h, c, o) sits at ((h ci + c) co + o): for every (h, c) the tile holds one 128-byte
// line (32 consecutive o).  In OIHW it sits at ((o ci + c) hw + h): for every o the tile holds ONE contiguous run of
// 32 m hw floats (1152 bytes = 9 whole lines for a 3x3 kernel).  Both sides are therefore read and written in whole lines:
//
//   HWIO side   thread (c_l = t / 8, o4 = t % 8) owns the float4 at o0 + 4 o4 of row (h, c0 + 32 j + c_l) in pass p = (j, h);
//               every pass's loads are issued up front (<= 9 passes, all independent);
//   LDS         the tile transposed: word [o_l][k], k = (c - c0) hw + h, row stride 289 (odd): the 4-byte transposed
//               accesses of a wave (8 channels x 8 float4 columns) and the run-order accesses are both conflict-free;
//   OIHW side   thread t walks the tile's runs in run order, e = t + 256 i: consecutive lanes -> consecutive addresses.
//
// K1 writes `out` from registers and the companion through LDS; K2 takes dy from the OIHW gradient through LDS (or, for the
// plain entry points, from an HWIO gradient directly), writes dP in HWIO order and accumulates the vote.  The group of an
// element depends on (h, c) only for every orientation the reference has (custom_layers.py:147-197 on an HWIO kernel):
//   kind 0   channelwise: g = c           -- the 8 lanes that share a row reduce with DPP, one partial per (c, o-tile);
//   kind 1   rowwise / columnwise / scalar: g = (h / A) % G -- uniform over the block; the passes are visited group by group
//            and each wave leaves one partial per (group, tile, wave).
// Vote sums are exact (lq_common.hpp, Acc), so ds does not depend on this choice of partials: bit-identical to lq_fq_scale_grad.
#ifndef LQ_CONV_TILE_HPP_
#define LQ_CONV_TILE_HPP_
#include "lq_stream2.hpp"

namespace lq {

constexpr int kCtO = 32;                       // output channels per tile
constexpr int kCtRun = 288;                    // (c, h) pairs per tile: 32 m hw <= 288
constexpr int kCtStride = kCtRun + 1;          // LDS row stride in words (odd)
constexpr int kCtPass = 9;                     // passes of the HWIO side: m hw <= 9
constexpr int kCtIter = kCtO * kCtRun / kBlock;   // 36 run-order steps of the OIHW side
constexpr int kCtLdsWords = kCtO * kCtStride;  // 9248 words = 36 992 bytes

struct ConvTile {
    uint32_t hw, ci, co;
    uint32_t tc;                // input channels per tile (32 m)
    uint32_t nto, ntc;          // tiles along co / along ci
    uint32_t npass;             // m hw
    uint32_t kind;              // 0: g = c, 1: g = pass_g[p]
    FastDiv fnto;               // block -> (c-tile, o-tile)
    FastDiv frun, frun_edge;    // run length tc_eff * hw of a full / of the last c-tile
    uint16_t pass_c[kCtPass + 1];   // channel offset of pass p inside the tile (multiple of 32)
    uint8_t pass_h[kCtPass + 1];    // h of pass p
    uint8_t pass_g[kCtPass + 1];    // kind 1: group of pass p
    uint8_t pass_new[kCtPass + 1];  // the group of pass p differs from that of pass p - 1 (flush the accumulator before it)
};

struct CtGeom {             // what a thread needs to know about its tile
    uint32_t c0, o0, tc_eff, to_eff, c_l, o4, tile, to;
    bool ov;
};

__device__ __forceinline__ CtGeom ct_geom(const ConvTile& ct, uint32_t b) {
    CtGeom g;
    const uint32_t tci = fd_div(ct.fnto, b);
    g.to = b - tci * ct.nto;
    g.tile = b;
    g.c0 = tci * ct.tc;
    g.o0 = g.to * kCtO;
    g.tc_eff = ct.ci - g.c0 < ct.tc ? ct.ci - g.c0 : ct.tc;
    g.to_eff = ct.co - g.o0 < (uint32_t)kCtO ? ct.co - g.o0 : (uint32_t)kCtO;
    g.c_l = threadIdx.x >> 3;
    g.o4 = threadIdx.x & 7;
    g.ov = 4 * g.o4 < g.to_eff;
    return g;
}

// run-order index e of the tile -> (o_l, k); RL = tc_eff * hw
__device__ __forceinline__ void ct_run(const ConvTile& ct, const CtGeom& g, uint32_t e, uint32_t& o_l, uint32_t& k) {
    const bool edge = g.tc_eff != ct.tc;            // block-uniform
    o_l = edge ? fd_div(ct.frun_edge, e) : fd_div(ct.frun, e);
    k = e - o_l * (g.tc_eff * ct.hw);
}

// ------------------------------------------------------------------------------------------
//  K1 on a tile: out (HWIO) from registers, out_perm (OIHW) through LDS when p.out_perm is set.
// ------------------------------------------------------------------------------------------
__device__ __forceinline__ void conv_tile_fwd(const Params& p, const ConvTile& ct, uint32_t b, float* lds) {
    using O = OpT<OP_FWD>;
    const CtGeom g = ct_geom(ct, b);
    float4 x[kCtPass];
    uint32_t idx[kCtPass], crel[kCtPass];
    bool v[kCtPass];
#pragma unroll
    for (int q = 0; q < kCtPass; ++q) {
        crel[q] = ct.pass_c[q] + g.c_l;
        v[q] = q < (int)ct.npass && crel[q] < g.tc_eff && g.ov;
        idx[q] = ((uint32_t)ct.pass_h[q] * ct.ci + g.c0 + crel[q]) * ct.co + g.o0 + 4 * g.o4;
        x[q] = make_float4(0.f, 0.f, 0.f, 0.f);
        if (v[q]) x[q] = load4<0>(p.P + idx[q]);
    }
    __builtin_amdgcn_sched_barrier(0);            // every load of the tile is in flight before the first scale fetch
    Acc none = O::template init<Acc>();
    Ctx ctx = O::ctx(p, 0);
    uint32_t gprev = 0;
#pragma unroll
    for (int q = 0; q < kCtPass; ++q) {
        if (q < (int)ct.npass) {                    // block-uniform
            if (v[q]) {
                const uint32_t gq = ct.kind == 0 ? g.c0 + crel[q] : (uint32_t)ct.pass_g[q];
                if (gq != gprev) {
                    ctx = O::ctx(p, gq);
                    gprev = gq;
                }
                const float4 o = O::elem4(p, ctx, idx[q], x[q], x[q], none);
                store4<0>(p.out + idx[q], o);
                if (p.out_perm) {
                    float* w = lds + (4 * g.o4) * kCtStride + crel[q] * ct.hw + ct.pass_h[q];
                    w[0] = o.x;
                    w[kCtStride] = o.y;
                    w[2 * kCtStride] = o.z;
                    w[3 * kCtStride] = o.w;
                }
            }
        }
    }
    if (p.out_perm) {
        __syncthreads();
        const uint32_t RL = g.tc_eff * ct.hw, E = g.to_eff * RL;
        float* dst = p.out_perm + ((size_t)g.o0 * ct.ci + g.c0) * ct.hw;
        const uint32_t ostride = ct.ci * ct.hw;
#pragma unroll 4
        for (int i = 0; i < kCtIter; ++i) {
            const uint32_t e = threadIdx.x + (uint32_t)i * kBlock;
            if (e < E) {
                uint32_t o_l, k;
                ct_run(ct, g, e, o_l, k);
                dst[o_l * ostride + k] = lds[o_l * kCtStride + k];
            }
        }
    }
}

// ------------------------------------------------------------------------------------------
//  K2 on a tile.  OIHW = true: dy arrives in OIHW order (p.dy_perm), goes through LDS and is written back in HWIO order
//  to p.dp_out (dP == dy, custom_layers.py:118).  OIHW = false: dy is an HWIO tensor (p.dy), read like P.
// ------------------------------------------------------------------------------------------
template <int OP>
__device__ __forceinline__ void ct_flush(const Params& p, const ConvTile& ct, const CtGeom& g, Acc& acc, uint32_t gq, bool mine) {
    if (ct.kind == 0) {
        dpp_team_reduce<3>(acc);                      // the 8 lanes of a row (same c): every lane ends with the row total
        if (mine && g.o4 == 0) write_partial(p, (int64_t)gq * ct.nto + g.to, acc);
    } else {
        dpp_wave_reduce(acc);                         // lane 63 <- wave total
        if ((threadIdx.x & 63) == 63)
            write_partial(p, ((int64_t)gq * (ct.nto * ct.ntc) + g.tile) * kWavesPerBlock + (threadIdx.x >> 6), acc);
    }
}

template <bool OIHW>
__device__ __forceinline__ void conv_tile_bwd(const Params& p, const ConvTile& ct, uint32_t b, float* lds) {
    using O = OpT<OP_BWD>;
    const CtGeom g = ct_geom(ct, b);
    float4 x[kCtPass], d[kCtPass];
    uint32_t idx[kCtPass], crel[kCtPass];
    bool v[kCtPass];
#pragma unroll
    for (int q = 0; q < kCtPass; ++q) {
        crel[q] = ct.pass_c[q] + g.c_l;
        v[q] = q < (int)ct.npass && crel[q] < g.tc_eff && g.ov;
        idx[q] = ((uint32_t)ct.pass_h[q] * ct.ci + g.c0 + crel[q]) * ct.co + g.o0 + 4 * g.o4;
        x[q] = make_float4(0.f, 0.f, 0.f, 0.f);
        d[q] = x[q];
        if (v[q]) {
            x[q] = load4<0>(p.P + idx[q]);
            if (!OIHW) d[q] = load4<0>(p.dy + idx[q]);
        }
    }
    if (OIHW) {
        const uint32_t RL = g.tc_eff * ct.hw, E = g.to_eff * RL;
        const float* src = p.dy_perm + ((size_t)g.o0 * ct.ci + g.c0) * ct.hw;
        const uint32_t ostride = ct.ci * ct.hw;
        // the run-order loads in four rounds of nine (36 further registers in flight would cost a wave of occupancy)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            float t[kCtIter / 4];
            uint32_t a[kCtIter / 4];
#pragma unroll
            for (int i = 0; i < kCtIter / 4; ++i) {
                const uint32_t e = threadIdx.x + (uint32_t)(r * (kCtIter / 4) + i) * kBlock;
                uint32_t o_l = 0, k = 0;
                t[i] = 0.f;
                a[i] = 0xffffffffu;
                if (e < E) {
                    ct_run(ct, g, e, o_l, k);
                    t[i] = src[o_l * ostride + k];
                    a[i] = o_l * kCtStride + k;
                }
            }
#pragma unroll
            for (int i = 0; i < kCtIter / 4; ++i)
                if (a[i] != 0xffffffffu) lds[a[i]] = t[i];
        }
        __syncthreads();
    }
    __builtin_amdgcn_sched_barrier(0);
    Acc acc = O::template init<Acc>();
    Ctx ctx = O::ctx(p, 0);
    uint32_t gprev = 0xffffffffu;
    bool mine = false;                                // this thread accumulated something for gprev
#pragma unroll
    for (int q = 0; q < kCtPass; ++q) {
        if (q < (int)ct.npass) {                      // block-uniform
            if (q > 0 && ct.pass_new[q]) {            // block-uniform: the group changes here for every thread
                ct_flush<OP_BWD>(p, ct, g, acc, gprev, mine);
                acc = O::template init<Acc>();
                mine = false;
                gprev = 0xffffffffu;
            }
            if (v[q]) {
                const uint32_t gq = ct.kind == 0 ? g.c0 + crel[q] : (uint32_t)ct.pass_g[q];
                if (gq != gprev) {
                    ctx = O::ctx(p, gq);
                    gprev = gq;
                }
                mine = true;
                if (OIHW) {
                    const float* r = lds + (4 * g.o4) * kCtStride + crel[q] * ct.hw + ct.pass_h[q];
                    d[q] = make_float4(r[0], r[kCtStride], r[2 * kCtStride], r[3 * kCtStride]);
                    store4<0>(p.dp_out + idx[q], d[q]);
                }
                O::elem4(p, ctx, idx[q], x[q], d[q], acc);
            } else if (ct.kind == 1) {
                gprev = ct.pass_g[q];                 // idle lanes still take part in the wave reduction of this group
            }
        }
    }
    ct_flush<OP_BWD>(p, ct, g, acc, gprev, mine);
}

// single-tensor launches (lq_fq_forward_oihw / lq_fq_scale_grad_oihw)
__global__ __launch_bounds__(kBlock) void k_conv_tile_fwd(Params p, ConvTile ct) {
    __shared__ float lds[kCtLdsWords];
    conv_tile_fwd(p, ct, blockIdx.x, lds);
}
template <bool OIHW>
__global__ __launch_bounds__(kBlock) void k_conv_tile_bwd(Params p, ConvTile ct) {
    __shared__ float lds[OIHW ? kCtLdsWords : 1];
    conv_tile_bwd<OIHW>(p, ct, blockIdx.x, lds);
}

}  // namespace lq

#endif
