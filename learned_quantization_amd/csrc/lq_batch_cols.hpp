// lq_batch_cols.hpp -- column traversals of the multi-tensor batch's scale-gradient pass: one partial per (row block, group FRAGMENT)
#ifndef LQ_BATCH_COLS_HPP_
#define LQ_BATCH_COLS_HPP_
#include "lq_frag.hpp"
#include "lq_traverse.hpp"

namespace lq {

// ------------------------------------------------------------------------------------------
//  A conv kernel stored in the order the convolution consumes (layers.py kernel_storage="oihw") with channel-wise scales
//  (custom_layers.py:170-180) is the matrix [co][ci * hw] whose group changes every `hw` columns: (outer, G, inner) =
//  (co, ci, hw).  The generic column tile (lq_traverse.hpp col_tile_body) leaves one partial per (row block, COLUMN) and sizes
//  its row blocks for a single launch (about 512 blocks per tensor): on the ResNet-18-like set that was 7.7 MB of partials
//  written and 10.9 MB read back for 89.4 MB of algorithmic traffic, 1968 blocks of which 1280 are resident at 96 VGPRs, and
//  a finalize that took 5.6 us to emit a few KB (profiles/r03/batch_oihw_storage_imagenette_channelwise/summary.txt).
//
//  Here (multi-tensor batch, OP_BWD, float4 tiles):
//   * the block merges the columns of a group before it writes: one partial per (row block, FRAGMENT), a fragment being the
//     part of a group inside one 256-column tile -- a group of `finner` columns has one fragment, or two where a tile boundary
//     cuts it.  Fragments are numbered in column order, so a group's fragments are adjacent:
//         index of the fragment that contains column x   F(x) = x / finner + x / 256 - x / lcm(finner, 256)
//         fragments per row block                        F    = G + #{tile boundaries that cut a group}
//     finner = 1 (kernels of 1 x 1 taps; `inner` > 64) degenerates to one fragment per column: the generic layout.
//   * rows are addressed as a wave-uniform 64-bit base (SGPR pair) plus a 32-bit lane offset; row indices, loop counters and the
//     clamp of the last round live in SGPRs.
//   * row blocks are sized by the BATCH (lq_kernels.hip batch_rows_per_block): every task gets about the same number of
//     elements per block and the whole launch is one resident round.
//  Sums are exact for lambda < 4e-4 (lq_common.hpp Acc), max|q| and counts are order-free: ds equals the single-tensor result
//  bit for bit there; above, the f64 sums differ below 2^-50 relative as between any two traversals.
// ------------------------------------------------------------------------------------------
#ifndef LQ_BATCH_U
#define LQ_BATCH_U 4          // rows per wave and stage (float4 of each stream per lane), per-column form
#endif
#ifndef LQ_BATCH_U2
#define LQ_BATCH_U2 4         // the same for the two-group form
#endif
#ifndef LQ_BATCH_PIPE
#define LQ_BATCH_PIPE 0       // 1: the loads of the next stage are issued before the current stage is consumed (two register sets)
#endif
#ifndef LQ_BATCH_GF2
#define LQ_BATCH_GF2 1        // 0: the per-column form for every `inner`
#endif
#ifndef LQ_BATCH_NT
#define LQ_BATCH_NT 0         // 1: nontemporal loads
#endif

// GF = 4: a context and an accumulator per COLUMN of the lane's float4 (any `inner`; the only form for inner == 1, where the four
//         columns are four groups, and for the per-column layout);
// GF = 2: two contexts and two accumulators per lane (lq_math.hpp Ctx2: inner >= 2 -- four adjacent columns touch at most two
//         groups), grouped layout only.
// PIPE:   (compile-time experiment, off) the loads of the next stage issued before the current stage is consumed, two register
//         sets.  Every block of a batch launch starts at t = 0 and is resident to the end -- one round of blocks -- and the
//         unpipelined launch runs 1.2-1.7 us faster math-free (profiles/r04/experiments/), so overlapping a wave's arithmetic with its
//         own loads looked like the lever.  Measured (profiles/r04/experiments/pipelined_variants_sweep.jsonl): it is not -- the
//         ResNet-18-like traversal 17.5 us pipelined against 17.4, the ResNet-50-like one 35.4 against 33.0 (99 VGPRs: four waves
//         per SIMD instead of five); with one row per stage (72 VGPRs, seven waves) 40.3 us per step against 39.6.  The ISA is
//         what was intended (stage loads, then s_waitcnt vmcnt(7) .. (4) while the other set is consumed).  Three details of it
//         are kept because the unpipelined loop needs them too: row addresses are wave-uniform POINTERS carried from stage to
//         stage (a row * pitch product would be formed per lane with v_mad_u64_u32, in registers that alias loads in flight); the
//         lane offset is re-materialised as a 32-bit value in the block that issues the loads (instruction selection is per
//         basic block: only there does it see `uniform base + zext(VGPR)` and emit global_load ... v_off, s[base:base+1]); the
//         prefetch is unconditional (row clamped) -- a conditional load leaves a pending-counter state on one path that the
//         compiler waits out on every path.
template <int OP, int U, int GF, int PIPE>
__device__ __forceinline__ void col_frag_tile_body(const Params& p, uint32_t C, uint32_t RB, uint32_t bx, uint32_t by, const FragGeom fg,
                                                   Acc* lds) {
    using O = OpT<OP>;
    static_assert(OP == OP_BWD, "scale-gradient traversal");
    const uint32_t lane = threadIdx.x & 63u;
    const uint32_t w = (uint32_t)__builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));      // wave-uniform: row arithmetic stays scalar
    const uint32_t col0 = (bx * 64u + lane) * 4u;
    const bool active = col0 < C;                          // C % 4 == 0: a float4 never straddles a row end
    uint32_t voff = (active ? col0 : 0u) * 4u;             // idle lanes of the last tile re-read the row's first float4: loads stay unconditional
    const uint32_t outer = (uint32_t)p.outer;
    const uint32_t r0 = by * RB;
    const uint32_t r1 = (r0 + RB < outer) ? r0 + RB : outer;
    const size_t row_bytes = (size_t)C * 4u;
    const char* const Pb = reinterpret_cast<const char*>(p.P);
    const char* const Db = reinterpret_cast<const char*>(p.dy);
    float4 x0[U], d0[U], x1[PIPE ? U : 1], d1[PIPE ? U : 1];
    // Row addresses are wave-uniform POINTERS carried from stage to stage (SGPR pairs: global_load ... v_off, s[base:base+1]); a 64-bit
    // row * pitch product per load would be done per lane in VGPRs (v_mad_u64_u32) -- and the address registers then alias the
    // destinations of loads still in flight, which makes the compiler wait for those before it can issue the prefetch
    const size_t step4 = 4u * row_bytes;                   // four waves interleave the rows of a block
    const char* const lastP = Pb + (size_t)(r1 - 1u) * row_bytes;
    const char* const lastD = Db + (size_t)(r1 - 1u) * row_bytes;
    auto load_stage = [&](float4* x, float4* d, uint32_t r, const char* rp, const char* rd) {
        // the lane offset is re-materialised as a 32-bit value in the block that issues the loads: instruction selection works per
        // basic block and recognises `uniform base + zext(32-bit VGPR)` (the saddr form) only when it sees the zero-extension there;
        // hoisted out of the loop as a 64-bit pair the sum is formed per lane (v_lshl_add_u64) in registers that alias pending loads
        asm volatile("" : "+v"(voff));                     // (in place: a copy would be written into a register that a load in flight still owns)
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const bool in = r + 4u * (uint32_t)u < r1;     // scalar; beyond the block: the block's last row again (loads stay unconditional)
            const char* const qp = in ? rp + (size_t)u * step4 : lastP;
            const char* const qd = in ? rd + (size_t)u * step4 : lastD;
            x[u] = load4<LQ_BATCH_NT>(reinterpret_cast<const float*>(qp + voff));
            d[u] = load4<LQ_BATCH_NT>(reinterpret_cast<const float*>(qd + voff));
        }
    };
    // the lane's scales first -- a few words, cache-resident -- then the first stage: vector loads return in order (one vmcnt), so
    // with the scales ahead of the data the contexts (reciprocals, thresholds) are formed while the first stage is in flight
    const uint32_t inner = (uint32_t)p.inner;
    const uint32_t cg = active ? col0 : 0u;
    const uint32_t g0 = cg / inner, k0 = cg - g0 * inner;
    float sv[GF == 4 ? 4 : 2];
    if constexpr (GF == 4) {
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            uint32_t g = g0, kk = k0 + (uint32_t)k;        // group of column col0 + k without another division (inner may be < 4)
            while (kk >= inner) {
                kk -= inner;
                ++g;
            }
            sv[k] = p.s[g];
        }
    } else {
        sv[0] = p.s[g0];
        sv[1] = p.s[(g0 + 1u < (uint32_t)p.G) ? g0 + 1u : g0];
    }
    uint32_t r = r0 + w;
    const bool any = r < r1;                               // wave-uniform (a row block shorter than four rows leaves waves without work)
    const char* rp = Pb + (size_t)r * row_bytes;           // row r of either stream
    const char* rd = Db + (size_t)r * row_bytes;
    const size_t stage = (size_t)U * step4;
    load_stage(x0, d0, r, rp, rd);                         // also in a wave without rows (clamped to the block's last row): one path, so
    __builtin_amdgcn_sched_barrier(0);                     // that the wait for the scales below does not have to cover the data as well
    // stage loop: `consume(x, d, r)` folds the rows r, r + 4, ... of one stage
    auto run = [&](auto&& consume) {
        if (!any) return;
        if constexpr (PIPE != 0) {
            for (;;) {
                load_stage(x1, d1, r + 4u * U, rp + stage, rd + stage);
                __builtin_amdgcn_sched_barrier(0);         // nothing that reads the current stage moves above the prefetch
                consume(x0, d0, r);
                r += 4u * U;
                rp += stage;
                rd += stage;
                if (r >= r1) break;
                load_stage(x0, d0, r + 4u * U, rp + stage, rd + stage);
                __builtin_amdgcn_sched_barrier(0);
                consume(x1, d1, r);
                r += 4u * U;
                rp += stage;
                rd += stage;
                if (r >= r1) break;
            }
        } else {
            for (;;) {
                consume(x0, d0, r);
                r += 4u * U;
                rp += stage;
                rd += stage;
                if (r >= r1) break;
                load_stage(x0, d0, r, rp, rd);
            }
        }
    };
    const uint32_t X0 = bx * 256u;
    const uint32_t X1 = (X0 + 256u < C) ? X0 + 256u : C;
    const uint32_t gA = X0 / fg.finner;
    const uint32_t nf = (X1 - 1u) / fg.finner - gA + 1u;
    const uint32_t t = threadIdx.x;
    const int64_t pbase = (int64_t)by * fg.F + frag_index(X0, fg.finner, fg.lq);
    if constexpr (GF == 4) {
        Ctx ctx[4];
        Acc acc[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            acc[k] = O::template init<Acc>();
            ctx[k] = O::ctx_of(p, sv[k]);
        }
        run([&](const float4* x, const float4* d, uint32_t rs) {
#pragma unroll
            for (int u = 0; u < U; ++u) {
                if (rs + 4u * (uint32_t)u < r1) {          // scalar condition
                    if (LQ_ABLATE(1)) acc[0].a |= __float_as_uint(x[u].x) ^ __float_as_uint(d[u].w) ^ __float_as_uint(x[u].z + d[u].y);      // development: math-free
                    else O::elem4c(p, ctx, 0, x[u], d[u], acc);
                }
            }
        });
        if (LQ_ABLATE(8)) {                                // development: no epilogue (one dummy word keeps the accumulators alive)
            if ((acc[0].a ^ acc[1].a ^ acc[2].a ^ acc[3].a) == 0x12345u && acc[0].c + acc[1].c + acc[2].c + acc[3].c == 1.5) p.pa[pbase] = 1u;
            return;
        }
        // the four waves meet in LDS; thread t merges fragment t of the tile (its columns x its waves, fixed order)
#pragma unroll
        for (int k = 0; k < 4; ++k) lds[w * 256u + lane * 4u + (uint32_t)k] = acc[k];
        __syncthreads();
        if (t < nf) {
            const uint32_t gs = (gA + t) * fg.finner;
            const uint32_t cs = gs > X0 ? gs : X0;
            const uint32_t ce = (gs + fg.finner < X1) ? gs + fg.finner : X1;
            Acc tot = lds[cs - X0];
#pragma unroll
            for (uint32_t ww = 1; ww < 4; ++ww) O::merge(tot, lds[ww * 256u + (cs - X0)]);
            for (uint32_t c = cs + 1u; c < ce; ++c)
#pragma unroll
                for (uint32_t ww = 0; ww < 4; ++ww) O::merge(tot, lds[ww * 256u + (c - X0)]);
            write_partial(p, pbase + t, tot);
        }
    } else {
        // two groups per lane: A = group of column col0 (elements 0 .. nA-1), B = the next one
        const uint32_t nA = inner - k0;                    // >= 1; >= 4: the whole float4 is A's
        const bool m1 = nA < 2u, m2 = nA < 3u, m3 = nA < 4u;
        const Ctx cA = O::ctx_of(p, sv[0]), cB = O::ctx_of(p, sv[1]);
        Ctx2 c2;
        c2.sA = cA.s;
        c2.rA = cA.r;
        c2.sB = cB.s;
        c2.rB = cB.r;
        c2.lam_hi = cA.lam_hi;
        c2.ok = (cA.fast & cB.fast & 1) | ((cA.sure_ok & cB.sure_ok & 1) << 1);
        Acc A = O::template init<Acc>(), B = O::template init<Acc>();
        run([&](const float4* x, const float4* d, uint32_t rs) {
#pragma unroll
            for (int u = 0; u < U; ++u) {
                if (rs + 4u * (uint32_t)u < r1) {          // scalar condition
                    if (LQ_ABLATE(1)) {                    // development: math-free
                        A.a |= __float_as_uint(x[u].x) ^ __float_as_uint(d[u].w) ^ __float_as_uint(x[u].z + d[u].y);
                    } else {
                        float4 q, o;
                        fq_core4g(x[u], c2, m1, m2, m3, q, o);
                        nq_accumulate4g(q, o, d[u], c2, p.lam, p.tmode, m1, m2, m3, A, B);
                    }
                }
            }
        });
        if (LQ_ABLATE(8)) {                                // development: no epilogue
            if ((A.a ^ B.a) == 0x12345u && A.c + B.c == 1.5) p.pa[pbase] = 1u;
            return;
        }
        lds[w * 128u + lane * 2u] = A;
        lds[w * 128u + lane * 2u + 1u] = B;
        __syncthreads();
        if (t < nf) {
            // fragment t = columns [cs, ce) of group g: held by lanes (cs - X0) / 4 .. (ce - 1 - X0) / 4, in slot A where the lane's
            // first column belongs to g, else in slot B
            const uint32_t g = gA + t;
            const uint32_t gs = g * fg.finner;
            const uint32_t cs = gs > X0 ? gs : X0;
            const uint32_t ce = (gs + fg.finner < X1) ? gs + fg.finner : X1;
            const uint32_t l0 = (cs - X0) >> 2, l1 = (ce - 1u - X0) >> 2;
            Acc tot = O::template init<Acc>();
            for (uint32_t l = l0; l <= l1; ++l) {
                const uint32_t slot = ((X0 + 4u * l) / fg.finner == g) ? 0u : 1u;
#pragma unroll
                for (uint32_t ww = 0; ww < 4; ++ww) O::merge(tot, lds[ww * 128u + l * 2u + slot]);
            }
            write_partial(p, pbase + t, tot);
        }
    }
}

// The forward of the same tiles (multi-tensor batch, OP_FWD, float4 tiles): the addressing of col_frag_tile_body -- wave-uniform row
// pointers, a 32-bit lane offset, the lane's scales requested before the first stage's data -- with one context per column (the
// forward reduces nothing: no accumulators, no LDS).  `out` has the parameter's layout.
template <int U>
__device__ __forceinline__ void col_fwd_tile_body(const Params& p, uint32_t C, uint32_t RB, uint32_t bx, uint32_t by) {
    using O = OpT<OP_FWD>;
    const uint32_t lane = threadIdx.x & 63u;
    const uint32_t w = (uint32_t)__builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const uint32_t col0 = (bx * 64u + lane) * 4u;
    const bool active = col0 < C;
    uint32_t voff = (active ? col0 : 0u) * 4u;
    const uint32_t outer = (uint32_t)p.outer;
    const uint32_t r0 = by * RB;
    const uint32_t r1 = (r0 + RB < outer) ? r0 + RB : outer;
    const size_t row_bytes = (size_t)C * 4u;
    const char* const Pb = reinterpret_cast<const char*>(p.P);
    char* const Ob = reinterpret_cast<char*>(p.out);
    const size_t step4 = 4u * row_bytes;
    const char* const lastP = Pb + (size_t)(r1 - 1u) * row_bytes;
    const uint32_t inner = (uint32_t)p.inner;
    const uint32_t cg = active ? col0 : 0u;
    const uint32_t g0 = cg / inner, k0 = cg - g0 * inner;
    float sv[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        uint32_t g = g0, kk = k0 + (uint32_t)k;
        while (kk >= inner) {
            kk -= inner;
            ++g;
        }
        sv[k] = p.s[g];
    }
    float4 x[U];
    auto load_stage = [&](uint32_t r, const char* rp) {
        asm volatile("" : "+v"(voff));
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const bool in = r + 4u * (uint32_t)u < r1;
            const char* const qp = in ? rp + (size_t)u * step4 : lastP;
            x[u] = *reinterpret_cast<const float4*>(qp + voff);
        }
    };
    uint32_t r = r0 + w;
    const char* rp = Pb + (size_t)r * row_bytes;
    char* ro = Ob + (size_t)r * row_bytes;
    const size_t stage = (size_t)U * step4;
    load_stage(r, rp);
    __builtin_amdgcn_sched_barrier(0);
    Ctx ctx[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) ctx[k] = O::ctx_of(p, sv[k]);
    if (r >= r1) return;
    Acc* const none = nullptr;
    for (;;) {
#pragma unroll
        for (int u = 0; u < U; ++u) {
            if (r + 4u * (uint32_t)u < r1) {               // scalar condition
                const int64_t i = (int64_t)(r + 4u * (uint32_t)u) * C + col0;
                const float4 o = O::elem4c(p, ctx, i, x[u], x[u], none);
                if (active) *reinterpret_cast<float4*>(ro + (size_t)u * step4 + voff) = o;
            }
        }
        r += 4u * U;
        rp += stage;
        ro += stage;
        if (r >= r1) break;
        load_stage(r, rp);
    }
}

// Finalize of that layout: the partials are a matrix [n1 row blocks][F fragments].  A 256-thread block takes `gpb` consecutive groups:
// lane l of every wave owns fragment column c0 + l (every load a contiguous run of a partial row), wave w walks row blocks
// w, w + 4, ...; the four waves' totals meet in LDS and thread t < gpb merges the one or two fragments of group g0 + t in a
// fixed order.  Returns true in the threads that hold a finished group.
template <int OP>
__device__ __forceinline__ bool finalize_frag_body(const Params& p, const FragGeom fg, int64_t n1, int64_t groups, uint32_t g0, AccW* lds,
                                                   int64_t& g, AccW& acc) {
    using O = OpT<OP>;
    const uint32_t lane = threadIdx.x & 63u, wv = threadIdx.x >> 6;
    const uint32_t gend = (g0 + fg.gpb < (uint32_t)groups) ? g0 + fg.gpb : (uint32_t)groups;
    const uint32_t c0 = frag_index(g0 * fg.finner, fg.finner, fg.lq);
    const uint32_t c1 = gend < (uint32_t)groups ? frag_index(gend * fg.finner, fg.finner, fg.lq) : fg.F;      // one past this block's fragments
    acc = O::template init<AccW>();
    if (c0 + lane < c1) {
        if (n1 <= 32) finalize_cols_walk<OP, 8>(p, (int64_t)(c0 + lane), (int64_t)fg.F, n1, (int)wv, acc);
        else finalize_cols_walk<OP, 16>(p, (int64_t)(c0 + lane), (int64_t)fg.F, n1, (int)wv, acc);
    }
    lds[threadIdx.x] = acc;
    __syncthreads();
    const uint32_t gg = g0 + threadIdx.x;
    g = (int64_t)gg;
    const bool emits = gg < gend;
    if (emits) {
        const uint32_t xs = gg * fg.finner;
        const uint32_t fa = frag_index(xs, fg.finner, fg.lq) - c0;
        const uint32_t n = 1u + (((xs + fg.finner - 1u) >> 8) - (xs >> 8));        // a tile boundary strictly inside the group cuts it in two
        acc = O::template init<AccW>();
        for (uint32_t k = 0; k < n; ++k)
#pragma unroll
            for (uint32_t ww = 0; ww < 4; ++ww) O::merge(acc, lds[ww * 64u + fa + k]);      // fixed order
    }
    return emits;
}

}  // namespace lq

#endif
