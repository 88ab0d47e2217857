// lq_stream2.hpp -- streaming-size (>= 4 M elements) forms of the column and tiny-row traversals (round 2)
//
// What rocprofv3 showed for the round-1 kernels of these modes (profiles/r02/shapes_baseline_counters.json; MI355X):
//   * every kernel moves exactly its algorithmic bytes (traffic / algorithmic <= 1.03): nothing is re-read;
//   * the read-only K2 tile already streams at 5.6-5.7 TB/s, but the SAME loop with stores in it (K1: 5.0, K4: 4.9 TB/s)
//     is a fifth slower.  On gfx9-family hardware loads and stores share one in-order counter (vmcnt): a wave that
//     issues the stores of iteration i and then the loads of iteration i+1 cannot see those loads before the stores
//     have been acknowledged by the memory side, so every iteration pays write latency + read latency back to back;
//   * tiny rows (1 M rows x 32): K2 60 us against 47 us for the same bytes in rows of 512 -- the per-row epilogue (team
//     reduction through ds_bpermute, f64 mean, direct emit from 8 of 64 lanes) is paid once per 128 bytes of each stream.
// Hence:
//   k_flat_fwd        K1 (no reduction) of ANY descriptor as a flat stream, one float4 per thread and no loop -- the shape
//                     of the row-stream kernel; the group of an element comes from its flat index (invariant-divisor
//                     arithmetic, lq_fastdiv.hpp).  Since the end of round 2 also the forward of long aligned rows (BENCH).
//   k_col_pipe        column tile with a software pipeline: the loads of iteration i+1 are issued BEFORE the stores of
//                     iteration i, so the vmcnt wait that covers them does not include those stores.
//   k_col_periodic_pipe   the same pipeline for the periodic float4 form (C <= 64; also 64 < C <= 512 where a tile would
//                     leave lanes idle or start its rows inside a 128-byte line).
//   k_flat_cols       C = 8, 16, 32, 64 as a flat one-shot stream with an xor-shuffle tree over the lanes that share columns
//                     (K1 and K4 with two float4 per thread and stream, K2 with four).
//   k_row_win         rows of 65..1023 elements of any alignment, and of 1153..1945 where the row stream's two chunks fill
//                     poorly (scale-gradient ops): a team (up to one wave) per row reads the aligned float4 window of the row.
//   k_row_seg         rows of 5..64 elements off the 16-byte grid: one flat window per block, segmented reduction through LDS.
//   k_row_tiny        rows of <= 64 elements: U passes of rows per wave with all loads up front, DPP team reductions
//                     (VALU only), ONE emit per wave in which lane (team, u) finishes row (u, team).
// All of them are used for tensors of >= 4 M elements only (kPeriodic4Min): below that the round-1 bodies run, the same
// code the multi-tensor batch kernels inline, so batched and single-tensor results stay bit-identical there.
#ifndef LQ_STREAM2_HPP_
#define LQ_STREAM2_HPP_
#include "lq_traverse.hpp"

namespace lq {

// ------------------------------------------------------------------------------------------
//  K1 as a flat stream.  GM: how the group of element i is found
//     0: g = (i / inner) % G with 32-bit arithmetic (numel < 2^32), one group per float4 (inner % 4 == 0)
//     1: the same, per element (a float4 may straddle groups)
//     2 / 3: as 0 / 1 with 64-bit arithmetic
//     4 / 5: inner == 1 and G % 4 == 0 (NHWC per-channel activations, column-wise Dense): the four scales of a float4 are
//            one aligned float4 of the scale vector at column (i % G) -- one 32-bit (4) or 64-bit (5) modulo per thread;
//            the quotient is the IEEE `/` itself (correctly rounded, ~11 VALU each): building four reciprocal contexts
//            per thread costs more than it saves when each is used for a single element
// ------------------------------------------------------------------------------------------
//     8 / 9: inner == 1 and G % 4 != 0 (G >= 4): as 4 / 5 with a dword-aligned scale float4, element by element where it wraps
//     10 / 11: any descriptor below 2^32 elements (10: inner == 1), a scale gathered per element -- the forward of the column
//            modes whose float4s hold several groups (C = 3, 5, 10, 30; inner = 2, 3, 5, 15)
//     6 / 7: long rows that do not start on 16-byte lines (row mode with inner % 4 != 0, e.g. rows of 4100 or 4099 elements):
//            the row-stream kernel would start every block in the middle of a 128-byte line (each 1 KB wave access then touches
//            9 lines instead of 8: K1 5.4-5.9 TB/s); as a flat stream every access is line-aligned, a float4 has one group
//            unless it straddles a row end -- checked from the groups of its first and last element (32-bit / 64-bit)
#ifdef LQ_DEV_KNOBS
// development builds: lq_dev_set_flags(bits) -- 1: k_row_win stores with the default cache policy (loads stay nontemporal)
__device__ int g_dev_flags = 0;
#endif

template <bool WIDE>
__device__ __forceinline__ int64_t flat_group_w(const Params& p, const FlatIdx& fx, int64_t i) {
    if (WIDE) return (i / p.inner) % p.G;
    return (int64_t)fd_mod(fx.G, fd_div(fx.inner, (uint32_t)i));        // exact: see FastDiv
}
template <int GM>
__device__ __forceinline__ int64_t flat_group(const Params& p, const FlatIdx& fx, int64_t i) {
    return flat_group_w<(GM == 2 || GM == 3 || GM == 5 || GM == 7)>(p, fx, i);
}

template <int OP, int BS, int NT, int GM>
__global__ __launch_bounds__(BS) void k_flat_fwd(Params p, FlatIdx fx, int64_t nv, int rem) {
    using O = OpT<OP>;
    const int64_t v = (int64_t)blockIdx.x * BS + threadIdx.x;
    if (v < nv) {
        const int64_t i = v * 4;
        const float4 x = load4<NT>(p.P + i);
        __builtin_amdgcn_sched_barrier(0);      // the load first; index arithmetic and contexts while it is in flight
        Acc none = O::template init<Acc>();
        float4 o;
        if (GM == 4 || GM == 5) {
            const int64_t c = GM == 4 ? (int64_t)fd_mod(fx.G, (uint32_t)i) : i % p.G;
            const float4 sv = *reinterpret_cast<const float4*>(p.s + c);
            float4 q;
            q.x = floorf(x.x / sv.x); q.y = floorf(x.y / sv.y); q.z = floorf(x.z / sv.z); q.w = floorf(x.w / sv.w);   // custom_layers.py:56-59
            o.x = q.x * sv.x; o.y = q.y * sv.y; o.z = q.z * sv.z; o.w = q.w * sv.w;                                   // :60
            if (p.q) {
                store_q(p.q, p.q_dtype, i + 0, q.x);
                store_q(p.q, p.q_dtype, i + 1, q.y);
                store_q(p.q, p.q_dtype, i + 2, q.z);
                store_q(p.q, p.q_dtype, i + 3, q.w);
            }
        } else if (GM == 8 || GM == 9) {
            // inner == 1, any G >= 4 (column matrices whose rows are off the 16-byte grid): the scales of a float4 are the four
            // consecutive entries at column i % G -- one dword-aligned float4 load -- unless the float4 wraps around a row end
            const int64_t c = GM == 8 ? (int64_t)fd_mod(fx.G, (uint32_t)i) : i % p.G;
            float4 sv;
            if (__builtin_expect(c + 4 <= p.G, 1)) {
                sv = load4x<0, 1>(p.s + c);
            } else {
                sv.x = p.s[c];
                sv.y = p.s[c + 1 < p.G ? c + 1 : c + 1 - p.G];
                sv.z = p.s[c + 2 < p.G ? c + 2 : c + 2 - p.G];
                sv.w = p.s[c + 3 < p.G ? c + 3 : c + 3 - p.G];
            }
            float4 q;
            q.x = floorf(x.x / sv.x); q.y = floorf(x.y / sv.y); q.z = floorf(x.z / sv.z); q.w = floorf(x.w / sv.w);   // custom_layers.py:56-59
            o.x = q.x * sv.x; o.y = q.y * sv.y; o.z = q.z * sv.z; o.w = q.w * sv.w;                                   // :60
            if (p.q) {
                store_q(p.q, p.q_dtype, i + 0, q.x);
                store_q(p.q, p.q_dtype, i + 1, q.y);
                store_q(p.q, p.q_dtype, i + 2, q.z);
                store_q(p.q, p.q_dtype, i + 3, q.w);
            }
        } else if (GM == 10 || GM == 11) {
            // ANY descriptor below 2^32 elements: the group of each element from its flat index (invariant-divisor arithmetic,
            // ~12 VALU per element), its scale gathered from the (cache-resident) scale vector, the IEEE `/`.  GM 10: inner == 1.
            const uint32_t iu = (uint32_t)i;
            float4 sv;
            if (GM == 10) {
                sv.x = p.s[fd_mod(fx.G, iu)];
                sv.y = p.s[fd_mod(fx.G, iu + 1)];
                sv.z = p.s[fd_mod(fx.G, iu + 2)];
                sv.w = p.s[fd_mod(fx.G, iu + 3)];
            } else {
                sv.x = p.s[fd_mod(fx.G, fd_div(fx.inner, iu))];
                sv.y = p.s[fd_mod(fx.G, fd_div(fx.inner, iu + 1))];
                sv.z = p.s[fd_mod(fx.G, fd_div(fx.inner, iu + 2))];
                sv.w = p.s[fd_mod(fx.G, fd_div(fx.inner, iu + 3))];
            }
            float4 q;
            q.x = floorf(x.x / sv.x); q.y = floorf(x.y / sv.y); q.z = floorf(x.z / sv.z); q.w = floorf(x.w / sv.w);   // custom_layers.py:56-59
            o.x = q.x * sv.x; o.y = q.y * sv.y; o.z = q.z * sv.z; o.w = q.w * sv.w;                                   // :60
            if (p.q) {
                store_q(p.q, p.q_dtype, i + 0, q.x);
                store_q(p.q, p.q_dtype, i + 1, q.y);
                store_q(p.q, p.q_dtype, i + 2, q.z);
                store_q(p.q, p.q_dtype, i + 3, q.w);
            }
        } else if (GM == 0 || GM == 2) {
            const Ctx ctx = O::ctx(p, flat_group<GM>(p, fx, i));
            o = O::elem4(p, ctx, i, x, x, none);
        } else if (GM == 6 || GM == 7) {
            // one division: row of the first element and its offset in that row; the float4 straddles a row end iff off + 3 >= inner
            int64_t g0, g3;
            uint32_t off6 = 0;
            int64_t off7 = 0;
            if (GM == 6) {
                const uint32_t r0 = fd_div(fx.inner, (uint32_t)i);
                off6 = (uint32_t)i - r0 * fx.inner.d;
                const uint32_t a = fd_mod(fx.G, r0), b = a + 1 == fx.G.d ? 0u : a + 1;
                g0 = a;
                g3 = off6 + 3 >= fx.inner.d ? b : a;
            } else {
                const int64_t r0 = i / p.inner;
                off7 = i - r0 * p.inner;
                g0 = r0 % p.G;
                g3 = off7 + 3 >= p.inner ? (g0 + 1 == p.G ? 0 : g0 + 1) : g0;
            }
            // ONE straight-line pass with a scale per element and the IEEE `/` (as group modes 4 / 8): a float4 that straddles
            // a row end takes the same instructions as the others.  (With rows of 49 elements every 12th float4 straddles;
            // as a divergent branch the per-element path ran in nearly every wave: 4.8 TB/s.)
            const float s0 = p.s[g0];
            float s3 = s0;
            if (g3 != g0) s3 = p.s[g3];
            float4 sv;
            sv.x = s0;
            sv.y = (GM == 6 ? off6 + 1 >= fx.inner.d : off7 + 1 >= p.inner) ? s3 : s0;
            sv.z = (GM == 6 ? off6 + 2 >= fx.inner.d : off7 + 2 >= p.inner) ? s3 : s0;
            sv.w = s3;
            float4 q;
            q.x = floorf(x.x / sv.x); q.y = floorf(x.y / sv.y); q.z = floorf(x.z / sv.z); q.w = floorf(x.w / sv.w);   // custom_layers.py:56-59
            o.x = q.x * sv.x; o.y = q.y * sv.y; o.z = q.z * sv.z; o.w = q.w * sv.w;                                   // :60
            if (p.q) {
                store_q(p.q, p.q_dtype, i + 0, q.x);
                store_q(p.q, p.q_dtype, i + 1, q.y);
                store_q(p.q, p.q_dtype, i + 2, q.z);
                store_q(p.q, p.q_dtype, i + 3, q.w);
            }
        } else {
            Ctx ctx[4];
            Acc acc[4];
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                ctx[k] = O::ctx(p, flat_group<GM>(p, fx, i + k));
                acc[k] = none;
            }
            o = O::elem4c(p, ctx, i, x, x, acc);
        }
        if (O::kStore) store4<NT>(p.out + i, o);
    } else if (v == nv && rem) {                // numel % 4 trailing elements
        Acc none = O::template init<Acc>();
        for (int k = 0; k < rem; ++k) {
            const int64_t i = nv * 4 + k;
            const Ctx ctx = O::ctx(p, (i / p.inner) % p.G);
            const float r = O::elem(p, ctx, i, p.P[i], 0.f, none);
            if (O::kStore) p.out[i] = r;
        }
    }
}

// ------------------------------------------------------------------------------------------
//  Column tile, software-pipelined.  Block of NW waves owns RB rows x 256 columns; a lane keeps 4 fixed columns; wave w
//  walks rows w, w + NW, ...; U rows per iteration and stream.  Partial layout as in col_tile_body: (row-block, column).
// ------------------------------------------------------------------------------------------
template <int OP, int NT, int U, int NW, int UA = 0>
__global__ __launch_bounds__(NW * 64) void k_col_pipe(Params p, int64_t C, int64_t RB, int64_t nbx) {
    using O = OpT<OP>;
    __shared__ Acc lds[O::kReduce ? NW * 256 : 1];
    const int lane = threadIdx.x & 63;
    const int w = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    // Rows that are not whole 128-byte lines (C % 32 != 0): the first / last line of a block's 1 KB row segment is shared with the
    // neighbouring column block.  Consecutive block indices go to consecutive XCDs (separate L2s), so each of the two fetches
    // that line from the fabric; remapped so that one XCD owns a contiguous range of tiles, neighbours share an L2.
    int64_t b = blockIdx.x;
    if (UA >= 1 && OP == OP_BWD) {      // measured: the read-only K2 gains 3-5 % (C = 1000, 2000, 3000), the storing kernels LOSE 3-7 %
        const int64_t nb = gridDim.x, q = nb >> 3, r = nb & 7;
        const int64_t xcd = b & 7, slot = b >> 3;
        b = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + slot;
    }
    const int64_t by = b / nbx, bx = b - by * nbx;
    const int64_t col0n = (bx * 64 + lane) * 4;
    const bool active = col0n < C;
    // UA = 0 / 2: C % 4 == 0, a float4 never straddles a row end (2: with the XCD remap above).  UA = 1 (any C >= 4): rows start on any 4-byte phase
    // (dword-aligned float4 access), and the lane whose float4 would cross the row end takes the LAST four columns instead:
    // its first `kdup` elements repeat the previous lane's -- same inputs, same outputs (the stores write identical
    // values), and their accumulators are simply not emitted.
    const int64_t col0 = (UA == 1 && col0n + 4 > C) ? C - 4 : col0n;
    const int kdup = (int)(col0n - col0);
    const int64_t r0 = by * RB;
    const int64_t r1 = (r0 + RB < p.outer) ? r0 + RB : p.outer;
    Acc acc[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) acc[k] = O::template init<Acc>();
    if (active && r0 + w < r1) {
        const int64_t step = (int64_t)NW * C;                 // elements between consecutive rows of this wave
        int64_t i0 = (r0 + w) * C + col0;                     // element index of this lane's float4 in the wave's next row
        const int cnt = (int)((r1 - (r0 + w) + NW - 1) / NW); // rows of this wave (wave-uniform)
        const int groups = cnt / U;                           // complete groups of U rows: the pipelined part
        float4 xa[U], da[U], xb[U], db[U];
        if (groups > 0) {
#pragma unroll
            for (int u = 0; u < U; ++u) {
                xa[u] = load4x<NT, (UA == 1)>(p.P + i0 + u * step);
                da[u] = xa[u];
                if (O::kDy) da[u] = load4x<NT, (UA == 1)>(p.dy + i0 + u * step);
            }
        }
        __builtin_amdgcn_sched_barrier(0);
        Ctx ctx[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) ctx[k] = O::ctx(p, (col0 + k) / p.inner);
        // One phase: issue the loads of the NEXT group into (xn, dn), then compute and store the current group (xc, dc).
        // The two register sets swap roles every phase (the loop is unrolled twice): copying "next" into "current"
        // would make the compiler wait for the loads -- and, vmcnt being in-order, for the stores before them -- at
        // the copy instead of at the first use one phase later.  Every loaded group is consumed on every path (no
        // per-row validity inside the loop), so no wait is carried around the back edge.
        int g = 0;
        auto phase = [&](float4 (&xc)[U], float4 (&dc)[U], float4 (&xn)[U], float4 (&dn)[U]) -> bool {
            const bool more = g + 1 < groups;                 // wave-uniform
            if (more) {
                const int64_t in0 = i0 + U * step;
#pragma unroll
                for (int u = 0; u < U; ++u) {
                    xn[u] = load4x<NT, (UA == 1)>(p.P + in0 + u * step);
                    dn[u] = xn[u];
                    if (O::kDy) dn[u] = load4x<NT, (UA == 1)>(p.dy + in0 + u * step);
                }
            }
            __builtin_amdgcn_sched_barrier(0);                // the next group's loads stay ahead of this phase's stores
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const int64_t i = i0 + u * step;
                const float4 ov = O::elem4c(p, ctx, i, xc[u], dc[u], acc);
                if (O::kStore) store4x<NT, (UA == 1)>(p.out + i, ov);
            }
            ++g;
            i0 += U * step;
            return more;
        };
        if (groups > 0) {
            for (;;) {
                if (!phase(xa, da, xb, db)) break;
                if (!phase(xb, db, xa, da)) break;
            }
        }
        for (int t = groups * U; t < cnt; ++t) {              // at most U - 1 leftover rows
            const float4 x = load4x<NT, (UA == 1)>(p.P + i0);
            float4 d = x;
            if (O::kDy) d = load4x<NT, (UA == 1)>(p.dy + i0);
            const float4 ov = O::elem4c(p, ctx, i0, x, d, acc);
            if (O::kStore) store4x<NT, (UA == 1)>(p.out + i0, ov);
            i0 += step;
        }
    }
    if (O::kReduce) {
#pragma unroll
        for (int k = 0; k < 4; ++k) lds[w * 256 + lane * 4 + k] = acc[k];
        __syncthreads();
        if (w == 0 && active) {
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                Acc r = lds[lane * 4 + k];
#pragma unroll
                for (int ww = 1; ww < NW; ++ww) O::merge(r, lds[ww * 256 + lane * 4 + k]);   // fixed wave order
                if (UA != 1 || k >= kdup) write_partial_t<OP>(p, by * C + col0 + k, r);
            }
        }
    }
}

// ------------------------------------------------------------------------------------------
//  Periodic float4 form (C <= 64), software-pipelined.  Geometry and partial layout of col_periodic4_body: `nblk` (a
//  multiple of C) blocks of 256 threads; thread tid sees vectors tid, tid + T, ...; one partial per (block, column).
// ------------------------------------------------------------------------------------------
template <int OP, int NT, int U, int BS = kBlock>
__global__ __launch_bounds__(BS, (U <= 2 ? 4 : 0)) void k_col_periodic_pipe(Params p, int C, int64_t nblk) {
    using O = OpT<OP>;
    __shared__ Acc lds[O::kReduce ? BS * 5 : 1];
    const int64_t blk = blockIdx.x;
    const int64_t numel = p.outer * (int64_t)C;
    const int64_t T = nblk * BS;
    const int64_t tid = blk * BS + threadIdx.x;
    const int64_t nv = numel >> 2;
    Acc acc[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[j] = O::template init<Acc>();
    const int c0 = (int)((tid * 4) % C);
    // every thread owns at least nv / T vectors (threads below nv % T one more): the pipelined part runs over the groups
    // of U vectors that EVERY thread has -- no per-lane validity inside the loop (see k_col_pipe) -- the rest follows
    const int64_t common = nv / T;
    const int64_t groups = common / U;
    int64_t v = tid;
    float4 xa[U], da[U], xb[U], db[U];
    if (groups > 0) {
#pragma unroll
        for (int u = 0; u < U; ++u) {
            xa[u] = load4<NT>(p.P + (v + u * T) * 4);
            da[u] = xa[u];
            if (O::kDy) da[u] = load4<NT>(p.dy + (v + u * T) * 4);
        }
    }
    __builtin_amdgcn_sched_barrier(0);          // the first loads before the scale fetches
    Ctx ctx[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) ctx[j] = O::ctx(p, ((c0 + j) % C) / p.inner);
    if (groups > 0) {
        int64_t g = 0;
        auto phase = [&](float4 (&xc)[U], float4 (&dc)[U], float4 (&xn)[U], float4 (&dn)[U]) -> bool {
            const bool more = g + 1 < groups;                 // grid-uniform
            const int64_t vn = v + (int64_t)U * T;
            if (more) {
#pragma unroll
                for (int u = 0; u < U; ++u) {
                    xn[u] = load4<NT>(p.P + (vn + u * T) * 4);
                    dn[u] = xn[u];
                    if (O::kDy) dn[u] = load4<NT>(p.dy + (vn + u * T) * 4);
                }
            }
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const int64_t i = (v + u * T) * 4;
                const float4 o = O::elem4c(p, ctx, i, xc[u], dc[u], acc);
                if (O::kStore) store4<NT>(p.out + i, o);
            }
            v = vn;
            ++g;
            return more;
        };
        for (;;) {
            if (!phase(xa, da, xb, db)) break;
            if (!phase(xb, db, xa, da)) break;
        }
    }
    for (; v < nv; v += T) {                                  // at most U leftover vectors of this thread
        const float4 x = load4<NT>(p.P + v * 4);
        float4 d = x;
        if (O::kDy) d = load4<NT>(p.dy + v * 4);
        const float4 o = O::elem4c(p, ctx, v * 4, x, d, acc);
        if (O::kStore) store4<NT>(p.out + v * 4, o);
    }
    const int rem = (int)(numel & 3);
    if (rem && tid == nv % T) {                     // ragged tail: the thread that would own the next vector
        const int64_t i = nv * 4;
#pragma unroll
        for (int j = 0; j < 3; ++j) {
            if (j < rem) {
                const float o = O::elem(p, ctx[j], i + j, p.P[i + j], O::kDy ? p.dy[i + j] : 0.f, acc[j]);
                if (O::kStore) p.out[i + j] = o;
            }
        }
    }
    if (O::kReduce) {
        Acc* lds2 = lds + BS * 4;
#pragma unroll
        for (int j = 0; j < 4; ++j) lds[threadIdx.x * 4 + j] = acc[j];
        __syncthreads();
        // entry e of this block belongs to column (blk*BS*4 + e) % C; H helpers per column walk them in a fixed order
        const int H = BS / C;
        const int c = (int)threadIdx.x % C, h = (int)threadIdx.x / C;
        const int b0 = (int)((blk * (BS * 4)) % C);
        if (h < H) {
            int e0 = c - b0;
            if (e0 < 0) e0 += C;
            Acc r = O::template init<Acc>();
            for (int e = e0 + h * C; e < BS * 4; e += H * C) O::merge(r, lds[e]);
            lds2[h * C + c] = r;
        }
        __syncthreads();
        if ((int)threadIdx.x < C) {
            Acc r = lds2[threadIdx.x];
            for (int hh = 1; hh < H; ++hh) O::merge(r, lds2[hh * C + (int)threadIdx.x]);                  // fixed order
            write_partial_t<OP>(p, blk * C + threadIdx.x, r);
        }
    }
}

// ------------------------------------------------------------------------------------------
//  C = 8, 16, 32, 64 columns as a flat ONE-SHOT stream (power-of-two channel counts of NHWC activations).  The looping periodic
//  form keeps a wave alive through 5-9 dependent load rounds; here, as in the row-stream kernel, a thread owns U float4 of each
//  stream, issued together, and the block ends after ONE barrier: a lane's four columns are the same in every vector it owns
//  ((4*512) % C == 0) and the lanes of a wave that share them are C/4 apart, so an xor-shuffle tree over the upper lane bits
//  leaves the wave totals in lanes 0 .. C/4-1 (fixed order: run-to-run bit-stable); one LDS hop across the block's 8 waves,
//  merged in wave order by the first C threads: one partial per (block, column).
//  Measured (profiles/r02/tuning_flat_cols.txt): K4 5.8-6.1 TB/s against 5.4-5.8 for the periodic form, K1 6.2-6.3 against
//  6.1-6.2 for the scale-float4 flat form; the read-only K2 is SLOWER this way with U = 2 (4.3-5.4 against 5.6-5.9: nothing hides
//  the shuffle tree and the barrier at the end of so short a wave) but FASTER with U = 4 (6.0-6.2 against 5.5-5.6:
//  profiles/r02/offgrid/flat_cols_k2_four_float4.txt).  Every other C <= 64 keeps the periodic form (a per-column select +
//  wave-reduction variant for C <= 4 ran at 1.2-2.2 TB/s).
// ------------------------------------------------------------------------------------------
constexpr int kFlatColsBlock = 512;

template <int OP, int NT, int U>
__global__ __launch_bounds__(kFlatColsBlock) void k_flat_cols(Params p, int C, int64_t nv) {
    using O = OpT<OP>;
    constexpr int BS = kFlatColsBlock, NW = BS / 64;
    __shared__ Acc part[O::kReduce ? NW * 64 : 1];
    const int lane = threadIdx.x & 63;
    const int wv = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const int64_t v0 = (int64_t)blockIdx.x * (BS * U) + threadIdx.x;
    float4 x[U], d[U];
    bool valid[U];
#pragma unroll
    for (int u = 0; u < U; ++u) {
        const int64_t v = v0 + (int64_t)u * BS;
        valid[u] = v < nv;
        const int64_t i = (valid[u] ? v : nv - 1) * 4;        // clamp: the loads stay unconditional (nv >= 1 at streaming sizes)
        x[u] = load4<NT>(p.P + i);
        d[u] = x[u];
        if (O::kDy) d[u] = load4<NT>(p.dy + i);
    }
    __builtin_amdgcn_sched_barrier(0);
    const uint32_t uC = (uint32_t)C;
    const uint32_t cblk = (uint32_t)(((int64_t)blockIdx.x * (BS * U) * 4) % C);      // block-uniform
    const uint32_t c0 = (cblk + (uint32_t)threadIdx.x * 4u) & (uC - 1u);             // four fixed, consecutive columns per lane
    Ctx ctx[4];
    Acc acc[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        ctx[k] = O::ctx(p, (int64_t)(c0 + (uint32_t)k) / p.inner);
        acc[k] = O::template init<Acc>();
    }
#pragma unroll
    for (int u = 0; u < U; ++u) {
        const int64_t i = (v0 + (int64_t)u * BS) * 4;
        if (valid[u]) {
            const float4 o = O::elem4c(p, ctx, i, x[u], d[u], acc);
            if (O::kStore) store4<NT>(p.out + i, o);
        }
    }
    if (O::kReduce) {
        const int period = C >> 2;                    // lanes l and l + period share their columns
        for (int off = 32; off >= period; off >>= 1) {
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                Acc o;
                o.a = __shfl_xor(acc[k].a, off, 64);
                o.b = __shfl_xor(acc[k].b, off, 64);
                o.c = __shfl_xor(acc[k].c, off, 64);
                O::merge(acc[k], o);
            }
        }
        if (lane < period) {
#pragma unroll
            for (int k = 0; k < 4; ++k) part[wv * 64 + (int)((c0 + (uint32_t)k) & (uC - 1u))] = acc[k];
        }
        __syncthreads();
        if ((int)threadIdx.x < C) {
            Acc r = part[threadIdx.x];
#pragma unroll
            for (int w = 1; w < NW; ++w) O::merge(r, part[w * 64 + (int)threadIdx.x]);      // fixed wave order
            write_partial_t<OP>(p, (int64_t)blockIdx.x * C + threadIdx.x, r);
        }
    }
}

// ------------------------------------------------------------------------------------------
//  Tiny rows (L <= 64, L % 4 == 0, 16-B aligned bases): a team of lpr = 2^lg <= 16 lanes owns a row (one float4 per
//  lane), a wave takes U passes of 64/lpr consecutive rows with every load issued up front, reduces the U teams'
//  accumulators with DPP (every lane of a team ends with the team total), and then lane (team, u), u < U <= lpr,
//  finishes row (u, team): ONE emit / partial store per wave for U * 64/lpr rows.
//  GM: 0 -> g = row (outer == 1), 1 -> 32-bit row % G, 2 -> 64-bit.
// ------------------------------------------------------------------------------------------
template <int STEPS>
__device__ __forceinline__ void dpp_team_reduce(Acc& acc) {      // team of 2^STEPS lanes inside a 16-lane DPP row
    if (STEPS >= 1) dpp_step<0xB1, 0xf>(acc);     // quad_perm [1,0,3,2]
    if (STEPS >= 2) dpp_step<0x4E, 0xf>(acc);     // quad_perm [2,3,0,1]
    if (STEPS >= 3) dpp_step<0x141, 0xf>(acc);    // row_half_mirror
    if (STEPS >= 4) dpp_step<0x140, 0xf>(acc);    // row_mirror
}

template <int OP, int NT, int U, int LG, int GM>
__global__ __launch_bounds__(kBlock) void k_row_tiny(Params p, int64_t R, int L) {
    using O = OpT<OP>;
    constexpr int lpr = 1 << LG, tpw = 64 >> LG;       // lanes per row, rows per wave and pass
    static_assert(U <= lpr, "one finishing lane per (team, pass)");
    const int lane = threadIdx.x & 63;
    const int wv = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const int team = lane >> LG, li = lane & (lpr - 1);
    const int L4 = L >> 2;
    const int64_t wave_base = ((int64_t)blockIdx.x * kWavesPerBlock + wv) * (tpw * U);
    if (wave_base >= R) return;                        // wave-uniform
    const int lic = li < L4 ? li : 0;
    float4 x[U], d[U];
    int64_t row[U];
#pragma unroll
    for (int u = 0; u < U; ++u) {
        row[u] = wave_base + u * tpw + team;
        const int64_t rr = row[u] < R ? row[u] : R - 1;      // clamp: the loads stay unconditional
        const int64_t i = rr * (int64_t)L + lic * 4;
        x[u] = load4<NT>(p.P + i);
        d[u] = x[u];
        if (O::kDy) d[u] = load4<NT>(p.dy + i);
    }
    __builtin_amdgcn_sched_barrier(0);
    Acc acc[U];
#pragma unroll
    for (int u = 0; u < U; ++u) {
        acc[u] = O::template init<Acc>();
        if (row[u] < R && li < L4) {
            const int64_t g = GM == 0 ? row[u] : (GM == 1 ? (int64_t)((uint32_t)row[u] % (uint32_t)p.G) : row[u] % p.G);
            const Ctx ctx = O::ctx(p, g);
            const int64_t i = row[u] * (int64_t)L + li * 4;
            const float4 r = O::elem4(p, ctx, i, x[u], d[u], acc[u]);
            if (O::kStore) store4<NT>(p.out + i, r);
        }
    }
    if (O::kReduce) {
#pragma unroll
        for (int u = 0; u < U; ++u) dpp_team_reduce<LG>(acc[u]);
        Acc mine = acc[0];
#pragma unroll
        for (int u = 1; u < U; ++u)
            if (li == u) mine = acc[u];
        const int64_t myrow = wave_base + li * tpw + team;
        if (li < U && myrow < R) {
            int64_t idx;
            if (GM == 0) {
                idx = myrow;                            // outer == 1: partial index == row == group
            } else {
                const int64_t o = GM == 1 ? (int64_t)((uint32_t)myrow / (uint32_t)p.G) : myrow / p.G;
                const int64_t g = myrow - o * p.G;
                idx = g * p.outer + o;                  // group-major partials (see row_small_body); direct emit has outer == 1
            }
            write_partial_t<OP>(p, idx, mine);
        }
    }
}

// ------------------------------------------------------------------------------------------
//  Rows of 5..1023 elements that do NOT start on 16-byte lines (L % 4 != 0: 7x7 = 49-element activation planes, rows of
//  1001, ...), scale-gradient ops.  The round-1 row-small kernel walks such rows with 4-byte loads (2.4-4.4 TB/s).  Here a
//  team of 2^LG lanes owns a row and reads the ALIGNED window of float4s that covers it -- [row*L & ~3, (row*L + L + 3) & ~3) --
//  V float4 per lane, every load of the wave's U rows issued up front; the (at most two) edge float4s of a row run the same
//  float4 element path with their outside elements replaced by neutral ones, and store their inside elements one by one, so
//  neighbouring rows never write the same address.  The window's last float4 may reach past the END OF THE TENSOR when
//  numel % 4 != 0: that one float4 is loaded element by element.  Reduction and emit as in k_row_tiny.
// ------------------------------------------------------------------------------------------
template <int LG>
__device__ __forceinline__ void team_reduce_std(Acc& acc) {     // every lane of a 2^LG-lane team <- team total (standard merge)
    if (LG <= 4) {
        dpp_team_reduce<LG>(acc);
    } else {
        dpp_row_reduce(acc);
#pragma unroll
        for (int off = 16; off < (1 << LG); off <<= 1) {
            const uint32_t a = __shfl_xor(acc.a, off, 64), b = __shfl_xor(acc.b, off, 64);
            const double c = __shfl_xor(acc.c, off, 64);
            acc.a = a > acc.a ? a : acc.a;
            acc.b += b;
            acc.c += c;
        }
    }
}

template <class O>
__device__ __forceinline__ float4 apply4(const Params& p, const Ctx& ctx, int64_t i, const float4& x, const float4& d, Acc& acc) {
    float4 r;
    if constexpr (O::kVec4) {
        r = O::elem4(p, ctx, i, x, d, acc);
    } else {
        r.x = O::elem(p, ctx, i + 0, x.x, O::kDy ? d.x : 0.f, acc);
        r.y = O::elem(p, ctx, i + 1, x.y, O::kDy ? d.y : 0.f, acc);
        r.z = O::elem(p, ctx, i + 2, x.z, O::kDy ? d.z : 0.f, acc);
        r.w = O::elem(p, ctx, i + 3, x.w, O::kDy ? d.w : 0.f, acc);
    }
    return r;
}

// rarely executed paths of k_row_win as rolled loops (the kernel is instantiated for 10 geometries; unrolled they are 40 % of its code)
__device__ __forceinline__ float4 load_tail4(const float* base, int64_t i0, int64_t n) {
    float4 t = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma clang loop unroll(disable)
    for (int k = 0; k < 4; ++k) {
        const float v = i0 + k < n ? base[i0 + k] : 0.f;
        t.x = k == 0 ? v : t.x;
        t.y = k == 1 ? v : t.y;
        t.z = k == 2 ? v : t.z;
        t.w = k == 3 ? v : t.w;
    }
    return t;
}
template <class O>
__device__ __forceinline__ void edge_scalar4(const Params& p, const Ctx& ctx, int64_t i0, int64_t b, int64_t e, const float4& x,
                                             const float4& d, Acc& acc) {
#pragma clang loop unroll(disable)
    for (int k = 0; k < 4; ++k) {
        const int64_t i = i0 + k;
        const float xv = k == 0 ? x.x : (k == 1 ? x.y : (k == 2 ? x.z : x.w));
        const float dv = k == 0 ? d.x : (k == 1 ? d.y : (k == 2 ? d.z : d.w));
        if (i >= b && i < e) {
            const float r = O::elem(p, ctx, i, xv, O::kDy ? dv : 0.f, acc);
            if (O::kStore) p.out[i] = r;
        }
    }
}

template <int OP, int NT, int LG, int V, int U>
__global__ __launch_bounds__(kBlock) void k_row_win(Params p, FastDiv fG, int64_t R, int L, int64_t n) {
    using O = OpT<OP>;
    static_assert(O::kStdMerge, "DPP / shuffle team reduction of the standard accumulator");
    constexpr int lpr = 1 << LG, tpw = 64 >> LG;
    static_assert(U <= lpr, "one finishing lane per (team, pass)");
    const int lane = threadIdx.x & 63;
    const int wv = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const int team = lane >> LG, li = lane & (lpr - 1);
    const int64_t wave_base = ((int64_t)blockIdx.x * kWavesPerBlock + wv) * (tpw * U);
    if (wave_base >= R) return;                        // wave-uniform
    float4 x[U][V], d[U][V];
    int64_t row[U], b[U], w0[U];
    int nwin[U];
#pragma unroll
    for (int u = 0; u < U; ++u) {
        row[u] = wave_base + u * tpw + team;
        const int64_t rr = row[u] < R ? row[u] : R - 1;      // clamp: the loads stay unconditional
        b[u] = rr * (int64_t)L;
        w0[u] = b[u] >> 2;
        nwin[u] = (int)(((b[u] + L + 3) >> 2) - w0[u]);
#pragma unroll
        for (int v = 0; v < V; ++v) {
            const int j = li + v * lpr;
            const int64_t i0 = (w0[u] + (j < nwin[u] ? j : 0)) * 4;
            if (__builtin_expect(i0 + 4 <= n, 1)) {
                x[u][v] = load4<NT>(p.P + i0);
                d[u][v] = x[u][v];
                if (O::kDy) d[u][v] = load4<NT>(p.dy + i0);
            } else {                                          // the one float4 that straddles the end of the tensor
                x[u][v] = load_tail4(p.P, i0, n);
                d[u][v] = O::kDy ? load_tail4(p.dy, i0, n) : x[u][v];
            }
        }
    }
    __builtin_amdgcn_sched_barrier(0);
    Acc acc[U];
#pragma unroll
    for (int u = 0; u < U; ++u) {
        acc[u] = O::template init<Acc>();
        if (row[u] < R) {
            const int64_t g = p.outer == 1 ? row[u] : (int64_t)fd_mod(fG, (uint32_t)row[u]);
            const Ctx ctx = O::ctx(p, g);
            const int64_t e = b[u] + L;
            const int ph = (int)(b[u] & 3);               // phase of the row start inside its first float4
            const bool neutral_ok = ctx.fast != 0 && p.lam == p.lam;
#pragma unroll
            for (int v = 0; v < V; ++v) {
                const int j = li + v * lpr;
                bool act = j < nwin[u];
                const int off = j * 4 - ph;                 // offset of this float4's first element from the row start
                const bool full = off >= 0 && off + 4 <= L;
                const int64_t i0 = (w0[u] + j) * 4;
                float4 xx = x[u][v], dd = d[u][v];
                // An edge float4 runs the SAME float4 path with its outside elements made neutral: x = s/2 gives q = 0,
                // out = 0 (max|q| unchanged) and dy = +Inf is "ratio >= lambda" for every lambda, so it casts no vote --
                // valid when s is inside the exact-division window (ctx.fast) and lambda is not NaN; else element by element.
                // Wave-uniform branch: waves without an edge float4 in this slot (all of them when L % 4 == 0) skip it.
                if (__builtin_amdgcn_ballot_w64(act && !full) != 0) {
                    if (act && !full) {
                        if (__builtin_expect(neutral_ok, 1)) {
                            const float xn = 0.5f * ctx.s, dn = __builtin_inff();
                            if (off + 0 < 0 || off + 0 >= L) { xx.x = xn; dd.x = dn; }
                            if (off + 1 < 0 || off + 1 >= L) { xx.y = xn; dd.y = dn; }
                            if (off + 2 < 0 || off + 2 >= L) { xx.z = xn; dd.z = dn; }
                            if (off + 3 < 0 || off + 3 >= L) { xx.w = xn; dd.w = dn; }
                        } else {
                            edge_scalar4<O>(p, ctx, i0, b[u], e, xx, dd, acc[u]);
                            act = false;
                        }
                    }
                }
                if (act) {
                    const float4 r = apply4<O>(p, ctx, i0, xx, dd, acc[u]);
                    if (O::kStore) {
                        if (__builtin_expect(full, 1)) {
#ifdef LQ_DEV_KNOBS
                            if (g_dev_flags & 1) store4<0>(p.out + i0, r);
                            else
#endif
                            store4<NT>(p.out + i0, r);
                        } else {                              // neighbouring rows own the other elements of this float4
                            if (off + 0 >= 0 && off + 0 < L) p.out[i0 + 0] = r.x;
                            if (off + 1 >= 0 && off + 1 < L) p.out[i0 + 1] = r.y;
                            if (off + 2 >= 0 && off + 2 < L) p.out[i0 + 2] = r.z;
                            if (off + 3 >= 0 && off + 3 < L) p.out[i0 + 3] = r.w;
                        }
                    }
                }
            }
        }
    }
    if (O::kReduce) {
#pragma unroll
        for (int u = 0; u < U; ++u) team_reduce_std<LG>(acc[u]);
        Acc mine = acc[0];
#pragma unroll
        for (int u = 1; u < U; ++u)
            if (li == u) mine = acc[u];
        const int64_t myrow = wave_base + li * tpw + team;
        if (li < U && myrow < R) {
            int64_t idx = myrow;                        // outer == 1: partial index == row == group
            if (p.outer != 1) {
                const uint32_t o = fd_div(fG, (uint32_t)myrow), g = (uint32_t)myrow - o * fG.d;
                idx = (int64_t)g * p.outer + o;         // group-major partials (see row_small_body)
            }
            write_partial_t<OP>(p, idx, mine);
        }
    }
}

// ------------------------------------------------------------------------------------------
//  Short rows off the 16-byte grid (5 <= L <= 64, L % 4 != 0: 7x7 activation planes, ...), scale-gradient ops.  A team per
//  row (k_row_win) leaves lanes idle (13 of 16 for L = 49, 5 of 8 for L = 17) and reads the float4 shared by two rows twice.
//  Here a BLOCK owns `rpb` whole consecutive rows -- one contiguous span -- and reads the aligned window that covers it as a
//  flat stream, U float4 per thread, all loads up front.  A float4 lies in at most two rows (L >= 4): each element gets the
//  context of its row (field-wise selects, no arrays of structs), elements outside the span become neutral (see k_row_win),
//  and the thread ends with one accumulator for "its first row" (A) and one for the next (B).  Both go to LDS; thread r then
//  walks the entries of local row r in ascending order -- a fixed order, so the result is run-to-run bit-stable -- and
//  writes ONE partial (or the finished outputs) for the row.
// ------------------------------------------------------------------------------------------
__device__ __forceinline__ Ctx ctx_select(bool b, const Ctx& x, const Ctx& y) {      // b ? x : y, field by field
    Ctx c;
    c.s = b ? x.s : y.s;
    c.r = b ? x.r : y.r;
    c.fast = b ? x.fast : y.fast;
    c.k0 = b ? x.k0 : y.k0;
    c.k1 = b ? x.k1 : y.k1;
    c.lam_hi = b ? x.lam_hi : y.lam_hi;
    c.sure_ok = b ? x.sure_ok : y.sure_ok;
    return c;
}
__device__ __forceinline__ void acc_merge_std(Acc& r, const Acc& o) {
    r.a = o.a > r.a ? o.a : r.a;
    r.b += o.b;
    r.c += o.c;
}

template <int OP, int NT, int U>
__global__ __launch_bounds__(kBlock) void k_row_seg(Params p, FastDiv fL, FastDiv fG, int64_t R, int L, int rpb, int64_t n) {
    using O = OpT<OP>;
    static_assert(O::kStdMerge && O::kVec4c, "standard accumulator, per-element contexts");
    __shared__ Acc lds[O::kReduce ? kBlock * U * 2 : 1];
    const int t = (int)threadIdx.x;
    const int64_t row0 = (int64_t)blockIdx.x * rpb;                      // < R by the grid size
    const int rows = (int)(R - row0 < (int64_t)rpb ? R - row0 : (int64_t)rpb);
    const int64_t b0 = row0 * (int64_t)L;
    const int ph = (int)(b0 & 3);
    const int64_t w0 = b0 >> 2;
    const int span = rows * L;                                           // elements of this block: (ph + span + 3) / 4 <= kBlock * U float4
    const int nwin = (ph + span + 3) >> 2;
    float4 x[U], d[U];
#pragma unroll
    for (int u = 0; u < U; ++u) {
        const int j = t + u * kBlock;
        const int64_t i0 = (w0 + (j < nwin ? j : 0)) * 4;
        if (__builtin_expect(i0 + 4 <= n, 1)) {
            x[u] = load4<NT>(p.P + i0);
            d[u] = x[u];
            if (O::kDy) d[u] = load4<NT>(p.dy + i0);
        } else {                                                         // the one float4 that straddles the end of the tensor
            x[u] = load_tail4(p.P, i0, n);
            d[u] = O::kDy ? load_tail4(p.dy, i0, n) : x[u];
        }
    }
    __builtin_amdgcn_sched_barrier(0);
    const bool lam_ok = p.lam == p.lam;
#pragma unroll
    for (int u = 0; u < U; ++u) {
        const int j = t + u * kBlock;
        Acc aA = O::template init<Acc>(), aB = aA;
        if (j < nwin) {
            const int off = j * 4 - ph;                                  // local index of this float4's first element (>= -3)
            const int e0 = off < 0 ? 0 : off;
            const int rA = (int)fd_div(fL, (uint32_t)e0);                // local row of the first inside element
            const int endA = (rA + 1) * L;
            const bool hasB = off + 3 >= endA && endA < span;            // the float4 reaches into the next row of this block
            const int64_t rowA = row0 + rA;
            const int64_t gA = p.outer == 1 ? rowA : (int64_t)fd_mod(fG, (uint32_t)rowA);
            const Ctx cA = O::ctx(p, gA);
            Ctx cB = cA;
            if (hasB) cB = O::ctx(p, p.outer == 1 ? rowA + 1 : (gA + 1 == p.G ? 0 : gA + 1));
            const int64_t i0 = (w0 + j) * 4;
            const bool inB1 = off + 1 >= endA, inB2 = off + 2 >= endA, inB3 = off + 3 >= endA;      // element 0 is never in B
            const bool v0 = off >= 0 && off < span, v1 = off + 1 >= 0 && off + 1 < span, v2 = off + 2 >= 0 && off + 2 < span,
                       v3 = off + 3 < span;                              // off + 3 >= 0 always
            const bool full = v0 && v3;
            if (__builtin_expect(full || (cA.fast != 0 && cB.fast != 0 && lam_ok), 1)) {
                Ctx cx[4];
                cx[0] = cA;
                cx[1] = ctx_select(inB1, cB, cA);
                cx[2] = ctx_select(inB2, cB, cA);
                cx[3] = ctx_select(inB3, cB, cA);
                float4 xx = x[u], dd = d[u];
                if (!full) {                                             // outside the block's span: neutral elements (see k_row_win)
                    const float dn = __builtin_inff();
                    if (!v0) { xx.x = 0.5f * cx[0].s; dd.x = dn; }
                    if (!v1) { xx.y = 0.5f * cx[1].s; dd.y = dn; }
                    if (!v2) { xx.z = 0.5f * cx[2].s; dd.z = dn; }
                    if (!v3) { xx.w = 0.5f * cx[3].s; dd.w = dn; }
                }
                Acc ac[4];
#pragma unroll
                for (int k = 0; k < 4; ++k) ac[k] = O::template init<Acc>();
                const float4 r = O::elem4c(p, cx, i0, xx, dd, ac);
                const Acc z = O::template init<Acc>();
                aA = ac[0];
                acc_merge_std(aA, inB1 ? z : ac[1]);
                acc_merge_std(aA, inB2 ? z : ac[2]);
                acc_merge_std(aA, inB3 ? z : ac[3]);
                acc_merge_std(aB, inB1 ? ac[1] : z);
                acc_merge_std(aB, inB2 ? ac[2] : z);
                acc_merge_std(aB, inB3 ? ac[3] : z);
                if (O::kStore) {
                    if (__builtin_expect(full, 1)) {
                        store4<NT>(p.out + i0, r);
                    } else {                                             // neighbouring blocks own the other elements
                        if (v0) p.out[i0 + 0] = r.x;
                        if (v1) p.out[i0 + 1] = r.y;
                        if (v2) p.out[i0 + 2] = r.z;
                        if (v3) p.out[i0 + 3] = r.w;
                    }
                }
            } else {                                                     // a scale outside the exact-division window at a block edge
#pragma clang loop unroll(disable)
                for (int k = 0; k < 4; ++k) {
                    const int e = off + k;
                    const float xv = k == 0 ? x[u].x : (k == 1 ? x[u].y : (k == 2 ? x[u].z : x[u].w));
                    const float dv = k == 0 ? d[u].x : (k == 1 ? d[u].y : (k == 2 ? d[u].z : d[u].w));
                    if (e >= 0 && e < span) {
                        const bool inB = e >= endA;
                        const Ctx c = ctx_select(inB, cB, cA);
                        Acc one = O::template init<Acc>();
                        const float r = O::elem(p, c, i0 + k, xv, O::kDy ? dv : 0.f, one);
                        if (inB) acc_merge_std(aB, one);
                        else acc_merge_std(aA, one);
                        if (O::kStore) p.out[i0 + k] = r;
                    }
                }
            }
        }
        if (O::kReduce) {
            lds[j * 2 + 0] = aA;
            lds[j * 2 + 1] = aB;
        }
    }
    if (O::kReduce) {
        __syncthreads();
        for (int r = t; r < rows; r += kBlock) {
            const int e_lo = r * L;
            const int j_lo = (ph + e_lo) >> 2, j_hi = (ph + e_lo + L - 1) >> 2;
            // the first float4 of the row holds it as ITS first row (A) unless it begins in the previous row (then as B)
            const bool firstB = r > 0 && j_lo * 4 - ph < e_lo;
            Acc acc = lds[j_lo * 2 + (firstB ? 1 : 0)];
            for (int j = j_lo + 1; j <= j_hi; ++j) acc_merge_std(acc, lds[j * 2]);
            const int64_t row = row0 + r;
            int64_t idx = row;                          // outer == 1: partial index == row == group
            if (p.outer != 1) {
                const uint32_t o = fd_div(fG, (uint32_t)row), g = (uint32_t)row - o * fG.d;
                idx = (int64_t)g * p.outer + o;         // group-major partials (see row_small_body)
            }
            write_partial_t<OP>(p, idx, acc);
        }
    }
}

}  // namespace lq

#endif
