// lq_traverse.hpp -- the three traversal modes (row stream, row small, column) and the finalize kernels
#ifndef LQ_TRAVERSE_HPP_
#define LQ_TRAVERSE_HPP_
#include "lq_reduce.hpp"

namespace lq {

// ------------------------------------------------------------------------------------------
//  Traversal 1 -- "row stream": rows of length L >= 1024.  One block per (row, chunk) unit of
//  BS*4 elements; every thread owns exactly ONE float4 of each stream (measured on MI355X,
//  tools/membench.hip: this shape with nontemporal accesses streams 2 reads at 6.8 TB/s, read+write
//  at 6.5 TB/s, 2 reads + write at 6.5 TB/s; multi-float4-per-thread loops and persistent blocks
//  are 5-15 % slower).  The scale is block-uniform.  Grid is 3-D (chunk, g, outer) so that no integer
//  division is needed; a 1-D grid with division is the fallback for huge G / outer.
//  VEC = 4: float4 accesses (L % 4 == 0 or a single flat row; 16-B aligned bases).
//  NT: nontemporal loads/stores (streamed-once tensors far larger than the caches).
// ------------------------------------------------------------------------------------------
typedef float v4f __attribute__((ext_vector_type(4)));

template <int NT>
__device__ __forceinline__ float4 load4(const float* p) {
    if (NT) {
        const v4f v = __builtin_nontemporal_load(reinterpret_cast<const v4f*>(p));
        return make_float4(v.x, v.y, v.z, v.w);
    }
    return *reinterpret_cast<const float4*>(p);
}
template <int NT>
__device__ __forceinline__ void store4(float* p, const float4& v) {
    if (NT) {
        const v4f t = {v.x, v.y, v.z, v.w};
        __builtin_nontemporal_store(t, reinterpret_cast<v4f*>(p));
    } else {
        *reinterpret_cast<float4*>(p) = v;
    }
}

// float4 access at an address that is only 4-byte aligned (one global_load_dwordx4 / global_store_dwordx4: global memory
// needs dword alignment only; a wave's 1 KB access then touches 9 lines instead of 8)
typedef float v4f_u __attribute__((ext_vector_type(4), aligned(4)));
template <int NT, int UA>
__device__ __forceinline__ float4 load4x(const float* p) {
    if (!UA) return load4<NT>(p);
    const v4f_u v = NT ? __builtin_nontemporal_load(reinterpret_cast<const v4f_u*>(p)) : *reinterpret_cast<const v4f_u*>(p);
    return make_float4(v.x, v.y, v.z, v.w);
}
template <int NT, int UA>
__device__ __forceinline__ void store4x(float* p, const float4& v) {
    if (!UA) {
        store4<NT>(p, v);
        return;
    }
    const v4f_u t = {v.x, v.y, v.z, v.w};
    if (NT) __builtin_nontemporal_store(t, reinterpret_cast<v4f_u*>(p));
    else *reinterpret_cast<v4f_u*>(p) = t;
}

// TAIL = 1: rows whose last chunk carries a folded tail, or whose chunks do not start on a 16-byte line (scalar head / tail
// elements).  Rows without either (the BENCH shape, rows of 2^k elements, ...) run the TAIL = 0 instantiation, which keeps the
// registers of the round-1 kernel (the hoisted tail loads cost 7-14 VGPRs).
template <int OP, int VEC, int BS, int NT, int U = 1, int TAIL = 1>
__device__ __forceinline__ void row_stream_body(const Params& p, int64_t L, int64_t nc, int64_t row, int64_t ck, int64_t g) {
    using O = OpT<OP>;
    constexpr int CH = BS * 4 * U;
    const int64_t base = row * L + ck * (int64_t)CH;
    const int64_t rem = L - ck * (int64_t)CH;
    // The last chunk of a row takes everything that remains: at most CH, or up to CH + CH/8 when the host folded a short
    // tail into it (row_chunks(): a row of 4100 elements is 2 chunks of 2048 + 2052, not 3 with an almost empty block).
    const int len = (ck == nc - 1) ? (int)rem : CH;
    Acc acc = O::template init<Acc>();

    if (VEC == 4) {
        // Rows whose length is not a multiple of 4 start at any of the 4 phases of a 16-byte line: `head` scalar
        // elements bring the chunk to the next float4 boundary (the base pointers are 16-B aligned on this path), the
        // body is float4, up to 3 elements remain as a tail.  head == 0 whenever L % 4 == 0.
        const int phase = (int)(base & 3);
        int head = (4 - phase) & 3;
        if (head > len) head = len;
        const int64_t vbase = base + head;
        const int vlen = len - head;
        const int len4 = vlen >> 2;
        // Issue the streaming loads FIRST (they depend only on the kernel arguments and the block index);
        // the per-group context (scale fetch, reciprocal, thresholds) is computed while they are in flight.
        // Inactive lanes of a partial chunk re-read float4 0 of the chunk instead of branching.
        // A chunk without a complete float4 (len4 == 0; only the last chunk of a row, so ck > 0)
        // reads the float4 just before it: always in bounds, never used.
        // U = 2 (two float4 per thread and stream): more bytes in flight per wave-lifetime; used by the backward always and
        // by the fused kernel when lambda >= 4e-4 (every element then takes the exact-ratio + tanh branch).
        int64_t i[U];
        float4 x[U], d[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int j = u * BS + (int)threadIdx.x;
            i[u] = vbase + (int64_t)(j < len4 ? j : 0) * 4;
            const int64_t il = len4 > 0 ? i[u] : vbase - 4;
            x[u] = load4<NT>(p.P + il);
            d[u] = x[u];
            if (O::kDy) d[u] = load4<NT>(p.dy + il);
        }
        // The folded tail (one more float4 for a few threads of a row's last chunk) and the scalar head / tail elements of rows
        // that do not start on a 16-byte line are loaded HERE, together with the main loads: as a second and third dependent
        // load round after the main pass they doubled the lifetime of every last-chunk block (rows of 4100 / 4099 elements:
        // K1 5.8, K4 5.3 TB/s against 6.3 / 6.0 for rows of 2048)
        const bool fold = TAIL && len4 > BS * U;          // block-uniform
        const int je = BS * U + (int)threadIdx.x;
        float4 xe = make_float4(0.f, 0.f, 0.f, 0.f), de = xe;
        if (__builtin_expect(fold && je < len4, 0)) {     // only the lanes that own a folded float4 (a few of wave 0) issue these loads
            const int64_t ie0 = vbase + (int64_t)je * 4;
            xe = load4<NT>(p.P + ie0);
            de = xe;
            if (O::kDy) de = load4<NT>(p.dy + ie0);
        }
        const int tail = vlen & 3;
        const int t = (int)threadIdx.x;
        const bool edge = TAIL && (t < head || (t >= 8 && t - 8 < tail));     // threads 0..head-1 and 8..8+tail-1: at most 6 elements per chunk
        const int64_t is = t < head ? base + t : vbase + (int64_t)len4 * 4 + (t - 8);
        float xs = 0.f, dsc = 0.f;
        if (edge) {
            xs = p.P[is];
            if (O::kDy) dsc = p.dy[is];
        }
        __builtin_amdgcn_sched_barrier(0);   // keep the loads ahead of the scale fetch + reciprocal below
        const Ctx ctx = O::ctx(p, g);
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int j = u * BS + (int)threadIdx.x;
            if (j < len4) {
                float4 r;
                if constexpr (O::kVec4) {
                    r = O::elem4(p, ctx, i[u], x[u], d[u], acc);
                } else {
                    r.x = O::elem(p, ctx, i[u] + 0, x[u].x, O::kDy ? d[u].x : 0.f, acc);
                    r.y = O::elem(p, ctx, i[u] + 1, x[u].y, O::kDy ? d[u].y : 0.f, acc);
                    r.z = O::elem(p, ctx, i[u] + 2, x[u].z, O::kDy ? d[u].z : 0.f, acc);
                    r.w = O::elem(p, ctx, i[u] + 3, x[u].w, O::kDy ? d[u].w : 0.f, acc);
                }
                if (O::kStore) store4<NT>(p.out + i[u], r);
            }
        }
        if (__builtin_expect(fold, 0)) {                  // folded tail: fewer than BS*U/8 further float4
            if (je < len4) {
                const int64_t ie = vbase + (int64_t)je * 4;
                float4 r;
                if constexpr (O::kVec4) {
                    r = O::elem4(p, ctx, ie, xe, de, acc);
                } else {
                    r.x = O::elem(p, ctx, ie + 0, xe.x, O::kDy ? de.x : 0.f, acc);
                    r.y = O::elem(p, ctx, ie + 1, xe.y, O::kDy ? de.y : 0.f, acc);
                    r.z = O::elem(p, ctx, ie + 2, xe.z, O::kDy ? de.z : 0.f, acc);
                    r.w = O::elem(p, ctx, ie + 3, xe.w, O::kDy ? de.w : 0.f, acc);
                }
                if (O::kStore) store4<NT>(p.out + ie, r);
            }
        }
        if (edge) {
            float r = O::elem(p, ctx, is, xs, dsc, acc);
            if (O::kStore) p.out[is] = r;
        }
    } else {
        float x[4 * U], d[4 * U];
#pragma unroll
        for (int u = 0; u < 4 * U; ++u) {
            const int j = u * BS + (int)threadIdx.x;
            const int jc = j < len ? j : len - 1;      // clamp instead of predicate: the loads stay in flight together
            x[u] = p.P[base + jc];
            d[u] = O::kDy ? p.dy[base + jc] : 0.f;
        }
        const Ctx ctx = O::ctx(p, g);
#pragma unroll
        for (int u = 0; u < 4 * U; ++u) {
            const int j = u * BS + (int)threadIdx.x;
            if (j < len) {
                float r = O::elem(p, ctx, base + j, x[u], d[u], acc);
                if (O::kStore) p.out[base + j] = r;
            }
        }
        for (int j = 4 * U * BS + (int)threadIdx.x; j < len; j += BS) {      // folded tail
            float r = O::elem(p, ctx, base + j, p.P[base + j], O::kDy ? p.dy[base + j] : 0.f, acc);
            if (O::kStore) p.out[base + j] = r;
        }
    }
    if (O::kReduce) {
        if constexpr (O::kStdMerge) {
            block_reduce_dpp<BS>(acc);
        } else {
            block_reduce<O, Acc, BS>(acc);
        }
        if (threadIdx.x == 0) write_partial_t<OP>(p, row * nc + ck, acc);
    }
}

template <int OP, int VEC, int BS, int NT, int U = 1, int TAIL = 1>
__global__ __launch_bounds__(BS, (BS == 1024 ? 8 : 0)) void k_row_stream(Params p, int64_t L, int64_t nc, int grid3d) {
    int64_t row, ck, g;
    if (grid3d) {
        ck = blockIdx.x;
        g = blockIdx.y;
        row = (int64_t)blockIdx.z * p.G + g;
    } else {
        const int64_t unit = blockIdx.x;
        row = unit / nc;
        ck = unit - row * nc;
        g = row % p.G;
    }
    row_stream_body<OP, VEC, BS, NT, U, TAIL>(p, L, nc, row, ck, g);
}

// ------------------------------------------------------------------------------------------
//  Traversal 2 -- "row small": rows of length L < 1024.  A team of 2^lpr_log2 lanes (<= 64,
//  inside one wave) owns a row; 256 >> lpr_log2 rows per block; one partial per row.
// ------------------------------------------------------------------------------------------
template <int OP, int VEC>
__device__ __forceinline__ void row_small_body(const Params& p, int64_t R, int L, int lpr_log2, int64_t blk) {
    using O = OpT<OP>;
    const int lpr = 1 << lpr_log2;
    const int team = (int)threadIdx.x >> lpr_log2;
    const int lane = (int)threadIdx.x & (lpr - 1);
    const int64_t row = blk * (kBlock >> lpr_log2) + team;
    const bool valid = row < R;
    Acc acc = O::template init<Acc>();
    if (valid) {
        const Ctx ctx = O::ctx(p, row % p.G);
        const int64_t base = row * (int64_t)L;
        if (VEC == 4) {   // L % 4 == 0 and 16-B aligned bases: every row starts on a float4 boundary
            const int L4 = L >> 2;
#pragma unroll 2
            for (int j = lane; j < L4; j += lpr) {
                const int64_t i = base + (int64_t)j * 4;
                const float4 x = *reinterpret_cast<const float4*>(p.P + i);
                float4 d = x;
                if (O::kDy) d = *reinterpret_cast<const float4*>(p.dy + i);
                float4 r;
                if constexpr (O::kVec4) {
                    r = O::elem4(p, ctx, i, x, d, acc);
                } else {
                    r.x = O::elem(p, ctx, i + 0, x.x, O::kDy ? d.x : 0.f, acc);
                    r.y = O::elem(p, ctx, i + 1, x.y, O::kDy ? d.y : 0.f, acc);
                    r.z = O::elem(p, ctx, i + 2, x.z, O::kDy ? d.z : 0.f, acc);
                    r.w = O::elem(p, ctx, i + 3, x.w, O::kDy ? d.w : 0.f, acc);
                }
                if (O::kStore) *reinterpret_cast<float4*>(p.out + i) = r;
            }
        } else {
#pragma unroll 4
            for (int j = lane; j < L; j += lpr) {
                const float x = p.P[base + j];
                const float d = O::kDy ? p.dy[base + j] : 0.f;
                float r = O::elem(p, ctx, base + j, x, d, acc);
                if (O::kStore) p.out[base + j] = r;
            }
        }
    }
    if (O::kReduce) {
        wave_reduce<O>(acc, lpr);   // all 64 lanes execute the shuffles; teams never mix (xor < lpr)
        if (valid && lane == 0) {
            // group-major partials (index g*outer + o for row o*G + g): the finalize of group g then reads one contiguous
            // run instead of every G-th word (16 groups x 16384 rows: 29.5 us of strided gathers)
            const int64_t o = row / p.G, g = row - o * p.G;
            write_partial_t<OP>(p, (p.G > 1 ? g * p.outer + o : row), acc);
        }
    }
}

template <int OP, int VEC>
__global__ __launch_bounds__(kBlock) void k_row_small(Params p, int64_t R, int L, int lpr_log2) {
    row_small_body<OP, VEC>(p, R, L, lpr_log2, (int64_t)blockIdx.x);
}

// ------------------------------------------------------------------------------------------
//  Traversal 3 -- "column": inner < 16 and outer > 1 (column-wise Dense, NHWC per-channel activations).
//  The tensor is a matrix [outer][C], C = G*inner, the group changes along the contiguous axis.
//  A block of 4 waves owns a tile of RB rows x (64*VW) columns; a lane keeps VW fixed columns (VW = 4:
//  one float4 per row, when C % 4 == 0 and the bases are 16-B aligned), so its scales and accumulators
//  are loop-invariant; wave w walks rows w, w+4, ...; the 4 waves' accumulators meet in LDS and one
//  partial per (row-block, column) goes to the workspace.  For C <= 64 a whole wave would cover more
//  than one row: there a wave scans floor(64/C) complete rows per load ("periodic" form, lane -> column
//  lane % C), which keeps 60-64 of the 64 lanes busy for any C.
//  (The first version -- one thread per column walking a slice of rows, 4-B loads, no tiling -- reached
//  1.7 TB/s on a 6144 x 6144 column-wise matrix and 0.25 TB/s on NHWC C = 3.)
// ------------------------------------------------------------------------------------------
constexpr int kColUnroll = 4;

template <class O>
__device__ __forceinline__ void col_cross_wave(Acc* lds, const Acc& mine, int slot, int slots) {
    lds[(threadIdx.x >> 6) * slots + slot] = mine;
}

template <int OP, int VW, int NT>
__device__ __forceinline__ void col_tile_body(const Params& p, int64_t C, int64_t RB, int64_t bx, int64_t by, Acc* lds) {
    using O = OpT<OP>;
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const int64_t col0 = (bx * 64 + lane) * VW;
    const bool active = col0 < C;          // VW == 4 implies C % 4 == 0: a float4 never straddles a row end
    Ctx ctx[VW];
    Acc acc[VW];
#pragma unroll
    for (int k = 0; k < VW; ++k) {
        acc[k] = O::template init<Acc>();
        ctx[k] = O::ctx(p, active ? (col0 + k) / p.inner : 0);
    }
    const int64_t r0 = by * RB;
    const int64_t r1 = (r0 + RB < p.outer) ? r0 + RB : p.outer;
    if (active) {
        for (int64_t r = r0 + w; r < r1; r += 4 * kColUnroll) {
            float x[kColUnroll][VW], d[kColUnroll][VW];
#pragma unroll
            for (int u = 0; u < kColUnroll; ++u) {
                const int64_t rr = (r + 4 * u < r1) ? r + 4 * u : r1 - 1;     // clamp: loads stay unconditional
                const int64_t i = rr * C + col0;
                if (VW == 4) {
                    const float4 v = load4<NT>(p.P + i);
                    x[u][0] = v.x; x[u][1 % VW] = v.y; x[u][2 % VW] = v.z; x[u][3 % VW] = v.w;
                    if (O::kDy) {
                        const float4 e = load4<NT>(p.dy + i);
                        d[u][0] = e.x; d[u][1 % VW] = e.y; d[u][2 % VW] = e.z; d[u][3 % VW] = e.w;
                    }
                } else {
                    x[u][0] = p.P[i];
                    if (O::kDy) d[u][0] = p.dy[i];
                }
            }
#pragma unroll
            for (int u = 0; u < kColUnroll; ++u) {
                if (r + 4 * u < r1) {
                    const int64_t i = (r + 4 * u) * C + col0;
                    if constexpr (VW == 4 && O::kVec4c) {
                        const float4 xv = make_float4(x[u][0], x[u][1 % VW], x[u][2 % VW], x[u][3 % VW]);
                        const float4 dv = O::kDy ? make_float4(d[u][0], d[u][1 % VW], d[u][2 % VW], d[u][3 % VW]) : xv;
                        const float4 ov = O::elem4c(p, ctx, i, xv, dv, acc);
                        if (O::kStore) store4<NT>(p.out + i, ov);
                    } else {
                        float o[VW];
#pragma unroll
                        for (int k = 0; k < VW; ++k) o[k] = O::elem(p, ctx[k], i + k, x[u][k], O::kDy ? d[u][k] : 0.f, acc[k]);
                        if (O::kStore) {
                            if (VW == 4) store4<NT>(p.out + i, make_float4(o[0], o[1 % VW], o[2 % VW], o[3 % VW]));
                            else p.out[i] = o[0];
                        }
                    }
                }
            }
        }
    }
    if (O::kReduce) {
#pragma unroll
        for (int k = 0; k < VW; ++k) lds[w * (64 * VW) + lane * VW + k] = acc[k];
        __syncthreads();
        if (w == 0 && active) {
#pragma unroll
            for (int k = 0; k < VW; ++k) {
                Acc r = lds[lane * VW + k];
#pragma unroll
                for (int ww = 1; ww < 4; ++ww) O::merge(r, lds[ww * (64 * VW) + lane * VW + k]);   // fixed wave order
                write_partial_t<OP>(p, by * C + col0 + k, r);
            }
        }
    }
}

// C <= 64: lane -> (row rl = lane / C, column c = lane % C); a wave reads k = 64 / C whole rows per load.
template <int OP>
__device__ __forceinline__ void col_small_body(const Params& p, int C, int64_t RB, int64_t by, Acc* lds) {
    using O = OpT<OP>;
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const int k = 64 / C;
    const bool active = lane < k * C;
    const int rl = lane / C, c = lane - rl * C;
    const Ctx ctx = O::ctx(p, active ? c / p.inner : 0);
    Acc acc = O::template init<Acc>();
    const int64_t r0 = by * RB;
    const int64_t r1 = (r0 + RB < p.outer) ? r0 + RB : p.outer;
    if (active) {
        for (int64_t r = r0 + (int64_t)w * k + rl; r < r1; r += (int64_t)4 * k * kColUnroll) {
            float x[kColUnroll], d[kColUnroll];
#pragma unroll
            for (int u = 0; u < kColUnroll; ++u) {
                const int64_t rq = r + (int64_t)4 * k * u;
                const int64_t rr = rq < r1 ? rq : r1 - 1;
                x[u] = p.P[rr * C + c];
                d[u] = O::kDy ? p.dy[rr * C + c] : 0.f;
            }
#pragma unroll
            for (int u = 0; u < kColUnroll; ++u) {
                const int64_t rq = r + (int64_t)4 * k * u;
                if (rq < r1) {
                    const int64_t i = rq * C + c;
                    const float o = O::elem(p, ctx, i, x[u], d[u], acc);
                    if (O::kStore) p.out[i] = o;
                }
            }
        }
    }
    if (O::kReduce) {
        lds[threadIdx.x] = acc;
        __syncthreads();
        if ((int)threadIdx.x < C) {
            Acc r = O::template init<Acc>();
            for (int ww = 0; ww < 4; ++ww)
                for (int q = 0; q < k; ++q) O::merge(r, lds[ww * 64 + q * C + (int)threadIdx.x]);      // fixed order
            write_partial_t<OP>(p, by * C + threadIdx.x, r);
        }
    }
}

// C <= 64, streaming sizes: grid-stride over float4 vectors with a thread count T that is a multiple of C.  Thread
// `tid` sees vectors tid, tid + T, ... whose four columns (4*tid + j) % C never change, so -- as in the tile variant --
// scales and accumulators stay in registers and every load is a fully coalesced float4.  `nblk` (a multiple of C)
// blocks; a ragged tail of numel % 4 elements is taken by the thread that would own the next vector.
template <int OP, int NT>
__device__ __forceinline__ void col_periodic4_body(const Params& p, int C, int64_t nblk, int64_t blk, Acc* lds) {
    using O = OpT<OP>;
    constexpr int kPerU = O::kDy ? 2 : 4;      // float4 per stream in flight (measured: one stream wants 4, two streams 2)
    const int64_t numel = p.outer * (int64_t)C;
    const int64_t T = nblk * kBlock;
    const int64_t tid = blk * kBlock + threadIdx.x;
    const int c0 = (int)((tid * 4) % C);
    Ctx ctx[4];
    Acc acc[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        acc[j] = O::template init<Acc>();
        ctx[j] = O::ctx(p, ((c0 + j) % C) / p.inner);
    }
    const int64_t nv = numel >> 2;
    for (int64_t v = tid; v < nv; v += T * kPerU) {
        float4 x[kPerU], d[kPerU];
#pragma unroll
        for (int u = 0; u < kPerU; ++u) {
            const int64_t vv = (v + u * T < nv) ? v + u * T : v;          // clamp: loads stay unconditional
            x[u] = load4<NT>(p.P + vv * 4);
            d[u] = x[u];
            if (O::kDy) d[u] = load4<NT>(p.dy + vv * 4);
        }
#pragma unroll
        for (int u = 0; u < kPerU; ++u) {
            if (v + u * T < nv) {
                const int64_t i = (v + u * T) * 4;
                float4 o;
                if constexpr (O::kVec4c) {
                    o = O::elem4c(p, ctx, i, x[u], d[u], acc);
                } else {
                    o.x = O::elem(p, ctx[0], i + 0, x[u].x, d[u].x, acc[0]);
                    o.y = O::elem(p, ctx[1], i + 1, x[u].y, d[u].y, acc[1]);
                    o.z = O::elem(p, ctx[2], i + 2, x[u].z, d[u].z, acc[2]);
                    o.w = O::elem(p, ctx[3], i + 3, x[u].w, d[u].w, acc[3]);
                }
                if (O::kStore) store4<NT>(p.out + i, o);
            }
        }
    }
    const int rem = (int)(numel & 3);
    if (rem && tid == nv % T) {
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
        Acc* lds2 = lds + kBlock * 4;
#pragma unroll
        for (int j = 0; j < 4; ++j) lds[threadIdx.x * 4 + j] = acc[j];
        __syncthreads();
        // entry e of this block belongs to column (blk*1024 + e) % C; H helpers per column walk them in a fixed order
        const int H = kBlock / C;
        const int c = (int)threadIdx.x % C, h = (int)threadIdx.x / C;
        const int b0 = (int)((blk * (kBlock * 4)) % C);
        if (h < H) {
            int e0 = c - b0;
            if (e0 < 0) e0 += C;
            Acc r = O::template init<Acc>();
            for (int e = e0 + h * C; e < kBlock * 4; e += H * C) O::merge(r, lds[e]);
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

// variant: 0 = periodic (C <= 64), 1 = tile with scalar columns, 4 = tile with float4 (4 columns per lane),
// 5 = float4 tile with nontemporal accesses (tensors >= 64 MiB), 6 / 7 = periodic float4 grid-stride (7: nontemporal)
// One LDS scratch for whichever variant runs (the variants are all inlined into one kernel: private static arrays
// would add up -- 61 KB per block, two blocks per CU -- instead of overlaying).
constexpr int kColLdsAcc = kBlock * 5;
template <int OP, bool P4 = true>      // P4 = false (multi-tensor batches): variants 6 / 7 compiled out, scratch of kBlock*4
__device__ __forceinline__ void col_body(const Params& p, int64_t C, int64_t RB, int64_t nbx, int variant, int64_t b, int64_t nby, Acc* lds) {
    if (variant == 0) {
        col_small_body<OP>(p, (int)C, RB, b, lds);
    } else if (P4 && variant == 6) {
        if constexpr (P4) col_periodic4_body<OP, 0>(p, (int)C, nby, b, lds);
    } else if (P4 && variant == 7) {
        if constexpr (P4) col_periodic4_body<OP, 1>(p, (int)C, nby, b, lds);
    } else {
        const int64_t by = b / nbx, bx = b - by * nbx;
        if (variant == 5) col_tile_body<OP, 4, 1>(p, C, RB, bx, by, lds);
        else if (variant == 4) col_tile_body<OP, 4, 0>(p, C, RB, bx, by, lds);
        else col_tile_body<OP, 1, 0>(p, C, RB, bx, by, lds);
    }
}

template <int OP>
__global__ __launch_bounds__(kBlock) void k_col(Params p, int64_t C, int64_t RB, int64_t nbx, int variant, int64_t nby) {
    __shared__ Acc lds[OpT<OP>::kReduce ? kColLdsAcc : 1];
    col_body<OP>(p, C, RB, nbx, variant, (int64_t)blockIdx.x, nby, lds);
}

// ------------------------------------------------------------------------------------------
//  Finalize: merge the partials of each group in a fixed order (wide accumulator) and emit.
//  Partial index of group g:  g*gstride + i1*stride1 + i2 ,  i1 < n1, i2 < n2.
// ------------------------------------------------------------------------------------------
struct FinGeom {
    int64_t groups;
    int64_t gstride, n1, stride1, n2;
    double count;        // elements per group (outer * inner), or numel for a global reduction
    float* o0;           // op-specific outputs
    float* o1;
    uint32_t* o2;
    int accum;           // OP_DIFF_BWD: add to o0 instead of overwriting (LQ_PENALTY_ACCUMULATE_DS)
};

template <int OP>
struct FinT;

template <>
struct FinT<OP_BWD> {
    __device__ static float emit(const Params& p, const FinGeom& f, int64_t g, const AccW& a) {
        const float maxq = __uint_as_float(a.a);
        float mean;
        if (a.b == 0.0) {
            mean = -1.0f * fabsf(tanhf(p.lam));                 // custom_layers.py:79 / :105
        } else {
            mean = (float)(a.c / f.count);                      // :87 / :113
        }
        const float ds = mean * maxq;                           // :116
        f.o0[g] = ds;
        if (f.o1) {
            f.o1[g] = maxq;
            f.o1[f.groups + g] = mean;
            f.o1[2 * f.groups + g] = (float)a.b;
        }
        return ds;
    }
};
template <>
struct FinT<OP_FUSED> : FinT<OP_BWD> {};
template <>
struct FinT<OP_BWD_PERM> : FinT<OP_BWD> {};

template <>
struct FinT<OP_MAXBIN_FWD> {
    __device__ static float emit(const Params&, const FinGeom& f, int64_t g, const AccW& a) {
        f.o0[g] = __uint_as_float(a.a);
        f.o2[g] = (uint32_t)(a.b > 4294967295.0 ? 4294967295.0 : a.b);
        return 0.f;
    }
};

template <>
struct FinT<OP_DIFF_FWD> {
    __device__ static float emit(const Params&, const FinGeom& f, int64_t g, const AccW& a) {
        f.o0[g] = (float)(a.c / f.count);                       // custom_loss_functions.py:175 reduce_mean
        return 0.f;
    }
};

template <>
struct FinT<OP_DIFF_BWD> {
    __device__ static float emit(const Params&, const FinGeom& f, int64_t g, const AccW& a) {
        const float v = (float)a.c;
        f.o0[g] = f.accum ? f.o0[g] + v : v;
        return 0.f;
    }
};

template <int OP>
__device__ __forceinline__ void emit_direct(const Params& p, int64_t g, const Acc& acc) {
    FinGeom f;
    f.groups = p.G;
    f.gstride = f.n1 = f.stride1 = f.n2 = 0;
    f.count = p.ecount;
    f.o0 = p.e0;
    f.o1 = p.e1;
    f.o2 = nullptr;
    f.accum = 0;
    AccW w;
    w.a = acc.a;
    w.b = (double)acc.b;
    w.c = acc.c;
    FinT<OP>::emit(p, f, g, w);        // same arithmetic as the finalize of a single partial: bit-identical outputs
}

template <class O>
__device__ __forceinline__ AccW load_partial(const Params& p, int64_t idx) {
    AccW w;
    w.a = p.pa[idx];
    w.b = (double)p.pb[idx];
    w.c = p.pc[idx];
    return w;
}

// DPP reduction of the wide standard accumulator (max, add, add); same lane pattern as dpp_wave_reduce.
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ void dpp_step_w(AccW& acc) {
    const uint32_t a = dpp_u32<CTRL, ROW_MASK>(0u, acc.a);
    const double b = dpp_f64<CTRL, ROW_MASK>(acc.b);
    const double c = dpp_f64<CTRL, ROW_MASK>(acc.c);
    acc.a = a > acc.a ? a : acc.a;
    acc.b += b;
    acc.c += c;
}
__device__ __forceinline__ void dpp_row_reduce_w(AccW& acc) {
    dpp_step_w<0xB1, 0xf>(acc);
    dpp_step_w<0x4E, 0xf>(acc);
    dpp_step_w<0x141, 0xf>(acc);
    dpp_step_w<0x140, 0xf>(acc);
}

// One block of BS threads per group.  Index arithmetic is 32-bit whenever the partial count allows (a 64-bit
// division per loaded partial used to dominate this kernel).
template <int OP, int BS>
__device__ __forceinline__ float finalize_block_body(const Params& p, const FinGeom& f, int64_t g, int tid) {     // tid: 0..BS-1; the
    // emitting thread (lane 63 of a one-wave finalize, thread 0 otherwise) returns the value it wrote to o0[g]
    float ret = 0.f;
    using O = OpT<OP>;
    const int64_t n = f.n1 * f.n2;
    const int64_t gbase = g * f.gstride;
    AccW acc = O::template init<AccW>();
    if (n < 0x7fffffffll) {
        const uint32_t n32 = (uint32_t)n, n2 = (uint32_t)f.n2;
        for (uint32_t k = (uint32_t)tid; k < n32; k += BS) {
            const uint32_t i1 = k / n2, i2 = k - i1 * n2;
            O::merge(acc, load_partial<O>(p, gbase + (int64_t)i1 * f.stride1 + i2));
        }
    } else {
        for (int64_t k = tid; k < n; k += BS) {
            const int64_t i1 = k / f.n2, i2 = k - i1 * f.n2;
            O::merge(acc, load_partial<O>(p, gbase + i1 * f.stride1 + i2));
        }
    }
    if constexpr (O::kStdMerge) {
        constexpr int NW = BS / 64;
        __shared__ uint32_t sa[NW];
        __shared__ double sb[NW], sc[NW];
        dpp_row_reduce_w(acc);
        dpp_step_w<0x142, 0xa>(acc);
        dpp_step_w<0x143, 0xc>(acc);
        const int lane = tid & 63, wid = tid >> 6;
        if (NW > 1) {
            if (lane == 63) {
                sa[wid] = acc.a;
                sb[wid] = acc.b;
                sc[wid] = acc.c;
            }
            __syncthreads();
            if (wid == 0) {
                AccW r;
                r.a = lane < NW ? sa[lane] : 0u;
                r.b = lane < NW ? sb[lane] : 0.0;
                r.c = lane < NW ? sc[lane] : 0.0;
                dpp_row_reduce_w(r);
                if (lane == 0) ret = FinT<OP>::emit(p, f, g, r);
            }
        } else if (lane == 63) {
            ret = FinT<OP>::emit(p, f, g, acc);
        }
    } else {
        block_reduce<O, AccW, BS>(acc);
        if (tid == 0) ret = FinT<OP>::emit(p, f, g, acc);
    }
    return ret;
}

template <int OP, int BS>
__global__ __launch_bounds__(BS) void k_finalize_block(Params p, FinGeom f) {
    finalize_block_body<OP, BS>(p, f, (int64_t)blockIdx.x, (int)threadIdx.x);
}

// One thread per group (few partials per group, possibly very many groups).
template <int OP>
__device__ __forceinline__ float finalize_thread_body(const Params& p, const FinGeom& f, int64_t g) {
    using O = OpT<OP>;
    AccW acc = O::template init<AccW>();
    for (int64_t i1 = 0; i1 < f.n1; ++i1)
        for (int64_t i2 = 0; i2 < f.n2; ++i2) O::merge(acc, load_partial<O>(p, g * f.gstride + i1 * f.stride1 + i2));
    return FinT<OP>::emit(p, f, g, acc);
}

// which form finalizes a group with this geometry: 0 = a thread, 1 = a wave (64), 2 = 256 threads, 3 = 1024 threads
__host__ __device__ inline int finalize_form(int64_t groups, int64_t n1, int64_t stride1, int64_t n2) {
    const int64_t n = n1 * n2;
    // one thread per group walks its partials one after the other: right for very many groups (throughput) or a handful of
    // partials, a latency trap otherwise (32 partials: ~10 us) -- few groups get one wave each instead
    if (n <= 4 || (n <= 32 && groups >= 2048)) return 0;
    (void)stride1;
    return n <= 256 ? 1 : (n <= 1024 ? 2 : 3);
}

// Column form: the partials form a matrix [n1][groups * n2] (a traversal that leaves one partial per (row block, column), or per
// (layer, group, chunk)) and group g owns the n2 adjacent columns g*n2 ... of every row.  A wave or a thread per group would
// gather its words `stride1` apart -- a whole line for every 4-byte word (the finalize of the ResNet-50-like batch read 88 MB for
// 20 MB of partials; a 4096 x 4096 column-wise scale gradient spent 10 of 44 us here).  A 256-thread block instead takes
// floor(64 / n2) groups: lane l of every wave owns column c0 + l, so every load is a contiguous run of a partial row; wave w walks
// rows w, w + 4, ... with eight or sixteen partials in flight per thread; the 4 x n2 totals of a group meet in LDS, merged in a fixed order.
__host__ __device__ inline bool finalize_cols_ok(int64_t groups, int64_t gstride, int64_t n1, int64_t stride1, int64_t n2) {
    return n1 > 1 && n2 >= 1 && n2 <= 64 && gstride == n2 && stride1 == groups * n2 && groups * n2 >= 64;
}

// wave `wv` of four merges rows wv, wv + 4, ... of column `col`, AH partials in flight per thread (clamped, unconditional loads)
template <int OP, int AH>
__device__ __forceinline__ void finalize_cols_walk(const Params& p, int64_t col, int64_t C, int64_t n1, int wv, AccW& acc) {
    using O = OpT<OP>;
    for (int64_t i = wv; i < n1; i += 4 * AH) {
        AccW v[AH];
#pragma unroll
        for (int u = 0; u < AH; ++u) v[u] = load_partial<O>(p, col + (i + 4 * u < n1 ? i + 4 * u : i) * C);
#pragma unroll
        for (int u = 0; u < AH; ++u)
            if (i + 4 * u < n1) O::merge(acc, v[u]);
    }
}

// returns true in the threads that hold a finished group (`g`, `acc`); `lds`: 256 AccW
template <int OP>
__device__ __forceinline__ bool finalize_cols_body(const Params& p, const FinGeom& f, int64_t g0, AccW* lds, int64_t& g, AccW& acc) {
    using O = OpT<OP>;
    const int inner = (int)f.n2, gpb = 64 / inner;
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const int64_t col = g0 * inner + lane, C = f.stride1, n1 = f.n1;
    acc = O::template init<AccW>();
    if (lane < gpb * inner && col < C) {
        // 64 row blocks are one dependent round with sixteen partials in flight; up to 32 row blocks eight do (the clamped
        // loads of rows that do not exist cost issue slots and L2 reads: 1.3 us on the ResNet-18-like set)
        if (n1 <= 32) finalize_cols_walk<OP, 8>(p, col, C, n1, wv, acc);
        else finalize_cols_walk<OP, 16>(p, col, C, n1, wv, acc);
    }
    lds[threadIdx.x] = acc;
    __syncthreads();
    g = g0 + threadIdx.x;
    const bool emits = (int)threadIdx.x < gpb && g < f.groups;
    if (emits) {
        acc = O::template init<AccW>();
        for (int k = 0; k < inner; ++k)
            for (int ww = 0; ww < 4; ++ww) O::merge(acc, lds[ww * 64 + threadIdx.x * inner + k]);      // fixed order
    }
    return emits;
}

template <int OP>
__global__ __launch_bounds__(kBlock) void k_finalize_cols(Params p, FinGeom f) {
    __shared__ AccW lds[kBlock];
    int64_t g;
    AccW acc;
    if (finalize_cols_body<OP>(p, f, (int64_t)blockIdx.x * (64 / (int)f.n2), lds, g, acc)) FinT<OP>::emit(p, f, g, acc);
}

template <int OP>
__global__ __launch_bounds__(kBlock) void k_finalize_thread(Params p, FinGeom f) {
    const int64_t g = (int64_t)blockIdx.x * kBlock + threadIdx.x;
    if (g >= f.groups) return;
    finalize_thread_body<OP>(p, f, g);
}

}  // namespace lq

#endif
