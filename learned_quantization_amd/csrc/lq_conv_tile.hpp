// lq_conv_tile.hpp -- conv kernels (HWIO parameter, OIHW consumer) as LDS tiles (round 3)
//
// The reference keeps conv kernels in HWIO (custom_layers.py:321) and MIOpen consumes OIHW.  Round 2 emitted the OIHW
// companion of K1 (and gathered K2's upstream gradient) element by element: 4-byte accesses 4*ci*hw bytes apart.
// rocprofv3 on the ResNet-18-like weight set with the round-2 sources (profiles/r03/base_r02_*, base_r02_oihw_*): forward
// 66.6 us with 184.6 MB written for 89.4 MB of output (every 128-byte line of the companion reaches memory in pieces),
// gathered backward 59.8 us -- 2.5-3x the plain traversals.  Here a block owns a TILE
//
//      c in [c0, c0 + 32 m)   x   o in [o0, o0 + 32)   x   every h = (kh, kw)            32 m hw <= 288 elements per o
//
// of the kernel.  In HWIO, element (h, c, o) sits at ((h ci + c) co + o): for every (h, c) the tile holds one 128-byte
// line (32 consecutive o).  In OIHW it sits at ((o ci + c) hw + h): for every o the tile holds ONE contiguous run of
// 32 m hw floats (1152 bytes = 9 whole lines for a 3x3 kernel).  Both sides are therefore read and written in whole lines:
//
//   HWIO side   wave w owns the 8 m channels [8 m w, 8 m (w + 1)) of the tile; its lane (c_lw = lane / 8, o4 = lane % 8) owns the
//               float4 at o0 + 4 o4 of row (h, c) in pass p = (j, h), c = c0 + 8 m w + 8 j + c_lw; every pass's loads are
//               issued up front (<= 9 passes, all independent);
//   LDS         per WAVE, no block barrier (a wave's LDS accesses execute in order): its slice transposed, word [o_l][k],
//               k = (c - first channel of the wave) hw + h < 72, row stride 73 (odd): the 4-byte transposed accesses of a
//               wave (8 channels x 8 float4 columns) and the run-order accesses are both conflict-free;
//   OIHW side   the wave walks its runs (72 floats = 288 bytes per o; the block's four waves together cover the 1152-byte
//               run of the tile) in run order, e = lane + 64 i: consecutive lanes -> consecutive addresses.
//
// K1 writes `out` from registers and the companion through LDS; K2 takes dy from the OIHW gradient through LDS (or, for the
// plain entry points, from an HWIO gradient directly), writes dP in HWIO order and accumulates the vote.  The group of an
// element depends on (h, c) only for every orientation the reference has (custom_layers.py:147-197 on an HWIO kernel):
//   kind 0   channelwise: g = c           -- the 8 lanes that share a row reduce with DPP, one partial per (c, o-tile);
//   kind 1   rowwise / columnwise / scalar: g = (h / A) % G -- uniform over the block; the passes are visited group by group
//            and each wave leaves one partial per (group, tile, wave).
// Vote sums are exact (lq_common.hpp, Acc), so ds does not depend on this choice of partials: bit-identical to lq_fq_scale_grad.
//
// Measured with these tiles (profiles/r03/batch_*, quick*): forward 23.3 us (20.0 us without the optional HWIO output), OIHW
// scale gradient 32.4 us (30.1 us since the run-order offsets are carried from access to access, profiles/r03/experiments/ring/).  K2 holds 37 KB of LDS and 113 VGPRs per block -- four blocks per CU, 1024 of the set's 1250 tiles at a
// time; the 226 others start as the first finish and cost 10 of the 32 us (profiles/r03/timelines/).  Two attempts at that tail
// were measured and dropped, patches and numbers kept under profiles/r03/experiments/: the tile through LDS in two halves (five
// blocks per CU, every tile resident: 33-34 us -- each wave then pays two dependent gradient round trips) and half-size tiles for
// the tasks that start last (34 us -- a half-size block lives as long as a full one).  The kernel is bound by a wave's latency
// chain, not by residency.
#ifndef LQ_CONV_TILE_HPP_
#define LQ_CONV_TILE_HPP_
#include "lq_stream2.hpp"

namespace lq {

// development builds (make dev): lq_dev_set_ablate(mask) switches parts of the tile kernels off, to see what each costs
// (1: the element arithmetic, 2: the OIHW side, 4: the HWIO stores, 8: flush / partial stores).  Compiled out of the product.
#ifdef LQ_DEV_KNOBS
__device__ int g_ablate = 0;
#define LQ_ABLATE(bit) ((g_ablate & (bit)) != 0)
// lq_dev_set_trace(buf): every block of a batch traversal records wall_clock64() (100 MHz) at its start and end in
// buf[2 * blockIdx.x], buf[2 * blockIdx.x + 1], and its XCC / CU / SIMD id word in buf's second half -- the block timeline
__device__ unsigned long long* g_trace = nullptr;
#define LQ_TRACE(slot) do { if (g_trace && threadIdx.x == 0) g_trace[2 * blockIdx.x + (slot)] = wall_clock64(); } while (0)
#else
#define LQ_ABLATE(bit) false
#define LQ_TRACE(slot) do { } while (0)
#endif

constexpr int kCtO = 32;                       // output channels per tile
constexpr int kCtPass = 9;                     // passes of the HWIO side: m hw <= 9
constexpr int kCtRunW = 72;                    // (c, h) pairs per wave: 8 m hw <= 72
constexpr int kCtStrideW = kCtRunW + 1;        // LDS row stride in words (odd: the transposed accesses are conflict-free)
constexpr int kCtFwdStage = 8;                 // K1 stages 8 output channels at a time: 4 waves x 8 x 73 words = 9.3 KB per block
constexpr int kCtLdsWordsFwd = kWavesPerBlock * kCtFwdStage * kCtStrideW;
constexpr int kCtLdsWordsBwd = kWavesPerBlock * kCtO * kCtStrideW;          // K2 (OIHW gradient): the wave's whole tile, 37 KB per block

struct ConvTile {
    uint32_t hw, ci, co;
    uint32_t tc;                // input channels per tile (32 m): wave w owns channels [8 m w, 8 m (w + 1)) of it
    uint32_t nto, ntc;          // tiles along co / along ci
    uint32_t npass;             // m hw
    uint32_t kind;              // 0: g = c, 1: g = pass_g[p]
    FastDiv fnto;               // block -> (c-tile, o-tile)
    // Per-pass tables, one DWORD per entry: the rolled loop of K2 indexes them with its (uniform) pass counter, which must
    // compile to scalar loads (lgkmcnt).  Sub-dword entries are fetched with VECTOR loads followed by s_waitcnt vmcnt(0) --
    // a wait for every tile load in flight, once per pass.
    uint32_t pass_info[kCtPass + 1];    // bits 0-7 pass_c (channel offset inside the wave's slice, multiple of 8), 8-15 pass_h,
                                        // 16-23 pass_g (kind 1: group of the pass), 24 pass_new (group differs from pass p - 1: flush)
    uint32_t pass_off[kCtPass + 1];     // (pass_h ci + pass_c) co: HWIO offset of pass p relative to the lane's first row
};
__device__ __forceinline__ uint32_t ct_pass_c(const ConvTile& ct, int q) { return ct.pass_info[q] & 0xffu; }
__device__ __forceinline__ uint32_t ct_pass_h(const ConvTile& ct, int q) { return (ct.pass_info[q] >> 8) & 0xffu; }
__device__ __forceinline__ uint32_t ct_pass_g(const ConvTile& ct, int q) { return (ct.pass_info[q] >> 16) & 0xffu; }
__device__ __forceinline__ bool ct_pass_new(const ConvTile& ct, int q) { return (ct.pass_info[q] >> 24) != 0u; }
__device__ __forceinline__ uint32_t ct_pass_k(const ConvTile& ct, int q) { return ct_pass_c(ct, q) * ct.hw + ct_pass_h(ct, q); }

struct CtGeom {             // what a thread needs to know about its tile
    uint32_t c0, o0, tc_eff, to_eff, to, tile;
    uint32_t w, lane, c_lw, o4;     // wave, lane, channel row of the lane inside a pass (0..7), float4 column (0..7)
    uint32_t wc0, cw, RLw, M;       // first channel of the wave's slice (relative to c0), its channel count, run length cw * hw, magic
    bool ov;
};

__device__ __forceinline__ CtGeom ct_geom(const ConvTile& ct, uint32_t b) {
    CtGeom g;
    const uint32_t tci = fd_div(ct.fnto, b);
    // o-tile rotated by the c-tile index: consecutive blocks go to consecutive XCDs, and with nto a multiple of 8 an XCD would
    // otherwise only ever see o-tiles x and x + 8 -- two of the 16 values of (address / 128) % 16 in its L2
    g.to = b - tci * ct.nto + tci;
    g.to = g.to >= ct.nto ? g.to - ct.nto * fd_div(ct.fnto, g.to) : g.to;
    g.tile = b;
    g.c0 = tci * ct.tc;
    g.o0 = g.to * kCtO;
    g.tc_eff = ct.ci - g.c0 < ct.tc ? ct.ci - g.c0 : ct.tc;
    g.to_eff = ct.co - g.o0 < (uint32_t)kCtO ? ct.co - g.o0 : (uint32_t)kCtO;
    g.w = (uint32_t)__builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    g.lane = threadIdx.x & 63;
    g.c_lw = g.lane >> 3;
    g.o4 = g.lane & 7;
    g.ov = 4 * g.o4 < g.to_eff;
    const uint32_t m8 = ct.tc >> 2;
    g.wc0 = g.w * m8;
    g.cw = g.tc_eff > g.wc0 ? (g.tc_eff - g.wc0 < m8 ? g.tc_eff - g.wc0 : m8) : 0u;
    g.RLw = g.cw * ct.hw;
    // e / RLw for e < 32 * 72 as (e * M) >> 20 with M = floor(2^20 / RLw) + 1: exact while e * RLw < 2^20 (2304 * 72 = 165 888)
    g.M = g.RLw ? (1u << 20) / g.RLw + 1u : 0u;
    return g;
}

// Run-order walk of a wave's slice, e = lane + 64 i -> (o_l, k) = divmod(e, RLw): the starting offsets of a lane and what one
// step of 64 elements adds to the byte offset in an OIHW tensor (row stride `ostride` floats) and in the wave's LDS slice
// (row stride kCtStrideW words), without / with a wrap of k.
struct CtStep {
    uint32_t sr, k0, g0, l0, ginc0, ginc1, linc0, linc1;
};
__device__ __forceinline__ CtStep ct_step(const CtGeom& g, uint32_t ostride) {
    CtStep s;
    const uint32_t sq = (64u * g.M) >> 20;                         // 64 / RLw (exact: see ct_geom)
    s.sr = 64u - sq * g.RLw;                                       // 64 % RLw
    const uint32_t o0 = (g.lane * g.M) >> 20;
    s.k0 = g.lane - o0 * g.RLw;
    s.g0 = (o0 * ostride + s.k0) * 4u;
    s.l0 = (o0 * (uint32_t)kCtStrideW + s.k0) * 4u;
    s.ginc0 = (sq * ostride + s.sr) * 4u;
    s.ginc1 = s.ginc0 + (ostride - g.RLw) * 4u;
    s.linc0 = (sq * (uint32_t)kCtStrideW + s.sr) * 4u;
    s.linc1 = s.linc0 + ((uint32_t)kCtStrideW - g.RLw) * 4u;
    return s;
}

// ------------------------------------------------------------------------------------------
//  K1 on a tile: out (HWIO) from registers, out_perm (OIHW) through the wave's LDS slice when p.out_perm is set.
//  No block barrier: a wave transposes its own 8 m channels (LDS executes a wave's accesses in order), kCtFwdStage output
//  channels at a time.
// ------------------------------------------------------------------------------------------
__device__ __forceinline__ void conv_tile_fwd(const Params& p, const ConvTile& ct, uint32_t b, float* lds) {
    using O = OpT<OP_FWD>;
    const CtGeom g = ct_geom(ct, b);
    // Every load is UNCONDITIONAL: a lane without an element in pass q re-reads the tile's first float4 (always in bounds,
    // never used).  Predicated loads compile to branches with s_waitcnt vmcnt(0) between them -- one memory round trip per
    // load instead of one per tile.
    float4 x[kCtPass];
    bool v[kCtPass];
    const uint32_t idx_safe = g.c0 * ct.co + g.o0;
#pragma unroll
    for (int q = 0; q < kCtPass; ++q) {
        const uint32_t crel = g.wc0 + ct_pass_c(ct, q) + g.c_lw;
        v[q] = q < (int)ct.npass && crel < g.tc_eff && g.ov;
        const uint32_t idx = (ct_pass_h(ct, q) * ct.ci + g.c0 + crel) * ct.co + g.o0 + 4 * g.o4;
        x[q] = load4<0>(p.P + (v[q] ? idx : idx_safe));
    }
    // the scale of every pass is fetched NOW, with the tile's loads: fetched inside the pass loop, each new channel block of a
    // 1x1 kernel (m = 9: nine different channels per lane) stalled the wave for a memory round trip
    float sc[kCtPass];
#pragma unroll
    for (int q = 0; q < kCtPass; ++q) {
        const uint32_t gq = ct.kind == 0 ? g.c0 + g.wc0 + ct_pass_c(ct, q) + g.c_lw : ct_pass_g(ct, q);
        sc[q] = p.s[v[q] ? gq : 0u];
    }
    __builtin_amdgcn_sched_barrier(0);            // every load of the tile is in flight before the first use
    Acc none = O::template init<Acc>();
    Ctx ctx = O::ctx_of(p, 1.0f);
#pragma unroll
    for (int q = 0; q < kCtPass; ++q) {
        if (q < (int)ct.npass) {                    // block-uniform
            if (v[q]) {
                const uint32_t crel = g.wc0 + ct_pass_c(ct, q) + g.c_lw;
                const uint32_t idx = (ct_pass_h(ct, q) * ct.ci + g.c0 + crel) * ct.co + g.o0 + 4 * g.o4;
                if (q == 0 || ct_pass_new(ct, q)) ctx = O::ctx_of(p, sc[q]);      // block-uniform condition
                if (!LQ_ABLATE(1)) x[q] = O::elem4(p, ctx, idx, x[q], x[q], none);      // the outputs take the inputs' registers
                if (p.out && !LQ_ABLATE(4)) store4<0>(p.out + idx, x[q]);        // block-uniform: the HWIO output is optional
            }
        }
    }
    if (p.out_perm && g.RLw && !LQ_ABLATE(2)) {
        float* lw = lds + g.w * (kCtFwdStage * kCtStrideW);
        float* dst = p.out_perm + ((size_t)g.o0 * ct.ci + g.c0 + g.wc0) * ct.hw;
        const uint32_t ostride = ct.ci * ct.hw;
        const CtStep st = ct_step(g, ostride);                  // carried offsets of the run-order walk (see conv_tile_bwd)
#pragma unroll 1
        for (uint32_t r = 0; r < (uint32_t)(kCtO / kCtFwdStage); ++r) {
            if (r * kCtFwdStage >= g.to_eff) break;             // block-uniform
            if ((g.o4 >> 1) == r) {                             // the lanes whose four output channels belong to this stage
                float* wr = lw + ((4 * g.o4) & (kCtFwdStage - 1)) * kCtStrideW;
#pragma unroll
                for (int q = 0; q < kCtPass; ++q) {
                    if (v[q]) {
                        const uint32_t k = (ct_pass_c(ct, q) + g.c_lw) * ct.hw + ct_pass_h(ct, q);
                        wr[k] = x[q].x;
                        wr[k + kCtStrideW] = x[q].y;
                        wr[k + 2 * kCtStrideW] = x[q].z;
                        wr[k + 3 * kCtStrideW] = x[q].w;
                    }
                }
            }
            __builtin_amdgcn_wave_barrier();
            const uint32_t no = g.to_eff - r * kCtFwdStage < (uint32_t)kCtFwdStage ? g.to_eff - r * kCtFwdStage : (uint32_t)kCtFwdStage;
            const uint32_t E = no * g.RLw;
            {
                uint32_t k = st.k0, goff = st.g0 + r * (uint32_t)kCtFwdStage * ostride * 4u, loff = st.l0;
                char* dstb = reinterpret_cast<char*>(dst);
                const char* lwb = reinterpret_cast<const char*>(lw);
#pragma unroll
                for (int i = 0; i < kCtFwdStage * kCtRunW / 64; ++i) {
                    const uint32_t e = g.lane + 64u * (uint32_t)i;
                    if (e < E) *reinterpret_cast<float*>(dstb + goff) = *reinterpret_cast<const float*>(lwb + loff);
                    k += st.sr;
                    const bool wrap = k >= g.RLw;
                    k -= wrap ? g.RLw : 0u;
                    goff += wrap ? st.ginc1 : st.ginc0;
                    loff += wrap ? st.linc1 : st.linc0;
                }
            }
            __builtin_amdgcn_wave_barrier();
        }
    }
}

// ------------------------------------------------------------------------------------------
//  K2 on a tile: dy arrives in OIHW order (p.dy_perm), goes through the wave's LDS slice and is written back in HWIO order
//  to p.dp_out (dP == dy, custom_layers.py:118).  (An HWIO gradient needs no tile: the plain entry points keep the generic
//  traversals -- measured as fast on the ResNet-18-like set, a quarter faster on the ResNet-50-like one.)
// ------------------------------------------------------------------------------------------
template <int OP>
__device__ __forceinline__ void ct_flush(const Params& p, const ConvTile& ct, const CtGeom& g, Acc& acc, uint32_t gq, bool mine) {
    if (LQ_ABLATE(8)) {
        if (acc.c == 12345.0) write_partial(p, 0, acc);
        return;
    }
    if (ct.kind == 0) {
        dpp_team_reduce<3>(acc);                      // the 8 lanes of a row (same c): every lane ends with the row total
        if (mine && g.o4 == 0) write_partial(p, (int64_t)gq * ct.nto + g.to, acc);
    } else {
        dpp_wave_reduce(acc);                         // lane 63 <- wave total
        if (g.lane == 63) write_partial(p, ((int64_t)gq * (ct.nto * ct.ntc) + g.tile) * kWavesPerBlock + g.w, acc);
    }
}

// The pass loop is ROLLED and software-pipelined with a ring of kCtAhead passes in flight.  Block timelines (tools/
// block_timeline.py) of the first, unrolled version -- nine copies of the vote arithmetic, 10 000 instructions, 140 VGPRs,
// three waves per SIMD -- showed every block alive for the whole launch; with two passes in flight a block that starts late
// (the second round of a launch with more tiles than the chip holds) needed nine dependent memory round trips: 15 us alone
// on an idle chip.
constexpr int kCtAhead = 4;      // (six and nine passes ahead, with the registers the carried offsets freed: 31.9 / 33.3 us against 30.1)

__device__ __forceinline__ void conv_tile_bwd(const Params& p, const ConvTile& ct, uint32_t b, float* lds) {
    using O = OpT<OP_BWD>;
    const CtGeom g = ct_geom(ct, b);
    float* lw = lds + g.w * (kCtO * kCtStrideW);
    const uint32_t tconst = (g.c0 + g.wc0 + g.c_lw) * ct.co + g.o0 + 4 * g.o4;      // + pass_off[q] = HWIO index of the lane's float4
    const uint32_t climit = g.tc_eff > g.wc0 + g.c_lw ? g.tc_eff - g.wc0 - g.c_lw : 0u;   // pass q is this lane's iff pass_c[q] < climit
    const int np = (int)ct.npass;
    auto valid = [&](int q) { return g.ov && ct_pass_c(ct, q) < climit; };
    auto group = [&](int q) { return ct.kind == 0 ? g.c0 + g.wc0 + g.c_lw + ct_pass_c(ct, q) : ct_pass_g(ct, q); };
    // ring of the next kCtAhead passes: P and the scale of the pass's group (a scale fetched inside the loop stalls the wave)
    float4 x[kCtAhead];
    float sc[kCtAhead];
    // every load is unconditional (clamped to the tile's first float4 / scale 0): see conv_tile_fwd
    const uint32_t idx_safe = g.c0 * ct.co + g.o0;
#pragma unroll
    for (int a = 0; a < kCtAhead; ++a) {
        const bool va = a < np && valid(a);
        x[a] = load4<0>(p.P + (va ? tconst + ct.pass_off[a] : idx_safe));
        sc[a] = p.s[va ? group(a) : 0u];
    }
    if (!LQ_ABLATE(2)) {
        // the wave's slice of the OIHW gradient into LDS in run order: 36 independent 256-byte loads, all in flight at once
        // (in four rounds of nine every wave paid four memory round trips before its first pass)
        const uint32_t E = g.to_eff * g.RLw;
        // (a wave without channels -- the last c-tile of a kernel with few input channels -- reads element 0 of the tensor)
        const float* src = E ? p.dy_perm + ((size_t)g.o0 * ct.ci + g.c0 + g.wc0) * ct.hw : p.dy_perm;
        const uint32_t ostride = ct.ci * ct.hw;
        constexpr int NI = kCtO * kCtRunW / 64;      // 36
        float t[NI];
        // Element e = lane + 64 i sits at (o_l, k) = divmod(e, RLw).  Both byte offsets -- (o_l ostride + k) 4 from the uniform
        // gradient base, (o_l 73 + k) 4 into the LDS slice -- are CARRIED from one access to the next (64 = sq RLw + sr: k += sr
        // with one conditional wrap, 4 VALU per access) instead of being formed from e by a magic division and 64-bit
        // address arithmetic each time (9 + 7): a quarter of the kernel's vector instructions went there.
        const CtStep st = ct_step(g, ostride);
        {
            uint32_t k = st.k0, goff = st.g0;
#pragma unroll
            for (int i = 0; i < NI; ++i) {
                const uint32_t e = g.lane + 64u * (uint32_t)i;
                // clamped to the slice's first word: the loads stay unconditional (E > 0 whenever a lane is valid)
                t[i] = *reinterpret_cast<const float*>(reinterpret_cast<const char*>(src) + (e < E ? goff : 0u));
                k += st.sr;
                const bool wrap = k >= g.RLw;
                k -= wrap ? g.RLw : 0u;
                goff += wrap ? st.ginc1 : st.ginc0;
            }
        }
        {
            uint32_t k = st.k0, loff = st.l0;
            char* lwb = reinterpret_cast<char*>(lw);
#pragma unroll
            for (int i = 0; i < NI; ++i) {
                const uint32_t e = g.lane + 64u * (uint32_t)i;
                // unconditional: a word past the slice's data goes to the pad word of row 0 (column 72 is never read) -- a
                // predicated write costs a compare, an exec save / restore and a branch each
                *reinterpret_cast<float*>(lwb + (e < E ? loff : (uint32_t)kCtRunW * 4u)) = t[i];
                k += st.sr;
                const bool wrap = k >= g.RLw;
                k -= wrap ? g.RLw : 0u;
                loff += wrap ? st.linc1 : st.linc0;
            }
        }
        __builtin_amdgcn_wave_barrier();
    }
    Acc acc = O::template init<Acc>();
    Ctx ctx = O::ctx_of(p, 1.0f);
    uint32_t gcur = 0;
    bool mine = false;                                // this lane accumulated something for the current group
    const float* lr = lw + (4 * g.o4) * kCtStrideW + g.c_lw * ct.hw;
    // slot a of the ring serves passes a, a + kCtAhead, ...: it is refilled right after its pass was computed, so a pass's
    // loads are issued kCtAhead - 1 passes before their use.  (A ring that SHIFTS its registers every pass makes the compiler
    // wait for the newest load at the copy: no pipelining at all -- the first version of this loop.)
#pragma unroll 1
    for (int q0 = 0; q0 < np; q0 += kCtAhead) {
#pragma unroll
        for (int a = 0; a < kCtAhead; ++a) {
            const int q = q0 + a;
            if (q < np) {                                 // block-uniform
                if (q == 0 || ct_pass_new(ct, q)) {           // block-uniform: a new group (kind 1) / a new channel block (kind 0)
                    if (q > 0) {
                        ct_flush<OP_BWD>(p, ct, g, acc, gcur, mine);
                        acc = O::template init<Acc>();
                        mine = false;
                    }
                    gcur = group(q);
                    // kind 1: the group is block-uniform, but an idle lane carries no scale: it takes a valid lane's
                    float sq = sc[a];
                    if (ct.kind == 1) sq = __shfl(sc[a], __builtin_ctzll(__builtin_amdgcn_ballot_w64(valid(q)) | (1ull << 63)), 64);
                    ctx = O::ctx_of(p, sq);
                }
                if (valid(q)) {
                    const uint32_t idx = tconst + ct.pass_off[q];
                    mine = true;
                    const float* r = lr + ct_pass_k(ct, q);
                    float4 d = x[a];
                    if (!LQ_ABLATE(2)) d = make_float4(r[0], r[kCtStrideW], r[2 * kCtStrideW], r[3 * kCtStrideW]);
                    if (!LQ_ABLATE(4)) store4<0>(p.dp_out + idx, d);
                    if (!LQ_ABLATE(1)) O::elem4(p, ctx, idx, x[a], d, acc);
                    else acc.c += (double)(x[a].x + d.x);
                }
                if (q + kCtAhead < np) {                  // block-uniform
                    const bool vn = valid(q + kCtAhead);
                    x[a] = load4<0>(p.P + (vn ? tconst + ct.pass_off[q + kCtAhead] : idx_safe));
                    sc[a] = p.s[vn ? group(q + kCtAhead) : 0u];
                }
                __builtin_amdgcn_sched_barrier(0);        // the refill is issued here, not sunk towards its use
            }
        }
    }
    ct_flush<OP_BWD>(p, ct, g, acc, gcur, mine);
}

// single-tensor launches (lq_fq_forward_oihw / lq_fq_scale_grad_oihw)
__global__ __launch_bounds__(kBlock) void k_conv_tile_fwd(Params p, ConvTile ct) {
    __shared__ float lds[kCtLdsWordsFwd];
    conv_tile_fwd(p, ct, blockIdx.x, lds);
}
__global__ __launch_bounds__(kBlock) void k_conv_tile_bwd(Params p, ConvTile ct) {
    __shared__ float lds[kCtLdsWordsBwd];
    conv_tile_bwd(p, ct, blockIdx.x, lds);
}

}  // namespace lq

#endif
