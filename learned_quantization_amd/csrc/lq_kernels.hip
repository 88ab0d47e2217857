// lq_kernels.hip -- hand-written gfx950 (MI355X / CDNA4) kernels for the learned-quantization
// hot path, and the C ABI of include/lq_hip.h on top of them.  One translation unit:
//
//   lq_common.hpp       constants, kernel parameter block, accumulator types
//   lq_math.hpp         exact fp32 arithmetic: uniform-divisor division, in-window ratio division, |tanh|, vote
//   lq_ops.hpp          per-operation traits (K1 fwd, K2 bwd, K4 fused, K5 penalties, integer view)
//   lq_reduce.hpp       wave/block reductions (DPP for the standard accumulator, shuffles for custom merges)
//   lq_traverse.hpp     traversal modes (row stream / row small / column) + finalize kernels
//   lq_stream2.hpp      streaming-size forms of the column and tiny-row modes (round 2): flat K1, pipelined column tile, ...
//   lq_aux_kernels.hpp  scale-sized kernels (K5c, K6), integer statistics, device self-test
//   lq_batch.hpp        device side of the multi-tensor batch
//   this file           host side: traversal plan, launchers, the extern "C" entry points
//
// The path is elementwise + per-group reductions: HBM-bound, no MFMA.  Design rules
// (cdna_hip_programming.md G2/G11/G12/G13, MI355X_MICROARCH.md HBM), each settled by measurement on the
// device (tools/membench.hip, tools/bench_shapes.py):
//   * one 16-byte coalesced access per lane and stream (float4), nontemporal for tensors >= 64 MiB; blocks of
//     512 threads for streaming-size tensors; loads issued before the per-group context is computed;
//   * the per-group scale is block-uniform in the streaming kernels (one scalar load, SGPR broadcast) -- the
//     group index never costs per-element integer division (3-D grid / loop-invariant lane->column maps);
//   * reductions: DPP inside the 64-lane wave -> LDS across the block's waves -> one partial per block in a
//     caller-provided workspace -> a finalize kernel.  No float atomics: every sum is taken in a fixed
//     order, results are run-to-run bit-stable;
//   * max|q| is reduced on the uint32 bit pattern of |q| (order-independent, exact);
//   * the integers must match the reference bit for bit: IEEE-754 correctly rounded fp32 division followed
//     by floor (never a reciprocal multiply; the fast forms are proven/validated exact), -ffp-contract=off,
//     no fast-math.
//
// Reference semantics restated here (file:line relative to /root/reference):
//   forward           MNIST/nested_quantization_layer/custom_components/custom_layers.py:55-60
//   NQ backward       custom_layers.py:62-118
//   penalties         CIFAR-10/custom_loss_terms/custom_components/custom_loss_functions.py:75-116,161-195,240-275
//   constraint        custom_layers.py:35-46
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>
#include <stdarg.h>
#include <stdlib.h>
#include <stdint.h>
#include <stdio.h>
#include <string.h>

#include <new>
#include <algorithm>
#include <vector>

#include "lq_hip.h"

#include "lq_batch.hpp"
#include "lq_stream2.hpp"

namespace lq {

// ------------------------------------------------------------------------------------------
//  Host side: plan, launchers, C ABI.
// ------------------------------------------------------------------------------------------
enum Mode { MODE_ROW_BIG = 0, MODE_ROW_SMALL = 1, MODE_COL = 2 };

// Rows per block of the pipelined column tile, by operation (MI355X sweep, profiles/r02/column_tile_rows_per_block.txt):
// the read-only scale-gradient traversal wants long blocks (128 rows: 5.7-5.9 TB/s; 32 rows: 4.5), the kernels that
// also store want short ones (48 rows: K4 5.6-5.8 TB/s; 128 rows: 5.3-5.5)
constexpr int64_t kColRbFused = 48;
constexpr int64_t kColRbBwd = 128;

struct Plan {
    int mode;
    int64_t R, L;       // row modes
    int bs;             // threads per block in the streaming traversal (unit = bs*4 elements)
    int CH;
    int64_t nc;
    int lpr_log2;
    int64_t C, rps, ysplit;   // column mode
    int per4;           // column mode, C <= 64: geometry admits the float4 grid-stride variant (ysplit % C == 0)
    int64_t np;         // number of partial triples
    int64_t np_ws;      // row-big at streaming size: partial count of the smallest chunk the launch may choose (workspace bound)
    // finalize geometry (per group)
    int64_t gstride, n1, stride1, n2;
};

static int64_t ceil_div(int64_t a, int64_t b) { return (a + b - 1) / b; }

// Development knobs.  The shipped library reads NO environment variable: the traversal plan, the partial layout and with
// them the workspace size are pure functions of the descriptor and the pointers' alignment.  `make dev` (-DLQ_DEV_KNOBS,
// tools/ only) turns LQ_KNOB into a getenv lookup and compiles the alternative launches the tuning sweeps of profiles/r02
// were made with; every use is marked.
#ifdef LQ_DEV_KNOBS
static int knob_env(const char* name, int dflt) {
    const char* e = getenv(name);
    return e ? atoi(e) : dflt;
}
#define LQ_KNOB(var, name, dflt) static const int var = knob_env(name, dflt)
#else
#define LQ_KNOB(var, name, dflt) constexpr int var = (dflt)
#endif

static bool flat_cols_ok(int64_t C) { return C == 8 || C == 16 || C == 32 || C == 64; }
static int64_t colx_max_inner() {        // rows shorter than this may run in column mode (see make_plan)
    LQ_KNOB(v, "LQ_TUNE_COLX_INNER", 200);
    return v;
}
static int64_t periodic_target() {      // block count of the periodic column form
    LQ_KNOB(v, "LQ_TUNE_PER_NB", 2048);
    return v > 0 ? v : 2048;
}
static int64_t periodic_blocks(int64_t C) { return C * ((periodic_target() + C / 2) / C); }      // ~2048 blocks, a multiple of C

// Chunks of CH elements per row of length L.  A tail of at most CH/8 elements is folded into the previous chunk (the
// traversal's last chunk takes whatever remains) instead of getting a block of its own.
static int64_t row_chunks(int64_t L, int64_t CH) {
    int64_t nc = ceil_div(L, CH);
    const int64_t r = L % CH;
    if (nc > 1 && r != 0 && r * 8 <= CH) nc -= 1;
    return nc;
}

static Plan make_plan(int64_t outer, int64_t G, int64_t inner, int force_bs = 0) {
    Plan pl;
    memset(&pl, 0, sizeof(pl));
    const int64_t N = outer * G * inner;
    int64_t R, L, f_outer;
    bool col = false;
    if (G == 1) {
        R = 1;
        L = N;
        f_outer = 1;
    } else {
        R = outer * G;
        L = inner;
        f_outer = outer;
        // Short rows that come in many layers (outer >= 32: per-channel scales of NCHW activations with small planes, e.g.
        // (256, 2048, 7 x 7)) are instruction-bound as rows -- per-row context, team reduction and emit every 100-400 bytes --
        // but as the matrix [outer][G * inner] a lane keeps its four columns, contexts and accumulators across all layers:
        // column mode for them too, at streaming size and when the matrix rows are whole 128-byte lines.
        LQ_KNOB(colx, "LQ_TUNE_COLX", 1);      // 0 = rows
        const bool short_rows_many_layers = colx && inner >= 16 && inner < colx_max_inner() && outer >= 32 && (G * inner) % 32 == 0 &&
                                            G * inner > 64 && (double)N >= (double)kPeriodic4Min && !(inner % 4 == 0 && (inner & (inner - 1)) == 0);
        if ((inner < 16 && outer > 1) || short_rows_many_layers) {
            // column mode unless thread-per-row yields fewer partials
            const int64_t C = G * inner;
            // rows per block: ~16 K elements per block for narrow matrices, 128 rows for wide ones
            int64_t RB;
            int64_t nby = 0;
            int per4 = 0;
            if (C <= 64 && (double)outer * (double)C >= (double)kPeriodic4Min) {
                per4 = 1;
                // streaming size: a block count that is a multiple of C (what the float4 grid-stride variant needs);
                // the scalar periodic variant runs on the same geometry (trailing blocks may own no rows)
                nby = periodic_blocks(C);
                const int64_t k = 64 / C;
                RB = ceil_div(ceil_div(outer, nby), 4 * k) * 4 * k;
            } else if (C <= 64) {
                // ~16 K elements per block, but at least ~1024 blocks' worth of parallelism for small matrices: a block whose
                // waves walk 1656 rows of 10 columns needs 17 dependent load rounds (12 us for 64 K elements)
                const int64_t k = 64 / C;
                RB = ceil_div(ceil_div(16384, C), 4 * k) * 4 * k;
                const int64_t rb_par = ceil_div(ceil_div(outer, 1024), 4 * k) * 4 * k;
                if (rb_par < RB) RB = rb_par;
            } else {
                // 128 rows per block for streaming sizes; small matrices get shorter row blocks (about 512 blocks in all) so
                // that a wave walks its rows in one or two dependent load rounds instead of eight (784 x 128 Dense, column-wise:
                // scale-gradient traversal 12.5 -> 5.9 us, forward 9.4 -> 4.3 us)
                RB = ceil_div(outer * ceil_div(C, 256), 512);
                RB = ceil_div(RB, 16) * 16;
                if (RB > 128) RB = 128;
                // streaming sizes with float4 columns: the round-2 pipelined tile (lq_stream2.hpp) picks its rows per block per
                // operation at launch (kColRbFused / kColRbBwd); the plan carries the smaller one, i.e. the larger partial
                // count, so the workspace bound holds for both
                // (C % 4 != 0: the same tile with dword-aligned float4 access, k_col_pipe<..., UA = 1>)
                if ((double)outer * (double)C >= (double)kPeriodic4Min) RB = kColRbFused;
                // short rows in many layers (see above): one partial per column and 128 layers keeps the partial count near the
                // row count (a partial per column and 48 layers would be 5-10 % of the traffic for rows of 100+ elements)
                // -- only when 48 layers per block would not fit the partial budget below (K4 prefers 48: 7 x 7 planes 5.4 against 5.2)
                if (short_rows_many_layers && ceil_div(outer, kColRbFused) * C > 2 * R) RB = kColRbBwd;
            }
            if (RB > outer) RB = outer;
            if (!nby) nby = ceil_div(outer, RB);
            if (nby * C <= R || (short_rows_many_layers && nby * C <= 2 * R)) {
                col = true;
                pl.mode = MODE_COL;
                pl.C = C;
                pl.rps = RB;
                pl.ysplit = nby;
                pl.per4 = per4;
                pl.np = nby * C;
                // the one-shot flat column kernel (lq_stream2.hpp k_flat_cols: C <= 4 or C = 8, 16, 32, 64 at streaming sizes)
                // writes one partial per (block of 1024 float4, column): size the workspace for it
                if (per4 && flat_cols_ok(C)) {
                    const int64_t fb = ceil_div(ceil_div(outer * C, 4) + 1, (int64_t)kFlatColsBlock * 2);
                    if (fb * C > pl.np) pl.np = fb * C;
                }
                // 64 < C <= 256 at streaming size may run the periodic form (a column tile narrower than 256 columns leaves
                // lanes idle): one partial per (block, column) for periodic_blocks(C) blocks
                if (C > 64 && C <= 256 && (double)outer * (double)C >= (double)kPeriodic4Min && periodic_blocks(C) * C > pl.np)
                    pl.np = periodic_blocks(C) * C;
                pl.gstride = inner;
                pl.n1 = nby;
                pl.stride1 = C;
                pl.n2 = inner;
            }
        }
    }
    if (!col) {
        pl.R = R;
        pl.L = L;
        if (L >= 1024) {
            pl.mode = MODE_ROW_BIG;
            // 512 threads: measured best on MI355X (BENCH bwd 44.9 us vs 53.6 us at 1024 -- fewer waves held at the
            // closing barrier -- and vs 48.4 us at 256; the forward is insensitive)
            // Below 4 M elements a tensor is latency-bound and keeps 256-thread units, the geometry the
            // multi-tensor batch kernels use -- so batched and single-tensor results are bit-identical there.
            pl.bs = (L >= 2048 && (double)R * (double)L >= 4194304.0) ? 512 : 256;
            if (force_bs) pl.bs = force_bs;
#ifdef LQ_DEV_KNOBS
            else {                                             // force the streaming block size
                LQ_KNOB(v, "LQ_TUNE_BS", 0);
                if ((v == 256 || v == 512 || v == 1024) && L >= v * 4) pl.bs = v;
            }
#endif
            pl.CH = pl.bs * 4;
            pl.nc = row_chunks(L, pl.CH);
            if (pl.bs == 512) pl.np_ws = R * row_chunks(L, 1024);      // launch_traverse may cut such rows into 1024-element chunks
        } else {
            pl.mode = MODE_ROW_SMALL;
            pl.CH = (int)L;
            pl.nc = 1;
            int lpr = 1, lg = 0;   // lanes per row = pow2floor(max(L/2, 1)) clipped to one wave
            const int64_t half = L / 2 > 1 ? L / 2 : 1;
            while (lpr * 2 <= half && lpr < 64) {
                lpr *= 2;
                ++lg;
            }
            pl.lpr_log2 = lg;
        }
        pl.np = R * pl.nc;
        if (pl.mode == MODE_ROW_SMALL) {   // one partial per row, stored group-major: g * outer + o
            pl.gstride = f_outer;
            pl.n1 = f_outer;
            pl.stride1 = 1;
            pl.n2 = 1;
        } else {
            pl.gstride = pl.nc;
            pl.n1 = f_outer;
            pl.stride1 = G * pl.nc;
            pl.n2 = pl.nc;
        }
    }
    return pl;
}

static thread_local char g_err[512] = "";
static thread_local hipEvent_t g_prof_start = nullptr, g_prof_stop = nullptr;      // lq_profile_events

static int fail(int code, const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
    return code;
}

static int check_hip(const char* what) {
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return fail(LQ_EHIP, "%s: %s", what, hipGetErrorString(e));
    return LQ_OK;
}

static bool aligned(const void* p, size_t a) { return (reinterpret_cast<uintptr_t>(p) % a) == 0; }

static int check_desc(int64_t outer, int64_t G, int64_t inner) {
    if (outer <= 0 || G <= 0 || inner <= 0) return fail(LQ_EINVAL, "descriptor extents must be positive (outer=%lld G=%lld inner=%lld)", (long long)outer, (long long)G, (long long)inner);
    const double n = (double)outer * (double)G * (double)inner;
    if (n > 9.0e15) return fail(LQ_EINVAL, "tensor too large");
    return LQ_OK;
}

static size_t ws_bytes_for(const Plan& pl) {
    size_t np = (size_t)(pl.np_ws > pl.np ? pl.np_ws : pl.np);
    np = (np + 63) / 64 * 64;
    return np * 16 + 256;
}

static int bind_ws(Params& p, const Plan& pl, void* ws, size_t ws_bytes) {
    if (!ws) return fail(LQ_EWORKSPACE, "workspace is NULL (need %zu bytes)", ws_bytes_for(pl));
    if (!aligned(ws, 16)) return fail(LQ_EALIGN, "workspace must be 16-byte aligned");
    if (ws_bytes < ws_bytes_for(pl)) return fail(LQ_EWORKSPACE, "workspace too small: %zu < %zu bytes", ws_bytes, ws_bytes_for(pl));
    size_t np = ((size_t)(pl.np_ws > pl.np ? pl.np_ws : pl.np) + 63) / 64 * 64;
    p.pa = reinterpret_cast<uint32_t*>(ws);
    p.pb = p.pa + np;
    p.pc = reinterpret_cast<double*>(p.pb + np);
    return LQ_OK;
}

// row-small: lanes per row.  Scalar form: pow2floor(max(L/2,1)) (plan); float4 form: pow2ceil(L/4), both <= 64.
static int row_small_lpr_log2_vec(int64_t L) {
    const int64_t l4 = L / 4;
    int lg = 0;
    while ((1 << lg) < l4 && lg < 6) ++lg;
    return lg;
}
static bool row_small_vec(const Plan& pl, const void* P, const void* dy, const void* out) {
    return pl.L % 4 == 0 && pl.L >= 8 && aligned(P, 16) && (!dy || aligned(dy, 16)) && (!out || aligned(out, 16));
}

// column-mode kernel variant and blocks along the columns: 0 = periodic (C <= 64), 4 = float4 tile, 1 = scalar tile
static void col_variant(const Plan& pl, const void* P, const void* dy, const void* out, int& variant, int64_t& nbx,
                        bool allow_periodic4 = true) {
    const bool al16 = aligned(P, 16) && (!dy || aligned(dy, 16)) && (!out || aligned(out, 16));
    const double bytes = (double)pl.C * (double)pl.ysplit * (double)pl.rps * 4.0;
    if (pl.C <= 64) {
        variant = (allow_periodic4 && pl.per4 && al16) ? (bytes >= (double)kNtBytes ? 7 : 6) : 0;
        nbx = 1;
    } else if (pl.C % 4 == 0 && aligned(P, 16) && (!dy || aligned(dy, 16)) && (!out || aligned(out, 16))) {
        variant = ((double)pl.C * (double)pl.ysplit * (double)pl.rps * 4.0 >= (double)kNtBytes) ? 5 : 4;
        nbx = ceil_div(pl.C, 256);
    } else {
        variant = 1;
        nbx = ceil_div(pl.C, 64);
    }
}

// development knobs of the streaming forms: LQ_TUNE_S2 = bit mask of forms to DISABLE (1 flat K1, 2 pipelined column tile,
// 4 pipelined periodic columns, 8 tiny-row passes, ...); LQ_TUNE_PIPE = U*10 + NW of the column tile; LQ_TUNE_TINY_U = passes

#ifdef LQ_DEV_KNOBS
constexpr bool kDevKnobs = true;       // tools/ builds: the LQ_TUNE_* switches can route a descriptor to any form
#else
constexpr bool kDevKnobs = false;
#endif

// Streaming-size (>= 4 M elements) forms of lq_stream2.hpp.  Returns 1 when it launched the traversal, 0 when the
// round-1 traversal should run, < 0 on error.
template <int OP>
static int launch_stream2_impl(Plan& pl, const Params& p, hipStream_t st);

template <int OP>
static int launch_stream2(Plan& pl, const Params& p, hipStream_t st) {
    if constexpr (OP == OP_FWD || OP == OP_BWD || OP == OP_FUSED) return launch_stream2_impl<OP>(pl, p, st);
    else return 0;
}

template <int OP>
static int launch_stream2_impl(Plan& pl, const Params& p, hipStream_t st) {
    using O = OpT<OP>;
    const double numel = (double)p.outer * (double)p.G * (double)p.inner;
    if (numel < (double)kPeriodic4Min) return 0;
    if (p.out_perm || p.dy_perm) return 0;
    FlatIdx fx;              // 32-bit forms only (numel < 2^32); the wide forms divide in 64 bits
    fx.inner = make_fastdiv((uint32_t)(p.inner < 4294967296ll ? p.inner : 1));
    fx.G = make_fastdiv((uint32_t)(p.G < 4294967296ll ? p.G : 1));
    (void)fx;
    if (pl.mode == MODE_ROW_BIG) {
        // long rows off the 16-byte grid: K1 as a line-aligned flat stream (lq_stream2.hpp k_flat_fwd, group mode 6 / 7)
        if constexpr (OP == OP_FWD) {
            LQ_KNOB(off_rb, "LQ_TUNE_S2", 0);
            const int64_t nn = p.outer * p.G * p.inner;
            // K1 of long rows as the flat one-shot stream too: one group per float4 when L % 4 == 0 (group mode 0 / 2) -- rows that
            // are not whole 128-byte lines (4100: 5.4 -> 6.3 TB/s), rows that fill their chunks poorly (1600 = 1024 + 576: 5.65 ->
            // 6.3) and, by 2 %, the well-filled aligned rows as well (BENCH tensor, K1 alone on four buffer sets, three interleaved
            // runs: 48.6-48.9 us against 49.8-49.9 for the row stream; profiles/r02/bench_k1_row_stream_vs_flat.txt; development
            // knob 32768 restores the row stream for them).  L % 4 != 0: a float4 may straddle a row end (6 / 7), see below.
            const bool poor_fill = (double)pl.L / (double)(pl.nc * pl.CH) < 0.95;
            if (!(off_rb & 256) && (pl.L % 32 != 0 || poor_fill || !(off_rb & 32768)) && aligned(p.P, 16) && aligned(p.out, 16)) {
                const int64_t nv = nn >> 2;
                const int rem = (int)(nn & 3);
                const int64_t blocks = ceil_div(nv + (rem ? 1 : 0), 512);
                if (blocks <= 2147483647ll) {
                    const bool ntb = numel * 4.0 >= (double)kNtBytes;
                    const bool wide = nn >= 4294967296ll;
                    // hipExtLaunchKernelGGL: with lq_profile_events() set, the events take the kernel's own begin / end timestamps
#define LQ_FLATR(NT_, GM_) hipExtLaunchKernelGGL((k_flat_fwd<OP, 512, NT_, GM_>), dim3((unsigned)blocks), dim3(512), 0, st, g_prof_start, g_prof_stop, 0, p, fx, nv, rem)
                    if (pl.L % 4 == 0) {                    // (2^32 elements are 16 GiB: `wide` implies `ntb`)
                        if (ntb) { if (wide) LQ_FLATR(1, 2); else LQ_FLATR(1, 0); }
                        else LQ_FLATR(0, 0);
                    } else if (!(off_rb & 512) && (double)pl.L / (double)(pl.nc * pl.CH) >= 0.8) {
                        // long rows with L % 4 != 0 keep the row stream (TAIL instantiation): K1 5.8-6.0 TB/s on rows of 1025,
                        // 2047, 4099, 50177 against 5.4-5.9 for the straddling flat form (development knob 512 selects the latter)
                        // -- unless their chunks are poorly filled (rows of 1225 = 1024 + 201 elements: 4.3 TB/s)
                        return 0;
                    } else {
                        if (ntb) { if (wide) LQ_FLATR(1, 7); else LQ_FLATR(1, 6); }
                        else LQ_FLATR(0, 6);
                    }
#undef LQ_FLATR
                    return check_hip("flat forward launch") ? -1 : 1;
                }
            }
        } else {
            // Rows of 1153..1533 elements are two 1024-chunks, the second one 13-50 % full (35 x 35 planes: 1225 = 1024 + 201) --
            // too few bytes in flight per resident thread: K2 / K4 4.0 / 3.9 TB/s.  One wave per row through the row-window
            // kernel instead (5 or 6 float4 per lane: 80-100 % of the lanes carry data); partials as in the row-small mode.
            LQ_KNOB(off_rb, "LQ_TUNE_S2", 0);
            const int nwin = (int)(pl.L % 4 ? (pl.L + 6) / 4 : pl.L / 4);
            const bool al = aligned(p.P, 16) && aligned(p.dy, 16) && (!O::kStore || aligned(p.out, 16));
            // (rows of 1534..2047 elements: two chunks that fill >= 75 % -- the row stream is as good or better (1800: K2 / K4 6.4 / 5.8
            // against 6.0 / 5.5 with 8 float4 per lane) unless the rows are off the 16-byte grid (1535, 1537: 5.0 / 4.7 -> 6.1 / 5.6))
            LQ_KNOB(win_max, "LQ_TUNE_WIN_MAX", 384);      // development knob: widest window (float4) for rows ON the grid
            const bool wide_ok = pl.L % 4 != 0 && (double)pl.L / 2048.0 < 0.95;
            if (!(off_rb & 128) && pl.bs == 256 && pl.nc == 2 && (nwin <= win_max || wide_ok) && nwin <= 512 && al && !p.direct && pl.R < 4294967296ll) {
                const int64_t blocks = ceil_div(pl.R, (int64_t)kWavesPerBlock);
                if (blocks <= 2147483647ll) {
                    const int64_t n = p.outer * p.G * p.inner;
                    const bool ntb = numel * 4.0 >= (double)kNtBytes;
                    const FastDiv fG = make_fastdiv((uint32_t)p.G);
                    const int64_t f_outer = pl.R / p.G;
                    pl.nc = 1;                 // the finalize that follows must walk the partial layout this launch produces:
                    pl.np = pl.R;              // one partial per row, group-major (g * outer + o), as in the row-small mode
                    pl.gstride = f_outer;
                    pl.n1 = f_outer;
                    pl.stride1 = 1;
                    pl.n2 = 1;
#define LQ_WINB(NT_, V_) hipLaunchKernelGGL((k_row_win<OP, NT_, 6, V_, 1>), dim3((unsigned)blocks), dim3(kBlock), 0, st, p, fG, pl.R, (int)pl.L, n)
                    if (nwin <= 320) { if (ntb) LQ_WINB(1, 5); else LQ_WINB(0, 5); }
                    else if (nwin <= 384) { if (ntb) LQ_WINB(1, 6); else LQ_WINB(0, 6); }
                    else if (nwin <= 448) { if (ntb) LQ_WINB(1, 7); else LQ_WINB(0, 7); }
                    else { if (ntb) LQ_WINB(1, 8); else LQ_WINB(0, 8); }
#undef LQ_WINB
                    return check_hip("row-window launch") ? -1 : 1;
                }
            }
        }
        return 0;
    }
    const bool al16 = aligned(p.P, 16) && (!O::kDy || aligned(p.dy, 16)) && (!O::kStore || aligned(p.out, 16));
    if (!al16) return 0;
    LQ_KNOB(off, "LQ_TUNE_S2", 0);
    const bool nt = numel * 4.0 >= (double)kNtBytes;
    const int64_t n = p.outer * p.G * p.inner;
    // ---- K1 as a flat stream: whenever a float4 has one group (inner % 4 == 0) -- tiny rows, rows of 8..1020 elements,
    // column matrices with inner = 4, 8, 12
    if constexpr (OP == OP_FWD) {
        const bool one_group = p.inner % 4 == 0;                                             // a float4 never straddles groups
        const bool scale4 = p.inner == 1 && p.G % 4 == 0 && aligned(p.s, 16) && !(off & 32);   // its 4 scales are one float4
        const bool scale4u = p.inner == 1 && p.G % 4 != 0 && p.G > 64 && !(off & 8192);       // the same, dword-aligned, may wrap
        const bool cols_pow2 = pl.mode == MODE_COL && pl.per4 && flat_cols_ok(pl.C) && !(off & 64);   // k_flat_cols below: 1-2 % faster
        // rows of 4..1023 elements off the 16-byte grid (row-small mode, L % 4 != 0): a float4 straddles at most one row end
        const bool straddle = pl.mode == MODE_ROW_SMALL && pl.L >= 4 && pl.L % 4 != 0 && !(off & 256);
        if (straddle) {
            const int64_t nv = n >> 2;
            const int rem = (int)(n & 3);
            const int64_t blocks = ceil_div(nv + (rem ? 1 : 0), 512);
            if (blocks <= 2147483647ll) {
                const bool wide = n >= 4294967296ll;
#define LQ_FLATS(NT_, GM_) hipLaunchKernelGGL((k_flat_fwd<OP, 512, NT_, GM_>), dim3((unsigned)blocks), dim3(512), 0, st, p, fx, nv, rem)
                if (nt) { if (wide) LQ_FLATS(1, 7); else LQ_FLATS(1, 6); }
                else LQ_FLATS(0, 6);
#undef LQ_FLATS
                return check_hip("flat forward launch") ? -1 : 1;
            }
        }
        if (scale4u && !(off & 1)) {
            const int64_t nv = n >> 2;
            const int rem = (int)(n & 3);
            const int64_t blocks = ceil_div(nv + (rem ? 1 : 0), 512);
            if (blocks <= 2147483647ll) {
                const bool wide = n >= 4294967296ll;
#define LQ_FLATU(NT_, GM_) hipLaunchKernelGGL((k_flat_fwd<OP, 512, NT_, GM_>), dim3((unsigned)blocks), dim3(512), 0, st, p, fx, nv, rem)
                if (nt) { if (wide) LQ_FLATU(1, 9); else LQ_FLATU(1, 8); }
                else LQ_FLATU(0, 8);
#undef LQ_FLATU
                return check_hip("flat forward launch") ? -1 : 1;
            }
        }
        // every other column-mode forward (several groups inside a float4: C = 3, 5, 10, 30; inner = 2, 3, 5, 15): gathered scales
        LQ_KNOB(gather_on, "LQ_TUNE_GATHER", 1);
        if (gather_on && !(off & 1) && pl.mode == MODE_COL && !one_group && !scale4 && !scale4u && !cols_pow2 && n < 4294967296ll) {
            const int64_t nv = n >> 2;
            const int rem = (int)(n & 3);
            const int64_t blocks = ceil_div(nv + (rem ? 1 : 0), 512);
            if (blocks <= 2147483647ll) {
#define LQ_FLATG(NT_, GM_) hipLaunchKernelGGL((k_flat_fwd<OP, 512, NT_, GM_>), dim3((unsigned)blocks), dim3(512), 0, st, p, fx, nv, rem)
                if (p.inner == 1) { if (nt) LQ_FLATG(1, 10); else LQ_FLATG(0, 10); }
                else { if (nt) LQ_FLATG(1, 11); else LQ_FLATG(0, 11); }
#undef LQ_FLATG
                return check_hip("flat forward launch") ? -1 : 1;
            }
        }
        if (!(off & 1) && (one_group || scale4) && !cols_pow2) {
            const int64_t nv = n >> 2;
            const int64_t blocks = ceil_div(nv, 512);
            if (blocks <= 2147483647ll) {
                const bool wide = n >= 4294967296ll;
#define LQ_FLAT(NT_, GM_) hipLaunchKernelGGL((k_flat_fwd<OP, 512, NT_, GM_>), dim3((unsigned)blocks), dim3(512), 0, st, p, fx, nv, 0)
                if (one_group) {
                    if (nt) {
                        if (wide) LQ_FLAT(1, 2); else LQ_FLAT(1, 0);
                    } else {
                        LQ_FLAT(0, 0);
                    }
                } else {
                    if (nt) {
                        if (wide) LQ_FLAT(1, 5); else LQ_FLAT(1, 4);
                    } else {
                        LQ_FLAT(0, 4);
                    }
                }
#undef LQ_FLAT
                return check_hip("flat forward launch") ? -1 : 1;
            }
        }
    }
    if (pl.mode == MODE_COL) {
        // K1 reaches this point only with 2^32 or more elements (every smaller column-mode forward is one of the flat forms
        // above): the round-1 column kernel serves it -- no pipelined column instantiation exists for the forward
        if constexpr (OP == OP_FWD) {
            return 0;
        } else {
        constexpr int kUp = 2;                                 // float4 per stream in flight (two streams)
        // K2 of C = 8, 16, 32, 64 as a one-shot stream with FOUR float4 per thread and stream: with two, nothing hid the shuffle
        // tree and the barrier at the end of so short a wave (4.3-5.4 TB/s); with four the epilogue is paid once per 8 KB of each
        // stream: 6.0-6.2 TB/s against 5.5-5.6 for the periodic form.  (K4 with four: 5.6-5.9 against 5.7-6.1 with two.)
        if constexpr (OP == OP_BWD) {
            LQ_KNOB(fc_k2, "LQ_TUNE_FC_K2", 4);      // development knob: 0 = the periodic form
            if (fc_k2 == 4 && pl.C <= 64 && pl.per4 && flat_cols_ok(pl.C) && nt && !(off & 64)) {
                const int64_t nv = n >> 2;
                const int64_t fb = ceil_div(nv, (int64_t)kFlatColsBlock * 4);
                if (fb <= 2147483647ll && fb * pl.C <= pl.np) {
                    pl.ysplit = fb;           // the finalize that follows must walk the partial layout this launch produces
                    pl.np = fb * pl.C;
                    pl.n1 = fb;
                    hipLaunchKernelGGL((k_flat_cols<OP, 1, 4>), dim3((unsigned)fb), dim3(kFlatColsBlock), 0, st, p, (int)pl.C, nv);
                    return check_hip("flat column launch") ? -1 : 1;
                }
            }
        }
        if constexpr (OP != OP_BWD) if (pl.C <= 64 && pl.per4 && flat_cols_ok(pl.C) && !(off & 64)) {
            // C = 8, 16, 32, 64 as a flat one-shot stream: K1 and K4 (the read-only K2 is faster in the periodic form)
            const int64_t nv = n >> 2;                        // numel = outer * C is a multiple of 8
            constexpr int kU = 2;
            const int64_t fb = ceil_div(nv, (int64_t)kFlatColsBlock * kU);
            if (fb <= 2147483647ll && (OP == OP_FWD || fb * pl.C <= pl.np)) {
                if (OP != OP_FWD) {
                    pl.ysplit = fb;           // the finalize that follows must walk the partial layout this launch produces
                    pl.np = fb * pl.C;
                    pl.n1 = fb;
                }
                if (nt) hipLaunchKernelGGL((k_flat_cols<OP, 1, kU>), dim3((unsigned)fb), dim3(kFlatColsBlock), 0, st, p, (int)pl.C, nv);
                else hipLaunchKernelGGL((k_flat_cols<OP, 0, kU>), dim3((unsigned)fb), dim3(kFlatColsBlock), 0, st, p, (int)pl.C, nv);
                return check_hip("flat column launch") ? -1 : 1;
            }
        }
        if (pl.C <= 64) {
            if ((off & 4) || !pl.per4) return 0;
#ifdef LQ_DEV_KNOBS
            LQ_KNOB(per, "LQ_TUNE_PERIODIC", 0);      // U*10 + (block size / 256)
            if (nt && (per == 12 || per == 22)) {
                // 512-thread blocks, half as many: the finalize that follows must walk the partial layout this launch produces
                const int64_t nb2 = ((pl.ysplit / 2) / pl.C) * pl.C;
                if (nb2 >= pl.C) {
                    pl.ysplit = nb2;
                    pl.np = nb2 * pl.C;
                    pl.n1 = nb2;
                    if (per == 12) hipLaunchKernelGGL((k_col_periodic_pipe<OP, 1, 1, 512>), dim3((unsigned)nb2), dim3(512), 0, st, p, (int)pl.C, nb2);
                    else hipLaunchKernelGGL((k_col_periodic_pipe<OP, 1, 2, 512>), dim3((unsigned)nb2), dim3(512), 0, st, p, (int)pl.C, nb2);
                    return check_hip("periodic column launch") ? -1 : 1;
                }
            }
            if (nt && (per == 11 || per == 21 || per == 41)) {
                if (per == 11) hipLaunchKernelGGL((k_col_periodic_pipe<OP, 1, 1, 256>), dim3((unsigned)pl.ysplit), dim3(256), 0, st, p, (int)pl.C, pl.ysplit);
                else if (per == 21) hipLaunchKernelGGL((k_col_periodic_pipe<OP, 1, 2, 256>), dim3((unsigned)pl.ysplit), dim3(256), 0, st, p, (int)pl.C, pl.ysplit);
                else hipLaunchKernelGGL((k_col_periodic_pipe<OP, 1, 4, 256>), dim3((unsigned)pl.ysplit), dim3(256), 0, st, p, (int)pl.C, pl.ysplit);
                return check_hip("periodic column launch") ? -1 : 1;
            }
#endif
            // K4 with C >= 16: one float4 per stream in flight (98 VGPRs, 5 waves per SIMD) measured 5.67-5.75 TB/s at C = 64 against
            // 5.30-5.49 with two (128 VGPRs); at C = 3 the other way round (5.57 against 5.24)
            if constexpr (OP == OP_FUSED) {
                if (nt && pl.C >= 16) {
                    hipLaunchKernelGGL((k_col_periodic_pipe<OP, 1, 1>), dim3((unsigned)pl.ysplit), dim3(kBlock), 0, st, p, (int)pl.C, pl.ysplit);
                    return check_hip("periodic column launch") ? -1 : 1;
                }
            }
            if (nt) hipLaunchKernelGGL((k_col_periodic_pipe<OP, 1, kUp>), dim3((unsigned)pl.ysplit), dim3(kBlock), 0, st, p, (int)pl.C, pl.ysplit);
            else hipLaunchKernelGGL((k_col_periodic_pipe<OP, 0, kUp>), dim3((unsigned)pl.ysplit), dim3(kBlock), 0, st, p, (int)pl.C, pl.ysplit);
            return check_hip("periodic column launch") ? -1 : 1;
        }
        LQ_KNOB(per_cmax, "LQ_TUNE_PER_CMAX", 256);      // development knob: 0 = always the tile
        if (pl.C <= per_cmax && pl.C <= 256 && (pl.C < 192 || pl.C % 32 != 0) && periodic_blocks(pl.C) * pl.C <= pl.np) {
            // 64 < C <= 256: a tile narrower than 256 columns leaves lanes idle and, when C % 32 != 0, starts every row in the
            // middle of a 128-byte line; the periodic form is a flat line-aligned stream for any C.  Measured (K2 / K4 TB/s,
            // tile -> periodic): C = 68: 3.2 / 2.9 -> 5.6 / 5.7; 100: 4.3 / 4.0 -> 5.7 / 5.6; 130: 5.0 / 4.3 -> 5.4 / 5.1;
            // 99 (inner 3): 3.9 / 3.6 -> 5.5 / 5.0; 200, 250: +-3 %; 192 (3 full lines per row): 5.5 / 5.6 -> 5.1 / 5.1, keeps the tile
            const int64_t nb = periodic_blocks(pl.C);
            pl.ysplit = nb;               // the finalize that follows must walk the partial layout this launch produces
            pl.np = nb * pl.C;
            pl.n1 = nb;
            if constexpr (OP == OP_FUSED) {
                if (nt) hipLaunchKernelGGL((k_col_periodic_pipe<OP, 1, 1>), dim3((unsigned)nb), dim3(kBlock), 0, st, p, (int)pl.C, nb);
                else hipLaunchKernelGGL((k_col_periodic_pipe<OP, 0, kUp>), dim3((unsigned)nb), dim3(kBlock), 0, st, p, (int)pl.C, nb);
            } else {
                if (nt) hipLaunchKernelGGL((k_col_periodic_pipe<OP, 1, kUp>), dim3((unsigned)nb), dim3(kBlock), 0, st, p, (int)pl.C, nb);
                else hipLaunchKernelGGL((k_col_periodic_pipe<OP, 0, kUp>), dim3((unsigned)nb), dim3(kBlock), 0, st, p, (int)pl.C, nb);
            }
            return check_hip("periodic column launch") ? -1 : 1;
        }
        // 256 < C <= 512 whose second column block is mostly empty (C = 320: 16 of 64 lanes in half the blocks, K2 / K4 4.7 / 4.5 TB/s)
        // or whose rows start inside a 128-byte line: the periodic form with 512-thread blocks (its LDS combine needs C <= block
        // size): C = 300: 4.2 / 4.0 -> 5.5 / 5.4; 320: -> 5.6 / 5.5; 450: 4.8 / 4.5 -> 5.4 / 5.4; 500: 5.3 / 4.9 -> 5.5 / 5.6
        if (nt && pl.C > 256 && pl.C <= 512 && pl.C <= per_cmax * 2 && ((double)pl.C / 512.0 < 0.8 || pl.C % 32 != 0)) {
            const int64_t nb = pl.C * ((1024 + pl.C / 2) / pl.C);          // ~1024 blocks of 512 threads, a multiple of C
            if (nb * pl.C <= pl.np) {
                pl.ysplit = nb;               // the finalize that follows must walk the partial layout this launch produces
                pl.np = nb * pl.C;
                pl.n1 = nb;
                if constexpr (OP == OP_FUSED) hipLaunchKernelGGL((k_col_periodic_pipe<OP, 1, 1, 512>), dim3((unsigned)nb), dim3(512), 0, st, p, (int)pl.C, nb);
                else hipLaunchKernelGGL((k_col_periodic_pipe<OP, 1, 2, 512>), dim3((unsigned)nb), dim3(512), 0, st, p, (int)pl.C, nb);
                return check_hip("periodic column launch") ? -1 : 1;
            }
        }
        const bool ua = pl.C % 4 != 0;
        if ((off & 2) || (ua && (off & 2048))) return 0;
        const int64_t nbx = ceil_div(pl.C, 256);
        {
            // rows per block by operation; never fewer than the plan's (the workspace was sized for the plan's partial count)
            LQ_KNOB(rb_bwd, "LQ_TUNE_COL_RB_K2", (int)kColRbBwd);
            LQ_KNOB(rb_st, "LQ_TUNE_COL_RB_K4", (int)kColRbFused);
            int64_t rb = (OP == OP_BWD) ? rb_bwd : rb_st;
            if (rb < pl.rps) rb = pl.rps;
            if (rb > p.outer) rb = p.outer;
            const int64_t nby = ceil_div(p.outer, rb);
            pl.rps = rb;                      // the finalize that follows must walk the partial layout this launch produces
            pl.ysplit = nby;
            pl.np = nby * pl.C;
            pl.n1 = nby;
        }
        const int64_t blocks = nbx * pl.ysplit;
        if (blocks > 2147483647ll) return 0;
        LQ_KNOB(pipe_r, "LQ_TUNE_PIPE", 28);      // U*10 + NW (scale-gradient ops)
        const int pipe = pipe_r;
#define LQ_PIPE(NT_, U_, NW_) hipLaunchKernelGGL((k_col_pipe<OP, NT_, U_, NW_>), dim3((unsigned)blocks), dim3(NW_ * 64), 0, st, p, pl.C, pl.rps, nbx)
        LQ_KNOB(xcd_remap, "LQ_TUNE_XCD", 1);      // development knob: 0 = natural block order for C % 32 != 0
        if (ua) {
            if (nt) hipLaunchKernelGGL((k_col_pipe<OP, 1, 2, 8, 1>), dim3((unsigned)blocks), dim3(512), 0, st, p, pl.C, pl.rps, nbx);
            else hipLaunchKernelGGL((k_col_pipe<OP, 0, 2, 8, 1>), dim3((unsigned)blocks), dim3(512), 0, st, p, pl.C, pl.rps, nbx);
        } else if (OP == OP_BWD && nt && pl.C % 32 != 0 && nbx > 1 && xcd_remap && pipe == 28) {
            // rows that are not whole lines: neighbouring column blocks share a line -- keep them on one XCD (lq_stream2.hpp)
            if constexpr (OP == OP_BWD) hipLaunchKernelGGL((k_col_pipe<OP, 1, 2, 8, 2>), dim3((unsigned)blocks), dim3(512), 0, st, p, pl.C, pl.rps, nbx);
        } else if (nt) {
#ifdef LQ_DEV_KNOBS
            if (pipe == 14) LQ_PIPE(1, 1, 4); else if (pipe == 44) LQ_PIPE(1, 4, 4); else if (pipe == 18) LQ_PIPE(1, 1, 8);
            else if (pipe == 48) LQ_PIPE(1, 4, 8); else if (pipe == 24) LQ_PIPE(1, 2, 4); else
#endif
            LQ_PIPE(1, 2, 8);
        } else {
            LQ_PIPE(0, 2, 8);
        }
#undef LQ_PIPE
        return check_hip("pipelined column launch") ? -1 : 1;
        }      // OP != OP_FWD
    }
    if constexpr (OP == OP_FWD) {
        return 0;                                      // a forward none of the flat forms above took (e.g. 2^32 or more elements in column mode)
    } else {
        // MODE_ROW_SMALL, scale-gradient ops.  Rows off the 16-byte grid (or, development knob 1024, any row of 68..1020
        // elements): aligned float4 windows (lq_stream2.hpp k_row_win)
        // short rows off the 16-byte grid: one flat window per block (lq_stream2.hpp k_row_seg)
        // -- when the team-per-row form below would leave too many lanes idle (widest window of a row / team size): measured
        // K2 / K4 TB/s, team -> block: L = 63 (17 of 32 lanes): 3.6 / 3.8 -> 4.2 / 4.9; 30: 3.5 / 3.9 -> 4.2 / 4.8; 17: 3.9 / 3.9 ->
        // 4.1 / 4.7; 49 (13 of 16): 4.8 / 4.6 -> 3.5 / 4.3 (keeps the team); 9 with outer = 4: 3.0 / 3.2 -> 2.8 / 3.5
        bool seg = false;
        if (pl.L >= 5 && pl.L <= 64 && pl.L % 4 != 0) {
            const int nw = (int)(pl.L + 6) / 4;
            int tl = 1;
            while ((1 << tl) < nw) ++tl;
            const double eff = (double)nw / (double)(1 << tl);
            seg = eff < (OP == OP_FUSED ? 0.8 : 0.7);
        }
        if (!(off & 16384) && seg && pl.R < 4294967296ll) {
            constexpr int kSegU = 2;
            const int rpb = (kBlock * kSegU * 4 - 3) / (int)pl.L;
            const int64_t blocks = ceil_div(pl.R, (int64_t)rpb);
            if (blocks <= 2147483647ll) {
                const FastDiv fL = make_fastdiv((uint32_t)pl.L), fG = make_fastdiv((uint32_t)p.G);
                if (nt) hipLaunchKernelGGL((k_row_seg<OP, 1, kSegU>), dim3((unsigned)blocks), dim3(kBlock), 0, st, p, fL, fG, pl.R, (int)pl.L, rpb, n);
                else hipLaunchKernelGGL((k_row_seg<OP, 0, kSegU>), dim3((unsigned)blocks), dim3(kBlock), 0, st, p, fL, fG, pl.R, (int)pl.L, rpb, n);
                return check_hip("row-segment launch") ? -1 : 1;
            }
        }
        LQ_KNOB(win_all, "LQ_TUNE_WIN", 1);      // 0: rows with L % 4 == 0 and L > 64 keep the round-1 row-small kernel
        if (!(off & 128) && pl.L >= 5 && pl.R < 4294967296ll && (pl.L % 4 != 0 || (win_all && pl.L > 64))) {
            const int nwin = (int)(pl.L % 4 ? (pl.L + 3 + 3) / 4 : pl.L / 4);     // float4s of the widest window of a row
            int lg = 1;
            while ((1 << lg) < nwin && lg < 6) ++lg;
            int V = lg == 6 ? (nwin + 63) / 64 : 1;
            // Round 3: rows of 65..340 elements.  The smallest power-of-two team with one float4 per lane leaves half the lanes
            // idle just above a power of two (rows of 68: 17 of 32 lanes) -- and, what costs more, pays the per-row work (context,
            // team reduction, emit) once per 32 or 64 lanes.  Teams of 16 / 32 lanes with 2-3 float4 per lane put two to four times
            // as many rows into a wave at 256- / 512-byte pieces per team and load.  Measured on 33.5 M elements, outer == 1
            // (tools/r03_win_geom.sh, profiles/r03/short_rows/): K2 rows of 68: 4.0 -> 5.0 TB/s, 88: 5.0 -> 6.0, 131: 3.7 -> 5.7,
            // 160: 4.9 -> 6.3, 260: 5.4 -> 6.1; K4 keeps the wider team where its stores want it (rows of 84..124 unless they are
            // whole 64-byte multiples): 68: 4.5 -> 4.9, 132: 4.5 -> 5.0, 160: 5.2 -> 6.0, 200: 5.5 -> 5.8.  (Teams of 8 lanes --
            // 128-byte pieces -- and two rows per team next to several float4 per lane both measured worse.)
            if constexpr (OP == OP_BWD) {
                if (nwin >= 17 && nwin <= 32) { lg = 4; V = 2; }
                else if (nwin >= 33 && nwin <= 48) { lg = 4; V = 3; }
                else if (nwin >= 49 && nwin <= 55) { lg = 5; V = 2; }
                else if (nwin >= 65 && nwin <= 84) { lg = 5; V = 3; }
            } else {
                if (nwin >= 17 && nwin <= 32 && (nwin <= 19 || pl.L % 16 == 0)) { lg = 4; V = 2; }
                else if (nwin >= 33 && nwin <= 64) { lg = 5; V = 2; }
            }
            const int U = V == 1 ? 2 : 1;
            const int64_t rows_per_block = (int64_t)kWavesPerBlock * (64 >> lg) * U;
            const int64_t blocks = ceil_div(pl.R, rows_per_block);
            if (blocks <= 2147483647ll && V <= 5) {
                const FastDiv fG = make_fastdiv((uint32_t)p.G);
#define LQ_WIN(NT_, LG_, V_, U_) hipLaunchKernelGGL((k_row_win<OP, NT_, LG_, V_, U_>), dim3((unsigned)blocks), dim3(kBlock), 0, st, p, fG, pl.R, (int)pl.L, n)
                // (only the combinations the rules above can produce are compiled: K2 never runs 32 lanes x 1, K4 never 64 x 1)
#define LQ_WIN2(NT_) do { \
                switch (lg * 8 + V) { \
                    case 1 * 8 + 1: LQ_WIN(NT_, 1, 1, 2); break; \
                    case 2 * 8 + 1: LQ_WIN(NT_, 2, 1, 2); break; \
                    case 3 * 8 + 1: LQ_WIN(NT_, 3, 1, 2); break; \
                    case 4 * 8 + 1: LQ_WIN(NT_, 4, 1, 2); break; \
                    case 4 * 8 + 2: LQ_WIN(NT_, 4, 2, 1); break; \
                    case 4 * 8 + 3: if constexpr (OP == OP_BWD || kDevKnobs) LQ_WIN(NT_, 4, 3, 1); break; \
                    case 5 * 8 + 1: if constexpr (OP != OP_BWD || kDevKnobs) LQ_WIN(NT_, 5, 1, 2); break; \
                    case 5 * 8 + 2: LQ_WIN(NT_, 5, 2, 1); break; \
                    case 5 * 8 + 3: if constexpr (OP == OP_BWD || kDevKnobs) LQ_WIN(NT_, 5, 3, 1); break; \
                    case 6 * 8 + 1: if constexpr (OP == OP_BWD || kDevKnobs) LQ_WIN(NT_, 6, 1, 2); break; \
                    case 6 * 8 + 2: LQ_WIN(NT_, 6, 2, 1); break; \
                    case 6 * 8 + 3: LQ_WIN(NT_, 6, 3, 1); break; \
                    case 6 * 8 + 4: LQ_WIN(NT_, 6, 4, 1); break; \
                    default: LQ_WIN(NT_, 6, 5, 1); break; \
                } } while (0)
#ifdef LQ_DEV_KNOBS
                // development: LQ_TUNE_WIN_GEOM = LG * 100 + V * 10 + U forces the team geometry (teams of 2^LG lanes, V float4 per lane,
                // U rows per team and wave)
                LQ_KNOB(geom, "LQ_TUNE_WIN_GEOM", 0);
                if (geom >= 100 && nt && ((geom / 10) % 10) * (1 << (geom / 100)) >= nwin) {
                    const int glg = geom / 100, gu = geom % 10;
                    const int64_t gblocks = ceil_div(pl.R, (int64_t)kWavesPerBlock * (64 >> glg) * gu);
                    bool done = true;
#define LQ_WING(LG_, V_, U_) hipLaunchKernelGGL((k_row_win<OP, 1, LG_, V_, U_>), dim3((unsigned)gblocks), dim3(kBlock), 0, st, p, fG, pl.R, (int)pl.L, n)
                    switch (geom) {
                        case 421: LQ_WING(4, 2, 1); break;
                        case 422: LQ_WING(4, 2, 2); break;
                        case 431: LQ_WING(4, 3, 1); break;
                        case 432: LQ_WING(4, 3, 2); break;
                        case 521: LQ_WING(5, 2, 1); break;
                        case 522: LQ_WING(5, 2, 2); break;
                        case 531: LQ_WING(5, 3, 1); break;
                        case 532: LQ_WING(5, 3, 2); break;
                        case 541: LQ_WING(5, 4, 1); break;
                        case 414: LQ_WING(4, 1, 4); break;
                        case 514: LQ_WING(5, 1, 4); break;
                        default: done = false;
                    }
#undef LQ_WING
                    if (done) return check_hip("row-window launch") ? -1 : 1;
                }
#endif
                if (nt) LQ_WIN2(1); else LQ_WIN2(0);
#undef LQ_WIN2
#undef LQ_WIN
                return check_hip("row-window launch") ? -1 : 1;
            }
        }
        if ((off & 8) || pl.L > 64 || pl.L < 8 || pl.L % 4 != 0) return 0;
        const int lg = row_small_lpr_log2_vec(pl.L);         // 1..4
        LQ_KNOB(tiny_u, "LQ_TUNE_TINY_U", 2);
        int U = (1 << lg) < tiny_u ? (1 << lg) : tiny_u;
        if (U != 2 && U != 4) U = 2;
        const int64_t rows_per_block = (int64_t)kWavesPerBlock * (64 >> lg) * U;
        const int64_t blocks = ceil_div(pl.R, rows_per_block);
        if (blocks > 2147483647ll) return 0;
        // (2^32 or more rows -- 128 GiB per stream at 8 elements a row -- keep the round-1 kernel: a 64-bit row modulo form of
        // this kernel could not be exercised by any test)
        if (pl.R >= 4294967296ll) return 0;
        const int gm = (p.outer == 1) ? 0 : 1;
#define LQ_TINY3(NT_, U_, LG_) do { \
            if (gm == 0) hipLaunchKernelGGL((k_row_tiny<OP, NT_, U_, LG_, 0>), dim3((unsigned)blocks), dim3(kBlock), 0, st, p, pl.R, (int)pl.L); \
            else hipLaunchKernelGGL((k_row_tiny<OP, NT_, U_, LG_, 1>), dim3((unsigned)blocks), dim3(kBlock), 0, st, p, pl.R, (int)pl.L); } while (0)
#ifdef LQ_DEV_KNOBS
#define LQ_TINY2(NT_) do { \
            if (lg == 1) LQ_TINY3(NT_, 2, 1); \
            else if (lg == 2) { if (U == 2) LQ_TINY3(NT_, 2, 2); else LQ_TINY3(NT_, 4, 2); } \
            else if (lg == 3) { if (U == 2) LQ_TINY3(NT_, 2, 3); else LQ_TINY3(NT_, 4, 3); } \
            else { if (U == 2) LQ_TINY3(NT_, 2, 4); else LQ_TINY3(NT_, 4, 4); } } while (0)
#else       // two passes of rows per wave (four measured no better: profiles/r02/tuning_sweeps.txt)
#define LQ_TINY2(NT_) do { \
            if (lg == 1) LQ_TINY3(NT_, 2, 1); \
            else if (lg == 2) LQ_TINY3(NT_, 2, 2); \
            else if (lg == 3) LQ_TINY3(NT_, 2, 3); \
            else LQ_TINY3(NT_, 2, 4); } while (0)
#endif
        if (nt) LQ_TINY2(1); else LQ_TINY2(0);
#undef LQ_TINY2
#undef LQ_TINY3
        return check_hip("tiny-row launch") ? -1 : 1;
    }
}

// The streaming geometry of long rows (512-thread units, nontemporal accesses, two float4 per thread) exists for K1, K2 and K4 --
// the operations that run on activation-sized tensors.  The penalty terms, the integer view and the element-wise OIHW
// companion work on weight-sized tensors: they keep the 256-thread units at every size (their entry points plan with
// make_plan(..., kBlock)), which is a third of the row-stream instantiations and none that a parity test could not reach.
template <int OP>
constexpr bool kStreamOp = OP == OP_FWD || OP == OP_BWD || OP == OP_FUSED;

template <int OP>
static int launch_traverse(Plan& pl, const Params& p, hipStream_t st) {
    using O = OpT<OP>;
    if (!kStreamOp<OP> && pl.mode == MODE_ROW_BIG && pl.bs != kBlock) return fail(LQ_EINVAL, "internal: operation %d planned with %d-thread units", OP, pl.bs);
    {
        const int r2 = launch_stream2<OP>(pl, p, st);
        if (r2 < 0) return LQ_EHIP;
        if (r2 > 0) return LQ_OK;
    }
    if (pl.mode == MODE_ROW_BIG) {
        // float4 path: 16-byte aligned bases; rows of any length (a scalar head/tail of <= 3 elements re-aligns each chunk)
        const bool vec = aligned(p.P, 16) && (!O::kDy || aligned(p.dy, 16)) && (!O::kStore || aligned(p.out, 16));
        // Chunk size by row length (streaming sizes, where the plan says 512 threads).  A row is cut into chunks of CH
        // elements and its last chunk is whatever remains: rows of 2500 elements fill 61 % of two 2048-chunks but 81 % of
        // three 1024-chunks.  Measured, K4 at 512 -> 256 threads: rows of 2500: 4.4 -> 6.0 TB/s, 3000: 5.1 -> 6.1, 5000: 5.8 ->
        // 6.3, 6000 (same fill): 6.2 = 6.2; K2 (two float4 per thread, CH = 4096): 5000: 5.7 -> 6.5, 6000: 5.9 -> 6.6, but a
        // row that is ONE chunk keeps it (2500: 6.8 against 6.3).  The BENCH rows (50176 = 24.5 x 2048) keep 512 threads.
        LQ_KNOB(tune_chunk, "LQ_TUNE_CHUNK", 1);     // development knob: 0 = always the plan's block size
        LQ_KNOB(forced_bs, "LQ_TUNE_BS", 0);
        if (pl.bs == 512 && vec && tune_chunk && !p.direct && !forced_bs && pl.np_ws >= pl.R * row_chunks(pl.L, 1024)) {
            auto fill = [&](int64_t CH) { return (double)pl.L / (double)(row_chunks(pl.L, CH) * CH); };
            const bool k2_form = (OP == OP_BWD || (OP == OP_FUSED && p.tmode >= 1)) && (double)pl.R * (double)pl.L * 4.0 >= (double)kNtBytes;
            const bool small = k2_form ? (row_chunks(pl.L, 4096) > 1 && fill(1024) > fill(4096) + 0.1) : (fill(1024) > fill(2048) + 0.05);
            if (small) {
                pl.bs = 256;
                pl.CH = 1024;
                pl.nc = row_chunks(pl.L, 1024);
                pl.np = pl.R * pl.nc;              // the finalize that follows must walk the partial layout this launch produces
                pl.gstride = pl.nc;
                pl.stride1 = p.G * pl.nc;
                pl.n2 = pl.nc;
            }
        }
        const int64_t units = pl.R * pl.nc;
        if (units > 2147483647ll) return fail(LQ_EINVAL, "too many work units (%lld)", (long long)units);
        const int64_t outer_f = pl.R / p.G;
        const int grid3d = (p.G <= 65535 && outer_f <= 65535 && pl.R == outer_f * p.G) ? 1 : 0;
        const dim3 grid = grid3d ? dim3((unsigned)pl.nc, (unsigned)p.G, (unsigned)outer_f) : dim3((unsigned)units);
        const bool nt = kStreamOp<OP> && vec && (double)pl.R * (double)pl.L * 4.0 >= (double)kNtBytes;
        // Two float4 per thread and stream (see row_stream_body): the backward always (measured 100.3 vs 101.4 us per
        // BENCH step, and 105 vs 116 us at lambda = 1e-3 where every element takes the exact-ratio + tanh branch; it
        // also halves the partials the finalize walks); the fused kernel only when lambda >= 4e-4 (77.4 vs 81.7 us
        // there, but 77.7 vs 75.4 us at small lambda).  The unit doubles, so the chunk count halves; the workspace
        // bound (computed for one float4 per thread) still holds.  Four float4 per thread in K2 were measured in round 2:
        // 49.5 against 48.0 us on the BENCH tensor -- not better.
        LQ_KNOB(tune_u2, "LQ_TUNE_U2", -1);   // force 0/1
        const bool want_u2 = tune_u2 >= 0 ? tune_u2 == 1 : (OP == OP_BWD || p.tmode >= 1);
        const bool u2 = (OP == OP_BWD || OP == OP_FUSED) && want_u2 && vec && nt && pl.bs == 512;
        // rows with a folded tail or with chunks off the 16-byte grid take the TAIL instantiation (see row_stream_body)
        auto needs_tail = [&](int64_t CH) {
            const int64_t r = pl.L % CH;
            return pl.L % 4 != 0 || (pl.L > CH && r != 0 && r * 8 <= CH);
        };
        if constexpr (OP == OP_BWD || OP == OP_FUSED) if (u2) {
            const int64_t nc2 = row_chunks(pl.L, (int64_t)pl.bs * 8);
            const dim3 grid2 = grid3d ? dim3((unsigned)nc2, (unsigned)p.G, (unsigned)outer_f) : dim3((unsigned)(pl.R * nc2));
            if (needs_tail((int64_t)pl.bs * 8))
                hipExtLaunchKernelGGL((k_row_stream<OP, 4, 512, 1, 2, 1>), grid2, dim3(512), 0, st, g_prof_start, g_prof_stop, 0, p, pl.L, nc2, grid3d);
            else
                hipExtLaunchKernelGGL((k_row_stream<OP, 4, 512, 1, 2, 0>), grid2, dim3(512), 0, st, g_prof_start, g_prof_stop, 0, p, pl.L, nc2, grid3d);
            // the finalize that follows must walk the partial layout this launch produced
            pl.CH = pl.bs * 8;
            pl.nc = nc2;
            pl.np = pl.R * nc2;
            pl.gstride = nc2;
            pl.stride1 = p.G * nc2;
            pl.n2 = nc2;
            return check_hip("traversal launch");
        }
        // hipExtLaunchKernelGGL with NULL events is a plain launch; with lq_profile_events() set, the events take the kernel's own
        // begin / end timestamps (what rocprofv3 reports as its duration)
        const bool tail1 = needs_tail((int64_t)pl.CH);
        // (TAIL = 0 forms that no descriptor reaches are not compiled: K1 beyond 256-thread default-policy units, K2 with
        // nontemporal 512-thread units -- see the two internal errors below)
#define LQ_LAUNCH_STREAM(VEC_, BS_, NT_) do { \
        constexpr bool kTail0 = kDevKnobs || (VEC_ == 4 && !(OP == OP_FWD && (BS_ != kBlock || NT_ != 0)) && !(OP == OP_BWD && BS_ == 512 && NT_ == 1)); \
        constexpr bool kTail1 = kDevKnobs || !(OP == OP_BWD && VEC_ == 4 && BS_ == 512 && NT_ == 1); \
        if (VEC_ == 1 || tail1) { if constexpr (kTail1) hipExtLaunchKernelGGL((k_row_stream<OP, VEC_, BS_, NT_, 1, 1>), grid, dim3(BS_), 0, st, g_prof_start, g_prof_stop, 0, p, pl.L, pl.nc, grid3d); } \
        else { if constexpr (kTail0) hipExtLaunchKernelGGL((k_row_stream<OP, VEC_, BS_, NT_, 1, 0>), grid, dim3(BS_), 0, st, g_prof_start, g_prof_stop, 0, p, pl.L, pl.nc, grid3d); } } while (0)
        if constexpr (!kStreamOp<OP>) {
            if (vec) LQ_LAUNCH_STREAM(4, 256, 0);
            else LQ_LAUNCH_STREAM(1, 256, 0);
        } else if (!kDevKnobs && OP == OP_FWD && vec && !tail1 && (pl.bs != kBlock || nt)) {
            // K1 of aligned rows with L % 4 == 0 at streaming size is the flat one-shot stream (launch_stream2): no row-stream
            // instantiation exists for it
            return fail(LQ_EINVAL, "internal: streaming-size forward of aligned rows reached the row stream");
        } else if (!kDevKnobs && OP == OP_BWD && vec && nt && pl.bs == 512) {
            // K2 with nontemporal 512-thread units always takes two float4 per thread (u2 above)
            return fail(LQ_EINVAL, "internal: streaming-size scale gradient reached the one-float4 row stream");
        } else if (vec) {
#ifdef LQ_DEV_KNOBS
            if (pl.bs == 1024) {
                if (nt) LQ_LAUNCH_STREAM(4, 1024, 1);
                else LQ_LAUNCH_STREAM(4, 1024, 0);
            } else
#endif
            if (pl.bs == 512) {
                if (nt) LQ_LAUNCH_STREAM(4, 512, 1);
                else LQ_LAUNCH_STREAM(4, 512, 0);
            } else {
                if (nt) LQ_LAUNCH_STREAM(4, 256, 1);
                else LQ_LAUNCH_STREAM(4, 256, 0);
            }
        } else {
#ifdef LQ_DEV_KNOBS
            if (pl.bs == 1024) LQ_LAUNCH_STREAM(1, 1024, 0);
            else
#endif
            if (pl.bs == 512) LQ_LAUNCH_STREAM(1, 512, 0);
            else LQ_LAUNCH_STREAM(1, 256, 0);
        }
#undef LQ_LAUNCH_STREAM
    } else if (pl.mode == MODE_ROW_SMALL) {
        const bool vec = row_small_vec(pl, p.P, O::kDy ? p.dy : nullptr, O::kStore ? p.out : nullptr);
        const int lg = vec ? row_small_lpr_log2_vec(pl.L) : pl.lpr_log2;
        const int64_t blocks = ceil_div(pl.R, kBlock >> lg);
        if (blocks > 2147483647ll) return fail(LQ_EINVAL, "too many blocks (%lld)", (long long)blocks);
        if (vec) hipLaunchKernelGGL((k_row_small<OP, 4>), dim3((unsigned)blocks), dim3(kBlock), 0, st, p, pl.R, (int)pl.L, lg);
        else hipLaunchKernelGGL((k_row_small<OP, 1>), dim3((unsigned)blocks), dim3(kBlock), 0, st, p, pl.R, (int)pl.L, lg);
    } else {
        int variant;
        int64_t nbx;
        col_variant(pl, p.P, O::kDy ? p.dy : nullptr, O::kStore ? p.out : nullptr, variant, nbx);
        const int64_t blocks = nbx * pl.ysplit;
        if (blocks > 2147483647ll) return fail(LQ_EINVAL, "too many blocks (%lld)", (long long)blocks);
        hipLaunchKernelGGL((k_col<OP>), dim3((unsigned)blocks), dim3(kBlock), 0, st, p, pl.C, pl.rps, nbx, variant, pl.ysplit);
    }
    return check_hip("traversal launch");
}

// The column form (lq_traverse.hpp) of a single launch pays four waves per 64 columns walking n1 rows eight at a time: it wins
// where the other forms gather megabytes of scattered words (6144 x 6144 column-wise: 8.7 -> 5.6 us; 4096 x 4096, whose 32
// partials per group used to go to the thread form: 10.3 -> 4.8 us) and loses on small partial sets or many rows per column
// (16384 x 1001: 4.9 -> 8.4 us, eleven dependent rounds in 16 blocks).  The multi-tensor batch uses it wherever the geometry
// allows (its tasks share one launch, the amplified gathers add up: lq_batch.hpp).
static bool finalize_cols_single(const FinGeom& f) {
    return finalize_cols_ok(f.groups, f.gstride, f.n1, f.stride1, f.n2) && f.n1 <= 64 && f.n1 * f.groups * f.n2 >= 131072;
}

template <int OP, bool COLS = true>      // COLS = false: global reductions (one group) never take the column form
static int launch_finalize(const Params& p, FinGeom f, hipStream_t st) {
    const int form = finalize_form(f.groups, f.n1, f.stride1, f.n2);       // lq_traverse.hpp
    if (COLS && finalize_cols_single(f)) {
        if constexpr (COLS) hipLaunchKernelGGL((k_finalize_cols<OP>), dim3((unsigned)ceil_div(f.groups, 64 / f.n2)), dim3(kBlock), 0, st, p, f);
    } else if (form == 0) {
        hipLaunchKernelGGL((k_finalize_thread<OP>), dim3((unsigned)ceil_div(f.groups, kBlock)), dim3(kBlock), 0, st, p, f);
    } else if (form == 1) {
        hipLaunchKernelGGL((k_finalize_block<OP, 64>), dim3((unsigned)f.groups), dim3(64), 0, st, p, f);
    } else if (form == 2) {
        hipLaunchKernelGGL((k_finalize_block<OP, 256>), dim3((unsigned)f.groups), dim3(256), 0, st, p, f);
    } else {
        hipLaunchKernelGGL((k_finalize_block<OP, 1024>), dim3((unsigned)f.groups), dim3(1024), 0, st, p, f);
    }
    return check_hip("finalize launch");
}

static FinGeom group_geom(const Plan& pl, int64_t outer, int64_t G, int64_t inner) {
    FinGeom f;
    memset(&f, 0, sizeof(f));
    f.groups = G;
    f.gstride = pl.gstride;
    f.n1 = pl.n1;
    f.stride1 = pl.stride1;
    f.n2 = pl.n2;
    f.count = (double)outer * (double)inner;
    return f;
}

static FinGeom global_geom(const Plan& pl, int64_t outer, int64_t G, int64_t inner) {
    FinGeom f;
    memset(&f, 0, sizeof(f));
    f.groups = 1;
    f.gstride = 0;
    f.n1 = 1;
    f.stride1 = 0;
    f.n2 = pl.np;
    f.count = (double)outer * (double)G * (double)inner;
    return f;
}

static Params base_params(const float* P, const float* s, int64_t outer, int64_t G, int64_t inner) {
    Params p;
    memset(&p, 0, sizeof(p));
    p.P = P;
    p.s = s;
    p.outer = outer;
    p.G = G;
    p.inner = inner;
    return p;
}

static bool make_conv_tile(ConvTile& ct, int64_t hw, int64_t ci, int64_t co, int64_t outer, int64_t G, int64_t inner);
static int64_t conv_tile_partials(const ConvTile& ct, int64_t G);
static void conv_tile_fin(const ConvTile& ct, int64_t G, int64_t& gstride, int64_t& n1, int64_t& stride1, int64_t& n2);

}  // namespace lq

using namespace lq;

#define LQ_REQUIRE_PTR(x)                                                      \
    do {                                                                       \
        if (!(x)) return fail(LQ_EINVAL, "%s: argument '%s' is NULL", __func__, #x); \
        if (!aligned((x), 4)) return fail(LQ_EALIGN, "%s: argument '%s' is not 4-byte aligned", __func__, #x); \
    } while (0)

extern "C" {

int lq_version(void) { return LQ_ABI_VERSION; }

#ifdef LQ_DEV_KNOBS
int lq_dev_set_ablate(int mask) {      // development builds only (see lq_conv_tile.hpp)
    return hipMemcpyToSymbol(HIP_SYMBOL(lq::g_ablate), &mask, sizeof(int)) == hipSuccess ? LQ_OK : LQ_EHIP;
}
#endif

#ifdef LQ_DEV_KNOBS
int lq_dev_set_flags(int bits) {       // development builds only (see lq_stream2.hpp)
    return hipMemcpyToSymbol(HIP_SYMBOL(lq::g_dev_flags), &bits, sizeof(int)) == hipSuccess ? LQ_OK : LQ_EHIP;
}
#endif

#ifdef LQ_DEV_KNOBS
int lq_dev_set_trace(void* buf) {      // development builds only: block timeline of the batch traversals (lq_conv_tile.hpp)
    unsigned long long* b = (unsigned long long*)buf;
    return hipMemcpyToSymbol(HIP_SYMBOL(lq::g_trace), &b, sizeof(b)) == hipSuccess ? LQ_OK : LQ_EHIP;
}
#endif

int lq_profile_events(void* start, void* stop) {
    g_prof_start = (hipEvent_t)start;
    g_prof_stop = (hipEvent_t)stop;
    return LQ_OK;
}

const char* lq_last_error(void) { return g_err; }

const char* lq_status_string(int status) {
    switch (status) {
        case LQ_OK: return "LQ_OK";
        case LQ_EINVAL: return "LQ_EINVAL";
        case LQ_EHIP: return "LQ_EHIP";
        case LQ_EWORKSPACE: return "LQ_EWORKSPACE";
        case LQ_EALIGN: return "LQ_EALIGN";
        default: return "LQ_UNKNOWN";
    }
}

size_t lq_workspace_bytes(int64_t outer, int64_t G, int64_t inner) {
    if (outer <= 0 || G <= 0 || inner <= 0) return 0;
    return ws_bytes_for(make_plan(outer, G, inner));
}

int lq_fq_forward(const float* P, const float* s, float* out, void* q, int q_dtype, int64_t outer, int64_t G,
                  int64_t inner, void* stream) {
    int rc = check_desc(outer, G, inner);
    if (rc) return rc;
    LQ_REQUIRE_PTR(P);
    LQ_REQUIRE_PTR(s);
    if (!out && !q) return fail(LQ_EINVAL, "lq_fq_forward: both out and q are NULL");
    if (q_dtype < LQ_Q_NONE || q_dtype > LQ_Q_I8) return fail(LQ_EINVAL, "lq_fq_forward: bad q_dtype %d", q_dtype);
    if ((q != nullptr) != (q_dtype != LQ_Q_NONE)) return fail(LQ_EINVAL, "lq_fq_forward: q and q_dtype disagree");
    Plan pl = make_plan(outer, G, inner);
    Params p = base_params(P, s, outer, G, inner);
    p.q = q;
    p.q_dtype = q_dtype;
    if (out) {
        if (!aligned(out, 4)) return fail(LQ_EALIGN, "lq_fq_forward: out is not 4-byte aligned");
        p.out = out;
        return launch_traverse<OP_FWD>(pl, p, (hipStream_t)stream);
    }
    pl = make_plan(outer, G, inner, kBlock);      // the integer view alone (callbacks / export): 256-thread units (kStreamOp)
    return launch_traverse<OP_QONLY>(pl, p, (hipStream_t)stream);
}

static int check_conv(const char* fn, int64_t hw, int64_t ci, int64_t co, int64_t outer, int64_t G, int64_t inner) {
    if (hw <= 0 || ci <= 0 || co <= 0) return fail(LQ_EINVAL, "%s: hw, ci, co must be positive", fn);
    const double n = (double)hw * (double)ci * (double)co;
    if (n != (double)outer * (double)G * (double)inner) return fail(LQ_EINVAL, "%s: hw*ci*co does not match the tensor", fn);
    if (n >= 4294967296.0) return fail(LQ_EINVAL, "%s: conv kernels must have fewer than 2^32 elements", fn);
    return LQ_OK;
}

int lq_fq_forward_oihw(const float* P, const float* s, float* out, float* out_oihw, int64_t hw, int64_t ci, int64_t co,
                       int64_t outer, int64_t G, int64_t inner, void* stream) {
    int rc = check_desc(outer, G, inner);
    if (rc) return rc;
    if ((rc = check_conv("lq_fq_forward_oihw", hw, ci, co, outer, G, inner))) return rc;
    LQ_REQUIRE_PTR(P);
    LQ_REQUIRE_PTR(s);
    LQ_REQUIRE_PTR(out_oihw);
    if (!out && !(aligned(P, 16) && lq_conv_tile_supported(hw, ci, co, outer, G, inner)))
        return fail(LQ_EINVAL, "lq_fq_forward_oihw: out may be NULL only for kernels the LDS-tile path takes (lq_conv_tile_supported) with P 16-byte aligned");
    Plan pl = make_plan(outer, G, inner, kBlock);      // weight-sized operation: 256-thread units at every size (kStreamOp)
    Params p = base_params(P, s, outer, G, inner);
    p.out = out;
    p.out_perm = out_oihw;
    p.perm_hw = (uint32_t)hw;
    p.perm_ci = (uint32_t)ci;
    p.perm_co = (uint32_t)co;
    ConvTile ct;
    if (aligned(P, 16) && aligned(out, 16) && make_conv_tile(ct, hw, ci, co, outer, G, inner)) {      // LDS tiles (lq_conv_tile.hpp)
        hipLaunchKernelGGL(k_conv_tile_fwd, dim3(ct.ntc * ct.nto), dim3(kBlock), 0, (hipStream_t)stream, p, ct);
        return check_hip("conv tile forward launch");
    }
    return launch_traverse<OP_FWD_PERM>(pl, p, (hipStream_t)stream);      // element-wise companion (kernels with > 9 taps, co % 4 != 0, ...)
}

int lq_conv_tile_supported(int64_t hw, int64_t ci, int64_t co, int64_t outer, int64_t G, int64_t inner) {
    if (outer <= 0 || G <= 0 || inner <= 0 || hw <= 0 || ci <= 0 || co <= 0) return 0;
    if ((double)hw * (double)ci * (double)co != (double)outer * (double)G * (double)inner) return 0;
    ConvTile ct;
    return make_conv_tile(ct, hw, ci, co, outer, G, inner) ? 1 : 0;
}

size_t lq_conv_workspace_bytes(int64_t hw, int64_t ci, int64_t co, int64_t outer, int64_t G, int64_t inner) {
    if (outer <= 0 || G <= 0 || inner <= 0) return 0;
    size_t need = ws_bytes_for(make_plan(outer, G, inner));
    ConvTile ct;
    if (hw > 0 && ci > 0 && co > 0 && make_conv_tile(ct, hw, ci, co, outer, G, inner)) {
        Plan tp;
        memset(&tp, 0, sizeof(tp));
        tp.np = conv_tile_partials(ct, G);
        const size_t t = ws_bytes_for(tp);
        if (t > need) need = t;
    }
    return need;
}

int lq_fq_scale_grad_oihw(const float* P, const float* s, const float* dy_oihw, float lambda, float* ds, float* dP,
                          void* ws, size_t ws_bytes, int64_t hw, int64_t ci, int64_t co,
                          int64_t outer, int64_t G, int64_t inner, void* stream) {
    int rc = check_desc(outer, G, inner);
    if (rc) return rc;
    if ((rc = check_conv("lq_fq_scale_grad_oihw", hw, ci, co, outer, G, inner))) return rc;
    LQ_REQUIRE_PTR(P);
    LQ_REQUIRE_PTR(s);
    LQ_REQUIRE_PTR(dy_oihw);
    LQ_REQUIRE_PTR(ds);
    LQ_REQUIRE_PTR(dP);
    Plan pl = make_plan(outer, G, inner, kBlock);      // weight-sized operation: 256-thread units at every size (kStreamOp)
    Params p = base_params(P, s, outer, G, inner);
    p.dy = P;                     // valid dummy for the traversals' dy loads; the op gathers from dy_perm
    p.dy_perm = dy_oihw;
    p.dp_out = dP;
    p.perm_hw = (uint32_t)hw;
    p.perm_ci = (uint32_t)ci;
    p.perm_co = (uint32_t)co;
    p.lam = lambda;
    p.tmode = (lambda < 4.0e-4f) ? 0 : ((lambda <= 0.25f) ? 1 : 2);
    ConvTile ct;
    if (aligned(P, 16) && aligned(dP, 16) && make_conv_tile(ct, hw, ci, co, outer, G, inner)) {       // LDS tiles (lq_conv_tile.hpp)
        Plan tp;
        memset(&tp, 0, sizeof(tp));
        tp.np = conv_tile_partials(ct, G);
        conv_tile_fin(ct, G, tp.gstride, tp.n1, tp.stride1, tp.n2);
        if (ws && ws_bytes < ws_bytes_for(tp))
            return fail(LQ_EWORKSPACE, "lq_fq_scale_grad_oihw: workspace too small: %zu < %zu bytes (size it with lq_conv_workspace_bytes)",
                        ws_bytes, ws_bytes_for(tp));
        if ((rc = bind_ws(p, tp, ws, ws_bytes))) return rc;
        hipLaunchKernelGGL(k_conv_tile_bwd, dim3(ct.ntc * ct.nto), dim3(kBlock), 0, (hipStream_t)stream, p, ct);
        if ((rc = check_hip("conv tile backward launch"))) return rc;
        FinGeom f = group_geom(tp, outer, G, inner);
        f.o0 = ds;
        return launch_finalize<OP_BWD>(p, f, (hipStream_t)stream);
    }
    if ((rc = bind_ws(p, pl, ws, ws_bytes))) return rc;
    const bool direct = pl.n1 * pl.n2 == 1;
    if (direct) {
        p.direct = 1;
        p.e0 = ds;
        p.e1 = nullptr;
        p.ecount = (double)outer * (double)inner;
    }
    if ((rc = launch_traverse<OP_BWD_PERM>(pl, p, (hipStream_t)stream))) return rc;
    if (direct) return LQ_OK;
    FinGeom f = group_geom(pl, outer, G, inner);
    f.o0 = ds;
    return launch_finalize<OP_BWD>(p, f, (hipStream_t)stream);
}

int lq_fq_scale_grad(const float* P, const float* s, const float* dy, float lambda, float* ds, float* parts, void* ws,
                     size_t ws_bytes, int64_t outer, int64_t G, int64_t inner, void* stream) {
    int rc = check_desc(outer, G, inner);
    if (rc) return rc;
    LQ_REQUIRE_PTR(P);
    LQ_REQUIRE_PTR(s);
    LQ_REQUIRE_PTR(dy);
    LQ_REQUIRE_PTR(ds);
    Plan pl = make_plan(outer, G, inner);
    Params p = base_params(P, s, outer, G, inner);
    p.dy = dy;
    p.lam = lambda;
    p.tmode = (lambda < 4.0e-4f) ? 0 : ((lambda <= 0.25f) ? 1 : 2);   // NaN lambda -> 2
    if ((rc = bind_ws(p, pl, ws, ws_bytes))) return rc;
    const bool direct = pl.n1 * pl.n2 == 1;      // one partial per group: the traversal emits ds itself, no finalize launch
    if (direct) {
        p.direct = 1;
        p.e0 = ds;
        p.e1 = parts;
        p.ecount = (double)outer * (double)inner;
    }
    if ((rc = launch_traverse<OP_BWD>(pl, p, (hipStream_t)stream))) return rc;
    if (direct) return LQ_OK;
    FinGeom f = group_geom(pl, outer, G, inner);
    f.o0 = ds;
    f.o1 = parts;
    return launch_finalize<OP_BWD>(p, f, (hipStream_t)stream);
}

int lq_fq_fwd_bwd_fused(const float* P, const float* s, const float* dy, float lambda, float* out, float* ds, void* ws,
                        size_t ws_bytes, int64_t outer, int64_t G, int64_t inner, void* stream) {
    int rc = check_desc(outer, G, inner);
    if (rc) return rc;
    LQ_REQUIRE_PTR(P);
    LQ_REQUIRE_PTR(s);
    LQ_REQUIRE_PTR(dy);
    LQ_REQUIRE_PTR(out);
    LQ_REQUIRE_PTR(ds);
    Plan pl = make_plan(outer, G, inner);
    Params p = base_params(P, s, outer, G, inner);
    p.dy = dy;
    p.lam = lambda;
    p.tmode = (lambda < 4.0e-4f) ? 0 : ((lambda <= 0.25f) ? 1 : 2);   // NaN lambda -> 2
    p.out = out;
    if ((rc = bind_ws(p, pl, ws, ws_bytes))) return rc;
    const bool direct = pl.n1 * pl.n2 == 1;
    if (direct) {
        p.direct = 1;
        p.e0 = ds;
        p.e1 = nullptr;
        p.ecount = (double)outer * (double)inner;
    }
    if ((rc = launch_traverse<OP_FUSED>(pl, p, (hipStream_t)stream))) return rc;
    if (direct) return LQ_OK;
    FinGeom f = group_geom(pl, outer, G, inner);
    f.o0 = ds;
    return launch_finalize<OP_FUSED>(p, f, (hipStream_t)stream);
}

int lq_penalty_maxbin_fwd(const float* P, const float* s, float* mb, uint32_t* ties, float* term, void* ws,
                          size_t ws_bytes, int64_t outer, int64_t G, int64_t inner, void* stream) {
    int rc = check_desc(outer, G, inner);
    if (rc) return rc;
    LQ_REQUIRE_PTR(P);
    LQ_REQUIRE_PTR(s);
    LQ_REQUIRE_PTR(mb);
    LQ_REQUIRE_PTR(ties);
    LQ_REQUIRE_PTR(term);
    Plan pl = make_plan(outer, G, inner, kBlock);      // weight-sized operation: 256-thread units at every size (kStreamOp)
    Params p = base_params(P, s, outer, G, inner);
    if ((rc = bind_ws(p, pl, ws, ws_bytes))) return rc;
    if ((rc = launch_traverse<OP_MAXBIN_FWD>(pl, p, (hipStream_t)stream))) return rc;
    FinGeom f = group_geom(pl, outer, G, inner);
    f.o0 = mb;
    f.o2 = ties;
    if ((rc = launch_finalize<OP_MAXBIN_FWD>(p, f, (hipStream_t)stream))) return rc;
    hipLaunchKernelGGL((k_vec_mean<0>), dim3(1), dim3(kBlock), 0, (hipStream_t)stream, mb, G, term);   // :110 reduce_mean(maxbin)
    return check_hip("maxbin mean launch");
}

int lq_penalty_maxbin_bwd(const float* P, const float* s, const float* mb, const uint32_t* ties, const float* c_dev,
                          float c_scale, float* dP, float* ds, int64_t outer, int64_t G, int64_t inner, void* stream) {
    int rc = check_desc(outer, G, inner);
    if (rc) return rc;
    LQ_REQUIRE_PTR(P);
    LQ_REQUIRE_PTR(s);
    LQ_REQUIRE_PTR(mb);
    LQ_REQUIRE_PTR(ties);
    LQ_REQUIRE_PTR(c_dev);
    LQ_REQUIRE_PTR(dP);
    LQ_REQUIRE_PTR(ds);
    Plan pl = make_plan(outer, G, inner, kBlock);      // weight-sized operation: 256-thread units at every size (kStreamOp)
    Params p = base_params(P, s, outer, G, inner);
    p.mb = mb;
    p.ties = ties;
    p.c_dev = c_dev;
    p.c_scale = c_scale;
    p.out = dP;
    if ((rc = launch_traverse<OP_MAXBIN_BWD>(pl, p, (hipStream_t)stream))) return rc;
    hipLaunchKernelGGL(k_maxbin_ds, dim3((unsigned)ceil_div(G, kBlock)), dim3(kBlock), 0, (hipStream_t)stream, s, mb, c_dev, c_scale, ds, G);
    return check_hip("maxbin ds launch");
}

int lq_penalty_difference_fwd(const float* P, const float* s, float* term, void* ws, size_t ws_bytes, int64_t outer,
                              int64_t G, int64_t inner, void* stream) {
    int rc = check_desc(outer, G, inner);
    if (rc) return rc;
    LQ_REQUIRE_PTR(P);
    LQ_REQUIRE_PTR(s);
    LQ_REQUIRE_PTR(term);
    Plan pl = make_plan(outer, G, inner, kBlock);      // weight-sized operation: 256-thread units at every size (kStreamOp)
    Params p = base_params(P, s, outer, G, inner);
    if ((rc = bind_ws(p, pl, ws, ws_bytes))) return rc;
    if ((rc = launch_traverse<OP_DIFF_FWD>(pl, p, (hipStream_t)stream))) return rc;
    FinGeom f = global_geom(pl, outer, G, inner);
    f.o0 = term;
    return launch_finalize<OP_DIFF_FWD, false>(p, f, (hipStream_t)stream);
}

int lq_penalty_difference_bwd(const float* P, const float* s, const float* c_dev, float c_scale, float* dP, float* ds,
                              void* ws, size_t ws_bytes, int64_t outer, int64_t G, int64_t inner, void* stream) {
    int rc = check_desc(outer, G, inner);
    if (rc) return rc;
    LQ_REQUIRE_PTR(P);
    LQ_REQUIRE_PTR(s);
    LQ_REQUIRE_PTR(c_dev);
    LQ_REQUIRE_PTR(dP);
    LQ_REQUIRE_PTR(ds);
    Plan pl = make_plan(outer, G, inner, kBlock);      // weight-sized operation: 256-thread units at every size (kStreamOp)
    Params p = base_params(P, s, outer, G, inner);
    p.c_dev = c_dev;
    p.c_scale = c_scale;
    p.out = dP;
    if ((rc = bind_ws(p, pl, ws, ws_bytes))) return rc;
    if ((rc = launch_traverse<OP_DIFF_BWD>(pl, p, (hipStream_t)stream))) return rc;
    FinGeom f = group_geom(pl, outer, G, inner);
    f.o0 = ds;
    return launch_finalize<OP_DIFF_BWD>(p, f, (hipStream_t)stream);
}

int lq_penalty_inverse_fwd(const float* s, float* term, int64_t G, void* stream) {
    if (G <= 0) return fail(LQ_EINVAL, "lq_penalty_inverse_fwd: G must be positive");
    LQ_REQUIRE_PTR(s);
    LQ_REQUIRE_PTR(term);
    hipLaunchKernelGGL((k_vec_mean<1>), dim3(1), dim3(kBlock), 0, (hipStream_t)stream, s, G, term);
    return check_hip("inverse fwd launch");
}

int lq_penalty_inverse_bwd(const float* s, const float* c_dev, float c_scale, float* ds, int64_t G, void* stream) {
    if (G <= 0) return fail(LQ_EINVAL, "lq_penalty_inverse_bwd: G must be positive");
    LQ_REQUIRE_PTR(s);
    LQ_REQUIRE_PTR(c_dev);
    LQ_REQUIRE_PTR(ds);
    hipLaunchKernelGGL(k_inverse_bwd, dim3((unsigned)ceil_div(G, kBlock)), dim3(kBlock), 0, (hipStream_t)stream, s, c_dev, c_scale, ds, G);
    return check_hip("inverse bwd launch");
}

int lq_scale_adam_step(float* s, const float* ds, float* m, float* v, int64_t n, double lr, double beta1, double beta2,
                       double eps, int64_t step, float min_value, int mode, void* stream) {
    if (n <= 0) return fail(LQ_EINVAL, "lq_scale_adam_step: n must be positive");
    if (step < 1) return fail(LQ_EINVAL, "lq_scale_adam_step: step is 1-based");
    if (mode != LQ_ADAM_KERAS && mode != LQ_ADAM_TORCH) return fail(LQ_EINVAL, "lq_scale_adam_step: bad mode %d", mode);
    LQ_REQUIRE_PTR(s);
    LQ_REQUIRE_PTR(ds);
    LQ_REQUIRE_PTR(m);
    LQ_REQUIRE_PTR(v);
    // the bias corrections are formed on the device from `step` with the arithmetic of lq_scale_adam_step_dev (Keras 2.11: beta
    // powers and alpha in fp32, tf.pow on the cast hyper-parameter; torch: python doubles), so that a step replayed from a
    // hipGraph (device-side counter) and an eager step give the same bits -- the thresholded scale gradient amplifies a
    // last-bit difference of a scale into visibly different trajectories
    hipLaunchKernelGGL(k_adam_dev, dim3((unsigned)ceil_div(n, kBlock)), dim3(kBlock), 0, (hipStream_t)stream, s, ds, m, v, n,
                       (float)lr, (float)beta1, (float)beta2, lr, beta1, beta2, (float)(1.0 - beta1), (float)(1.0 - beta2),
                       (float)eps, (const int64_t*)nullptr, step, min_value, mode);
    return check_hip("adam launch");
}

int lq_scale_adam_step_dev(float* s, const float* ds, float* m, float* v, int64_t n, double lr, double beta1, double beta2,
                           double eps, const int64_t* step_dev, float min_value, int mode, void* stream) {
    if (n <= 0) return fail(LQ_EINVAL, "lq_scale_adam_step_dev: n must be positive");
    if (mode != LQ_ADAM_KERAS && mode != LQ_ADAM_TORCH) return fail(LQ_EINVAL, "lq_scale_adam_step_dev: bad mode %d", mode);
    LQ_REQUIRE_PTR(s);
    LQ_REQUIRE_PTR(ds);
    LQ_REQUIRE_PTR(m);
    LQ_REQUIRE_PTR(v);
    if (!step_dev || !aligned(step_dev, 8)) return fail(LQ_EINVAL, "lq_scale_adam_step_dev: step_dev must be an 8-byte aligned device pointer");
    hipLaunchKernelGGL(k_adam_dev, dim3((unsigned)ceil_div(n, kBlock)), dim3(kBlock), 0, (hipStream_t)stream, s, ds, m, v, n,
                       (float)lr, (float)beta1, (float)beta2, lr, beta1, beta2, (float)(1.0 - beta1), (float)(1.0 - beta2),
                       (float)eps, step_dev, (int64_t)0, min_value, mode);
    return check_hip("adam (device step) launch");
}

int lq_min_value_project(float* w, int64_t n, float min_value, void* stream) {
    if (n <= 0) return fail(LQ_EINVAL, "lq_min_value_project: n must be positive");
    LQ_REQUIRE_PTR(w);
    hipLaunchKernelGGL(k_min_project, dim3((unsigned)ceil_div(n, kBlock)), dim3(kBlock), 0, (hipStream_t)stream, w, n, min_value);
    return check_hip("min project launch");
}

int lq_q_absmax_over_axis(const float* P, const float* s, float* result, int64_t pre, int64_t n_axis, int64_t post,
                          int64_t outer, int64_t G, int64_t inner, void* stream) {
    int rc = check_desc(outer, G, inner);
    if (rc) return rc;
    if (pre <= 0 || n_axis <= 0 || post <= 0 || pre * n_axis * post != outer * G * inner)
        return fail(LQ_EINVAL, "lq_q_absmax_over_axis: (pre,n_axis,post) does not match the tensor");
    LQ_REQUIRE_PTR(P);
    LQ_REQUIRE_PTR(s);
    LQ_REQUIRE_PTR(result);
    const int64_t n = pre * n_axis * post;
    hipStream_t st = (hipStream_t)stream;
    if (hipMemsetAsync(result, 0, (size_t)(pre * post) * sizeof(float), st) != hipSuccess) return check_hip("absmax axis memset");
    const int64_t blocks = ceil_div(n, (int64_t)kAbsChunk);
    if (blocks > 2147483647ll) return fail(LQ_EINVAL, "lq_q_absmax_over_axis: tensor too large");
    // outputs one chunk can touch: (chunks of slices it overlaps) * post entries, starting at its first slice
    const int64_t slice = n_axis * post;
    int64_t span = (ceil_div((int64_t)kAbsChunk, slice) + 1) * post;
    if (span > pre * post) span = pre * post;
    const int use_lds = span <= kAbsTab ? 1 : 0;
    if (n < 4294967296ll)
        hipLaunchKernelGGL(k_q_absmax_axis<uint32_t>, dim3((unsigned)blocks), dim3(kBlock), 0, st, P, s, reinterpret_cast<uint32_t*>(result),
                           (uint32_t)n, (uint32_t)n_axis, (uint32_t)post, (uint32_t)G, (uint32_t)inner, use_lds, (int)(use_lds ? span : 0));
    else
        hipLaunchKernelGGL(k_q_absmax_axis<int64_t>, dim3((unsigned)blocks), dim3(kBlock), 0, st, P, s, reinterpret_cast<uint32_t*>(result),
                           n, n_axis, post, G, inner, use_lds, (int)(use_lds ? span : 0));
    return check_hip("absmax axis launch");
}

}  // extern "C"

// ------------------------------------------------------------------------------------------
//  lq_batch: host object + C ABI
// ------------------------------------------------------------------------------------------
struct lq_task_table {                    // one device-resident task table
    std::vector<lq::Task> h;
    std::vector<int> index;               // table position -> descriptor index (tasks are ordered by work per block)
    lq::Task* d = nullptr;
    uint32_t* prefix_d = nullptr;         // [n] first block of every task, then [n] first group
    uint32_t* block_task_d = nullptr;     // [blocks] task of every traversal block (a DWORD each: one scalar load; sub-dword entries are fetched with vector loads)
    lq::FinRec* fin_blocks_d = nullptr;   // [fin_blocks] one self-contained record per finalize block (scale-gradient tables)
    std::vector<lq::FinRec> fin_h;        // host copy: the Adam-state pointers are filled in by lq_batch_create before the upload
    uint32_t fin_blocks = 0;
    uint32_t blocks = 0, groups = 0;
    bool has_tile = false;                // some task runs the conv tile (needs the tile kernel's LDS)
    int64_t ws_words = 0;
};

struct lq_batch {
    int n = 0;
    lq_task_table fwd, bwd, bwd_o, pen;   // bwd: HWIO gradients (generic traversals); bwd_o: OIHW gradients of the conv kernels (tiles);
                                          // pen: every tensor, with workspace slices and mb/ties buffers (penalty passes)
    std::vector<lq::AdamTask> adam_h;
    float* mb_d = nullptr;
    uint32_t* ties_d = nullptr;
    lq::AdamTask* adam_d = nullptr;
    size_t ws_bytes = 256;
    bool has_perm = false;               // some conv kernel has an OIHW companion
};

namespace lq {

// Conv-tile geometry of an HWIO kernel (lq_conv_tile.hpp), or false when the tensor keeps the generic traversal: kernels
// with more than 9 taps, co % 4 != 0, or a descriptor whose groups are not a function of (h, c) the tile knows.
static bool make_conv_tile(ConvTile& ct, int64_t hw, int64_t ci, int64_t co, int64_t outer, int64_t G, int64_t inner) {
    memset(&ct, 0, sizeof(ct));
    if (hw < 1 || hw > kCtPass || co % 4 != 0 || ci < 1) return false;
    if ((double)hw * (double)ci * (double)co >= 4294967296.0) return false;
    int kind;
    int64_t A = 1;
    if (G == 1) {
        kind = 1;
        A = hw;
    } else if (inner == co && G == ci) {
        kind = 0;
    } else if (inner % (ci * co) == 0 && G <= 255) {
        kind = 1;
        A = inner / (ci * co);
        if (outer * G * A != hw) return false;
    } else {
        return false;
    }
    int64_t m = kCtPass / hw;
    const int64_t mc = ceil_div(ci, 32);
    if (m > mc) m = mc;
    if (m < 1) m = 1;
    ct.hw = (uint32_t)hw;
    ct.ci = (uint32_t)ci;
    ct.co = (uint32_t)co;
    ct.tc = (uint32_t)(32 * m);
    ct.ntc = (uint32_t)ceil_div(ci, 32 * m);
    ct.nto = (uint32_t)ceil_div(co, kCtO);
    if ((double)ct.ntc * (double)ct.nto > 2147483647.0) return false;
    ct.npass = (uint32_t)(m * hw);
    ct.kind = (uint32_t)kind;
    ct.fnto = make_fastdiv(ct.nto);
    // pass order: kind 0 by channel block, kind 1 group by group (one accumulator flush per group)
    struct PassKey { int g, j, h; };
    std::vector<PassKey> ps;
    for (int64_t j = 0; j < m; ++j)
        for (int64_t h = 0; h < hw; ++h) ps.push_back({kind == 0 ? (int)j : (int)((h / A) % G), (int)j, (int)h});
    std::stable_sort(ps.begin(), ps.end(), [](const PassKey& a, const PassKey& b) { return a.g < b.g; });
    for (size_t q = 0; q < ps.size(); ++q) {
        const uint32_t is_new = (q > 0 && ps[q].g != ps[q - 1].g) ? 1u : 0u;
        ct.pass_info[q] = (uint32_t)(8 * ps[q].j) | ((uint32_t)ps[q].h << 8) | ((uint32_t)(kind == 0 ? 0 : ps[q].g) << 16) | (is_new << 24);
        ct.pass_off[q] = (uint32_t)(((int64_t)ps[q].h * ci + 8 * ps[q].j) * co);
    }
    return true;
}

static int64_t conv_tile_partials(const ConvTile& ct, int64_t G) {
    return ct.kind == 0 ? (int64_t)ct.ci * ct.nto : G * (int64_t)ct.nto * ct.ntc * kWavesPerBlock;
}

// finalize geometry of the tile's partials (see ct_flush)
static void conv_tile_fin(const ConvTile& ct, int64_t G, int64_t& gstride, int64_t& n1, int64_t& stride1, int64_t& n2) {
    n1 = 1;
    stride1 = 0;
    n2 = ct.kind == 0 ? (int64_t)ct.nto : (int64_t)ct.nto * ct.ntc * kWavesPerBlock;
    gstride = n2;
    (void)G;
}

// Row blocks of the batch's scale-gradient column tiles are sized by the BATCH, not per tensor: make_plan aims at about 512 blocks
// per tensor (right for a launch of its own); 40 or 108 tensors in one launch then make 2000-5300 blocks of 2-8 K elements, more than
// the chip holds at once -- the second round of blocks ran on a nearly idle chip (profiles/r03/timelines/) -- and 4-8 times the
// partials the launch needs.  Every float4 column tile of the scale-gradient table gets about `w` elements per block instead, with
// `w` chosen so that the whole launch is one resident round.
static double batch_block_elements(double total_elements) {
    LQ_KNOB(w, "LQ_TUNE_BATCH_W", 0);                  // development: force the elements per block
    if (w > 0) return (double)w;
    LQ_KNOB(nb, "LQ_TUNE_BATCH_NB", 1280);             // blocks the chip holds at once: 256 CUs x 5 (k_batch_traverse<OP_BWD>: 5 waves per SIMD)
    const double per = total_elements / (double)nb;
    return per < 8192.0 ? 8192.0 : per;
}

static void batch_rows_per_block(Plan& pl, int64_t outer, double w) {
    const int64_t tilew = pl.C < 256 ? pl.C : 256;
    int64_t rb = ceil_div((int64_t)w, tilew);
    rb = ceil_div(rb, 16) * 16;                        // whole rounds of four waves x four rows
    if (rb > outer) rb = outer;
    const int64_t nby = ceil_div(outer, rb);
    rb = ceil_div(ceil_div(outer, nby), 4) * 4;        // even out the row blocks
    if (rb > outer) rb = outer;
    pl.rps = rb;
    pl.ysplit = ceil_div(outer, rb);
}

// Narrow column matrices (C <= 64: row-wise / column-wise scales of a conv kernel stored OIHW are [co ci][kh kw] and [co ci kh][kw],
// groups of period 9 and 3) as a periodic float4 stream (lq_traverse.hpp col_periodic4_body): `nblk` blocks, a multiple of C, so that
// a thread's four columns never change; about `w` elements per block.  The scalar form it replaces in the batch reads 4 bytes per
// lane (col_small_body: 252 bytes per wave and load for C = 9).
static int64_t batch_periodic_blocks(int64_t C, double elements, double w) {
    int64_t k = (int64_t)(elements / w / (double)C + 0.5);
    if (k < 1) k = 1;
    return C * k;
}
constexpr double kBatchPeriodicMin = 32768.0;      // elements from which a C <= 64 task of the batch takes the periodic form

static int fill_task(Task& t, const lq_tensor_desc& d, bool bwd, bool allow_tile, lq_task_table& tb, double batch_w = 0.0) {
    memset(&t, 0, sizeof(t));
    Plan pl = make_plan(d.outer, d.G, d.inner, kBlock);
    t.p = base_params(d.P, d.s, d.outer, d.G, d.inner);
    t.p.out = d.out;
    t.p.dy = d.dy;
    t.p.lam = d.lambda;
    t.p.tmode = (d.lambda < 4.0e-4f) ? 0 : ((d.lambda <= 0.25f) ? 1 : 2);
    bool tile = false;
    if (d.conv_co > 0) {
        t.p.out_perm = bwd ? nullptr : d.out_oihw;
        t.p.perm_hw = (uint32_t)d.conv_hw;
        t.p.perm_ci = (uint32_t)d.conv_ci;
        t.p.perm_co = (uint32_t)d.conv_co;
        t.dp = d.dp;
        tile = allow_tile && aligned(d.P, 16) && aligned(bwd ? (const void*)d.dp : (const void*)d.out, 16) &&
               make_conv_tile(t.ct, d.conv_hw, d.conv_ci, d.conv_co, d.outer, d.G, d.inner);
    }
    t.ds = d.ds;
    t.mode = tile ? MODE_CONV_TILE : pl.mode;
    t.lpr_log2 = pl.lpr_log2;
    t.R = pl.R;
    t.L = pl.L;
    t.nc = pl.nc;
    t.C = pl.C;
    t.nbx = 0;
    if (!tile && pl.mode == MODE_COL) {
        col_variant(pl, d.P, nullptr, bwd ? nullptr : d.out, t.col_variant, t.nbx, false);
        // scale-gradient table of the plain (non-companion) pass: float4 tiles run lq_batch_cols.hpp -- batch-sized row blocks, one
        // partial per (row block, group fragment)
        if (bwd && batch_w > 0.0 && t.col_variant >= 4 && pl.C < (1ll << 30) && d.outer < (1ll << 31)) {
            batch_rows_per_block(pl, d.outer, batch_w);
            LQ_KNOB(frag, "LQ_TUNE_BATCH_FRAG", 1);    // development: 0 = a partial per column (the generic layout)
            const bool grouped = frag && d.inner > 1 && d.inner <= 64;
            t.fg = make_frag_geom(d.G, d.inner, pl.C, grouped);
            if (!grouped) t.fg.F = (uint32_t)pl.C;
            pl.np = pl.ysplit * (int64_t)t.fg.F;
            pl.n1 = pl.ysplit;
            if (grouped) {                             // read by finalize_frag_body through t.fg; kept consistent for the workspace bound
                pl.gstride = 0;
                pl.stride1 = t.fg.F;
                pl.n2 = 1;
            } else {
                t.fg.gpb = 0;
            }
        } else if (!bwd && batch_w > 0.0 && t.col_variant >= 4) {
            // forward tiles keep make_plan's row blocks (16-32 rows: 1968 blocks for the ResNet-18-like set, all resident at 62 VGPRs)
            LQ_KNOB(fw, "LQ_TUNE_BATCH_FWD_W", 0);     // development: elements per block of the forward's tiles
            if (fw > 0) batch_rows_per_block(pl, d.outer, (double)fw);
        } else if (bwd && batch_w > 0.0 && t.col_variant >= 4) {
            // a matrix beyond the 32-bit extents of that form (2^30 columns): the scalar column tile, generic layout
            t.col_variant = 1;
            t.nbx = ceil_div(pl.C, 64);
        } else if (batch_w > 0.0 && t.col_variant == 0 && (double)d.outer * (double)pl.C >= kBatchPeriodicMin && aligned(d.P, 16) &&
                   (bwd || aligned(d.out, 16))) {
            t.col_variant = 6;
            pl.ysplit = batch_periodic_blocks(pl.C, (double)d.outer * (double)pl.C, batch_w);
            pl.rps = pl.ysplit;                        // variant 6: the block count travels in `rps`
            t.nbx = 1;
            pl.np = pl.ysplit * pl.C;                  // one partial per (block, column)
            pl.n1 = pl.ysplit;
        }
    }
    t.rps = pl.rps;
    int64_t blocks;
    if (tile) {
        blocks = (int64_t)t.ct.ntc * t.ct.nto;
        t.vec = 1;
    } else if (pl.mode == MODE_ROW_BIG) {
        // float4 path: forward needs P and out 16-byte aligned; backward needs P (dy is checked at every launch)
        t.vec = (aligned(d.P, 16) && (bwd || aligned(d.out, 16))) ? 1 : 0;
        t.ru = 1;
        if (batch_w > 0.0 && t.vec && d.conv_co == 0) {
            // batch-sized units for long rows (forward and scale-gradient tables): 4096 elements where a row has at least two of them
            const int ru = pl.L >= 8192 ? 4 : (pl.L >= 4096 ? 2 : 1);
            if (ru > 1) {
                t.ru = ru;
                pl.nc = row_chunks(pl.L, (int64_t)kBlock * 4 * ru);
                pl.np = pl.R * pl.nc;
                pl.gstride = pl.nc;                    // partial (row o * G + g, chunk c) at (o * G + g) * nc + c
                pl.stride1 = d.G * pl.nc;
                pl.n2 = pl.nc;
                t.nc = pl.nc;
            }
        }
        blocks = pl.R * pl.nc;
    } else if (pl.mode == MODE_ROW_SMALL) {
        t.vec = row_small_vec(pl, d.P, nullptr, bwd ? nullptr : d.out) ? 1 : 0;
        if (t.vec) t.lpr_log2 = row_small_lpr_log2_vec(pl.L);
        blocks = ceil_div(pl.R, kBlock >> t.lpr_log2);
    } else {
        blocks = t.nbx * pl.ysplit;
    }
    if (blocks <= 0 || blocks > 0x7fffffffll) return fail(LQ_EINVAL, "lq_batch_create: too many blocks");
    t.first_block = (uint32_t)blocks;            // block COUNT until finish_table() turns it into a prefix
    if (bwd) {
        t.first_group = (uint32_t)d.G;           // likewise
        int64_t np = pl.np;
        t.gstride = pl.gstride;
        t.n1 = pl.n1;
        t.stride1 = pl.stride1;
        t.n2 = pl.n2;
        if (tile) {
            np = conv_tile_partials(t.ct, d.G);
            conv_tile_fin(t.ct, d.G, t.gstride, t.n1, t.stride1, t.n2);
        }
        t.np_pad = (np + 63) / 64 * 64;
        t.count = (double)d.outer * (double)d.inner;
    }
    if (tile) tb.has_tile = true;
    return LQ_OK;
}

// Orders the table by decreasing work per block (tiles first), assigns block / group prefixes and workspace slices, uploads.
static int finish_table(lq_task_table& tb, bool bwd) {
    const size_t n = tb.h.size();
    if (!n) return LQ_OK;
    std::vector<size_t> order(n);
    for (size_t i = 0; i < n; ++i) order[i] = i;
    auto weight = [&](size_t i) {
        const Task& t = tb.h[i];
        const double el = (double)t.p.outer * (double)t.p.G * (double)t.p.inner;
        return el / (double)t.first_block;                     // elements per block
    };
    std::stable_sort(order.begin(), order.end(), [&](size_t a, size_t b) { return weight(a) > weight(b); });
    std::vector<Task> h2(n);
    std::vector<int> idx2(n);
    std::vector<uint32_t> prefix(2 * n);
    uint64_t bp = 0, gp = 0;
    int64_t words = 0;
    for (size_t k = 0; k < n; ++k) {
        Task t = tb.h[order[k]];
        const uint32_t nb = t.first_block, ng = bwd ? t.first_group : 0u;
        if (bp + nb > 0xffffffffull || gp + ng > 0xffffffffull) return fail(LQ_EINVAL, "lq_batch_create: too many blocks or groups");
        t.first_block = (uint32_t)bp;
        t.first_group = (uint32_t)gp;
        prefix[k] = (uint32_t)bp;
        prefix[n + k] = (uint32_t)gp;
        bp += nb;
        gp += ng;
        if (bwd) {
            t.ws_off = words;
            words += 4 * t.np_pad;
        }
        h2[k] = t;
        idx2[k] = tb.index[order[k]];
    }
    tb.h.swap(h2);
    tb.index.swap(idx2);
    tb.blocks = (uint32_t)bp;
    tb.groups = (uint32_t)gp;
    tb.ws_words = words;
    std::vector<uint32_t> bt((size_t)bp);
    for (size_t k = 0; k < n; ++k) {
        const uint64_t end = k + 1 < n ? prefix[k + 1] : bp;
        for (uint64_t b = prefix[k]; b < end; ++b) bt[b] = (uint32_t)k;
    }
    hipError_t e = hipMalloc(&tb.prefix_d, 2 * n * sizeof(uint32_t));
    if (e == hipSuccess) e = hipMemcpy(tb.prefix_d, prefix.data(), 2 * n * sizeof(uint32_t), hipMemcpyHostToDevice);
    if (e == hipSuccess) e = hipMalloc(&tb.block_task_d, bt.size() * sizeof(uint32_t));
    if (e == hipSuccess) e = hipMemcpy(tb.block_task_d, bt.data(), bt.size() * sizeof(uint32_t), hipMemcpyHostToDevice);
    if (e == hipSuccess && bwd && gp > 0) {
        // finalize blocks: column traversals (one partial per (row block, column), a group's partials `C` words apart) get a
        // block per floor(64 / inner) groups that reads contiguous runs of the rows of partials; otherwise four groups per block (a wave each)
        // where a group has at most 256 partials, else a block per group
        std::vector<FinRec>& fb = tb.fin_h;
        fb.clear();
        for (size_t k = 0; k < n; ++k) {
            const Task& t = tb.h[k];
            const bool frag = t.fg.gpb != 0;                                                  // lq_batch_cols.hpp: fragments of fg.gpb groups per block
            const bool cols = !frag && finalize_cols_ok(t.p.G, t.gstride, t.n1, t.stride1, t.n2);      // the rule of the single-tensor finalize
            const bool wide = !frag && !cols && t.n1 * t.n2 > 256;
            const int64_t per = frag ? (int64_t)t.fg.gpb : (cols ? 64 / t.n2 : (wide ? 1 : 4));
            FinRec r;
            memset(&r, 0, sizeof(r));
            r.form = frag ? 3u : (cols ? 2u : (wide ? 1u : 0u));
            r.G = (uint32_t)t.p.G;
            r.lam = t.p.lam;
            r.ws_off = t.ws_off;
            r.np_pad = t.np_pad;
            r.gstride = t.gstride;
            r.n1 = t.n1;
            r.stride1 = t.stride1;
            r.n2 = t.n2;
            r.fg = t.fg;
            r.count = t.count;
            r.ds = t.ds;
            r.s = const_cast<float*>(t.p.s);
            r.pad = (uint32_t)k;                       // table position of the task: fill_fin_state() finds its Adam state through it
            for (int64_t g = 0; g < t.p.G; g += per) {
                r.g0 = (uint32_t)g;
                fb.push_back(r);
            }
        }
        if (fb.size() > 0x7fffffffull) return fail(LQ_EINVAL, "lq_batch_create: too many groups");
        tb.fin_blocks = (uint32_t)fb.size();
        e = hipMalloc(&tb.fin_blocks_d, fb.size() * sizeof(FinRec));
    }
    if (e == hipSuccess) e = hipMalloc(&tb.d, n * sizeof(Task));
    if (e != hipSuccess) return fail(LQ_EHIP, "lq_batch_create: %s", hipGetErrorString(e));
    return LQ_OK;
}

static int upload_table(lq_task_table& tb) {
    if (tb.h.empty()) return LQ_OK;
    hipError_t e = hipMemcpy(tb.d, tb.h.data(), tb.h.size() * sizeof(Task), hipMemcpyHostToDevice);
    if (e == hipSuccess && !tb.fin_h.empty()) {
        for (FinRec& r : tb.fin_h) {                   // Adam state of the task (set after finish_table ordered the tasks)
            const Task& t = tb.h[r.pad];
            r.am = t.am;
            r.av = t.av;
            r.amin = t.amin;
        }
        e = hipMemcpy(tb.fin_blocks_d, tb.fin_h.data(), tb.fin_h.size() * sizeof(FinRec), hipMemcpyHostToDevice);
    }
    if (e != hipSuccess) return fail(LQ_EHIP, "lq_batch_create: %s", hipGetErrorString(e));
    return LQ_OK;
}

static void free_table(lq_task_table& tb) {
    if (tb.d) (void)hipFree(tb.d);
    if (tb.prefix_d) (void)hipFree(tb.prefix_d);
    if (tb.block_task_d) (void)hipFree(tb.block_task_d);
    if (tb.fin_blocks_d) (void)hipFree(tb.fin_blocks_d);
    tb.block_task_d = nullptr;
    tb.fin_blocks_d = nullptr;
    tb.d = nullptr;
    tb.prefix_d = nullptr;
}

}  // namespace lq

extern "C" {

int lq_batch_create(const lq_tensor_desc* descs, int n, lq_batch** out) {
    if (!descs || !out) return fail(LQ_EINVAL, "lq_batch_create: NULL argument");
    if (n <= 0 || n > kBatchMax) return fail(LQ_EINVAL, "lq_batch_create: n must be in 1..%d", kBatchMax);
    lq_batch* b = new (std::nothrow) lq_batch();
    if (!b) return fail(LQ_EHIP, "lq_batch_create: out of host memory");
    b->n = n;
    int rc = LQ_OK;
    double bwd_elements = 0.0;             // elements the scale-gradient pass traverses: sizes its row blocks (batch_block_elements)
    for (int i = 0; i < n; ++i)
        if (descs[i].lambda == descs[i].lambda && descs[i].outer > 0 && descs[i].G > 0 && descs[i].inner > 0)
            bwd_elements += (double)descs[i].outer * (double)descs[i].G * (double)descs[i].inner;
    const double batch_w = batch_block_elements(bwd_elements);
    // the forward table keeps make_plan's float4 tiles (its blocks are all resident: 61 VGPRs); `b_fwd_w` only sizes the periodic
    // streams of narrow column matrices -- and only where no OIHW companion is emitted (OP_FWD; OP_FWD_PERM has no float4 element path)
    bool any_perm = false;
    for (int i = 0; i < n; ++i) any_perm = any_perm || descs[i].conv_co > 0;
    const double b_fwd_w = any_perm ? 0.0 : 8192.0;
    for (int i = 0; i < n && !rc; ++i) {
        const lq_tensor_desc& d = descs[i];
        rc = check_desc(d.outer, d.G, d.inner);
        // `out` (HWIO) is optional for a conv kernel whose OIHW companion is the only consumer, where the LDS tile serves it
        const bool out_optional = !rc && d.conv_co > 0 && d.out_oihw && aligned(d.P, 16) &&
                                  lq_conv_tile_supported(d.conv_hw, d.conv_ci, d.conv_co, d.outer, d.G, d.inner);
        if (!rc && (!d.P || !d.s || (!d.out && !out_optional)))
            rc = fail(LQ_EINVAL, "lq_batch_create: tensor %d has a NULL P/s/out (out may be NULL only next to an out_oihw the LDS-tile path writes)", i);
        if (!rc && (!aligned(d.P, 4) || !aligned(d.s, 4) || !aligned(d.out, 4))) rc = fail(LQ_EALIGN, "lq_batch_create: tensor %d misaligned", i);
        if (!rc && d.conv_co > 0) {
            rc = check_conv("lq_batch_create", d.conv_hw, d.conv_ci, d.conv_co, d.outer, d.G, d.inner);
            if (!rc && (!d.out_oihw || !d.dp || !aligned(d.out_oihw, 4) || !aligned(d.dp, 4)))
                rc = fail(LQ_EINVAL, "lq_batch_create: tensor %d has conv extents but no out_oihw / dp buffer", i);
            if (!rc) b->has_perm = true;
        }
        Task t;
        if (!rc) rc = fill_task(t, d, false, true, b->fwd, b_fwd_w);
        if (rc) break;
        b->fwd.h.push_back(t);
        b->fwd.index.push_back(i);
        if (d.ds) {                   // penalty passes need a scale-gradient destination
            Task tp;
            if ((rc = fill_task(tp, d, true, false, b->pen))) break;
            b->pen.h.push_back(tp);
            b->pen.index.push_back(i);
        }
        if (d.lambda == d.lambda) {   // not NaN: nested-quantization tensor with a scale gradient
            if (!d.ds) {
                rc = fail(LQ_EINVAL, "lq_batch_create: tensor %d has lambda but no ds", i);
                break;
            }
            Task tb;
            if ((rc = fill_task(tb, d, true, false, b->bwd, batch_w))) break;
            if (d.m && d.v) {           // the finalize can apply the scale's Adam step itself (lq_batch_scale_grad_step)
                tb.am = d.m;
                tb.av = d.v;
                tb.amin = d.min_value;
            }
            b->bwd.h.push_back(tb);
            b->bwd.index.push_back(i);
            if ((rc = fill_task(tb, d, true, true, b->bwd_o))) break;
            if (d.m && d.v) {
                tb.am = d.m;
                tb.av = d.v;
                tb.amin = d.min_value;
            }
            b->bwd_o.h.push_back(tb);
            b->bwd_o.index.push_back(i);
        }
        if (d.m && d.v) {
            AdamTask a;
            memset(&a, 0, sizeof(a));
            a.s = const_cast<float*>(d.s);
            a.ds = d.ds;
            a.m = d.m;
            a.v = d.v;
            a.n = d.G;
            a.min_value = d.min_value;
            if (!d.ds) {
                rc = fail(LQ_EINVAL, "lq_batch_create: tensor %d has Adam state but no ds", i);
                break;
            }
            b->adam_h.push_back(a);
        }
    }
    if (!rc) rc = finish_table(b->fwd, false);
    if (!rc) rc = finish_table(b->bwd, true);
    if (!rc) rc = finish_table(b->bwd_o, true);
    if (!rc) rc = finish_table(b->pen, true);
    if (!rc && !b->pen.h.empty()) {
        hipError_t e = hipMalloc(&b->mb_d, (size_t)b->pen.groups * sizeof(float));
        if (e == hipSuccess) e = hipMalloc(&b->ties_d, (size_t)b->pen.groups * sizeof(uint32_t));
        if (e != hipSuccess) rc = fail(LQ_EHIP, "lq_batch_create: %s", hipGetErrorString(e));
        for (auto& tp : b->pen.h) {
            tp.mb = b->mb_d + tp.first_group;
            tp.ties = b->ties_d + tp.first_group;
        }
    }
    if (!rc) rc = upload_table(b->fwd);
    if (!rc) rc = upload_table(b->bwd);
    if (!rc) rc = upload_table(b->bwd_o);
    if (!rc) rc = upload_table(b->pen);
    if (!rc && !b->adam_h.empty()) {
        hipError_t e = hipMalloc(&b->adam_d, b->adam_h.size() * sizeof(AdamTask));
        if (e == hipSuccess) e = hipMemcpy(b->adam_d, b->adam_h.data(), b->adam_h.size() * sizeof(AdamTask), hipMemcpyHostToDevice);
        if (e != hipSuccess) rc = fail(LQ_EHIP, "lq_batch_create: %s", hipGetErrorString(e));
    }
    if (rc) {
        char keep[sizeof(g_err)];
        memcpy(keep, g_err, sizeof(keep));
        lq_batch_destroy(b);
        memcpy(g_err, keep, sizeof(keep));
        return rc;
    }
    int64_t words = b->bwd.ws_words > b->pen.ws_words ? b->bwd.ws_words : b->pen.ws_words;
    if (b->bwd_o.ws_words > words) words = b->bwd_o.ws_words;
    b->ws_bytes = (size_t)words * 4 + 256;
    *out = b;
    return LQ_OK;
}

int lq_batch_destroy(lq_batch* b) {
    if (!b) return LQ_OK;
    free_table(b->fwd);
    free_table(b->bwd);
    free_table(b->bwd_o);
    free_table(b->pen);
    if (b->adam_d) (void)hipFree(b->adam_d);
    if (b->mb_d) (void)hipFree(b->mb_d);
    if (b->ties_d) (void)hipFree(b->ties_d);
    delete b;
    return LQ_OK;
}

size_t lq_batch_workspace_bytes(const lq_batch* b) { return b ? b->ws_bytes : 0; }

int lq_batch_forward(const lq_batch* b, void* stream) {
    if (!b) return fail(LQ_EINVAL, "lq_batch_forward: NULL batch");
    PtrPack pk;
    CoefPack cf;
    const lq_task_table& tb = b->fwd;
    const int nt = (int)tb.h.size();
    if (b->has_perm)
        hipLaunchKernelGGL((k_batch_traverse<OP_FWD_PERM>), dim3(tb.blocks), dim3(kBlock), 0, (hipStream_t)stream, tb.d, tb.block_task_d, nt,
                           (uint32_t*)nullptr, pk, 0, cf);
    else
        hipLaunchKernelGGL((k_batch_traverse<OP_FWD>), dim3(tb.blocks), dim3(kBlock), 0, (hipStream_t)stream, tb.d, tb.block_task_d, nt,
                           (uint32_t*)nullptr, pk, 0, cf);
    return check_hip("batch forward launch");
}

}  // extern "C"
extern "C" {

static int batch_scale_grad(const lq_batch* b, const float* const* dy, void* ws, size_t ws_bytes, void* stream, bool oihw, const AdamHyper& ah);

static AdamHyper adam_hyper(double lr, double beta1, double beta2, double eps, int64_t step, const int64_t* step_dev, int mode, int on) {
    AdamHyper h;
    memset(&h, 0, sizeof(h));
    h.lr_d = lr;
    h.b1_d = beta1;
    h.b2_d = beta2;
    h.step_dev = step_dev;
    h.step_host = step;
    h.lr = (float)lr;
    h.b1 = (float)beta1;
    h.b2 = (float)beta2;
    h.f0 = (float)(1.0 - beta1);
    h.f1 = (float)(1.0 - beta2);
    h.eps = (float)eps;
    h.mode = mode;
    h.on = on;
    return h;
}

int lq_batch_scale_grad(const lq_batch* b, const float* const* dy, void* ws, size_t ws_bytes, void* stream) {
    return batch_scale_grad(b, dy, ws, ws_bytes, stream, false, adam_hyper(0, 0, 0, 0, 1, nullptr, LQ_ADAM_KERAS, 0));
}

int lq_batch_scale_grad_oihw(const lq_batch* b, const float* const* dy, void* ws, size_t ws_bytes, void* stream) {
    return batch_scale_grad(b, dy, ws, ws_bytes, stream, b && b->has_perm, adam_hyper(0, 0, 0, 0, 1, nullptr, LQ_ADAM_KERAS, 0));
}

int lq_batch_scale_grad_step(const lq_batch* b, const float* const* dy, int dy_oihw, void* ws, size_t ws_bytes, double lr, double beta1,
                             double beta2, double eps, int64_t step, const int64_t* step_dev, int mode, void* stream) {
    if (!b) return fail(LQ_EINVAL, "lq_batch_scale_grad_step: NULL batch");
    if (!step_dev && step < 1) return fail(LQ_EINVAL, "lq_batch_scale_grad_step: step is 1-based");
    if (mode != LQ_ADAM_KERAS && mode != LQ_ADAM_TORCH) return fail(LQ_EINVAL, "lq_batch_scale_grad_step: bad mode %d", mode);
    // every scale the separate Adam launch would update must be one whose gradient this pass computes
    size_t with_state = 0;
    for (const lq::Task& t : b->bwd.h) with_state += t.am ? 1 : 0;
    if (with_state != b->adam_h.size() || with_state != b->bwd.h.size())
        return fail(LQ_EINVAL, "lq_batch_scale_grad_step: every tensor of the batch needs lambda, ds and Adam state (%zu of %zu have them)",
                    with_state, b->adam_h.size() > b->bwd.h.size() ? b->adam_h.size() : b->bwd.h.size());
    return batch_scale_grad(b, dy, ws, ws_bytes, stream, dy_oihw && b->has_perm, adam_hyper(lr, beta1, beta2, eps, step, step_dev, mode, 1));
}

}  // extern "C"

static int batch_scale_grad(const lq_batch* b, const float* const* dy, void* ws, size_t ws_bytes, void* stream, bool oihw, const AdamHyper& ah) {
    if (!b) return fail(LQ_EINVAL, "lq_batch_scale_grad: NULL batch");
    const lq_task_table& tb = oihw ? b->bwd_o : b->bwd;
    if (tb.h.empty()) return LQ_OK;
    if (!ws) return fail(LQ_EWORKSPACE, "lq_batch_scale_grad: workspace is NULL (need %zu bytes)", b->ws_bytes);
    if (!aligned(ws, 16)) return fail(LQ_EALIGN, "lq_batch_scale_grad: workspace must be 16-byte aligned");
    if (ws_bytes < b->ws_bytes) return fail(LQ_EWORKSPACE, "lq_batch_scale_grad: workspace too small: %zu < %zu bytes", ws_bytes, b->ws_bytes);
    PtrPack pk;
    memset(&pk, 0, sizeof(pk));
    bool all_aligned = true;
    for (size_t i = 0; i < tb.h.size(); ++i) {
        const lq::Task& t = tb.h[i];
        const float* d = dy ? dy[tb.index[i]] : t.p.dy;
        if (!d) return fail(LQ_EINVAL, "lq_batch_scale_grad: no upstream gradient for tensor %d", tb.index[i]);
        if (!aligned(d, 4)) return fail(LQ_EALIGN, "lq_batch_scale_grad: dy of tensor %d misaligned", tb.index[i]);
        const bool gathered = oihw && t.p.perm_co != 0;      // read through the permutation (LDS tile or element-wise): no vector loads
        const bool wants16 = t.mode == MODE_CONV_TILE || (t.mode != MODE_COL && t.vec) || (t.mode == MODE_COL && t.col_variant >= 4);
        if (!gathered && wants16 && !aligned(d, 16)) all_aligned = false;
        pk.dy[i] = d;
    }
    if (!all_aligned) return fail(LQ_EALIGN, "lq_batch_scale_grad: a 16-byte aligned tensor got a dy that is not 16-byte aligned");
    CoefPack cf;
    const int nt = (int)tb.h.size();
    if (oihw)
        hipLaunchKernelGGL((k_batch_traverse<OP_BWD_PERM>), dim3(tb.blocks), dim3(kBlock), 0, (hipStream_t)stream, tb.d, tb.block_task_d, nt,
                           (uint32_t*)ws, pk, oihw ? 3 : 1, cf);
    else
        hipLaunchKernelGGL((k_batch_traverse<OP_BWD>), dim3(tb.blocks), dim3(kBlock), 0, (hipStream_t)stream, tb.d, tb.block_task_d, nt,
                           (uint32_t*)ws, pk, 1, cf);
    int rc = check_hip("batch scale-grad launch");
    if (rc) return rc;
    hipLaunchKernelGGL((k_batch_finalize_t<OP_BWD>), dim3(tb.fin_blocks), dim3(256), 0, (hipStream_t)stream, tb.fin_blocks_d, (uint32_t*)ws, ah);
    return check_hip("batch finalize launch");
}

extern "C" {

int lq_batch_penalty_grads(const lq_batch* b, int kind, const float* coeff, float* const* grad, void* ws, size_t ws_bytes,
                           void* stream) {
    if (!b) return fail(LQ_EINVAL, "lq_batch_penalty_grads: NULL batch");
    const int accum = (kind & LQ_PENALTY_ACCUMULATE_DS) ? 1 : 0;
    kind &= ~LQ_PENALTY_ACCUMULATE_DS;
    if (kind < LQ_PENALTY_MAXBIN || kind > LQ_PENALTY_INVERSE) return fail(LQ_EINVAL, "lq_batch_penalty_grads: bad kind %d", kind);
    if (!coeff) return fail(LQ_EINVAL, "lq_batch_penalty_grads: coeff is NULL");
    const lq_task_table& tb = b->pen;
    if (tb.h.size() != (size_t)b->n) return fail(LQ_EINVAL, "lq_batch_penalty_grads: every tensor of the batch needs a ds buffer");
    CoefPack cf;
    PtrPack pk;
    memset(&pk, 0, sizeof(pk));
    memset(&cf, 0, sizeof(cf));
    const int nt = (int)tb.h.size();
    for (int i = 0; i < nt; ++i) cf.c[i] = coeff[tb.index[i]];
    hipStream_t st = (hipStream_t)stream;
    if (kind != LQ_PENALTY_INVERSE) {
        if (!grad) return fail(LQ_EINVAL, "lq_batch_penalty_grads: grad pointers are NULL");
        if (!ws) return fail(LQ_EWORKSPACE, "lq_batch_penalty_grads: workspace is NULL (need %zu bytes)", b->ws_bytes);
        if (!aligned(ws, 16)) return fail(LQ_EALIGN, "lq_batch_penalty_grads: workspace must be 16-byte aligned");
        if (ws_bytes < b->ws_bytes) return fail(LQ_EWORKSPACE, "lq_batch_penalty_grads: workspace too small: %zu < %zu bytes", ws_bytes, b->ws_bytes);
        for (int i = 0; i < nt; ++i) {
            float* gi = grad[tb.index[i]];
            if (!gi || !aligned(gi, 4)) return fail(LQ_EINVAL, "lq_batch_penalty_grads: gradient buffer of tensor %d missing", tb.index[i]);
            const Task& t = tb.h[i];
            const bool wants16 = (t.mode != MODE_COL && t.vec) || (t.mode == MODE_COL && t.col_variant >= 4);
            if (wants16 && !aligned(gi, 16)) return fail(LQ_EALIGN, "lq_batch_penalty_grads: gradient buffer of tensor %d is not 16-byte aligned", tb.index[i]);
            pk.dy[i] = gi;
        }
    }
    const uint32_t* gpre = tb.prefix_d + nt;
    if (kind == LQ_PENALTY_MAXBIN) {
        PtrPack none;
        memset(&none, 0, sizeof(none));
        hipLaunchKernelGGL((k_batch_traverse<OP_MAXBIN_FWD>), dim3(tb.blocks), dim3(kBlock), 0, st, tb.d, tb.block_task_d, nt, (uint32_t*)ws, none, 0, cf);
        hipLaunchKernelGGL((k_batch_finalize<OP_MAXBIN_FWD>), dim3(tb.groups), dim3(64), 0, st, tb.d, gpre, nt, (uint32_t*)ws, 0);
        hipLaunchKernelGGL((k_batch_traverse<OP_MAXBIN_BWD>), dim3(tb.blocks), dim3(kBlock), 0, st, tb.d, tb.block_task_d, nt, (uint32_t*)ws, pk, 2, cf);
        hipLaunchKernelGGL(k_batch_penalty_ds, dim3((unsigned)ceil_div(tb.groups, kBlock)), dim3(kBlock), 0, st, tb.d, gpre, nt, tb.groups, 0, cf, accum);
    } else if (kind == LQ_PENALTY_DIFFERENCE) {
        hipLaunchKernelGGL((k_batch_traverse<OP_DIFF_BWD>), dim3(tb.blocks), dim3(kBlock), 0, st, tb.d, tb.block_task_d, nt, (uint32_t*)ws, pk, 2, cf);
        hipLaunchKernelGGL((k_batch_finalize<OP_DIFF_BWD>), dim3(tb.groups), dim3(64), 0, st, tb.d, gpre, nt, (uint32_t*)ws, accum);
    } else {
        hipLaunchKernelGGL(k_batch_penalty_ds, dim3((unsigned)ceil_div(tb.groups, kBlock)), dim3(kBlock), 0, st, tb.d, gpre, nt, tb.groups, 2, cf, accum);
    }
    return check_hip("batch penalty launch");
}

int lq_batch_scale_adam(const lq_batch* b, double lr, double beta1, double beta2, double eps, int64_t step,
                        const int64_t* step_dev, int mode, void* stream) {
    if (!b) return fail(LQ_EINVAL, "lq_batch_scale_adam: NULL batch");
    if (b->adam_h.empty()) return LQ_OK;
    if (!step_dev && step < 1) return fail(LQ_EINVAL, "lq_batch_scale_adam: step is 1-based");
    if (mode != LQ_ADAM_KERAS && mode != LQ_ADAM_TORCH) return fail(LQ_EINVAL, "lq_batch_scale_adam: bad mode %d", mode);
    hipLaunchKernelGGL(k_batch_adam, dim3((unsigned)b->adam_h.size()), dim3(kBlock), 0, (hipStream_t)stream, b->adam_d,
                       adam_hyper(lr, beta1, beta2, eps, step, step_dev, mode, 1));
    return check_hip("batch adam launch");
}

int lq_selftest_ratio_division(uint64_t seed, uint32_t blocks, uint32_t pairs_per_thread, uint64_t* mismatches_dev, void* stream) {
    if (!mismatches_dev || !aligned(mismatches_dev, 8)) return fail(LQ_EINVAL, "lq_selftest_ratio_division: mismatches_dev must be an 8-byte aligned device pointer");
    if (blocks == 0 || pairs_per_thread == 0) return fail(LQ_EINVAL, "lq_selftest_ratio_division: empty test");
    hipLaunchKernelGGL(k_selftest_ratio_div, dim3(blocks), dim3(kBlock), 0, (hipStream_t)stream, seed, pairs_per_thread,
                       reinterpret_cast<unsigned long long*>(mismatches_dev));
    return check_hip("selftest launch");
}

int lq_q_minmax(const float* P, const float* s, int32_t* minmax_dev, int64_t outer, int64_t G, int64_t inner, void* stream) {
    int rc = check_desc(outer, G, inner);
    if (rc) return rc;
    LQ_REQUIRE_PTR(P);
    LQ_REQUIRE_PTR(s);
    LQ_REQUIRE_PTR(minmax_dev);
    const int64_t n = outer * G * inner;
    int64_t blocks = ceil_div(n, (int64_t)kBlock * 8);
    if (blocks > 1024) blocks = 1024;      // 4 blocks per CU; 2 x 1024 same-address atomics at the end
    // grid-stride index + stride stays below 2^32 for n < 2^31 (stride <= 1024 * 256)
    if (n < 2147483648ll) hipLaunchKernelGGL(k_q_minmax<uint32_t>, dim3((unsigned)blocks), dim3(kBlock), 0, (hipStream_t)stream, P, s, minmax_dev, n, G, inner);
    else hipLaunchKernelGGL(k_q_minmax<int64_t>, dim3((unsigned)blocks), dim3(kBlock), 0, (hipStream_t)stream, P, s, minmax_dev, n, G, inner);
    return check_hip("q minmax launch");
}

int lq_q_histogram(const float* P, const float* s, int32_t qmin, int64_t nbins, uint32_t* bins_dev, int64_t outer, int64_t G,
                   int64_t inner, void* stream) {
    int rc = check_desc(outer, G, inner);
    if (rc) return rc;
    if (nbins <= 0) return fail(LQ_EINVAL, "lq_q_histogram: nbins must be positive");
    LQ_REQUIRE_PTR(P);
    LQ_REQUIRE_PTR(s);
    LQ_REQUIRE_PTR(bins_dev);
    const int64_t n = outer * G * inner;
    int64_t blocks = ceil_div(n, (int64_t)kBlock * 16);
    if (blocks > 2048) blocks = 2048;
    if (n < 2147483648ll) hipLaunchKernelGGL(k_q_histogram<uint32_t>, dim3((unsigned)blocks), dim3(kBlock), 0, (hipStream_t)stream, P, s, qmin, nbins, bins_dev, n, G, inner);
    else hipLaunchKernelGGL(k_q_histogram<int64_t>, dim3((unsigned)blocks), dim3(kBlock), 0, (hipStream_t)stream, P, s, qmin, nbins, bins_dev, n, G, inner);
    return check_hip("q histogram launch");
}

int lq_selftest_uniform_division(uint64_t seed, uint32_t blocks, uint32_t pairs_per_thread, uint64_t* mismatches_dev, void* stream) {
    if (!mismatches_dev || !aligned(mismatches_dev, 8)) return fail(LQ_EINVAL, "lq_selftest_uniform_division: mismatches_dev must be an 8-byte aligned device pointer");
    if (blocks == 0 || pairs_per_thread == 0) return fail(LQ_EINVAL, "lq_selftest_uniform_division: empty test");
    hipLaunchKernelGGL(k_selftest_uniform_div, dim3(blocks), dim3(kBlock), 0, (hipStream_t)stream, seed, pairs_per_thread,
                       reinterpret_cast<unsigned long long*>(mismatches_dev));
    return check_hip("selftest launch");
}

}  // extern "C" (reopened below)

struct lq_adam_set {
    std::vector<lq::VecTask> h;
    lq::VecTask* d = nullptr;
    uint32_t blocks = 0;
};

extern "C" {

int lq_adam_set_create(float* const* w, float* const* m, float* const* v, const int64_t* n, const float* min_value, int count,
                       lq_adam_set** out) {
    if (!w || !m || !v || !n || !out) return fail(LQ_EINVAL, "lq_adam_set_create: NULL argument");
    if (count <= 0 || count > kBatchMax) return fail(LQ_EINVAL, "lq_adam_set_create: count must be in 1..%d", kBatchMax);
    lq_adam_set* a = new (std::nothrow) lq_adam_set();
    if (!a) return fail(LQ_EHIP, "lq_adam_set_create: out of host memory");
    for (int i = 0; i < count; ++i) {
        if (!w[i] || !m[i] || !v[i] || n[i] <= 0 || !aligned(w[i], 4) || !aligned(m[i], 4) || !aligned(v[i], 4)) {
            delete a;
            return fail(LQ_EINVAL, "lq_adam_set_create: tensor %d has a NULL/misaligned pointer or a non-positive size", i);
        }
        VecTask t;
        t.w = w[i];
        t.m = m[i];
        t.v = v[i];
        t.n = n[i];
        t.min_value = min_value ? min_value[i] : -INFINITY;
        t.first_block = a->blocks;
        const int64_t nb = ceil_div(n[i], (int64_t)kBlock * 4);
        if ((uint64_t)a->blocks + (uint64_t)nb > 0x7fffffffull) {
            delete a;
            return fail(LQ_EINVAL, "lq_adam_set_create: too many blocks");
        }
        a->blocks += (uint32_t)nb;
        a->h.push_back(t);
    }
    hipError_t e = hipMalloc(&a->d, a->h.size() * sizeof(VecTask));
    if (e == hipSuccess) e = hipMemcpy(a->d, a->h.data(), a->h.size() * sizeof(VecTask), hipMemcpyHostToDevice);
    if (e != hipSuccess) {
        const int rc = fail(LQ_EHIP, "lq_adam_set_create: %s", hipGetErrorString(e));
        if (a->d) (void)hipFree(a->d);
        delete a;
        return rc;
    }
    *out = a;
    return LQ_OK;
}

int lq_adam_set_destroy(lq_adam_set* a) {
    if (!a) return LQ_OK;
    if (a->d) (void)hipFree(a->d);
    delete a;
    return LQ_OK;
}

int lq_adam_set_step(const lq_adam_set* a, const float* const* grads, double lr, double beta1, double beta2, double eps, int64_t step,
                     const int64_t* step_dev, int mode, void* stream) {
    if (!a || !grads) return fail(LQ_EINVAL, "lq_adam_set_step: NULL argument");
    if (!step_dev && step < 1) return fail(LQ_EINVAL, "lq_adam_set_step: step is 1-based");
    if (mode != LQ_ADAM_KERAS && mode != LQ_ADAM_TORCH) return fail(LQ_EINVAL, "lq_adam_set_step: bad mode %d", mode);
    PtrPack pk;
    memset(&pk, 0, sizeof(pk));
    for (size_t i = 0; i < a->h.size(); ++i) {
        if (grads[i] && !aligned(grads[i], 4)) return fail(LQ_EINVAL, "lq_adam_set_step: gradient %zu is misaligned", i);
        pk.dy[i] = grads[i];     // NULL = no gradient this step: that tensor is skipped
    }
    hipLaunchKernelGGL(k_multi_adam, dim3(a->blocks), dim3(kBlock), 0, (hipStream_t)stream, a->d, (int)a->h.size(), pk, (float)lr,
                       (float)beta1, (float)beta2, lr, beta1, beta2, (float)(1.0 - beta1), (float)(1.0 - beta2), (float)eps, step_dev, step,
                       mode);
    return check_hip("multi adam launch");
}

}  // extern "C"
