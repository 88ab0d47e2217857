// lq_batch.hpp -- device side of the multi-tensor batch (task table, batch kernels)
#ifndef LQ_BATCH_HPP_
#define LQ_BATCH_HPP_
#include "lq_aux_kernels.hpp"
#include "lq_conv_tile.hpp"
#include "lq_batch_cols.hpp"

namespace lq {

// ------------------------------------------------------------------------------------------
//  Multi-tensor batch (SURVEY f-4): the 4 / 12 / 40 weight-sized tensors a training step fake-quantises
//  are latency-bound one by one (each launch costs more than its work).  A batch is a device-resident
//  table of tasks; ONE launch covers every tensor's traversal (each 256-thread block finds its task in a per-block
//  table) and ONE launch finalizes every group of every tensor -- and can apply the scales' Adam step (k_batch_finalize_t).
//  The per-tensor code is the single-tensor traversal bodies, or -- conv kernels with an OIHW companion -- the LDS tile
//  of lq_conv_tile.hpp; forward outputs, max|q| and vote counts are bit-identical to the single-tensor entry points by
//  construction, the vote sums because they are exact (lq_common.hpp, Acc).
//  Tasks are ordered by decreasing work per block, so the heavy tiles start first and the light blocks fill the tail.
// ------------------------------------------------------------------------------------------
constexpr int MODE_CONV_TILE = 3;
struct Task {
    Params p;                 // pa/pb/pc are rebound to the batch workspace inside the kernel
    ConvTile ct;              // mode == MODE_CONV_TILE
    float* ds;                // scale gradient output [G]
    float* dp;                // dP in HWIO order (conv kernels with an OIHW companion), else NULL
    float* am;                // Adam moments of the scale (fused update in the finalize), or NULL
    float* av;
    float amin;               // MinValueConstraint of the scale
    int pad2;
    float* mb;                // batch-owned per-group max(|P|/s)      (MaxBin penalty)
    uint32_t* ties;           // batch-owned per-group tie counts
    int mode, vec, lpr_log2, ru;   // ru: float4 per thread and stream of a row-stream unit (1, 2 or 4: units of 1024 / 2048 / 4096 elements)
    int64_t R, L, nc;         // row modes (block size 256)
    int64_t C, rps, nbx;      // column mode (rps = rows per block, nbx = blocks along the columns; col_variant 6: rps = number of blocks)
    int col_variant, pad1;
    FragGeom fg;              // scale-gradient tables, float4 column tiles: partial layout of lq_batch_cols.hpp (fg.F == 0: generic layout)
    int64_t np_pad;           // padded partial count; this task's workspace slice is 4 * np_pad words (u32, u32, f64)
    int64_t ws_off;           // offset of the slice in uint32 words
    int64_t gstride, n1, stride1, n2;   // finalize geometry
    double count;             // elements per group
    uint32_t first_block;     // prefix over traversal blocks
    uint32_t first_group;     // prefix over groups
};

// Traversal blocks find their task in a per-block table (one 2-byte load; a search costs 5-7 dependent loads at the head of
// every block, before its first streaming load can be issued).  Groups and Adam blocks search the contiguous prefix array:
// prefix[i] = first group of task i.
__device__ __forceinline__ int find_task(const uint32_t* __restrict__ prefix, int ntasks, uint32_t b) {
    int lo = 0, hi = ntasks - 1;
    while (lo < hi) {
        const int mid = (lo + hi + 1) >> 1;
        if (prefix[mid] <= b) lo = mid;
        else hi = mid - 1;
    }
    return lo;
}

// Upstream-gradient pointers change every step (autograd allocates them): they travel in the kernel
// arguments (captured at launch, no staging buffer to race on), at most kBatchMax per launch.
constexpr int kBatchMax = 256;
struct PtrPack {
    const float* dy[kBatchMax];
};
struct CoefPack {           // per-tensor upstream coefficient of a penalty term (host constants)
    float c[kBatchMax];
};

// use_pack: 0 = pointers from the task table; 1 = pk.dy[] are the upstream gradients (scale-gradient pass);
//           2 = penalty pass: pk.dy[] are the gradient buffers to ACCUMULATE into, cf.c[] the upstream coefficients;
//           3 = as 1, but the gradients of conv kernels with an OIHW companion arrive in OIHW order (OP_BWD_PERM)
template <int OP>
__global__ __launch_bounds__(kBlock) void k_batch_traverse(const Task* __restrict__ tasks, const uint32_t* __restrict__ block_task, int ntasks,
                                                           uint32_t* ws, PtrPack pk, int use_pack, CoefPack cf) {
    // one LDS scratch: the column traversal's accumulators, or the transposed tile of a conv kernel
    constexpr int kColBytes = OpT<OP>::kReduce ? (OP == OP_BWD ? kColLdsAcc : kBlock * 4) * (int)sizeof(Acc) : 16;      // OP_BWD: + the periodic form's second stage
    constexpr int kTileBytes = OP == OP_FWD_PERM ? kCtLdsWordsFwd * 4 : (OP == OP_BWD_PERM ? kCtLdsWordsBwd * 4 : 16);
    __shared__ __align__(16) unsigned char smem[kColBytes > kTileBytes ? kColBytes : kTileBytes];
    LQ_TRACE(0);
    const int ti = (int)block_task[blockIdx.x];
    (void)ntasks;
    const Task& t = tasks[ti];
    Params p = t.p;
    if (use_pack == 1) p.dy = pk.dy[ti];
    if (use_pack == 3) {
        if (p.perm_co) {
            p.dy_perm = pk.dy[ti];
            p.dy = p.P;               // the traversals' unconditional dy loads need a valid address; the op gathers from dy_perm
            p.dp_out = t.dp;
        } else {
            p.dy = pk.dy[ti];
        }
    }
    if (use_pack == 2) {
        p.out = const_cast<float*>(pk.dy[ti]);
        p.accum = 1;
        p.c_dev = nullptr;
        p.c_scale = cf.c[ti];
        p.mb = t.mb;
        p.ties = t.ties;
    }
    p.pa = ws + t.ws_off;
    p.pb = p.pa + t.np_pad;
    p.pc = reinterpret_cast<double*>(p.pb + t.np_pad);
    const uint32_t b = blockIdx.x - t.first_block;
    // (Tried, round 4: pinning every task field the traversal forms read to SGPRs right here, so that the seven dependent
    // s_load / s_waitcnt rounds the compiler spreads over the control flow below become one.  32 SGPRs spill to VGPR lanes and the
    // step is 1.3-2 us SLOWER on all three sets measured -- profiles/r04/experiments/pinned_task_fields.jsonl.)
    if (t.mode == MODE_CONV_TILE) {
        if constexpr (OP == OP_FWD_PERM) {
            conv_tile_fwd(p, t.ct, b, reinterpret_cast<float*>(smem));
        } else if constexpr (OP == OP_BWD_PERM) {        // every tile task has an OIHW companion: its gradient arrives in OIHW order
            conv_tile_bwd(p, t.ct, b, reinterpret_cast<float*>(smem));
        }
    } else if (t.mode == 0) {
        const uint32_t nc = (uint32_t)t.nc;
        const uint32_t row = b / nc, ck = b - row * nc;
        const int64_t g = (int64_t)(row % (uint32_t)p.G);
        if (t.vec) {
            // long rows (per-tensor scales: the whole tensor is one row) take units of 2048 / 4096 elements in the forward and the
            // scale-gradient tables: 1024-element blocks are 11 K blocks for the ResNet-18-like set, each paying the head of a block
            // (block -> task -> first load) for 4 KB per stream
            if constexpr (OP == OP_FWD || OP == OP_BWD) {
                if (t.ru == 4) row_stream_body<OP, 4, kBlock, 0, 4>(p, t.L, t.nc, (int64_t)row, (int64_t)ck, g);
                else if (t.ru == 2) row_stream_body<OP, 4, kBlock, 0, 2>(p, t.L, t.nc, (int64_t)row, (int64_t)ck, g);
                else row_stream_body<OP, 4, kBlock, 0>(p, t.L, t.nc, (int64_t)row, (int64_t)ck, g);
            } else {
                row_stream_body<OP, 4, kBlock, 0>(p, t.L, t.nc, (int64_t)row, (int64_t)ck, g);
            }
        } else {
            row_stream_body<OP, 1, kBlock, 0>(p, t.L, t.nc, (int64_t)row, (int64_t)ck, g);
        }
    } else if (t.mode == 1) {
        if (t.vec) row_small_body<OP, 4>(p, t.R, (int)t.L, t.lpr_log2, (int64_t)b);
        else row_small_body<OP, 1>(p, t.R, (int)t.L, t.lpr_log2, (int64_t)b);
    } else if constexpr (OP == OP_BWD) {
        // scale-gradient pass: float4 tiles run the fragment form (lq_batch_cols.hpp); the generic float4 tile is not instantiated here
        if (t.col_variant == 0) {
            col_small_body<OP>(p, (int)t.C, t.rps, (int64_t)b, reinterpret_cast<Acc*>(smem));
        } else if (t.col_variant == 6) {       // C <= 64: the periodic float4 stream (a thread's four columns never change), t.rps blocks
            col_periodic4_body<OP, 0>(p, (int)t.C, t.rps, (int64_t)b, reinterpret_cast<Acc*>(smem));
        } else {
            const uint32_t nbx = (uint32_t)t.nbx;
            const uint32_t by = b / nbx, bx = b - by * nbx;
            if (t.col_variant >= 4) {
                if (LQ_BATCH_GF2 && t.fg.finner >= 2u) col_frag_tile_body<OP, LQ_BATCH_U2, 2, LQ_BATCH_PIPE>(p, (uint32_t)t.C, (uint32_t)t.rps, bx, by, t.fg, reinterpret_cast<Acc*>(smem));
                else col_frag_tile_body<OP, LQ_BATCH_U, 4, LQ_BATCH_PIPE>(p, (uint32_t)t.C, (uint32_t)t.rps, bx, by, t.fg, reinterpret_cast<Acc*>(smem));
            }
            else col_tile_body<OP, 1, 0>(p, t.C, t.rps, (int64_t)bx, (int64_t)by, reinterpret_cast<Acc*>(smem));
        }
#ifndef LQ_BATCH_FWD_TILE
#define LQ_BATCH_FWD_TILE 1   // the forward's float4 column tiles with the addressing of lq_batch_cols.hpp (0: the generic col_tile_body):
#endif                        // forward launch 16.4 -> 15.8 us, step -1.0 us on both ResNet sets (profiles/r04/experiments/forward_tile_addressing.jsonl)
    } else if constexpr (LQ_BATCH_FWD_TILE && OP == OP_FWD) {
        if (t.col_variant == 4 && !p.q && t.C < (1ll << 30) && p.outer < (1ll << 31)) {
            const uint32_t nbx = (uint32_t)t.nbx;
            const uint32_t by = b / nbx, bx = b - by * nbx;
            col_fwd_tile_body<4>(p, (uint32_t)t.C, (uint32_t)t.rps, bx, by);
        } else {
            col_body<OP, true>(p, t.C, t.rps, t.nbx, t.col_variant, (int64_t)b, t.rps, reinterpret_cast<Acc*>(smem));
        }
    } else {
        // (the periodic float4 stream, variant 6, is planned for the forward and the scale-gradient tables only)
        col_body<OP, OP == OP_FWD>(p, t.C, t.rps, t.nbx, t.col_variant, (int64_t)b, t.rps, reinterpret_cast<Acc*>(smem));
    }
#ifdef LQ_DEV_KNOBS
    __syncthreads();
    __builtin_amdgcn_s_waitcnt(0);
    LQ_TRACE(1);
#endif
}

template <int OP, int BS = 64>     // BS = 256 when some group of the batch has more than 256 partials (host decides)
__global__ __launch_bounds__(BS) void k_batch_finalize(const Task* __restrict__ tasks, const uint32_t* __restrict__ gprefix, int ntasks,
                                                       uint32_t* ws, int accum = 0) {
    const int ti = find_task(gprefix, ntasks, blockIdx.x);
    const Task& t = tasks[ti];
    Params p = t.p;
    p.pa = ws + t.ws_off;
    p.pb = p.pa + t.np_pad;
    p.pc = reinterpret_cast<double*>(p.pb + t.np_pad);
    FinGeom f;
    f.groups = p.G;
    f.gstride = t.gstride;
    f.n1 = t.n1;
    f.stride1 = t.stride1;
    f.n2 = t.n2;
    f.count = t.count;
    f.o0 = (OP == OP_MAXBIN_FWD) ? t.mb : t.ds;
    f.o1 = nullptr;
    f.o2 = (OP == OP_MAXBIN_FWD) ? t.ties : nullptr;
    f.accum = accum;
    finalize_block_body<OP, BS>(p, f, (int64_t)(blockIdx.x - t.first_group), (int)threadIdx.x);
}

// Scale gradients of the MaxBin (kind 0) and Inverse (kind 2) penalty terms for every group of every tensor:
//   maxbin   ds[g] = -((c/G) * mb[g]) / s[g]          (custom_loss_functions.py:92,110; reduce_max + RealDiv gradients)
//   inverse  ds[g] = s[g] == 0 ? 0 : -((c/G) / s[g]) / s[g]                     (:252-255)
__global__ __launch_bounds__(kBlock) void k_batch_penalty_ds(const Task* __restrict__ tasks, const uint32_t* __restrict__ gprefix, int ntasks,
                                                             uint32_t total_groups, int kind, CoefPack cf, int accum) {
    const uint32_t gg = blockIdx.x * kBlock + threadIdx.x;
    if (gg >= total_groups) return;
    const int ti = find_task(gprefix, ntasks, gg);
    const Task& t = tasks[ti];
    const uint32_t g = gg - t.first_group;
    const float up = cf.c[ti], sg = t.p.s[g], G = (float)t.p.G;
    const float v = (kind == 0) ? -((up / G) * t.mb[g]) / sg : ((sg == 0.0f) ? 0.0f : -((up / G) / sg) / sg);
    t.ds[g] = accum ? t.ds[g] + v : v;
}

// K6 for every scale of the batch in one launch: block per tensor.
struct AdamTask {
    float* s;
    const float* ds;
    float* m;
    float* v;
    int64_t n;
    float min_value;
    int pad;
};

// Hyper-parameters of one Adam step (kernel argument).  `on` = 0: the finalize only emits ds.
struct AdamHyper {
    double lr_d, b1_d, b2_d;
    const int64_t* step_dev;
    int64_t step_host;
    float lr, b1, b2, f0, f1, eps;
    int mode, on;
};
struct AdamCoef {
    float alpha, step_size, sq_bc2;
};
__device__ __forceinline__ AdamCoef adam_coef(const AdamHyper& h) {
    const int64_t step = h.step_dev ? h.step_dev[0] : h.step_host;
    AdamCoef c;
    c.alpha = 0.f;
    c.step_size = 0.f;
    c.sq_bc2 = 1.f;
    if (h.mode == LQ_ADAM_KERAS) {
        const float b1p = powf(h.b1, (float)step), b2p = powf(h.b2, (float)step);
        c.alpha = h.lr * sqrtf(1.0f - b2p) / (1.0f - b1p);
    } else {
        const double bc1 = 1.0 - pow(h.b1_d, (double)step), bc2 = 1.0 - pow(h.b2_d, (double)step);
        c.step_size = (float)(h.lr_d / bc1);
        c.sq_bc2 = (float)sqrt(bc2);
    }
    return c;
}
// one scale element: Adam (Keras 2.11 or torch arithmetic) + MinValueConstraint (custom_layers.py:42-43, 158)
__device__ __forceinline__ void adam_value(const AdamHyper& h, const AdamCoef& c, float g, float& mi, float& vi, float& w, float min_value) {
    mi = mi + (g - mi) * h.f0;
    vi = vi + (g * g - vi) * h.f1;
    if (h.mode == LQ_ADAM_KERAS) w = w - (mi * c.alpha) / (sqrtf(vi) + h.eps);
    else w = w - c.step_size * (mi / (sqrtf(vi) / c.sq_bc2 + h.eps));
    w = (w < min_value) ? min_value : w;
}
__device__ __forceinline__ void adam_element(const AdamHyper& h, const AdamCoef& c, float g, float* m, float* v, float* s, int64_t i, float min_value) {
    float mi = m[i], vi = v[i], w = s[i];
    adam_value(h, c, g, mi, vi, w, min_value);
    m[i] = mi;
    v[i] = vi;
    s[i] = w;
}

__global__ __launch_bounds__(kBlock) void k_batch_adam(const AdamTask* __restrict__ tasks, AdamHyper h) {
    const AdamTask t = tasks[blockIdx.x];
    const AdamCoef c = adam_coef(h);
    for (int64_t i = threadIdx.x; i < t.n; i += kBlock) adam_element(h, c, t.ds[i], t.m, t.v, t.s, i, t.min_value);
}

// Finalize of the scale-gradient pass for every group of every tensor, and -- `ah.on` -- the Adam step of that group's scale in
// the same launch (nothing may read or change ds in between: no loss term, no exchange of ds; the caller decides).
// A 256-thread block serves, by its entry of a per-block table (one 8-byte load instead of a binary search over the prefix
// array: 5-7 dependent loads at the head of every block were a third of this launch):
//   * wave form: FOUR groups of a task whose groups have at most 256 contiguous partials (a wave per group, no block barrier: the
//     summation order of the 64-thread finalize of the single-tensor entry points);
//   * wide form: ONE group of a task with more partials per group (the 256-thread finalize);
//   * column form: floor(64 / inner) groups of a column traversal.  Those write one partial per (row block, column), so a
//     group's partials sit `C` words apart and a wave per group gathers a whole line for every 4-byte word (the finalize of the
//     ResNet-50-like set, mostly 1 x 1 kernels, read 88 MB for 20 MB of partials and took 21 us).  Here lane l of each
//     wave walks partials of column c0 + l -- every load a contiguous run of a row of partials, the four waves taking the row
//     blocks in turn -- and the 4 x inner totals of a group meet in LDS.  Vote sums are exact (lq_common.hpp), max|q| and counts are order-free: the outputs are those of the
//     other forms bit for bit for lambda < 4e-4, within an ulp of the fp32 mean above.
// The emitting thread fetches the scale's Adam state BEFORE it walks the partials, so that after the reduction only arithmetic
// and three stores remain.
// One record per finalize block holds EVERYTHING the block needs (round 4): with a (task, first group) pair the block first fetched
// its pair, then -- a dependent trip to the L2 -- the task's geometry and pointers, and only then could issue the loads of the
// partials and of the Adam state; the launch is a chain of such trips (profiles/r04: 5.0 us for 1 MB of partials).
struct FinRec {
    uint32_t form;        // 0: wave form, 1: wide form, 2: column form, 3: fragment form (lq_batch_cols.hpp)
    uint32_t g0;          // first group of this block within its task
    uint32_t G;           // groups of the task
    float lam;            // FinT<OP_BWD>::emit: -|tanh(lambda)| where no element voted
    int64_t ws_off, np_pad;               // the task's slice of the workspace (uint32 words), padded partial count
    int64_t gstride, n1, stride1, n2;     // partial geometry (lq_traverse.hpp FinGeom); fragment form: n1 row blocks
    FragGeom fg;
    double count;         // elements per group
    float* ds;
    float* am;            // Adam moments of the scale, or NULL
    float* av;
    float* s;
    float amin;
    uint32_t pad;
};
static_assert(sizeof(FinRec) == 128, "one record = two 64-byte scalar loads");

template <int OP>
__global__ __launch_bounds__(256) void k_batch_finalize_t(const FinRec* __restrict__ recs, uint32_t* ws, AdamHyper ah) {
    using O = OpT<OP>;
    static_assert(O::kStdMerge, "wave-per-group finalize needs the DPP merge (no block barrier)");
    __shared__ AccW col_tot[256];
    const FinRec& t = recs[blockIdx.x];
    const bool wide = t.form == 1u, cols = t.form == 2u, frag = t.form == 3u;
    Params p;
    p.lam = t.lam;
    p.G = (int64_t)t.G;
    p.pa = ws + t.ws_off;
    p.pb = p.pa + t.np_pad;
    p.pc = reinterpret_cast<double*>(p.pb + t.np_pad);
    FinGeom f;
    f.groups = (int64_t)t.G;
    f.gstride = t.gstride;
    f.n1 = t.n1;
    f.stride1 = t.stride1;
    f.n2 = t.n2;
    f.count = t.count;
    f.o0 = t.ds;
    f.o1 = nullptr;
    f.o2 = nullptr;
    f.accum = 0;
    if (cols || frag) {                       // finalize_cols_body (lq_traverse.hpp): 64 / n2 groups per block, whole partial rows;
                                              // finalize_frag_body (lq_batch_cols.hpp): fg.gpb groups per block, one or two fragments each
        const int gpb = frag ? (int)t.fg.gpb : 64 / (int)t.n2;
        const int64_t gp = (int64_t)t.g0 + threadIdx.x;                       // the group this thread will hold (threads < gpb)
        const bool update = ah.on && t.am && (int)threadIdx.x < gpb && gp < (int64_t)t.G;
        AdamCoef c;
        float mi = 0.f, vi = 0.f, w = 0.f;
        if (update) {
            c = adam_coef(ah);
            mi = t.am[gp];
            vi = t.av[gp];
            w = t.s[gp];
        }
        int64_t g;
        AccW acc;
        const bool holds = frag ? finalize_frag_body<OP>(p, t.fg, t.n1, (int64_t)t.G, t.g0, col_tot, g, acc)
                                : finalize_cols_body<OP>(p, f, (int64_t)t.g0, col_tot, g, acc);
        if (holds) {
            const float dsg = FinT<OP>::emit(p, f, g, acc);
            if (update) {
                adam_value(ah, c, dsg, mi, vi, w, t.amin);
                t.am[g] = mi;
                t.av[g] = vi;
                t.s[g] = w;
            }
        }
        return;
    }
    const int64_t g = (int64_t)t.g0 + (wide ? 0 : (int64_t)(threadIdx.x >> 6));
    if (g >= (int64_t)t.G) return;           // wave-uniform, wave form only (it has no block barrier)
    const int tid = wide ? (int)threadIdx.x : (int)(threadIdx.x & 63);
    // the thread that will emit ds[g]: lane 63 of a one-wave finalize, thread 0 of the wide one
    const bool update = ah.on && t.am && tid == (wide ? 0 : 63);
    AdamCoef c;
    float mi = 0.f, vi = 0.f, w = 0.f;
    if (update) {
        c = adam_coef(ah);
        mi = t.am[g];
        vi = t.av[g];
        w = t.s[g];
    }
    const float dsg = wide ? finalize_block_body<OP, 256>(p, f, g, tid) : finalize_block_body<OP, 64>(p, f, g, tid);
    if (update) {
        adam_value(ah, c, dsg, mi, vi, w, t.amin);
        t.am[g] = mi;
        t.av[g] = vi;
        t.s[g] = w;
    }
}

// Multi-tensor Adam for arbitrary parameter tensors (SURVEY f-4): one launch for the whole parameter set.  Blocks of
// 256 threads x 4 elements; a block finds its tensor by binary search over the block prefix; gradients arrive through
// the kernel-argument pointer pack (autograd re-allocates them every step).
struct VecTask {
    float* w;
    float* m;
    float* v;
    int64_t n;
    float min_value;        // -inf: no projection
    uint32_t first_block;
};

__global__ __launch_bounds__(kBlock) void k_multi_adam(const VecTask* __restrict__ tasks, int ntasks, PtrPack grads, float lr, float b1, float b2,
                                                       double lr_d, double b1_d, double b2_d, float f0, float f1, float eps,
                                                       const int64_t* step_dev, int64_t step_host, int mode) {
    int lo = 0, hi = ntasks - 1;
    while (lo < hi) {
        const int mid = (lo + hi + 1) >> 1;
        if (tasks[mid].first_block <= blockIdx.x) lo = mid;
        else hi = mid - 1;
    }
    const VecTask t = tasks[lo];
    const float* __restrict__ g = grads.dy[lo];
    if (!g) return;          // no gradient this step: the tensor is skipped (block-uniform)
    const int64_t step = step_dev ? step_dev[0] : step_host;
    float alpha = 0.f, step_size = 0.f, sq_bc2 = 1.f;
    if (mode == LQ_ADAM_KERAS) {
        const float b1p = powf(b1, (float)step), b2p = powf(b2, (float)step);
        alpha = lr * sqrtf(1.0f - b2p) / (1.0f - b1p);
    } else {
        const double bc1 = 1.0 - pow(b1_d, (double)step), bc2 = 1.0 - pow(b2_d, (double)step);
        step_size = (float)(lr_d / bc1);
        sq_bc2 = (float)sqrt(bc2);
    }
    const int64_t base = (int64_t)(blockIdx.x - t.first_block) * (kBlock * 4);
#pragma unroll
    for (int u = 0; u < 4; ++u) {
        const int64_t i = base + u * kBlock + threadIdx.x;
        if (i < t.n) {
            const float gi = g[i];
            float mi = t.m[i], vi = t.v[i], w = t.w[i];
            mi = mi + (gi - mi) * f0;
            vi = vi + (gi * gi - vi) * f1;
            if (mode == LQ_ADAM_KERAS) w = w - (mi * alpha) / (sqrtf(vi) + eps);
            else w = w - step_size * (mi / (sqrtf(vi) / sq_bc2 + eps));
            w = (w < t.min_value) ? t.min_value : w;
            t.m[i] = mi;
            t.v[i] = vi;
            t.w[i] = w;
        }
    }
}

}  // namespace lq

#endif
