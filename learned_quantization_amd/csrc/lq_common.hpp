// lq_common.hpp -- constants, kernel parameter block, accumulator types (overview: lq_kernels.hip)
#ifndef LQ_COMMON_HPP_
#define LQ_COMMON_HPP_
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "lq_hip.h"
#include "lq_fastdiv.hpp"

namespace lq {

constexpr int kBlock = 256;
constexpr int kWavesPerBlock = kBlock / 64;
constexpr float kEpsF32 = 1.1920928955078125e-07f;   // np.finfo(np.float32).eps, custom_layers.py:11
constexpr int64_t kPeriodic4Min = 4ll << 20;         // column mode, C <= 64: elements from which the float4 grid-stride variant is used
                                                     // (the same 4 M boundary as the 512-thread streaming units: below it single-tensor
                                                     // and batched launches share one geometry and give bit-identical results)
constexpr int64_t kNtBytes = 64ll << 20;             // tensors at least this large are streamed with nontemporal accesses

// ------------------------------------------------------------------------------------------
//  Parameters shared by every kernel (passed by value in the kernarg segment).
// ------------------------------------------------------------------------------------------
struct Params {
    const float* P;
    const float* s;
    const float* dy;
    float* out;          // primary dense output (out for K1/K4, dP for penalty backward)
    void* q;             // optional integer view
    int q_dtype;
    float lam;
    int tmode;           // 0: lambda < 4e-4 (tanh(d) == d), 1: lambda <= 0.25 (polynomial), 2: general (ocml tanhf)
    const float* mb;     // per-group max(|P|/s)      (maxbin backward)
    const uint32_t* ties;
    const float* c_dev;  // upstream gradient, device scalar
    float c_scale;
    int accum;           // penalty backward: add to the existing contents of `out` instead of overwriting
    uint32_t* pa;        // partials, SoA
    uint32_t* pb;
    double* pc;          // vote sums are carried in f64 (exact for lambda < 4e-4, see Acc below)
    int64_t outer, G, inner;
    // direct emit (scale-gradient ops whose groups have exactly one partial: biases, row-wise Dense, one row per group):
    // the traversal writes the op's outputs itself and the finalize launch is skipped
    int direct;
    float* e0;           // ds[G]
    float* e1;           // optional parts[3*G]
    double ecount;       // elements per group
    // conv kernels: the parameter is HWIO (custom_layers.py:321), MIOpen consumes OIHW.  Element i = (h*ci + c)*co + o of the
    // HWIO tensor (h = kh*KW + kw) is element (o*ci + c)*hw + h of the OIHW one.  K1 can emit the OIHW tensor next to
    // `out`; K2 can take the upstream gradient in OIHW order and write dP (= dy, custom_layers.py:118) back in HWIO order.
    float* out_perm;       // K1: second output in OIHW order, or NULL
    const float* dy_perm;  // K2: upstream gradient in OIHW order (then `dy` is only a valid dummy), or NULL
    float* dp_out;         // K2: dP in HWIO order, written when dy_perm is set
    uint32_t perm_hw, perm_ci, perm_co;
};

__device__ __forceinline__ uint32_t perm_index(const Params& p, int64_t i) {      // conv kernels are far below 2^32 elements
    const uint32_t iu = (uint32_t)i;
    const uint32_t t = iu / p.perm_co, o = iu - t * p.perm_co;
    const uint32_t h = t / p.perm_ci, c = t - h * p.perm_ci;
    return (o * p.perm_ci + c) * p.perm_hw + h;
}

struct FlatIdx {          // group of flat element i: (i / inner) % G
    FastDiv inner, G;
};

// Per-group context, loaded once per row / column.
struct Ctx {
    float s;
    float r;      // RN(1/s)
    int fast;     // s is inside the window where the uniform-divisor division is exact
    float k0;
    float k1;
    float lam_hi; // RN(lambda * 1.000001): a >= lam_hi*b  =>  RN(a/b) >= lambda for sure
    int sure_ok;  // lam_hi*b cannot underflow for any b this row can produce (b >= min(s, eps_f32))
};

// Narrow accumulator (inside the traversal kernels) and wide accumulator (finalize).
//
// The sum `c` is carried in f64 from the first addition on.  For the nested-quantization vote (custom_layers.py:84 / :110) with
// lambda < 4e-4 -- every published threshold; there |tanh(d)| == d in fp32 -- this makes the sum EXACT and therefore independent
// of the order in which a traversal visits the elements: with lambda in [2^e, 2^(e+1)) every term d = RN(lambda - ratio),
// 0 <= ratio < lambda, is a multiple of 2^(e-24) (Sterbenz for ratio >= lambda/2; one rounding at ulp(lambda/2) below) and
// at most lambda, i.e. an integer below 2^25 in units of 2^(e-24); a group of up to 2^28 elements sums to less than 2^53 units.
// Every traversal (row stream, tiles, columns, multi-tensor batch, conv tiles, any block size, any number of ranks' partials
// merged in any order) therefore yields the SAME bits of ds -- and that value is the correctly rounded mean.  For lambda >= 4e-4
// the terms are polynomial / ocml tanh values without a common quantum: there the f64 sum still depends on the order, below
// 2^-50 relative, which survives the final rounding to fp32 only when the mean sits on a rounding boundary.
template <typename TB, typename TC>
struct AccT {
    uint32_t a;
    TB b;
    TC c;
};
using Acc = AccT<uint32_t, double>;
using AccW = AccT<double, double>;   // finalize: count in f64 as well (counts are exact below 2^53)

enum OpKind {
    OP_FWD = 0,        // K1
    OP_BWD = 1,        // K2
    OP_FUSED = 2,      // K4
    OP_MAXBIN_FWD = 3, // K5a
    OP_MAXBIN_BWD = 4,
    OP_DIFF_FWD = 5,   // K5b
    OP_DIFF_BWD = 6,
    OP_QONLY = 7,      // integer view only (callbacks / export)
    OP_FWD_PERM = 8,   // K1 that can also emit the OIHW companion of an HWIO conv kernel
    OP_BWD_PERM = 9,   // K2 that can take dy in OIHW order and write dP in HWIO order
};

}  // namespace lq

#endif
