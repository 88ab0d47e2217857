/* lq_hip.h -- C ABI of the MI355X (gfx950) learned-quantization hot path.
 *
 * The reference (anuunchin/learned-quantization) is pure Python on TensorFlow 2.11
 * and has no native interface of its own; the operator surface it exposes for this
 * path is the Python op `my_custom_gradient` and the loss-term methods.  Every entry
 * point below names the reference interface it replaces (file:line relative to
 * /root/reference).  The ctypes binding a maintainer adds is shown in INTEGRATION.md.
 *
 * Conventions (all entry points):
 *   - plain pointers and sizes, no framework types; all tensor pointers are DEVICE
 *     pointers to contiguous float32 unless stated otherwise;
 *   - group descriptor (outer, G, inner): element i of a contiguous tensor uses scale
 *     element  g = (i / inner) % G ; numel = outer * G * inner.  It encodes the four
 *     `orientation`s of CustomQuantizedScaleLayer.build
 *     (MNIST/nested_quantization_layer/custom_components/custom_layers.py:147-197);
 *   - `stream` is a hipStream_t passed as void* (NULL = default stream); kernels are
 *     enqueued and the call returns without synchronising; nothing is allocated;
 *   - `ws` is caller-owned DEVICE scratch of at least lq_workspace_bytes(outer,G,inner)
 *     bytes, 16-byte aligned; it carries deterministic two-stage reduction partials
 *     (no float atomics: results are run-to-run bit-stable);
 *   - return value: LQ_OK (0) or a negative lq_status; lq_last_error() returns a
 *     thread-local message for the last failing call on this thread.
 */
#ifndef LQ_HIP_H_
#define LQ_HIP_H_

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define LQ_ABI_VERSION 3

typedef enum lq_status {
    LQ_OK = 0,
    LQ_EINVAL = -1,      /* bad argument (null pointer, non-positive extent, bad enum) */
    LQ_EHIP = -2,        /* a HIP runtime call / kernel launch failed                  */
    LQ_EWORKSPACE = -3,  /* workspace missing or too small                             */
    LQ_EALIGN = -4       /* pointer not 4-byte aligned / workspace not 16-byte aligned */
} lq_status;

typedef enum lq_qdtype {
    LQ_Q_NONE = 0,
    LQ_Q_F32 = 1,        /* integer-valued float32, exactly the reference's tf.floor output */
    LQ_Q_I32 = 2,        /* saturating float->int32                                          */
    LQ_Q_I8 = 3          /* two's-complement wrap of the integer, = numpy .astype(int8) of
                            CIFAR-10/nested_quantization_layer/utils/log_scripts.py:74-79    */
} lq_qdtype;

typedef enum lq_adam_mode {
    LQ_ADAM_KERAS = 0,   /* Keras 2.11: var -= lr*sqrt(1-b2^t)/(1-b1^t) * m / (sqrt(v) + eps) */
    LQ_ADAM_TORCH = 1    /* torch.optim.Adam: m_hat / (sqrt(v_hat) + eps)                      */
} lq_adam_mode;

int lq_version(void);                 /* LQ_ABI_VERSION of the loaded library            */
const char* lq_last_error(void);      /* thread-local, never NULL                        */
const char* lq_status_string(int status);

/* Scratch bytes needed by every reducing entry point for this descriptor. */
size_t lq_workspace_bytes(int64_t outer, int64_t G, int64_t inner);

/* ---- K1: fake-quant forward -------------------------------------------------------
 * Replaces the forward of `my_custom_gradient`
 *   MNIST/nested_quantization_layer/custom_components/custom_layers.py:55-60,120
 *   CIFAR-10/custom_loss_terms/custom_components/custom_layers.py:53-59,64
 * t = P / s (IEEE RN), q = floor(t), out = q * s.  `out` may be NULL when only q is
 * wanted (callbacks/export: custom_callbacks.py:85-87, log_scripts.py:74-79).       */
int lq_fq_forward(const float* P, const float* s, float* out, void* q, int q_dtype,
                  int64_t outer, int64_t G, int64_t inner, void* stream);

/* ---- K2+K3: nested-quantization scale gradient ------------------------------------
 * Replaces `custom_grad` of the NQ op, custom_layers.py:62-118:
 *   ratio = |dy| / |where(out==0, eps_f32, out)|            (:63-64)
 *   m_g = max|q| , A_g = all(ratio >= lambda)                (:68-73, :94-99)
 *   mean_g = A_g ? -|tanh(lambda)| : mean(ratio>=lambda ? 0 : -|tanh(lambda-ratio)|)  (:77-88,:103-114)
 *   ds[g] = mean_g * m_g                                     (:116)
 * dP is `dy` itself (STE, :118) and is therefore not an output.
 * `parts` (optional, DEVICE float[3*G]) receives m_g, mean_g and the count of elements
 * with ratio < lambda (as float) for diagnostics / tests.                             */
int lq_fq_scale_grad(const float* P, const float* s, const float* dy, float lambda,
                     float* ds, float* parts, void* ws, size_t ws_bytes,
                     int64_t outer, int64_t G, int64_t inner, void* stream);

/* ---- K4: forward and NQ backward of one tensor in a single pass (benchmark path) ---
 * Same results as lq_fq_forward followed by lq_fq_scale_grad (out, max|q| and the vote count bit for bit; on
 * streaming-size tensors the vote sum may differ by fp32 summation order, ~1e-7 relative); P is read once.  */
int lq_fq_fwd_bwd_fused(const float* P, const float* s, const float* dy, float lambda,
                        float* out, float* ds, void* ws, size_t ws_bytes,
                        int64_t outer, int64_t G, int64_t inner, void* stream);

/* ---- K5a: MaxBin penalty tensor term ------------------------------------------------
 * Replaces the per-tensor part of SCCEMaxBin.compute_maxbin_penalty,
 *   CIFAR-10/custom_loss_terms/custom_components/custom_loss_functions.py:90-100,110:
 *   mb[g] = max_{i in g} |P_i| / s[g] ;  *term = mean_g mb[g].
 * `ties[g]` = number of elements attaining the maximum (TensorFlow's reduce_max gradient
 * splits evenly over ties).  Backward for upstream c = (*c_dev) * c_scale on `term`:
 *   dP_i = (|P_i|/s == mb[g]) ? sign(P_i) * c / (G * ties[g] * s[g]) : 0
 *   ds[g] = -c * mb[g] / (G * s[g])                                                   */
int lq_penalty_maxbin_fwd(const float* P, const float* s, float* mb, uint32_t* ties, float* term,
                          void* ws, size_t ws_bytes,
                          int64_t outer, int64_t G, int64_t inner, void* stream);
int lq_penalty_maxbin_bwd(const float* P, const float* s, const float* mb, const uint32_t* ties,
                          const float* c_dev, float c_scale, float* dP, float* ds,
                          int64_t outer, int64_t G, int64_t inner, void* stream);

/* ---- K5b: Difference penalty tensor term --------------------------------------------
 * Replaces the per-tensor part of SCCEDifference.compute_difference_penalty,
 *   custom_loss_functions.py:172-176:  *term = mean_i |P_i - P_i / s[g]|.
 * Backward, u = P - P/s, gi = sign(u) * c / N:
 *   dP_i = gi - gi / s[g] ;  ds[g] = sum_{i in g} gi * (P_i / s[g]) / s[g]           */
int lq_penalty_difference_fwd(const float* P, const float* s, float* term,
                              void* ws, size_t ws_bytes,
                              int64_t outer, int64_t G, int64_t inner, void* stream);
int lq_penalty_difference_bwd(const float* P, const float* s, const float* c_dev, float c_scale,
                              float* dP, float* ds, void* ws, size_t ws_bytes,
                              int64_t outer, int64_t G, int64_t inner, void* stream);

/* ---- K5c: Inverse penalty tensor term -----------------------------------------------
 * Replaces the per-tensor part of SCCEInverse.compute_inverse_penalty,
 *   custom_loss_functions.py:252-256:  *term = mean_g 1 / where(s[g]==0, eps_f32, s[g]).
 * Backward: ds[g] = s[g]==0 ? 0 : -c / (G * s[g]^2).                                  */
int lq_penalty_inverse_fwd(const float* s, float* term, int64_t G, void* stream);
int lq_penalty_inverse_bwd(const float* s, const float* c_dev, float c_scale, float* ds,
                           int64_t G, void* stream);

/* ---- K6: scale update ---------------------------------------------------------------
 * Adam step on a scale vector fused with the projection of MinValueConstraint
 *   custom_layers.py:35-46, attached at :158,170,182,191 with min_value = 100*eps_f32:
 *   s <- max(adam(s, ds), min_value).  `step` is the 1-based iteration count.  Hyper-parameters are
 *   doubles because Keras/torch form (1-beta) and the bias corrections from Python doubles before
 *   the fp32 tensor arithmetic; passing floats would change the last bits of every update.
 * lq_min_value_project is the bare constraint  w <- max(w, min_value)  (:42-43).      */
int lq_scale_adam_step(float* s, const float* ds, float* m, float* v, int64_t n,
                       double lr, double beta1, double beta2, double eps, int64_t step,
                       float min_value, int mode, void* stream);
/* Capture-safe form: the 1-based step is read from DEVICE memory at execution time (int64), so the launch can
 * live inside a hipGraph and be replayed while the caller increments the counter on the device.           */
int lq_scale_adam_step_dev(float* s, const float* ds, float* m, float* v, int64_t n,
                           double lr, double beta1, double beta2, double eps, const int64_t* step_dev,
                           float min_value, int mode, void* stream);
int lq_min_value_project(float* w, int64_t n, float min_value, void* stream);

/* ---- integer-view statistics (callbacks) ----------------------------------------------
 * max |floor(P/s)| over all elements sharing the index along `axis` of a tensor viewed
 * as (pre, n_axis, post) -- the reduce_max(abs(q), axis=1) of
 *   CIFAR-10/nested_quantization_layer/custom_components/custom_callbacks.py:98-99
 * is the complementary reduction: it keeps every axis but 1.  Here `keep_pre`/`keep_post`
 * select the kept extents: result[pre_i, post_j] = max over the middle axis.
 * The scale descriptor (outer,G,inner) is independent of the reduction geometry.      */
int lq_q_absmax_over_axis(const float* P, const float* s, float* result,
                          int64_t pre, int64_t n_axis, int64_t post,
                          int64_t outer, int64_t G, int64_t inner, void* stream);

/* ---- multi-tensor batch -----------------------------------------------------------------
 * A training step fake-quantises every custom layer's kernel and bias (4 / 12 / 40 tensors in the
 * reference's MNIST / CIFAR-10 / Imagenette models); one by one they are pure launch latency.
 * An lq_batch is a device-resident table of per-tensor tasks built ONCE from stable pointers:
 *   lq_batch_forward     = lq_fq_forward of every tensor in ONE launch                   (custom_layers.py:55-60)
 *   lq_batch_scale_grad  = lq_fq_scale_grad of every tensor with a finite lambda in TWO  (custom_layers.py:62-118)
 *                          launches; `dy` (optional, [n] device pointers, indexed like the descriptors)
 *                          overrides the descriptors' dy -- the upstream gradients move every step
 *   lq_batch_scale_adam  = lq_scale_adam_step(_dev) of every scale with Adam state in ONE launch
 *   lq_batch_scale_grad_step = the two above in the two launches of the first
 * q, out, max|q| and vote counts are bit-identical to the single-tensor entry points, and so is ds for lambda < 4e-4 (every
 * published threshold: the vote sums are exact in float64, any traversal and finalize order gives the same bits); for larger
 * lambda the float64 summation order of the traversal (tensors from 4 M elements) or of the finalize form may differ, which
 * reaches the fp32 result only when the mean sits on a rounding boundary.
 * lq_batch_create/destroy allocate/free the table (hipMalloc; not stream-ordered, never inside a capture);
 * the other calls only enqueue.  lambda = NaN marks an STE-only tensor (custom_loss_terms variant).        */
typedef struct lq_tensor_desc {
    const float* P;      /* parameter, dense fp32; (outer, G, inner) describes its elements in MEMORY order */
    const float* s;      /* scale [G]                                            */
    const float* dy;     /* default upstream gradient (may be NULL)              */
    float* out;          /* fake-quantised output (may be NULL next to out_oihw when lq_conv_tile_supported) */
    float* ds;           /* scale gradient [G] (required when lambda is finite)  */
    float* m;            /* Adam first moment of the scale [G] or NULL           */
    float* v;            /* Adam second moment of the scale [G] or NULL          */
    int64_t outer, G, inner;
    float lambda;        /* penalty_threshold, or NaN for STE-only               */
    float min_value;     /* MinValueConstraint bound used by lq_batch_scale_adam */
    /* conv kernels (HWIO, custom_layers.py:321): optional OIHW companions, see lq_fq_forward_oihw below; conv_co = 0: none */
    float* out_oihw;     /* second forward output in OIHW order, or NULL          */
    float* dp;           /* dP in HWIO order, written by lq_batch_scale_grad_oihw */
    int64_t conv_hw, conv_ci, conv_co;   /* kh*kw, input channels, output channels  */
} lq_tensor_desc;

typedef struct lq_batch lq_batch;

int lq_batch_create(const lq_tensor_desc* descs, int n, lq_batch** out);
int lq_batch_destroy(lq_batch* batch);
size_t lq_batch_workspace_bytes(const lq_batch* batch);
int lq_batch_forward(const lq_batch* batch, void* stream);
int lq_batch_scale_grad(const lq_batch* batch, const float* const* dy, void* ws, size_t ws_bytes, void* stream);
/* As lq_batch_scale_grad, but for every tensor that has an OIHW companion dy[i] is in OIHW order (MIOpen's weight gradient
 * as it stands) and dP (= dy, custom_layers.py:118) is written in HWIO order to the descriptor's `dp` buffer.            */
int lq_batch_scale_grad_oihw(const lq_batch* batch, const float* const* dy, void* ws, size_t ws_bytes, void* stream);
int lq_batch_scale_adam(const lq_batch* batch, double lr, double beta1, double beta2, double eps, int64_t step,
                        const int64_t* step_dev, int mode, void* stream);
/* lq_batch_scale_grad (dy_oihw = 0) or lq_batch_scale_grad_oihw (dy_oihw = 1) followed by lq_batch_scale_adam, in the TWO
 * launches of the former: the finalize that emits ds[g] applies the Adam step and the MinValueConstraint to s[g] itself
 * (custom_layers.py:116, 158).  Same ds, m, v and s, bit for bit, as the two calls one after the other.  For the step in which
 * nothing reads or changes ds between its computation and the update (nested-quantization training without a loss term and
 * without an exchange of ds); every tensor of the batch must have lambda, ds and Adam state (LQ_EINVAL otherwise).          */
int lq_batch_scale_grad_step(const lq_batch* batch, const float* const* dy, int dy_oihw, void* ws, size_t ws_bytes,
                             double lr, double beta1, double beta2, double eps, int64_t step, const int64_t* step_dev,
                             int mode, void* stream);

/* Custom-loss-term gradients of every tensor of the batch in 2-4 launches
 *   (CIFAR-10/custom_loss_terms/custom_components/custom_loss_functions.py:75-116, 161-195, 240-275).
 * The total loss is mean(SCCE) + gamma * penalty (:58), so the upstream coefficient of tensor i's term is the HOST
 * constant coeff[i] = gamma * numel_i / normalizer -- no autograd node per tensor is needed:
 *   grad[i][...] += coeff[i] * d term_i / d P_i     (MaxBin, Difference; accumulated into the existing gradient)
 *   ds_i[...]     = coeff[i] * d term_i / d s_i     (written to the descriptors' ds buffers)
 * coeff: host array [n]; grad: host array [n] of device pointers (unused for LQ_PENALTY_INVERSE).
 * kind | LQ_PENALTY_ACCUMULATE_DS: ds_i[...] += ... instead of = (a nested-quantization layer trained WITH a loss term:
 * the penalty's scale gradient is added to the hand-written one already in the buffer; no reference call site).     */
typedef enum lq_penalty_kind { LQ_PENALTY_MAXBIN = 0, LQ_PENALTY_DIFFERENCE = 1, LQ_PENALTY_INVERSE = 2 } lq_penalty_kind;
#define LQ_PENALTY_ACCUMULATE_DS 0x100
int lq_batch_penalty_grads(const lq_batch* batch, int kind, const float* coeff, float* const* grad,
                           void* ws, size_t ws_bytes, void* stream);

/* ---- multi-tensor Adam for the ordinary parameters (SURVEY f-4) -------------------------------------------
 * The reference trains everything with Keras 2.11 Adam(learning_rate=1e-4)
 *   CIFAR-10/nested_quantization_layer/experiment.py:435-443.
 * An lq_adam_set holds (w, m, v, n[, min_value]) of up to 256 tensors; lq_adam_set_step applies one Adam step to all of
 * them in ONE launch with the arithmetic of lq_scale_adam_step (mode LQ_ADAM_KERAS / LQ_ADAM_TORCH); `grads` is a host
 * array of device pointers (they move every step; a NULL entry skips that tensor), `step_dev` (optional) a device int64 for hipGraph capture.        */
typedef struct lq_adam_set lq_adam_set;
int lq_adam_set_create(float* const* w, float* const* m, float* const* v, const int64_t* n, const float* min_value, int count,
                       lq_adam_set** out);
int lq_adam_set_destroy(lq_adam_set* set);
int lq_adam_set_step(const lq_adam_set* set, const float* const* grads, double lr, double beta1, double beta2, double eps,
                     int64_t step, const int64_t* step_dev, int mode, void* stream);

/* ---- integer-view range and histogram (tracking callbacks) -----------------------------------------
 * The reference's callbacks pull floor(P/s) to the host and run np.unique on it every epoch
 *   CIFAR-10/nested_quantization_layer/custom_components/custom_callbacks.py:85-96, 131-207.
 * lq_q_minmax: minmax_dev[0] = min q, minmax_dev[1] = max q over the tensor (int32; the caller initialises the
 *   pair to {INT32_MAX, INT32_MIN}; NaN/Inf/out-of-int32 quotients are skipped).
 * lq_q_histogram: bins_dev[q - qmin] += count for qmin <= q < qmin + nbins (uint32 bins, zeroed by the caller).
 *   #unique = number of non-zero bins; (value, count) pairs = the non-zero bins.  Integer atomics: exact.      */
int lq_q_minmax(const float* P, const float* s, int32_t* minmax_dev, int64_t outer, int64_t G, int64_t inner, void* stream);
int lq_q_histogram(const float* P, const float* s, int32_t qmin, int64_t nbins, uint32_t* bins_dev,
                   int64_t outer, int64_t G, int64_t inner, void* stream);

/* ---- device self-test -------------------------------------------------------------------
 * Compares the kernels' in-window ratio division (rcp + Newton + fma chain, see lq_kernels.hip window_div)
 * with the IEEE `/` on blocks*256*pairs_per_thread pseudo-random operand pairs in [2^-40, 2^40]; ADDS the
 * number of bit mismatches to *mismatches_dev (uint64 on the device, zeroed by the caller). Expected: 0.   */
int lq_selftest_ratio_division(uint64_t seed, uint32_t blocks, uint32_t pairs_per_thread, uint64_t* mismatches_dev,
                               void* stream);
/* The same for the uniform-divisor division of the streaming kernels (r = RN(1/s) + two fma refinements, see
 * lq_math.hpp): random s in [2^-40, 2^40], random x in [2^-80, 2^81), both signs.  Expected: 0 mismatches.      */
int lq_selftest_uniform_division(uint64_t seed, uint32_t blocks, uint32_t pairs_per_thread, uint64_t* mismatches_dev,
                                 void* stream);

/* ---- conv kernels: HWIO parameter, OIHW consumer ---------------------------------------------------------------
 * The reference keeps conv kernels in HWIO (custom_layers.py:321) and `orientation` names HWIO axes; MIOpen consumes OIHW.
 * Instead of a separate transpose launch after K1 and another before K2 (what `qk.permute(3, 2, 0, 1)` costs in torch),
 *   lq_fq_forward_oihw     writes out (HWIO, as lq_fq_forward) AND out_oihw[(o*ci + c)*hw + h] = out[(h*ci + c)*co + o];
 *   lq_fq_scale_grad_oihw  takes dy in OIHW order, computes ds exactly as lq_fq_scale_grad would on the un-permuted dy
 *                          (same traversal, same summation order: bit-identical) and writes dP = dy in HWIO order.
 * hw = kh*kw; hw*ci*co must equal outer*G*inner.  Weight-sized tensors (below 2^32 elements).
 * Kernels of at most 9 taps with co % 4 == 0 and one of the reference's four orientations run as LDS tiles (whole 128-byte
 * lines on both the HWIO and the OIHW side); lq_fq_scale_grad_oihw then needs lq_conv_workspace_bytes() of scratch.
 * When the OIHW tensor is the only consumer (tf.nn.conv2d(x, qk) at custom_layers.py:340-348 is the kernel's ONLY use in a
 * training step), `out` may be NULL for kernels the tile path takes -- lq_conv_tile_supported() == 1 and P 16-byte aligned:
 * the forward then moves 8 bytes per element instead of 12.                                                             */
int lq_conv_tile_supported(int64_t hw, int64_t ci, int64_t co, int64_t outer, int64_t G, int64_t inner);
size_t lq_conv_workspace_bytes(int64_t hw, int64_t ci, int64_t co, int64_t outer, int64_t G, int64_t inner);
int lq_fq_forward_oihw(const float* P, const float* s, float* out, float* out_oihw, int64_t hw, int64_t ci, int64_t co,
                       int64_t outer, int64_t G, int64_t inner, void* stream);
int lq_fq_scale_grad_oihw(const float* P, const float* s, const float* dy_oihw, float lambda, float* ds, float* dP,
                          void* ws, size_t ws_bytes, int64_t hw, int64_t ci, int64_t co,
                          int64_t outer, int64_t G, int64_t inner, void* stream);

/* ---- profiling hook ----------------------------------------------------------------------
 * lq_profile_events(start, stop): while set (both hipEvent_t, created with timing enabled), every row-stream traversal
 * kernel this host thread launches (K1 / K2 / K4 of streaming-size tensors with rows >= 1024 elements: the BENCH path) is
 * launched through hipExtLaunchKernelGGL with these events, which then carry the kernel's OWN begin and end timestamps --
 * hipEventElapsedTime(start, stop) is the kernel duration rocprofv3 reports, without the cost of bracketing event records
 * and without the finalize launch of the two-launch calls.  lq_profile_events(NULL, NULL) switches it off.            */
int lq_profile_events(void* start, void* stop);

#ifdef __cplusplus
}
#endif
#endif /* LQ_HIP_H_ */
