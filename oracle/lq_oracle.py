"""CPU oracle for the learned-quantization hot path  --  TEST INFRASTRUCTURE ONLY.

This module is a float32 NumPy restatement, op for op and in the reference's
own evaluation order, of

  * the nested-quantization-layer fake-quant op and its hand-written backward
      /root/reference/MNIST/nested_quantization_layer/custom_components/custom_layers.py:49-120
      (CIFAR-10 / IMAGENETTE copies are identical apart from line 98)
  * the STE-only variant of the same op
      /root/reference/CIFAR-10/custom_loss_terms/custom_components/custom_layers.py:49-64
  * MinValueConstraint                      custom_layers.py:35-46
  * the scale-shape rule of CustomQuantizedScaleLayer.build   custom_layers.py:147-197
  * the three penalties and compute_total_loss
      /root/reference/CIFAR-10/custom_loss_terms/custom_components/custom_loss_functions.py:47-116,161-195,240-275
  * the integer view / export cast
      /root/reference/CIFAR-10/nested_quantization_layer/utils/log_scripts.py:72-79
      /root/reference/CIFAR-10/nested_quantization_layer/custom_components/custom_callbacks.py:85-129

Only ``tests/``, ``__graft_entry__.smoke()`` and the ``cpu_baseline`` leg of
``bench.py`` may import it, and only as the *checker*.  Nothing under
``learned_quantization_amd/`` imports it; the product path has no CPU fallback.

PARITY PINNING STATUS: **parity unpinned by upstream** -- the reference ships no
tests, golden vectors or fixtures for this path, and its arithmetic lives in a
third-party dependency that is absent here (TensorFlow 2.11.0, pinned at
/root/reference/requirements.txt:1; not installed, no network).  What pins this
restatement instead (see tests/test_oracle.py):
  1. the hand-derived known-answer vector of SURVEY.md section 8(c)
     (tests/golden/kat_survey.json -- values typed in from the survey, not
     produced by this file),
  2. an independent float64 restatement of the thesis math
     (oracle/lq_oracle_f64.py, thesis/chapters/chapter3.tex:30-37,84-167,242-337),
  3. an independent scalar C restatement (oracle/lq_oracle.c),
  4. the source-implied invariants of SURVEY.md section 4.

TensorFlow semantics that matter and how they are restated:
  * ``/`` on float32 tensors = IEEE-754 round-to-nearest division (np.divide on
    float32 arrays is the same operation); ``tf.floor`` exact; ``*`` IEEE RN.
  * ``tf.where(cond, python_float, tensor)`` converts the Python float to the
    tensor dtype (float32).
  * ``tf.math.tanh`` on float32: Eigen's rational approximation on CPU, CUDA
    ``tanhf`` on GPU; both within a few ulp of the correctly rounded value and
    both return ``x`` for |x| < 4e-4 (to fp32 precision tanh(x) == x there).
    NumPy's float32 tanh is used here; comparisons of tanh-derived quantities
    use rtol 1e-5, everything else is compared bit for bit.
  * ``tf.reduce_mean`` = float32 sum / count with unspecified summation order;
    here: pairwise float32 summation (np.sum) / count.  Compared with rtol 1e-5.
  * ``tf.reduce_max`` / ``tf.reduce_all``: exact, order independent.
  * ``tf.reduce_max`` gradient splits evenly across ties (math_grad._MinOrMaxGrad).
"""
from __future__ import annotations

import numpy as np

F32 = np.float32
#: custom_layers.py:11  eps_float32 = np.finfo(np.float32).eps
EPS_F32 = np.finfo(np.float32).eps            # 1.1920929e-07
#: custom_layers.py:156,158  initial value and lower bound of every scale
SCALE_INIT = F32(EPS_F32 * 100)               # 1.1920929e-05
SCALE_MIN = F32(EPS_F32 * 100)

ORIENTATIONS = ("rowwise", "columnwise", "channelwise", "scalar")


def _f32(x):
    return np.asarray(x, dtype=F32)


# --------------------------------------------------------------------------- #
#  CustomQuantizedScaleLayer.build  (custom_layers.py:147-197)
# --------------------------------------------------------------------------- #
def scale_shape(input_shape, orientation):
    """Shape of the trainable scale for a parameter of ``input_shape``.

    rowwise -> axis 0 kept (custom_layers.py:149-152), columnwise -> axis 1
    (:162-164), channelwise -> axis 2 (:174-176), scalar -> (1,) (:186-189);
    anything else raises ValueError (:194-197).
    """
    input_shape = tuple(int(d) for d in input_shape)
    n = len(input_shape)
    if orientation == "rowwise":
        return tuple(input_shape[i] if i == 0 else 1 for i in range(n))
    if orientation == "columnwise":
        return tuple(input_shape[i] if i == 1 else 1 for i in range(n))
    if orientation == "channelwise":
        return tuple(input_shape[i] if i == 2 else 1 for i in range(n))
    if orientation == "scalar":
        return (1,)
    raise ValueError(
        f"Invalid scaler application: {orientation}. Expected rowwise, columnwise or scalar."
    )


def min_value_constraint(w, min_value=SCALE_MIN):
    """MinValueConstraint.__call__: tf.maximum(w, min_value)  (custom_layers.py:42-43)."""
    return np.maximum(_f32(w), F32(min_value))


# --------------------------------------------------------------------------- #
#  my_custom_gradient forward  (custom_layers.py:55-60)
# --------------------------------------------------------------------------- #
def fq_forward(parameter, scale):
    """Returns (q, out): q = floor(P / s) (integer-valued float32), out = q * s."""
    parameter = _f32(parameter)
    scale = _f32(scale)
    nonrounded = parameter / scale              # :56-58  IEEE RN float32 divide, broadcast
    rounded = np.floor(nonrounded)              # :59
    scaled_back = rounded * scale               # :60
    return rounded.astype(F32), scaled_back.astype(F32)


def quantized_integers(parameter, scale):
    """floor(P/s) as used by callbacks/export (custom_callbacks.py:85-87, log_scripts.py:74-79)."""
    return np.floor(_f32(parameter) / _f32(scale))


def export_int8(parameter, scale):
    """log_scripts.py:74-79: np.floor(k / s).astype(np.int8) -- wraps modulo 256 like a C cast.

    NumPy >= 2 raises/undefined-warns on out-of-range float->int8; the reference
    environment (NumPy 1.x on x86-64) goes float -> int32/64 -> truncation.  Restated
    explicitly so the result is defined: two's-complement wrap of the integer value.
    """
    q = np.floor(_f32(parameter) / _f32(scale)).astype(np.float64)
    q = np.where(np.isfinite(q), q, 0.0)
    return (q.astype(np.int64) & 0xFF).astype(np.uint8).view(np.int8)


# --------------------------------------------------------------------------- #
#  custom_grad, nested-quantization variant  (custom_layers.py:62-118)
# --------------------------------------------------------------------------- #
def _scale_grads_elementwise(ratio, all_above_broad, lam):
    """The nested tf.where of custom_layers.py:77-85 / 103-111."""
    lam = F32(lam)
    inner = np.where(
        ratio >= lam,
        F32(0.0),
        F32(-1.0) * np.abs(np.tanh((lam - ratio).astype(F32))).astype(F32),
    ).astype(F32)
    const = F32(-1.0) * np.abs(np.tanh(lam)).astype(F32)
    return np.where(all_above_broad, const, inner).astype(F32)


def nq_backward(parameter, scale, penalty_threshold, dy, return_intermediates=False):
    """custom_grad(dy) of the NQ op.  Returns (dP, ds) -- dP *is* dy (STE, :118)."""
    parameter = _f32(parameter)
    scale = _f32(scale)
    dy = _f32(dy)
    lam = F32(penalty_threshold)                                  # :55 (python float -> f32)
    rounded, scaled_back = fq_forward(parameter, scale)

    with np.errstate(divide="ignore", invalid="ignore", over="ignore"):
        non_zero_param = np.where(scaled_back == F32(0.0), F32(EPS_F32), scaled_back)  # :63
        ratio = (np.abs(dy) / np.abs(non_zero_param)).astype(F32)                      # :64

        if scale.ndim == 1:                                        # :67 bias / scalar scale
            maxvalue = np.max(np.abs(rounded))                     # :68
            all_above = np.all(ratio >= lam)                       # :70
            all_above_broad = np.broadcast_to(all_above, parameter.shape)   # :71-73
            sg = _scale_grads_elementwise(ratio, all_above_broad, lam)      # :77-85
            reduced = (np.sum(sg, dtype=F32) / F32(sg.size)).astype(F32)    # :87 reduce_mean
            reduced = np.reshape(reduced, scale.shape)             # :88
            maxvalue = F32(maxvalue)
        else:
            axes = tuple(i for i in range(scale.ndim) if scale.shape[i] == 1)   # :92
            maxvalue = np.max(np.abs(rounded), axis=axes).reshape(scale.shape)  # :94-95
            all_above = np.all(ratio >= lam, axis=axes)                         # :97
            all_above_broad = np.broadcast_to(all_above.reshape(scale.shape), parameter.shape)  # :98-99
            sg = _scale_grads_elementwise(ratio, all_above_broad, lam)          # :103-111
            count = 1
            for a in axes:
                count *= parameter.shape[a]
            # pairwise float32 summation needs the reduced axes contiguous and last: np.sum over a leading axis of a
            # C-ordered array accumulates row after row (error ~ N*eps, 2e-3 at N = 350 k), which is not a property
            # of the reference (TF's reduction order is unspecified) but of that NumPy loop
            kept = [i for i in range(scale.ndim) if scale.shape[i] != 1]
            sg_g = np.ascontiguousarray(np.moveaxis(sg, kept, list(range(len(kept)))))
            sg_g = sg_g.reshape(tuple(parameter.shape[i] for i in kept) + (-1,))
            reduced = (np.sum(sg_g, axis=-1, dtype=F32) / F32(count)).astype(F32)  # :113
            reduced = reduced.reshape(scale.shape)                               # :114

        ds = (reduced * maxvalue).astype(F32)                      # :116
    if return_intermediates:
        return dy, ds, dict(q=rounded, out=scaled_back, ratio=ratio, sg=sg,
                            maxvalue=np.asarray(maxvalue, F32), mean=np.asarray(reduced, F32))
    return dy, ds                                                  # :118


def ste_backward(parameter, scale, dy):
    """custom_grad of the STE-only op: (dy, zeros_like(scale))  (CL custom_layers.py:61-62)."""
    return _f32(dy), np.zeros_like(_f32(scale))


# --------------------------------------------------------------------------- #
#  (outer, G, inner) descriptor used by the C ABI (include/lq_hip.h)
# --------------------------------------------------------------------------- #
def group_descriptor(param_shape, scale_shape_):
    """Element i of a C-contiguous parameter belongs to scale element (i // inner) % G.

    A scale has at most one non-unit axis (custom_layers.py:147-192); a (1,) scale
    or an all-ones shape is G = 1.
    """
    param_shape = tuple(int(d) for d in param_shape)
    scale_shape_ = tuple(int(d) for d in scale_shape_)
    numel = int(np.prod(param_shape)) if param_shape else 1
    if len(scale_shape_) == 1 and scale_shape_[0] == 1:
        return 1, 1, numel
    if len(scale_shape_) != len(param_shape):
        raise ValueError("scale rank must equal parameter rank (or be (1,))")
    axes = [i for i, d in enumerate(scale_shape_) if d != 1]
    if not axes:
        return 1, 1, numel
    if len(axes) > 1:
        raise ValueError("scale may have at most one non-unit axis")
    a = axes[0]
    if scale_shape_[a] != param_shape[a]:
        raise ValueError("scale axis length must match parameter axis length")
    outer = int(np.prod(param_shape[:a])) if a > 0 else 1
    inner = int(np.prod(param_shape[a + 1:])) if a + 1 < len(param_shape) else 1
    return outer, param_shape[a], inner


# --------------------------------------------------------------------------- #
#  custom loss terms  (custom_loss_functions.py)
#  ``layers`` is a sequence of (kernel, kernel_scale, b, b_scale) float32 arrays.
# --------------------------------------------------------------------------- #
def _dim(a):
    d = 1.0
    for n in a.shape:                                              # :102-108
        d *= n
    return d


def _maxbin(p, s):
    axes = [i for i in range(s.ndim) if s.shape[i] == 1 and s.ndim > 1]   # :90, :96
    t = (np.abs(p) / s).astype(F32)
    if axes != []:
        return np.max(t, axis=tuple(axes))                         # :92
    return np.max(t)                                               # :94


def _mean_f32(a):
    a = _f32(a)
    return F32(np.sum(a, dtype=F32) / F32(a.size))


def maxbin_penalty(layers):
    """SCCEMaxBin.compute_maxbin_penalty  (custom_loss_functions.py:75-116)."""
    total = F32(0.0)
    normalizer = 0.0
    for k, ks, b, bs in layers:
        k, ks, b, bs = _f32(k), _f32(ks), _f32(b), _f32(bs)
        k_dim, b_dim = _dim(k), _dim(b)
        layer_penalty = _mean_f32(_maxbin(k, ks)) * F32(k_dim) + _mean_f32(_maxbin(b, bs)) * F32(b_dim)  # :110
        total = F32(total + layer_penalty)                         # :112
        normalizer += k_dim + b_dim                                # :114
    return F32(total / F32(normalizer))                            # :116


def difference_penalty(layers):
    """SCCEDifference.compute_difference_penalty  (custom_loss_functions.py:161-195)."""
    total = F32(0.0)
    normalizer = 0.0
    for k, ks, b, bs in layers:
        k, ks, b, bs = _f32(k), _f32(ks), _f32(b), _f32(bs)
        kq = (k / ks).astype(F32)                                  # :172
        bq = (b / bs).astype(F32)                                  # :173
        k_pen = _mean_f32(np.abs(k - kq))                          # :175
        b_pen = _mean_f32(np.abs(b - bq))                          # :176
        k_dim, b_dim = _dim(k), _dim(b)
        total = F32(total + (k_pen * F32(k_dim) + b_pen * F32(b_dim)))   # :186-188
        normalizer += k_dim + b_dim
    return F32(total / F32(normalizer))                            # :195


def inverse_penalty(layers):
    """SCCEInverse.compute_inverse_penalty  (custom_loss_functions.py:240-275)."""
    total = F32(0.0)
    normalizer = 0.0
    for k, ks, b, bs in layers:
        k, ks, b, bs = _f32(k), _f32(ks), _f32(b), _f32(bs)
        ks_nz = np.where(ks == F32(0.0), F32(EPS_F32), ks)         # :252
        bs_nz = np.where(bs == F32(0.0), F32(EPS_F32), bs)         # :253
        k_inv = _mean_f32(F32(1.0) / ks_nz)                        # :255
        b_inv = _mean_f32(F32(1.0) / bs_nz)                        # :256
        k_dim, b_dim = _dim(k), _dim(b)
        total = F32(total + (k_inv * F32(k_dim) + b_inv * F32(b_dim)))   # :266-268
        normalizer += k_dim + b_dim
    return F32(total / F32(normalizer))                            # :275


def sparse_categorical_crossentropy(y_true, y_pred):
    """tf.keras.losses.sparse_categorical_crossentropy(y_true, y_pred), from_logits=False.

    Keras 2.11 backend: probabilities are clipped to [1e-7, 1 - 1e-7] and the
    per-sample loss is -log(p[y])  (third-party; keras/backend.py
    sparse_categorical_crossentropy, epsilon() = 1e-7).  Returns shape (B,).
    """
    y_pred = _f32(y_pred)
    y_true = np.asarray(y_true).astype(np.int64).reshape(-1)
    eps = F32(1e-7)
    p = np.clip(y_pred, eps, F32(1.0) - eps)
    return (-np.log(p[np.arange(p.shape[0]), y_true])).astype(F32)


def total_loss(y_true, y_pred, penalty_rate, penalty):
    """compute_total_loss: SCCE (B,) + penalty_rate * penalty (scalar)  (custom_loss_functions.py:52-58)."""
    ce = sparse_categorical_crossentropy(y_true, y_pred)
    return (ce + F32(penalty_rate) * F32(penalty)).astype(F32)


# --------------------------------------------------------------------------- #
#  analytic float32 gradients of the penalties (what TF autodiff produces)
#    d|x| = sign(x); d(x/y) = (g/y, -g*x/y/y); reduce_max: even split over ties;
#    reduce_mean: g / N.   Returned per tensor for an upstream scalar ``c`` that
#    multiplies the *tensor term*  mean(.)  (i.e. c = gamma * dim / normalizer).
# --------------------------------------------------------------------------- #
def maxbin_term_grads(p, s, c):
    p, s = _f32(p), _f32(s)
    full = np.broadcast_to(s, p.shape) if s.ndim == p.ndim else np.broadcast_to(s.reshape((1,) * p.ndim), p.shape)
    t = (np.abs(p) / full).astype(F32)
    axes = tuple(i for i in range(s.ndim) if s.shape[i] == 1 and s.ndim > 1)
    if axes:
        m = np.max(t, axis=axes, keepdims=True)
        n_groups = m.size
    else:
        m = np.max(t).reshape((1,) * p.ndim)
        n_groups = 1
    ind = (t == m)
    cnt = np.sum(ind, axis=axes if axes else None, keepdims=True).astype(F32)
    g = (ind.astype(F32) / cnt) * F32(c) / F32(n_groups)           # grad wrt t
    dp = (g / full * np.sign(p)).astype(F32)
    ds_full = (-g * t / full).astype(F32)
    if axes:
        ds = np.sum(ds_full, axis=axes, keepdims=True, dtype=F32).reshape(s.shape)
    else:
        ds = np.sum(ds_full, dtype=F32).reshape(s.shape)
    return dp, ds.astype(F32)


def difference_term_grads(p, s, c):
    p, s = _f32(p), _f32(s)
    full = np.broadcast_to(s, p.shape) if s.ndim == p.ndim else np.broadcast_to(s.reshape((1,) * p.ndim), p.shape)
    pq = (p / full).astype(F32)
    u = (p - pq).astype(F32)
    g = np.sign(u).astype(F32) * F32(c) / F32(p.size)              # grad wrt u
    dp = (g - g / full).astype(F32)
    ds_full = (g * pq / full).astype(F32)                          # -(-g) * (p/s)/s
    axes = tuple(i for i in range(s.ndim) if s.shape[i] == 1) if s.ndim == p.ndim else None
    if s.ndim == p.ndim and s.size > 1:
        ds = np.sum(ds_full, axis=axes, keepdims=True, dtype=F32).reshape(s.shape)
    else:
        ds = np.sum(ds_full, dtype=F32).reshape(s.shape)
    return dp, ds.astype(F32)


def inverse_term_grads(s, c):
    s = _f32(s)
    s_nz = np.where(s == F32(0.0), F32(EPS_F32), s)
    ds = (-(F32(c) / F32(s.size)) / s_nz / s_nz).astype(F32)
    ds = np.where(s == F32(0.0), F32(0.0), ds)                     # where() routes no gradient to s when s == 0
    return ds.astype(F32)


# --------------------------------------------------------------------------- #
#  Optimizer step for the scales: Keras 2.11 Adam + MinValueConstraint
#  (third-party keras/optimizers/optimizer_experimental/adam.py update_step,
#   constraint applied after the update: custom_layers.py:158)
# --------------------------------------------------------------------------- #
def keras_adam_step(var, grad, m, v, step, lr=1e-4, beta_1=0.9, beta_2=0.999, epsilon=1e-7,
                    min_value=None):
    """One Adam step in float32 as Keras 2.11 does it; ``step`` is 1-based.  Returns (var, m, v)."""
    var, grad, m, v = _f32(var), _f32(grad), _f32(m), _f32(v)
    b1p = F32(np.power(F32(beta_1), F32(step)))
    b2p = F32(np.power(F32(beta_2), F32(step)))
    alpha = F32(F32(lr) * np.sqrt(F32(1.0) - b2p) / (F32(1.0) - b1p))
    m = (m + (grad - m) * F32(1.0 - beta_1)).astype(F32)
    v = (v + (grad * grad - v) * F32(1.0 - beta_2)).astype(F32)
    var = (var - (m * alpha) / (np.sqrt(v) + F32(epsilon))).astype(F32)
    if min_value is not None:
        var = np.maximum(var, F32(min_value))
    return var, m, v
