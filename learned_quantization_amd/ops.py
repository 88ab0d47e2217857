"""Operators of the learned-quantization hot path on top of the C ABI (include/lq_hip.h).

``my_custom_gradient`` keeps the reference's name, argument order and meaning:
  * 3-argument form ``(parameter, scale, penalty_threshold)`` = nested-quantization op
    /root/reference/MNIST/nested_quantization_layer/custom_components/custom_layers.py:49-120
  * 2-argument form ``(parameter, scale)`` = STE-only op
    /root/reference/CIFAR-10/custom_loss_terms/custom_components/custom_layers.py:49-64

Raw (non-autograd) wrappers ``fq_forward``, ``fq_scale_grad``, ``fq_fwd_bwd_fused``,
``quantized_integers`` ... are thin: argument checking + one C-ABI call each.
Everything runs on the HIP device; there is no CPU fallback.
"""
from __future__ import annotations

from typing import Optional, Tuple

import torch

from . import _hip
from .descriptor import group_descriptor, memory_descriptor

_QDTYPES = {
    torch.float32: _hip.LQ_Q_F32,
    torch.int32: _hip.LQ_Q_I32,
    torch.int8: _hip.LQ_Q_I8,
}


def _desc(parameter: torch.Tensor, scale: torch.Tensor) -> Tuple[int, int, int]:
    return group_descriptor(tuple(parameter.shape), tuple(scale.shape))


def _param(parameter: torch.Tensor, scale: torch.Tensor):
    """(P, s, descriptor) as the kernels read them.  A parameter whose memory is a dense permutation of its logical axes -- a
    conv kernel shaped HWIO (custom_layers.py:321) and stored OIHW, layers.py ``kernel_storage`` -- is NOT copied: the groups
    are described in memory order, and every same-shaped output is allocated with the parameter's strides, so that logically
    (index by index) the results are those of the contiguous tensor."""
    p = _hip.require_device_f32(parameter, "parameter", dense_ok=True)
    s = _hip.require_device_f32(scale, "scale")
    return p, s, memory_descriptor(tuple(p.shape), p.stride(), tuple(s.shape))


# --------------------------------------------------------------------------- raw wrappers
def fq_forward(parameter: torch.Tensor, scale: torch.Tensor, q_dtype: Optional[torch.dtype] = None,
               want_out: bool = True):
    """K1.  Returns ``out`` (and ``q`` when ``q_dtype`` is given).  custom_layers.py:55-60."""
    lib = _hip.load()
    p, s, (outer, G, inner) = _param(parameter, scale)
    out = torch.empty_like(p) if want_out else None
    q = None
    qd = _hip.LQ_Q_NONE
    if q_dtype is not None:
        if q_dtype not in _QDTYPES:
            raise TypeError(f"q_dtype must be one of {list(_QDTYPES)}, got {q_dtype}")
        q = torch.empty_like(p, dtype=q_dtype)
        qd = _QDTYPES[q_dtype]
    if out is None and q is None:
        raise ValueError("nothing to compute: want_out=False and q_dtype=None")
    _hip.check(lib.lq_fq_forward(_hip.ptr(p), _hip.ptr(s), _hip.ptr(out), _hip.ptr(q), qd,
                                 outer, G, inner, _hip.stream_ptr(p.device)), "lq_fq_forward")
    if q is None:
        return out
    return (out, q) if want_out else q


def quantized_integers(parameter: torch.Tensor, scale: torch.Tensor, dtype: torch.dtype = torch.float32):
    """floor(P/s): the integer view of callbacks/export (custom_callbacks.py:85-87, log_scripts.py:74-79)."""
    return fq_forward(parameter, scale, q_dtype=dtype, want_out=False)


def fq_scale_grad(parameter: torch.Tensor, scale: torch.Tensor, dy: torch.Tensor, penalty_threshold: float,
                  return_parts: bool = False):
    """K2+K3: the hand-written scale gradient of custom_layers.py:62-118.  Returns ds (shape of scale)."""
    lib = _hip.load()
    p, s, (outer, G, inner) = _param(parameter, scale)
    d = _hip.require_device_f32(dy, "dy", like=p)
    ds = torch.empty_like(s)
    parts = torch.empty(3 * G, dtype=torch.float32, device=p.device) if return_parts else None
    ws = _hip.workspace_for(p.device, outer, G, inner)
    _hip.check(lib.lq_fq_scale_grad(_hip.ptr(p), _hip.ptr(s), _hip.ptr(d), float(penalty_threshold),
                                    _hip.ptr(ds), _hip.ptr(parts), _hip.ptr(ws), ws.numel(),
                                    outer, G, inner, _hip.stream_ptr(p.device)), "lq_fq_scale_grad")
    if return_parts:
        return ds, parts.view(3, G)
    return ds


def fq_fwd_bwd_fused(parameter: torch.Tensor, scale: torch.Tensor, dy: torch.Tensor, penalty_threshold: float,
                     out: Optional[torch.Tensor] = None, ds: Optional[torch.Tensor] = None):
    """K4: forward and NQ backward in one pass over P (benchmark path).  Returns (out, ds)."""
    lib = _hip.load()
    p, s, (outer, G, inner) = _param(parameter, scale)
    d = _hip.require_device_f32(dy, "dy", like=p)
    if out is None:
        out = torch.empty_like(p)
    elif not _hip.same_layout(out, p):
        raise ValueError("out must have the parameter's shape and strides")
    if ds is None:
        ds = torch.empty_like(s)
    ws = _hip.workspace_for(p.device, outer, G, inner)
    _hip.check(lib.lq_fq_fwd_bwd_fused(_hip.ptr(p), _hip.ptr(s), _hip.ptr(d), float(penalty_threshold),
                                       _hip.ptr(out), _hip.ptr(ds), _hip.ptr(ws), ws.numel(),
                                       outer, G, inner, _hip.stream_ptr(p.device)), "lq_fq_fwd_bwd_fused")
    return out, ds


def q_absmax_over_axis(parameter: torch.Tensor, scale: torch.Tensor, axis: int) -> torch.Tensor:
    """max |floor(P/s)| reduced over ``axis`` (custom_callbacks.py:98-99 uses axis=1)."""
    lib = _hip.load()
    p = _hip.require_device_f32(parameter, "parameter")
    s = _hip.require_device_f32(scale, "scale")
    outer, G, inner = _desc(p, s)
    shape = tuple(p.shape)
    axis = axis % len(shape)
    pre = 1
    for d in shape[:axis]:
        pre *= d
    post = 1
    for d in shape[axis + 1:]:
        post *= d
    res = torch.empty(shape[:axis] + shape[axis + 1:], dtype=torch.float32, device=p.device)
    _hip.check(lib.lq_q_absmax_over_axis(_hip.ptr(p), _hip.ptr(s), _hip.ptr(res), pre, shape[axis], post,
                                         outer, G, inner, _hip.stream_ptr(p.device)), "lq_q_absmax_over_axis")
    return res


_MAX_BINS = 1 << 26   # 256 MiB of uint32 bins; beyond that the integers are not "a few quantisation levels"


def q_unique(parameter: torch.Tensor, scale: torch.Tensor):
    """np.unique(floor(P/s), return_counts=True) of the callbacks (custom_callbacks.py:92, 142) computed on the
    device: range by lq_q_minmax, counts by lq_q_histogram.  Returns (values int32, counts int64), both on the device.
    One small device->host read (the range) is needed to size the histogram -- this is a per-epoch statistic."""
    lib = _hip.load()
    p = _hip.require_device_f32(parameter, "parameter")
    s = _hip.require_device_f32(scale, "scale")
    outer, G, inner = _desc(p, s)
    st = _hip.stream_ptr(p.device)
    mm = torch.tensor([2 ** 31 - 1, -(2 ** 31)], dtype=torch.int32, device=p.device)
    _hip.check(lib.lq_q_minmax(_hip.ptr(p), _hip.ptr(s), _hip.ptr(mm), outer, G, inner, st), "lq_q_minmax")
    lo, hi = (int(v) for v in mm.tolist())
    if lo > hi:                                            # nothing countable (all NaN/Inf)
        return (torch.empty(0, dtype=torch.int32, device=p.device), torch.empty(0, dtype=torch.int64, device=p.device))
    nbins = hi - lo + 1
    if nbins > _MAX_BINS:
        raise ValueError(f"integer range [{lo}, {hi}] too wide for a histogram ({nbins} bins)")
    bins = torch.zeros(nbins, dtype=torch.int32, device=p.device)
    _hip.check(lib.lq_q_histogram(_hip.ptr(p), _hip.ptr(s), lo, nbins, _hip.ptr(bins), outer, G, inner, st), "lq_q_histogram")
    nz = torch.nonzero(bins, as_tuple=False).flatten()
    return (nz + lo).to(torch.int32), bins[nz].to(torch.int64)


def min_value_project_(w: torch.Tensor, min_value: float) -> torch.Tensor:
    """In-place MinValueConstraint: w <- max(w, min_value)  (custom_layers.py:42-43)."""
    lib = _hip.load()
    if not w.is_contiguous():
        raise ValueError("min_value_project_ needs a contiguous tensor (in-place)")
    _hip.require_device_f32(w, "w")
    _hip.check(lib.lq_min_value_project(_hip.ptr(w), w.numel(), float(min_value), _hip.stream_ptr(w.device)),
               "lq_min_value_project")
    return w


def scale_adam_step_(scale: torch.Tensor, grad: torch.Tensor, m: torch.Tensor, v: torch.Tensor, step: int,
                     lr: float = 1e-4, beta1: float = 0.9, beta2: float = 0.999, eps: float = 1e-7,
                     min_value: float = 0.0, mode: str = "keras") -> None:
    """K6: Adam + MinValueConstraint projection in one launch (custom_layers.py:158; Keras 2.11 Adam)."""
    lib = _hip.load()
    for name, t in (("scale", scale), ("grad", grad), ("m", m), ("v", v)):
        _hip.require_device_f32(t, name)
        if not t.is_contiguous() or t.numel() != scale.numel():
            raise ValueError(f"{name} must be contiguous with {scale.numel()} elements")
    md = {"keras": _hip.LQ_ADAM_KERAS, "torch": _hip.LQ_ADAM_TORCH}[mode]
    _hip.check(lib.lq_scale_adam_step(_hip.ptr(scale), _hip.ptr(grad), _hip.ptr(m), _hip.ptr(v), scale.numel(),
                                      lr, beta1, beta2, eps, int(step), float(min_value), md,
                                      _hip.stream_ptr(scale.device)), "lq_scale_adam_step")


def scale_adam_step_dev_(scale: torch.Tensor, grad: torch.Tensor, m: torch.Tensor, v: torch.Tensor, step_dev: torch.Tensor,
                         lr: float = 1e-4, beta1: float = 0.9, beta2: float = 0.999, eps: float = 1e-7,
                         min_value: float = 0.0, mode: str = "keras") -> None:
    """K6, hipGraph-capturable form: ``step_dev`` is a 1-element int64 device tensor holding the 1-based step."""
    lib = _hip.load()
    for name, t in (("scale", scale), ("grad", grad), ("m", m), ("v", v)):
        _hip.require_device_f32(t, name)
        if not t.is_contiguous() or t.numel() != scale.numel():
            raise ValueError(f"{name} must be contiguous with {scale.numel()} elements")
    if step_dev.dtype != torch.int64 or not step_dev.is_cuda or step_dev.numel() != 1:
        raise TypeError("step_dev must be a 1-element int64 device tensor")
    md = {"keras": _hip.LQ_ADAM_KERAS, "torch": _hip.LQ_ADAM_TORCH}[mode]
    _hip.check(lib.lq_scale_adam_step_dev(_hip.ptr(scale), _hip.ptr(grad), _hip.ptr(m), _hip.ptr(v), scale.numel(),
                                          lr, beta1, beta2, eps, _hip.ptr(step_dev), float(min_value), md,
                                          _hip.stream_ptr(scale.device)), "lq_scale_adam_step_dev")


# --------------------------------------------------------------------------- autograd ops
class _NestedQuantFn(torch.autograd.Function):
    """custom_layers.py:49-120 -- forward K1, backward (dy, K2+K3, None)."""

    @staticmethod
    def forward(ctx, parameter, scale, penalty_threshold, defer_scale_grad=False):
        ctx.save_for_backward(parameter, scale)
        ctx.penalty_threshold = float(penalty_threshold)
        ctx.defer = bool(defer_scale_grad)
        return fq_forward(parameter, scale)

    @staticmethod
    def backward(ctx, dy):
        parameter, scale = ctx.saved_tensors
        ds = None
        # defer: exact data-parallel mode recomputes ds from the all-reduced dP after the exchange (ddp.py, mode B)
        if ctx.needs_input_grad[1] and not ctx.defer:
            ds = fq_scale_grad(parameter, scale, dy, ctx.penalty_threshold)
        return (dy if ctx.needs_input_grad[0] else None), ds, None, None      # :118  dP is dy itself (STE)


def _conv_extents(shape):
    kh, kw, ci, co = (int(d) for d in shape)
    return kh * kw, ci, co


def fq_forward_oihw(kernel: torch.Tensor, scale: torch.Tensor, hwio_out: bool = True):
    """K1 on an HWIO conv kernel (custom_layers.py:321, 340) that also emits the OIHW tensor MIOpen consumes: returns
    (out_hwio, out_oihw) with out_oihw == out_hwio.permute(3, 2, 0, 1) bit for bit, in ONE launch (no transpose kernel).
    ``hwio_out=False``: where the LDS-tile kernel takes the tensor, only the OIHW tensor is written (8 bytes per element
    instead of 12) and out_hwio is the permuted view of it."""
    lib = _hip.load()
    p = _hip.require_device_f32(kernel, "kernel")
    s = _hip.require_device_f32(scale, "scale")
    if p.dim() != 4:
        raise ValueError("fq_forward_oihw needs an HWIO conv kernel (kh, kw, ci, co)")
    hw, ci, co = _conv_extents(p.shape)
    outer, G, inner = _desc(p, s)
    companion_only = (not hwio_out) and p.data_ptr() % 16 == 0 and lib.lq_conv_tile_supported(hw, ci, co, outer, G, inner) == 1
    out = None if companion_only else torch.empty_like(p)
    out_oihw = torch.empty((co, ci, p.shape[0], p.shape[1]), dtype=torch.float32, device=p.device)
    _hip.check(lib.lq_fq_forward_oihw(_hip.ptr(p), _hip.ptr(s), _hip.ptr(out), _hip.ptr(out_oihw), hw, ci, co,
                                      outer, G, inner, _hip.stream_ptr(p.device)), "lq_fq_forward_oihw")
    return (out if out is not None else out_oihw.permute(2, 3, 1, 0)), out_oihw


def fq_scale_grad_oihw(kernel: torch.Tensor, scale: torch.Tensor, dy_oihw: torch.Tensor, penalty_threshold: float):
    """K2+K3 with the upstream gradient in OIHW order (MIOpen's weight gradient as it stands): returns (ds, dP_hwio);
    ds is bit-identical to ``fq_scale_grad(kernel, scale, dy_oihw.permute(2, 3, 1, 0))``, dP is that permuted dy."""
    lib = _hip.load()
    p = _hip.require_device_f32(kernel, "kernel")
    s = _hip.require_device_f32(scale, "scale")
    d = _hip.require_device_f32(dy_oihw, "dy_oihw")
    hw, ci, co = _conv_extents(p.shape)
    if tuple(d.shape) != (co, ci, p.shape[0], p.shape[1]):
        raise ValueError(f"dy_oihw shape {tuple(d.shape)} does not match the kernel {tuple(p.shape)}")
    outer, G, inner = _desc(p, s)
    ds = torch.empty_like(s)
    dP = torch.empty_like(p)
    ws = _hip.workspace(p.device, lib.lq_conv_workspace_bytes(hw, ci, co, outer, G, inner))
    _hip.check(lib.lq_fq_scale_grad_oihw(_hip.ptr(p), _hip.ptr(s), _hip.ptr(d), float(penalty_threshold), _hip.ptr(ds),
                                         _hip.ptr(dP), _hip.ptr(ws), ws.numel(), hw, ci, co, outer, G, inner,
                                         _hip.stream_ptr(p.device)), "lq_fq_scale_grad_oihw")
    return ds, dP


class _NestedQuantConvFn(torch.autograd.Function):
    """The nested-quantization op on an HWIO conv kernel, handing MIOpen its OIHW tensor directly: forward K1 with the OIHW
    companion store, backward K2+K3 reading MIOpen's OIHW weight gradient and returning dP in HWIO order.  Same q, out and
    ds as ``_NestedQuantFn`` bit for bit (custom_layers.py:49-120, 338-350); two transposition launches fewer per layer."""

    @staticmethod
    def forward(ctx, kernel, scale, penalty_threshold, defer_scale_grad=False):
        ctx.save_for_backward(kernel, scale)
        ctx.penalty_threshold = float(penalty_threshold)
        ctx.defer = bool(defer_scale_grad)
        return fq_forward_oihw(kernel, scale, hwio_out=False)[1]

    @staticmethod
    def backward(ctx, dy_oihw):
        kernel, scale = ctx.saved_tensors
        if ctx.defer or not ctx.needs_input_grad[1]:
            return (dy_oihw.permute(2, 3, 1, 0) if ctx.needs_input_grad[0] else None), None, None, None
        ds, dP = fq_scale_grad_oihw(kernel, scale, dy_oihw, ctx.penalty_threshold)
        return (dP if ctx.needs_input_grad[0] else None), ds, None, None


def my_custom_gradient_oihw(kernel, scale, penalty_threshold, *, defer_scale_grad=False):
    """``my_custom_gradient`` for an HWIO conv kernel whose consumer wants OIHW: returns the fake-quantised kernel in OIHW
    layout (co, ci, kh, kw), contiguous.  Gradients flow back to the HWIO parameter."""
    if isinstance(penalty_threshold, torch.Tensor):
        penalty_threshold = float(penalty_threshold)
    return _NestedQuantConvFn.apply(kernel, scale, penalty_threshold, defer_scale_grad)


class _STEQuantFn(torch.autograd.Function):
    """CL custom_layers.py:49-64 -- forward K1, backward (dy, zeros_like(scale))."""

    @staticmethod
    def forward(ctx, parameter, scale):
        ctx.save_for_backward(scale)
        return fq_forward(parameter, scale)

    @staticmethod
    def backward(ctx, dy):
        (scale,) = ctx.saved_tensors
        ds = torch.zeros_like(scale) if ctx.needs_input_grad[1] else None    # :62
        return (dy if ctx.needs_input_grad[0] else None), ds


def my_custom_gradient(parameter, scale, penalty_threshold=None, *, defer_scale_grad=False):
    """The reference op.  With ``penalty_threshold`` -> nested-quantization variant
    (custom_layers.py:49-120); without -> STE-only variant (CL custom_layers.py:49-64).
    ``defer_scale_grad`` (not in the reference): backward returns dP only; the caller computes ds later from the
    all-reduced dP (exact data-parallel mode, ddp.py)."""
    if penalty_threshold is None:
        return _STEQuantFn.apply(parameter, scale)
    if isinstance(penalty_threshold, torch.Tensor):
        penalty_threshold = float(penalty_threshold)      # tf.stop_gradient(penalty_threshold), :55
    return _NestedQuantFn.apply(parameter, scale, penalty_threshold, defer_scale_grad)


# --------------------------------------------------------------------------- penalty terms
def _up(grad_out: torch.Tensor) -> torch.Tensor:
    g = grad_out.reshape(1)
    return _hip.require_device_f32(g, "upstream gradient")


class _MaxBinTerm(torch.autograd.Function):
    """mean_g max_{i in g} |P_i|/s_g  (custom_loss_functions.py:90-100,110) -- K5a."""

    @staticmethod
    def forward(ctx, parameter, scale):
        lib = _hip.load()
        p, s, (outer, G, inner) = _param(parameter, scale)
        mb = torch.empty(G, dtype=torch.float32, device=p.device)
        ties = torch.empty(G, dtype=torch.int32, device=p.device)
        term = torch.empty((), dtype=torch.float32, device=p.device)
        ws = _hip.workspace_for(p.device, outer, G, inner)
        _hip.check(lib.lq_penalty_maxbin_fwd(_hip.ptr(p), _hip.ptr(s), _hip.ptr(mb), _hip.ptr(ties), _hip.ptr(term),
                                             _hip.ptr(ws), ws.numel(), outer, G, inner,
                                             _hip.stream_ptr(p.device)), "lq_penalty_maxbin_fwd")
        ctx.save_for_backward(p, s, mb, ties)
        ctx.desc = (outer, G, inner)
        return term

    @staticmethod
    def backward(ctx, grad_out):
        lib = _hip.load()
        p, s, mb, ties = ctx.saved_tensors
        outer, G, inner = ctx.desc
        c = _up(grad_out)
        dP = torch.empty_like(p)
        ds = torch.empty_like(s)
        _hip.check(lib.lq_penalty_maxbin_bwd(_hip.ptr(p), _hip.ptr(s), _hip.ptr(mb), _hip.ptr(ties), _hip.ptr(c), 1.0,
                                             _hip.ptr(dP), _hip.ptr(ds), outer, G, inner,
                                             _hip.stream_ptr(p.device)), "lq_penalty_maxbin_bwd")
        return dP, ds


class _DifferenceTerm(torch.autograd.Function):
    """mean |P - P/s|  (custom_loss_functions.py:172-176) -- K5b."""

    @staticmethod
    def forward(ctx, parameter, scale):
        lib = _hip.load()
        p, s, (outer, G, inner) = _param(parameter, scale)
        term = torch.empty((), dtype=torch.float32, device=p.device)
        ws = _hip.workspace_for(p.device, outer, G, inner)
        _hip.check(lib.lq_penalty_difference_fwd(_hip.ptr(p), _hip.ptr(s), _hip.ptr(term), _hip.ptr(ws), ws.numel(),
                                                 outer, G, inner, _hip.stream_ptr(p.device)),
                   "lq_penalty_difference_fwd")
        ctx.save_for_backward(p, s)
        ctx.desc = (outer, G, inner)
        return term

    @staticmethod
    def backward(ctx, grad_out):
        lib = _hip.load()
        p, s = ctx.saved_tensors
        outer, G, inner = ctx.desc
        c = _up(grad_out)
        dP = torch.empty_like(p)
        ds = torch.empty_like(s)
        ws = _hip.workspace_for(p.device, outer, G, inner)
        _hip.check(lib.lq_penalty_difference_bwd(_hip.ptr(p), _hip.ptr(s), _hip.ptr(c), 1.0, _hip.ptr(dP), _hip.ptr(ds),
                                                 _hip.ptr(ws), ws.numel(), outer, G, inner,
                                                 _hip.stream_ptr(p.device)), "lq_penalty_difference_bwd")
        return dP, ds


class _InverseTerm(torch.autograd.Function):
    """mean 1/where(s==0, eps, s)  (custom_loss_functions.py:252-256) -- K5c."""

    @staticmethod
    def forward(ctx, scale):
        lib = _hip.load()
        s = _hip.require_device_f32(scale, "scale")
        term = torch.empty((), dtype=torch.float32, device=s.device)
        _hip.check(lib.lq_penalty_inverse_fwd(_hip.ptr(s), _hip.ptr(term), s.numel(), _hip.stream_ptr(s.device)),
                   "lq_penalty_inverse_fwd")
        ctx.save_for_backward(s)
        return term

    @staticmethod
    def backward(ctx, grad_out):
        lib = _hip.load()
        (s,) = ctx.saved_tensors
        c = _up(grad_out)
        ds = torch.empty_like(s)
        _hip.check(lib.lq_penalty_inverse_bwd(_hip.ptr(s), _hip.ptr(c), 1.0, _hip.ptr(ds), s.numel(),
                                              _hip.stream_ptr(s.device)), "lq_penalty_inverse_bwd")
        return ds


def maxbin_term(parameter: torch.Tensor, scale: torch.Tensor) -> torch.Tensor:
    return _MaxBinTerm.apply(parameter, scale)


def difference_term(parameter: torch.Tensor, scale: torch.Tensor) -> torch.Tensor:
    return _DifferenceTerm.apply(parameter, scale)


def inverse_term(scale: torch.Tensor) -> torch.Tensor:
    return _InverseTerm.apply(scale)
