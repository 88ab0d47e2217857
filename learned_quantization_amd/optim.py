"""Scale optimizer (K6) and constraint application.

Keras applies a variable's ``constraint`` right after the optimizer update; for the scales
that is ``MinValueConstraint(100 * eps_f32)``
(/root/reference/MNIST/nested_quantization_layer/custom_components/custom_layers.py:35-46,158).
``ScaleAdam`` fuses the Adam update of a scale with that projection in one HIP launch.
Its default arithmetic is Keras 2.11's Adam (epsilon 1e-7 outside the bias correction), the
optimizer the reference trains with (``Adam(learning_rate=1e-4)``,
/root/reference/CIFAR-10/nested_quantization_layer/experiment.py:435-443).
"""
from __future__ import annotations

from typing import Iterable

import torch

from . import ops


def scale_parameters(module: torch.nn.Module):
    return [p for p in module.parameters() if getattr(p, "lq_is_scale", False)]


def non_scale_parameters(module: torch.nn.Module):
    return [p for p in module.parameters() if not getattr(p, "lq_is_scale", False)]


def apply_constraints(params_or_module) -> None:
    """s <- max(s, min_value) for every parameter carrying an ``lq_constraint`` (in place)."""
    params = params_or_module.parameters() if isinstance(params_or_module, torch.nn.Module) else params_or_module
    for p in params:
        c = getattr(p, "lq_constraint", None)
        if c is not None:
            c.project_(p.data)


def _adopt_layout(state: dict, like: torch.Tensor) -> None:
    """Brings the moments ``state["m"]``, ``state["v"]`` into the element order of ``like`` (the parameter's memory).  Moments that
    arrived through ``load_state_dict`` keep the CHECKPOINT's strides -- a model saved with another ``kernel_storage``, or a
    contiguous round-2 one -- while the update kernel walks w, m, v and g as flat memory: left alone they would be paired with the
    wrong weights, silently."""
    for k in ("m", "v"):
        t = state[k]
        if t.shape != like.shape or t.dtype != like.dtype or t.device != like.device:
            raise ValueError(f"KerasAdam: state '{k}' of a parameter of shape {tuple(like.shape)} ({like.dtype}, {like.device}) has "
                             f"shape {tuple(t.shape)} ({t.dtype}, {t.device})")
        same = all(d == 1 or x == y for d, x, y in zip(t.shape, t.stride(), like.stride()))
        if not same:
            state[k] = torch.empty_like(like).copy_(t)


class KerasAdam(torch.optim.Optimizer):
    """Adam for ANY fp32 parameters with Keras 2.11 arithmetic (the reference's optimizer,
    /root/reference/CIFAR-10/nested_quantization_layer/experiment.py:435-443) in ONE launch per <= 256 tensors
    (``lq_adam_set_step``, SURVEY f-4).  ``mode="torch"`` gives torch.optim.Adam's arithmetic instead.  Parameters
    carrying an ``lq_constraint`` (the learned scales) are projected in the same launch."""

    _CHUNK = 256

    def __init__(self, params: Iterable[torch.nn.Parameter], lr: float = 1e-4, betas=(0.9, 0.999), eps: float = 1e-7,
                 mode: str = "keras", capturable: bool = False):
        if mode not in ("keras", "torch"):
            raise ValueError("mode must be 'keras' or 'torch'")
        super().__init__(params, dict(lr=lr, betas=betas, eps=eps, mode=mode))
        self.capturable = capturable
        self._step = 0
        self._step_t = None
        self._sets = None

    def _build(self):
        import ctypes
        from . import _hip
        lib = _hip.load()
        self._sets = []
        for group in self.param_groups:
            ps = [p for p in group["params"] if p.requires_grad]
            for p in ps:
                # element-wise update: any dense memory order will do as long as w, m, v and the gradient share it
                d = p.data
                if _hip.require_device_f32(d, "parameter", dense_ok=True) is not d:
                    raise ValueError("KerasAdam needs dense parameters (contiguous, or a permutation of a contiguous array)")
                st = self.state[p]
                if "m" not in st:
                    st["m"] = torch.zeros_like(p.data)
                    st["v"] = torch.zeros_like(p.data)
                _adopt_layout(st, p.data)
            for k in range(0, len(ps), self._CHUNK):
                chunk = ps[k:k + self._CHUNK]
                n = len(chunk)
                w = (ctypes.c_void_p * n)(*[p.data_ptr() for p in chunk])
                m = (ctypes.c_void_p * n)(*[self.state[p]["m"].data_ptr() for p in chunk])
                v = (ctypes.c_void_p * n)(*[self.state[p]["v"].data_ptr() for p in chunk])
                cnt = (ctypes.c_int64 * n)(*[p.numel() for p in chunk])
                mv = (ctypes.c_float * n)(*[float(getattr(p, "lq_constraint").min_value) if getattr(p, "lq_constraint", None)
                                            is not None else float("-inf") for p in chunk])
                handle = ctypes.c_void_p()
                _hip.check(lib.lq_adam_set_create(w, m, v, cnt, mv, n, ctypes.byref(handle)), "lq_adam_set_create")
                self._sets.append((group, chunk, handle, (ctypes.c_void_p * n)(), [p.data_ptr() for p in chunk]))

    def _destroy_sets(self):
        try:
            from . import _hip
            lib = _hip.load()
            for _, _, handle, _, _ in (self._sets or []):
                lib.lq_adam_set_destroy(handle)
        except Exception:
            pass
        self._sets = None

    def __del__(self):
        self._destroy_sets()

    def state_dict(self):
        sd = super().state_dict()
        sd["lq_step"] = int(self._step_t.item()) if (self.capturable and self._step_t is not None) else int(self._step)
        return sd

    def load_state_dict(self, state_dict):
        """torch replaces the state tensors: the launch tables are rebuilt at the next step (fresh pointers, moments brought into
        the parameters' element order), and the step counter of the bias correction is restored."""
        sd = dict(state_dict)
        step = sd.pop("lq_step", None)
        super().load_state_dict(sd)
        self._destroy_sets()
        # now, not at the next step: torch's load_state_dict copies shallowly -- until they are re-laid out the loaded moments may
        # still be the tensors of the optimizer the state came from
        for group in self.param_groups:
            for p in group["params"]:
                st = self.state.get(p)
                if st and "m" in st and "v" in st:
                    _adopt_layout(st, p.data)
        if step is not None:
            self._step = int(step)
            if self._step_t is not None:
                self._step_t.fill_(int(step))
            elif self.capturable:
                self._pending_step = int(step)

    @torch.no_grad()
    def step(self, closure=None):
        from . import _hip
        loss = None
        if closure is not None:
            with torch.enable_grad():
                loss = closure()
        if self._sets is None:
            self._build()
        lib = _hip.load()
        if self.capturable:
            if self._step_t is None:
                self._step_t = torch.full((1,), int(getattr(self, "_pending_step", 0)), dtype=torch.int64,
                                          device=self._sets[0][1][0].device)
            self._step_t += 1
        else:
            self._step += 1
        keep = []
        f32 = torch.float32
        for group, chunk, handle, gptrs, ptrs in self._sets:
            for i, p in enumerate(chunk):
                d = p.data
                if d.data_ptr() != ptrs[i]:
                    raise RuntimeError("a parameter was re-allocated after the optimizer was built; create a new KerasAdam")
                g = p.grad
                if g is None:
                    gptrs[i] = None
                else:
                    # this loop runs every eager step for every parameter: the cheap comparisons first (autograd's own gradients
                    # pass them), the full checks and a relayout only where one fails
                    if g.dtype is not f32 or g.shape != d.shape or g.stride() != d.stride() or g.device != d.device:
                        g = _hip.require_device_f32(g, "gradient", like=d)
                    keep.append(g)
                    gptrs[i] = g.data_ptr()
            b1, b2 = group["betas"]
            md = {"keras": _hip.LQ_ADAM_KERAS, "torch": _hip.LQ_ADAM_TORCH}[group["mode"]]
            _hip.check(lib.lq_adam_set_step(handle, gptrs, group["lr"], b1, b2, group["eps"], self._step,
                                            _hip.ptr(self._step_t), md, _hip.stream_ptr(chunk[0].device)), "lq_adam_set_step")
        return loss


class ScaleAdam(torch.optim.Optimizer):
    """Adam for the learned scales with the MinValueConstraint projection fused (K6)."""

    def __init__(self, params: Iterable[torch.nn.Parameter], lr: float = 1e-4, betas=(0.9, 0.999),
                 eps: float = 1e-7, mode: str = "keras", capturable: bool = False):
        """``capturable=True`` keeps the step counter on the device (``lq_scale_adam_step_dev``) so that
        ``step()`` can be recorded into a hipGraph and replayed."""
        if mode not in ("keras", "torch"):
            raise ValueError("mode must be 'keras' or 'torch'")
        super().__init__(params, dict(lr=lr, betas=betas, eps=eps, mode=mode))
        self.capturable = capturable
        self._step_t = None

    @torch.no_grad()
    def step(self, closure=None):
        loss = None
        if closure is not None:
            with torch.enable_grad():
                loss = closure()
        if self.capturable:
            if self._step_t is None:
                dev = self.param_groups[0]["params"][0].device
                self._step_t = torch.zeros(1, dtype=torch.int64, device=dev)
            self._step_t += 1                                  # device-side counter: replay-safe
        for group in self.param_groups:
            b1, b2 = group["betas"]
            for p in group["params"]:
                if p.grad is None:
                    continue
                st = self.state[p]
                if not st:
                    st["step"] = 0
                    st["m"] = torch.zeros_like(p, memory_format=torch.contiguous_format)
                    st["v"] = torch.zeros_like(p, memory_format=torch.contiguous_format)
                st["step"] += 1
                c = getattr(p, "lq_constraint", None)
                min_value = float(c.min_value) if c is not None else float("-inf")
                if self.capturable:
                    ops.scale_adam_step_dev_(p.data, p.grad.contiguous(), st["m"], st["v"], self._step_t, lr=group["lr"],
                                             beta1=b1, beta2=b2, eps=group["eps"], min_value=min_value, mode=group["mode"])
                else:
                    ops.scale_adam_step_(p.data, p.grad.contiguous(), st["m"], st["v"], st["step"], lr=group["lr"],
                                         beta1=b1, beta2=b2, eps=group["eps"], min_value=min_value, mode=group["mode"])
        return loss
