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
