"""PyTorch-CPU restatement of the reference path as the UNFUSED op sequence
TensorFlow executes  --  TEST INFRASTRUCTURE ONLY.

Two uses, both as checker / baseline, never as product:
  * ``bench.py``'s ``cpu_baseline`` leg times ``nq_forward_backward`` on the GPU
    box's host cores ("reference-equivalent CPU path
    (restated; TF 2.11 unavailable)").
  * tests differentiate the penalty restatements with torch autograd on CPU to
    obtain the gradients TF autodiff would produce (``torch.amax`` splits the
    gradient evenly across ties exactly like ``tf.reduce_max``).

Each function mirrors the same-named function of oracle/lq_oracle.py (which
cites the reference file:line per statement); tests/test_oracle.py checks the
two agree bit for bit on q/out/max and within rtol 1e-5 on tanh/mean terms.
Sources: /root/reference/MNIST/nested_quantization_layer/custom_components/custom_layers.py:49-120,
/root/reference/CIFAR-10/custom_loss_terms/custom_components/custom_loss_functions.py:75-116,161-195,240-275.
"""
from __future__ import annotations

import torch

EPS_F32 = float(torch.finfo(torch.float32).eps)


def fq_forward(parameter: torch.Tensor, scale: torch.Tensor):
    nonrounded = torch.div(parameter, scale)           # RealDiv
    rounded = torch.floor(nonrounded)                  # Floor
    scaled_back = torch.mul(rounded, scale)            # Mul
    return rounded, scaled_back


def nq_backward(parameter, scale, penalty_threshold, dy, rounded=None, scaled_back=None):
    """custom_grad(dy): one torch op per TF op of custom_layers.py:62-118."""
    if rounded is None:
        rounded, scaled_back = fq_forward(parameter, scale)
    lam = torch.tensor(float(penalty_threshold), dtype=torch.float32)
    eps = torch.tensor(EPS_F32, dtype=torch.float32)
    is_zero = torch.eq(scaled_back, 0.0)                               # Equal
    non_zero_param = torch.where(is_zero, eps, scaled_back)            # Select
    ratio = torch.div(torch.abs(dy), torch.abs(non_zero_param))        # Abs, Abs, RealDiv
    above = torch.ge(ratio, lam)                                       # GreaterEqual
    absq = torch.abs(rounded)                                          # Abs
    if scale.dim() == 1:
        maxvalue = torch.max(absq)                                     # Max
        all_above = torch.all(above)                                   # All
        all_above_broad = all_above.expand(parameter.shape)            # ExpandDims + BroadcastTo
        axes = None
    else:
        axes = [i for i in range(scale.dim()) if scale.shape[i] == 1]
        maxvalue = torch.amax(absq, dim=axes).reshape(scale.shape)
        all_above = _all_over(above, axes)                             # All
        all_above_broad = all_above.reshape(scale.shape).expand(parameter.shape)
    const = -1.0 * torch.abs(torch.tanh(lam))                          # Tanh, Abs, Mul
    inner = torch.where(above, torch.zeros((), dtype=torch.float32),
                        -1.0 * torch.abs(torch.tanh(torch.sub(lam, ratio))))   # Sub, Tanh, Abs, Mul, Select
    scale_grads = torch.where(all_above_broad, const, inner)           # Select
    if axes is None:
        reduced = torch.mean(scale_grads).reshape(scale.shape)         # Mean, Reshape
    else:
        reduced = torch.mean(scale_grads, dim=axes).reshape(scale.shape)
    return dy, torch.mul(reduced, maxvalue)                            # Mul


def _all_over(mask, axes):
    out = mask
    for a in sorted(axes, reverse=True):
        out = torch.all(out, dim=a, keepdim=True)
    return out


def nq_forward_backward(parameter, scale, penalty_threshold, dy):
    """One training-step worth of the op: forward, then custom_grad. Returns (out, dP, ds)."""
    rounded, scaled_back = fq_forward(parameter, scale)
    dp, ds = nq_backward(parameter, scale, penalty_threshold, dy, rounded, scaled_back)
    return scaled_back, dp, ds


# ----------------------------- penalties (differentiable) ----------------------------- #
def _dim(t):
    d = 1.0
    for n in t.shape:
        d *= n
    return d


def _maxbin(p, s):
    axes = [i for i in range(s.dim()) if s.shape[i] == 1 and s.dim() > 1]
    t = torch.abs(p) / s
    if axes:
        return torch.amax(t, dim=axes)
    return torch.amax(t)


def maxbin_penalty(layers):
    total, normalizer = 0, 0
    for k, ks, b, bs in layers:
        k_dim, b_dim = _dim(k), _dim(b)
        total = total + (torch.mean(_maxbin(k, ks)) * k_dim + torch.mean(_maxbin(b, bs)) * b_dim)
        normalizer += k_dim + b_dim
    return total / normalizer


def difference_penalty(layers):
    total, normalizer = 0, 0
    for k, ks, b, bs in layers:
        k_pen = torch.mean(torch.abs(k - k / ks))
        b_pen = torch.mean(torch.abs(b - b / bs))
        k_dim, b_dim = _dim(k), _dim(b)
        total = total + (k_pen * k_dim + b_pen * b_dim)
        normalizer += k_dim + b_dim
    return total / normalizer


def inverse_penalty(layers):
    total, normalizer = 0, 0
    eps = torch.tensor(EPS_F32, dtype=torch.float32)
    for k, ks, b, bs in layers:
        ks_nz = torch.where(ks == 0.0, eps, ks)
        bs_nz = torch.where(bs == 0.0, eps, bs)
        k_dim, b_dim = _dim(k), _dim(b)
        total = total + (torch.mean(1.0 / ks_nz) * k_dim + torch.mean(1.0 / bs_nz) * b_dim)
        normalizer += k_dim + b_dim
    return total / normalizer
