"""Custom loss terms: SCCEMaxBin / SCCEDifference / SCCEInverse.

Mirrors /root/reference/CIFAR-10/custom_loss_terms/custom_components/custom_loss_functions.py
(CL-F; the MNIST copy reads ``layer.W`` / ``nested_q_w_layer`` instead of ``layer.kernel`` /
``nested_q_k_layer`` -- both attribute sets are accepted here):

    Cls(layers, penalty_rate, log_dir, l2_lambda=0.01)                 CL-F:34,120,199
    .compute_total_loss(y_true, y_pred)  -> (B,) tensor                 CL-F:47-73
    .compute_{maxbin,difference,inverse}_penalty() -> scalar tensor     CL-F:75-116,161-195,240-275

The per-tensor terms (the O(#parameters) part) run in the HIP kernels K5a/K5b/K5c with analytic
backward; the per-layer scalar glue ``term * dim``, the sum over layers and ``/ normalizer``
are 0-dim device tensor arithmetic.  The sparse categorical cross-entropy is stock.

The reference writes three ``tf.print`` lines per step into ``<log_dir>/custom_losses/*.log``
(CL-F:60-71).  That forces a device->host sync per step, so it is opt-in here
(``log_every_step=True``); the files are always created, as ``setup_logger`` does (CL-F:13-30).
"""
from __future__ import annotations

import os
from typing import List, Sequence

import torch

from . import ops

_EPSILON = 1e-7   # keras.backend.epsilon()


def setup_logger(new_log_dir: str, logs: Sequence[str]) -> None:
    """Creates/truncates the log files (CL-F:13-30)."""
    os.makedirs(new_log_dir, exist_ok=True)
    for log in logs:
        with open(os.path.join(new_log_dir, log), "w"):
            pass


def softmax(logits: torch.Tensor, dim: int = -1) -> torch.Tensor:
    """tf.keras.activations.softmax: the probabilities, with the logits attached as ``_keras_logits`` -- exactly what
    Keras 2.11 does (keras/activations.py: ``output._keras_logits = x``) so that a later cross-entropy can be computed
    from the logits.  The models of this package end in it, as the reference's models end in
    ``Dense(..., activation='softmax')`` / ``tf.keras.activations.softmax`` (MNIST/nested_quantization_layer/
    experiment.py:116,160; CIFAR-10/custom_loss_terms/experiment.py:269,428)."""
    p = torch.softmax(logits, dim=dim)
    p._keras_logits = logits
    return p


def sparse_categorical_crossentropy(y_true: torch.Tensor, y_pred: torch.Tensor) -> torch.Tensor:
    """tf.keras.losses.sparse_categorical_crossentropy(y_true, y_pred) (from_logits=False), per sample.

    Keras 2.11 (keras/backend.py ``sparse_categorical_crossentropy`` -> ``_get_logits``): when ``y_pred`` is the output
    of a softmax activation it carries ``_keras_logits`` and the loss is computed FROM THE LOGITS
    (``tf.nn.sparse_softmax_cross_entropy_with_logits`` = -log_softmax(logits)[y]), with no clipping -- the reference's
    models all end in a softmax, so this is the branch its training runs (custom_loss_functions.py:52-54 passes the model
    output straight through).  With the reference's raw 0..255 inputs the nets saturate at initialisation; from the logits
    the gradient there is still softmax - onehot, where the clipped form below would give exactly zero.
    Probabilities that do not come from ``softmax()`` take Keras' other branch: clip to [1e-7, 1 - 1e-7], then -log."""
    y = y_true.reshape(-1).long()
    logits = getattr(y_pred, "_keras_logits", None)
    if logits is not None:
        return -torch.gather(torch.log_softmax(logits, dim=-1), 1, y.unsqueeze(1)).squeeze(1)
    p = torch.clamp(y_pred, _EPSILON, 1.0 - _EPSILON)
    return -torch.log(torch.gather(p, 1, y.unsqueeze(1)).squeeze(1))


def _kernel_and_scales(layer):
    if hasattr(layer, "kernel"):
        kernel, kernel_scale = layer.kernel, layer.nested_q_k_layer.scale        # CL-F:85-86
    else:
        kernel, kernel_scale = layer.W, layer.nested_q_w_layer.scale             # MNIST CL-F variant
    if hasattr(layer, "b") and layer.b is not None:
        return kernel, kernel_scale, layer.b, layer.nested_q_b_layer.scale       # CL-F:87-88
    return kernel, kernel_scale, None, None


def _dim(t: torch.Tensor) -> float:
    d = 1.0
    for n in t.shape:                                                            # CL-F:102-108
        d *= n
    return d


class _SCCEBase:
    _penalty_log = "penalty_loss.log"

    def __init__(self, layers, penalty_rate, log_dir, l2_lambda=0.01, log_every_step: bool = False):
        self.layers: List = list(layers)
        self.penalty_rate = penalty_rate
        self.l2_lambda = l2_lambda
        self.custom_loss_dir = os.path.join(log_dir, "custom_losses")
        self.log_every_step = log_every_step
        setup_logger(self.custom_loss_dir, ["total_loss.log", "scce_loss.log", self._penalty_log])

    def _penalty(self) -> torch.Tensor:
        raise NotImplementedError

    def _tensor_terms(self, kernel, kernel_scale, b, b_scale):
        raise NotImplementedError

    def _accumulate(self) -> torch.Tensor:
        total_penalty = None
        normalizer = 0.0
        for layer in self.layers:
            kernel, kernel_scale, b, b_scale = _kernel_and_scales(layer)
            k_term, b_term = self._tensor_terms(kernel, kernel_scale, b, b_scale)
            kernel_dim = _dim(kernel)
            layer_penalty = k_term * kernel_dim
            b_dim = 0.0
            if b is not None:
                b_dim = _dim(b)
                layer_penalty = layer_penalty + b_term * b_dim                   # CL-F:110
            total_penalty = layer_penalty if total_penalty is None else total_penalty + layer_penalty   # :112
            normalizer += kernel_dim + b_dim                                     # :114
        if total_penalty is None:
            raise ValueError("no layers given")
        return total_penalty / normalizer                                        # :116

    def compute_total_loss(self, y_true, y_pred):
        cross_entropy_loss = sparse_categorical_crossentropy(y_true, y_pred)     # CL-F:52-54
        penalty = self._penalty()
        total_loss = cross_entropy_loss + self.penalty_rate * penalty            # CL-F:58
        if self.log_every_step:                                                  # CL-F:60-71
            self._append("total_loss.log", float(total_loss.mean()))
            self._append("scce_loss.log", float(cross_entropy_loss.mean()))
            self._append(self._penalty_log, float(self.penalty_rate * penalty))
        return total_loss

    def _append(self, name, value):
        with open(os.path.join(self.custom_loss_dir, name), "a") as f:
            f.write(f"{value}\n")


class SCCEMaxBin(_SCCEBase):
    """SCCE + penalty_rate * MaxBin penalty (CL-F:33-116)."""
    _penalty_log = "maxbin_loss.log"

    def _tensor_terms(self, kernel, kernel_scale, b, b_scale):
        return ops.maxbin_term(kernel, kernel_scale), (ops.maxbin_term(b, b_scale) if b is not None else None)

    def compute_maxbin_penalty(self):
        return self._accumulate()

    _penalty = compute_maxbin_penalty


class SCCEDifference(_SCCEBase):
    """SCCE + penalty_rate * Difference penalty (CL-F:119-195)."""
    _penalty_log = "difference_loss.log"

    def _tensor_terms(self, kernel, kernel_scale, b, b_scale):
        return ops.difference_term(kernel, kernel_scale), (ops.difference_term(b, b_scale) if b is not None else None)

    def compute_difference_penalty(self):
        return self._accumulate()

    _penalty = compute_difference_penalty


class SCCEInverse(_SCCEBase):
    """SCCE + penalty_rate * Inverse penalty (CL-F:198-275)."""
    _penalty_log = "inverse_loss.log"

    def _tensor_terms(self, kernel, kernel_scale, b, b_scale):
        return ops.inverse_term(kernel_scale), (ops.inverse_term(b_scale) if b is not None else None)

    def compute_inverse_penalty(self):
        return self._accumulate()

    _penalty = compute_inverse_penalty
