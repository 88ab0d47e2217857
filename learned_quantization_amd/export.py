"""Integer export + compression (SURVEY f-1): the reference's end-of-run artefact.

Mirrors ``save_compress_parameters`` of
/root/reference/CIFAR-10/nested_quantization_layer/utils/log_scripts.py:61-97:

    weights[layer.name + '/W'] = floor(kernel / scale).astype(int8)      (:74-76)
    weights[layer.name + '/b'] = floor(b / b_scale).astype(int8)         (:77-79)
    np.save(log_dir/weights.npy, weights)  -> zip(ZIP_DEFLATED) -> file_sizes.log in MB  (:83-97)

The integers come from the HIP kernel K1 (``q_dtype = int8``: the same two's-complement wrap as
``astype(np.int8)``; the reference's cast silently wraps when |q| > 127, e.g. at the initial scale).
Layouts are the reference's (HWIO / (in,out)), so the arrays are byte-compatible.  Plain (non-custom)
conv/dense layers are stored in float32 like the reference does for "conv2d" layers (:80-82).
The reference does not store the scales (thesis chapter4.tex:328-330 notes they must be kept separately);
``scales.npz`` is written next to the archive as an extension.
"""
from __future__ import annotations

import os
import zipfile
from typing import Dict

import numpy as np
import torch

from . import ops
from .layers import CustomDenseLayer, _ConvBase


def _host(t: torch.Tensor) -> np.ndarray:
    """C-contiguous host array in the tensor's LOGICAL index order.  A conv kernel stored in OIHW order behind its HWIO shape comes
    back from ``.cpu().numpy()`` with permuted strides -- a 1x1 kernel even F-contiguous, which numpy pickles in Fortran byte order:
    the bytes of weights.npy (and the compressed size in file_sizes.log) would then depend on the storage."""
    return np.ascontiguousarray(t.detach().cpu().numpy())


def quantized_state(model: torch.nn.Module) -> Dict[str, np.ndarray]:
    weights: Dict[str, np.ndarray] = {}

    def q(param, nested):
        return _host(ops.quantized_integers(param.data, nested.scale.data, torch.int8))
    for layer in model.modules():
        if isinstance(layer, _ConvBase):
            weights[layer.name + "/W"] = q(layer.kernel, layer.nested_q_k_layer)
            if layer._has_bias:
                weights[layer.name + "/b"] = q(layer.b, layer.nested_q_b_layer)
        elif isinstance(layer, CustomDenseLayer):
            weights[layer.name + "/W"] = q(layer.W, layer.nested_q_w_layer)
            weights[layer.name + "/b"] = q(layer.b, layer.nested_q_b_layer)
    return weights


def scale_state(model: torch.nn.Module) -> Dict[str, np.ndarray]:
    out = {}
    for layer in model.modules():
        for attr, tag in (("nested_q_k_layer", "/W_scale"), ("nested_q_w_layer", "/W_scale"), ("nested_q_b_layer", "/b_scale")):
            nested = getattr(layer, attr, None)
            if nested is not None and getattr(nested, "scale", None) is not None and hasattr(layer, "name"):
                out[layer.name + tag] = _host(nested.scale)
    return out


def save_compress_parameters(model: torch.nn.Module, log_dir: str) -> Dict[str, float]:
    """Writes weights.npy, weights.zip, file_sizes.log (reference format) and scales.npz; returns sizes in MB."""
    os.makedirs(log_dir, exist_ok=True)
    weights_path = os.path.join(log_dir, "weights.npy")
    weights = quantized_state(model)
    for name, layer in model.named_modules():
        if isinstance(layer, (torch.nn.Conv2d, torch.nn.Linear)):            # log_scripts.py:80-82
            weights[name + "/W"] = _host(layer.weight)
            if layer.bias is not None:
                weights[name + "/b"] = _host(layer.bias)
    np.save(weights_path, weights)                                             # pickled dict, like the reference
    zip_file_path = os.path.join(log_dir, "weights.zip")
    with zipfile.ZipFile(zip_file_path, "w", compression=zipfile.ZIP_DEFLATED) as zipf:
        zipf.write(weights_path, arcname="weights.npy")
    np.savez(os.path.join(log_dir, "scales.npz"), **scale_state(model))
    size = os.path.getsize(weights_path) / (1024 * 1024)
    zip_size = os.path.getsize(zip_file_path) / (1024 * 1024)
    with open(os.path.join(log_dir, "file_sizes.log"), "w") as log_file:
        log_file.write(f"Weights size: {size:.4f} MB\n")
        log_file.write(f"Compressed weights size: {zip_size:.4f} MB\n")
    return {"weights_mb": size, "zip_mb": zip_size}
