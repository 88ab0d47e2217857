"""Input pipeline pieces of the reference's IMAGENETTE driver, on device tensors (no dataset is reachable offline; the
harness feeds synthetic batches through the same transformations).

  augment_image(image, label)              /root/reference/IMAGENETTE/nested_quantization_layer/experiment.py:834-849
  preprocess_for_validation(image, label)  /root/reference/IMAGENETTE/nested_quantization_layer/experiment.py:852-855

Images are float tensors in 0..255 (the reference never normalises, experiment.py:512-513), layout NCHW (or CHW for one
image); plain torch ops -- this is data preparation, not the hot path.
"""
from __future__ import annotations

from typing import Optional

import torch
import torch.nn.functional as F


def _batched(image: torch.Tensor):
    return (image.unsqueeze(0), True) if image.dim() == 3 else (image, False)


def _rgb_to_hsv(rgb: torch.Tensor):
    r, g, b = rgb.unbind(1)
    mx, mn = rgb.max(1).values, rgb.min(1).values
    d = mx - mn
    s = torch.where(mx > 0, d / mx.clamp_min(1e-30), torch.zeros_like(mx))
    dz = d.clamp_min(1e-30)
    h = torch.where(mx == r, ((g - b) / dz) % 6.0, torch.where(mx == g, (b - r) / dz + 2.0, (r - g) / dz + 4.0))
    h = torch.where(d == 0, torch.zeros_like(h), h) / 6.0
    return h, s, mx


def _hsv_to_rgb(h, s, v):
    i = torch.floor(h * 6.0)
    f = h * 6.0 - i
    p, q, t = v * (1 - s), v * (1 - f * s), v * (1 - (1 - f) * s)
    i = i.long() % 6
    sel = lambda a0, a1, a2, a3, a4, a5: torch.stack([a0, a1, a2, a3, a4, a5], 0).gather(0, i.unsqueeze(0)).squeeze(0)   # noqa: E731
    return torch.stack([sel(v, q, p, p, t, v), sel(t, v, v, q, p, p), sel(p, p, t, v, v, q)], 1)


def augment_image(image: torch.Tensor, label, generator: Optional[torch.Generator] = None):
    """experiment.py:834-849: resize up to 256x256 (bilinear), random 224x224 crop, random horizontal flip, random
    brightness (delta in [-0.2, 0.2) ADDED to the 0..255 values, as tf.image.random_brightness does on float images), random
    saturation (factor in [0.8, 1.2), through HSV like tf.image.adjust_saturation), clip to [0, 255]."""
    x, single = _batched(image)
    n, dev = x.shape[0], x.device
    x = F.interpolate(x, size=(256, 256), mode="bilinear", align_corners=False, antialias=False)
    rnd = lambda *shape: torch.rand(*shape, device=dev, generator=generator)                      # noqa: E731
    top = (rnd(n) * 33).long().clamp_(max=32)
    left = (rnd(n) * 33).long().clamp_(max=32)
    rows = top.view(n, 1) + torch.arange(224, device=dev).view(1, 224)
    cols = left.view(n, 1) + torch.arange(224, device=dev).view(1, 224)
    flip = rnd(n) < 0.5
    cols = torch.where(flip.view(n, 1), cols.flip(1), cols)                                        # random_flip_left_right
    x = x[torch.arange(n, device=dev).view(n, 1, 1, 1), torch.arange(3, device=dev).view(1, 3, 1, 1), rows.view(n, 1, 224, 1), cols.view(n, 1, 1, 224)]
    x = x + (rnd(n) * 0.4 - 0.2).view(n, 1, 1, 1)                                                  # random_brightness(max_delta=0.2)
    h, s, v = _rgb_to_hsv(x)
    s = (s * (0.8 + rnd(n) * 0.4).view(n, 1, 1)).clamp_(0.0, 1.0)                                  # random_saturation(0.8, 1.2)
    x = _hsv_to_rgb(h, s, v).clamp_(0.0, 255.0)                                                    # clip_by_value
    return (x[0] if single else x), label


def preprocess_for_validation(image: torch.Tensor, label):
    """experiment.py:852-855: resize to 224x224 (bilinear)."""
    x, single = _batched(image)
    x = F.interpolate(x, size=(224, 224), mode="bilinear", align_corners=False, antialias=False)
    return (x[0] if single else x), label
