"""Scale shapes and the (outer, G, inner) group descriptor of the C ABI.

``scale_shape`` restates CustomQuantizedScaleLayer.build
(/root/reference/MNIST/nested_quantization_layer/custom_components/custom_layers.py:147-197):
rowwise keeps axis 0, columnwise axis 1, channelwise axis 2, scalar is ``(1,)``;
any other string raises ``ValueError`` with the reference's message (:194-197).

``group_descriptor`` maps (parameter shape, scale shape) to the descriptor of
include/lq_hip.h: element ``i`` of the contiguous parameter uses scale element
``(i // inner) % G``.
"""
from __future__ import annotations

import math
from typing import Sequence, Tuple

ORIENTATIONS = ("rowwise", "columnwise", "channelwise", "scalar")

_AXIS = {"rowwise": 0, "columnwise": 1, "channelwise": 2}


def scale_shape(input_shape: Sequence[int], orientation: str) -> Tuple[int, ...]:
    input_shape = tuple(int(d) for d in input_shape)
    if orientation in _AXIS:
        axis = _AXIS[orientation]
        return tuple(input_shape[i] if i == axis else 1 for i in range(len(input_shape)))
    if orientation == "scalar":
        return (1,)
    raise ValueError(
        f"Invalid scaler application: {orientation}. Expected rowwise, columnwise or scalar."
    )


def group_descriptor(param_shape: Sequence[int], scale_shape_: Sequence[int]) -> Tuple[int, int, int]:
    param_shape = tuple(int(d) for d in param_shape)
    scale_shape_ = tuple(int(d) for d in scale_shape_)
    numel = math.prod(param_shape) if param_shape else 1
    if numel <= 0:
        raise ValueError(f"parameter must be non-empty, got shape {param_shape}")
    if len(scale_shape_) == 1 and scale_shape_[0] == 1:
        return 1, 1, numel
    if len(scale_shape_) != len(param_shape):
        raise ValueError(
            f"scale shape {scale_shape_} is not broadcast-compatible with parameter shape {param_shape}: "
            "rank must match (or the scale must be (1,))"
        )
    axes = [i for i, d in enumerate(scale_shape_) if d != 1]
    if not axes:
        return 1, 1, numel
    if len(axes) > 1:
        raise ValueError(f"scale shape {scale_shape_} has more than one non-unit axis")
    a = axes[0]
    if scale_shape_[a] != param_shape[a]:
        raise ValueError(f"scale axis {a} has length {scale_shape_[a]}, parameter has {param_shape[a]}")
    outer = math.prod(param_shape[:a]) if a > 0 else 1
    inner = math.prod(param_shape[a + 1:]) if a + 1 < len(param_shape) else 1
    return outer, param_shape[a], inner
