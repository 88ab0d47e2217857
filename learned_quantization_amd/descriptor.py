"""Scale shapes and the (outer, G, inner) group descriptor of the C ABI.

``scale_shape`` restates CustomQuantizedScaleLayer.build
(/root/reference/MNIST/nested_quantization_layer/custom_components/custom_layers.py:147-197):
rowwise keeps axis 0, columnwise axis 1, channelwise axis 2, scalar is ``(1,)``;
any other string raises ``ValueError`` with the reference's message (:194-197).

``group_descriptor`` maps (parameter shape, scale shape) to the descriptor of
include/lq_hip.h: element ``i`` of the contiguous parameter uses scale element
``(i // inner) % G``.

``memory_order`` / ``memory_descriptor`` do the same for a parameter whose MEMORY is a permutation of its logical axes
(a conv kernel presented HWIO like the reference's, custom_layers.py:321, but stored in the OIHW order the convolution
library consumes): the descriptor then speaks about the element order in memory, which is all the kernels see.
"""
from __future__ import annotations

import math
from typing import Optional, Sequence, Tuple

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


def memory_order(shape: Sequence[int], strides: Sequence[int]) -> Optional[Tuple[int, ...]]:
    """Logical axes from the slowest- to the fastest-varying one in memory, or ``None`` when the tensor is not a dense
    permutation of a contiguous array (gaps, overlaps, broadcast strides).  Unit axes keep their logical position."""
    shape, strides = tuple(int(d) for d in shape), tuple(int(d) for d in strides)
    order = sorted(range(len(shape)), key=lambda a: (-strides[a] if shape[a] != 1 else 0, a))
    # unit axes may carry any stride: place them by position only (their stride never addresses anything)
    real = [a for a in order if shape[a] != 1]
    expect = 1
    for a in reversed(real):
        if strides[a] != expect:
            return None
        expect *= shape[a]
    units = [a for a in range(len(shape)) if shape[a] == 1]
    return tuple(real + units) if real else tuple(range(len(shape)))


def memory_descriptor(shape: Sequence[int], strides: Sequence[int], scale_shape_: Sequence[int]) -> Optional[Tuple[int, int, int]]:
    """(outer, G, inner) over the parameter's elements IN MEMORY ORDER, or ``None`` when the memory is not a dense permutation
    of the logical axes.  Equals ``group_descriptor`` for a contiguous parameter."""
    shape = tuple(int(d) for d in shape)
    scale_shape_ = tuple(int(d) for d in scale_shape_)
    order = memory_order(shape, strides)
    if order is None:
        return None
    if order == tuple(range(len(shape))) or (len(scale_shape_) == 1 and scale_shape_[0] == 1):
        return group_descriptor(shape, scale_shape_)
    if len(scale_shape_) != len(shape):
        return group_descriptor(shape, scale_shape_)          # raises with the usual message
    return group_descriptor(tuple(shape[a] for a in order), tuple(scale_shape_[a] for a in order))
