"""Host layers: the reference's constructor / attribute / call surface on PyTorch-ROCm.

Mirrors /root/reference/MNIST/nested_quantization_layer/custom_components/custom_layers.py
(NQ-L) and /root/reference/CIFAR-10/custom_loss_terms/custom_components/custom_layers.py (CL-L):

  MinValueConstraint(min_value); __call__(w); get_config()                       NQ-L:35-46
  CustomQuantizedScaleLayer(penalty_threshold|penalty_rate, initializer, orientation)
      .build(input_shape) .call(inputs) .scale .orientation                      NQ-L:123-200, CL-L:67-144
  CustomDenseLayer(seed, units, penalty_threshold, orientation, initializer, name,
                   regularizer, trained_weights=None, **kw)
      .W .b .nested_q_w_layer .nested_q_b_layer .units                           NQ-L:203-268
  CustomConv2DLayer(seed, penalty_threshold, orientation, initializer, filters, kernel_size,
                    strides, padding, name, regularizer, trained_weights=None, **kw)
      .kernel .b .nested_q_k_layer .nested_q_b_layer                             NQ-L:271-350
  CustomConv2DLayerNoBias(...)  .kernel .nested_q_k_layer
      /root/reference/CIFAR-10/paper_implementation/custom_components/custom_layers.py:299-365

Parameters have the REFERENCE's shapes -- Dense ``W`` = (in, out), conv ``kernel`` =
HWIO (kh, kw, ci, co) -- so ``orientation`` means the same axes as in the reference
(rowwise = axis 0, columnwise = axis 1, channelwise = axis 2 = input channels), the loss terms
and callbacks see the same shapes, and the integer export is byte-compatible.  The matmul /
convolution themselves are stock (rocBLAS / MIOpen through torch) and out of scope; only
the fake-quant of W/kernel/b runs in this package's HIP kernels.

``kernel_storage`` (conv layers; not in the reference) chooses the MEMORY order behind the HWIO shape:
  "oihw" (default)  the kernel's elements lie in the order MIOpen consumes.  ``kernel.permute(3, 2, 0, 1)`` is a contiguous
                    OIHW tensor, so the fake-quantised kernel goes to the convolution as it is written, MIOpen's weight
                    gradient IS dP (custom_layers.py:118, 0 bytes), and the fake-quant kernels stream both (8 B per element
                    forward, 8 B backward -- SURVEY 8d's figures) with the groups described in memory order
                    (descriptor.memory_descriptor).  Index by index every tensor equals the HWIO-stored one.
  "hwio"            contiguous HWIO like a TensorFlow variable (zero-copy hand-over of the raw buffer to code that expects
                    that); the kernels then transpose through LDS tiles (lq_fq_forward_oihw / lq_fq_scale_grad_oihw).

Like Keras layers, these build lazily on first call (``build(input_shape)``); pass
``input_shape=`` (Dense: last dim, Conv: channels) to build eagerly so an optimizer can be
created before the first forward.

Conscious deviations (documented in DESIGN.md): activations are NCHW by default
(``data_format="NHWC"`` accepts the reference's layout); the ``setup_logger`` side effects
at construction (NQ-L:133-145) are dropped.
"""
from __future__ import annotations

import math
from typing import Optional, Tuple

import numpy as np
import torch
import torch.nn as nn
import torch.nn.functional as F

from . import ops
from .descriptor import scale_shape

eps_float32 = float(np.finfo(np.float32).eps)            # NQ-L:11
SCALE_INIT = float(np.float32(eps_float32 * 100))        # NQ-L:156
_NAME_COUNTERS = {}


def _auto_name(prefix: str) -> str:
    """Keras auto-naming: custom_dense_layer, custom_dense_layer_1, ... (the reference's
    name-substring selection relies on these because `name` is not forwarded, NQ-L:219,290)."""
    n = _NAME_COUNTERS.get(prefix, 0)
    _NAME_COUNTERS[prefix] = n + 1
    return prefix if n == 0 else f"{prefix}_{n}"


def reset_layer_names() -> None:
    """Counterpart of tf.keras.backend.clear_session() for the auto-name counters."""
    _NAME_COUNTERS.clear()


class RandomNormal:
    """tf.keras.initializers.RandomNormal(mean=0.0, stddev=0.05, seed=None) counterpart: callable(shape)."""

    def __init__(self, mean: float = 0.0, stddev: float = 0.05, seed: Optional[int] = None):
        self.mean, self.stddev, self.seed = mean, stddev, seed
        self._calls = 0

    def __call__(self, shape, dtype=torch.float32, device=None):
        gen = None
        if self.seed is not None:
            gen = torch.Generator(device="cpu")
            gen.manual_seed(int(self.seed) + self._calls)   # distinct tensors per call, reproducible per seed
            self._calls += 1
        t = torch.empty(tuple(shape), dtype=dtype).normal_(self.mean, self.stddev, generator=gen)
        return t.to(device) if device is not None else t


class L2:
    """tf.keras.regularizers.l2(l2): l2 * sum(w^2)."""

    def __init__(self, l2: float = 0.01):
        self.l2 = float(l2)

    def __call__(self, w: torch.Tensor) -> torch.Tensor:
        return self.l2 * torch.sum(torch.square(w))


def l2(value: float = 0.01) -> L2:
    return L2(value)


class MinValueConstraint:
    """Ensures the scale factor values stay above a defined minimum value (NQ-L:35-46)."""

    def __init__(self, min_value):
        self.min_value = min_value

    def __call__(self, w):
        """Returns max(w, min_value) (NQ-L:42-43).  Out of place, like tf.maximum."""
        out = w.detach().clone()
        return ops.min_value_project_(out, self.min_value)

    def project_(self, w: torch.Tensor) -> torch.Tensor:
        """In-place form used after an optimizer step (Keras applies constraints by assignment)."""
        with torch.no_grad():
            return ops.min_value_project_(w, self.min_value)

    def get_config(self):
        return {"min_value": self.min_value}


class CustomQuantizedScaleLayer(nn.Module):
    """Nested layer that owns the trainable scale and applies the fake-quant op (NQ-L:123-200).

    ``penalty_threshold`` given  -> nested-quantization op (hand-written scale gradient).
    ``penalty_rate`` given instead (CL-L:71) -> STE-only op; the scale learns through a loss term.
    """

    _SCALE_NAMES = {"rowwise": "Rowwise-scaler", "columnwise": "Columnwise-scaler",
                    "channelwise": "Columnwise-scaler",      # sic, NQ-L:178
                    "scalar": "Scalar-scaler"}

    def __init__(self, penalty_threshold=None, initializer=None, orientation="scalar", *, penalty_rate=None):
        super().__init__()
        self.initializer = initializer
        self.orientation = orientation
        self.penalty_threshold = penalty_threshold
        self.penalty_rate = penalty_rate
        self.constraint = MinValueConstraint(SCALE_INIT)          # NQ-L:158
        self.scale: Optional[nn.Parameter] = None
        self.scale_name: Optional[str] = None
        self.built = False
        self.defer_scale_grad = False          # set by DataParallel(mode="B"): ds is recomputed after the all-reduce

    def build(self, input_shape, device=None):
        shape = scale_shape(tuple(input_shape), self.orientation)  # raises ValueError like NQ-L:194-197
        self.scale = nn.Parameter(torch.full(shape, SCALE_INIT, dtype=torch.float32, device=device))  # NQ-L:156
        self.scale.lq_constraint = self.constraint                 # found by optim.apply_constraints
        self.scale.lq_is_scale = True
        if self.penalty_rate is not None and not hasattr(self, "penalty_rate_weight"):
            # the MNIST loss-term variant keeps the rate as a NON-trainable weight of the nested layer
            # (MNIST/custom_loss_terms/custom_components/custom_layers.py:97-102): a buffer here, saved with the model
            self.register_buffer("penalty_rate_weight", torch.tensor(float(self.penalty_rate), dtype=torch.float32, device=device))
        self.scale_name = self._SCALE_NAMES[self.orientation]
        self.built = True

    def call(self, inputs):
        if not self.built:
            self.build(tuple(inputs.shape), device=inputs.device)
        if self.penalty_threshold is None:
            return ops.my_custom_gradient(inputs, self.scale)                        # CL-L:143-144
        return ops.my_custom_gradient(inputs, self.scale, self.penalty_threshold,   # NQ-L:199-200
                                      defer_scale_grad=self.defer_scale_grad)

    forward = call

    def extra_repr(self):
        return f"orientation={self.orientation!r}, penalty_threshold={self.penalty_threshold}, penalty_rate={self.penalty_rate}"


def _nested(penalty_threshold, penalty_rate, orientation):
    return CustomQuantizedScaleLayer(penalty_threshold=penalty_threshold, initializer=None,
                                     orientation=orientation, penalty_rate=penalty_rate)


def _as_tensor(a, shape, device):
    t = torch.as_tensor(np.asarray(a) if not isinstance(a, torch.Tensor) else a, dtype=torch.float32)
    if tuple(t.shape) != tuple(shape):
        raise ValueError(f"trained weight has shape {tuple(t.shape)}, expected {tuple(shape)}")
    return t.to(device).contiguous()


class _HostLayer(nn.Module):
    def _init_value(self, initializer, shape, device):
        if initializer is None:
            raise ValueError("initializer is required when trained_weights is not given")
        v = initializer(shape)
        if not isinstance(v, torch.Tensor):
            v = torch.as_tensor(np.asarray(v), dtype=torch.float32)
        return v.to(dtype=torch.float32, device=device).contiguous()

    def regularization_loss(self) -> Optional[torch.Tensor]:
        """Keras adds regularizer(variable) of every regularised weight to the loss (NQ-L:251,259)."""
        if self.regularizer is None or not self.built:
            return None
        total = None
        for w in self._regularized():
            r = self.regularizer(w)
            total = r if total is None else total + r
        return total


class CustomDenseLayer(_HostLayer):
    """Standard dense layer with nested quantization layers for weights and bias (NQ-L:203-268)."""

    def __init__(self, seed=None, units=None, penalty_threshold=None, orientation="scalar", initializer=None,
                 name=None, regularizer=None, trained_weights=None, *, penalty_rate=None, input_shape=None,
                 device=None, **kwargs):
        super().__init__()
        self.seed = seed
        self.nested_q_w_layer = _nested(penalty_threshold, penalty_rate, orientation)       # NQ-L:222-224
        self.nested_q_b_layer = _nested(penalty_threshold, penalty_rate, "scalar")          # NQ-L:225-227
        self.units = units
        self.initializer = initializer
        self.regularizer = regularizer
        self.trained_weights = trained_weights
        self.name = _auto_name("custom_dense_layer")     # `name` is not forwarded in the reference (NQ-L:219)
        self.requested_name = name
        self.built = False
        if input_shape is not None:
            shape = (input_shape,) if isinstance(input_shape, int) else tuple(input_shape)
            self.build(shape, device=device)

    def build(self, input_shape, device=None):
        in_features = int(input_shape[-1])
        w_shape, b_shape = (in_features, self.units), (self.units,)                         # NQ-L:249,257
        if self.trained_weights:                                                            # NQ-L:240-245
            w = _as_tensor(self.trained_weights[0], w_shape, device)
            b = _as_tensor(self.trained_weights[1], b_shape, device)
        else:
            w = self._init_value(self.initializer, w_shape, device)
            b = self._init_value(self.initializer, b_shape, device)
        self.W = nn.Parameter(w)
        self.b = nn.Parameter(b)
        self.nested_q_w_layer.build(w_shape, device=device)
        self.nested_q_b_layer.build(b_shape, device=device)
        self.built = True

    def _regularized(self):
        return (self.W, self.b)

    def quantized_parameters(self):
        """(qW, qb) of this call: what FakeQuantBatch.quantize_all() left for the layer (one launch for all tensors, consumed
        once), else the nested layers' own ops (NQ-L:265-266)."""
        pre, self._q_pre = getattr(self, "_q_pre", None), None
        if pre is not None:
            return pre[0], pre[1]
        return self.nested_q_w_layer(self.W), self.nested_q_b_layer(self.b)

    def call(self, inputs):
        if not self.built:
            self.build(tuple(inputs.shape), device=inputs.device)
        qw, qb = self.quantized_parameters()
        return torch.add(torch.matmul(inputs, qw), qb)           # NQ-L:268

    forward = call


_KERNEL_STORAGE = ["oihw"]


class default_kernel_storage:
    """``with default_kernel_storage("hwio"): model = build_model(...)`` -- the memory order of conv kernels built inside."""

    def __init__(self, storage: str):
        if storage not in ("oihw", "hwio"):
            raise ValueError("kernel_storage must be 'oihw' or 'hwio'")
        self.storage = storage

    def __enter__(self):
        _KERNEL_STORAGE.append(self.storage)
        return self

    def __exit__(self, *exc):
        _KERNEL_STORAGE.pop()
        return False


def _pair(v) -> Tuple[int, int]:
    if isinstance(v, int):
        return (v, v)
    v = tuple(int(x) for x in v)
    return v if len(v) == 2 else (v[0], v[0])


def _same_padding(size: int, k: int, s: int) -> Tuple[int, int]:
    """TensorFlow 'SAME': total = max((ceil(n/s)-1)*s + k - n, 0), extra pixel goes after."""
    out = math.ceil(size / s)
    total = max((out - 1) * s + k - size, 0)
    return total // 2, total - total // 2


class _ConvBase(_HostLayer):
    _prefix = "custom_conv2d_layer"
    _has_bias = True

    def __init__(self, seed=None, penalty_threshold=None, orientation="scalar", initializer=None, filters=None,
                 kernel_size=(3, 3), strides=(1, 1), padding="same", name=None, regularizer=None,
                 trained_weights=None, *, penalty_rate=None, input_shape=None, data_format="NCHW", device=None,
                 kernel_storage=None, **kwargs):
        super().__init__()
        self.seed = seed
        if kernel_storage is None:
            kernel_storage = _KERNEL_STORAGE[-1]
        if kernel_storage not in ("oihw", "hwio"):
            raise ValueError("kernel_storage must be 'oihw' or 'hwio'")
        self.kernel_storage = kernel_storage
        self.nested_q_k_layer = _nested(penalty_threshold, penalty_rate, orientation)      # NQ-L:293-295
        if self._has_bias:
            self.nested_q_b_layer = _nested(penalty_threshold, penalty_rate, "scalar")     # NQ-L:296-298
        self.initializer = initializer
        self.filters = filters
        self.kernel_size = _pair(kernel_size)
        self.strides = _pair(strides)
        self.padding = str(padding).upper()                                                # NQ-L:305
        if self.padding not in ("SAME", "VALID"):
            raise ValueError(f"padding must be 'same' or 'valid', got {padding!r}")
        self.regularizer = regularizer
        self.trained_weights = trained_weights
        self.name = _auto_name(self._prefix)      # conv forwards **kwargs but not `name` (NQ-L:290)
        self.requested_name = name
        if data_format not in ("NCHW", "NHWC"):
            raise ValueError("data_format must be 'NCHW' or 'NHWC'")
        self.data_format = data_format
        self.built = False
        if input_shape is not None:
            ci = input_shape if isinstance(input_shape, int) else (
                input_shape[-1] if data_format == "NHWC" else input_shape[-3])
            self._build_channels(int(ci), device)

    def build(self, input_shape, device=None):
        ci = input_shape[-1] if self.data_format == "NHWC" else input_shape[-3]
        self._build_channels(int(ci), device)

    def _build_channels(self, ci: int, device):
        kernel_shape = (*self.kernel_size, ci, self.filters)                               # NQ-L:321  HWIO
        if self.trained_weights:                                                           # NQ-L:314-319
            k = _as_tensor(self.trained_weights[0], kernel_shape, device)
            b = _as_tensor(self.trained_weights[1], (self.filters,), device) if self._has_bias else None
        else:
            k = self._init_value(self.initializer, kernel_shape, device)
            b = self._init_value(self.initializer, (self.filters,), device) if self._has_bias else None
        if self.kernel_storage == "oihw":          # same shape and values, elements in (co, ci, kh, kw) order
            k = k.permute(3, 2, 0, 1).contiguous().permute(2, 3, 1, 0)
        self.kernel = nn.Parameter(k)
        self.nested_q_k_layer.build(kernel_shape, device=device)
        if self._has_bias:
            self.b = nn.Parameter(b)
            self.nested_q_b_layer.build((self.filters,), device=device)
        self.built = True

    def _regularized(self):
        return (self.kernel, self.b) if self._has_bias else (self.kernel,)

    def quantized_parameters(self):
        """(w, qb) of this call: ``w`` is the fake-quantised kernel as the OIHW-shaped tensor the convolution consumes (NQ-L:340),
        ``qb`` the fake-quantised bias (NQ-L:341; None without one) -- from FakeQuantBatch.quantize_all() when it ran (one launch
        for all tensors, consumed once), else from the nested layers' own ops."""
        pre, self._q_pre = getattr(self, "_q_pre", None), None
        nested = self.nested_q_k_layer
        if pre is not None and len(pre) > 2 and pre[2] is not None:
            w = pre[2]                                   # the batch emitted the OIHW companion with the same launch
        elif pre is not None:
            w = pre[0].permute(3, 2, 0, 1)                                                 # HWIO -> OIHW view
        elif (self.kernel_storage == "hwio" and nested.built and nested.penalty_threshold is not None
              and self.kernel.is_cuda and self.kernel.is_contiguous()):
            # NQ-L:340 with K1 writing the OIHW tensor MIOpen consumes (and K2 reading its OIHW weight gradient): same
            # values as nested(kernel).permute(3, 2, 0, 1), without the two transposition launches
            w = ops.my_custom_gradient_oihw(self.kernel, nested.scale, nested.penalty_threshold,
                                            defer_scale_grad=nested.defer_scale_grad)
        else:
            # NQ-L:340; HWIO -> OIHW view -- contiguous when the kernel is stored "oihw": MIOpen takes it as it is and its
            # weight gradient comes back with the parameter's own strides
            w = nested(self.kernel).permute(3, 2, 0, 1)
        if not self._has_bias:
            return w, None
        return w, (pre[1] if pre is not None else self.nested_q_b_layer(self.b))

    def call(self, inputs):
        if not self.built:
            self.build(tuple(inputs.shape), device=inputs.device)
        x = inputs.permute(0, 3, 1, 2) if self.data_format == "NHWC" else inputs
        w, qb = self.quantized_parameters()
        if self.padding == "SAME":
            ph = _same_padding(x.shape[-2], self.kernel_size[0], self.strides[0])
            pw = _same_padding(x.shape[-1], self.kernel_size[1], self.strides[1])
            if ph[0] == ph[1] and pw[0] == pw[1]:
                y = F.conv2d(x, w, None, self.strides, (ph[0], pw[0]))
            else:
                y = F.conv2d(F.pad(x, (pw[0], pw[1], ph[0], ph[1])), w, None, self.strides, 0)
        else:
            y = F.conv2d(x, w, None, self.strides, 0)                                      # NQ-L:343-348
        if qb is not None:
            y = torch.add(y, qb.view(1, -1, 1, 1))                                         # NQ-L:350
        return y.permute(0, 2, 3, 1) if self.data_format == "NHWC" else y

    forward = call


class CustomConv2DLayer(_ConvBase):
    """Standard convolutional layer with nested quantization layers for kernel and bias (NQ-L:271-350)."""


class CustomConv2DLayerNoBias(_ConvBase):
    """Bias-free variant (paper_implementation custom_layers.py:299-365)."""
    _has_bias = False


def custom_layers_of(model: nn.Module):
    """The reference selects layers by name substring (experiment.py:73, custom_loss_terms/experiment.py:446)."""
    return [m for m in model.modules()
            if isinstance(m, (CustomDenseLayer, _ConvBase))]
