"""The three topologies of the reference's experiments, built from this package's custom layers (SURVEY f-3).

  mnist_dense      /root/reference/MNIST/nested_quantization_layer/experiment.py:119-162
                   Flatten -> CustomDense(128) -> relu -> CustomDense(10) -> softmax
  cifar_cnn        /root/reference/CIFAR-10/nested_quantization_layer/experiment.py:279-432
                   3 x [CustomConv(f) relu BN CustomConv(f) relu BN MaxPool Dropout], f = 32/64/128,
                   Dense(128, relu) BN Dropout(0.5) Dense(10, softmax)
  resnet18_like    /root/reference/IMAGENETTE/nested_quantization_layer/experiment.py:627-760
                   CustomConv 7x7/2 BN relu MaxPool3x3/2 Dropout(0.2); 8 basic blocks 64,64,128,128,256,256,512,512
                   (stride 2 + quantised 1x1 shortcut conv + BN at each widening, l2(1e-4) on block convs);
                   Dropout GAP Dropout Dense(softmax)

Only the custom layers are this package's product; BatchNorm / pooling / dropout / plain Dense / the
convolution itself are stock torch (MIOpen / rocBLAS).  Keras defaults are mirrored where they affect
semantics: BatchNormalization(momentum=0.99, epsilon=1e-3) == torch momentum 0.01, eps 1e-3;
RandomNormal stddev 0.05 for kernels AND biases of custom layers (custom_layers.py:318-319); plain Dense
layers get RandomNormal kernels and zero biases.  Models output class probabilities (softmax), as the
reference's do, so ``compute_total_loss(y_true, y_pred)`` keeps its meaning; like Keras' softmax activation the output
carries its logits (``losses.softmax``), from which the cross-entropy is computed.  Activations are NCHW.

``mode="nq"``: ``value`` is the penalty_threshold (nested-quantization scale gradient);
``mode="cl"``: ``value`` is the penalty_rate (STE-only op; scales learn through a custom loss term).
"""
from __future__ import annotations

from typing import Optional

import torch
import torch.nn as nn
import torch.nn.functional as F

from .layers import CustomConv2DLayer, CustomDenseLayer, RandomNormal, l2
from .losses import softmax


def _bn(c: int) -> nn.BatchNorm2d:
    return nn.BatchNorm2d(c, eps=1e-3, momentum=0.01)


def _kw(mode: str, value: float):
    if mode == "nq":
        return dict(penalty_threshold=value)
    if mode == "cl":
        return dict(penalty_threshold=None, penalty_rate=value)
    if mode == "nqcl":      # (penalty_threshold, penalty_rate): nested-quantization op + a loss term (extension, train.py)
        return dict(penalty_threshold=value[0], penalty_rate=value[1])
    raise ValueError("mode must be 'nq', 'cl' or 'nqcl'")


def _plain_dense(n_in: int, n_out: int, seed: Optional[int]) -> nn.Linear:
    lin = nn.Linear(n_in, n_out)
    with torch.no_grad():
        lin.weight.copy_(RandomNormal(seed=seed)((n_out, n_in)))
        lin.bias.zero_()
    return lin


class MNISTDense(nn.Module):
    """784 -> 128 -> 10, both layers quantised (101 770 quantised elements)."""

    def __init__(self, mode="nq", value=1e-10, seed=42, orientation="rowwise", input_shape=(28, 28, 1), device=None,
                 trained_weights=None):
        super().__init__()
        n_in = 1
        for d in input_shape:
            n_in *= d
        init = RandomNormal(seed=seed)
        tw = trained_weights or [None, None]
        self.dense_1 = CustomDenseLayer(seed=seed, units=128, orientation=orientation, initializer=init,
                                        name="custom_dense_layer_1", regularizer=None, trained_weights=tw[0],
                                        input_shape=n_in, device=device, **_kw(mode, value))
        self.dense_2 = CustomDenseLayer(seed=seed, units=10, orientation=orientation, initializer=init,
                                        name="custom_dense_layer_2", regularizer=None, trained_weights=tw[1],
                                        input_shape=128, device=device, **_kw(mode, value))

    def forward(self, x):
        x = torch.flatten(x, 1)
        x = F.relu(self.dense_1(x))
        return softmax(self.dense_2(x), dim=1)


class CIFARCNN(nn.Module):
    """Six quantised 3x3 convs (287 008 quantised elements) + two plain Dense layers."""

    def __init__(self, mode="nq", value=1e-11, seed=42, orientation="channelwise", input_shape=(3, 32, 32),
                 num_classes=10, device=None):
        super().__init__()
        init = RandomNormal(seed=seed)
        kw = _kw(mode, value)
        chans = [(input_shape[0], 32, "32_1"), (32, 32, "32_2"), (32, 64, "64_1"), (64, 64, "64_2"),
                 (64, 128, "128_1"), (128, 128, "128_2")]
        self.convs = nn.ModuleList()
        self.bns = nn.ModuleList()
        for ci, co, tag in chans:
            self.convs.append(CustomConv2DLayer(seed=seed, orientation=orientation, initializer=init, filters=co,
                                                kernel_size=(3, 3), strides=(1, 1), padding="same",
                                                name=f"custom_conv2d_layer_{tag}", regularizer=None, trained_weights=None,
                                                input_shape=ci, device=device, **kw))
            self.bns.append(_bn(co))
        self.drops = nn.ModuleList([nn.Dropout(0.2), nn.Dropout(0.3), nn.Dropout(0.4)])
        h, w = input_shape[1] // 8, input_shape[2] // 8
        self.dense_1 = _plain_dense(128 * h * w, 128, seed)
        self.bn_dense = nn.BatchNorm1d(128, eps=1e-3, momentum=0.01)
        self.drop_dense = nn.Dropout(0.5)
        self.out = _plain_dense(128, num_classes, seed)
        if device is not None:
            self.to(device)

    def forward(self, x):
        for blk in range(3):
            for j in range(2):
                k = 2 * blk + j
                x = self.bns[k](F.relu(self.convs[k](x)))      # conv -> relu -> BN (experiment.py:309-311)
            x = self.drops[blk](F.max_pool2d(x, 2))
        x = torch.flatten(x.permute(0, 2, 3, 1), 1)           # Keras flattens NHWC
        x = self.drop_dense(self.bn_dense(F.relu(self.dense_1(x))))
        return softmax(self.out(x), dim=1)


class _ResidualBlock(nn.Module):
    """ResNet v1 basic block of IMAGENETTE/.../experiment.py:627-704."""

    def __init__(self, ci, filters, stride, idx, mode, value, seed, orientation, device):
        super().__init__()
        init = RandomNormal(seed=seed)
        reg = l2(1e-4)
        kw = _kw(mode, value)
        mk = lambda cin, k, s, name: CustomConv2DLayer(                                      # noqa: E731
            seed=seed, orientation=orientation, initializer=init, filters=filters, kernel_size=k, strides=s,
            padding="same", name=name, regularizer=reg, input_shape=cin, device=device, **kw)
        self.conv1 = mk(ci, (3, 3), stride, f"custom_conv2d_layer_{filters}_{idx}_0")
        self.bn1 = _bn(filters)
        self.conv2 = mk(filters, (3, 3), (1, 1), f"custom_conv2d_layer_{filters}_{idx}_1")
        self.bn2 = _bn(filters)
        self.shortcut = None
        if tuple(stride) != (1, 1):
            self.shortcut = mk(ci, (1, 1), stride, "custom_conv2d_layer_shortcut")
            self.bn_s = _bn(filters)

    def forward(self, x):
        sc = x
        y = F.relu(self.bn1(self.conv1(x)))
        y = self.bn2(self.conv2(y))
        if self.shortcut is not None:
            sc = self.bn_s(self.shortcut(x))
        return F.relu(y + sc)


class ResNet18Like(nn.Module):
    """20 quantised convs: kernels 11 166 912 + biases 4 800 = 11 171 712 quantised elements."""

    def __init__(self, mode="nq", value=1e-11, seed=42, orientation="channelwise", input_shape=(3, 224, 224),
                 num_classes=10, device=None):
        super().__init__()
        init = RandomNormal(seed=seed)
        self.stem = CustomConv2DLayer(seed=seed, orientation=orientation, initializer=init, filters=64, kernel_size=(7, 7),
                                      strides=(2, 2), padding="same", name="custom_conv2d_layer_64_0", regularizer=None,
                                      input_shape=input_shape[0], device=device, **_kw(mode, value))
        self.bn0 = _bn(64)
        self.drop0 = nn.Dropout(0.2)
        cfg = [(64, 64, (1, 1), 1), (64, 64, (1, 1), 2), (64, 128, (2, 2), 1), (128, 128, (1, 1), 2),
               (128, 256, (2, 2), 1), (256, 256, (1, 1), 2), (256, 512, (2, 2), 1), (512, 512, (1, 1), 2)]
        self.blocks = nn.ModuleList([_ResidualBlock(ci, co, st, idx, mode, value, seed, orientation, device)
                                     for ci, co, st, idx in cfg])
        self.drop1 = nn.Dropout(0.3)
        self.drop2 = nn.Dropout(0.3)
        self.out = _plain_dense(512, num_classes, seed)
        if device is not None:
            self.to(device)

    def forward(self, x):
        x = F.relu(self.bn0(self.stem(x)))
        # MaxPooling2D(pool 3, stride 2, padding "same"): for even extents TF pads (0, 1); -inf padding
        h, w = x.shape[-2:]
        ph = max((-(-h // 2) - 1) * 2 + 3 - h, 0)
        pw = max((-(-w // 2) - 1) * 2 + 3 - w, 0)
        x = F.pad(x, (pw // 2, pw - pw // 2, ph // 2, ph - ph // 2), value=float("-inf"))
        x = self.drop0(F.max_pool2d(x, 3, 2))
        for b in self.blocks:
            x = b(x)
        x = self.drop2(torch.mean(self.drop1(x), dim=(2, 3)))
        return softmax(self.out(x), dim=1)


class _BottleneckBlock(nn.Module):
    """ResNet v1 bottleneck (1x1 reduce, 3x3, 1x1 expand x4) in the style of ``_ResidualBlock``: quantised convs with bias,
    BN after every conv, quantised 1x1 projection shortcut where the shape changes, l2(1e-4) on the block convs."""

    def __init__(self, ci, width, stride, project, idx, values, seed, orientation, device, mode):
        super().__init__()
        init = RandomNormal(seed=seed)
        reg = l2(1e-4)
        co = 4 * width

        def mk(cin, cout, k, s, name, value):
            return CustomConv2DLayer(seed=seed, orientation=orientation, initializer=init, filters=cout, kernel_size=k, strides=s,
                                     padding="same", name=name, regularizer=reg, input_shape=cin, device=device, **_kw(mode, value))
        self.conv1 = mk(ci, width, (1, 1), (1, 1), f"custom_conv2d_layer_{width}_{idx}_0", values[0])
        self.bn1 = _bn(width)
        self.conv2 = mk(width, width, (3, 3), stride, f"custom_conv2d_layer_{width}_{idx}_1", values[1])
        self.bn2 = _bn(width)
        self.conv3 = mk(width, co, (1, 1), (1, 1), f"custom_conv2d_layer_{width}_{idx}_2", values[2])
        self.bn3 = _bn(co)
        self.shortcut = None
        if project:
            self.shortcut = mk(ci, co, (1, 1), stride, "custom_conv2d_layer_shortcut", values[3])
            self.bn_s = _bn(co)

    def forward(self, x):
        sc = x
        y = F.relu(self.bn1(self.conv1(x)))
        y = F.relu(self.bn2(self.conv2(y)))
        y = self.bn3(self.conv3(y))
        if self.shortcut is not None:
            sc = self.bn_s(self.shortcut(x))
        return F.relu(y + sc)


class ResNet50Like(nn.Module):
    """BASELINE.json configs[4] ("IMAGENETTE ResNet-50, mixed 4/8-bit learned scales").  EXTENSION WITHOUT A REFERENCE CALL
    SITE: the reference's only ResNets are the ResNet-18-like net above and CIFAR-10/paper_implementation/resnet.py:139-140;
    there is no bit-width parameter anywhere -- quantisation intensity follows from the threshold / rate, so
    "mixed" is modelled as a per-layer ``value``: ``value`` may be a pair (coarse, fine); 3x3 convs get the first entry,
    1x1 convs, the stem and the classifier the second.  The fake-quant op itself is shape-generic and is checked against
    the oracle on exactly these tensors (tests/test_gpu_parity.py::test_resnet50_kernel_shapes_full_size).
    53 quantised convs (with bias) + a quantised Dense classifier = 108 quantised tensors, 23.5 M quantised elements."""

    def __init__(self, mode="nq", value=(1e-10, 1e-11), seed=42, orientation="channelwise", input_shape=(3, 224, 224),
                 num_classes=10, device=None):
        super().__init__()
        # mode "nqcl": value is (penalty_threshold, penalty_rate) for every layer, not a (coarse, fine) pair
        coarse, fine = (value if (isinstance(value, (tuple, list)) and mode != "nqcl") else (value, value))
        init = RandomNormal(seed=seed)
        self.stem = CustomConv2DLayer(seed=seed, orientation=orientation, initializer=init, filters=64, kernel_size=(7, 7),
                                      strides=(2, 2), padding="same", name="custom_conv2d_layer_64_0", regularizer=None,
                                      input_shape=input_shape[0], device=device, **_kw(mode, fine))
        self.bn0 = _bn(64)
        blocks = []
        ci = 64
        for stage, (width, n) in enumerate(((64, 3), (128, 4), (256, 6), (512, 3))):
            for j in range(n):
                stride = (2, 2) if (j == 0 and stage > 0) else (1, 1)
                blocks.append(_BottleneckBlock(ci, width, stride, j == 0, j + 1, (fine, coarse, fine, fine), seed, orientation,
                                               device, mode))
                ci = 4 * width
        self.blocks = nn.ModuleList(blocks)
        self.out = CustomDenseLayer(seed=seed, units=num_classes, orientation="rowwise" if orientation == "channelwise" else orientation,
                                    initializer=init, name="custom_dense_layer_out", regularizer=None, input_shape=2048,
                                    device=device, **_kw(mode, fine))
        if device is not None:
            self.to(device)

    def forward(self, x):
        x = F.relu(self.bn0(self.stem(x)))
        h, w = x.shape[-2:]
        ph = max((-(-h // 2) - 1) * 2 + 3 - h, 0)
        pw = max((-(-w // 2) - 1) * 2 + 3 - w, 0)
        x = F.pad(x, (pw // 2, pw - pw // 2, ph // 2, ph - ph // 2), value=float("-inf"))
        x = F.max_pool2d(x, 3, 2)
        for b in self.blocks:
            x = b(x)
        return softmax(self.out(torch.mean(x, dim=(2, 3))), dim=1)


def build_model(config: str, kernel_storage: str = None, **kw) -> nn.Module:
    """config: 'mnist' (C1), 'cifar' (C2/C4), 'imagenette' (C3), 'resnet50' (C5, extension).
    ``kernel_storage``: memory order of the conv kernels, "oihw" (default) or "hwio" (layers.py)."""
    if kernel_storage is not None:
        from .layers import default_kernel_storage
        with default_kernel_storage(kernel_storage):
            return build_model(config, **kw)
    if config == "resnet50":
        return ResNet50Like(**kw)
    if config == "mnist":
        return MNISTDense(**kw)
    if config == "cifar":
        return CIFARCNN(**kw)
    if config == "imagenette":
        return ResNet18Like(**kw)
    raise ValueError(f"unknown config {config!r}")


INPUT_SHAPES = {"mnist": (1, 28, 28), "cifar": (3, 32, 32), "imagenette": (3, 224, 224), "resnet50": (3, 224, 224)}
