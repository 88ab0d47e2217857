"""CPU tests of the model topologies (construction only; forward needs the GPU)."""
import pytest
import torch

import learned_quantization_amd as lq


def _quantized_elements(model):
    return sum(p.numel() for l in lq.custom_layers_of(model) for p in l._regularized())


def test_mnist_dense_sizes():
    lq.reset_layer_names()
    m = lq.build_model("mnist", mode="nq", value=1e-10, seed=42, orientation="rowwise")
    assert _quantized_elements(m) == 101770                               # C1
    names = [l.name for l in lq.custom_layers_of(m)]
    assert names == ["custom_dense_layer", "custom_dense_layer_1"]
    assert tuple(m.dense_1.nested_q_w_layer.scale.shape) == (784, 1)
    assert len(lq.scale_parameters(m)) == 4


@pytest.mark.parametrize("orient,kshape", [("rowwise", (3, 1, 1, 1)), ("columnwise", (1, 3, 1, 1)),
                                           ("channelwise", (1, 1, 32, 1)), ("scalar", (1,))])
def test_cifar_cnn_sizes(orient, kshape):
    lq.reset_layer_names()
    m = lq.build_model("cifar", mode="nq", value=1e-11, seed=42, orientation=orient)
    assert _quantized_elements(m) == 287008                               # C2/C4
    convs = lq.custom_layers_of(m)
    assert len(convs) == 6 and all("custom_conv2d_layer" in c.name for c in convs)   # name-substring selection
    assert tuple(convs[1].kernel.shape) == (3, 3, 32, 32)
    assert tuple(convs[1].nested_q_k_layer.scale.shape) == kshape
    assert m.dense_1.in_features == 128 * 4 * 4


def test_resnet18_like_sizes():
    lq.reset_layer_names()
    m = lq.build_model("imagenette", mode="cl", value=1e-7, seed=42, orientation="channelwise")
    convs = lq.custom_layers_of(m)
    assert len(convs) == 20                                               # C3
    kernels = sum(c.kernel.numel() for c in convs)
    biases = sum(c.b.numel() for c in convs)
    assert (kernels, biases) == (11166912, 4800)
    assert sum(1 for c in convs if c.regularizer is not None) == 19       # every block conv, not the stem
    assert all(c.nested_q_k_layer.penalty_threshold is None and c.nested_q_k_layer.penalty_rate == 1e-7 for c in convs)
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        m(torch.zeros(1, 3, 64, 64))
