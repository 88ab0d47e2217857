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


def test_resnet50_like_sizes():
    """BASELINE configs[4] topology (extension: the reference has no ResNet-50): 53 quantised convs + quantised classifier."""
    import learned_quantization_amd as lq
    lq.reset_layer_names()
    m = lq.ResNet50Like(value=(1e-10, 1e-11))
    layers = lq.custom_layers_of(m)
    assert len(layers) == 54 and sum(len(l._regularized()) for l in layers) == 108
    assert sum(p.numel() for l in layers for p in l._regularized()) == 23501962
    thr = {tuple(l.kernel.shape)[:2]: l.nested_q_k_layer.penalty_threshold for l in layers if hasattr(l, "kernel")}
    assert thr[(3, 3)] == 1e-10 and thr[(1, 1)] == 1e-11 and thr[(7, 7)] == 1e-11        # "mixed": coarse on the 3x3 kernels


def test_input_pipeline_counterparts():
    """augment_image / preprocess_for_validation (IMAGENETTE experiment.py:834-855) on tensors."""
    import torch
    import learned_quantization_amd as lq
    g = torch.Generator().manual_seed(0)
    x = torch.rand(4, 3, 300, 280, generator=g) * 255
    y, lab = lq.augment_image(x, 7, generator=g)
    assert y.shape == (4, 3, 224, 224) and lab == 7 and float(y.min()) >= 0.0 and float(y.max()) <= 255.0
    z, _ = lq.preprocess_for_validation(x[0], 1)
    assert z.shape == (3, 224, 224)
    same, _ = lq.preprocess_for_validation(torch.rand(3, 224, 224) * 255, 0)
    assert same.shape == (3, 224, 224)


def test_loss_term_layers_keep_the_rate_as_a_non_trainable_weight():
    """MNIST/custom_loss_terms/custom_components/custom_layers.py:97-102."""
    from learned_quantization_amd.custom_loss_terms import custom_layers as CL
    nested = CL.CustomQuantizedScaleLayer(1e-7, None, "rowwise")
    import numpy as np
    nested.build((8, 4))
    assert float(nested.penalty_rate_weight) == np.float32(1e-7) and not nested.penalty_rate_weight.requires_grad
    assert "penalty_rate_weight" in dict(nested.named_buffers()) and [n for n, _ in nested.named_parameters()] == ["scale"]
