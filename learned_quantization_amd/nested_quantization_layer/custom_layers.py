"""Same names as /root/reference/MNIST/nested_quantization_layer/custom_components/custom_layers.py
(3-argument ``my_custom_gradient``; ``CustomQuantizedScaleLayer(penalty_threshold, initializer, orientation)``)."""
from ..layers import (CustomConv2DLayer, CustomConv2DLayerNoBias, CustomDenseLayer, CustomQuantizedScaleLayer,
                      MinValueConstraint, eps_float32)
from ..ops import my_custom_gradient as _op


def my_custom_gradient(parameter, scale, penalty_threshold):
    """custom_layers.py:49-120."""
    return _op(parameter, scale, penalty_threshold)


__all__ = ["my_custom_gradient", "MinValueConstraint", "CustomQuantizedScaleLayer", "CustomDenseLayer",
           "CustomConv2DLayer", "CustomConv2DLayerNoBias", "eps_float32"]
