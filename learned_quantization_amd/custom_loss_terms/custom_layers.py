"""Same names as /root/reference/CIFAR-10/custom_loss_terms/custom_components/custom_layers.py:
2-argument STE-only ``my_custom_gradient`` (:49-64) and layers whose first nested-layer argument is
``penalty_rate`` (:71); the scale gets a zero gradient from the op and learns through a loss term."""
from .. import layers as _l
from ..layers import MinValueConstraint, eps_float32
from ..ops import my_custom_gradient as _op


def my_custom_gradient(parameter, scale):
    """CL custom_layers.py:49-64."""
    return _op(parameter, scale)


class CustomQuantizedScaleLayer(_l.CustomQuantizedScaleLayer):
    def __init__(self, penalty_rate, initializer, orientation):
        super().__init__(penalty_threshold=None, initializer=initializer, orientation=orientation,
                         penalty_rate=penalty_rate)


class CustomDenseLayer(_l.CustomDenseLayer):
    def __init__(self, seed, units, penalty_rate, orientation, initializer, name, regularizer,
                 trained_weights=None, **kwargs):
        super().__init__(seed=seed, units=units, penalty_threshold=None, orientation=orientation,
                         initializer=initializer, name=name, regularizer=regularizer,
                         trained_weights=trained_weights, penalty_rate=penalty_rate, **kwargs)


class CustomConv2DLayer(_l.CustomConv2DLayer):
    def __init__(self, seed, penalty_rate, orientation, initializer, filters, kernel_size, strides, padding, name,
                 regularizer, trained_weights=None, **kwargs):
        super().__init__(seed=seed, penalty_threshold=None, orientation=orientation, initializer=initializer,
                         filters=filters, kernel_size=kernel_size, strides=strides, padding=padding, name=name,
                         regularizer=regularizer, trained_weights=trained_weights, penalty_rate=penalty_rate,
                         **kwargs)


__all__ = ["my_custom_gradient", "MinValueConstraint", "CustomQuantizedScaleLayer", "CustomDenseLayer",
           "CustomConv2DLayer", "eps_float32"]
