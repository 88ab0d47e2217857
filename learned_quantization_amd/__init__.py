"""learned_quantization_amd -- MI355X-native (gfx950) learned-quantization hot path.

The nested-quantization fake-quant op (forward, STE, hand-written scale gradient), the three
custom loss terms and the scale update of anuunchin/learned-quantization, behind the
reference's own Python layer surface, executed by hand-written HIP kernels through the C ABI of
``include/lq_hip.h``.  See DESIGN.md / INTEGRATION.md.
"""
from .descriptor import ORIENTATIONS, group_descriptor, memory_descriptor, memory_order, scale_shape
from .layers import (default_kernel_storage, CustomConv2DLayer, CustomConv2DLayerNoBias, CustomDenseLayer, CustomQuantizedScaleLayer,
                     L2, MinValueConstraint, RandomNormal, SCALE_INIT, custom_layers_of, eps_float32, l2,
                     reset_layer_names)
from .losses import SCCEDifference, SCCEInverse, SCCEMaxBin, sparse_categorical_crossentropy
from .ops import (difference_term, fq_forward, fq_fwd_bwd_fused, fq_scale_grad, inverse_term, maxbin_term,
                  my_custom_gradient, q_absmax_over_axis, q_unique, quantized_integers)
from .optim import KerasAdam, ScaleAdam, apply_constraints, non_scale_parameters, scale_parameters
from .ddp import DataParallel, GradBucket
from .batch import BatchedScaleAdam, FakeQuantBatch
from .models import CIFARCNN, MNISTDense, ResNet18Like, ResNet50Like, build_model
from .data import augment_image, preprocess_for_validation
from .export import save_compress_parameters
from .tracking import AccuracyLossTrackingCallBack, NestedScaleTrackingCallback

__version__ = "0.1.0"
