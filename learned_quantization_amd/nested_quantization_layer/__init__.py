"""Import-path mirror of the reference's ``*/nested_quantization_layer/custom_components``."""
from . import custom_layers  # noqa: F401
