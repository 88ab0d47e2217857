"""Import-path mirror of the reference's ``*/custom_loss_terms/custom_components``."""
from . import custom_layers, custom_loss_functions  # noqa: F401
