"""Same names as /root/reference/CIFAR-10/custom_loss_terms/custom_components/custom_loss_functions.py."""
from ..losses import SCCEDifference, SCCEInverse, SCCEMaxBin, setup_logger

__all__ = ["SCCEMaxBin", "SCCEDifference", "SCCEInverse", "setup_logger"]
