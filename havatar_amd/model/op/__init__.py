"""Mirror of the reference's model/op package (model/op/__init__.py:1-2)."""
from .fused_act import FusedLeakyReLU, fused_leaky_relu
from .upfirdn2d import upfirdn2d
from . import conv2d_gradfix

__all__ = ["FusedLeakyReLU", "fused_leaky_relu", "upfirdn2d", "conv2d_gradfix"]
