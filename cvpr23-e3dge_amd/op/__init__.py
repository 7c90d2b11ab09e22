"""StyleGAN2 custom ops on gfx950 -- same names as the reference's project/models/op/__init__.py:2-3."""
from .fused_act import FusedLeakyReLU, fused_leaky_relu, fused_bias_act, noise_bias_act
from .upfirdn2d import upfirdn2d, upfirdn2d_raw

__all__ = ["FusedLeakyReLU", "fused_leaky_relu", "fused_bias_act", "noise_bias_act", "upfirdn2d", "upfirdn2d_raw"]
