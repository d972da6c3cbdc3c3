"""Drop-in for the reference's ``models.networks.stylegan2_op`` package: the
three names its callers import (stylegan2_op/__init__.py:1-2, used by
stylegan2_layers.py:17 and encoder_*.py:9-10), backed by HIP kernels of
libppst_hip.so instead of the JIT-built CUDA extensions.  There is no
``is_custom_kernel_supported`` switch and no pure-PyTorch fallback: tensors
that are not on the GPU raise (CHECK_CUDA, upfirdn2d.cpp:9, fused_bias_act.cpp:8).
"""
from .bias_act import FusedLeakyReLU, fused_leaky_relu
from .fir import upfirdn2d

__all__ = ["FusedLeakyReLU", "fused_leaky_relu", "upfirdn2d"]
