"""MI355X-native perception hot path (ray generation, occupancy-grid marching, hash-grid field,
volumetric compositing, predictive-information scoring) behind the reference's call surface.

Everything here drives libmi355nerf.so (hand-written HIP for gfx950) through its C ABI
(include/mi355nerf.h).  There is no CPU or PyTorch fallback: without the library or without a
GPU the compute entry points raise.
"""
from . import _lib  # noqa: F401
from ._lib import MnfError, lib_path, load_library  # noqa: F401
