"""jxl_rs_amd -- MI355X-native JPEG XL reconstruction hot path (VarDCT dequant/IDCT/Gaborish/EPF,
Modular RCT/Palette/Squeeze) behind the C ABI of include/jxl_hip.h.

The compute lives in hand-written HIP kernels (csrc/*.hip, built for gfx950 into
libjxl_hip.so).  Importing the package loads that library and fails loudly if it is missing:
there is no CPU fallback in the product path.
"""
from . import lib as _lib

_LIB = _lib.load()  # raises ImportError when libjxl_hip.so has not been built

from .lib import Context, FrameParams, JxlHipError, Plane  # noqa: E402,F401
from . import synth  # noqa: E402,F401

__all__ = ["Context", "FrameParams", "JxlHipError", "Plane", "synth"]
