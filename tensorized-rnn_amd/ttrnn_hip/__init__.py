"""Host-side binding of libttrnn.so (C ABI in include/ttrnn.h) for PyTorch-ROCm tensors."""
from . import _lib  # noqa: F401
from ._lib import (TtrnnError, device_status, fp32_math, get_fp32_math, get_option, load, option, set_fp32_math,  # noqa: F401
                   set_option)
from .graph import CapturedTrainStep, adam_for_capture  # noqa: F401,E402
