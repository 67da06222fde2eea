"""ptam_cg_amd — MI355X-native PTAM tracking + bundle-adjustment hot path.

Product = ptam_cg_amd/csrc (HIP kernels + C ABI of include/ptam_hip.h, built as libptam_hip.so);
this package is the thin host-side mirror of the reference's class surface used by tests and bench.
"""
from . import _abi  # noqa: F401
from .host import (Bundle, Context, KeyFrame, PatchFinder, PtamError, Tracker, DEFAULT_CAMERA)  # noqa: F401
