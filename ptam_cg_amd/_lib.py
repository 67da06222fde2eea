"""Loads libptam_hip.so (the HIP/gfx950 build of include/ptam_hip.h).  There is no CPU fallback:
if the library is missing, or no GPU is present when a context is created, this fails loudly."""
import ctypes
import os
import sys

from . import _abi

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("PTAM_HIP_LIB", os.path.join(_HERE, "csrc", "libptam_hip.so"))   # override: instrumented builds
_bound = None


def load():
    """Returns the bound HIP library (cached)."""
    global _bound
    if _bound is not None:
        return _bound
    if not os.path.exists(LIB_PATH):
        raise RuntimeError(
            f"{LIB_PATH} not found: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
            "or `make -C ptam_cg_amd/csrc` (hipcc --offload-arch=gfx950). There is no CPU fallback.")
    # PyTorch wheels bundle their own libamdhip64.so.7 / librccl.so.1.  If torch is (or will be) in
    # this process, it must be loaded FIRST so that both share one HIP runtime (SONAME match);
    # otherwise two runtimes would coexist in the process.
    if "torch" not in sys.modules and os.environ.get("PTAM_NO_TORCH_PRELOAD", "0") != "1":
        try:
            import torch  # noqa: F401
        except Exception:  # pragma: no cover - torch absent: system ROCm is used
            pass
    lib = ctypes.CDLL(LIB_PATH, mode=ctypes.RTLD_GLOBAL)
    _bound = _abi.bind(lib, "ptam_")
    if _bound.missing:
        raise RuntimeError(f"libptam_hip.so lacks symbols: {_bound.missing}")
    return _bound
