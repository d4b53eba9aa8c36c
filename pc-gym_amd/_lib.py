"""Loader for libpcgym_hip.so (the C ABI of include/pcgym_hip.h).

There is no fallback: if the HIP library is missing or cannot be loaded this
raises.  The product path never routes through oracle/ or any CPU implementation.
"""
from __future__ import annotations

import ctypes
import os

from . import _abi

_HERE = os.path.dirname(os.path.abspath(__file__))
# PCGYM_HIP_LIB: another build of the same library (same ABI version, checked below) -- A/B measurements of kernel variants
LIB_PATH = os.environ.get("PCGYM_HIP_LIB") or os.path.join(_HERE, "libpcgym_hip.so")
_lib = None


class PcgError(RuntimeError):
    def __init__(self, status, where):
        self.status = status
        msg = "?"
        if _lib is not None:
            msg = _lib.pcg_strerror(int(status)).decode()
        super().__init__(f"{where}: status {status} ({msg})")


def load():
    """Load (once) and return the ctypes handle; raises if the .so is absent."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise RuntimeError(
                f"{LIB_PATH} not found: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
                "or `make -C pc-gym_amd/csrc` (hipcc --offload-arch=gfx950). There is no CPU fallback.")
        lib = ctypes.CDLL(LIB_PATH)
        _abi.declare(lib)
        v = lib.pcg_version()
        if v != _abi.PCG_ABI_VERSION:
            raise RuntimeError(f"libpcgym_hip.so ABI {v} != python side {_abi.PCG_ABI_VERSION}")
        _lib = lib
    return _lib


def check(status, where):
    if status != 0:
        raise PcgError(status, where)
