"""Loader of the C-ABI HIP library (libsaf_hip.so, built in-tree by csrc/Makefile).

There is deliberately NO CPU fallback: if the library is missing, was built for another ABI
version, or lacks a symbol declared in include/saf.h, importing the ops fails loudly.
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess
import threading

from . import _abi

_HERE = os.path.dirname(os.path.abspath(__file__))
# SAF_LIB_PATH selects another build of the same ABI (same-box A/B measurements of two kernel versions)
LIB_PATH = os.environ.get("SAF_LIB_PATH") or os.path.join(_HERE, "libsaf_hip.so")
CSRC = os.path.join(_HERE, "csrc")

_lib = None
_lock = threading.Lock()


class SafError(RuntimeError):
    """A C-ABI call returned a negative status."""


def build(verbose: bool = False) -> str:
    """Compile csrc/*.hip for gfx950 into libsaf_hip.so (hipcc cross-compiles without a GPU)."""
    out = None if verbose else subprocess.DEVNULL
    subprocess.check_call(["make", "-C", CSRC, "-j4"], stdout=out)
    return LIB_PATH


def lib():
    """The loaded library with prototypes attached."""
    global _lib
    if _lib is None:
        with _lock:
            if _lib is None:
                if not os.path.exists(LIB_PATH):
                    raise SafError(
                        f"{LIB_PATH} is missing: build it with `make -C {CSRC}` "
                        "(or __graft_entry__.build()); there is no CPU fallback for the fused path"
                    )
                # torch first: it brings its own libamdhip64.so, and the library must bind to THAT runtime (same soname) --
                # loaded before torch it would pull in /opt/rocm's copy, and two HIP runtimes in one process do not know
                # each other's allocations (hipMemsetAsync on a torch tensor: invalid value)
                import torch  # noqa: F401

                l = C.CDLL(LIB_PATH)
                _abi.declare(l)
                got = l.saf_abi_version()
                if got != _abi.ABI_VERSION:
                    raise SafError(f"libsaf_hip.so ABI version {got} != expected {_abi.ABI_VERSION}; rebuild")
                _lib = l
    return _lib


def check(rc: int, what: str = ""):
    if rc != 0:
        msg = lib().saf_last_error()
        raise SafError(f"{what or 'saf call'} failed ({rc}): {msg.decode() if msg else ''}")


def current_stream_ptr():
    import torch

    # the raw handle of the current device's current stream (no Stream object, no is_available() round trip)
    return torch._C._cuda_getCurrentRawStream(torch._C._cuda_getDevice())


def require_cuda(t, name):
    if not t.is_cuda:
        raise SafError(
            f"{name} is on {t.device}: the fused path runs only on the MI355X (HIP) device; "
            "there is no CPU fallback -- move the module and its inputs to 'cuda'"
        )
