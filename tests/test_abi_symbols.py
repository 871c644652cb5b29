"""CPU: the C-ABI library builds for gfx950, loads, and exports every function that
include/saf.h declares (no compute calls -- there is no GPU here)."""
import ctypes
import os
import re

from spatially_aware_ai_amd import _abi, _lib

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared_functions():
    src = open(os.path.join(ROOT, "include", "saf.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(saf_[a-z_0-9]+)\s*\(", src)))


def test_header_and_ctypes_prototypes_agree():
    assert _declared_functions() == sorted(_abi.PROTOTYPES)


def test_library_builds_loads_and_exports_every_symbol():
    _lib.build()
    lib = ctypes.CDLL(_lib.LIB_PATH)
    for name in _declared_functions():
        assert hasattr(lib, name), f"{name} missing from libsaf_hip.so"
    l = _lib.lib()
    assert l.saf_abi_version() == _abi.ABI_VERSION
    # pure host-side helpers are safe without a GPU
    assert l.saf_fuse_workspace_bytes(64**3, 64, 5, 7) > 64**3 * 4
    assert l.saf_fuse_workspace_bytes(0, 64, 5, 7) == 0
    assert l.saf_query_workspace_bytes(7, _abi.SAF_Q_SURGERY) >= 28
    assert l.saf_query_workspace_bytes(7, _abi.SAF_Q_SOFTMAX) == 0


def test_struct_layout_matches_header():
    # 8 x 4-byte scalars, then 9 pointers
    assert ctypes.sizeof(_abi.SafVolume) == 8 * 4 + 9 * 8
    assert _abi.SafVolume.axis_x.offset == 32
    assert _abi.SafFrame.depth.offset == 8
    assert ctypes.sizeof(_abi.SafFrame) == 8 + 5 * 8 + 8 + 8 + 8


def test_invalid_arguments_are_rejected_on_the_host():
    l = _lib.lib()
    vol = _abi.SafVolume()  # all zeros
    fr = _abi.SafFrame()
    rc = l.saf_fuse_frame(ctypes.byref(vol), ctypes.byref(fr), None, 0, None, None)
    assert rc == _abi.SAF_E_INVALID
    assert b"bad volume shape" in l.saf_last_error()


def test_product_has_no_cpu_fallback():
    import pytest
    import torch

    from spatially_aware_ai_amd import clipfusion

    class FakeClip:
        feature_dim = 8

        def img_inference_tiled(self, rgb, patch_size, patch_stride):
            return torch.zeros(1, 8, 2, 3)

    f = clipfusion.ClipFusion(torch.zeros(3), 0.1, torch.tensor([4, 4, 4]), 0.3, False, FakeClip(), None, 10, 10)
    with pytest.raises(_lib.SafError, match="no CPU fallback"):
        f.integrate(torch.ones(1, 30, 40), torch.zeros(1, 30, 40, 3), torch.eye(4)[None], torch.eye(3)[None])
