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


def test_workspace_is_sized_for_the_volume(monkeypatch):
    """the brick form's segment pools (6.5 GB at 256^3) are reserved only for volumes whose row kernel is the
    brick form -- by width, dtype and SAF_WIN_FORM -- not for every volume (host-only: no compute)."""
    monkeypatch.delenv("SAF_WIN_FORM", raising=False)
    l = _lib.lib()
    n = 256

    def vol(d, dt):
        v = _abi.SafVolume()
        v.nx = v.ny = v.nz = n
        v.feat_dim, v.feat_dtype, v.accum_mode, v.trunc = d, dt, _abi.SAF_RUNNING_MEAN, 0.03
        for f in ("axis_x", "axis_y", "axis_z", "tsdf", "tsdf_weight", "weight", "rgb", "clip_feat"):
            setattr(v, f, 4096)  # non-NULL, aligned: the descriptor is only inspected
        return v

    size = lambda d, dt: l.saf_fuse_workspace_bytes_for(ctypes.byref(vol(d, dt)), 5, 7)
    rows_f32, rows_bf16 = size(512, _abi.SAF_F32), size(512, _abi.SAF_BF16)
    assert 0 < rows_f32 < 1.5e9 and rows_f32 == rows_bf16, "a 512-channel volume needs no brick pools"
    assert size(320, _abi.SAF_F32) > 4e9 and size(256, _abi.SAF_BF16) > 4e9, "widths only the brick form takes"
    monkeypatch.setenv("SAF_WIN_FORM", "bricks")
    assert size(512, _abi.SAF_F32) > 4e9
    monkeypatch.setenv("SAF_WIN_FORM", "rows")
    assert size(320, _abi.SAF_F32) < 1.5e9
    assert l.saf_fuse_workspace_bytes_for(None, 5, 7) == 0
    # the form-agnostic entry stays conservative: enough for whatever dtype the volume turns out to have
    monkeypatch.delenv("SAF_WIN_FORM")
    assert l.saf_fuse_workspace_bytes(n ** 3, 512, 5, 7) == rows_f32
    assert l.saf_fuse_workspace_bytes(n ** 3, 256, 5, 7) >= size(256, _abi.SAF_BF16)


def test_slab_entry_rejects_bad_lists_on_the_host():
    l = _lib.lib()
    vol = _abi.SafVolume()
    fr = _abi.SafFrame()
    rc = l.saf_fuse_frames_slabs(ctypes.byref(vol), ctypes.byref(fr), 1, None, None, 0, None, 0, None, 0, None, None, None)
    assert rc == _abi.SAF_E_INVALID


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


def test_frame_descriptors_built_by_columns_match_the_struct(monkeypatch):
    """_make_frames fills `struct saf_frame[]` as int64 columns of a numpy array (no Python object per frame); every field
    must land where include/saf.h puts it.  Host-only: the CUDA requirement is lifted for the test."""
    import torch

    from spatially_aware_ai_amd import clipfusion as cf

    monkeypatch.setattr(cf, "require_cuda", lambda t, name: None)

    class Host(cf._FusionVolumeMixin):
        n_clip_feats = 6

    h, w, b, npy, npx = 5, 7, 4, 2, 3
    depth, rgb = torch.rand(b, h, w), torch.rand(b, h, w, 3)
    poses, ks, feat = torch.rand(b, 4, 4), torch.rand(b, 3, 3), torch.rand(b, 8, npy, npx)
    labs = [torch.rand(h, w) for _ in range(b)]
    for label_maps, bilinear in ((None, False), (labs, True), (torch.stack(labs), True)):
        arr, keep, gy, gx = Host()._make_frames(depth, rgb, poses, ks, feat, label_maps, bilinear)
        assert (gy, gx) == (npy, npx) and len(arr) == b and ctypes.sizeof(arr) == b * ctypes.sizeof(_abi.SafFrame)
        for i in range(b):
            f = arr[i]
            assert (f.height, f.width, f.npy, f.npx, f.rgb_bilinear) == (h, w, npy, npx, int(bilinear))
            assert f.depth == keep[0][i].data_ptr() and f.rgb == keep[1][i].data_ptr()
            assert f.pose == keep[2][i].data_ptr() and f.K == keep[3][i].data_ptr() and f.feat_map == keep[4][i].data_ptr()
            if label_maps is None:
                assert not f.label_map
            elif torch.is_tensor(label_maps):
                assert f.label_map == keep[5][0][i].data_ptr()
            else:
                assert f.label_map == keep[5][i].data_ptr()


def test_committed_bench_line_keeps_the_contract():
    """The bench line committed under profiles/ (what `python bench.py` printed on an MI355X) carries every field of the
    driver's contract, the roofline and cpu_baseline objects included."""
    import json

    path = os.path.join(ROOT, "profiles", "r03", "bench_n1.json")
    d = json.loads(open(path).read().strip().splitlines()[-1])
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline",
              "dtype", "data", "config", "roofline", "cpu_baseline"):
        assert k in d, k
    assert d["n_gpus"] == 1 and d["higher_is_better"] is True and d["vs_baseline"] is None and d["data"] == "synthetic"
    assert "workload" in d["config"] and "model" not in d["config"]
    r = d["roofline"]
    for k in ("bound", "achieved", "peak", "unit", "frac", "traffic"):
        assert k in r, k
    assert r["bound"] in ("hbm", "mfma") and abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-3
    c = d["cpu_baseline"]
    for k in ("value", "unit", "cores", "kind", "sample"):
        assert k in c, k
    assert c["kind"] in ("port", "reference")
    assert abs(d["value"] - d["config"]["frames_per_rank"] * d["steps"] / (d["ms_per_step"] * d["steps"] / 1e3)) / d["value"] < 1e-3
    # round 3: the side evidence rides on the driver's line
    assert d["end_to_end"] is not None and d["end_to_end"]["value"] > 0
    assert d["hbm_copy_GBps"] > 1000 and r["traffic"] is not None
    side = d["side_workloads"]
    for k in ("config2_128cube_f32", "config3_256cube_bf16_labels", "coherent_scene_depth_B", "config5_row_argmax",
              "config5_heat_maps", "config5_query_max", "config3_end_to_end"):
        assert k in side and side[k]["value"] is not None, k
    for k in ("config2_128cube_f32", "config5_row_argmax"):
        rr = side[k]["roofline"]
        assert rr["bound"] in ("hbm", "mfma") and abs(rr["frac"] - rr["achieved"] / rr["peak"]) < 1e-3
    assert d["slab_by_slab_fuse"]["ratio"] < 1.25


def test_no_kernel_of_the_library_uses_scratch_memory(tmp_path):
    """Every kernel of libsaf_hip.so keeps its working set in registers: no private segment, no vector register spilled to
    memory (the code objects' metadata notes; scalar registers parked in lanes of a vector register are not memory).  Round 5: locals of HIP's `uint4` -- a struct around a union -- that are loaded under a condition
    stayed in scratch memory (the brick form's segment image, the wide scan's staged text pieces); native vectors do not."""
    import shutil
    import subprocess

    llvm = "/opt/rocm/lib/llvm/bin"
    if not os.path.exists(os.path.join(llvm, "llvm-objdump")):
        import pytest

        pytest.skip("no llvm-objdump in this image")
    _lib.build()
    so = shutil.copy(_lib.LIB_PATH, tmp_path / "lib.so")  # (the bundles are extracted next to the file)
    subprocess.run([os.path.join(llvm, "llvm-objdump"), "--offloading", str(so)], check=True, capture_output=True, cwd=tmp_path)
    objects = sorted(p for p in os.listdir(tmp_path) if "gfx950" in p)
    assert objects, "no gfx950 code object in the library"
    kernels, bad = 0, []
    for obj in objects:
        notes = subprocess.run([os.path.join(llvm, "llvm-readelf"), "--notes", str(tmp_path / obj)], check=True,
                               capture_output=True, text=True).stdout
        name = None
        for line in notes.splitlines():
            m = re.match(r"\s*\.name:\s+(\S+)", line)
            if m:
                name = m.group(1)
                kernels += 1
            m = re.match(r"\s*\.(private_segment_fixed_size|vgpr_spill_count):\s+(\d+)", line)
            if m and int(m.group(2)) != 0:
                bad.append((name, m.group(1), int(m.group(2))))
    assert kernels > 100, kernels
    assert not bad, bad
