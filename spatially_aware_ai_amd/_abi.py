"""ctypes mirror of include/saf.h (structs, enums, prototypes).

Pure declarations -- no library is loaded here, so CPU-only tooling can import it.
"""
from __future__ import annotations

import ctypes as C

ABI_VERSION = 3

SAF_OK = 0
SAF_E_INVALID = -1
SAF_E_WORKSPACE = -2
SAF_E_HIP = -3
SAF_E_UNSUPPORTED = -4

SAF_F32, SAF_BF16, SAF_F16 = 0, 1, 2
SAF_RUNNING_MEAN, SAF_SUM = 0, 1
SAF_Q_SCORES, SAF_Q_SOFTMAX, SAF_Q_SURGERY = 0, 1, 2
SAF_NORM_NONE, SAF_NORM_L2, SAF_NORM_L2_CLAMP = 0, 1, 2
SAF_QW_SCORES, SAF_QW_VS_BACKGROUND, SAF_QW_ROW_ARGMAX, SAF_QW_QUERY_MAX = 0, 1, 2, 3
SAF_STATS_WORDS = 16
SAF_WINDOW_FRAMES = 128

_fp = C.c_void_p  # every data pointer travels as an integer address


class SafVolume(C.Structure):
    _fields_ = [
        ("nx", C.c_int32),
        ("ny", C.c_int32),
        ("nz", C.c_int32),
        ("feat_dim", C.c_int32),
        ("n_classes", C.c_int32),
        ("feat_dtype", C.c_int32),
        ("accum_mode", C.c_int32),
        ("trunc", C.c_float),
        ("axis_x", _fp),
        ("axis_y", _fp),
        ("axis_z", _fp),
        ("tsdf", _fp),
        ("tsdf_weight", _fp),
        ("weight", _fp),
        ("rgb", _fp),
        ("clip_feat", _fp),
        ("labels_one_hot", _fp),
    ]


class SafFrame(C.Structure):
    _fields_ = [
        ("height", C.c_int32),
        ("width", C.c_int32),
        ("depth", _fp),
        ("rgb", _fp),
        ("pose", _fp),
        ("K", _fp),
        ("feat_map", _fp),
        ("npy", C.c_int32),
        ("npx", C.c_int32),
        ("label_map", _fp),
        ("rgb_bilinear", C.c_int32),
    ]


# name -> (restype, argtypes); the exported symbols of libsaf_hip.so (include/saf.h)
PROTOTYPES = {
    "saf_last_error": (C.c_char_p, []),
    "saf_abi_version": (C.c_int, []),
    "saf_fuse_workspace_bytes": (C.c_size_t, [C.c_int64, C.c_int32, C.c_int32, C.c_int32]),
    "saf_fuse_workspace_bytes_for": (C.c_size_t, [C.POINTER(SafVolume), C.c_int32, C.c_int32]),
    "saf_fuse_workspace_bytes_for_frames": (C.c_size_t, [C.POINTER(SafVolume), C.c_int32, C.c_int32, C.c_int32, C.c_int32]),
    "saf_fuse_frame": (
        C.c_int,
        [C.POINTER(SafVolume), C.POINTER(SafFrame), _fp, C.c_size_t, _fp, _fp],
    ),
    "saf_fuse_frames": (
        C.c_int,
        [C.POINTER(SafVolume), C.POINTER(SafFrame), C.c_int32, _fp, C.c_size_t, _fp, _fp],
    ),
    "saf_poll_async_error": (C.c_int, []),
    "saf_stage_frame": (C.c_int, [C.POINTER(SafFrame), C.c_int32, C.c_int64, C.c_int64, C.c_int64, C.POINTER(SafFrame), _fp]),
    "saf_fuse_frames_slabs": (
        C.c_int,
        [C.POINTER(SafVolume), C.POINTER(SafFrame), C.c_int32, C.POINTER(C.c_int32), C.POINTER(C.c_int32), C.c_int32,
         C.POINTER(C.c_void_p), C.c_int32, _fp, C.c_size_t, _fp, C.c_void_p, C.c_void_p],
    ),
    "saf_fuse_path": (C.c_int, [C.POINTER(SafVolume), C.POINTER(SafFrame), C.c_int32, C.c_size_t]),
    "saf_fuse_session_create": (C.c_void_p, []),
    "saf_fuse_session_ok": (C.c_int, [C.POINTER(SafVolume), C.POINTER(SafFrame), C.c_int32, C.c_size_t]),
    "saf_fuse_session_push": (C.c_int, [C.c_void_p, C.POINTER(SafVolume), C.POINTER(SafFrame), C.c_int32, _fp, C.c_size_t, _fp, _fp, _fp, _fp]),
    "saf_fuse_session_prepare": (C.c_int, [C.c_void_p, C.POINTER(SafVolume), C.POINTER(SafFrame), C.c_int32, _fp, C.c_size_t, _fp]),
    "saf_fuse_session_finish": (C.c_int, [C.c_void_p, _fp]),
    "saf_fuse_session_abandon": (C.c_int, [C.c_void_p]),
    "saf_fuse_session_pending": (C.c_int, [C.c_void_p]),
    "saf_fuse_session_destroy": (None, [C.c_void_p]),
    "saf_clear_unwritten_rows": (C.c_int, [C.POINTER(SafVolume), C.c_int64, C.c_int64, _fp]),
    "saf_profiler_create": (_fp, [C.c_int32]),
    "saf_profiler_destroy": (None, [_fp]),
    "saf_profiler_reset": (None, [_fp]),
    "saf_profiler_set_stride": (None, [_fp, C.c_int32]),
    "saf_profiler_read": (C.c_int, [_fp, C.c_int32, C.POINTER(C.c_double), C.POINTER(C.c_int64)]),
    "saf_fuse_frames_profiled": (
        C.c_int,
        [C.POINTER(SafVolume), C.POINTER(SafFrame), C.c_int32, _fp, C.c_size_t, _fp, _fp, _fp],
    ),
    "saf_fuse_frames_recycled": (
        C.c_int,
        [C.POINTER(SafVolume), C.POINTER(SafFrame), C.c_int32, _fp, C.c_size_t, _fp, _fp, _fp],
    ),
    "saf_backproject_lattice": (
        C.c_int,
        [_fp, C.c_int32, C.c_int32, _fp, _fp, _fp, C.c_int32, _fp, C.c_int32, C.c_float, _fp, _fp, _fp],
    ),
    "saf_query_scan": (
        C.c_int,
        [
            _fp, C.c_int32, C.c_int64, C.c_int64, C.c_int32,
            _fp, C.c_int32, C.c_int64,
            C.c_int32, C.c_float, C.c_int32, _fp, _fp, _fp, C.c_size_t, _fp,
        ],
    ),
    "saf_query_workspace_bytes": (C.c_size_t, [C.c_int32, C.c_int32]),
    "saf_query_wide_workspace_bytes": (C.c_size_t, [C.c_int32, C.c_int32]),
    "saf_query_scan_wide": (
        C.c_int,
        [
            _fp, C.c_int32, C.c_int64, C.c_int64, C.c_int32,
            _fp, C.c_int32, C.c_int64,
            C.c_float, C.c_int32, _fp, C.c_int32, C.c_int64, _fp, C.c_size_t, _fp,
        ],
    ),
    "saf_query_wide_ex_workspace_bytes": (C.c_size_t, [C.c_int32, C.c_int32, C.c_int32, C.c_int32]),
    "saf_query_scan_wide_ex": (
        C.c_int,
        [
            _fp, C.c_int32, C.c_int64, C.c_int64, C.c_int32,
            _fp, C.c_int32, C.c_int64, C.c_float, C.c_int32,
            C.c_int32, C.c_int32, C.c_int32, _fp, C.c_int32, C.c_int64, _fp, _fp, _fp, C.c_int64,
            _fp, C.c_size_t, _fp,
        ],
    ),
    "saf_merge_finalize": (C.c_int, [C.POINTER(SafVolume), C.c_int64, C.c_int64, _fp]),
    "saf_mean_to_sum": (C.c_int, [C.POINTER(SafVolume), C.c_int64, C.c_int64, _fp]),
    "saf_merge_scan_workspace_bytes": (C.c_size_t, [C.c_int64, C.c_int32]),
    "saf_merge_scan_touched": (C.c_int, [_fp, C.c_int64, _fp, _fp, C.c_int32, _fp, _fp, C.c_size_t, _fp]),
    "saf_merge_pack_rows": (C.c_int, [_fp, C.c_int64, _fp, _fp, C.c_int64, C.c_int64, _fp, _fp]),
    "saf_merge_add_packed": (C.c_int, [_fp, C.c_int64, C.c_int32, _fp, _fp, C.c_int64, C.c_int64, _fp, C.c_int64, C.c_int32, _fp]),
    "saf_label_argmax": (C.c_int, [_fp, C.c_int64, C.c_int32, _fp, _fp]),
    "saf_sample_vertices": (C.c_int, [C.POINTER(SafVolume), _fp, C.c_int64, _fp, _fp, _fp, _fp, _fp, _fp, _fp]),
    "saf_clip_tiles": (
        C.c_int,
        [_fp, C.c_int32, C.c_int32, C.c_int32, C.c_int64, C.c_int64, C.c_int64, C.c_int64, C.c_int32, C.c_int32, C.c_int32,
         C.POINTER(C.c_float), C.POINTER(C.c_float), _fp, C.c_int32, _fp],
    ),
    "saf_save_npy": (C.c_int, [_fp, C.c_int32, C.c_int32, C.POINTER(C.c_int64), C.c_int32, C.c_char_p, _fp]),
    "saf_dwconv7x7_nhwc": (C.c_int, [_fp, _fp, _fp, _fp, C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_int32, _fp]),
    "saf_mesh_json": (C.c_int, [_fp, C.c_int64, _fp, C.c_int64, _fp, C.c_int32, C.POINTER(C.c_void_p), C.POINTER(C.c_int64)]),
    "saf_array_json": (C.c_int, [_fp, C.c_int32, C.c_int64, C.c_int32, C.POINTER(C.c_void_p), C.POINTER(C.c_int64)]),
    "saf_free": (None, [C.c_void_p]),
    "saf_save_ply": (C.c_int, [C.c_char_p, _fp, C.c_int64, _fp, C.c_int64, _fp, C.c_int32]),
    "saf_marching_cubes_workspace_bytes": (C.c_size_t, [C.c_int64]),
    "saf_marching_cubes_count": (C.c_int, [_fp, _fp, C.c_int32, C.c_int32, C.c_int32, C.c_float, _fp, C.c_size_t, _fp, _fp]),
    "saf_marching_cubes_emit": (
        C.c_int, [_fp, _fp, C.c_int32, C.c_int32, C.c_int32, C.c_float, _fp, C.c_size_t, _fp, C.c_int64, _fp, C.c_int64, _fp]),
    "saf_label_components_workspace_bytes": (C.c_size_t, [C.c_int64]),
    "saf_label_components": (
        C.c_int,
        [_fp, C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_int32, _fp, _fp, C.c_int32, _fp, _fp, _fp, _fp, C.c_size_t, _fp],
    ),
}


def declare(lib, prototypes=PROTOTYPES):
    """Attach restype/argtypes; raises AttributeError if a symbol is missing."""
    for name, (res, args) in prototypes.items():
        fn = getattr(lib, name)
        fn.restype = res
        fn.argtypes = args
    return lib


def ptr(t):
    """Address of a torch tensor's / numpy array's first element (or None)."""
    if t is None:
        return None
    if hasattr(t, "data_ptr"):
        return t.data_ptr()
    return t.ctypes.data
