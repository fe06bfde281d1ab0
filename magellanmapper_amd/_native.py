"""ctypes binding of ``libmmx_hip.so`` (the C ABI declared in ``include/mmx.h``).

There is NO CPU fallback: if the shared library is missing or a call fails the
functions raise.  Build the library with ``python -c "import __graft_entry__ as g; g.build()"``
or ``make -C magellanmapper_amd/csrc -j8``.
"""
from __future__ import annotations

import ctypes
import os
from ctypes import (POINTER, Structure, c_char_p, c_double, c_float, c_int, c_int32, c_int64,
                    c_uint8, c_uint32, c_void_p)

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
#: ``MMX_LIB_PATH`` selects an experimental build of the same ABI (kernel tuning only)
LIB_PATH = os.environ.get("MMX_LIB_PATH") or os.path.join(_HERE, "libmmx_hip.so")

MMX_ABI_VERSION = 16
MMX_U8, MMX_U16, MMX_F32, MMX_F64 = 0, 1, 2, 3
MMX_MAX_RADIUS_FAST = 24
MMX_MAX_RADIUS_GENERIC = 255
MMX_CAND_CONTESTED = 1
MMX_CAND_BAND = 2
MMX_CAND_PROBE = 4
#: ``mmx_zx_mode``: how mmx_log_batch_f32 runs its Z and X passes (a per-call argument)
MMX_ZX_AUTO, MMX_ZX_SEPARATE, MMX_ZX_PACKED, MMX_ZX_MFMA_F32, MMX_ZX_MFMA_F16, MMX_ZX_MFMA_F16_LDS = -1, 0, 2, 3, 4, 5
MMX_ZX_TILED, MMX_ZX_TILED_Q16, MMX_ZX_PREPACKED, MMX_ZX_Y_VALU = 6, 7, 0x100, 0x200
#: NMS entry layouts ``mmx_log_batch_f32`` reports and ``mmx_peaks_batch`` takes
MMX_MASK_ROWS, MMX_MASK_QUADS = 1, 2
MMX_MAX_BLOCKS = 65535
MMX_COLOC_BALL = 33

#: NumPy mirror of ``mmx_block`` (32 bytes).
BLOCK_DTYPE = np.dtype([("src_off", "<i8"), ("nz", "<i4"), ("ny", "<i4"), ("nx", "<i4"),
                        ("slot", "<i4"), ("px", "<i4"), ("_pad", "<i4")], align=True)
MMX_ROW_ALIGN = 32
#: NumPy mirror of ``mmx_cand`` (48 bytes).
CAND_DTYPE = np.dtype([("slot", "<i4"), ("s", "<i4"), ("z", "<i4"), ("y", "<i4"), ("x", "<i4"),
                       ("flags", "<u4"), ("v", "<f4"), ("nbr_max", "<f4"), ("v64", "<f8"),
                       ("band", "<u8")], align=True)
assert BLOCK_DTYPE.itemsize == 32 and CAND_DTYPE.itemsize == 48
#: NumPy mirrors of ``mmx_subblock`` (40 bytes), ``mmx_quantile_class`` (32), ``mmx_subblock_info`` (32)
SUBBLOCK_DTYPE = np.dtype([("src_off", "<i8"), ("dst_off", "<i8"), ("scratch_off", "<i8"),
                           ("nz", "<i4"), ("ny", "<i4"), ("nx", "<i4"), ("qclass", "<i4")], align=True)
QCLASS_DTYPE = np.dtype([("lo_prev", "<i4"), ("lo_next", "<i4"), ("hi_prev", "<i4"), ("hi_next", "<i4"),
                         ("lo_gamma", "<f8"), ("hi_gamma", "<f8")], align=True)
SUBINFO_DTYPE = np.dtype([("vmin", "<f8"), ("vmax", "<f8"), ("mean", "<f8"), ("flags", "<i4"),
                          ("_pad", "<i4")], align=True)
assert SUBBLOCK_DTYPE.itemsize == 40 and QCLASS_DTYPE.itemsize == 32 and SUBINFO_DTYPE.itemsize == 32
MMX_PP_IDENTITY, MMX_PP_ERODED, MMX_PP_EXACT_MEAN = 1, 2, 4
MMX_PP_AUTO, MMX_PP_SINGLE, MMX_PP_PIPELINED = 0, 1, 2     # kernel choice of mmx_preprocess_batch_mode
#: NumPy mirror of ``mmx_resize_block`` (48 bytes)
RESIZE_DTYPE = np.dtype([("src_off", "<i8"), ("in_nz", "<i4"), ("in_ny", "<i4"), ("in_nx", "<i4"),
                         ("out_nz", "<i4"), ("out_ny", "<i4"), ("out_nx", "<i4"), ("slot", "<i4"),
                         ("tz", "<i4"), ("ty", "<i4"), ("tx", "<i4")], align=True)
assert RESIZE_DTYPE.itemsize == 48


class Volume(Structure):
    """``mmx_volume``."""
    _fields_ = [("d_data", c_void_p), ("dtype", c_int32), ("value_range", c_float),
                ("stride_z", c_int64), ("stride_y", c_int64), ("stride_x", c_int64)]


class DetectArgs(Structure):
    """``mmx_detect_args`` (one batch from voxels to the re-scored candidate table: ``mmx_detect_batch``)."""
    _fields_ = [("vol32", POINTER(Volume)), ("vol_exact", POINTER(Volume)), ("d_blocks", c_void_p),
                ("h_blocks", c_void_p), ("n_blocks", c_int32), ("n_sigma", c_int32), ("slot_elems", c_int64),
                ("h_w0", c_void_p), ("h_w2", c_void_p), ("d_w0", c_void_p), ("d_w2", c_void_p),
                ("h_radius", c_void_p), ("h_norm", c_void_p), ("d_work", c_void_p), ("work_bytes", ctypes.c_size_t),
                ("thr", c_float), ("eps", c_float), ("d_cands", c_void_p), ("cap", c_uint32), ("h_prefix", c_uint32),
                ("d_count", c_void_p), ("h_count", c_void_p), ("h_cands", c_void_p),
                ("zx_mode", c_int32), ("zx_flags", c_int32), ("store_f32", c_int32), ("exact", c_int32),
                ("expand", c_int32), ("_pad", c_int32),
                ("stream", c_void_p), ("tail_stream", c_void_p), ("pack_stream", c_void_p),
                ("ev_work_free", c_void_p), ("ev_work_read", c_void_p), ("ev_done", c_void_p)]


class FinishStackArgs(Structure):
    """``mmx_finish_stack_args`` (a one-batch stack from the re-scored candidates to its final table: ``mmx_host_finish_stack``)."""
    _fields_ = [("cands", c_void_p), ("n_cands", c_uint32), ("n_total", c_uint32),
                ("blocks", c_void_p), ("n_blocks", c_int32), ("n_sigma", c_int32),
                ("thr", c_double), ("eps", c_double),
                ("sigmas", c_void_p), ("overlap", c_double), ("overlap_band", c_double),
                ("channel", c_double),
                ("block_offsets", c_void_p), ("block_tags", c_void_p), ("interior", c_void_p),
                ("store", c_void_p), ("ld", c_int64), ("zyx", c_void_p), ("tag", c_void_p), ("abs_zyx", c_void_p),
                ("capacity", c_int64), ("rows_per_block", c_void_p), ("any_before", c_void_p),
                ("n_sections", c_void_p), ("bounds", c_void_p), ("last_end", c_void_p), ("tol", c_void_p),
                ("nxt_lo", c_void_p), ("nxt_hi", c_void_p),
                ("n_slab", c_void_p), ("n_after", c_void_p), ("n_next", c_void_p), ("stat_ld", c_int64),
                ("src_cols", c_void_p), ("n_out", c_int32), ("abs_dst0", c_int32),
                ("out", c_void_p), ("out_capacity", c_int64), ("out_rows", c_void_p),
                ("stats", c_void_p)]


MMX_DEFERRED = 6


class DetectInfo(Structure):
    """``mmx_detect_info``."""
    _fields_ = [("zx_path", c_int32), ("mask_layout", c_int32), ("n_pass_rounds", c_int32), ("_pad", c_int32),
                ("q16_bound", c_double)]


class PreprocParams(Structure):
    """``mmx_preproc_params``."""
    _fields_ = [("clip_min", c_double), ("clip_max", c_double), ("max_thresh", c_double),
                ("unsharp_strength", c_double), ("erosion_threshold", c_double),
                ("radius", c_int32), ("rgb_guess", c_int32), ("tv_weight", c_double), ("tv_factor", c_double)]


class MmxError(RuntimeError):
    """A C-ABI call returned a non-zero status."""


_lib = None

#: every symbol ``include/mmx.h`` declares
SYMBOLS = (
    "mmx_abi_version", "mmx_strerror", "mmx_last_hip_error", "mmx_device_count",
    "mmx_detect_batch", "mmx_detect_batch_capture", "mmx_graph_launch", "mmx_graph_destroy", "mmx_detect_last_error",
    "mmx_event_synchronize", "mmx_stream_wait_event", "mmx_timing_is_enabled",
    "mmx_log_batch_f32", "mmx_log_batch_f32_generic", "mmx_zx_pack", "mmx_tiled_q16_error_bound", "mmx_workspace_bytes", "mmx_peaks_batch", "mmx_rescore_f64",
    "mmx_overlap_pairs", "mmx_close_pairs", "mmx_event_create", "mmx_event_destroy",
    "mmx_event_record", "mmx_event_elapsed_ms", "mmx_timing_enable", "mmx_timing_read",
    "mmx_calib_stream", "mmx_host_prune_axis",
    "mmx_preprocess_fast_lds", "mmx_preprocess_batch", "mmx_preprocess_batch_mode", "mmx_preprocess_work_bytes", "mmx_preprocess_batch_generic",
    "mmx_coloc_means", "mmx_coloc_voxels", "mmx_host_take_rows", "mmx_host_map_columns", "mmx_resize_batch_as", "mmx_gauss_axis_batch", "mmx_unmix_batch", "mmx_minmax_batch", "mmx_resize_batch",
    "mmx_cdist_f64", "mmx_host_lsap", "mmx_expand_probes", "mmx_host_resolve_peaks", "mmx_host_overlap_prune",
    "mmx_host_emit_tables", "mmx_host_prune_region", "mmx_host_prune_parts", "mmx_host_rows_in_boxes", "mmx_host_append_rows", "mmx_host_emit_survivors", "mmx_host_merge_by_key", "mmx_host_merge_parts_by_key", "mmx_host_gather_by_key", "mmx_host_take_rows_final", "mmx_host_emit_survivors_final", "mmx_host_emit_parts_final", "mmx_host_gather_parts_by_key_final", "mmx_host_take_rows_split", "mmx_host_gather_parts_by_key_split", "mmx_host_emit_tables_multi", "mmx_host_coloc_flags", "mmx_host_finish_stack", "mmx_copy_rect_h2d", "mmx_host_stage_upload", "mmx_event_query",
)
KERNEL_KINDS = ("zpass", "ypass", "xpass", "generic", "peaks", "rescore", "overlap_pairs",
                "close_pairs", "zxpass", "y2pass", "preproc", "coloc", "zxpack")


def lib() -> ctypes.CDLL:
    """Load the library once; raise loudly when it is not there."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise MmxError(
            f"{LIB_PATH} not found: the HIP extension has not been built "
            "(run __graft_entry__.build() or make -C magellanmapper_amd/csrc). "
            "There is no CPU fallback for this path.")
    # PyTorch-ROCm wheels bundle their own libamdhip64: when this library is loaded first, the loader binds it to the
    # system's copy and a process that imports torch afterwards ends up with two HIP runtimes -- every launch on a torch
    # stream then fails with "HIP runtime error" (seen with __graft_entry__.build() followed by smoke() in one
    # process).  With torch imported first the runtime already in the process satisfies the dependency.
    try:
        import torch  # noqa: F401
    except ImportError:
        pass
    L = ctypes.CDLL(LIB_PATH)
    vp = c_void_p
    L.mmx_abi_version.restype = c_int
    L.mmx_strerror.restype = c_char_p
    L.mmx_strerror.argtypes = [c_int]
    L.mmx_last_hip_error.restype = c_char_p
    L.mmx_device_count.restype = c_int
    log_args = [POINTER(Volume), vp, vp, c_int, c_int64, POINTER(c_double), POINTER(c_double),
                c_int, c_double, vp, vp]
    L.mmx_log_batch_f32.argtypes = log_args + [vp, c_float, c_float, POINTER(c_int), c_int, POINTER(c_int), vp]
    L.mmx_log_batch_f32_generic.argtypes = log_args + [vp]
    L.mmx_workspace_bytes.argtypes = [c_int, c_int64, c_int, c_int]
    L.mmx_workspace_bytes.restype = ctypes.c_size_t
    L.mmx_zx_pack.argtypes = [POINTER(Volume), vp, vp, c_int, c_int64, vp, vp]
    L.mmx_tiled_q16_error_bound.argtypes = [POINTER(c_double), POINTER(c_double), c_int, c_double]
    L.mmx_tiled_q16_error_bound.restype = c_double
    L.mmx_peaks_batch.argtypes = [vp, vp, c_int, c_int, vp, vp, c_int, c_int64, c_float, c_float, vp,
                                  c_uint32, vp, vp]
    L.mmx_rescore_f64.argtypes = [POINTER(Volume), vp, c_int, vp, c_uint32, vp, vp, vp,
                                  POINTER(c_int32), POINTER(c_double), c_int, c_int, vp]
    L.mmx_overlap_pairs.argtypes = [vp, vp, c_int, c_double, c_double, c_double, vp, vp, c_uint32, vp, vp]
    L.mmx_close_pairs.argtypes = [vp, c_int, vp, c_int, POINTER(c_int32), vp, vp, vp]
    L.mmx_detect_batch.argtypes = [POINTER(DetectArgs), POINTER(DetectInfo)]
    L.mmx_detect_batch_capture.argtypes = [POINTER(DetectArgs), POINTER(DetectInfo), POINTER(vp)]
    L.mmx_graph_launch.argtypes = [vp, vp, vp, POINTER(DetectInfo)]
    L.mmx_graph_destroy.argtypes = [vp]
    L.mmx_detect_last_error.restype = c_char_p
    L.mmx_event_synchronize.argtypes = [vp]
    L.mmx_stream_wait_event.argtypes = [vp, vp]
    L.mmx_event_create.argtypes = [POINTER(vp)]
    L.mmx_event_destroy.argtypes = [vp]
    L.mmx_event_record.argtypes = [vp, vp]
    L.mmx_event_elapsed_ms.argtypes = [vp, vp, POINTER(c_float)]
    L.mmx_timing_enable.argtypes = [c_int]
    L.mmx_timing_read.argtypes = [POINTER(c_double), POINTER(c_int64), c_int]
    L.mmx_calib_stream.argtypes = [c_int, vp, vp, c_int64, vp]
    L.mmx_host_prune_axis.argtypes = [vp, vp, vp, vp, c_int64, c_int, c_int, vp, c_double,
                                      POINTER(c_int32), vp, vp, vp, POINTER(c_int64), vp, vp, vp]
    L.mmx_preprocess_fast_lds.argtypes = [c_int, c_int, c_int]
    L.mmx_preprocess_fast_lds.restype = c_int64
    pre_args = [POINTER(Volume), vp, vp, c_int, vp, c_int, POINTER(PreprocParams), vp,
                c_int64, c_int64, vp, vp, vp]
    L.mmx_preprocess_batch.argtypes = pre_args + [vp]
    L.mmx_preprocess_batch_mode.argtypes = pre_args + [c_int, c_int, vp, c_int64, vp]
    L.mmx_preprocess_work_bytes.argtypes = [vp, c_int]
    L.mmx_preprocess_work_bytes.restype = c_int64
    L.mmx_preprocess_batch_generic.argtypes = pre_args + [vp, c_int64, vp]
    L.mmx_minmax_batch.argtypes = [POINTER(Volume), vp, vp, c_int, vp, vp]
    L.mmx_minmax_batch.restype = c_int
    L.mmx_resize_batch.argtypes = [POINTER(Volume), vp, vp, c_int, vp, vp, vp, c_int64, c_int64, c_int64,
                                   vp, vp, vp]
    L.mmx_resize_batch.restype = c_int
    L.mmx_resize_batch_as.argtypes = [POINTER(Volume), vp, vp, c_int, vp, vp, vp, c_int64, c_int64, c_int64,
                                      c_int, vp, vp, vp]
    L.mmx_resize_batch_as.restype = c_int
    L.mmx_gauss_axis_batch.argtypes = [POINTER(Volume), vp, vp, c_int, c_int, vp, vp, c_int, c_int,
                                       c_int64, c_int64, c_int64, vp, vp]
    L.mmx_gauss_axis_batch.restype = c_int
    L.mmx_unmix_batch.argtypes = [POINTER(Volume), POINTER(Volume), POINTER(c_double), c_int, vp, vp, c_int,
                                  c_int64, c_int64, c_int64, vp, vp, vp]
    L.mmx_unmix_batch.restype = c_int
    L.mmx_host_take_rows.argtypes = [vp, c_int64, vp, c_int64, c_int64, vp, POINTER(c_int32), vp]
    L.mmx_host_map_columns.argtypes = [vp, c_int64, c_int64, POINTER(c_int32), c_int32, vp, c_int64, c_int32]
    L.mmx_cdist_f64.argtypes = [vp, c_int64, vp, c_int64, c_int, vp, vp]
    L.mmx_cdist_f64.restype = c_int
    L.mmx_host_lsap.argtypes = [vp, c_int64, c_int64, vp, vp]
    L.mmx_host_lsap.restype = c_int
    L.mmx_coloc_means.argtypes = [POINTER(Volume), vp, c_int, vp, vp, c_int, vp, vp, vp]
    L.mmx_coloc_means.restype = c_int
    L.mmx_coloc_voxels.argtypes = [POINTER(Volume), vp, c_int, vp, vp, c_int, vp, vp, vp, vp]
    L.mmx_coloc_voxels.restype = c_int
    L.mmx_expand_probes.argtypes = [vp, c_uint32, vp, vp, vp, c_int, c_int, vp]
    L.mmx_expand_probes.restype = c_int
    L.mmx_host_resolve_peaks.argtypes = [vp, c_uint32, c_uint32, vp, c_int, c_int, c_double, vp, vp, vp, vp, vp, vp, vp]
    L.mmx_host_overlap_prune.argtypes = [vp, vp, c_int, vp, c_int, c_double, c_double, vp, vp, vp, vp, c_int64,
                                         POINTER(c_int64), POINTER(c_int64)]
    L.mmx_host_prune_region.argtypes = [vp, vp, vp, vp, c_int64, c_int64, c_int64, POINTER(c_int32), POINTER(vp),
                                        POINTER(c_double), POINTER(c_int32), POINTER(vp), POINTER(vp), vp, vp,
                                        POINTER(c_int64), vp, vp, vp, c_int64]
    L.mmx_host_prune_parts.argtypes = [vp, vp, vp, vp, c_int64, vp, c_int, c_int, POINTER(c_int32), POINTER(c_int32),
                                       vp, c_int, POINTER(c_int32), POINTER(vp), POINTER(c_double), POINTER(c_int32),
                                       POINTER(vp), POINTER(vp), c_int64, vp, vp, vp, POINTER(c_int64), vp, vp, vp,
                                       c_int64]
    L.mmx_host_rows_in_boxes.argtypes = [vp, vp, vp, vp, c_int64, c_int64, vp, vp, c_int, vp, c_int64, POINTER(c_int64)]
    L.mmx_host_append_rows.argtypes = [vp, c_int64, POINTER(c_int32), POINTER(c_int32), vp, vp, vp, vp, c_int64,
                                       c_int64, c_int64, POINTER(c_int64)]
    L.mmx_host_emit_survivors.argtypes = [vp, c_int64, vp, vp, c_int64, c_int64, vp, POINTER(c_int32), vp]
    L.mmx_host_merge_by_key.argtypes = [vp, c_int64, vp, c_int64, c_int64, c_int64, vp]
    L.mmx_host_merge_parts_by_key.argtypes = [vp, vp, c_int32, c_int64, c_int64, c_int64, vp, c_int64]
    L.mmx_host_gather_by_key.argtypes = [vp, c_int64, vp, vp, c_int64, c_int64, c_int64, vp, POINTER(c_int32), vp]
    L.mmx_host_emit_parts_final.argtypes = [vp, c_int64, c_int32, vp, vp, vp, vp, POINTER(c_int32), c_int32, c_int32, vp, c_int64]
    L.mmx_host_emit_survivors_final.argtypes = [vp, c_int64, vp, vp, c_int64, POINTER(c_int32), c_int32, vp, c_int32, vp]
    L.mmx_host_take_rows_final.argtypes = [vp, c_int64, vp, c_int64, POINTER(c_int32), c_int32, vp, c_int32, vp]
    L.mmx_host_gather_parts_by_key_final.argtypes = [vp, c_int64, c_int32, vp, vp, vp, vp, c_int64, POINTER(c_int32), c_int32,
                                                     c_int32, vp, c_int64]
    L.mmx_host_emit_tables_multi.argtypes = [c_int32, vp, vp, vp, c_int, vp, vp, vp, vp, vp, vp, vp, c_int64, c_int32, vp, vp,
                                             vp, c_int64, c_int64, vp, vp, vp]
    L.mmx_host_coloc_flags.argtypes = [vp, vp, c_int32, c_int64, vp, vp, c_int, vp, c_int32, vp, c_int64]
    L.mmx_host_finish_stack.argtypes = [POINTER(FinishStackArgs)]
    L.mmx_host_stage_upload.argtypes = [vp, vp, vp, c_int32, c_int64, c_int64, c_int64, vp, c_int64, c_int32, vp, vp, c_int32,
                                        vp, vp, c_int32]
    L.mmx_event_query.argtypes = [vp]
    L.mmx_copy_rect_h2d.argtypes = [vp, ctypes.c_size_t, vp, ctypes.c_size_t, ctypes.c_size_t, ctypes.c_size_t, vp]
    L.mmx_host_take_rows_split.argtypes = [vp, c_int64, vp, c_int64, POINTER(c_int32), c_int32, vp, c_int32, vp, c_int32, vp]
    L.mmx_host_gather_parts_by_key_split.argtypes = [vp, c_int64, c_int32, vp, vp, vp, vp, c_int64, POINTER(c_int32), c_int32,
                                                     c_int32, vp, c_int64, c_int32, vp]
    L.mmx_host_emit_tables.argtypes = [vp, vp, vp, c_int, vp, c_int, c_double, vp, vp, vp, vp, c_int64, vp, vp, vp,
                                       c_int64, c_int64, vp]
    L.mmx_preprocess_batch.restype = c_int
    L.mmx_preprocess_batch_mode.restype = c_int
    L.mmx_preprocess_batch_generic.restype = c_int
    for name in SYMBOLS:
        fn = getattr(L, name)
        if name in ("mmx_preprocess_fast_lds", "mmx_workspace_bytes", "mmx_preprocess_work_bytes"):
            continue
        if fn.restype is None or name.startswith(("mmx_log", "mmx_peaks", "mmx_rescore",
                                                  "mmx_overlap", "mmx_close", "mmx_event", "mmx_timing", "mmx_calib", "mmx_host")):
            fn.restype = c_int
    if L.mmx_abi_version() != MMX_ABI_VERSION:
        raise MmxError("libmmx_hip.so ABI version mismatch")
    _lib = L
    return L


def check(status: int, what: str) -> None:
    if status != 0:
        L = lib()
        msg = L.mmx_strerror(status).decode()
        hip = L.mmx_last_hip_error().decode()
        raise MmxError(f"{what}: {msg}" + (f" ({hip})" if hip else ""))


def as_double_ptr(a: np.ndarray):
    assert a.dtype == np.float64 and a.flags.c_contiguous
    return a.ctypes.data_as(POINTER(c_double))


def as_int32_ptr(a: np.ndarray):
    assert a.dtype == np.int32 and a.flags.c_contiguous
    return a.ctypes.data_as(POINTER(c_int32))


def timing_enable(on: bool = True, kinds=None) -> None:
    """Per-kernel HIP-event timing on / off; ``kinds`` (names from ``KERNEL_KINDS``): only those families."""
    mask = 1 if on else 0
    if on and kinds is not None:
        mask = 0
        for k in kinds:
            mask |= 1 << (KERNEL_KINDS.index(k) + 1)
    check(lib().mmx_timing_enable(mask), "mmx_timing_enable")


def timing_read() -> dict:
    """``{kind: (milliseconds, launches)}`` since the last read (synchronises the events)."""
    ms = (c_double * len(KERNEL_KINDS))()
    n = (c_int64 * len(KERNEL_KINDS))()
    check(lib().mmx_timing_read(ms, n, len(KERNEL_KINDS)), "mmx_timing_read")
    return {k: (ms[i], n[i]) for i, k in enumerate(KERNEL_KINDS)}


def keep_host_heap(mmap_threshold: int = 1 << 30, trim_threshold: int = (1 << 31) - 1) -> bool:
    """glibc ``mallopt``: serve large host arrays from the heap and keep freed heap memory mapped.

    A whole-volume detection builds tables of tens of MB per call (the merged blob table, its compact
    columns, the pruned output).  By default glibc ``mmap``s every allocation above 32 MiB: each call then pays
    page faults for fresh zero pages and an ``munmap`` when the arrays die -- ~8 ms per 2048 x 2048 x 1024 volume.
    Long-running detection processes (and ``bench.py``) call this once; it changes the allocator of the WHOLE
    process (memory is returned to the OS later), which is why importing the package does not do it.
    Returns False where ``mallopt`` is not available (non-glibc)."""
    try:
        libc = ctypes.CDLL("libc.so.6")
        ok1 = libc.mallopt(-3, int(mmap_threshold))      # M_MMAP_THRESHOLD
        ok2 = libc.mallopt(-1, int(trim_threshold))      # M_TRIM_THRESHOLD
        return bool(ok1 and ok2)
    except (OSError, AttributeError):
        return False
