"""ctypes binding of libbulklmm_hip.so (include/bulklmm_hip.h).

There is no CPU fallback: if the HIP library is missing or cannot be loaded, importing this module
raises.  `build()` compiles it in-tree with hipcc for gfx950 (works without a GPU)."""
from __future__ import annotations

import ctypes as C
import os
import subprocess

_HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(_HERE, "csrc")
LIB_PATH = os.path.join(CSRC, "libbulklmm_hip.so")

# every symbol include/bulklmm_hip.h declares
EXPORTS = [
    "blmm_version", "blmm_device_count", "blmm_create", "blmm_destroy", "blmm_last_error", "blmm_err_string",
    "blmm_set_stream", "blmm_set_timing", "blmm_read_timings", "blmm_lowrank_profile", "blmm_synchronize", "blmm_default_opts",
    "blmm_kinship", "blmm_kinship_dev", "blmm_bulkscan", "blmm_bulkscan_dev", "blmm_scan_perms",
    "blmm_scan_perms_dev", "blmm_scan_perms_f32", "blmm_scan_perms_f32_dev", "blmm_lod_colmax", "blmm_lod_colmax_dev", "blmm_rotate", "blmm_null_h2_brent", "blmm_null_loglik_grid",
    "blmm_weighted_liteqtl", "blmm_liteqtl_given_h2",
    "blmm_create_multi", "blmm_destroy_multi", "blmm_multi_ndev", "blmm_multi_last_error", "blmm_default_multi_opts",
    "blmm_multi_shard", "blmm_bulkscan_multi", "blmm_multi_device_result",
    "blmm_host_register", "blmm_host_unregister", "blmm_host_alloc", "blmm_host_free",
    "blmm_lod2log10p", "blmm_lod2log10p_dev", "blmm_lod_threshold", "blmm_lod_threshold_dev", "blmm_get_thresholds", "blmm_get_thresholds_dev",
    "blmm_last_log10p", "blmm_last_lod_threshold", "blmm_last_get_thresholds", "blmm_set_log10p_output",
    "blmm_read_csv", "blmm_read_he", "blmm_table_rows", "blmm_table_cols", "blmm_table_copy", "blmm_table_free",
    "blmm_kinship_rounded", "blmm_scan_alt", "blmm_scan_alt_dev", "blmm_bulkscan_alt_exact", "blmm_bulkscan_alt_exact_dev",
    "blmm_prepare_dev", "blmm_rotated_rows", "blmm_rotate_block_dev", "blmm_bulkscan_prerotated_dev", "blmm_scan_perms_prerotated_dev",
    "blmm_set_tuning", "blmm_get_tuning", "blmm_lowrank_columns", "blmm_bulkscan_reduced", "blmm_bulkscan_reduced_dev", "blmm_last_reduced_route",
    "blmm_last_dims", "blmm_last_lod_colmax", "blmm_last_lod_columns", "blmm_multi_last_colmax", "blmm_multi_last_lod_threshold",
]

BLMM_NULL_EXACT, BLMM_NULL_GRID, BLMM_ALT_GRID = 0, 1, 2
BLMM_EIGEN, BLMM_SVD = 0, 1
BLMM_COMPAT_ALT_COUNTER = 1
BLMM_COMPAT_ALT_TRUE_WEIGHTS = 2
BLMM_FLAG_H2_AUDIT = 4
BLMM_GATHER_NONE, BLMM_GATHER_HOST_SHARDS, BLMM_GATHER_ALLGATHER = 0, 1, 2

ERR_ZERO_NORM_MSG = "Dividing by zeros: the input vector can not contain any zeros!"


class blmm_opts(C.Structure):
    _fields_ = [("method", C.c_int32), ("reml", C.c_int32), ("add_intercept", C.c_int32), ("decomp_scheme", C.c_int32),
                ("optim_interval", C.c_int32), ("compat_flags", C.c_int32), ("prior_variance", C.c_double),
                ("prior_sample_size", C.c_double)]


class blmm_multi_opts(C.Structure):
    _fields_ = [("gather_mode", C.c_int32), ("reserved", C.c_int32)]


class blmm_status(C.Structure):
    _fields_ = [("n_neg_eig", C.c_int64), ("n_nonpos_weight", C.c_int64), ("n_zero_norm", C.c_int64),
                ("n_nan_lod", C.c_int64), ("n_brent_maxiter", C.c_int64), ("jacobi_sweeps", C.c_int64),
                ("jacobi_cycles", C.c_int64), ("jacobi_ticks_100mhz", C.c_int64),
                ("lowrank_rank", C.c_int64), ("lowrank_fallback", C.c_int64), ("lowrank_shared", C.c_int64), ("lowrank_resid", C.c_double),
                ("t_eigen_ms", C.c_double), ("t_rotate_ms", C.c_double), ("t_h2_ms", C.c_double),
                ("t_prep_ms", C.c_double), ("t_scan_ms", C.c_double), ("t_total_ms", C.c_double),
                ("n_h2_boundary", C.c_int64), ("n_h2_multimodal", C.c_int64), ("n_illcond_rescan", C.c_int64)]


class blmm_reduced(C.Structure):
    """blmm_bulkscan_reduced[_dev]: pointers as integers (host addresses for the host form, device addresses for _dev)."""
    _fields_ = [("colmax", C.c_void_p), ("argmax", C.c_void_p), ("want_triplets", C.c_int64), ("thr", C.c_double),
                ("cap", C.c_int64), ("ti", C.c_void_p), ("tj", C.c_void_p), ("tlod", C.c_void_p), ("count", C.c_void_p)]


def build(force: bool = False) -> str:
    """Compile libbulklmm_hip.so in-tree (hipcc --offload-arch=gfx950).  Returns the library path."""
    srcs = [os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith((".hip", ".h"))]
    srcs.append(os.path.join(_HERE, "..", "include", "bulklmm_hip.h"))
    if not force and os.path.exists(LIB_PATH):
        lib_m = os.path.getmtime(LIB_PATH)
        if all(os.path.getmtime(s) <= lib_m for s in srcs if os.path.exists(s)):
            return LIB_PATH
    r = subprocess.run(["make", "-C", CSRC, "-j4"], capture_output=True, text=True)
    if r.returncode != 0:
        raise RuntimeError("building libbulklmm_hip.so failed:\n" + r.stdout[-4000:] + r.stderr[-4000:])
    return LIB_PATH


_lib = None


def load():
    """dlopen the library and declare prototypes.  Raises OSError if it is not built."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise OSError(f"{LIB_PATH} is missing: run `python -c 'import __graft_entry__ as g; g.build()'` "
                      "(there is no CPU fallback for the bulkscan path)")
    lib = C.CDLL(LIB_PATH)
    dp, ip = C.POINTER(C.c_double), C.POINTER(C.c_int32)
    i64, vp = C.c_int64, C.c_void_p
    op, sp = C.POINTER(blmm_opts), C.POINTER(blmm_status)
    lib.blmm_version.restype = C.c_int
    lib.blmm_device_count.restype = C.c_int
    lib.blmm_create.argtypes = [C.c_int, vp, C.POINTER(vp)]
    lib.blmm_destroy.argtypes = [vp]
    lib.blmm_destroy.restype = None
    lib.blmm_last_error.argtypes = [vp]
    lib.blmm_last_error.restype = C.c_char_p
    lib.blmm_err_string.argtypes = [C.c_int]
    lib.blmm_err_string.restype = C.c_char_p
    lib.blmm_set_stream.argtypes = [vp, vp]
    lib.blmm_set_timing.argtypes = [vp, C.c_int]
    lib.blmm_synchronize.argtypes = [vp]
    lib.blmm_read_timings.argtypes = [vp, dp, C.POINTER(C.c_int64)]
    lib.blmm_lowrank_profile.argtypes = [vp, C.POINTER(C.c_int64)]
    lib.blmm_default_opts.argtypes = [op]
    lib.blmm_default_opts.restype = None
    # data pointers are passed as void* so that both host (numpy) and device (torch data_ptr) addresses fit
    lib.blmm_kinship.argtypes = [vp, vp, i64, i64, vp]
    lib.blmm_kinship_dev.argtypes = [vp, vp, i64, i64, vp]
    lib.blmm_bulkscan.argtypes = [vp, op, vp, i64, i64, vp, i64, vp, i64, vp, vp, vp, i64, vp, vp, sp]
    lib.blmm_bulkscan_dev.argtypes = [vp, op, vp, i64, i64, vp, i64, vp, i64, vp, vp, vp, i64, vp, i64, vp, sp]
    lib.blmm_scan_perms.argtypes = [vp, op, vp, i64, vp, i64, vp, i64, vp, vp, i64, C.c_uint64, vp, vp, vp, vp, sp]
    lib.blmm_scan_perms_dev.argtypes = [vp, op, vp, i64, vp, i64, vp, i64, vp, vp, i64, C.c_uint64, vp, vp, vp, vp, sp]
    lib.blmm_scan_perms_f32.argtypes = [vp, op, vp, i64, vp, i64, vp, i64, vp, vp, i64, C.c_uint64, vp, vp, vp, vp, sp]
    lib.blmm_scan_perms_f32_dev.argtypes = [vp, op, vp, i64, vp, i64, vp, i64, vp, vp, i64, C.c_uint64, vp, vp, vp, vp, sp]
    lib.blmm_scan_alt.argtypes = [vp, op, vp, i64, vp, i64, vp, i64, vp, vp, vp, vp, vp, sp]
    lib.blmm_scan_alt_dev.argtypes = [vp, op, vp, i64, vp, i64, vp, i64, vp, vp, vp, vp, vp, sp]
    lib.blmm_bulkscan_alt_exact.argtypes = [vp, C.POINTER(blmm_opts), vp, i64, i64, vp, i64, vp, i64, vp, vp, vp, vp, vp, vp, C.POINTER(blmm_status)]
    lib.blmm_bulkscan_alt_exact_dev.argtypes = [vp, C.POINTER(blmm_opts), vp, i64, i64, vp, i64, vp, i64, vp, vp, vp, i64, vp, i64, vp, vp, C.POINTER(blmm_status)]
    lib.blmm_prepare_dev.argtypes = [vp, op, i64, vp, i64, vp, vp, sp]
    lib.blmm_rotated_rows.argtypes = [vp]
    lib.blmm_rotated_rows.restype = i64
    lib.blmm_rotate_block_dev.argtypes = [vp, vp, i64, vp, i64]
    lib.blmm_bulkscan_prerotated_dev.argtypes = [vp, op, vp, i64, i64, vp, i64, i64, i64, vp, i64, vp, i64, vp, sp]
    lib.blmm_scan_perms_prerotated_dev.argtypes = [vp, op, vp, i64, vp, i64, i64, i64, i64, C.c_uint64, vp, vp, vp, vp, vp, sp]
    lib.blmm_lod_colmax.argtypes = [vp, vp, i64, i64, vp, vp]
    lib.blmm_lod_colmax_dev.argtypes = [vp, vp, i64, i64, i64, vp, vp]
    lib.blmm_rotate.argtypes = [vp, op, vp, i64, i64, vp, i64, vp, i64, vp, vp, vp, vp, sp]
    lib.blmm_null_h2_brent.argtypes = [vp, op, vp, i64, i64, vp, i64, vp, vp, vp, vp, sp]
    lib.blmm_null_loglik_grid.argtypes = [vp, op, vp, i64, i64, vp, i64, vp, vp, i64, vp, sp]
    lib.blmm_weighted_liteqtl.argtypes = [vp, vp, i64, i64, vp, i64, i64, vp, C.c_double, vp, sp]
    lib.blmm_liteqtl_given_h2.argtypes = [vp, vp, i64, i64, vp, i64, i64, vp, vp, vp, sp]
    mp = C.POINTER(blmm_multi_opts)
    lib.blmm_create_multi.argtypes = [ip, C.c_int, C.POINTER(vp)]
    lib.blmm_destroy_multi.argtypes = [vp]
    lib.blmm_destroy_multi.restype = None
    lib.blmm_multi_ndev.argtypes = [vp]
    lib.blmm_multi_last_error.argtypes = [vp]
    lib.blmm_multi_last_error.restype = C.c_char_p
    lib.blmm_default_multi_opts.argtypes = [mp]
    lib.blmm_default_multi_opts.restype = None
    lib.blmm_multi_shard.argtypes = [i64, C.c_int, C.c_int, C.POINTER(i64), C.POINTER(i64)]
    lib.blmm_multi_shard.restype = None
    lib.blmm_bulkscan_multi.argtypes = [vp, op, mp, vp, i64, i64, vp, i64, vp, i64, vp, vp, vp, i64, vp, vp, sp]
    lib.blmm_multi_device_result.argtypes = [vp, C.c_int, C.POINTER(vp), C.POINTER(i64), C.POINTER(i64), C.POINTER(i64),
                                             C.POINTER(vp)]
    lib.blmm_lod2log10p.argtypes = [vp, vp, i64, i64, i64, vp]
    lib.blmm_lod2log10p_dev.argtypes = [vp, vp, i64, i64, i64, i64, vp, i64]
    lib.blmm_lod_threshold.argtypes = [vp, vp, i64, i64, C.c_double, i64, vp, vp, vp, C.POINTER(i64)]
    lib.blmm_lod_threshold_dev.argtypes = [vp, vp, i64, i64, i64, C.c_double, i64, vp, vp, vp, vp]
    lib.blmm_get_thresholds.argtypes = [vp, vp, i64, i64, vp, i64, vp]
    lib.blmm_get_thresholds_dev.argtypes = [vp, vp, i64, i64, i64, vp, i64, vp]
    lib.blmm_last_log10p.argtypes = [vp, i64, vp]
    lib.blmm_set_log10p_output.argtypes = [vp, vp, i64, i64]
    lib.blmm_last_lod_threshold.argtypes = [vp, C.c_double, i64, vp, vp, vp, C.POINTER(i64)]
    lib.blmm_last_get_thresholds.argtypes = [vp, vp, i64, vp]
    lib.blmm_read_csv.argtypes = [C.c_char_p, i64, i64, i64, i64, C.POINTER(vp)]
    lib.blmm_read_he.argtypes = [C.c_char_p, C.POINTER(vp)]
    lib.blmm_table_rows.argtypes = [vp]
    lib.blmm_table_rows.restype = i64
    lib.blmm_table_cols.argtypes = [vp]
    lib.blmm_table_cols.restype = i64
    lib.blmm_table_copy.argtypes = [vp, vp]
    lib.blmm_table_free.argtypes = [vp]
    lib.blmm_table_free.restype = None
    lib.blmm_kinship_rounded.argtypes = [vp, vp, i64, i64, i64, vp]
    lib.blmm_host_register.argtypes = [vp, C.c_uint64]
    lib.blmm_host_unregister.argtypes = [vp]
    lib.blmm_host_alloc.argtypes = [C.c_uint64]
    lib.blmm_host_alloc.restype = vp
    lib.blmm_host_free.argtypes = [vp]
    lib.blmm_host_free.restype = None
    lib.blmm_lowrank_columns.argtypes = [vp, i64, vp, C.POINTER(i64), C.POINTER(i64)]
    lib.blmm_set_tuning.argtypes = [vp, C.c_char_p, C.c_double]
    lib.blmm_get_tuning.argtypes = [vp, C.c_char_p, dp]
    rp = C.POINTER(blmm_reduced)
    lib.blmm_bulkscan_reduced.argtypes = [vp, op, vp, i64, i64, vp, i64, vp, i64, vp, vp, vp, i64, rp, vp, sp]
    lib.blmm_bulkscan_reduced_dev.argtypes = [vp, op, vp, i64, i64, vp, i64, vp, i64, vp, vp, vp, i64, rp, vp, sp]
    lib.blmm_last_reduced_route.argtypes = [vp]
    lib.blmm_last_dims.argtypes = [vp, C.POINTER(i64), C.POINTER(i64)]
    lib.blmm_last_lod_colmax.argtypes = [vp, vp, vp]
    lib.blmm_last_lod_columns.argtypes = [vp, vp, i64, vp]
    lib.blmm_multi_last_colmax.argtypes = [vp, vp, vp]
    lib.blmm_multi_last_lod_threshold.argtypes = [vp, C.c_double, i64, vp, vp, vp, C.POINTER(i64)]
    for name in EXPORTS:
        getattr(lib, name)  # AttributeError if a declared symbol is not exported
    _lib = lib
    return lib
