"""bulklmm.jl_amd -- MI355X-native drop-in for the bulkscan hot path of senresearch/BulkLMM.jl.

Import name: `bulklmm_jl_amd` (via the shim at the repo root; the directory name contains a dot).
Everything is computed by libbulklmm_hip.so (HIP, gfx950); importing fails loudly if it is missing."""
from . import _lib
from ._lib import build, load, LIB_PATH, EXPORTS
from .sharding import trait_shard, shard_sizes, allgather_lod
from . import api
from .api import (BulkLMMError, Context, MultiContext, bulkscan_multi, host_register, host_unregister, bulkscan_into, readGenoProb, readGenoProb_ExcludeComplements, readBXDpheno, readBXDgeno, readhe, default_context, calcKinship, bulkscan, bulkscan_null, bulkscan_null_grid,
                  bulkscan_alt_grid, bulkscan_alt_exact, scan, transform_rotation, fitlmm_bulk, null_loglik_grid, weighted_liteqtl,
                  liteqtl_given_h2, bulkscan_dev, scan_perms_dev, lod2log10p, lod_colmax, get_thresholds, lod_threshold,
                  prepare_dev, rotated_rows, rotate_block_dev, bulkscan_prerotated_dev, scan_perms_prerotated_dev,
                  bulkscan_reduced, bulkscan_reduced_dev, DeviceLOD)

__all__ = ["BulkLMMError", "Context", "MultiContext", "bulkscan_multi", "default_context", "calcKinship", "bulkscan", "bulkscan_null", "bulkscan_null_grid",
           "bulkscan_alt_grid", "bulkscan_alt_exact", "scan", "transform_rotation", "fitlmm_bulk", "null_loglik_grid", "weighted_liteqtl",
           "liteqtl_given_h2", "bulkscan_dev", "scan_perms_dev", "lod2log10p", "build", "load", "LIB_PATH", "EXPORTS",
           "trait_shard", "shard_sizes", "allgather_lod", "lod_colmax", "get_thresholds", "lod_threshold",
           "prepare_dev", "rotated_rows", "rotate_block_dev", "bulkscan_prerotated_dev", "scan_perms_prerotated_dev",
           "bulkscan_reduced", "bulkscan_reduced_dev", "DeviceLOD"]
