"""ctypes binding of include/grandplus.h (libgrandplus.so, hand-written HIP for gfx950).

There is no CPU fallback: if the shared library is missing this module raises at first
use, and if no GPU is visible `gp_graph_create` returns GP_ERR_NO_DEVICE (RuntimeError).
"""
from __future__ import annotations

import ctypes
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
# GRANDPLUS_DIAG=1 selects the diagnostic build (in-kernel phase stamps); never the default.
# GRANDPLUS_LIB=<file name in this directory> selects another HIP build of the same sources (A/B runs of
# two kernel variants on the same GPU box, see tools/ab.sh); there is no non-HIP implementation to select.
LIB_PATH = os.path.join(_HERE, os.environ.get("GRANDPLUS_LIB") or
                        ("libgrandplus_diag.so" if os.environ.get("GRANDPLUS_DIAG") == "1" else "libgrandplus.so"))

GP_OK = 0
GP_ERR_NULL, GP_ERR_INVALID_CSR, GP_ERR_INVALID_SEED, GP_ERR_INVALID_ARG = 1, 2, 3, 4
GP_ERR_NO_DEVICE, GP_ERR_HIP, GP_ERR_NOMEM, GP_ERR_OVERFLOW = 5, 6, 7, 8
GP_MAX_K = 1024

# every symbol include/grandplus.h declares (tests check the library exports all of them)
EXPORTS = (
    "gp_abi_version", "gp_strerror", "gp_last_error", "gp_device_count",
    "gp_graph_create", "gp_graph_destroy", "gp_graph_num_nodes", "gp_graph_nnz",
    "gp_graph_device", "gp_gfpush", "gp_gfpush_device", "gp_get_stats", "gp_reset_stats",
    "gp_set_option", "gp_random_prop_rows", "gp_random_prop_coo", "gp_internal_set_error",
    "gp_propagate_features", "gp_internal_graph_csr", "gp_internal_diag_counters",
    "gp_graph_create_multi", "gp_graph_num_gpus", "gp_internal_multi_plan", "gp_internal_graph_acsr", "gp_graph_create_multi_on",
    "gp_seed_positions", "gp_batch_positions", "gp_internal_create_ms", "gp_internal_warm_device",
)


class GpStats(ctypes.Structure):
    _fields_ = [
        ("rows", ctypes.c_int64), ("pushes", ctypes.c_int64), ("edges", ctypes.c_int64),
        ("filled", ctypes.c_int64), ("support", ctypes.c_int64), ("frontier", ctypes.c_int64),
        ("lds_levels", ctypes.c_int64), ("global_levels", ctypes.c_int64),
        ("failed_rows", ctypes.c_int64), ("degree_lookups", ctypes.c_int64), ("kernel_ms", ctypes.c_double),
        ("workgroups", ctypes.c_int32), ("block_threads", ctypes.c_int32),
        ("lds_bytes", ctypes.c_int32), ("lds_slots", ctypes.c_int32),
        ("workspace_bytes", ctypes.c_int64),
        ("diag_ticks_scan", ctypes.c_int64), ("diag_ticks_expand", ctypes.c_int64),
        ("diag_ticks_topk", ctypes.c_int64), ("diag_ticks_total", ctypes.c_int64),
        ("diag_ticks_scan_hbm", ctypes.c_int64), ("diag_ticks_expand_hbm", ctypes.c_int64),
        ("diag_sub", ctypes.c_int64 * 16),
        ("retried_rows", ctypes.c_int64), ("max_level_edges", ctypes.c_int64), ("max_log_records", ctypes.c_int64),
        ("kernel", ctypes.c_int32), ("sketch_pad", ctypes.c_int32),
        ("sketch_candidate_edges", ctypes.c_int64), ("sketch_second_sweeps", ctypes.c_int64),
        ("choice_ms", ctypes.c_float * 3), ("choice_pad", ctypes.c_int32),
    ]

    def as_dict(self):
        d = {name: getattr(self, name) for name, _ in self._fields_}
        d["diag_sub"] = list(self.diag_sub)
        d["choice_ms"] = list(self.choice_ms)
        return d


_LIB = None


def _optional(L, name, argtypes):
    """Declare an entry point an older build (GRANDPLUS_LIB=... for an A/B run) may lack: calling it there raises a clear error."""
    try:
        f = getattr(L, name)
    except AttributeError:
        def missing(*_a, **_k):
            raise RuntimeError(f"{LIB_PATH} does not export {name} (an older build?): rebuild the library")
        setattr(L, name, missing)
        return
    f.restype = ctypes.c_int
    f.argtypes = argtypes


def lib():
    """Load libgrandplus.so (once).  Raises RuntimeError if it has not been built."""
    global _LIB
    if _LIB is not None:
        return _LIB
    if not os.path.exists(LIB_PATH):
        raise RuntimeError(
            f"{LIB_PATH} is missing: the HIP extension has not been built "
            "(run `python -c 'import __graft_entry__ as g; g.build()'`). There is no CPU fallback.")
    # PyTorch-ROCm bundles its own libamdhip64 / libhsa-runtime64.  If this library were loaded first it
    # would pull in /opt/rocm's copies and the process would hold TWO HIP runtimes; whichever initialises
    # second can then report "no ROCm-capable device".  Loading torch first makes the dynamic loader bind
    # libgrandplus.so to the runtime torch already mapped, so device pointers and streams are shared.
    try:
        import torch  # noqa: F401
    except ImportError:
        pass
    L = ctypes.CDLL(LIB_PATH)
    i32p, f64p = ctypes.POINTER(ctypes.c_int32), ctypes.POINTER(ctypes.c_double)
    vp = ctypes.c_void_p
    L.gp_abi_version.restype = ctypes.c_int
    L.gp_strerror.restype = ctypes.c_char_p
    L.gp_strerror.argtypes = [ctypes.c_int]
    L.gp_last_error.restype = ctypes.c_char_p
    L.gp_device_count.restype = ctypes.c_int
    L.gp_graph_create.restype = ctypes.c_int
    L.gp_graph_create.argtypes = [i32p, ctypes.c_int64, i32p, ctypes.c_int64, ctypes.c_int,
                                  ctypes.POINTER(vp)]
    L.gp_graph_create_multi.restype = ctypes.c_int
    L.gp_graph_create_multi.argtypes = [i32p, ctypes.c_int64, i32p, ctypes.c_int64, ctypes.c_int, ctypes.POINTER(vp)]
    _optional(L, "gp_graph_create_multi_on", [i32p, ctypes.c_int64, i32p, ctypes.c_int64, ctypes.POINTER(ctypes.c_int), ctypes.c_int, ctypes.POINTER(vp)])
    L.gp_graph_num_gpus.restype = ctypes.c_int
    L.gp_graph_num_gpus.argtypes = [vp]
    L.gp_graph_destroy.restype = None
    L.gp_graph_destroy.argtypes = [vp]
    L.gp_graph_num_nodes.restype = ctypes.c_int64
    L.gp_graph_num_nodes.argtypes = [vp]
    L.gp_graph_nnz.restype = ctypes.c_int64
    L.gp_graph_nnz.argtypes = [vp]
    L.gp_graph_device.restype = ctypes.c_int
    L.gp_graph_device.argtypes = [vp]
    L.gp_gfpush.restype = ctypes.c_int
    L.gp_gfpush.argtypes = [vp, i32p, ctypes.c_int64, f64p, ctypes.c_int, ctypes.c_double,
                            ctypes.c_int, i32p, i32p, f64p]
    # device pointers travel as integers (tensor.data_ptr())
    L.gp_gfpush_device.restype = ctypes.c_int
    L.gp_gfpush_device.argtypes = [vp, vp, ctypes.c_int64, f64p, ctypes.c_int, ctypes.c_double,
                                   ctypes.c_int, vp, vp, vp, vp, vp]
    L.gp_get_stats.restype = ctypes.c_int
    L.gp_get_stats.argtypes = [vp, ctypes.POINTER(GpStats)]
    L.gp_reset_stats.restype = ctypes.c_int
    L.gp_reset_stats.argtypes = [vp]
    L.gp_random_prop_rows.restype = ctypes.c_int
    L.gp_random_prop_rows.argtypes = [ctypes.c_int, vp, ctypes.c_int64, ctypes.c_int32, vp, vp, vp, ctypes.c_int32,
                                      vp, ctypes.c_int32, ctypes.c_float, ctypes.c_int, ctypes.c_uint64, vp, vp, vp]
    L.gp_random_prop_coo.restype = ctypes.c_int
    L.gp_random_prop_coo.argtypes = [ctypes.c_int, vp, ctypes.c_int64, ctypes.c_int32, vp, vp, ctypes.c_int64,
                                     ctypes.c_float, ctypes.c_int, ctypes.c_uint64, vp, vp, vp]
    _optional(L, "gp_seed_positions", [ctypes.c_int, vp, ctypes.c_int64, ctypes.c_int64, vp, vp, vp])
    _optional(L, "gp_batch_positions", [ctypes.c_int, vp, ctypes.c_int64, vp, ctypes.c_int64, vp, vp, vp])
    L.gp_propagate_features.restype = ctypes.c_int
    L.gp_propagate_features.argtypes = [vp, vp, ctypes.c_int32, vp, ctypes.c_int, ctypes.c_int, ctypes.c_double, vp, vp]
    L.gp_internal_diag_counters.restype = ctypes.c_int
    L.gp_internal_diag_counters.argtypes = [vp, ctypes.POINTER(ctypes.c_int64), ctypes.c_int]
    _optional(L, "gp_internal_graph_acsr", [vp, vp, vp, vp, ctypes.POINTER(ctypes.c_int64), ctypes.POINTER(ctypes.c_int), ctypes.POINTER(ctypes.c_uint32)])
    _optional(L, "gp_internal_warm_device", [ctypes.c_int])
    try:
        L.gp_internal_create_ms.restype = None
        L.gp_internal_create_ms.argtypes = [ctypes.POINTER(ctypes.c_double)]
    except AttributeError:
        pass
    L.gp_set_option.restype = ctypes.c_int
    L.gp_set_option.argtypes = [vp, ctypes.c_char_p, ctypes.c_int64]
    # (ABI 3 is accepted only for an older build named explicitly through GRANDPLUS_LIB for an A/B run: its gp_stats is a prefix of
    #  this one, and the entry points it lacks raise a clear error where they are called -- _optional)
    abi = L.gp_abi_version()
    if abi != 4 and not (abi == 3 and os.environ.get("GRANDPLUS_LIB")):
        raise RuntimeError(f"libgrandplus.so ABI version {abi}, this package needs 4: rebuild (python -c 'import __graft_entry__ as g; g.build()')")
    _LIB = L
    return L


def raise_for_status(status: int):
    """Map a C-ABI status to the Python exception the shims document."""
    if status == GP_OK:
        return
    L = lib()
    detail = L.gp_last_error().decode() or L.gp_strerror(status).decode()
    if status in (GP_ERR_INVALID_CSR, GP_ERR_INVALID_SEED, GP_ERR_INVALID_ARG, GP_ERR_NULL):
        raise ValueError(detail)
    if status == GP_ERR_NOMEM:
        raise MemoryError(detail)
    raise RuntimeError(detail)
