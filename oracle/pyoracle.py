"""ctypes front-end for the CHECKERS in oracle/ (test infrastructure only).

Importable from tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg.
Nothing under grand_plus_amd/ may import this module.
"""
from __future__ import annotations

import ctypes
import importlib.util
import os
import subprocess
import sysconfig

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None


def build(quiet: bool = True) -> None:
    """(Re)build libgfpush_oracle.so and, when /root/reference exists, oracle/_ref."""
    subprocess.run(["make", "-C", _HERE], check=True,
                   stdout=subprocess.DEVNULL if quiet else None)


def _lib():
    global _LIB
    if _LIB is None:
        path = os.path.join(_HERE, "libgfpush_oracle.so")
        if not os.path.exists(path):
            build()
        lib = ctypes.CDLL(path)
        i32p = ctypes.POINTER(ctypes.c_int32)
        f64p = ctypes.POINTER(ctypes.c_double)
        lib.gfpush_oracle.restype = ctypes.c_int
        lib.gfpush_oracle.argtypes = [i32p, ctypes.c_int64, i32p, i32p, ctypes.c_int64,
                                      f64p, ctypes.c_int, ctypes.c_double, ctypes.c_int,
                                      i32p, i32p, f64p, ctypes.c_int,
                                      ctypes.POINTER(ctypes.c_int64)]
        lib.gfpush_oracle_ex.restype = ctypes.c_int
        lib.gfpush_oracle_ex.argtypes = lib.gfpush_oracle.argtypes + [f64p]
        lib.gfpush_oracle_max_threads.restype = ctypes.c_int
        _LIB = lib
    return _LIB


def _p(a, ct):
    return a.ctypes.data_as(ctypes.POINTER(ct))


STAT_NAMES = ("pushes", "edges", "filled", "support_sum", "support_max",
              "frontier_max", "dangling", "frontier_sum")


def gfpush(indptr, indices, seeds, coef, rmax, K, row_idx=None, col_idx=None, value=None,
           threads: int = 0, want_next: bool = False):
    """Run the CPU restatement.  Returns (row_idx, col_idx, value, stats dict); with want_next the stats dict also
    carries "next_value": the (K+1)-th largest reserve value of every row (what proves a K-th-position tie)."""
    lib = _lib()
    indptr = np.ascontiguousarray(indptr, dtype=np.int32)
    indices = np.ascontiguousarray(indices, dtype=np.int32)
    seeds = np.ascontiguousarray(seeds, dtype=np.int32)
    coef = np.ascontiguousarray(coef, dtype=np.float64)
    S = len(seeds)
    if row_idx is None:
        row_idx = np.zeros(S * K, dtype=np.int32)
        col_idx = np.zeros(S * K, dtype=np.int32)
        value = np.zeros(S * K, dtype=np.float64)
    stats = np.zeros(8, dtype=np.int64)
    if threads <= 0:
        threads = lib.gfpush_oracle_max_threads()
    nxt = np.zeros(S if want_next else 0, dtype=np.float64)
    rc = lib.gfpush_oracle_ex(_p(indptr, ctypes.c_int32), len(indptr) - 1, _p(indices, ctypes.c_int32),
                              _p(seeds, ctypes.c_int32), S, _p(coef, ctypes.c_double), len(coef),
                              float(rmax), int(K), _p(row_idx, ctypes.c_int32),
                              _p(col_idx, ctypes.c_int32), _p(value, ctypes.c_double),
                              int(threads), _p(stats, ctypes.c_int64), _p(nxt, ctypes.c_double) if want_next else None)
    if rc != 0:
        raise ValueError(f"gfpush_oracle failed with status {rc}")
    st = dict(zip(STAT_NAMES, stats.tolist()))
    if want_next:
        st["next_value"] = nxt
    return row_idx, col_idx, value, st


def max_threads() -> int:
    return _lib().gfpush_oracle_max_threads()


def load_reference_module():
    """The REAL reference pybind11 module compiled into oracle/_ref (None if absent)."""
    # exactly the binary built for THIS interpreter (oracle/Makefile names it with the same suffix); a directory listing
    # would pick whatever ABI tag happens to come first
    path = os.path.join(_HERE, "_ref", "propagation" + sysconfig.get_config_var("EXT_SUFFIX"))
    if not os.path.exists(path):
        return None
    spec = importlib.util.spec_from_file_location("propagation", path)
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod
