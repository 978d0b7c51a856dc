"""Deterministic synthetic power-law CSR graphs of the shapes BASELINE.json names.

Thin ctypes front-end over csrc/synth_graph.cpp (host C++/OpenMP, integer-only, so the
same parameters regenerate the same CSR bit for bit here and on the GPU box).
"""
from __future__ import annotations

import ctypes
import os
from dataclasses import dataclass

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None
GEN_VERSION = 1          # bump when csrc/synth_graph.cpp changes what it generates


def _lib():
    global _LIB
    if _LIB is None:
        path = os.path.join(_HERE, "libgpsynth.so")
        if not os.path.exists(path):
            raise RuntimeError(f"{path} is missing: run `python -c 'import __graft_entry__ as g; g.build()'`")
        lib = ctypes.CDLL(path)
        lib.gp_synth_powerlaw_csr.restype = ctypes.c_int
        lib.gp_synth_powerlaw_csr.argtypes = [
            ctypes.c_int64, ctypes.c_int64, ctypes.c_uint64, ctypes.c_int64,
            ctypes.POINTER(ctypes.POINTER(ctypes.c_int32)),
            ctypes.POINTER(ctypes.POINTER(ctypes.c_int32)),
            ctypes.POINTER(ctypes.c_int64)]
        lib.gp_synth_free.argtypes = [ctypes.c_void_p]
        lib.gp_synth_seeds.restype = ctypes.c_int
        lib.gp_synth_seeds.argtypes = [ctypes.c_int64, ctypes.c_int64, ctypes.c_uint64,
                                       ctypes.POINTER(ctypes.c_int32)]
        lib.gp_synth_set_threads.restype = None
        lib.gp_synth_set_threads.argtypes = [ctypes.c_int]
        lib.gp_checksum64.restype = ctypes.c_uint64
        lib.gp_checksum64.argtypes = [ctypes.c_void_p, ctypes.c_int64]
        _LIB = lib
    return _LIB


@dataclass(frozen=True)
class Shape:
    """A named synthetic shape: n nodes, `samples` undirected edge draws, RNG seed, hub offset."""
    name: str
    n_nodes: int
    samples: int
    seed: int = 42
    offset: int = 10


# Sample counts are calibrated so that the directed nnz BEFORE self-loops lands within
# ~1 % of the edge counts BASELINE.json names (duplicates between hubs are merged).
SHAPES = {
    "tiny":      Shape("tiny", 2_000, 8_000),
    "small":     Shape("small", 100_000, 700_000),
    "reddit":    Shape("reddit", 232_965, 5_940_000),
    "amazon2m":  Shape("amazon2m", 2_449_029, 30_750_000),
    "mag":       Shape("mag", 12_400_000, 86_850_000),
}


def set_threads(n: int) -> None:
    """Host threads the generator may use (several ranks of one node generate at once)."""
    _lib().gp_synth_set_threads(int(n))


def powerlaw_csr(n_nodes: int, samples: int, seed: int = 42, offset: int = 10):
    """Return (indptr int32[n+1], indices int32[nnz]) of the symmetric power-law graph + I."""
    lib = _lib()
    p_ptr = ctypes.POINTER(ctypes.c_int32)()
    p_idx = ctypes.POINTER(ctypes.c_int32)()
    nnz = ctypes.c_int64(0)
    rc = lib.gp_synth_powerlaw_csr(n_nodes, samples, seed, offset,
                                   ctypes.byref(p_ptr), ctypes.byref(p_idx), ctypes.byref(nnz))
    if rc != 0:
        raise ValueError(f"gp_synth_powerlaw_csr failed with status {rc}")
    try:
        indptr = np.ctypeslib.as_array(p_ptr, shape=(n_nodes + 1,)).copy()
        indices = np.ctypeslib.as_array(p_idx, shape=(max(nnz.value, 1),))[:nnz.value].copy()
    finally:
        lib.gp_synth_free(p_ptr)
        lib.gp_synth_free(p_idx)
    return indptr, indices


def shape_csr(name: str):
    """The named shape.  GRANDPLUS_SYNTH_CACHE=<dir> keeps generated graphs as .npy files there (A/B tooling that
    starts many processes on one box; e.g. /dev/shm/gp); without it every call regenerates."""
    s = SHAPES[name]
    cache = os.environ.get("GRANDPLUS_SYNTH_CACHE")
    if cache:
        # every generator parameter is part of the file name (a changed SHAPES entry must not load a stale graph);
        # GEN_VERSION changes with csrc/synth_graph.cpp
        tag = f"{name}_n{s.n_nodes}_m{s.samples}_s{s.seed}_o{s.offset}_v{GEN_VERSION}"
        fp, fi = (os.path.join(cache, f"{tag}_{k}.npy") for k in ("indptr", "indices"))
        if os.path.exists(fp) and os.path.exists(fi):
            ip, ix = np.load(fp), np.load(fi)
            if len(ip) == s.n_nodes + 1 and int(ip[-1]) == len(ix):
                return ip, ix
        indptr, indices = powerlaw_csr(s.n_nodes, s.samples, s.seed, s.offset)
        os.makedirs(cache, exist_ok=True)
        for path, arr in ((fp, indptr), (fi, indices)):
            tmp = f"{path}.{os.getpid()}.tmp.npy"
            np.save(tmp, arr)
            os.replace(tmp, path)
        return indptr, indices
    return powerlaw_csr(s.n_nodes, s.samples, s.seed, s.offset)


def seeds(n_nodes: int, n_seeds: int, seed: int = 42) -> np.ndarray:
    out = np.empty(n_seeds, dtype=np.int32)
    rc = _lib().gp_synth_seeds(n_nodes, n_seeds, seed, out.ctypes.data_as(ctypes.POINTER(ctypes.c_int32)))
    if rc != 0:
        raise ValueError(f"gp_synth_seeds failed with status {rc}")
    return out


def checksum64(a: np.ndarray) -> int:
    a = np.ascontiguousarray(a)
    return int(_lib().gp_checksum64(a.ctypes.data, a.nbytes))
