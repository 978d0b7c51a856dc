"""Host-side mirror of the reference's operator interface for the GFPush hot path.

`Graph` has the reference's surface (`precompute/propagation.cpp:9-11`):
    Graph(indptr, indices, seed)                                   -- graph.h:32-47
    Graph.gfpush_omp(node_idx, row_idx, col_idx, value, coef, rmax, K) -> None   -- graph.h:53-131
with the same positional arguments and in-place output convention, on top of the C ABI
(include/grandplus.h).  `gfpush_device` is the device-resident form used by bench.py and
by the multi-GPU driver (grand_plus_amd/sharded.py).

Differences from the reference, all supersets:
  * the CSR is copied to the GPU (the reference borrows the numpy buffers, graph.h:35-36);
  * wrong-dtype / non-contiguous OUTPUT arrays raise TypeError (the reference silently
    writes into a temporary and the results are lost);
  * invalid CSR, out-of-range seeds, K < 1 raise ValueError (the reference reads out of bounds).
"""
from __future__ import annotations

import ctypes

import numpy as np

from . import _native


def _as_i32_readonly(a, name):
    a = np.asarray(a)
    if a.dtype.kind not in "iu":
        raise TypeError(f"{name} must be an integer array, got {a.dtype}")
    # (only a wider or unsigned-32 dtype can hold a value outside int32: two passes over 185 M column ids otherwise -- ~50 ms of a
    #  0.16 s constructor on the MAG shape)
    wider = a.dtype.itemsize > 4 or (a.dtype.kind == "u" and a.dtype.itemsize == 4)
    if wider and a.size and (a.min() < -2**31 or a.max() >= 2**31):
        raise ValueError(f"{name} does not fit int32")
    return np.ascontiguousarray(a, dtype=np.int32).reshape(-1)     # pybind11 force-cast semantics


def _check_out(a, dtype, name, need):
    if not isinstance(a, np.ndarray) or a.dtype != dtype or not a.flags.c_contiguous or not a.flags.writeable:
        raise TypeError(f"{name} must be a writeable C-contiguous numpy array of dtype {np.dtype(dtype).name} "
                        "(the reference silently drops results otherwise)")
    if a.size < need:
        raise ValueError(f"{name} has {a.size} slots, need len(node_idx)*K = {need}")


def _ptr(a, ct):
    return a.ctypes.data_as(ctypes.POINTER(ct))


class Graph:
    """CSR graph resident in one MI355X's HBM.  Mirrors `propagation.Graph`."""

    def __init__(self, indptr, indices, seed=0, device=None, n_gpus=None, devices=None):
        """`n_gpus` = None: one GPU (`device`, the form every rank of the torch.distributed driver uses).
        `n_gpus` = 0 / N: a multi-GPU handle over all / the first N visible GPUs (`gp_graph_create_multi`):
        `gfpush_omp` then shards its seeds over them inside the one call (RCCL all-gather of the rows).
        `devices` = [0, 0, ...]: such a handle over an explicit device list (`gp_graph_create_multi_on`); a repeated device
        gives several parts on one GPU, gathered through the host -- the whole sharded path on a one-GPU box."""
        L = _native.lib()
        ip = _as_i32_readonly(indptr, "indptr")
        ix = _as_i32_readonly(indices, "indices")
        if ip.size < 1:
            raise ValueError("indptr must have at least one element")
        self.seed = int(seed)                  # accepted and unused, as in the reference (graph.h:40)
        if device is None:
            import os
            device = int(os.environ.get("GRANDPLUS_DEVICE", os.environ.get("LOCAL_RANK", "0")))
            if device >= max(L.gp_device_count(), 1):
                device = 0
        h = ctypes.c_void_p()
        if devices is not None:
            devs = (ctypes.c_int * len(devices))(*[int(d) for d in devices])
            _native.raise_for_status(L.gp_graph_create_multi_on(_ptr(ip, ctypes.c_int32), ip.size - 1, _ptr(ix, ctypes.c_int32), ix.size,
                                                                devs, len(devices), ctypes.byref(h)))
            device = L.gp_graph_device(h)
        elif n_gpus is None:
            _native.raise_for_status(L.gp_graph_create(_ptr(ip, ctypes.c_int32), ip.size - 1,
                                                       _ptr(ix, ctypes.c_int32), ix.size, int(device),
                                                       ctypes.byref(h)))
        else:
            _native.raise_for_status(L.gp_graph_create_multi(_ptr(ip, ctypes.c_int32), ip.size - 1,
                                                             _ptr(ix, ctypes.c_int32), ix.size, int(n_gpus),
                                                             ctypes.byref(h)))
            device = L.gp_graph_device(h)
        self._h = h
        self.num_nodes = ip.size - 1
        self.nnz = ix.size
        self.device = int(device)
        self.n_gpus = L.gp_graph_num_gpus(h)

    def close(self):
        if getattr(self, "_h", None):
            _native.lib().gp_graph_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # -- the reference's method ------------------------------------------------------
    def gfpush_omp(self, node_idx, row_idx, col_idx, value, coef, rmax, K):
        """In-place GFPush, exactly the reference's call (`model.py:268`).  Returns None."""
        L = _native.lib()
        seeds = _as_i32_readonly(node_idx, "node_idx")
        cf = np.ascontiguousarray(np.asarray(coef, dtype=np.float64)).reshape(-1)
        K = int(K)
        if K < 1:
            raise ValueError("K must be >= 1")
        need = seeds.size * K
        _check_out(row_idx, np.int32, "row_idx", need)
        _check_out(col_idx, np.int32, "col_idx", need)
        _check_out(value, np.float64, "value", need)
        _native.raise_for_status(L.gp_gfpush(self._h, _ptr(seeds, ctypes.c_int32), seeds.size,
                                             _ptr(cf, ctypes.c_double), cf.size, float(rmax), K,
                                             _ptr(row_idx, ctypes.c_int32), _ptr(col_idx, ctypes.c_int32),
                                             _ptr(value, ctypes.c_double)))
        return None

    # -- device-resident form ----------------------------------------------------------
    def gfpush_device(self, seeds, coef, rmax, K, row=None, col=None, val=None, filled=None, stream=None):
        """GFPush on torch CUDA tensors living on this graph's GPU; asynchronous on `stream`
        (default: torch's current stream).  Returns (row, col, val, filled); rows are written
        densely for the first filled[it] slots, the remaining slots keep their contents."""
        import torch

        L = _native.lib()
        dev = torch.device("cuda", self.device)
        if seeds.device != dev or seeds.dtype != torch.int32 or not seeds.is_contiguous():
            raise TypeError(f"seeds must be a contiguous int32 tensor on {dev}")
        S = seeds.numel()
        K = int(K)
        if row is None:
            row = torch.zeros(S * K, dtype=torch.int32, device=dev)
            col = torch.zeros(S * K, dtype=torch.int32, device=dev)
            val = torch.zeros(S * K, dtype=torch.float64, device=dev)
        if filled is None:
            filled = torch.zeros(S, dtype=torch.int32, device=dev)
        for t, dt, n, nm in ((row, torch.int32, S * K, "row"), (col, torch.int32, S * K, "col"),
                             (val, torch.float64, S * K, "val"), (filled, torch.int32, S, "filled")):
            if t.device != dev or t.dtype != dt or not t.is_contiguous() or t.numel() < n:
                raise TypeError(f"{nm} must be a contiguous {dt} tensor on {dev} with >= {n} elements")
        cf = np.ascontiguousarray(np.asarray(coef, dtype=np.float64)).reshape(-1)
        if stream is None:
            stream = torch.cuda.current_stream(dev).cuda_stream
        _native.raise_for_status(L.gp_gfpush_device(
            self._h, seeds.data_ptr(), S, _ptr(cf, ctypes.c_double), cf.size, float(rmax), K,
            row.data_ptr(), col.data_ptr(), val.data_ptr(), filled.data_ptr(), ctypes.c_void_p(stream)))
        return row, col, val, filled

    # -- exact inference propagation ----------------------------------
    def propagate_features(self, features, prop_mode, order, alpha=0.2, edge_weight=None, out=None, stream=None):
        """Exact full-graph propagation of `predict()` (`model.py:186-210`) on this graph's CSR.

        features: float32 CUDA tensor [N, F]; prop_mode 'ppr' | 'avg' | 'single' (`args.prop_mode`);
        order = `args.order`; edge_weight: float32 CUDA tensor [nnz] with the stored values of adj + I, or
        None when they are all ones (every shipped dataset).  Returns the float32 [N, F] matrix the reference
        feeds to the MLP (`model.py:212-213`)."""
        import torch
        dev = torch.device("cuda", self.device)
        if features.device != dev or features.dtype != torch.float32 or not features.is_contiguous() or features.dim() != 2:
            raise TypeError(f"features must be a contiguous float32 [N, F] tensor on {dev}")
        if features.shape[0] != self.num_nodes:
            raise ValueError("features must have one row per node")
        modes = {"ppr": 0, "avg": 1, "single": 2}
        if prop_mode not in modes:
            raise ValueError(f"Unknown propagation mode: {prop_mode}")            # model.py:210
        if edge_weight is not None and (edge_weight.device != dev or edge_weight.dtype != torch.float32
                                        or not edge_weight.is_contiguous() or edge_weight.numel() != self.nnz):
            raise TypeError("edge_weight must be a contiguous float32 [nnz] tensor on the graph's device")
        if out is None:
            out = torch.empty_like(features)
        if stream is None:
            stream = torch.cuda.current_stream(dev).cuda_stream
        _native.raise_for_status(_native.lib().gp_propagate_features(
            self._h, features.data_ptr(), features.shape[1], edge_weight.data_ptr() if edge_weight is not None else None,
            modes[prop_mode], int(order), float(alpha), out.data_ptr(), ctypes.c_void_p(stream)))
        return out

    def reset_stats(self):
        _native.raise_for_status(_native.lib().gp_reset_stats(self._h))

    def stats(self):
        """Counters since the last reset_stats() (gfpush_omp resets them itself); waits for the
        outstanding call.  Raises if a row hit a workspace bound."""
        st = _native.GpStats()
        rc = _native.lib().gp_get_stats(self._h, ctypes.byref(st))
        _native.raise_for_status(rc)
        return st.as_dict()

    def self_addressed_csr(self):
        """The sketch kernel's copy of the CSR, as host arrays (tests): (acsr int32[32 * n_units + 1] incl. the sentinel word,
        node_pos uint32[N + 1], unit_info int32[n_units], unit_bits, deg_sat); None when the graph does not allow the layout."""
        import numpy as np
        nu, bits, sat = ctypes.c_int64(), ctypes.c_int(), ctypes.c_uint32()
        L = _native.lib()
        _native.raise_for_status(L.gp_internal_graph_acsr(self._h, None, None, None, ctypes.byref(nu), ctypes.byref(bits), ctypes.byref(sat)))
        if nu.value == 0:
            return None
        acsr = np.empty(32 * nu.value + 1, np.int32); pos = np.empty(self.num_nodes + 1, np.uint32); info = np.empty(nu.value, np.int32)
        _native.raise_for_status(L.gp_internal_graph_acsr(self._h, acsr.ctypes.data_as(ctypes.c_void_p), pos.ctypes.data_as(ctypes.c_void_p),
                                                          info.ctypes.data_as(ctypes.c_void_p), ctypes.byref(nu), ctypes.byref(bits), ctypes.byref(sat)))
        return acsr, pos, info, bits.value, sat.value

    def diag_counters(self):
        """Extended counters of the diagnostic build (GRANDPLUS_DIAG=1); zeros in the product library."""
        buf = (ctypes.c_int64 * 256)()
        _native.raise_for_status(_native.lib().gp_internal_diag_counters(self._h, buf, 256))
        return list(buf)

    def set_option(self, key: str, value: int):
        _native.raise_for_status(_native.lib().gp_set_option(self._h, key.encode(), int(value)))


def algorithmic_bytes(stats: dict) -> int:
    """bytes_algo = 8*P + 4*E + 16*filled + 4*rows (compulsory CSR reads of the
    pushed nodes, the output slots, and the seed ids)."""
    return 8 * stats["pushes"] + 4 * stats["edges"] + 16 * stats["filled"] + 4 * stats["rows"]
