"""Feature augmentation ("random propagation") of GRAND+ on MI355X -- SURVEY.md 8(f) next-1.

Host-side mirror of `Grand_Plus.random_prop(feats, mat_scores, mat_idx, dropnode_rate)`
(`model.py:80-87`, `model_mag.py:80-86`) over the fused HIP kernels of csrc/augment.hip:

  * `random_prop`       -- the reference's own argument shape (gathered feats, scores, sorted ids);
  * `random_prop_rows`  -- the MI355X-native form: reads the `[S x K]` rows `Graph.gfpush_device`
                           left in HBM and the node-feature matrix, so the per-step scipy slicing,
                           host-side feature gather and upload of `model.py:310-316` disappear.

The reference's dropout draws from torch's global generator (`F.dropout`, `model.py:82`); here the
keep decision of entry e is a counter-based RNG of (seed, e), or an explicit `keep` mask.  Both keep
an entry with probability 1 - dropnode_rate and scale kept scores by 1/(1 - dropnode_rate).
"""
from __future__ import annotations

import ctypes
import itertools

from . import _native

_seed_counter = itertools.count(0x5EED)


def _dev_index(t):
    if not t.is_cuda:
        raise TypeError("random_prop runs on the GPU only: tensors must be CUDA tensors (no CPU fallback)")
    return t.device.index


def _check(t, dtype, name):
    import torch
    if not isinstance(t, torch.Tensor) or t.dtype != dtype or not t.is_contiguous() or not t.is_cuda:
        raise TypeError(f"{name} must be a contiguous CUDA tensor of dtype {dtype}")


def random_prop(feats, mat_scores, mat_idx, dropnode_rate, training=True, seed=None, keep=None, stream=None):
    """Drop-in for `Grand_Plus.random_prop` (`model.py:80-87`) on CUDA tensors.

    feats [M, F] float32, mat_scores [M] float32, mat_idx [M] int64 sorted ascending (the order
    scipy's `.nonzero()` yields, `model.py:312`).  Returns [mat_idx[-1] + 1, F] float32.
    `training` plays the role of `self.training`.
    """
    import torch
    _check(feats, torch.float32, "feats")
    _check(mat_scores, torch.float32, "mat_scores")
    _check(mat_idx, torch.int64, "mat_idx")
    M, F = feats.shape
    if mat_scores.numel() != M or mat_idx.numel() != M:
        raise ValueError("feats, mat_scores and mat_idx must have the same number of entries")
    if M == 0:
        return feats.new_zeros((0, F))
    n_out = int(mat_idx[-1].item()) + 1                                   # model.py:84 dim_size
    out = torch.empty((n_out, F), dtype=torch.float32, device=feats.device)
    if keep is not None:
        _check(keep, torch.uint8, "keep")
    if seed is None:
        seed = next(_seed_counter) * 0x9E3779B97F4A7C15 & (2**64 - 1)
    if stream is None:
        stream = torch.cuda.current_stream(feats.device).cuda_stream
    rc = _native.lib().gp_random_prop_coo(
        _dev_index(feats), feats.data_ptr(), M, F, mat_scores.data_ptr(), mat_idx.data_ptr(), n_out,
        float(dropnode_rate), int(bool(training)), ctypes.c_uint64(seed), keep.data_ptr() if keep is not None else None,
        out.data_ptr(), ctypes.c_void_p(stream))
    _native.raise_for_status(rc)
    return out


def random_prop_rows(features, col, val, filled, K, batch_rows=None, dropnode_rate=0.5, training=True,
                     seed=None, keep=None, stream=None):
    """Fused augmentation straight from the GFPush row matrix.

    features [N, F] float32 (node features resident on the GPU); col int32 [S*K], val float64 [S*K],
    filled int32 [S] as returned by `Graph.gfpush_device`; batch_rows int32 [B] = positions of the
    batch's seeds in the seed list (None = all S rows).  Returns [B, F] float32:
        out[b] = sum_k w_k X[col[r,k]] / (sum_k w_k + 1e-12),  r = batch_rows[b]
    """
    import torch
    _check(features, torch.float32, "features")
    _check(col, torch.int32, "col")
    _check(val, torch.float64, "val")
    N, F = features.shape
    S = col.numel() // K
    if filled is not None:
        _check(filled, torch.int32, "filled")
    if batch_rows is not None:
        _check(batch_rows, torch.int32, "batch_rows")
    B = S if batch_rows is None else batch_rows.numel()
    out = torch.empty((B, F), dtype=torch.float32, device=features.device)
    if keep is not None:
        _check(keep, torch.uint8, "keep")
    if seed is None:
        seed = next(_seed_counter) * 0x9E3779B97F4A7C15 & (2**64 - 1)
    if stream is None:
        stream = torch.cuda.current_stream(features.device).cuda_stream
    rc = _native.lib().gp_random_prop_rows(
        _dev_index(features), features.data_ptr(), N, F, col.data_ptr(), val.data_ptr(),
        filled.data_ptr() if filled is not None else None, int(K),
        batch_rows.data_ptr() if batch_rows is not None else None, B,
        float(dropnode_rate), int(bool(training)), ctypes.c_uint64(seed), keep.data_ptr() if keep is not None else None,
        out.data_ptr(), ctypes.c_void_p(stream))
    _native.raise_for_status(rc)
    return out


def algorithmic_bytes(n_kept_entries: int, n_out: int, feat_dim: int) -> int:
    """HBM gather bound of SURVEY.md 8(f): one feature row per kept neighbour + the output rows."""
    return 4 * feat_dim * (n_kept_entries + n_out) + 16 * n_kept_entries
