"""Seeds sharded over the GPUs of one node, one process per GPU (torch.distributed).

GFPush rows are independent (`precompute/graph.h:73-127` keeps no state across seeds), so
the CSR is replicated on every GPU and the seed list is cut into `world` contiguous blocks
of ceil(S/world) rows.  There is no collective inside the computation; the only exchange is
ONE all-gather (RCCL over xGMI when the backend is "nccl") of a packed, fixed-stride buffer
per rank -- [value f64 | row i32 | col i32 | filled i32] -- that reassembles the sparse row
matrix on every rank.  The kernel writes straight into typed views of that buffer.
"""
from __future__ import annotations

import numpy as np


def shard_range(n_seeds: int, world: int, rank: int):
    """Contiguous block [lo, hi) of rank `rank`; every rank owns `per` slots (last ones padded)."""
    per = -(-n_seeds // world) if n_seeds else 0
    lo = min(rank * per, n_seeds)
    hi = min(lo + per, n_seeds)
    return lo, hi, per


def packed_stride(per: int, K: int) -> int:
    """Bytes one rank contributes to the all-gather: [val f64 | row i32 | col i32 | filled i32], 16-B padded."""
    return (16 * per * K + 4 * per + 15) // 16 * 16


class PackedRows:
    """One contiguous byte buffer holding `per` rows of K slots, with typed views into it."""

    def __init__(self, per: int, K: int, device):
        import torch
        n = per * K
        self.per, self.K = per, K
        self.nbytes = packed_stride(per, K)          # padded so that every rank's slice stays 16-byte aligned
        self.buf = torch.zeros(max(self.nbytes, 8), dtype=torch.uint8, device=device)
        self.val = self.buf[0:8 * n].view(torch.float64)
        self.row = self.buf[8 * n:12 * n].view(torch.int32)
        self.col = self.buf[12 * n:16 * n].view(torch.int32)
        self.filled = self.buf[16 * n:16 * n + 4 * per].view(torch.int32)


def unpack_gathered(gathered, world: int, per: int, K: int, n_seeds: int):
    """Split the all-gathered byte buffer back into (row, col, val, filled) of the first n_seeds rows."""
    import torch
    n = per * K
    stride = packed_stride(per, K)
    rows, cols, vals, fills = [], [], [], []
    for r in range(world):
        part = gathered[r * stride:(r + 1) * stride]
        vals.append(part[0:8 * n].view(torch.float64))
        rows.append(part[8 * n:12 * n].view(torch.int32))
        cols.append(part[12 * n:16 * n].view(torch.int32))
        fills.append(part[16 * n:16 * n + 4 * per].view(torch.int32))
    row = torch.cat(rows)[:n_seeds * K]
    col = torch.cat(cols)[:n_seeds * K]
    val = torch.cat(vals)[:n_seeds * K]
    filled = torch.cat(fills)[:n_seeds]
    return row, col, val, filled


def gfpush_sharded(compute, seeds_local, per: int, K: int, n_seeds: int, device, group=None,
                   packed: PackedRows | None = None, gathered=None, failed_rows=None):
    """Run `compute` on this rank's shard and all-gather the packed rows.

    compute(seeds_local, row, col, val, filled) must fill the first len(seeds_local) rows of
    the given views (dense per row for the first filled[it] slots) -- `Graph.gfpush_device`
    bound to its coef/rmax/K is the product path; tests inject a CPU stand-in under gloo.
    Returns (row, col, val, filled) for all n_seeds rows, on every rank.  Slots i >= filled[it] of a row hold
    whatever the buffer held before (a re-used `packed` keeps the previous step's data there): consumers mask by
    `filled`.  `failed_rows` (a callable returning this rank's count of rows that hit a workspace bound, e.g.
    `lambda: graph.stats()["failed_rows"]` -- it synchronises) makes a failure on ANY rank raise on EVERY rank
    instead of travelling through the gather as valid-looking empty rows.
    """
    import torch
    import torch.distributed as dist

    world = dist.get_world_size(group) if dist.is_initialized() else 1
    if packed is None:
        packed = PackedRows(per, K, device)
    else:
        packed.filled.zero_()
    if seeds_local.numel() > 0:
        compute(seeds_local, packed.row, packed.col, packed.val, packed.filled)
    if failed_rows is not None:
        try:
            bad = int(failed_rows())
        except RuntimeError:                            # Graph.stats() raises for failed rows on this rank
            bad = 1
        if world > 1:
            t = torch.tensor([bad], dtype=torch.int64, device=device if dist.get_backend(group) == "nccl" else "cpu")
            dist.all_reduce(t, op=dist.ReduceOp.SUM, group=group)
            bad = int(t.item())
        if bad:
            raise RuntimeError(f"{bad} row(s) hit a workspace bound on some rank (GP_ERR_OVERFLOW); the gathered rows are incomplete")
    if world == 1:
        return unpack_gathered(packed.buf[:packed.nbytes], 1, per, K, n_seeds)
    if gathered is None:
        gathered = torch.empty(world * packed.nbytes, dtype=torch.uint8, device=device)
    dist.all_gather_into_tensor(gathered, packed.buf[:packed.nbytes], group=group)
    return unpack_gathered(gathered, world, per, K, n_seeds)


def scatter_filled_to_numpy(row, col, val, filled, K, row_idx, col_idx, value):
    """Honour 'write only v > 0 slots' (`graph.h:121`) when handing the gathered rows to the
    caller's numpy arrays: slot it*K+i is written only for i < filled[it]."""
    f = filled.cpu().numpy()
    S = len(f)
    mask = (np.arange(K)[None, :] < f[:, None]).reshape(-1)
    row_idx[:S * K][mask] = row.cpu().numpy()[mask]
    col_idx[:S * K][mask] = col.cpu().numpy()[mask]
    value[:S * K][mask] = val.cpu().numpy()[mask]
