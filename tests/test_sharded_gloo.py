"""The N > 1 path on CPU: world_size-2 gloo run of the seed-sharding driver.

The compute callback is the only thing swapped (the CPU oracle stands in for the HIP
kernel, which tests may do); partitioning, padding of the last shard, the packed
all-gather and the unpacking are the product code of grand_plus_amd/sharded.py.
"""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank, world, port, n_seeds, K, out_dir):
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from grand_plus_amd import synth
    from grand_plus_amd.recipes import make_coef
    from grand_plus_amd.sharded import gfpush_sharded, scatter_filled_to_numpy, shard_range
    from oracle import pyoracle

    indptr, indices = synth.shape_csr("tiny")
    seeds = synth.seeds(len(indptr) - 1, n_seeds)
    coef = make_coef("ppr", 4, 0.2)
    lo, hi, per = shard_range(n_seeds, world, rank)

    def compute(seeds_local, row, col, val, filled):            # CPU stand-in for Graph.gfpush_device
        s = seeds_local.numpy()
        r, c, v, _ = pyoracle.gfpush(indptr, indices, s, coef, 1e-4, K)
        nf = (v.reshape(len(s), K) > 0).sum(1)
        row[:len(s) * K] = torch.from_numpy(r); col[:len(s) * K] = torch.from_numpy(c)
        val[:len(s) * K] = torch.from_numpy(v); filled[:len(s)] = torch.from_numpy(nf.astype(np.int32))

    local = torch.from_numpy(seeds[lo:hi].copy())
    row, col, val, filled = gfpush_sharded(compute, local, per, K, n_seeds, torch.device("cpu"))
    ro = np.full(n_seeds * K, -7, np.int32); co = np.full(n_seeds * K, -7, np.int32); vo = np.full(n_seeds * K, -7.0)
    scatter_filled_to_numpy(row, col, val, filled, K, ro, co, vo)
    np.savez(os.path.join(out_dir, f"rank{rank}.npz"), row=ro, col=co, val=vo)
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("n_seeds", [101, 64])           # ragged last shard, and an even split
def test_two_rank_gloo_matches_single_process(tmp_path, n_seeds):
    K, world = 8, 2
    mp.spawn(_worker, args=(world, _free_port(), n_seeds, K, str(tmp_path)), nprocs=world, join=True)
    from grand_plus_amd import synth
    from grand_plus_amd.recipes import make_coef
    from oracle import pyoracle
    indptr, indices = synth.shape_csr("tiny")
    seeds = synth.seeds(len(indptr) - 1, n_seeds)
    er, ec, ev, _ = pyoracle.gfpush(indptr, indices, seeds, make_coef("ppr", 4, 0.2), 1e-4, K,
                                    np.full(n_seeds * K, -7, np.int32), np.full(n_seeds * K, -7, np.int32),
                                    np.full(n_seeds * K, -7.0))
    for rank in range(world):
        z = np.load(tmp_path / f"rank{rank}.npz")
        np.testing.assert_array_equal(z["row"], er)         # every rank holds the whole matrix,
        np.testing.assert_array_equal(z["col"], ec)         # unfilled slots keep the caller's pre-fill
        np.testing.assert_array_equal(z["val"], ev)


def test_shard_range_covers_all_seeds():
    from grand_plus_amd.sharded import shard_range
    for S in (0, 1, 7, 8, 1000):
        for world in (1, 2, 3, 8):
            spans = [shard_range(S, world, r) for r in range(world)]
            assert spans[0][0] == 0 and spans[-1][1] == S
            assert all(a[1] == b[0] for a, b in zip(spans, spans[1:]))
            assert all(hi - lo <= per for lo, hi, per in spans)
