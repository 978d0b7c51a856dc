"""bench.py's multi-rank orchestration on CPU: `--gpus 2` rank code end to end under gloo -- sharding of every
step's batch, the fixed warm-up, the fences, the gather, the max-over-ranks reductions, ONE JSON line on rank 0 -- with the
compute swapped for the CPU oracle: this test passes its own `platform` (Graph class, events, device) to bench.run_rank.  The product
Graph class is not involved: this pins the launcher contract before an 8-GPU driver run meets it."""
import json
import os
import socket
import sys

import numpy as np
import torch
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


class _OracleGraph:
    """Stand-in with the slice of grand_plus_amd.api.Graph that bench.run_rank uses; rows come from the CPU oracle."""

    def __init__(self, indptr, indices, seed, device=0):
        self.indptr, self.indices = indptr, indices
        self._st = None
        self.reset_stats()

    def set_option(self, key, value):
        pass

    def reset_stats(self):
        self._st = {k: 0 for k in ("rows", "pushes", "edges", "filled", "support", "frontier", "lds_levels", "global_levels", "failed_rows",
                                   "degree_lookups", "retried_rows", "max_level_edges", "max_log_records", "workspace_bytes",
                                   "sketch_candidate_edges", "sketch_second_sweeps", "diag_ticks_total")}
        self._st.update(kernel_ms=0.0, workgroups=1, block_threads=64, lds_bytes=0, lds_slots=0, kernel=1, diag_sub=[0] * 16)

    def gfpush_device(self, seeds, coef, rmax, K, row, col, val, filled):
        from oracle import pyoracle
        s = seeds.numpy()
        r, c, v, st = pyoracle.gfpush(self.indptr, self.indices, s, coef, rmax, K)
        nf = (v.reshape(len(s), K) > 0).sum(1)
        row[:len(s) * K] = torch.from_numpy(r); col[:len(s) * K] = torch.from_numpy(c)
        val[:len(s) * K] = torch.from_numpy(v); filled[:len(s)] = torch.from_numpy(nf.astype(np.int32))
        self._st["rows"] += len(s); self._st["pushes"] += int(st["pushes"]); self._st["edges"] += int(st["edges"]); self._st["filled"] += int(st["filled"])

    def stats(self):
        return dict(self._st)


class _HostEvent:
    """torch.cuda.Event stand-in (perf_counter stamps)."""
    def __init__(self):
        self.t = 0.0

    def record(self):
        import time
        self.t = time.perf_counter()

    def elapsed_time(self, other):
        return (other.t - self.t) * 1e3


class _CpuPlatform:
    """The stand-in for bench.CudaPlatform: CPU tensors, host clocks, the oracle-backed Graph above."""
    name = "cpu test stand-in"
    graph_class = staticmethod(lambda: _OracleGraph)
    event = staticmethod(lambda: _HostEvent())
    device = staticmethod(lambda local_rank: torch.device("cpu"))
    sync = staticmethod(lambda dev: None)


def _rank(rank, world, port, out_dir):
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), WORLD_SIZE=str(world), RANK=str(rank), LOCAL_RANK=str(rank),
                      GRANDPLUS_BENCH_BACKEND="gloo")
    import bench
    sys.argv = ["bench.py", "--gpus", str(world), "--steps", "2", "--warmup", "1", "--prewarm", "2", "--workload", "pubmed",
                "--seeds-per-gpu", "96", "--no-cpu-baseline", "--no-host-api", "--no-next-rows"]
    out = open(os.path.join(out_dir, f"rank{rank}.out"), "w")
    sys.stdout = out
    rc = bench.run_rank(bench.parse_args(), platform=_CpuPlatform)
    out.flush()
    assert rc == 0


def test_bench_two_ranks_under_gloo_print_one_json_line(tmp_path):
    world = 2
    mp.spawn(_rank, args=(world, _free_port(), str(tmp_path)), nprocs=world, join=True)
    lines0 = [ln for ln in open(tmp_path / "rank0.out").read().splitlines() if ln.startswith("{")]
    lines1 = [ln for ln in open(tmp_path / "rank1.out").read().splitlines() if ln.startswith("{")]
    assert len(lines0) == 1 and not lines1                      # rank 0 prints ONE line, the others none
    line = json.loads(lines0[0])
    assert line["n_gpus"] == 2 and line["steps"] == 2 and line["warmup"] == 1 and line["warmup_effective"] == 3
    assert line["scaling"] == "weak" and line["unit"] == "rows/s" and line["value"] > 0 and line["anomaly"] is False
    assert line["config"]["rows_per_step"] == 192 and line["config"]["seeds_per_gpu"] == 96
    assert len(line["kernel_ms_per_rank"]) == 2 and line["rccl_ranks"] == 0        # gloo stand-in, not RCCL
    assert abs(line["value"] - 192 * 2 / (line["ms_per_step"] * 2 * 1e-3)) < 1e-3 * line["value"]
    assert "roofline" in line and line["roofline"]["bound"] == "hbm" and line["vs_baseline"] is None


def test_bench_has_no_cpu_path_of_its_own(monkeypatch):
    """bench.py selects nothing but the product: with the default platform and no GPU the rank stops at GP_ERR_NO_DEVICE."""
    import pytest
    if torch.cuda.is_available():
        pytest.skip("a GPU is present: the default platform would run the benchmark")
    sys.path.insert(0, ROOT)
    import bench
    assert not hasattr(bench, "_GRAPH_FACTORY") and "GRANDPLUS_BENCH_DEVICE" not in open(bench.__file__).read()
    monkeypatch.setattr(sys, "argv", ["bench.py", "--workload", "pubmed", "--seeds-per-gpu", "8", "--steps", "1", "--warmup", "0"])
    assert bench.run_rank(bench.parse_args()) == 3
