#!/usr/bin/env python3
"""bench.py -- GFPush propagation-matrix rows/s on MI355X (one process per GPU).

A "step" is one pass of the hot path over one batch of seeds: per GPU `--seeds-per-gpu`
rows of the workload's recipe; with N > 1 ranks the seed batch is sharded (weak scaling:
per-GPU rows fixed) and ONE RCCL all-gather reassembles the sparse row matrix on every rank.
Inputs (CSR, seeds) are resident in HBM before the timed region.

Default workload = the configuration BASELINE.json's metric and targets are quoted on:
MAG-Scholar-C-shape synthetic power-law CSR (12.4 M nodes / 173 M edges + self-loops),
ppr order 10 alpha 0.2 rmax 1e-5 top-k 32 (scripts/run_mag.sh:7 of the reference).

Prints ONE JSON line on rank 0 (see the driver contract in the task statement), carrying
`roofline` (HBM, algorithmic bytes of SURVEY.md 8d / kernel time from HIP events on the
launch stream) and, at N = 1, `cpu_baseline` (the CPU checker timed on this box's cores).
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

import numpy as np  # noqa: E402

HBM_PEAK_GBS = 8000.0       # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec

WORKLOADS = {
    # name: (graph source, recipe key, description)
    "mag": ("synth:mag", ("mag", "ppr"), "MAG-Scholar-C-shape synthetic power-law CSR 12.4M nodes / 173M edges (+I), ppr order 10 alpha 0.2 rmax 1e-5 K 32"),
    "amazon2m": ("synth:amazon2m", ("amazon2m", "ppr"), "Amazon2M-shape synthetic power-law CSR 2.45M nodes / 61M edges (+I), ppr order 6 alpha 0.2 rmax 1e-6 K 64"),
    "reddit": ("synth:reddit", ("reddit", "avg"), "Reddit-shape synthetic power-law CSR 233k nodes / 11.6M edges (+I), avg order 6 rmax 1e-5 K 64"),
    "pubmed": ("golden:pubmed", ("pubmed", "ppr"), "Pubmed graph fixture 19.7k nodes / 88.6k edges (+I), ppr order 6 alpha 0.5 rmax 1e-5 K 16, seeds cycled"),
    "cora": ("golden:cora", ("cora", "ppr"), "Cora graph fixture 2.7k nodes / 10.6k edges (+I), ppr order 20 alpha 0.2 rmax 1e-7 K 32, seeds cycled"),
    "small": ("synth:small", ("mag", "ppr"), "100k-node synthetic power-law CSR, MAG recipe (debug)"),
}


def load_graph(source: str, threads: int):
    from grand_plus_amd import synth
    kind, name = source.split(":")
    if kind == "synth":
        synth.set_threads(threads)                 # world ranks generate the same graph concurrently
        return synth.shape_csr(name)
    z = np.load(os.path.join(ROOT, "tests", "golden", f"{name}.npz"))
    return z["indptr"], z["indices"]


def make_seeds(source: str, n_nodes: int, total: int):
    from grand_plus_amd import synth
    if total <= n_nodes:
        return synth.seeds(n_nodes, total)
    base = synth.seeds(n_nodes, n_nodes)          # small graphs: cycle a permutation of all nodes
    reps = -(-total // n_nodes)
    return np.tile(base, reps)[:total].copy()


def cpu_baseline(indptr, indices, seeds, recipe, budget_s: float = 12.0):
    """Time the CPU checker on a bounded sample of the same workload (rank 0, N = 1 only)."""
    from oracle import pyoracle
    coef = recipe.coef()
    cores = os.cpu_count() or 1
    out = {}
    # size the sample from a small probe so that the timed part is ~budget_s
    probe = seeds[:min(len(seeds), 256)]
    t = time.perf_counter()
    pyoracle.gfpush(indptr, indices, probe, coef, recipe.rmax, recipe.top_k, threads=cores)
    rate = len(probe) / max(time.perf_counter() - t, 1e-6)
    n = int(min(len(seeds), max(256, rate * budget_s)))
    sample = seeds[:n]
    best = 0.0
    for _ in range(2):
        t = time.perf_counter()
        pyoracle.gfpush(indptr, indices, sample, coef, recipe.rmax, recipe.top_k, threads=cores)
        best = max(best, n / (time.perf_counter() - t))
    out = {"value": round(best, 1), "unit": "rows/s", "cores": cores, "kind": "port",
           "sample": f"{n} seeds of the same workload, oracle/gfpush_oracle.cpp (unordered_map + OpenMP dynamic), {cores} threads, best of 2"}
    ref = pyoracle.load_reference_module()
    if ref is not None:
        # the reference itself (compiled from its own sources into oracle/_ref); 40 threads hard-coded (graph.h:41)
        g = ref.Graph(indptr, indices, 0)
        K = recipe.top_k
        row = np.zeros(n * K, np.int32); col = np.zeros(n * K, np.int32); val = np.zeros(n * K, np.float64)
        rbest = 0.0
        for _ in range(2):
            t = time.perf_counter()
            g.gfpush_omp(sample, row, col, val, coef, recipe.rmax, K)
            rbest = max(rbest, n / (time.perf_counter() - t))
        out = {"value": round(rbest, 1), "unit": "rows/s", "cores": 40, "host_cores": cores, "kind": "reference",
               "sample": f"{n} seeds of the same workload, reference precompute/propagation.cpp compiled as oracle/_ref, 40 OpenMP threads as shipped (graph.h:41) on {cores} cores, best of 2",
               "port_value": round(best, 1), "port_cores": cores}
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--workload", default="mag", choices=sorted(WORKLOADS))
    ap.add_argument("--seeds-per-gpu", type=int, default=65536)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-budget-s", type=float, default=12.0)
    ap.add_argument("--block-threads", type=int, default=0)
    ap.add_argument("--lds-bytes", type=int, default=0)
    ap.add_argument("--force-global", type=int, default=0)
    ap.add_argument("--diag-flags", type=int, default=0, help="GRANDPLUS_DIAG=1 builds only: bit 0 skips TOP-K (instruction attribution)")
    args = ap.parse_args()

    import torch
    import torch.distributed as dist
    from grand_plus_amd import Graph, RECIPES, algorithmic_bytes
    from grand_plus_amd.sharded import PackedRows, gfpush_sharded

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if os.environ.get("GRANDPLUS_BENCH_FORCE_DEVICE") is not None:      # debugging aid: several ranks on one GPU
        local_rank = int(os.environ["GRANDPLUS_BENCH_FORCE_DEVICE"])
    # RCCL ("nccl") is the product path.  GRANDPLUS_BENCH_BACKEND=gloo is a debugging aid for boxes with a
    # single GPU (RCCL refuses two ranks on one device): same orchestration, the gather is staged through the host.
    backend = os.environ.get("GRANDPLUS_BENCH_BACKEND", "nccl")
    if world > 1:
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
        else:
            dist.init_process_group(backend)
    if world != args.gpus and rank == 0:
        print(f"[bench] note: --gpus {args.gpus} but WORLD_SIZE={world}; using WORLD_SIZE", file=sys.stderr)
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)

    source, rkey, desc = WORKLOADS[args.workload]
    recipe = RECIPES[rkey]
    coef = recipe.coef()
    K = recipe.top_k
    threads = max(1, (os.cpu_count() or 1) // world)
    t0 = time.perf_counter()
    indptr, indices = load_graph(source, threads)
    n_nodes = len(indptr) - 1
    t_gen = time.perf_counter() - t0
    t0 = time.perf_counter()
    graph = Graph(indptr, indices, 0, device=local_rank)
    t_upload = time.perf_counter() - t0
    if args.block_threads:
        graph.set_option("block_threads", args.block_threads)
    if args.lds_bytes:
        graph.set_option("lds_bytes", args.lds_bytes)
    if args.force_global:
        graph.set_option("force_global", 1)
    if args.diag_flags:
        graph.set_option("diag_flags", args.diag_flags)

    per = args.seeds_per_gpu
    S_step = per * world
    n_steps_total = args.warmup + args.steps
    all_seeds = make_seeds(source, n_nodes, S_step * n_steps_total)
    # this rank's shard of every step's batch, resident in HBM before timing starts
    shards = []
    for i in range(n_steps_total):
        batch = all_seeds[i * S_step:(i + 1) * S_step]
        shards.append(torch.from_numpy(batch[rank * per:(rank + 1) * per].copy()).to(dev))
    packed = PackedRows(per, K, dev)
    gathered = torch.empty(world * packed.nbytes, dtype=torch.uint8, device=dev) if world > 1 else None

    def compute(seeds_local, row, col, val, filled):
        graph.gfpush_device(seeds_local, coef, recipe.rmax, K, row, col, val, filled)

    def step(i):
        if backend == "nccl" or world == 1:
            return gfpush_sharded(compute, shards[i], per, K, S_step, dev, packed=packed, gathered=gathered)
        packed.filled.zero_()
        compute(shards[i], packed.row, packed.col, packed.val, packed.filled)
        gather_rows()

    def gather_rows():
        if backend == "nccl":
            dist.all_gather_into_tensor(gathered, packed.buf[:packed.nbytes])
        else:                                           # debugging aid only
            host = torch.empty(world * packed.nbytes, dtype=torch.uint8)
            dist.all_gather_into_tensor(host, packed.buf[:packed.nbytes].cpu())
            gathered.copy_(host)

    def fence():
        torch.cuda.synchronize(dev)
        if world > 1:
            dist.barrier()
            torch.cuda.synchronize(dev)

    for i in range(args.warmup):
        step(i)
    fence()
    graph.reset_stats()
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(args.steps)]
    t_start = time.perf_counter()
    for j in range(args.steps):
        i = args.warmup + j
        packed.filled.zero_()
        ev[j][0].record()                       # torch's current stream == the launch stream
        compute(shards[i], packed.row, packed.col, packed.val, packed.filled)
        ev[j][1].record()
        if world > 1:
            gather_rows()
    fence()
    elapsed = time.perf_counter() - t_start
    if world > 1:
        t = torch.tensor([elapsed], dtype=torch.float64, device=dev if backend == "nccl" else "cpu")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())

    stats = graph.stats()                       # counters of the timed steps on this rank
    kernel_ms = [a.elapsed_time(b) for a, b in ev]
    if rank == 0:
        rows_total = S_step * args.steps
        value = rows_total / elapsed
        avg_ms = sum(kernel_ms) / len(kernel_ms)
        bytes_per_launch = algorithmic_bytes(stats) / args.steps
        achieved = bytes_per_launch / (avg_ms * 1e-3) / 1e9
        line = {
            "metric": "propagation-matrix rows/sec (whole node)", "value": round(value, 1), "unit": "rows/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(elapsed / args.steps * 1e3, 3), "higher_is_better": True,
            "scaling": "weak", "vs_baseline": None, "dtype": "f64", "data": "synthetic",
            "config": {"workload": desc, "recipe": f"{recipe.prop_mode} order {recipe.order} alpha {recipe.alpha} rmax {recipe.rmax} K {K}",
                       "seeds_per_gpu": per, "rows_per_step": S_step, "n_nodes": n_nodes, "nnz": int(len(indices)),
                       "sharding": "seeds block-partitioned, CSR replicated" + (", 1 RCCL all-gather of packed rows per step" if world > 1 else "")},
            "roofline": {"bound": "hbm", "achieved": round(achieved, 2), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": round(achieved / HBM_PEAK_GBS, 5), "traffic": None,
                         "kernel": "gfpush_kernel", "kernel_ms_avg": round(avg_ms, 3),
                         "algorithmic_bytes_per_launch": int(bytes_per_launch),
                         "bytes_per_row": round(bytes_per_launch / per, 1)},
            "detail": {"pushes_per_row": round(stats["pushes"] / stats["rows"], 1),
                       "edges_per_row": round(stats["edges"] / stats["rows"], 1),
                       "support_per_row": round(stats["support"] / stats["rows"], 1),
                       "frontier_per_row": round(stats["frontier"] / stats["rows"], 1),
                       "degree_lookups_per_row": round(stats["degree_lookups"] / stats["rows"], 1),
                       "edge_pushes_per_s_per_gpu": round(stats["edges"] / args.steps / (avg_ms * 1e-3), 0),
                       "lds_levels": stats["lds_levels"], "global_levels": stats["global_levels"],
                       "workgroups": stats["workgroups"], "block_threads": stats["block_threads"],
                       "lds_bytes": stats["lds_bytes"], "workspace_gb": round(stats["workspace_bytes"] / 2**30, 2),
                       "graph_gen_s": round(t_gen, 2), "csr_upload_s": round(t_upload, 3)},
        }
        # HBM-side traffic per launch: measured with rocprofv3 --pmc (separate FETCH_SIZE / WRITE_SIZE passes of
        # this same command) and committed under profiles/; a bench run cannot profile itself.
        try:
            prof = json.load(open(os.path.join(ROOT, "profiles", "r01_mag_pmc_summary.json")))
            if args.workload == prof.get("workload") and per == prof.get("seeds_per_gpu") and world == 1:
                line["roofline"]["traffic"] = int(prof["derived"]["hbm_read_bytes_raw"] + prof["derived"]["hbm_write_bytes"])
                line["roofline"]["traffic_source"] = "profiles/r01_mag_pmc_summary.json (rocprofv3 FETCH_SIZE+WRITE_SIZE, raw; see note there)"
        except (OSError, KeyError, ValueError):
            pass
        if stats.get("diag_ticks_total"):
            tot = stats["diag_ticks_total"]
            line["detail"]["diag_phase_share"] = {k: round(stats[f"diag_ticks_{k}"] / tot, 3) for k in ("scan", "expand", "topk", "scan_hbm", "expand_hbm")}
            names = ["agg_init", "agg_insert", "agg_scan", "sel_hist", "sel_pick", "sel_compact", "sel_collect", "final"]
            line["detail"]["diag_topk_sub_share"] = {n: round(stats["diag_sub"][i] / tot, 3) for i, n in enumerate(names)}
            line["detail"]["diag_counts_per_row"] = {"agg_parts": round(stats["diag_sub"][8] / stats["rows"], 2), "sel_passes_hbm": round(stats["diag_sub"][9] / stats["rows"], 2), "sel_passes_lds": round(stats["diag_sub"][10] / stats["rows"], 2)}
        if world == 1 and not args.no_cpu_baseline:
            cb = cpu_baseline(indptr, indices, all_seeds[args.warmup * S_step:], recipe, args.cpu_budget_s)
            line["cpu_baseline"] = cb
            line["detail"]["gpu_over_cpu"] = round(value / cb["value"], 1)
        print(json.dumps(line), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
