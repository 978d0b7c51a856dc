#!/usr/bin/env python3
"""bench.py -- GFPush propagation-matrix rows/s on MI355X (one process per GPU).

A "step" is one pass of the hot path over one batch of seeds: per GPU `--seeds-per-gpu`
rows of the workload's recipe; with N > 1 ranks the seed batch is sharded (weak scaling:
per-GPU rows fixed) and ONE RCCL all-gather reassembles the sparse row matrix on every rank.
Inputs (CSR, seeds) are resident in HBM before the timed region.

Default workload = the configuration BASELINE.json's metric and targets are quoted on:
MAG-Scholar-C-shape synthetic power-law CSR (12.4 M nodes / 173 M edges + self-loops),
ppr order 10 alpha 0.2 rmax 1e-5 top-k 32 (scripts/run_mag.sh:7 of the reference).

`python bench.py --gpus N` with no WORLD_SIZE in the environment LAUNCHES the N ranks itself
(one child process per GPU, rendezvous on 127.0.0.1) before anything in this process touches
HIP; under `torch.distributed.run` (WORLD_SIZE set) it is one of the ranks.  N = 1 runs the
same rank code in-process.

Prints ONE JSON line on rank 0 (see the driver contract in the task statement), carrying
`roofline` (HBM, algorithmic bytes of SURVEY.md 8d / kernel time from HIP events on the
launch stream), `issue_bound` (edge inserts per clock and CU against the measured LDS-atomic
insert rate), `host_api` (the metric's own clock: host buffers in -> host buffers out through
gp_gfpush, N = 1) and, at N = 1, `cpu_baseline` (the CPU checker timed on this box's cores).
"""
from __future__ import annotations

import argparse
import hashlib
import json
import os
import socket
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

import numpy as np  # noqa: E402

HBM_PEAK_GBS = 8000.0       # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec
SHADER_CLOCK_HZ = 2.4e9     # MI355X_MICROARCH.md: max clock (the chip holds less under load: the fraction below is a lower bound)
N_CUS = 256
# tools/micro/lds_random.hip on MI355X (profiles/r02_lds_random.txt): one residue-table insert = LDS compare-and-swap
# + ds_add_f64 on random slots, 16 waves per CU -> 38.2 clk per 64-lane wave-insert = 1.675 edge inserts / clk / CU
LDS_INSERT_PEAK = 64.0 / 38.2

WORKLOADS = {
    # name: (graph source, recipe key, description)
    "mag": ("synth:mag", ("mag", "ppr"), "MAG-Scholar-C-shape synthetic power-law CSR 12.4M nodes / 173M edges (+I), ppr order 10 alpha 0.2 rmax 1e-5 K 32"),
    "amazon2m": ("synth:amazon2m", ("amazon2m", "ppr"), "Amazon2M-shape synthetic power-law CSR 2.45M nodes / 61M edges (+I), ppr order 6 alpha 0.2 rmax 1e-6 K 64"),
    "reddit": ("synth:reddit", ("reddit", "avg"), "Reddit-shape synthetic power-law CSR 233k nodes / 11.6M edges (+I), avg order 6 rmax 1e-5 K 64"),
    "pubmed": ("golden:pubmed", ("pubmed", "ppr"), "Pubmed graph fixture 19.7k nodes / 88.6k edges (+I), ppr order 6 alpha 0.5 rmax 1e-5 K 16, seeds cycled"),
    "cora": ("golden:cora", ("cora", "ppr"), "Cora graph fixture 2.7k nodes / 10.6k edges (+I), ppr order 20 alpha 0.2 rmax 1e-7 K 32, seeds cycled"),
    "small": ("synth:small", ("mag", "ppr"), "100k-node synthetic power-law CSR, MAG recipe (debug)"),
}


# S of the reference's ONE call per process (model.py:244-248 with the shipped --unlabel_num)
REFERENCE_S = {"mag": 10400, "amazon2m": 12350, "reddit": 12050, "pubmed": 1559, "cora": 1639, "small": 2048}


def kernel_source_sha16() -> str:
    """Identity of the GFPush kernel sources: profiles/*_pmc_summary.json carries the hash it was measured on."""
    h = hashlib.sha256()
    for f in ("grand_plus_amd/csrc/gfpush_kernels.hpp", "grand_plus_amd/csrc/gfpush_sketch.hpp", "grand_plus_amd/csrc/gfpush.hip"):
        h.update(open(os.path.join(ROOT, f), "rb").read())
    return h.hexdigest()[:16]


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
    """Time the CPU checker on a bounded sample of the same workload (rank 0, N = 1 only): the reference itself
    (oracle/_ref, 40 OpenMP threads as shipped, graph.h:41) when it was built, and this repo's restatement at
    40 threads and on all cores; best of 3 each, on >= 16 384 seeds when the budget allows."""
    from oracle import pyoracle
    coef = recipe.coef()
    cores = os.cpu_count() or 1
    K = recipe.top_k
    probe = seeds[:min(len(seeds), 512)]
    t = time.perf_counter()
    pyoracle.gfpush(indptr, indices, probe, coef, recipe.rmax, K, threads=min(40, cores))
    rate = len(probe) / max(time.perf_counter() - t, 1e-6)
    n = int(min(len(seeds), max(16384 if rate * budget_s / 3 >= 16384 else 512, rate * budget_s / 3)))
    sample = seeds[:n]

    def best_of(fn, reps=3):
        best = 0.0
        for _ in range(reps):
            t0 = time.perf_counter()
            fn()
            best = max(best, n / (time.perf_counter() - t0))
        return round(best, 1)

    port40 = best_of(lambda: pyoracle.gfpush(indptr, indices, sample, coef, recipe.rmax, K, threads=min(40, cores)))
    port_all = best_of(lambda: pyoracle.gfpush(indptr, indices, sample, coef, recipe.rmax, K, threads=cores)) if cores > 40 else port40
    out = {"value": port40, "unit": "rows/s", "cores": min(40, cores), "host_cores": cores, "kind": "port",
           "sample": f"{n} seeds of the same workload, oracle/gfpush_oracle.cpp (unordered_map + OpenMP dynamic), {min(40, cores)} threads, best of 3",
           "port_40_threads": port40, "port_all_cores": port_all, "port_all_cores_threads": cores}
    ref = pyoracle.load_reference_module()
    if ref is not None:
        g = ref.Graph(indptr, indices, 0)
        row = np.zeros(n * K, np.int32); col = np.zeros(n * K, np.int32); val = np.zeros(n * K, np.float64)
        refv = best_of(lambda: g.gfpush_omp(sample, row, col, val, coef, recipe.rmax, K))
        out.update({"value": refv, "cores": 40, "kind": "reference",
                    "sample": f"{n} seeds of the same workload, reference precompute/propagation.cpp compiled as oracle/_ref, 40 OpenMP threads as shipped (graph.h:41) on {cores} cores, best of 3"})
    return out


def cold_call_child(workload: str) -> int:
    """The drop-in's real call pattern (model.py:251,268): a FRESH process makes ONE constructor and ONE gfpush_omp
    call of the reference's own S through the pybind11 module, wall-clock -- HIP runtime start, CSR upload, device-side
    validation, the self-addressed copy, workspace and pinned slabs, code upload, kernel -- and, beside it, the reference
    (oracle/_ref, 40 OpenMP threads as shipped) making the same two calls on this box's cores.  Prints one JSON line."""
    from grand_plus_amd import RECIPES, _native
    from precompute import propagation                       # the drop-in module, same import path as model.py:9
    source, rkey, _ = WORKLOADS[workload]
    recipe = RECIPES[rkey]
    coef, K = recipe.coef(), recipe.top_k
    indptr, indices = load_graph(source, os.cpu_count() or 8)
    n_nodes = len(indptr) - 1
    S = min(REFERENCE_S.get(workload, 10000), 4 * n_nodes)
    node_idx = make_seeds(source, n_nodes, S).astype(np.int64)           # (the shipped caller passes int64: model.py:244-248)
    out = {"S": S, "K": K, "what": "fresh process: propagation.Graph(indptr, indices, 0) + ONE gfpush_omp of the reference's own S, wall-clock"}

    def one_call(g):
        row = np.zeros(S * K, np.int32); col = np.zeros(S * K, np.int32); val = np.zeros(S * K, np.float64)     # model.py:252-254
        t = time.perf_counter()
        g.gfpush_omp(node_idx, row, col, val, coef, recipe.rmax, K)
        return time.perf_counter() - t, (row, col, val)

    t0 = time.perf_counter()
    n_dev = _native.lib().gp_device_count()                  # the first HIP calls of the process: the runtime and the device context start here
    if n_dev > 0:
        _native.lib().gp_internal_warm_device(0)
    t_init = time.perf_counter() - t0
    if n_dev <= 0:
        print(json.dumps({"error": "no HIP device"}), flush=True)
        return 3
    t0 = time.perf_counter()
    g = propagation.Graph(indptr, indices, 0)
    t_ctor = time.perf_counter() - t0
    import ctypes
    parts = (ctypes.c_double * 5)()
    try:
        _native.lib().gp_internal_create_ms(parts)
    except Exception:
        pass
    t_call, got = one_call(g)
    t_call2, _ = one_call(g)
    out.update({"gpu_s": round(t_init + t_ctor + t_call, 4), "hip_runtime_start_s": round(t_init, 4), "ctor_s": round(t_ctor, 4),
                "ctor_ms_parts": {k: round(float(v), 2) for k, v in zip(("device", "alloc", "upload", "validate", "objects"), list(parts))},
                "first_call_s": round(t_call, 4), "second_call_s": round(t_call2, 4)})
    del g
    from oracle import pyoracle
    ref = pyoracle.load_reference_module()
    if ref is not None:
        t0 = time.perf_counter()
        rg = ref.Graph(indptr, indices, 0)
        rt_ctor = time.perf_counter() - t0
        rt_call, exp = one_call(rg)
        out.update({"cpu_reference_s": round(rt_ctor + rt_call, 4), "cpu_reference_ctor_s": round(rt_ctor, 4), "cpu_reference_call_s": round(rt_call, 4),
                    "cpu_reference": f"oracle/_ref (the reference compiled as it lies), 40 OpenMP threads as shipped (graph.h:41) on {os.cpu_count()} cores",
                    "speedup": round((rt_ctor + rt_call) / (t_init + t_ctor + t_call), 2)})
        from grand_plus_amd.parity import compare_rows
        rep = compare_rows(node_idx, K, got, exp)
        out["rows_equal_the_references"] = bool(rep.ok)
    else:
        threads = min(40, os.cpu_count() or 1)
        t0 = time.perf_counter()
        pyoracle.gfpush(indptr, indices, node_idx, coef, recipe.rmax, K, threads=threads)
        out.update({"cpu_reference_s": round(time.perf_counter() - t0, 4), "cpu_reference": f"oracle/gfpush_oracle.cpp (port), {threads} threads", })
        out["speedup"] = round(out["cpu_reference_s"] / out["gpu_s"], 2)
    print(json.dumps(out), flush=True)
    return 0


def cold_call(workload: str):
    """Runs cold_call_child in a child process.  Called BEFORE this process touches HIP (the child starts its own runtime)."""
    try:
        r = subprocess.run([sys.executable, os.path.abspath(__file__), "--cold-call-child", "--workload", workload],
                           capture_output=True, text=True, timeout=900)
        lines = [ln for ln in r.stdout.strip().splitlines() if ln.startswith("{")]
        if lines:
            return json.loads(lines[-1])
        return {"error": f"child exited with code {r.returncode}", "stderr": r.stderr[-500:]}
    except Exception as e:                                   # never lose the headline line to a side measurement
        return {"error": f"{type(e).__name__}: {e}"}


def next_rows(graph, packed, per, K, n_nodes, nnz, dev):
    """One measured line each for the two SURVEY.md 8(f) kernels that consume this path's output, on the data the
    benchmark has resident: the fused feature augmentation (random_prop, model.py:80-87) over the rows GFPush just
    wrote, validation-sized batch (model.py:143), and two steps of predict()'s exact propagation (model.py:186-210)
    on the resident CSR.  HBM-gather bound, algorithmic bytes as in bench_augment.py / bench_propagate.py."""
    import torch
    from grand_plus_amd.augment import algorithmic_bytes as aug_bytes, random_prop_rows
    out = {}

    def timed(fn, iters):
        """min and median of `iters` individually timed calls after three warm-ups (two iterations after two
        warm-ups once read 3x slow on the driver's box and nothing in the line could say why)."""
        for _ in range(3):
            fn()
        torch.cuda.synchronize(dev)
        ms = []
        for _ in range(iters):
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record(); fn(); b.record(); torch.cuda.synchronize(dev)
            ms.append(a.elapsed_time(b))
        ms.sort()
        return ms[0], ms[len(ms) // 2], ms[-1]

    def clock_mhz():
        try:
            sys.path.insert(0, os.path.join(ROOT, "tools"))
            import device_probe
            return round(device_probe.shader_clock_mhz(dev.index or 0), 0)
        except Exception:                                    # the probe is a tools/ aid: its absence must not cost the line
            return None

    try:
        F, B = 128, min(10000, per)
        X = torch.randn((n_nodes, F), device=dev)
        rows = (torch.arange(B, dtype=torch.int64) * 7919 % per).to(torch.int32).to(dev)
        out["shader_clock_mhz_before"] = clock_mhz()
        lo, ms, hi = timed(lambda: random_prop_rows(X, packed.col, packed.val, packed.filled, K, batch_rows=rows, training=False), 20)
        kept = int(packed.filled[rows.long()].sum().item())
        by = aug_bytes(kept, B, F)
        out["augment"] = {"kernel": "random_prop_rows_kernel", "batch_rows": B, "K": K, "feat_dim": F, "ms": round(ms, 4), "ms_min": round(lo, 4), "ms_max": round(hi, 4),
                          "iters": 20, "achieved_GBps": round(by / ms / 1e6, 1), "frac_of_8TBps": round(by / ms / 1e6 / HBM_PEAK_GBS, 4),
                          "anomaly": bool(ms > 1.5 * lo), "parity": "unpinned (torch_scatter is un-vendored: oracle/random_prop_ref.py is this repo's restatement)"}
        Fp, steps = 32, 2
        Xp = X[:, :Fp].contiguous()
        outp = torch.empty_like(Xp)
        lo, ms, hi = timed(lambda: graph.propagate_features(Xp, "ppr", steps, 0.2, out=outp), 10)
        by = (4 * Fp * nnz + 4 * nnz + 12 * n_nodes * Fp) * steps
        out["propagate"] = {"kernel": "spmm_kernel", "feat_dim": Fp, "steps": steps, "ms_per_step": round(ms / steps, 3), "ms_per_step_min": round(lo / steps, 3),
                            "ms_per_step_max": round(hi / steps, 3), "iters": 10,
                            "achieved_GBps": round(by / ms / 1e6, 1), "frac_of_8TBps": round(by / ms / 1e6 / HBM_PEAK_GBS, 4),
                            "anomaly": bool(ms > 1.5 * lo), "parity": "unpinned (model.py:14 imports the un-vendored torch_scatter: oracle/predict_ref.py is this repo's restatement)"}
        out["shader_clock_mhz_after"] = clock_mhz()
    except Exception as e:                                   # never lose the headline line to a side measurement
        out["error"] = f"{type(e).__name__}: {e}"
    return out


def free_port() -> int:
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def launch_ranks(n: int) -> int:
    """`--gpus N` without a launcher: start N rank processes (this file, one per GPU) and wait.  Runs before this
    process imports torch or touches HIP, and it never replaces itself with another program."""
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    env.update({"MASTER_ADDR": "127.0.0.1", "MASTER_PORT": str(free_port()), "WORLD_SIZE": str(n)})
    print(f"[bench] launching {n} ranks (one process per GPU, rendezvous 127.0.0.1:{env['MASTER_PORT']})", file=sys.stderr, flush=True)
    procs = []
    for r in range(n):
        e = dict(env, RANK=str(r), LOCAL_RANK=str(r))
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env=e))
    rc = 0
    for r, p in enumerate(procs):
        code = p.wait()
        if code != 0:
            print(f"[bench] rank {r} exited with code {code}", file=sys.stderr, flush=True)
            rc = rc or code
    return rc


def parse_args():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--workload", default="mag", choices=sorted(WORKLOADS))
    ap.add_argument("--seeds-per-gpu", type=int, default=65536)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-host-api", action="store_true")
    ap.add_argument("--no-next-rows", action="store_true", help="skip the augmentation / propagation lines (SURVEY.md 8f)")
    ap.add_argument("--prewarm", type=int, default=50, help="untimed launches of the first warmup batch in front of the W warmup steps (a fixed count: "
                    "the same on every rank, no collective); reported as part of warmup_effective")
    ap.add_argument("--no-settle", action="store_true", help="same as --prewarm 0")
    ap.add_argument("--cpu-budget-s", type=float, default=15.0)
    ap.add_argument("--block-threads", type=int, default=0)
    ap.add_argument("--lds-bytes", type=int, default=0)
    ap.add_argument("--force-global", type=int, default=0)
    ap.add_argument("--opt", action="append", default=[], metavar="KEY=VALUE", help="gp_set_option on the graph (A/B runs), e.g. --opt seedrow=0")
    ap.add_argument("--diag-flags", type=int, default=0, help="GRANDPLUS_DIAG=1 builds only: bit 0 skips TOP-K (instruction attribution)")
    ap.add_argument("--no-cold-call", action="store_true", help="skip the cold_call block (a fresh child process: constructor + one reference-sized call)")
    ap.add_argument("--cold-call-child", action="store_true", help=argparse.SUPPRESS)
    return ap.parse_args()


class CudaPlatform:
    """What run_rank needs from the machine: the product Graph class, HIP events on torch's current stream, device handles.
    tests/test_bench_gloo.py passes its own stand-in to run_rank to drive the rank orchestration on CPU tensors; nothing in
    this file selects anything else."""
    name = "cuda"

    @staticmethod
    def graph_class():
        from grand_plus_amd import Graph
        return Graph

    @staticmethod
    def event():
        import torch
        return torch.cuda.Event(enable_timing=True)

    @staticmethod
    def device(local_rank):
        import torch
        torch.cuda.set_device(local_rank)
        return torch.device("cuda", local_rank)

    @staticmethod
    def sync(dev):
        import torch
        torch.cuda.synchronize(dev)


def run_rank(args, platform=CudaPlatform, cold=None) -> int:
    import torch
    import torch.distributed as dist
    from grand_plus_amd import RECIPES, algorithmic_bytes
    from grand_plus_amd.sharded import PackedRows, gfpush_sharded
    Graph = platform.graph_class()
    Event = lambda enable_timing=True: platform.event()       # noqa: E731
    dev_sync = platform.sync

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if os.environ.get("GRANDPLUS_BENCH_FORCE_DEVICE") is not None:      # debugging aid: several ranks on one GPU
        local_rank = int(os.environ["GRANDPLUS_BENCH_FORCE_DEVICE"])
    # RCCL ("nccl") is the product path.  GRANDPLUS_BENCH_BACKEND=gloo is a debugging aid for boxes with a
    # single GPU (RCCL refuses two ranks on one device): same orchestration, the gather is staged through the host.
    backend = os.environ.get("GRANDPLUS_BENCH_BACKEND", "nccl")
    rccl_ranks = 1
    if world > 1:
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
            rccl_ranks = dist.get_world_size()
        else:
            dist.init_process_group(backend)
            rccl_ranks = 0
    if world != args.gpus and rank == 0:
        print(f"[bench] note: --gpus {args.gpus} but WORLD_SIZE={world}; using WORLD_SIZE", file=sys.stderr)

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
    try:
        from grand_plus_amd import _native
        if _native.lib().gp_device_count() > local_rank:             # the first HIP calls of the process: the runtime and the device context start here,
            _native.lib().gp_internal_warm_device(local_rank)        # not inside the constructor's clock
    except Exception:
        pass
    t_runtime = time.perf_counter() - t0
    t0 = time.perf_counter()
    try:
        graph = Graph(indptr, indices, 0, device=local_rank)        # no GPU => GP_ERR_NO_DEVICE (there is no CPU path)
    except RuntimeError as e:
        print(f"[bench] rank {rank}: {e}", file=sys.stderr, flush=True)
        if world > 1:
            dist.destroy_process_group()
        return 3
    t_upload = time.perf_counter() - t0
    dev = platform.device(local_rank)
    if args.block_threads:
        graph.set_option("block_threads", args.block_threads)
    if args.lds_bytes:
        graph.set_option("lds_bytes", args.lds_bytes)
    if args.force_global:
        graph.set_option("force_global", 1)
    if args.diag_flags:
        graph.set_option("diag_flags", args.diag_flags)
    for kv in args.opt:
        k, v = kv.split("=")
        graph.set_option(k, int(v))

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
        dev_sync(dev)
        if world > 1:
            dist.barrier()
            dev_sync(dev)

    # Untimed launches in front of the W warmup steps the driver asks for: a freshly started process spends its first launches on
    # first-touch page faults of the workspace, code-object upload and clock ramp-up (the first MAG launch takes 38-41 ms against
    # 27 ms, profiles/r04_slow_phase_run1.jsonl).  A FIXED count -- the same on every rank, no collective, nothing keyed on an
    # earlier result -- reported in the JSON line as part of `warmup_effective`.  Nothing is ever discarded: the timed block is the
    # first and only one, and `anomaly` says when it ran more than 1.5x slower than those launches (reported, not acted on).
    prewarm = 0 if args.no_settle else max(0, args.prewarm)
    pre_ms = []
    for _ in range(prewarm):
        a, b = Event(enable_timing=True), Event(enable_timing=True)
        a.record(); step(0); b.record(); dev_sync(dev)
        pre_ms.append(a.elapsed_time(b))
    for i in range(args.warmup):
        step(i)
    fence()
    graph.reset_stats()
    ev = [(Event(enable_timing=True), Event(enable_timing=True)) for _ in range(args.steps)]
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
    stats = graph.stats()                       # counters of the timed steps on this rank; raises if a row failed
    kernel_ms = [a.elapsed_time(b) for a, b in ev]
    avg_ms = sum(kernel_ms) / len(kernel_ms)
    pre_med = sorted(pre_ms[len(pre_ms) // 2:])[len(pre_ms[len(pre_ms) // 2:]) // 2] if pre_ms else None     # median of the later half
    per_rank_ms = [round(avg_ms, 3)]
    if world > 1:
        red_dev = dev if backend == "nccl" else "cpu"
        t = torch.tensor([elapsed], dtype=torch.float64, device=red_dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
        km = torch.zeros(world, dtype=torch.float64, device=red_dev)
        km[rank] = avg_ms
        dist.all_reduce(km, op=dist.ReduceOp.SUM)
        per_rank_ms = [round(float(x), 3) for x in km.tolist()]

    if rank == 0:
        rows_total = S_step * args.steps
        value = rows_total / elapsed
        bytes_per_launch = algorithmic_bytes(stats) / args.steps
        achieved = bytes_per_launch / (avg_ms * 1e-3) / 1e9
        edges_per_launch = stats["edges"] / args.steps
        edges_per_clk_cu = edges_per_launch / (avg_ms * 1e-3) / SHADER_CLOCK_HZ / N_CUS
        sha = kernel_source_sha16()
        sk_line = stats.get("kernel") == 2
        insert_share = stats.get("sketch_candidate_edges", 0) / max(stats["edges"], 1) if sk_line else 1.0
        line = {
            "metric": "propagation-matrix rows/sec (whole node)", "value": round(value, 1), "unit": "rows/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "warmup_effective": args.warmup + prewarm,
            "prewarm_step_ms_median": None if pre_med is None else round(pre_med, 3),
            "anomaly": bool(pre_med is not None and world == 1 and avg_ms > 1.5 * pre_med),
            "ms_per_step": round(elapsed / args.steps * 1e3, 3), "higher_is_better": True,
            "scaling": "weak", "vs_baseline": None, "dtype": "f64",
            "data": "synthetic" if source.startswith("synth:") else "fixture graph (tests/golden), seeds cycled",
            "value_clock": "inputs resident in HBM when the timed region starts (task contract); the metric's own clock -- host buffers in -> "
                           "host buffers out through gp_gfpush, PCIe included -- is host_api.rows_per_s in this same line",
            "rccl_ranks": rccl_ranks, "kernel_ms_per_rank": per_rank_ms,
            "config": {"workload": desc, "recipe": f"{recipe.prop_mode} order {recipe.order} alpha {recipe.alpha} rmax {recipe.rmax} K {K}",
                       "seeds_per_gpu": per, "rows_per_step": S_step, "n_nodes": n_nodes, "nnz": int(len(indices)),
                       "sharding": "seeds block-partitioned, CSR replicated" + (", 1 RCCL all-gather of packed rows per step" if world > 1 else "")},
            "roofline": {"bound": "hbm", "achieved": round(achieved, 2), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": round(achieved / HBM_PEAK_GBS, 5), "traffic": None,
                         "kernel": "gfpush_sk_kernel (+ gfpush_retry_kernel for the rows it hands back)" if stats.get("kernel") == 2 else "gfpush_kernel",
                         "kernel_ms_avg": round(avg_ms, 3), "kernel_sha16": sha,
                         "algorithmic_bytes_per_launch": int(bytes_per_launch),
                         "bytes_per_row": round(bytes_per_launch / per, 1)},
            # the path is not HBM-bound: second yardstick = residue-table inserts per clock and CU against the
            # LDS-atomic insert rate measured by tools/micro/lds_random.hip (clock taken at its 2.4 GHz maximum)
            "issue_bound": {"bound": "lds_atomic_insert", "achieved": round(edges_per_clk_cu * insert_share, 5), "peak": round(LDS_INSERT_PEAK, 3),
                            "unit": "table inserts/clk/CU", "frac": round(edges_per_clk_cu * insert_share / LDS_INSERT_PEAK, 5),
                            "counts": "edges that reach the exact table (sketch kernel: sketch_candidate_edges; the other edges are keyless ds_add_u32)" if sk_line else "every traversed edge",
                            "clock_hz": SHADER_CLOCK_HZ, "source": "tools/micro/lds_random.hip -> profiles/r02_lds_random.txt"},
            "detail": {"pushes_per_row": round(stats["pushes"] / stats["rows"], 1),
                       "edges_per_row": round(stats["edges"] / stats["rows"], 1),
                       "support_per_row": None if sk_line else round(stats["support"] / stats["rows"], 1),        # (not counted by the sketch kernel)
                       "frontier_per_row": None if sk_line else round(stats["frontier"] / stats["rows"], 1),
                       "degree_lookups_per_row": round(stats["degree_lookups"] / stats["rows"], 1),
                       "edge_pushes_per_s_per_gpu": round(stats["edges"] / args.steps / (avg_ms * 1e-3), 0),
                       "lds_levels": stats["lds_levels"], "global_levels": stats["global_levels"],
                       "workgroups": stats["workgroups"], "block_threads": stats["block_threads"],
                       "lds_bytes": stats["lds_bytes"], "workspace_gb": round(stats["workspace_bytes"] / 2**30, 2),
                       "retried_rows": stats["retried_rows"], "kernel_kind": stats.get("kernel"),
                       "sketch_candidate_edge_fraction": round(stats.get("sketch_candidate_edges", 0) / max(stats["edges"], 1), 4),
                       "sketch_second_rounds_per_row": round(stats.get("sketch_second_sweeps", 0) / max(stats["rows"], 1), 4),
                       "max_level_edges": stats["max_level_edges"],
                       "max_log_records": stats["max_log_records"],
                       "graph_gen_s": round(t_gen, 2), "csr_upload_s": round(t_upload, 3), "hip_runtime_start_s": round(t_runtime, 3)},
        }
        # HBM-side traffic per launch: measured with rocprofv3 --pmc (separate FETCH_SIZE / WRITE_SIZE passes of this
        # same command, tools/collect_pmc.sh) and committed under profiles/ together with the hash of the kernel
        # sources it was measured on; a bench run cannot profile itself, and a profile of OTHER sources is not reported.
        try:
            import glob
            cands = sorted(glob.glob(os.path.join(ROOT, "profiles", f"r*_{args.workload}_pmc_summary.json")), reverse=True)
            seen_other = None
            for path in cands:                  # newest round first: the profile of THESE kernel sources, if there is one
                prof = json.load(open(path))
                if args.workload != prof.get("workload") or per != prof.get("seeds_per_gpu") or world != 1:
                    continue
                pname = os.path.basename(path)
                if prof.get("kernel_sha16") == sha:
                    dd = prof["derived"]
                    line["roofline"]["traffic"] = int(dd["hbm_read_bytes_corrected"] + dd["hbm_write_bytes"])
                    line["roofline"]["traffic_over_algorithmic"] = round(line["roofline"]["traffic"] / bytes_per_launch, 2)
                    line["roofline"]["traffic_source"] = (f"profiles/{pname}: rocprofv3 --pmc, separate FETCH_SIZE / WRITE_SIZE passes of this command on the same "
                                                          "kernel sources; reads = TCC_EA0_RDREQ x 128 B (one request per 128-byte L2 line on gfx950, "
                                                          "calibrated by tools/fetch_calib.sh -> profiles/r03_fetch_calib.json; FETCH_SIZE tallies 64), writes = WRITE_SIZE")
                    seen_other = None
                    break
                seen_other = seen_other or f"none: profiles/{pname} was measured on kernel sources {prof.get('kernel_sha16')}, this run is {sha}"
            if seen_other:
                line["roofline"]["traffic_source"] = seen_other
        except (OSError, KeyError, ValueError):
            pass
        if stats.get("diag_ticks_total"):
            tot = stats["diag_ticks_total"]
            line["detail"]["diag_phase_share"] = {k: round(stats[f"diag_ticks_{k}"] / tot, 3) for k in ("scan", "expand", "topk", "scan_hbm", "expand_hbm")}
            names = ["agg_init", "agg_insert", "agg_scan", "sel_hist", "sel_pick", "sel_compact", "sel_collect", "final"]
            line["detail"]["diag_topk_sub_share"] = {n: round(stats["diag_sub"][i] / tot, 3) for i, n in enumerate(names)}
            line["detail"]["diag_counts_per_row"] = {"agg_parts": round(stats["diag_sub"][8] / stats["rows"], 2), "sel_passes_hbm": round(stats["diag_sub"][9] / stats["rows"], 2), "sel_passes_lds": round(stats["diag_sub"][10] / stats["rows"], 2)}
        if world == 1 and not args.no_host_api:
            # The metric's own clock: the gfpush_omp call, host buffers in -> host buffers out
            # (model.py:268), through gp_gfpush: seed upload, kernel, one packed D2H, scatter of the v > 0 slots.
            hs = all_seeds[args.warmup * S_step:args.warmup * S_step + per].astype(np.int64)
            row = np.zeros(per * K, np.int32); col = np.zeros(per * K, np.int32); val = np.zeros(per * K, np.float64)
            ts = []
            for _ in range(4):
                t1 = time.perf_counter()
                graph.gfpush_omp(hs, row, col, val, coef, recipe.rmax, K)
                ts.append(time.perf_counter() - t1)
            med = sorted(ts[1:])[1]
            k_same = graph.stats()["kernel_ms"]                  # (gp_gfpush reports its own call: the kernel time of THIS seed batch, not the timed steps' average)
            line["host_api"] = {"rows_per_s": round(per / med, 1), "ms_per_call": round(med * 1e3, 3), "rows_per_call": per,
                                "first_call_ms": round(ts[0] * 1e3, 3), "kernel_ms_same_batch": round(k_same, 3),
                                "call_over_kernel": round(med * 1e3 / k_same, 4) if k_same else None,
                                "what": "Graph.gfpush_omp (gp_gfpush): int64 seeds on the host -> numpy row/col/value filled in place, median of 3 calls after one warm-up"}
        if cold is not None:
            line["cold_call"] = cold
        if world == 1 and not args.no_next_rows:
            line["next_rows"] = next_rows(graph, packed, per, K, n_nodes, len(indices), dev)
        if world == 1 and not args.no_cpu_baseline:
            cb = cpu_baseline(indptr, indices, all_seeds[args.warmup * S_step:], recipe, args.cpu_budget_s)
            line["cpu_baseline"] = cb
            # like clocks: the CPU number is host buffers in -> host buffers out, so it is compared with the host-API rate
            # (gp_gfpush, PCIe included); the device-resident `value` over the same CPU number is kept beside it
            if "host_api" in line:
                line["detail"]["gpu_over_cpu"] = round(line["host_api"]["rows_per_s"] / cb["value"], 1)
                line["detail"]["gpu_over_cpu_clock"] = "host buffers in -> host buffers out on both sides (host_api.rows_per_s / cpu_baseline.value)"
            line["detail"]["gpu_device_resident_over_cpu"] = round(value / cb["value"], 1)
        print(json.dumps(line), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()
    return 0


def main():
    args = parse_args()
    if args.cold_call_child:
        sys.exit(cold_call_child(args.workload))
    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        sys.exit(launch_ranks(args.gpus))
    cold = None
    if "WORLD_SIZE" not in os.environ and args.gpus == 1 and not args.no_cold_call:
        cold = cold_call(args.workload)                      # (a child process, started before this one touches HIP)
    sys.exit(run_rank(args, cold=cold))


if __name__ == "__main__":
    main()
