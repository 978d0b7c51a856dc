#!/usr/bin/env python3
"""Time-bounded randomised differential test of the sketch kernel against the CPU oracle on graphs large enough for sketch
levels, partition walks and partitioned TOP-K rounds: random power-law graphs (65 k - 400 k nodes, average degree 4 - 40),
random recipes (levels, coefficients, rmax 5e-6 .. 2e-4, K 1 .. 128), random sketch geometries, duplicate and hub seeds.
Rows tie-aware identical, pushes / edges / filled exactly the oracle's.   Usage: python tools/fuzz_sketch.py [seconds] [first case]   (FUZZ_KERNEL=1: the general kernel under its launch shapes, rmax down to 1e-6)"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
from grand_plus_amd import synth
from test_gpu_parity import _assert_parity, _oracle, _run_gpu

budget = float(sys.argv[1]) if len(sys.argv) > 1 else 300.0
case = int(sys.argv[2]) if len(sys.argv) > 2 else 0
t0 = time.time(); done = 0; kinds = {}
while time.time() - t0 < budget:
    rng = np.random.default_rng(77000 + case)
    n = int(rng.integers(65536, 400000))
    indptr, indices = synth.powerlaw_csr(n, int(n * rng.uniform(2, 20)), seed=int(rng.integers(1, 1 << 30)), offset=int(rng.integers(2, 30)))
    L = int(rng.integers(2, 11))
    mode = rng.choice(["ppr", "avg", "rand"])
    if mode == "ppr":
        a = float(rng.uniform(0.1, 0.6)); coef = a * (1 - a) ** np.arange(L + 1)
    elif mode == "avg":
        coef = np.ones(L + 1)
    else:
        coef = rng.random(L + 1); coef[rng.random(L + 1) < 0.2] = 0.0
        if coef.sum() == 0: coef[-1] = 1.0
    coef = coef / coef.sum() * float(rng.choice([1.0, 1.0, 0.3, 1.7]))
    rmax = float(10.0 ** rng.uniform(np.log10(5e-6), np.log10(2e-4)))
    K = int(rng.choice([1, 8, 16, 32, 64, 100, 128]))
    S = 384
    seeds = synth.seeds(n, S, seed=int(rng.integers(1, 1 << 30))).astype(np.int64)
    deg = np.diff(indptr)
    seeds[:4] = np.argsort(deg)[-4:]                       # hub seeds
    seeds[10:13] = seeds[10]                               # duplicates
    opts = {"kernel": 2}
    g = int(rng.integers(0, 6))
    if os.environ.get("FUZZ_KERNEL") == "1":               # the general kernel under its launch shapes instead
        opts = {"kernel": 1}; g = -1
        shape = int(rng.integers(0, 4))
        if shape == 1: opts.update(block_threads=768, lds_bytes=81920)
        if shape == 2: opts.update(block_threads=1024, lds_bytes=163840)
        if shape == 3: opts.update(block_threads=512, lds_bytes=40960)
        rmax = float(10.0 ** rng.uniform(np.log10(1e-6), np.log10(2e-4)))
    if g == 1: opts.update(sk_block_threads=512)
    if g == 2: opts.update(sk_lg_mu=int(rng.integers(10, 13)), sk_lg_mr=int(rng.integers(9, 12)))
    if g == 3: opts.update(sk_target=int(rng.choice([1, 8, 64, 1024])))
    if g == 4: opts.update(sk_direct_max=int(rng.choice([1, 64, 100000])))
    if g == 5: opts.update(sk_lg_mu=9, sk_lg_mr=8)
    if case % 5 == 4: opts.update(sk_seed_merge=0) if opts["kernel"] == 2 else opts.update(gk_acsr=0)    # (round 6) the paths the new defaults replace
    if case % 2: opts.update(max_workgroups=int(rng.choice([3, 8, 64])))       # a workgroup takes MANY rows one after the other (round 5: state left behind by a row)
    label = f"fuzz {case}: n {n} nnz {len(indices)} {mode} L{L} rmax {rmax:.2e} K{K} {opts}"
    got, st = _run_gpu(indptr, indices, seeds, coef, rmax, K, options=opts)
    exp, ost = _oracle(indptr, indices, seeds, coef, rmax, K)
    _assert_parity(seeds, K, got, exp, label=label)
    assert (st["pushes"], st["edges"], st["filled"]) == (ost["pushes"], ost["edges"], ost["filled"]), (label, st, ost)
    assert st["failed_rows"] == 0, label
    kinds[st["kernel"]] = kinds.get(st["kernel"], 0) + 1
    print(f"ok {label}: kernel {st['kernel']} retried {st['retried_rows']} cand {st['sketch_candidate_edges']} sweeps2 {st['sketch_second_sweeps']} edges/row {st['edges'] / S:.0f}", flush=True)
    done += 1; case += 1
print(f"fuzz_sketch: {done} cases passed in {time.time() - t0:.0f} s; kernels used {kinds}")
