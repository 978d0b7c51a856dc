#!/usr/bin/env python3
"""Rows/s per workgroup as the number of persistent workgroups shrinks: if a workgroup runs no faster with most of the
chip idle, nothing chip-wide (HBM, fabric, L2) binds it.  Round 4: the round-3 launch shape (3 x 512 threads x 52 KB per CU),
(a) fewer workgroups on the whole chip (`max_workgroups`), (b) 1 / 2 / 3 workgroups on EVERY CU at the same table size
(`lds_pad`: dynamic LDS the tables do not use).  Usage: python tools/exp_wg_scale.py [workload]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch, bench
from grand_plus_amd import Graph
from grand_plus_amd.recipes import RECIPES
name = sys.argv[1] if len(sys.argv) > 1 else "mag"
source, rkey, _ = bench.WORKLOADS[name]
ip, ix = bench.load_graph(source, os.cpu_count() or 8)
r = RECIPES[rkey]
g = Graph(ip, ix, 0)
bt, lds = 512, 53248


def run(tag, n_wg, pad, S):
    seeds = torch.from_numpy(bench.make_seeds(source, len(ip) - 1, S).astype(np.int32)).cuda()
    g.set_option("block_threads", bt); g.set_option("lds_bytes", lds)
    g.set_option("max_workgroups", n_wg); g.set_option("lds_pad", pad)
    best = 1e9
    for _ in range(4):
        g.reset_stats(); g.gfpush_device(seeds, r.coef(), r.rmax, r.top_k); torch.cuda.synchronize()
        st = g.stats(); best = min(best, st["kernel_ms"])
    wg = st["workgroups"]
    print(f"{name} {bt}x{lds} {tag}: workgroups {wg:4d} rows {S:6d} kernel {best:8.3f} ms -> {S / best:9.1f} rows/ms, "
          f"{best * 1e3 * wg / S:7.1f} us per row and workgroup, retried {st['retried_rows']}", flush=True)


for n_wg in (0, 512, 256, 64, 8):
    run("max_workgroups", n_wg, 0, 32768 if n_wg == 0 or n_wg >= 256 else 64 * n_wg)
for per_cu, pad in ((3, 0), (2, 81920 - lds), (1, 163840 - lds)):
    run(f"{per_cu} per CU on all CUs (lds_pad {pad})", 0, pad, 32768)
g.close()
