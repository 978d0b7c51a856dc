#!/usr/bin/env python3
"""Rows/s per workgroup as the number of persistent workgroups shrinks: if a workgroup runs no faster with most of the
chip idle, nothing chip-wide (HBM, fabric, L2) binds it.  Usage: python tools/exp_wg_scale.py [workload]"""
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
for bt, lds in ((512, 81920), (1024, 163840)):
    for n_wg in (0, 256, 128, 64, 32, 8):
        S = 16384 if n_wg == 0 or n_wg >= 128 else 64 * n_wg
        seeds = torch.from_numpy(bench.make_seeds(source, len(ip) - 1, S).astype(np.int32)).cuda()
        g = Graph(ip, ix, 0)
        g.set_option("block_threads", bt); g.set_option("lds_bytes", lds)
        if n_wg: g.set_option("max_workgroups", n_wg)
        best = 1e9
        for _ in range(3):
            g.reset_stats(); g.gfpush_device(seeds, r.coef(), r.rmax, r.top_k); torch.cuda.synchronize()
            st = g.stats(); best = min(best, st["kernel_ms"])
        wg = st["workgroups"]
        print(f"{name} {bt}x{lds}: workgroups {wg:4d} rows {S:6d} kernel {best:8.3f} ms -> {S / best:9.1f} rows/ms, {S / best / wg * 1000:8.1f} rows/s per workgroup ({best * 1e3 * wg / S:7.1f} us per row and workgroup)", flush=True)
        g.close()
