#!/usr/bin/env python3
"""Times the GFPush kernel under the two launch shapes (1 x 1024 threads / 160 KB LDS per CU and
2 x 512 threads / 80 KB) on the fixture graphs and synthetic shapes, every recipe of scripts/run_*.sh.
The result table is what the automatic choice in gp_gfpush_device (gfpush.hip) is calibrated on."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch
import bench
from grand_plus_amd import Graph
from grand_plus_amd.recipes import RECIPES

CASES = [("golden:cora", "cora"), ("golden:citeseer", "citeseer"), ("golden:pubmed", "pubmed"),
         ("synth:small", "mag"), ("synth:reddit", "reddit"), ("synth:mag", "mag")]
only = sys.argv[1:]
for source, ds in CASES:
    if only and ds not in only and source.split(":")[1] not in only:
        continue
    ip, ix = bench.load_graph(source, os.cpu_count() or 8)
    n = len(ip) - 1
    seeds = torch.from_numpy(bench.make_seeds(source, n, 16384).astype(np.int32)).cuda()
    for mode in ("ppr", "avg", "single"):
        r = RECIPES[(ds, mode)]
        res = {}
        for name, (bt, lds) in {"auto": (0, 0), "1x1024": (1024, 163840), "2x512": (512, 81920)}.items():
            g = Graph(ip, ix, 0)
            if bt:
                g.set_option("block_threads", bt); g.set_option("lds_bytes", lds)
            best = 1e9
            for _ in range(4):
                g.reset_stats(); g.gfpush_device(seeds, r.coef(), r.rmax, r.top_k); torch.cuda.synchronize()
                best = min(best, g.stats()["kernel_ms"])
            st = g.stats()
            res[name] = best
            shape = f"{st['workgroups']}x{st['block_threads']}"
            extra = f"frontier/row {st['frontier'] / st['rows']:.0f} levels/row {(st['lds_levels'] + st['global_levels']) / st['rows']:.1f} edges/row {st['edges'] / st['rows']:.0f}"
            if name == "auto":
                auto_shape = shape
        print(f"{source:16s} {mode:6s} avg_deg {len(ix) / n:5.1f} rmax {r.rmax:g}: auto({auto_shape}) {res['auto']:.3f} ms | 1x1024 {res['1x1024']:.3f} | 2x512 {res['2x512']:.3f} | {extra}", flush=True)
