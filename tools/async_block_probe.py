#!/usr/bin/env python3
"""MAG launches in blocks of 5 WITHOUT host synchronisation in between (what bench.py's timed block does), alternating with
synchronised launches: does the 4-5x anomaly of DESIGN.md section 4 belong to back-to-back launches?"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import bench
from grand_plus_amd import Graph
from grand_plus_amd.recipes import RECIPES

source, rkey, _ = bench.WORKLOADS["mag"]
ip, ix = bench.load_graph(source, os.cpu_count() or 8)
r = RECIPES[rkey]
seeds = bench.make_seeds(source, len(ip) - 1, 65536 * 8)
batches = [torch.from_numpy(seeds[i * 65536:(i + 1) * 65536].astype(np.int32)).cuda() for i in range(8)]
K = r.top_k
row = torch.zeros(65536 * K, dtype=torch.int32, device="cuda"); col = torch.zeros_like(row)
val = torch.zeros(65536 * K, dtype=torch.float64, device="cuda"); filled = torch.zeros(65536, dtype=torch.int32, device="cuda")
g = Graph(ip, ix, 0)
t00 = time.time()
for blk in range(int(sys.argv[1]) if len(sys.argv) > 1 else 40):
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(5)]
    for j in range(5):
        filled.zero_(); ev[j][0].record(); g.gfpush_device(batches[(blk + j) % 8], r.coef(), r.rmax, K, row, col, val, filled); ev[j][1].record()
    torch.cuda.synchronize()
    a = [x.elapsed_time(y) for x, y in ev]
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(); g.gfpush_device(batches[blk % 8], r.coef(), r.rmax, K, row, col, val, filled); e1.record(); torch.cuda.synchronize()
    print("t=%.1f block %d async ms %s  sync ms %.1f  lib_kernel_ms %.1f" % (time.time() - t00, blk, " ".join("%.1f" % x for x in a), e0.elapsed_time(e1), g.stats()["kernel_ms"]), flush=True)
