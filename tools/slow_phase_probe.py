#!/usr/bin/env python3
"""Times MAG launches batch by batch; when one is slow (> 60 ms), re-creates the Graph (new CSR + workspace allocations) and
times the same batches again: tells an allocation-placement effect from a time-based one (settling)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import bench, device_probe
from grand_plus_amd import Graph, _native
from grand_plus_amd.recipes import RECIPES

source, rkey, _ = bench.WORKLOADS["mag"]
ip, ix = bench.load_graph(source, os.cpu_count() or 8)
r = RECIPES[rkey]
seeds = bench.make_seeds(source, len(ip) - 1, 65536 * 8)
batches = [torch.from_numpy(seeds[i * 65536:(i + 1) * 65536].astype(np.int32)).cuda() for i in range(8)]
t00 = time.time()

def run(g, tag):
    slow = False
    for rnd in range(2):
        for i, b in enumerate(batches):
            a, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record(); g.gfpush_device(b, r.coef(), r.rmax, r.top_k); e.record(); torch.cuda.synchronize()
            ms = a.elapsed_time(e); st = g.stats()
            print(tag, "t=%.1f" % (time.time() - t00), "round", rnd, "batch", i, "ms %.2f" % ms, "workspace_gb %.2f" % (st["workspace_bytes"] / 2**30), "retried", st["retried_rows"], flush=True)
            slow = slow or (rnd == 1 and ms > 60.0)
    return slow

for attempt in range(4):
    g = Graph(ip, ix, 0)
    slow = run(g, "graph%d" % attempt)
    print("graph%d" % attempt, "slow" if slow else "fast", "probe", device_probe.speed_probe(0), flush=True)
    g.close()
