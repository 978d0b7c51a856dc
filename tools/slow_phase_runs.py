#!/usr/bin/env python3
"""One fresh process = one sample: 70 MAG launches of 65 536 rows, each timed with HIP events; prints min / median / max and
how many launches ran slower than 2 x the minimum.  Options on the command line as key=value (workspace_mb, pretouch, ...)."""
import os, sys, time, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import bench
from grand_plus_amd import Graph
from grand_plus_amd.recipes import RECIPES
opts = dict(kv.split("=") for kv in sys.argv[1:])
source, rkey, _ = bench.WORKLOADS["mag"]
ip, ix = bench.load_graph(source, os.cpu_count() or 8)
r = RECIPES[rkey]
seeds = bench.make_seeds(source, len(ip) - 1, 65536 * 4)
batches = [torch.from_numpy(seeds[i * 65536:(i + 1) * 65536].astype(np.int32)).cuda() for i in range(4)]
g = Graph(ip, ix, 0)
for k, v in opts.items():
    g.set_option(k, int(v))
ms = []
t0 = time.time()
for i in range(70):
    a, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record(); g.gfpush_device(batches[i % 4], r.coef(), r.rmax, r.top_k); e.record(); torch.cuda.synchronize()
    ms.append(a.elapsed_time(e))
st = g.stats()
m = np.array(ms[2:])
print(json.dumps({"opts": opts, "min": round(float(m.min()), 2), "median": round(float(np.median(m)), 2), "max": round(float(m.max()), 2),
                  "slow": int((m > 2 * m.min()).sum()), "n": len(m), "first": [round(x, 1) for x in ms[:4]], "workspace_gb": round(st["workspace_bytes"] / 2**30, 2),
                  "retried": st["retried_rows"], "wall_s": round(time.time() - t0, 1)}), flush=True)
g.close()
