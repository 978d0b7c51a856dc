#!/usr/bin/env python3
"""the driver's bench line once showed `propagate_features` at 50 ms per step where every other run has 16.5 ms.
One FRESH process per run, doing what bench.py does in front of that measurement -- graph upload, device calls, the host-API calls
(gp_gfpush: pinned slabs, zero-copy writes, the slab-reset thread), then the propagation steps, each timed on its own with HIP
events -- and the shader clock before and after.  Usage: python tools/slow_propagate_runs.py [n_processes] [workload]
Prints one JSON line per process: {"run", "clock_mhz_before", "clock_mhz_after", "ms_per_step": [...], "slow": bool}."""
import json, os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

CHILD = r'''
import json, os, sys, time
sys.path.insert(0, %r); sys.path.insert(0, os.path.join(%r, "tools"))
import numpy as np, torch, bench, device_probe
from grand_plus_amd import Graph
from grand_plus_amd.recipes import RECIPES
name = sys.argv[1]
source, rkey, _ = bench.WORKLOADS[name]
ip, ix = bench.load_graph(source, os.cpu_count() or 8)
r = RECIPES[rkey]; K = r.top_k; S = 65536
n = len(ip) - 1
seeds = bench.make_seeds(source, n, S)
g = Graph(ip, ix, 0)
d = torch.from_numpy(seeds.astype(np.int32)).cuda()
for _ in range(3):
    g.gfpush_device(d, r.coef(), r.rmax, K)
torch.cuda.synchronize()
row = np.zeros(S * K, np.int32); col = np.zeros(S * K, np.int32); val = np.zeros(S * K)
for _ in range(4):
    g.gfpush_omp(seeds.astype(np.int64), row, col, val, r.coef(), r.rmax, K)
c0 = device_probe.shader_clock_mhz(0)
X = torch.randn((n, 32), device="cuda"); out = torch.empty_like(X)
ms = []
for i in range(12):
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record(); g.propagate_features(X, "ppr", 2, 0.2, out=out); b.record(); torch.cuda.synchronize()
    ms.append(round(a.elapsed_time(b) / 2, 3))
c1 = device_probe.shader_clock_mhz(0)
print(json.dumps({"clock_mhz_before": round(c0), "clock_mhz_after": round(c1), "ms_per_step": ms, "slow": max(ms[2:]) > 1.5 * min(ms)}))
''' % (ROOT, ROOT)

n = int(sys.argv[1]) if len(sys.argv) > 1 else 8
w = sys.argv[2] if len(sys.argv) > 2 else "mag"
slow = 0
for i in range(n):
    p = subprocess.run([sys.executable, "-c", CHILD, w], capture_output=True, text=True)
    line = [l for l in p.stdout.splitlines() if l.startswith("{")]
    if not line:
        print(json.dumps({"run": i, "error": p.stderr[-300:]}), flush=True); continue
    d = json.loads(line[-1]); d["run"] = i; slow += bool(d["slow"])
    print(json.dumps(d), flush=True)
print(json.dumps({"processes": n, "slow_processes": slow, "workload": w}), flush=True)
