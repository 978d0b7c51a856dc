#!/usr/bin/env python3
"""Per-phase time of the sketch kernel (thread 0's 100 MHz stamps, -DGP_SK_TIMING build = libgrandplus_skt.so; the stamps cost a
few per cent).  Usage: GRANDPLUS_LIB=libgrandplus_skt.so python tools/sk_phases.py [workload] [rows] [key=value ...]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch, bench
from grand_plus_amd import Graph
from grand_plus_amd.recipes import RECIPES
name = sys.argv[1] if len(sys.argv) > 1 else "mag"
S = int(sys.argv[2]) if len(sys.argv) > 2 else 65536
extra = dict(kv.split("=") for kv in sys.argv[3:])
source, rkey, _ = bench.WORKLOADS[name]
ip, ix = bench.load_graph(source, os.cpu_count() or 8)
r = RECIPES[rkey]
seeds = torch.from_numpy(bench.make_seeds(source, len(ip) - 1, S).astype(np.int32)).cuda()
g = Graph(ip, ix, 0)
g.set_option("kernel", 2)
for k, v in extra.items():
    g.set_option(k, int(v))
for _ in range(3):
    g.reset_stats(); g.gfpush_device(seeds, r.coef(), r.rmax, r.top_k); torch.cuda.synchronize()
    st = g.stats()
d = st["diag_sub"]
dx = g.diag_counters()
rows = max(d[15], 1)
names = ["prologue+level0", "stream small", "stream sketch", "stream last", "filter", "scan", "level loop rest", "topk R+threshold", "topk sweeps", "topk select", "output+row end"]
tot = sum(d[:11])
print(f"{name} {extra}: kernel {st['kernel_ms']:.3f} ms, {st['block_threads']}x{st['lds_bytes']} wgs {st['workgroups']}, rows stamped {rows}, retried {st['retried_rows']}, "
      f"{tot / rows / 100:.1f} us per row and workgroup; small levels/row {d[13] / rows:.2f} sketch levels/row {d[14] / rows:.2f}")
for i, n in enumerate(names):
    print(f"   {n:18s} {d[i] / rows / 100:8.2f} us/row  {d[i] / max(tot, 1):6.3f}")
x = dx
n2 = ["filter: call -> entry", "filter: body", "filter: -> returned", "filter: -> behind the barrier",
      "scan: call -> entry", "scan: -> compaction done", "scan: -> cheap test done", "scan: -> lookups + push entries done", "scan: -> end of function",
      "scan: -> returned", "scan: -> behind the barrier", "stream (levels with a push): call -> returned", "stream: -> behind the barrier"]
print(f"   filter calls/row {d[11] / rows:.2f}  scan calls/row {d[12] / rows:.2f}  second rounds {st['sketch_second_sweeps']}  candidate edges {st['sketch_candidate_edges'] / max(st['edges'], 1):.3f}")
for i, n in enumerate(n2):
    print(f"   {n:55s} {x[i] / rows / 100:8.2f} us/row")
g.close()
