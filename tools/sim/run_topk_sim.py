#!/usr/bin/env python3
"""Drives tools/sim/topk_sim.cpp (CPU only).  Usage: run_topk_sim.py [workload] [rows]"""
import ctypes, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np
import bench
from grand_plus_amd.recipes import RECIPES
name = sys.argv[1] if len(sys.argv) > 1 else "mag"
rows = int(sys.argv[2]) if len(sys.argv) > 2 else 2048
source, rkey, _ = bench.WORKLOADS[name]
ip, ix = bench.load_graph(source, os.cpu_count() or 8)
r = RECIPES[rkey]; n = len(ip) - 1
seeds = bench.make_seeds(source, n, rows).astype(np.int32)
coef = r.coef()
lib = ctypes.CDLL(os.path.join(ROOT, "tools", "sim", "libtopk_sim.so"))
P = lambda a: a.ctypes.data_as(ctypes.c_void_p)
lib.topk_sim.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int64, ctypes.c_void_p, ctypes.c_int64, ctypes.c_void_p, ctypes.c_int,
                         ctypes.c_double, ctypes.c_int, ctypes.c_uint32, ctypes.c_uint32, ctypes.c_void_p, ctypes.c_void_p]
print(f"{name}: rows {rows} K {r.top_k}")
for M in (2048, 4096, 8192):
    for target in (2 * r.top_k, 4 * r.top_k):
        out = np.zeros(16); pr = np.zeros((rows, 4), np.uint32)
        lib.topk_sim(P(ip), P(ix), n, P(seeds), rows, P(coef), len(coef), r.rmax, r.top_k, M, target, P(out), P(pr))
        print(f"M {M:5d} target {target:4d}: sweep-1 nodes {out[0]/rows:8.1f} (max {out[7]:.0f}, p99 {np.percentile(pr[:,0],99):.0f}) records {out[1]/rows:8.1f} of {out[6]/rows:8.1f}; second sweep in {out[2]/rows*100:5.1f} % of rows, "
              f"nodes {out[3]/max(out[2],1):8.1f} (max {out[8]:.0f}); support {out[4]/rows:8.1f}; mismatches {out[5]:.0f}; mean kth {out[9]/rows:.3e} mean t_c {out[10]/rows:.3e}")
