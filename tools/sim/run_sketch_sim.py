#!/usr/bin/env python3
"""Drives tools/sim/sketch_sim.cpp on one of bench.py's workloads (CPU only).  Usage: run_sketch_sim.py [workload] [rows]"""
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
r = RECIPES[rkey]
n = len(ip) - 1
seeds = bench.make_seeds(source, n, rows).astype(np.int32)
id_bits = 1
while id_bits < 31 and (1 << id_bits) < n: id_bits += 1
deg_sat = (1 << (31 - id_bits)) - 1
sizes = np.array([2048, 4096, 8192, 16384], dtype=np.uint32)
coef = r.coef(); L = len(coef) - 1
stride = 4 + 4 * len(sizes)
out = np.zeros((L + 1, stride)); rmc = np.zeros((rows, len(sizes)), np.uint32); rme = np.zeros(rows, np.uint32)
lib = ctypes.CDLL(os.path.join(ROOT, "tools", "sim", "libsketch_sim.so"))
P = lambda a: a.ctypes.data_as(ctypes.c_void_p)
lib.sketch_sim.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int64, ctypes.c_void_p, ctypes.c_int64, ctypes.c_void_p, ctypes.c_int,
                           ctypes.c_double, ctypes.c_uint32, ctypes.c_void_p, ctypes.c_int, ctypes.c_void_p, ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p]
lib.sketch_sim(P(ip), P(ix), n, P(seeds), rows, P(coef), len(coef), r.rmax, deg_sat, P(sizes), len(sizes), P(out), stride, P(rmc), P(rme))
out /= rows
print(f"{name}: N {n} nnz {len(ix)} rows {rows} rmax {r.rmax} L {L} deg_sat {deg_sat}")
print("level   edges  targets pushers | per sketch size M: cand nodes 1h / cand edges 1h / cand nodes 2h / cand edges 2h")
for l in range(1, L + 1):
    o = out[l]
    print(f"{l:5d} {o[0]:8.1f} {o[1]:8.1f} {o[2]:7.1f} | " + " | ".join(f"M={sizes[s]}: {o[4+4*s]:7.1f} {o[5+4*s]:7.1f} {o[6+4*s]:7.1f} {o[7+4*s]:7.1f}" for s in range(len(sizes))))
t = out.sum(0)
print(f"  sum {t[0]:8.1f} {t[1]:8.1f} {t[2]:7.1f} | " + " | ".join(f"M={sizes[s]}: {t[4+4*s]:7.1f} {t[5+4*s]:7.1f} {t[6+4*s]:7.1f} {t[7+4*s]:7.1f}" for s in range(len(sizes))))
for q in (50, 90, 99, 99.9, 100):
    print(f"  row quantile {q}: max level edges {np.percentile(rme, q):9.0f}; max level candidate nodes (1h) " + " ".join(f"M={sizes[s]}: {np.percentile(rmc[:, s], q):8.0f}" for s in range(len(sizes))))
