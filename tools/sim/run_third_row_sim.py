#!/usr/bin/env python3
"""Drives tools/sim/third_row_sim.cpp on one of bench.py's workloads (CPU only).  Usage: run_third_row_sim.py [workload] [rows] [lg cells]"""
import ctypes, os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np
import bench
from grand_plus_amd.recipes import RECIPES
name = sys.argv[1] if len(sys.argv) > 1 else "mag"
rows = int(sys.argv[2]) if len(sys.argv) > 2 else 2048
lg = int(sys.argv[3]) if len(sys.argv) > 3 else 13
so = os.path.join(ROOT, "tools", "sim", "libthird_row_sim.so")
src = os.path.join(ROOT, "tools", "sim", "third_row_sim.cpp")
if not os.path.exists(so) or os.path.getmtime(so) < os.path.getmtime(src):
    subprocess.run(["g++", "-O3", "-fopenmp", "-shared", "-fPIC", "-o", so, src], check=True)
source, rkey, _ = bench.WORKLOADS[name]
ip, ix = bench.load_graph(source, os.cpu_count() or 8)
r = RECIPES[rkey]
n = len(ip) - 1
seeds = bench.make_seeds(source, n, rows).astype(np.int32)
deg_sat = 127 if name == "mag" else 255                     # what the self-addressed copy leaves for the degree field on these shapes
coef = r.coef(); L = len(coef) - 1
out = np.zeros((L + 1, 10)); ra = np.zeros(rows, np.uint32); rb = np.zeros(rows, np.uint32); rs = np.zeros(rows, np.uint32)
lib = ctypes.CDLL(so)
P = lambda a: a.ctypes.data_as(ctypes.c_void_p)
lib.third_row_sim.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int64, ctypes.c_void_p, ctypes.c_int64, ctypes.c_void_p, ctypes.c_int,
                              ctypes.c_double, ctypes.c_uint32, ctypes.c_uint32, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p]
lib.third_row_sim(P(ip), P(ix), n, P(seeds), rows, P(coef), len(coef), r.rmax, deg_sat, lg, P(out), P(ra), P(rb), P(rs))
out /= rows
print(f"{name}: N {n} nnz {len(ix)} rows {rows} rmax {r.rmax} L {L} deg_sat {deg_sat}, {1 << lg} cells; per row:")
print("level    edges  targets  pushers | 32-bit cells: cand edges  cand nodes | 16-bit cells, per-level quantum: cand edges  cand nodes | fp32 superset  missed pushers  quantum doubled")
for l in range(1, L + 1):
    o = out[l]
    print(f"{l:5d} {o[0]:8.1f} {o[1]:8.1f} {o[2]:8.1f} | {o[3]:10.1f} {o[4]:10.1f} | {o[5]:10.1f} {o[6]:10.1f} | {o[7]:9.1f} {o[8]:9.3f} {o[9]:7.3f}")
t = out.sum(0)
print(f"  sum {t[0]:8.1f} {t[1]:8.1f} {t[2]:8.1f} | {t[3]:10.1f} {t[4]:10.1f} | {t[5]:10.1f} {t[6]:10.1f} | {t[7]:9.1f} {t[8]:9.3f} {t[9]:7.3f}")
print(f"  candidate edges / edges: 32-bit {t[3] / t[0]:.3f}, 16-bit {t[5] / t[0]:.3f}; superset / pushers {t[7] / max(t[2], 1):.3f}")
for q in (50, 90, 99, 100):
    print(f"  row quantile {q}: largest level's candidate nodes 32-bit {np.percentile(ra, q):7.0f}, 16-bit {np.percentile(rb, q):7.0f}, fp32 superset {np.percentile(rs, q):7.0f}")
