#!/usr/bin/env python3
"""Drives tools/sim/layout_sim.cpp on one of bench.py's workloads (CPU only): cache lines per row under the packed CSR and under
the self-addressed 64-B-aligned CSR, and the sizes of the per-row structures.  Usage: run_layout_sim.py [workload] [rows]"""
import ctypes, os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np
import bench
from grand_plus_amd.recipes import RECIPES
name = sys.argv[1] if len(sys.argv) > 1 else "mag"
rows = int(sys.argv[2]) if len(sys.argv) > 2 else 2048
UW = int(os.environ.get("UW", "32"))
here = os.path.dirname(os.path.abspath(__file__))
so = os.path.join(here, "liblayout_sim.so")
src = os.path.join(here, "layout_sim.cpp")
if not os.path.exists(so) or os.path.getmtime(so) < os.path.getmtime(src):
    subprocess.run(["g++", "-O3", "-fopenmp", "-shared", "-fPIC", "-o", so, src], check=True)
source, rkey, _ = bench.WORKLOADS[name]
ip, ix = bench.load_graph(source, os.cpu_count() or 8)
r = RECIPES[rkey]
n = len(ip) - 1
seeds = bench.make_seeds(source, n, rows).astype(np.int32)
deg = np.diff(ip).astype(np.int64)
units = np.maximum(1, -(-deg // UW))
pos = np.zeros(n + 1, np.int64); np.cumsum(units, out=pos[1:])
n_units = int(pos[-1])
pb = max(1, int(n_units).bit_length())
sat_new = (1 << (31 - pb)) - 1
id_bits = max(1, int(n - 1).bit_length())
sat_old = (1 << (31 - id_bits)) - 1
coef = r.coef(); L = len(coef) - 1
RS, LS = 12, 4
lib = ctypes.CDLL(so)
P = lambda a: a.ctypes.data_as(ctypes.c_void_p)
lib.layout_sim.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int64, ctypes.c_void_p, ctypes.c_int64, ctypes.c_int, ctypes.c_double,
                           ctypes.c_uint32, ctypes.c_void_p, ctypes.c_uint32, ctypes.c_void_p, ctypes.c_int, ctypes.c_void_p, ctypes.c_int]
print(f"{name}: N {n} nnz {len(ix)} rows {rows} rmax {r.rmax} L {L}; units of {UW} words: {n_units} ({n_units * UW * 4 / 2**30:.2f} GB against {len(ix) * 4 / 2**30:.2f} GB packed), "
      f"position bits {pb} -> degree field saturates at {sat_new} (node ids: {id_bits} bits, {sat_old}); dangling nodes {int((deg == 0).sum())}")
for sat in (sat_new, sat_old):
    ro = np.zeros((rows, RS), np.int64); lo = np.zeros((L + 1, LS))
    lib.layout_sim(P(ip), P(ix), n, P(seeds), rows, L, r.rmax, UW, P(pos), sat, P(ro), RS, P(lo), LS)
    m = ro.mean(0)
    print(f" deg_sat {sat}: saturated pushers / row {m[9]:.1f}, candidates at saturation (exact-degree lookups) / row {m[10]:.1f}")
lo /= rows
print(" level    edges  pushers  targets")
for l in range(1, L + 1):
    print(f" {l:5d} {lo[l][0]:8.1f} {lo[l][1]:8.1f} {lo[l][2]:8.1f}")
print(f" per row: pushes {m[0]:.0f} edges {m[1]:.0f} levels {m[11]:.1f}")
print(f" packed CSR : runs {m[4]:.0f} x 128-B lines = {m[4] * 128 / 1e3:.0f} KB ({m[5]:.0f} x 64-B sectors = {m[5] * 64 / 1e3:.0f} KB), indptr {m[6]:.0f} lines = {m[6] * 128 / 1e3:.0f} KB; useful {m[1] * 4 / 1e3:.0f} + {m[0] * 8 / 1e3:.0f} KB")
print(f" aligned CSR: runs {m[7]:.0f} x 128-B lines = {m[7] * 128 / 1e3:.0f} KB ({m[8]:.0f} x 64-B sectors = {m[8] * 64 / 1e3:.0f} KB), no indptr")
for q in (50, 90, 99, 99.9, 100):
    print(f"  row quantile {q:5}: pushes {np.percentile(ro[:, 0], q):7.0f} edges {np.percentile(ro[:, 1], q):8.0f} max level pushers {np.percentile(ro[:, 2], q):6.0f} max level edges {np.percentile(ro[:, 3], q):7.0f}")
