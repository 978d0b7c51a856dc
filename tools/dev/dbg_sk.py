#!/usr/bin/env python3
"""Debug aid: the failing sketch-kernel case against the oracle, row by row."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
from grand_plus_amd import synth, Graph
from grand_plus_amd.recipes import make_coef
from oracle import pyoracle
indptr, indices = synth.shape_csr("small")
seeds = synth.seeds(len(indptr) - 1, 1024)
coef = make_coef("ppr", 10, 0.2); rmax = 1e-4; K = 64
deg = np.diff(indptr)
exp = pyoracle.gfpush(indptr, indices, seeds, coef, rmax, K)
for block in (768,):
    for extra in ({}, {"sk_direct_max": 64}, {"sk_direct_max": 300}, {"sk_direct_max": 700}, {"sk_direct_max": 1200}, {"sk_direct_max": 1200, "solo_levels": 0}, {"max_workgroups": 8}, {"max_workgroups": 1}):
        g = Graph(indptr, indices, 0)
        g.set_option("kernel", 2); g.set_option("sk_block_threads", block)
        for k, v in extra.items(): g.set_option(k, v)
        S = len(seeds)
        row = np.zeros(S * K, np.int32); col = np.zeros(S * K, np.int32); val = np.zeros(S * K)
        g.gfpush_omp(seeds.astype(np.int64), row, col, val, coef, rmax, K)
        st = g.stats()
        bad = []
        for i in range(S):
            a = dict(zip(col[i*K:(i+1)*K].tolist(), val[i*K:(i+1)*K].tolist()))
            b = dict(zip(exp[1][i*K:(i+1)*K].tolist(), exp[2][i*K:(i+1)*K].tolist()))
            d = [(c, a[c], b[c]) for c in a if c in b and abs(a[c] - b[c]) > 1e-9 * b[c]]
            if d: bad.append((i, int(seeds[i]), int(deg[seeds[i]]), d[:3], len(d)))
        print(block, extra, "bad rows", len(bad), "pushes", st["pushes"], exp[3]["pushes"], "edges", st["edges"], exp[3]["edges"], "retried", st["retried_rows"], st["diag_sub"][1:7])
        for b_ in bad[:6]: print("   ", b_)
        g.close()

