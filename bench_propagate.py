#!/usr/bin/env python3
"""bench_propagate.py -- roofline measurement of gp_propagate_features, the exact
full-graph propagation of the reference's predict() (model.py:186-210).

Bound: HBM gather.  Algorithmic bytes per step = 4*F*nnz (one feature row per stored edge) + 4*nnz
(column ids) + 3*4*N*F (write X_next, read-modify-write the running sum).  CPU column: the scipy/numpy
float64 restatement (oracle/predict_ref.py = the reference's own calls) on this box, one process
(scipy's CSR x dense product is single-threaded), on the same graph with fewer steps, scaled per step.
"""
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

import numpy as np  # noqa: E402
import torch  # noqa: E402

from grand_plus_amd import Graph, synth  # noqa: E402

CASES = [  # name, graph source, F, mode, order, alpha, cpu_steps
    ("pubmed graph, F=500, ppr order 6", "golden:pubmed", 500, "ppr", 6, 0.5, 6),
    ("reddit-shape, F=602, ppr order 6", "synth:reddit", 602, "ppr", 6, 0.05, 1),
    ("amazon2m-shape, F=100, ppr order 6", "synth:amazon2m", 100, "ppr", 6, 0.2, 1),
    ("mag-shape, F=64 (embedded), ppr order 10", "synth:mag", 64, "ppr", 10, 0.2, 0),
]


def main():
    for name, src, F, mode, order, alpha, cpu_steps in CASES:
        kind, gname = src.split(":")
        if kind == "synth":
            indptr, indices = synth.shape_csr(gname)
        else:
            z = np.load(os.path.join(ROOT, "tests", "golden", f"{gname}.npz")); indptr, indices = z["indptr"], z["indices"]
        n, nnz = len(indptr) - 1, len(indices)
        g = Graph(indptr, indices, 0)
        X = torch.randn((n, F), device="cuda")
        out = torch.empty_like(X)
        for _ in range(2):
            g.propagate_features(X, mode, order, alpha, out=out)
        torch.cuda.synchronize()
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        iters = 3
        a.record()
        for _ in range(iters):
            g.propagate_features(X, mode, order, alpha, out=out)
        b.record(); torch.cuda.synchronize()
        ms = a.elapsed_time(b) / iters
        bytes_step = 4 * F * nnz + 4 * nnz + 12 * n * F
        gbps = bytes_step * order / ms / 1e6
        line = {"case": name, "n_nodes": n, "nnz": nnz, "feat_dim": F, "steps": order, "ms_total": round(ms, 3),
                "ms_per_step": round(ms / order, 3), "algorithmic_GB_per_step": round(bytes_step / 1e9, 3),
                "achieved_GBps": round(gbps, 1), "frac_of_8TBps": round(gbps / 8000, 4)}
        if cpu_steps:
            import scipy.sparse as sp
            from oracle.predict_ref import propagate_ref
            adj = sp.csr_matrix((np.ones(nnz), indices, indptr), shape=(n, n))
            Xh = X.cpu().numpy()
            t = time.perf_counter(); propagate_ref(adj, Xh, mode, cpu_steps, alpha); dt = time.perf_counter() - t
            line["cpu_scipy_ms_per_step"] = round(dt / cpu_steps * 1e3, 1)
            line["gpu_over_cpu"] = round(dt / cpu_steps * 1e3 / (ms / order), 1)
        print(json.dumps(line), flush=True)
        g.close(); del X, out


if __name__ == "__main__":
    main()
