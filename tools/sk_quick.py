#!/usr/bin/env python3
"""Quick A/B of the general kernel against the sketch kernel on one workload: best kernel_ms of a few launches of 65 536 rows.
Usage: python tools/sk_quick.py [workload] [rows] [key=value ...]  (extra options are applied to the sketch runs)"""
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
variants = [("general", {"kernel": 1}), ("sketch 768", {"kernel": 2, "sk_block_threads": 768}), ("sketch 512", {"kernel": 2, "sk_block_threads": 512})]
if os.environ.get("SKQ_ONLY"):
    variants = [v for v in variants if v[0] in os.environ["SKQ_ONLY"].split(",")]
for tag, opts in variants:
    for k, v in opts.items():
        g.set_option(k, int(v))
    if opts["kernel"] == 2:
        for k, v in extra.items():
            g.set_option(k, int(v))
    ms = []
    for _ in range(6):
        g.reset_stats(); g.gfpush_device(seeds, r.coef(), r.rmax, r.top_k); torch.cuda.synchronize()
        st = g.stats(); ms.append(st["kernel_ms"])
    print(f"{name} {tag:11s} {extra if opts['kernel'] == 2 else ''}: kernel {st['kernel']} {st['block_threads']}x{st['lds_bytes']} wgs {st['workgroups']} best {min(ms):8.3f} ms median {sorted(ms)[len(ms)//2]:8.3f} -> {S / min(ms) / 1e3:7.3f} M rows/s; "
          f"retried {st['retried_rows']} cand_edges {st['sketch_candidate_edges'] / max(st['edges'], 1):.3f} sweeps2 {st['sketch_second_sweeps']} pushes {st['pushes']} edges {st['edges']} filled {st['filled']} ws {st['workspace_bytes'] / 2**30:.2f} GB handed back why {st['diag_sub'][1:6]}", flush=True)
# the metric's own clock: host buffers in -> host buffers out (gp_gfpush), last variant's options
import time
hs = seeds.cpu().numpy().astype(np.int64); K = r.top_k
row = np.zeros(S * K, np.int32); col = np.zeros(S * K, np.int32); val = np.zeros(S * K)
ts = []
for _ in range(5):
    t = time.perf_counter(); g.gfpush_omp(hs, row, col, val, r.coef(), r.rmax, K); ts.append((time.perf_counter() - t) * 1e3)
st = g.stats()
print(f"{name} host API ({tag}): calls {[round(x, 2) for x in ts]} ms, kernel of the last {st['kernel_ms']:.3f} ms -> {S / sorted(ts[1:])[1] / 1e3:.3f} M rows/s", flush=True)
g.close()
