"""Per-row fixed cost of the GFPush kernel: the MAG-shape launch with rmax large enough that no node pushes beyond the
first levels, against the real recipe.  usage: python tools/fixed_cost.py [workload]"""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from grand_plus_amd import Graph
from grand_plus_amd.recipes import RECIPES

name = sys.argv[1] if len(sys.argv) > 1 else "mag"
source, rkey, _ = bench.WORKLOADS[name]
ip, ix = bench.load_graph(source, 8)
r = RECIPES[rkey]
S = 65536
seeds = torch.from_numpy(bench.make_seeds(source, len(ip) - 1, S).astype(np.int32)).cuda()
g = Graph(ip, ix, 0)
for rmax in (r.rmax, 1e-3, 1e-2, 1.0, 10.0):
    for _ in range(2):
        g.reset_stats(); g.gfpush_device(seeds, r.coef(), rmax, r.top_k); torch.cuda.synchronize()
    st = g.stats()
    print(f"{name} rmax {rmax:g}: kernel {st['kernel_ms']:.3f} ms per {S} rows = {st['kernel_ms'] * 1e3 * st['workgroups'] / S:.2f} us per row and workgroup; "
          f"pushes/row {st['pushes'] / S:.1f} edges/row {st['edges'] / S:.1f} frontier/row {st['frontier'] / S:.1f}", flush=True)
