import sys, time, numpy as np, torch
sys.path.insert(0, '.')
from grand_plus_amd import Graph, synth
from grand_plus_amd.recipes import make_coef
indptr, indices = synth.shape_csr(sys.argv[1] if len(sys.argv) > 1 else "mag")
n = len(indptr) - 1
g = Graph(indptr, indices, 0)
S = 16384
seeds = torch.from_numpy(synth.seeds(n, S)).cuda()
def run(name, coef, rmax, K=32):
    for _ in range(2):
        g.reset_stats()
        g.gfpush_device(seeds, coef, rmax, K)
        st = g.stats()
    per_row_us = st["kernel_ms"] * 1e3 * st["workgroups"] / S
    print(f"{name:28s} kernel {st['kernel_ms']:.3f} ms  per-row-per-CU {per_row_us:7.1f} us  levels/row {(st['lds_levels']+st['global_levels'])/S:.2f} edges/row {st['edges']/S:.0f} frontier/row {st['frontier']/S:.0f} rows/s {S/st['kernel_ms']*1e3:.0f}")
run("no push (rmax=2)", make_coef("ppr", 10, 0.2), 2.0)
run("L=0 (coef=[1])", np.array([1.0]), 1e-5)
run("L=1", make_coef("ppr", 1, 0.2), 1e-5)
run("L=2", make_coef("ppr", 2, 0.2), 1e-5)
run("L=3", make_coef("ppr", 3, 0.2), 1e-5)
run("L=4", make_coef("ppr", 4, 0.2), 1e-5)
run("L=6", make_coef("ppr", 6, 0.2), 1e-5)
run("L=10", make_coef("ppr", 10, 0.2), 1e-5)
run("L=10 rmax=1e-4", make_coef("ppr", 10, 0.2), 1e-4)
run("L=10 rmax=1e-3", make_coef("ppr", 10, 0.2), 1e-3)
run("L=10 rmax=1e-2", make_coef("ppr", 10, 0.2), 1e-2)
