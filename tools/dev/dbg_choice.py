import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np, torch, bench
from grand_plus_amd import Graph
from grand_plus_amd.recipes import RECIPES
name = sys.argv[1] if len(sys.argv) > 1 else "mag"
S = int(sys.argv[2]) if len(sys.argv) > 2 else 65536
source, rkey, _ = bench.WORKLOADS[name]
ip, ix = bench.load_graph(source, os.cpu_count() or 8)
r = RECIPES[rkey]
seeds = torch.from_numpy(bench.make_seeds(source, len(ip) - 1, S).astype(np.int32)).cuda()
g = Graph(ip, ix, 0)
for i in range(3):
    g.reset_stats(); g.gfpush_device(seeds, r.coef(), r.rmax, r.top_k); torch.cuda.synchronize()
    st = g.stats()
    print(name, "call", i, "kernel", st["kernel"], st["block_threads"], "ms", round(st["kernel_ms"], 3), "choice_ms", [round(x, 3) for x in st["choice_ms"]], flush=True)
