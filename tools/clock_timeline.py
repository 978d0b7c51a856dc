import sys, time, os
sys.path.insert(0, os.getcwd())
t00 = time.time()
import numpy as np, torch
from grand_plus_amd import Graph, _native
from grand_plus_amd.recipes import RECIPES
import bench, device_probe
print('import done', round(time.time()-t00,1), flush=True)
source, rkey, _ = bench.WORKLOADS['pubmed']
ip, ix = bench.load_graph(source, 8)
r = RECIPES[rkey]
g = Graph(ip, ix, 0)
seeds = torch.from_numpy(bench.make_seeds(source, len(ip)-1, 65536).astype(np.int32)).cuda()
print('graph up', round(time.time()-t00,1), 'mhz', device_probe.shader_clock_mhz(0), flush=True)
t0 = time.time()
while time.time() - t0 < float(sys.argv[1]):
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record(); g.gfpush_device(seeds, r.coef(), r.rmax, r.top_k); b.record(); torch.cuda.synchronize()
    alu, gbs = device_probe.speed_probe(0)
    print(round(time.time()-t00,1), 'ms', round(a.elapsed_time(b),2), 'mhz', round(device_probe.shader_clock_mhz(0)), 'alu_iters_per_us', round(alu,1), 'copy_gb_s', round(gbs), flush=True)
    time.sleep(float(sys.argv[2]))
