#!/usr/bin/env python3
"""Per-phase time of the GFPush kernel from the diagnostic build (in-kernel 100 MHz phase stamps).

Run as `GRANDPLUS_DIAG=1 python tools/phase_breakdown.py [workload ...]`: selects
libgrandplus_diag.so (same sources, -DGP_DIAG) and prints, per bench workload, the share of SCAN /
EXPAND / TOP-K in the workgroups' time plus per-row work counts.  Diagnostic only -- the stamps cost
~7 % -- never used by bench.py or the tests.
"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402
import torch  # noqa: E402
import bench  # noqa: E402
from grand_plus_amd import Graph  # noqa: E402
from grand_plus_amd.recipes import RECIPES  # noqa: E402


def main():
    if os.environ.get("GRANDPLUS_DIAG") != "1":
        sys.exit("set GRANDPLUS_DIAG=1 (the product library carries no phase stamps)")
    names = sys.argv[1:] or ["mag", "reddit", "pubmed", "cora", "amazon2m"]
    for name in names:
        source, rkey, _ = bench.WORKLOADS[name]
        ip, ix = bench.load_graph(source, os.cpu_count() or 8)
        r = RECIPES[rkey]
        S = 4096 if name == "amazon2m" else 16384
        seeds = torch.from_numpy(bench.make_seeds(source, len(ip) - 1, S).astype(np.int32)).cuda()
        g = Graph(ip, ix, 0)
        for _ in range(3):
            g.reset_stats(); g.gfpush_device(seeds, r.coef(), r.rmax, r.top_k); torch.cuda.synchronize()
        st = g.stats(); tot = st["diag_ticks_total"]; rows = st["rows"]
        print(f"{name}: kernel {st['kernel_ms']:.2f} ms, {tot / rows / 100:.1f} us/row/workgroup; "
              f"scan {st['diag_ticks_scan'] / tot:.3f} expand {st['diag_ticks_expand'] / tot:.3f} "
              f"topk {st['diag_ticks_topk'] / tot:.3f}; per row: levels {st['lds_levels'] / rows:.1f} "
              f"frontier {st['frontier'] / rows:.0f} pushes {st['pushes'] / rows:.0f} edges {st['edges'] / rows:.0f}", flush=True)
        print("   topk sub-phases (share):", [round(x / tot, 4) for x in st["diag_sub"][:8]], "counts:", st["diag_sub"][8:12], flush=True)


if __name__ == "__main__":
    main()
