#!/usr/bin/env python3
"""Per-level time and work of the GFPush kernel + the share of wave time spent at workgroup barriers
(diagnostic build only: `GRANDPLUS_DIAG=1 python tools/level_breakdown.py [workload ...]`)."""
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
        sys.exit("set GRANDPLUS_DIAG=1")
    names = sys.argv[1:] or ["mag"]
    for name in names:
        source, rkey, _ = bench.WORKLOADS[name]
        ip, ix = bench.load_graph(source, os.cpu_count() or 8)
        r = RECIPES[rkey]
        S = 4096 if name == "amazon2m" else 16384
        seeds = torch.from_numpy(bench.make_seeds(source, len(ip) - 1, S).astype(np.int32)).cuda()
        g = Graph(ip, ix, 0)
        for o in os.environ.get("GP_OPTS", "").split(","):
            if o:
                k, v = o.split("="); g.set_option(k, int(v))
        for _ in range(2):
            g.reset_stats(); g.gfpush_device(seeds, r.coef(), r.rmax, r.top_k); torch.cuda.synchronize()
        st = g.stats(); dx = g.diag_counters(); rows = st["rows"]; tot = st["diag_ticks_total"]
        print(f"{name}: kernel {st['kernel_ms']:.2f} ms, {tot / rows / 100:.1f} us/row/wg; scan {st['diag_ticks_scan'] / tot:.3f} "
              f"expand {st['diag_ticks_expand'] / tot:.3f} topk {st['diag_ticks_topk'] / tot:.3f}; "
              f"barrier wait {dx[1] / max(dx[0], 1):.3f} of wave cycles, {dx[2] / 16 / rows:.1f} barriers/row "
              f"({dx[1] / max(dx[2], 1):.0f} cyc waited per wave per barrier); wave cycles/row {dx[0] / 16 / rows:.0f}", flush=True)
        print(f"   per row (us): prologue {dx[4] / rows / 100:.2f}  level0 {dx[5] / rows / 100:.2f}  level-loop outside expand/scan {dx[6] / rows / 100:.2f}  "
              f"table restore {dx[7] / rows / 100:.2f}  topk {st['diag_ticks_topk'] / rows / 100:.2f}", flush=True)
        print("   SCAN of wave 0 (cycles/row): compact (a,b) %.0f  records (c) %.0f  lookups+entries (d) %.0f  tail %.0f" % tuple(dx[8 + i] / rows for i in range(4)), flush=True)
        print("   EXPAND stream of wave 0 (cycles/row): prepare %.0f  wait for columns %.0f  inserts %.0f  total %.0f;  per row: steps %.1f  first-step chain (cycles) %.0f  calls %.1f" % tuple(dx[112 + i] / rows for i in (0, 1, 2, 5, 3, 4, 6)), flush=True)
        print("   EXPAND of wave 0 (cycles/row): call -> first instruction of edge_stream %.0f, its end -> behind the barrier %.0f" % (dx[122] / rows, dx[123] / rows), flush=True)
        print("   EXPAND calls: longest wave %.0f cycles/row, mean wave %.0f cycles/row (sum over waves / waves)" % (dx[120] / rows, dx[121] / rows / (st["block_threads"] / 64)), flush=True)
        names = ["agg_init", "agg_insert", "agg_scan", "sel_hist", "sel_pick", "sel_compact", "sel_collect", "final"]
        print("   topk sub-phases (us/row):", {n: round(st["diag_sub"][i] / rows / 100, 2) for i, n in enumerate(names)}, flush=True)
        if len(st["diag_sub"]) >= 13:
            print("   topk counts per row: full-aggregation table passes %.3f  select histogram passes over HBM %.3f / over LDS %.3f  rows finished by the pruned path %.3f"
                  % tuple(st["diag_sub"][i] / rows for i in (8, 9, 10, 11)), flush=True)
        if os.environ.get("GP_SITES"):
            # barrier sites in source order: GP_SYNC() occurrences after the macro definitions
            src = open(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "grand_plus_amd", "csrc", "gfpush_kernels.hpp")).read().split("\n")
            lines = [i + 1 for i, l in enumerate(src) if "GP_SYNC();" in l and "#define" not in l]
            tot_w = sum(dx[128:192])
            print("   barrier sites (line: waited us/row/wave, arrivals/row/wave, share of all barrier wait):")
            for site in range(64):
                if dx[192 + site]:
                    print(f"     {lines[site] if site < len(lines) else '?':>5}: {dx[128 + site] / rows / 8 / 2400:7.2f} us  {dx[192 + site] / rows / 8:6.2f}  {dx[128 + site] / max(tot_w, 1):.3f}   {src[lines[site] - 1].strip()[:60] if site < len(lines) else ''}")
        for lvl in range(1, 16):
            e, s, ed, nd, pe, ps = dx[16 + 6 * lvl:16 + 6 * lvl + 6]
            if ps == 0:
                continue
            print(f"   level {lvl:2d}: expand {e / rows / 100:6.2f} us  scan {s / rows / 100:6.2f} us  edges {ed / rows:8.1f}  "
                  f"nodes {nd / rows:8.1f}  entries {pe / rows:7.1f}  passes {ps / rows:5.2f}", flush=True)


if __name__ == "__main__":
    main()
