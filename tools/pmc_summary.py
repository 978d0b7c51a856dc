#!/usr/bin/env python3
"""Summarise the counter CSVs written by tools/collect_pmc.sh into one JSON: per-CALL means over the timed steps of the bench
command (which runs with --prewarm 0, so its gfpush launches are exactly (warmup + steps) calls; a call is 2 launches of the
general kernel or 3 with the sketch kernel in front -- all of a call's launches are summed, the first `--warmup` calls dropped).
Usage: python tools/pmc_summary.py <dir> <out.json> [--warmup 2] [--steps 5] [--rows 65536]"""
import argparse, csv, glob, json, os, sys
from collections import defaultdict

ap = argparse.ArgumentParser(); ap.add_argument("dir"); ap.add_argument("out")
ap.add_argument("--warmup", type=int, default=2); ap.add_argument("--steps", type=int, default=5); ap.add_argument("--rows", type=int, default=65536)
ap.add_argument("--workload", default="mag")
a = ap.parse_args()
per = {}; n_avg = {}; kernels = set()
for f in sorted(glob.glob(os.path.join(a.dir, "**", "*counter_collection.csv"), recursive=True)):
    by = defaultdict(lambda: defaultdict(float))          # counter -> dispatch -> value
    first = {}                                            # dispatch -> does it start a call (the kernel that takes every row)
    for r in csv.DictReader(open(f)):
        if "gfpush_" not in r["Kernel_Name"] or "kernel<" not in r["Kernel_Name"]:
            continue
        kernels.add(r["Kernel_Name"].split("(")[0])
        d = int(r["Dispatch_Id"])
        by[r["Counter_Name"]][d] += float(r["Counter_Value"])
        first[d] = "gfpush_retry_kernel" not in r["Kernel_Name"]
    # a call = one launch of gfpush_kernel / gfpush_sk_kernel and the retry launches behind it; the timed calls are the LAST
    # `steps` calls of the process (in front of them: warm-up calls and, once per recipe, the short launches that time the
    # candidates of the measured choice)
    order = sorted(first)
    calls = []
    for d in order:
        if first[d]: calls.append([d])
        elif calls: calls[-1].append(d)
    if len(calls) < a.steps:
        print(f"{f}: {len(calls)} gfpush calls, fewer than the {a.steps} timed steps -- skipped", file=sys.stderr)
        continue
    timed = calls[-a.steps:]
    for c, dd in by.items():
        per[c] = sum(dd[d] for call in timed for d in call) / a.steps
        n_avg[c] = {"calls": a.steps, "launches_per_call": len(timed[-1]), "calls_in_process": len(calls)}
g = per.get
d = {}
if g("TCC_EA0_RDREQ_sum"): d["hbm_read_bytes_raw"] = g("TCC_EA0_RDREQ_sum") * 64
elif g("FETCH_SIZE"): d["hbm_read_bytes_raw"] = g("FETCH_SIZE") * 1024
# Calibrated (tools/fetch_calib.sh -> profiles/r03_fetch_calib.json): on gfx950 every TCC_EA0_RDREQ is the fill of one
# 128-byte L2 line -- streams of 16 B/lane and of a 4-B + 8-B SoA pair read 128.0 bytes per request, random runs of 4 / 14 /
# 64 words and single-word gathers produce exactly one request per 128-byte line they touch -- and FETCH_SIZE tallies it
# at 64 B (TCC_BUBBLE, its 128-byte term, stays 0).  So the bytes that cross the L2 -> fabric boundary are 2 x raw.
if "hbm_read_bytes_raw" in d: d["hbm_read_bytes_corrected"] = 2.0 * d["hbm_read_bytes_raw"]
if g("WRITE_SIZE"): d["hbm_write_bytes"] = g("WRITE_SIZE") * 1024
for k in ("hbm_read_bytes_corrected", "hbm_write_bytes"):
    if k in d: d[k.replace("bytes", "kb_per_row").replace("_corrected", "")] = d[k] / 1024 / a.rows
if "hbm_read_bytes_corrected" in d and "hbm_write_bytes" in d: d["hbm_traffic_bytes"] = d["hbm_read_bytes_corrected"] + d["hbm_write_bytes"]
if g("TCC_HIT_sum") and g("TCC_MISS_sum"): d["l2_hit_rate"] = g("TCC_HIT_sum") / (g("TCC_HIT_sum") + g("TCC_MISS_sum"))
for c, n in (("SQ_INSTS_VALU", "valu_per_row"), ("SQ_INSTS_SALU", "salu_per_row"), ("SQ_INSTS_LDS", "lds_insts_per_row"),
             ("SQ_INSTS_VMEM_RD", "vmem_rd_per_row"), ("SQ_INSTS_VMEM_WR", "vmem_wr_per_row")):
    if g(c): d[n] = g(c) / a.rows
if g("SQ_WAVE_CYCLES"):
    if g("SQ_WAIT_ANY"): d["wave_wait_fraction"] = g("SQ_WAIT_ANY") / g("SQ_WAVE_CYCLES")
    if g("SQ_ACTIVE_INST_ANY"): d["wave_active_fraction"] = g("SQ_ACTIVE_INST_ANY") / g("SQ_WAVE_CYCLES")
if g("SQ_LDS_BANK_CONFLICT") and g("SQ_LDS_IDX_ACTIVE"): d["lds_bank_conflict_fraction"] = g("SQ_LDS_BANK_CONFLICT") / g("SQ_LDS_IDX_ACTIVE")
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench
json.dump({"workload": a.workload, "seeds_per_gpu": a.rows, "kernel_sha16": bench.kernel_source_sha16(),
           "command": "tools/collect_pmc.sh: rocprofv3 --pmc <group> --output-format csv -- python3 bench.py --prewarm 0 --steps 5 --warmup 2 --no-cpu-baseline --no-host-api --no-next-rows (one pass per group; per-call mean over the 5 timed calls, all launches of a call summed)",
           "kernels": sorted(kernels), "averaged_over": n_avg, "per_launch": per, "derived": d,
           "note": "hbm_read_bytes_raw = TCC_EA0_RDREQ x 64 B (= FETCH_SIZE); hbm_read_bytes_corrected = TCC_EA0_RDREQ x 128 B: tools/fetch_calib.sh (profiles/r03_fetch_calib.json) shows one request per 128-byte L2 line for every access shape of this kernel (16-B/lane and SoA streams, runs of 4 / 14 / 64 words, single-word gathers). WRITE_SIZE is taken as it reads (exact for 16-B/lane streaming stores per MI355X_MICROARCH.md). Infinity-Cache hits are included in both. SQ_* cycle counters are in quad-cycles; SQ_BUSY_CYCLES is summed over the 32 shader engines."},
          open(a.out, "w"), indent=1)
print(json.dumps(d, indent=1))
