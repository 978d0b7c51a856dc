#!/bin/bash
# Calibrates the gfx950 read counter on the GFPush kernel's access shapes: tools/fetch_calib.sh [out_dir]
# One rocprofv3 --pmc pass per counter group (never combined with a trace option); the program itself follows "--".
OUT=${1:-gpurun_out/fetch_calib}
export TMPDIR=/tmp
mkdir -p $OUT
BIN=$(dirname "$0")/micro/fetch_calib
SRC=$(dirname "$0")/micro/fetch_calib.hip
if [ ! -x "$BIN" ] || [ "$SRC" -nt "$BIN" ]; then hipcc --offload-arch=gfx950 -O3 -o "$BIN" "$SRC" || exit 1; fi     # never a checked-in binary
$BIN > $OUT/requested.txt
i=0
for grp in "TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum" "TCC_BUBBLE_sum TCC_EA0_RDREQ_DRAM_sum" "FETCH_SIZE" "TCC_HIT_sum TCC_MISS_sum"; do
  i=$((i+1))
  timeout -k 5 120 rocprofv3 --pmc $grp --output-format csv -d $OUT/g$i -- $BIN > $OUT/g$i.log 2>&1
  echo "group $i ($grp): rc=$?"
done
python3 - <<PY
import csv, glob, json, os
from collections import defaultdict
cnt = defaultdict(dict)
for f in sorted(glob.glob(os.path.join("$OUT", "**", "*counter_collection.csv"), recursive=True)):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"].split("(")[0]
        disp = int(r["Dispatch_Id"])
        cnt[(k, disp)][r["Counter_Name"]] = cnt[(k, disp)].get(r["Counter_Name"], 0.0) + float(r["Counter_Value"])
req = [l.split() for l in open(os.path.join("$OUT", "requested.txt")) if " requested " in l]
# dispatches in launch order within each pass: gather4, runs4 x3, soa12, stream16
by_name = defaultdict(list)
for (k, disp), c in sorted(cnt.items(), key=lambda x: x[0][1]):
    by_name[k].append(c)
out = {}
names = [("gather4", "gather4", 0), ("runs4_4", "runs4", 0), ("runs4_14", "runs4", 1), ("runs4_64", "runs4", 2), ("soa12", "soa12", 0), ("stream16", "stream16", 0)]
for (label, kern, idx), r in zip(names, req):
    cs = {}
    for kname, lst in by_name.items():
        if kern in kname:
            # the same kernel appears once per pass; merge the idx-th launch of every pass
            per_pass = defaultdict(list)
            for c in lst:
                per_pass[tuple(sorted(c))].append(c)
            for key, launches in per_pass.items():
                if idx < len(launches):
                    cs.update(launches[idx])
    requested, sector = int(r[2]), int(r[4])
    rd = cs.get("TCC_EA0_RDREQ_sum", 0.0)
    out[label] = {"requested_bytes": requested, "bytes_if_64B_sectors": sector, "TCC_EA0_RDREQ": rd, "RDREQ_32B": cs.get("TCC_EA0_RDREQ_32B_sum"),
                  "TCC_BUBBLE": cs.get("TCC_BUBBLE_sum"), "RDREQ_DRAM": cs.get("TCC_EA0_RDREQ_DRAM_sum"), "FETCH_SIZE_KB": cs.get("FETCH_SIZE"),
                  "rdreq_x64_over_requested": round(rd * 64 / requested, 4) if rd else None,
                  "requested_over_rdreq_x64": round(requested / (rd * 64), 4) if rd else None,
                  "bytes_per_request": round(requested / rd, 2) if rd else None,
                  "l2_hit_rate": round(cs["TCC_HIT_sum"] / (cs["TCC_HIT_sum"] + cs["TCC_MISS_sum"]), 4) if cs.get("TCC_MISS_sum") else None}
    print(label, json.dumps(out[label]))
json.dump(out, open(os.path.join("$OUT", "fetch_calib.json"), "w"), indent=1)
PY
