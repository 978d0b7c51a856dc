#!/bin/bash
# The driver's own command on this lease, kept as it prints (quote what the driver will see):
#   tools/driver_style.sh <tag>      -> gpurun_out/<tag>/driver_style.json  (one JSON line; run it on three leases, commit min / median / max)
TAG=${1:-driver_style}
mkdir -p gpurun_out/$TAG
python bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/$TAG/driver_style.json 2> gpurun_out/$TAG/driver_style.err
python - <<PY
import json
d = json.loads(open("gpurun_out/$TAG/driver_style.json").read().strip().splitlines()[-1])
print({k: d[k] for k in ("value", "ms_per_step")}, d["roofline"]["frac"], d["roofline"]["kernel_ms_avg"], d.get("cold_call"), d["host_api"]["rows_per_s"], d["detail"]["csr_upload_s"])
PY
