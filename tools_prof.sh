R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out/prof2
cd /tmp && export TMPDIR=/tmp
ARGS="$R/bench.py --steps 5 --warmup 2 --no-cpu-baseline"
python3 $ARGS 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.readline()); print(d['value'], d['detail'])"
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $R/gpurun_out/prof2/pmc_fetch -- python3 $ARGS > /dev/null 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $R/gpurun_out/prof2/pmc_write -- python3 $ARGS > /dev/null 2>&1
rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_ACTIVE_INST_ANY --output-format csv -d $R/gpurun_out/prof2/pmc_sq -- python3 $ARGS > /dev/null 2>&1
cd $R
python3 - <<'PY'
import csv, glob, collections
for d in ["pmc_fetch","pmc_write","pmc_sq"]:
    f=sorted(glob.glob(f"gpurun_out/prof2/{d}/*/*_counter_collection.csv"))[-1]
    agg=collections.defaultdict(list)
    for r in csv.DictReader(open(f)):
        if "gfpush_kernel" in r["Kernel_Name"]: agg[r["Counter_Name"]].append(float(r["Counter_Value"]))
    for k,v in agg.items(): print(d,k,"per-row=",round(sum(v[-5:])/5/16384,1))
PY
