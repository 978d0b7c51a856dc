timeout 900 python -m pytest tests -m gpu -x -q 2>&1 | tail -3
run() { GRANDPLUS_DIAG=$1 timeout 900 python bench.py --workload $2 --steps 3 --warmup 1 --seeds-per-gpu 8192 --no-cpu-baseline 2>&1 | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.readline()); print('$2 diag=$1', d['value'], 'rows/s', d['detail'].get('diag_phase_share'), d['detail'].get('diag_topk_sub_share'), d['detail'].get('diag_counts_per_row'))"; }
for w in mag reddit pubmed cora; do run 0 $w; run 1 $w; done
R=$GRAFT_REPO_ROOT
rm -rf $R/gpurun_out/prof; mkdir -p $R/gpurun_out/prof
cd /tmp && export TMPDIR=/tmp
ARGS="$R/bench.py --workload mag --steps 3 --warmup 1 --seeds-per-gpu 8192 --no-cpu-baseline"
rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_ACTIVE_INST_ANY --output-format csv -d $R/gpurun_out/prof/pmc2 -- python3 $ARGS > $R/gpurun_out/prof/pmc2.log 2>&1
cd $R
python3 - <<'PY'
import csv, glob, collections
for d in ["pmc2"]:
    f=sorted(glob.glob(f"gpurun_out/prof/{d}/*/*_counter_collection.csv"))[-1]
    agg=collections.defaultdict(list)
    for r in csv.DictReader(open(f)):
        if "gfpush_kernel" in r["Kernel_Name"]: agg[r["Counter_Name"]].append(float(r["Counter_Value"]))
    for k,v in agg.items(): print(d,k,"per-row=",round(sum(v[-3:])/3/8192))
PY
