#!/bin/bash
# Collects the PMC counters behind profiles/*_pmc_summary.json: one rocprofv3 --pmc pass per counter
# group (never combined with a trace option), on `python3 bench.py` itself (no wrapper process).
#   tools/collect_pmc.sh [workload] [out_dir] [extra bench.py arguments]      then: python tools/pmc_summary.py <out_dir> <json>
#   (--opt measure_choice=0: the launches that time the candidates of the measured choice stay out of the counters; the thresholds
#    pick the same kernel as the measurement on all five workloads)
#   PMC_GROUPS=traffic collects the four memory-side groups only (the other workloads beside the headline one)
W=${1:-mag}; OUT=${2:-gpurun_out/pmc}; shift; shift
export TMPDIR=/tmp
mkdir -p $OUT
i=0
# FETCH_SIZE and WRITE_SIZE do not fit one pass ("exceeds the capabilities of the hardware"), and a failed
# rocprofv3 can hang while finalizing: every pass runs under its own timeout.
PMC_SETS=("FETCH_SIZE" "WRITE_SIZE" "TCC_EA0_RDREQ_sum TCC_EA0_WRREQ_sum" "TCC_HIT_sum TCC_MISS_sum" \
        "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS")
[ "$PMC_GROUPS" != "traffic" ] && PMC_SETS+=("SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SMEM" \
        "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_ACTIVE_INST_ANY" "SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA" \
        "SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS")
for grp in "${PMC_SETS[@]}"; do
  i=$((i+1))
  timeout -k 5 100 rocprofv3 --pmc $grp --output-format csv -d $OUT/g$i -- python3 bench.py --workload $W --prewarm 0 --steps 5 --warmup 2 --no-cpu-baseline --no-host-api --no-next-rows --no-cold-call --opt measure_choice=0 "$@" > $OUT/g$i.log 2>&1
  echo "group $i ($grp): rc=$?"
done
