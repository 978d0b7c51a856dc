set -x
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out/prof
cd /tmp && export TMPDIR=/tmp
ARGS="$R/bench.py --workload ${WL:-mag} --steps 3 --warmup 1 --seeds-per-gpu 8192 --no-cpu-baseline"
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof/stats -- python3 $ARGS > $R/gpurun_out/prof/stats.log 2>&1
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAVES --output-format csv -d $R/gpurun_out/prof/pmc1 -- python3 $ARGS > $R/gpurun_out/prof/pmc1.log 2>&1
rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SMEM SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM --output-format csv -d $R/gpurun_out/prof/pmc2 -- python3 $ARGS > $R/gpurun_out/prof/pmc2.log 2>&1
rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_INST_CYCLES_VMEM --output-format csv -d $R/gpurun_out/prof/pmc3 -- python3 $ARGS > $R/gpurun_out/prof/pmc3.log 2>&1
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $R/gpurun_out/prof/pmc4 -- python3 $ARGS > $R/gpurun_out/prof/pmc4.log 2>&1
rocprofv3 --pmc WRITE_SIZE TCC_HIT_sum TCC_MISS_sum --output-format csv -d $R/gpurun_out/prof/pmc5 -- python3 $ARGS > $R/gpurun_out/prof/pmc5.log 2>&1
cd $R/gpurun_out/prof; find . -name "*.csv" | head -30; tail -2 *.log | cut -c1-300
