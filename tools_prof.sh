R=$GRAFT_REPO_ROOT
rm -rf $R/gpurun_out/prof_r01; mkdir -p $R/gpurun_out/prof_r01
cd /tmp && export TMPDIR=/tmp
ARGS="$R/bench.py --steps 5 --warmup 2 --no-cpu-baseline"
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_r01/stats -- python3 $ARGS > $R/gpurun_out/prof_r01/stats_bench.json 2> $R/gpurun_out/prof_r01/stats.err
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $R/gpurun_out/prof_r01/pmc_fetch -- python3 $ARGS > /dev/null 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $R/gpurun_out/prof_r01/pmc_write -- python3 $ARGS > /dev/null 2>&1
rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_ACTIVE_INST_ANY --output-format csv -d $R/gpurun_out/prof_r01/pmc_sq -- python3 $ARGS > /dev/null 2>&1
rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_sum TCC_EA0_WRREQ_sum --output-format csv -d $R/gpurun_out/prof_r01/pmc_tcc -- python3 $ARGS > /dev/null 2>&1
rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS --output-format csv -d $R/gpurun_out/prof_r01/pmc_lds -- python3 $ARGS > /dev/null 2>&1
cd $R
python3 bench.py --steps 5 --warmup 2 > gpurun_out/prof_r01/bench_mag.json 2>/dev/null
for w in pubmed reddit cora; do python3 bench.py --workload $w --steps 5 --warmup 2 > gpurun_out/prof_r01/bench_$w.json 2>/dev/null; done
python3 bench.py --workload amazon2m --steps 3 --warmup 1 --seeds-per-gpu 4096 > gpurun_out/prof_r01/bench_amazon2m.json 2>gpurun_out/prof_r01/bench_amazon2m.err
for f in mag pubmed reddit cora amazon2m; do python3 -c "
import json
d=json.loads(open('gpurun_out/prof_r01/bench_$f.json').read().strip().splitlines()[-1]); cb=d.get('cpu_baseline',{})
print('$f', d['value'],'rows/s kernel_ms',d['roofline']['kernel_ms_avg'],'GB/s',d['roofline']['achieved'],'frac',d['roofline']['frac'],'cpu',cb.get('value'),cb.get('kind'),'port',cb.get('port_value'),'x',d['detail'].get('gpu_over_cpu'))"; done
