export TMPDIR=/tmp GRANDPLUS_DIAG=1
for f in 0 1 2; do
 timeout -k 5 100 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS --output-format csv -d gpurun_out/pmc_skip/f$f -- python3 bench.py --steps 5 --warmup 2 --no-cpu-baseline --diag-flags $f > gpurun_out/pmc_skip/f$f.log 2>&1
 python tools/pmc_summary.py gpurun_out/pmc_skip/f$f gpurun_out/pmc_skip/f$f.json | grep per_row
 tail -1 gpurun_out/pmc_skip/f$f.log | cut -c1-200
done
