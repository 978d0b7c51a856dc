export GRANDPLUS_SYNTH_CACHE=/dev/shm/gp_synth SKQ_ONLY="sketch 768" GRANDPLUS_LIB=libgrandplus.so.t1
mkdir -p gpurun_out; : > gpurun_out/sk_ab.txt
for r in 1 2 3; do for o in "" "sk_lg_mu=12" "sk_lg_mr=10" "sk_lg_mu=12 sk_lg_mr=10" "sk_target=256"; do
  echo -n "[$o] " >> gpurun_out/sk_ab.txt
  timeout 300 python tools/sk_quick.py mag 65536 $o 2>&1 | grep "sketch 768 " | cut -c1-150 >> gpurun_out/sk_ab.txt
done; done; cat gpurun_out/sk_ab.txt
