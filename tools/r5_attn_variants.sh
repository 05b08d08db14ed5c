#!/bin/bash
# flash_attn64 with 48 / 32 queries per wave and three workgroups per CU against the product (64 queries, two per CU): tools/bench_attn.py
# per build, two alternating rounds.  Builds: tools/exp_build.sh qt3o3 -DONEDC_ACT_F16 -DFA64_QT=3 -DFA64_OCC=3 (etc.)
out=${1:-gpurun_out/r5_attn}; mkdir -p "$out"
for r in 1 2; do
  for v in base qt3o3 qt2o3 qt3o2; do
    if [ $v = base ]; then unset ONEDC_HIP_F16_SO; else export ONEDC_HIP_F16_SO="$(pwd)/tools/exp/libonedc_hip_$v.so"; fi
    echo "== $v (round $r)" >> "$out/attn.txt"
    timeout 200 python3 tools/bench_attn.py 2> /dev/null | head -2 >> "$out/attn.txt"
  done
done
unset ONEDC_HIP_F16_SO
cat "$out/attn.txt"
