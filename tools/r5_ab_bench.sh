#!/bin/bash
# Same-box A/B of the whole decode bench: the in-tree library against an experiment build (tools/exp_build.sh NAME ...),
# alternating processes:  tools/r5_ab_bench.sh NAME [rounds] [outdir]
name=$1; rounds=${2:-2}; out=${3:-gpurun_out/r5_ab_$name}; mkdir -p "$out"
for r in $(seq 1 $rounds); do
  for v in base $name; do
    if [ $v = base ]; then unset ONEDC_HIP_F16_SO; else export ONEDC_HIP_F16_SO="$(pwd)/tools/exp/libonedc_hip_$v.so"; fi
    timeout 400 python3 bench.py --no-cpu-baseline --no-extras 2> /dev/null | python3 -c "import sys, json; d = json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$v', d['value'], 'images/s', d['ms_per_step'], 'ms/step')" >> "$out/ab.txt"
  done
done
unset ONEDC_HIP_F16_SO
cat "$out/ab.txt"
