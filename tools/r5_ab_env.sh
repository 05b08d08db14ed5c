#!/bin/bash
# Same-box A/B of the whole decode bench by an environment switch, alternating processes:
#   tools/r5_ab_env.sh ONEDC_COMPACT_HEADS 0 1 [rounds] [outdir]
var=$1; a=$2; b=$3; rounds=${4:-3}; out=${5:-gpurun_out/r5_ab_$var}; mkdir -p "$out"
for r in $(seq 1 $rounds); do
  for v in $a $b; do
    env $var=$v timeout 400 python3 bench.py --no-cpu-baseline --no-extras 2> /dev/null | python3 -c "import sys, json; d = json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$var=$v', d['value'], 'images/s', d['ms_per_step'], 'ms/step')" >> "$out/ab.txt"
  done
done
cat "$out/ab.txt"
