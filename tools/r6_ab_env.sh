#!/bin/bash
# Same-box A/B of the whole decode bench over values of ONE environment switch, alternating processes:
#   tools/r6_ab_env.sh VAR "v1 v2 v3" [rounds] [outfile]     e.g.  tools/r6_ab_env.sh ONEDC_GN_FUSION_MAX_COUT "128 256 320" 3
var=$1; values=$2; rounds=${3:-3}; out=${4:-gpurun_out/r6_ab_$var.txt}; mkdir -p "$(dirname "$out")"
for r in $(seq 1 $rounds); do
  for v in $values; do
    env $var=$v timeout 400 python3 bench.py --no-cpu-baseline --no-extras 2> /dev/null | python3 -c "import sys, json; d = json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$var=$v', d['value'], 'images/s', d['ms_per_step'], 'ms/step', 'median of 3:', d.get('value_median_of_3'))" >> "$out"
  done
done
cat "$out"
