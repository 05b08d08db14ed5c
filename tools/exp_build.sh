#!/bin/bash
# Build an experiment variant of libonedc_hip.so with extra -D flags:  tools/exp_build.sh NAME -DEXP_FOO ...
# -> tools/exp/libonedc_hip_NAME.so ; run with ONEDC_HIP_SO=tools/exp/libonedc_hip_NAME.so
set -e
cd "$(dirname "$0")/.."
name=$1; shift
mkdir -p tools/exp
objs=""
for f in onedc_amd/csrc/*.hip; do
  o=tools/exp/$(basename ${f%.hip})_$name.o
  ff=""; case "$(basename $f)" in attention.hip|xattn_block.hip) ff="-fno-honor-nans";; esac
  case "$(basename $f)" in attention.hip|xattn_block.hip|gemm_conv*.hip) ff="$ff -fno-slp-vectorize";; esac   # as onedc_amd/build.py
  hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -c -Wno-unused-result $ff "$@" -o $o $f 2>/dev/null &
  objs="$objs $o"
done
wait
hipcc --offload-arch=gfx950 -shared -fPIC -o tools/exp/libonedc_hip_$name.so $objs
rm -f $objs
echo built tools/exp/libonedc_hip_$name.so
