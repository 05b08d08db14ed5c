#!/bin/bash
# round 6, first GPU call: the bench line on this round's box + the epilogue timeline of the 4-wave window kernel
cd "$(dirname "$0")/.."
mkdir -p gpurun_out/r6a
export TMPDIR=/tmp
timeout 900 python bench.py > gpurun_out/r6a/bench.json 2> gpurun_out/r6a/bench.err
echo "bench rc=$?"
for v in "" "--res" "--gn --bf" "--gn --bf --res"; do
  echo "=== 16 768 768 128 128 $v" >> gpurun_out/r6a/timeline.txt
  ONEDC_HIP_F16_SO=tools/exp/libonedc_hip_w4etl.so timeout 300 python tools/timeline_win4.py 16 768 768 128 128 --epilogue $v >> gpurun_out/r6a/timeline.txt 2>&1
done
echo "=== 16 192 192 512 512 --res" >> gpurun_out/r6a/timeline.txt
ONEDC_HIP_F16_SO=tools/exp/libonedc_hip_w4etl.so timeout 300 python tools/timeline_win4.py 16 192 192 512 512 --epilogue --res >> gpurun_out/r6a/timeline.txt 2>&1
tail -5 gpurun_out/r6a/bench.err
cut -c1-600 gpurun_out/r6a/bench.json
