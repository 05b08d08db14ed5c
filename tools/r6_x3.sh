#!/bin/bash
cd "$(dirname "$0")/.."
python -c "from onedc_amd import kernels as K; K.lib(\"f16\")" || { echo "STALE LIBRARY"; exit 9; }
mkdir -p gpurun_out/r6g
export TMPDIR=/tmp
timeout 1200 python -m pytest tests/test_gpu_parity.py -x -q -s -k "compact_attention or fused_cross" > gpurun_out/r6g/pytest_a.txt 2>&1
grep -a "PSNR(fused\|passed\|failed\|Error" gpurun_out/r6g/pytest_a.txt | tail -8
timeout 2400 python -m pytest tests/test_gpu_parity.py tests/test_gpu_output.py tests/test_ref_arith.py tests/test_ref_stream_portability.py -x -q > gpurun_out/r6g/pytest_b.txt 2>&1
tail -5 gpurun_out/r6g/pytest_b.txt
for a in "--s-streams 2" "--s-streams 3" "--s-streams 2 --free-vae" "--s-streams 2" "--s-streams 3"; do
  echo "bench.py $a: $(timeout 400 python3 bench.py --no-cpu-baseline --no-extras $a 2> /dev/null | python3 -c "import sys, json; d = json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], 'images/s', d['ms_per_step'], 'ms/step', 'median of 3:', d.get('value_median_of_3'))")" >> gpurun_out/r6g/ab_streams.txt
done
cat gpurun_out/r6g/ab_streams.txt
