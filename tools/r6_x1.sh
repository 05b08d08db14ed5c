#!/bin/bash
cd "$(dirname "$0")/.."
python -c "from onedc_amd import kernels as K; K.lib(\"f16\")" || { echo "STALE LIBRARY"; exit 9; }
mkdir -p gpurun_out/r6e
timeout 600 python -m pytest tests/test_gpu_kernels.py -x -q -k "cross_attention_block" > gpurun_out/r6e/test_xattn.txt 2>&1
tail -25 gpurun_out/r6e/test_xattn.txt
timeout 300 python tools/bench_xattn.py > gpurun_out/r6e/bench_xattn.txt 2>&1
cat gpurun_out/r6e/bench_xattn.txt
ONEDC_HIP_F16_SO=tools/exp/libonedc_hip_xatl.so timeout 300 python tools/timeline_xattn.py > gpurun_out/r6e/timeline_xattn.txt 2>&1
cat gpurun_out/r6e/timeline_xattn.txt
