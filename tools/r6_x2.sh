#!/bin/bash
# round 6: the GPU suite with the fused cross-attention block + GroupNorm on load for two-tile layers, then whole-bench A/Bs
cd "$(dirname "$0")/.."
python -c "from onedc_amd import kernels as K; K.lib(\"f16\")" || { echo "STALE LIBRARY"; exit 9; }
mkdir -p gpurun_out/r6f
export TMPDIR=/tmp
timeout 2400 python -m pytest tests -m gpu -x -q > gpurun_out/r6f/pytest_gpu.txt 2>&1
tail -12 gpurun_out/r6f/pytest_gpu.txt
bash tools/r6_ab_env.sh ONEDC_XATTN "0 1" 3 gpurun_out/r6f/ab_xattn.txt
