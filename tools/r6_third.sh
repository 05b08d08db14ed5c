#!/bin/bash
# round 6, third GPU call: LDS-read mix vs clock, the 256-channel-tile probe, then the GPU test suite
cd "$(dirname "$0")/.."
python -c "from onedc_amd import kernels as K; K.lib(\"f16\")" || { echo "STALE LIBRARY"; exit 9; }
mkdir -p gpurun_out/r6c
export TMPDIR=/tmp
timeout 300 tools/micro/mfma_lds_mix > gpurun_out/r6c/power_mix.txt 2>&1
cat gpurun_out/r6c/power_mix.txt
AB_SHAPES=384:256:256,384:512:256,768:128:128 ONEDC_GN_FUSION_MAX_COUT=256 timeout 900 python tools/ab_libs.py tools/exp/libonedc_hip_normt0.so > gpurun_out/r6c/ab_tile256_probe.txt 2>&1
cat gpurun_out/r6c/ab_tile256_probe.txt
timeout 2400 python -m pytest tests -m gpu -x -q > gpurun_out/r6c/pytest_gpu.txt 2>&1
tail -15 gpurun_out/r6c/pytest_gpu.txt
