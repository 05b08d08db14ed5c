#!/bin/bash
# round 6, second GPU call: epilogue experiments A/B (interleaved, one process) + the lone-workgroup K loop
cd "$(dirname "$0")/.."
python -c "from onedc_amd import kernels as K; K.lib(\"f16\")" || { echo "STALE LIBRARY"; exit 9; }
mkdir -p gpurun_out/r6b
export TMPDIR=/tmp
timeout 1500 python tools/ab_libs.py tools/exp/libonedc_hip_episc.so tools/exp/libonedc_hip_epiov.so tools/exp/libonedc_hip_epiboth.so > gpurun_out/r6b/ab_epilogue.txt 2>&1
cat gpurun_out/r6b/ab_epilogue.txt
