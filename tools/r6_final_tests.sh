#!/bin/bash
# round 6: the GPU suite, the entry point's smoke and the large-size probe on the final tree
cd "$(dirname "$0")/.."
python -c "from onedc_amd import kernels as K; K.lib(\"f16\"); K.lib(\"bf16\")" || { echo "STALE LIBRARY"; exit 9; }
mkdir -p gpurun_out/r6h
timeout 3000 python -m pytest tests -m gpu -x -q > gpurun_out/r6h/pytest_gpu.txt 2>&1
tail -6 gpurun_out/r6h/pytest_gpu.txt
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -3
timeout 900 python tests/devtools/large_sizes.py 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r6h/large_sizes.txt
