#!/bin/bash
# bf16 operands under f16 storage: kernel tests, accuracy at both operating points, same-box A/B of the decode bench
out=${1:-gpurun_out/r5_bf16ops}; mkdir -p "$out"
timeout 900 python3 -m pytest tests/test_gpu_kernels.py -x -q -m gpu -k "bf16_operands or groupnorm_on_load or second_source" > "$out/pytest.txt" 2>&1; tail -4 "$out/pytest.txt"
timeout 900 python3 tests/devtools/mixed_precision.py --rounds 0 --only none 2>&1 | grep -v amdgpu > "$out/accuracy_on.txt"; cat "$out/accuracy_on.txt"
ONEDC_BF16_OPERANDS=0 timeout 900 python3 tests/devtools/mixed_precision.py --rounds 0 --only none 2>&1 | grep -v amdgpu > "$out/accuracy_off.txt"; cat "$out/accuracy_off.txt"
tools/r5_ab_env.sh ONEDC_BF16_OPERANDS 0 1 3 "$out"
