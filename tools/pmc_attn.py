#!/usr/bin/env python3
"""Three launches of the d = 40 self-attention at 9216 tokens (for rocprofv3 --pmc runs)."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from onedc_amd import kernels as K
from onedc_amd import pack

DEV = "cuda:0"
B, H, d, N = 16, 8, 40, 9216
DQ, DV = pack.HEAD_PAD[d]
g = torch.Generator(device=DEV).manual_seed(0)
q = torch.randn(B * N * H * DQ, generator=g, device=DEV).to(K.a16())
k = torch.randn(B * N * H * DQ, generator=g, device=DEV).to(K.a16())
vt = torch.randn(B * H * DV * N, generator=g, device=DEV).to(K.flash_vt_dtype(DQ, DV, N))
o = torch.empty(B * N * H * DV, dtype=K.a16(), device=DEV)
for _ in range(3):
    K.flash_attn(q, k, vt, o, B, H, N, N, N, DQ, DV, d ** -0.5)
torch.cuda.synchronize()
print("done")
