#!/usr/bin/env python3
"""A handful of launches of one conv shape (for rocprofv3 --pmc runs, which serialise and slow every dispatch).
usage: pmc_conv.py B H W Cin Cout k up tile [launches [res]]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from onedc_amd import kernels as K
from onedc_amd import pack

B, H, W, Cin, Cout, k, up, tile = map(int, sys.argv[1:9])
n = int(sys.argv[9]) if len(sys.argv) > 9 else 3
res = len(sys.argv) > 10 and sys.argv[10] == "1"
DEV = "cuda:0"
g = torch.Generator(device=DEV).manual_seed(0)
x = K.View(torch.randn((B * H * W + 128) * Cin, generator=g, device=DEV).to(K.a16()), B, H, W, Cin)
w = pack.pack_matrix(torch.randn(Cout, k * k * Cin, generator=g, device=DEV) * (k * k * Cin) ** -0.5, DEV)
OH, OW = (2 * H, 2 * W) if up else (H, W)
y = K.new_view(B, OH, OW, Cout, DEV)
r = K.new_view(B, OH, OW, Cout, DEV) if res else None
if r is not None:
    r.t.normal_()
for _ in range(n):
    K.conv_gemm(x, w, None, y, KH=k, up=up, tile=tile, res=r)
torch.cuda.synchronize()
print("done")
