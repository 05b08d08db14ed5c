#!/usr/bin/env python3
"""Diagnostic (EXP_STAMP build only): where a wave of the 128x256 conv kernel spends its cycles.
usage: ONEDC_HIP_SO=tools/exp/libonedc_hip_stamp.so stamp_conv.py B H W Cin Cout k up"""
import ctypes as C
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from onedc_amd import kernels as K
from onedc_amd import pack

B, H, W, Cin, Cout, k, up = map(int, sys.argv[1:8])
TILE = int(sys.argv[8]) if len(sys.argv) > 8 else 129
DEV = "cuda:0"
g = torch.Generator(device=DEV).manual_seed(0)
x = K.View(torch.randn((B * H * W + 128) * Cin, generator=g, device=DEV).to(K.a16()), B, H, W, Cin)
w = pack.pack_matrix(torch.randn(Cout, k * k * Cin, generator=g, device=DEV) * (k * k * Cin) ** -0.5, DEV)
OH, OW = (2 * H, 2 * W) if up else (H, W)
y = K.new_view(B, OH, OW, Cout, DEV)
dbg = torch.zeros(2048 * 8 * 8, dtype=torch.int64, device=DEV)


class FakeStats:
    t = dbg

    def mark(self, *a):
        pass


y.stats = FakeStats()
for _ in range(3):
    K.conv_gemm(x, w, None, y, KH=k, up=up, tile=TILE, stats=True)
torch.cuda.synchronize()
d = dbg.view(2048, 8, 8).cpu().double()
names = ["prologue", "load-seg issue", "barrier1", "lgkmcnt wait", "mfma seg", "barrier2", "vmcnt wait", "epilogue"]
tot = d.sum(-1).mean()
print(f"mean cycles per wave (first 2048 blocks): total {tot:.0f}")
for i, n in enumerate(names):
    print(f"  {n:16s} {d[..., i].mean():9.0f}  {100 * d[..., i].mean() / tot:5.1f} %   (wave row 0: {d[:, :4, i].mean():8.0f}, row 1: {d[:, 4:, i].mean():8.0f})")
