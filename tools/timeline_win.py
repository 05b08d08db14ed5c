#!/usr/bin/env python3
"""Diagnostic (EXP_WIN_TIMELINE build only): where the time of a window-kernel tile goes, from s_memtime stamps of every
workgroup: prologue, K loop, epilogue, and the gap between two consecutive workgroups on the same CU.
usage: ONEDC_HIP_SO=tools/exp/libonedc_hip_timeline.so timeline_win.py B H W Cin Cout"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

from onedc_amd import kernels as K
from onedc_amd import pack

B, H, W, Cin, Cout = map(int, sys.argv[1:6])
DEV = "cuda:0"
g = torch.Generator(device=DEV).manual_seed(0)
x = K.View(torch.randn((B * H * W + 128) * Cin, generator=g, device=DEV).to(K.a16()), B, H, W, Cin)
w = pack.pack_matrix(torch.randn(Cout, 9 * Cin, generator=g, device=DEV) * (9 * Cin) ** -0.5, DEV)
y = K.new_view(B, H, W, Cout, DEV)
blocks = (Cout + 127) // 128 * (H // 16) * (W // 16) * B
dbg = torch.zeros(blocks * 8 * 6, dtype=torch.int64, device=DEV)


class FakeStats:
    t = dbg
    gran = 4

    def mark(self, *a):
        pass


y.stats = FakeStats()
for _ in range(2):
    K.conv_gemm(x, w, None, y, KH=3, tile=130, stats=True)
torch.cuda.synchronize()
d = dbg.view(blocks, 8, 6).cpu().numpy().astype(np.int64)
t0, t1, t2, t3 = d[:, :, 1].min(1), d[:, :, 2].max(1), d[:, :, 3].max(1), d[:, :, 4].max(1)
print(f"{blocks} workgroups; s_memtime ticks per workgroup (min start .. max end over its 8 waves):")
print(f"  prologue (entry -> first barrier)  {np.mean(t1 - t0):9.0f}")
print(f"  K loop                             {np.mean(t2 - t1):9.0f}")
t4 = d[:, :, 5].max(1)
print(f"  epilogue: arithmetic + LDS writes  {np.mean(t4 - t2):9.0f}")
print(f"  epilogue: LDS reads + line stores  {np.mean(t3 - t4):9.0f}")
print(f"  whole workgroup                    {np.mean(t3 - t0):9.0f}")
