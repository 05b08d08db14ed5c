#!/usr/bin/env python3
"""GroupNorm apply (+SiLU) pass on the VAE's big maps: effective bandwidth (read + write)."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from onedc_amd import kernels as K

DEV = "cuda:0"
for B, H, W, C in [(16, 768, 768, 128), (16, 768, 768, 256), (16, 384, 384, 256), (16, 192, 192, 512), (16, 96, 96, 320)]:
    x = K.new_view(B, H, W, C, DEV)
    x.t.normal_()
    y = K.new_view(B, H, W, C, DEV)
    ab = torch.randn(B * C * 2, device=DEV)
    lib = K.lib()
    def run():
        K._ok(lib.onedc_scale_shift_act(x.ptr, ab.data_ptr(), y.ptr, B, H * W, C, x.cs, x.co, y.cs, y.co, 1, K.stream()), "ssa")
    for _ in range(3):
        run()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10):
        run()
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 10
    print(f"{B}x{H}x{W}x{C}: {ms:7.3f} ms  {B * H * W * C * 4 / ms / 1e9:6.2f} TB/s")
