#!/usr/bin/env python3
"""GroupNorm apply (+SiLU) pass on the VAE's big maps: effective bandwidth (read + write)."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from onedc_amd import kernels as K

DEV = "cuda:0"
SHAPES = [(16, 768, 768, 128), (16, 768, 768, 256), (16, 384, 384, 256), (16, 384, 384, 512), (16, 192, 192, 512), (16, 96, 96, 512),
          (16, 96, 96, 320), (16, 96, 96, 640), (16, 96, 96, 960), (16, 48, 48, 640), (16, 48, 48, 1280), (16, 48, 48, 1920),
          (16, 24, 24, 1280), (16, 24, 24, 2560), (16, 12, 12, 1280), (16, 12, 12, 2560), (16, 48, 48, 256), (16, 12, 12, 768)]
for B, H, W, C in SHAPES:
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
    print(f"{B}x{H}x{W}x{C}: {ms * 1e3:8.1f} us  {B * H * W * C * 4 / ms / 1e9:6.2f} TB/s  ({B * H * W * C * 4 / 1e6:7.1f} MB read + written)")
