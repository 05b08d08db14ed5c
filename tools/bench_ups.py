#!/usr/bin/env python3
"""x2 upsample + 3x3: fused-upsample window kernel (up=1) against the four 2x2 phase convolutions (up=2)."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from onedc_amd import kernels as K
from onedc_amd import pack

DEV = "cuda:0"
g = torch.Generator(device=DEV).manual_seed(0)
for name, B, H, W, C in [("vae 512 96->192", 16, 96, 96, 512), ("vae 512 192->384", 16, 192, 192, 512),
                         ("vae 256 384->768", 16, 384, 384, 256), ("unet 640 48->96", 16, 48, 48, 640)]:
    x = K.View(torch.randn((B * H * W + 128) * C, generator=g, device=DEV).to(K.a16()), B, H, W, C)
    w = torch.randn(C, C, 3, 3, generator=g, device=DEV) * (9 * C) ** -0.5
    w1 = pack.pack_matrix(pack.conv_matrix(w), DEV)
    w2 = pack.pack_matrix(pack.upsample_phase_matrix(w), DEV)
    y = K.new_view(B, 2 * H, 2 * W, C, DEV)
    res = []
    for up, wp, tile in ((1, w1, 0), (2, w2, 130), (2, w2, 131)):      # fused upsample; phases on the 8-wave / 4-wave kernel
        for _ in range(2):
            K.conv_gemm(x, wp, None, y, KH=3, up=up, Cin=C, Cout=C, tile=tile)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(5):
            K.conv_gemm(x, wp, None, y, KH=3, up=up, Cin=C, Cout=C, tile=tile)
        e1.record()
        torch.cuda.synchronize()
        res.append(e0.elapsed_time(e1) / 5)
    f9 = 2.0 * B * 4 * H * W * C * 9 * C
    print(f"{name:20s} up=1 {res[0]:7.3f} ms {f9 / res[0] / 1e9:7.1f} TF | phases, 8 waves {res[1]:7.3f} ms "
          f"{f9 * 4 / 9 / res[1] / 1e9:7.1f} TF executed | phases, 4 waves {res[2]:7.3f} ms {f9 * 4 / 9 / res[2] / 1e9:7.1f} TF")
