#!/usr/bin/env python3
"""GroupNorm+SiLU -> 3x3 conv: fused (normalise on load) against separate pass + conv, B=16."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from onedc_amd import kernels as K
from onedc_amd import pack

DEV = "cuda:0"
g = torch.Generator(device=DEV).manual_seed(0)


class WS:
    def __init__(self):
        self.t = {}

    def f32(self, name, n):
        if name not in self.t or self.t[name].numel() < n:
            self.t[name] = torch.empty(n, dtype=torch.float32, device=DEV)
        return self.t[name]


for H, Cin, Cout in [(768, 128, 128), (768, 256, 128), (384, 256, 256)]:
    B = 16
    x = K.new_view(B, H, H, Cin, DEV, stats=True)
    x.t.normal_()
    x.stats.t.normal_().abs_()
    x.stats.mark(0, Cin)
    w = pack.pack_matrix(torch.randn(Cout, 9 * Cin, generator=g, device=DEV) * (9 * Cin) ** -0.5, DEV)
    gam, bet = torch.ones(Cin, device=DEV), torch.zeros(Cin, device=DEV)
    y = K.new_view(B, H, H, Cout, DEV)
    ws = WS()
    K.GN_FUSION_MAX_COUT = 1 << 20
    for tile in (130, 131):                          # 8-wave / 4-wave window kernel
        out = []
        for fuse in (True, False):
            K.USE_GN_FUSION = fuse
            for _ in range(2):
                K.conv_gemm(x, w, None, y, KH=3, gn=(gam, bet, 1e-6, True, ws), tile=tile)
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(5):
                K.conv_gemm(x, w, None, y, KH=3, gn=(gam, bet, 1e-6, True, ws), tile=tile)
            e1.record()
            torch.cuda.synchronize()
            out.append(e0.elapsed_time(e1) / 5)
        print(f"{H}^2 {Cin}->{Cout} tile {tile}: fused {out[0]:7.3f} ms   separate pass + conv {out[1]:7.3f} ms")
