#!/usr/bin/env python3
"""The V^T projection GEMMs of the self-attention layers (batched, weights as the pixel operand) with the bf16-storing generic
epilogue (what the f16 build runs: out_bf16) against the fast f16 epilogue: what a fast bf16-storing epilogue would be worth."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from onedc_amd import kernels as K
from onedc_amd import pack
from onedc_amd.kernels import View

DEV = "cuda:0"
g = torch.Generator(device=DEV).manual_seed(0)
for name, C, Nk in (("level 0: 320 x 320 -> V^T [320][9216]", 320, 9216), ("level 1: 640 x 640 -> V^T [640][2304]", 640, 2304)):
    B = 16
    wv = View(pack.pack_matrix(torch.randn(C, C, generator=g, device=DEV) * C ** -0.5, DEV), 1, 1, C, C)
    ctx = torch.randn(B * Nk * C + 128 * C, generator=g, device=DEV).to(K.a16())
    for obf in (True, False):
        vt = torch.empty(B * C * Nk, dtype=torch.bfloat16 if obf else K.a16(), device=DEV)
        def run():
            K.conv_gemm(wv, ctx, None, View(vt, 1, 1, C, Nk, Nk), Cin=C, Cout=Nk, batch=B, x_bs=0, w_bs=Nk * C, y_bs=C * Nk, out_bf16=obf)
        for _ in range(3):
            run()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(20):
            run()
        e1.record()
        torch.cuda.synchronize()
        print(f"{name}: out_bf16={obf}: {e0.elapsed_time(e1) / 20 * 1e3:7.1f} us")
