#!/usr/bin/env python3
"""The cross-attention half of a transformer block at the UNet's first level (96 x 96, C = 320, 144 context tokens, B = 16):
the fused launch (csrc/xattn_block.hip) against the four launches it replaces, interleaved rounds in one process."""
import os
import statistics
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from onedc_amd import kernels as K
from onedc_amd import layers as L
from onedc_amd.kernels import from_nchw

DEV = "cuda:0"
B, HW, C = int(os.environ.get("B", 16)), int(os.environ.get("HW", 96)), 320
g = torch.Generator().manual_seed(0)
sd = {f"a.{n}.weight": torch.randn(C, 768 if n in ("to_k", "to_v") else C, generator=g) * (768 if n in ("to_k", "to_v") else C) ** -0.5
      for n in ("to_q", "to_k", "to_v", "to_out.0")}
sd["a.to_out.0.bias"] = torch.randn(C, generator=g) * 0.1
sd["n.weight"], sd["n.bias"] = torch.ones(C), torch.zeros(C)
h = from_nchw(torch.randn(B, C, HW, HW, generator=g).to(DEV))
ctx = from_nchw(torch.randn(B, 768, HW // 8, HW // 8, generator=g).to(DEV))
ln = L.LayerNorm(sd, "n", DEV)
layer = L.MultiHeadAttention(sd, "a", DEV, C, 768, self_attention=False).prepare_fused(ln, DEV)
ws = L.Workspace(DEV)
kv = L.ContextKV([layer], DEV)(ctx, ws)[0]
y = K.new_view(B, HW, HW, C, DEV)


def fused():
    assert layer.block_fused(h, ln, ctx, kv, y) is not None


def unfused():
    layer(ln(h), ctx, h, y, ws=ws, kv=kv)


times = {"fused": [], "unfused": []}
n = 20
for rnd in range(7):
    for name, fn in (("fused", fused), ("unfused", unfused)) if rnd % 2 == 0 else (("unfused", unfused), ("fused", fused)):
        K.USE_XATTN_BLOCK = name == "fused"
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(n):
            fn()
        e1.record()
        torch.cuda.synchronize()
        if rnd:
            times[name].append(e0.elapsed_time(e1) / n * 1e3)
rows = B * HW * HW
gf = (2 * 2.0 * rows * C * C + 4.0 * rows * (HW // 8) ** 2 * C) / 1e9
for name in times:
    us = statistics.median(times[name])
    print(f"{name:8s} {us:8.1f} us   {gf / us * 1e3:7.1f} TFLOP/s algorithmic   {3 * rows * C * 2 / us / 1e6:6.2f} TB/s of (h read twice + y written)")
print(f"fused / unfused time {statistics.median(times['fused']) / statistics.median(times['unfused']):.3f}")
