#!/usr/bin/env python3
"""Diagnostic (library built with -DEXP_XA_TIMELINE): cycles of the phases of one workgroup of xattn_block_kernel, from the s_memtime
stamps wave 0 leaves in its tile's first output row.
usage: tools/exp_build.sh xatl -DONEDC_ACT_F16 -DEXP_XA_TIMELINE; ONEDC_HIP_F16_SO=tools/exp/libonedc_hip_xatl.so python tools/timeline_xattn.py"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

from onedc_amd import kernels as K
from onedc_amd import layers as L
from onedc_amd.kernels import from_nchw

DEV = "cuda:0"
B, HW, C = 16, 96, 320
g = torch.Generator().manual_seed(0)
sd = {f"a.{n}.weight": torch.randn(C, 768 if n in ("to_k", "to_v") else C, generator=g) * 0.04 for n in ("to_q", "to_k", "to_v", "to_out.0")}
sd["a.to_out.0.bias"] = torch.randn(C, generator=g) * 0.1
sd["n.weight"], sd["n.bias"] = torch.ones(C), torch.zeros(C)
h = from_nchw(torch.randn(B, C, HW, HW, generator=g).to(DEV))
ctx = from_nchw(torch.randn(B, 768, HW // 8, HW // 8, generator=g).to(DEV))
ln = L.LayerNorm(sd, "n", DEV)
layer = L.MultiHeadAttention(sd, "a", DEV, C, 768, self_attention=False).prepare_fused(ln, DEV)
kv = L.ContextKV([layer], DEV)(ctx, L.Workspace(DEV))[0]
y = K.new_view(B, HW, HW, C, DEV)
for _ in range(30):
    assert layer.block_fused(h, ln, ctx, kv, y) is not None
torch.cuda.synchronize()
rows = B * HW * HW
raw = y.t[:rows * C].view(rows // 64, 64 * C)[:, :32].contiguous().view(torch.int64).cpu().numpy()      # 8 stamps per tile
d = np.diff(raw, axis=1).astype(np.float64)
names = ["P0 load + LayerNorm -> LDS", "P1 q = Wq . n (10 K-steps)", "P2 q -> LDS, first head's K / V^T loaded", "P3 attention, 8 heads",
         "P4 y = Wo . o (10 K-steps, residual loads in front)", "P5 epilogue through LDS, stores issued", "stores drained"]
tot = (raw[:, 7] - raw[:, 0]).mean()
print(f"{raw.shape[0]} workgroups; s_memtime ticks of wave 0 per phase (mean, p10 / p90), whole workgroup {tot:.0f}:")
for i, n in enumerate(names):
    print(f"  {d[:, i].mean():8.0f}  ({np.percentile(d[:, i], 10):7.0f} / {np.percentile(d[:, i], 90):7.0f})  {100 * d[:, i].mean() / tot:5.1f} %  {n}")
