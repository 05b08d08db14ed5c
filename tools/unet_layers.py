#!/usr/bin/env python3
"""Per-layer-shape device time of the conv launches of one UNet step (B=16): where the generic kernels' time goes."""
import collections
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from onedc_amd import kernels as K
from onedc_amd import synth
from onedc_amd.decoder_unet import UNet
from onedc_amd.kernels import new_view

DEV = "cuda:0"
unet = UNet(synth.unet_state_dict(0), DEV)
g = torch.Generator(device=DEV).manual_seed(0)
x = new_view(16, 96, 96, 320, DEV)
x.t.copy_(torch.randn(x.t.numel(), generator=g, device=DEV).to(K.a16()))
ctx = new_view(16, 12, 12, 768, DEV)
ctx.t.copy_(torch.randn(ctx.t.numel(), generator=g, device=DEV).to(K.a16()))
unet(x, 999, ctx)
torch.cuda.synchronize()
rec = []
orig = K.conv_gemm


def wrapped(x, w, bias, y, **kw):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    r = orig(x, w, bias, y, **kw)
    e1.record()
    if r is None:                      # (a second source the dispatcher refused: nothing was launched)
        return r
    cout = kw.get("Cout") or y.C
    cin, up = kw.get("Cin") or x.C, kw.get("up", 0)
    # rows are keyed by the OUTPUT map (a stride-2 or upsampling layer is not the layer of its input size), FLOPs are the
    # EXECUTED ones: the phase-decomposed upsample conv (up == 2) runs 4 taps per output, not the reference's 9
    taps = 4 if up == 2 else kw.get("KH", 1) ** 2
    cin2 = kw["x2"].C if kw.get("x2") is not None else 0
    tag = ("res" if kw.get("res") is not None else "") + (f" up{up}" if up else "") + (f" +1x1 {cin2}" if cin2 else "") + \
        (f" s{kw['stride']}" if kw.get("stride", 1) != 1 else "")
    rec.append(((kw.get("KH", 1), y.H, cin, cout, kw.get("epi", 0), kw.get("batch", 1), tag), e0, e1,
                2.0 * x.B * y.H * y.W * cout * (taps * cin + cin2) * max(1, kw.get("batch", 1))))
    return r


xrec = []
orig_x = K.xattn_block


def wrapped_x(p):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    orig_x(p)
    e1.record()
    rows = p.B * p.HW
    xrec.append((e0, e1, 2 * 2.0 * rows * p.C * p.C + 4.0 * rows * p.Nk * p.C, 3.0 * rows * p.C * 2))


K.conv_gemm = wrapped
K.xattn_block = wrapped_x
import onedc_amd.layers as L
L.K.conv_gemm = wrapped
unet(x, 999, ctx)
torch.cuda.synchronize()
agg = collections.OrderedDict()
for key, e0, e1, fl in rec:
    d = agg.setdefault(key, [0, 0.0, 0.0])
    d[0] += 1
    d[1] += e0.elapsed_time(e1)
    d[2] += fl
tot = sum(v[1] for v in agg.values())
print(f"conv launches {len(rec)}, {tot:.2f} ms")
for key, (n, ms, fl) in sorted(agg.items(), key=lambda kv: -kv[1][1])[:32]:
    print(f"k{key[0]} out {key[1]:3d}x{key[1]:<3d} {key[2]:5d}->{key[3]:<5d} epi{key[4]} batch{key[5]:<3d} {key[6]:14s} x{n:3d} {ms:7.3f} ms {fl / ms / 1e9:7.1f} TF executed")
if xrec:
    ms = sum(e0.elapsed_time(e1) for e0, e1, _, _ in xrec)
    fl, by = sum(r[2] for r in xrec), sum(r[3] for r in xrec)
    print(f"fused cross-attention block (norm2 + attn2.to_q + attention + attn2.to_out + residual, csrc/xattn_block.hip) x{len(xrec):3d} "
          f"{ms:7.3f} ms  {ms / len(xrec) * 1e3:6.1f} us each  {fl / ms / 1e9:6.1f} TF algorithmic  {by / ms / 1e9:5.2f} TB/s of (h read twice + y written)")
