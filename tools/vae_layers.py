#!/usr/bin/env python3
"""Per-layer-shape device time of the conv launches of one VAE decode (B=16, 768x768)."""
import collections
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from onedc_amd import kernels as K
from onedc_amd import synth
from onedc_amd.kernels import new_view
from onedc_amd.vae import VAEDecoder

DEV = "cuda:0"
vae = VAEDecoder(synth.vae_state_dict(0), DEV, attn_patch=16)
x0 = new_view(16, 96, 96, 4, DEV, torch.float32)
x0.t.normal_()
vae.decode(x0)
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
    taps = 4 if kw.get("up") == 2 else kw.get("KH", 1) ** 2
    cin2 = kw["x2"].C if kw.get("x2") is not None else 0      # conv_shortcut riding on conv2 (round 4)
    rec.append(((kw.get("KH", 1), y.H, kw.get("Cin") or x.C, cout, kw.get("up", 0), "gn" if kw.get("gn") is not None else "",
                 ("res" if kw.get("res") is not None else "") + (f"+1x1 {cin2}" if cin2 else "")), e0, e1,
                2.0 * x.B * y.H * y.W * cout * (taps * (kw.get("Cin") or x.C) + cin2) * max(1, kw.get("batch", 1))))
    return r


K.conv_gemm = wrapped
import onedc_amd.layers as L
L.K.conv_gemm = wrapped
t0, t1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
t0.record()
vae.decode(x0)
t1.record()
torch.cuda.synchronize()
agg = collections.OrderedDict()
for key, e0, e1, fl in rec:
    d = agg.setdefault(key, [0, 0.0, 0.0])
    d[0] += 1
    d[1] += e0.elapsed_time(e1)
    d[2] += fl
tot = sum(v[1] for v in agg.values())
print(f"VAE decode {t0.elapsed_time(t1):.2f} ms; conv launches {len(rec)}, {tot:.2f} ms")
for key, (n, ms, fl) in sorted(agg.items(), key=lambda kv: -kv[1][1])[:22]:
    print(f"k{key[0]} out {key[1]:3d}^2 {key[2]:4d}->{key[3]:<4d} up{key[4]} {key[5]:2s} {key[6]:9s} x{n:3d} {ms:7.3f} ms {fl / ms / 1e9:7.1f} TF executed")
