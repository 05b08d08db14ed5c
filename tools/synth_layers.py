#!/usr/bin/env python3
"""Per-layer-shape device time of the conv launches of the codec's synthesis stage (B=16, 768x768)."""
import collections
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

from onedc_amd import kernels as K
from onedc_amd.model_sd15_with_codec_stage1 import SD15_1step_codec_stage1

B, S = 16, 768
model = SD15_1step_codec_stage1({"vae_attn_patch": 16}).load_synthetic(0)
model.codec_model.update(force=True)
idx = torch.from_numpy(np.random.default_rng(1).integers(0, 16384, (B, S // 64, S // 64)).astype(np.int32)).cuda()
blobs, _ = model.codec_model.encode_latents(idx, None, noise_seed=2)
ent = model.codec_model.decode_entropy(stream=blobs)
cm = model.codec_model
cm.synthesis(ent[0], ent[1])
torch.cuda.synchronize()
rec = []
orig = K.conv_gemm


def wrapped(x, w, bias, y, **kw):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    r = orig(x, w, bias, y, **kw)
    e1.record()
    cout = kw.get("Cout") or y.C
    rec.append(((kw.get("KH", 1), x.H, kw.get("Cin") or x.C, cout, kw.get("epi", 0), kw.get("pixel_shuffle", 0),
                 "f32" if x.f32 else "bf16"), e0, e1,
                2.0 * x.B * x.H * x.W * cout * kw.get("KH", 1) ** 2 * (kw.get("Cin") or x.C)))
    return r


K.conv_gemm = wrapped
import onedc_amd.layers as L
L.K.conv_gemm = wrapped
t0, t1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
t0.record()
cm.synthesis(ent[0], ent[1])
t1.record()
torch.cuda.synchronize()
agg = collections.OrderedDict()
for key, e0, e1, fl in rec:
    d = agg.setdefault(key, [0, 0.0, 0.0])
    d[0] += 1
    d[1] += e0.elapsed_time(e1)
    d[2] += fl
tot = sum(v[1] for v in agg.values())
print(f"synthesis {t0.elapsed_time(t1):.2f} ms; conv launches {len(rec)}, {tot:.2f} ms")
for key, (n, ms, fl) in sorted(agg.items(), key=lambda kv: -kv[1][1])[:20]:
    print(f"k{key[0]} in {key[1]:3d}^2 {key[2]:4d}->{key[3]:<4d} epi{key[4]} ps{key[5]} {key[6]:4s} x{n:3d} {ms:7.3f} ms {fl / ms / 1e9:7.1f} TF")
