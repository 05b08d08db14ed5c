#!/usr/bin/env python3
"""BASELINE config 5 on one GPU: full encode (SD-VAE encoder, analysis transform, hyper analysis, four-part prior,
host rANS) + decode of 512x512 pictures, batch 32 (the 4-GPU config shards 8 per GPU; here one GPU takes 32).
Synthetic weights and pictures.  Prints encode and decode images/s and the bpp of the synthetic streams."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from onedc_amd.model_sd15_with_codec_stage1 import SD15_1step_codec_stage1

B, S = int(os.environ.get("B", 32)), int(os.environ.get("S", 512))
DEV = "cuda:0"
model = SD15_1step_codec_stage1({"vae_attn_patch": 16}).load_synthetic(0).load_synthetic_encoder(0)
model.codec_model.update(force=True)
g = torch.Generator().manual_seed(3)
yy, xx = torch.meshgrid(torch.linspace(-1, 1, S), torch.linspace(-1, 1, S), indexing="ij")
img = (0.5 * torch.rand(B, 3, S, S, generator=g) - 0.25 + 0.5 * (yy + xx)[None, None] * 0.5).clamp(-1, 1).to(DEV)
noise = torch.randn(B, S // 8, S // 8, 4, generator=g).to(DEV)
blobs, bpp = model.encode(img, noise=noise)          # warm-up
out = model.decode(stream=blobs)
torch.cuda.synchronize()
n = 3
t0 = time.perf_counter()
for _ in range(n):
    blobs, bpp = model.encode(img, noise=noise)
torch.cuda.synchronize()
t1 = time.perf_counter()
for _ in range(n):
    out = model.decode(stream=blobs)
torch.cuda.synchronize()
t2 = time.perf_counter()
assert out.shape == (B, 3, S, S) and bool(torch.isfinite(out).all())
print(f"{S}x{S} batch {B}: encode {B * n / (t1 - t0):7.1f} images/s ({(t1 - t0) / n * 1e3:7.1f} ms/batch), "
      f"decode {B * n / (t2 - t1):7.1f} images/s ({(t2 - t1) / n * 1e3:7.1f} ms/batch), "
      f"mean bpp {sum(b['bpp'] for b in bpp) / B:.4f} (synthetic weights)")
