#!/usr/bin/env python3
"""Wall time of the two decode stages in isolation (B=16, 768x768): entropy stage E and synthesis/UNet/VAE stage S."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

from onedc_amd.model_sd15_with_codec_stage1 import SD15_1step_codec_stage1

B, S = 16, 768
model = SD15_1step_codec_stage1({"vae_attn_patch": 16}).load_synthetic(0)
model.codec_model.update(force=True)
idx = torch.from_numpy(np.random.default_rng(1).integers(0, 16384, (B, S // 64, S // 64)).astype(np.int32)).cuda()
blobs, _ = model.codec_model.encode_latents(idx, None, noise_seed=2)


def timed(fn, n=4):
    fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e3


ent = model.codec_model.decode_entropy(stream=blobs)
print(f"stage E (entropy)            {timed(lambda: model.codec_model.decode_entropy(stream=blobs)):8.2f} ms")
print(f"stage S (synth + UNet + VAE) {timed(lambda: model._decode_after_entropy(ent)):8.2f} ms")
print(f"decode() sequential          {timed(lambda: model.decode(stream=blobs)):8.2f} ms")
pipe = model.pipeline()
print(f"pipeline, per batch          {timed(lambda: pipe.map([blobs] * 4), 2) / 4:8.2f} ms")
pipe.close()
# host-side launch cost of stage S: enqueue only
torch.cuda.synchronize()
t0 = time.perf_counter()
model._decode_after_entropy(ent)
t1 = time.perf_counter()
torch.cuda.synchronize()
print(f"stage S host enqueue time    {(t1 - t0) * 1e3:8.2f} ms")

# ---- device time per sub-stage of S (events on the current stream, sequential) -----------------------------
cm = model.codec_model
ev = [torch.cuda.Event(enable_timing=True) for _ in range(5)]
torch.cuda.synchronize()
ev[0].record()
h, ysem = cm.synthesis(ent[0], ent[1])
ev[1].record()
eps, reduced = model.feedforward_model(h, model.conditioning_timestep, ysem)
ev[2].record()
from onedc_amd import kernels as K
from onedc_amd.kernels import new_view
x0 = new_view(eps.B, eps.H, eps.W, 4, model.device, torch.float32)
n = eps.B * eps.H * eps.W * 4
K.x0_from_noise(reduced.t[:n], eps.t[:n], x0.t[:n], float(model.alphas_cumprod[999].double()))
img = model.vae_large.decode(x0)
ev[3].record()
torch.cuda.synchronize()
gf = {"codec synthesis": 16 * (11.5 + 222.6), "UNet step": 16 * 2191.7, "VAE decoder": 16 * 5585.2}
for i, name in enumerate(["codec synthesis", "UNet step", "VAE decoder"]):
    ms = ev[i].elapsed_time(ev[i + 1])
    print(f"{name:18s} {ms:8.2f} ms   {gf[name] / ms:8.1f} TFLOP/s (algorithmic)")
