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
