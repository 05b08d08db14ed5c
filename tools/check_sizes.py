#!/usr/bin/env python3
"""Full decode at a few picture sizes: finite output, and a picture alone == the same picture inside a batch (bits)."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

from onedc_amd.model_sd15_with_codec_stage1 import SD15_1step_codec_stage1

model = SD15_1step_codec_stage1({"vae_attn_patch": 16}).load_synthetic(0)
model.codec_model.update(force=True)
for (H, W, B) in [(512, 768, 3), (1024, 1024, 2), (256, 256, 5), (704, 576, 2), (1280, 768, 2), (1408, 2048, 2), (2048, 2048, 1)]:   # (the last two: the CLIC 2K class)
    rng = np.random.default_rng(H + W)
    idx = torch.from_numpy(rng.integers(0, 16384, (B, H // 64, W // 64)).astype(np.int32)).cuda()
    blobs, _ = model.codec_model.encode_latents(idx, None, noise_seed=3)
    whole = model.decode(stream=blobs)
    alone = model.decode(stream=blobs[B - 1:B])
    torch.cuda.synchronize()
    ok = bool(torch.isfinite(whole).all()) and whole.shape == (B, 3, H, W)
    same = bool(torch.equal(whole[B - 1:B], alone))
    print(f"{H}x{W} B={B}: finite {ok}, alone == in batch: {same}, range [{float(whole.min()):.2f}, {float(whole.max()):.2f}]")
