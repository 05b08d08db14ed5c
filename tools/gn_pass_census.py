#!/usr/bin/env python3
"""Which GroupNorms of one decode (B = 16, 768x768) still run the separate statistics pass, and why: the producer left no partial
sums (stats is None), the sums do not cover the normalised channels, or the group is not whole quads / pairs of the recorded
granularity.  Prints one line per (shape, reason) with the launch count."""
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
model.decode(stream=blobs)
torch.cuda.synchronize()
census = collections.Counter()
orig = K.group_norm_ab


def wrapped(x, gamma, beta, eps, ws):
    if x.stats is None:
        why = "no sums"
    elif not x.stats.covers(x.co, x.C):
        why = "sums do not cover the channels"
    elif (x.C // 32) % x.stats.gran or x.co % x.stats.gran:
        why = f"group of {x.C // 32} channels, sums per {x.stats.gran}"
    else:
        why = None
    census[(x.H, x.W, x.C, why or "from the producers' sums")] += 1
    return orig(x, gamma, beta, eps, ws)


K.group_norm_ab = wrapped
import onedc_amd.layers as L
L.K.group_norm_ab = wrapped
model.decode(stream=blobs)
torch.cuda.synchronize()
for (H, W, C, why), n in sorted(census.items(), key=lambda kv: (kv[0][3], -kv[0][0] * kv[0][1] * kv[0][2])):
    print(f"{H:4d}x{W:<4d} {C:5d} channels  x {n:2d}   {why}")
