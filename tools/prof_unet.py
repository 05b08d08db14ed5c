#!/usr/bin/env python3
"""Only the one-step UNet (B=16, 96x96 latents, 144 context tokens), three times: for rocprofv3 --stats."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

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
for _ in range(3):
    unet(x, 999, ctx)
torch.cuda.synchronize()
print("done")
