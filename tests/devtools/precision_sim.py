#!/usr/bin/env python3
"""CPU simulation of storage / operand precisions on the oracle (no GPU): which roundings of the HIP path spend the
0.01 dB budget?  Patches oracle.torch_ref so that every conv / linear rounds its input (`op`), its weights (`w`) and
its output (`store`), and every residual block rounds the stream it hands on (`stream`), then compares the decoded
picture with the pure-fp32 oracle.  Formats: f32, bf16, f16.

    python tests/devtools/precision_sim.py --size 256 --stage vae
"""
import argparse
import os
import re
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))

import torch
import torch.nn.functional as F

from onedc_amd import synth
from oracle import torch_ref as R

FMT = {"f32": None, "bf16": torch.bfloat16, "f16": torch.float16}


def q(x, fmt, dim=1):
    """Round to a storage / operand format.  mxfp8 / mxfp4-style: OCP e4m3 elements with one power-of-two scale per block
    of 32 consecutive elements along `dim` (the K direction of the contraction), as v_mfma_scale_f32_*_f8f6f4 consumes."""
    if fmt in ("mxfp8", "fp8"):
        if fmt == "fp8":                               # one scale for the whole tensor
            amax = x.abs().max().clamp_min(1e-30)
            sc = torch.exp2(torch.floor(torch.log2(amax)) - 8)
            return (x / sc).clamp(-448, 448).to(torch.float8_e4m3fn).float() * sc
        xm = x.movedim(dim, -1)
        shp = xm.shape
        pad = (-shp[-1]) % 32
        xp = torch.nn.functional.pad(xm, (0, pad)).reshape(*shp[:-1], -1, 32)
        amax = xp.abs().amax(dim=-1, keepdim=True).clamp_min(1e-30)
        sc = torch.exp2(torch.floor(torch.log2(amax)) - 8)
        xq = ((xp / sc).clamp(-448, 448).to(torch.float8_e4m3fn).float() * sc).reshape(*shp[:-1], -1)[..., :shp[-1]]
        return xq.movedim(-1, dim)
    return x if FMT[fmt] is None else x.to(FMT[fmt]).float()


class Sim:
    def __init__(self, op, w, store, stream, low=None, low_layers=None):
        """low = (op, w) formats for the layers whose name matches the regular expression `low_layers` (the fp8
        experiment: only some VAE up-block convolutions drop to 8 bits)."""
        self.op, self.w, self.store, self.stream = op, w, store, stream
        self.low, self.low_layers = low, (re.compile(low_layers) if low_layers else None)
        self.on = True

    def install(self):
        self.orig = {n: getattr(R, n) for n in ("conv", "linear", "unet_resnet_block", "vq_resnet_block",
                                                "depth_conv_block4", "residual_block_upsample", "vq_attn_block")}
        sim = self

        def conv(sd, p, x, stride=1, padding=0, groups=1):
            if not sim.on:
                return sim.orig["conv"](sd, p, x, stride, padding, groups)
            op, wf = sim.op, sim.w
            if sim.low_layers is not None and sim.low_layers.search(p) and sd[f"{p}.weight"].shape[-1] == 3:
                op, wf = sim.low
            wt = sd[f"{p}.weight"] if groups > 1 else q(sd[f"{p}.weight"], wf)
            y = F.conv2d(q(x, op), wt, sd.get(f"{p}.bias"), stride=stride, padding=padding, groups=groups)
            return q(y, sim.store)

        def linear(sd, p, x):
            if not sim.on:
                return sim.orig["linear"](sd, p, x)
            return q(F.linear(q(x, sim.op), q(sd[f"{p}.weight"], sim.w), sd.get(f"{p}.bias")), sim.store)

        def wrap(name):
            f = self.orig[name]

            def g(*a, **k):
                y = f(*a, **k)
                return q(y, sim.stream) if sim.on else y
            return g

        R.conv, R.linear = conv, linear
        for n in ("unet_resnet_block", "vq_resnet_block", "depth_conv_block4", "residual_block_upsample", "vq_attn_block"):
            setattr(R, n, wrap(n))

    def remove(self):
        for n, f in self.orig.items():
            setattr(R, n, f)


def psnr_pair(img, ref):
    out = [R.psnr(img, ref)]
    for level in (25.0, 27.0):
        sigma = 10 ** (-level / 20)
        orig = ref + sigma * torch.randn(ref.shape, generator=torch.Generator().manual_seed(3))
        out.append(abs(R.psnr(img, orig) - R.psnr(ref, orig)))
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--size", type=int, default=256)
    ap.add_argument("--stage", default="vae", choices=["vae", "synthesis", "unet", "all"])
    ap.add_argument("--threads", type=int, default=8)
    ap.add_argument("--x0-rms", type=float, default=0.0,
                    help="vae stage: rescale the oracle's x0 to this rms first (random-init weights produce x0 ~ 6e4, where the "
                         "decoder's residual branches hardly matter; trained models have x0 ~ 1)")
    a = ap.parse_args()
    torch.set_num_threads(a.threads)
    S = a.size
    codec_sd, unet_sd, vae_sd = synth.codec_state_dict(0), R.fold_lora(synth.unet_state_dict(0)), synth.vae_state_dict(0)
    g = torch.Generator().manual_seed(0)
    y_hat = torch.randn(1, 128, S // 16, S // 16, generator=g).round()
    z_sem = torch.randn(1, 128, S // 64, S // 64, generator=g)
    with torch.no_grad():
        ref = R.decode_from_latents(codec_sd, unet_sd, vae_sd, y_hat, z_sem)
    if a.x0_rms > 0 and a.stage == "vae":
        with torch.no_grad():
            ref["x0"] = ref["x0"] * (a.x0_rms / ref["x0"].pow(2).mean().sqrt())
            ref["image"] = R.vae_decode(vae_sd, ref["x0"], 16)
        print(f"x0 rescaled to rms {a.x0_rms}; image rms {float(ref['image'].pow(2).mean().sqrt()):.3f}")
    ref_img = R.to_unit_range(ref["image"])
    t = torch.full((1,), 999, dtype=torch.long)
    acp = R.ddim_alphas_cumprod().double()

    def run(stage):
        with torch.no_grad():
            if stage == "vae":
                return R.vae_decode(vae_sd, ref["x0"], 16)
            if stage == "synthesis":
                ys = R.semantic_adaptor(codec_sd, z_sem)
                hd = R.codec_decoder(codec_sd, y_hat, ys)
                sim.on = False
                ctx = ys.permute(0, 2, 3, 1).reshape(1, -1, ys.shape[1])
                eps, red = R.unet_forward(unet_sd, hd, t, ctx)
                x0 = R.get_x0_from_noise(red.double(), eps.double(), acp, t).float()
                img = R.vae_decode(vae_sd, x0, 16)
                sim.on = True
                return img
            if stage == "unet":
                ctx = ref["y_semantic"].permute(0, 2, 3, 1).reshape(1, -1, 768)
                eps, red = R.unet_forward(unet_sd, ref["h_detail"], t, ctx)
                sim.on = False
                x0 = R.get_x0_from_noise(red.double(), eps.double(), acp, t).float()
                img = R.vae_decode(vae_sd, x0, 16)
                sim.on = True
                return img
            return R.decode_from_latents(codec_sd, unet_sd, vae_sd, y_hat, z_sem)["image"]

    ap_cfg = os.environ.get("SIM_CONFIGS")
    configs = [("bf16", "bf16", "bf16", "bf16"), ("bf16", "bf16", "bf16", "f32"), ("bf16", "bf16", "f32", "f32"),
               ("f16", "bf16", "bf16", "bf16"), ("f16", "f16", "bf16", "bf16"), ("f16", "f16", "f16", "f16"),
               ("f16", "f16", "f16", "f32"), ("f16", "f16", "f32", "f32"), ("bf16", "f32", "bf16", "bf16"),
               ("f32", "bf16", "f32", "f32"), ("f32", "f32", "bf16", "bf16"), ("f32", "f32", "f32", "bf16")]
    if ap_cfg:
        configs = [tuple(c.split("/")) for c in ap_cfg.split(",")]
    print(f"stage {a.stage} at {S}x{S}: conv operand / weights / conv-output store / residual stream")
    low_layers = os.environ.get("SIM_LOW_LAYERS")            # e.g. "up_blocks\\.[23]\\.resnets" with SIM_LOW=mxfp8/mxfp8
    low = tuple(os.environ.get("SIM_LOW", "mxfp8/mxfp8").split("/"))
    if low_layers:
        print(f"layers matching /{low_layers}/ (3x3 convs): operand {low[0]}, weights {low[1]}")
    for op, w, store, stream in configs:
        sim = Sim(op, w, store, stream, low, low_layers)
        sim.install()
        try:
            img = R.to_unit_range(run(a.stage))
        finally:
            sim.remove()
        p, d25, d27 = psnr_pair(img, ref_img)
        print(f"  op {op:5s} w {w:5s} store {store:5s} stream {stream:5s}   PSNR {p:6.2f} dB   dPSNR@25 {d25:.4f} @27 {d27:.4f}",
              flush=True)


if __name__ == "__main__":
    main()
