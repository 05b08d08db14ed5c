#!/usr/bin/env python3
"""Where does the product-vs-oracle error of a decoded picture come from?

Runs the decode path after the entropy stage on one synthetic picture (random-init weights) on the GPU and in the
fp32 oracle, then swaps ONE product stage at a time into the oracle's chain:

    synthesis (semantic adaptor + codec decoder)   product -> oracle UNet -> oracle VAE
    UNet step (incl. vae_reduction)                oracle h_detail -> product -> oracle VAE
    VAE decoder                                    oracle x0 -> product

and prints PSNR(picture, oracle picture) on [0,1] for each, plus the north-star measure
|PSNR(product, orig) - PSNR(oracle, orig)| for synthetic "originals" at 25 and 27 dB from the oracle picture
(reference src/test_quality.py:234-235).  python tests/devtools/error_budget.py [--size 256] [--seed 0]
"""
import argparse
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))

import numpy as np
import torch

from onedc_amd import kernels as K
from onedc_amd import synth
from onedc_amd.kernels import View, from_nchw, new_view
from onedc_amd.model_sd15_with_codec_stage1 import SD15_1step_codec_stage1
from oracle import torch_ref as R


def nhwc_f32(t, dev):
    B, C, H, W = t.shape
    return View(t.permute(0, 2, 3, 1).contiguous().to(dev, torch.float32).reshape(-1), B, H, W, C)


def delta_psnr(img, ref, level_db, seed=0):
    """|PSNR(img, orig) - PSNR(ref, orig)| with orig = ref + white noise such that PSNR(ref, orig) = level_db."""
    g = torch.Generator().manual_seed(seed)
    sigma = 10 ** (-level_db / 20)
    orig = ref + sigma * torch.randn(ref.shape, generator=g)
    return abs(R.psnr(img, orig) - R.psnr(ref, orig))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--size", type=int, default=256)
    ap.add_argument("--seed", type=int, default=0)
    ap.add_argument("--threads", type=int, default=32)
    ap.add_argument("--act", default=None, help="bf16 | f16 (default: ONEDC_ACT / f16)")
    a = ap.parse_args()
    if a.act:
        K.set_act(a.act)
        K.DEFAULT_ACT = a.act
    print(f"activation format: {K.act()}")
    torch.set_num_threads(a.threads)
    dev = torch.device("cuda", 0)
    model = SD15_1step_codec_stage1({"vae_attn_patch": 16, "act_dtype": K.act()}, device=dev).load_synthetic(0)
    model.codec_model.update(force=True)
    S = a.size
    zh = S // 64
    idx = torch.from_numpy(np.random.default_rng(a.seed).integers(0, 16384, (1, zh, zh)).astype(np.int32)).to(dev)
    _, y_hat = model.codec_model.encode_latents(idx, None, noise_seed=a.seed + 1)
    codec_sd, unet_sd, vae_sd = synth.codec_state_dict(0), R.fold_lora(synth.unet_state_dict(0)), synth.vae_state_dict(0)
    with torch.no_grad():
        _, z_sem = R.z_indices_to_params(codec_sd, idx.cpu())
        y_hat_cpu = y_hat.torch_nchw().cpu()
        ref = R.decode_from_latents(codec_sd, unet_sd, vae_sd, y_hat_cpu, z_sem)
    ref_img = R.to_unit_range(ref["image"])
    t999 = torch.full((1,), 999, dtype=torch.long)
    acp = R.ddim_alphas_cumprod().double()

    def report(name, img):
        img = R.to_unit_range(img)
        print(f"{name:46s} PSNR vs oracle {R.psnr(img, ref_img):6.2f} dB   dPSNR@25 {delta_psnr(img, ref_img, 25):.4f}  "
              f"@27 {delta_psnr(img, ref_img, 27):.4f}", flush=True)

    def rel(got, want):
        return float((got.double() - want.double()).pow(2).mean().sqrt() / want.double().pow(2).mean().sqrt())

    # whole product chain
    h_p, ysem_p = model.codec_model.synthesis(y_hat, from_nchw(z_sem.to(dev)))
    out = model.decode_latents(h_p, ysem_p)
    print(f"stage rel. RMS vs oracle: y_semantic {rel(ysem_p.torch_nchw().cpu(), ref['y_semantic']):.2e}  h_detail "
          f"{rel(h_p.torch_nchw().cpu(), ref['h_detail']):.2e}  eps {rel(out['eps'].torch_nchw().cpu(), ref['eps']):.2e}  "
          f"reduced {rel(out['reduced'].torch_nchw().cpu(), ref['reduced']):.2e}  x0 {rel(out['x0'].torch_nchw().cpu(), ref['x0']):.2e}  "
          f"image {rel(out['image'].torch_nchw().cpu(), ref['image']):.2e}")
    print(f"magnitudes (oracle rms): eps {ref['eps'].pow(2).mean().sqrt():.3f} reduced {ref['reduced'].pow(2).mean().sqrt():.3f} "
          f"x0 {ref['x0'].pow(2).mean().sqrt():.3f} image {ref['image'].pow(2).mean().sqrt():.3f}")
    report("all product stages", out["image"].torch_nchw().cpu())
    with torch.no_grad():
        # 1. product synthesis only
        ctx = ysem_p.torch_nchw().cpu().permute(0, 2, 3, 1).reshape(1, -1, 768)
        eps, red = R.unet_forward(unet_sd, h_p.torch_nchw().cpu(), t999, ctx)
        x0 = R.get_x0_from_noise(red.double(), eps.double(), acp, t999).float()
        report("product synthesis -> oracle UNet, VAE", R.vae_decode(vae_sd, x0, 16))
        # 2. product UNet only (oracle inputs rounded to bf16, as the product stores them)
        eps_v, red_v = model.feedforward_model(from_nchw(ref["h_detail"].to(dev)), 999, from_nchw(ref["y_semantic"].to(dev)))
        x0 = R.get_x0_from_noise(red_v.torch_nchw().cpu().double(), eps_v.torch_nchw().cpu().double(), acp, t999).float()
        report("oracle synthesis -> product UNet -> oracle VAE", R.vae_decode(vae_sd, x0, 16))
        print(f"    product UNet on oracle inputs: eps rel {rel(eps_v.torch_nchw().cpu(), ref['eps']):.2e}  reduced rel "
              f"{rel(red_v.torch_nchw().cpu(), ref['reduced']):.2e}  x0 rel {rel(x0, ref['x0']):.2e}")
        # 3. product VAE only
        img = model.vae_large.decode(nhwc_f32(ref["x0"], dev))
        report("oracle x0 -> product VAE", img.torch_nchw().cpu())


if __name__ == "__main__":
    main()
