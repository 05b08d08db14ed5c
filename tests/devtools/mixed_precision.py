#!/usr/bin/env python3
"""Round 5: per-stage 16-bit formats inside one f16 model.  For every candidate configuration, at both synthetic
operating points, one 768x768 picture: PSNR(product, oracle) and |dPSNR| at 25 / 27 dB (budget 0.01), then the candidates'
pipelines timed against the default's in ONE process, alternating (boxes differ by 3-5 %).

    python tests/devtools/mixed_precision.py [--no-accuracy] [--rounds 3] [--steps 6]
"""
import argparse
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import torch

from onedc_amd import synth
from onedc_amd.model_sd15_with_codec_stage1 import SD15_1step_codec_stage1
from oracle import torch_ref as R

CANDIDATES = {"f16 (default)": {}, "unet bf16": {"unet_act": "bf16"}, "unet + vae up3 bf16": {"unet_act": "bf16", "vae_up3_act": "bf16"},
              "vae up3 bf16": {"vae_up3_act": "bf16"}}


def delta_psnr(img, ref, level_db, seed):
    orig = ref + 10 ** (-level_db / 20) * torch.randn(ref.shape, generator=torch.Generator().manual_seed(seed))
    return abs(R.psnr(img, orig) - R.psnr(ref, orig))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--no-accuracy", action="store_true")
    ap.add_argument("--rounds", type=int, default=3)
    ap.add_argument("--steps", type=int, default=6)
    ap.add_argument("--only", nargs="*", default=None)
    a = ap.parse_args()
    cands = {k: v for k, v in CANDIDATES.items() if a.only is None or k in a.only or k == "f16 (default)"}
    dev = "cuda:0"
    torch.set_num_threads(16)
    for preset in ("trained_like", "random_init"):
        sds = (synth.unet_state_dict(0, preset=preset), synth.codec_state_dict(0, preset=preset, scale_bias=-0.2 if preset == "trained_like" else None),
               synth.vae_state_dict(0, preset=preset))
        models = {}
        for name, cfg in cands.items():
            try:
                m = SD15_1step_codec_stage1(dict({"vae_attn_patch": 16}, **cfg), device=dev).load_state_dicts(*sds)
                m.codec_model.update(force=True)
                models[name] = m
            except Exception as ex:                                # noqa: BLE001
                print(f"[{preset}] {name}: not built ({type(ex).__name__}: {ex})", flush=True)
        base = models["f16 (default)"]
        if not a.no_accuracy:
            idx = torch.from_numpy(np.random.default_rng(177).integers(0, 16384, (1, 12, 12)).astype(np.int32)).to(dev)
            blobs, y_hat = base.codec_model.encode_latents(idx, None, noise_seed=178)
            with torch.no_grad():
                _, z_ref = R.z_indices_to_params(sds[1], idx.cpu())
                ref = R.to_unit_range(R.decode_from_latents(sds[1], R.fold_lora(sds[0]), sds[2], y_hat.torch_nchw().cpu(), z_ref)["image"])
            for name, m in models.items():
                img = m.decode(stream=blobs, unit_range=True).cpu()
                print(f"[{preset}] {name:22s} PSNR(product, oracle) {R.psnr(img, ref):6.2f} dB   |dPSNR| at 25 / 27 dB "
                      f"{delta_psnr(img, ref, 25.0, 1000):.4f} / {delta_psnr(img, ref, 27.0, 1000):.4f} dB", flush=True)
        if preset == "trained_like" and a.rounds:
            idx = torch.from_numpy(np.random.default_rng(1000).integers(0, 16384, (16, 12, 12)).astype(np.int32)).to(dev)
            blobs, _ = base.codec_model.encode_latents(idx, None, noise_seed=2)
            pipes = {n: m.pipeline() for n, m in models.items()}
            times = {n: [] for n in models}
            for r in range(a.rounds + 1):
                for n in (list(pipes) if r % 2 == 0 else list(pipes)[::-1]):
                    torch.cuda.synchronize()
                    t0 = time.perf_counter()
                    pipes[n].map([blobs] * a.steps)
                    torch.cuda.synchronize()
                    if r:
                        times[n].append(16 * a.steps / (time.perf_counter() - t0))
            for n, p in pipes.items():
                p.close()
                print(f"[speed] {n:22s} images/s per round: " + " ".join(f"{v:7.2f}" for v in times[n]) +
                      f"   median {float(np.median(times[n])):7.2f}", flush=True)
        del models
        torch.cuda.empty_cache()


if __name__ == "__main__":
    main()
