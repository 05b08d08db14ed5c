"""Probe: pictures far above the benchmarked size (decode only; properties, and the oracle for one ragged size)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from onedc_amd import synth
from onedc_amd.model_sd15_with_codec_stage1 import SD15_1step_codec_stage1
from oracle import torch_ref as R
DEV = "cuda:0"
t = "trained_like"
m = SD15_1step_codec_stage1({"vae_attn_patch": 16}).load_synthetic(0, preset=t)
m.codec_model.update(force=True)
for (B, zh, zw, oracle) in [(1, 32, 32, False), (2, 24, 40, False), (1, 31, 17, True), (1, 48, 48, False)]:
    idx = torch.from_numpy(np.random.default_rng(zh).integers(0, 16384, (B, zh, zw)).astype(np.int32)).to(DEV)
    try:
        torch.cuda.reset_peak_memory_stats()
        t0 = time.time()
        blobs, y_hat = m.codec_model.encode_latents(idx, None, noise_seed=5)
        img = m.decode(stream=blobs, unit_range=True)
        torch.cuda.synchronize()
        dt = time.time() - t0
        alone = m.decode(stream=blobs[:1], unit_range=True)
        msg = f"{B} x {zh * 64}x{zw * 64}: finite {bool(torch.isfinite(img).all())}, alone == in batch {bool(torch.equal(alone[0], img[0]))}, {dt:.2f} s, peak HBM {torch.cuda.max_memory_allocated() / 1e9:.1f} GB"
        if oracle:
            codec_sd, unet_sd, vae_sd = synth.codec_state_dict(0, preset=t), R.fold_lora(synth.unet_state_dict(0, preset=t)), synth.vae_state_dict(0, preset=t)
            torch.set_num_threads(32)
            with torch.no_grad():
                _, z_sem = R.z_indices_to_params(codec_sd, idx.cpu())
                ref = R.to_unit_range(R.decode_from_latents(codec_sd, unet_sd, vae_sd, y_hat.torch_nchw().cpu(), z_sem)["image"])
            msg += f", PSNR vs oracle {R.psnr(img.cpu(), ref):.2f} dB"
        print(msg, flush=True)
        del img, alone
        torch.cuda.empty_cache()
    except Exception as e:
        import traceback
        print(f"{B} x {zh * 64}x{zw * 64}: FAILED {type(e).__name__}: {e}", flush=True)
        traceback.print_exc(limit=6)
