#!/usr/bin/env python3
"""First contact with a real checkpoint: everything one wants to know before trusting a decode, in one command.

    python tools/first_contact.py --checkpoint_path CKPT [--vae_path vae.safetensors] [--config_path cfg.yaml]
                                  [--bin picture.bin | --size 768] [--out report.json] [--cpu-only]

No OneDC checkpoint exists offline (reference readme.md:64 links a OneDrive folder; the SD-2.1 VAE comes from the HF hub,
src/models/sd15_onedc_codec_stage1/model_sd15_with_codec_stage1.py:43-45), so every parity number of this repository was
measured on synthetic weights and IEEE half is the default 16-bit format on the ASSUMPTION that a trained model's
activations fit its range.  This script checks that assumption the day the files exist:

  1. strict key / shape contract of the three files (the loader contract of src/inference.py:87-93: model.safetensors =
     UNet + LoRA + conv_in + vae_reduction, model_1.safetensors = codec; plus the VAE file) -- on the CPU, nothing is built
  2. weight magnitudes after the LoRA fold against the half range (+-65504), including the conv_shortcut weights that ride
     on conv2 divided by the VAE's 2^-6 stream scale -- on the CPU
  3. one picture through the HIP decoder in f16 AND bf16 (a .bin the reference or this package wrote, or a synthetic stream
     from the model's own prior): abs-max of every conv / linear output AS STORED (the VAE stream times 2^-6) against 65504,
     rms / abs-max of the stage tensors (y_hat, h_detail, eps, reduced, x0, picture), non-finite flags
  4. PSNR between the two pictures, and the recommendation: `act_dtype: f16` (default), or `bf16` when a layer comes within
     a factor 8 of the half range or the f16 picture is not finite
  5. (round 6) the bf16-OPERAND switch on THIS checkpoint: the f16 model's resnet convs multiply bf16 operands by default since
     round 5 (kernels.BF16_OPERANDS, + 2.3 % images/s), justified by |dPSNR| 0.0003 dB measured on synthetic weights only.  The same
     stream is decoded with the switch on and off; PSNR between the two f32 pictures P gives the |dPSNR| it can cost a picture
     that sits Q dB from its original: 4.34 x 10^((Q - P) / 10) dB when the two errors are uncorrelated (measured on the synthetic
     points: P = 63.6 dB -> 0.0010 dB at Q = 27), and with --original the measured figure itself.  More than
     half the 0.01 dB budget at Q = 30 dB -> the report recommends ONEDC_BF16_OPERANDS=0.

Exit status 0 = the files load and both decodes are finite; the JSON report says the rest.
"""
import argparse
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

HALF_MAX = 65504.0
HEADROOM_WANTED = 8.0          # recommend bf16 when a stored activation comes within this factor of the half range
DPSNR_BUDGET = 0.01            # dB (BASELINE.json north_star)


def dpsnr_uncorrelated(p_between, q):
    """|dPSNR| (dB) of a picture Q dB from its original when an independent error P dB below full scale is added to it."""
    return 4.3429 * 10.0 ** ((q - p_between) / 10.0)


def bf16_operand_check(model, blobs, original=None):
    """Step 5: the same stream through the f16 model with bf16 operands on and off (a run-time switch: both weight packings exist).
    -> report dict.  original: optional [1,3,H,W] tensor in [0,1]."""
    from onedc_amd import kernels as K
    from onedc_amd.quality import psnr
    pics = {}
    keep = K.BF16_OPERANDS
    try:
        for on in (True, False):
            K.BF16_OPERANDS = on
            pics[on] = model.decode(stream=blobs, unit_range=True).float().clamp(0, 1).cpu()
    finally:
        K.BF16_OPERANDS = keep
    p = float(psnr(pics[True], pics[False]))
    rep = {"psnr_between_db": p, "dpsnr_uncorrelated_db": {str(q): dpsnr_uncorrelated(p, q) for q in (25, 27, 30, 35)},
           "built_with_bf16_operand_weights": bool(keep)}
    if original is not None:
        q_on, q_off = float(psnr(pics[True], original)), float(psnr(pics[False], original))
        rep["psnr_vs_original_db"] = {"bf16_operands": q_on, "f16_operands": q_off, "abs_delta": abs(q_on - q_off)}
        cost = abs(q_on - q_off)
    else:
        cost = rep["dpsnr_uncorrelated_db"]["30"]
    rep["recommendation"] = ("ONEDC_BF16_OPERANDS=1 (default)" if cost <= DPSNR_BUDGET / 2 else
                             f"ONEDC_BF16_OPERANDS=0: the switch costs {cost:.4f} dB on this checkpoint, more than half the "
                             f"{DPSNR_BUDGET} dB budget")
    return rep


def weight_table(sd, top=8):
    """abs-max of every floating tensor of a state dict -> (global max, the `top` largest as (key, max))."""
    rows = [(k, float(v.abs().max())) for k, v in sd.items() if torch.is_floating_point(v) and v.numel()]
    rows.sort(key=lambda r: -r[1])
    return (rows[0][1] if rows else 0.0), rows[:top]


def check_files(checkpoint_path, vae_path=None):
    """Step 1 + 2 on the CPU.  -> (report dict, (unet_sd, codec_sd, vae_sd))"""
    from onedc_amd import pack
    from onedc_amd.model_sd15_with_codec_stage1 import check_state_dicts, load_safetensor
    from onedc_amd.vae import F16_STREAM_SCALE
    ff = load_safetensor(os.path.join(checkpoint_path, "model.safetensors"))
    codec = load_safetensor(os.path.join(checkpoint_path, "model_1.safetensors"))
    vae_path = vae_path or os.path.join(checkpoint_path, "vae", "diffusion_pytorch_model.safetensors")
    vae = load_safetensor(vae_path)
    rep = {"files": {"model.safetensors": len(ff), "model_1.safetensors": len(codec), os.path.basename(vae_path): len(vae)},
           "parameters": {"unet_file": sum(v.numel() for v in ff.values()), "codec_file": sum(v.numel() for v in codec.values()),
                          "vae_file": sum(v.numel() for v in vae.values())}}
    try:
        check_state_dicts(ff, codec, vae)
        rep["strict_load"] = "ok"
    except RuntimeError as e:
        rep["strict_load"] = str(e).splitlines()
    lora = any(".lora_A." in k for k in ff)
    folded = pack.fold_lora(ff, 64, 8.0) if lora else ff
    rep["lora_tensors_present"] = lora
    weights = {}
    for name, sd in (("unet (LoRA folded)", folded), ("codec", codec), ("vae", vae)):
        mx, top = weight_table(sd)
        weights[name] = {"abs_max": mx, "largest": [{"key": k, "abs_max": m} for k, m in top], "fits_half": mx < HALF_MAX}
    # the VAE's conv_shortcut weights are packed divided by the stream scale when they ride on conv2 (layers.ResnetBlock2D)
    sc = [(k, float(v.abs().max()) / F16_STREAM_SCALE) for k, v in vae.items() if k.endswith("conv_shortcut.weight")]
    weights["vae conv_shortcut / stream scale"] = {"abs_max": max((m for _, m in sc), default=0.0),
                                                   "fits_half": all(m < HALF_MAX for _, m in sc)}
    rep["weights"] = weights
    return rep, (ff, codec, vae)


def stage_stats(view):
    t = view.torch_nchw().float()
    return {"rms": float(t.pow(2).mean().sqrt()), "abs_max": float(t.abs().max()), "finite": bool(torch.isfinite(t).all())}


def decode_instrumented(model, blobs):
    """One decode with the abs-max of every conv / linear output recorded as STORED.  -> (picture u8 tensor, report dict)"""
    from onedc_amd import kernels as K
    from onedc_amd import layers as L
    records = []
    orig = K.conv_gemm

    def wrapped(x, w, bias, y, **kw):
        r = orig(x, w, bias, y, **kw)
        if r is not None:
            n = y.B * y.H * y.W
            stored = y.t[:n * y.cs].view(n, y.cs)[:, y.co:y.co + y.C]
            records.append((f"k{kw.get('KH', 1)} {x.H}x{x.W} {kw.get('Cin') or x.C}->{kw.get('Cout') or y.C}"
                            + (" f32" if y.f32 else ""), float(stored.float().abs().max()), bool(y.f32)))
        return r
    K.conv_gemm = L.K.conv_gemm = wrapped
    try:
        with K.act_format(model.act), torch.no_grad(), torch.cuda.device(model.device):
            ent = model.codec_model.decode_entropy(stream=blobs)
            y_hat, z_sem = ent[0], ent[1]
            h_detail, y_semantic = model.codec_model.synthesis(y_hat, z_sem)
            out = model.decode_latents(h_detail, y_semantic)
            H, W = ent[2][0]
            pic, flags = model._image_out(out["image"], H, W, False, "u8")
            torch.cuda.synchronize()
    finally:
        K.conv_gemm = L.K.conv_gemm = orig
    sixteen = [(n, m) for n, m, f32 in records if not f32]
    worst = sorted(sixteen, key=lambda r: -r[1])[:10]
    mx = max((m for _, m in sixteen), default=0.0)
    rep = {"act_dtype": model.act, "conv_launches": len(records),
           "stored_16bit_abs_max": mx, "headroom_to_half_max": (HALF_MAX / mx) if mx > 0 else float("inf"),
           "largest_stored_outputs": [{"layer": n, "abs_max": m} for n, m in worst],
           "stages": {"y_hat": stage_stats(y_hat), "h_detail": stage_stats(h_detail), "y_semantic": stage_stats(y_semantic),
                      "eps": stage_stats(out["eps"]), "reduced": stage_stats(out["reduced"]), "x0": stage_stats(out["x0"]),
                      "image": stage_stats(out["image"])},
           "nonfinite_pictures": torch.nonzero(flags).flatten().tolist()}
    return pic, rep


def main(argv=None):
    ap = argparse.ArgumentParser(description=__doc__.split("\n")[0])
    ap.add_argument("--checkpoint_path", required=True)
    ap.add_argument("--vae_path", default=None)
    ap.add_argument("--config_path", default=None)
    ap.add_argument("--bin", default=None, help="a container to decode (default: a synthetic stream from the model's own prior)")
    ap.add_argument("--size", type=int, default=768, help="picture size of the synthetic stream")
    ap.add_argument("--out", default=None, help="write the JSON report here (default: stdout only)")
    ap.add_argument("--cpu-only", action="store_true", help="steps 1 and 2 only")
    ap.add_argument("--original", default=None, help="the picture --bin was coded from (PNG / JPEG): step 5 then reports the measured "
                                                     "|dPSNR| of the bf16-operand switch instead of the uncorrelated estimate")
    args = ap.parse_args(argv)
    report, sds = check_files(args.checkpoint_path, args.vae_path)
    ok = report["strict_load"] == "ok"
    if not args.cpu_only and ok:
        if not torch.cuda.is_available():
            report["decode"] = "skipped: no GPU (run with --cpu-only to silence)"
        else:
            import numpy as np
            import yaml
            from onedc_amd.model_sd15_with_codec_stage1 import SD15_1step_codec_stage1
            from onedc_amd.quality import psnr
            cfg = {}
            if args.config_path:
                with open(args.config_path) as f:
                    cfg = yaml.safe_load(f) or {}
            cfg.setdefault("vae_attn_patch", 16)
            pics, decodes = {}, {}
            blobs = None
            for fmt in ("f16", "bf16"):
                c = dict(cfg, act_dtype=fmt)
                model = SD15_1step_codec_stage1(c, device="cuda:0").load_state_dicts(*sds)
                model.codec_model.update(force=True)
                model.check_finite = False
                if blobs is None:
                    if args.bin:
                        with open(args.bin, "rb") as f:
                            blobs = [f.read()]
                    else:
                        zh = args.size // 64
                        idx = torch.from_numpy(np.random.default_rng(0).integers(0, 16384, (1, zh, zh)).astype(np.int32)).cuda()
                        blobs, _ = model.codec_model.encode_latents(idx, None, noise_seed=1)
                    report["stream"] = {"bytes": len(blobs[0]), "source": args.bin or f"synthetic, {args.size}x{args.size}"}
                pics[fmt], decodes[fmt] = decode_instrumented(model, blobs)
                if fmt == "f16":
                    original = None
                    if args.original:
                        from PIL import Image
                        original = torch.from_numpy(np.asarray(Image.open(args.original).convert("RGB"))).permute(2, 0, 1)[None].float() / 255
                    model.check_finite = True
                    try:
                        report["bf16_operands"] = bf16_operand_check(model, blobs, original)
                    except Exception as ex:                        # noqa: BLE001 -- (a non-finite f16 decode is reported by step 3 / 4)
                        report["bf16_operands"] = {"error": f"{type(ex).__name__}: {ex}"}
                del model
                torch.cuda.empty_cache()
            report["decode"] = decodes
            a, b = (pics[f].float().div(255).permute(0, 3, 1, 2) for f in ("f16", "bf16"))
            report["psnr_f16_vs_bf16_db"] = float(psnr(a, b))
            f16 = decodes["f16"]
            finite = not f16["nonfinite_pictures"] and all(s["finite"] for s in f16["stages"].values())
            if not finite:
                rec, why = "bf16", "the f16 decode of this picture is not finite"
            elif f16["headroom_to_half_max"] < HEADROOM_WANTED:
                rec, why = "bf16", (f"a stored f16 activation reaches {f16['stored_16bit_abs_max']:.3g}: only "
                                    f"{f16['headroom_to_half_max']:.1f}x below the half range (wanted: {HEADROOM_WANTED:.0f}x); or keep "
                                    f"f16 with f16_range_fallback: true")
            else:
                rec, why = "f16", (f"largest stored f16 activation {f16['stored_16bit_abs_max']:.3g} "
                                   f"({f16['headroom_to_half_max']:.0f}x headroom), picture finite")
            report["recommendation"] = {"act_dtype": rec, "why": why}
            ok = ok and not decodes["bf16"]["nonfinite_pictures"]
    text = json.dumps(report, indent=1)
    print(text)
    if args.out:
        with open(args.out, "w") as f:
            f.write(text)
    return 0 if ok else 1


if __name__ == "__main__":
    sys.exit(main())
