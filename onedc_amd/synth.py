"""Deterministic synthetic weights in checkpoint naming (no checkpoints exist offline).

Every tensor is a pure function of (seed, key, shape): a torch CPU generator seeded with
crc32(key) ^ seed draws it, so the CPU oracle, the GPU path and any rank produce the same
weights without shipping ~1e9 floats.  Scales follow 1/sqrt(fan_in) so activations stay O(1)
through ~60 layers; norm gains sit near 1; LoRA factors are small but non-zero so that the
fold W += (alpha/r) B A is exercised.

Two presets (`preset=`):

  "random_init"   the plain 1/sqrt(fan_in) draw.  Every un-normalised residual add of the codec's DepthConvBlock4 stacks
                  doubles the variance, so h_detail comes out ~3e3, x0 ~5e4 (a trained model: ~1) and the four-part
                  prior's (scales, means) grow ~4x per hop until most scales leave the 0.11..64 table.
  "trained_like"  the same draw with the OUTPUT convolutions of the residual branches damped (DepthConvBlock4 conv2 /
                  conv_out, ResnetBlock conv2, attention out-projections, FF out) and the three heads that form x0 scaled,
                  so that the operating point resembles a trained checkpoint's: x0 rms ~1 (the SD-VAE decoder's residual
                  branches then matter against its stream, which is the regime that stresses f16 rounding and range), the
                  prior nets pass a perturbation of y_hat on with a gain of about 1 per hop instead of growing, and the
                  scales stay inside the Gaussian table.  Which tensors, by how much: TRAINED_LIKE_RULES below.
"""
import math
import re
import zlib

import torch

from . import arch

PRESETS = ("random_init", "trained_like")

# (regex on the checkpoint key, factor) -- first match wins; applied to weights AND biases of the matched layer
# (for LoRA'd layers: base_layer.* and lora_B, so that the folded W + (alpha/r) B A scales as a whole).
_BR = 0.3          # residual-branch output gain of the un-normalised codec blocks
_BRN = 0.5         # ... of the blocks that start with a GroupNorm / LayerNorm
TRAINED_LIKE_RULES = {
    "codec": (
        (r"\.block\.0\.conv2\.", _BR), (r"\.block\.1\.conv_out\.", _BR),          # DepthConvBlock4 (dcvc.py:242-265, 353-367)
        (r"(?<!block\.1)\.(conv|upsample\.0)\.(weight|bias)$", 0.7),                 # ResidualBlockUpsample: two summed branches
        (r"(blocks|to_semantic|trans_coding)\.\d+\.conv2\.", _BRN),                  # vqgan ResnetBlock
        (r"\.proj_out\.", _BRN),                                                     # vqgan AttnBlock
    ),
    "unet": (
        (r"^conv_out\.", 0.04), (r"^vae_reduction\.(blocks\.5|short_cut)\.", 0.04),  # the heads of x0 = (reduced - .998 eps)/.068
        (r"resnets\.\d+\.conv2\.", _BRN), (r"\.to_out\.0\.", _BRN), (r"\.ff\.net\.2\.", _BRN), (r"\.proj_out\.", _BRN),
    ),
    # SD-VAE decoders carry a residual stream in the hundreds (thousands in places): conv_in lifts the stream there and
    # the branch outputs are raised with it, so that a branch still is ~7 % of the stream it joins (as with x 0.5 on rms 7)
    "vae": (
        (r"^decoder\.conv_in\.", 8.0), (r"resnets\.\d+\.conv2\.", 4.0), (r"\.to_out\.0\.", 4.0),
    ),
}


def _apply_preset(sd, family, preset):
    if preset in (None, "random_init"):
        return sd
    if preset != "trained_like":
        raise ValueError(f"unknown synthetic-weight preset {preset!r} (known: {PRESETS})")
    rules = [(re.compile(pat), f) for pat, f in TRAINED_LIKE_RULES[family]]
    for k in sd:
        if ".lora_A." in k:
            continue
        for pat, f in rules:
            if pat.search(k):
                sd[k] = sd[k] * f
                break
    return sd


def _gen(key, seed):
    g = torch.Generator(device="cpu")
    g.manual_seed((zlib.crc32(key.encode()) ^ (seed * 0x9E3779B1)) & 0x7FFFFFFF)
    return g


def make_tensor(key, shape, seed=0, dtype=torch.float32):
    g = _gen(key, seed)
    leaf = key.rsplit(".", 1)[-1]
    if "lora_A" in key:
        fan_in = math.prod(shape[1:])
        return torch.randn(shape, generator=g, dtype=dtype) * (0.5 / math.sqrt(fan_in))
    if "lora_B" in key:
        return torch.randn(shape, generator=g, dtype=dtype) * (0.5 / math.sqrt(shape[1]))
    if leaf == "bias":
        return torch.randn(shape, generator=g, dtype=dtype) * 0.05
    if len(shape) == 1:  # norm gains
        return 1.0 + 0.1 * torch.randn(shape, generator=g, dtype=dtype)
    fan_in = math.prod(shape[1:])
    return torch.randn(shape, generator=g, dtype=dtype) * (1.0 / math.sqrt(fan_in))


def make_state_dict(spec, seed=0, dtype=torch.float32):
    return {k: make_tensor(k, tuple(s), seed, dtype) for k, s in spec.items()}


def codec_state_dict(seed=0, scale_bias=None, preset=None):
    """Codec decode-side weights.  The last 1x1 of the two nets that emit (scales, means) gets
    +scale_bias on its `scales` half so synthetic sigmas land inside the table range
    (0.11..64, reference entropy_models.py:262-271) and streams cost a realistic ~2 bit/symbol."""
    sd = _apply_preset(make_state_dict(arch.codec_decode_spec(), seed), "codec", preset)
    if scale_bias is None:
        scale_bias = 1.0 if preset in (None, "random_init") else 1.5
    for k in ("y_prior_fusion.1.block.1.conv_out.bias", "y_spatial_prior.2.block.1.conv_out.bias"):
        n = sd[k].numel() // 2
        sd[k][:n] += scale_bias
    return sd


def hyper_enc_state_dict(seed=0):
    """HyperEncoder weights (encode side).  The last 1x1 gets a x3 gain so the synthetic z spreads over
    all four FSQ levels per dimension instead of sitting at the two central ones."""
    sd = make_state_dict(arch.hyper_enc_spec(), seed)
    sd["hyper_enc.fusion.5.weight"] *= 3.0
    return sd


def codec_encoder_state_dict(seed=0):
    return make_state_dict(arch.codec_encoder_spec(), seed)


def vae_encoder_state_dict(seed=0):
    return make_state_dict(arch.vae_encoder_spec(), seed)


def unet_state_dict(seed=0, lora=True, preset=None):
    return _apply_preset(make_state_dict(arch.unet_spec(lora=lora), seed), "unet", preset)


def vae_state_dict(seed=0, preset=None):
    return _apply_preset(make_state_dict(arch.vae_decoder_spec(), seed), "vae", preset)
