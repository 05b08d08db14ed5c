"""Deterministic synthetic weights in checkpoint naming (no checkpoints exist offline).

Every tensor is a pure function of (seed, key, shape): a torch CPU generator seeded with
crc32(key) ^ seed draws it, so the CPU oracle, the GPU path and any rank produce the same
weights without shipping ~1e9 floats.  Scales follow 1/sqrt(fan_in) so activations stay O(1)
through ~60 layers; norm gains sit near 1; LoRA factors are small but non-zero so that the
fold W += (alpha/r) B A is exercised.
"""
import math
import zlib

import torch

from . import arch


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


def codec_state_dict(seed=0, scale_bias=1.0):
    """Codec decode-side weights.  The last 1x1 of the two nets that emit (scales, means) gets
    +scale_bias on its `scales` half so synthetic sigmas land inside the table range
    (0.11..64, reference entropy_models.py:262-271) and streams cost a realistic ~2 bit/symbol."""
    sd = make_state_dict(arch.codec_decode_spec(), seed)
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


def unet_state_dict(seed=0, lora=True):
    return make_state_dict(arch.unet_spec(lora=lora), seed)


def vae_state_dict(seed=0):
    return make_state_dict(arch.vae_decoder_spec(), seed)
