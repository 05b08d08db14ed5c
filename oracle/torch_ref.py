"""oracle/torch_ref.py — TEST INFRASTRUCTURE, NOT PRODUCT CODE.

fp32 torch-CPU restatement of the reference's decode path (and of the encode-side prior pass
that is needed to create valid bitstreams), written functionally over a flat state dict in
checkpoint naming.  Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may
import this module; the product (onedc_amd/) never does.

Parity status
  * codec-side functions (DCB4, RBU, ResnetBlock, AttnBlock, HyperDecoder, prior nets, four-part
    prior, SemanticAdaptor, Decoder, container, FSQ packing): PINNED — checked against outputs of
    the reference's own Python modules imported in the build container with synthetic weights
    (tests/golden/make_golden.py -> tests/golden/*.npz) and, for the entropy coder, against the
    reference's C++ compiled into oracle/_ref.
  * SD1.5 UNet body, SD-VAE decoder, peft LoRA fold, FSQ code book: PARITY UNPINNED.  Their
    arithmetic lives in third-party packages that are neither vendored in the reference nor
    installed here (diffusers==0.32.1, peft==0.14.0, vector-quantize-pytorch==1.21.2:
    reference requirements.txt:4,21,29).  They are restated from the published architecture of
    those packages and anchored on the reference's call sites (decoder_unet.py:32-305, 331-408;
    model_sd15_with_codec_stage1.py:43-48, 103-106, 184-188; autoencoders_patch_attn.py:9-80;
    codec_module.py:202, 431).

All file:line citations are relative to the reference repository root, `src/` omitted.
"""
import math

import numpy as np
import torch
import torch.nn.functional as F


# ------------------------------------------------------------------ small helpers
def conv(sd, p, x, stride=1, padding=0, groups=1):
    return F.conv2d(x, sd[f"{p}.weight"], sd.get(f"{p}.bias"), stride=stride, padding=padding,
                    groups=groups)


def linear(sd, p, x):
    return F.linear(x, sd[f"{p}.weight"], sd.get(f"{p}.bias"))


def gn(sd, p, x, eps, groups=32):
    return F.group_norm(x, groups, sd[f"{p}.weight"], sd[f"{p}.bias"], eps)


def ln(sd, p, x, eps=1e-5):
    return F.layer_norm(x, (x.shape[-1],), sd[f"{p}.weight"], sd[f"{p}.bias"], eps)


# ------------------------------------------------------------------ codec building blocks
def depth_conv_block4(sd, p, x):
    """modules/dcvc.py:424-433 = DepthConv :242-265 then ConvFFN3 :353-367."""
    c = x.shape[1]
    identity = conv(sd, f"{p}.block.0.adaptor", x) if f"{p}.block.0.adaptor.weight" in sd else x
    out = F.leaky_relu(conv(sd, f"{p}.block.0.conv1.0", x), 0.01)
    out = conv(sd, f"{p}.block.0.depth_conv", out, padding=1, groups=c)
    out = conv(sd, f"{p}.block.0.conv2", out)
    x = out + identity
    x1, x2 = conv(sd, f"{p}.block.1.conv", x).chunk(2, 1)
    out = F.leaky_relu(x1, 0.1) + F.leaky_relu(x2, 0.01)
    return x + conv(sd, f"{p}.block.1.conv_out", out)


def residual_block_upsample(sd, p, x):
    """modules/dcvc.py:183-208."""
    out = F.pixel_shuffle(conv(sd, f"{p}.subpel_conv.0", x), 2)
    out = F.leaky_relu(out, 0.01)
    out = conv(sd, f"{p}.conv", out, padding=1)
    out = F.leaky_relu(out, 0.1)
    identity = F.pixel_shuffle(conv(sd, f"{p}.upsample.0", x), 2)
    return out + identity


def vq_resnet_block(sd, p, x):
    """modules/vqgan/blocks.py:15-52 (in == out on every decode-path use, so :49-50 is dead)."""
    h = conv(sd, f"{p}.conv1", F.silu(gn(sd, f"{p}.norm1", x, 1e-6)), padding=1)
    h = conv(sd, f"{p}.conv2", F.silu(gn(sd, f"{p}.norm2", h, 1e-6)), padding=1)
    return h + x


def vq_attn_block(sd, p, x):
    """modules/vqgan/blocks.py:55-106 — single head over all H*W positions."""
    h_ = gn(sd, f"{p}.norm", x, 1e-6)
    q, k, v = conv(sd, f"{p}.q", h_), conv(sd, f"{p}.k", h_), conv(sd, f"{p}.v", h_)
    b, c, h, w = q.shape
    q = q.reshape(b, c, h * w).permute(0, 2, 1)
    k = k.reshape(b, c, h * w)
    w_ = torch.softmax(torch.bmm(q, k) * (int(c) ** (-0.5)), dim=2)
    v = v.reshape(b, c, h * w)
    h_ = torch.bmm(v, w_.permute(0, 2, 1)).reshape(b, c, h, w)
    return x + conv(sd, f"{p}.proj_out", h_)


# ------------------------------------------------------------------ codec nets
def fsq_indices_to_codes(indices, levels=(4,) * 7):
    """vector-quantize-pytorch FSQ.indices_to_codes for even level counts (call site
    codec_module.py:431): digit_d = (idx // prod(levels[:d])) % L_d ; code = (digit - L//2)/(L//2).
    indices (B,H,W) int -> (B,len(levels),H,W) float.  PARITY UNPINNED (third-party)."""
    idx = indices.long()
    codes, basis = [], 1
    for L in levels:
        digit = (idx // basis) % L
        codes.append((digit.float() - (L // 2)) / (L // 2))
        basis *= L
    return torch.stack(codes, dim=1)


def fsq_quantize(z, levels=(4,) * 7, eps=1e-3):
    """vector-quantize-pytorch 1.21.2 FSQ.forward for dim == len(levels) (no projections; call sites
    codec_module.py:235, :301, :379): bound -> round -> codes and their mixed-radix index.
        half_l = (L-1)(1+eps)/2; offset = 0.5 for even L; shift = atanh(offset/half_l)
        q = round(tanh(z + shift) * half_l - offset);  code = q / (L//2);  digit = q + L//2
    z (B,len(levels),H,W) float -> (z_hat (B,D,H,W) float32, indices (B,H,W) int32).
    PARITY UNPINNED (third-party formula restated)."""
    z = z.float()
    codes, index, basis = [], 0, 1
    for d, L in enumerate(levels):
        half_l = (L - 1) * (1 + eps) / 2
        offset = 0.5 if L % 2 == 0 else 0.0
        shift = math.atanh(offset / half_l)
        q = torch.round(torch.tanh(z[:, d] + shift) * half_l - offset)
        codes.append(q / (L // 2))
        index = index + (q + (L // 2)).to(torch.int64) * basis
        basis *= L
    return torch.stack(codes, 1), index.to(torch.int32)


def hyper_encoder(sd, y, sem, p="hyper_enc"):
    """codec_module.py:119-142: z = fusion(cat(y_trans_coding(y), sem)).  y (B,128,H/16,W/16),
    sem (B,768,H/64,W/64) -> z (B,7,H/64,W/64)."""
    z = depth_conv_block4(sd, f"{p}.y_trans_coding.0", y)
    z = conv(sd, f"{p}.y_trans_coding.1", z, stride=2, padding=1)
    z = depth_conv_block4(sd, f"{p}.y_trans_coding.2", z)
    z = conv(sd, f"{p}.y_trans_coding.3", z, stride=2, padding=1)
    z = torch.cat([z, sem], 1)
    z = depth_conv_block4(sd, f"{p}.fusion.0", z)
    z = vq_attn_block(sd, f"{p}.fusion.1", z)
    z = depth_conv_block4(sd, f"{p}.fusion.2", z)
    z = vq_attn_block(sd, f"{p}.fusion.3", z)
    z = depth_conv_block4(sd, f"{p}.fusion.4", z)
    return conv(sd, f"{p}.fusion.5", z)


def hyper_decoder(sd, z_hat, p="hyper_dec"):
    """codec_module.py:145-166 -> (params, z_semantic)."""
    z = F.leaky_relu(conv(sd, f"{p}.feat_in.0", z_hat), 0.01)
    h = depth_conv_block4(sd, f"{p}.to_entropy.0", z)
    h = residual_block_upsample(sd, f"{p}.to_entropy.1", h)
    h = depth_conv_block4(sd, f"{p}.to_entropy.2", h)
    h = residual_block_upsample(sd, f"{p}.to_entropy.3", h)
    h = depth_conv_block4(sd, f"{p}.to_entropy.4", h)
    return h, z


def y_prior_fusion(sd, params):
    """codec_module.py:205-208."""
    return depth_conv_block4(sd, "y_prior_fusion.1", depth_conv_block4(sd, "y_prior_fusion.0", params))


def y_spatial_prior(sd, x):
    for i in range(3):
        x = depth_conv_block4(sd, f"y_spatial_prior.{i}", x)
    return x


def semantic_adaptor(sd, z_semantic, p="semantic_adaptor.to_semantic"):
    """codec_module.py:169-181."""
    x = depth_conv_block4(sd, f"{p}.0", z_semantic)
    x = vq_resnet_block(sd, f"{p}.1", x)
    x = vq_attn_block(sd, f"{p}.2", x)
    x = vq_attn_block(sd, f"{p}.3", x)
    x = vq_resnet_block(sd, f"{p}.4", x)
    x = vq_attn_block(sd, f"{p}.5", x)
    x = vq_attn_block(sd, f"{p}.6", x)
    return depth_conv_block4(sd, f"{p}.7", x)


def codec_decoder(sd, y_hat, sem_hat, p="dec"):
    """codec_module.py:88-116."""
    y = depth_conv_block4(sd, f"{p}.trans_coding.1", depth_conv_block4(sd, f"{p}.trans_coding.0", y_hat))
    for i in range(3):
        y = vq_resnet_block(sd, f"{p}.blocks.{i}", y)
    y = F.pixel_shuffle(conv(sd, f"{p}.blocks.3", y), 2)
    y = conv(sd, f"{p}.blocks.5", y, padding=1)
    for i in (6, 7, 8):
        y = vq_resnet_block(sd, f"{p}.blocks.{i}", y)
    s = residual_block_upsample(sd, f"{p}.sem_up.0", sem_hat)
    s = depth_conv_block4(sd, f"{p}.sem_up.1", s)
    s = residual_block_upsample(sd, f"{p}.sem_up.2", s)
    s = depth_conv_block4(sd, f"{p}.sem_up.3", s)
    s = residual_block_upsample(sd, f"{p}.sem_up.4", s)
    return depth_conv_block4(sd, f"{p}.conv_out", torch.cat([y, s], dim=1))


# ------------------------------------------------------------------ four-part prior
# phase table, modules/entropy/compression_model.py:269-283: step k puts channel group g on the
# 2x2 phase PHASE[k][g]; phases 0..3 = (row%2, col%2) = (0,0),(0,1),(1,0),(1,1) (:241-267)
PHASE = ((0, 1, 2, 3), (3, 2, 1, 0), (2, 3, 0, 1), (1, 0, 3, 2))


def four_part_masks(B, C, H, W, dtype=torch.float32):
    ys = torch.arange(H).view(H, 1) % 2
    xs = torch.arange(W).view(1, W) % 2
    phase = (ys * 2 + xs)                       # (H, W) in 0..3
    masks = []
    for k in range(4):
        m = torch.zeros(B, C, H, W, dtype=dtype)
        q = C // 4
        for g in range(4):
            m[:, g * q:(g + 1) * q] = (phase == PHASE[k][g]).to(dtype)
        masks.append(m)
    return masks


def combine_for_writing(x):
    """compression_model.py:296-301."""
    x0, x1, x2, x3 = x.chunk(4, 1)
    return (x0 + x1) + (x2 + x3)


LOG_SCALE_MIN = math.log(0.11)
LOG_SCALE_STEP = (math.log(64.0) - math.log(0.11)) / 255


def build_indexes(scales):
    """modules/entropy/entropy_models.py:355-362 (skip_thres is None on this path)."""
    scales = torch.maximum(scales, torch.zeros_like(scales) + 1e-5)
    indexes = (torch.log(scales) - LOG_SCALE_MIN) / LOG_SCALE_STEP
    return indexes.clamp_(0, 255).int()


def _prior_step(sd, k, y_hat_so_far, common):
    params = torch.cat((y_hat_so_far, common), dim=1)
    a = depth_conv_block4(sd, f"y_spatial_prior_adaptor_{k}", params)
    return y_spatial_prior(sd, a).chunk(2, 1)


def compress_four_part_prior(sd, y, common_params):
    """compression_model.py:303-367 with write=True (eval mode: quant = round).
    Returns (symbol planes [4 x (B,C/4,H,W)], scale planes [4 x ...], y_hat)."""
    scales, means = common_params.chunk(2, 1)
    common = conv(sd, "y_spatial_prior_reduction", common_params)
    B, C, H, W = y.shape
    masks = four_part_masks(B, C, H, W, y.dtype)
    y_q_w, s_w = [], []
    y_hat_so_far = None
    for k in range(4):
        if k > 0:
            scales, means = _prior_step(sd, k, y_hat_so_far, common)
        m = masks[k]
        scales_hat, means_hat = scales * m, means * m          # process_with_mask :224-239
        y_q = torch.round((y - means_hat) * m)
        y_hat = y_q + means_hat
        y_q_w.append(combine_for_writing(y_q))
        s_w.append(combine_for_writing(scales_hat))
        y_hat_so_far = y_hat if k == 0 else y_hat_so_far + y_hat
    return y_q_w, s_w, y_hat_so_far


def decompress_four_part_prior(sd, common_params, decode_symbols):
    """compression_model.py:369-407.  decode_symbols(k, index_plane int32 (B,C/4,H,W)) must
    return the symbols of step k as a float tensor of the same shape (the rANS hop,
    entropy_models.py:369-374)."""
    scales, means = common_params.chunk(2, 1)
    common = conv(sd, "y_spatial_prior_reduction", common_params)
    B, C, H, W = means.shape
    masks = four_part_masks(B, C, H, W, means.dtype)
    y_hat_so_far = None
    for k in range(4):
        if k > 0:
            scales, means = _prior_step(sd, k, y_hat_so_far, common)
        m = masks[k]
        scales_r = combine_for_writing(scales * m)
        y_q_r = decode_symbols(k, build_indexes(scales_r))
        cur = (torch.cat((y_q_r, y_q_r, y_q_r, y_q_r), dim=1) + means) * m
        y_hat_so_far = cur if k == 0 else y_hat_so_far + cur
    return y_hat_so_far


def recon_with_z_four_part_prior(sd, common_params):
    """z-only model: compression_model.py:421-465 — y_hat_k = means * mask_k (y_q * 0 + means)."""
    scales, means = common_params.chunk(2, 1)
    common = conv(sd, "y_spatial_prior_reduction", common_params)
    B, C, H, W = means.shape
    masks = four_part_masks(B, C, H, W, means.dtype)
    y_hat_so_far = None
    for k in range(4):
        if k > 0:
            scales, means = _prior_step(sd, k, y_hat_so_far, common)
        cur = means * masks[k]
        y_hat_so_far = cur if k == 0 else y_hat_so_far + cur
    return y_hat_so_far


# ------------------------------------------------------------------ LoRA fold (peft, UNPINNED)
def fold_lora(sd, rank=64, alpha=8.0):
    """W = base + (alpha/r) * B @ A for every `<m>.base_layer.weight` (peft LoRA Linear / Conv2d:
    lora_A is (r,in[,k,k]), lora_B is (out,r[,1,1]); call site decoder_unet.py:360-367)."""
    out = {}
    scale = alpha / rank
    for k, v in sd.items():
        if ".lora_A." in k or ".lora_B." in k:
            continue
        if ".base_layer." in k:
            name, leaf = k.split(".base_layer.")
            if leaf == "weight":
                A = sd[f"{name}.lora_A.default.weight"].double()
                Bm = sd[f"{name}.lora_B.default.weight"].double()
                delta = (Bm.reshape(Bm.shape[0], -1) @ A.reshape(A.shape[0], -1)).reshape(v.shape)
                v = (v.double() + scale * delta).float()
            out[f"{name}.{leaf}"] = v
        else:
            out[k] = v
    return out


# ------------------------------------------------------------------ SD1.5 UNet (diffusers, UNPINNED)
def timestep_embedding(t, dim=320, max_period=10000):
    """diffusers get_timestep_embedding with flip_sin_to_cos=True, downscale_freq_shift=0
    (SD1.5 config; call site decoder_unet.py:138)."""
    half = dim // 2
    exponent = -math.log(max_period) * torch.arange(0, half, dtype=torch.float32) / half
    emb = t[:, None].float() * torch.exp(exponent)[None, :]
    return torch.cat([torch.cos(emb), torch.sin(emb)], dim=-1)


def unet_resnet_block(sd, p, x, temb, eps=1e-5):
    h = conv(sd, f"{p}.conv1", F.silu(gn(sd, f"{p}.norm1", x, eps)), padding=1)
    if temb is not None:
        h = h + linear(sd, f"{p}.time_emb_proj", F.silu(temb))[:, :, None, None]
    h = conv(sd, f"{p}.conv2", F.silu(gn(sd, f"{p}.norm2", h, eps)), padding=1)
    if f"{p}.conv_shortcut.weight" in sd:
        x = conv(sd, f"{p}.conv_shortcut", x)
    return x + h


def _mha(sd, p, x, ctx, heads):
    q, k, v = linear(sd, f"{p}.to_q", x), linear(sd, f"{p}.to_k", ctx), linear(sd, f"{p}.to_v", ctx)
    B, N, C = q.shape
    d = C // heads
    q = q.view(B, N, heads, d).transpose(1, 2)
    k = k.view(B, -1, heads, d).transpose(1, 2)
    v = v.view(B, -1, heads, d).transpose(1, 2)
    a = torch.softmax(q @ k.transpose(-1, -2) * (d ** -0.5), dim=-1) @ v
    return linear(sd, f"{p}.to_out.0", a.transpose(1, 2).reshape(B, N, C))


def unet_transformer(sd, p, x, ctx, heads=8):
    """Transformer2DModel (conv projections) around one BasicTransformerBlock with GEGLU."""
    B, C, H, W = x.shape
    res = x
    h = conv(sd, f"{p}.proj_in", gn(sd, f"{p}.norm", x, 1e-6))
    h = h.permute(0, 2, 3, 1).reshape(B, H * W, C)
    t = f"{p}.transformer_blocks.0"
    n = ln(sd, f"{t}.norm1", h)
    h = _mha(sd, f"{t}.attn1", n, n, heads) + h
    h = _mha(sd, f"{t}.attn2", ln(sd, f"{t}.norm2", h), ctx, heads) + h
    n = ln(sd, f"{t}.norm3", h)
    hid, gate = linear(sd, f"{t}.ff.net.0.proj", n).chunk(2, dim=-1)
    h = linear(sd, f"{t}.ff.net.2", hid * F.gelu(gate)) + h
    h = h.reshape(B, H, W, C).permute(0, 3, 1, 2)
    return conv(sd, f"{p}.proj_out", h) + res


def vae_reduction(sd, x, p="vae_reduction"):
    """decoder_unet.py:14-29."""
    h = conv(sd, f"{p}.blocks.2", F.silu(gn(sd, f"{p}.blocks.0", x, 1e-6)), padding=1)
    h = conv(sd, f"{p}.blocks.5", F.silu(gn(sd, f"{p}.blocks.3", h, 1e-6)), padding=1)
    return h + conv(sd, f"{p}.short_cut", x)


def unet_forward(sd, sample, timestep, ctx):
    """decoder_unet.py:32-305 over the SD1.5 topology; `sd` has LoRA already folded.
    Returns (eps, reduced_sample)."""
    reduced = vae_reduction(sd, sample)
    temb = timestep_embedding(timestep)
    temb = linear(sd, "time_embedding.linear_2", F.silu(linear(sd, "time_embedding.linear_1", temb)))
    h = conv(sd, "conv_in", sample, padding=1)
    skips = [h]
    for i in range(4):
        for j in range(2):
            h = unet_resnet_block(sd, f"down_blocks.{i}.resnets.{j}", h, temb)
            if i < 3:
                h = unet_transformer(sd, f"down_blocks.{i}.attentions.{j}", h, ctx)
            skips.append(h)
        if i < 3:
            h = conv(sd, f"down_blocks.{i}.downsamplers.0.conv", h, stride=2, padding=1)
            skips.append(h)
    h = unet_resnet_block(sd, "mid_block.resnets.0", h, temb)
    h = unet_transformer(sd, "mid_block.attentions.0", h, ctx)
    h = unet_resnet_block(sd, "mid_block.resnets.1", h, temb)
    for i in range(4):
        for j in range(3):
            h = torch.cat([h, skips.pop()], dim=1)
            h = unet_resnet_block(sd, f"up_blocks.{i}.resnets.{j}", h, temb)
            if i > 0:
                h = unet_transformer(sd, f"up_blocks.{i}.attentions.{j}", h, ctx)
        if i < 3:
            h = F.interpolate(h, scale_factor=2.0, mode="nearest")
            h = conv(sd, f"up_blocks.{i}.upsamplers.0.conv", h, padding=1)
    h = conv(sd, "conv_out", F.silu(gn(sd, "conv_norm_out", h, 1e-5)), padding=1)
    return h, reduced


# ------------------------------------------------------------------ x0 (dmd/utils.py:279-284)
def ddim_alphas_cumprod(n=1000, beta_start=0.00085, beta_end=0.012):
    """DDIMScheduler 'scaled_linear' schedule of SD1.5 (model_sd15_with_codec_stage1.py:103-106)."""
    betas = torch.linspace(beta_start ** 0.5, beta_end ** 0.5, n, dtype=torch.float32) ** 2
    return torch.cumprod(1.0 - betas, dim=0)


def get_x0_from_noise(sample, model_output, alphas_cumprod, timestep):
    alpha_prod_t = alphas_cumprod[timestep].reshape(-1, 1, 1, 1)
    beta_prod_t = 1 - alpha_prod_t
    return (sample - beta_prod_t ** 0.5 * model_output) / alpha_prod_t ** 0.5


# ------------------------------------------------------------------ SD-VAE decoder (diffusers, UNPINNED)
def vae_attention(sd, p, x):
    """diffusers Attention (1 head, group norm, residual) applied to one window."""
    B, C, H, W = x.shape
    h = x.view(B, C, H * W)
    h = F.group_norm(h, 32, sd[f"{p}.group_norm.weight"], sd[f"{p}.group_norm.bias"], 1e-6)
    h = h.transpose(1, 2)
    q, k, v = linear(sd, f"{p}.to_q", h), linear(sd, f"{p}.to_k", h), linear(sd, f"{p}.to_v", h)
    a = torch.softmax(q @ k.transpose(1, 2) * (C ** -0.5), dim=-1) @ v
    a = linear(sd, f"{p}.to_out.0", a)
    return a.transpose(1, 2).reshape(B, C, H, W) + x


def windowed_attn(sd, p, x, patch):
    """modules/vae/autoencoders_patch_attn.py:9-29, eval branch: the whole attention module
    (its group norm included) runs per patch x patch window."""
    x = x.clone()
    _, _, H, W = x.shape
    for i in range(0, H, patch):
        for j in range(0, W, patch):
            x[:, :, i:i + patch, j:j + patch] = vae_attention(
                sd, p, x[:, :, i:i + patch, j:j + patch].contiguous())
    return x


def vae_decode(sd, latents, attn_patch=16, scaling_factor=0.18215):
    """model_sd15_with_codec_stage1.py:184-188 -> AutoencoderKL.decode with the windowed mid block."""
    z = latents / scaling_factor
    h = conv(sd, "post_quant_conv", z)
    h = conv(sd, "decoder.conv_in", h, padding=1)
    h = unet_resnet_block(sd, "decoder.mid_block.resnets.0", h, None, 1e-6)
    h = windowed_attn(sd, "decoder.mid_block.attentions.0", h, attn_patch)
    h = unet_resnet_block(sd, "decoder.mid_block.resnets.1", h, None, 1e-6)
    for i in range(4):
        for j in range(3):
            h = unet_resnet_block(sd, f"decoder.up_blocks.{i}.resnets.{j}", h, None, 1e-6)
        if i < 3:
            h = F.interpolate(h, scale_factor=2.0, mode="nearest")
            h = conv(sd, f"decoder.up_blocks.{i}.upsamplers.0.conv", h, padding=1)
    h = F.silu(gn(sd, "decoder.conv_norm_out", h, 1e-6))
    return conv(sd, "decoder.conv_out", h, padding=1)


# ------------------------------------------------------------------ encode side: SD-VAE encoder (diffusers, UNPINNED)
def vae_encode(sd, image, attn_patch=16, noise=None, scaling_factor=0.18215):
    """model_sd15_with_codec_stage1.py:191-195: AutoencoderKL.encode(image).latent_dist.sample() * scaling.
    diffusers 0.32.1 Encoder: conv_in, 4 DownEncoderBlock2D (2 resnets, eps 1e-6; Downsample2D = pad
    (0,1,0,1) + conv 3x3 stride 2 on the first three), mid block with the windowed attention patch of
    modules/vae/autoencoders_patch_attn.py, GroupNorm-SiLU-conv_out (8 ch), quant_conv 1x1, then
    DiagonalGaussianDistribution.sample: mean + exp(0.5 clamp(logvar, -30, 20)) * noise.
    `noise` (B,4,h,w) replaces the reference's global-RNG draw (SURVEY hard part 4); None = zeros (the mode)."""
    h = conv(sd, "encoder.conv_in", image, padding=1)
    for i in range(4):
        for j in range(2):
            h = unet_resnet_block(sd, f"encoder.down_blocks.{i}.resnets.{j}", h, None, 1e-6)
        if i < 3:
            h = conv(sd, f"encoder.down_blocks.{i}.downsamplers.0.conv", F.pad(h, (0, 1, 0, 1)), stride=2)
    h = unet_resnet_block(sd, "encoder.mid_block.resnets.0", h, None, 1e-6)
    h = windowed_attn(sd, "encoder.mid_block.attentions.0", h, attn_patch)
    h = unet_resnet_block(sd, "encoder.mid_block.resnets.1", h, None, 1e-6)
    h = conv(sd, "encoder.conv_out", F.silu(gn(sd, "encoder.conv_norm_out", h, 1e-6)), padding=1)
    moments = conv(sd, "quant_conv", h)
    mean, logvar = moments.chunk(2, 1)
    std = torch.exp(0.5 * logvar.clamp(-30.0, 20.0))
    z = mean if noise is None else mean + std * noise
    return z * scaling_factor


# ------------------------------------------------------------------ encode side: analysis transform `enc`
def enc_own_resnet(sd, p, x):
    """encoder_unet.py:14-52 — the reference's own ResnetBlock: the 1x1 `nin_shortcut` is applied to the
    block OUTPUT (hidden_states), not to the input, when in != out (:47-50)."""
    h = conv(sd, f"{p}.conv1", F.silu(gn(sd, f"{p}.norm1", x, 1e-6)), padding=1)
    h = conv(sd, f"{p}.conv2", F.silu(gn(sd, f"{p}.norm2", h, 1e-6)), padding=1)
    res = conv(sd, f"{p}.nin_shortcut", h) if f"{p}.nin_shortcut.weight" in sd else x
    return h + res


def enc_attention(sd, p, x, head_dim=8, eps=1e-5):
    """diffusers Attention of AttnDownBlock2D / AttnUpBlock2D / UNetMidBlock2D in UNet2DModel (UNPINNED):
    GroupNorm(32) over (C, HW), heads = C / 8, softmax(q k^T / sqrt(8)) v, to_out.0, + residual."""
    B, C, H, W = x.shape
    h = F.group_norm(x.view(B, C, H * W), 32, sd[f"{p}.group_norm.weight"], sd[f"{p}.group_norm.bias"], eps)
    h = h.transpose(1, 2)
    heads = C // head_dim
    q, k, v = (linear(sd, f"{p}.{n}", h).view(B, H * W, heads, head_dim).transpose(1, 2)
               for n in ("to_q", "to_k", "to_v"))
    a = torch.softmax(q @ k.transpose(-1, -2) * head_dim ** -0.5, dim=-1) @ v
    a = linear(sd, f"{p}.to_out.0", a.transpose(1, 2).reshape(B, H * W, C))
    return a.transpose(1, 2).reshape(B, C, H, W) + x


def enc_unet_forward(sd, sample, p="enc.unet"):
    """forward_enc (encoder_unet.py:54-134) over the patched UNet2DModel of :137-161 -> (y_sample, z_sample).
    Timestep is the constant 999; Timesteps(512, flip_sin_to_cos=True, shift 0) -> linear_1, SiLU, linear_2."""
    B = sample.shape[0]
    temb = timestep_embedding(torch.full((B,), 999.0), 512)
    temb = linear(sd, f"{p}.time_embedding.linear_2", F.silu(linear(sd, f"{p}.time_embedding.linear_1", temb)))
    h = sample
    for i in range(3):
        h = enc_own_resnet(sd, f"{p}.conv_in.{i}", h)
    h = conv(sd, f"{p}.conv_in.3", h, stride=2, padding=1)
    skips = [h]
    for i in range(3):
        for j in range(2):
            h = unet_resnet_block(sd, f"{p}.down_blocks.{i}.resnets.{j}", h, temb)
            if i < 2:
                h = enc_attention(sd, f"{p}.down_blocks.{i}.attentions.{j}", h)
            skips.append(h)
        if i < 2:
            h = conv(sd, f"{p}.down_blocks.{i}.downsamplers.0.conv", h, stride=2, padding=1)
            skips.append(h)
    h = unet_resnet_block(sd, f"{p}.mid_block.resnets.0", h, temb)
    h = enc_attention(sd, f"{p}.mid_block.attentions.0", h)
    h = unet_resnet_block(sd, f"{p}.mid_block.resnets.1", h, temb)
    z_sample = h
    for i in range(3):
        for j in range(3):
            h = unet_resnet_block(sd, f"{p}.up_blocks.{i}.resnets.{j}", torch.cat([h, skips.pop()], 1), temb)
            if i < 2:
                h = enc_attention(sd, f"{p}.up_blocks.{i}.attentions.{j}", h)
        if i < 2:
            h = conv(sd, f"{p}.up_blocks.{i}.upsamplers.0.conv", F.interpolate(h, scale_factor=2.0, mode="nearest"),
                     padding=1)
    h = conv(sd, f"{p}.conv_out", F.silu(gn(sd, f"{p}.conv_norm_out", h, 1e-5)), padding=1)
    return h, z_sample


def codec_encoder(sd, x, cond, p="enc"):
    """Encoder.forward (codec_module.py:75-85): x (B,3,H,W) picture in [-1,1], cond (B,4,H/8,W/8) VAE latent
    -> (y (B,128,H/16,W/16), sem (B,768,H/64,W/64))."""
    x_emb = conv(sd, f"{p}.pix_emb", x, stride=8)
    x_emb = conv(sd, f"{p}.pix_fusion", torch.cat([x_emb, cond], 1))
    y, sem = enc_unet_forward(sd, x_emb, f"{p}.unet")
    y = vq_resnet_block(sd, f"{p}.trans_coding.0", y)
    y = vq_attn_block(sd, f"{p}.trans_coding.1", y)
    y = vq_resnet_block(sd, f"{p}.trans_coding.2", y)
    y = depth_conv_block4(sd, f"{p}.trans_coding.3", y)
    y = depth_conv_block4(sd, f"{p}.trans_coding.4", y)
    return y, sem


# ------------------------------------------------------------------ whole decode / encode-side prior
def z_indices_to_params(codec_sd, indices):
    z_hat = fsq_indices_to_codes(indices)
    params, z_semantic = hyper_decoder(codec_sd, z_hat)
    return y_prior_fusion(codec_sd, params), z_semantic


def decode_from_latents(codec_sd, unet_sd, vae_sd, y_hat, z_semantic, timestep=999, attn_patch=16):
    """Everything after the entropy decode: codec_module.py:452-453 then
    model_sd15_with_codec_stage1.py:303-324.  Returns a dict of stage outputs."""
    y_semantic = semantic_adaptor(codec_sd, z_semantic)
    h_detail = codec_decoder(codec_sd, y_hat, y_semantic)
    B = y_hat.shape[0]
    ctx = y_semantic.permute(0, 2, 3, 1).reshape(B, -1, y_semantic.shape[1])
    t = torch.full((B,), timestep, dtype=torch.long)
    eps, reduced = unet_forward(unet_sd, h_detail, t, ctx)
    x0 = get_x0_from_noise(reduced.double(), eps.double(), ddim_alphas_cumprod().double(), t).float()
    image = vae_decode(vae_sd, x0, attn_patch)
    return {"y_semantic": y_semantic, "h_detail": h_detail, "eps": eps, "reduced": reduced,
            "x0": x0, "image": image}


def psnr(a, b):
    """test_quality.py:234-235 on [0,1] images."""
    mse = torch.mean((a.double() - b.double()) ** 2)
    return float(-10 * torch.log10(mse))


def to_unit_range(img):
    """inference.py:106 — clamp(-1,1)*0.5+0.5"""
    return img.clamp(-1.0, 1.0) * 0.5 + 0.5
