"""SD-2.1 VAE decoder with windowed mid-block attention, on HIP kernels.

Mirrors reference `vae_decode_image` (src/models/sd15_onedc_codec_stage1/
model_sd15_with_codec_stage1.py:184-188) -> diffusers `AutoencoderKL.decode` as patched by
src/modules/vae/autoencoders_patch_attn.py:9-80 (attention, including its GroupNorm, runs per
`attn_patch` x `attn_patch` window; here all windows of the batch go through one set of batched
launches instead of a Python double loop).  The decoder body is third-party and restated.
"""
import torch

from . import arch
from . import kernels as K
from .kernels import View, new_view
from .layers import Conv, GroupNorm, ResnetBlock2D, SingleHeadAttention, Workspace


F16_STREAM_SCALE = 2.0 ** -6


class VAEDecoder:
    def __init__(self, sd, device, attn_patch=16, scaling_factor=0.18215, stream_scale=None, up3_act=None):
        """stream_scale: the decoder's residual stream (everything between conv_in and conv_norm_out) is stored times
        this power of two.  With IEEE-half activations the default 2^-6 moves the stream's headroom from 65504 to 4.2e6
        (SD-VAE decoder activations reach the thousands; with random-init weights and their huge x0, 2e4) at no cost:
        every consumer of the stream is a GroupNorm (scale invariant, eps scaled by s^2) or a linear map back into the
        stream, a power of two changes no rounding, and values below 2^-14 / s = 0.004 are far under the noise of a
        group's statistics.  bf16 has f32's range: 1."""
        sd = dict(sd)
        if stream_scale is None:
            stream_scale = F16_STREAM_SCALE if K.act() == "f16" else 1.0
        self.stream_scale = s = float(stream_scale)
        a = "decoder.mid_block.attentions.0"
        for old, new in arch.VAE_LEGACY_ATTN.items():      # pre-0.15 diffusers attention names
            for leaf in ("weight", "bias"):
                if f"{a}.{old}.{leaf}" in sd:
                    sd[f"{a}.{new}.{leaf}"] = sd.pop(f"{a}.{old}.{leaf}").reshape(512, -1).squeeze(-1)
        self.device = device
        self.act = K.act()                 # the format the weights below are packed in
        self.attn_patch = attn_patch
        self.scaling_factor = scaling_factor
        self.ws = Workspace(device)
        # post_quant_conv (1x1, 4->4) and conv_in (3x3, 4->512): tiny-channel direct kernels in f32
        self.pq_w = sd["post_quant_conv.weight"].permute(0, 2, 3, 1).reshape(4, -1).float().contiguous().to(device)
        self.pq_b = sd["post_quant_conv.bias"].float().contiguous().to(device)
        # conv_in (3x3, 4 -> 512), input channels zero-padded 4 -> 32, f32 activations.  Round 4: on split-half operands (the 16-bit
        # matrix rate with f32-grade sums, layers.Conv / DESIGN.md section 3) instead of the f32 matrix instructions (0.46 -> 0.25 ms
        # per batch).  Its operands must lie inside the half range, and x0 of a random-init model reaches 1e6 and more, so the
        # latent travels multiplied by 2^-10 (post_quant_conv's input scale and bias below) and conv_in's weights by 2^10: powers of
        # two, the products and sums are the same numbers.  What that costs: a latent component below 0.06 keeps an ABSOLUTE
        # precision of 3e-8 instead of 22 bits; a latent beyond 6.7e7 becomes inf and is reported like any other non-finite
        # picture; weights of 64 and more (never seen in a VAE) keep the f32 instructions, as does ONEDC_PRIOR_SPLIT16=0.
        self.latent_prescale = 2.0 ** -10
        wci = torch.zeros(512, 32, 3, 3)
        wci[:, :4] = sd["decoder.conv_in.weight"].float() / self.latent_prescale
        fits = bool(wci.abs().max() < 6.0e4)
        if not fits:
            self.latent_prescale = 1.0
            wci[:, :4] = sd["decoder.conv_in.weight"].float()
        self.pq_b = (self.pq_b * self.latent_prescale).contiguous()
        self.conv_in = Conv({}, "", device, weight=wci, bias=sd["decoder.conv_in.bias"] * s, f32=True, exact=not fits, flop_scale=4 / 32)
        self.mid_res = [ResnetBlock2D(sd, f"decoder.mid_block.resnets.{j}", device, 1e-6, s) for j in range(2)]
        self.attn_norm = GroupNorm(sd, f"{a}.group_norm", device, 1e-6 * s * s)
        self.attn = SingleHeadAttention(sd, (f"{a}.to_q", f"{a}.to_k", f"{a}.to_v", f"{a}.to_out.0"), device, 512)
        self.attn.scale_output(s)
        # up3_act (round 5): the 16-bit format of the LAST up block (three resnets at picture resolution) + conv_norm_out +
        # conv_out, "bf16" inside an f16 decoder: what those layers round meets no further GroupNorm-and-amplify stage (the
        # MXFP8 experiment of round 3 found this block alone inside the 0.01 dB budget even at 3 mantissa bits), and the chip
        # holds a 3 - 6 % higher clock on 8-bit mantissas in exactly these MFMA-bound launches.  The upsampler in front of
        # it stores bf16 (onedc_conv_params.out_bf16).  None: the decoder's own format throughout.
        self.up3_act = up3_act if (up3_act and up3_act != self.act) else None
        self.up = []
        for i, (cin, cout) in enumerate(arch.VAE_UP):
            with K.act_format(self.up3_act if (i == 3 and self.up3_act) else self.act):
                blk = {"res": [ResnetBlock2D(sd, f"decoder.up_blocks.{i}.resnets.{j}", device, 1e-6, s) for j in range(3)]}
            blk["up"] = Conv(sd, f"decoder.up_blocks.{i}.upsamplers.0.conv", device, upsample=True) if i < 3 else None
            if blk["up"] is not None:
                blk["up"].b = (blk["up"].b * s).contiguous()       # a linear map of the stream: only its bias scales
            self.up.append(blk)
        with K.act_format(self.up3_act or self.act):
            self.norm_out = GroupNorm(sd, "decoder.conv_norm_out", device, 1e-6 * s * s)
            self.conv_out = Conv(sd, "decoder.conv_out", device)

    MXFP8_GROUPS = ("mid", "up0", "up1", "up2", "up3")

    def set_mxfp8(self, groups=()):
        """MXFP8 experiment: run the 3x3 convs of the named parts of the decoder -- "mid" (two resnets, 96^2 at
        768x768), "up0" .. "up3" (three resnets + the upsampler conv each: 96^2, 192^2, 384^2, 768^2) -- on MXFP8-quantised
        operands (layers.Conv.set_mxfp8).  () switches it off again.  Off by default: DESIGN.md section 5 has the measured
        cost in dB."""
        groups = set(groups)
        unknown = groups - set(self.MXFP8_GROUPS)
        if unknown:
            raise ValueError(f"unknown decoder parts {sorted(unknown)} (known: {self.MXFP8_GROUPS})")
        parts = {"mid": [(r, None) for r in self.mid_res]}
        for i, blk in enumerate(self.up):
            parts[f"up{i}"] = [(r, None) for r in blk["res"]] + ([(None, blk["up"])] if blk["up"] is not None else [])
        for name, items in parts.items():
            on = name in groups
            for r, c in items:
                for conv in ((r.c1, r.c2) if r is not None else (c,)):
                    conv.set_mxfp8(on)
        self.mxfp8_groups = tuple(sorted(groups))
        return self

    def set_attn_patch(self, attn_patch):
        assert attn_patch > 0, "attn_patch must be greater than 0"
        self.attn_patch = attn_patch

    def _windowed_attention(self, x):
        from .encoder import windowed_attention
        return windowed_attention(x, self.attn_norm, self.attn, self.attn_patch, self.ws, self.device)

    @K.in_own_format
    @torch.no_grad()
    def decode(self, latents):
        """latents: f32 View (B,h,w,4) -> f32 View (B,8h,8w,3) picture in [-1,1] (unclamped)."""
        ws, dev = self.ws, self.device
        zbuf = torch.zeros((latents.B * latents.H * latents.W + K.SLACK_ROWS) * 32, dtype=torch.float32, device=dev)
        z = View(zbuf, latents.B, latents.H, latents.W, 32)
        K.direct_conv(latents, self.pq_w, self.pq_b, z.slice(0, 4), K=1, in_scale=self.latent_prescale / self.scaling_factor)
        h = self.conv_in(z, new_view(latents.B, latents.H, latents.W, 512, dev), stats=True, alpha=self.stream_scale)
        h = self.mid_res[0](h, ws)
        h = self._windowed_attention(h)
        h = self.mid_res[1](h, ws)
        for i, blk in enumerate(self.up):
            if i == 3 and self.up3_act:
                if h.t.dtype != (torch.bfloat16 if self.up3_act == "bf16" else torch.float16):
                    raise K.KernelError("vae_up3_act: the upsampler in front of the last block did not run as upsample phases "
                                        "(picture size not whole 16x16 patches at 1/2 resolution); use the default format")
                with K.act_format(self.up3_act):
                    for r in blk["res"]:
                        h = r(h, ws)
                    return self._tail(h, ws, dev)
            for r in blk["res"]:
                h = r(h, ws)
            if blk["up"] is not None:
                h = blk["up"](h, up=1, stats=True, out_bf16=bool(i == 2 and self.up3_act == "bf16"))
        return self._tail(h, ws, dev)

    def _tail(self, h, ws, dev):
        if h.H % 16 == 0 and h.W % 16 == 0 and h.C % 32 == 0 and K.USE_GN_FUSION and h.H * h.W * h.cs < 2 ** 31 - 1:
            # conv_norm_out + SiLU + conv_out in one pass over the 128-channel picture-resolution tensor
            y = new_view(h.B, h.H, h.W, 3, dev, torch.float32)
            return K.conv3x3_few_out(h, self.conv_out.w, self.conv_out.b, y,
                                     gn=(self.norm_out.g, self.norm_out.b, self.norm_out.eps, True, ws))
        h = self.norm_out(h, ws, True)
        return self.conv_out(h, out_f32=True)
