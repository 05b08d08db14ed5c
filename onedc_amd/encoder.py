"""Encode side: the analysis path on HIP kernels.

    VAEEncoder     diffusers AutoencoderKL.encode of the SD-2.1 VAE with the windowed mid-block attention
                   patch (reference model_sd15_with_codec_stage1.py:191-195,
                   modules/vae/autoencoders_patch_attn.py:9-80) -> latent_dist.sample() * scaling_factor
    CodecEncoder   reference `Encoder` (codec_module.py:51-85): pix_emb (8x8/8), pix_fusion, the patched
                   UNet2DModel of encoder_unet.py:137-161 run by forward_enc (:54-134), trans_coding

The bodies of AutoencoderKL / UNet2DModel are third-party (diffusers 0.32.1): restated from their
architecture, parity unpinned.  Everything reuses the decode path's kernels: implicit-GEMM
convs (stride 2 with the one-sided padding of Downsample2D is the kernel's own bounds check), epilogue
GroupNorm statistics, flash attention (head dim 8 padded to 32/16), concat-free skip buffers.
"""
import torch

from . import arch, pack
from . import kernels as K
from .kernels import View, new_view
from .layers import (Conv, DepthConvBlock4, GroupNorm, ResnetBlock2D, SingleHeadAttention, VQAttnBlock, VQResnetBlock,
                     Workspace, silu_cpu, timestep_embedding_cpu)


def windowed_attention(x, attn_norm, attn, patch, ws, device):
    """windowed_attn (autoencoders_patch_attn.py:9-29, eval branch) for a batch: the reference slices
    [i:i+patch, j:j+patch], so maps that are not a multiple of the patch get narrower windows at the
    right / bottom edge; each of the up to four window shapes is one batched pass."""
    B, H, W, C = x.B, x.H, x.W, x.C
    p = patch
    out = new_view(B, H, W, C, device)
    rows = [(0, H - H % p, p)] + ([(H - H % p, H % p, H % p)] if H % p else [])
    cols = [(0, W - W % p, p)] + ([(W - W % p, W % p, W % p)] if W % p else [])
    for y0, rh, wh in rows:
        for x0, rw, ww in cols:
            if rh == 0 or rw == 0:
                continue
            nwin = (rh // wh) * (rw // ww)
            xw = new_view(B * nwin, wh, ww, C, device)
            K.window_partition(x.t, xw.t, B, H, W, C, p, region=(y0, x0, rh, rw), win_hw=(wh, ww))
            hn = attn_norm(xw, ws, False)                     # statistics per window
            ow = attn(hn, xw, wh * ww)
            K.window_partition(ow.t, out.t, B, H, W, C, p, reverse=True, region=(y0, x0, rh, rw), win_hw=(wh, ww))
    return out


class VAEEncoder:
    def __init__(self, sd, device, attn_patch=16, scaling_factor=0.18215):
        sd = dict(sd)
        a = "encoder.mid_block.attentions.0"
        for old, new in arch.VAE_LEGACY_ATTN.items():
            for leaf in ("weight", "bias"):
                if f"{a}.{old}.{leaf}" in sd:
                    sd[f"{a}.{new}.{leaf}"] = sd.pop(f"{a}.{old}.{leaf}").reshape(512, -1).squeeze(-1)
        self.device, self.attn_patch, self.scaling_factor = device, attn_patch, scaling_factor
        self.act = K.act()
        self.ws = Workspace(device)
        # conv_in (3x3, 3 -> 128), input channels zero-padded 3 -> 32, f32 picture in [-1, 1] (reflect-padded): on split-half
        # operands (the 16-bit matrix rate with f32-grade sums: layers.Conv) -- the picture is inside the half
        # range by construction; weights of 6e4 and more (never seen) keep the f32 instructions, as does ONEDC_PRIOR_SPLIT16=0
        wci = torch.zeros(128, 32, 3, 3)
        wci[:, :3] = sd["encoder.conv_in.weight"].float()
        self.conv_in = Conv({}, "", device, weight=wci, bias=sd["encoder.conv_in.bias"], f32=True,
                            exact=bool(wci.abs().max() >= 6.0e4), flop_scale=3 / 32)
        self.down = []
        for i in range(4):
            blk = {"res": [ResnetBlock2D(sd, f"encoder.down_blocks.{i}.resnets.{j}", device, 1e-6, bf16_operands=False) for j in range(2)]}
            blk["down"] = Conv(sd, f"encoder.down_blocks.{i}.downsamplers.0.conv", device) if i < 3 else None
            self.down.append(blk)
        self.mid_res = [ResnetBlock2D(sd, f"encoder.mid_block.resnets.{j}", device, 1e-6, bf16_operands=False) for j in range(2)]
        self.attn_norm = GroupNorm(sd, f"{a}.group_norm", device, 1e-6)
        self.attn = SingleHeadAttention(sd, (f"{a}.to_q", f"{a}.to_k", f"{a}.to_v", f"{a}.to_out.0"), device, 512)
        self.norm_out = GroupNorm(sd, "encoder.conv_norm_out", device, 1e-6)
        self.conv_out = Conv(sd, "encoder.conv_out", device)
        self.q_w = sd["quant_conv.weight"].reshape(8, 8).float().contiguous().to(device)
        self.q_b = sd["quant_conv.bias"].float().contiguous().to(device)

    def set_attn_patch(self, attn_patch):
        assert attn_patch > 0, "attn_patch must be greater than 0"
        self.attn_patch = attn_patch

    @torch.no_grad()
    @K.in_own_format
    def encode(self, image, noise=None):
        """image: f32 NHWC View (B,H,W,32) holding the picture in channels 0..2 (3..31 zero) ->
        f32 View (B,H/8,W/8,4): latent_dist.sample() * scaling_factor.  noise: f32 device tensor
        (B,H/8,W/8,4) standing in for the reference's global-RNG draw, None = the distribution's mode."""
        ws, dev = self.ws, self.device
        h = self.conv_in(image, new_view(image.B, image.H, image.W, 128, dev), stats=True)
        for blk in self.down:
            for r in blk["res"]:
                h = r(h, ws)
            if blk["down"] is not None:
                h = blk["down"](h, stride=2, pad=0, stats=True)
        h = self.mid_res[0](h, ws)
        h = windowed_attention(h, self.attn_norm, self.attn, self.attn_patch, ws, dev)
        h = self.mid_res[1](h, ws)
        h = self.norm_out(h, ws, True)
        m = View(torch.empty((h.B * h.H * h.W + K.SLACK_ROWS) * 8, dtype=torch.float32, device=dev), h.B, h.H, h.W, 8)
        self.conv_out(h, m)
        mom = new_view(h.B, h.H, h.W, 8, dev, torch.float32)
        K.direct_conv(m, self.q_w, self.q_b, mom, K=1)
        z = new_view(h.B, h.H, h.W, 4, dev, torch.float32)
        return K.latent_sample(mom, noise, z, self.scaling_factor)


class OwnResnetBlock:
    """The reference's own ResnetBlock of the analysis UNet's conv_in (encoder_unet.py:14-52): bias-free
    convs and, when in != out, a 1x1 `nin_shortcut` applied to the block OUTPUT (:47-50)."""

    def __init__(self, sd, p, device):
        self.n1 = GroupNorm(sd, f"{p}.norm1", device, 1e-6)
        self.c1 = Conv(sd, f"{p}.conv1", device)
        self.n2 = GroupNorm(sd, f"{p}.norm2", device, 1e-6)
        self.c2 = Conv(sd, f"{p}.conv2", device)
        self.nin = Conv(sd, f"{p}.nin_shortcut", device) if f"{p}.nin_shortcut.weight" in sd else None

    def __call__(self, x, ws, y=None):
        h = self.c1(self.n1(x, ws, True), stats=True)
        if self.nin is None:
            return self.c2(self.n2(h, ws, True), y, res=x, stats=True)
        h = self.c2(self.n2(h, ws, True))
        return self.nin(h, y, res=h, stats=True)


class GroupNormAttention:
    """diffusers `Attention` of AttnDownBlock2D / AttnUpBlock2D / UNetMidBlock2D inside UNet2DModel:
    GroupNorm(32) -> q, k, v Linear (with bias) -> heads of 8 channels -> to_out.0 -> + residual."""

    def __init__(self, sd, p, device, C, head_dim=arch.ENC_HEAD_DIM, eps=1e-5):
        H = C // head_dim
        self.H, self.d, self.C = H, head_dim, C
        self.DQ, self.DV = pack.HEAD_PAD[head_dim]
        self.norm = GroupNorm(sd, f"{p}.group_norm", device, eps)

        def padded_bias(b, dp):
            return pack.pad_heads_rows(b.float()[:, None], H, head_dim, dp)[:, 0]

        self.q = Conv(sd, "", device, weight=pack.pad_heads_rows(sd[f"{p}.to_q.weight"], H, head_dim, self.DQ),
                      bias=padded_bias(sd[f"{p}.to_q.bias"], self.DQ), flop_scale=head_dim / self.DQ)
        self.k = Conv(sd, "", device, weight=pack.pad_heads_rows(sd[f"{p}.to_k.weight"], H, head_dim, self.DQ),
                      bias=padded_bias(sd[f"{p}.to_k.bias"], self.DQ), flop_scale=head_dim / self.DQ)
        wv = pack.pad_heads_rows(sd[f"{p}.to_v.weight"], H, head_dim, self.DV)
        self.wv = View(pack.pack_matrix(wv, device), 1, 1, H * self.DV, C)   # V^T = Wv . h^T: Wv is the "pixel" operand
        # softmax rows sum to 1, so the v bias rides on the output: W_o (P V + 1 b_v^T) + b_o = W_o P V + (W_o b_v + b_o)
        wo = sd[f"{p}.to_out.0.weight"].float()
        bo = sd[f"{p}.to_out.0.bias"].float() + wo @ sd[f"{p}.to_v.bias"].float()
        self.o = Conv(sd, "", device, weight=pack.pad_heads_cols(wo, H, head_dim, self.DV), bias=bo,
                      flop_scale=head_dim / self.DV)

    def __call__(self, x, ws, y=None):
        dev = x.t.device
        B, N = x.B, x.H * x.W
        n_pad = (N + 63) // 64 * 64
        hn = self.norm(x, ws, False)
        q, k = self.q(hn), self.k(hn)
        vt_dtype = K.flash_vt_dtype(self.DQ, self.DV, N)
        vt = torch.zeros(B * self.H * self.DV * n_pad, dtype=vt_dtype, device=dev)
        K.conv_gemm(self.wv, hn.t, None, View(vt, 1, 1, self.H * self.DV, N, n_pad), Cin=self.C, Cout=N, batch=B,
                    x_bs=0, w_bs=N * hn.cs, y_bs=self.H * self.DV * n_pad, flop_scale=self.d / self.DV,
                    out_bf16=(vt_dtype == torch.bfloat16))
        o = new_view(B, x.H, x.W, self.H * self.DV, dev)
        K.flash_attn(q.t, k.t, vt, o.t, B, self.H, N, N, n_pad, self.DQ, self.DV, self.d ** -0.5)
        return self.o(o, y, res=x, stats=True)


class AnalysisUNet:
    """The patched UNet2DModel (encoder_unet.py:137-161) as run by forward_enc (:54-134) -> (y_sample, z_sample).
    The timestep is the constant 999, so every resnet's time_emb_proj(SiLU(emb)) is folded into its conv1 bias."""

    def __init__(self, sd, device, p="enc.unet"):
        self.device, self.ws = device, Workspace(device)
        self.act = K.act()
        ch = arch.ENC_UNET_CH
        self.conv_in = [OwnResnetBlock(sd, f"{p}.conv_in.{i}", device) for i in range(3)]
        self.conv_in_down = Conv(sd, f"{p}.conv_in.3", device)
        self.down = []
        for i, c in enumerate(ch):
            blk = {"res": [ResnetBlock2D(sd, f"{p}.down_blocks.{i}.resnets.{j}", device, 1e-5, bf16_operands=False) for j in range(2)]}
            blk["attn"] = [GroupNormAttention(sd, f"{p}.down_blocks.{i}.attentions.{j}", device, c)
                           for j in range(2)] if i < 2 else None
            blk["down"] = Conv(sd, f"{p}.down_blocks.{i}.downsamplers.0.conv", device) if i < 2 else None
            self.down.append(blk)
        self.mid_res = [ResnetBlock2D(sd, f"{p}.mid_block.resnets.{j}", device, 1e-5, bf16_operands=False) for j in range(2)]
        self.mid_attn = GroupNormAttention(sd, f"{p}.mid_block.attentions.0", device, ch[-1])
        self.up = []
        plan = arch.enc_unet_up_plan()
        for i in range(3):
            blk = {"res": [ResnetBlock2D(sd, f"{p}.up_blocks.{i}.resnets.{j}", device, 1e-5, bf16_operands=False) for j in range(3)],
                   "io": [plan[(i, j)] for j in range(3)]}
            blk["attn"] = [GroupNormAttention(sd, f"{p}.up_blocks.{i}.attentions.{j}", device, plan[(i, j)][1])
                           for j in range(3)] if i < 2 else None
            blk["up"] = Conv(sd, f"{p}.up_blocks.{i}.upsamplers.0.conv", device, upsample=True) if i < 2 else None
            self.up.append(blk)
        self.norm_out = GroupNorm(sd, f"{p}.conv_norm_out", device, 1e-5)
        self.conv_out = Conv(sd, f"{p}.conv_out", device)
        # time embedding of t = 999 (Timesteps(512, flip_sin_to_cos, shift 0) -> linear_1 -> SiLU -> linear_2)
        w1, b1, w2, b2 = (sd[f"{p}.time_embedding.linear_{i}.{n}"].float() for i in (1, 2) for n in ("weight", "bias"))
        emb = timestep_embedding_cpu(999, ch[0]).to(w1.device)
        act = silu_cpu(w2 @ silu_cpu(w1 @ emb + b1) + b2)
        for blk in self.down + self.up:
            for r in blk["res"]:
                r.set_temb(act)
        for r in self.mid_res:
            r.set_temb(act)

    @torch.no_grad()
    @K.in_own_format
    def __call__(self, x):
        ws, dev, B = self.ws, self.device, x.B
        h = x
        for r in self.conv_in:
            h = r(h, ws)
        # skips in production order: conv_in, (r0, r1, down) x2, r0, r1; the up path pops them from the end and
        # reads cat(h, skip): every skip is written straight into the tail slice of its consumer's buffer
        H0, W0 = (h.H + 1) // 2, (h.W + 1) // 2
        res = [(H0, W0), ((H0 + 1) // 2, (W0 + 1) // 2), ((H0 + 3) // 4, (W0 + 3) // 4)]
        skip_c = [512, 512, 512, 512, 768, 768, 768, 768, 768]
        skip_lvl = [0, 0, 0, 1, 1, 1, 2, 2, 2]
        cat_bufs, order = {}, []
        s = len(skip_c) - 1
        for i, blk in enumerate(self.up):
            for j in range(3):
                cin_total = blk["io"][j][0]
                lvl = skip_lvl[s]
                buf = new_view(B, res[lvl][0], res[lvl][1], cin_total, dev, stats=True)
                cat_bufs[s] = (buf, cin_total - skip_c[s])
                order.append(s)
                s -= 1

        def skip_slot(idx):
            buf, h_c = cat_bufs[idx]
            return buf.slice(h_c, skip_c[idx])

        def head_slot(idx):
            buf, h_c = cat_bufs[idx]
            return buf.slice(0, h_c)

        si = 0
        h = self.conv_in_down(h, skip_slot(si), stride=2, stats=True)
        si += 1
        for i, blk in enumerate(self.down):
            for j in range(2):
                if blk["attn"] is not None:
                    h = blk["res"][j](h, ws)
                    h = blk["attn"][j](h, ws, skip_slot(si))
                else:
                    h = blk["res"][j](h, ws, skip_slot(si))
                si += 1
            if blk["down"] is not None:
                h = blk["down"](h, skip_slot(si), stride=2, stats=True)
                si += 1
        h = self.mid_res[0](h, ws)
        h = self.mid_attn(h, ws)
        z_sample = self.mid_res[1](h, ws)
        K.copy_channels(z_sample, head_slot(order[0]))        # z_sample is also an output: keep it whole
        oi = 0
        for i, blk in enumerate(self.up):
            for j in range(3):
                idx = order[oi]
                oi += 1
                cat = cat_bufs[idx][0]
                has_next = oi < len(order)
                dst = None
                if has_next and not (j == 2 and blk["up"] is not None):
                    dst = head_slot(order[oi])
                if blk["attn"] is not None:
                    h = blk["res"][j](cat, ws)
                    h = blk["attn"][j](h, ws, dst)
                else:
                    h = blk["res"][j](cat, ws, dst)
            if blk["up"] is not None:
                h = blk["up"](h, head_slot(order[oi]), up=1, stats=True)
        h = self.norm_out(h, ws, True)
        return self.conv_out(h, stats=True), z_sample


class CodecEncoder:
    """reference `Encoder` (codec_module.py:51-85) -> (y (B,H/16,W/16,128), sem (B,H/64,W/64,768))."""

    def __init__(self, sd, device, p="enc"):
        self.device, self.ws = device, Workspace(device)
        self.act = K.act()
        self.pix_emb = Conv(sd, f"{p}.pix_emb", device)       # (192, 3, 8, 8) -> [192][ky][kx][c]: a 1x1 GEMM on patches
        self.pix_emb.k = 1
        self.pix_emb.cin = 192
        wf = sd[f"{p}.pix_fusion.weight"].reshape(320, -1).float()
        wfp = torch.zeros(320, 256)                           # cat(x_emb 192, cond 4) zero-padded to 256 channels
        wfp[:, :wf.shape[1]] = wf
        self.pix_fusion = Conv({}, "", device, weight=wfp, bias=sd[f"{p}.pix_fusion.bias"], flop_scale=196 / 256)
        self.unet = AnalysisUNet(sd, device, f"{p}.unet")
        t = f"{p}.trans_coding"
        self.t0 = VQResnetBlock(sd, f"{t}.0", device)
        self.t1 = VQAttnBlock(sd, f"{t}.1", device, 512)
        self.t2 = VQResnetBlock(sd, f"{t}.2", device)
        self.t3 = DepthConvBlock4(sd, f"{t}.3", device)
        self.t4 = DepthConvBlock4(sd, f"{t}.4", device)

    @torch.no_grad()
    @K.in_own_format
    def __call__(self, image, cond):
        """image: f32 NHWC View (B,H,W,>=3); cond: f32 View (B,H/8,W/8,4) (the VAE latent)."""
        dev, ws = self.device, self.ws
        B, h8, w8 = image.B, image.H // 8, image.W // 8
        patches = new_view(B, h8, w8, 192, dev)
        K.patchify8(image, patches)
        cat = View(torch.zeros((B * h8 * w8 + K.SLACK_ROWS) * 256, dtype=K.a16(), device=dev), B, h8, w8, 256)
        self.pix_emb(patches, cat.slice(0, 192))
        K.copy_channels(cond, cat.slice(192, 4))
        x_emb = self.pix_fusion(cat, stats=True)
        y, sem = self.unet(x_emb)
        y = self.t2(self.t1(self.t0(y, ws), ws), ws)
        y = self.t4(self.t3(y))
        return y, sem
