"""The one-step SD1.5 UNet of the decoder on HIP kernels.

Mirrors reference src/models/sd15_onedc_codec_stage1/decoder_unet.py: `prepare_unet_for_codec`
:371-408 (SD1.5 topology, new 320->320 `conv_in`, `vae_reduction`), `forward_unet` :32-305 (returns
(eps, reduced_sample)), `reduce_resblock` :14-29, `unet_add_lora` :331-368 (folded at load).
The body of UNet2DConditionModel is third-party (diffusers 0.32.1) and restated from its
architecture.  channel concatenation with the skip connections is realised by writing producers
into channel slices of shared buffers.
"""
import torch

from . import arch, pack
from . import kernels as K
from .kernels import View, new_view
from .layers import (ContextKV, Conv, GroupNorm, ResnetBlock2D, Transformer2D, Workspace, silu_cpu,
                     timestep_embedding_cpu)


class ReduceResblock:
    """reduce_resblock (decoder_unet.py:14-29): GN-SiLU-conv3x3-GN-SiLU-conv3x3 + 1x1 shortcut."""

    def __init__(self, sd, device, p="vae_reduction"):
        self.n0 = GroupNorm(sd, f"{p}.blocks.0", device, 1e-6)
        self.c2 = Conv(sd, f"{p}.blocks.2", device)
        self.n3 = GroupNorm(sd, f"{p}.blocks.3", device, 1e-6)
        self.c5 = Conv(sd, f"{p}.blocks.5", device)
        self.short = Conv(sd, f"{p}.short_cut", device)

    def __call__(self, x, ws):
        h = self.c2(self.n0(x, ws, True), stats=True)
        h = self.n3(h, ws, True)
        sc = self.short(x)                                  # bf16 (B,H,W,4)
        return self.c5(h, res=sc, out_f32=True)             # f32: feeds the x0 formula (x14.65 gain)


class UNet:
    def __init__(self, sd, device, lora_rank=64, lora_alpha=8.0):
        """sd: checkpoint-named state dict of model.safetensors (LoRA tensors still separate)."""
        self.device = device
        self.act = K.act()                 # the format the weights below are packed in
        if any(".lora_A." in k for k in sd):
            sd = pack.fold_lora(sd, lora_rank, lora_alpha)
        self.ws = Workspace(device)
        self.vae_reduction = ReduceResblock(sd, device)
        self.conv_in = Conv(sd, "conv_in", device)
        self._te = [sd[f"time_embedding.linear_{i}.{n}"].float() for i in (1, 2) for n in ("weight", "bias")]
        ch = arch.UNET_CH
        self.down = []
        for i, c in enumerate(ch):
            blk = {"res": [ResnetBlock2D(sd, f"down_blocks.{i}.resnets.{j}", device, 1e-5) for j in range(2)]}
            blk["attn"] = [Transformer2D(sd, f"down_blocks.{i}.attentions.{j}", device, c) for j in range(2)] if i < 3 else None
            blk["down"] = Conv(sd, f"down_blocks.{i}.downsamplers.0.conv", device) if i < 3 else None
            self.down.append(blk)
        self.mid_res = [ResnetBlock2D(sd, f"mid_block.resnets.{j}", device, 1e-5) for j in range(2)]
        self.mid_attn = Transformer2D(sd, "mid_block.attentions.0", device, 1280)
        self.up = []
        for i, (c, ins) in enumerate(arch.UNET_UP_IN):
            blk = {"res": [ResnetBlock2D(sd, f"up_blocks.{i}.resnets.{j}", device, 1e-5) for j in range(3)],
                   "ins": ins, "c": c}
            blk["attn"] = [Transformer2D(sd, f"up_blocks.{i}.attentions.{j}", device, c) for j in range(3)] if i > 0 else None
            blk["up"] = Conv(sd, f"up_blocks.{i}.upsamplers.0.conv", device, upsample=True) if i < 3 else None
            self.up.append(blk)
        self.norm_out = GroupNorm(sd, "conv_norm_out", device, 1e-5)
        self.conv_out = Conv(sd, "conv_out", device)
        self._timestep = None
        # the 16 cross-attention layers read one context: their to_k / to_v run as three GEMMs per step (layers.ContextKV)
        self._transformers = [t for blk in self.down if blk["attn"] for t in blk["attn"]] + [self.mid_attn] + \
                             [t for blk in self.up if blk["attn"] for t in blk["attn"]]
        self.ctx_kv = ContextKV([t.attn2 for t in self._transformers], device)

    def _all_resnets(self):
        for blk in self.down:
            yield from blk["res"]
        yield from self.mid_res
        for blk in self.up:
            yield from blk["res"]

    def set_timestep(self, t):
        """time_embedding(sinusoid(t)) (decoder_unet.py:138-139) is batch independent and the codec
        uses one constant timestep, so each resnet's `time_emb_proj(SiLU(emb))` is folded into
        conv1's bias once per timestep."""
        if self._timestep == t:
            return
        w1, b1, w2, b2 = self._te
        emb = timestep_embedding_cpu(t, arch.UNET_CH[0]).to(w1.device)
        emb = w2 @ silu_cpu(w1 @ emb + b1) + b2
        act = silu_cpu(emb)
        for r in self._all_resnets():
            r.set_temb(act)
        self._timestep = t

    @K.in_own_format
    @torch.no_grad()
    def __call__(self, sample, timestep, ctx):
        """sample: View (B,H,W,320) bf16; ctx: View (B,h,w,768) tokens. -> (eps f32 View, reduced f32 View)"""
        self.set_timestep(int(timestep))
        ws, dev = self.ws, self.device
        B = sample.B
        reduced = self.vae_reduction(sample, ws)
        kv = {id(t): v for t, v in zip(self._transformers, self.ctx_kv(ctx, ws))}

        # Skip tensors are written straight into the channel tail of the buffer the matching up
        # resnet will read (torch.cat([h, skip]) in diffusers), so plan those buffers first.
        # skip order as produced: conv_in, (r0, r1, down) x3, r0, r1
        res = [(sample.H >> i, sample.W >> i) for i in range(4)]
        skip_c = [320, 320, 320, 320, 640, 640, 640, 1280, 1280, 1280, 1280, 1280]
        skip_lvl = [0, 0, 0, 1, 1, 1, 2, 2, 2, 3, 3, 3]
        # consumer of skip s (popped from the end): up block i, resnet j
        cat_bufs = {}
        order = []
        s = len(skip_c) - 1
        for i, blk in enumerate(self.up):
            for j in range(3):
                cin_total = blk["ins"][j]
                h_c = cin_total - skip_c[s]
                lvl = skip_lvl[s]
                # statistics granularity: pairs when the whole buffer or one of its two slices is normalised
                # with groups that are not multiples of 4 channels (320 -> 10, 960 -> 30 channels per group)
                gran = 2 if any((c // 32) % 4 for c in (cin_total, h_c, skip_c[s])) else 4
                buf = new_view(B, res[lvl][0], res[lvl][1], cin_total, dev, stats=gran)
                cat_bufs[s] = (buf, h_c)
                order.append(s)
                s -= 1

        def skip_slot(idx):
            buf, h_c = cat_bufs[idx]
            return buf.slice(h_c, skip_c[idx])

        def head_slot(idx):
            buf, h_c = cat_bufs[idx]
            return buf.slice(0, h_c)

        si = 0
        h = self.conv_in(sample, skip_slot(si), stats=True)
        si += 1
        for i, blk in enumerate(self.down):
            for j in range(2):
                if blk["attn"] is not None:
                    h = blk["res"][j](h, ws)
                    h = blk["attn"][j](h, ctx, ws, skip_slot(si), kv=kv[id(blk["attn"][j])])
                else:
                    h = blk["res"][j](h, ws, skip_slot(si))
                si += 1
            if blk["down"] is not None:
                h = blk["down"](h, skip_slot(si), stride=2, stats=True)
                si += 1

        h = self.mid_res[0](h, ws)
        h = self.mid_attn(h, ctx, ws, kv=kv[id(self.mid_attn)])
        # the first up resnet reads cat(h, skip): h must land in the head of that buffer
        nxt = order[0]
        h = self.mid_res[1](h, ws, head_slot(nxt))

        oi = 0
        for i, blk in enumerate(self.up):
            for j in range(3):
                idx = order[oi]
                oi += 1
                cat = cat_bufs[idx][0]
                last_in_block = j == 2
                has_next = oi < len(order)
                # where does this resnet(+attn) output go?  into the head of the next cat buffer unless
                # an upsampler conv sits in between (then the upsampler writes there)
                dst = None
                if has_next and not (last_in_block and blk["up"] is not None):
                    dst = head_slot(order[oi])
                if blk["attn"] is not None:
                    h = blk["res"][j](cat, ws)
                    h = blk["attn"][j](h, ctx, ws, dst, kv=kv[id(blk["attn"][j])])
                else:
                    h = blk["res"][j](cat, ws, dst)
            if blk["up"] is not None:
                h = blk["up"](h, head_slot(order[oi]), up=1, stats=True)

        h = self.norm_out(h, ws, True)
        eps = self.conv_out(h, out_f32=True)
        return eps, reduced
