"""Device-side building blocks of the decode path, expressed as sequences of HIP kernel launches.

Each class is built from checkpoint-named tensors (`sd`, a flat state dict with LoRA already
folded) and keeps only packed device weights.  Names follow the reference modules they stand for:

    DepthConvBlock4, ResidualBlockUpsample   reference src/modules/dcvc.py:424-433, 183-208
    VQResnetBlock, VQAttnBlock               src/modules/vqgan/blocks.py:15-52, 55-106
    ResnetBlock2D, Transformer2D             diffusers blocks used by decoder_unet.py:204-299 and
                                             by AutoencoderKL.decode (third-party, restated)
"""
import math

import torch

from . import kernels as K
from . import pack
from .kernels import EPI_GEGLU, EPI_GLU_LEAKY, EPI_LEAKY, EPI_LINEAR, EPI_SILU, View, new_view


class Workspace:
    """Named scratch tensors that only ever grow (stable addresses).  One set per HIP stream: the pipeline runs
    several batches at once, each on its own stream, through the same layers."""

    def __init__(self, device):
        self.device = device
        self._t = {}

    def _get(self, name, n, dtype):
        key = (torch.cuda.current_stream(self.device).cuda_stream, name)
        t = self._t.get(key)
        if t is None or t.numel() < n or t.dtype != dtype:
            t = torch.empty(max(n, 1), dtype=dtype, device=self.device)
            self._t[key] = t
        return t

    def f32(self, name, n):
        return self._get(name, n, torch.float32)

    def bf16(self, name, n):
        return self._get(name, n, K.a16())

    def zeros(self, name, n, dtype):
        """A buffer that is zero when it is made and never re-zeroed: for tensors whose writers always touch the SAME
        positions and whose other positions must read as zero (the padding keys of V^T).  The caller puts everything that
        decides the layout into `name`, so a buffer is never shared between two layouts."""
        key = (torch.cuda.current_stream(self.device).cuda_stream, "zeros", name)
        t = self._t.get(key)
        if t is None or t.numel() != n or t.dtype != dtype:
            # a service that sees many picture sizes must not collect one buffer per layout for ever: keep the most
            # recently made MAX_ZERO_LAYOUTS per stream (a dropped one is simply zeroed again when its layout returns)
            # (list(dict) is one atomic call: the pipeline's stream threads share this object)
            mine = [k for k in list(self._t) if len(k) == 3 and k[0] == key[0]]
            for k in mine[:max(0, len(mine) + 1 - self.MAX_ZERO_LAYOUTS)]:
                self._t.pop(k, None)
            t = torch.zeros(max(n, 1), dtype=dtype, device=self.device)
            self._t[key] = t
        return t

    MAX_ZERO_LAYOUTS = 24

    def drop_stream(self, cuda_stream):
        """Forget the scratch tensors of one HIP stream (a DecodePipeline that is closed never launches on its streams again)."""
        for k in [k for k in list(self._t) if k[0] == cuda_stream]:
            self._t.pop(k, None)


USE_UPSAMPLE_PHASES = True    # tests switch it off to compare with the plain fused-upsample path
FUSE_SHORTCUT = __import__("os").environ.get("ONEDC_FUSE_SHORTCUT", "1") != "0"   # ResnetBlock2D.conv_shortcut inside conv2's launch
# f32 layers built while this is True compute on split operands (pairs of halves, 22 bits) at the 16-bit matrix rate; False: the
# exact f32 matrix instructions (a fixed-order fp32 fma chain).  codec_module.IntraNoAR sets it from `prior_f32` while it builds.
PRIOR_SPLIT16 = __import__("os").environ.get("ONEDC_PRIOR_SPLIT16", "1") != "0"


class Conv:
    """nn.Conv2d / nn.Linear as one implicit-GEMM launch."""

    def __init__(self, sd, name, device, *, perm=None, glu=False, shuffle=False, weight=None, bias=None,
                 flop_scale=1.0, f32=False, upsample=False, exact=False):
        self.flop_scale = flop_scale          # < 1 when rows/cols are zero padding (padded heads)
        self.f32 = f32                        # exact-f32 path: f32 weights, activations, output
        w = sd[f"{name}.weight"] if weight is None else weight
        b = sd.get(f"{name}.bias") if bias is None and weight is None else bias
        self.k = w.shape[2] if w.dim() == 4 else 1
        self.cout, self.cin = w.shape[0], w.shape[1]
        m = pack.conv_matrix(w)
        if glu:
            perm = pack.glu_perm(self.cout // 2)
        elif shuffle:
            perm = pack.shuffle_perm(self.cout)
        # f32 layers (the prior side): PRIOR_SPLIT16 -> weights as pairs of halves, products on the 16-bit matrix cores with
        # f32 activations and accumulation (onedc_conv_params.in_f32 == 2); exact=True keeps the f32 matrix instructions
        self.split16 = bool(f32 and PRIOR_SPLIT16 and not exact and self.cin % 32 == 0)
        if self.split16:
            self.w = pack.pack_matrix_split16(m, device, perm)
        else:
            self.w = pack.pack_matrix(m, device, perm, torch.float32 if f32 else K.a16())
        self.b = pack.f32(b, device, perm)
        self.glu, self.shuffle = glu, shuffle
        # conv behind a nearest x2 upsample: also keep the four 2x2 phase convolutions (2.25x fewer MACs); used
        # whenever the input map is whole 16x16 patches, the plain fused-upsample path otherwise
        self.w_ups = None
        if upsample and self.k == 3 and self.cout % 128 == 0 and self.cin % 64 == 0 and not f32:
            self.w_ups = pack.pack_matrix(pack.upsample_phase_matrix(w), device)
        self.mxfp8 = False
        self._mx_src = (m, perm, device) if (not f32 and not glu and not shuffle and self.cin % 32 == 0) else None
        # the same weights in bf16 for layers that may multiply bf16 operands (kernels.conv_gemm: w_bf16; asked for by the block
        # that owns the conv -- ResnetBlock2D -- through want_bf16_operands())
        self.w_bf = None
        self._bf_src = (m, perm, device) if (not f32 and not glu and not shuffle and self.k == 3) else None

    def want_bf16_operands(self):
        if self._bf_src is not None and self.w_bf is None and K.act() == "f16" and K.BF16_OPERANDS:
            m, perm, device = self._bf_src
            self.w_bf = pack.pack_matrix(m, device, perm, torch.bfloat16)
        self._bf_src = None
        return self

    def set_mxfp8(self, on=True):
        """MXFP8 experiment: the layer computes on operands quantised to e4m3 with one power-of-two scale per
        32 along K -- the weights once, here; the activations per call (kernels.mxfp8_roundtrip) -- through the SAME 16-bit
        kernel: the values, products and f32 accumulation of a block-scaled fp8 MFMA layer, without its speed."""
        if on and self._mx_src is None:
            raise ValueError("this layer has no MXFP8 form (f32 path, GLU / pixel-shuffle rows, or Cin % 32 != 0)")
        if on and not hasattr(self, "w_mx"):
            m, perm, device = self._mx_src
            self.w_mx = pack.pack_matrix(pack.mxfp8_roundtrip(m), device, perm)
        self.mxfp8 = bool(on)
        return self

    def __call__(self, x, y=None, *, epi=EPI_LINEAR, slope=0.01, res=None, stride=1, up=0, out_f32=False,
                 alpha=1.0, stats=False, pad=None, gn=None, x2=None, w2=None, bias=None, out_bf16=False):
        """stats: the output feeds a GroupNorm -- leave its partial sums in the epilogue.
        gn = (GroupNorm layer, workspace, silu): the conv reads that norm of x (fused into the operand staging
        where the kernel can, kernels.conv_gemm).
        x2, w2, bias: a second (1x1) source accumulated into the same output and the combined bias (kernels.conv_gemm);
        returns None, with nothing launched, when the kernel this layer runs on does not take one."""
        if x2 is not None and (self.mxfp8 or up or self.shuffle or self.glu):
            return None
        if y is None:
            oh, ow = x.H, x.W
            if stride == 2:
                oh, ow = (x.H + 1) // 2, (x.W + 1) // 2
                if pad == 0:                       # diffusers Downsample2D(padding=0): pad (0,1,0,1), conv stride 2
                    oh, ow = x.H // 2, x.W // 2
            if up:
                oh, ow = 2 * x.H, 2 * x.W
            c = self.cout
            if self.glu:
                c //= 2
            if self.shuffle:
                c //= 4
                oh, ow = 2 * oh, 2 * ow
            y = new_view(x.B, oh, ow, c, x.t.device, torch.float32 if (out_f32 or self.f32) else
                         torch.bfloat16 if out_bf16 else K.a16())
        if stats:
            y.want_stats()
        if self.mxfp8:
            if gn is not None:                     # the NORMALISED tensor is the operand: its own pass, then quantise
                x = gn[0](x, gn[1], gn[2])
                gn = None
            xq = K.mxfp8_roundtrip(x, new_view(x.B, x.H, x.W, x.C, x.t.device))
            return K.conv_gemm(xq, self.w_mx, self.b, y, KH=self.k, stride=stride, up=up, epi=epi, slope=slope,
                               alpha=alpha, res=res, Cin=self.cin, Cout=self.cout, flop_scale=self.flop_scale, stats=stats,
                               pad=pad)
        if (up and self.w_ups is not None and USE_UPSAMPLE_PHASES and res is None and x.H % 16 == 0 and
                x.W % 16 == 0 and not y.f32 and y.cs % 8 == 0 and y.co % 8 == 0 and
                epi in (EPI_LINEAR, EPI_LEAKY, EPI_SILU)):
            return K.conv_gemm(x, self.w_ups, self.b, y, KH=3, up=2, epi=epi, slope=slope, alpha=alpha, Cin=self.cin,
                               Cout=self.cout, stats=stats, out_bf16=out_bf16)
        if out_bf16 and K.act() != "bf16":
            raise K.KernelError("out_bf16 on a conv that does not run as upsample phases (only that kernel stores the other format)")
        if gn is not None:
            gn = (gn[0].g, gn[0].b, gn[0].eps, gn[2], gn[1])
        return K.conv_gemm(x, self.w, self.b if bias is None else bias, y, KH=self.k, stride=stride, up=up, epi=epi,
                           slope=slope, alpha=alpha, res=res, pixel_shuffle=int(self.shuffle), Cin=self.cin,
                           Cout=self.cout, flop_scale=self.flop_scale, stats=stats, pad=pad, gn=gn, x2=x2, w2=w2,
                           w_bf16=self.w_bf if gn is not None else None)


_Conv = Conv


class GroupNorm:
    def __init__(self, sd, name, device, eps):
        self.g = pack.f32(sd[f"{name}.weight"], device)
        self.b = pack.f32(sd[f"{name}.bias"], device)
        self.eps = eps

    def __call__(self, x, ws, silu, y=None):
        if y is None:
            y = new_view(x.B, x.H, x.W, x.C, x.t.device)
        return K.group_norm(x, self.g, self.b, y, self.eps, silu, ws)


class DepthConvBlock4:
    def __init__(self, sd, p, device, f32=False):
        self.f32 = f32
        Conv = lambda *a, **k: _Conv(*a, f32=f32, **k)          # noqa: E731
        self.conv1 = Conv(sd, f"{p}.block.0.conv1.0", device)
        w = sd[f"{p}.block.0.depth_conv.weight"]                       # (C,1,3,3) -> [9][C]
        self.dw_w = w[:, 0].permute(1, 2, 0).reshape(9, -1).float().contiguous().to(device)
        self.dw_b = pack.f32(sd[f"{p}.block.0.depth_conv.bias"], device)
        self.conv2 = Conv(sd, f"{p}.block.0.conv2", device)
        self.adaptor = Conv(sd, f"{p}.block.0.adaptor", device) if f"{p}.block.0.adaptor.weight" in sd else None
        self.ffn_in = Conv(sd, f"{p}.block.1.conv", device, glu=True)
        self.ffn_out = Conv(sd, f"{p}.block.1.conv_out", device)

    def __call__(self, x, y=None, out_f32=False):
        identity = self.adaptor(x) if self.adaptor is not None else x
        t = self.conv1(x, epi=EPI_LEAKY, slope=0.01)
        t2 = K.dwconv3x3(t, self.dw_w, self.dw_b,
                         new_view(t.B, t.H, t.W, t.C, t.t.device, torch.float32 if self.f32 else K.a16()))
        h = self.conv2(t2, res=identity)
        u = self.ffn_in(h, epi=EPI_GLU_LEAKY)
        return self.ffn_out(u, y, res=h, out_f32=out_f32)


class ResidualBlockUpsample:
    def __init__(self, sd, p, device, f32=False):
        Conv = lambda *a, **k: _Conv(*a, f32=f32, **k)          # noqa: E731
        self.subpel = Conv(sd, f"{p}.subpel_conv.0", device, shuffle=True)
        self.conv = Conv(sd, f"{p}.conv", device)
        self.upsample = Conv(sd, f"{p}.upsample.0", device, shuffle=True)

    def __call__(self, x, y=None):
        t = self.subpel(x, epi=EPI_LEAKY, slope=0.01)
        identity = self.upsample(x)
        return self.conv(t, y, epi=EPI_LEAKY, slope=0.1, res=identity)


class VQResnetBlock:
    def __init__(self, sd, p, device):
        self.n1 = GroupNorm(sd, f"{p}.norm1", device, 1e-6)
        self.c1 = Conv(sd, f"{p}.conv1", device)
        self.n2 = GroupNorm(sd, f"{p}.norm2", device, 1e-6)
        self.c2 = Conv(sd, f"{p}.conv2", device)

    def __call__(self, x, ws, y=None):
        h = self.c1(x, stats=True, gn=(self.n1, ws, True))
        return self.c2(h, y, res=x, stats=True, gn=(self.n2, ws, True))


class SingleHeadAttention:
    """softmax(q k^T / sqrt(C)) v over `n` tokens per group, unfused (scores in f32):
    VQ AttnBlock (1x1-conv projections) and the VAE mid-block Attention (Linear projections)."""

    def __init__(self, sd, names, device, C):
        qn, kn, vn, on = names
        self.C = C
        self.q = Conv(sd, qn, device)
        self.k = Conv(sd, kn, device)
        wv = sd[f"{vn}.weight"].reshape(C, C)
        self.wv = View(pack.pack_matrix(wv, device), 1, 1, C, C)       # V^T = Wv . h^T: Wv is the "pixel" operand
        # rows of softmax sum to 1, so the v bias rides on the output: P (V + 1 bv^T) = P V + bv
        self.bv = pack.f32(sd[f"{vn}.bias"], device)
        self.o = Conv(sd, on, device)

    def scale_output(self, s):
        """The block's output joins a residual stream that is stored times `s` (vae.py): out = s * (o(a) + b)."""
        self.out_scale = s
        self.o.b = (self.o.b * s).contiguous()

    def __call__(self, hn, res, n, y=None, ws=None):
        """hn: normalised tokens View with B groups of n = H*W tokens; res: residual View."""
        dev, C, G = hn.t.device, self.C, hn.B
        n_pad = (n + 63) // 64 * 64
        rows = G * n
        q = self.q(hn)
        k = self.k(hn)
        kbuf = k.t
        # V^T per group: [C][n_pad], zero padded keys (a workspace buffer of exactly this layout is zeroed once: the GEMM
        # below only ever writes the n real keys)
        if n == n_pad:
            vt = torch.empty(G * C * n_pad, dtype=K.a16(), device=dev)
        elif ws is not None:
            vt = ws.zeros(f"sha_vt_{G}_{C}_{n}", G * C * n_pad, K.a16())
        else:
            vt = torch.zeros(G * C * n_pad, dtype=K.a16(), device=dev)
        hn_flat = hn
        K.conv_gemm(self.wv, hn_flat.t, None, View(vt, 1, 1, C, n, n_pad), Cin=C, Cout=n, batch=G,
                    x_bs=0, w_bs=n * C, y_bs=C * n_pad)
        s = torch.empty(rows * n, dtype=torch.float32, device=dev)
        K.conv_gemm(View(q.t, 1, 1, n, C), kbuf, None, View(s, 1, 1, n, n), Cin=C, Cout=n, batch=G,
                    x_bs=n * C, w_bs=n * C, y_bs=n * n, alpha=C ** -0.5)
        p = torch.empty(rows * n_pad, dtype=K.a16(), device=dev)
        K.softmax_rows(s, p, rows, n, n_pad)
        a = torch.empty(rows * C, dtype=K.a16(), device=dev)
        K.conv_gemm(View(p, 1, 1, n, n_pad), vt, self.bv, View(a, 1, 1, n, C), Cin=n_pad, Cout=C, batch=G,
                    x_bs=n * n_pad, w_bs=C * n_pad, y_bs=n * C)
        return self.o(View(a, hn.B, hn.H, hn.W, C), y, res=res, alpha=getattr(self, "out_scale", 1.0))


class VQAttnBlock:
    def __init__(self, sd, p, device, C):
        self.norm = GroupNorm(sd, f"{p}.norm", device, 1e-6)
        self.attn = SingleHeadAttention(sd, (f"{p}.q", f"{p}.k", f"{p}.v", f"{p}.proj_out"), device, C)

    def __call__(self, x, ws, y=None):
        return self.attn(self.norm(x, ws, False), x, x.H * x.W, y, ws)


class ResnetBlock2D:
    """diffusers ResnetBlock2D; `temb` (B-independent, timestep is a constant of the codec) is
    folded into conv1's bias by `set_temb`."""

    def __init__(self, sd, p, device, eps, stream_scale=1.0, bf16_operands=True):
        """bf16_operands: conv1 / conv2 may multiply bf16 operands where the 4-wave window kernel runs them (kernels.BF16_OPERANDS:
        their inputs are GroupNorm outputs, intermediates nobody else reads; everything stored stays f16).  The decode path's
        blocks do (+2.3 % images/s for 0.0003 dB); the ENCODE side's blocks do not: they write the streams, and their arithmetic
        stays what the parity figures of rounds 2-4 were measured on.
        stream_scale s (a power of two): the residual stream this block reads and writes is stored times s.
        GroupNorm is scale invariant up to eps (norm1 gets eps * s^2), the branch output and the biases of the linear
        maps of the stream are multiplied by s in the epilogue (alpha) / at load."""
        self.p, self.device = p, device
        self.stream_scale = s = float(stream_scale)
        cin = sd[f"{p}.norm1.weight"].shape[0]
        self.n1 = GroupNorm(sd, f"{p}.norm1", device, eps * s * s)
        self.c1 = Conv(sd, f"{p}.conv1", device)
        self.n2 = GroupNorm(sd, f"{p}.norm2", device, eps)
        self.c2 = Conv(sd, f"{p}.conv2", device)
        if bf16_operands:                                                   # both convs read a GroupNorm output: an intermediate
            self.c1.want_bf16_operands()
            self.c2.want_bf16_operands()
        self.short = Conv(sd, f"{p}.conv_shortcut", device) if f"{p}.conv_shortcut.weight" in sd else None
        if s != 1.0:
            self.c2.b = (self.c2.b * s).contiguous()
            if self.short is not None:
                self.short.b = (self.short.b * s).contiguous()
        # conv_shortcut riding on conv2 (onedc_conv_params.x2): its weights divided by conv2's alpha (a power of two), the
        # two biases added; used wherever conv2 runs on the 4-wave window kernel (the dispatcher is asked per call)
        self.short_w2 = self.short_b2 = None
        if self.short is not None and self.short.k == 1 and FUSE_SHORTCUT:
            self.short_w2 = pack.pack_matrix(pack.conv_matrix(sd[f"{p}.conv_shortcut.weight"]) / s, device)
            self.short_b2 = (self.c2.b + self.short.b).contiguous()
        self.cin = cin
        self._c1_bias = sd[f"{p}.conv1.bias"].float()
        if f"{p}.time_emb_proj.weight" in sd:
            self._tw, self._tb = sd[f"{p}.time_emb_proj.weight"].float(), sd[f"{p}.time_emb_proj.bias"].float()
        else:
            self._tw = None

    def set_temb(self, temb_act):
        """temb_act = SiLU(time_embedding(t)) as a CPU f32 vector [1280]."""
        if self._tw is not None:
            self.c1.b = (self._c1_bias + self._tw @ temb_act.to(self._tw.device) + self._tb).contiguous().to(self.device)

    def __call__(self, x, ws, y=None):
        h = self.c1(x, stats=True, gn=(self.n1, ws, True))
        if self.short_w2 is not None and FUSE_SHORTCUT:
            out = self.c2(h, y, stats=True, gn=(self.n2, ws, True), alpha=self.stream_scale, x2=x, w2=self.short_w2,
                          bias=self.short_b2)
            if out is not None:
                return out
        res = self.short(x) if self.short is not None else x
        return self.c2(h, y, res=res, stats=True, gn=(self.n2, ws, True), alpha=self.stream_scale)


# Compact heads (round 5): q / k / V^T / o tensors of the d = 40 and d = 80 attention layers hold the REAL head dims back to back
# (the flash kernels make the padding on their DMA path: onedc_flash_attn with DQ = DV = d), so the projections compute and move
# no padded columns -- to_q|k 320 -> 640 instead of -> 1024, to_out with K = 320 instead of 384.  Bit-identical to the padded
# layout (the MFMAs see the same values).  Only the two ring kernels read it, i.e. 64 keys and more: a layer keeps both packings
# and decides per call.  ONEDC_COMPACT_HEADS=0: the padded layout everywhere (A/B, tests).
COMPACT_HEADS = __import__("os").environ.get("ONEDC_COMPACT_HEADS", "1") != "0"


class MultiHeadAttention:
    """diffusers Attention (8 heads) on flash attention; the projections zero-pad the head dim, or -- head dims 40 / 80 with 64
    keys and more -- leave it compact (COMPACT_HEADS above).

    Self-attention (`ctx_dim == C`, keys = queries): to_q and to_k are ONE GEMM whose output rows are [q | k]
    (`onedc_flash_attn_strided` reads both as column slices of it).  Cross-attention: K and V^T depend on the context only,
    which all 16 transformer blocks of the UNet share -- `ContextKV` projects it for all of them in one K GEMM and one or
    two V^T GEMMs per UNet call, and this layer is handed its slices (`kv`)."""

    def __init__(self, sd, p, device, C, ctx_dim, heads=8, self_attention=None):
        d = C // heads
        self.H, self.d = heads, d
        self.DQ, self.DV = pack.HEAD_PAD[d]
        self.C, self.ctx_dim = C, ctx_dim
        self.self_attention = (ctx_dim == C) if self_attention is None else self_attention
        self.compact = COMPACT_HEADS and d in (40, 80)
        wq = pack.pad_heads_rows(sd[f"{p}.to_q.weight"], heads, d, self.DQ)
        self.wk_rows = pack.pad_heads_rows(sd[f"{p}.to_k.weight"], heads, d, self.DQ)      # (kept on the host for ContextKV)
        self.wv_rows = pack.pad_heads_rows(sd[f"{p}.to_v.weight"], heads, d, self.DV)
        self.wk_rows_c = self.wv_rows_c = None
        if self.self_attention:
            self.qk = Conv(sd, "", device, weight=torch.cat([wq, self.wk_rows], 0), flop_scale=d / self.DQ)
            self.wv = View(pack.pack_matrix(self.wv_rows, device), 1, 1, heads * self.DV, ctx_dim)
            self.wk_rows = self.wv_rows = None
            if self.compact:
                self.qk_c = Conv(sd, "", device, weight=torch.cat([sd[f"{p}.to_q.weight"], sd[f"{p}.to_k.weight"]], 0))
                self.wv_c = View(pack.pack_matrix(sd[f"{p}.to_v.weight"], device), 1, 1, heads * d, ctx_dim)
        else:
            self.q = Conv(sd, "", device, weight=wq, flop_scale=d / self.DQ)
            if self.compact:
                self.q_c = Conv(sd, "", device, weight=sd[f"{p}.to_q.weight"])
                self.wk_rows_c, self.wv_rows_c = sd[f"{p}.to_k.weight"], sd[f"{p}.to_v.weight"]   # (for ContextKV)
            # the fused block (block_fused) takes to_q with the LayerNorm's affine folded in: the f32 matrix is kept on the host
            # until the first call knows that norm (only where the kernel exists: the UNet's first level)
            self._wq_host = sd[f"{p}.to_q.weight"].detach().float().cpu() if (self.compact and C == 320 and heads == 8) else None
            self._xq = self._xbq = self._x_ln = None
        self.o = Conv(sd, "", device, weight=pack.pad_heads_cols(sd[f"{p}.to_out.0.weight"], heads, d, self.DV),
                      bias=sd[f"{p}.to_out.0.bias"], flop_scale=d / self.DV)
        if self.compact:
            self.o_c = Conv(sd, "", device, weight=sd[f"{p}.to_out.0.weight"], bias=sd[f"{p}.to_out.0.bias"])

    def __call__(self, x, ctx, res, y=None, ws=None, kv=None):
        """x: query tokens View (B, H, W, C); ctx: key/value tokens View; res: residual.
        kv = (k tensor, k row stride, V^T tensor, V^T image stride, compact) from ContextKV (cross-attention)."""
        dev = x.t.device
        B, Nq, Nk = x.B, x.H * x.W, ctx.H * ctx.W
        Nk_pad = (Nk + 63) // 64 * 64
        if self.self_attention or kv is None:
            compact = self.compact and Nk >= 64
        else:
            compact = bool(kv[4])
        DQ, DV = (self.d, self.d) if compact else (self.DQ, self.DV)
        HDQ, HDV = self.H * DQ, self.H * DV
        if self.self_attention:
            qk = (self.qk_c if compact else self.qk)(x)              # rows [q (H DQ) | k (H DQ)]
            q_t, k_t, q_rs, k_rs = qk.t, qk.t[HDQ:], 2 * HDQ, 2 * HDQ
            vt_dtype = K.flash_vt_dtype(DQ, DV, Nk)
            n = B * HDV * Nk_pad
            if Nk == Nk_pad:
                vt = torch.empty(n, dtype=vt_dtype, device=dev)
            elif ws is not None:                                     # zero padding keys: zeroed once per layout
                vt = ws.zeros(f"mha_vt_{B}_{HDV}_{Nk}_{vt_dtype}", n, vt_dtype)
            else:
                vt = torch.zeros(n, dtype=vt_dtype, device=dev)
            K.conv_gemm(self.wv_c if compact else self.wv, ctx.t, None, View(vt, 1, 1, HDV, Nk, Nk_pad), Cin=self.ctx_dim,
                        Cout=Nk, batch=B, x_bs=0, w_bs=Nk * ctx.cs, y_bs=HDV * Nk_pad,
                        flop_scale=self.d / DV, out_bf16=(vt_dtype == torch.bfloat16))
            vt_bs = 0
        else:
            if kv is None:                       # stand-alone use: this layer's own projections, packed once
                if getattr(self, "_own_kv", None) is None:
                    if self.wk_rows is None:
                        raise RuntimeError("this cross-attention layer's to_k / to_v live in a shared ContextKV: pass its `kv`")
                    self._own_kv = ContextKV([self], dev)
                kv = self._own_kv(ctx, ws)[0]
                compact = bool(kv[4])
                DQ, DV = (self.d, self.d) if compact else (self.DQ, self.DV)
                HDQ, HDV = self.H * DQ, self.H * DV
            k_t, k_rs, vt, vt_bs = kv[:4]
            q = (self.q_c if compact else self.q)(x)
            q_t, q_rs = q.t, HDQ
        o = new_view(x.B, x.H, x.W, HDV, dev)
        K.flash_attn(q_t, k_t, vt, o.t, B, self.H, Nq, Nk, Nk_pad, DQ, DV, self.d ** -0.5,
                     q_rs=q_rs, k_rs=k_rs, vt_bs=vt_bs)
        return (self.o_c if compact else self.o)(o, y, res=res)

    def prepare_fused(self, ln, device):
        """Called once by the block that owns this layer and the LayerNorm `ln` in front of it (Transformer2D.__init__; not lazily:
        the pipeline's stream threads run the same layers side by side).  to_q(LayerNorm(h)) = (Wq . diag(gamma)) . hhat + Wq . beta
        with hhat = (h - mean) * rstd: the fused kernel centres and scales, the affine rides in the weights (rounded to 16 bits
        once, like Wq itself) and in a bias of q."""
        if self._wq_host is None:
            return self
        gam, bet = ln.g.detach().float().cpu(), ln.b.detach().float().cpu()
        self._xq = pack.pack_matrix(self._wq_host * gam[None, :], device)
        self._xbq = pack.f32(self._wq_host @ bet, device)
        self._wq_host, self._x_ln = None, ln
        return self

    def block_fused(self, h, ln, ctx, kv, y=None):
        """Cross-attention only: y = h + to_out(attention(to_q(LayerNorm(h)), K, V)) as ONE launch (kernels.xattn_block) --
        `ln` is the block's LayerNorm in front of this layer (the one prepare_fused folded into to_q), kv its slices of the shared
        context projections.  Returns None, with nothing launched, where the kernel does not take the layer (the caller runs
        LayerNorm + __call__)."""
        if self.self_attention or kv is None or not kv[4] or not self.compact or not K.USE_XATTN_BLOCK:
            return None
        Nk = ctx.H * ctx.W
        Nk_pad = (Nk + 63) // 64 * 64
        k_t, k_rs, vt, vt_bs = kv[:4]
        if y is None:
            y = new_view(h.B, h.H, h.W, self.C, h.t.device)
        if y.t.data_ptr() == h.t.data_ptr():
            return None                                                # (the kernel re-reads h for the residual: no aliasing)
        if self._xq is None or ln is not self._x_ln:
            return None                                                # (prepare_fused was not called for this block's norm)
        p = K.xattn_params(h, None, None, 1e-5, self._xq, self._xbq, k_t, k_rs, vt, vt_bs, self.o_c.w, self.o_c.b, y, self.H,
                           self.d, Nk, Nk_pad, self.d ** -0.5)
        if h.C != self.C or not K.xattn_block_ok(p):
            return None
        K.xattn_block(p)
        return y


class _ContextKVSet:
    """One packing of the shared context projections: K of all layers as one GEMM, V^T of all layers as one batched GEMM (two in
    the IEEE-half build: the d = 40 / 80 kernels read V^T as bf16, the d = 160 kernel as f16)."""

    def __init__(self, layers, order, device, compact):
        self.compact = compact
        self.ctx_dim = layers[0].ctx_dim
        wk, wv, self.k_off, self.v_off = [], [], {}, {}
        ko = vo = 0
        real = padded = 0
        for i in order:
            l = layers[i]
            c = compact and l.compact
            wk.append(l.wk_rows_c if c else l.wk_rows)
            wv.append(l.wv_rows_c if c else l.wv_rows)
            self.k_off[i], self.v_off[i] = ko, vo
            ko += l.H * (l.d if c else l.DQ)
            vo += l.H * (l.d if c else l.DV)
            real += l.H * l.d
            padded += l.H * (l.d if c else l.DQ)
        self.k_cols, self.v_rows = ko, vo
        self.v_split = sum(l.H * (l.d if (compact and l.compact) else l.DV) for l in layers if l.d < 160)   # rows [0, v_split): head dims 40 / 80
        self.k = Conv({}, "", device, weight=torch.cat(wk, 0), flop_scale=real / padded)
        self.wv = View(pack.pack_matrix(torch.cat(wv, 0), device), 1, 1, vo, self.ctx_dim)
        self.v_scale = sum(l.H * l.d for l in layers) / vo


class ContextKV:
    """to_k / to_v of every cross-attention layer that reads ONE context (the UNet's 16 transformer blocks all attend to
    y_semantic: reference decoder_unet.py:204-293 passes the same `encoder_hidden_states` down): K of all layers is one GEMM
    [tokens][sum H DQ], V^T of all layers one batched GEMM [image][sum H DV][keys] (two in the IEEE-half build when the
    d = 40 / 80 kernels take their V^T as bf16 and the d = 160 kernel as f16) -- 3 launches per UNet step instead of 32.
    Two packings where layers have compact heads (COMPACT_HEADS): the padded one for contexts of fewer than 64 tokens."""

    def __init__(self, layers, device):
        # V^T rows: the layers whose kernel may want bf16 V^T first (head dims 40, 80), so that each dtype is one row range
        self.layers = list(layers)
        order = sorted(range(len(self.layers)), key=lambda i: (self.layers[i].d >= 160, i))
        self.ctx_dim = self.layers[0].ctx_dim
        self.padded = _ContextKVSet(self.layers, order, device, False)
        self.compact = _ContextKVSet(self.layers, order, device, True) if any(l.compact for l in self.layers) else None
        for l in self.layers:
            l.wk_rows = l.wv_rows = l.wk_rows_c = l.wv_rows_c = None              # packed: drop the host copies

    def __call__(self, ctx, ws=None):
        """-> per layer (k tensor, k row stride, V^T tensor, V^T image stride, compact), in the order the layers were given."""
        dev = ctx.t.device
        B, Nk = ctx.B, ctx.H * ctx.W
        Nk_pad = (Nk + 63) // 64 * 64
        S = self.compact if (self.compact is not None and Nk >= 64) else self.padded
        k = S.k(ctx)                                                # (B, h, w, sum H DQ)
        small = [l for l in self.layers if l.d < 160]
        dqv = (lambda l: (l.d, l.d) if (S.compact and l.compact) else (l.DQ, l.DV))
        bf16_rows = S.v_split if small and K.flash_vt_dtype(*dqv(small[0]), Nk) == torch.bfloat16 else 0
        if K.a16() == torch.bfloat16:
            bf16_rows = 0                                           # one format anyway
        # ONE buffer of 16-bit words for all layers (bf16 and f16 are both 2 bytes: the row ranges are written as what
        # their readers expect)
        n = B * S.v_rows * Nk_pad
        if Nk == Nk_pad:
            vt = torch.empty(n, dtype=K.a16(), device=dev)
        elif ws is not None:
            vt = ws.zeros(f"ctx_vt_{B}_{S.v_rows}_{Nk}", n, K.a16())
        else:
            vt = torch.zeros(n, dtype=K.a16(), device=dev)
        vt_bs = S.v_rows * Nk_pad
        for r0, r1, as_bf16 in ((0, bf16_rows, True), (bf16_rows, S.v_rows, False)):
            if r1 > r0:
                K.conv_gemm(View(S.wv.t.view(-1)[r0 * self.ctx_dim:], 1, 1, r1 - r0, self.ctx_dim), ctx.t, None,
                            View(vt[r0 * Nk_pad:], 1, 1, r1 - r0, Nk, Nk_pad), Cin=self.ctx_dim, Cout=Nk, batch=B, x_bs=0,
                            w_bs=Nk * ctx.cs, y_bs=vt_bs, flop_scale=S.v_scale, out_bf16=as_bf16)
        out = []
        for i, l in enumerate(self.layers):
            vt_l = vt[S.v_off[i] * Nk_pad:]
            if S.v_off[i] < bf16_rows:
                vt_l = vt_l.view(torch.bfloat16)                    # same bytes, the dtype its kernel asks for
            out.append((k.t[S.k_off[i]:], S.k_cols, vt_l, vt_bs, S.compact and l.compact))
        return out


class LayerNorm:
    def __init__(self, sd, name, device):
        self.g = pack.f32(sd[f"{name}.weight"], device)
        self.b = pack.f32(sd[f"{name}.bias"], device)

    def __call__(self, x):
        y = new_view(x.B, x.H, x.W, x.C, x.t.device)
        K.layernorm(x.t, self.g, self.b, y.t, x.B * x.H * x.W, x.C)
        return y


class Transformer2D:
    """diffusers Transformer2DModel (conv projections) with one BasicTransformerBlock (GEGLU FF)."""

    def __init__(self, sd, p, device, C, ctx_dim=768):
        t = f"{p}.transformer_blocks.0"
        self.norm = GroupNorm(sd, f"{p}.norm", device, 1e-6)
        self.proj_in = Conv(sd, f"{p}.proj_in", device)
        self.n1, self.n2, self.n3 = (LayerNorm(sd, f"{t}.norm{i}", device) for i in (1, 2, 3))
        self.attn1 = MultiHeadAttention(sd, f"{t}.attn1", device, C, C, self_attention=True)
        self.attn2 = MultiHeadAttention(sd, f"{t}.attn2", device, C, ctx_dim, self_attention=False).prepare_fused(self.n2, device)
        self.ff1 = Conv(sd, f"{t}.ff.net.0.proj", device, glu=True)
        self.ff2 = Conv(sd, f"{t}.ff.net.2", device)
        self.proj_out = Conv(sd, f"{p}.proj_out", device)

    def __call__(self, x, ctx, ws, y=None, kv=None):
        """kv: this block's slices of the shared context projections (ContextKV), or None to project here."""
        h = self.proj_in(self.norm(x, ws, False))
        n = self.n1(h)
        h = self.attn1(n, n, h, ws=ws)
        # norm2 -> attn2 -> + h: one launch where the fused kernel takes the level (csrc/xattn_block.hip), four otherwise
        h2 = self.attn2.block_fused(h, self.n2, ctx, kv) if kv is not None else None
        h = h2 if h2 is not None else self.attn2(self.n2(h), ctx, h, ws=ws, kv=kv)
        f = self.ff1(self.n3(h), epi=EPI_GEGLU)
        h = self.ff2(f, res=h)
        return self.proj_out(h, y, res=x, stats=True)


def silu_cpu(v):
    return v * torch.sigmoid(v)


def timestep_embedding_cpu(t, dim=320):
    half = dim // 2
    exponent = -math.log(10000.0) * torch.arange(0, half, dtype=torch.float32) / half
    emb = torch.tensor([float(t)], dtype=torch.float32)[:, None] * torch.exp(exponent)[None, :]
    return torch.cat([torch.cos(emb), torch.sin(emb)], dim=-1)[0]
