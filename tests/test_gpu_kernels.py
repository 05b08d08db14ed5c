"""Kernel-level numerics on a real MI355X: every HIP kernel (called through the C ABI) against a
plain PyTorch fp32 reference of the same op.  Inputs and weights are rounded to bf16 first, so the
only differences left are accumulation order (fp32 MFMA) and the final bf16 store.

Tolerances (written here, checked by `close`): relative RMS error <= 4e-3 (bf16 has 8 significant
bits: a rounded output carries ~2e-3 relative error) and max abs error <= 2.5e-2 * max|ref|.
"""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu

from onedc_amd import kernels as K
from onedc_amd import pack
from onedc_amd.kernels import View, from_nchw, new_view

DEV = "cuda:0"


def rnd(*shape, seed=0, scale=1.0):
    g = torch.Generator().manual_seed(seed)
    # representable in bf16 AND in the current 16-bit activation format (f16 has subnormals where bf16 still has 8 bits)
    return (torch.randn(*shape, generator=g) * scale).to(torch.bfloat16).float().to(K.a16()).float()


def close(got, ref, rms=4e-3, mx=2.5e-2, what=""):
    got, ref = got.double().cpu(), ref.double().cpu()
    assert got.shape == ref.shape, (got.shape, ref.shape)
    assert torch.isfinite(got).all(), f"{what}: non-finite output"
    err = got - ref
    scale = ref.abs().max().clamp_min(1e-6)
    rel_rms = float(err.pow(2).mean().sqrt() / ref.pow(2).mean().sqrt().clamp_min(1e-9))
    rel_max = float(err.abs().max() / scale)
    assert rel_rms <= rms and rel_max <= mx, f"{what}: rel_rms={rel_rms:.2e} rel_max={rel_max:.2e}"


def packed(w, perm=None):
    return pack.pack_matrix(pack.conv_matrix(w), DEV, perm)


@pytest.mark.parametrize("B,H,W,Cin,Cout,k", [
    (2, 16, 16, 128, 128, 3),     # VAE-style 3x3, 128x128 tile
    (1, 24, 20, 64, 320, 3),      # Cout=320 -> 64-wide channel tile, ragged M
    (2, 12, 12, 256, 256, 1),     # 1x1
    (1, 9, 7, 192, 128, 3),       # odd spatial size: halo + tail rows
    (3, 8, 8, 1280, 640, 1),      # long K
])
def test_conv_gemm_plain(B, H, W, Cin, Cout, k):
    x, w, b = rnd(B, Cin, H, W, seed=1), rnd(Cout, Cin, k, k, seed=2, scale=(Cin * k * k) ** -0.5), rnd(Cout, seed=3)
    y = new_view(B, H, W, Cout, DEV)
    K.conv_gemm(from_nchw(x.to(DEV)), packed(w), b.to(DEV), y, KH=k)
    close(y.torch_nchw(), F.conv2d(x, w, b, padding=k // 2), what="conv")


def test_conv_gemm_stride2_and_upsample():
    x, w, b = rnd(2, 128, 16, 12, seed=4), rnd(192, 128, 3, 3, seed=5, scale=1 / 34), rnd(192, seed=6)
    xv = from_nchw(x.to(DEV))
    y = new_view(2, 8, 6, 192, DEV)
    K.conv_gemm(xv, packed(w), b.to(DEV), y, KH=3, stride=2)
    close(y.torch_nchw(), F.conv2d(x, w, b, stride=2, padding=1), what="stride2")
    y = new_view(2, 32, 24, 192, DEV)
    K.conv_gemm(xv, packed(w), b.to(DEV), y, KH=3, up=1)
    close(y.torch_nchw(), F.conv2d(F.interpolate(x, scale_factor=2.0, mode="nearest"), w, b, padding=1), what="up")


def test_conv_gemm_epilogues_residual_slices():
    B, H, W, Cin, Cout = 2, 10, 10, 128, 128
    x, w, b = rnd(B, Cin, H, W, seed=7), rnd(Cout, Cin, 1, 1, seed=8, scale=Cin ** -0.5), rnd(Cout, seed=9)
    r = rnd(B, Cout, H, W, seed=10)
    # input read from a channel slice, output + residual in slices of wider buffers
    xbuf = from_nchw(torch.cat([rnd(B, 64, H, W, seed=11), x], 1).to(DEV))
    rbuf = from_nchw(torch.cat([r, rnd(B, 8, H, W, seed=12)], 1).to(DEV))
    ybuf = new_view(B, H, W, Cout + 64, DEV)
    ybuf.t.zero_()
    K.conv_gemm(xbuf.slice(64, Cin), packed(w), b.to(DEV), ybuf.slice(64, Cout), KH=1, epi=K.EPI_LEAKY,
                slope=0.1, res=rbuf.slice(0, Cout))
    close(ybuf.slice(64, Cout).torch_nchw(), F.leaky_relu(F.conv2d(x, w, b), 0.1) + r, what="leaky+res")
    assert float(ybuf.slice(0, 64).torch_nchw().abs().max()) == 0.0       # neighbours untouched
    y = new_view(B, H, W, Cout, DEV)
    K.conv_gemm(from_nchw(x.to(DEV)), packed(w), b.to(DEV), y, KH=1, epi=K.EPI_SILU, alpha=0.5)
    close(y.torch_nchw(), F.silu(0.5 * F.conv2d(x, w) + b.view(1, -1, 1, 1)), what="silu/alpha")


@pytest.mark.parametrize("epi", ["glu_leaky", "geglu"])
def test_conv_gemm_glu(epi):
    B, H, W, Cin, Ch = 1, 12, 12, 128, 256
    x, w, b = rnd(B, Cin, H, W, seed=13), rnd(2 * Ch, Cin, 1, 1, seed=14, scale=Cin ** -0.5), rnd(2 * Ch, seed=15)
    perm = pack.glu_perm(Ch)
    y = new_view(B, H, W, Ch, DEV)
    K.conv_gemm(from_nchw(x.to(DEV)), packed(w, perm), b[perm].to(DEV), y, KH=1,
                epi=K.EPI_GLU_LEAKY if epi == "glu_leaky" else K.EPI_GEGLU)
    a, g = F.conv2d(x, w, b).chunk(2, 1)
    ref = F.leaky_relu(a, 0.1) + F.leaky_relu(g, 0.01) if epi == "glu_leaky" else a * F.gelu(g)
    close(y.torch_nchw(), ref, what=epi)


def test_conv_gemm_pixel_shuffle():
    B, H, W, Cin, Cq = 2, 6, 5, 128, 64
    x, w, b = rnd(B, Cin, H, W, seed=16), rnd(4 * Cq, Cin, 1, 1, seed=17, scale=Cin ** -0.5), rnd(4 * Cq, seed=18)
    perm = pack.shuffle_perm(4 * Cq)
    y = new_view(B, 2 * H, 2 * W, Cq, DEV)
    K.conv_gemm(from_nchw(x.to(DEV)), packed(w, perm), b[perm].to(DEV), y, KH=1, pixel_shuffle=1)
    close(y.torch_nchw(), F.pixel_shuffle(F.conv2d(x, w, b), 2), what="pixel_shuffle")


@pytest.mark.parametrize("Cout", [3, 4])
def test_conv_gemm_tiny_cout_f32(Cout):
    """conv_out of the VAE (128->3) and the UNet (320->4): 16-channel tile, f32 output."""
    B, H, W, Cin = 1, 20, 24, 128
    x, w, b = rnd(B, Cin, H, W, seed=19), rnd(Cout, Cin, 3, 3, seed=20, scale=1 / 34), rnd(Cout, seed=21)
    y = new_view(B, H, W, Cout, DEV, torch.float32)
    K.conv_gemm(from_nchw(x.to(DEV)), packed(w), b.to(DEV), y, KH=3)
    close(y.torch_nchw(), F.conv2d(x, w, b, padding=1), rms=1e-4, mx=1e-4, what="tiny cout f32")


def test_conv_gemm_batched_nt():
    """Batched activation x activation product (QK^T of the single-head attention)."""
    G, n, Cc = 5, 144, 128
    q, k = rnd(G, n, Cc, seed=22), rnd(G, n, Cc, seed=23)
    qd = torch.zeros(G * n + 128, Cc, dtype=K.a16(), device=DEV)
    kd = torch.zeros(G * n + 128, Cc, dtype=K.a16(), device=DEV)
    qd[:G * n] = q.view(-1, Cc).to(DEV)
    kd[:G * n] = k.view(-1, Cc).to(DEV)
    s = torch.empty(G * n * n, dtype=torch.float32, device=DEV)
    K.conv_gemm(View(qd, 1, 1, n, Cc), kd, None, View(s, 1, 1, n, n), Cin=Cc, Cout=n, batch=G, x_bs=n * Cc,
                w_bs=n * Cc, y_bs=n * n, alpha=0.25)
    close(s.view(G, n, n), 0.25 * q @ k.transpose(1, 2), rms=1e-4, mx=1e-4, what="batched nt")


def test_conv_gemm_batch_invariant():
    """Same image alone or inside a batch -> bit-identical output (determinism requirement)."""
    x, w = rnd(4, 128, 12, 12, seed=24), rnd(256, 128, 3, 3, seed=25, scale=1 / 34)
    pw = packed(w)
    y4 = new_view(4, 12, 12, 256, DEV)
    K.conv_gemm(from_nchw(x.to(DEV)), pw, None, y4, KH=3)
    y1 = new_view(1, 12, 12, 256, DEV)
    K.conv_gemm(from_nchw(x[2:3].to(DEV)), pw, None, y1, KH=3)
    assert torch.equal(y4.torch_nchw()[2:3], y1.torch_nchw())


def test_conv_gemm_rejects_bad_arguments():
    y = new_view(1, 4, 4, 128, DEV)
    with pytest.raises(K.KernelError):
        K.conv_gemm(new_view(1, 4, 4, 100, DEV), torch.zeros(128, 100, dtype=K.a16(), device=DEV), None, y)


class _WS:
    def __init__(self):
        self.t = {}

    def f32(self, name, n):
        if name not in self.t or self.t[name].numel() < n:
            self.t[name] = torch.empty(n, dtype=torch.float32, device=DEV)
        return self.t[name]


@pytest.mark.parametrize("B,C,H,W,silu", [(2, 128, 40, 36, True), (1, 320, 9, 9, False), (2, 1920, 6, 6, True),
                                          (1, 2560, 4, 4, True), (3, 512, 16, 16, False), (2, 1280, 12, 12, True),
                                          (2, 768, 12, 12, False)])       # (the last four: the single-launch statistics of small maps)
def test_group_norm(B, C, H, W, silu):
    x, g, b = rnd(B, C, H, W, seed=26) * 2 + 0.5, rnd(C, seed=27), rnd(C, seed=28)
    x = x.to(torch.bfloat16).float()
    y = new_view(B, H, W, C, DEV)
    K.group_norm(from_nchw(x.to(DEV)), g.to(DEV), b.to(DEV), y, 1e-6, silu, _WS())
    ref = F.group_norm(x, 32, g, b, 1e-6)
    close(y.torch_nchw(), F.silu(ref) if silu else ref, what="group_norm")
    if B > 1:                                                                  # a picture alone == inside the batch (bits)
        y1 = new_view(1, H, W, C, DEV)
        K.group_norm(from_nchw(x[B - 1:].to(DEV)), g.to(DEV), b.to(DEV), y1, 1e-6, silu, _WS())
        assert torch.equal(y1.t[:H * W * C], y.t[(B - 1) * H * W * C:B * H * W * C])


def test_group_norm_large_map_many_chunks():
    x, g, b = rnd(1, 128, 96, 128, seed=29), rnd(128, seed=30), rnd(128, seed=31)
    y = new_view(1, 96, 128, 128, DEV)
    K.group_norm(from_nchw(x.to(DEV)), g.to(DEV), b.to(DEV), y, 1e-5, True, _WS())
    close(y.torch_nchw(), F.silu(F.group_norm(x, 32, g, b, 1e-5)), what="group_norm chunks")


@pytest.mark.parametrize("C", [320, 640, 1280])
def test_layernorm(C):
    x, g, b = rnd(300, C, seed=32) * 3, rnd(C, seed=33), rnd(C, seed=34)
    y = torch.empty(300, C, dtype=K.a16(), device=DEV)
    K.layernorm(x.to(K.a16()).to(DEV), g.to(DEV), b.to(DEV), y, 300, C)
    close(y.float(), F.layer_norm(x, (C,), g, b, 1e-5), what="layernorm")


@pytest.mark.parametrize("n,n_pad", [(144, 192), (256, 256), (16, 64)])
def test_softmax_rows(n, n_pad):
    s = torch.randn(77, n, generator=torch.Generator().manual_seed(35)) * 4
    p = torch.empty(77, n_pad, dtype=K.a16(), device=DEV)
    K.softmax_rows(s.to(DEV), p, 77, n, n_pad)
    close(p[:, :n].float(), torch.softmax(s, -1), what="softmax")
    assert float(p[:, n:].float().abs().max() if n_pad > n else 0.0) == 0.0


def test_dwconv3x3():
    B, C, H, W = 2, 256, 11, 13
    x, w, b = rnd(B, C, H, W, seed=36), torch.randn(C, 1, 3, 3, generator=torch.Generator().manual_seed(37)), rnd(C, seed=38)
    y = new_view(B, H, W, C, DEV)
    K.dwconv3x3(from_nchw(x.to(DEV)), w[:, 0].permute(1, 2, 0).reshape(9, C).contiguous().to(DEV), b.to(DEV), y)
    close(y.torch_nchw(), F.conv2d(x, w, b, padding=1, groups=C), what="dwconv")


def test_direct_conv_small_cin():
    x = torch.randn(2, 4, 12, 12, generator=torch.Generator().manual_seed(39))
    w, b = torch.randn(512, 4, 3, 3, generator=torch.Generator().manual_seed(40)) * 0.2, rnd(512, seed=41)
    xv = View(x.permute(0, 2, 3, 1).contiguous().to(DEV).reshape(-1), 2, 12, 12, 4)
    y = new_view(2, 12, 12, 512, DEV)
    K.direct_conv(xv, w.permute(0, 2, 3, 1).reshape(512, -1).contiguous().to(DEV), b.to(DEV), y, K=3, in_scale=1 / 0.18215)
    close(y.torch_nchw(), F.conv2d(x / 0.18215, w, b, padding=1), what="direct conv")


@pytest.mark.parametrize("d,Nq,Nk", [(40, 300, 300), (40, 200, 144), (80, 576, 576), (80, 300, 144), (80, 130, 16),
                                     (160, 144, 144), (160, 130, 77)])
def test_flash_attention(d, Nq, Nk):
    """d = 80 is laid out as two half-heads (pack.head_positions); with fewer than 64 keys it takes the generic kernel."""
    B, Hh = 2, 8
    DQ, DV = pack.HEAD_PAD[d]
    pq, pv = pack.head_positions(d, DQ), pack.head_positions(d, DV)
    g = torch.Generator().manual_seed(42)
    q, k, v = (torch.randn(B, Hh, n, d, generator=g).to(torch.bfloat16).float() for n in (Nq, Nk, Nk))
    Nk_pad = (Nk + 63) // 64 * 64

    def padded(t, dp):                      # (B,H,N,d) -> (B,N,H*dp) with zero padding
        out = torch.zeros(B, t.shape[2], Hh, dp)
        out[..., pq] = t.permute(0, 2, 1, 3)
        return out.reshape(B, t.shape[2], Hh * dp).to(K.a16()).to(DEV).contiguous()

    vt = torch.zeros(B, Hh, DV, Nk_pad)
    vt[:, :, pv, :Nk] = v.transpose(2, 3)
    o = torch.empty(B, Nq, Hh * DV, dtype=K.a16(), device=DEV)
    K.flash_attn(padded(q, DQ), padded(k, DQ), vt.reshape(B, Hh * DV, Nk_pad).to(K.flash_vt_dtype(DQ, DV, Nk)).to(DEV).contiguous(),
                 o, B, Hh, Nq, Nk, Nk_pad, DQ, DV, d ** -0.5)
    ref = torch.softmax(q @ k.transpose(-1, -2) * d ** -0.5, -1) @ v               # (B,H,Nq,d)
    full = o.float().cpu().view(B, Nq, Hh, DV)
    close(full[..., pv].permute(0, 2, 1, 3), ref, rms=6e-3, mx=3e-2, what="flash attention")
    pad = torch.ones(DV, dtype=torch.bool)
    pad[pv] = False
    assert float(full[..., pad].abs().max() if bool(pad.any()) else 0.0) == 0.0


@pytest.mark.parametrize("d,Nq,Nk", [(40, 320, 320), (40, 200, 144), (80, 576, 576), (80, 130, 16), (160, 144, 100)])
def test_flash_attention_strided_operands_equal_dense(d, Nq, Nk):
    """onedc_flash_attn_strided: q and k as column slices of one wider buffer (the fused [to_q | to_k] GEMM output, or the
    to_k of all layers that share a context), V^T blocks of consecutive images further apart than one layer's rows (the to_v of
    several layers from one batched GEMM) -- the same bits as the dense call on the same values."""
    B, Hh = 2, 8
    DQ, DV = pack.HEAD_PAD[d]
    g = torch.Generator().manual_seed(420 + d)
    Nk_pad = (Nk + 63) // 64 * 64
    HDQ, HDV = Hh * DQ, Hh * DV
    vdt = K.flash_vt_dtype(DQ, DV, Nk)
    q = (torch.randn(B, Nq, HDQ, generator=g) * 0.5).to(K.a16()).to(DEV)
    k = (torch.randn(B, Nk, HDQ, generator=g) * 0.5).to(K.a16()).to(DEV)
    vt = torch.zeros(B, HDV, Nk_pad)
    vt[:, :, :Nk] = torch.randn(B, HDV, Nk, generator=g)
    vt = vt.to(vdt).to(DEV)
    dense = torch.empty(B, Nq, HDV, dtype=K.a16(), device=DEV)
    K.flash_attn(q, k, vt, dense, B, Hh, Nq, Nk, Nk_pad, DQ, DV, d ** -0.5)
    # strided: k lives in columns [HDQ + 64, ..) of a 3 HDQ + 128 wide buffer, q in columns [0, HDQ) of another wide one (when
    # Nq == Nk: of the same one); V^T image blocks sit inside a taller [B][HDV + 3 * 16][Nk_pad] buffer at row 16
    wide = 3 * HDQ + 128
    kbuf = torch.randn(B, Nk, wide, generator=g).to(K.a16()).to(DEV)
    kbuf[:, :, HDQ + 64:2 * HDQ + 64] = k
    if Nq == Nk:
        qbuf = kbuf
        qbuf[:, :, :HDQ] = q
    else:
        qbuf = torch.randn(B, Nq, wide, generator=g).to(K.a16()).to(DEV)
        qbuf[:, :, :HDQ] = q
    tall = torch.randn(B, HDV + 48, Nk_pad, generator=g).to(vdt).to(DEV)
    tall[:, 16:16 + HDV] = vt
    got = torch.empty(B, Nq, HDV, dtype=K.a16(), device=DEV)
    K.flash_attn(qbuf.view(-1), kbuf.view(-1)[HDQ + 64:], tall.view(-1)[16 * Nk_pad:], got, B, Hh, Nq, Nk, Nk_pad, DQ, DV,
                 d ** -0.5, q_rs=wide, k_rs=wide, vt_bs=(HDV + 48) * Nk_pad)
    torch.cuda.synchronize()
    assert torch.equal(got, dense)
    # strides below the dense ones, or not multiples of 8, are refused
    with pytest.raises(K.KernelError):
        K.flash_attn(q, k, vt, got, B, Hh, Nq, Nk, Nk_pad, DQ, DV, d ** -0.5, q_rs=HDQ - 8)
    with pytest.raises(K.KernelError):
        K.flash_attn(q, k, vt, got, B, Hh, Nq, Nk, Nk_pad, DQ, DV, d ** -0.5, k_rs=HDQ + 4)


def test_context_kv_equals_per_layer_projections():
    """layers.ContextKV (to_k / to_v of every cross-attention layer of the UNet in three GEMMs) hands each layer the bits its own
    projection GEMMs would have produced: same kernel, same k order -- only the grid differs."""
    from onedc_amd.layers import ContextKV, Conv, MultiHeadAttention, Workspace
    g = torch.Generator().manual_seed(77)
    ws = Workspace(DEV)
    for hw in ((12, 12), (2, 3)):                              # 144 keys (bf16 V^T for d = 40 / 80 in the f16 build) and 6 keys
        layers, sds = [], []
        for C in (320, 1280, 640, 320):
            sd = {f"a.{n}.weight": torch.randn(C, 768 if n in ("to_k", "to_v") else C, generator=g) * 0.03 for n in
                  ("to_q", "to_k", "to_v", "to_out.0")}
            sd["a.to_out.0.bias"] = torch.zeros(C)
            sds.append(sd)
            layers.append(MultiHeadAttention(sd, "a", DEV, C, 768, self_attention=False))
        ctx = from_nchw((torch.randn(2, 768, *hw, generator=g)).to(DEV))
        Nk = hw[0] * hw[1]
        Nk_pad = (Nk + 63) // 64 * 64
        kvs = ContextKV(layers, DEV)(ctx, ws)
        for l, sd, (k_t, k_rs, vt, vt_bs, compact) in zip(layers, sds, kvs):
            # 144 keys: the d = 40 / 80 layers get the COMPACT packing (real head dims back to back, round 5), 6 keys: the padded one
            assert bool(compact) == (l.compact and Nk >= 64)
            DQ, DV = (l.d, l.d) if compact else (l.DQ, l.DV)
            rows = (lambda w, dp: w if compact else pack.pad_heads_rows(w, 8, l.d, dp))
            k_own = Conv({}, "", DEV, weight=rows(sd["a.to_k.weight"], l.DQ))(ctx)
            HDQ, HDV = 8 * DQ, 8 * DV
            got_k = torch.as_strided(k_t, (2 * Nk, HDQ), (k_rs, 1))
            assert torch.equal(got_k, k_own.t[:2 * Nk * HDQ].view(2 * Nk, HDQ))
            vdt = K.flash_vt_dtype(DQ, DV, Nk)
            assert vt.dtype == vdt
            own = torch.zeros(2 * HDV * Nk_pad, dtype=vdt, device=DEV)
            wv = View(pack.pack_matrix(rows(sd["a.to_v.weight"], l.DV), DEV), 1, 1, HDV, 768)
            K.conv_gemm(wv, ctx.t, None, View(own, 1, 1, HDV, Nk, Nk_pad), Cin=768, Cout=Nk, batch=2, x_bs=0, w_bs=Nk * ctx.cs,
                        y_bs=HDV * Nk_pad, out_bf16=(vdt == torch.bfloat16))
            got_v = torch.as_strided(vt, (2, HDV, Nk_pad), (vt_bs, Nk_pad, 1))
            assert torch.equal(got_v, own.view(2, HDV, Nk_pad))


@pytest.mark.parametrize("d,Nq,Nk", [(40, 300, 300), (40, 256, 144), (40, 1024, 1024), (80, 300, 300), (80, 192, 144), (80, 64, 64)])
def test_flash_attention_compact_heads_are_bit_identical_to_padded(d, Nq, Nk):
    """Round 5: q / k / V^T / o with the REAL head dims back to back (onedc_flash_attn with DQ = DV = d: the ring kernels make the
    padding on their DMA path -- zero page, constant-1 chunk, ones row) against the same values in the padded layout: the MFMAs
    see the same operands, so every output bit is equal.  Ragged key tails, strided q / k (column slices of one [q | k] buffer)."""
    from onedc_amd import pack
    B, Hh = 2, 8
    DQ, DV = pack.HEAD_PAD[d]
    pos = pack.head_positions(d, DQ).to(DEV)
    posv = pack.head_positions(d, DV).to(DEV)
    g = torch.Generator(device=DEV).manual_seed(91)
    Nk_pad = (Nk + 63) // 64 * 64
    qk_c = (torch.randn(B * max(Nq, Nk), 2 * Hh * d, generator=g, device=DEV)).to(K.a16())        # rows [q | k], compact
    vt_dt = K.flash_vt_dtype(d, d, Nk)
    assert vt_dt == K.flash_vt_dtype(DQ, DV, Nk)
    v_c = torch.zeros(B, Hh * d, Nk_pad, dtype=vt_dt, device=DEV)
    v_c[:, :, :Nk] = torch.randn(B, Hh * d, Nk, generator=g, device=DEV).to(vt_dt)
    # the same values, padded
    qk_p = torch.zeros(B * max(Nq, Nk), 2 * Hh * DQ, dtype=K.a16(), device=DEV)
    qk_p.view(-1, 2, Hh, DQ)[:, :, :, pos] = qk_c.view(-1, 2, Hh, d)
    v_p = torch.zeros(B, Hh, DV, Nk_pad, dtype=vt_dt, device=DEV)
    v_p[:, :, posv] = v_c.view(B, Hh, d, Nk_pad)
    o_c = torch.full((B * Nq * Hh * d,), 7.0, dtype=K.a16(), device=DEV)
    o_p = torch.empty(B * Nq * Hh * DV, dtype=K.a16(), device=DEV)
    # q / k as column slices of the [q | k] rows: q rows are the first Nq (per image B * N rows are laid out image-major)
    def rows(t, N, w):      # image-major [B][N][w] out of [B * max][w]
        return t.view(B, max(Nq, Nk), -1)[:, :N].contiguous().view(-1)
    qc, kc = rows(qk_c[:, :Hh * d], Nq, Hh * d), rows(qk_c[:, Hh * d:], Nk, Hh * d)
    qp, kp = rows(qk_p[:, :Hh * DQ], Nq, Hh * DQ), rows(qk_p[:, Hh * DQ:], Nk, Hh * DQ)
    K.flash_attn(qc, kc, v_c.view(-1), o_c, B, Hh, Nq, Nk, Nk_pad, d, d, d ** -0.5)
    K.flash_attn(qp, kp, v_p.view(-1), o_p, B, Hh, Nq, Nk, Nk_pad, DQ, DV, d ** -0.5)
    torch.cuda.synchronize()
    assert torch.equal(o_c.view(B * Nq, Hh, d), o_p.view(B * Nq, Hh, DV)[:, :, posv])
    ref = torch.nn.functional.scaled_dot_product_attention(
        qc.float().view(B, Nq, Hh, d).transpose(1, 2), kc.float().view(B, Nk, Hh, d).transpose(1, 2),
        v_c[:, :, :Nk].float().view(B, Hh, d, Nk).transpose(2, 3))
    close(o_c.float().view(B, Nq, Hh, d).transpose(1, 2).cpu(), ref.cpu(), rms=8e-3, mx=6e-2, what=f"compact flash attention d={d}")
    # the layout is only read by the ring kernels: fewer than 64 keys are refused (the layers keep the padded packing for them)
    with pytest.raises(K.KernelError):
        K.flash_attn(qc, kc, v_c.view(-1), o_c, B, Hh, Nq, 48, 64, d, d, d ** -0.5)


@pytest.mark.parametrize("C,self_attn,hw", [(320, True, (16, 20)), (640, True, (12, 12)), (320, False, (16, 16)), (640, False, (8, 10))])
def test_attention_layer_with_compact_heads_equals_the_padded_layer(C, self_attn, hw):
    """layers.MultiHeadAttention built with and without COMPACT_HEADS on the same weights: the same output bits (projection GEMMs
    with fewer columns compute the same dot products in the same k order; the attention kernel sees the same operands)."""
    from onedc_amd import layers as L
    g = torch.Generator().manual_seed(93)
    ctx_dim = C if self_attn else 768
    sd = {f"a.{n}.weight": torch.randn(C, ctx_dim if n in ("to_k", "to_v") else C, generator=g) * 0.05 for n in ("to_q", "to_k", "to_v", "to_out.0")}
    sd["a.to_out.0.bias"] = torch.randn(C, generator=g) * 0.1
    x = from_nchw(torch.randn(2, C, *hw, generator=g).to(DEV))
    ctx = x if self_attn else from_nchw(torch.randn(2, 768, 9, 9, generator=g).to(DEV))       # 81 context tokens
    res = from_nchw(torch.randn(2, C, *hw, generator=g).to(DEV))
    outs = []
    for flag in (True, False):
        before, L.COMPACT_HEADS = L.COMPACT_HEADS, flag
        try:
            layer = L.MultiHeadAttention(sd, "a", DEV, C, ctx_dim, self_attention=self_attn)
        finally:
            L.COMPACT_HEADS = before
        assert layer.compact == flag
        y = layer(x, ctx, res, ws=L.Workspace(DEV))
        outs.append(y.t[:2 * hw[0] * hw[1] * C].clone())
    assert torch.equal(outs[0], outs[1])


@pytest.mark.parametrize("hw,ctx_hw", [((16, 16), (12, 12)), ((24, 16), (8, 8)), ((8, 16), (16, 10)), ((8, 8), (4, 4))])
def test_cross_attention_block_as_one_launch(hw, ctx_hw):
    """csrc/xattn_block.hip (round 6): norm2 -> attn2.to_q -> attention over the context -> attn2.to_out + residual of a diffusers
    BasicTransformerBlock (reference src/models/sd15_onedc_codec_stage1/decoder_unet.py:204-293) as ONE launch, against (a) a plain
    torch fp32 reference of the same ops on the same 16-bit inputs and weights and (b) the four separate launches it replaces
    (LayerNorm, to_q, flash attention, to_out + residual).  C = 320, 8 heads of 40 (the UNet's first level); contexts of 144 (the
    768 x 768 picture), 64, 160 (the kernel's largest: every row of its K image is a real key) and 16 tokens.  Tolerances: the layer bound of this file against the reference; 3e-3 rel. RMS between the
    two device paths (both round q, o and y to 16 bits once; the fused kernel keeps the row maximum where flash_attn64 uses a lazy
    one, and sums the projections' K loop in another order)."""
    from onedc_amd import layers as L
    C, Hh, d, B = 320, 8, 40, 2
    g = torch.Generator().manual_seed(97)
    sd = {f"a.{n}.weight": rnd(C, 768 if n in ("to_k", "to_v") else C, seed=101 + i, scale=(768 if n in ("to_k", "to_v") else C) ** -0.5)
          for i, n in enumerate(("to_q", "to_k", "to_v", "to_out.0"))}
    sd["a.to_out.0.bias"] = rnd(C, seed=110, scale=0.1)
    sd["n.weight"], sd["n.bias"] = 1.0 + rnd(C, seed=111, scale=0.2), rnd(C, seed=112, scale=0.2)
    h_t = rnd(B, C, *hw, seed=113)
    h_t[:, :40] += 3.0                                           # a mean that is not zero: the LayerNorm has something to subtract
    h_t = h_t.to(K.a16()).float()
    ctx_t = rnd(B, 768, *ctx_hw, seed=114)
    h, ctx = from_nchw(h_t.to(DEV)), from_nchw(ctx_t.to(DEV))
    Nk = ctx_hw[0] * ctx_hw[1]
    ln = L.LayerNorm(sd, "n", DEV)
    layer = L.MultiHeadAttention(sd, "a", DEV, C, 768, self_attention=False).prepare_fused(ln, DEV)
    ws = L.Workspace(DEV)
    ckv = L.ContextKV([layer], DEV)
    kv = ckv(ctx, ws)[0]
    keep = K.USE_XATTN_BLOCK
    try:
        K.USE_XATTN_BLOCK = True
        fused = layer.block_fused(h, ln, ctx, kv)
        K.USE_XATTN_BLOCK = False
        assert layer.block_fused(h, ln, ctx, kv) is None
        unfused = layer(ln(h), ctx, h, ws=ws, kv=kv)
    finally:
        K.USE_XATTN_BLOCK = keep
    if Nk < 64 or Nk % 16:                                      # compact heads need 64 keys, the kernel whole 16s
        assert fused is None
        return
    assert fused is not None
    torch.cuda.synchronize()
    # fp32 reference on the same rounded inputs
    x = h_t.permute(0, 2, 3, 1).reshape(B, -1, C)
    n = F.layer_norm(x, (C,), sd["n.weight"], sd["n.bias"], 1e-5)
    c = ctx_t.permute(0, 2, 3, 1).reshape(B, -1, 768)
    q = (n @ sd["a.to_q.weight"].T).view(B, -1, Hh, d).transpose(1, 2)
    k = (c @ sd["a.to_k.weight"].T).view(B, -1, Hh, d).transpose(1, 2)
    v = (c @ sd["a.to_v.weight"].T).view(B, -1, Hh, d).transpose(1, 2)
    o = F.scaled_dot_product_attention(q, k, v).transpose(1, 2).reshape(B, -1, C)
    ref = (x + o @ sd["a.to_out.0.weight"].T + sd["a.to_out.0.bias"]).view(B, *hw, C).permute(0, 3, 1, 2)
    close(fused.torch_nchw(), ref, rms=4e-3, mx=2.5e-2, what=f"fused cross-attention block {hw} x {Nk} keys")
    close(unfused.torch_nchw(), ref, rms=4e-3, mx=2.5e-2, what="the four launches")
    close(fused.torch_nchw(), unfused.torch_nchw(), rms=3e-3, mx=2.5e-2, what="fused against unfused")
    # the kernel's other form: the LayerNorm's affine applied inside (ln_g / ln_b given, to_q's own weights, no q bias) -- what a C
    # host that does not fold would call
    k_t, k_rs, vt, vt_bs = kv[:4]
    y2 = new_view(B, *hw, C, DEV)
    p2 = K.xattn_params(h, ln.g, ln.b, 1e-5, layer.q_c.w, None, k_t, k_rs, vt, vt_bs, layer.o_c.w, layer.o_c.b, y2, Hh, d, Nk,
                        (Nk + 63) // 64 * 64, d ** -0.5)
    assert K.lib().onedc_xattn_block_ok(__import__("ctypes").byref(p2)) == 1
    K.xattn_block(p2)
    close(y2.torch_nchw(), ref, rms=4e-3, mx=2.5e-2, what="fused block, affine inside")
    p2.ln_b = None                                               # gamma without beta: refused, nothing launched
    assert K.lib().onedc_xattn_block_ok(__import__("ctypes").byref(p2)) == 0
    # batch invariance (the path's determinism rule): picture 1 alone == picture 1 inside the batch
    K.USE_XATTN_BLOCK = True
    try:
        one = layer.block_fused(h.batch_slice(1, 2), ln, ctx.batch_slice(1, 2), ckv(ctx.batch_slice(1, 2), ws)[0])
    finally:
        K.USE_XATTN_BLOCK = keep
    npx = hw[0] * hw[1] * C
    assert torch.equal(one.t[:npx], fused.t[npx:2 * npx])


def test_mxfp8_roundtrip_kernel_matches_the_ocp_definition():
    """onedc_mxfp8_roundtrip against the definition written with torch's own float8_e4m3fn cast (pack.mxfp8_roundtrip: e4m3,
    round to nearest even, saturating, one power-of-two scale per 32 channels): bit-equal, including blocks of zeros, values
    below the e4m3 subnormal step, and a block whose largest value lands between 448 and 512 of its scale (saturation)."""
    g = torch.Generator().manual_seed(81)
    B, H, W, C = 2, 5, 7, 96
    x = torch.randn(B, C, H, W, generator=g) * torch.logspace(-4, 3, C)[None, :, None, None]
    x[0, :32, 0, 0] = 0.0
    x[0, 32:64, 0, 0] = torch.linspace(-1, 1, 32) * 1e-6
    x[1, :32, 1, 1] = torch.linspace(0.01, 1.0, 32) * 500.0 / 512.0 * 2.0       # amax just under a power of two: > 448 scaled
    xv = from_nchw(x.to(DEV), cs=128)                                           # a channel slice of a wider buffer
    yv = new_view(B, H, W, C, DEV, cs=160).slice(32, C)
    K.mxfp8_roundtrip(xv, yv)
    got = yv.torch_nchw().cpu()
    xin = xv.torch_nchw().cpu()                                                 # the 16-bit values the kernel saw
    want = pack.mxfp8_roundtrip(xin.permute(0, 2, 3, 1).reshape(-1, C)).reshape(B, H, W, C).permute(0, 3, 1, 2)
    want = want.to(K.a16()).float()
    assert torch.equal(got, want)
    rel = float((got - xin).pow(2).mean().sqrt() / xin.pow(2).mean().sqrt())
    assert 5e-3 < rel < 5e-2                                                    # ~2^-5: a 4-bit significand


def test_x0_and_image_out_and_fsq():
    g = torch.Generator().manual_seed(43)
    red, eps = torch.randn(2 * 6 * 5 * 4, generator=g), torch.randn(2 * 6 * 5 * 4, generator=g)
    x0 = torch.empty_like(red, device=DEV)
    a = 0.004660095
    K.x0_from_noise(red.to(DEV), eps.to(DEV), x0, a)
    ref = ((red.double() - (1 - a) ** 0.5 * eps.double()) / a ** 0.5).float()
    assert torch.equal(x0.cpu(), ref)
    img = torch.randn(2, 3, 16, 24, generator=g) * 1.5
    v = View(img.permute(0, 2, 3, 1).contiguous().to(DEV).reshape(-1), 2, 16, 24, 3)
    out = torch.empty(2, 3, 13, 20, device=DEV)
    K.image_out(v, out, 13, 20, True)
    assert torch.equal(out.cpu(), img[:, :, :13, :20].clamp(-1, 1) * 0.5 + 0.5)
    idx = torch.arange(16384, dtype=torch.int32)
    codes = torch.empty(16384 * 8, device=DEV)
    K.fsq_codes(idx.to(DEV), codes)
    c = codes.cpu().view(-1, 8)
    want = torch.stack([((idx // 4 ** dd) % 4 - 2).float() / 2 for dd in range(7)], 1)
    assert torch.equal(c[:, :7], want) and float(c[:, 7].abs().max()) == 0.0


def test_window_partition_round_trip():
    x = rnd(2, 64, 32, 48, seed=44)
    v = from_nchw(x.to(DEV))
    w = torch.empty_like(v.t)
    K.window_partition(v.t, w, 2, 32, 48, 64, 16)
    back = torch.zeros_like(v.t)
    K.window_partition(w, back, 2, 32, 48, 64, 16, reverse=True)
    n = 2 * 32 * 48 * 64
    assert torch.equal(back[:n], v.t[:n])
    ref = x.view(2, 64, 2, 16, 3, 16).permute(0, 2, 4, 3, 5, 1).reshape(12, 256, 64)   # (b wy wx) (iy ix) c
    assert torch.equal(w[:n].view(12, 256, 64).float().cpu(), ref)
    # ragged edge region: rows 16..31, cols 32..47 as 16x8 windows
    e = torch.zeros_like(v.t)
    K.window_partition(v.t, e, 2, 32, 48, 64, 16, region=(16, 32, 16, 16), win_hw=(16, 8))
    ref = x[:, :, 16:32, 32:48].reshape(2, 64, 1, 16, 2, 8).permute(0, 2, 4, 3, 5, 1).reshape(4, 128, 64)
    assert torch.equal(e[:4 * 128 * 64].view(4, 128, 64).float().cpu(), ref)


# ---- exact-f32 path (prior networks): v_mfma_f32_16x16x4_f32, tolerance 2e-6 rel. RMS vs fp64 ----------
def _f64_conv(x, w, b, **kw):
    return F.conv2d(x.double(), w.double(), None if b is None else b.double(), **kw)


@pytest.mark.parametrize("B,H,W,Cin,Cout,k", [
    (2, 12, 12, 128, 128, 1),
    (1, 9, 7, 96, 256, 3),        # Cin = 3 x 32, odd spatial size
    (3, 8, 8, 256, 512, 1),
])
def test_conv_gemm_f32_exact(B, H, W, Cin, Cout, k):
    g = torch.Generator().manual_seed(21)
    x = torch.randn(B, Cin, H, W, generator=g)
    w = torch.randn(Cout, Cin, k, k, generator=g) * (Cin * k * k) ** -0.5
    b = torch.randn(Cout, generator=g)
    r = torch.randn(B, Cout, H, W, generator=g)
    y = new_view(B, H, W, Cout, DEV, torch.float32)
    K.conv_gemm(from_nchw(x.to(DEV), torch.float32), pack.pack_matrix(pack.conv_matrix(w), DEV, None, torch.float32),
                b.to(DEV), y, KH=k, epi=K.EPI_LEAKY, slope=0.01, res=from_nchw(r.to(DEV), torch.float32))
    ref = F.leaky_relu(_f64_conv(x, w, b, padding=k // 2), 0.01) + r.double()
    close(y.torch_nchw(), ref, rms=2e-6, mx=2e-6, what="f32 conv")
    # these grids are small: the dispatcher took the 64-row f32 tile; the 128-row tile (tile=128) must give the same BITS
    # (same k order per output) -- the prior's decisions may not depend on which tile a map size selects
    y2 = new_view(B, H, W, Cout, DEV, torch.float32)
    K.conv_gemm(from_nchw(x.to(DEV), torch.float32), pack.pack_matrix(pack.conv_matrix(w), DEV, None, torch.float32),
                b.to(DEV), y2, KH=k, epi=K.EPI_LEAKY, slope=0.01, res=from_nchw(r.to(DEV), torch.float32), tile=128)
    assert torch.equal(y.torch_nchw(), y2.torch_nchw())


def test_conv_gemm_f32_glu_and_shuffle():
    g = torch.Generator().manual_seed(22)
    B, H, W, C = 2, 6, 10, 128
    x = torch.randn(B, C, H, W, generator=g)
    w = torch.randn(4 * C, C, 1, 1, generator=g) * C ** -0.5
    b = torch.randn(4 * C, generator=g)
    xv = from_nchw(x.to(DEV), torch.float32)
    perm = pack.glu_perm(2 * C)
    y = new_view(B, H, W, 2 * C, DEV, torch.float32)
    K.conv_gemm(xv, pack.pack_matrix(pack.conv_matrix(w), DEV, perm, torch.float32), b[perm].to(DEV), y, KH=1,
                epi=K.EPI_GLU_LEAKY)
    t = _f64_conv(x, w, b)
    close(y.torch_nchw(), F.leaky_relu(t[:, :2 * C], 0.1) + F.leaky_relu(t[:, 2 * C:], 0.01), rms=2e-6, mx=2e-6,
          what="f32 glu")
    perm = pack.shuffle_perm(4 * C)
    y = new_view(B, 2 * H, 2 * W, C, DEV, torch.float32)
    K.conv_gemm(xv, pack.pack_matrix(pack.conv_matrix(w), DEV, perm, torch.float32), b[perm].to(DEV), y, KH=1,
                pixel_shuffle=1)
    close(y.torch_nchw(), F.pixel_shuffle(t, 2), rms=2e-6, mx=2e-6, what="f32 pixel shuffle")


def test_conv_gemm_f32_batch_invariant():
    """Image 1 alone == image 1 inside a batch of 5, bit for bit (fixed k order, no split-K)."""
    g = torch.Generator().manual_seed(23)
    x = torch.randn(5, 256, 12, 12, generator=g)
    w = pack.pack_matrix(torch.randn(256, 256, generator=g) / 16, DEV, None, torch.float32)
    y5 = new_view(5, 12, 12, 256, DEV, torch.float32)
    y1 = new_view(1, 12, 12, 256, DEV, torch.float32)
    K.conv_gemm(from_nchw(x.to(DEV), torch.float32), w, None, y5, KH=1)
    K.conv_gemm(from_nchw(x[1:2].to(DEV), torch.float32), w, None, y1, KH=1)
    assert torch.equal(y5.torch_nchw()[1:2], y1.torch_nchw())


def test_dwconv3x3_f32():
    g = torch.Generator().manual_seed(24)
    x = torch.randn(2, 128, 9, 11, generator=g)
    w = torch.randn(128, 1, 3, 3, generator=g) / 3
    b = torch.randn(128, generator=g)
    y = new_view(2, 9, 11, 128, DEV, torch.float32)
    K.dwconv3x3(from_nchw(x.to(DEV), torch.float32), w[:, 0].permute(1, 2, 0).reshape(9, -1).contiguous().to(DEV),
                b.to(DEV), y)
    close(y.torch_nchw(), _f64_conv(x, w, b, padding=1, groups=128), rms=1e-6, mx=1e-6, what="f32 dwconv")


# ---- the entropy-parameter head: integer decisions, bit-exact against the oracle's formulas ----------------
def test_prior_indexes_and_update_match_oracle_exactly():
    """build_indexes (entropy_models.py:355-362) over the whole scale range incl. the clamps, and the
    four-part mask / combine_for_writing geometry (compression_model.py:241-301): the int16 index planes
    and the decoder-side y_hat update must equal the oracle's for identical f32 parameters.
    (logf differs from torch's in the last ulp for a few inputs: <= 2e-5 of the positions may differ by 1.)"""
    from oracle import torch_ref as R
    B, N, H, W = 2, 128, 6, 10
    g = torch.Generator().manual_seed(50)
    scales = torch.exp(torch.rand(B, N, H, W, generator=g) * 9 - 4)            # 0.018 .. 148: both clamps hit
    scales[0, :, 0, 0] = 0.0
    scales[0, :, 0, 1] = -1.0
    means = torch.randn(B, N, H, W, generator=g) * 3
    params = torch.cat([scales, means], 1)
    pv = from_nchw(params.to(DEV), torch.float32)
    masks = R.four_part_masks(B, N, H, W)
    n = (N // 4) * H * W
    idx_dev = torch.empty(B * n, dtype=torch.int16, device=DEV)
    y_hat = new_view(B, H, W, N, DEV, torch.float32)
    ref_hat = None
    bad = 0
    for k in range(4):
        K.prior_indexes(pv, idx_dev, B, H, W, N, k)
        ref = R.build_indexes(R.combine_for_writing(scales * masks[k])).to(torch.int16)
        got = idx_dev.view(B, N // 4, H, W).cpu()
        d = (got.int() - ref.int()).abs()
        assert int(d.max()) <= 1
        bad += int((d != 0).sum())
        sym = torch.randint(-40, 41, (B, N // 4, H, W), generator=g).to(torch.int16)
        K.prior_update_dec(pv, sym.to(DEV).reshape(-1), y_hat, k)
        cur = (torch.cat([sym.float()] * 4, 1) + means) * masks[k]
        ref_hat = cur if k == 0 else ref_hat + cur
    assert bad <= 2e-5 * 4 * B * n + 1, bad
    assert torch.equal(y_hat.torch_nchw().cpu(), ref_hat)


# ---- the 256 x 256-tile, 8-phase kernel (forced with tile=256) against the same references ---------------
@pytest.mark.parametrize("B,H,W,Cin,Cout,k", [
    (2, 16, 16, 64, 256, 1),      # one K-tile (prologue/tail only)
    (2, 16, 16, 128, 256, 1),     # two K-tiles
    (1, 24, 20, 192, 256, 3),     # 27 K-tiles (odd), ragged M (480 pixels), halo rows
    (2, 12, 12, 256, 512, 3),     # two channel tiles
    (3, 9, 7, 320, 256, 1),       # 5 K-tiles, M = 189 < one tile
])
def test_conv_gemm256_plain(B, H, W, Cin, Cout, k):
    x, w, b = rnd(B, Cin, H, W, seed=61), rnd(Cout, Cin, k, k, seed=62, scale=(Cin * k * k) ** -0.5), rnd(Cout, seed=63)
    y = new_view(B, H, W, Cout, DEV)
    K.conv_gemm(from_nchw(x.to(DEV)), packed(w), b.to(DEV), y, KH=k, tile=256)
    close(y.torch_nchw(), F.conv2d(x, w, b, padding=k // 2), what="conv256")
    # same bits as the 128-wide kernel?  Both accumulate each output in the same k order (fixed, no split-K)
    y2 = new_view(B, H, W, Cout, DEV)
    K.conv_gemm(from_nchw(x.to(DEV)), packed(w), b.to(DEV), y2, KH=k, tile=128)
    assert torch.equal(y.torch_nchw(), y2.torch_nchw())


def test_conv_gemm256_stride_up_shuffle_glu_residual():
    x, w, b = rnd(2, 128, 16, 12, seed=64), rnd(256, 128, 3, 3, seed=65, scale=1 / 34), rnd(256, seed=66)
    xv = from_nchw(x.to(DEV))
    y = new_view(2, 8, 6, 256, DEV)
    K.conv_gemm(xv, packed(w), b.to(DEV), y, KH=3, stride=2, tile=256)
    close(y.torch_nchw(), F.conv2d(x, w, b, stride=2, padding=1), what="256 stride2")
    y = new_view(2, 32, 24, 256, DEV)
    r = rnd(2, 256, 32, 24, seed=67)
    K.conv_gemm(xv, packed(w), b.to(DEV), y, KH=3, up=1, tile=256, res=from_nchw(r.to(DEV)), epi=K.EPI_SILU)
    close(y.torch_nchw(), F.silu(F.conv2d(F.interpolate(x, scale_factor=2.0, mode="nearest"), w, b, padding=1)) + r,
          what="256 up+silu+res")
    # pixel shuffle
    w1, b1 = rnd(512, 128, 1, 1, seed=68, scale=128 ** -0.5), rnd(512, seed=69)
    perm = pack.shuffle_perm(512)
    y = new_view(2, 32, 24, 128, DEV)
    K.conv_gemm(xv, packed(w1, perm), b1[perm].to(DEV), y, KH=1, pixel_shuffle=1, tile=256)
    close(y.torch_nchw(), F.pixel_shuffle(F.conv2d(x, w1, b1), 2), what="256 pixel shuffle")
    # GLU (ConvFFN3) and GEGLU
    perm = pack.glu_perm(256)
    t = F.conv2d(x, w1, b1)
    y = new_view(2, 16, 12, 256, DEV)
    K.conv_gemm(xv, packed(w1, perm), b1[perm].to(DEV), y, KH=1, epi=K.EPI_GLU_LEAKY, tile=256)
    close(y.torch_nchw(), F.leaky_relu(t[:, :256], 0.1) + F.leaky_relu(t[:, 256:], 0.01), what="256 glu")
    K.conv_gemm(xv, packed(w1, perm), b1[perm].to(DEV), y, KH=1, epi=K.EPI_GEGLU, tile=256)
    close(y.torch_nchw(), t[:, :256] * F.gelu(t[:, 256:]), what="256 geglu")


def test_conv_gemm256_race_screen():
    """The prefetch pipeline leaves DMA in flight across barriers: a misplaced wait shows up as rare
    wrong tiles.  Same launch 40 times on a shape with many K-tiles and many workgroups: identical bits."""
    x, w = rnd(4, 256, 48, 48, seed=70), rnd(512, 256, 3, 3, seed=71, scale=1 / 48)
    xv, wp = from_nchw(x.to(DEV)), packed(w)
    ref = None
    for it in range(40):
        y = new_view(4, 48, 48, 512, DEV)
        K.conv_gemm(xv, wp, None, y, KH=3, tile=256)
        got = y.t[:4 * 48 * 48 * 512].clone()
        if ref is None:
            ref = got
            close(y.torch_nchw(), F.conv2d(x, w, None, padding=1), what="256 race ref")
        else:
            assert torch.equal(got, ref), f"run {it} differs"


# ---- GroupNorm statistics from the producing conv's epilogue ------------------------------------------------
@pytest.mark.parametrize("tile,Cout,HW", [(128, 128, (16, 16)), (256, 512, (16, 24)), (128, 640, (8, 8)), (128, 64, (8, 16))])
def test_group_norm_from_conv_epilogue_stats(tile, Cout, HW):
    """conv (+bias, residual) with stats=True, then GroupNorm+SiLU fed by those partial sums, against
    F.group_norm of the conv result; also the two producers-one-buffer (concat) case."""
    H, W = HW
    B, Cin = 3, 128
    x, w, b = rnd(B, Cin, H, W, seed=80), rnd(Cout, Cin, 3, 3, seed=81, scale=1 / 34), rnd(Cout, seed=82)
    r = rnd(B, Cout, H, W, seed=83)
    g, be = rnd(Cout, seed=84), rnd(Cout, seed=85)
    y = new_view(B, H, W, Cout, DEV, stats=True)
    K.conv_gemm(from_nchw(x.to(DEV)), packed(w), b.to(DEV), y, KH=3, res=from_nchw(r.to(DEV)), tile=tile, stats=True)
    assert y.stats is not None and y.stats.covers(0, Cout)
    out = new_view(B, H, W, Cout, DEV)
    K.group_norm(y, g.to(DEV), be.to(DEV), out, 1e-6, True, _WS())
    ref = F.silu(F.group_norm(F.conv2d(x, w, b, padding=1) + r, 32, g, be, 1e-6))
    close(out.torch_nchw(), ref, rms=6e-3, mx=4e-2, what="gn from epilogue stats")
    # the statistics themselves: same (a, b) as the separate pass over the stored bf16 tensor, to 1e-3
    ws1, ws2 = _WS(), _WS()
    K.group_norm(y, g.to(DEV), be.to(DEV), out, 1e-6, True, ws1)
    K.USE_EPILOGUE_STATS = False
    try:
        K.group_norm(y, g.to(DEV), be.to(DEV), out, 1e-6, True, ws2)
    finally:
        K.USE_EPILOGUE_STATS = True
    a1, a2 = ws1.t["gn_ab"][:B * Cout * 2].cpu(), ws2.t["gn_ab"][:B * Cout * 2].cpu()
    assert float((a1 - a2).abs().max() / a2.abs().max()) <= 2e-3


def test_group_norm_stats_two_producers_one_buffer():
    B, H, W = 2, 16, 16
    xa, xb = rnd(B, 128, H, W, seed=86), rnd(B, 64, H, W, seed=87)
    wa, wb = rnd(256, 128, 1, 1, seed=88, scale=0.09), rnd(128, 64, 1, 1, seed=89, scale=0.12)
    cat = new_view(B, H, W, 384, DEV, stats=True)
    K.conv_gemm(from_nchw(xa.to(DEV)), packed(wa), None, cat.slice(0, 256), KH=1, stats=True)
    assert not cat.stats.covers(0, 384)
    K.conv_gemm(from_nchw(xb.to(DEV)), packed(wb), None, cat.slice(256, 128), KH=1, stats=True)
    assert cat.stats.covers(0, 384)
    g, be = rnd(384, seed=90), rnd(384, seed=91)
    out = new_view(B, H, W, 384, DEV)
    K.group_norm(cat, g.to(DEV), be.to(DEV), out, 1e-5, False, _WS())
    ref = F.group_norm(torch.cat([F.conv2d(xa, wa), F.conv2d(xb, wb)], 1), 32, g, be, 1e-5)
    close(out.torch_nchw(), ref, rms=6e-3, mx=4e-2, what="gn stats, concat buffer")


# ---- the 128 x 256-tile kernel (8 waves, 3-deep LDS ring; forced with tile=129) ---------------------------
@pytest.mark.parametrize("B,H,W,Cin,Cout,k", [
    (2, 16, 16, 64, 128, 1),      # one K-tile
    (2, 16, 16, 128, 128, 1),     # two K-tiles
    (2, 16, 16, 192, 128, 1),     # three K-tiles (ring wraps)
    (1, 24, 20, 128, 128, 3),     # 18 K-tiles, ragged M
    (2, 12, 12, 256, 320, 3),     # 320 channels: three tiles, the last one 5/8 padding rows
    (3, 9, 7, 320, 640, 1),       # five channel tiles, M = 189
])
def test_conv_gemm128x256_plain(B, H, W, Cin, Cout, k):
    x, w, b = rnd(B, Cin, H, W, seed=101), rnd(Cout, Cin, k, k, seed=102, scale=(Cin * k * k) ** -0.5), rnd(Cout, seed=103)
    y = new_view(B, H, W, Cout, DEV)
    K.conv_gemm(from_nchw(x.to(DEV)), packed(w), b.to(DEV), y, KH=k, tile=129)
    close(y.torch_nchw(), F.conv2d(x, w, b, padding=k // 2), what="conv128x256")
    y2 = new_view(B, H, W, Cout, DEV)
    K.conv_gemm(from_nchw(x.to(DEV)), packed(w), b.to(DEV), y2, KH=k, tile=128)
    assert torch.equal(y.torch_nchw(), y2.torch_nchw())


def test_conv_gemm128x256_variants_and_race_screen():
    x, w, b = rnd(2, 128, 16, 12, seed=104), rnd(128, 128, 3, 3, seed=105, scale=1 / 34), rnd(128, seed=106)
    xv = from_nchw(x.to(DEV))
    y = new_view(2, 8, 6, 128, DEV)
    K.conv_gemm(xv, packed(w), b.to(DEV), y, KH=3, stride=2, tile=129)
    close(y.torch_nchw(), F.conv2d(x, w, b, stride=2, padding=1), what="128x256 stride2")
    y = new_view(2, 32, 24, 128, DEV, stats=True)
    r = rnd(2, 128, 32, 24, seed=107)
    K.conv_gemm(xv, packed(w), b.to(DEV), y, KH=3, up=1, tile=129, res=from_nchw(r.to(DEV)), stats=True)
    ref = F.conv2d(F.interpolate(x, scale_factor=2.0, mode="nearest"), w, b, padding=1) + r
    close(y.torch_nchw(), ref, what="128x256 up+res")
    g, be = rnd(128, seed=108), rnd(128, seed=109)
    out = new_view(2, 32, 24, 128, DEV)
    K.group_norm(y, g.to(DEV), be.to(DEV), out, 1e-6, True, _WS())
    close(out.torch_nchw(), F.silu(F.group_norm(ref, 32, g, be, 1e-6)), rms=6e-3, mx=4e-2, what="128x256 stats")
    # race screen: many K-tiles, many workgroups, identical bits over 40 launches
    x, w = rnd(4, 256, 48, 48, seed=110), rnd(128, 256, 3, 3, seed=111, scale=1 / 48)
    xv, wp = from_nchw(x.to(DEV)), packed(w)
    ref = None
    for it in range(40):
        y = new_view(4, 48, 48, 128, DEV)
        K.conv_gemm(xv, wp, None, y, KH=3, tile=129)
        got = y.t[:4 * 48 * 48 * 128].clone()
        if ref is None:
            ref = got
            close(y.torch_nchw(), F.conv2d(x, w, None, padding=1), what="128x256 race ref")
        else:
            assert torch.equal(got, ref), f"run {it} differs"


@pytest.mark.parametrize("d", [40, 80])
def test_flash_attention_reference_point_moves(d):
    """The d = 40 / 80 kernels keep a lazily moved softmax reference (set from tile 0, moved only when a later
    tile exceeds it by 2^12): force that rare path -- logits that grow tile after tile and one spiked key
    late in the sequence -- and an all-very-negative first tile; full-tensor check against fp64."""
    B, Hh, Nq, Nk = 1, 8, 320, 640
    DQ, DV = pack.HEAD_PAD[d]
    pq, pv = pack.head_positions(d, DQ), pack.head_positions(d, DV)
    g = torch.Generator().manual_seed(77)
    q = torch.randn(B, Hh, Nq, d, generator=g)
    k = torch.randn(B, Hh, Nk, d, generator=g)
    v = torch.randn(B, Hh, Nk, d, generator=g)
    ramp = torch.linspace(0.2, 6.0, Nk).view(1, 1, Nk, 1)                   # later keys give larger logits
    k = k * ramp
    k[:, :, 500] = q[:, :, 7] * 4.0                                         # a spike for query 7 (and friends)
    k[:, 3, :64] = -q[:, 3, :1] * 3.0                                       # head 3: first tile strongly negative for query 0
    q, k, v = (t.to(torch.bfloat16).float() for t in (q, k, v))

    def padded(t, dp):
        out = torch.zeros(B, t.shape[2], Hh, dp)
        out[..., pq] = t.permute(0, 2, 1, 3)
        return out.reshape(B, t.shape[2], Hh * dp).to(K.a16()).to(DEV).contiguous()

    vt = torch.zeros(B, Hh, DV, Nk)
    vt[:, :, pv] = v.transpose(2, 3)
    o = torch.empty(B, Nq, Hh * DV, dtype=K.a16(), device=DEV)
    K.flash_attn(padded(q, DQ), padded(k, DQ), vt.reshape(B, Hh * DV, Nk).to(K.flash_vt_dtype(DQ, DV, Nk)).to(DEV).contiguous(),
                 o, B, Hh, Nq, Nk, Nk, DQ, DV, d ** -0.5)
    ref = torch.softmax(q.double() @ k.double().transpose(-1, -2) * d ** -0.5, -1) @ v.double()
    got = o.float().cpu().view(B, Nq, Hh, DV)[..., pv].permute(0, 2, 1, 3)
    assert torch.isfinite(got).all()
    close(got, ref, rms=1e-2, mx=4e-2, what="flash attention, moving reference")


@pytest.mark.parametrize("d", [40, 80])
def test_flash_attention_fast_pass_falls_back_to_the_moving_reference(d):
    """The d = 40 / 80 kernels first run WITHOUT looking for tile maxima (reference point fixed by tile 0) and repeats the workgroup
    with the moving reference only when a row sum leaves (2^-100, 2^100).  Head 2 here has logits that climb by ~2^30 per tile of
    64 keys to ~2^320 above tile 0 -- exp2 overflows in the fast pass, the moving reference follows it (a jump of more than 2^127
    inside ONE tile is beyond either pass: P is bf16, f32's exponent range); the other heads never leave the fast pass.
    Full-tensor check against fp64."""
    B, Hh, Nq, Nk = 1, 8, 512, 704
    DQ, DV = pack.HEAD_PAD[d]
    pq, pv = pack.head_positions(d, DQ), pack.head_positions(d, DV)
    g = torch.Generator().manual_seed(78)
    q = torch.randn(B, Hh, Nq, d, generator=g)
    k = torch.randn(B, Hh, Nk, d, generator=g) * 0.5
    v = torch.randn(B, Hh, Nk, d, generator=g)
    q[:, 2] = 1.0 + 0.05 * q[:, 2]                                          # head 2: every query close to the all-ones direction
    k[:, 2] = torch.linspace(0.0, 35.0 * (40 / d) ** 0.5, Nk).view(1, Nk, 1) * torch.ones(d) + 0.05 * k[:, 2]   # keys climb along it
    q, k, v = (t.to(torch.float16).float() for t in (q, k, v))
    assert float((q[:, 2] @ k[:, 2].transpose(-1, -2)).max()) * d ** -0.5 * 1.4427 > 250

    def padded(t, dp):
        out = torch.zeros(B, t.shape[2], Hh, dp)
        out[..., pq] = t.permute(0, 2, 1, 3)
        return out.reshape(B, t.shape[2], Hh * dp).to(K.a16()).to(DEV).contiguous()

    vt = torch.zeros(B, Hh, DV, Nk)
    vt[:, :, pv] = v.transpose(2, 3)
    o = torch.empty(B, Nq, Hh * DV, dtype=K.a16(), device=DEV)
    K.flash_attn(padded(q, DQ), padded(k, DQ), vt.reshape(B, Hh * DV, Nk).to(K.flash_vt_dtype(DQ, DV, Nk)).to(DEV).contiguous(),
                 o, B, Hh, Nq, Nk, Nk, DQ, DV, d ** -0.5)
    qa, ka = (t.to(K.a16()).double() for t in (q, k))
    ref = torch.softmax(qa @ ka.transpose(-1, -2) * d ** -0.5, -1) @ v.to(torch.bfloat16).double()
    got = o.float().cpu().view(B, Nq, Hh, DV)[..., pv].permute(0, 2, 1, 3)
    assert torch.isfinite(got).all()
    close(got, ref, rms=1e-2, mx=4e-2, what="flash attention, fast pass + fallback")


# ---- 3x3 window kernel (16x16-pixel patches, every input pixel staged once per chunk; tile=130) -----------
@pytest.mark.parametrize("B,H,W,Cin,Cout", [
    (1, 16, 16, 64, 128),         # one patch, one chunk: nine K-tiles, borders on all four sides
    (2, 32, 48, 128, 128),        # interior patches, two chunks (window double buffer)
    (1, 48, 32, 192, 320),        # three chunks, 320 channels (last channel tile mostly padding rows)
    (3, 16, 32, 256, 256),        # two channel tiles, batch
])
def test_conv_win3x3_plain(B, H, W, Cin, Cout):
    x, w, b = rnd(B, Cin, H, W, seed=121), rnd(Cout, Cin, 3, 3, seed=122, scale=(Cin * 9) ** -0.5), rnd(Cout, seed=123)
    y = new_view(B, H, W, Cout, DEV)
    K.conv_gemm(from_nchw(x.to(DEV)), packed(w), b.to(DEV), y, KH=3, tile=130)
    close(y.torch_nchw(), F.conv2d(x, w, b, padding=1), what="win3x3")
    # against the generic kernel: same products, different summation order (chunk-major vs tap-major)
    y2 = new_view(B, H, W, Cout, DEV)
    K.conv_gemm(from_nchw(x.to(DEV)), packed(w), b.to(DEV), y2, KH=3, tile=128)
    close(y.torch_nchw(), y2.torch_nchw(), rms=3e-3, mx=2e-2, what="win3x3 vs generic")


# ---- the same layers on the 4-wave window kernel (two workgroups per CU; tile=131) --------------------------------
@pytest.mark.parametrize("B,H,W,Cin,Cout", [
    (1, 16, 16, 64, 128),         # one patch, two 32-channel chunks, borders on all four sides
    (2, 32, 48, 128, 128),        # interior patches, four chunks (two passes of the chunk-pair loop)
    (1, 48, 32, 192, 320),        # six chunks, 320 channels (the last channel tile has one live wave row)
    (3, 16, 32, 256, 256),        # two channel tiles, batch
])
def test_conv_win3x3_4wave_plain(B, H, W, Cin, Cout):
    x, w, b = rnd(B, Cin, H, W, seed=321), rnd(Cout, Cin, 3, 3, seed=322, scale=(Cin * 9) ** -0.5), rnd(Cout, seed=323)
    y = new_view(B, H, W, Cout, DEV)
    K.conv_gemm(from_nchw(x.to(DEV)), packed(w), b.to(DEV), y, KH=3, tile=131)
    close(y.torch_nchw(), F.conv2d(x, w, b, padding=1), what="win3x3 4-wave")
    y2 = new_view(B, H, W, Cout, DEV)
    K.conv_gemm(from_nchw(x.to(DEV)), packed(w), b.to(DEV), y2, KH=3, tile=130)
    close(y.torch_nchw(), y2.torch_nchw(), rms=3e-3, mx=2e-2, what="win3x3 4-wave vs 8-wave")


@pytest.mark.parametrize("gran", [4, 2])
def test_conv_win3x3_4wave_residual_stats_slices(gran):
    """Residual, bias, input and output in channel slices, GroupNorm statistics (quads; pairs for a 320-channel buffer)
    left by the epilogue and consumed by group_norm without a statistics pass."""
    B, H, W, Cin = 2, 32, 16, 128
    Cout = 128 if gran == 4 else 320
    x, w, b = rnd(B, Cin, H, W, seed=324), rnd(Cout, Cin, 3, 3, seed=325, scale=1 / 34), rnd(Cout, seed=326)
    r = rnd(B, Cout, H, W, seed=327)
    xbuf = from_nchw(torch.cat([rnd(B, 64, H, W, seed=328), x], 1).to(DEV))
    if gran == 4:
        ybuf = new_view(B, H, W, Cout + 64, DEV, stats=True)
        ybuf.t.zero_()
        yv = ybuf.slice(64, Cout)
    else:
        ybuf = new_view(B, H, W, Cout, DEV, stats=True)
        assert ybuf.stats.gran == 2
        yv = ybuf
    K.conv_gemm(xbuf.slice(64, Cin), packed(w), b.to(DEV), yv, KH=3, tile=131, res=from_nchw(r.to(DEV)), stats=True)
    ref = F.conv2d(x, w, b, padding=1) + r
    close(yv.torch_nchw(), ref, what="win3x3 4-wave res")
    if gran == 4:
        assert float(ybuf.slice(0, 64).torch_nchw().abs().max()) == 0.0
    g, be = rnd(Cout, seed=329), rnd(Cout, seed=330)
    out = new_view(B, H, W, Cout, DEV)
    ws = _WS()
    K.group_norm(yv, g.to(DEV), be.to(DEV), out, 1e-6, True, ws)
    assert "gn_partial" not in ws.t
    close(out.torch_nchw(), F.silu(F.group_norm(ref, 32, g, be, 1e-6)), rms=6e-3, mx=4e-2, what="win3x3 4-wave stats")
    # the statistics are the 8-wave kernel's to the last bit wherever the outputs are (same 64 pixels, same order)
    y8 = new_view(B, H, W, Cout, DEV, stats=True)
    K.conv_gemm(xbuf.slice(64, Cin), packed(w), b.to(DEV), y8, KH=3, tile=130, res=from_nchw(r.to(DEV)), stats=True)
    close(yv.torch_nchw(), y8.torch_nchw(), rms=3e-3, mx=2e-2, what="4-wave vs 8-wave with residual")


@pytest.mark.parametrize("B,H,W,Cin,Cout,silu", [(2, 32, 16, 128, 128, True), (1, 16, 48, 320, 128, True),
                                                 (2, 16, 16, 64, 128, False), (1, 48, 48, 512, 128, True)])
def test_conv_win3x3_4wave_groupnorm_on_load(B, H, W, Cin, Cout, silu):
    """conv(act(GroupNorm(x))) + residual on the 4-wave kernel (window blocks normalised in place in LDS): against
    F.conv2d(F.group_norm), and bit for bit against the same kernel behind the separate scale/shift pass."""
    x, w, b = rnd(B, Cin, H, W, seed=341), rnd(Cout, Cin, 3, 3, seed=342, scale=(9 * Cin) ** -0.5), rnd(Cout, seed=343)
    g, be, r = 1 + 0.2 * rnd(Cin, seed=344), 0.5 * rnd(Cin, seed=345), rnd(B, Cout, H, W, seed=346)
    xbuf = from_nchw(torch.cat([rnd(B, 64, H, W, seed=347), x], 1).to(DEV))        # input is a channel slice
    outs = []
    for fuse in (True, False):
        K.USE_GN_FUSION = fuse
        try:
            y = new_view(B, H, W, Cout, DEV, stats=True)
            ws = _WS()
            K.conv_gemm(xbuf.slice(64, Cin), packed(w), b.to(DEV), y, KH=3, res=from_nchw(r.to(DEV)), stats=True,
                        gn=(g.to(DEV), be.to(DEV), 1e-6, silu, ws), tile=131)
        finally:
            K.USE_GN_FUSION = True
        outs.append((y.t[:B * H * W * Cout].clone(), y.stats.t.clone()))
    n = F.group_norm(x, 32, g, be, 1e-6)
    ref = F.conv2d(F.silu(n) if silu else n, w, b, padding=1) + r
    close(y.torch_nchw(), ref, rms=6e-3, mx=4e-2, what="4-wave gn on load (unfused)")
    assert torch.equal(outs[0][0], outs[1][0]) and torch.equal(outs[0][1], outs[1][1])


def test_conv_win3x3_4wave_groupnorm_on_load_race_screen():
    B, H, W, Cin, Cout = 4, 48, 48, 256, 128
    x, w = rnd(B, Cin, H, W, seed=351), rnd(Cout, Cin, 3, 3, seed=352, scale=(9 * Cin) ** -0.5)
    g, be = 1 + 0.2 * rnd(Cin, seed=353), 0.5 * rnd(Cin, seed=354)
    xv, wp, ws = from_nchw(x.to(DEV)), packed(w), _WS()
    ref = None
    for it in range(40):
        y = new_view(B, H, W, Cout, DEV)
        K.conv_gemm(xv, wp, None, y, KH=3, gn=(g.to(DEV), be.to(DEV), 1e-6, True, ws), tile=131)
        got = y.t[:B * H * W * Cout].clone()
        if ref is None:
            ref = got
            close(y.torch_nchw(), F.conv2d(F.silu(F.group_norm(x, 32, g, be, 1e-6)), w, padding=1), rms=6e-3, mx=4e-2,
                  what="4-wave gn on load race ref")
        else:
            assert torch.equal(got, ref), f"run {it} differs"


def test_conv_win3x3_4wave_race_screen_and_batch_invariance():
    x, w = rnd(4, 256, 48, 48, seed=331), rnd(128, 256, 3, 3, seed=332, scale=1 / 48)
    xv, wp = from_nchw(x.to(DEV)), packed(w)
    ref = None
    for it in range(40):
        y = new_view(4, 48, 48, 128, DEV)
        K.conv_gemm(xv, wp, None, y, KH=3, tile=131)
        got = y.t[:4 * 48 * 48 * 128].clone()
        if ref is None:
            ref = got
            close(y.torch_nchw(), F.conv2d(x, w, None, padding=1), what="win3x3 4-wave race ref")
        else:
            assert torch.equal(got, ref), f"run {it} differs"
    y1 = new_view(1, 48, 48, 128, DEV)
    K.conv_gemm(from_nchw(x[2:3].to(DEV)), wp, None, y1, KH=3, tile=131)
    assert torch.equal(y1.t[:48 * 48 * 128], ref.view(4, -1)[2])


def test_conv_win3x3_upsample_residual_stats_slices():
    B, H, W, Cin, Cout = 2, 16, 24, 128, 128
    x, w, b = rnd(B, Cin, H, W, seed=124), rnd(Cout, Cin, 3, 3, seed=125, scale=1 / 34), rnd(Cout, seed=126)
    r = rnd(B, Cout, 2 * H, 2 * W, seed=127)
    xbuf = from_nchw(torch.cat([rnd(B, 64, H, W, seed=128), x], 1).to(DEV))          # input from a channel slice
    ybuf = new_view(B, 2 * H, 2 * W, Cout + 64, DEV, stats=True)
    ybuf.t.zero_()
    K.conv_gemm(xbuf.slice(64, Cin), packed(w), b.to(DEV), ybuf.slice(64, Cout), KH=3, up=1, tile=130,
                res=from_nchw(r.to(DEV)), epi=K.EPI_SILU, stats=True)
    ref = F.silu(F.conv2d(F.interpolate(x, scale_factor=2.0, mode="nearest"), w, b, padding=1)) + r
    close(ybuf.slice(64, Cout).torch_nchw(), ref, what="win3x3 up+silu+res")
    assert float(ybuf.slice(0, 64).torch_nchw().abs().max()) == 0.0
    g, be = rnd(Cout, seed=129), rnd(Cout, seed=130)
    out = new_view(B, 2 * H, 2 * W, Cout, DEV)
    K.group_norm(ybuf.slice(64, Cout), g.to(DEV), be.to(DEV), out, 1e-6, True, _WS())
    close(out.torch_nchw(), F.silu(F.group_norm(ref, 32, g, be, 1e-6)), rms=6e-3, mx=4e-2, what="win3x3 stats")


def test_conv_win3x3_race_screen_and_batch_invariance():
    x, w = rnd(4, 256, 48, 48, seed=131), rnd(128, 256, 3, 3, seed=132, scale=1 / 48)
    xv, wp = from_nchw(x.to(DEV)), packed(w)
    ref = None
    for it in range(40):
        y = new_view(4, 48, 48, 128, DEV)
        K.conv_gemm(xv, wp, None, y, KH=3, tile=130)
        got = y.t[:4 * 48 * 48 * 128].clone()
        if ref is None:
            ref = got
            close(y.torch_nchw(), F.conv2d(x, w, None, padding=1), what="win3x3 race ref")
        else:
            assert torch.equal(got, ref), f"run {it} differs"
    y1 = new_view(1, 48, 48, 128, DEV)
    K.conv_gemm(from_nchw(x[2:3].to(DEV)), wp, None, y1, KH=3, tile=130)
    assert torch.equal(y1.t[:48 * 48 * 128], ref.view(4, -1)[2])


def test_conv_win3x3_is_the_automatic_choice_whatever_the_batch():
    """tile=0 (automatic) takes a window kernel for every eligible 3x3 layer, for one picture as for sixteen -- the
    4-wave one where the epilogue is the plain linear one, else the 8-wave one: same bits as the forced kernel, output
    and statistics."""
    x, w, b = rnd(3, 128, 32, 32, seed=171), rnd(256, 128, 3, 3, seed=172, scale=1 / 34), rnd(256, seed=173)
    for epi, forced, name in ((K.EPI_LINEAR, 131, "win3x3_4w"), (K.EPI_SILU, 130, "win3x3")):
        for xs in (x, x[1:2]):
            outs = []
            for tile in (forced, 0):
                y = new_view(xs.shape[0], 32, 32, 256, DEV, stats=True)
                K.conv_gemm(from_nchw(xs.to(DEV)), packed(w), b.to(DEV), y, KH=3, tile=tile, stats=True, epi=epi)
                outs.append((y.t[:xs.shape[0] * 32 * 32 * 256].clone(), y.stats.t.clone()))
            assert torch.equal(outs[0][0], outs[1][0]) and torch.equal(outs[0][1], outs[1][1]), name
            if xs.shape[0] == 3:
                whole = outs[0][0].view(3, -1)[1].clone()
        assert torch.equal(outs[0][0].view(-1), whole), name


# ---- x2 upsample + 3x3 as four 2x2 phase convolutions (up=2, gemm_conv_ups.hip) -------------------------------
@pytest.mark.parametrize("B,H,W,Cin,C,epi", [(2, 16, 32, 128, 128, "linear"), (1, 32, 16, 64, 256, "silu"),
                                             (3, 16, 16, 320, 128, "linear")])
def test_conv_upsample_phases(B, H, W, Cin, C, epi):
    """Against F.conv2d(F.interpolate(nearest)); input and output in channel slices, statistics for the consumer's
    GroupNorm, and agreement with the plain fused-upsample path (up=1)."""
    from onedc_amd import pack
    x, w, b = rnd(B, Cin, H, W, seed=181), rnd(C, Cin, 3, 3, seed=182, scale=(9 * Cin) ** -0.5), rnd(C, seed=183)
    xbuf = from_nchw(torch.cat([rnd(B, 64, H, W, seed=184), x], 1).to(DEV))
    ybuf = new_view(B, 2 * H, 2 * W, C + 64, DEV, stats=True)
    ybuf.t.zero_()
    wp = pack.pack_matrix(pack.upsample_phase_matrix(w), DEV)
    e = K.EPI_SILU if epi == "silu" else K.EPI_LINEAR
    K.conv_gemm(xbuf.slice(64, Cin), wp, b.to(DEV), ybuf.slice(64, C), KH=3, up=2, epi=e, Cin=Cin, Cout=C, stats=True)
    ref = F.conv2d(F.interpolate(x, scale_factor=2.0, mode="nearest"), w, b, padding=1)
    if epi == "silu":
        ref = F.silu(ref)
    close(ybuf.slice(64, C).torch_nchw(), ref, what="upsample phases")
    assert float(ybuf.slice(0, 64).torch_nchw().abs().max()) == 0.0
    g, be = rnd(C, seed=185), rnd(C, seed=186)
    out = new_view(B, 2 * H, 2 * W, C, DEV)
    ws = _WS()
    K.group_norm(ybuf.slice(64, C), g.to(DEV), be.to(DEV), out, 1e-6, True, ws)
    assert "gn_partial" not in ws.t
    close(out.torch_nchw(), F.silu(F.group_norm(ref, 32, g, be, 1e-6)), rms=6e-3, mx=4e-2, what="upsample phases stats")
    y1 = new_view(B, 2 * H, 2 * W, C, DEV)
    K.conv_gemm(xbuf.slice(64, Cin), packed(w), b.to(DEV), y1, KH=3, up=1, epi=e)
    close(ybuf.slice(64, C).torch_nchw(), y1.torch_nchw(), rms=4e-3, mx=3e-2, what="phases vs fused upsample")
    # linear epilogue: the automatic choice is the 4-wave kernel (tile 131); the 8-wave one (tile 130) agrees to rounding
    want = "ups2x2_4w" if epi == "linear" else "ups2x2"
    y8 = new_view(B, 2 * H, 2 * W, C, DEV, stats=True)
    K.conv_gemm(xbuf.slice(64, Cin), wp, b.to(DEV), y8, KH=3, up=2, epi=e, Cin=Cin, Cout=C, stats=True, tile=130)
    close(ybuf.slice(64, C).torch_nchw(), y8.torch_nchw(), rms=3e-3, mx=2e-2, what="phases: 4 waves vs 8 waves")
    assert K.PROFILE is None
    K.PROFILE = []
    try:
        K.conv_gemm(xbuf.slice(64, Cin), wp, b.to(DEV), y8, KH=3, up=2, epi=e, Cin=Cin, Cout=C)
        assert K.PROFILE[0][0] == want
    finally:
        K.PROFILE = None


def test_conv_upsample_phases_store_bf16_in_the_f16_build():
    """onedc_conv_params.out_bf16 on the upsample phases (round 5: the VAE's last up block may run in bf16 inside an f16 model, and
    its input is this kernel's output): the f16 library's launch with out_bf16 must store exactly round-to-nearest-even bf16 of
    the values its f16 launch rounds to f16 -- both are roundings of the same f32 sums -- and leave the same statistics."""
    from onedc_amd import pack
    if K.act() != "f16":
        pytest.skip("the flag only means something in the IEEE-half build")
    B, H, W, Cin, C = 2, 16, 32, 128, 256
    x, w, b = rnd(B, Cin, H, W, seed=187), rnd(C, Cin, 3, 3, seed=188, scale=(9 * Cin) ** -0.5), rnd(C, seed=189)
    xv, wp = from_nchw(x.to(DEV)), pack.pack_matrix(pack.upsample_phase_matrix(w), DEV)
    y16 = new_view(B, 2 * H, 2 * W, C, DEV, stats=True)
    K.conv_gemm(xv, wp, b.to(DEV), y16, KH=3, up=2, Cin=Cin, Cout=C, stats=True)
    ybf = new_view(B, 2 * H, 2 * W, C, DEV, torch.bfloat16, stats=True)
    K.conv_gemm(xv, wp, b.to(DEV), ybf, KH=3, up=2, Cin=Cin, Cout=C, stats=True, out_bf16=True)
    n = B * 4 * H * W * C
    ref = F.conv2d(F.interpolate(x, scale_factor=2.0, mode="nearest"), w, b, padding=1)
    got = ybf.t[:n].float().view(B, 2 * H, 2 * W, C).permute(0, 3, 1, 2).cpu()
    close(got, ref, rms=6e-3, mx=5e-2, what="upsample phases, bf16 stored by the f16 build")
    # bf16(f32 sum) against bf16(f16(f32 sum)): f16 keeps 3 bits more than bf16, so one f16 value in 8 sits exactly on a bf16 tie and
    # the direct rounding went the other way for half of those: 6.25 % of the elements differ, each by exactly one bf16 ulp
    twice = y16.t[:n].float().to(torch.bfloat16)
    differ = twice != ybf.t[:n]
    assert 0.03 < float(differ.float().mean()) < 0.09
    a, b = twice.view(torch.int16).int(), ybf.t[:n].view(torch.int16).int()
    big = y16.t[:n].float().abs() > 2.0 ** -10          # (below, f16's own precision falls under bf16's: subnormals)
    assert int((a - b).abs()[big].max()) == 1 and int((a - b).abs().max()) <= 8
    assert torch.equal(y16.stats.t, ybf.stats.t)                       # statistics come from the f32 values
    with pytest.raises(K.KernelError):                                 # the 8-wave phase kernel stores the build's format only
        K.conv_gemm(xv, wp, b.to(DEV), ybf, KH=3, up=2, Cin=Cin, Cout=C, out_bf16=True, tile=130)


def test_conv_upsample_phases_race_screen_and_batch_invariance():
    from onedc_amd import pack
    x, w = rnd(4, 256, 32, 32, seed=191), rnd(128, 256, 3, 3, seed=192, scale=1 / 48)
    xv, wp = from_nchw(x.to(DEV)), pack.pack_matrix(pack.upsample_phase_matrix(w), DEV)
    ref = None
    for it in range(30):
        y = new_view(4, 64, 64, 128, DEV)
        K.conv_gemm(xv, wp, None, y, KH=3, up=2, Cin=256, Cout=128)
        got = y.t[:4 * 64 * 64 * 128].clone()
        if ref is None:
            ref = got
        else:
            assert torch.equal(got, ref), f"run {it} differs"
    y1 = new_view(1, 64, 64, 128, DEV)
    K.conv_gemm(from_nchw(x[2:3].to(DEV)), wp, None, y1, KH=3, up=2, Cin=256, Cout=128)
    assert torch.equal(y1.t[:64 * 64 * 128], ref.view(4, -1)[2])


# ---- GroupNorm (+SiLU) applied while the window kernel stages its operand (gn=...) --------------------------
@pytest.mark.parametrize("B,H,W,Cin,Cout,silu", [(2, 32, 16, 128, 128, True), (1, 16, 48, 320, 128, True),
                                                 (2, 16, 16, 64, 128, False), (1, 48, 48, 512, 128, True)])
def test_conv_win3x3_groupnorm_on_load(B, H, W, Cin, Cout, silu):
    """conv(act(GroupNorm(x))) + residual: the fused path against F.conv2d(F.group_norm), and bit for bit against
    the unfused path (separate scale/shift pass, same kernel) -- halo pixels must stay zero, not act(b)."""
    x, w, b = rnd(B, Cin, H, W, seed=201), rnd(Cout, Cin, 3, 3, seed=202, scale=(9 * Cin) ** -0.5), rnd(Cout, seed=203)
    g, be, r = 1 + 0.2 * rnd(Cin, seed=204), 0.5 * rnd(Cin, seed=205), rnd(B, Cout, H, W, seed=206)
    xbuf = from_nchw(torch.cat([rnd(B, 64, H, W, seed=207), x], 1).to(DEV))        # input is a channel slice
    outs = []
    for fuse in (True, False):
        K.USE_GN_FUSION = fuse
        try:
            y = new_view(B, H, W, Cout, DEV, stats=True)
            ws = _WS()
            K.conv_gemm(xbuf.slice(64, Cin), packed(w), b.to(DEV), y, KH=3, res=from_nchw(r.to(DEV)), stats=True,
                        gn=(g.to(DEV), be.to(DEV), 1e-6, silu, ws))
        finally:
            K.USE_GN_FUSION = True
        outs.append((y.t[:B * H * W * Cout].clone(), y.stats.t.clone()))
    n = F.group_norm(x, 32, g, be, 1e-6)
    ref = F.conv2d(F.silu(n) if silu else n, w, b, padding=1) + r
    close(y.torch_nchw(), ref, rms=6e-3, mx=4e-2, what="gn on load (unfused)")
    assert torch.equal(outs[0][0], outs[1][0]) and torch.equal(outs[0][1], outs[1][1])


@pytest.mark.parametrize("B,H,W,Cin,Cout,x2c", [(2, 32, 16, 128, 128, 0), (1, 16, 48, 256, 256, 0), (1, 32, 32, 128, 128, 256), (2, 16, 16, 320, 320, 0)])
def test_conv_win3x3_bf16_operands_under_f16_storage(B, H, W, Cin, Cout, x2c):
    """Round 5: conv(silu(GroupNorm(x))) multiplying bf16 OPERANDS (onedc_conv_params.x_bf16: the normalised tensor -- an
    intermediate only this conv reads -- and the weights in bf16) while y, the residual, the statistics and a second source stay
    f16.  Against F.conv2d on operands rounded the same way; the fused form (normalised to bf16 on load) and the unfused one (the
    scale/shift pass stores bf16) give the same bits; with the switch off the launch is the f16 one."""
    if K.act() != "f16":
        pytest.skip("operand format of the IEEE-half build")
    x, w, b = rnd(B, Cin, H, W, seed=211), rnd(Cout, Cin, 3, 3, seed=212, scale=(9 * Cin) ** -0.5), rnd(Cout, seed=213)
    g, be, r = 1 + 0.2 * rnd(Cin, seed=214), 0.5 * rnd(Cin, seed=215), rnd(B, Cout, H, W, seed=216)
    xv, rv = from_nchw(x.to(DEV)), from_nchw(r.to(DEV))
    wp, wbf = packed(w), pack.pack_matrix(pack.conv_matrix(w), DEV, None, torch.bfloat16)
    x2 = w2 = x2v = w2p = None
    if x2c:
        x2, w2 = rnd(B, x2c, H, W, seed=217), rnd(Cout, x2c, 1, 1, seed=218, scale=x2c ** -0.5)
        x2v, w2p = from_nchw(x2.to(DEV)), packed(w2)
    outs = {}
    for name, fuse, bf in (("fused", True, True), ("pass", False, True), ("f16", True, False)):
        K.USE_GN_FUSION, before = fuse, K.BF16_OPERANDS
        K.BF16_OPERANDS = bf
        try:
            y = new_view(B, H, W, Cout, DEV, stats=True)
            ws = _WS()
            assert K.conv_gemm(xv, wp, b.to(DEV), y, KH=3, res=None if x2c else rv, stats=True, gn=(g.to(DEV), be.to(DEV), 1e-6, True, ws),
                               x2=x2v, w2=w2p, w_bf16=wbf) is not None
        finally:
            K.USE_GN_FUSION, K.BF16_OPERANDS = True, before
        outs[name] = (y.t[:B * H * W * Cout].clone(), y.stats.t.clone(), y)
    if Cout <= K.GN_FUSION_MAX_COUT:                  # (wider layers take the pass either way)
        assert torch.equal(outs["fused"][0], outs["pass"][0]) and torch.equal(outs["fused"][1], outs["pass"][1])
    assert not torch.equal(outs["fused"][0], outs["f16"][0])          # the switch changes the operands
    xn = F.silu(F.group_norm(x.to(torch.float16).float(), 32, g, be, 1e-6))
    ref16 = F.conv2d(xn.to(torch.float16).double(), w.to(torch.float16).double(), b.double(), padding=1)
    refbf = F.conv2d(xn.to(torch.bfloat16).double(), w.to(torch.bfloat16).double(), b.double(), padding=1)
    tail = F.conv2d(x2.to(torch.float16).double(), w2.to(torch.float16).double()) if x2c else r.to(torch.float16).double()
    # each form against ITS reference far more closely than the two references differ
    e_bf = float((outs["fused"][2].torch_nchw().double().cpu() - (refbf + tail)).pow(2).mean().sqrt())
    e_16 = float((outs["f16"][2].torch_nchw().double().cpu() - (ref16 + tail)).pow(2).mean().sqrt())
    gap = float((refbf - ref16).pow(2).mean().sqrt())
    scale = float((ref16 + tail).pow(2).mean().sqrt())
    print(f"[bf16 operands] rms error vs own reference: bf16 {e_bf / scale:.2e}, f16 {e_16 / scale:.2e}; the references differ by {gap / scale:.2e}")
    assert e_bf <= 0.5 * gap and e_16 <= 0.5 * gap and gap / scale < 5e-3
    close(outs["fused"][2].torch_nchw(), refbf + tail, rms=1.5e-3, mx=1e-2, what="bf16 operands")


def test_conv_win3x3_groupnorm_on_load_race_screen():
    """The fused path mixes register loads, LDS writes and LDS-DMA under hand-counted waits: same bits 40 times,
    on a shape with several chunk transitions and many workgroups."""
    B, H, W, Cin, Cout = 4, 48, 48, 256, 128
    x, w = rnd(B, Cin, H, W, seed=241), rnd(Cout, Cin, 3, 3, seed=242, scale=(9 * Cin) ** -0.5)
    g, be = 1 + 0.2 * rnd(Cin, seed=243), 0.5 * rnd(Cin, seed=244)
    xv, wp, ws = from_nchw(x.to(DEV)), packed(w), _WS()
    ref = None
    for it in range(40):
        y = new_view(B, H, W, Cout, DEV)
        K.conv_gemm(xv, wp, None, y, KH=3, gn=(g.to(DEV), be.to(DEV), 1e-6, True, ws))
        got = y.t[:B * H * W * Cout].clone()
        if ref is None:
            ref = got
            close(y.torch_nchw(), F.conv2d(F.silu(F.group_norm(x, 32, g, be, 1e-6)), w, padding=1), rms=6e-3, mx=4e-2,
                  what="gn on load race ref")
        else:
            assert torch.equal(got, ref), f"run {it} differs"


@pytest.mark.parametrize("B,H,W,Cin,Cout,gn,f32", [(2, 32, 48, 128, 3, True, True), (1, 16, 16, 64, 4, False, False),
                                                    (3, 48, 16, 32, 1, True, True)])
def test_conv3x3_few_out(B, H, W, Cin, Cout, gn, f32):
    """conv_out-style layer (<= 4 output channels, dot2 accumulation), with and without the GroupNorm+SiLU on load,
    input in a channel slice."""
    x, w, b = rnd(B, Cin, H, W, seed=271), rnd(Cout, Cin, 3, 3, seed=272, scale=(9 * Cin) ** -0.5), rnd(Cout, seed=273)
    g, be = 1 + 0.2 * rnd(Cin, seed=274), 0.5 * rnd(Cin, seed=275)
    xbuf = from_nchw(torch.cat([rnd(B, 32, H, W, seed=276), x], 1).to(DEV))
    y = new_view(B, H, W, Cout, DEV, torch.float32 if f32 else K.a16())
    K.conv3x3_few_out(xbuf.slice(32, Cin), packed(w), b.to(DEV), y, gn=(g.to(DEV), be.to(DEV), 1e-6, True, _WS()) if gn else None)
    xin = F.silu(F.group_norm(x, 32, g, be, 1e-6)) if gn else x
    close(y.torch_nchw(), F.conv2d(xin, w, b, padding=1), rms=6e-3, mx=4e-2, what="few-out conv")


def test_conv_groupnorm_argument_on_other_kernels():
    """gn= on a layer the window kernel cannot take (1x1): falls back to the separate pass, same result."""
    B, H, W, Cin, Cout = 2, 12, 12, 128, 256
    x, w = rnd(B, Cin, H, W, seed=211), rnd(Cout, Cin, 1, 1, seed=212, scale=Cin ** -0.5)
    g, be = 1 + 0.2 * rnd(Cin, seed=213), 0.5 * rnd(Cin, seed=214)
    y = new_view(B, H, W, Cout, DEV)
    K.conv_gemm(from_nchw(x.to(DEV)), packed(w), None, y, KH=1, gn=(g.to(DEV), be.to(DEV), 1e-5, True, _WS()))
    close(y.torch_nchw(), F.conv2d(F.silu(F.group_norm(x, 32, g, be, 1e-5)), w), rms=6e-3, mx=4e-2, what="gn fallback")


def test_conv_stats_identical_across_kernel_families():
    """Which conv kernel runs depends on the problem size (hence on the batch size).  Outputs AND the
    GroupNorm partial sums left by the epilogue must not: 4-wave, 128x256 and 256x256 kernels, bit for bit."""
    x, w, b = rnd(2, 128, 32, 32, seed=141), rnd(256, 128, 3, 3, seed=142, scale=1 / 34), rnd(256, seed=143)
    outs = []
    for tile in (128, 129, 256):
        y = new_view(2, 32, 32, 256, DEV, stats=True)
        K.conv_gemm(from_nchw(x.to(DEV)), packed(w), b.to(DEV), y, KH=3, tile=tile, stats=True)
        outs.append((y.t[:2 * 32 * 32 * 256].clone(), y.stats.t.clone()))
    for o, s in outs[1:]:
        assert torch.equal(o, outs[0][0]) and torch.equal(s, outs[0][1])


@pytest.mark.parametrize("n,n_pad", [(1024, 1024), (2304, 2304), (600, 640)])
def test_softmax_rows_long(n, n_pad):
    s = torch.randn(9, n, generator=torch.Generator().manual_seed(151)) * 4
    p = torch.empty(9, n_pad, dtype=K.a16(), device=DEV)
    K.softmax_rows(s.to(DEV), p, 9, n, n_pad)
    close(p[:, :n].float(), torch.softmax(s, -1), what="softmax long rows")
    assert float(p[:, n:].float().abs().max() if n_pad > n else 0.0) == 0.0


@pytest.mark.parametrize("tile", [128, 129, 130, 131])
def test_group_norm_pair_statistics_320_and_960_channels(tile):
    """Group sizes of 10 / 30 channels (320 / 960-channel buffers of the UNet): the epilogue leaves one entry per
    channel PAIR; concat buffer 640 + 320 = 960 normalised as a whole, and its 320-channel tail on its own."""
    B, H, W = 2, 16, 16
    xa, xb = rnd(B, 128, H, W, seed=161), rnd(B, 64, H, W, seed=162)
    wa, wb = rnd(640, 128, 3, 3, seed=163, scale=1 / 34), rnd(320, 64, 3, 3, seed=164, scale=1 / 24)
    cat = new_view(B, H, W, 960, DEV, stats=2)
    K.conv_gemm(from_nchw(xa.to(DEV)), packed(wa), None, cat.slice(0, 640), KH=3, stats=True, tile=tile)
    K.conv_gemm(from_nchw(xb.to(DEV)), packed(wb), None, cat.slice(640, 320), KH=3, stats=True, tile=tile)
    assert cat.stats.gran == 2 and cat.stats.covers(0, 960)
    ref_cat = torch.cat([F.conv2d(xa, wa, padding=1), F.conv2d(xb, wb, padding=1)], 1)
    for view, ref in ((cat, ref_cat), (cat.slice(640, 320), ref_cat[:, 640:])):
        C = view.C
        g, be = rnd(C, seed=165), rnd(C, seed=166)
        out = new_view(B, H, W, C, DEV)
        ws = _WS()
        K.group_norm(view, g.to(DEV), be.to(DEV), out, 1e-5, True, ws)
        assert "gn_partial" not in ws.t                       # the statistics pass really was skipped
        close(out.torch_nchw(), F.silu(F.group_norm(ref, 32, g, be, 1e-5)), rms=6e-3, mx=4e-2, what=f"pair stats C={C}")
    # a 320-channel buffer of its own picks pairs by itself
    y = new_view(B, H, W, 320, DEV, stats=True)
    assert y.stats.gran == 2


# ---- round 4: split-K for the 12x12 level, conv_shortcut as a second source of the window kernel ------------------------
@pytest.mark.parametrize("Cin,Cout,res", [(1280, 1280, True), (2560, 1280, False), (1280, 256, True), (768, 768, True)])
def test_conv_splitk_small_maps(Cin, Cout, res):
    """3x3 layers of the UNet's 12x12 level (decoder_unet.py:204-293) on the split-K form: against F.conv2d, against the
    unsplit kernels, the same bits on every launch, and a picture alone == inside a batch (the number of parts follows
    from the layer's shape, never from B)."""
    B, H, W = 3, 12, 12
    x, w, b = rnd(B, Cin, H, W, seed=401), rnd(Cout, Cin, 3, 3, seed=402, scale=(9 * Cin) ** -0.5), rnd(Cout, seed=403)
    r = rnd(B, Cout, H, W, seed=404) if res else None
    xv, wp, rv = from_nchw(x.to(DEV)), packed(w), (from_nchw(r.to(DEV)) if res else None)
    runs = []
    for split in (0, 1, 0):
        K.SPLIT_K = split
        try:
            y = new_view(B, H, W, Cout, DEV)
            K.conv_gemm(xv, wp, b.to(DEV), y, KH=3, res=rv)
        finally:
            K.SPLIT_K = 0
        runs.append(y.t[:B * H * W * Cout].clone())
        if split == 0:
            yk = y
    ref = F.conv2d(x, w, b, padding=1) + (r if res else 0)
    close(yk.torch_nchw(), ref, what="split-K")
    assert torch.equal(runs[0], runs[2])                                       # deterministic
    close(runs[0].float().view(B, H, W, Cout), runs[1].float().view(B, H, W, Cout), rms=2e-3, mx=1e-2, what="split vs unsplit")
    y1 = new_view(1, H, W, Cout, DEV)
    K.conv_gemm(from_nchw(x[1:2].to(DEV)), wp, b.to(DEV), y1, KH=3, res=(from_nchw(r[1:2].to(DEV)) if res else None))
    assert torch.equal(y1.t[:H * W * Cout], runs[0].view(B, -1)[1])            # batch-invariant


def test_conv_splitk_dispatch_is_by_shape():
    """The dispatcher's own answer: split for the 12x12 level's 3x3 layers at any B, not for short K, large maps or 1x1."""
    def kid(B, H, Cin, Cout, k, ws=True):
        x = new_view(B, H, H, Cin, DEV)
        y = new_view(B, H, H, Cout, DEV)
        p = K.ConvParams()
        p.x, p.w, p.y, p.zeros = x.ptr, x.ptr, y.ptr, K.zeros_page(DEV).data_ptr()
        p.B, p.IH, p.IW, p.Cin, p.x_cs = B, H, H, Cin, Cin
        p.OH, p.OW, p.Cout, p.y_cs = H, H, Cout, Cout
        p.KH = p.KW = k
        p.stride, p.pad, p.alpha = 1, k // 2, 1.0
        need = K.lib().onedc_conv_splitk_ws_bytes(p)
        if need and ws:
            t = K.splitk_workspace(DEV, need)
            p.ws, p.ws_bytes = t.data_ptr(), t.numel()
        return K.conv_kernel_name(p), need
    for B in (1, 2, 16):
        name, need = kid(B, 12, 1280, 1280, 3)
        assert name == "splitk_256" and need == 5 * B * 144 * 1280 * 4
    name, need = kid(16, 12, 768, 768, 3)                                      # the codec's 768-channel layers: four ranges
    assert name == "splitk_256" and need == 4 * 16 * 144 * 768 * 4
    with pytest.raises(K.KernelError, match="-15"):                            # no scratch: refused since ABI 5, never run unsplit
        kid(16, 12, 1280, 1280, 3, ws=False)
    assert kid(16, 12, 1280, 1280, 1)[1] == 0 and kid(16, 24, 1280, 1280, 3)[1] == 0 and kid(16, 12, 320, 1280, 3)[1] == 0


@pytest.mark.parametrize("B,H,W,Cin,Cout,Cin2,gn,alpha", [(2, 32, 16, 128, 128, 256, True, 2.0 ** -6), (1, 16, 48, 256, 256, 512, False, 1.0),
                                                          (2, 16, 16, 320, 320, 960, False, 1.0), (1, 32, 32, 128, 128, 32, True, 1.0),
                                                          (1, 16, 16, 64, 128, 96, False, 0.25)])
def test_conv_win4_second_source(B, H, W, Cin, Cout, Cin2, gn, alpha):
    """conv3x3([silu(GroupNorm(.))] h) + conv1x1(x2) in ONE launch of the 4-wave window kernel (diffusers ResnetBlock2D: conv2
    + conv_shortcut, called from model_sd15_with_codec_stage1.py:184-188) against F.conv2d and against the two separate
    launches; x2 is a channel slice of a wider buffer, GroupNorm statistics of the output left by the epilogue."""
    h, w, b = rnd(B, Cin, H, W, seed=411), rnd(Cout, Cin, 3, 3, seed=412, scale=(9 * Cin) ** -0.5), rnd(Cout, seed=413)
    x2, w2, b2 = rnd(B, Cin2, H, W, seed=414), rnd(Cout, Cin2, 1, 1, seed=415, scale=Cin2 ** -0.5), rnd(Cout, seed=416)
    g, be = 1 + 0.2 * rnd(Cin, seed=417), 0.5 * rnd(Cin, seed=418)
    hv = from_nchw(h.to(DEV))
    x2buf = from_nchw(torch.cat([rnd(B, 64, H, W, seed=419), x2], 1).to(DEV))
    x2v = x2buf.slice(64, Cin2)
    ws = _WS()
    gnarg = (g.to(DEV), be.to(DEV), 1e-6, True, ws) if gn else None
    y = new_view(B, H, W, Cout, DEV, stats=True)
    w2p = pack.pack_matrix(pack.conv_matrix(w2) / alpha, DEV)
    out = K.conv_gemm(hv, packed(w), (b + b2).to(DEV), y, KH=3, alpha=alpha, stats=True, gn=gnarg, x2=x2v, w2=w2p)
    assert out is not None, "the dispatcher refused the second source"
    hn = F.silu(F.group_norm(h, 32, g, be, 1e-6)) if gn else h
    ref = alpha * F.conv2d(hn, w, None, padding=1) + b.view(1, -1, 1, 1) + F.conv2d(x2, w2, b2)
    close(y.torch_nchw(), ref, rms=6e-3 if gn else 4e-3, mx=4e-2, what="second source")
    # the two separate launches (shortcut first, its output as the residual; the generic kernels read 64-channel K-tiles)
    if Cin2 % 64 == 0:
        sc = new_view(B, H, W, Cout, DEV)
        K.conv_gemm(x2v, packed(w2), b2.to(DEV), sc, KH=1)
        y2 = new_view(B, H, W, Cout, DEV, stats=True)
        K.conv_gemm(hv, packed(w), b.to(DEV), y2, KH=3, alpha=alpha, stats=True, gn=gnarg, res=sc)
        close(y.torch_nchw(), y2.torch_nchw(), rms=3e-3, mx=2e-2, what="one launch vs two")
    # the statistics the epilogue left are those of the stored values
    gg, bb = rnd(Cout, seed=420), rnd(Cout, seed=421)
    o = new_view(B, H, W, Cout, DEV)
    ws2 = _WS()
    K.group_norm(y, gg.to(DEV), bb.to(DEV), o, 1e-6, True, ws2)
    assert "gn_partial" not in ws2.t
    close(o.torch_nchw(), F.silu(F.group_norm(y.torch_nchw().cpu(), 32, gg, bb, 1e-6)), rms=6e-3, mx=4e-2, what="stats behind x2")


def test_conv_win4_second_source_race_screen_and_refusal():
    B, H, W, Cin, Cout, Cin2 = 4, 48, 48, 128, 128, 256
    h, w = rnd(B, Cin, H, W, seed=431), rnd(Cout, Cin, 3, 3, seed=432, scale=(9 * Cin) ** -0.5)
    x2, w2 = rnd(B, Cin2, H, W, seed=433), rnd(Cout, Cin2, 1, 1, seed=434, scale=Cin2 ** -0.5)
    hv, x2v, wp, w2p = from_nchw(h.to(DEV)), from_nchw(x2.to(DEV)), packed(w), packed(w2)
    ref = None
    for it in range(30):
        y = new_view(B, H, W, Cout, DEV)
        assert K.conv_gemm(hv, wp, None, y, KH=3, x2=x2v, w2=w2p) is not None
        got = y.t[:B * H * W * Cout].clone()
        if ref is None:
            ref = got
            close(y.torch_nchw(), F.conv2d(h, w, None, padding=1) + F.conv2d(x2, w2), what="second source race ref")
        else:
            assert torch.equal(got, ref), f"run {it} differs"
    y1 = new_view(1, H, W, Cout, DEV)
    K.conv_gemm(from_nchw(h[2:3].to(DEV)), wp, None, y1, KH=3, x2=from_nchw(x2[2:3].to(DEV)), w2=w2p)
    assert torch.equal(y1.t[:H * W * Cout], ref.view(B, -1)[2])
    # a layer the window kernel does not take (12x12 map): nothing is launched, the caller is told
    y = new_view(1, 12, 12, Cout, DEV)
    y.t.fill_(7.0)
    assert K.conv_gemm(from_nchw(h[:1, :, :12, :12].to(DEV)), wp, None, y, KH=3, x2=from_nchw(x2[:1, :, :12, :12].to(DEV)), w2=w2p) is None
    assert float((y.t[:144 * Cout].float() - 7.0).abs().max()) == 0.0


@pytest.mark.parametrize("B,H,W,Cin,Cout,epi,res,stats", [
    (2, 24, 24, 320, 320, "linear", True, True),       # proj_out: residual + statistics (pairs: 320 channels), 64-row tail tile
    (1, 20, 20, 320, 2560, "geglu", False, False),     # ff.net.0 (GEGLU pairs), ragged M (400 pixels)
    (3, 8, 8, 1280, 640, "linear", True, False),       # long K
    (1, 16, 16, 64, 128, "silu", False, True),         # two K-steps: shorter than the ring
    (2, 12, 12, 384, 320, "linear", True, False),      # attention out-projection (padded heads), 12 K-steps
    (1, 9, 7, 96 + 32, 128, "leaky", False, False),    # 63 pixels: one partial tile
    (2, 24, 24, 640, 5120, "geglu", False, False),     # round 5: the gated layers run on the 128 x 256 tile (automatic)
    (1, 12, 12, 1280, 1024, "geglu", False, False),    # ... K = 1280, a map smaller than one pixel tile
    (1, 16, 20, 320, 256, "linear", True, True),       # ... and the same tile forced (136) on a plain layer with statistics
])
def test_conv_lin_kernel_equals_the_64_channel_kernel(B, H, W, Cin, Cout, epi, res, stats):
    """1x1 layers on conv_lin_kernel (32-channel K-steps, four-stage ring; tile 133) against F.conv2d and, BIT
    FOR BIT -- outputs and the GroupNorm statistics of the epilogue -- against conv_gemm_kernel (tile=128): same tiles, same k
    order.  The layers are the UNet's Linear / 1x1 convs (decoder_unet.py:204-293)."""
    glu = epi == "geglu"
    x, w, b = rnd(B, Cin, H, W, seed=441), rnd(Cout, Cin, 1, 1, seed=442, scale=Cin ** -0.5), rnd(Cout, seed=443)
    cstore = Cout // 2 if glu else Cout
    r = rnd(B, cstore, H, W, seed=444) if res else None
    perm = pack.glu_perm(Cout // 2) if glu else None
    xv, wp = from_nchw(x.to(DEV)), packed(w, perm)
    bd = (b[perm] if glu else b).to(DEV)
    rv = from_nchw(r.to(DEV)) if res else None
    code = {"linear": K.EPI_LINEAR, "geglu": K.EPI_GEGLU, "silu": K.EPI_SILU, "leaky": K.EPI_LEAKY}[epi]
    outs = []
    # the ring kernel, the 64-channel kernel, the same without line stores, 3-stage ring, automatic, the 128 x 256 tile (round 5)
    for tile in (133, 128, 132, 134, 0) + ((136,) if Cout % 128 == 0 else ()):
        y = new_view(B, H, W, cstore, DEV, stats=stats)
        K.conv_gemm(xv, wp, bd, y, KH=1, epi=code, slope=0.1, res=rv, stats=stats, tile=tile)
        outs.append((y.t[:B * H * W * cstore].clone(), y.stats.t.clone() if (stats and y.stats is not None) else None, y))
    full = F.conv2d(x, w, b)
    if glu:
        a, g = full.chunk(2, 1)
        ref = a * F.gelu(g)
    else:
        ref = {"linear": full, "silu": F.silu(full), "leaky": F.leaky_relu(full, 0.1)}[epi]
    ref = ref + (r if res else 0)
    close(outs[0][2].torch_nchw(), ref, what="conv_lin")
    for o in outs[1:]:
        assert torch.equal(outs[0][0], o[0])
        if outs[0][1] is not None:
            assert torch.equal(outs[0][1], o[1])


def test_conv_lin_kernel_batched_and_dispatch():
    """The dispatcher's answer for the 1x1 layers, and the batched form (blockIdx.z: V^T of the attention layers)."""
    G, n, Cc = 5, 144, 128
    q, k = rnd(G, n, Cc, seed=451), rnd(G, n, Cc, seed=452)
    qd = torch.zeros(G * n + 128, Cc, dtype=K.a16(), device=DEV)
    kd = torch.zeros(G * n + 128, Cc, dtype=K.a16(), device=DEV)
    qd[:G * n] = q.view(-1, Cc).to(DEV)
    kd[:G * n] = k.view(-1, Cc).to(DEV)
    outs = []
    for tile in (133, 128):
        s = torch.empty(G * n * n, dtype=torch.float32, device=DEV)
        K.conv_gemm(View(qd, 1, 1, n, Cc), kd, None, View(s, 1, 1, n, n), Cin=Cc, Cout=n, batch=G, x_bs=n * Cc,
                    w_bs=n * Cc, y_bs=n * n, alpha=0.25, tile=tile)
        outs.append(s)
    close(outs[0].view(G, n, n), 0.25 * q @ k.transpose(1, 2), rms=1e-4, mx=1e-4, what="batched nt on conv_lin")
    assert torch.equal(outs[0], outs[1])
    # the dispatcher's answers
    x = new_view(16, 12, 12, 1280, DEV)
    y = new_view(16, 12, 12, 1280, DEV)
    p = K.ConvParams()
    p.x, p.w, p.y, p.zeros = x.ptr, x.ptr, y.ptr, K.zeros_page(DEV).data_ptr()
    p.B, p.IH, p.IW, p.Cin, p.x_cs, p.OH, p.OW, p.Cout, p.y_cs = 16, 12, 12, 1280, 1280, 12, 12, 1280, 1280
    p.KH = p.KW = p.stride = 1
    p.alpha = 1.0
    assert K.conv_kernel_name(p) == "64x128"            # (the ring is a measured negative: never the automatic choice)
    p.tile = 133
    assert K.conv_kernel_name(p) == "lin_64x128"
    p.tile = 128
    assert K.conv_kernel_name(p) == "64x128"
    # 24x24 level, K = 1280 (20 K-tiles): four waves since round 4 (the 8-wave tiles from 32 K-tiles on)
    p.tile, p.IH, p.IW, p.OH, p.OW = 0, 24, 24, 24, 24
    assert K.conv_kernel_name(p) == "128x128"
    p.Cin = p.x_cs = 5120
    assert K.conv_kernel_name(p) == "256x256"
    # level 0, K = 320, wide output: the three-stage ring (three workgroups per CU); narrow output: the 64-row tile as before
    p.IH = p.IW = p.OH = p.OW = 96
    p.Cin, p.x_cs, p.Cout, p.y_cs = 320, 320, 1024, 1024
    assert K.conv_kernel_name(p) == "lin_128x128"
    p.Cout = p.y_cs = 320
    assert K.conv_kernel_name(p) == "64x128"
    # round 5: gated layers (GEGLU) with fewer than 32 K-tiles take the 128 x 256 tile of the ring kernel
    p.Cout, p.y_cs, p.epi = 2560, 1280, K.EPI_GEGLU
    assert K.conv_kernel_name(p) == "lin_128x256"


# ---- round 4: the f32 path on split operands (pairs of halves, 16-bit matrix rate) ---------------------------------------
@pytest.mark.parametrize("B,H,W,Cin,Cout,k,scale", [
    (2, 12, 12, 128, 128, 1, 1.0),
    (1, 9, 7, 96, 256, 3, 1.0),           # Cin = 3 x 32, odd spatial size
    (3, 8, 8, 256, 512, 1, 1.0),
    (2, 16, 16, 256, 256, 1, 300.0),      # activations in the hundreds (the random-init prior reaches 540)
    (2, 16, 16, 256, 256, 1, 1e-4),       # ... and tiny ones (low halves in the subnormal range)
])
def test_conv_gemm_f32_split16(B, H, W, Cin, Cout, k, scale):
    """onedc_conv_params.in_f32 == 2: f32 activations, weights and activations as pairs of halves (22 bits), three f16 MFMAs
    per 32 channels, f32 accumulation -- against an fp64 convolution at the tolerance of the exact-f32 kernel, against that
    kernel itself, bit-identical between the two tile shapes and between a picture alone and inside a batch."""
    g = torch.Generator().manual_seed(31)
    x = torch.randn(B, Cin, H, W, generator=g) * scale
    w = torch.randn(Cout, Cin, k, k, generator=g) * (Cin * k * k) ** -0.5
    b = torch.randn(Cout, generator=g) * scale
    r = torch.randn(B, Cout, H, W, generator=g) * scale
    xv, rv = from_nchw(x.to(DEV), torch.float32), from_nchw(r.to(DEV), torch.float32)
    ws = pack.pack_matrix_split16(pack.conv_matrix(w), DEV)
    wf = pack.pack_matrix(pack.conv_matrix(w), DEV, None, torch.float32)
    outs = {}
    for name, wt, tile in (("split", ws, 0), ("split128", ws, 128), ("exact", wf, 0)):
        y = new_view(B, H, W, Cout, DEV, torch.float32)
        K.conv_gemm(xv, wt, b.to(DEV), y, KH=k, epi=K.EPI_LEAKY, slope=0.01, res=rv, tile=tile)
        outs[name] = y.torch_nchw()
    ref = F.leaky_relu(_f64_conv(x, w, b, padding=k // 2), 0.01) + r.double()
    close(outs["split"], ref, rms=2e-6, mx=2e-6, what="split f32 conv")
    close(outs["split"], outs["exact"], rms=1e-6, mx=2e-6, what="split vs exact f32 conv")
    assert torch.equal(outs["split"], outs["split128"])
    y1 = new_view(1, H, W, Cout, DEV, torch.float32)
    K.conv_gemm(from_nchw(x[-1:].to(DEV), torch.float32), ws, b.to(DEV), y1, KH=k, epi=K.EPI_LEAKY, slope=0.01,
                res=from_nchw(r[-1:].to(DEV), torch.float32))
    assert torch.equal(y1.torch_nchw(), outs["split"][-1:])


def test_conv_gemm_f32_split16_glu_shuffle_and_accuracy_against_the_exact_kernel():
    g = torch.Generator().manual_seed(32)
    B, H, W, C = 2, 6, 10, 128
    x = torch.randn(B, C, H, W, generator=g)
    w = torch.randn(4 * C, C, 1, 1, generator=g) * C ** -0.5
    b = torch.randn(4 * C, generator=g)
    xv = from_nchw(x.to(DEV), torch.float32)
    perm = pack.glu_perm(2 * C)
    y = new_view(B, H, W, 2 * C, DEV, torch.float32)
    K.conv_gemm(xv, pack.pack_matrix_split16(pack.conv_matrix(w), DEV, perm), b[perm].to(DEV), y, KH=1, epi=K.EPI_GLU_LEAKY)
    t = _f64_conv(x, w, b)
    close(y.torch_nchw(), F.leaky_relu(t[:, :2 * C], 0.1) + F.leaky_relu(t[:, 2 * C:], 0.01), rms=2e-6, mx=2e-6, what="split glu")
    perm = pack.shuffle_perm(4 * C)
    y = new_view(B, 2 * H, 2 * W, C, DEV, torch.float32)
    K.conv_gemm(xv, pack.pack_matrix_split16(pack.conv_matrix(w), DEV, perm), b[perm].to(DEV), y, KH=1, pixel_shuffle=1)
    close(y.torch_nchw(), F.pixel_shuffle(t, 2), rms=2e-6, mx=2e-6, what="split pixel shuffle")
    # error of the two f32 forms against fp64 on a long sum: the split form is not the worse one by more than a factor 2
    xl = torch.randn(1, 512, 16, 16, generator=g)
    wl = torch.randn(256, 512, 1, 1, generator=g) / 512 ** 0.5
    ref = _f64_conv(xl, wl, None)
    errs = {}
    for name, wt in (("split", pack.pack_matrix_split16(pack.conv_matrix(wl), DEV)), ("exact", pack.pack_matrix(pack.conv_matrix(wl), DEV, None, torch.float32))):
        yl = new_view(1, 16, 16, 256, DEV, torch.float32)
        K.conv_gemm(from_nchw(xl.to(DEV), torch.float32), wt, None, yl, KH=1)
        errs[name] = float((yl.torch_nchw().double().cpu() - ref).pow(2).mean().sqrt() / ref.pow(2).mean().sqrt())
    print(f"[split16] rel. RMS error against fp64, K = 512: split {errs['split']:.2e}, exact f32 chain {errs['exact']:.2e}")
    assert errs["split"] <= max(2 * errs["exact"], 3e-7)
