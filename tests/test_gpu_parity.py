"""Parity of the HIP decode path (through the C ABI) with the fp32 oracle (oracle/torch_ref.py,
itself pinned to the reference by tests/test_oracle_codec.py), stage by stage and end to end, with
the deterministic synthetic weights of onedc_amd.synth.

Floating-point tolerances are written at each check and KEYED BY THE 16-BIT FORMAT (`tol(f16, bf16)`): the product
computes in IEEE half (default) or bf16 (ONEDC_ACT=bf16, BASELINE config "bf16") with fp32 accumulation, the oracle is
fp32.  bf16 rounds 8x coarser than f16, so one number for both would let a 5-10x regression of the f16 kernels pass: the
f16 bounds are 2-3x the values measured on MI355X (printed by every check; round 3: semantic adaptor 9.5e-4,
synthesis 1.0e-3, vae_reduction 3.5e-4, UNet eps 1.5e-3, VAE decoder 1.5e-3, VAE encoder 7.5e-4, analysis transform 1.3e-3 / 1.4e-3,
end-to-end 67.8 dB), the bf16 bounds the round-1 ones.
Integer/byte results (symbols, bitstreams, round trips) are checked bit-exactly.
"""
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from conftest import golden
from onedc_amd import kernels as K
from onedc_amd import synth
from onedc_amd.codec_module import IntraNoAR
from onedc_amd.decoder_unet import UNet
from onedc_amd.kernels import View, from_nchw, new_view
from onedc_amd.layers import (DepthConvBlock4, ResidualBlockUpsample, VQAttnBlock, VQResnetBlock, Workspace)
from onedc_amd.model_sd15_with_codec_stage1 import SD15_1step_codec_stage1
from onedc_amd.vae import VAEDecoder
from oracle import rans_port
from oracle import torch_ref as R

DEV = "cuda:0"


def rel_rms(got, ref):
    got, ref = got.double().cpu(), ref.double().cpu()
    return float((got - ref).pow(2).mean().sqrt() / ref.pow(2).mean().sqrt().clamp_min(1e-12))


def tol(f16, bf16):
    """Tolerance for the library under test (`K.act()`)."""
    return f16 if K.act() == "f16" else bf16


def check(got, ref, tol, what):
    assert got.shape == ref.shape, (what, got.shape, ref.shape)
    assert torch.isfinite(got).all(), what
    e = rel_rms(got, ref)
    print(f"[parity] {what}: rel_rms = {e:.3e} (tol {tol:.1e})")
    assert e <= tol, f"{what}: rel_rms {e:.3e} > {tol:.1e}"


@pytest.fixture(scope="module")
def codec_sd():
    return synth.codec_state_dict(0)


@pytest.fixture(scope="module")
def codec(codec_sd):
    m = IntraNoAR(codec_sd, DEV)          # default: prior networks in exact f32
    m.update(force=True)
    return m


@pytest.fixture(scope="module")
def full_model():
    """The whole codec (decode + encode side) with synthetic weights, built once for the module."""
    m = SD15_1step_codec_stage1({"vae_attn_patch": 16}).load_synthetic(0).load_synthetic_encoder(0)
    m.codec_model.update(force=True)
    return m


@pytest.fixture(scope="module")
def codec_exact(codec_sd):
    m = IntraNoAR(codec_sd, DEV, prior_f32="exact")      # the f32 matrix instructions: rounds 1-3's arithmetic
    m.update(force=True)
    return m


@pytest.fixture(scope="module")
def codec_bf16(codec_sd):
    m = IntraNoAR(codec_sd, DEV, prior_f32=False)
    m.update(force=True)
    return m


def test_codec_blocks(codec_sd):
    """DCB4 / RBU / ResnetBlock / AttnBlock on the golden block inputs. tol: 1e-2 rel. RMS."""
    g = golden("g5_blocks.npz")
    ws = Workspace(DEV)
    x128 = torch.from_numpy(g["x128"]).to(torch.bfloat16).float()
    x768 = torch.from_numpy(g["x768"]).to(torch.bfloat16).float()
    with torch.no_grad():
        y = DepthConvBlock4(codec_sd, "hyper_dec.to_entropy.0", DEV)(from_nchw(x128.to(DEV)))
        check(y.torch_nchw(), R.depth_conv_block4(codec_sd, "hyper_dec.to_entropy.0", x128), tol(2e-3, 1e-2), "DCB4 128->128")
        y = DepthConvBlock4(codec_sd, "y_prior_fusion.0", DEV)(from_nchw(x128.to(DEV)))
        check(y.torch_nchw(), R.depth_conv_block4(codec_sd, "y_prior_fusion.0", x128), tol(2e-3, 1e-2), "DCB4 128->256")
        y = ResidualBlockUpsample(codec_sd, "hyper_dec.to_entropy.1", DEV)(from_nchw(x128.to(DEV)))
        check(y.torch_nchw(), R.residual_block_upsample(codec_sd, "hyper_dec.to_entropy.1", x128), tol(2e-3, 1e-2), "RBU")
        y = VQResnetBlock(codec_sd, "semantic_adaptor.to_semantic.1", DEV)(from_nchw(x768.to(DEV)), ws)
        check(y.torch_nchw(), R.vq_resnet_block(codec_sd, "semantic_adaptor.to_semantic.1", x768), tol(2e-3, 1e-2), "ResnetBlock")
        y = VQAttnBlock(codec_sd, "semantic_adaptor.to_semantic.2", DEV, 768)(from_nchw(x768.to(DEV)), ws)
        check(y.torch_nchw(), R.vq_attn_block(codec_sd, "semantic_adaptor.to_semantic.2", x768), tol(2e-3, 1e-2), "AttnBlock")


def test_hyper_decoder_and_prior_params(codec, codec_sd):
    """FSQ codes -> HyperDecoder -> y_prior_fusion against the oracle AND the reference-made golden."""
    g = golden("g5_codec_path.npz")
    idx = torch.from_numpy(g["indices"]).to(DEV)
    fused, z_sem = codec._params_from_indices(idx)
    check(z_sem.torch_nchw(), torch.from_numpy(g["z_semantic"]), tol(1e-3, 4e-3), "z_semantic vs reference golden")
    # the prior networks run in exact f32 (f32 MFMA): tol 1e-5 rel. RMS against the reference's fp32 CPU result
    check(fused.torch_nchw(), torch.from_numpy(g["fused"]), 1e-5, "prior params (f32 path) vs reference golden")
    with torch.no_grad():
        ref, _ = R.z_indices_to_params(codec_sd, torch.from_numpy(g["indices"]))
    check(fused.torch_nchw(), ref, 1e-5, "prior params (f32 path) vs oracle")


def test_prior_f32_modes_pick_their_kernels_and_round_trip(codec_sd):
    """prior_f32 True (default): f32 activations, operands as pairs of halves on the 16-bit matrix cores (in_f32 = 2);
    "exact": the f32 matrix instructions of rounds 1-3.  Both round-trip their own streams bit for bit; their parameters agree
    to 1e-6 (and both with the oracle to 1e-5, test_hyper_decoder_and_prior_params runs on the default)."""
    names = {}
    orig = K.conv_gemm

    def spy(x, w, bias, y, **kw):
        r = orig(x, w, bias, y, **kw)
        if x.f32:
            names.setdefault(w.dtype, 0)
            names[w.dtype] += 1
        return r
    import onedc_amd.layers as L
    outs = {}
    for mode in (True, "exact"):
        m = IntraNoAR(codec_sd, DEV, prior_f32=mode, prior_arith="mfma")
        m.update(force=True)
        names.clear()
        K.conv_gemm = L.K.conv_gemm = spy
        try:
            idx = torch.from_numpy(np.random.default_rng(7).integers(0, 16384, (2, 2, 3)).astype(np.int32)).to(DEV)
            fused, _ = m._params_from_indices(idx)
            blobs, y_hat = m.encode_latents(idx, None, noise_seed=3)
            from onedc_amd.entropy.utils import decode_i
            y_dec, _ = m.decompress([decode_i(b, 14, 64)["bit_stream_y"] for b in blobs], idx)
        finally:
            K.conv_gemm = L.K.conv_gemm = orig
        assert torch.equal(y_dec.torch_nchw(), y_hat.torch_nchw())
        assert set(names) == ({torch.float16} if mode is True else {torch.float32}), names
        outs[mode] = fused.torch_nchw()
    check(outs[True], outs["exact"], 2e-6, "split-half prior parameters vs the exact-f32 ones")


def test_prior_params_bf16_mode(codec_bf16, codec_sd):
    """prior_f32=False (the reference's GPU-autocast precision): bf16 tolerance 2e-2."""
    g = golden("g5_codec_path.npz")
    fused, _ = codec_bf16._params_from_indices(torch.from_numpy(g["indices"]).to(DEV))
    check(fused.torch_nchw(), torch.from_numpy(g["fused"]), 2e-2, "prior params (bf16 mode) vs reference golden")


@pytest.mark.parametrize("mode", ["f32", "f32_exact", "bf16"])
def test_entropy_round_trip_and_oracle_agreement(mode, codec, codec_exact, codec_bf16, codec_sd, gaussian_tables):
    """encode_latents -> container -> decode: y_hat must come back BIT-EXACT (integer path), for
    a ragged batch size, and image 1 alone must give the same bytes as inside the batch.
    Against the fp32 oracle the quantised latent can only differ by rounding flips
    (y_hat = round(y - mu) + mu):
      f32 prior nets (default): a seed flip needs y - mu within ~1e-6 of a half-integer (about one per
      1e5 symbols, measured by test_prior_steps_teacher_forced) but then changes the next steps'
      parameters in its neighbourhood, so free-running flips are asserted <= 2 %, stream size within 0.5 %;
      bf16 prior nets: flips are a property of bf16 matrix products (reported; the reference's GPU
      autocast path has them too), |delta| <= 1.25 everywhere, stream size within 3 %."""
    codec = {"f32": codec, "f32_exact": codec_exact, "bf16": codec_bf16}[mode]
    g = golden("g5_codec_path.npz")
    B = 3
    rng = np.random.default_rng(5)
    idx = torch.from_numpy(rng.integers(0, 16384, (B, 2, 3)).astype(np.int32)).to(DEV)
    fused, _ = codec._params_from_indices(idx)
    H, W = fused.H, fused.W
    y = (fused.torch_nchw().cpu()[:, 128:] + 3.0 * torch.randn(B, 128, H, W, generator=torch.Generator().manual_seed(6)))
    y_nhwc = y.permute(0, 2, 3, 1).contiguous().to(DEV)
    blobs, y_hat_enc = codec.encode_latents(idx, y_nhwc, heights=[128, 100, 128], widths=[192, 192, 131])
    x_hat, y_sem, orig_hw, pad_hw, pad = codec.decode(stream=blobs)
    assert orig_hw == [(128, 192), (100, 192), (128, 131)] and pad_hw == (128, 192)
    # decode again, recovering y_hat, and compare with the encoder's
    from onedc_amd.entropy.utils import decode_i, unpack_indices
    infos = [decode_i(b, 14, 64) for b in blobs]
    y_hat_dec, _ = codec.decompress([d["bit_stream_y"] for d in infos], idx)
    assert torch.equal(y_hat_dec.torch_nchw(), y_hat_enc.torch_nchw()), "decoder did not reproduce the encoder's y_hat"
    assert np.array_equal(np.stack([unpack_indices(d["bit_stream_z"], 6, 14) for d in infos]).reshape(B, 2, 3),
                          idx.cpu().numpy())
    # batch invariance of the whole entropy path: image 1 alone gives the same bytes
    blob1, _ = codec.encode_latents(idx[1:2], y_nhwc[1:2].contiguous(), heights=[100], widths=[192])
    assert blob1[0] == blobs[1]

    # oracle on the same inputs
    with torch.no_grad():
        fused_ref, _ = R.z_indices_to_params(codec_sd, idx.cpu())
        y_q_w, s_w, y_hat_ref = R.compress_four_part_prior(codec_sd, y, fused_ref)
    sym_ref = torch.stack(y_q_w, 1).to(torch.int16).numpy()                # (B,4,32,H,W)
    # product symbols: decode our own streams with the ORACLE coder and the oracle's index planes is not
    # possible when indexes differ, so compare y_hat instead: y_hat = round(y - mu) + mu
    diff = (y_hat_dec.torch_nchw().cpu() - y_hat_ref).abs()
    frac = float((diff > 0.5).float().mean())
    print(f"[parity] y_hat rounding flips vs oracle: {frac:.4%}, max |delta| = {float(diff.max()):.3f}, "
          f"mean |delta| = {float(diff.mean()):.4f}")
    assert float(diff.max()) <= 1.25
    if mode == "f32":
        # (which seed flips occur, and how far each cascades through the random-init nets, is a chaotic function of the last
        # bits of the arithmetic: 1.x % with the f32 matrix instructions of rounds 1-3, 3.0 % with round 4's split-half
        # products -- which are CLOSER to fp64 than the f32 chain, test_conv_gemm_f32_split16_... -- on these 36 864 symbols;
        # the measure of the arithmetic itself is the teacher-forced test below, <= 5e-4)
        assert frac <= 5e-2, frac
    if mode == "f32_exact":
        assert frac <= 2e-2, frac                    # (rounds 1-3's bound, kept for the arithmetic it was measured on)
    bits_ref = sum(len(rans_port.encode(sym_ref[b].reshape(-1),
                                        torch.stack([R.build_indexes(s[b:b + 1]) for s in s_w]).to(torch.int16).numpy().reshape(-1),
                                        rans_port.Tables(*gaussian_tables))) for b in range(B))
    bits_got = sum(len(d["bit_stream_y"]) for d in infos)
    print(f"[parity] y stream bytes: product {bits_got}, oracle {bits_ref} ({bits_got / bits_ref - 1:+.3%})")
    assert abs(bits_got / bits_ref - 1) <= (0.03 if mode == "bf16" else 5e-3)


def test_auto_decodes_a_stream_written_in_the_exact_f32_arithmetic(codec, codec_exact):
    """A y stream written by this package with `prior_f32: exact` (rounds 1-3's encoder) on a model in the default split-half
    arithmetic: where the two arithmetics take different table-index decisions the first attempt desynchronises, and
    `prior_arith: auto` must reach the exact-f32 twin BEFORE the reference's CPU classes and return the writer's y_hat bit for
    bit.  (The container records no arithmetic: ADVICE round 4.)"""
    B = 2
    rng = np.random.default_rng(77)
    idx = torch.from_numpy(rng.integers(0, 16384, (B, 12, 12)).astype(np.int32)).to(DEV)
    blobs, y_hat_enc = codec_exact.encode_latents(idx, None, noise_seed=5)
    from onedc_amd.entropy.utils import decode_i
    streams = [decode_i(b, 14, 64)["bit_stream_y"] for b in blobs]
    assert codec.prior_arith == "auto" and codec.prior_f32 and not codec.prior_exact
    y_hat, _ = codec.decompress(streams, idx)
    assert torch.equal(y_hat.torch_nchw(), y_hat_enc.torch_nchw())
    assert codec.last_auto_arith in ("mfma", "mfma_exact"), codec.last_auto_arith
    # the writer's own arithmetic decodes at the first try; and the default model is back on its own networks afterwards
    y2, _ = codec_exact.decompress(streams, idx)
    assert torch.equal(y2.torch_nchw(), y_hat_enc.torch_nchw()) and codec_exact.last_auto_arith == "mfma"
    blobs_d, y_d = codec.encode_latents(idx, None, noise_seed=5)
    y3, _ = codec.decompress([decode_i(b, 14, 64)["bit_stream_y"] for b in blobs_d], idx)
    assert torch.equal(y3.torch_nchw(), y_d.torch_nchw()) and codec.last_auto_arith == "mfma"


def test_prior_steps_teacher_forced(codec, codec_sd):
    """The four prior steps one at a time, each fed the ORACLE's y_hat_so_far (so one early rounding
    flip cannot cascade): parameters within 1e-5 rel. RMS of the fp32 oracle, and the integer decisions
    taken from them -- symbols round((y - mu) m) and rANS table indexes (build_indexes,
    entropy_models.py:355-362) -- equal to the oracle's except for <= 5e-4 of the positions (a 1e-6
    relative difference of sigma moves (ln sigma - ln 0.11)/0.025 by 4e-5: ~1e-4 of the truncations flip)."""
    B, N = 3, 128
    rng = np.random.default_rng(15)
    idx = torch.from_numpy(rng.integers(0, 16384, (B, 2, 3)).astype(np.int32))
    with torch.no_grad():
        fused_ref, _ = R.z_indices_to_params(codec_sd, idx)
        common = R.conv(codec_sd, "y_spatial_prior_reduction", fused_ref)
    H, W = fused_ref.shape[2:]
    y = fused_ref[:, N:] + 3.0 * torch.randn(B, N, H, W, generator=torch.Generator().manual_seed(16))
    masks = R.four_part_masks(B, N, H, W)
    fused, _ = codec._params_from_indices(idx.to(DEV))
    common_cat, y_hat = codec._prior_setup(fused)
    n = (N // 4) * H * W
    idx_dev = torch.empty(B * n, dtype=torch.int16, device=DEV)
    yhs = None
    bad_sym = bad_idx = total = 0
    for k in range(4):
        if k == 0:
            params_ref, params = fused_ref, fused
        else:
            with torch.no_grad():
                params_ref = torch.cat(R._prior_step(codec_sd, k, yhs, common), 1)
            y_hat.t[:B * H * W * y_hat.cs].view(B, H, W, y_hat.cs)[..., :N] = yhs.permute(0, 2, 3, 1).to(DEV)
            params = codec._next_params(k, y_hat, common_cat)
        got = params.torch_nchw().cpu()
        check(got, params_ref, 1e-5, f"prior params step {k} (teacher forced)")
        m = masks[k]
        sym_ref = torch.round((y - params_ref[:, N:] * m) * m)
        sym_got = torch.round((y - got[:, N:] * m) * m)
        bad_sym += int((sym_ref != sym_got).sum())
        K.prior_indexes(params, idx_dev, B, H, W, N, k)
        idx_ref = R.build_indexes(R.combine_for_writing(params_ref[:, :N] * m)).to(torch.int16)
        bad_idx += int((idx_dev.view(B, N // 4, H, W).cpu() != idx_ref).sum())
        total += B * n
        cur = sym_ref + params_ref[:, N:] * m
        yhs = cur if k == 0 else yhs + cur
    print(f"[parity] teacher-forced decisions: {bad_sym} symbol and {bad_idx} index mismatches in {total}")
    assert total == 4 * B * n and bad_sym <= 5e-4 * total and bad_idx <= 5e-4 * total


def test_z_only_prior(codec, codec_sd):
    g = golden("g5_codec_path.npz")
    idx = torch.from_numpy(g["indices"])
    y_hat, _ = codec.recon_with_z(idx.to(DEV))
    with torch.no_grad():
        fused_ref, _ = R.z_indices_to_params(codec_sd, idx)
        ref = R.recon_with_z_four_part_prior(codec_sd, fused_ref)
    check(y_hat.torch_nchw(), ref, 1e-5, "z-only y_hat (f32 prior) vs oracle")
    check(y_hat.torch_nchw(), torch.from_numpy(g["y_hat_zonly"]), 1e-5, "z-only y_hat (f32 prior) vs reference golden")


def test_semantic_adaptor_and_synthesis(codec, codec_sd):
    g = golden("g5_codec_path.npz")
    y_hat = torch.from_numpy(g["y_hat"])
    z_sem = torch.from_numpy(g["z_semantic"])
    yv = View(y_hat.permute(0, 2, 3, 1).contiguous().to(DEV).reshape(-1), 1, y_hat.shape[2], y_hat.shape[3], 128)
    x_hat, y_sem = codec.synthesis(yv, from_nchw(z_sem.to(DEV)))
    with torch.no_grad():
        ys_ref = R.semantic_adaptor(codec_sd, z_sem)
        xh_ref = R.codec_decoder(codec_sd, y_hat, ys_ref)
    check(y_sem.torch_nchw(), ys_ref, tol(2.5e-3, 2e-2), "semantic adaptor")
    check(x_hat.torch_nchw(), xh_ref, tol(2.5e-3, 3e-2), "codec synthesis (h_detail_hat)")
    flat = x_hat.torch_nchw().reshape(-1).cpu()
    gi = torch.from_numpy(g["x_hat.idx"])
    assert rel_rms(flat[gi], torch.from_numpy(g["x_hat.val"])) <= tol(2.5e-3, 3e-2)       # vs the reference-made golden


@pytest.fixture(scope="module")
def unet_pair():
    sd = synth.unet_state_dict(0)
    return UNet(sd, DEV), R.fold_lora(sd)


def test_unet_step(unet_pair):
    """One UNet step at 16x24 latents (128x192 picture) vs the oracle: eps and reduced.
    tol: 3e-2 rel. RMS over ~60 bf16 layers."""
    unet, sd = unet_pair
    g = torch.Generator().manual_seed(7)
    sample = torch.randn(2, 320, 16, 24, generator=g).to(torch.bfloat16).float()
    ctx = torch.randn(2, 768, 2, 3, generator=g).to(torch.bfloat16).float()
    eps, reduced = unet(from_nchw(sample.to(DEV)), 999, from_nchw(ctx.to(DEV)))
    with torch.no_grad():
        eps_ref, red_ref = R.unet_forward(sd, sample, torch.tensor([999, 999]), ctx.permute(0, 2, 3, 1).reshape(2, 6, 768))
    check(reduced.torch_nchw(), red_ref, tol(1e-3, 1e-2), "vae_reduction")
    check(eps.torch_nchw(), eps_ref, tol(4e-3, 3e-2), "UNet eps")


@pytest.mark.parametrize("h,w", [(32, 48), (24, 40)])
def test_vae_decoder_windowed(h, w):
    """VAE decoder vs the oracle on a 32x48 latent (six 16x16 attention windows) and on a 24x40
    latent whose right/bottom windows are ragged (16x8, 8x16, 8x8), as the reference's slicing
    produces for pictures that are not a multiple of 128. tol 3e-2."""
    sd = synth.vae_state_dict(0)
    vae = VAEDecoder(sd, DEV, attn_patch=16)
    lat = torch.randn(1, 4, h, w, generator=torch.Generator().manual_seed(8)) * 0.18215
    lv = View(lat.permute(0, 2, 3, 1).contiguous().to(DEV).reshape(-1), 1, h, w, 4)
    with torch.no_grad():
        ref = R.vae_decode(sd, lat, 16)
    # Two screens: the f16-OPERAND kernels keep their tight bound (f16 build: 1.7e-3 ... 3e-3 measured, bound
    # 4e-3 as in rounds 2-4); the default since round 5 -- the decoder's resnet convs multiply bf16 OPERANDS under f16 storage -- reads
    # 5.2e-3 at the random-init point of this small test and gets 8e-3 (the budget that counts, |dPSNR| <= 0.01 dB at full size, moved
    # by 0.0003 dB: DESIGN.md section 5).  The switch is a run-time one: both weight packings exist in the layers.
    from onedc_amd import kernels as Kmod
    keep = Kmod.BF16_OPERANDS
    try:
        for on, bound in ((False, 4e-3), (True, 8e-3)):
            if on and not keep:
                continue                                    # (ONEDC_BF16_OPERANDS=0 in the environment: no bf16 packing was made)
            Kmod.BF16_OPERANDS = on
            img = vae.decode(lv)
            check(img.torch_nchw(), ref, tol(bound, 3e-2), f"VAE decoder {h}x{w}, {'bf16' if on else 'f16'} operands")
    finally:
        Kmod.BF16_OPERANDS = keep


def test_end_to_end_decode_psnr(full_model, gaussian_tables):
    """Whole decode of two 128x192 streams (made by the product's own encoder side) against the
    oracle run from the same decoded latents.  Reported: PSNR(product, oracle) on [0,1] pictures
    (reference metric, test_quality.py:234-235).  tol: >= 60 dB with f16 activations (measured 67), >= 45 dB with bf16
    (measured 49-50; bf16 UNet + x14.65 eps->x0 gain)."""
    model = full_model
    rng = np.random.default_rng(9)
    idx = torch.from_numpy(rng.integers(0, 16384, (2, 2, 3)).astype(np.int32)).to(DEV)
    blobs, y_hat = model.codec_model.encode_latents(idx, None, noise_seed=1)
    img = model.decode(stream=blobs, unit_range=True)
    assert img.shape == (2, 3, 128, 192)
    # oracle from the same y_hat / z indices (the entropy path is integer-exact, checked above)
    codec_sd, unet_sd, vae_sd = synth.codec_state_dict(0), R.fold_lora(synth.unet_state_dict(0)), synth.vae_state_dict(0)
    with torch.no_grad():
        _, z_sem = R.z_indices_to_params(codec_sd, idx.cpu())
        out = R.decode_from_latents(codec_sd, unet_sd, vae_sd, y_hat.torch_nchw().cpu(), z_sem)
    ref = R.to_unit_range(out["image"])
    p = R.psnr(img.cpu(), ref)
    print(f"[parity] end-to-end PSNR(product, oracle) = {p:.2f} dB")
    assert p >= tol(62.0, 45.0)


def test_decode_pipeline_matches_sequential_decode(full_model):
    """DecodePipeline (entropy stage of batch n+1 on its own stream/thread while batch n is in the
    UNet/VAE) must return bit-identical pictures to back-to-back decode() calls."""
    model = full_model
    rng = np.random.default_rng(19)
    batches = []
    for k in range(3):
        idx = torch.from_numpy(rng.integers(0, 16384, (2, 2, 3)).astype(np.int32)).to(DEV)
        blobs, _ = model.codec_model.encode_latents(idx, None, noise_seed=20 + k)
        batches.append(blobs)
    seq = [model.decode(stream=b) for b in batches]
    pipe = model.pipeline()                 # two stage-S streams; an odd last batch is split over them
    try:
        got = pipe.map(batches)
        got2 = pipe.map(batches[::-1])
        got3 = pipe.map(batches[:2])        # even number of batches: no split
    finally:
        pipe.close()
    pipe1 = model.pipeline(s_streams=1)
    try:
        got4 = pipe1.map(batches)
    finally:
        pipe1.close()
    torch.cuda.synchronize()
    for a, b in zip(seq, got):
        assert torch.equal(a, b)
    for a, b in zip(seq[::-1], got2):
        assert torch.equal(a, b)
    for a, b in zip(seq[:2], got3):
        assert torch.equal(a, b)
    for a, b in zip(seq, got4):
        assert torch.equal(a, b)


# ---- encode side -----------------------------------------------------------------------------
def test_hyper_encoder_fsq_and_compress_round_trip(codec, codec_sd):
    """HyperEncoder on HIP vs the oracle (bf16 nets: 3e-2 rel. RMS), the FSQ quantisation kernel vs the
    oracle's formula on identical z (integer result: equal except for values within float rounding of a
    cell edge, <= 1e-4 of the entries), and compress() -> decode(): the decoder recovers the encoder's
    indices and y_hat bit for bit."""
    g = golden("g7_hyper_enc.npz")
    henc_sd = synth.hyper_enc_state_dict(0)
    codec.load_hyper_encoder(henc_sd)
    y, sem = torch.from_numpy(g["y"]), torch.from_numpy(g["sem"])
    yv, sv = from_nchw(y.to(DEV)), from_nchw(sem.to(DEV))
    z = codec.hyper_enc(yv, sv, codec.ws)
    with torch.no_grad():
        z_ref = R.hyper_encoder(henc_sd, y.to(torch.bfloat16).float(), sem.to(torch.bfloat16).float())
    check(z.slice(0, 7).torch_nchw(), z_ref, tol(4e-3, 3e-2), "hyper encoder z vs oracle")
    check(z.slice(0, 7).torch_nchw(), torch.from_numpy(g["z"]), tol(3e-3, 3e-2), "hyper encoder z vs reference golden")
    # FSQ kernel on a dense sweep of z values
    zz = (torch.rand(4, 7, 16, 16, generator=torch.Generator().manual_seed(31)) * 6 - 3)
    zzv = from_nchw(zz.to(DEV), torch.float32, pad_to=8)
    idx = torch.empty(4 * 16 * 16, dtype=torch.int32, device=DEV)
    K.fsq_quantize(zzv.slice(0, 7), idx)
    _, idx_ref = R.fsq_quantize(zz)
    bad = int((idx.cpu().view(4, 16, 16) != idx_ref).sum())
    assert bad <= 1, bad
    # compress -> decode
    yn = (2.0 * torch.randn(2, 8, 12, 128, generator=torch.Generator().manual_seed(32))).to(DEV)
    sn = sem.permute(0, 2, 3, 1).contiguous().to(DEV)
    blobs, indices, y_hat_enc = codec.compress(yn, sn, heights=[128, 120], widths=[192, 180])
    y_hat_dec, z_sem, orig_hw, pad_hw, _ = codec.decode_entropy(stream=blobs)
    assert orig_hw == [(128, 192), (120, 180)] and pad_hw == (128, 192)
    assert torch.equal(y_hat_dec.torch_nchw(), y_hat_enc.torch_nchw())
    from onedc_amd.entropy.utils import decode_i, unpack_indices
    got = np.stack([unpack_indices(decode_i(b, 14, 64)["bit_stream_z"], 6, 14) for b in blobs]).reshape(2, 2, 3)
    assert np.array_equal(got, indices.cpu().numpy())
    assert len(set(indices.cpu().numpy().reshape(-1).tolist())) > 1         # the synthetic z really spreads


def test_vae_encoder_vs_oracle():
    """SD-VAE encoder (windowed mid attention, one-sided stride-2 padding, latent sample with injected noise)
    vs the oracle on a 128x192 picture (16x24 latent: one full 16x16 window + a ragged 16x8 one). tol 3e-2."""
    from onedc_amd.encoder import VAEEncoder
    sd = synth.vae_encoder_state_dict(0)
    enc = VAEEncoder(sd, DEV, attn_patch=16)
    g = torch.Generator().manual_seed(41)
    img = torch.rand(2, 3, 128, 192, generator=g) * 2 - 1
    noise = torch.randn(2, 4, 16, 24, generator=g)
    iv = View(torch.zeros((2 * 128 * 192 + K.SLACK_ROWS) * 32, dtype=torch.float32, device=DEV), 2, 128, 192, 32)
    K.nchw_to_nhwc(img.to(DEV), iv.slice(0, 3))
    z = enc.encode(iv, noise.permute(0, 2, 3, 1).contiguous().to(DEV))
    with torch.no_grad():
        ref = R.vae_encode(sd, img, 16, noise)
    check(z.torch_nchw(), ref, tol(2e-3, 3e-2), "VAE encoder latents")


def test_flash_attention_head_dim_8():
    """The analysis UNet's attention: 64 heads of 8 channels (DQ 32 / DV 16 padding)."""
    from onedc_amd import pack
    B, Hh, d, N = 2, 64, 8, 144
    DQ, DV = pack.HEAD_PAD[d]
    g = torch.Generator().manual_seed(42)
    q, k, v = (torch.randn(B, Hh, N, d, generator=g).to(torch.bfloat16).float() for _ in range(3))
    n_pad = 192

    def padded(t, dp):
        out = torch.zeros(B, N, Hh, dp)
        out[..., :d] = t.permute(0, 2, 1, 3)
        return out.reshape(B, N, Hh * dp).to(K.a16()).to(DEV).contiguous()

    vt = torch.zeros(B, Hh, DV, n_pad)
    vt[:, :, :d, :N] = v.transpose(2, 3)
    o = torch.empty(B, N, Hh * DV, dtype=K.a16(), device=DEV)
    K.flash_attn(padded(q, DQ), padded(k, DQ), vt.reshape(B, Hh * DV, n_pad).to(K.flash_vt_dtype(DQ, DV, N)).to(DEV).contiguous(),
                 o, B, Hh, N, N, n_pad, DQ, DV, d ** -0.5)
    ref = torch.softmax(q @ k.transpose(-1, -2) * d ** -0.5, -1) @ v
    got = o.float().cpu().view(B, N, Hh, DV)[..., :d].permute(0, 2, 1, 3)
    check(got, ref, tol(1e-3, 8e-3), "flash attention d=8")


def test_analysis_transform_vs_oracle():
    """Encoder (pix_emb 8x8/8 as patchify + GEMM, pix_fusion, the patched UNet2DModel, trans_coding) vs the
    oracle on a 128x192 picture: y and sem. tol 5e-2 rel. RMS (~80 bf16 layers, two up/down scales)."""
    from onedc_amd.encoder import CodecEncoder
    sd = synth.codec_encoder_state_dict(0)
    enc = CodecEncoder(sd, DEV)
    g = torch.Generator().manual_seed(43)
    img = torch.rand(1, 3, 128, 192, generator=g) * 2 - 1
    cond = torch.randn(1, 4, 16, 24, generator=g) * 0.2
    iv = View(torch.zeros((128 * 192 + K.SLACK_ROWS) * 32, dtype=torch.float32, device=DEV), 1, 128, 192, 32)
    K.nchw_to_nhwc(img.to(DEV), iv.slice(0, 3))
    cv = from_nchw(cond.to(DEV), torch.float32)
    y, sem = enc(iv, cv)
    with torch.no_grad():
        y_ref, sem_ref = R.codec_encoder(sd, img, cond)
    check(sem.torch_nchw(), sem_ref, tol(4e-3, 5e-2), "analysis transform: sem (UNet mid output)")
    check(y.torch_nchw(), y_ref, tol(4e-3, 5e-2), "analysis transform: y")


def test_model_encode_then_decode_round_trip(full_model, tmp_path):
    """model.encode(picture) -> container bytes -> model.decode: the whole codec on HIP.  Checks the container
    (sizes, replicate padding to 64, bpp bookkeeping), that decode reproduces the encoder's latents bit-exactly,
    and that a picture alone encodes to the same bytes as inside a batch."""
    model = full_model
    g = torch.Generator().manual_seed(44)
    img = (torch.rand(2, 3, 120, 180, generator=g) * 2 - 1).to(DEV)          # padded to 128x192
    noise = torch.randn(2, 16, 24, 4, generator=g).to(DEV)
    blobs, bpp = model.encode(img, noise=noise)
    assert len(blobs) == 2 and bpp[0]["bits_total"] == len(blobs[0]) * 8
    assert abs(bpp[0]["bpp"] - len(blobs[0]) * 8 / (120 * 180)) < 1e-12 and bpp[0]["bits_z"] == 88   # 6 indices x 14 bit
    one, bpp1 = model.encode(img[1:2], noise=noise[1:2], fp=str(tmp_path / "a.bin"))
    assert one == blobs[1] and (tmp_path / "a.bin").read_bytes() == one
    out = model.decode(stream=blobs)
    assert out.shape == (2, 3, 120, 180) and bool(torch.isfinite(out).all())
    out1 = model.decode(fp=str(tmp_path / "a.bin"))
    assert torch.equal(out1[0], out[1])


def test_cli_harness_encode_decode_and_decoder_only(tmp_path):
    """The reference's command-line surface end to end on synthetic checkpoint FILES: strict loader of
    model.safetensors / model_1.safetensors / the VAE file (decode + encode side), evaluate loop over a picture
    folder (stream/, recon/, the two CSVs with the reference's bpp keys), then --decoder_only on a written stream."""
    import pandas as pd
    import yaml
    from PIL import Image
    from safetensors.torch import save_file
    from onedc_amd import inference
    ckpt = tmp_path / "ckpt"
    ckpt.mkdir()
    save_file(synth.unet_state_dict(0), str(ckpt / "model.safetensors"))
    codec_full = dict(synth.codec_state_dict(0))
    codec_full.update(synth.codec_encoder_state_dict(0))
    codec_full.update(synth.hyper_enc_state_dict(0))
    save_file(codec_full, str(ckpt / "model_1.safetensors"))
    vae_full = dict(synth.vae_state_dict(0))
    vae_full.update(synth.vae_encoder_state_dict(0))
    save_file(vae_full, str(ckpt / "vae.safetensors"))
    (tmp_path / "cfg.yaml").write_text(yaml.safe_dump({"vae_attn_patch": 16, "conditioning_timestep": 999}))
    pics = tmp_path / "pics"
    pics.mkdir()
    rng = np.random.default_rng(51)
    for name, (h, w) in (("a", (96, 150)), ("b", (128, 192))):
        Image.fromarray(rng.integers(0, 256, (h, w, 3), dtype=np.uint8)).save(pics / f"{name}.png")
    out = tmp_path / "out"
    avg = inference.main(["--config_path", str(tmp_path / "cfg.yaml"), "--checkpoint_path", str(ckpt),
                          "--vae_path", str(ckpt / "vae.safetensors"), "--output_path", str(out),
                          "--eval_image_path", str(pics)])
    for name, (h, w) in (("a", (96, 150)), ("b", (128, 192))):
        assert (out / "stream" / f"{name}.bin").stat().st_size > 16
        assert Image.open(out / "recon" / f"{name}.png").size == (w, h)
    detail = pd.read_csv(out / "detail_bpp_caption.xlsx")
    assert set(["bits_y", "bits_z", "bits_caption", "bits_total", "bpp_y", "bpp_z", "bpp_caption", "bpp", "bpp_file",
                "image_name"]) <= set(detail.columns) and len(detail) == 2
    assert abs(detail["bpp_file"][0] - (out / "stream" / "a.bin").stat().st_size * 8 / (96 * 150)) < 1e-9
    # --decoder_only reproduces the picture the evaluate loop wrote
    inference.main(["--config_path", str(tmp_path / "cfg.yaml"), "--checkpoint_path", str(ckpt),
                    "--vae_path", str(ckpt / "vae.safetensors"), "--decoder_only",
                    "--decoder_bin_path", str(out / "stream" / "b.bin"), "--decoder_out_path", str(tmp_path / "b2.png")])
    assert np.array_equal(np.asarray(Image.open(tmp_path / "b2.png")), np.asarray(Image.open(out / "recon" / "b.png")))
    # a DIRECTORY of streams goes through the overlapped pipeline (model.pipeline().map(..., out="u8", to_host=True), PNG
    # writing on a thread pool): the same pictures, bit for bit, as the one-by-one decodes of the evaluate loop
    inference.main(["--config_path", str(tmp_path / "cfg.yaml"), "--checkpoint_path", str(ckpt),
                    "--vae_path", str(ckpt / "vae.safetensors"), "--decoder_only",
                    "--decoder_bin_path", str(out / "stream"), "--decoder_out_path", str(tmp_path / "dec")])
    for name in ("a", "b"):
        assert np.array_equal(np.asarray(Image.open(tmp_path / "dec" / f"{name}.png")),
                              np.asarray(Image.open(out / "recon" / f"{name}.png"))), name
    # tools/first_contact.py on the same checkpoint files: strict contract, weight magnitudes, one picture in f16 and bf16
    # with the stored abs-max of every layer, PSNR between the two, a recommendation (what one runs the day real weights exist)
    import importlib.util
    import json
    spec = importlib.util.spec_from_file_location("first_contact", os.path.join(os.path.dirname(os.path.dirname(__file__)), "tools", "first_contact.py"))
    fc = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(fc)
    rc = fc.main(["--checkpoint_path", str(ckpt), "--vae_path", str(ckpt / "vae.safetensors"), "--config_path", str(tmp_path / "cfg.yaml"),
                  "--bin", str(out / "stream" / "b.bin"), "--out", str(tmp_path / "report.json")])
    rep = json.loads((tmp_path / "report.json").read_text())
    assert rc == 0 and rep["strict_load"] == "ok" and rep["lora_tensors_present"] and rep["weights"]["vae"]["fits_half"]
    assert rep["decode"]["f16"]["conv_launches"] > 300 and not rep["decode"]["f16"]["nonfinite_pictures"]
    assert rep["decode"]["f16"]["stages"]["x0"]["finite"] and rep["psnr_f16_vs_bf16_db"] > 35.0
    assert rep["recommendation"]["act_dtype"] in ("f16", "bf16") and rep["stream"]["bytes"] == (out / "stream" / "b.bin").stat().st_size
    # round 6, step 5: the bf16-operand switch measured on THESE weights (same stream, switch on / off), and a recommendation from it
    bo = rep["bf16_operands"]
    assert "error" not in bo, bo
    assert bo["psnr_between_db"] > 45.0 and set(bo["dpsnr_uncorrelated_db"]) == {"25", "27", "30", "35"}
    assert bo["dpsnr_uncorrelated_db"]["25"] < bo["dpsnr_uncorrelated_db"]["35"]
    assert bo["recommendation"].startswith("ONEDC_BF16_OPERANDS=")
    assert fc.dpsnr_uncorrelated(63.57, 27.0) == pytest.approx(0.00095, rel=0.05)           # (63.57 dB -> 0.0010 dB at 27 dB)
    # a file that breaks the contract is reported, not loaded
    bad = dict(codec_full)
    bad.pop("dec.conv_out.weight", None) if "dec.conv_out.weight" in bad else bad.pop(sorted(bad)[0])
    save_file(bad, str(ckpt / "model_1.safetensors"))
    assert fc.main(["--checkpoint_path", str(ckpt), "--vae_path", str(ckpt / "vae.safetensors"), "--cpu-only"]) == 1


def test_z_only_harness_forward(full_model):
    """Ex-low-rate harness: reflect padding, indices from the analysis side, picture from the indices alone; the
    252-byte payload at 768x768 (readme: "0.24 KB") = 144 indices x 14 bit, here 6 indices -> 11 bytes."""
    from onedc_amd.entropy.utils import decode_i, unpack_indices
    model = full_model
    img = (torch.rand(1, 3, 100, 170, generator=torch.Generator().manual_seed(61)) * 2 - 1).to(DEV)
    enc, recon = model.forward_z_only(img)
    assert recon.shape == (1, 3, 100, 170) and bool(torch.isfinite(recon).all())
    d = decode_i(enc["stream"][0], 14, 64)
    assert (d["height"], d["width"], d["pad_height"], d["pad_width"]) == (100, 170, 128, 192)
    assert len(d["bit_stream_y"]) == 0 and len(d["bit_stream_z"]) == 11
    assert np.array_equal(unpack_indices(d["bit_stream_z"], 6, 14).reshape(2, 3), enc["indices"][0].cpu().numpy())
    assert abs(enc["bpp_z"] - 6 * 14 / (100 * 170)) < 1e-12
    # the same indices through the decode-only entry give the same picture
    again = model.decode_z_only(enc["indices"])[:, :, :100, :170]
    assert torch.equal(again, recon)
    # reflect padding really is a reflection
    iv = View(torch.zeros((128 * 192 + K.SLACK_ROWS) * 32, dtype=torch.float32, device=DEV), 1, 128, 192, 32)
    K.nchw_to_nhwc(img, iv.slice(0, 3), reflect=True)
    ref = torch.nn.functional.pad(img.cpu(), (0, 22, 0, 28), mode="reflect")
    assert torch.equal(iv.slice(0, 3).torch_nchw().cpu(), ref)


def test_z_only_cli_harness_on_a_folder(tmp_path):
    """`python -m onedc_amd.inference_z_only` (reference: models/sd15_onedc_codec_z_only/inference.py:93-161) on
    synthetic checkpoint FILES and a picture folder: recon/, the two tables with the reference's columns, the
    indices-only .bin (16 B + ceil(14 n / 8) B), and --decoder_only on those files reproduces the pictures."""
    import pandas as pd
    import yaml
    from PIL import Image
    from safetensors.torch import save_file
    from onedc_amd import inference_z_only
    ckpt = tmp_path / "ckpt"
    ckpt.mkdir()
    save_file(synth.unet_state_dict(0), str(ckpt / "model.safetensors"))
    codec_full = dict(synth.codec_state_dict(0))
    codec_full.update(synth.codec_encoder_state_dict(0))
    codec_full.update(synth.hyper_enc_state_dict(0))
    save_file(codec_full, str(ckpt / "model_1.safetensors"))
    vae_full = dict(synth.vae_state_dict(0))
    vae_full.update(synth.vae_encoder_state_dict(0))
    save_file(vae_full, str(ckpt / "vae.safetensors"))
    (tmp_path / "cfg.yaml").write_text(yaml.safe_dump({"vae_attn_patch": 16, "conditioning_timestep": 999}))
    pics = tmp_path / "pics"
    pics.mkdir()
    rng = np.random.default_rng(71)
    sizes = {"a": (100, 170), "b": (128, 192), "c": (128, 192)}
    for name, (h, w) in sizes.items():
        Image.fromarray(rng.integers(0, 256, (h, w, 3), dtype=np.uint8)).save(pics / f"{name}.png")
    out = tmp_path / "out"
    common = ["--config_path", str(tmp_path / "cfg.yaml"), "--checkpoint_path", str(ckpt), "--vae_path", str(ckpt / "vae.safetensors")]
    avg = inference_z_only.main(common + ["--output_path", str(out), "--eval_image_path", str(pics)])
    detail = pd.read_csv(out / "detail_bpp_caption.xlsx")
    assert list(detail["image_name"]) == ["a", "b", "c"] and {"bpp_hard_y", "bpp_z"} <= set(detail.columns)
    assert abs(detail["bpp_z"][0] - 6 * 14 / (100 * 170)) < 1e-12 and abs(detail["bpp_z"][1] - 6 * 14 / (128 * 192)) < 1e-12
    assert float(detail["bpp_hard_y"].abs().max()) == 0.0 and abs(avg["bpp_z"] - detail["bpp_z"].mean()) < 1e-12
    for name, (h, w) in sizes.items():
        assert Image.open(out / "recon" / f"{name}.png").size == (w, h)
        assert (out / "stream" / f"{name}.bin").stat().st_size == 16 + 11          # 6 indices x 14 bit -> 11 bytes
    # decoder only: one file, then the folder (b and c share a size and go through one batch)
    inference_z_only.main(common + ["--decoder_only", "--decoder_bin_path", str(out / "stream" / "a.bin"),
                                    "--decoder_out_path", str(tmp_path / "a2.png")])
    assert np.array_equal(np.asarray(Image.open(tmp_path / "a2.png")), np.asarray(Image.open(out / "recon" / "a.png")))
    inference_z_only.main(common + ["--decoder_only", "--decoder_bin_path", str(out / "stream"),
                                    "--decoder_out_path", str(tmp_path / "dec")])
    for name in sizes:
        assert np.array_equal(np.asarray(Image.open(tmp_path / "dec" / f"{name}.png")),
                              np.asarray(Image.open(out / "recon" / f"{name}.png")))


def test_encode_decode_large_picture(full_model):
    """A picture above 0.65 MP: the analysis transform's single-head attention runs over (H/16)(W/16) = 4096 tokens at
    1024x1024, past the 2560 columns the register-resident row softmax holds (onedc_softmax_rows now streams longer rows);
    the reference's folder loop handles any size (inference.py:109-127).  Round trip: the decoder recovers the encoder's
    latent bit for bit and returns a finite picture of the original size."""
    model = full_model
    g = torch.Generator().manual_seed(91)
    img = (torch.rand(1, 3, 1000, 1024, generator=g) * 2 - 1).to(DEV)        # padded to 1024 x 1024
    noise = torch.randn(1, 128, 128, 4, generator=g).to(DEV)
    blob, bpp = model.encode(img, noise=noise)
    assert bpp["bits_z"] == ((16 * 16 * 14 + 7) // 8) * 8 and bpp["bits_total"] == len(blob) * 8
    out = model.decode(stream=[blob])
    assert out.shape == (1, 3, 1000, 1024) and bool(torch.isfinite(out).all())
    again, _ = model.encode(img, noise=noise)
    assert again == blob                                                      # deterministic


def test_softmax_rows_long_rows_match_torch():
    """onedc_softmax_rows beyond the register-resident width (n_pad > 2560): streaming kernel vs torch.softmax."""
    rows, n = 37, 4099
    n_pad = (n + 63) // 64 * 64
    s = torch.randn(rows, n, generator=torch.Generator().manual_seed(92)) * 4
    p = torch.empty(rows * n_pad, dtype=K.a16(), device=DEV)
    K.softmax_rows(s.to(DEV).contiguous(), p, rows, n, n_pad)
    got = p.view(rows, n_pad).float().cpu()
    assert float(got[:, n:].abs().max()) == 0.0
    ref = torch.softmax(s.double(), -1)
    t = tol(2e-3, 1e-2)                                        # output rounding: 11 / 8 significant bits
    assert float((got[:, :n].double() - ref).abs().max()) <= t * float(ref.max())
    assert float((got[:, :n].double().sum(-1) - 1).abs().max()) <= t


def test_compact_attention_heads_do_not_change_a_decoded_bit():
    """Round 5: the UNet's d = 40 / 80 attention layers keep q / k / V^T / o compact (no padded head columns in HBM) wherever their
    kernel sees 64 keys and more.  A 512x512 picture (64 context tokens; self-attention over 4 096 / 1 024 tokens) decoded by a model
    built with the switch off and by one built with it on: the same picture, bit for bit -- and both paths were actually taken."""
    from onedc_amd import layers as L
    idx = torch.from_numpy(np.random.default_rng(31).integers(0, 16384, (1, 8, 8)).astype(np.int32)).to(DEV)
    pics = []
    # (round 6: the fused cross-attention block only exists for compact heads and has its own summation order -- this test is about
    # the LAYOUT, so both models run the four separate launches; the fused block has test_fused_cross_attention_block_in_a_decode)
    keep_x, K.USE_XATTN_BLOCK = K.USE_XATTN_BLOCK, False
    for flag in (False, True):
        before, L.COMPACT_HEADS = L.COMPACT_HEADS, flag
        try:
            m = SD15_1step_codec_stage1({"vae_attn_patch": 16}).load_synthetic(0, preset="trained_like")
        finally:
            L.COMPACT_HEADS = before
        m.codec_model.update(force=True)
        t = m.feedforward_model._transformers
        assert all(x.attn1.compact == flag and x.attn2.compact == flag for x in t if x.attn1.d in (40, 80))
        assert (m.feedforward_model.ctx_kv.compact is not None) == flag
        if not pics:
            blobs, _ = m.codec_model.encode_latents(idx, None, noise_seed=32)
        try:
            pics.append(m.decode(stream=blobs))
        except BaseException:
            K.USE_XATTN_BLOCK = keep_x
            raise
        del m
        torch.cuda.empty_cache()
    K.USE_XATTN_BLOCK = keep_x
    assert torch.equal(pics[0], pics[1])


def test_fused_cross_attention_block_in_a_decode():
    """Round 6: norm2 -> attn2 of the UNet's first-level transformer blocks as one launch (csrc/xattn_block.hip).  One 512x512 picture
    (64 context tokens, 4 096 pixels per picture at the first level) decoded with the fused launches and with the four separate ones:
    the fused path is actually taken (five blocks x two batches... counted through the kernel wrapper), the two pictures agree to
    >= 66 dB PSNR on [0, 1] (measured ~75: both round q, o and the block output to 16 bits once; other summation order and the plain
    row maximum in the softmax), i.e. far inside the 0.01 dB budget the full-size tests hold against the oracle with the fused path on."""
    from onedc_amd.quality import psnr
    idx = torch.from_numpy(np.random.default_rng(33).integers(0, 16384, (1, 8, 8)).astype(np.int32)).to(DEV)
    m = SD15_1step_codec_stage1({"vae_attn_patch": 16}).load_synthetic(0, preset="trained_like")
    m.codec_model.update(force=True)
    blobs, _ = m.codec_model.encode_latents(idx, None, noise_seed=34)
    calls, orig = [], K.xattn_block

    def counted(p):
        calls.append(p.Nk)
        return orig(p)
    keep = K.USE_XATTN_BLOCK
    try:
        K.xattn_block = counted
        K.USE_XATTN_BLOCK = True
        fused = m.decode(stream=blobs, unit_range=True)
        n_fused = len(calls)
        K.USE_XATTN_BLOCK = False
        plain = m.decode(stream=blobs, unit_range=True)
    finally:
        K.xattn_block, K.USE_XATTN_BLOCK = orig, keep
    assert n_fused == 5 and len(calls) == 5 and set(calls) == {64}     # the five transformer blocks of the 64x64 level, only with the switch on
    assert not torch.equal(fused, plain)
    db = float(psnr(fused.clamp(0, 1).cpu(), plain.clamp(0, 1).cpu()))
    print(f"PSNR(fused, separate launches) = {db:.2f} dB")
    assert db >= 66.0, db
