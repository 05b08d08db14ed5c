"""The output hand-off of the decode path and the half-precision range guard.

  * onedc_image_out / onedc_image_out_u8 against torch: crop, clamp(-1,1)*0.5+0.5 (reference src/inference.py:106), and
    torchvision.utils.save_image's 8-bit quantisation (x255 + 0.5, clamp, truncation: inference.py:107,120) on the device;
  * the non-finite flag: a NaN / Inf that the clamp would turn into a valid-looking pixel is reported per picture;
  * model.decode() / DecodePipeline.map() raise kernels.HalfRangeError when an f16 activation overflowed upstream
    (forced here by scaling a packed weight), name act_dtype: bf16, and write no picture through the CLI.
"""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from onedc_amd import kernels as K
from onedc_amd.kernels import View
from onedc_amd.model_sd15_with_codec_stage1 import SD15_1step_codec_stage1

DEV = "cuda:0"


def _picture(B, PH, PW, cs, f32, seed):
    g = torch.Generator().manual_seed(seed)
    x = torch.randn(B, PH, PW, cs, generator=g) * 0.8
    t = x.to(DEV).to(torch.float32 if f32 else K.a16())
    return View(t.reshape(-1).contiguous(), B, PH, PW, 3, cs, 0), t.float().cpu()


@pytest.mark.parametrize("f32,cs", [(True, 3), (False, 8)])
def test_image_out_u8_matches_save_image_quantisation(f32, cs):
    B, PH, PW, H, W = 3, 64, 80, 50, 77
    v, x = _picture(B, PH, PW, cs, f32, 5)
    ref01 = x[:, :H, :W, :3].clamp(-1, 1) * 0.5 + 0.5
    ref_u8 = ref01.mul(255).add_(0.5).clamp_(0, 255).to(torch.uint8)                # torchvision save_image
    flags = torch.zeros(B, dtype=torch.int32, device=DEV)
    hwc = torch.empty(B, H, W, 3, dtype=torch.uint8, device=DEV)
    K.image_out_u8(v, hwc, H, W, True, flags)
    chw = torch.empty(B, 3, H, W, dtype=torch.uint8, device=DEV)
    K.image_out_u8(v, chw, H, W, False, flags)
    f = torch.empty(B, 3, H, W, dtype=torch.float32, device=DEV)
    K.image_out(v, f, H, W, True, flags)
    torch.cuda.synchronize()
    assert torch.equal(hwc.cpu(), ref_u8)
    assert torch.equal(chw.cpu(), ref_u8.permute(0, 3, 1, 2))
    assert torch.equal(f.cpu(), ref01.permute(0, 3, 1, 2))
    assert flags.cpu().tolist() == [0] * B


def test_nonfinite_values_are_flagged_per_picture_not_clamped_away():
    B, PH, PW = 4, 32, 32
    v, x = _picture(B, PH, PW, 3, True, 6)
    flat = v.t.view(B, PH, PW, 3)
    flat[1, 3, 4, 2] = float("nan")
    flat[3, 0, 0, 0] = float("inf")
    flat[2, 31, 31, 1] = float("-inf")           # outside the 20x20 crop: not part of the picture
    for u8 in (False, True):
        flags = torch.zeros(B, dtype=torch.int32, device=DEV)
        if u8:
            K.image_out_u8(v, torch.empty(B, 20, 20, 3, dtype=torch.uint8, device=DEV), 20, 20, True, flags)
        else:
            out = torch.empty(B, 3, 20, 20, dtype=torch.float32, device=DEV)
            K.image_out(v, out, 20, 20, True, flags)
            assert bool(torch.isfinite(out).all())          # the clamp hides them: this is why the flag exists
        assert flags.cpu().tolist() == [0, 1, 0, 1]


@pytest.fixture(scope="module")
def model():
    m = SD15_1step_codec_stage1({"vae_attn_patch": 16}).load_synthetic(0)
    m.codec_model.update(force=True)
    return m


def _streams(model, B, seed):
    idx = torch.from_numpy(np.random.default_rng(seed).integers(0, 16384, (B, 2, 3)).astype(np.int32)).to(DEV)
    return model.codec_model.encode_latents(idx, None, noise_seed=seed + 1)[0]


def test_decode_u8_and_host_handoff(model):
    """out="u8" equals save_image's quantisation of the f32 result; the pipeline's to_host pictures are the same bytes in
    page-locked host memory."""
    blobs = _streams(model, 4, 31)
    f = model.decode(stream=blobs, unit_range=True)
    u = model.decode(stream=blobs, out="u8")
    assert u.shape == (4, 128, 192, 3) and u.dtype == torch.uint8
    assert torch.equal(u.cpu(), f.cpu().mul(255).add_(0.5).clamp_(0, 255).to(torch.uint8).permute(0, 2, 3, 1))
    pipe = model.pipeline()
    try:
        host = pipe.map([blobs, blobs[:2], blobs], out="u8", to_host=True)
        dev = pipe.map([blobs], out="u8")
    finally:
        pipe.close()
    assert all(h.device.type == "cpu" and h.is_pinned() for h in host[:2])
    assert torch.equal(host[0], u.cpu()) and torch.equal(host[1], u.cpu()[:2]) and torch.equal(host[2], u.cpu())
    assert torch.equal(dev[0].cpu(), u.cpu())


def test_half_range_overflow_raises_with_the_fix_named(model, tmp_path):
    """Force an activation out of the IEEE-half range (the packed UNet conv_in weights x 6e4: f16 saturates to Inf, the
    GroupNorm behind it makes NaN of it) and see the API refuse to return the clamped garbage."""
    blobs = _streams(model, 2, 41)
    good = model.decode(stream=blobs, unit_range=True)
    w = model.feedforward_model.conv_in.w
    saved = w.clone()
    try:
        w.mul_(6e4)
        if model.act == "f16":
            with pytest.raises(K.HalfRangeError, match="act_dtype: bf16") as ei:
                model.decode(stream=blobs, unit_range=True)
            assert ei.value.images == [0, 1] and isinstance(ei.value, FloatingPointError)
            with pytest.raises(K.HalfRangeError):
                model.decode(stream=blobs, out="u8")
            pipe = model.pipeline()
            try:
                with pytest.raises(K.HalfRangeError):
                    pipe.map([blobs, blobs])
            finally:
                pipe.close()
            # the CLI path: nothing is written
            from onedc_amd import inference
            from onedc_amd.entropy.utils import write_to_file
            write_to_file(blobs[0], str(tmp_path / "a.bin"))
            ev = inference.Evaluator.__new__(inference.Evaluator)
            ev.args = {"decoder_only": True, "decoder_bin_path": str(tmp_path / "a.bin"),
                       "decoder_out_path": str(tmp_path / "a.png"), "batch_size": 16}
            ev.model = model
            ev.rank, ev.world = 0, 1
            with pytest.raises(K.HalfRangeError):
                ev.evaluate()
            assert not (tmp_path / "a.png").exists()
            # f16_range_fallback: the batch is decoded again by a bf16 twin of the model instead of raising
            model.range_fallback = True
            try:
                fb = model.decode(stream=blobs, unit_range=True)
                # ... and in the throughput mode too: the affected batches go to the twin, map() returns every batch
                pipe = model.pipeline()
                try:
                    fbm = pipe.map([blobs, blobs[:1]], unit_range=True)
                finally:
                    pipe.close()
                assert torch.equal(fbm[0], fb) and torch.equal(fbm[1], fb[:1])
            finally:
                model.range_fallback = False
            assert model._twin is not None and model._twin.act == "bf16" and bool(torch.isfinite(fb).all())
            from oracle import torch_ref as R
            assert R.psnr(fb.cpu(), good.cpu()) >= 44.0            # the bf16 picture of the same streams (measured 49.5 dB)
            model._twin = None
            torch.cuda.empty_cache()
            model.check_finite = False                     # opting out restores the old behaviour: a finite, wrong picture
            try:
                out = model.decode(stream=blobs, unit_range=True)
                assert bool(torch.isfinite(out).all())
            finally:
                model.check_finite = True
        else:                                              # bf16 has f32's exponent range: same weights, no overflow
            out = model.decode(stream=blobs, unit_range=True)
            assert bool(torch.isfinite(out).all())
    finally:
        w.copy_(saved)
    assert torch.equal(model.decode(stream=blobs, unit_range=True), good)


def test_quality_metrics_on_device_tensors(model):
    """onedc_amd.quality (PSNR of reference src/test_quality.py:234-235, MS-SSIM restated) on pictures that stay on the GPU --
    what an evaluation loop behind model.decode() calls -- against the same functions on the CPU."""
    import math
    from onedc_amd import quality as Q
    g = torch.Generator().manual_seed(5)
    yy, xx = torch.meshgrid(torch.linspace(0, 1, 192), torch.linspace(0, 1, 256), indexing="ij")
    ref = (0.5 + 0.3 * torch.sin(12 * xx) * torch.cos(9 * yy))[None, None].repeat(2, 3, 1, 1)
    rec = (ref + 0.03 * torch.randn(ref.shape, generator=g)).clamp(0, 1)
    p_cpu, s_cpu = Q.psnr(ref, rec), Q.ms_ssim(ref, rec)
    p_gpu, s_gpu = Q.psnr(ref.to(DEV), rec.to(DEV)), Q.ms_ssim(ref.to(DEV), rec.to(DEV))
    assert abs(p_cpu - p_gpu) < 1e-6 and abs(s_cpu - s_gpu) < 1e-5 and 0.5 < s_gpu < 1.0
    assert abs(Q.psnr(ref.to(DEV), ref.to(DEV) + 0.1) - 20.0) < 1e-4
    # straight from the decoder: the 8-bit hand-off costs what 8-bit quantisation costs (uniform error of 1/255: 58.9 dB)
    blobs = _streams(model, 2, 51)
    f = model.decode(stream=blobs, unit_range=True)
    u = model.decode(stream=blobs, out="u8").permute(0, 3, 1, 2).float() / 255.0
    p = Q.psnr(f, u)
    assert p > 10 * math.log10(12 * 255 ** 2) - 1.5, p


def test_pipeline_with_mixed_picture_sizes_is_stable(model):
    """A service sees batches of different sizes back to back: six (batch, size) combinations in shuffled order through ONE
    pipeline, four rounds, 8-bit pictures in host memory -- every picture equals the sequential decode's, neither the device
    memory held between rounds nor the workspaces' zero-once buffers grow from round to round, and closing the pipeline
    releases its streams' scratch tensors."""
    from onedc_amd.layers import Workspace
    rng = np.random.default_rng(0)
    sets = []
    for i, (B, zh, zw) in enumerate([(4, 3, 3), (2, 2, 3), (5, 2, 2), (3, 3, 2), (1, 4, 4), (4, 3, 3)]):
        idx = torch.from_numpy(rng.integers(0, 16384, (B, zh, zw)).astype(np.int32)).to(DEV)
        sets.append(model.codec_model.encode_latents(idx, None, noise_seed=60 + i)[0])
    ref = [model.decode(stream=s, out="u8").cpu() for s in sets]
    pipe = model.pipeline()
    ids = {st.cuda_stream for st in pipe.streams_s} | {pipe.stream_e.cuda_stream}
    held = []
    try:
        for rnd in range(4):
            order = rng.permutation(len(sets))
            outs = pipe.map([sets[i] for i in order], out="u8", to_host=True)
            for i, o in zip(order, outs):
                assert torch.equal(o, ref[i]), (rnd, i)
            del outs
            torch.cuda.synchronize()
            held.append(torch.cuda.memory_allocated())            # what is still held when a round is over
        per_stream = [sum(1 for k in model.feedforward_model.ws._t if len(k) == 3 and k[0] == i) for i in ids]
        assert 0 < max(per_stream) <= Workspace.MAX_ZERO_LAYOUTS, per_stream
    finally:
        pipe.close()
    print(f"[mixed sizes] device memory held after rounds 0..3: {[round(h / 1e6) for h in held]} MB")
    assert held[-1] <= held[1] + 128e6, held                      # (the workspaces reach their sizes in rounds 0 / 1)
    for ws in (model.feedforward_model.ws, model.vae_large.ws, model.codec_model.ws):
        assert not any(k[0] in ids for k in ws._t)                # a closed pipeline leaves nothing behind


def test_pipeline_survives_a_corrupt_stream(model):
    """One damaged container in the middle of a call: map() raises the decoder's own error for it (never a picture), and the
    same pipeline decodes the next call correctly -- its worker threads and streams are not left in a broken state."""
    from onedc_amd.codec_module import StreamDesyncError
    good = _streams(model, 3, 71)
    ref = model.decode(stream=good, out="u8").cpu()
    bad = list(good)
    blob = bytearray(bad[1])
    for k in range(40, 60):                       # inside the y stream (16-byte header first): flips symbols -> desynchronises
        blob[k] ^= 0x5A
    bad[1] = bytes(blob)
    pipe = model.pipeline()
    try:
        with pytest.raises(StreamDesyncError) as ei:
            pipe.map([good, bad, good], out="u8")
        assert ei.value.images == [1]
        with pytest.raises(StreamDesyncError):
            model.decode(stream=bad)
        truncated = [good[0][:len(good[0]) // 2]] + good[1:]
        with pytest.raises((StreamDesyncError, ValueError, RuntimeError)):
            pipe.map([truncated], out="u8")
        again = pipe.map([good, good], out="u8")
        assert torch.equal(again[0].cpu(), ref) and torch.equal(again[1].cpu(), ref)
    finally:
        pipe.close()
