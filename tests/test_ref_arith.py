"""The prior in the reference's CPU arithmetic (IntraNoAR(prior_arith="torch_cpu"); onedc_amd/ref_arith.py, csrc/ref_arith.hip).

torch-CPU's f32 convolutions are fixed arrangements of fused-multiply-add chains (oneDNN jit_1x1 / jit_dw / jit direct, or aten's
native path); the GPU kernels evaluate the same chains, calibrated against the local torch.  What is asserted:

  * every convolution shape of the prior nets at 768x768 has a structure that reproduces torch-CPU bit for bit (calibration),
  * the two 768x768 streams written by the REFERENCE's own modules (tests/golden/g8_ref_stream_768.npz random-init weights,
    g9_ref_stream_768_trained.npz trained-like weights; written under torch-CPU with 8 threads) decode on the GPU:
    teacher-forced table-index flips [0, 0, 0, 0], free-running decode clean, y_hat equal to the reference's sample points
    EXACTLY -- where the MFMA prior flips 91 / 6 of the 294 912 decisions and raises StreamDesyncError,
  * the other direction: a stream this package writes in that mode carries the symbols and table indexes the reference's CPU
    arithmetic (the oracle on torch-CPU) computes for the same latent -- "identical bitstreams on encode" for the prior side,
  * index thresholds == torch's build_indexes (CPU test), small pictures are refused, not mis-decoded.
"""
import numpy as np
import pytest
import torch

from conftest import golden
from onedc_amd import synth


def test_index_thresholds_equal_torch_build_indexes():
    """index(sigma) = #{thresholds <= sigma} against the reference's formula (entropy_models.py:355-362) evaluated by torch on
    the CPU: random sigmas over the table's range and the two floats around every threshold."""
    from onedc_amd.ref_arith import index_thresholds
    from oracle import torch_ref as R
    thr = index_thresholds()
    assert thr.shape == (255,) and bool((thr[1:] > thr[:-1]).all())
    g = torch.Generator().manual_seed(0)
    s = torch.exp(torch.rand(200000, generator=g) * (np.log(80.0) - np.log(0.05)) + np.log(0.05)).float()
    bits = thr.view(torch.int32)
    s = torch.cat([s, thr, (bits - 1).view(torch.float32), (bits + 1).view(torch.float32), torch.tensor([-1.0, 0.0, 1e-6, 1e-5, 64.0, 1e3])])
    want = R.build_indexes(s)
    got = torch.searchsorted(thr, s, right=True).int()
    assert torch.equal(got, want)


def test_cpu_probe_fits_torch_and_profiles_round_trip(tmp_path):
    """tools/cpu_chain_probe.py (what a WRITER'S machine runs, no GPU): every probed shape gets a structure that reproduces the
    local torch bit for bit; on an Intel host the native shapes are the ones ref_arith.intel_profile() states; the JSON it writes
    loads into the dict form Calibration takes."""
    import importlib.util
    import json
    import os
    from onedc_amd.ref_arith import Calibration, intel_profile, load_profile
    spec = importlib.util.spec_from_file_location(
        "cpu_chain_probe", os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools", "cpu_chain_probe.py"))
    probe = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(probe)
    intel = any("GenuineIntel" in line for line in open("/proc/cpuinfo"))
    prev = torch.get_num_threads()
    try:
        with __import__("tempfile").TemporaryDirectory() as tmp:
            lib = probe.build_helper(tmp)
            for threads in (8, 1):
                torch.set_num_threads(threads)
                want = intel_profile(threads)
                for C, OC, ks, H, W in [(128, 128, 1, 4, 6), (256, 128, 1, 4, 4), (128, 128, 3, 4, 6), (128, 128, 3, 8, 8),
                                        (256, 128, 1, 8, 12), (128, 128, 3, 16, 16)]:
                    got = probe.fit(lib, C, OC, ks, H, W)
                    assert got is not None, (threads, C, OC, ks, H, W)
                    known = want(C, OC, H, W, ks)
                    if intel and known is not None:
                        assert tuple(got) == tuple(known), (threads, C, OC, ks, H, W, got, known)
                    if ks == 3 and got[0] == "im2col":
                        assert tuple(got[1:]) in Calibration.im2col_candidates(9 * C)     # the GPU side tries it too
    finally:
        torch.set_num_threads(prev)
    path = tmp_path / "writer.json"
    json.dump({"host": "x", "threads": 8, "1x1": {"256,128,4,4": [256, 0], "7,128,4,4": None},
               "3x3": {"128,128,8,8": ["im2col", 4, 288, 2], "128,128,16,16": ["direct", 16, 0]}}, open(path, "w"))
    prof = load_profile(str(path))
    assert prof == {(256, 128, 4, 4): (256, 0), ("3x3", 128, 128, 8, 8): ("im2col", 4, 288, 2),
                    ("3x3", 128, 128, 16, 16): ("direct", 16, 0)}
    cal = Calibration("cpu", threads=8, profile=prof)             # a calibration that only knows the profile: saved == loaded
    assert cal.conv1x1(256, 128, 4, 4) == (256, 0) and cal.conv3x3(128, 128, 8, 8) == ("im2col", 4, 288, 2)
    assert load_profile(cal.save_profile(str(tmp_path / "again.json"))) == {(256, 128, 4, 4): (256, 0),
                                                                            ("3x3", 128, 128, 8, 8): ("im2col", 4, 288, 2)}


@pytest.fixture(scope="module")
def eight_threads():
    prev = torch.get_num_threads()
    torch.set_num_threads(8)                    # the configuration class that wrote the golden streams (2-8 threads, AVX-512)
    yield
    torch.set_num_threads(prev)


# (golden file, synthetic-weight preset, torch threads of its writer)
REF_STREAMS = [("g8_ref_stream_768.npz", None, 8), ("g9_ref_stream_768_trained.npz", "trained_like", 8),
               ("g10_ref_stream_768_trained_1thread.npz", "trained_like", 1),      # one thread: the harness's own setting
               ("g11_ref_stream_512x768_trained_4threads.npz", "trained_like", 4)]  # non-square (Kodak size), four threads


@pytest.mark.gpu
@pytest.mark.parametrize("name,preset,threads", REF_STREAMS)
def test_reference_made_768_streams_decode_in_the_reference_arithmetic(eight_threads, name, preset, threads):
    from onedc_amd import kernels as K
    from onedc_amd.codec_module import IntraNoAR, StreamDesyncError
    DEV = "cuda:0"
    g = golden(name)
    # the one-thread stream was written on an Intel host: torch's native gemm path (MKL) blocks K by the CPU it runs on -- the
    # writer's structure is given as a profile there; the oneDNN configurations are the same on Intel and AMD hosts: calibrated
    from onedc_amd.ref_arith import native_gemm_profile
    profile = native_gemm_profile() if threads == 1 else None
    codec = IntraNoAR(synth.codec_state_dict(0, preset=preset), DEV, prior_arith="torch_cpu", ref_threads=threads,
                      ref_profile=profile)
    codec.update(force=True)
    idx = torch.from_numpy(g["indices"]).to(DEV)
    # teacher-forced: the reference's symbols are fed back, every table-index decision is compared
    fused, _ = codec._params_from_indices(idx)
    common_cat, y_hat = codec._prior_setup(fused)
    B, H, W, N = fused.B, fused.H, fused.W, codec.N
    n = (N // 4) * H * W
    idx_dev = torch.empty(B * n, dtype=torch.int16, device=DEV)
    flips, params = [], fused
    for k in range(4):
        if k > 0:
            params = codec._next_params(k, y_hat, common_cat)
        codec._prior_indexes(params, idx_dev, B, H, W, k)
        flips.append(int((idx_dev.view(B, N // 4, H, W).cpu().numpy() != g["index_planes"][k]).sum()))
        K.prior_update_dec(params, torch.from_numpy(np.ascontiguousarray(g["symbols"][k])).to(DEV).reshape(-1), y_hat, k)
    print(f"[ref-arith] {name} (writer: {threads} thread(s)): teacher-forced table-index flips per hop {flips}; calibrated structures: "
          f"{dict(sorted(codec.ref_prior.cal.table.items(), key=str))}")
    assert flips == [0, 0, 0, 0]
    flat = y_hat.torch_nchw().cpu().reshape(-1)
    assert torch.equal(flat[torch.from_numpy(g["y_hat.idx"])], torch.from_numpy(g["y_hat.val"]))      # bit for bit
    # free-running: the reference's stream through the whole entropy decode
    trace = []
    y2, _ = codec.decompress([g["stream"].tobytes()], idx, trace=trace)
    assert all(np.array_equal(sy, g["symbols"][k]) for k, (ix, sy) in enumerate(trace))
    assert torch.equal(y2.torch_nchw(), y_hat.torch_nchw())
    # in a batch (the pixel groups of the chain kernel run across picture borders): the same bits per picture
    yb, _ = codec.decompress([g["stream"].tobytes()] * 3, idx.repeat(3, 1, 1))
    assert all(torch.equal(yb.torch_nchw()[i], y_hat.torch_nchw()[0]) for i in range(3))
    # the MFMA prior on the same stream: desynchronises (the reason this path exists)
    mfma = IntraNoAR(synth.codec_state_dict(0, preset=preset), DEV, prior_arith="mfma")
    mfma.update(force=True)
    with pytest.raises(StreamDesyncError):
        mfma.decompress([g["stream"].tobytes()], idx)
    # ... and so does the reference's arithmetic calibrated for the OTHER configuration class (one thread <-> several): loud
    other = IntraNoAR(synth.codec_state_dict(0, preset=preset), DEV, prior_arith="torch_cpu", ref_threads=1 if threads > 1 else 8)
    other.update(force=True)
    with pytest.raises(StreamDesyncError):
        other.decompress([g["stream"].tobytes()], idx)
    # prior_arith="auto" finds the class by itself (MFMA first, then the calibrations), and still raises for a corrupt stream
    auto = IntraNoAR(synth.codec_state_dict(0, preset=preset), DEV, prior_arith="auto", ref_profile=profile)
    auto.update(force=True)
    y3, _ = auto.decompress([g["stream"].tobytes()], idx)
    assert torch.equal(y3.torch_nchw(), y_hat.torch_nchw()) and auto.ref_prior is None
    bad = bytearray(g["stream"].tobytes())
    bad[len(bad) // 2] ^= 0x3C
    with pytest.raises(StreamDesyncError):
        auto.decompress([bytes(bad)], idx)
    blobs, own = auto.encode_latents(idx, None, noise_seed=3)          # its own streams: the MFMA prior, first try
    from onedc_amd.entropy.utils import decode_i
    y4, _ = auto.decompress([decode_i(blobs[0], 14, 64)["bit_stream_y"]], idx)
    assert torch.equal(y4.torch_nchw(), own.torch_nchw())


@pytest.mark.gpu
def test_z_only_latents_in_the_reference_arithmetic_are_the_references_bits(eight_threads):
    """The ex-low-rate model (forward_four_part_prior_recon_with_z, compression_model.py:421-465: y_hat = the prior's means, no
    symbols) in the reference's arithmetic: y_hat of tests/golden/g5 (written by the reference's modules, 128x192, 8 threads)
    bit for bit -- with the exact-f32 MFMA prior the same tensor agrees to 1e-5 (tests/test_gpu_parity.py::test_z_only_prior)."""
    from onedc_amd.codec_module import IntraNoAR
    from onedc_amd.ref_arith import intel_profile
    g = golden("g5_codec_path.npz")
    codec = IntraNoAR(synth.codec_state_dict(0), "cuda:0", prior_arith="torch_cpu", ref_threads=8, ref_profile=intel_profile(8))
    codec.update(force=True)
    y_hat, _ = codec.recon_with_z(torch.from_numpy(g["indices"]).to("cuda:0"))
    assert torch.equal(y_hat.torch_nchw().cpu(), torch.from_numpy(g["y_hat_zonly"]))


@pytest.mark.gpu
def test_auto_mode_decodes_a_batch_of_mixed_producers(eight_threads):
    """prior_arith="auto" on a batch that holds a reference-made stream BETWEEN two streams of this package's encoder: no single
    arithmetic decodes the batch, every picture is decoded in its writer's; a corrupt stream among them is named."""
    from onedc_amd.codec_module import IntraNoAR, StreamDesyncError
    from onedc_amd.entropy.utils import decode_i
    DEV = "cuda:0"
    g = golden("g9_ref_stream_768_trained.npz")
    auto = IntraNoAR(synth.codec_state_dict(0, preset="trained_like"), DEV, prior_arith="auto")
    auto.update(force=True)
    idx_ref = torch.from_numpy(g["indices"]).to(DEV)
    idx_own = torch.from_numpy(np.random.default_rng(3).integers(0, 16384, (2, 12, 12)).astype(np.int32)).to(DEV)
    blobs, own = auto.encode_latents(idx_own, None, noise_seed=7)
    own_streams = [decode_i(b, 14, 64)["bit_stream_y"] for b in blobs]
    streams = [own_streams[0], g["stream"].tobytes(), own_streams[1]]
    idx = torch.cat([idx_own[0:1], idx_ref, idx_own[1:2]])
    y, z = auto.decompress(streams, idx)
    got = y.torch_nchw().cpu()
    assert torch.equal(got[0], own.torch_nchw().cpu()[0]) and torch.equal(got[2], own.torch_nchw().cpu()[1])
    assert torch.equal(got[1].reshape(-1)[torch.from_numpy(g["y_hat.idx"])], torch.from_numpy(g["y_hat.val"]))
    assert z.B == 3 and auto.ref_prior is None
    bad = bytearray(own_streams[1])
    bad[len(bad) // 3] ^= 0x55
    with pytest.raises(StreamDesyncError) as e:
        auto.decompress([own_streams[0], g["stream"].tobytes(), bytes(bad)], idx)
    assert e.value.images == [2]


@pytest.mark.gpu
def test_model_decodes_container_files_of_both_producers_in_one_batch(eight_threads):
    """The whole decoder (`SD15_1step_codec_stage1.decode`, config key `prior_arith: auto`) on three 768x768 container files in one
    batch -- own, reference-made (tests/golden/g9: the y stream the reference wrote, in the reference's container format), own:
    three finite pictures, each bit-identical to the picture the same file decodes to alone."""
    from onedc_amd.entropy.utils import decode_i, encode_i, pack_indices
    from onedc_amd.model_sd15_with_codec_stage1 import SD15_1step_codec_stage1
    DEV = "cuda:0"
    g = golden("g9_ref_stream_768_trained.npz")
    model = SD15_1step_codec_stage1({"vae_attn_patch": 16, "prior_arith": "auto"}, device=DEV).load_synthetic(0, preset="trained_like")
    model.codec_model.update(force=True)
    idx_own = torch.from_numpy(np.random.default_rng(4).integers(0, 16384, (2, 12, 12)).astype(np.int32)).to(DEV)
    own_blobs, _ = model.codec_model.encode_latents(idx_own, None, noise_seed=9)
    ref_blob = encode_i(768, 768, g["stream"].tobytes(), pack_indices(g["indices"].reshape(-1), 14), b"", 0)
    assert decode_i(ref_blob, 14, 64)["bit_stream_y"] == g["stream"].tobytes()
    files = [own_blobs[0], ref_blob, own_blobs[1]]
    batch = model.decode(stream=files, unit_range=True)
    assert batch.shape == (3, 3, 768, 768) and bool(torch.isfinite(batch).all())
    for b, f in enumerate(files):
        alone = model.decode(stream=[f], unit_range=True)
        assert torch.equal(alone[0], batch[b]), b


@pytest.mark.gpu
def test_cli_decodes_a_reference_made_container_with_the_default_config(eight_threads, tmp_path):
    """`python -m onedc_amd.inference --decoder_only` on a container whose y stream THE REFERENCE wrote (tests/golden/g9: 768x768,
    trained-like weights, torch-CPU fp32 at 8 threads; wrapped by the reference's own container format, utils.py:95-105), with a
    config file that names NO prior_arith: the default ("auto") decodes it -- the MFMA prior desynchronises, the reference's
    arithmetic takes over (codec_module.py:418-454 is what the file was made for) -- to the picture the explicit compatibility
    mode decodes it to."""
    import yaml
    from PIL import Image
    from safetensors.torch import save_file
    from onedc_amd import inference
    from onedc_amd.entropy.utils import encode_i, pack_indices
    from onedc_amd.model_sd15_with_codec_stage1 import SD15_1step_codec_stage1
    g = golden("g9_ref_stream_768_trained.npz")
    ckpt = tmp_path / "ckpt"
    ckpt.mkdir()
    save_file(synth.unet_state_dict(0, preset="trained_like"), str(ckpt / "model.safetensors"))
    save_file(synth.codec_state_dict(0, preset="trained_like"), str(ckpt / "model_1.safetensors"))
    save_file(synth.vae_state_dict(0, preset="trained_like"), str(ckpt / "vae.safetensors"))
    (tmp_path / "cfg.yaml").write_text(yaml.safe_dump({"vae_attn_patch": 16, "conditioning_timestep": 999}))
    (tmp_path / "ref.bin").write_bytes(encode_i(768, 768, g["stream"].tobytes(), pack_indices(g["indices"].reshape(-1), 14), b"", 0))
    inference.main(["--config_path", str(tmp_path / "cfg.yaml"), "--checkpoint_path", str(ckpt), "--vae_path", str(ckpt / "vae.safetensors"),
                    "--decoder_only", "--decoder_bin_path", str(tmp_path / "ref.bin"), "--decoder_out_path", str(tmp_path / "ref.png")])
    got = np.asarray(Image.open(tmp_path / "ref.png"))
    assert got.shape == (768, 768, 3)
    explicit = SD15_1step_codec_stage1({"vae_attn_patch": 16, "prior_arith": "torch_cpu", "ref_threads": 8},
                                       device="cuda:0").load_synthetic(0, preset="trained_like")
    explicit.codec_model.update(force=True)
    want = explicit.decode(fp=str(tmp_path / "ref.bin"), out="u8")[0].cpu().numpy()
    assert np.array_equal(got, want)
    # the default really is "auto", and the MFMA prior alone refuses this file
    from onedc_amd.codec_module import StreamDesyncError
    assert SD15_1step_codec_stage1({"vae_attn_patch": 16}, device="cuda:0").load_synthetic(0, preset="trained_like").codec_model.prior_arith == "auto"
    mfma = SD15_1step_codec_stage1({"vae_attn_patch": 16, "prior_arith": "mfma"}, device="cuda:0").load_synthetic(0, preset="trained_like")
    mfma.codec_model.update(force=True)
    with pytest.raises(StreamDesyncError):
        mfma.decode(fp=str(tmp_path / "ref.bin"))


@pytest.mark.gpu
def test_reference_made_1024_stream_decodes_to_the_references_digest(eight_threads):
    """A larger picture (tests/golden/g12: 1024x1024, 524 288 symbols, written at 8 threads): free-running decode in the
    reference's arithmetic ends on the encoder's state, and the WHOLE y_hat (sha256 of its f32 bytes, NCHW) and all symbol planes
    (sha256) are the reference's."""
    import hashlib
    from onedc_amd.codec_module import IntraNoAR
    g = golden("g12_ref_stream_1024_trained.npz")
    codec = IntraNoAR(synth.codec_state_dict(0, preset="trained_like"), "cuda:0", prior_arith="torch_cpu", ref_threads=8)
    codec.update(force=True)
    trace = []
    y_hat, _ = codec.decompress([g["stream"].tobytes()], torch.from_numpy(g["indices"]).to("cuda:0"), trace=trace)
    got = y_hat.torch_nchw().cpu().contiguous()
    assert torch.equal(got.reshape(-1)[torch.from_numpy(g["y_hat.idx"])], torch.from_numpy(g["y_hat.val"]))
    assert hashlib.sha256(got.numpy().tobytes()).digest() == g["y_hat_sha256"].tobytes()
    sym = np.stack([np.asarray(sy).reshape(1, 32, 64, 64) for _, sy in trace]).astype(np.int16)
    assert hashlib.sha256(sym.tobytes()).digest() == g["symbols_sha256"].tobytes()


@pytest.mark.gpu
def test_stream_of_the_reference_harness_class_256_one_thread(eight_threads):
    """BASELINE config 1's class: a 256x256 picture written at ONE thread (what src/inference.py itself does, :29) -- small maps and
    torch's native path everywhere (tests/golden/g14).  Native structures from ref_arith.intel_profile(1); whole y_hat and symbol
    planes to the reference's sha256."""
    import hashlib
    from onedc_amd.codec_module import IntraNoAR
    from onedc_amd.ref_arith import intel_profile
    g = golden("g14_ref_stream_256_trained_1thread.npz")
    codec = IntraNoAR(synth.codec_state_dict(0, preset="trained_like"), "cuda:0", prior_arith="torch_cpu", ref_threads=1,
                      ref_profile=intel_profile(1))
    codec.update(force=True)
    trace = []
    y_hat, _ = codec.decompress([g["stream"].tobytes()], torch.from_numpy(g["indices"]).to("cuda:0"), trace=trace)
    got = y_hat.torch_nchw().cpu().contiguous()
    assert hashlib.sha256(got.numpy().tobytes()).digest() == g["y_hat_sha256"].tobytes()
    sym = np.stack([np.asarray(sy).reshape(1, 32, 16, 16) for _, sy in trace]).astype(np.int16)
    assert hashlib.sha256(sym.tobytes()).digest() == g["symbols_sha256"].tobytes()
    print(f"[ref-arith] g14: structures {dict(sorted(codec.ref_prior.cal.table.items(), key=str))}")


@pytest.mark.gpu
def test_stream_of_an_avx2_writer_decodes_with_the_writers_profile(eight_threads):
    """A writer WITHOUT AVX-512 (tests/golden/g13: the reference with oneDNN capped to AVX2, 8 threads): its 1x1 blockings and its
    one-chain dense 3x3 come from the profile tools/cpu_chain_probe.py wrote on the writer's machine
    (tests/golden/profile_intel_avx2_8threads_768.json) -- this host's torch, whatever its instruction set, is not asked.
    Whole y_hat and symbol planes to the reference's sha256; the avx512 calibration of this host raises StreamDesyncError."""
    import hashlib
    import os
    from onedc_amd.codec_module import IntraNoAR, StreamDesyncError
    from onedc_amd.ref_arith import load_profile
    g = golden("g13_ref_stream_768_trained_avx2.npz")
    prof = load_profile(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "profile_intel_avx2_8threads_768.json"))
    assert prof[("3x3", 128, 128, 48, 48)] == ("direct", 8, 1) and prof["gaussian_tables_sha16"] == "681191522681e23f"
    wrong = dict(prof, gaussian_tables_sha16="0" * 16)          # a writer whose torch built other CDF tables: refused by name
    with pytest.raises(RuntimeError, match="Gaussian CDF tables"):
        IntraNoAR(synth.codec_state_dict(0, preset="trained_like"), "cuda:0", prior_arith="torch_cpu", ref_profile=wrong).update(force=True)
    idx = torch.from_numpy(g["indices"]).to("cuda:0")
    codec = IntraNoAR(synth.codec_state_dict(0, preset="trained_like"), "cuda:0", prior_arith="torch_cpu", ref_threads=8,
                      ref_profile=prof)
    codec.update(force=True)
    trace = []
    y_hat, _ = codec.decompress([g["stream"].tobytes()], idx, trace=trace)
    got = y_hat.torch_nchw().cpu().contiguous()
    assert hashlib.sha256(got.numpy().tobytes()).digest() == g["y_hat_sha256"].tobytes()
    sym = np.stack([np.asarray(sy).reshape(1, 32, 48, 48) for _, sy in trace]).astype(np.int16)
    assert hashlib.sha256(sym.tobytes()).digest() == g["symbols_sha256"].tobytes()
    if torch.backends.cpu.get_cpu_capability() == "AVX512":
        local = IntraNoAR(synth.codec_state_dict(0, preset="trained_like"), "cuda:0", prior_arith="torch_cpu", ref_threads=8)
        local.update(force=True)
        with pytest.raises(StreamDesyncError):
            local.decompress([g["stream"].tobytes()], idx)


@pytest.mark.gpu
def test_streams_written_in_the_reference_arithmetic_carry_the_references_decisions(eight_threads):
    """The other direction (BASELINE north_star: "identical bitstreams on encode"): for a given latent y, the encoder side of the
    prior in torch_cpu mode produces the symbols, the table indexes and therefore the y stream that the reference's CPU
    arithmetic produces (oracle compress_four_part_prior on torch-CPU + the C port of the coder)."""
    from onedc_amd.codec_module import IntraNoAR
    from oracle import rans_port
    from oracle import torch_ref as R
    DEV = "cuda:0"
    sd = synth.codec_state_dict(0, preset="trained_like")
    codec = IntraNoAR(sd, DEV, prior_arith="torch_cpu")
    codec.update(force=True)
    idx = torch.from_numpy(np.random.default_rng(5).integers(0, 16384, (1, 12, 12)).astype(np.int32))
    with torch.no_grad():
        fused, _ = R.z_indices_to_params(sd, idx.long())
        means = fused.chunk(2, 1)[1]
        y = means + 1.5 * torch.randn(means.shape, generator=torch.Generator().manual_seed(6))
        y_q_w, s_w, y_hat_ref = R.compress_four_part_prior(sd, y, fused)
        sym = torch.stack(y_q_w).to(torch.int16).numpy().reshape(-1)
        planes = torch.stack([R.build_indexes(s) for s in s_w]).to(torch.int16).numpy().reshape(-1)
    want = rans_port.encode(sym, planes, rans_port.Tables(*codec._tables))
    blobs, y_hat = codec.encode_latents(idx.to(DEV), y.permute(0, 2, 3, 1).contiguous().to(DEV))
    from onedc_amd.entropy.utils import decode_i
    got = decode_i(blobs[0], 14, 64)["bit_stream_y"]
    assert got == want, (len(got), len(want))
    assert torch.equal(y_hat.torch_nchw().cpu(), y_hat_ref)


# reference-made streams of SMALL pictures (128x192, 256x256; written at 8 threads on the Intel build host).  Under ~0.16 MP torch
# sends the hyper decoder's dense 3x3 convolutions and most 1x1 convolutions through its native path (im2col + MKL sgemm), whose K
# split depends on the thread count and on the CPU MKL dispatches for: the writer's structures come from
# ref_arith.intel_profile(8), the oneDNN shapes from the local calibration.
SMALL_STREAMS = [("g5_codec_path.npz", None), ("g8_ref_stream_256.npz", None), ("g9_ref_stream_256_trained.npz", "trained_like")]


@pytest.mark.gpu
@pytest.mark.parametrize("name,preset", SMALL_STREAMS)
def test_reference_made_small_streams_decode_in_the_reference_arithmetic(eight_threads, name, preset):
    from onedc_amd import kernels as K
    from onedc_amd.codec_module import IntraNoAR
    from onedc_amd.ref_arith import intel_profile
    DEV = "cuda:0"
    g = golden(name)
    codec = IntraNoAR(synth.codec_state_dict(0, preset=preset), DEV, prior_arith="torch_cpu", ref_threads=8,
                      ref_profile=intel_profile(8))
    codec.update(force=True)
    idx = torch.from_numpy(g["indices"]).to(DEV)
    fused, _ = codec._params_from_indices(idx)
    common_cat, y_hat = codec._prior_setup(fused)
    B, H, W, N = fused.B, fused.H, fused.W, codec.N
    idx_dev = torch.empty(B * (N // 4) * H * W, dtype=torch.int16, device=DEV)
    flips, params = [], fused
    for k in range(4):
        if k > 0:
            params = codec._next_params(k, y_hat, common_cat)
        codec._prior_indexes(params, idx_dev, B, H, W, k)
        flips.append(int((idx_dev.view(B, N // 4, H, W).cpu().numpy() != g["index_planes"][k]).sum()))
        K.prior_update_dec(params, torch.from_numpy(np.ascontiguousarray(g["symbols"][k])).to(DEV).reshape(-1), y_hat, k)
    print(f"[ref-arith] {name}: teacher-forced table-index flips per hop {flips}; structures: "
          f"{dict(sorted(codec.ref_prior.cal.table.items(), key=str))}")
    assert flips == [0, 0, 0, 0]
    got = y_hat.torch_nchw().cpu()
    if "y_hat" in g:
        assert torch.equal(got, torch.from_numpy(g["y_hat"]))
    else:
        assert torch.equal(got.reshape(-1)[torch.from_numpy(g["y_hat.idx"])], torch.from_numpy(g["y_hat.val"]))
    trace = []
    y2, _ = codec.decompress([g["stream"].tobytes()], idx, trace=trace)                 # free-running
    assert all(np.array_equal(sy, g["symbols"][k]) for k, (ix, sy) in enumerate(trace))
    assert torch.equal(y2.torch_nchw().cpu(), got)


@pytest.mark.gpu
def test_shapes_without_a_fitting_structure_are_refused_not_misdecoded(eight_threads, monkeypatch):
    """A convolution whose torch-CPU summation none of the candidate structures reproduces raises
    ReferenceArithmeticUnavailable (here: the native dense 3x3 of a 128x192 picture with the candidate list emptied) instead of
    decoding with some other arithmetic."""
    from onedc_amd.codec_module import IntraNoAR
    from onedc_amd.ref_arith import Calibration, ReferenceArithmeticUnavailable
    monkeypatch.setattr(Calibration, "im2col_candidates", staticmethod(lambda K: []))
    g = golden("g5_codec_path.npz")
    codec = IntraNoAR(synth.codec_state_dict(0), "cuda:0", prior_arith="torch_cpu")
    codec.update(force=True)
    with pytest.raises(ReferenceArithmeticUnavailable):
        codec.decompress([g["stream"].tobytes()], torch.from_numpy(g["indices"]).to("cuda:0"))
