"""Report of the reference-arithmetic prior (IntraNoAR(prior_arith="torch_cpu")) on this machine: the calibrated structures of
this host's torch at the writer's thread count, and for the reference-made golden streams that carry their planes the teacher-forced table-index flips,
the free-running verdict and the time of the entropy decode in both arithmetics.  Output: stdout (copy to profiles/).

    python tools/ref_arith_report.py
"""
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from onedc_amd import kernels as K  # noqa: E402
from onedc_amd import synth  # noqa: E402
from onedc_amd.codec_module import IntraNoAR, StreamDesyncError  # noqa: E402
from onedc_amd.ref_arith import intel_profile  # noqa: E402

GOLDEN = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden")
STREAMS = [("g5_codec_path.npz", None, 8, "intel"), ("g8_ref_stream_256.npz", None, 8, "intel"),
           ("g9_ref_stream_256_trained.npz", "trained_like", 8, "intel"),
           ("g11_ref_stream_512x768_trained_4threads.npz", "trained_like", 4, None),
           ("g8_ref_stream_768.npz", None, 8, None), ("g9_ref_stream_768_trained.npz", "trained_like", 8, None),
           ("g10_ref_stream_768_trained_1thread.npz", "trained_like", 1, None),
           ("g10_ref_stream_768_trained_1thread.npz", "trained_like", 1, "intel")]
DEV = "cuda:0"


def cpu_name():
    for line in open("/proc/cpuinfo"):
        if line.startswith("model name"):
            return line.split(":", 1)[1].strip()
    return "?"


def flips_of(codec, g):
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
    flat = y_hat.torch_nchw().cpu().reshape(-1)
    if "y_hat" in g:
        same = bool(torch.equal(flat, torch.from_numpy(g["y_hat"]).reshape(-1)))
    else:
        same = bool(torch.equal(flat[torch.from_numpy(g["y_hat.idx"])], torch.from_numpy(g["y_hat.val"])))
    return flips, same


def timed(codec, stream, idx, n=15):
    try:
        codec.decompress([stream], idx)
    except StreamDesyncError:
        return "DESYNC (raised)", None
    ts = []
    for _ in range(n):
        torch.cuda.synchronize()
        t = time.perf_counter()
        codec.decompress([stream], idx)
        torch.cuda.synchronize()
        ts.append((time.perf_counter() - t) * 1e3)
    return "clean", float(np.median(ts))


def main():
    print(f"[host] torch {torch.__version__}, cpu: {cpu_name()}, gpu: {torch.cuda.get_device_name(0)}")
    for name, preset, threads, prof in STREAMS:
        g = dict(np.load(os.path.join(GOLDEN, name)))
        idx = torch.from_numpy(g["indices"]).to(DEV)
        stream = g["stream"].tobytes()
        sd = synth.codec_state_dict(0, preset=preset)
        ref = IntraNoAR(sd, DEV, prior_arith="torch_cpu", ref_threads=threads,
                        ref_profile=intel_profile(threads) if prof == "intel" else None)
        ref.update(force=True)
        flips, same = flips_of(ref, g)
        verdict, ms = timed(ref, stream, idx)
        tag = f"{name} (writer: {threads} thread(s), Intel AVX-512 host; structures: " + \
              (f"native shapes from the writer's profile intel_profile({threads}), oneDNN shapes calibrated on this host" if prof else f"calibrated on this host at {threads} thread(s)") + ")"
        print(f"[ref-arith] {tag}")
        print(f"    torch_cpu prior: table-index flips per hop {flips} of {g['index_planes'].size}; y_hat bit-equal: {same}; "
              f"free-running {verdict}" + (f"; entropy decode {ms:.1f} ms / picture (median of 15)" if ms else ""))
        print(f"    structures (C, OC, H, W): (K block, bias first): {dict(sorted(ref.ref_prior.cal.table.items(), key=str))}")
        mf = IntraNoAR(sd, DEV)
        mf.update(force=True)
        flips, same = flips_of(mf, g)
        verdict, ms = timed(mf, stream, idx)
        print(f"    mfma prior:      table-index flips per hop {flips}; y_hat bit-equal: {same}; free-running {verdict}")
        # the MFMA prior's time on a stream it can decode: its own
        blobs, _ = mf.encode_latents(idx, None, noise_seed=3)
        from onedc_amd.entropy.utils import decode_i
        own = decode_i(blobs[0], 14, 64)["bit_stream_y"]
        _, ms_m = timed(mf, own, idx)
        _, ms_r = timed(ref, own, idx) if prof is None else (None, None)
        print(f"    own stream ({len(own)} bytes): entropy decode mfma {ms_m:.1f} ms" +
              (f", torch_cpu {ms_r:.1f} ms" if ms_r else "") + " / picture (batch 1, sequential, host rANS included)")


if __name__ == "__main__":
    main()
