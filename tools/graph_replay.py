#!/usr/bin/env python3
"""Stage S (synthesis + UNet + VAE) replayed from a captured HIP graph against eager launches; external timing
events inside the capture."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

from onedc_amd import kernels as K
from onedc_amd.model_sd15_with_codec_stage1 import SD15_1step_codec_stage1

B, S = 16, 768
model = SD15_1step_codec_stage1({"vae_attn_patch": 16}).load_synthetic(0)
model.codec_model.update(force=True)
idx = torch.from_numpy(np.random.default_rng(1).integers(0, 16384, (B, S // 64, S // 64)).astype(np.int32)).cuda()
blobs, _ = model.codec_model.encode_latents(idx, None, noise_seed=2)
ent = model.codec_model.decode_entropy(stream=blobs)


def timed(fn, n=4):
    fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e3


with torch.no_grad():
    ref = model._decode_after_entropy(ent)
    print(f"eager stage S      {timed(lambda: model._decode_after_entropy(ent)):8.2f} ms")
    s = torch.cuda.Stream()
    s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s):
        for _ in range(2):
            model._decode_after_entropy(ent)
    torch.cuda.current_stream().wait_stream(s)
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    t0 = time.perf_counter()
    with torch.cuda.graph(g, stream=s):
        out = model._decode_after_entropy(ent)
    print(f"capture took       {(time.perf_counter() - t0) * 1e3:8.2f} ms")
    g.replay()
    torch.cuda.synchronize()
    print("graph output equals eager:", bool(torch.equal(out, ref)))
    print(f"graph replay       {timed(lambda: g.replay()):8.2f} ms")

    # external timing events inside a capture
    try:
        g2 = torch.cuda.CUDAGraph()
        evs = []
        K.PROFILE_EVENT_KW = {"external": True}
        K.PROFILE = []
        with torch.cuda.graph(g2, stream=s):
            out2 = model._decode_after_entropy(ent)
        prof, K.PROFILE = K.PROFILE, None
        g2.replay()
        torch.cuda.synchronize()
        tot = sum(e0.elapsed_time(e1) for _, _, e0, e1, _ in prof)
        print(f"graph with {len(prof)} external event pairs: conv time {tot:8.2f} ms; replay {timed(lambda: g2.replay()):8.2f} ms")
    except Exception as e:   # noqa
        print("external events in capture failed:", type(e).__name__, str(e)[:300])
