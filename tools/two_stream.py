#!/usr/bin/env python3
"""Stage S of one batch of 16 on one stream against two half batches of 8 on two streams (two host threads):
do independent kernel chains fill each other's dispatch gaps?"""
import os
import sys
import threading
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

from onedc_amd.model_sd15_with_codec_stage1 import SD15_1step_codec_stage1

S = 768
model = SD15_1step_codec_stage1({"vae_attn_patch": 16}).load_synthetic(0)
model.codec_model.update(force=True)


def entropy(B, seed):
    idx = torch.from_numpy(np.random.default_rng(seed).integers(0, 16384, (B, S // 64, S // 64)).astype(np.int32)).cuda()
    blobs, _ = model.codec_model.encode_latents(idx, None, noise_seed=2)
    return model.codec_model.decode_entropy(stream=blobs)


ent16 = entropy(16, 1)
ent8 = [entropy(8, 2), entropy(8, 3)]
ent4 = [entropy(4, 4 + i) for i in range(4)]
torch.cuda.synchronize()


def run_parallel(ents, n):
    streams = [torch.cuda.Stream() for _ in ents]

    def work(e, s):
        with torch.cuda.stream(s), torch.no_grad():
            for _ in range(n):
                model._decode_after_entropy(e)

    ths = [threading.Thread(target=work, args=(e, s)) for e, s in zip(ents, streams)]
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for t in ths:
        t.start()
    for t in ths:
        t.join()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e3


with torch.no_grad():
    for e in (ent16, *ent8, *ent4):
        model._decode_after_entropy(e)
    print(f"1 stream  x 16 pictures: {run_parallel([ent16], 4):8.2f} ms per 16 pictures")
    print(f"2 streams x  8 pictures: {run_parallel(ent8, 4):8.2f} ms per 16 pictures")
    print(f"4 streams x  4 pictures: {run_parallel(ent4, 4):8.2f} ms per 16 pictures")
    print(f"2 streams x 16 pictures: {run_parallel([ent16, ent16], 3) / 2:8.2f} ms per 16 pictures")
