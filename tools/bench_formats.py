#!/usr/bin/env python3
"""bf16 vs IEEE-half builds of the same conv kernels, interleaved rounds in one process on the same random data
(cdna_hip_programming.md rule 24): is there a price for the 11-bit mantissa?  Prints median ms and TFLOP/s per format."""
import os
import statistics
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from onedc_amd import kernels as K
from onedc_amd import pack

DEV = "cuda:0"
SHAPES = [  # (name, B, H, W, Cin, Cout, k)
    ("vae 128->128 3x3 @768", 16, 768, 768, 128, 128, 3),
    ("vae 256->256 3x3 @384", 16, 384, 384, 256, 256, 3),
    ("vae 512->512 3x3 @192", 16, 192, 192, 512, 512, 3),
    ("unet 1280->1280 3x3 @24", 16, 24, 24, 1280, 1280, 3),
    ("unet ff 320->2560 1x1 @96", 16, 96, 96, 320, 2560, 1),
]
ROUNDS = 7


def main():
    g = torch.Generator(device=DEV).manual_seed(0)
    for name, B, H, W, Cin, Cout, k in SHAPES:
        xf = torch.randn((B * H * W + 128) * Cin, generator=g, device=DEV)
        wf = torch.randn(Cout, k * k * Cin, generator=g, device=DEV) * (k * k * Cin) ** -0.5
        fl = 2.0 * B * H * W * Cout * k * k * Cin
        data = {}
        for fmt in ("bf16", "f16"):
            with K.act_format(fmt):
                data[fmt] = (K.View(xf.to(K.a16()), B, H, W, Cin), pack.pack_matrix(wf, DEV), K.new_view(B, H, W, Cout, DEV))
        times = {"bf16": [], "f16": []}
        n = min(200, max(5, int(20.0 / (fl / 800e9))))
        for r in range(ROUNDS + 1):
            for fmt in ("bf16", "f16") if r % 2 == 0 else ("f16", "bf16"):
                x, w, y = data[fmt]
                with K.act_format(fmt):
                    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                    e0.record()
                    for _ in range(n):
                        K.conv_gemm(x, w, None, y, KH=k)
                    e1.record()
                    torch.cuda.synchronize()
                if r:
                    times[fmt].append(e0.elapsed_time(e1) / n)
        mb, mf = statistics.median(times["bf16"]), statistics.median(times["f16"])
        print(f"{name:28s} bf16 {mb:7.3f} ms {fl / mb / 1e9:7.1f} TF   f16 {mf:7.3f} ms {fl / mf / 1e9:7.1f} TF   "
              f"f16/bf16 time {mf / mb:.3f}  (min {min(times['f16']) / min(times['bf16']):.3f})", flush=True)
        del data, xf, wf


if __name__ == "__main__":
    main()
