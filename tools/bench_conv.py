#!/usr/bin/env python3
"""Micro-benchmark of onedc_conv_gemm on the layer shapes that dominate a 768x768, B=16 decode.
Random bf16 data (zero-filled operands read ~15-20 % high on this chip).  Prints TFLOP/s per shape."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from onedc_amd import kernels as K
from onedc_amd import pack

DEV = "cuda:0"
SHAPES = [  # (name, B, H, W, Cin, Cout, k, up)
    ("vae 128->128 3x3 @768", 16, 768, 768, 128, 128, 3, 0),
    ("vae 256->128 3x3 @768", 16, 768, 768, 256, 128, 3, 0),
    ("vae 256->256 3x3 up->768", 16, 384, 384, 256, 256, 3, 1),
    ("vae 256->256 3x3 @384", 16, 384, 384, 256, 256, 3, 0),
    ("vae 512->256 3x3 @384", 16, 384, 384, 512, 256, 3, 0),
    ("vae 512->512 3x3 @192", 16, 192, 192, 512, 512, 3, 0),
    ("vae 512->512 3x3 @96", 16, 96, 96, 512, 512, 3, 0),
    ("unet 320->320 3x3 @96", 16, 96, 96, 320, 320, 3, 0),
    ("unet 640->640 3x3 @48", 16, 48, 48, 640, 640, 3, 0),
    ("unet 1280->1280 3x3 @24", 16, 24, 24, 1280, 1280, 3, 0),
    ("unet 2560->1280 3x3 @12", 16, 12, 12, 2560, 1280, 3, 0),
    ("unet ff 320->2560 1x1 @96", 16, 96, 96, 320, 2560, 1, 0),
    ("unet ff 1280->320 1x1 @96", 16, 96, 96, 1280, 320, 1, 0),
    ("codec 256->256 1x1 @48", 16, 48, 48, 256, 256, 1, 0),
]


TILES = tuple(int(t) for t in os.environ.get("BENCH_TILES", "129,256,130,131").split(","))


def main():
    only = sys.argv[1] if len(sys.argv) > 1 else None
    g = torch.Generator(device=DEV).manual_seed(0)
    tot_f = tot_t = 0.0
    for name, B, H, W, Cin, Cout, k, up in SHAPES:
        if only and only not in name:
            continue
        x = K.View(torch.randn((B * H * W + 128) * Cin, generator=g, device=DEV).to(K.a16()), B, H, W, Cin)
        w = pack.pack_matrix(torch.randn(Cout, k * k * Cin, generator=g, device=DEV) * (k * k * Cin) ** -0.5, DEV)
        OH, OW = (2 * H, 2 * W) if up else (H, W)
        y = K.new_view(B, OH, OW, Cout, DEV)
        fl = 2.0 * B * OH * OW * Cout * k * k * Cin
        res = []
        for tile in TILES:                      # interleaved in one process (same data, same clocks)
            if ((tile == 256 and Cout % 256) or (tile in (129, 130, 131) and (Cout + 127) // 128 * 128 * 5 > Cout * 6) or
                    (tile in (130, 131) and (k != 3 or OH % 16 or OW % 16)) or (tile == 131 and (up or Cout % 64))):
                res.append(float("nan"))
                continue
            for _ in range(2):
                K.conv_gemm(x, w, None, y, KH=k, up=up, tile=tile)
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            n = max(5, int(20.0 / (fl / 800e9)))   # ~20 ms of sustained work per measurement
            n = min(n, 200)
            e0.record()
            for _ in range(n):
                K.conv_gemm(x, w, None, y, KH=k, up=up, tile=tile)
            e1.record()
            torch.cuda.synchronize()
            res.append(e0.elapsed_time(e1) / n)
        ms = min(r for r in res if r == r)
        tot_f += fl
        tot_t += ms
        print(f"{name:32s} " + "  ".join(f"tile{t}: {r:8.3f} ms {fl / r / 1e9:7.1f} TF" for t, r in zip(TILES, res)))
        del x, y, w
    print(f"{'sum':32s} {tot_t:9.3f} ms  {tot_f / tot_t / 1e9:8.1f} TFLOP/s")


if __name__ == "__main__":
    main()
