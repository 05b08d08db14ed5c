#!/usr/bin/env python3
"""1x1 / small-K layers of the UNet (B=16) on the 128x256 (tile=129), 256x256 (256) and 4-wave (128) kernels."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from onedc_amd import kernels as K
from onedc_amd import pack

DEV = "cuda:0"
g = torch.Generator(device=DEV).manual_seed(0)
SH = [("ff-in geglu 320->2560 @96", 96, 320, 2560, 1, K.EPI_GEGLU, False), ("ff-in geglu 640->5120 @48", 48, 640, 5120, 1, K.EPI_GEGLU, False),
      ("ff-in geglu 1280->10240 @24", 24, 1280, 10240, 1, K.EPI_GEGLU, False), ("qkv 320->512 @96", 96, 320, 512, 1, 0, False),
      ("out 384->320 @96 +res", 96, 384, 320, 1, 0, True), ("ff-out 1280->320 @96 +res", 96, 1280, 320, 1, 0, True),
      ("qkv 640->768 @48", 48, 640, 768, 1, 0, False), ("proj 640->640 @48 +res", 48, 640, 640, 1, 0, True),
      ("3x3 1280->1280 @12 +res", 12, 1280, 1280, 3, 0, True), ("3x3 2560->1280 @12", 12, 2560, 1280, 3, 0, False),
      ("1x1 1280->1280 @24", 24, 1280, 1280, 1, 0, False), ("3x3 1280->1280 @24 +res", 24, 1280, 1280, 3, 0, True),
      # round 4: the rest of the UNet's 1x1 shapes (profiles/r4_unet_layers.txt), to check the dispatcher's choice per shape
      ("qk 320->1024 @96", 96, 320, 1024, 1, 0, False), ("proj 320->320 @96 +res", 96, 320, 320, 1, 0, True),
      ("qk 640->2048 @48", 48, 640, 2048, 1, 0, False), ("out 768->640 @48 +res", 48, 768, 640, 1, 0, True),
      ("ff-out 2560->640 @48 +res", 48, 2560, 640, 1, 0, True), ("q 640->1024 @48", 48, 640, 1024, 1, 0, False),
      ("qk 1280->2560 @24", 24, 1280, 2560, 1, 0, False), ("out 1280->1280 @24 +res", 24, 1280, 1280, 1, 0, True),
      ("ff-out 5120->1280 @24 +res", 24, 5120, 1280, 1, 0, True), ("3x3 2560->1280 @24", 24, 2560, 1280, 3, 0, False),
      ("3x3 1280->1280 @24", 24, 1280, 1280, 3, 0, False), ("1x1 1280->1280 @12 +res", 12, 1280, 1280, 1, 0, True),
      ("ff-in geglu 1280->10240 @12", 12, 1280, 10240, 1, K.EPI_GEGLU, False), ("ff-out 5120->1280 @12 +res", 12, 5120, 1280, 1, 0, True)]
for name, H, Cin, Cout, k, epi, res in SH:
    B = 16
    x = K.View(torch.randn((B * H * H + 128) * Cin, generator=g, device=DEV).to(K.a16()), B, H, H, Cin)
    glu = epi in (K.EPI_GEGLU, K.EPI_GLU_LEAKY)
    wm = torch.randn(Cout, k * k * Cin, generator=g, device=DEV) * (k * k * Cin) ** -0.5
    w = pack.pack_matrix(wm, DEV, pack.glu_perm(Cout // 2).to(DEV) if glu else None)
    y = K.new_view(B, H, H, Cout // 2 if glu else Cout, DEV)
    r = K.new_view(B, H, H, Cout, DEV) if res else None
    if r is not None:
        r.t.normal_()
    fl = 2.0 * B * H * H * Cout * k * k * Cin
    out = []
    tiles = (0, 128, 129, 256) + tuple(int(a) for a in sys.argv[1:])        # extra ids: experiment builds (EXP_TILES: 132..135)
    for _ in range(10):                # (the first column used to be measured cold -- fresh tensors, the clock still ramping: round 5
        K.conv_gemm(x, w, None, y, KH=k, epi=epi, res=r)                     # read 14 % into a tile that an interleaved A/B puts level)
    torch.cuda.synchronize()
    for tile in tiles:
        if tile == 256 and Cout % 256:
            out.append("   n/a   ")
            continue
        try:
            for _ in range(3):
                K.conv_gemm(x, w, None, y, KH=k, epi=epi, res=r, tile=tile)
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(20):
                K.conv_gemm(x, w, None, y, KH=k, epi=epi, res=r, tile=tile)
            e1.record()
            torch.cuda.synchronize()
            ms = e0.elapsed_time(e1) / 20
            out.append(f"{ms * 1e3:6.0f} us {fl / ms / 1e9:5.0f}")
        except Exception:
            out.append("   err   ")
    print(f"{name:30s} auto {out[0]} | 4w {out[1]} | 129 {out[2]} | 256 {out[3]}" +
          "".join(f" | {t} {o}" for t, o in zip(tiles[4:], out[4:])))
