#!/usr/bin/env python3
"""The gated Linear layers (GEGLU) of the UNet on the in-tree library against an experiment build, interleaved rounds in one process
(cdna_hip_programming.md rule 24).  Round 5: tools/exp_build.sh r5base -DONEDC_ACT_F16 -DEXP_R5_BASELINE keeps round 4's dispatch
(128 x 128 tiles); the product sends these layers to the 128 x 256 tile of the ring kernel.
    python tools/ab_gated.py tools/exp/libonedc_hip_r5base.so"""
import ctypes
import os
import statistics
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from onedc_amd import kernels as K
from onedc_amd import pack

DEV, FMT = "cuda:0", "f16"
SHAPES = [(96, 320, 2560), (48, 640, 5120), (24, 1280, 10240), (12, 1280, 10240)]


def main():
    base = K.lib(FMT)
    exp = K.attach_signatures(ctypes.CDLL(os.path.abspath(sys.argv[1])))
    libs = {"product": base, "exp": exp}
    g = torch.Generator(device=DEV).manual_seed(0)
    B = 16
    with K.act_format(FMT):
        for H, Cin, Cout in SHAPES:
            x = K.View(torch.randn((B * H * H + 128) * Cin, generator=g, device=DEV).to(K.a16()), B, H, H, Cin)
            w = pack.pack_matrix(torch.randn(Cout, Cin, generator=g, device=DEV) * Cin ** -0.5, DEV, pack.glu_perm(Cout // 2).to(DEV))
            y = K.new_view(B, H, H, Cout // 2, DEV)
            times = {k: [] for k in libs}
            names = {}
            for rnd in range(7):
                for which in (("product", "exp") if rnd % 2 == 0 else ("exp", "product")):
                    K._libs[FMT] = libs[which]
                    for _ in range(3):
                        K.conv_gemm(x, w, None, y, KH=1, epi=K.EPI_GEGLU)
                    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                    e0.record()
                    for _ in range(20):
                        K.conv_gemm(x, w, None, y, KH=1, epi=K.EPI_GEGLU)
                    e1.record()
                    torch.cuda.synchronize()
                    if rnd:
                        times[which].append(e0.elapsed_time(e1) / 20 * 1e3)
                    p = K.ConvParams()
            K._libs[FMT] = base
            mp, me = statistics.median(times["product"]), statistics.median(times["exp"])
            print(f"GEGLU {Cin:5d} -> {Cout:6d} at {H}x{H}: product {mp:7.1f} us   exp {me:7.1f} us   product / exp {mp / me:.3f}", flush=True)


if __name__ == "__main__":
    main()
