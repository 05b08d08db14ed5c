#!/usr/bin/env python3
"""A/B of two BUILDS of the kernel library on the window-kernel layers, interleaved rounds in one process on the same data
(cdna_hip_programming.md rule 24: boxes and clocks differ by more than most effects).

    tools/exp_build.sh stag4 -DONEDC_ACT_F16 -DEXP_W4_STAGGER=4
    python tools/ab_libs.py tools/exp/libonedc_hip_stag4.so            # against the in-tree f16 library

Variants per shape: plain, + residual, + GroupNorm on load + residual + statistics (what the VAE's resnets run)."""
import ctypes
import os
import statistics
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from onedc_amd import kernels as K
from onedc_amd import pack

DEV = "cuda:0"
FMT = os.environ.get("ONEDC_ACT", "f16")
SHAPES = [(16, 768, 128, 128), (16, 768, 256, 128), (16, 384, 256, 256), (16, 384, 512, 256), (16, 192, 512, 512),
          (16, 96, 320, 320)]
ROUNDS = 5


class WS:
    def __init__(self):
        self.t = {}

    def f32(self, name, n):
        if name not in self.t or self.t[name].numel() < n:
            self.t[name] = torch.empty(n, dtype=torch.float32, device=DEV)
        return self.t[name]


def main():
    base = K.lib(FMT)
    exp = K.attach_signatures(ctypes.CDLL(os.path.abspath(sys.argv[1])))
    K.check_library_format(FMT, exp.onedc_hip_act_format(), exp.onedc_hip_version())
    libs = {"base": base, "exp": exp}
    g = torch.Generator(device=DEV).manual_seed(0)
    ws = WS()
    with K.act_format(FMT):
        for B, H, Cin, Cout in SHAPES:
            x = K.new_view(B, H, H, Cin, DEV, stats=True)
            x.t.normal_()
            x.stats.t.normal_().abs_()
            x.stats.mark(0, Cin)
            r = K.new_view(B, H, H, Cout, DEV)
            r.t.normal_()
            w = pack.pack_matrix(torch.randn(Cout, 9 * Cin, generator=g, device=DEV) * (9 * Cin) ** -0.5, DEV)
            gam, bet = torch.ones(Cin, device=DEV), torch.zeros(Cin, device=DEV)
            y = K.new_view(B, H, H, Cout, DEV, stats=True)
            fl = 2.0 * B * H * H * Cout * 9 * Cin
            n = min(100, max(5, int(20.0 / (fl / 800e9))))
            variants = {"plain": {}, "res": {"res": r}}
            if Cout <= K.GN_FUSION_MAX_COUT:
                variants["gn+res+stats"] = {"res": r, "gn": (gam, bet, 1e-6, True, ws), "stats": True}
            for vname, kw in variants.items():
                times = {"base": [], "exp": []}
                for rnd in range(ROUNDS + 1):
                    for which in ("base", "exp") if rnd % 2 == 0 else ("exp", "base"):
                        K._libs[FMT] = libs[which]
                        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                        e0.record()
                        for _ in range(n):
                            K.conv_gemm(x, w, None, y, KH=3, **kw)
                        e1.record()
                        torch.cuda.synchronize()
                        if rnd:
                            times[which].append(e0.elapsed_time(e1) / n)
                mb, me = statistics.median(times["base"]), statistics.median(times["exp"])
                print(f"{H}^2 {Cin}->{Cout} {vname:13s} base {mb:7.3f} ms {fl / mb / 1e9:7.1f} TF   exp {me:7.3f} ms "
                      f"{fl / me / 1e9:7.1f} TF   exp/base time {me / mb:.3f}", flush=True)
            K._libs[FMT] = base
            del x, r, y, w


if __name__ == "__main__":
    main()
