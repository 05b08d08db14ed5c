#!/usr/bin/env python3
"""A/B of two BUILDS of the kernel library on the window-kernel layers, interleaved rounds in one process on the same data
(cdna_hip_programming.md rule 24: boxes and clocks differ by more than most effects).

    tools/exp_build.sh stag4 -DONEDC_ACT_F16 -DEXP_W4_STAGGER=4
    python tools/ab_libs.py tools/exp/libonedc_hip_stag4.so            # against the in-tree f16 library
    python tools/ab_libs.py A.so B.so C.so                              # (round 6) several builds, all interleaved with the base
    AB_SHAPES=768:128:128,192:512:512 python tools/ab_libs.py A.so      # a subset of the shapes (H:Cin:Cout)

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
    libs = {"base": base}
    for path in sys.argv[1:]:
        exp = K.attach_signatures(ctypes.CDLL(os.path.abspath(path)))
        K.check_library_format(FMT, exp.onedc_hip_act_format(), exp.onedc_hip_version())
        libs[os.path.basename(path).replace("libonedc_hip_", "").replace(".so", "")] = exp
    names = list(libs)
    shapes = SHAPES
    if os.environ.get("AB_SHAPES"):
        shapes = [(16,) + tuple(int(v) for v in t.split(":")) for t in os.environ["AB_SHAPES"].split(",")]
    g = torch.Generator(device=DEV).manual_seed(0)
    ws = WS()
    with K.act_format(FMT):
        for B, H, Cin, Cout in shapes:
            x = K.new_view(B, H, H, Cin, DEV, stats=True)
            x.t.normal_()
            x.stats.t.normal_().abs_()
            x.stats.mark(0, Cin)
            r = K.new_view(B, H, H, Cout, DEV)
            r.t.normal_()
            wf = torch.randn(Cout, 9 * Cin, generator=g, device=DEV) * (9 * Cin) ** -0.5
            w = pack.pack_matrix(wf, DEV)
            w_bf = pack.pack_matrix(wf, DEV, None, torch.bfloat16) if FMT == "f16" else None
            gam, bet = torch.ones(Cin, device=DEV), torch.zeros(Cin, device=DEV)
            y = K.new_view(B, H, H, Cout, DEV, stats=True)
            fl = 2.0 * B * H * H * Cout * 9 * Cin
            n = min(100, max(5, int(20.0 / (fl / 800e9))))
            variants = {"plain": {}, "res": {"res": r}}
            if Cout <= K.GN_FUSION_MAX_COUT:
                variants["gn+res+stats"] = {"res": r, "gn": (gam, bet, 1e-6, True, ws), "stats": True}
                if w_bf is not None:            # what the decode path's resnets run since round 5: bf16 operands
                    variants["gn+bf+stats"] = {"gn": (gam, bet, 1e-6, True, ws), "stats": True, "w_bf16": w_bf}
                    variants["gn+bf+res+st"] = {"res": r, "gn": (gam, bet, 1e-6, True, ws), "stats": True, "w_bf16": w_bf}
                    # the same layer with the GroupNorm as its own scale / shift pass in front (what layers of more than
                    # GN_FUSION_MAX_COUT output channels run): pass + conv on the bf16 tensor it stores
                    variants["PASS+bf+stats"] = {"gn": (gam, bet, 1e-6, True, ws), "stats": True, "w_bf16": w_bf, "_nofuse": True}
            for vname, kw in variants.items():
                times = {k: [] for k in names}
                for rnd in range(ROUNDS + 1):
                    for which in (names if rnd % 2 == 0 else names[::-1]):
                        K._libs[FMT] = libs[which]
                        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                        e0.record()
                        kw2 = {k: v for k, v in kw.items() if not k.startswith("_")}
                        K.USE_GN_FUSION = not kw.get("_nofuse", False)
                        for _ in range(n):
                            K.conv_gemm(x, w, None, y, KH=3, **kw2)
                        K.USE_GN_FUSION = True
                        e1.record()
                        torch.cuda.synchronize()
                        if rnd:
                            times[which].append(e0.elapsed_time(e1) / n)
                mb = statistics.median(times["base"])
                line = f"{H}^2 {Cin}->{Cout} {vname:13s} base {mb:7.3f} ms {fl / mb / 1e9:7.1f} TF"
                for k in names[1:]:
                    me = statistics.median(times[k])
                    line += f" | {k} {me:7.3f} ms {fl / me / 1e9:7.1f} TF x{me / mb:.3f}"
                print(line, flush=True)
            K._libs[FMT] = base
            del x, r, y, w


if __name__ == "__main__":
    main()
