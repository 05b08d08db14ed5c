#!/usr/bin/env python3
"""Round-4 A/B on the layers the two new forms replace, interleaved rounds in ONE process on the same data
(cdna_hip_programming.md rule 24):
  split-K      3x3 layers of the UNet's 12x12 level (B = 16): kernels.SPLIT_K 0 (dispatcher: five K ranges) against 1 (never)
  second src   VAE up-block resnets whose conv_shortcut rides on conv2 (768x768 256 -> 128 with GroupNorm on load, 384x384
               512 -> 256) and the UNet's 96x96 / 48x48 up resnets: one launch against shortcut launch + conv2 with residual
"""
import os
import statistics
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from onedc_amd import kernels as K
from onedc_amd import pack

DEV = "cuda:0"
ROUNDS = 5


class WS:
    def __init__(self):
        self.t = {}

    def f32(self, name, n):
        if name not in self.t or self.t[name].numel() < n:
            self.t[name] = torch.empty(n, dtype=torch.float32, device=DEV)
        return self.t[name]


def timed(fns, n):
    times = {k: [] for k in fns}
    order = list(fns)
    for rnd in range(ROUNDS + 1):
        for which in (order if rnd % 2 == 0 else order[::-1]):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(n):
                fns[which]()
            e1.record()
            torch.cuda.synchronize()
            if rnd:
                times[which].append(e0.elapsed_time(e1) / n)
    return {k: statistics.median(v) for k, v in times.items()}


def main():
    g = torch.Generator(device=DEV).manual_seed(0)
    ws = WS()
    print("== split-K, 12x12, B = 16")
    for Cin, Cout, res in ((1280, 1280, True), (1280, 1280, False), (2560, 1280, False)):
        B, H = 16, 12
        x = K.new_view(B, H, H, Cin, DEV)
        x.t.normal_()
        r = K.new_view(B, H, H, Cout, DEV)
        r.t.normal_()
        w = pack.pack_matrix(torch.randn(Cout, 9 * Cin, generator=g, device=DEV) * (9 * Cin) ** -0.5, DEV)
        y = K.new_view(B, H, H, Cout, DEV)

        def run(split):
            K.SPLIT_K = split
            K.conv_gemm(x, w, None, y, KH=3, res=r if res else None)
            K.SPLIT_K = 0
        t = timed({"split": lambda: run(0), "unsplit": lambda: run(1)}, 50)
        fl = 2.0 * B * H * H * Cout * 9 * Cin
        print(f"3x3 {Cin}->{Cout} res={int(res)}: split {t['split'] * 1e3:7.1f} us {fl / t['split'] / 1e9:7.1f} TF   unsplit "
              f"{t['unsplit'] * 1e3:7.1f} us {fl / t['unsplit'] / 1e9:7.1f} TF   ratio {t['split'] / t['unsplit']:.3f}", flush=True)
    print("== 1x1 layers on the 4-wave tiles: line stores (round 4) and the 32-channel four-stage ring against round 3's kernel")
    for H, Cin, Cout, epi, res in ((96, 320, 2560, K.EPI_GEGLU, False), (96, 320, 1024, K.EPI_LINEAR, False), (96, 384, 320, K.EPI_LINEAR, True),
                                   (96, 1280, 320, K.EPI_LINEAR, True), (96, 320, 320, K.EPI_LINEAR, True), (96, 320, 512, K.EPI_LINEAR, False),
                                   (48, 640, 5120, K.EPI_GEGLU, False), (48, 2560, 640, K.EPI_LINEAR, True), (48, 768, 640, K.EPI_LINEAR, True),
                                   (48, 640, 2048, K.EPI_LINEAR, False), (24, 1280, 1280, K.EPI_LINEAR, True), (24, 5120, 1280, K.EPI_LINEAR, True)):
        B = 16
        glu = epi == K.EPI_GEGLU
        x = K.new_view(B, H, H, Cin, DEV)
        x.t.normal_()
        cs = Cout // 2 if glu else Cout
        r = K.new_view(B, H, H, cs, DEV)
        r.t.normal_()
        w = pack.pack_matrix(torch.randn(Cout, Cin, generator=g, device=DEV) * Cin ** -0.5, DEV)
        y = K.new_view(B, H, H, cs, DEV)
        names = {}

        def run(tile):
            K.conv_gemm(x, w, None, y, KH=1, epi=epi, res=r if res else None, tile=tile)
        t = timed({"auto": lambda: run(0), "r3": lambda: run(132), "k64": lambda: run(128), "k32": lambda: run(133), "k32x3": lambda: run(134)}, 40)
        fl = 2.0 * B * H * H * Cout * Cin
        print(f"{H}^2 1x1 {Cin}->{Cout} epi{epi} res={int(res)}: auto {t['auto'] * 1e3:7.1f} us {fl / t['auto'] / 1e9:7.1f} TF | round 3 (scattered "
              f"stores) {t['r3'] * 1e3:7.1f} us | 64-ch steps + line stores {t['k64'] * 1e3:7.1f} | 32-ch ring + line stores {t['k32'] * 1e3:7.1f} | 3-stage ring, 3 WG/CU {t['k32x3'] * 1e3:7.1f} | "
              f"auto / round 3 {t['auto'] / t['r3']:.3f}", flush=True)
        del x, r, y, w
    print("== prior-side GEMMs (f32 activations, 48x48, B = 16): exact f32 MFMA against split-half operands, 64-row and 128-row tiles")
    for Cin, Cout, epi, res in ((256, 256, K.EPI_LEAKY, False), (256, 256, K.EPI_LINEAR, True), (256, 1024, K.EPI_GLU_LEAKY, False),
                                (512, 256, K.EPI_LINEAR, True), (128, 128, K.EPI_LEAKY, False), (128, 512, K.EPI_GLU_LEAKY, False),
                                (256, 128, K.EPI_LINEAR, False)):
        B, H = 16, 48
        glu = epi == K.EPI_GLU_LEAKY
        x = K.new_view(B, H, H, Cin, DEV, torch.float32)
        x.t.normal_()
        cs = Cout // 2 if glu else Cout
        r = K.new_view(B, H, H, cs, DEV, torch.float32)
        r.t.normal_()
        wm = torch.randn(Cout, Cin, generator=g, device=DEV) * Cin ** -0.5
        perm = pack.glu_perm(Cout // 2).to(DEV) if glu else None
        wf = pack.pack_matrix(wm, DEV, perm, torch.float32)
        wsp = pack.pack_matrix_split16(wm, DEV, perm)
        y = K.new_view(B, H, H, cs, DEV, torch.float32)

        def run(wt, tile):
            K.conv_gemm(x, wt, None, y, KH=1, epi=epi, res=r if res else None, tile=tile)
        t = timed({"exact": lambda: run(wf, 0), "exact128": lambda: run(wf, 128), "split": lambda: run(wsp, 0), "split128": lambda: run(wsp, 128)}, 40)
        fl = 2.0 * B * H * H * Cout * Cin
        print(f"1x1 {Cin}->{Cout} epi{epi} res={int(res)}: exact {t['exact'] * 1e3:6.1f} us (128-row {t['exact128'] * 1e3:6.1f}) | split auto "
              f"{t['split'] * 1e3:6.1f} us {fl / t['split'] / 1e9:6.1f} TF | split 128-row {t['split128'] * 1e3:6.1f} us | split / exact {t['split'] / t['exact']:.3f}", flush=True)
        del x, r, y, wf, wsp
    print("== conv_shortcut as a second source of conv2")
    for B, H, Cin, Cout, Cin2, gn in ((16, 768, 128, 128, 256, True), (16, 384, 256, 256, 512, False), (16, 96, 320, 320, 960, False),
                                      (16, 96, 320, 320, 640, False), (16, 48, 640, 640, 1920, False), (16, 48, 640, 640, 960, False)):
        h = K.new_view(B, H, H, Cin, DEV, stats=True)
        h.t.normal_()
        h.stats.t.normal_().abs_()
        h.stats.mark(0, Cin)
        x2 = K.new_view(B, H, H, Cin2, DEV)
        x2.t.normal_()
        w = pack.pack_matrix(torch.randn(Cout, 9 * Cin, generator=g, device=DEV) * (9 * Cin) ** -0.5, DEV)
        w2 = pack.pack_matrix(torch.randn(Cout, Cin2, generator=g, device=DEV) * Cin2 ** -0.5, DEV)
        gam, bet = torch.ones(Cin, device=DEV), torch.zeros(Cin, device=DEV)
        y = K.new_view(B, H, H, Cout, DEV, stats=True)
        sc = K.new_view(B, H, H, Cout, DEV)
        gnarg = (gam, bet, 1e-6, True, ws) if gn else None

        def fused():
            assert K.conv_gemm(h, w, None, y, KH=3, stats=True, gn=gnarg, x2=x2, w2=w2) is not None

        def separate():
            K.conv_gemm(x2, w2, None, sc, KH=1)
            K.conv_gemm(h, w, None, y, KH=3, stats=True, gn=gnarg, res=sc)
        fl = 2.0 * B * H * H * Cout * (9 * Cin + Cin2)
        n = min(100, max(5, int(20.0 / (fl / 800e9))))
        t = timed({"fused": fused, "separate": separate}, n)
        print(f"{H}^2 3x3 {Cin}->{Cout} + 1x1 {Cin2}->{Cout} gn={int(gn)}: one launch {t['fused']:7.3f} ms {fl / t['fused'] / 1e9:7.1f} TF   "
              f"two launches {t['separate']:7.3f} ms   ratio {t['fused'] / t['separate']:.3f}", flush=True)
        del h, x2, y, sc, w, w2


if __name__ == "__main__":
    main()
