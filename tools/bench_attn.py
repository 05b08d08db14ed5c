#!/usr/bin/env python3
"""Micro-benchmark of onedc_flash_attn on the UNet's attention shapes at 768x768, B=16 (random data)."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from onedc_amd import kernels as K
from onedc_amd import pack

DEV = "cuda:0"
SHAPES = [("self d40 N9216", 40, 9216, 9216), ("cross d40 N9216 x144", 40, 9216, 144),
          ("self d80 N2304", 80, 2304, 2304), ("cross d80 N2304 x144", 80, 2304, 144),
          ("self d160 N576", 160, 576, 576), ("self d160 N144", 160, 144, 144)]


def main():
    B, H = 16, 8
    g = torch.Generator(device=DEV).manual_seed(0)
    for name, d, Nq, Nk in SHAPES:
        DQ, DV = pack.HEAD_PAD[d]
        Nk_pad = (Nk + 63) // 64 * 64
        q = torch.randn(B * Nq * H * DQ, generator=g, device=DEV).to(K.a16())
        k = torch.randn(B * Nk * H * DQ, generator=g, device=DEV).to(K.a16())
        vt = torch.randn(B * H * DV * Nk_pad, generator=g, device=DEV).to(K.flash_vt_dtype(DQ, DV, Nk))
        o = torch.empty(B * Nq * H * DV, dtype=K.a16(), device=DEV)
        for _ in range(2):
            K.flash_attn(q, k, vt, o, B, H, Nq, Nk, Nk_pad, DQ, DV, d ** -0.5)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        n = 5
        e0.record()
        for _ in range(n):
            K.flash_attn(q, k, vt, o, B, H, Nq, Nk, Nk_pad, DQ, DV, d ** -0.5)
        e1.record()
        torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / n
        fl = 4.0 * B * H * Nq * Nk * d
        print(f"{name:24s} {ms:8.3f} ms  {fl / ms / 1e9:8.1f} TFLOP/s (algorithmic, d={d})")


if __name__ == "__main__":
    main()
