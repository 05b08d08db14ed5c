#!/usr/bin/env python3
"""Micro-benchmark of the exact-f32 GEMM path (conv_gemm_kernel<Tile<128,128>, F32>, v_mfma_f32_16x16x4_f32) on the
prior networks' shapes at 768x768, B = 16 (48 x 48 maps): TFLOP/s against the 157 TF f32 matrix-core peak."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from onedc_amd import kernels as K
from onedc_amd import pack

DEV = "cuda:0"
SHAPES = [("1x1 256->256 @48", 16, 48, 256, 256, K.EPI_LINEAR), ("1x1 256->1024 GLU @48", 16, 48, 256, 1024, K.EPI_GLU_LEAKY),
          ("1x1 512->256 @48", 16, 48, 512, 256, K.EPI_LINEAR), ("1x1 128->128 @48", 16, 48, 128, 128, K.EPI_LINEAR),
          ("1x1 128->128 @24", 16, 24, 128, 128, K.EPI_LINEAR), ("1x1 256->256 @96 (4x px)", 16, 96, 256, 256, K.EPI_LINEAR)]


def main():
    g = torch.Generator(device=DEV).manual_seed(0)
    for name, B, H, Cin, Cout, epi in SHAPES:
        x = K.View(torch.randn((B * H * H + 128) * Cin, generator=g, device=DEV), B, H, H, Cin)
        w = pack.pack_matrix(torch.randn(Cout, Cin, generator=g, device=DEV) * Cin ** -0.5, DEV,
                             pack.glu_perm(Cout // 2) if epi == K.EPI_GLU_LEAKY else None, torch.float32)
        cy = Cout // 2 if epi == K.EPI_GLU_LEAKY else Cout
        y = K.new_view(B, H, H, cy, DEV, torch.float32)
        for _ in range(3):
            K.conv_gemm(x, w, None, y, epi=epi)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        n = 30
        e0.record()
        for _ in range(n):
            K.conv_gemm(x, w, None, y, epi=epi)
        e1.record()
        torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / n
        fl = 2.0 * B * H * H * Cin * Cout
        tiles = ((B * H * H + 127) // 128) * ((Cout + 127) // 128)
        print(f"{name:28s} {ms * 1e3:8.1f} us  {fl / ms / 1e9:7.1f} TFLOP/s  ({tiles} workgroups, {tiles / 512:.2f} rounds of 2 per CU)")


if __name__ == "__main__":
    main()
