"""Per-tile overhead of the 3x3 window kernel: time against the number of K-tiles at a fixed output tile count.
time = tiles * (a + 9 * nc * b)  ->  a (prologue + epilogue) and b (one K-tile) from a least-squares fit."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from onedc_amd import kernels as K, pack
DEV = "cuda:0"
g = torch.Generator(device=DEV).manual_seed(0)
B, H, W, Cout = 8, 768, 768, 128
pts = []
for Cin in (64, 128, 256, 512):
    x = K.View(torch.randn((B * H * W + 128) * Cin, generator=g, device=DEV).to(K.a16()), B, H, W, Cin)
    w = pack.pack_matrix(torch.randn(Cout, 9 * Cin, generator=g, device=DEV) * (9 * Cin) ** -0.5, DEV)
    y = K.new_view(B, H, W, Cout, DEV)
    for _ in range(3):
        K.conv_gemm(x, w, None, y, KH=3, tile=130)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10):
        K.conv_gemm(x, w, None, y, KH=3, tile=130)
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 10
    tiles_per_cu = B * H * W / 256 / 256
    us_tile = ms * 1e3 / tiles_per_cu
    pts.append((9 * Cin // 64, us_tile))
    print(f"Cin {Cin:4d}: {ms:7.3f} ms {2.0 * B * H * W * Cout * 9 * Cin / ms / 1e9:7.1f} TF  {us_tile:6.2f} us per tile per CU")
    del x, w, y
A = np.array([[1, k] for k, _ in pts], float)
a, b = np.linalg.lstsq(A, np.array([t for _, t in pts]), rcond=None)[0]
print(f"fit: a = {a:.2f} us per tile, b = {b:.3f} us per K-tile  (asymptote {2*128*256*64*256/b/1e6:.0f} TF)")
