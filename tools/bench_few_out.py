import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from onedc_amd import kernels as K, pack
DEV = "cuda:0"
B, H, C = 16, 768, 128
x = K.new_view(B, H, H, C, DEV, stats=True); x.t.normal_(); x.stats.t.normal_().abs_(); x.stats.mark(0, C)
w = pack.pack_matrix(torch.randn(3, 9 * C, device=DEV) * 0.03, DEV)
b = torch.zeros(3, device=DEV)
g, be = torch.ones(C, device=DEV), torch.zeros(C, device=DEV)
class WS:
    def __init__(self): self.t = {}
    def f32(self, n, k):
        if n not in self.t or self.t[n].numel() < k: self.t[n] = torch.empty(k, dtype=torch.float32, device=DEV)
        return self.t[n]
ws = WS()
y = K.new_view(B, H, H, 3, DEV, torch.float32)
def t(fn, n=10):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n
print("fused norm_out + conv_out (few-out kernel):", t(lambda: K.conv3x3_few_out(x, w, b, y, gn=(g, be, 1e-6, True, ws))))
print("few-out kernel without norm:", t(lambda: K.conv3x3_few_out(x, w, b, y)))
xn = K.new_view(B, H, H, C, DEV)
def old():
    K.group_norm(x, g, be, xn, 1e-6, True, ws)
    K.conv_gemm(xn, w, b, y, KH=3)
print("separate pass + 16x256 MFMA kernel:", t(old))
