import sys; sys.path.insert(0,'/root/repo'); sys.path.insert(0,'/root/repo/tests')
import torch
from onedc_amd import kernels as K, pack
from onedc_amd.kernels import from_nchw, new_view
DEV="cuda:0"
def rnd(*shape, seed=0, scale=1.0):
    g = torch.Generator().manual_seed(seed)
    return (torch.randn(*shape, generator=g) * scale).to(torch.bfloat16).float()
x, w, b = rnd(2, 128, 32, 32, seed=141), rnd(256, 128, 3, 3, seed=142, scale=1 / 34), rnd(256, seed=143)
outs=[]
for tile in (128,129,256):
    y = new_view(2, 32, 32, 256, DEV, stats=True)
    K.conv_gemm(from_nchw(x.to(DEV)), pack.pack_matrix(pack.conv_matrix(w), DEV), b.to(DEV), y, KH=3, tile=tile, stats=True)
    torch.cuda.synchronize()
    outs.append((y.t[:2*32*32*256].clone(), y.stats.t.clone()))
for i,(o,s) in enumerate(outs[1:],1):
    print(i, "out equal", torch.equal(o,outs[0][0]), "stats equal", torch.equal(s,outs[0][1]), "max stat diff", (s-outs[0][1]).abs().max().item(), "nonfinite", (~torch.isfinite(s)).sum().item())
    d=(s-outs[0][1]).abs().view(32,64,2)
    print("  rows with diff", (d.amax(dim=(1,2))>0).nonzero().flatten().tolist()[:10], "quads", (d.amax(dim=(0,2))>0).nonzero().flatten().tolist()[:10])
