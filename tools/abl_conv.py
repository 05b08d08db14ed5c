import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from onedc_amd import kernels as K, pack
DEV="cuda:0"
g = torch.Generator(device=DEV).manual_seed(0)
for name,B,H,W,Cin,Cout in [("512->512@192",16,192,192,512,512),("256->256@384",16,384,384,256,256)]:
    x = K.View(torch.randn((B*H*W+128)*Cin, generator=g, device=DEV).to(K.a16()), B,H,W,Cin)
    w = pack.pack_matrix(torch.randn(Cout, 9*Cin, generator=g, device=DEV)*(9*Cin)**-0.5, DEV)
    y = K.new_view(B,H,W,Cout,DEV)
    for _ in range(3): K.conv_gemm(x,w,None,y,KH=3,tile=256)
    torch.cuda.synchronize()
    e0,e1=torch.cuda.Event(enable_timing=True),torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10): K.conv_gemm(x,w,None,y,KH=3,tile=256)
    e1.record(); torch.cuda.synchronize()
    ms=e0.elapsed_time(e1)/10
    print(f"{name}: {ms:7.3f} ms  {2.0*B*H*W*Cout*9*Cin/ms/1e9:7.1f} TF")
