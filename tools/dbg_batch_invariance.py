import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from onedc_amd import kernels as K, synth
from onedc_amd.kernels import View
from onedc_amd.model_sd15_with_codec_stage1 import SD15_1step_codec_stage1
DEV = "cuda:0"
model = SD15_1step_codec_stage1({"vae_attn_patch": 16}).load_synthetic(0).load_synthetic_encoder(0)
model.codec_model.update(force=True)
g = torch.Generator().manual_seed(44)
img = (torch.rand(2, 3, 120, 180, generator=g) * 2 - 1).to(DEV)
noise = torch.randn(2, 16, 24, 4, generator=g).to(DEV)

def run(im, nz):
    B = im.shape[0]
    iv = View(torch.zeros((B * 128 * 192 + K.SLACK_ROWS) * 32, dtype=torch.float32, device=DEV), B, 128, 192, 32)
    K.nchw_to_nhwc(im.contiguous(), iv.slice(0, 3))
    lat = model.vae_encoder.encode(iv, nz)
    enc = model.codec_model.enc
    # stages of the analysis transform
    patches = K.new_view(B, 16, 24, 192, DEV); K.patchify8(iv, patches)
    y, sem = enc(iv, lat)
    return dict(iv=iv.torch_nchw()[:, :3], lat=lat.torch_nchw(), y=y.torch_nchw(), sem=sem.torch_nchw())
a = run(img, noise)
b = run(img[1:2], noise[1:2].contiguous())
for k in a:
    d = (a[k][1:2] - b[k]).abs().max().item()
    print(k, "max abs diff", d, "equal", torch.equal(a[k][1:2], b[k]))
