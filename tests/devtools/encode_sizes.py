"""Probe: model.encode at the ends of the size range (round trip through decode, determinism, alone == in batch)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from onedc_amd.model_sd15_with_codec_stage1 import SD15_1step_codec_stage1
DEV = "cuda:0"
m = SD15_1step_codec_stage1({"vae_attn_patch": 16}).load_synthetic(0).load_synthetic_encoder(0)
m.codec_model.update(force=True)
g = torch.Generator().manual_seed(3)
for (B, H, W) in [(1, 64, 64), (2, 50, 70), (3, 64, 128), (1, 192, 64), (2, 1536, 1536), (1, 2000, 1100)]:
    try:
        img = (torch.rand(B, 3, H, W, generator=g) * 2 - 1).to(DEV)
        Hp, Wp = (H + 63) // 64 * 64, (W + 63) // 64 * 64
        noise = torch.randn(B, Hp // 8, Wp // 8, 4, generator=g).to(DEV)
        t0 = time.time()
        blobs, bpp = m.encode(img, noise=noise)
        if B == 1:
            blobs, bpp = [blobs], [bpp]
        out = m.decode(stream=list(blobs))
        torch.cuda.synchronize()
        dt = time.time() - t0
        again, _ = m.encode(img[:1], noise=noise[:1])
        print(f"{B} x {H}x{W}: {[len(b) for b in blobs]} bytes, decoded {tuple(out.shape)}, finite {bool(torch.isfinite(out).all())}, "
              f"picture 0 alone -> same bytes {again == blobs[0]}, {dt:.2f} s", flush=True)
    except Exception as e:
        import traceback
        print(f"{B} x {H}x{W}: FAILED {type(e).__name__}: {e}", flush=True)
        traceback.print_exc(limit=8)
