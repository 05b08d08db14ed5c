"""Quality tooling: the two reference-free-of-networks metrics of src/test_quality.py.

    PSNR     -10 log10(mean((ref - recon)^2)) on [0,1] pictures                     (test_quality.py:234-235)
    MS-SSIM  pytorch_msssim.ms_ssim(ref, recon, data_range=1.)                      (:236; third-party, restated:
             5 scales, 11-tap Gaussian window sigma 1.5, weights .0448 .2856 .3001 .2363 .1333, 2x2 average
             pooling between scales, relu on the contrast terms -- parity unpinned)
LPIPS, DISTS, FID, KID and the Inception score need pretrained networks that do not exist offline.

    python -m onedc_amd.quality --ref_path REF_DIR --recon_path RECON_DIR [--out quality.csv]

This is evaluation tooling on plain torch (CPU or GPU), not part of the decode path.
"""
import argparse
import os

import numpy as np
import torch
import torch.nn.functional as F

MS_WEIGHTS = (0.0448, 0.2856, 0.3001, 0.2363, 0.1333)


def psnr(ref, recon):
    """ref, recon: [B,3,H,W] (or [3,H,W]) in [0,1]."""
    mse = torch.mean((ref.double() - recon.double()) ** 2)
    return float(-10.0 * torch.log10(mse))


def _gauss(size=11, sigma=1.5):
    x = torch.arange(size, dtype=torch.float32) - size // 2
    g = torch.exp(-(x ** 2) / (2 * sigma ** 2))
    return g / g.sum()


def _filter(x, win):
    c = x.shape[1]
    k = win.to(x)
    x = F.conv2d(x, k.view(1, 1, -1, 1).expand(c, 1, -1, 1), groups=c)
    return F.conv2d(x, k.view(1, 1, 1, -1).expand(c, 1, 1, -1), groups=c)


def _ssim_terms(x, y, win, data_range):
    c1, c2 = (0.01 * data_range) ** 2, (0.03 * data_range) ** 2
    mu1, mu2 = _filter(x, win), _filter(y, win)
    s1 = _filter(x * x, win) - mu1 * mu1
    s2 = _filter(y * y, win) - mu2 * mu2
    s12 = _filter(x * y, win) - mu1 * mu2
    cs = (2 * s12 + c2) / (s1 + s2 + c2)
    ssim = ((2 * mu1 * mu2 + c1) / (mu1 * mu1 + mu2 * mu2 + c1)) * cs
    return ssim.flatten(2).mean(-1), cs.flatten(2).mean(-1)          # per (image, channel)


def ms_ssim(ref, recon, data_range=1.0):
    """Multi-scale SSIM of [B,C,H,W] pictures (smaller side must exceed (11 - 1) * 2^4 = 160)."""
    assert ref.shape == recon.shape and min(ref.shape[-2:]) > 160, "MS-SSIM needs pictures larger than 160 pixels"
    x, y = ref.float(), recon.float()
    win = _gauss()
    mcs = []
    for i in range(len(MS_WEIGHTS)):
        ssim, cs = _ssim_terms(x, y, win, data_range)
        if i < len(MS_WEIGHTS) - 1:
            mcs.append(torch.relu(cs))
            pad = [s % 2 for s in x.shape[2:]]
            x = F.avg_pool2d(x, 2, padding=pad)
            y = F.avg_pool2d(y, 2, padding=pad)
    vals = torch.stack(mcs + [torch.relu(ssim)], 0)                   # (scales, B, C)
    w = torch.tensor(MS_WEIGHTS).to(vals).view(-1, 1, 1)
    return float(torch.prod(vals ** w, 0).mean())


def _load(path):
    from PIL import Image
    return torch.from_numpy(np.asarray(Image.open(path).convert("RGB"), dtype=np.float32) / 255.0).permute(2, 0, 1)[None]


def main(argv=None):
    import pandas as pd
    ap = argparse.ArgumentParser()
    ap.add_argument("--ref_path", required=True)
    ap.add_argument("--recon_path", required=True)
    ap.add_argument("--out", default=None)
    a = ap.parse_args(argv)
    rows = []
    for f in sorted(os.listdir(a.recon_path)):
        name = os.path.splitext(f)[0]
        cands = [os.path.join(a.ref_path, name + e) for e in (".png", ".jpg")]
        ref = next((c for c in cands if os.path.exists(c)), None)
        if ref is None or not f.endswith((".png", ".jpg")):
            continue
        r, x = _load(ref), _load(os.path.join(a.recon_path, f))
        row = {"name": name, "psnr": psnr(r, x)}
        if min(r.shape[-2:]) > 160:
            row["msssim"] = ms_ssim(r, x)
        rows.append(row)
    df = pd.DataFrame(rows)
    avg = df.mean(numeric_only=True).to_dict() if len(df) else {}
    if a.out:
        df.to_csv(a.out)
    print(avg)
    return avg


if __name__ == "__main__":
    main()
