"""Quality tooling on CPU: closed forms of PSNR (reference test_quality.py:234-235) and the basic properties of
the restated MS-SSIM."""
import math

import torch

from onedc_amd import quality as Q


def test_psnr_closed_forms():
    a = torch.zeros(1, 3, 8, 8)
    assert abs(Q.psnr(a, a + 0.1) - 20.0) < 1e-5                    # mse 0.01
    assert abs(Q.psnr(a, a + 1 / 255) - 20 * math.log10(255)) < 1e-4
    g = torch.Generator().manual_seed(0)
    x, y = torch.rand(2, 3, 16, 16, generator=g), torch.rand(2, 3, 16, 16, generator=g)
    assert abs(Q.psnr(x, y) - float(-10 * torch.log10(((x - y) ** 2).mean()))) < 1e-5


def test_ms_ssim_properties():
    g = torch.Generator().manual_seed(1)
    yy, xx = torch.meshgrid(torch.linspace(0, 1, 192), torch.linspace(0, 1, 224), indexing="ij")
    img = (0.5 + 0.3 * torch.sin(12 * xx) * torch.cos(9 * yy))[None, None].repeat(1, 3, 1, 1)
    assert abs(Q.ms_ssim(img, img) - 1.0) < 1e-6
    n1 = (img + 0.02 * torch.randn(img.shape, generator=g)).clamp(0, 1)
    n2 = (img + 0.10 * torch.randn(img.shape, generator=g)).clamp(0, 1)
    s1, s2 = Q.ms_ssim(img, n1), Q.ms_ssim(img, n2)
    assert 0.0 < s2 < s1 < 1.0
    assert abs(Q.ms_ssim(n1, img) - s1) < 1e-6                      # symmetric
