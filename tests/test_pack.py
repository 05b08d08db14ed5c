"""Host-side weight packing (pure torch, CPU)."""
import torch
import torch.nn.functional as F

from onedc_amd import pack


def test_upsample_phase_matrix_is_the_upsampled_conv():
    """conv3x3(nearest x2 upsample(x)) == four 2x2 phase convolutions of x with the summed weights
    (csrc/gemm_conv_ups.hip): evaluated here with plain torch ops in fp64."""
    g = torch.Generator().manual_seed(0)
    C, Cin, H, W = 6, 5, 7, 9
    x = torch.randn(2, Cin, H, W, generator=g, dtype=torch.float64)
    w = torch.randn(C, Cin, 3, 3, generator=g, dtype=torch.float64)
    ref = F.conv2d(F.interpolate(x, scale_factor=2.0, mode="nearest"), w, padding=1)
    m = pack.upsample_phase_matrix(w.float()).double().view(4, C, 4, Cin)      # [phase][c][tap][cin]
    out = torch.zeros_like(ref)
    xp = F.pad(x, (1, 1, 1, 1))                                                 # window origin (y - 1, x - 1)
    for py in range(2):
        for px in range(2):
            acc = torch.zeros(2, C, H, W, dtype=torch.float64)
            for ty in range(2):
                for tx in range(2):
                    # tap (ty, tx) of phase (py, px) reads low-res pixel (y + py + ty - 1, x + px + tx - 1)
                    patch = xp[:, :, py + ty:py + ty + H, px + tx:px + tx + W]
                    acc += torch.einsum("oc,bchw->bohw", m[py * 2 + px, :, ty * 2 + tx], patch)
            out[:, :, py::2, px::2] = acc
    assert float((out - ref).abs().max()) < 1e-5


def test_glu_and_shuffle_permutations_are_permutations():
    assert sorted(pack.glu_perm(64).tolist()) == list(range(128))
    assert sorted(pack.shuffle_perm(64).tolist()) == list(range(64))
