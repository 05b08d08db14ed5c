"""Host logic that needs no GPU: which kernel family onedc_conv_gemm picks (onedc_conv_gemm_kernel_id is the dispatcher's own
answer, computed on the host from the parameters alone), and the head-dim layouts the attention kernels expect."""
import ctypes as C

import pytest
import torch

from onedc_amd import kernels as K
from onedc_amd import pack


def params(B=2, H=32, W=32, Cin=128, Cout=128, k=3, *, up=0, epi=K.EPI_LINEAR, tile=0, stats=0, gn=False, res=False,
           y_co=0, y_cs=None, out_f32=0, stride=1, in_f32=0):
    p = K.ConvParams()
    dummy = C.cast(C.create_string_buffer(64), C.c_void_p)      # never dereferenced: no launch happens here
    p.x = p.w = p.y = p.zeros = dummy
    p.res = dummy if res else None
    p.stats = dummy if stats else None
    p.stats_gran = stats or 4
    p.gn_ab = dummy if gn else None
    p.B, p.IH, p.IW, p.Cin, p.x_cs, p.x_co = B, H, W, Cin, Cin, 0
    p.OH, p.OW = (2 * H, 2 * W) if up else (H // stride, W // stride)
    p.Cout, p.y_cs, p.y_co = Cout, (y_cs or Cout), y_co
    p.r_cs, p.r_co = Cout, 0
    p.KH = p.KW = k
    p.stride, p.pad, p.up = stride, k // 2, up
    p.epi, p.out_f32, p.in_f32 = epi, out_f32, in_f32
    p.alpha, p.slope, p.batch, p.tile = 1.0, 0.01, 1, tile
    return p


def name(p, fmt="f16"):
    lib = K.lib(fmt)
    kid = lib.onedc_conv_gemm_kernel_id(C.byref(p))
    return lib.onedc_conv_kernel_name(kid).decode() if kid >= 0 else kid


@pytest.mark.parametrize("fmt", ["f16", "bf16"])
def test_window_layers_go_to_the_4wave_kernel_where_its_epilogue_applies(fmt):
    assert name(params(), fmt) == "win3x3_4w"
    assert name(params(B=1), fmt) == name(params(B=16), fmt) == "win3x3_4w"          # by layer shape, never by batch size
    assert name(params(Cin=192, Cout=320, stats=2, res=True), fmt) == "win3x3_4w"    # pair statistics, ragged last channel tile
    assert name(params(gn=True, Cin=512), fmt) == "win3x3_4w"                        # GroupNorm on load: (a, b) fit behind the ring
    # what it does not take stays on the 8-wave kernel: activations, fused upsample, f32 output, unaligned output slices
    assert name(params(epi=K.EPI_SILU), fmt) == "win3x3"
    assert name(params(up=1), fmt) == "win3x3"
    assert name(params(out_f32=1), fmt) == "win3x3"
    assert name(params(y_co=4, y_cs=256), fmt) == "win3x3"
    # forcing: 130 = 8 waves, 131 = 4 waves (rejected where it cannot run)
    assert name(params(tile=130), fmt) == "win3x3"
    assert name(params(tile=131), fmt) == "win3x3_4w"
    assert name(params(tile=131, epi=K.EPI_SILU), fmt) == -11


def test_layers_no_window_kernel_takes():
    assert name(params(H=24, W=24, Cin=1280, Cout=1280, B=16)) == "256x256"          # map is not whole 16x16 patches
    p12 = params(H=12, W=12, Cin=1280, Cout=1280, B=16)
    assert name(p12) == -15                                                          # a split-K shape without scratch is refused (ABI 5)
    p12.split_k = 1
    assert name(p12) == "64x128"                                                     # 2304 pixels: a small grid
    assert name(params(k=1, H=96, W=96, Cin=320, Cout=2560, B=16)) == "lin_128x128"  # K <= 384, wide output: three-stage ring, 3 workgroups per CU
    # round 5: gated layers (fewer than 32 K-tiles) on the 128 x 256 tile of the ring kernel, by shape and epilogue only
    for B in (1, 16):
        assert name(params(k=1, H=96, W=96, Cin=320, Cout=2560, B=B, epi=K.EPI_GEGLU)) == "lin_128x256"
        assert name(params(k=1, H=24, W=24, Cin=1280, Cout=10240, B=B, epi=K.EPI_GEGLU)) == "lin_128x256"
    assert name(params(k=1, H=24, W=24, Cin=5120, Cout=1280, B=16, epi=K.EPI_GEGLU)) == "256x256"      # long K stays on eight waves
    assert name(params(k=1, H=96, W=96, Cin=320, Cout=320, B=16, tile=136)) == -11                       # forced where it cannot run
    assert name(params(k=1, H=96, W=96, Cin=320, Cout=320, B=16)) == "64x128"        # ... narrow output: the 64-row tile
    assert name(params(k=1, H=48, W=48, Cin=640, Cout=5120, B=16)) == "128x128"      # short K: the 4-wave kernels
    assert name(params(k=1, H=24, W=24, Cin=1280, Cout=1280, B=16)) == "128x128"     # 20 K-tiles: four waves since round 4
    assert name(params(k=1, H=24, W=24, Cin=5120, Cout=1280, B=16)) == "256x256"     # 80 K-tiles: eight
    assert name(params(k=1, H=96, W=96, Cin=320, Cout=2560, B=16, tile=133)) == "lin_128x128"
    assert name(params(k=1, H=96, W=96, Cin=320, Cout=2560, B=16, tile=128)) == "128x128"
    assert name(params(k=1, H=48, W=48, Cin=256, Cout=256, B=16, in_f32=1, out_f32=1)).startswith("f32_")
    assert name(params(k=1, H=48, W=48, Cin=256, Cout=256, B=16, in_f32=2, out_f32=1)).startswith("f32s_")   # split-half operands
    assert name(params(Cout=3)) == "16x256"                                           # conv_out-like


def test_split_k_is_a_function_of_the_layer_shape():
    """3x3 layers on maps of <= 256 pixels with >= 160 K-tiles (the UNet's 12x12 level) are split into five K ranges (four from 96
    K-tiles on: the codec's 768-channel layers) whatever the batch size.  A caller that brings no scratch (or too little for this
    batch) is REFUSED with -15 since ABI 5 -- never served by the unsplit kernel, whose sums round differently (a
    fixed-size scratch would have made a picture's bits depend on the batch); split_k = 1 asks for the unsplit form."""
    lib = K.lib()
    for B in (1, 2, 16):
        p = params(H=12, W=12, Cin=1280, Cout=1280, B=B)
        need = lib.onedc_conv_splitk_ws_bytes(C.byref(p))
        assert need == 5 * B * 144 * 1280 * 4
        assert lib.onedc_conv_gemm_kernel_id(C.byref(p)) == -15                       # no scratch: refused, not run unsplit
        p.ws, p.ws_bytes = C.cast(C.create_string_buffer(64), C.c_void_p), need - 1
        assert lib.onedc_conv_gemm_kernel_id(C.byref(p)) == -15                       # too small for this batch: the same
        p.ws_bytes = need
        assert name(p) == "splitk_256"
        p.split_k = 1
        assert name(p) == "64x128"
    assert lib.onedc_conv_splitk_ws_bytes(C.byref(params(H=12, W=12, Cin=768, Cout=768, B=16))) == 4 * 16 * 144 * 768 * 4   # 108 K-tiles
    assert lib.onedc_conv_splitk_ws_bytes(C.byref(params(H=12, W=12, Cin=512, Cout=768, B=16))) == 0       # 72 K-tiles
    assert lib.onedc_conv_splitk_ws_bytes(C.byref(params(H=24, W=24, Cin=1280, Cout=1280, B=16))) == 0     # 576 pixels per image
    assert lib.onedc_conv_splitk_ws_bytes(C.byref(params(H=12, W=12, Cin=320, Cout=1280, B=16))) == 0       # 45 K-tiles
    assert lib.onedc_conv_splitk_ws_bytes(C.byref(params(k=1, H=12, W=12, Cin=5120, Cout=1280, B=16))) == 0  # 1x1


def test_bf16_operands_exist_in_the_4wave_window_kernel_only():
    """onedc_conv_params.x_bf16 (round 5: bf16 operands under f16 storage): the 4-wave window kernel multiplies them, with or
    without GroupNorm on load and a second source; every other kernel refuses the layer (-17) and the caller keeps f16 operands."""
    p = params(gn=True)
    p.x_bf16 = 1
    assert name(p) == "win3x3_4w"
    p = params(Cin=256, Cout=256, res=True, stats=4)
    p.x_bf16 = 1
    assert name(p) == "win3x3_4w"
    for q in (params(H=24, W=24, Cin=1280, Cout=1280, B=16), params(k=1, H=96, W=96, Cin=320, Cout=320, B=16), params(epi=K.EPI_SILU),
              params(up=1)):
        q.x_bf16, q.split_k = 1, 1
        assert name(q) == -17, name(q)


def test_bf16_operands_are_asked_for_with_the_final_parameters():
    """with gn_ab the 4-wave window kernel needs room for the image's (a, b) pairs (Cin <= 768); a wider layer
    normalises on load on the 8-wave kernel, which has no bf16-operand form -- asked with gn_ab SET the dispatcher says so (-17) and
    kernels.conv_gemm keeps f16 operands; asked before (the round-5 order) it answered for another launch.  The bf16 library
    multiplies bf16 everywhere and ignores the flag (include/onedc_hip.h)."""
    wide = params(gn=True, Cin=1024, Cout=128)
    assert name(wide) == "win3x3"                              # GroupNorm on load, 8 waves
    wide.x_bf16 = 1
    assert name(wide) == -17
    before = params(Cin=1024, Cout=128)                        # the same layer before gn_ab is attached: another answer
    before.x_bf16 = 1
    assert name(before) == "win3x3_4w"
    for q in (wide, params(k=1, H=96, W=96, Cin=320, Cout=320, B=16)):
        q.x_bf16 = 1
        assert name(q, "bf16") in ("win3x3", "128x128", "64x128", "lin_128x128")     # never refused for the flag


def test_second_source_question():
    """conv_shortcut as a second source: taken where the layer runs on the 4-wave window kernel, refused elsewhere (the caller
    then launches the shortcut itself)."""
    lib = K.lib()
    dummy = C.cast(C.create_string_buffer(64), C.c_void_p)

    def with_x2(p, Cin2=256, cs=None):
        p.x2, p.w2, p.Cin2, p.x2_cs, p.x2_co = dummy, dummy, Cin2, cs or Cin2, 0
        return p
    assert lib.onedc_conv_win_takes_x2(C.byref(with_x2(params()))) == 1
    assert name(with_x2(params())) == "win3x3_4w"
    assert lib.onedc_conv_win_takes_x2(C.byref(with_x2(params(gn=True)))) == 1                 # GroupNorm on load too
    assert lib.onedc_conv_win_takes_x2(C.byref(with_x2(params(H=12, W=12)))) == 0              # not whole patches
    assert lib.onedc_conv_win_takes_x2(C.byref(with_x2(params(epi=K.EPI_SILU)))) == 0          # 8-wave kernel's layer
    assert lib.onedc_conv_win_takes_x2(C.byref(with_x2(params(), Cin2=48))) == 0               # not whole 32-channel chunks
    assert lib.onedc_conv_win_takes_x2(C.byref(with_x2(params(B=1, H=3072, W=3072), Cin2=256))) == 0   # 4.8 GB per image of x2
    assert name(with_x2(params(H=12, W=12))) == -16


def test_window_kernels_refuse_images_beyond_their_32bit_window_offsets():
    """A window kernel addresses one image through a buffer descriptor of at most 0xfffff000 bytes with 32-bit offsets:
    128 channels at 4096x4096 (4 GiB per image) must go to a 64-bit-addressed GEMM, not wrap."""
    assert name(params(B=1, H=2048, W=2048, Cin=128, Cout=128)) == "win3x3_4w"          # 1 GiB per image: fine
    assert name(params(B=1, H=4096, W=4096, Cin=128, Cout=128)) not in ("win3x3_4w", "win3x3")
    assert name(params(B=1, H=4096, W=4096, Cin=128, Cout=128, tile=131)) == -11        # forcing it is refused
    assert K.lib().onedc_conv_win_takes(C.byref(params(B=1, H=4096, W=4096, Cin=128, Cout=128))) == 0


def test_upsample_phases():
    p = params(H=32, W=32, Cin=256, Cout=256, up=2)
    assert name(p) == "ups2x2_4w"
    p.tile = 130
    assert name(p) == "ups2x2"
    assert name(params(H=32, W=32, Cin=256, Cout=256, up=2, epi=K.EPI_SILU)) == "ups2x2"
    assert name(params(H=24, W=24, Cin=256, Cout=256, up=2)) == -12                   # low-resolution map is not whole patches


def test_groupnorm_on_load_question():
    lib = K.lib()
    assert lib.onedc_conv_win_takes(C.byref(params())) == 1
    assert lib.onedc_conv_win_takes(C.byref(params(k=1))) == 0
    assert lib.onedc_conv_win_takes(C.byref(params(Cin=1024, Cout=128))) == 1         # too many (a, b) pairs for two workgroups
    assert name(params(Cin=1024, Cout=128, gn=True)) == "win3x3"                      # per CU: the 8-wave kernel has the room
    assert lib.onedc_conv_win_takes(C.byref(params(up=1))) == 0                       # the fused-upsample variant cannot


def test_head_layouts():
    assert pack.HEAD_PAD[80] == (128, 96) and pack.HEAD_PAD[40] == (64, 48)
    assert pack.head_positions(40, 64).tolist() == list(range(40))
    assert pack.head_positions(80, 128).tolist() == list(range(40)) + list(range(64, 104))
    assert pack.head_positions(80, 96).tolist() == list(range(40)) + list(range(48, 88))
    w = torch.arange(2 * 80 * 3, dtype=torch.float32).view(160, 3)
    rows = pack.pad_heads_rows(w, 2, 80, 128)
    assert rows.shape == (256, 3)
    assert torch.equal(rows[64:104], w[40:80]) and torch.equal(rows[128:168], w[80:120]) and float(rows[40:64].abs().max()) == 0
    cols = pack.pad_heads_cols(w.t().contiguous(), 2, 80, 96)
    assert cols.shape == (3, 192)
    assert torch.equal(cols[:, 48:88], w.t()[:, 40:80]) and float(cols[:, 88:96].abs().max()) == 0
    lib = K.lib()
    assert lib.onedc_flash_attn_vt_is_bf16(128, 96, 144) == 1 and lib.onedc_flash_attn_vt_is_bf16(128, 96, 16) == 0
    assert lib.onedc_flash_attn_vt_is_bf16(160, 160, 576) == 0


def test_fused_cross_attention_block_says_which_shapes_it_takes():
    """onedc_xattn_block_ok (csrc/xattn_block.hip, round 6) answers on the host, without a launch: the UNet's first level (C = 320, 8
    heads of 40, a context of 16 .. 160 tokens in whole 16s, pixel counts in whole 64s, channel slices inside their rows, no aliasing,
    the LayerNorm's affine given whole or folded away) and nothing else -- the caller keeps the four separate launches otherwise."""
    lib = K.lib()
    d1, d2 = C.cast(C.create_string_buffer(64), C.c_void_p), C.cast(C.create_string_buffer(64), C.c_void_p)

    def xp(**kw):
        p = K.XAttnParams()
        p.h, p.y = d1, d2
        p.ln_g = p.ln_b = p.wq = p.k = p.vt = p.wo = p.bo = d1
        p.B, p.HW, p.C, p.heads, p.d, p.Nk, p.Nk_pad, p.k_rs = 16, 96 * 96, 320, 8, 40, 144, 192, 16 * 320
        p.h_cs, p.h_co, p.y_cs, p.y_co, p.vt_bf16 = 320, 0, 320, 0, 1
        p.ln_eps, p.scale, p.vt_bs = 1e-5, 40 ** -0.5, 16 * 320 * 192
        for k, v in kw.items():
            setattr(p, k, v)
        return p
    ok = lambda p: lib.onedc_xattn_block_ok(C.byref(p))          # noqa: E731
    assert ok(xp()) == 1
    assert ok(xp(ln_g=None, ln_b=None, bq=d1)) == 1                       # the affine folded into to_q by the host
    assert ok(xp(Nk=64, Nk_pad=64)) == 1 and ok(xp(Nk=16, Nk_pad=64)) == 1 and ok(xp(Nk=160, Nk_pad=192)) == 1
    assert ok(xp(HW=64 * 64)) == 1 and ok(xp(h_cs=640, h_co=320, y_cs=960, y_co=640)) == 1
    for bad in (dict(C=640), dict(heads=4), dict(d=80), dict(Nk=150), dict(Nk=176, Nk_pad=192), dict(Nk=0), dict(Nk_pad=128),
                dict(HW=12 * 12), dict(h_co=4), dict(y_cs=324), dict(h_cs=320, h_co=8), dict(y_co=-8), dict(k_rs=312),
                dict(vt_bs=320 * 192 - 8), dict(ln_b=None), dict(y=d1), dict(wq=None), dict(k=None), dict(B=0)):
        assert ok(xp(**bad)) == 0, bad
    assert lib.onedc_xattn_block(C.byref(xp(C=640)), None) == -1           # a refused shape is a status, not a launch
