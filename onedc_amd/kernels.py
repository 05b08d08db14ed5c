"""ctypes signatures of libonedc_hip.so + thin torch-tensor wrappers.

PyTorch is plumbing here: device memory (`tensor.data_ptr()`), the current HIP stream and
nothing else.  Every wrapper launches on torch's current stream and raises on a non-zero status.
There is no CPU or eager fallback: without the built library `_native.hip()` raises.
"""
import contextlib
import ctypes as C
import os
import threading

import torch

from . import _native

# ---- 16-bit activation / weight format ------------------------------------------------------------------------------
# The kernel library exists twice (csrc/common.h): "bf16" and "f16" (IEEE half: 3 more mantissa bits in every stored
# activation, packed weight and MFMA operand at the same speed; finite range +-65504).  The format is a property of the
# CALLING CONTEXT (thread-local): layers pack their weights and allocate their outputs in the format that is current when
# they are built / called, and `lib()` hands out the matching library.
DEFAULT_ACT = os.environ.get("ONEDC_ACT", "f16")
_tls = threading.local()


def act():
    return getattr(_tls, "act", None) or DEFAULT_ACT


def a16():
    """torch dtype of 16-bit activations and packed weights in the current format."""
    return torch.float16 if act() == "f16" else torch.bfloat16


def set_act(name):
    if name not in ("bf16", "f16"):
        raise ValueError(f"activation format must be 'bf16' or 'f16', not {name!r}")
    _tls.act = name


def in_own_format(method):
    """Decorator: run a method under the activation format its object was built in (`self.act`)."""
    import functools

    @functools.wraps(method)
    def wrapper(self, *args, **kwargs):
        with act_format(self.act):
            return method(self, *args, **kwargs)
    return wrapper


@contextlib.contextmanager
def act_format(name):
    prev = getattr(_tls, "act", None)
    set_act(name)
    try:
        yield
    finally:
        _tls.act = prev

EPI_LINEAR, EPI_LEAKY, EPI_SILU, EPI_GLU_LEAKY, EPI_GEGLU = range(5)

_P, _I, _L, _F, _D = C.c_void_p, C.c_int, C.c_int64, C.c_float, C.c_double


class ConvParams(C.Structure):
    _fields_ = [
        ("x", _P), ("w", _P), ("bias", _P), ("res", _P), ("y", _P), ("zeros", _P), ("stats", _P),
        ("B", C.c_int32), ("IH", C.c_int32), ("IW", C.c_int32), ("Cin", C.c_int32),
        ("x_cs", C.c_int32), ("x_co", C.c_int32),
        ("OH", C.c_int32), ("OW", C.c_int32), ("Cout", C.c_int32), ("y_cs", C.c_int32), ("y_co", C.c_int32),
        ("r_cs", C.c_int32), ("r_co", C.c_int32),
        ("KH", C.c_int32), ("KW", C.c_int32), ("stride", C.c_int32), ("pad", C.c_int32), ("up", C.c_int32),
        ("epi", C.c_int32), ("out_f32", C.c_int32), ("pixel_shuffle", C.c_int32),
        ("alpha", _F), ("slope", _F),
        ("batch", C.c_int32), ("tile", C.c_int32), ("stats_gran", C.c_int32), ("patch16", C.c_int32),
        ("in_f32", C.c_int32),
        ("x_bs", _L), ("w_bs", _L), ("y_bs", _L), ("r_bs", _L),
        ("gn_ab", _P), ("gn_silu", C.c_int32), ("out_bf16", C.c_int32),
        ("ws", _P), ("ws_bytes", _L), ("split_k", C.c_int32),
        ("x2", _P), ("w2", _P), ("Cin2", C.c_int32), ("x2_cs", C.c_int32), ("x2_co", C.c_int32),
        ("x_bf16", C.c_int32),
    ]


class DirectConvParams(C.Structure):
    _fields_ = [
        ("x", _P), ("w", _P), ("bias", _P), ("y", _P),
        ("B", C.c_int32), ("H", C.c_int32), ("W", C.c_int32), ("Cin", C.c_int32), ("x_cs", C.c_int32),
        ("x_co", C.c_int32), ("Cout", C.c_int32), ("y_cs", C.c_int32), ("y_co", C.c_int32), ("K", C.c_int32),
        ("x_f32", C.c_int32), ("out_f32", C.c_int32), ("epi", C.c_int32),
        ("slope", _F), ("in_scale", _F),
    ]


class XAttnParams(C.Structure):
    """onedc_xattn_params of include/onedc_hip.h."""
    _fields_ = [
        ("h", _P), ("ln_g", _P), ("ln_b", _P), ("wq", _P), ("k", _P), ("vt", _P), ("wo", _P), ("bo", _P), ("y", _P),
        ("B", C.c_int32), ("HW", C.c_int32), ("C", C.c_int32), ("heads", C.c_int32), ("d", C.c_int32), ("Nk", C.c_int32),
        ("Nk_pad", C.c_int32), ("k_rs", C.c_int32),
        ("h_cs", C.c_int32), ("h_co", C.c_int32), ("y_cs", C.c_int32), ("y_co", C.c_int32), ("vt_bf16", C.c_int32),
        ("ln_eps", _F), ("scale", _F),
        ("vt_bs", _L), ("bq", _P),
    ]


SIGNATURES = {
    "onedc_conv_gemm": (_I, [C.POINTER(ConvParams), _P]),
    "onedc_conv_win_takes": (_I, [C.POINTER(ConvParams)]),
    "onedc_conv_win_takes_x2": (_I, [C.POINTER(ConvParams)]),
    "onedc_conv_splitk_ws_bytes": (_L, [C.POINTER(ConvParams)]),
    "onedc_conv_gemm_kernel_id": (_I, [C.POINTER(ConvParams)]),
    "onedc_conv_kernel_name": (C.c_char_p, [_I]),
    "onedc_direct_conv": (_I, [C.POINTER(DirectConvParams), _P]),
    "onedc_conv3x3_few_out": (_I, [_P, _P, _P, _P, _P, _I, _I, _I, _I, _I, _I, _I, _I, _I, _I, _I, _P]),
    "onedc_dwconv3x3": (_I, [_P, _P, _P, _P, _I, _I, _I, _I, _I, _I, _I, _I, _P]),
    "onedc_dwconv3x3_f32": (_I, [_P, _P, _P, _P, _I, _I, _I, _I, _I, _I, _I, _I, _P]),
    "onedc_gn_chunks": (_I, [_I]),
    "onedc_gn_stats": (_I, [_P, _P, _P, _P, _P, _I, _I, _I, _I, _I, _F, _P]),
    "onedc_scale_shift_act": (_I, [_P, _P, _P, _I, _I, _I, _I, _I, _I, _I, _I, _P]),
    "onedc_scale_shift_act_fmt": (_I, [_P, _P, _P, _I, _I, _I, _I, _I, _I, _I, _I, _I, _P]),
    "onedc_gn_from_stats": (_I, [_P, _P, _P, _P, _I, _I, _I, _I, _I, _F, _I, _P]),
    "onedc_layernorm": (_I, [_P, _P, _P, _P, _L, _I, _F, _P]),
    "onedc_softmax_rows": (_I, [_P, _P, _L, _I, _I, _P]),
    "onedc_flash_attn": (_I, [_P, _P, _P, _P, _I, _I, _I, _I, _I, _I, _I, _F, _P]),
    "onedc_flash_attn_strided": (_I, [_P, _P, _P, _P, _I, _I, _I, _I, _I, _I, _I, _F, _I, _I, _L, _P]),
    "onedc_flash_attn_vt_is_bf16": (_I, [_I, _I, _I]),
    "onedc_xattn_block_ok": (_I, [C.POINTER(XAttnParams)]),
    "onedc_xattn_block": (_I, [C.POINTER(XAttnParams), _P]),
    "onedc_fsq_codes": (_I, [_P, _P, _L, _P]),
    "onedc_fsq_quantize": (_I, [_P, _I, _I, _I, _P, _L, _P]),
    "onedc_nchw_to_nhwc": (_I, [_P, _P, _I, _I, _I, _I, _I, _I, _I, _I, _I, _I, _P]),
    "onedc_patchify8": (_I, [_P, _I, _P, _I, _I, _I, _P]),
    "onedc_latent_sample": (_I, [_P, _P, _P, _L, _F, _P]),
    "onedc_prior_indexes": (_I, [_P, _I, _P, _I, _I, _I, _I, _I, _P]),
    "onedc_prior_update_dec": (_I, [_P, _I, _P, _P, _I, _I, _I, _I, _I, _I, _I, _P]),
    "onedc_prior_update_enc": (_I, [_P, _I, _P, _P, _P, _I, _I, _I, _I, _I, _I, _I, _P]),
    "onedc_prior_update_zonly": (_I, [_P, _I, _P, _I, _I, _I, _I, _I, _I, _I, _P]),
    "onedc_x0_from_noise": (_I, [_P, _P, _P, _L, _D, _P]),
    "onedc_image_out": (_I, [_P, _I, _I, _P, _I, _I, _I, _I, _I, _I, _P]),
    "onedc_image_out_checked": (_I, [_P, _I, _I, _P, _I, _I, _I, _I, _I, _I, _P, _P]),
    "onedc_image_out_u8": (_I, [_P, _I, _I, _P, _I, _I, _I, _I, _I, _I, _P, _P]),
    "onedc_copy_channels": (_I, [_P, _I, _I, _I, _P, _I, _I, _I, _L, _I, _P]),
    "onedc_window_partition": (_I, [_P, _P, _I, _I, _I, _I, _I, _I, _I, _I, _I, _I, _I, _P]),
    "onedc_timestep_embedding": (_I, [_P, _P, _I, _I, _P]),
    "onedc_mxfp8_roundtrip": (_I, [_P, _P, _L, _I, _I, _I, _I, _I, _P]),
    "onedc_ref_conv1x1": (_I, [_P, _I, _I, _P, _P, _P, _I, _I, _P, _I, _I, _L, _I, _I, _I, _I, _I, _F, _F, _I, _I, _I, _P]),
    "onedc_ref_conv3x3": (_I, [_P, _I, _I, _P, _P, _P, _I, _I, _P, _I, _I, _I, _I, _I, _I, _I, _I, _F, _I, _I, _P]),
    "onedc_ref_conv3x3_im2col": (_I, [_P, _I, _I, _P, _P, _P, _I, _I, _P, _I, _I, _I, _I, _I, _I, _I, _I, _I, _I, _I, _F, _P]),
    "onedc_ref_prior_indexes": (_I, [_P, _P, _I, _P, _I, _I, _I, _I, _I, _P]),
    "onedc_add": (_I, [_P, _P, _P, _L, _P]),
    "onedc_hip_version": (C.c_char_p, []),
    "onedc_hip_abi_version": (_I, []),
    "onedc_hip_act_format": (C.c_char_p, []),
    "onedc_hip_build_id": (C.c_char_p, []),
}
HIP_SYMBOLS = list(SIGNATURES)

_libs = {}


def attach_signatures(l):
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(l, name)
        fn.restype = res
        fn.argtypes = args
    return l


ABI_VERSION = 5        # ONEDC_HIP_ABI_VERSION of include/onedc_hip.h this binding was written against (tests/test_abi.py compares)


def lib(fmt=None):
    fmt = fmt or act()
    l = _libs.get(fmt)
    if l is None:
        l = attach_signatures(_native.hip(fmt))
        check_library_format(fmt, l.onedc_hip_act_format(), l.onedc_hip_version())
        if l.onedc_hip_abi_version() != ABI_VERSION:          # a symbol-compatible break is worse than a link error
            raise _native.NativeLibraryError(f"the kernel library speaks ABI {l.onedc_hip_abi_version()} ({l.onedc_hip_version().decode()}), "
                                             f"this binding ABI {ABI_VERSION}")
        path = _native.HIP_SO if fmt == "bf16" else _native.HIP_F16_SO
        if not os.environ.get("ONEDC_HIP_SO" if fmt == "bf16" else "ONEDC_HIP_F16_SO"):    # (experiment builds: no id check)
            _native.check_build_id(path, l.onedc_hip_build_id().decode(), "hip")
        _libs[fmt] = l
    return l


def check_library_format(fmt, reported, version=b""):
    """The library loaded for activation format `fmt` must have been BUILT for it: an f16 tensor read as bf16 is silent
    garbage.  Exact token comparison (b"f16" is a substring of b"bf16")."""
    if reported != fmt.encode():
        raise _native.NativeLibraryError(f"the kernel library loaded for {fmt!r} activations was built for "
                                         f"{reported.decode()!r} ({version.decode()})")


class KernelError(RuntimeError):
    pass


class HalfRangeError(FloatingPointError):
    """A decoded picture holds NaN / Inf: an activation left the IEEE-half range (act_dtype "f16", +-65504; the VAE stream
    is stored x 2^-6) somewhere upstream.  Raised by model.decode() / DecodePipeline.map() instead of returning the
    clamped garbage; `.images` lists the batch positions.  The fix is in the message: act_dtype: bf16."""

    def __init__(self, images, act):
        self.images = list(images)
        super().__init__(f"decoded picture(s) {self.images} of the batch hold non-finite values "
                         f"({act} activations: half-precision range exceeded); set act_dtype: bf16 in the config")


def _ok(rc, name):
    if rc != 0:
        raise KernelError(f"{name} failed with status {rc}")


def stream():
    return torch.cuda.current_stream().cuda_stream


_zeros = {}


def zeros_page(device):
    key = str(device)
    if key not in _zeros:
        _zeros[key] = torch.zeros(1024, dtype=torch.uint8, device=device)
    return _zeros[key]


def _ptr(t):
    return None if t is None else t.data_ptr()


class Stats:
    """GroupNorm partial sums of a whole NHWC buffer, written by the epilogues of the convs that
    produce its channel slices: float2 [B*H*W/64][cs/gran], gran = 4 channels per entry, or 2 when the
    buffer's GroupNorm groups (cs/32 channels) are not multiples of 4 (320, 960 channels).  `have` records
    which entries a producer has filled since the buffer was made."""
    __slots__ = ("t", "have", "gran")

    def __init__(self, B, H, W, cs, device, gran=None):
        self.gran = gran or (4 if (cs // 32) % 4 == 0 or cs % 32 else 2)
        self.t = torch.empty((B * H * W // 64) * (cs // self.gran) * 2, dtype=torch.float32, device=device)
        self.have = [False] * (cs // self.gran)

    @staticmethod
    def possible(H, W, cs):
        return (H * W) % 64 == 0 and cs % 4 == 0

    def mark(self, co, C):
        for q in range(co // self.gran, (co + C) // self.gran):
            self.have[q] = True

    def covers(self, co, C):
        g = self.gran
        return co % 4 == 0 and C % 4 == 0 and all(self.have[co // g:(co + C) // g])


class View:
    """A channel slice [co, co+C) of an NHWC buffer with `cs` channels per pixel."""
    __slots__ = ("t", "B", "H", "W", "C", "cs", "co", "stats")

    def __init__(self, t, B, H, W, C, cs=None, co=0, stats=None):
        self.t, self.B, self.H, self.W, self.C = t, B, H, W, C
        self.cs = C if cs is None else cs
        self.co = co
        self.stats = stats         # Stats of the underlying buffer (shared by all its slices) or None

    def want_stats(self, gran=None):
        """Attach a Stats holder to this buffer (no-op if the map is not a multiple of 64 pixels).
        gran: 2 / 4 channels per entry; default: 4 unless this buffer's own group size needs pairs."""
        if self.stats is None and Stats.possible(self.H, self.W, self.cs):
            self.stats = Stats(self.B, self.H, self.W, self.cs, self.t.device, gran)
        return self

    @property
    def ptr(self):
        return self.t.data_ptr()

    @property
    def f32(self):
        return self.t.dtype == torch.float32

    def slice(self, co, C):
        return View(self.t, self.B, self.H, self.W, C, self.cs, self.co + co, self.stats)

    def batch_slice(self, b0, b1):
        """Images [b0, b1) of this view (same buffer, no copy; the statistics side array does not carry over)."""
        off = b0 * self.H * self.W * self.cs
        return View(self.t[off:], b1 - b0, self.H, self.W, self.C, self.cs, self.co, None)

    def torch_nchw(self):
        """fp32 NCHW copy, for tests."""
        n = self.B * self.H * self.W * self.cs
        v = self.t[:n].view(self.B, self.H, self.W, self.cs)[..., self.co:self.co + self.C]
        return v.permute(0, 3, 1, 2).float().contiguous()


SLACK_ROWS = 128   # a GEMM may read (never use) up to one tile of rows past its "weight" operand


def new_view(B, H, W, C, device, dtype=None, cs=None, stats=False):
    """dtype None: 16-bit activations in the current format (a16())."""
    cs = C if cs is None else cs
    dtype = dtype or a16()
    v = View(torch.empty((B * H * W + SLACK_ROWS) * cs, dtype=dtype, device=device), B, H, W, C, cs, 0)
    return v.want_stats(stats if stats in (2, 4) else None) if stats else v


def from_nchw(x, dtype=None, cs=None, pad_to=None):
    """torch NCHW tensor -> NHWC View (optionally zero-padded to `pad_to` channels)."""
    dtype = dtype or a16()
    B, Cc, H, W = x.shape
    Cp = pad_to or Cc
    cs = cs or Cp
    t = torch.zeros(B * H * W + SLACK_ROWS, cs, dtype=dtype, device=x.device)
    t[:B * H * W].view(B, H, W, cs)[..., :Cc] = x.permute(0, 2, 3, 1).to(dtype)
    return View(t.reshape(-1), B, H, W, Cp, cs, 0)


SPLIT_K = int(os.environ.get("ONEDC_SPLIT_K", "0"))       # onedc_conv_params.split_k of every launch: 0 = the dispatcher decides by layer shape, 1 = never (tests, A/B)
_splitk_ws = {}


def splitk_workspace(device, nbytes):
    """Scratch for the partial sums of a split-K launch: one buffer per (device, HIP stream), grown on demand."""
    key = (str(device), stream())
    t = _splitk_ws.get(key)
    if t is None or t.numel() < nbytes:
        t = torch.empty(nbytes, dtype=torch.uint8, device=device)
        _splitk_ws[key] = t
    return t


PROFILE = None   # bench.py sets this to a list: every conv_gemm launch is then bracketed by HIP events
PROFILE_ONLY = None   # ... or, with a set of kernel-family names here, only the launches the dispatcher sends to those


PROFILE_EVENT_KW = {}     # {"external": True} when the profiled launches are captured into a HIP graph


def conv_kernel_name(p):
    """The kernel family onedc_conv_gemm runs these parameters on: the C dispatcher's own answer."""
    kid = lib().onedc_conv_gemm_kernel_id(C.byref(p))
    if kid < 0:
        raise KernelError(f"onedc_conv_gemm would reject these parameters with status {kid}")
    return lib().onedc_conv_kernel_name(kid).decode()


def conv_gemm(x, w, bias, y, *, KH=1, stride=1, up=0, epi=EPI_LINEAR, slope=0.01, alpha=1.0, res=None,
              pixel_shuffle=0, Cin=None, Cout=None, batch=1, x_bs=0, w_bs=0, y_bs=0, r_bs=0, flop_scale=1.0,
              tile=0, stats=False, pad=None, gn=None, out_bf16=False, x2=None, w2=None, w_bf16=None):
    """x, y, res: View; w: packed bf16 weight tensor; bias: f32 tensor or None.
    x2, w2: a second source on the output grid and its packed 1x1 weights (pre-divided by alpha), accumulated behind the 3x3
    taps: y = alpha * (conv(x) + conv1x1(x2)) + bias + res.  Only the 4-wave window kernel does it; when the dispatcher
    would not, NOTHING is launched and None is returned (the caller runs the two layers separately).
    gn = (gamma, beta, eps, silu, ws): the conv reads GroupNorm(x) (+SiLU).  On the 3x3 window kernel the
    normalisation is applied while the operand is staged (x is read raw, the normalised tensor never exists);
    any other kernel gets it from the separate scale/shift pass, with identical values.
    w_bf16 (with gn, IEEE-half format only): the same packed weights in bf16 -- the layer then multiplies bf16 OPERANDS (the
    normalised tensor is an intermediate only this conv reads: it is normalised to bf16 on load, or stored as bf16 by the pass)
    while y, res and the statistics stay f16 (onedc_conv_params.x_bf16; BF16_OPERANDS).  Only where the 4-wave window kernel
    takes the layer; silently the f16 operands otherwise."""
    p = ConvParams()
    p.x, p.w, p.bias, p.res, p.y = x.ptr, w.data_ptr(), _ptr(bias), (res.ptr if res is not None else None), y.ptr
    p.zeros = zeros_page(x.t.device).data_ptr()
    p.B, p.IH, p.IW = x.B, x.H, x.W
    p.Cin = Cin if Cin is not None else x.C
    p.x_cs, p.x_co = x.cs, x.co
    if pixel_shuffle:
        p.OH, p.OW = y.H // 2, y.W // 2
    else:
        p.OH, p.OW = y.H, y.W
    glu = epi in (EPI_GLU_LEAKY, EPI_GEGLU)
    p.Cout = Cout if Cout is not None else (y.C * 2 if glu else (y.C * 4 if pixel_shuffle else y.C))
    p.y_cs, p.y_co = y.cs, y.co
    if res is not None:
        p.r_cs, p.r_co = res.cs, res.co
    p.KH = p.KW = KH
    p.stride, p.pad, p.up = stride, (KH // 2 if pad is None else pad), up
    p.epi, p.out_f32, p.pixel_shuffle = epi, int(y.f32), pixel_shuffle
    p.in_f32 = 2 if (x.f32 and w.dtype == torch.float16) else int(x.f32)       # f16 "weights" of an f32 layer: the split form
    p.out_bf16 = int(bool(out_bf16))
    p.tile = tile
    if stats and y.stats is not None and not pixel_shuffle and batch <= 1 and y.co % 4 == 0 and y.C % 4 == 0:
        p.stats = y.stats.t.data_ptr()       # the epilogue leaves GroupNorm partial sums for the consumer
        p.stats_gran = y.stats.gran
        y.stats.mark(y.co, y.C)
    if x.f32 and (w.dtype not in (torch.float32, torch.float16) or (res is not None and not res.f32)):
        raise KernelError("f32 conv_gemm needs f32 (or split-half) weights and an f32 residual")
    p.alpha, p.slope = alpha, slope
    p.batch, p.x_bs, p.w_bs, p.y_bs, p.r_bs = batch, x_bs, w_bs, y_bs, r_bs
    p.split_k = SPLIT_K
    if x2 is not None:
        p.x2, p.w2, p.Cin2, p.x2_cs, p.x2_co = x2.ptr, w2.data_ptr(), x2.C, x2.cs, x2.co
        if (x2.B, x2.H, x2.W) != (y.B, y.H, y.W) or not lib().onedc_conv_win_takes_x2(C.byref(p)):
            return None
    if gn is not None:
        gamma, beta, eps, silu, ws = gn
        # every 128-channel output tile normalises the window again: only worth it while there is one tile
        fused = bool(USE_GN_FUSION and x.C == p.Cin and p.Cout <= GN_FUSION_MAX_COUT and lib().onedc_conv_win_takes(C.byref(p)))
        ab = group_norm_ab(x, gamma, beta, eps, ws)
        if fused:
            p.gn_ab, p.gn_silu = ab.data_ptr(), int(silu)
        else:                                     # the normalised tensor is a dense buffer of x.C channels
            p.x_cs, p.x_co = x.C, 0
        # bf16 operands: asked of the dispatcher with the launch's FINAL parameters (gn_ab set or the pass's geometry: with gn_ab
        # the 4-wave window kernel has one more condition -- room for the (a, b) pairs, Cin <= 768 -- and a layer it then leaves to
        # the 8-wave kernel must keep f16 operands instead of being refused with -17)
        bf = False
        if BF16_OPERANDS and w_bf16 is not None and act() == "f16" and not x.f32:
            p.x_bf16, p.w = 1, w_bf16.data_ptr()
            bf = lib().onedc_conv_gemm_kernel_id(C.byref(p)) == KERNEL_WIN3X3_4W      # (the only kernel that multiplies them)
            if not bf:
                p.x_bf16, p.w = 0, w.data_ptr()
        if not fused:
            xn = new_view(x.B, x.H, x.W, x.C, x.t.device, torch.bfloat16 if bf else None)
            group_norm(x, gamma, beta, xn, eps, silu, ws, out_bf16=bf, ab=ab)
            p.x, p.x_cs, p.x_co = xn.ptr, xn.cs, xn.co
    if KH == 3 and p.OH * p.OW <= 256 and SPLIT_K != 1:        # (pre-filter only: the C dispatcher decides)
        need = lib().onedc_conv_splitk_ws_bytes(C.byref(p))
        if need:
            wsb = splitk_workspace(x.t.device, need)
            p.ws, p.ws_bytes = wsb.data_ptr(), wsb.numel()
    if PROFILE is None or (PROFILE_ONLY is not None and conv_kernel_name(p) not in PROFILE_ONLY):
        _ok(lib().onedc_conv_gemm(C.byref(p), stream()), "onedc_conv_gemm")
        return y
    e0, e1 = torch.cuda.Event(enable_timing=True, **PROFILE_EVENT_KW), torch.cuda.Event(enable_timing=True, **PROFILE_EVENT_KW)
    e0.record()
    _ok(lib().onedc_conv_gemm(C.byref(p), stream()), "onedc_conv_gemm")
    e1.record()
    # executed multiply-adds: the phase-decomposed upsample conv (up == 2) runs 4 taps per output, not 9
    flops = 2.0 * p.B * p.OH * p.OW * p.Cout * ((4 if up == 2 else p.KH * p.KW) * p.Cin + p.Cin2) * max(batch, 1) * flop_scale
    es_in = 4 if p.in_f32 else 2
    es_out = 4 if p.out_f32 else 2
    cout_store = p.Cout // 2 if glu else p.Cout
    nbytes = (p.B * p.IH * p.IW * (p.Cin + p.Cin2) * es_in + p.B * p.OH * p.OW * cout_store * es_out + p.Cout * p.Cin2 * es_in +
              p.Cout * p.KH * p.KW * p.Cin * es_in + (p.B * p.OH * p.OW * cout_store * es_in if res is not None else 0)) \
        * max(batch, 1)
    PROFILE.append((conv_kernel_name(p), flops, e0, e1, float(nbytes)))
    return y


def direct_conv(x, w, bias, y, *, K, epi=EPI_LINEAR, slope=0.01, in_scale=1.0):
    p = DirectConvParams()
    p.x, p.w, p.bias, p.y = x.ptr, w.data_ptr(), _ptr(bias), y.ptr
    p.B, p.H, p.W, p.Cin, p.x_cs, p.x_co = x.B, x.H, x.W, x.C, x.cs, x.co
    p.Cout, p.y_cs, p.y_co, p.K = y.C, y.cs, y.co, K
    p.x_f32, p.out_f32, p.epi, p.slope, p.in_scale = int(x.f32), int(y.f32), epi, slope, in_scale
    _ok(lib().onedc_direct_conv(C.byref(p), stream()), "onedc_direct_conv")
    return y


def conv3x3_few_out(x, w, bias, y, gn=None):
    """3x3 conv to y.C <= 4 channels; gn = (gamma, beta, eps, silu, ws): GroupNorm (+SiLU) of x applied on load."""
    ab, silu = None, False
    if gn is not None:
        gamma, beta, eps, silu, ws = gn
        ab = group_norm_ab(x, gamma, beta, eps, ws)
    _ok(lib().onedc_conv3x3_few_out(x.ptr, w.data_ptr(), _ptr(bias), _ptr(ab), y.ptr, x.B, x.H, x.W, x.C, x.cs, x.co,
                                    y.C, y.cs, y.co, int(y.f32), int(silu), stream()), "onedc_conv3x3_few_out")
    return y


def dwconv3x3(x, w, bias, y):
    if x.f32:
        _ok(lib().onedc_dwconv3x3_f32(x.ptr, w.data_ptr(), _ptr(bias), y.ptr, x.B, x.H, x.W, x.C, x.cs, x.co,
                                      y.cs, y.co, stream()), "onedc_dwconv3x3_f32")
        return y
    _ok(lib().onedc_dwconv3x3(x.ptr, w.data_ptr(), _ptr(bias), y.ptr, x.B, x.H, x.W, x.C, x.cs, x.co,
                              y.cs, y.co, stream()), "onedc_dwconv3x3")
    return y


def gn_chunks(HW):
    return lib().onedc_gn_chunks(HW)


USE_EPILOGUE_STATS = True    # False: always run the separate statistics pass (tests, A/B)


# bf16 OPERANDS under f16 storage (round 5): convs that read a GroupNorm output take it -- an intermediate -- and their weights in
# bf16 where the 4-wave window kernel runs them; ONEDC_BF16_OPERANDS=0: f16 operands everywhere (A/B, tests)
BF16_OPERANDS = __import__("os").environ.get("ONEDC_BF16_OPERANDS", "1") != "0"
KERNEL_WIN3X3_4W = 10              # ONEDC_KERNEL_WIN3X3_4W of include/onedc_hip.h
# GroupNorm on load up to this many output channels: 128 = one channel tile per window.  Round 6 priced 256 (two tiles: each tile's
# workgroup normalises the window again) on hardware: per layer it beats the separate scale / shift pass (256 -> 256 at 384 x 384 2.552
# -> 2.491 ms, 512 -> 256 4.932 -> 4.658 ms: profiles/r6_tile256_probe.txt), in the whole bench it is worth + 0.1 % (noise:
# profiles/r6_ab_gn_fusion_cout.txt) -- it moves 3 ms of an HBM-bound pass into the MFMA-bound window kernel, whose issue slots are the
# step's scarce resource.  Not adopted; ONEDC_GN_FUSION_MAX_COUT=256 is the switch.
GN_FUSION_MAX_COUT = int(__import__("os").environ.get("ONEDC_GN_FUSION_MAX_COUT", "128"))
USE_GN_FUSION = True         # False: GroupNorm in front of a conv always runs as its own scale/shift pass (tests, A/B)


def group_norm_ab(x, gamma, beta, eps, ws):
    """Per (image, channel) scale and shift of GroupNorm(32): float2 [B][C] in ws.  When the convs that
    produced x left partial sums (x.stats covers the normalised channels and a group is whole quads / pairs),
    the statistics pass over x is skipped."""
    HW = x.H * x.W
    ab = ws.f32("gn_ab", x.B * x.C * 2)
    if (USE_EPILOGUE_STATS and x.stats is not None and (x.C // 32) % x.stats.gran == 0 and
            x.co % x.stats.gran == 0 and x.stats.covers(x.co, x.C)):
        _ok(lib().onedc_gn_from_stats(x.stats.t.data_ptr(), gamma.data_ptr(), beta.data_ptr(), ab.data_ptr(),
                                      x.B, HW, x.C, x.cs, x.co, eps, x.stats.gran, stream()), "onedc_gn_from_stats")
    else:
        part = ws.f32("gn_partial", x.B * gn_chunks(HW) * 64)
        _ok(lib().onedc_gn_stats(x.ptr, gamma.data_ptr(), beta.data_ptr(), part.data_ptr(), ab.data_ptr(),
                                 x.B, HW, x.C, x.cs, x.co, eps, stream()), "onedc_gn_stats")
    return ab


def group_norm(x, gamma, beta, y, eps, silu, ws, out_bf16=False, ab=None):
    """ws: Workspace providing .gn_partial (f32) and .gn_ab (f32) scratch tensors.
    out_bf16: y is stored as bf16 whatever the current format (its reader is a conv with bf16 operands, conv_gemm).
    ab: the (a, b) pairs when the caller has run group_norm_ab already."""
    if ab is None:
        ab = group_norm_ab(x, gamma, beta, eps, ws)
    _ok(lib().onedc_scale_shift_act_fmt(x.ptr, ab.data_ptr(), y.ptr, x.B, x.H * x.W, x.C, x.cs, x.co, y.cs, y.co,
                                        int(silu), int(bool(out_bf16)), stream()), "onedc_scale_shift_act")
    return y


def layernorm(x, gamma, beta, y, rows, Cc, eps=1e-5):
    _ok(lib().onedc_layernorm(x.data_ptr(), gamma.data_ptr(), beta.data_ptr(), y.data_ptr(), rows, Cc, eps,
                              stream()), "onedc_layernorm")
    return y


def softmax_rows(s, p, rows, n, n_pad):
    _ok(lib().onedc_softmax_rows(s.data_ptr(), p.data_ptr(), rows, n, n_pad, stream()), "onedc_softmax_rows")
    return p


def flash_vt_dtype(DQ, DV, Nk):
    """torch dtype in which onedc_flash_attn wants V^T for this shape: bf16 for the d = 40 kernel in either library
    (its P / V^T product needs f32's exponent range), the current activation format otherwise."""
    return torch.bfloat16 if lib().onedc_flash_attn_vt_is_bf16(DQ, DV, Nk) else a16()


def flash_attn(q, k, vt, o, B, H, Nq, Nk, Nk_pad, DQ, DV, scale, q_rs=0, k_rs=0, vt_bs=0):
    """q_rs / k_rs: elements between consecutive query / key rows when q / k are column slices of a wider buffer (0: dense,
    H * DQ); vt_bs: elements between the V^T blocks of consecutive images (0: dense, H * DV * Nk_pad)."""
    if vt.dtype != flash_vt_dtype(DQ, DV, Nk):
        raise KernelError(f"flash_attn: V^T must be {flash_vt_dtype(DQ, DV, Nk)} for DQ={DQ}, DV={DV}, Nk={Nk}, not {vt.dtype}")
    _ok(lib().onedc_flash_attn_strided(q.data_ptr(), k.data_ptr(), vt.data_ptr(), o.data_ptr(), B, H, Nq, Nk, Nk_pad,
                                       DQ, DV, scale, q_rs, k_rs, vt_bs, stream()), "onedc_flash_attn_strided")
    return o


# The cross-attention half of a transformer block as one launch (csrc/xattn_block.hip, round 6) where the kernel takes the shape
# (the UNet's 96 x 96 level: C = 320, 8 heads of 40, compact heads, a context of 16 .. 160 tokens in whole 16s); ONEDC_XATTN=0: the
# four separate launches everywhere (A/B, tests)
USE_XATTN_BLOCK = os.environ.get("ONEDC_XATTN", "1") != "0"


def xattn_params(h, ln_g, ln_b, eps, wq, bq, k, k_rs, vt, vt_bs, wo, bo, y, heads, d, Nk, Nk_pad, scale):
    """ln_g / ln_b None: the LayerNorm's affine is folded into wq / bq (onedc_xattn_params.bq)."""
    p = XAttnParams()
    p.h, p.ln_g, p.ln_b, p.wq, p.k, p.vt, p.wo, p.bo, p.y = (h.ptr, _ptr(ln_g), _ptr(ln_b), wq.data_ptr(), k.data_ptr(),
                                                            vt.data_ptr(), wo.data_ptr(), _ptr(bo), y.ptr)
    p.bq = _ptr(bq)
    p.B, p.HW, p.C, p.heads, p.d, p.Nk, p.Nk_pad, p.k_rs = h.B, h.H * h.W, h.C, heads, d, Nk, Nk_pad, k_rs
    p.h_cs, p.h_co, p.y_cs, p.y_co = h.cs, h.co, y.cs, y.co
    p.vt_bf16 = int(vt.dtype == torch.bfloat16 and a16() != torch.bfloat16)
    p.ln_eps, p.scale, p.vt_bs = eps, scale, vt_bs
    return p


def xattn_block_ok(p):
    return bool(USE_XATTN_BLOCK and lib().onedc_xattn_block_ok(C.byref(p)))


def xattn_block(p):
    _ok(lib().onedc_xattn_block(C.byref(p), stream()), "onedc_xattn_block")


def fsq_codes(idx, codes):
    _ok(lib().onedc_fsq_codes(idx.data_ptr(), codes.data_ptr(), idx.numel(), stream()), "onedc_fsq_codes")
    return codes


def nchw_to_nhwc(x, y, reflect=False):
    """x: f32 NCHW tensor (B,C,H,W) -> View y (B,Hp,Wp,>=C), replicate (or reflect) padding to y's frame."""
    B, Cc, Hx, Wx = x.shape
    _ok(lib().onedc_nchw_to_nhwc(x.data_ptr(), y.ptr, int(y.f32), B, Cc, Hx, Wx, y.H, y.W, y.cs, y.co, int(reflect),
                                 stream()), "onedc_nchw_to_nhwc")
    return y


def patchify8(x, y):
    """x: f32 NHWC View (B,H,W,>=3) -> bf16 View y (B,H/8,W/8,192)."""
    _ok(lib().onedc_patchify8(x.ptr, x.cs, y.ptr, x.B, x.H, x.W, stream()), "onedc_patchify8")
    return y


def latent_sample(moments, noise, out, scale):
    n = moments.B * moments.H * moments.W
    _ok(lib().onedc_latent_sample(moments.ptr, _ptr(noise), out.ptr, n, float(scale), stream()), "onedc_latent_sample")
    return out


def fsq_quantize(z, idx):
    """z: View (B,h,w,>=7) -> int32 indices tensor [B*h*w]."""
    _ok(lib().onedc_fsq_quantize(z.ptr, int(z.f32), z.cs, z.co, idx.data_ptr(), z.B * z.H * z.W, stream()),
        "onedc_fsq_quantize")
    return idx


def prior_indexes(params, indexes, B, H, W, Cc, step):
    _ok(lib().onedc_prior_indexes(params.ptr, int(params.f32), indexes.data_ptr(), B, H, W, Cc, step,
                                  stream()), "onedc_prior_indexes")
    return indexes


def prior_update_dec(params, symbols, y_hat, step):
    _ok(lib().onedc_prior_update_dec(params.ptr, int(params.f32), symbols.data_ptr(), y_hat.ptr, int(y_hat.f32),
                                     y_hat.cs, y_hat.B, y_hat.H, y_hat.W, y_hat.C, step, stream()),
        "onedc_prior_update_dec")


def prior_update_enc(params, y, symbols, y_hat, step):
    _ok(lib().onedc_prior_update_enc(params.ptr, int(params.f32), y.data_ptr(), symbols.data_ptr(), y_hat.ptr,
                                     int(y_hat.f32), y_hat.cs, y_hat.B, y_hat.H, y_hat.W, y_hat.C, step,
                                     stream()), "onedc_prior_update_enc")


def prior_update_zonly(params, y_hat, step):
    _ok(lib().onedc_prior_update_zonly(params.ptr, int(params.f32), y_hat.ptr, int(y_hat.f32), y_hat.cs,
                                       y_hat.B, y_hat.H, y_hat.W, y_hat.C, step, stream()),
        "onedc_prior_update_zonly")


def x0_from_noise(reduced, eps, x0, alpha_bar):
    _ok(lib().onedc_x0_from_noise(reduced.data_ptr(), eps.data_ptr(), x0.data_ptr(), x0.numel(),
                                  float(alpha_bar), stream()), "onedc_x0_from_noise")
    return x0


def image_out(x, out, H, W, unit_range=False, nonfinite=None):
    """nonfinite: int32 device tensor [B], zero-filled: word b becomes 1 when picture b holds a NaN / Inf (before the clamp)."""
    _ok(lib().onedc_image_out_checked(x.ptr, int(x.f32), x.cs, out.data_ptr(), x.B, x.H, x.W, H, W, int(unit_range),
                                      _ptr(nonfinite), stream()), "onedc_image_out_checked")
    return out


def image_out_u8(x, out, H, W, nhwc=True, nonfinite=None):
    """8-bit samples of the cropped picture (save_image's quantisation on the device): out uint8 [B,H,W,3] or [B,3,H,W]."""
    _ok(lib().onedc_image_out_u8(x.ptr, int(x.f32), x.cs, out.data_ptr(), x.B, x.H, x.W, H, W, int(nhwc),
                                 _ptr(nonfinite), stream()), "onedc_image_out_u8")
    return out


def copy_channels(x, y):
    _ok(lib().onedc_copy_channels(x.ptr, int(x.f32), x.cs, x.co, y.ptr, int(y.f32), y.cs, y.co,
                                  x.B * x.H * x.W, x.C, stream()), "onedc_copy_channels")
    return y


def window_partition(x, y, B, H, W, Cc, win, reverse=False, region=None, win_hw=None):
    """region = (y0, x0, RH, RW) (default: whole map); win_hw = (wh, ww) (default: win x win)."""
    y0, x0, RH, RW = region or (0, 0, H, W)
    wh, ww = win_hw or (win, win)
    _ok(lib().onedc_window_partition(x.data_ptr(), y.data_ptr(), B, H, W, Cc, y0, x0, RH, RW, wh, ww,
                                     int(reverse), stream()), "onedc_window_partition")
    return y


def ref_conv1x1(x, wT, bias, y, *, blk, bias_first, act=0, slope=0.01, slope2=0.01, res=None, shuffle=False):
    """Reference-arithmetic 1x1 conv (csrc/ref_arith.hip): x, y, res f32 Views; wT f32 [C][OC]; y = act(chains) [+ res]."""
    C, OC = wT.shape
    _ok(lib().onedc_ref_conv1x1(x.ptr, x.cs, x.co, wT.data_ptr(), _ptr(bias), (res.ptr if res is not None else None),
                                (res.cs if res is not None else 0), (res.co if res is not None else 0), y.ptr, y.cs, y.co,
                                x.B * x.H * x.W, C, OC, int(blk), int(bool(bias_first)), int(act), slope, slope2,
                                int(bool(shuffle)), x.H, x.W, stream()), "onedc_ref_conv1x1")
    return y


def ref_conv3x3(x, w3T, bias, y, *, leaky=None, res=None, icb=16, one_chain=False):
    """Reference-arithmetic dense 3x3 conv (oneDNN's direct kernels): w3T f32 [9][C][OC]; (icb 16, block chains added to the bias)
    = avx512, (icb 8, one chain from the bias) = avx2; y = [LeakyReLU(leaky)](.) [+ res]."""
    _, C, OC = w3T.shape
    _ok(lib().onedc_ref_conv3x3(x.ptr, x.cs, x.co, w3T.data_ptr(), _ptr(bias), (res.ptr if res is not None else None),
                                (res.cs if res is not None else 0), (res.co if res is not None else 0), y.ptr, y.cs, y.co,
                                x.B, x.H, x.W, C, OC, int(leaky is not None), float(leaky or 0.0), int(icb), int(bool(one_chain)),
                                stream()), "onedc_ref_conv3x3")
    return y


def ref_conv3x3_im2col(x, wKT, bias, y, *, parts, blk, split, leaky=None, res=None):
    """Reference-arithmetic dense 3x3 conv on torch's native path (small maps): wKT f32 [9 C][OC], k = c * 9 + ky * 3 + kx."""
    K9, OC = wKT.shape
    _ok(lib().onedc_ref_conv3x3_im2col(x.ptr, x.cs, x.co, wKT.data_ptr(), _ptr(bias), (res.ptr if res is not None else None),
                                       (res.cs if res is not None else 0), (res.co if res is not None else 0), y.ptr, y.cs,
                                       y.co, x.B, x.H, x.W, K9 // 9, OC, int(parts), int(blk), int(split),
                                       int(leaky is not None), float(leaky or 0.0), stream()), "onedc_ref_conv3x3_im2col")
    return y


def ref_prior_indexes(params, thresholds, indexes, B, H, W, Cc, step):
    _ok(lib().onedc_ref_prior_indexes(params.ptr, thresholds.data_ptr(), thresholds.numel(), indexes.data_ptr(), B, H, W, Cc,
                                      step, stream()), "onedc_ref_prior_indexes")
    return indexes


def mxfp8_roundtrip(x, y):
    """x, y: 16-bit Views of one shape: y = x quantised to MXFP8 (e4m3, one power-of-two scale per 32 channels of a pixel)
    and dequantised again."""
    _ok(lib().onedc_mxfp8_roundtrip(x.ptr, y.ptr, x.B * x.H * x.W, x.C, x.cs, x.co, y.cs, y.co, stream()),
        "onedc_mxfp8_roundtrip")
    return y


def timestep_embedding(t, emb):
    _ok(lib().onedc_timestep_embedding(t.data_ptr(), emb.data_ptr(), t.numel(), emb.shape[-1], stream()),
        "onedc_timestep_embedding")
    return emb


def add(a, b, y):
    _ok(lib().onedc_add(a.data_ptr(), b.data_ptr(), y.data_ptr(), y.numel(), stream()), "onedc_add")
    return y
