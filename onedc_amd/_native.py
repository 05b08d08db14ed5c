"""ctypes loaders for the two in-tree shared libraries.

* ``libonedc_coder.so``  — host entropy coder, C ABI in ``include/onedc_coder.h``
* ``libonedc_hip.so``    — gfx950 HIP kernels, C ABI in ``include/onedc_hip.h``; 16-bit activations are bf16
* ``libonedc_hip_f16.so`` — the same sources built with IEEE-half activations (``-DONEDC_ACT_F16``)

There is deliberately no fallback: a missing library raises ``NativeLibraryError``
so that a product path can never silently run on something else.
"""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(_HERE, "csrc")
CODER_SO = os.path.join(CSRC, "libonedc_coder.so")
# ONEDC_HIP_SO: developer override for A/B experiment builds of the same sources (tools/exp_build.sh)
HIP_SO = os.environ.get("ONEDC_HIP_SO") or os.path.join(CSRC, "libonedc_hip.so")
HIP_F16_SO = os.environ.get("ONEDC_HIP_F16_SO") or os.path.join(CSRC, "libonedc_hip_f16.so")


class NativeLibraryError(RuntimeError):
    pass


class Container(C.Structure):
    _fields_ = [
        ("height", C.c_uint32), ("width", C.c_uint32),
        ("pad_height", C.c_uint32), ("pad_width", C.c_uint32),
        ("len_y", C.c_uint32), ("len_z", C.c_uint32), ("len_caption", C.c_uint32),
        ("off_y", C.c_int64), ("off_z", C.c_int64), ("off_caption", C.c_int64),
    ]


_coder = None
_hip = None
_hip_f16 = None

_P = C.c_void_p
_I = C.c_int
_L = C.c_int64


def coder():
    """The host coder library (built by ``onedc_amd.build.build_coder``)."""
    global _coder
    if _coder is not None:
        return _coder
    if not os.path.exists(CODER_SO):
        raise NativeLibraryError(
            f"{CODER_SO} is missing: run `python -c 'import __graft_entry__ as g; g.build()'`")
    lib = C.CDLL(CODER_SO)
    sig = {
        "onedc_rans_enc_create": (_P, [_I, _I]),
        "onedc_rans_enc_destroy": (None, [_P]),
        "onedc_rans_enc_add_cdf": (_I, [_P, _P, _I, _I, _P, _P]),
        "onedc_rans_enc_empty_cdf_buffer": (_I, [_P]),
        "onedc_rans_enc_reset": (_I, [_P]),
        "onedc_rans_enc_encode_with_indexes": (_I, [_P, _P, _P, _L, _I]),
        "onedc_rans_enc_flush": (_I, [_P]),
        "onedc_rans_enc_stream_size": (_L, [_P]),
        "onedc_rans_enc_get_stream": (_I, [_P, _P, _L]),
        "onedc_rans_dec_create": (_P, [_I]),
        "onedc_rans_dec_destroy": (None, [_P]),
        "onedc_rans_dec_add_cdf": (_I, [_P, _P, _I, _I, _P, _P]),
        "onedc_rans_dec_empty_cdf_buffer": (_I, [_P]),
        "onedc_rans_dec_set_stream": (_I, [_P, _P, _L]),
        "onedc_rans_dec_decode_stream": (_I, [_P, _P, _L, _I, _P]),
        "onedc_rans_dec_finished": (_I, [_P]),
        "onedc_rans_dec_decode_batch": (_I, [_P, _I, _P, _L, _I, _P, _I, _P]),
        "onedc_rans_enc_encode_batch": (_I, [_P, _I, _P, _P, _L, _I, _I]),
        "onedc_pmf_to_quantized_cdf": (_I, [_P, _I, _I, _P]),
        "onedc_container_parse": (_I, [_P, _L, _I, _I, C.POINTER(Container)]),
        "onedc_container_write": (_L, [C.c_uint32, C.c_uint32, _P, C.c_uint32, _P, C.c_uint32,
                                       _P, C.c_uint32, _P, _L]),
        "onedc_pack_indices": (_L, [_P, _L, _I, _P, _L]),
        "onedc_unpack_indices": (_I, [_P, _L, _L, _I, _P]),
        "onedc_coder_version": (C.c_char_p, []),
        "onedc_coder_build_id": (C.c_char_p, []),
    }
    for name, (res, args) in sig.items():
        fn = getattr(lib, name)
        fn.restype = res
        fn.argtypes = args
    lib._signatures = sig
    check_build_id(CODER_SO, lib.onedc_coder_build_id().decode(), "coder")
    _coder = lib
    return lib


def check_build_id(path, built, kind):
    """A library that was compiled from OTHER sources than the ones beside it is an error, not a wrong answer (the .so files
    travel untracked next to the tree).  Only the in-tree default paths are checked: ONEDC_HIP_SO / ONEDC_HIP_F16_SO point at
    experiment builds on purpose; ONEDC_ALLOW_STALE_LIB=1 switches the check off."""
    if os.environ.get("ONEDC_ALLOW_STALE_LIB") == "1":
        return
    from . import build
    want = build.coder_source_id() if kind == "coder" else build.hip_source_id()
    if built != want:
        raise NativeLibraryError(
            f"{path} was built from other sources (library id {built}, sources {want}): rebuild with "
            "`python -c 'import __graft_entry__ as g; g.build()'`")


CODER_SYMBOLS = [
    "onedc_rans_enc_create", "onedc_rans_enc_destroy", "onedc_rans_enc_add_cdf",
    "onedc_rans_enc_empty_cdf_buffer", "onedc_rans_enc_reset",
    "onedc_rans_enc_encode_with_indexes", "onedc_rans_enc_flush", "onedc_rans_enc_stream_size",
    "onedc_rans_enc_get_stream", "onedc_rans_dec_create", "onedc_rans_dec_destroy",
    "onedc_rans_dec_add_cdf", "onedc_rans_dec_empty_cdf_buffer", "onedc_rans_dec_set_stream",
    "onedc_rans_dec_decode_stream", "onedc_rans_dec_finished", "onedc_rans_dec_decode_batch", "onedc_rans_enc_encode_batch",
    "onedc_pmf_to_quantized_cdf", "onedc_container_parse", "onedc_container_write",
    "onedc_pack_indices", "onedc_unpack_indices", "onedc_coder_version", "onedc_coder_build_id",
]


def hip(act="bf16"):
    """The HIP kernel library for the given 16-bit activation format ("bf16" or "f16"); signatures are attached by
    ``onedc_amd.kernels``."""
    global _hip, _hip_f16
    if act not in ("bf16", "f16"):
        raise ValueError(f"unknown activation format {act!r}")
    cached = _hip if act == "bf16" else _hip_f16
    if cached is not None:
        return cached
    path = HIP_SO if act == "bf16" else HIP_F16_SO
    if not os.path.exists(path):
        raise NativeLibraryError(
            f"{path} is missing: the HIP extension must be built "
            "(`python -c 'import __graft_entry__ as g; g.build()'`); there is no CPU fallback")
    lib = C.CDLL(path)
    if act == "bf16":
        _hip = lib
    else:
        _hip_f16 = lib
    return lib
