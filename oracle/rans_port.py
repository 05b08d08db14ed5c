"""ctypes wrapper over oracle/liboracle_rans.so (rans_port.c) — TEST INFRASTRUCTURE ONLY."""
import ctypes as C
import os

import numpy as np

_SO = os.path.join(os.path.dirname(os.path.abspath(__file__)), "liboracle_rans.so")
_lib = None


class _Dec(C.Structure):
    _fields_ = [("ptr", C.c_void_p), ("state", C.c_uint32)]


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(_SO):
            raise RuntimeError(f"{_SO} missing: run `make -C oracle oracle`")
        _lib = C.CDLL(_SO)
        _lib.oracle_rans_encode.restype = C.c_long
        _lib.oracle_rans_encode.argtypes = [C.c_void_p, C.c_void_p, C.c_long, C.c_void_p, C.c_int,
                                            C.c_void_p, C.c_void_p, C.c_void_p, C.c_long]
        _lib.oracle_rans_dec_init.argtypes = [C.POINTER(_Dec), C.c_void_p]
        _lib.oracle_rans_decode.argtypes = [C.POINTER(_Dec), C.c_void_p, C.c_long, C.c_void_p,
                                            C.c_int, C.c_void_p, C.c_void_p, C.c_void_p]
        _lib.oracle_pmf_to_quantized_cdf.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_void_p]
    return _lib


class Tables:
    def __init__(self, cdf, sizes, offsets):
        self.cdf = np.ascontiguousarray(cdf, np.int32)
        self.sizes = np.ascontiguousarray(sizes, np.int32)
        self.offsets = np.ascontiguousarray(offsets, np.int32)


def encode(symbols, indexes, t):
    s = np.ascontiguousarray(symbols, np.int16).reshape(-1)
    i = np.ascontiguousarray(indexes, np.int16).reshape(-1)
    out = np.empty(s.size * 24 + 64, np.uint8)
    n = lib().oracle_rans_encode(s.ctypes.data, i.ctypes.data, s.size, t.cdf.ctypes.data,
                                 t.cdf.shape[1], t.sizes.ctypes.data, t.offsets.ctypes.data,
                                 out.ctypes.data, out.size)
    assert n > 0
    return out[:n].tobytes()


class Decoder:
    """Stateful like the reference decoder: set a stream, then pull symbol planes in order."""

    def __init__(self, stream, t):
        # pad so that the oracle (which, like the reference, does not bounds-check) stays in memory
        self._buf = np.frombuffer(bytes(stream) + b"\0" * 64, np.uint8).copy()
        self._t = t
        self._d = _Dec()
        lib().oracle_rans_dec_init(C.byref(self._d), self._buf.ctypes.data)

    def decode(self, indexes):
        i = np.ascontiguousarray(indexes, np.int16).reshape(-1)
        out = np.empty(i.size, np.int16)
        t = self._t
        lib().oracle_rans_decode(C.byref(self._d), i.ctypes.data, i.size, t.cdf.ctypes.data,
                                 t.cdf.shape[1], t.sizes.ctypes.data, t.offsets.ctypes.data,
                                 out.ctypes.data)
        return out


def pmf_to_quantized_cdf(pmf, precision=16):
    p = np.ascontiguousarray(pmf, np.float32)
    out = np.empty(p.size + 1, np.uint32)
    rc = lib().oracle_pmf_to_quantized_cdf(p.ctypes.data, p.size, precision, out.ctypes.data)
    assert rc == 0
    return out
