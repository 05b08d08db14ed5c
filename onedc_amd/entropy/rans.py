"""Python façade over the C-ABI coder with the reference's pybind method names.

Mirrors ``MLCodec_rans.RansEncoder`` / ``RansDecoder`` (reference
src/cpp/py_rans/py_rans.cpp:261-281) and ``MLCodec_CXX.pmf_to_quantized_cdf``
(src/cpp/ops/ops.cpp:84-91): same names, argument meaning and return types
(numpy arrays), so code written against the reference modules runs unchanged.
Errors surface as ``ValueError`` / ``RuntimeError`` instead of C++ asserts.
"""
import ctypes as C

import numpy as np

from .. import _native

_ERR = {-1: "bad argument", -2: "call order violated", -3: "bitstream too short or malformed",
        -4: "output buffer too small"}


class RansStreamError(RuntimeError):
    """decode_batch: one or more streams ran out of bytes or carried a malformed escape; `.status` has the per-stream
    status codes (0 = ok, -3 = ONEDC_ERR_STREAM)."""

    def __init__(self, status):
        self.status = list(status)
        super().__init__(f"decode_batch: bitstream too short or malformed (per-stream status {self.status})")


def _check(rc, what):
    if rc < 0:
        exc = ValueError if rc == -1 else RuntimeError
        raise exc(f"{what}: {_ERR.get(rc, rc)}")
    return rc


def _i16(a):
    return np.ascontiguousarray(np.asarray(a).reshape(-1), dtype=np.int16)


def _cdf_args(cdfs, sizes, offsets):
    cdfs = np.ascontiguousarray(cdfs, dtype=np.int32)
    sizes = np.ascontiguousarray(np.asarray(sizes).reshape(-1), dtype=np.int32)
    offsets = np.ascontiguousarray(np.asarray(offsets).reshape(-1), dtype=np.int32)
    if cdfs.ndim != 2 or cdfs.shape[0] != sizes.size or offsets.size != sizes.size:
        raise ValueError("add_cdf: expected cdfs[N,L], sizes[N], offsets[N]")
    return cdfs, sizes, offsets


class RansEncoder:
    def __init__(self, multiThread=False, streamPart=1):
        self._lib = _native.coder()
        self._h = self._lib.onedc_rans_enc_create(int(bool(multiThread)), int(streamPart))
        if not self._h:
            raise ValueError("RansEncoder: streamPart out of range")

    def __del__(self):
        if getattr(self, "_h", None):
            self._lib.onedc_rans_enc_destroy(self._h)
            self._h = None

    def encode_with_indexes(self, symbols, indexes, cdf_group_index):
        s, i = _i16(symbols), _i16(indexes)
        if s.size != i.size:
            raise ValueError("encode_with_indexes: symbols and indexes differ in length")
        _check(self._lib.onedc_rans_enc_encode_with_indexes(
            self._h, s.ctypes.data, i.ctypes.data, s.size, int(cdf_group_index)),
            "encode_with_indexes")

    def flush(self):
        _check(self._lib.onedc_rans_enc_flush(self._h), "flush")

    def get_encoded_stream(self):
        n = _check(self._lib.onedc_rans_enc_stream_size(self._h), "get_encoded_stream")
        out = np.empty(n, dtype=np.uint8)
        _check(self._lib.onedc_rans_enc_get_stream(self._h, out.ctypes.data, n),
               "get_encoded_stream")
        return out

    def reset(self):
        _check(self._lib.onedc_rans_enc_reset(self._h), "reset")

    def add_cdf(self, cdfs, cdfs_sizes, offsets):
        c, s, o = _cdf_args(cdfs, cdfs_sizes, offsets)
        return _check(self._lib.onedc_rans_enc_add_cdf(
            self._h, c.ctypes.data, c.shape[0], c.shape[1], s.ctypes.data, o.ctypes.data),
            "add_cdf")

    def empty_cdf_buffer(self):
        _check(self._lib.onedc_rans_enc_empty_cdf_buffer(self._h), "empty_cdf_buffer")


class RansDecoder:
    def __init__(self, streamPart=1):
        self._lib = _native.coder()
        self._h = self._lib.onedc_rans_dec_create(int(streamPart))
        if not self._h:
            raise ValueError("RansDecoder: streamPart out of range")

    def __del__(self):
        if getattr(self, "_h", None):
            self._lib.onedc_rans_dec_destroy(self._h)
            self._h = None

    def set_stream(self, encoded):
        buf = np.ascontiguousarray(np.frombuffer(encoded, dtype=np.uint8)
                                   if isinstance(encoded, (bytes, bytearray, memoryview))
                                   else encoded, dtype=np.uint8)
        _check(self._lib.onedc_rans_dec_set_stream(self._h, buf.ctypes.data, buf.size),
               "set_stream")

    def decode_stream(self, indexes, cdf_group_index):
        i = _i16(indexes)
        out = np.empty(i.size, dtype=np.int16)
        _check(self._lib.onedc_rans_dec_decode_stream(
            self._h, i.ctypes.data, i.size, int(cdf_group_index), out.ctypes.data),
            "decode_stream")
        return out

    def finished(self):
        """Extension: True when every byte of the stream has been consumed and the coder state is back at the
        encoder's initial state -- i.e. the (index, symbol) decisions made while decoding account for the stream."""
        rc = self._lib.onedc_rans_dec_finished(self._h)
        if rc == -2:
            raise RuntimeError("finished: call order violated (no stream set)")
        return rc == 0

    def add_cdf(self, cdfs, cdfs_sizes, offsets):
        c, s, o = _cdf_args(cdfs, cdfs_sizes, offsets)
        return _check(self._lib.onedc_rans_dec_add_cdf(
            self._h, c.ctypes.data, c.shape[0], c.shape[1], s.ctypes.data, o.ctypes.data),
            "add_cdf")

    def empty_cdf_buffer(self):
        _check(self._lib.onedc_rans_dec_empty_cdf_buffer(self._h), "empty_cdf_buffer")


def decode_batch(decoders, indexes, cdf_group_index, n_threads=8):
    """Extension: decode ``indexes[b]`` from ``decoders[b]`` for all b on a host thread pool.

    indexes: int16 [B, n] (C-contiguous) -> int16 [B, n].
    """
    lib = _native.coder()
    idx = np.ascontiguousarray(indexes, dtype=np.int16)
    B, n = idx.shape
    if B != len(decoders):
        raise ValueError("decode_batch: one decoder per row expected")
    out = np.empty((B, n), dtype=np.int16)
    handles = (C.c_void_p * B)(*[d._h for d in decoders])
    status = (C.c_int * B)()
    rc = lib.onedc_rans_dec_decode_batch(handles, B, idx.ctypes.data, n, int(cdf_group_index),
                                         out.ctypes.data, int(n_threads), status)
    if rc == -3:
        raise RansStreamError(status)
    _check(rc, f"decode_batch (per-stream status {list(status)})")
    return out


def encode_batch(encoders, symbols, indexes, cdf_group_index, n_threads=8):
    """Extension: push ``symbols[b]`` into ``encoders[b]`` for all b."""
    lib = _native.coder()
    sym = np.ascontiguousarray(symbols, dtype=np.int16)
    idx = np.ascontiguousarray(indexes, dtype=np.int16)
    B, n = sym.shape
    if idx.shape != sym.shape or B != len(encoders):
        raise ValueError("encode_batch: shape mismatch")
    handles = (C.c_void_p * B)(*[e._h for e in encoders])
    _check(lib.onedc_rans_enc_encode_batch(handles, B, sym.ctypes.data, idx.ctypes.data, n,
                                           int(cdf_group_index), int(n_threads)), "encode_batch")


def pmf_to_quantized_cdf(pmf, precision=16):
    """``MLCodec_CXX.pmf_to_quantized_cdf(list[float], int) -> list[int]``."""
    p = np.ascontiguousarray(pmf, dtype=np.float32).reshape(-1)
    out = np.empty(p.size + 1, dtype=np.uint32)
    _check(_native.coder().onedc_pmf_to_quantized_cdf(p.ctypes.data, p.size, int(precision),
                                                      out.ctypes.data), "pmf_to_quantized_cdf")
    return out.tolist()
