"""Container format and FSQ-index (de)serialisation, reference names.

Mirrors reference ``src/modules/entropy/utils.py`` (``get_padding_size`` :7-16,
``encode_i`` :95-105, ``decode_i`` :108-132, file helpers :85-92) and the index
bit-packing of ``codec_module.py`` :404-409 / :425-429, over the C ABI.
"""
import ctypes as C
from pathlib import Path

import numpy as np

from .. import _native
from .rans import _check


def get_padding_size(height, width, p=64):
    new_h = (height + p - 1) // p * p
    new_w = (width + p - 1) // p * p
    return 0, new_w - width, 0, new_h - height  # left, right, top, bottom


def filesize(filepath):
    if not Path(filepath).is_file():
        raise ValueError(f'Invalid file "{filepath}".')
    return Path(filepath).stat().st_size


def write_to_file(value, file_path):
    with Path(file_path).open("wb") as f:
        f.write(value)


def read_from_file(file_path):
    with Path(file_path).open("rb") as f:
        return f.read()


def _buf(b):
    a = np.frombuffer(bytes(b), dtype=np.uint8)
    return a, (a.ctypes.data if a.size else None)


def encode_i(pic_height, pic_width, bit_stream_y, bit_stream_z, bit_stream_caption, caption_length):
    lib = _native.coder()
    if isinstance(bit_stream_caption, str):
        bit_stream_caption = bit_stream_caption.encode("utf-8")
    y, py = _buf(bit_stream_y)
    z, pz = _buf(bit_stream_z)
    c, pc = _buf(bit_stream_caption)
    # the reference writes `caption_length` into the header and whatever bytes it was given
    # (utils.py:101-104); with the codec's empty caption both are zero
    need = 16 + y.size + z.size + c.size
    out = np.empty(need, dtype=np.uint8)
    n = lib.onedc_container_write(pic_height, pic_width, py, y.size, pz, z.size, pc, c.size,
                                  out.ctypes.data, need)
    _check(n, "encode_i")
    if caption_length != c.size:
        out[12:16] = np.frombuffer(int(caption_length).to_bytes(4, "big"), dtype=np.uint8)
    return out.tobytes()


def decode_i(data, index_unit_length, ds):
    lib = _native.coder()
    a, pa = _buf(data)
    info = _native.Container()
    _check(lib.onedc_container_parse(pa, a.size, index_unit_length, ds, C.byref(info)), "decode_i")
    pad_r = info.pad_width - info.width
    pad_b = info.pad_height - info.height
    raw = bytes(data)
    return {
        "height": info.height,
        "width": info.width,
        "pad_height": info.pad_height,
        "pad_width": info.pad_width,
        "pad_tuple": (0, pad_r, 0, pad_b),
        "bit_stream_y": raw[info.off_y:info.off_y + info.len_y],
        "bit_stream_z": raw[info.off_z:info.off_z + info.len_z],
        "bit_stream_caption": raw[info.off_caption:info.off_caption + info.len_caption],
    }


def pack_indices(indices, unit_bits):
    """codec_module.py:404-409 — right-aligned big-endian bit string."""
    lib = _native.coder()
    idx = np.ascontiguousarray(np.asarray(indices).reshape(-1), dtype=np.int32)
    n = _check(lib.onedc_pack_indices(idx.ctypes.data, idx.size, unit_bits, None, 0), "pack")
    out = np.empty(max(n, 1), dtype=np.uint8)
    _check(lib.onedc_pack_indices(idx.ctypes.data, idx.size, unit_bits, out.ctypes.data, n), "pack")
    return out[:n].tobytes()


def unpack_indices(data, count, unit_bits):
    """codec_module.py:425-429"""
    lib = _native.coder()
    a, pa = _buf(data)
    out = np.empty(count, dtype=np.int32)
    _check(lib.onedc_unpack_indices(pa, a.size, count, unit_bits, out.ctypes.data), "unpack")
    return out
