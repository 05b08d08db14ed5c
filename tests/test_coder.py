"""Host entropy-coder boundary: product coder (C ABI) and the oracle C port against the
reference's golden vectors (tests/golden/g1..g4, produced by the reference build in oracle/_ref),
plus properties and the edge cases the reference cannot even produce (tiny / empty streams)."""
import hashlib

import numpy as np
import pytest

from conftest import golden
from onedc_amd.entropy import rans, utils
from onedc_amd.entropy.entropy_models import EntropyCoder, GaussianEncoder
from oracle import rans_port

SINGLE_PART_CASES = ["plain", "escape_skip", "short"]


def _coder(tables, parts=1):
    ec = EntropyCoder(False, parts)
    assert ec.add_cdf(*tables) == 0
    return ec


def test_gaussian_tables_match_reference(gaussian_tables):
    """G1: GaussianEncoder.update rebuilds the reference's 256 tables bit-for-bit."""
    cdf, length, offset = gaussian_tables
    ge = GaussianEncoder()
    ge.update(force=True, entropy_coder=EntropyCoder())
    c, l, o = ge.get_cdf_info()
    assert np.array_equal(c, cdf) and np.array_equal(l, length) and np.array_equal(o, offset)
    assert hashlib.sha256(c.astype(np.int32).tobytes()).hexdigest()[:16] == "681191522681e23f"
    assert c.shape == (256, 103) and l.min() == 7 and l.max() == 103
    assert o.min() == -50 and o.max() == -2


@pytest.mark.parametrize("case", SINGLE_PART_CASES)
def test_encode_bytes_identical(gaussian_tables, case):
    """G2: product encoder and oracle port both reproduce the reference's bytes."""
    g = golden("g2_rans_streams.npz")
    sym, idx, want = g[f"{case}.symbols"], g[f"{case}.indexes"], g[f"{case}.stream"].tobytes()
    ec = _coder(gaussian_tables)
    ec.reset()
    ec.encoder.encode_with_indexes(sym, idx, 0)
    ec.flush()
    assert ec.get_encoded_stream() == want
    assert want[0] == 0x01
    assert rans_port.encode(sym, idx, rans_port.Tables(*gaussian_tables)) == want


@pytest.mark.parametrize("case", SINGLE_PART_CASES)
def test_decode_symbols_identical(gaussian_tables, case):
    g = golden("g2_rans_streams.npz")
    sym, idx, stream = g[f"{case}.symbols"], g[f"{case}.indexes"], g[f"{case}.stream"].tobytes()
    want = np.where(idx < 0, 0, sym)
    ec = _coder(gaussian_tables)
    ec.set_stream(stream)
    assert np.array_equal(ec.decoder.decode_stream(idx, 0), want)
    dec = rans_port.Decoder(stream, rans_port.Tables(*gaussian_tables))
    assert np.array_equal(dec.decode(idx), want)


@pytest.mark.parametrize("case,parts", [("two_parts", 2), ("three_parts", 3)])
def test_multi_part_framing(gaussian_tables, case, parts):
    """py_rans.cpp:91-142 / :156-187 framing with streamPart > 1."""
    g = golden("g2_rans_streams.npz")
    sym, idx, want = g[f"{case}.symbols"], g[f"{case}.indexes"], g[f"{case}.stream"].tobytes()
    ec = _coder(gaussian_tables, parts)
    ec.reset()
    ec.encoder.encode_with_indexes(sym, idx, 0)
    ec.flush()
    assert ec.get_encoded_stream() == want
    ec.set_stream(want)
    assert np.array_equal(ec.decoder.decode_stream(idx, 0), np.where(idx < 0, 0, sym))
    with pytest.raises(RuntimeError):
        _coder(gaussian_tables, 1).set_stream(want)    # decoder built for another part count


def test_four_sequential_calls(gaussian_tables):
    """The codec pushes four planes and pulls them back in four calls (codec_module.py:394-399)."""
    g = golden("g2_rans_streams.npz")
    sym, idx = g["escape_skip.symbols"], g["escape_skip.indexes"]
    q = sym.size // 4
    ec = _coder(gaussian_tables)
    ec.reset()
    for k in range(4):
        ec.encoder.encode_with_indexes(sym[k * q:(k + 1) * q], idx[k * q:(k + 1) * q], 0)
    ec.flush()
    stream = ec.get_encoded_stream()
    assert stream == g["four_calls.stream"].tobytes()
    ec.set_stream(stream)
    got = np.concatenate([ec.decoder.decode_stream(idx[k * q:(k + 1) * q], 0) for k in range(4)])
    assert np.array_equal(got, np.where(idx < 0, 0, sym))


@pytest.mark.parametrize("n", [0, 1, 2, 5, 17])
def test_tiny_streams_round_trip(gaussian_tables, n):
    """Empty / few-symbol inputs: the reference corrupts its heap here (rans.cpp:165-178), so the
    pin is the oracle port, which follows the same arithmetic with a safe buffer."""
    rng = np.random.default_rng(n)
    idx = rng.integers(0, 256, n).astype(np.int16)
    sym = rng.integers(-60, 60, n).astype(np.int16)
    ec = _coder(gaussian_tables)
    ec.reset()
    ec.encoder.encode_with_indexes(sym, idx, 0)
    ec.flush()
    stream = ec.get_encoded_stream()
    assert stream == rans_port.encode(sym, idx, rans_port.Tables(*gaussian_tables))
    assert len(stream) >= 5            # flag byte + 4 state bytes even when nothing was coded
    ec.set_stream(stream)
    assert np.array_equal(ec.decoder.decode_stream(idx, 0), sym)


def test_round_trip_random_tables():
    """Property: decode(encode(x)) == x for arbitrary valid tables, incl. escapes both ways."""
    rng = np.random.default_rng(7)
    n_tab, L = 9, 40
    cdf = np.zeros((n_tab, L), np.int32)
    sizes = np.zeros(n_tab, np.int32)
    offsets = rng.integers(-12, 3, n_tab).astype(np.int32)
    for t in range(n_tab):
        nsym = int(rng.integers(2, L - 1))
        pmf = rng.random(nsym).astype(np.float32) + 1e-3
        pmf /= pmf.sum()
        c = np.array(rans.pmf_to_quantized_cdf(pmf, 16), np.int32)
        cdf[t, :c.size] = c
        sizes[t] = c.size
    enc, dec = rans.RansEncoder(False, 1), rans.RansDecoder(1)
    assert enc.add_cdf(cdf, sizes, offsets) == 0 and dec.add_cdf(cdf, sizes, offsets) == 0
    idx = rng.integers(-1, n_tab, 20000).astype(np.int16)
    sym = rng.integers(-300, 300, 20000).astype(np.int16)
    enc.encode_with_indexes(sym, idx, 0)
    enc.flush()
    stream = enc.get_encoded_stream()
    dec.set_stream(stream)
    assert np.array_equal(dec.decode_stream(idx, 0), np.where(idx < 0, 0, sym))
    t = rans_port.Tables(cdf, sizes, offsets)
    assert rans_port.encode(sym, idx, t) == stream.tobytes()


def test_encoder_reciprocal_division_is_exact_for_extreme_frequencies():
    """The encoder replaces x / freq by a multiply-high with a precomputed reciprocal (coder.cpp: EncSym).  Tables whose
    frequencies are 1, 2, 3, powers of two and their neighbours, and 65 000+ (two-symbol rows) must still give the bytes of
    the C restatement, which divides (oracle/rans_port.c), for long random symbol runs that walk the state through its range."""
    rng = np.random.default_rng(11)
    rows = []
    for f in (1, 2, 3, 4, 5, 7, 8, 9, 255, 256, 257, 4095, 4096, 4097, 32767, 32768, 32769, 65533, 65534):
        rows.append([0, f, 65535, 65536] if f < 65535 else None)
    rows = [r for r in rows if r is not None]
    rows += [[0, 1, 2, 3, 65536], [0, 21845, 43690, 65535, 65536], [0, 65533, 65534, 65535, 65536]]
    n_tab, L = len(rows), max(len(r) for r in rows)
    cdf = np.zeros((n_tab, L), np.int32)
    sizes = np.zeros(n_tab, np.int32)
    for t, r in enumerate(rows):
        cdf[t, :len(r)] = r
        sizes[t] = len(r)
    offsets = np.zeros(n_tab, np.int32)
    enc, dec = rans.RansEncoder(False, 1), rans.RansDecoder(1)
    assert enc.add_cdf(cdf, sizes, offsets) == 0 and dec.add_cdf(cdf, sizes, offsets) == 0
    n = 200000
    idx = rng.integers(0, n_tab, n).astype(np.int16)
    sym = (rng.integers(0, 4, n) % (sizes[idx] - 2)).astype(np.int16)             # in-table symbols only (no escape)
    sym[::1001] = 7                                                               # ... and a few escapes
    enc.encode_with_indexes(sym, idx, 0)
    enc.flush()
    stream = enc.get_encoded_stream()
    assert rans_port.encode(sym, idx, rans_port.Tables(cdf, sizes, offsets)) == stream.tobytes()
    dec.set_stream(stream)
    assert np.array_equal(dec.decode_stream(idx, 0), sym) and dec.finished()


def test_batch_decode_matches_serial(gaussian_tables):
    rng = np.random.default_rng(3)
    B, n = 6, 3000
    idx = rng.integers(0, 200, (B, n)).astype(np.int16)
    sym = rng.integers(-9, 9, (B, n)).astype(np.int16)
    coders = [_coder(gaussian_tables) for _ in range(B)]
    rans.encode_batch([c.encoder for c in coders], sym, idx, 0, n_threads=3)
    streams = []
    for c in coders:
        c.flush()
        streams.append(c.get_encoded_stream())
        c.set_stream(streams[-1])
    got = rans.decode_batch([c.decoder for c in coders], idx, 0, n_threads=4)
    assert np.array_equal(got, sym)


def test_truncated_stream_reports_error(gaussian_tables):
    g = golden("g2_rans_streams.npz")
    idx, stream = g["plain.indexes"], g["plain.stream"].tobytes()
    ec = _coder(gaussian_tables)
    ec.set_stream(stream[: len(stream) // 2])
    with pytest.raises(RuntimeError):
        ec.decoder.decode_stream(idx, 0)


def test_error_paths(gaussian_tables):
    ec = _coder(gaussian_tables)
    with pytest.raises(ValueError):
        ec.encoder.encode_with_indexes(np.zeros(4, np.int16), np.full(4, 256, np.int16), 0)  # no such table
    with pytest.raises(ValueError):
        ec.encoder.encode_with_indexes(np.zeros(4, np.int16), np.zeros(4, np.int16), 5)      # no such group
    with pytest.raises(RuntimeError):
        rans.RansEncoder(False, 1).get_encoded_stream()                                     # before flush
    with pytest.raises(RuntimeError):
        d = rans.RansDecoder(1)
        d.add_cdf(*gaussian_tables)
        d.decode_stream(np.zeros(4, np.int16), 0)                                           # before set_stream


@pytest.mark.parametrize("i", range(5))
def test_pmf_to_quantized_cdf(i):
    """G3, incl. the frequency-stealing branch (ops.cpp:52-73)."""
    g = golden("g3_pmf_cdf.npz")
    pmf, want = g[f"pmf{i}"], g[f"cdf{i}"]
    assert rans.pmf_to_quantized_cdf(pmf, 16) == want.tolist()
    assert np.array_equal(rans_port.pmf_to_quantized_cdf(pmf, 16), want)
    assert want[0] == 0 and want[-1] == 65536 and np.all(np.diff(want.astype(np.int64)) > 0)


@pytest.mark.parametrize("i", range(4))
def test_container_and_index_packing(i):
    """G4: encode_i / decode_i bytes and the 14-bit right-aligned big-endian index string."""
    g = golden("g4_containers.npz")
    h, w, ph, pw = (int(v) for v in g[f"c{i}.hw"])
    indices, y, z, blob = g[f"c{i}.indices"], g[f"c{i}.y"].tobytes(), g[f"c{i}.z"].tobytes(), g[f"c{i}.blob"].tobytes()
    assert utils.pack_indices(indices, 14) == z
    assert np.array_equal(utils.unpack_indices(z, indices.size, 14), indices)
    assert utils.encode_i(h, w, y, z, b"", 0) == blob
    d = utils.decode_i(blob, 14, 64)
    assert (d["height"], d["width"], d["pad_height"], d["pad_width"]) == (h, w, ph, pw)
    assert d["pad_tuple"] == tuple(int(v) for v in g[f"c{i}.pad_tuple"])
    assert d["bit_stream_y"] == y and d["bit_stream_z"] == z and d["bit_stream_caption"] == b""
    assert utils.get_padding_size(h, w, 64) == d["pad_tuple"]


def test_container_errors():
    with pytest.raises(RuntimeError):
        utils.decode_i(b"\0" * 8, 14, 64)                       # shorter than the header
    blob = utils.encode_i(64, 64, b"abc", b"\x12\x34", b"", 0)
    with pytest.raises(RuntimeError):
        utils.decode_i(blob[:-2], 14, 64)                       # payload cut short
    with pytest.raises(ValueError):
        utils.pack_indices([1 << 14], 14)                       # index does not fit 14 bits


# ---- decoder hardening + end-of-stream check (no reference counterpart: extensions of the boundary) ----------
def test_decoder_finished_accepts_exact_decode_and_flags_desync(gaussian_tables):
    """After the last symbol the decoder must stand on the encoder's initial state with every byte consumed
    (onedc_rans_dec_finished).  One wrong table index anywhere (what a rounding flip of build_indexes does) must be
    reported -- by the end check or, earlier, by the stream running dry -- never pass silently."""
    g = golden("g2_rans_streams.npz")
    sym, idx, stream = g["escape_skip.symbols"], g["escape_skip.indexes"], g["escape_skip.stream"].tobytes()
    d = rans.RansDecoder(1)
    d.add_cdf(*gaussian_tables)
    d.set_stream(stream)
    assert not d.finished()                       # nothing decoded yet: bytes left
    assert np.array_equal(d.decode_stream(idx, 0), np.where(idx < 0, 0, sym))
    assert d.finished()
    # the codec's usage: four pushes decoded in four calls, checked once at the end
    d.set_stream(g["four_calls.stream"].tobytes())
    q = len(idx) // 4
    for k in range(4):
        d.decode_stream(idx[k * q:(k + 1) * q], 0)
        assert d.finished() == (k == 3)
    # desynchronise: move one table index by one at several positions
    flagged = 0
    for pos in (0, 17, len(idx) // 2, len(idx) - 5):
        bad = idx.copy()
        if bad[pos] < 0:
            continue
        bad[pos] = bad[pos] + 1 if bad[pos] < 255 else bad[pos] - 1
        d.set_stream(stream)
        try:
            d.decode_stream(bad, 0)
            flagged += not d.finished()
        except RuntimeError:
            flagged += 1
    assert flagged >= 3
    # a truncated stream is an error, not garbage
    d.set_stream(stream[:len(stream) // 2])
    with pytest.raises(RuntimeError):
        d.decode_stream(idx, 0)
    assert not d.finished()
    with pytest.raises(RuntimeError):
        rans.RansDecoder(1).finished()            # no stream set


def test_add_cdf_rejects_malformed_tables(gaussian_tables):
    cdf, length, offset = (np.array(a) for a in gaussian_tables)
    for mutate in ("first", "last", "flat", "descending"):
        c = cdf.copy()
        n = int(length[3])
        if mutate == "first":
            c[3, 0] = 1
        elif mutate == "last":
            c[3, n - 1] = 65535
        elif mutate == "flat":
            c[3, 2] = c[3, 1]
        else:
            c[3, 2], c[3, 3] = c[3, 3], c[3, 2]
        for coder in (rans.RansDecoder(1), rans.RansEncoder(False, 1)):
            with pytest.raises(ValueError):
                coder.add_cdf(c, length, offset)


def test_malformed_escape_is_an_error_not_undefined_behaviour(gaussian_tables):
    """A stream of 0xff bytes lands on the escape symbol and then announces an endless run of bypass digits: the
    decoder stops at 16 digits (an int16 needs 9) with ONEDC_ERR_STREAM instead of shifting by >= 32."""
    d = rans.RansDecoder(1)
    d.add_cdf(*gaussian_tables)
    d.set_stream(bytes([0x01]) + bytes([0xff]) * 4096)
    with pytest.raises(RuntimeError):
        d.decode_stream(np.full(64, 200, np.int16), 0)
    assert not d.finished()
    out = None
    d.set_stream(bytes([0x01]) + bytes([0xff]) * 4096)
    with pytest.raises(rans.RansStreamError) as e:
        out = rans.decode_batch([d], np.full((1, 64), 200, np.int16), 0, 2)
    assert e.value.status == [-3] and out is None


def test_gaussian_table_builder_pieces(gaussian_tables):
    """The vectorised table builder: half widths are the smallest i in [2, 50] with Phi(i / sigma) > 0.9999, every row
    is a valid 16-bit CDF of its own length, and pmf_to_cdf pads with zeros."""
    ge = GaussianEncoder()
    cdf, length, offset = ge.build_tables()
    cdf, length, offset = cdf.numpy(), length.numpy(), offset.numpy()
    assert np.array_equal(cdf, gaussian_tables[0])
    sig = ge.scale_table.double().numpy()
    from math import erf, sqrt
    for t in (0, 40, 128, 200, 255):
        c = -int(offset[t])
        phi = lambda x: 0.5 * (1 + erf(x / sig[t] / sqrt(2)))       # noqa: E731
        assert c == 50 or phi(c) > 0.9999
        assert c == 2 or phi(c - 1) <= 0.9999 + 1e-6
        n = int(length[t])
        assert n == 2 * c + 3 and cdf[t, 0] == 0 and cdf[t, n - 1] == 65536
        assert (np.diff(cdf[t, :n]) > 0).all() and (cdf[t, n:] == 0).all()
    idx = ge.build_indexes(ge.scale_table.clone())
    assert abs(int(idx[0]) - 0) <= 1 and abs(int(idx[255]) - 255) <= 1 and (np.diff(idx.numpy()) >= 0).all()
    assert int(ge.build_indexes(ge.scale_table[:4] * 0)[0]) == 0           # floored at 1e-5 -> clamped to table 0
    assert ge.build_indexes(ge.scale_table[:4], skip_thres=1.0).tolist() == [-1, -1, -1, -1]
