"""Host side of the entropy layer: the coder pair and the Gaussian table model.

Drop-in for the two classes callers use from reference ``src/modules/entropy/entropy_models.py``
(``EntropyCoder`` :32-94, ``GaussianEncoder`` :252-374): same class, method and attribute names, same
argument meaning and return types, so `codec_model.update(force=True)`, `gaussian_encoder.encode(...)`
and `entropy_coder.set_stream(...)` read exactly as in the reference's own call sites.  The bodies are
this repo's: the table builder is one vectorised pass over a (scale, sample) grid followed by one call
of the C quantiser per scale, and the coder pair is a thin adapter over the C ABI of
``include/onedc_coder.h``.  What MUST be reproduced to the bit is pinned by golden G1 (the reference's
256 tables, sha256[:16] = 681191522681e23f): the fp32 value of every pmf entry, i.e. torch-CPU's
``0.5 * (1 + erf(x / sigma / sqrt 2))`` evaluated in this operation order.
The per-symbol table lookup (`build_indexes`) has a device twin, ``kernels.prior_indexes``; the host
form here serves tests and callers that hold CPU tensors.
"""
import math

import numpy as np
import torch

from . import rans

_SYMBOL_LIMIT = 30000           # symbols are clamped to +-30000 before the int16 cast (reference :67)
_PRECISION = 16


def _as_i16(values, clamp=False):
    """torch tensor / ndarray -> flat contiguous int16 ndarray (optionally clamped like the reference's encode)."""
    if torch.is_tensor(values):
        if clamp:
            values = values.clamp(-_SYMBOL_LIMIT, _SYMBOL_LIMIT)
        return values.detach().to(torch.int16).cpu().numpy().reshape(-1)
    a = np.asarray(values)
    if clamp:
        a = a.clip(-_SYMBOL_LIMIT, _SYMBOL_LIMIT)
    return np.ascontiguousarray(a.reshape(-1), dtype=np.int16)


class EntropyCoder:
    """One rANS encoder and one decoder sharing their CDF tables (reference :32-94 over the pybind pair
    `RansEncoder` / `RansDecoder`; here over ``entropy.rans``, the ctypes face of libonedc_coder.so)."""

    def __init__(self, ec_thread=False, stream_part=1):
        self.encoder = rans.RansEncoder(ec_thread, stream_part)
        self.decoder = rans.RansDecoder(stream_part)

    # -- tables ---------------------------------------------------------------------------------------------
    @staticmethod
    def pmf_to_quantized_cdf(pmf, precision=_PRECISION):
        return torch.tensor(rans.pmf_to_quantized_cdf(np.asarray(pmf, dtype=np.float32), precision), dtype=torch.int32)

    @staticmethod
    def pmf_to_cdf(pmf, tail_mass, pmf_length, max_length):
        """pmf [T, >= max_length], tail_mass [T, 1], pmf_length [T] -> int32 [T, max_length + 2]: row t is the
        16-bit quantised CDF of (pmf[t, :pmf_length[t]], tail_mass[t]), zero padded (reference :47-55)."""
        pmf = np.asarray(pmf, dtype=np.float32)
        tail = np.asarray(tail_mass, dtype=np.float32).reshape(len(pmf), -1)[:, 0]
        lengths = np.asarray(pmf_length).reshape(-1).astype(np.int64)
        table = np.zeros((len(lengths), int(max_length) + 2), dtype=np.int32)
        for t, n in enumerate(lengths):
            row = rans.pmf_to_quantized_cdf(np.append(pmf[t, :n], tail[t]), _PRECISION)
            table[t, :len(row)] = row
        return torch.from_numpy(table)

    def add_cdf(self, cdf, cdf_length, offset):
        """Registers one table group with BOTH coders; they must agree on its index (reference :57-64)."""
        group = self.encoder.add_cdf(cdf, cdf_length, offset)
        if self.decoder.add_cdf(cdf, cdf_length, offset) != group:
            raise RuntimeError("encoder and decoder disagree on the CDF group index")
        return group

    # -- encode ---------------------------------------------------------------------------------------------
    def reset(self):
        self.encoder.reset()

    def encode_with_indexes(self, symbols, indexes, cdf_group_index):
        self.encoder.encode_with_indexes(_as_i16(symbols, clamp=True), _as_i16(indexes), cdf_group_index)

    encode_with_indexes_np = encode_with_indexes         # same entry for ndarrays

    def flush(self):
        self.encoder.flush()

    def get_encoded_stream(self):
        return self.encoder.get_encoded_stream().tobytes()

    # -- decode ---------------------------------------------------------------------------------------------
    def set_stream(self, stream):
        self.decoder.set_stream(stream)

    def decode_stream_np(self, indexes, cdf_group_index):
        return self.decoder.decode_stream(_as_i16(indexes), cdf_group_index)

    def decode_stream(self, indexes, cdf_group_index):
        """-> float32 torch tensor of the decoded symbols, flat (reference :85-89 returns torch.Tensor(...))."""
        return torch.from_numpy(self.decode_stream_np(indexes, cdf_group_index).astype(np.float32))


class GaussianEncoder:
    """Zero-mean Gaussian symbol model over 256 log-spaced scales 0.11 .. 64 (reference :252-374,
    ``distribution='gaussian'``: the only form the codec constructs, codec_module.py:192 ->
    compression_model.py:37).  Table i covers the integers [-c_i, c_i] plus one escape bin, with c_i the
    smallest i in [2, 50] whose Phi(i / sigma_i) exceeds 0.9999 (50 if none does)."""

    scale_min = 0.11
    scale_max = 64.0
    scale_level = 256
    _center_lo, _center_hi = 2, 50
    _center_mass = 0.9999

    def __init__(self, distribution="gaussian"):
        if distribution != "gaussian":
            raise ValueError("the OneDC codec only instantiates the gaussian model")
        self.distribution = distribution
        self.log_scale_min = math.log(self.scale_min)
        self.log_scale_max = math.log(self.scale_max)
        self.log_scale_step = (self.log_scale_max - self.log_scale_min) / (self.scale_level - 1)
        self.scale_table = torch.linspace(self.log_scale_min, self.log_scale_max, self.scale_level).exp()
        self.entropy_coder = None
        self.cdf_group_index = None
        self._quantized_cdf = self._cdf_length = self._offset = None

    # -- table bookkeeping ------------------------------------------------------------------------------------
    def set_cdf_info(self, quantized_cdf, cdf_length, offset):
        as_np = lambda t: t.cpu().numpy() if torch.is_tensor(t) else np.asarray(t)       # noqa: E731
        self._quantized_cdf = as_np(quantized_cdf).astype(np.int32)
        self._cdf_length = as_np(cdf_length).reshape(-1).astype(np.int32)
        self._offset = as_np(offset).reshape(-1).astype(np.int32)

    def get_cdf_info(self):
        return self._quantized_cdf, self._cdf_length, self._offset

    def set_entropy_coder(self, entropy_coder):
        self.entropy_coder = entropy_coder

    @staticmethod
    def _phi(x, sigma):
        """Normal(0, sigma).cdf(x) in torch-CPU fp32, operation for operation what torch.distributions evaluates
        (0.5 * (1 + erf(x * (1 / sigma) / sqrt 2))): golden G1 depends on these exact roundings."""
        return 0.5 * (1 + torch.erf(x * sigma.reciprocal() / math.sqrt(2)))

    def build_tables(self):
        """-> (int32 [256, 2 c_max + 3] quantised CDFs, lengths 2 c_i + 3, offsets -c_i)."""
        sigma = self.scale_table
        # half width per scale: first candidate whose cumulative mass passes the threshold
        cand = torch.arange(self._center_lo, self._center_hi + 1, dtype=torch.float32)
        passed = self._phi(cand[None, :].expand(len(sigma), -1).contiguous(), sigma[:, None]) > self._center_mass
        first = passed.int().argmax(dim=1) + self._center_lo
        center = torch.where(passed.any(dim=1), first, torch.full_like(first, self._center_hi)).int()
        width = 2 * center + 1
        widest = int(width.max())
        # probability of every integer bin [k - .5, k + .5], k = j - c_i, on the (scale, j) grid
        k = (torch.arange(widest)[None, :] - center[:, None]).float()
        sig = sigma[:, None].expand_as(k).contiguous()
        below = self._phi(k - 0.5, sig)
        pmf = self._phi(k + 0.5, sig) - below
        tail = 2 * below[:, :1]                        # mass outside [-c_i - .5, c_i + .5], by symmetry
        cdf = EntropyCoder.pmf_to_cdf(pmf.numpy(), tail.numpy(), width.numpy(), widest)
        return cdf, width + 2, -center

    def update(self, force=False, entropy_coder=None):
        if entropy_coder is None:
            raise ValueError("update needs the EntropyCoder that will hold the tables")
        self.entropy_coder = entropy_coder
        if self._offset is None or force:
            self.set_cdf_info(*self.build_tables())
            self.cdf_group_index = entropy_coder.add_cdf(*self.get_cdf_info())

    # -- per-symbol table choice --------------------------------------------------------------------------------
    def build_indexes(self, scales, skip_thres=None):
        """Table index per symbol: the log-scale position truncated to int, scales floored at 1e-5, clamped to
        [0, 255]; -1 (symbol skipped, decodes as 0) below `skip_thres` (reference :355-362).  The fp32 truncation is
        the desynchronisation hazard of the whole codec: same expression order as the device kernel
        (elementwise.hip: prior_indexes_kernel)."""
        s = scales.clamp_min(1e-5)
        pos = ((s.log() - self.log_scale_min) / self.log_scale_step).clamp_(0, self.scale_level - 1)
        if skip_thres is not None:
            pos = pos.masked_fill(s < skip_thres, -1.0)
        return pos.int()

    def encode(self, x, scales, skip_thres=None):
        self.entropy_coder.encode_with_indexes(x.reshape(-1), self.build_indexes(scales, skip_thres).reshape(-1),
                                               self.cdf_group_index)

    def decode_stream(self, scales, dtype, device, skip_thres=None):
        flat = self.entropy_coder.decode_stream(self.build_indexes(scales, skip_thres).reshape(-1), self.cdf_group_index)
        return flat.reshape(scales.shape).to(device=device, dtype=dtype)
