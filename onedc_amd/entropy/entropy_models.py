"""Host-side mirror of the reference's entropy layer for the decode path.

Same class and method names as reference ``src/modules/entropy/entropy_models.py``
(``EntropyCoder`` :32-94, ``GaussianEncoder`` :252-374) so callers and tests read
alike.  Table construction is init-time host work and stays on torch-CPU fp32
exactly as the reference does it; the per-symbol ``build_indexes`` hot loop has a
HIP twin (``onedc_amd.kernels.build_indexes``) that the decoder uses on device.
"""
import math

import numpy as np
import torch

from . import rans


class EntropyCoder:
    """reference entropy_models.py:32-94"""

    def __init__(self, ec_thread=False, stream_part=1):
        self.encoder = rans.RansEncoder(ec_thread, stream_part)
        self.decoder = rans.RansDecoder(stream_part)

    @staticmethod
    def pmf_to_quantized_cdf(pmf, precision=16):
        return torch.IntTensor(rans.pmf_to_quantized_cdf(pmf.tolist(), precision))

    @staticmethod
    def pmf_to_cdf(pmf, tail_mass, pmf_length, max_length):
        # :47-55 — one quantised table per scale, padded to max_length + 2
        cdf = torch.zeros((len(pmf_length), max_length + 2), dtype=torch.int32)
        for i, p in enumerate(pmf):
            prob = torch.cat((p[: pmf_length[i]], tail_mass[i]), dim=0)
            _cdf = EntropyCoder.pmf_to_quantized_cdf(prob, 16)
            cdf[i, : _cdf.size(0)] = _cdf
        return cdf

    def reset(self):
        self.encoder.reset()

    def add_cdf(self, cdf, cdf_length, offset):
        enc_cdf_idx = self.encoder.add_cdf(cdf, cdf_length, offset)
        dec_cdf_idx = self.decoder.add_cdf(cdf, cdf_length, offset)
        assert enc_cdf_idx == dec_cdf_idx
        return enc_cdf_idx

    def encode_with_indexes(self, symbols, indexes, cdf_group_index):
        # :66-69 — symbols are clamped to +-30000 before the int16 cast
        self.encoder.encode_with_indexes(
            symbols.clamp(-30000, 30000).to(torch.int16).cpu().numpy(),
            indexes.to(torch.int16).cpu().numpy(), cdf_group_index)

    def encode_with_indexes_np(self, symbols, indexes, cdf_group_index):
        self.encoder.encode_with_indexes(
            symbols.clip(-30000, 30000).astype(np.int16).reshape(-1),
            indexes.astype(np.int16).reshape(-1), cdf_group_index)

    def flush(self):
        self.encoder.flush()

    def get_encoded_stream(self):
        return self.encoder.get_encoded_stream().tobytes()

    def set_stream(self, stream):
        self.decoder.set_stream(np.frombuffer(stream, dtype=np.uint8))

    def decode_stream(self, indexes, cdf_group_index):
        rv = self.decoder.decode_stream(indexes.to(torch.int16).cpu().numpy(), cdf_group_index)
        return torch.Tensor(rv)

    def decode_stream_np(self, indexes, cdf_group_index):
        return self.decoder.decode_stream(indexes.astype(np.int16).reshape(-1), cdf_group_index)


class GaussianEncoder:
    """reference entropy_models.py:252-374 (``distribution='gaussian'`` branch only:
    the codec constructs it that way, codec_module.py:192 -> compression_model.py:37)."""

    scale_min = 0.11
    scale_max = 64.0
    scale_level = 256

    def __init__(self, distribution="gaussian"):
        assert distribution == "gaussian", "the OneDC codec only instantiates the gaussian model"
        self.distribution = distribution
        self.scale_table = torch.exp(torch.linspace(
            math.log(self.scale_min), math.log(self.scale_max), self.scale_level))
        self.log_scale_min = math.log(self.scale_min)
        self.log_scale_max = math.log(self.scale_max)
        self.log_scale_step = (self.log_scale_max - self.log_scale_min) / (self.scale_level - 1)
        self.entropy_coder = None
        self.cdf_group_index = None
        self._offset = None
        self._quantized_cdf = None
        self._cdf_length = None

    def set_cdf_info(self, quantized_cdf, cdf_length, offset):
        self._quantized_cdf = quantized_cdf.cpu().numpy()
        self._cdf_length = cdf_length.reshape(-1).int().cpu().numpy()
        self._offset = offset.reshape(-1).int().cpu().numpy()

    def get_cdf_info(self):
        return self._quantized_cdf, self._cdf_length, self._offset

    def set_entropy_coder(self, entropy_coder):
        self.entropy_coder = entropy_coder

    def build_tables(self):
        """:313-350 — 256 tables; c_i = smallest i in [2,50] with Phi(i/sigma) > 0.9999."""
        pmf_center = torch.zeros_like(self.scale_table) + 50
        scales = torch.zeros_like(pmf_center) + self.scale_table
        dist = torch.distributions.normal.Normal(torch.zeros_like(scales), scales)
        for i in range(50, 1, -1):
            samples = torch.zeros_like(pmf_center) + i
            probs = torch.squeeze(dist.cdf(samples))
            pmf_center = torch.where(probs > torch.zeros_like(pmf_center) + 0.9999,
                                     torch.zeros_like(pmf_center) + i, pmf_center)
        pmf_center = pmf_center.int()
        pmf_length = 2 * pmf_center + 1
        max_length = torch.max(pmf_length).item()
        samples = (torch.arange(max_length) - pmf_center[:, None]).float()
        scales = torch.zeros_like(samples) + self.scale_table[:, None]
        dist = torch.distributions.normal.Normal(torch.zeros_like(scales), scales)
        upper = dist.cdf(samples + 0.5)
        lower = dist.cdf(samples - 0.5)
        pmf = upper - lower
        tail_mass = 2 * lower[:, :1]
        quantized_cdf = EntropyCoder.pmf_to_cdf(pmf, tail_mass, pmf_length, max_length)
        return quantized_cdf, pmf_length + 2, -pmf_center

    def update(self, force=False, entropy_coder=None):
        assert entropy_coder is not None
        self.entropy_coder = entropy_coder
        if not force and self._offset is not None:
            return
        self.set_cdf_info(*self.build_tables())
        self.cdf_group_index = self.entropy_coder.add_cdf(*self.get_cdf_info())

    def build_indexes(self, scales, skip_thres=None):
        """:355-362, host (torch) form; the device form is kernels.build_indexes."""
        scales = torch.maximum(scales, torch.zeros_like(scales) + 1e-5)
        indexes = (torch.log(scales) - self.log_scale_min) / self.log_scale_step
        indexes = indexes.clamp_(0, self.scale_level - 1)
        if skip_thres is not None:
            indexes = torch.where(scales < skip_thres, torch.zeros_like(indexes) - 1, indexes)
        return indexes.int()

    def encode(self, x, scales, skip_thres=None):
        indexes = self.build_indexes(scales, skip_thres)
        return self.entropy_coder.encode_with_indexes(x.reshape(-1), indexes.reshape(-1),
                                                      self.cdf_group_index)

    def decode_stream(self, scales, dtype, device, skip_thres=None):
        indexes = self.build_indexes(scales, skip_thres)
        val = self.entropy_coder.decode_stream(indexes.reshape(-1), self.cdf_group_index)
        return val.reshape(scales.shape).to(device).to(dtype)
