"""The prior networks in the reference's CPU arithmetic, on the GPU (compatibility path, `IntraNoAR(prior_arith="torch_cpu")`).

Why: a y stream can only be decoded by a prior that takes every table-index decision its writer took (reference
src/modules/entropy/compression_model.py:369-407; build_indexes truncates: entropy_models.py:355-362).  The reference's
writer is torch-CPU fp32, and its convolutions turn out to be fixed arrangements of fused-multiply-add chains (measured;
DESIGN.md section 5, csrc/ref_arith.hip has the list).  The kernels of csrc/ref_arith.hip evaluate those chains with
v_fma_f32; this module

  * holds the prior-side layers (HyperDecoder, y_prior_fusion, y_spatial_prior_reduction, the three adaptors, y_spatial_prior:
    reference src/models/sd15_onedc_codec_stage1/codec_module.py:145-166, 205-217) on those kernels,
  * CALIBRATES them: which block length R and which bias position torch-CPU uses for a 1x1 convolution depends on the layer
    shape, the map size and torch's thread count (oneDNN's blocking heuristics, torch's use_mkldnn rule), so for every
    (C, OC, H, W) it meets, `Calibration` runs that convolution once on random data with the LOCAL torch-CPU and takes the
    (R, bias position) whose GPU result is bit-identical; the dense 3x3 layers are either oneDNN's direct kernel or -- on maps
    of <= 20 480 elements, i.e. pictures under ~0.16 megapixels -- aten's im2col + sgemm with a K split that depends on the
    thread count, found the same way; depthwise layers have one known structure that is checked.  A shape nothing
    reproduces raises `ReferenceArithmeticUnavailable`,
  * takes the WRITER'S structures where this host's differ (`profile`): oneDNN's kernels calibrate identically on Intel and
    AMD hosts, torch's native path (the BLAS sgemm: every 1x1 of a one-thread writer, small inputs of any writer) does not --
    `intel_profile(threads)` holds what an Intel AVX-512 writer does, `load_profile(path)` what tools/cpu_chain_probe.py
    measured on the writer's machine,
  * builds the index thresholds from torch's own logarithm.

Without a profile the calibration describes the torch of THIS host at `threads` threads (default `torch.get_num_threads()`):
a stream decodes when its writer ran the same configuration class -- e.g. the reference on an AVX-512 CPU with 2-8 threads,
which is what wrote tests/golden/g8_* / g9_*; one thread is the reference harness's own setting (src/inference.py:29,
tests/golden/g10_*).  Streams of this package's own encoder do not need any of this (exact-f32 MFMA prior on both sides).
"""
import math

import numpy as np
import torch

from . import kernels as K
from .kernels import View, new_view


class ReferenceArithmeticUnavailable(RuntimeError):
    pass


def _v(B, H, W, C, device):
    return new_view(B, H, W, C, device, torch.float32)


class _Threads:
    """torch's intra-op thread count for the duration of a calibration run (torch picks oneDNN or its native path by it)."""

    def __init__(self, n):
        self.n = n

    def __enter__(self):
        self.prev = torch.get_num_threads()
        if self.n != self.prev:
            torch.set_num_threads(self.n)

    def __exit__(self, *exc):
        if self.n != self.prev:
            torch.set_num_threads(self.prev)
        return False


class Calibration:
    """(block length, bias first) of torch-CPU's 1x1 convolution per (C, OC, H, W), found by trial against the local torch
    running with `threads` intra-op threads (default: what torch has now)."""

    def __init__(self, device, threads=None, profile=None):
        """profile: the WRITER'S structures where they are known and differ from this host's -- a dict {(C, OC, H, W): (block,
        bias_first)} or a callable (C, OC, H, W) -> (block, bias_first) (e.g. dumped from `Calibration.table` on the machine
        that writes the streams; torch's native gemm path blocks K differently on Intel and AMD CPUs, oneDNN's kernels do
        not).  Shapes the profile does not cover are calibrated against the local torch."""
        self.device = device
        self.table = {}
        self.checked = set()
        self.threads = int(threads or torch.get_num_threads())
        self.profile = profile

    def save_profile(self, path):
        """What this calibration has settled so far, in the JSON form of tools/cpu_chain_probe.py / `load_profile`: a decoder
        started with it asks no torch at all for these shapes."""
        import json
        out = {"host": "calibrated on the decoder", "threads": self.threads, "1x1": {}, "3x3": {}}
        for key, v in self.table.items():
            if key[0] == "3x3":
                out["3x3"][",".join(str(t) for t in key[1:])] = list(v)
            else:
                out["1x1"][",".join(str(t) for t in key)] = list(v)
        with open(path, "w") as f:
            json.dump(out, f, indent=1)
        return path

    def _rand(self, C, OC, H, W, k, groups=1):
        g = torch.Generator().manual_seed(C * 131 + OC * 7 + H)
        x = torch.randn(1, C, H, W, generator=g)
        w = torch.randn(OC, C // groups, k, k, generator=g) / math.sqrt(C * k * k / groups)
        b = torch.randn(OC, generator=g) * 0.05
        return x, w, b

    def conv1x1(self, C, OC, H, W):
        key = (C, OC, H, W)
        if key in self.table:
            return self.table[key]
        known = self._known(1, C, OC, H, W)
        if known is not None:
            self.table[key] = (int(known[0]), int(known[1]))
            return self.table[key]
        x, w, b = self._rand(C, OC, H, W, 1)
        with torch.no_grad(), _Threads(self.threads):
            ref = torch.nn.functional.conv2d(x, w, b)[0].permute(1, 2, 0).contiguous()             # (H, W, OC)
        dev = self.device
        cs = (C + 3) // 4 * 4
        xv = _v(1, H, W, cs, dev)
        xv.t[:H * W * cs].view(H, W, cs)[..., :C] = x[0].permute(1, 2, 0).to(dev)
        xv = xv.slice(0, C)
        wT = w.reshape(OC, C).t().contiguous().to(dev)
        bd = b.to(dev)
        yv = _v(1, H, W, OC, dev)
        found = None
        for bias_first in (1, 0):
            for blk in sorted(set([C] + list(range(16, C, 16))), reverse=True):
                K.ref_conv1x1(xv, wT, bd, yv, blk=blk, bias_first=bias_first)
                if torch.equal(yv.t[:H * W * OC].view(H, W, OC).cpu(), ref):
                    found = (blk, bias_first)
                    break
            if found:
                break
        if found is None:
            raise ReferenceArithmeticUnavailable(
                f"no fma-chain structure reproduces torch-CPU's 1x1 convolution {C}->{OC} on a {H}x{W} map with "
                f"{self.threads} thread(s) on this host")
        self.table[key] = found
        return found

    def _known(self, k, C, OC, H, W):
        if self.profile is None:
            return None
        if callable(self.profile):
            return self.profile(C, OC, H, W, k)
        return self.profile.get((C, OC, H, W) if k == 1 else ("3x3", C, OC, H, W))

    @staticmethod
    def im2col_candidates(K):
        """(parts, block, split) of the native dense 3x3 (csrc/ref_arith.hip header), the measured ones first."""
        out = [(1, 384, 1), (4, K // 4, 2), (3, K // 3, 1)]
        for T in (1, 2, 3, 4, 6, 8):
            if K % T:
                continue
            per = K // T
            for blk in sorted({per, 384, 256, 192, 128, 96, 64} & set(range(1, per + 1)), reverse=True):
                for split in range(T, 0, -1):
                    if (T, blk, split) not in out:
                        out.append((T, blk, split))
        return out

    def conv3x3(self, C, OC, H, W):
        """-> ("direct", 16, 0) (oneDNN avx512: 16-channel block chains added to the bias), ("direct", 8, 1) (oneDNN avx2: one chain
        from the bias) or ("im2col", parts, block, split) (torch's native path on maps of <= 20 480 elements), whichever reproduces
        the local torch (or what the writer's profile says; a bare ("direct",) there means the avx512 form)."""
        key = ("3x3", C, OC, H, W)
        if key in self.table:
            return self.table[key]
        known = self._known(3, C, OC, H, W)
        if known is not None:
            self.table[key] = tuple(known)
            return self.table[key]
        x, w, b = self._rand(C, OC, H, W, 3)
        with torch.no_grad(), _Threads(self.threads):
            ref = torch.nn.functional.conv2d(x, w, b, padding=1)[0].permute(1, 2, 0).contiguous()
        dev = self.device
        xv = _v(1, H, W, C, dev)
        xv.t[:H * W * C].view(H, W, C).copy_(x[0].permute(1, 2, 0))
        yv = _v(1, H, W, OC, dev)
        bd = b.to(dev)
        got = lambda: yv.t[:H * W * OC].view(H, W, OC).cpu()
        found = None
        w3T = w.permute(2, 3, 1, 0).reshape(9, C, OC).contiguous().to(dev)
        for icb, one in ((16, False), (8, True)):              # oneDNN's direct kernels: avx512_core, avx2
            if C % icb == 0 and found is None:
                K.ref_conv3x3(xv, w3T, bd, yv, icb=icb, one_chain=one)
                if torch.equal(got(), ref):
                    found = ("direct", icb, int(one))
        if found is None and C * H * W <= 20480:
            wKT = w.reshape(OC, C * 9).t().contiguous().to(dev)
            for T, blk, split in self.im2col_candidates(9 * C):
                K.ref_conv3x3_im2col(xv, wKT, bd, yv, parts=T, blk=blk, split=split)
                if torch.equal(got(), ref):
                    found = ("im2col", T, blk, split)
                    break
        if found is None:
            raise ReferenceArithmeticUnavailable(
                f"no fma-chain structure reproduces torch-CPU's dense 3x3 convolution {C}->{OC} on a {H}x{W} map with "
                f"{self.threads} thread(s) on this host (oneDNN's direct kernel, or aten's im2col + sgemm on maps of <= 20 480 "
                f"elements, whose K split depends on the thread count and the BLAS)")
        self.table[key] = found
        return found

    def check_dw3x3(self, C, H, W):
        key = ("dw", C, H, W)
        if key in self.checked:
            return
        x, w, b = self._rand(C, C, H, W, 3, groups=C)
        with torch.no_grad(), _Threads(self.threads):
            ref = torch.nn.functional.conv2d(x, w, b, padding=1, groups=C)[0].permute(1, 2, 0).contiguous()
        dev = self.device
        xv = _v(1, H, W, C, dev)
        xv.t[:H * W * C].view(H, W, C).copy_(x[0].permute(1, 2, 0))
        yv = _v(1, H, W, C, dev)
        K.dwconv3x3(xv, w[:, 0].permute(1, 2, 0).reshape(9, -1).contiguous().to(dev), b.to(dev), yv)
        if not torch.equal(yv.t[:H * W * C].view(H, W, C).cpu(), ref):
            raise ReferenceArithmeticUnavailable(f"torch-CPU's depthwise 3x3 convolution ({C} channels, {H}x{W}) is not the "
                                                 f"bias-first row-major chain on this host")
        self.checked.add(key)


def index_thresholds(scale_min=0.11, scale_max=64.0, levels=256):
    """thresholds[i] = the smallest f32 sigma whose index under the reference's build_indexes (entropy_models.py:355-362, run
    with the local torch-CPU) is >= i + 1; index(sigma) = #{i : thresholds[i] <= sigma}.  Bisection over the bit patterns of
    positive floats (monotone with the value), all 255 in parallel."""
    log_min = math.log(scale_min)
    log_step = (math.log(scale_max) - log_min) / (levels - 1)

    def index(s):
        # torch's vectorised loops finish a tensor's tail (n mod 2 x vector width) with the SCALAR libm function; the reference's
        # tensors are (B, 32, H, W), whole vectors always: pad to a multiple of 64 so that every element here takes the same path
        n = s.numel()
        s = torch.cat([s, s.new_ones((-n) % 64)])
        s = torch.maximum(s, torch.zeros_like(s) + 1e-5)
        return ((torch.log(s) - log_min) / log_step).clamp_(0, levels - 1).int()[:n]

    want = torch.arange(1, levels, dtype=torch.int32)
    lo = torch.full((levels - 1,), np.float32(1e-6).view(np.int32).item(), dtype=torch.int32)       # index 0 for sure
    hi = torch.full((levels - 1,), np.float32(128.0).view(np.int32).item(), dtype=torch.int32)      # index 255 for sure
    with torch.no_grad():
        while int((hi - lo).max()) > 1:
            mid = lo + (hi - lo) // 2
            ok = index(mid.view(torch.float32)) >= want
            hi = torch.where(ok, mid, hi)
            lo = torch.where(ok, lo, mid)
        thr = hi.view(torch.float32)
        assert bool((index(thr) >= want).all()) and bool((index(lo.view(torch.float32)) < want).all())
    return thr.contiguous()


class _Conv1x1:
    def __init__(self, sd, name, device, cal):
        w = sd[f"{name}.weight"].float()
        self.OC, self.C = w.shape[0], w.shape[1]
        self.wT = w.reshape(self.OC, self.C).t().contiguous().to(device)
        b = sd.get(f"{name}.bias")
        self.b = None if b is None else b.float().contiguous().to(device)
        self.cal, self.device = cal, device

    def __call__(self, x, y=None, *, act=0, slope=0.01, slope2=0.01, res=None, shuffle=False):
        blk, bias_first = self.cal.conv1x1(self.C, self.OC, x.H, x.W)
        if y is None:
            c = self.OC // 2 if act == 2 else self.OC
            y = _v(x.B, 2 * x.H, 2 * x.W, c // 4, self.device) if shuffle else _v(x.B, x.H, x.W, c, self.device)
        return K.ref_conv1x1(x, self.wT, self.b, y, blk=blk, bias_first=bias_first, act=act, slope=slope, slope2=slope2,
                             res=res, shuffle=shuffle)


class _DCB4:
    """DepthConvBlock4 (dcvc.py:424-433) in the reference's arithmetic."""

    def __init__(self, sd, p, device, cal):
        self.cal, self.device = cal, device
        self.conv1 = _Conv1x1(sd, f"{p}.block.0.conv1.0", device, cal)
        w = sd[f"{p}.block.0.depth_conv.weight"].float()
        self.dw_w = w[:, 0].permute(1, 2, 0).reshape(9, -1).contiguous().to(device)
        self.dw_b = sd[f"{p}.block.0.depth_conv.bias"].float().contiguous().to(device)
        self.conv2 = _Conv1x1(sd, f"{p}.block.0.conv2", device, cal)
        self.adaptor = _Conv1x1(sd, f"{p}.block.0.adaptor", device, cal) if f"{p}.block.0.adaptor.weight" in sd else None
        self.ffn_in = _Conv1x1(sd, f"{p}.block.1.conv", device, cal)
        self.ffn_out = _Conv1x1(sd, f"{p}.block.1.conv_out", device, cal)

    def __call__(self, x, y=None):
        self.cal.check_dw3x3(x.C, x.H, x.W)
        identity = self.adaptor(x) if self.adaptor is not None else x
        t = self.conv1(x, act=1, slope=0.01)
        t2 = K.dwconv3x3(t, self.dw_w, self.dw_b, _v(t.B, t.H, t.W, t.C, self.device))
        h = self.conv2(t2, res=identity)
        u = self.ffn_in(h, act=2, slope=0.1, slope2=0.01)
        return self.ffn_out(u, y, res=h)


class _RBU:
    """ResidualBlockUpsample (dcvc.py:183-208) in the reference's arithmetic."""

    def __init__(self, sd, p, device, cal):
        self.cal, self.device = cal, device
        self.subpel = _Conv1x1(sd, f"{p}.subpel_conv.0", device, cal)
        self.upsample = _Conv1x1(sd, f"{p}.upsample.0", device, cal)
        w = sd[f"{p}.conv.weight"].float()
        self.OC, self.C = w.shape[0], w.shape[1]
        self.w3T = w.permute(2, 3, 1, 0).reshape(9, self.C, self.OC).contiguous().to(device)
        self.wKT = w.reshape(self.OC, self.C * 9).t().contiguous().to(device)          # im2col order, for small maps
        self.b3 = sd[f"{p}.conv.bias"].float().contiguous().to(device)

    def __call__(self, x):
        t = self.subpel(x, act=1, slope=0.01, shuffle=True)
        identity = self.upsample(x, shuffle=True)
        how = self.cal.conv3x3(self.C, self.OC, t.H, t.W)
        out = _v(t.B, t.H, t.W, self.OC, self.device)
        if how[0] == "direct":
            icb, one = (how[1], how[2]) if len(how) > 2 else (16, 0)
            return K.ref_conv3x3(t, self.w3T, self.b3, out, leaky=0.1, res=identity, icb=icb, one_chain=one)
        return K.ref_conv3x3_im2col(t, self.wKT, self.b3, out, parts=how[1], blk=how[2], split=how[3], leaky=0.1, res=identity)


def intel_profile(threads, kc=256):
    """The WRITER'S structures of torch's NATIVE paths (aten _slow_conv2d_forward -> MKL sgemm with beta = 1 on a bias-filled
    output) as measured on an Intel AVX-512 host, for a decoder whose own host blocks differently (MKL dispatches by CPU: K
    blocks of 128 / 192 on an EPYC 9575F):
      1x1 (every shape at one thread, inputs of <= 20 480 elements otherwise): chains from 0 over blocks of `kc` input
          channels, bias added to the first block's sum;
      dense 3x3 on inputs of <= 20 480 elements (K = 9 * 128): the K split of the thread count -- 1, 4, 6 threads: blocks of
          384 in one sequence; 2, 8 threads: four parts of 288, (p0 + b + p1) + (p2 + p3); 3 threads: (p0 + b) + (p1 + p2).
    Shapes that go through oneDNN are not covered (None): its kernels are the same on Intel and AMD hosts and are
    calibrated against the local torch."""
    im2col = {1: (1, 384, 1), 4: (1, 384, 1), 6: (1, 384, 1), 2: (4, None, 2), 8: (4, None, 2), 3: (3, None, 1)}

    def profile(C, OC, H, W, k=1):
        small = C * H * W <= 20480
        if k == 1:
            return (min(C, kc), 0) if (small or threads == 1) else None
        if not small or threads not in im2col or C != 128:
            return None
        T, blk, split = im2col[threads]
        return ("im2col", T, blk or 9 * C // T, split)
    return profile


def load_profile(path):
    """A writer's profile written by tools/cpu_chain_probe.py on the machine that writes the streams (no GPU needed there) ->
    the dict form `Calibration` takes.  Shapes the probe could not fit (null) are left to the local calibration."""
    import json
    raw = json.load(open(path))
    out = {}
    for key, v in raw.get("1x1", {}).items():
        if v is not None:
            out[tuple(int(t) for t in key.split(","))] = (int(v[0]), int(v[1]))
    for key, v in raw.get("3x3", {}).items():
        if v is not None:
            out[("3x3",) + tuple(int(t) for t in key.split(","))] = tuple(v)
    if raw.get("gaussian_tables_sha16"):                     # the CDF tables the writer's torch builds (codec_module.update checks)
        out["gaussian_tables_sha16"] = str(raw["gaussian_tables_sha16"])
    if raw.get("index_thresholds_bits"):                     # the writer's logarithm (255 f32 bit patterns)
        out["index_thresholds"] = torch.tensor(raw["index_thresholds_bits"], dtype=torch.int32).view(torch.float32)
    return out


def native_gemm_profile(kc=256):
    """The one-thread writer on an Intel host (src/inference.py:29 sets one thread): `intel_profile(1, kc)`."""
    return intel_profile(1, kc)


class ReferencePrior:
    """The prior side of IntraNoAR on the reference-arithmetic kernels: same call surface as the MFMA path's pieces in
    codec_module.IntraNoAR (`params_from_indices`, `prior_setup`, `next_params`), same f32 NHWC Views out."""

    def __init__(self, sd, device, N=128, threads=None, profile=None):
        """threads: the torch thread count of the WRITER'S configuration to calibrate for (None: torch's current one);
        profile: known structures of the writer (`Calibration`)."""
        self.device, self.N = device, N
        self.cal = cal = Calibration(device, threads, profile)
        self.feat_in = _Conv1x1(sd, "hyper_dec.feat_in.0", device, cal)
        self.hyper = [_DCB4(sd, "hyper_dec.to_entropy.0", device, cal), _RBU(sd, "hyper_dec.to_entropy.1", device, cal),
                      _DCB4(sd, "hyper_dec.to_entropy.2", device, cal), _RBU(sd, "hyper_dec.to_entropy.3", device, cal),
                      _DCB4(sd, "hyper_dec.to_entropy.4", device, cal)]
        self.fusion = [_DCB4(sd, f"y_prior_fusion.{i}", device, cal) for i in range(2)]
        self.reduction = _Conv1x1(sd, "y_spatial_prior_reduction", device, cal)
        self.adaptors = [_DCB4(sd, f"y_spatial_prior_adaptor_{i}", device, cal) for i in (1, 2, 3)]
        self.spatial = [_DCB4(sd, f"y_spatial_prior.{i}", device, cal) for i in range(3)]
        thr = profile.get("index_thresholds") if isinstance(profile, dict) else None
        self.thresholds = (index_thresholds() if thr is None else thr).contiguous().to(device)

    def params_from_indices(self, indices):
        """indices int32 (B, h, w) -> (fused params f32 View (B, 4h, 4w, 2N), z_semantic 16-bit View)."""
        B, h, w = indices.shape
        codes = View(torch.empty(B * h * w * 8 + 8, dtype=torch.float32, device=self.device), B, h, w, 8)
        K.fsq_codes(indices.contiguous(), codes.t)
        z = self.feat_in(codes.slice(0, 7), act=1, slope=0.01)
        x = z
        for layer in self.hyper:
            x = layer(x)
        fused = self.fusion[1](self.fusion[0](x))
        z16 = K.copy_channels(z, new_view(z.B, z.H, z.W, z.C, self.device))
        return fused, z16

    def prior_setup(self, fused):
        B, H, W, N = fused.B, fused.H, fused.W, self.N
        common_cat = _v(B, H, W, 2 * N, self.device)
        self.reduction(fused, common_cat.slice(N, N))
        return common_cat, common_cat.slice(0, N)

    def next_params(self, k, common_cat):
        a = self.adaptors[k - 1](common_cat)
        for layer in self.spatial:
            a = layer(a)
        return a

    def indexes(self, params, idx_dev, B, H, W, step):
        return K.ref_prior_indexes(params, self.thresholds, idx_dev, B, H, W, self.N, step)
