"""IntraNoAR — the codec module of the decode path on HIP kernels (+ host rANS).

Mirrors reference src/models/sd15_onedc_codec_stage1/codec_module.py: `IntraNoAR.decode` :356-369,
`_decompress` :418-454, `HyperDecoder` :145-166, `SemanticAdaptor` :169-181, `Decoder` :88-116 and the
four-part prior of src/modules/entropy/compression_model.py:303-407.  The reference is batch = 1;
here every call takes a list of bitstreams and runs them as one batch (same arithmetic per image:
all kernels are batch-invariant), which is what makes the rANS hops and launches amortise.

Encode side built so far: `HyperEncoder` + FSQ quantisation + `compress` (hyper analysis
-> indices -> four-part prior -> rANS -> container) from the analysis transform's outputs (y, sem), and
`encode_latents` (the prior + entropy coder from given (z indices, y)).  The analysis transform `enc`
itself (SD-VAE encoder + UNet2DModel, third-party bodies) is not built yet.
"""
import os

import numpy as np
from types import SimpleNamespace
import torch

from . import kernels as K
from .entropy import rans
from .entropy.entropy_models import EntropyCoder, GaussianEncoder
from .entropy.utils import decode_i, encode_i, pack_indices, read_from_file, unpack_indices, write_to_file
from .kernels import View, new_view
from .layers import (Conv, DepthConvBlock4, ResidualBlockUpsample, VQAttnBlock, VQResnetBlock, Workspace)
from .ref_arith import ReferenceArithmeticUnavailable


class StreamDesyncError(RuntimeError):
    """The y stream of one or more pictures was not consumed exactly by the four rANS hops: somewhere a scale index
    (build_indexes truncation, reference entropy_models.py:355-362) or an earlier symbol differed from the ENCODER's.
    From the first such position on the decoded symbols are garbage, so no picture is returned.  Typical cause: the
    stream was written by a producer whose prior networks round differently (reference on another device or under
    bf16 autocast -- see `IntraNoAR.prior_f32` and INTEGRATION.md section 4), or the file is corrupt.
    `.images` lists the offending batch positions."""

    def __init__(self, images):
        self.images = list(images)
        super().__init__(f"rANS desynchronisation: the y stream of picture(s) {self.images} of the batch was not consumed "
                         "exactly (scale-index or symbol decisions differ from the encoder's, or the stream is corrupt)")


class HyperDecoder:
    def __init__(self, sd, device, p="hyper_dec", f32=False):
        self.f32 = f32
        w = sd[f"{p}.feat_in.0.weight"]                        # (128, 7, 1, 1) -> f32 [128][8] (pad cin)
        wf = torch.zeros(w.shape[0], 8)
        wf[:, :7] = w[:, :, 0, 0]
        self.feat_w = wf.contiguous().to(device)
        self.feat_b = sd[f"{p}.feat_in.0.bias"].float().contiguous().to(device)
        self.n = w.shape[0]
        self.b0 = DepthConvBlock4(sd, f"{p}.to_entropy.0", device, f32)
        self.u1 = ResidualBlockUpsample(sd, f"{p}.to_entropy.1", device, f32)
        self.b2 = DepthConvBlock4(sd, f"{p}.to_entropy.2", device, f32)
        self.u3 = ResidualBlockUpsample(sd, f"{p}.to_entropy.3", device, f32)
        self.b4 = DepthConvBlock4(sd, f"{p}.to_entropy.4", device, f32)

    def __call__(self, codes):
        """codes: f32 View (B, h, w, 8) -> (params View, z_semantic View (bf16: it feeds the synthesis))."""
        dev = codes.t.device
        z = new_view(codes.B, codes.H, codes.W, self.n, dev, torch.float32 if self.f32 else K.a16())
        K.direct_conv(codes, self.feat_w, self.feat_b, z, K=1, epi=K.EPI_LEAKY, slope=0.01)
        h = self.b4(self.u3(self.b2(self.u1(self.b0(z)))))
        if self.f32:
            z = K.copy_channels(z, new_view(z.B, z.H, z.W, z.C, dev))
        return h, z


class HyperEncoder:
    """Encode side: reference HyperEncoder, codec_module.py:119-142.
    z = fusion(cat(y_trans_coding(y), sem)); the FSQ quantisation of z is `IntraNoAR.compress`."""

    def __init__(self, sd, device, N=128, semantic_ch=768, internal_ch=512, p="hyper_enc"):
        self.N, self.semantic_ch = N, semantic_ch
        self.t0 = DepthConvBlock4(sd, f"{p}.y_trans_coding.0", device)
        self.t1 = Conv(sd, f"{p}.y_trans_coding.1", device)
        self.t2 = DepthConvBlock4(sd, f"{p}.y_trans_coding.2", device)
        self.t3 = Conv(sd, f"{p}.y_trans_coding.3", device)
        self.f0 = DepthConvBlock4(sd, f"{p}.fusion.0", device)
        self.f1 = VQAttnBlock(sd, f"{p}.fusion.1", device, semantic_ch)
        self.f2 = DepthConvBlock4(sd, f"{p}.fusion.2", device)
        self.f3 = VQAttnBlock(sd, f"{p}.fusion.3", device, internal_ch)
        self.f4 = DepthConvBlock4(sd, f"{p}.fusion.4", device)
        self.f5 = Conv(sd, f"{p}.fusion.5", device)

    def __call__(self, y, sem, ws):
        """y: bf16 View (B,H/16,W/16,N); sem: bf16 View (B,H/64,W/64,semantic_ch) -> z f32 View (.., 7 of 8)."""
        dev = y.t.device
        z = self.t1(self.t0(y), stride=2)
        cat = new_view(sem.B, sem.H, sem.W, self.N + self.semantic_ch, dev)     # torch.cat([z, sem], 1) without a copy
        self.t3(self.t2(z), cat.slice(0, self.N), stride=2)
        K.copy_channels(sem, cat.slice(self.N, self.semantic_ch))
        h = self.f1(self.f0(cat), ws)
        h = self.f3(self.f2(h), ws)
        h = self.f4(h)
        zbuf = View(torch.zeros((h.B * h.H * h.W + K.SLACK_ROWS) * 8, dtype=torch.float32, device=dev), h.B, h.H, h.W, 8)
        self.f5(h, zbuf.slice(0, 7))
        return zbuf


class SemanticAdaptor:
    def __init__(self, sd, device, C=768, p="semantic_adaptor.to_semantic"):
        self.b0 = DepthConvBlock4(sd, f"{p}.0", device)
        self.r1 = VQResnetBlock(sd, f"{p}.1", device)
        self.a2 = VQAttnBlock(sd, f"{p}.2", device, C)
        self.a3 = VQAttnBlock(sd, f"{p}.3", device, C)
        self.r4 = VQResnetBlock(sd, f"{p}.4", device)
        self.a5 = VQAttnBlock(sd, f"{p}.5", device, C)
        self.a6 = VQAttnBlock(sd, f"{p}.6", device, C)
        self.b7 = DepthConvBlock4(sd, f"{p}.7", device)

    def __call__(self, z_semantic, ws):
        x = self.b0(z_semantic)
        x = self.a3(self.a2(self.r1(x, ws), ws), ws)
        x = self.a6(self.a5(self.r4(x, ws), ws), ws)
        return self.b7(x)


class Decoder:
    def __init__(self, sd, device, p="dec"):
        self.t0 = DepthConvBlock4(sd, f"{p}.trans_coding.0", device)
        self.t1 = DepthConvBlock4(sd, f"{p}.trans_coding.1", device)
        self.res_a = [VQResnetBlock(sd, f"{p}.blocks.{i}", device) for i in range(3)]
        self.up_a = Conv(sd, f"{p}.blocks.3", device, shuffle=True)
        self.up_b = Conv(sd, f"{p}.blocks.5", device)
        self.res_b = [VQResnetBlock(sd, f"{p}.blocks.{i}", device) for i in (6, 7, 8)]
        self.s0 = ResidualBlockUpsample(sd, f"{p}.sem_up.0", device)
        self.s1 = DepthConvBlock4(sd, f"{p}.sem_up.1", device)
        self.s2 = ResidualBlockUpsample(sd, f"{p}.sem_up.2", device)
        self.s3 = DepthConvBlock4(sd, f"{p}.sem_up.3", device)
        self.s4 = ResidualBlockUpsample(sd, f"{p}.sem_up.4", device)
        self.conv_out = DepthConvBlock4(sd, f"{p}.conv_out", device)

    def __call__(self, y_hat, sem_hat, ws):
        dev = y_hat.t.device
        y = self.t1(self.t0(y_hat))
        for r in self.res_a:
            y = r(y, ws)
        y = self.up_b(self.up_a(y), stats=True)
        # torch.cat([y, sem], dim=1) without a copy: both branches write their last layer into
        # channel slices of one buffer
        cat = new_view(y.B, y.H, y.W, 2 * y.C, dev, stats=True)
        for i, r in enumerate(self.res_b):
            y = r(y, ws, cat.slice(0, y.C) if i == 2 else None)
        s = self.s3(self.s2(self.s1(self.s0(sem_hat))))
        self.s4(s, cat.slice(y.C, y.C))
        return self.conv_out(cat)


class IntraNoAR:
    def __init__(self, sd, device, bottleneck_ch=128, semantic_ch=768, z_fsq_levels=(4,) * 7, host_threads=None,
                 prior_f32=True, prior_arith="auto", ref_threads=None, ref_profile=None):
        """prior_f32: run everything that feeds the integer decisions of the entropy coder (hyper
        decoder, prior fusion, the four-part prior's nets: 36 GFLOP of the 8047 per 768x768 picture)
        on f32 activations with f32 accumulation, like the reference's CPU path.  True (default, round 4): the products on
        the 16-bit matrix cores with every operand split into a pair of halves (22 significant bits, exact products:
        csrc/gemm_conv.hip SPLIT; a sum differs from the f32 fma chain by less than that chain's own rounding; operands must
        lie inside the half range, which the prior networks do by two orders of magnitude); "exact": the f32 matrix
        instructions, a fixed-order fp32 fma chain (rounds 1-3; what a stream written by those rounds' encoder needs);
        False = 16-bit nets like the reference under GPU autocast (SURVEY hard part 2).
        prior_arith: "mfma" (this package's own prior, encoder and decoder side alike) or "torch_cpu": the prior
        networks evaluated in the REFERENCE'S CPU arithmetic -- the fused-multiply-add chains torch-CPU's convolutions are
        made of, calibrated against the local torch (onedc_amd/ref_arith.py) -- so that y streams written by the reference
        decode here, and streams written here in this mode decode there.  ~10x slower prior; pictures above ~0.16 MP.
        "auto" (the default: the only setting under which a .bin written by the reference decodes,
        compression_model.py:369-407 / entropy_models.py:355-362, and it costs nothing on this package's own streams, which
        pass at the first try; the ENCODER side always writes with the MFMA prior in this mode): decode with the MFMA prior;
        a stream that desynchronises is tried again in the reference's arithmetic,
        calibrated for `ref_threads` / torch's current thread count and then for 8, 1 and 16 threads (the classes torch-CPU's
        kernel choice falls into), a batch of mixed producers picture by picture, before StreamDesyncError is raised.
        ref_threads: the writer's torch thread count to calibrate "torch_cpu" for (None: torch's current count).
        ref_profile: the writer's 1x1 structures where known (ref_arith.Calibration: dict or callable), e.g.
        ref_arith.native_gemm_profile() for a one-thread reference on an Intel host."""
        self.device = device
        self.act = K.act()                 # the format the weights below are packed in
        self.N = N = bottleneck_ch
        self.prior_f32 = f32 = bool(prior_f32)
        self.prior_exact = isinstance(prior_f32, str) and prior_f32.lower() == "exact"
        if prior_arith not in ("mfma", "torch_cpu", "auto"):
            raise ValueError(f"prior_arith must be 'mfma', 'torch_cpu' or 'auto', not {prior_arith!r}")
        self.prior_arith = prior_arith
        self.ref_threads = ref_threads            # torch thread count of the writer's configuration (None: torch's current one)
        if isinstance(ref_profile, str):          # YAML form: "intel:<threads>" or the path of a tools/cpu_chain_probe.py JSON
            from .ref_arith import intel_profile, load_profile
            ref_profile = intel_profile(int(ref_profile[6:])) if ref_profile.startswith("intel:") else load_profile(ref_profile)
        self.ref_profile = ref_profile            # known native-path structures of the writer (ref_arith.Calibration), or None
        self.ref_prior = None                     # the reference-arithmetic prior in use (None: the MFMA prior)
        self.last_auto_arith = None               # "auto": which arithmetic decoded the last call (mfma / mfma_exact / torch_cpu:<threads>)
        self._ref_priors = {}                     # per calibrated thread count
        import threading
        self._auto_lock = threading.RLock()       # "auto" switches self.ref_prior while it retries: one decode at a time
        self._ref_sd = sd if prior_arith != "mfma" else None
        if prior_arith == "torch_cpu":
            self.ref_prior = self._reference_prior(ref_threads)
        self._prior_nets_default = self._build_prior_nets(sd, f32, self.prior_exact)
        self._prior_nets_exact = self._prior_nets_default if (self.prior_exact or not f32) else None   # built on first use ("auto")
        self._use_prior_nets(self._prior_nets_default)
        self.semantic_adaptor = SemanticAdaptor(sd, device, semantic_ch)
        self.dec = Decoder(sd, device)
        self.z_fsq_levels = list(z_fsq_levels)
        self.index_unit_length = int(round(np.log2(np.prod(self.z_fsq_levels))))   # 14
        self.ds = 64
        self.cond_ds = 8
        self.debug = False
        # after the fourth hop every decoder must stand on the encoder's initial state with all bytes consumed
        # (onedc_rans_dec_finished); False restores the reference's behaviour: a desynchronised stream decodes to a
        # wrong picture without any error
        self.check_stream_end = True
        self.ws = Workspace(device)
        self.host_threads = host_threads or min(16, os.cpu_count() or 1)
        self.gaussian_encoder = GaussianEncoder()
        self.entropy_coder = None
        self._tables = None
        self._pinned_bufs = {}

    @K.in_own_format
    def _build_prior_nets(self, sd, f32, exact):
        """The networks that feed the entropy coder's integer decisions (hyper decoder, prior fusion, the four-part prior's
        nets) in one arithmetic: f32 activations on split-half operands (exact=False, the default since round 4) or on the f32
        matrix instructions (exact=True: rounds 1-3's arithmetic, what a stream written by those rounds' encoder needs)."""
        from . import layers as _layers
        split_before = _layers.PRIOR_SPLIT16
        _layers.PRIOR_SPLIT16 = split_before and not exact
        try:
            dev = self.device
            return SimpleNamespace(
                hyper_dec=HyperDecoder(sd, dev, f32=f32),
                y_prior_fusion=[DepthConvBlock4(sd, f"y_prior_fusion.{i}", dev, f32) for i in range(2)],
                y_spatial_prior_reduction=Conv(sd, "y_spatial_prior_reduction", dev, f32=f32),
                y_spatial_prior_adaptor=[DepthConvBlock4(sd, f"y_spatial_prior_adaptor_{i}", dev, f32) for i in (1, 2, 3)],
                y_spatial_prior=[DepthConvBlock4(sd, f"y_spatial_prior.{i}", dev, f32) for i in range(3)])
        finally:
            _layers.PRIOR_SPLIT16 = split_before

    def _use_prior_nets(self, nets):
        self.hyper_dec, self.y_prior_fusion = nets.hyper_dec, nets.y_prior_fusion
        self.y_spatial_prior_reduction = nets.y_spatial_prior_reduction
        self.y_spatial_prior_adaptor, self.y_spatial_prior = nets.y_spatial_prior_adaptor, nets.y_spatial_prior

    def _exact_prior_nets(self):
        """The exact-f32 twin of the prior networks ("auto": second attempt for a stream that desynchronises under the
        split-half form -- a stream written by a round-1-3 encoder of this package, or with prior_f32: exact)."""
        if self._prior_nets_exact is None:
            self._prior_nets_exact = self._build_prior_nets(self._ref_sd, True, True)
        return self._prior_nets_exact

    # reference CompressionModel.update (compression_model.py:169-171)
    @K.in_own_format
    def update(self, force=False):
        self.entropy_coder = EntropyCoder(False, 1)
        self.gaussian_encoder.update(force=force, entropy_coder=self.entropy_coder)
        self._tables = self.gaussian_encoder.get_cdf_info()
        want = self.ref_profile.get("gaussian_tables_sha16") if isinstance(self.ref_profile, dict) else None
        if want:                               # a writer's profile names the CDF tables its torch built: they must be ours
            import hashlib
            import numpy as np
            got = hashlib.sha256(np.asarray(self._tables[0]).astype(np.int32).tobytes()).hexdigest()[:16]
            if got != want:
                raise RuntimeError(f"the writer's Gaussian CDF tables (sha256[:16] {want}) are not the ones this host's torch builds "
                                   f"({got}): torch versions differ in Normal.cdf; its streams cannot be decoded with these tables")

    # ------------------------------------------------------------------ prior nets
    @K.in_own_format
    def _params_from_indices(self, indices):
        """indices: int32 device tensor (B, h, w) -> (fused params f32 View (B,4h,4w,2N), z_semantic)."""
        if self.ref_prior is not None:
            return self.ref_prior.params_from_indices(indices)
        B, h, w = indices.shape
        codes = View(torch.empty(B * h * w * 8, dtype=torch.float32, device=self.device), B, h, w, 8)
        K.fsq_codes(indices.contiguous(), codes.t)
        params, z_semantic = self.hyper_dec(codes)
        fused = self.y_prior_fusion[0](params)
        fused = self.y_prior_fusion[1](fused, out_f32=True)
        return fused, z_semantic

    @K.in_own_format
    def _next_params(self, k, y_hat, common_cat):
        """(scales, means) of step k >= 1 from y_hat_so_far and the reduced common params."""
        if self.ref_prior is not None:
            return self.ref_prior.next_params(k, common_cat)
        if not self.prior_f32:
            K.copy_channels(y_hat, common_cat.slice(0, self.N))           # f32 -> bf16 into the cat buffer
        a = self.y_spatial_prior_adaptor[k - 1](common_cat)
        a = self.y_spatial_prior[0](a)
        a = self.y_spatial_prior[1](a)
        return self.y_spatial_prior[2](a, out_f32=True)

    @K.in_own_format
    def _prior_setup(self, fused):
        B, H, W, N = fused.B, fused.H, fused.W, self.N
        if self.ref_prior is not None:
            return self.ref_prior.prior_setup(fused)
        if self.prior_f32:
            # [y_hat_so_far | common] is ONE f32 buffer: y_hat is its first channel slice, so
            # torch.cat((y_hat_so_far, common_params)) (compression_model.py:387) costs nothing
            common_cat = new_view(B, H, W, 2 * N, self.device, torch.float32)
            self.y_spatial_prior_reduction(fused, common_cat.slice(N, N))
            y_hat = common_cat.slice(0, N)            # step 0 of the prior update initialises it
            return common_cat, y_hat
        fused_bf = new_view(B, H, W, 2 * N, self.device)
        K.copy_channels(fused, fused_bf)
        common_cat = new_view(B, H, W, 2 * N, self.device)                 # [y_hat_so_far | common]
        self.y_spatial_prior_reduction(fused_bf, common_cat.slice(N, N))
        y_hat = new_view(B, H, W, N, self.device, torch.float32)
        return common_cat, y_hat

    # ------------------------------------------------------------------ decode
    def decompress(self, streams_y, indices, trace=None):
        """`_decompress_once`, and with prior_arith "auto" the retries in the reference's arithmetic for streams that
        desynchronise under the MFMA prior."""
        if self.prior_arith != "auto":
            return self._decompress_once(streams_y, indices, trace)
        with self._auto_lock:
            return self._decompress_auto(streams_y, indices, trace)

    def _decompress_auto(self, streams_y, indices, trace):
        cur = torch.get_num_threads()
        # the MFMA prior, then the reference's arithmetic calibrated for the configuration classes a writer may have used: the
        # stated / current thread count, 8 (2-8 threads are one class on pictures above ~0.16 MP), 1 (the reference harness's own
        # setting, src/inference.py:29: torch's native gemm path), 16 (oneDNN blocks 1x1 differently from 16 threads on)
        # Second attempt (round 5): this package's OTHER MFMA arithmetic -- a model on split-half operands (the default since
        # round 4) tries the exact-f32 matrix instructions, which is what a .bin written by rounds 1-3 of this package (or with
        # prior_f32: exact) needs; the container records no arithmetic, so the order of the attempts is the only selector.
        attempts = [None]
        if self.prior_f32 and not self.prior_exact:
            attempts.append("exact")
        for t in (self.ref_threads or cur, 8, 1, 16):
            if t not in attempts:
                attempts.append(t)
        first_error = None
        for threads in attempts:
            try:
                # (building / calibrating the reference prior for a thread count can itself fail -- no candidate structure
                # matches this host's torch -- and must not end the attempts: the next class may still decode the stream)
                if threads == "exact":
                    self._use_prior_nets(self._exact_prior_nets())
                else:
                    self.ref_prior = None if threads is None else self._reference_prior(threads)
                if trace is not None:
                    del trace[:]
                out = self._decompress_once(streams_y, indices, trace)
                self.last_auto_arith = "mfma" if threads is None else "mfma_exact" if threads == "exact" else f"torch_cpu:{threads}"
                return out
            except StreamDesyncError as e:
                first_error = first_error or e
            except (ReferenceArithmeticUnavailable, AssertionError):         # nothing more to try in that class
                if threads is None:                             # (not a calibration problem: the MFMA attempt itself)
                    raise
            finally:
                self.ref_prior = None
                self._use_prior_nets(self._prior_nets_default)
        if len(streams_y) > 1 and trace is None:
            return self._decompress_one_by_one(streams_y, indices, first_error)
        raise first_error

    def _decompress_one_by_one(self, streams_y, indices, batch_error):
        """"auto" with a batch that no single arithmetic decodes: streams of different producers side by side (this package's
        and the reference's).  Every picture goes through the attempts alone; the results are gathered into one batch.
        Pictures that still fail are named in the StreamDesyncError."""
        outs, bad = [], []
        for b, s in enumerate(streams_y):
            try:
                outs.append(self.decompress([s], indices[b:b + 1]))
            except StreamDesyncError:
                outs.append(None)
                bad.append(b)
        if bad:
            raise StreamDesyncError(bad) from batch_error
        B = len(outs)
        gathered = []
        for part in range(2):                                   # y_hat (f32), z_semantic (16 bit)
            v0 = outs[0][part]
            n = v0.H * v0.W
            allv = new_view(B, v0.H, v0.W, v0.C, self.device, v0.t.dtype)
            for b, o in enumerate(outs):
                v = o[part]
                src = v.t[:n * v.cs].view(n, v.cs)[:, v.co:v.co + v.C]
                allv.t[b * n * v.C:(b + 1) * n * v.C].view(n, v.C).copy_(src)
            gathered.append(allv)
        return gathered[0], gathered[1]

    @K.in_own_format
    @torch.no_grad()
    def _decompress_once(self, streams_y, indices, trace=None):
        """decompress_four_part_prior (compression_model.py:369-407) for a batch.
        streams_y: list of B rANS byte strings; indices: int32 (B,h,w) device tensor.
        trace: optional list; receives per hop (table indexes, symbols) as int16 arrays [B, 32, H, W] (what crossed
        the host boundary), also for the hops completed before a StreamDesyncError."""
        B = len(streams_y)
        fused, z_semantic = self._params_from_indices(indices)
        common_cat, y_hat = self._prior_setup(fused)
        H, W, N = fused.H, fused.W, self.N
        n = (N // 4) * H * W
        decoders = []
        for s in streams_y:
            d = rans.RansDecoder(1)
            d.add_cdf(*self._tables)
            d.set_stream(s)
            decoders.append(d)
        idx_dev = torch.empty(B * n, dtype=torch.int16, device=self.device)
        sym_dev = torch.empty(B * n, dtype=torch.int16, device=self.device)
        idx_host, sym_host = self._pinned(B, n)
        params = fused
        for k in range(4):
            if k > 0:
                params = self._next_params(k, y_hat, common_cat)
            self._prior_indexes(params, idx_dev, B, H, W, k)
            idx_host.copy_(idx_dev.view(B, n), non_blocking=True)
            torch.cuda.current_stream().synchronize()
            try:
                sym_host.numpy()[...] = rans.decode_batch(decoders, idx_host.numpy(), 0, self.host_threads)
            except rans.RansStreamError as e:       # a stream ran dry in hop k: the same desynchronisation, seen early
                raise StreamDesyncError([b for b, st in enumerate(e.status) if st != 0]) from e
            if trace is not None:
                trace.append((idx_host.numpy().reshape(B, N // 4, H, W).copy(),
                              sym_host.numpy().reshape(B, N // 4, H, W).copy()))
            sym_dev.view(B, n).copy_(sym_host, non_blocking=True)
            K.prior_update_dec(params, sym_dev, y_hat, k)
        if self.check_stream_end:
            bad = [b for b, d in enumerate(decoders) if not d.finished()]
            if bad:
                raise StreamDesyncError(bad)
        return y_hat, z_semantic

    def _reference_prior(self, threads=None):
        import torch as _t
        from .ref_arith import ReferencePrior
        threads = int(threads or _t.get_num_threads())
        if threads not in self._ref_priors:
            self._ref_priors[threads] = ReferencePrior(self._ref_sd, self.device, self.N, threads, self.ref_profile)
        return self._ref_priors[threads]

    def _prior_indexes(self, params, idx_dev, B, H, W, k):
        """build_indexes of hop k (entropy_models.py:355-362): with this package's logarithm, or -- prior_arith "torch_cpu" --
        through thresholds taken from the reference's own (torch-CPU) logarithm."""
        if self.ref_prior is not None:
            return self.ref_prior.indexes(params, idx_dev, B, H, W, k)
        return K.prior_indexes(params, idx_dev, B, H, W, self.N, k)

    def _pinned(self, B, n):
        """Page-locked staging buffers of the rANS hops, one pair per (thread, shape), allocated once."""
        import threading
        key = (threading.get_ident(), B, n)
        bufs = self._pinned_bufs.get(key)
        if bufs is None:
            bufs = (torch.empty(B, n, dtype=torch.int16).pin_memory(), torch.empty(B, n, dtype=torch.int16).pin_memory())
            self._pinned_bufs[key] = bufs
        return bufs

    def _prior_guard(self):
        """Everything that evaluates the prior networks holds this while it does: "auto" decoding swaps self.ref_prior and the
        whole prior stack (hyper_dec, y_prior_fusion, y_spatial_prior*) to another arithmetic while it retries a desynchronised
        stream, and an encode (or a z-only reconstruction) on another thread must neither pick up the swapped networks -- it
        would write a stream in the wrong arithmetic -- nor be overtaken by the swap half way.  Re-entrant."""
        import contextlib
        return self._auto_lock if self.prior_arith == "auto" else contextlib.nullcontext()

    def recon_with_z(self, indices):
        with self._prior_guard():
            return self._recon_with_z(indices)

    @K.in_own_format
    @torch.no_grad()
    def _recon_with_z(self, indices):
        """z-only model: forward_four_part_prior_recon_with_z (compression_model.py:421-465)."""
        fused, z_semantic = self._params_from_indices(indices)
        common_cat, y_hat = self._prior_setup(fused)
        params = fused
        for k in range(4):
            if k > 0:
                params = self._next_params(k, y_hat, common_cat)
            K.prior_update_zonly(params, y_hat, k)
        return y_hat, z_semantic

    @K.in_own_format
    @torch.no_grad()
    def synthesis(self, y_hat, z_semantic):
        """codec_module.py:452-453: y_semantic = semantic_adaptor(z_semantic); x_hat = dec(y_hat, y_semantic)."""
        y_bf = new_view(y_hat.B, y_hat.H, y_hat.W, self.N, self.device)
        K.copy_channels(y_hat, y_bf)
        y_semantic = self.semantic_adaptor(z_semantic, self.ws)
        x_hat = self.dec(y_bf, y_semantic, self.ws)
        return x_hat, y_semantic

    @K.in_own_format
    @torch.no_grad()
    def decode(self, fp=None, stream=None):
        """Reference signature (codec_module.py:356-369), one stream or a list of streams."""
        y_hat, z_semantic, orig_hw, pad_hw, pad = self.decode_entropy(fp=fp, stream=stream)
        x_hat, y_semantic = self.synthesis(y_hat, z_semantic)
        return x_hat, y_semantic, orig_hw, pad_hw, pad

    @K.in_own_format
    @torch.no_grad()
    def decode_entropy(self, fp=None, stream=None):
        """The entropy stage of `decode`: container parse, z unpack, prior networks and the four host
        rANS hops -> (y_hat, z_semantic, sizes).  Split from the synthesis so that a pipeline can run
        it for the next batch while the previous one is in the UNet / VAE (DecodePipeline)."""
        assert fp or stream
        blobs = stream if stream else [read_from_file(f) for f in ([fp] if isinstance(fp, str) else fp)]
        if isinstance(blobs, (bytes, bytearray)):
            blobs = [blobs]
        infos = [decode_i(b, self.index_unit_length, self.ds) for b in blobs]
        hw = {(d["pad_height"], d["pad_width"]) for d in infos}
        if len(hw) != 1:
            raise ValueError("a batch must share one padded picture size")
        pad_h, pad_w = hw.pop()
        zh, zw = pad_h // self.ds, pad_w // self.ds
        idx = np.stack([unpack_indices(d["bit_stream_z"], zh * zw, self.index_unit_length) for d in infos])
        indices = torch.from_numpy(idx.reshape(len(blobs), zh, zw)).to(self.device)
        y_hat, z_semantic = self.decompress([d["bit_stream_y"] for d in infos], indices)
        orig_hw = [(d["height"], d["width"]) for d in infos]
        return y_hat, z_semantic, orig_hw, (pad_h, pad_w), infos[0]["pad_tuple"]

    @K.in_own_format
    @torch.no_grad()
    def decode_entropy_z_only(self, fp=None, stream=None):
        """The ex-low-rate model's entropy stage (sd15_onedc_codec_z_only: y is never transmitted): containers whose y
        part is empty -- 16 B header + ceil(14 n / 8) B of FSQ indices, 252 B of payload at 768x768 -- -> the same tuple as
        `decode_entropy`, with y_hat = the prior's means (forward_four_part_prior_recon_with_z,
        compression_model.py:421-465).  No rANS hop, no host synchronisation."""
        assert fp or stream
        blobs = stream if stream else [read_from_file(f) for f in ([fp] if isinstance(fp, str) else fp)]
        if isinstance(blobs, (bytes, bytearray)):
            blobs = [blobs]
        infos = [decode_i(b, self.index_unit_length, self.ds) for b in blobs]
        hw = {(d["pad_height"], d["pad_width"]) for d in infos}
        if len(hw) != 1:
            raise ValueError("a batch must share one padded picture size")
        if any(len(d["bit_stream_y"]) for d in infos):
            raise ValueError("a z-only container carries no y stream: decode it with decode_entropy / decode")
        pad_h, pad_w = hw.pop()
        zh, zw = pad_h // self.ds, pad_w // self.ds
        idx = np.stack([unpack_indices(d["bit_stream_z"], zh * zw, self.index_unit_length) for d in infos])
        indices = torch.from_numpy(idx.reshape(len(blobs), zh, zw)).to(self.device)
        y_hat, z_semantic = self.recon_with_z(indices)
        return y_hat, z_semantic, [(d["height"], d["width"]) for d in infos], (pad_h, pad_w), infos[0]["pad_tuple"]

    # ------------------------------------------------------------------ encode side
    @K.in_own_format
    def load_hyper_encoder(self, sd):
        """The hyper analysis weights (`hyper_enc.*` of model_1.safetensors; a decode-only load skips them)."""
        self.hyper_enc = HyperEncoder(sd, self.device, self.N)
        return self

    @K.in_own_format
    def load_encoder(self, sd):
        """The analysis transform (`enc.*` of model_1.safetensors)."""
        from .encoder import CodecEncoder
        self.enc = CodecEncoder(sd, self.device)
        return self

    @K.in_own_format
    @torch.no_grad()
    def encode(self, x, cond, pic_width, pic_height, fp=None):
        """Reference `IntraNoAR.encode` (codec_module.py:326-353) for a batch of equally sized pictures:
        x: f32 NHWC View of the padded pictures, cond: f32 View of their VAE latents.
        -> (list of container bytes, list of bpp dicts with the reference's keys :342-352)."""
        if getattr(self, "enc", None) is None:
            raise RuntimeError("encode needs the analysis transform: call load_encoder(state_dict)")
        y, sem = self.enc(x, cond)
        blobs, _, _ = self.compress(y, sem, heights=[pic_height] * x.B, widths=[pic_width] * x.B)
        out = []
        for i, blob in enumerate(blobs):
            if fp:
                write_to_file(blob, fp if isinstance(fp, str) else fp[i])
            d = decode_i(blob, self.index_unit_length, self.ds)
            n = pic_height * pic_width
            bits_y, bits_z, bits_c = (len(d[k]) * 8 for k in ("bit_stream_y", "bit_stream_z", "bit_stream_caption"))
            out.append({"bits_y": bits_y, "bits_z": bits_z, "bits_caption": bits_c, "bits_total": len(blob) * 8,
                        "bpp_y": bits_y / n, "bpp_z": bits_z / n, "bpp_caption": bits_c / n, "bpp": len(blob) * 8 / n,
                        "bpp_file": (len(blob) * 8 / n) if fp else 0.0})
        return blobs, out

    @K.in_own_format
    @torch.no_grad()
    def compress(self, y, sem, heights=None, widths=None):
        """codec_module.py:372-415 after the analysis transform `enc` (SD-VAE encoder + UNet2DModel: not
        built yet): hyper analysis, FSQ quantisation, four-part prior, rANS, index packing, container.
        y: f32 NHWC device tensor (B,H/16,W/16,N); sem: bf16/f32 NHWC tensor (B,H/64,W/64,768).
        -> (list of container bytes, indices int32 (B,h,w), y_hat View)."""
        if getattr(self, "hyper_enc", None) is None:
            raise RuntimeError("compress needs the hyper encoder weights: call load_hyper_encoder(state_dict)")
        if isinstance(y, View):                               # straight from the analysis transform (bf16 Views)
            yb, sb = y, sem
            B, H, W, N = y.B, y.H, y.W, y.C
            yf = new_view(B, H, W, N, self.device, torch.float32)
            K.copy_channels(yb, yf)
            y = yf.t[:B * H * W * N].view(B, H, W, N)
        else:
            B, H, W, N = y.shape
            yb = View(torch.empty((B * H * W + K.SLACK_ROWS) * N, dtype=K.a16(), device=self.device), B, H, W, N)
            yb.t[:B * H * W * N] = y.reshape(-1).to(K.a16())
            sb = View(torch.empty((sem.shape[0] * sem.shape[1] * sem.shape[2] + K.SLACK_ROWS) * sem.shape[3],
                                  dtype=K.a16(), device=self.device), sem.shape[0], sem.shape[1], sem.shape[2],
                      sem.shape[3])
            sb.t[:sem.numel()] = sem.reshape(-1).to(K.a16())
        z = self.hyper_enc(yb, sb, self.ws)
        indices = torch.empty(B * z.H * z.W, dtype=torch.int32, device=self.device)
        K.fsq_quantize(z.slice(0, 7), indices)
        indices = indices.view(B, z.H, z.W)
        blobs, y_hat = self.encode_latents(indices, y.contiguous().float(), heights=heights, widths=widths)
        return blobs, indices, y_hat

    # ------------------------------------------------------------------ encode side of the prior
    def encode_latents(self, indices, y=None, noise_seed=0, heights=None, widths=None):
        with self._prior_guard():
            return self._encode_latents(indices, y, noise_seed, heights, widths)

    @K.in_own_format
    @torch.no_grad()
    def _encode_latents(self, indices, y=None, noise_seed=0, heights=None, widths=None):
        """compress_four_part_prior + entropy encode (codec_module.py:372-415 minus the analysis
        transform): indices int32 (B,h,w); y f32 NHWC (B,4h,4w,N) device tensor, or None to draw a
        synthetic latent step by step as means + clamp(scales, .11, 8) * N(0,1).
        Returns (list of container bytes, y_hat View)."""
        B, zh, zw = indices.shape
        fused, _ = self._params_from_indices(indices)
        common_cat, y_hat = self._prior_setup(fused)
        H, W, N = fused.H, fused.W, self.N
        n = (N // 4) * H * W
        encoders = []
        for _ in range(B):
            e = rans.RansEncoder(False, 1)
            e.add_cdf(*self._tables)
            encoders.append(e)
        idx_dev = torch.empty(B * n, dtype=torch.int16, device=self.device)
        sym_dev = torch.empty(B * n, dtype=torch.int16, device=self.device)
        gen = torch.Generator(device=self.device).manual_seed(noise_seed)
        params = fused
        for k in range(4):
            if k > 0:
                params = self._next_params(k, y_hat, common_cat)
            if y is None:      # synthetic latent for this step's positions (bench / test input only)
                pv = params.t[:B * H * W * 2 * N].view(B, H, W, 2 * N)
                y_k = pv[..., N:] + pv[..., :N].clamp(0.11, 8.0) * torch.randn(
                    (B, H, W, N), generator=gen, device=self.device)
                y_k = y_k.contiguous()
            else:
                y_k = y
            self._prior_indexes(params, idx_dev, B, H, W, k)
            K.prior_update_enc(params, y_k, sym_dev, y_hat, k)
            torch.cuda.current_stream().synchronize()
            rans.encode_batch(encoders, sym_dev.view(B, n).cpu().numpy(), idx_dev.view(B, n).cpu().numpy(), 0,
                              self.host_threads)
        blobs = []
        idx_host = indices.cpu().numpy()
        for b, e in enumerate(encoders):
            e.flush()
            ybytes = e.get_encoded_stream().tobytes()
            zbytes = pack_indices(idx_host[b].reshape(-1), self.index_unit_length)
            hh = heights[b] if heights else zh * self.ds
            ww = widths[b] if widths else zw * self.ds
            blobs.append(encode_i(hh, ww, ybytes, zbytes, b"", 0))
        return blobs, y_hat
