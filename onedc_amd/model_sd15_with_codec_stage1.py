"""SD15_1step_codec_stage1 — drop-in decoder with the reference's model surface.

Mirrors reference src/models/sd15_onedc_codec_stage1/model_sd15_with_codec_stage1.py:
`decode(fp=None, stream=None)` :295-330, `vae_decode_image` :184-188, the DDIM alphas_cumprod of
:103-106 and the epsilon -> x0 step :320-322, plus `eval()` / `prepare()` and the
`codec_model.update(force=True)` call sequence of src/inference.py:69-74.  Differences, all
additive: `decode` also accepts lists (a batch of bitstreams of one padded size), and the SD-2.1
VAE weights come from a local file or state dict because the reference downloads them from the
HF hub (:43-45) which does not exist offline.
"""
import os
from types import SimpleNamespace

import torch

from . import kernels as K
from . import synth
from .codec_module import IntraNoAR
from .decoder_unet import UNet
from .kernels import View, new_view
from .vae import VAEDecoder


def load_safetensor(path, map_location="cpu"):
    """reference src/utils.py:25-33"""
    if path.endswith("safetensors"):
        from safetensors import safe_open
        tensors = {}
        with safe_open(path, framework="pt", device=map_location) as f:
            for k in f.keys():
                tensors[k] = f.get_tensor(k)
        return tensors
    return torch.load(path, map_location=map_location)


def ddim_alphas_cumprod(num_train_timesteps=1000, beta_start=0.00085, beta_end=0.012):
    """DDIMScheduler('scaled_linear') of runwayml/stable-diffusion-v1-5 (:103-106)."""
    betas = torch.linspace(beta_start ** 0.5, beta_end ** 0.5, num_train_timesteps, dtype=torch.float32) ** 2
    return torch.cumprod(1.0 - betas, dim=0)


def _get(args, name, default=None):
    if isinstance(args, dict):
        return args.get(name, default)
    return getattr(args, name, default)


class SD15_1step_codec_stage1:
    def __init__(self, args, accelerator=None, device=None):
        if not torch.cuda.is_available():
            raise RuntimeError("onedc_amd needs a ROCm device: the decode path has no CPU implementation")
        if device is None:
            device = getattr(accelerator, "device", None) or torch.device("cuda", torch.cuda.current_device())
        self.device = torch.device(device)
        self.args = args
        self.accelerator = accelerator
        # 16-bit storage / MFMA operand format of the device path: "f16" (default: the 0.01 dB budget needs its 11-bit
        # mantissa) or "bf16" (the reference's autocast format; unlimited range)
        self.act = _get(args, "act_dtype", None) or K.DEFAULT_ACT
        self.use_fp16 = _get(args, "use_fp16", True)
        self.vae_attn_patch = _get(args, "vae_attn_patch", 16)
        self.lora_config = _get(args, "lora_config", {"lora_rank": 64, "lora_alpha": 8.0})
        self.conditioning_timestep = _get(args, "conditioning_timestep", 999)
        self.num_train_timesteps = _get(args, "num_train_timesteps", 1000)
        self.alphas_cumprod = ddim_alphas_cumprod(self.num_train_timesteps)
        self.feedforward_model = None
        self.codec_model = None
        self.vae_large = None
        # f16_range_fallback: a batch whose f16 decode trips the range guard (kernels.HalfRangeError) is decoded again by a
        # bf16 twin of this model (same weights, bf16 storage: f32's exponent range, 8x coarser rounding) instead of raising.
        # The twin is built on first use from where the weights came from (checkpoint files / the synthetic generator).
        self.range_fallback = bool(_get(args, "f16_range_fallback", False))
        self._weight_source = None
        self._twin = None

    # -- weights -------------------------------------------------------------------------------
    def load_state_dicts(self, feedforward_sd, codec_sd, vae_sd):
        with K.act_format(self.act):
            return self._load_state_dicts(feedforward_sd, codec_sd, vae_sd)

    def _load_state_dicts(self, feedforward_sd, codec_sd, vae_sd):
        lc = self.lora_config or {}
        rank = _get(lc, "lora_rank", 64)
        alpha = _get(lc, "lora_alpha", 8.0)
        with torch.cuda.device(self.device):
            # unet_act: the UNet alone in another 16-bit format (round 5 experiment: "bf16" inside an f16 model -- the chip
            # holds a higher clock on 8-bit mantissas; tests/devtools/error_budget.py: a bf16 UNet alone sits 60.7 dB from the
            # oracle where the whole bf16 model sits 49.7)
            with K.act_format(_get(self.args, "unet_act", None) or self.act):
                self.feedforward_model = UNet(feedforward_sd, self.device, rank, alpha)
            self.codec_model = IntraNoAR(codec_sd, self.device, prior_f32=_get(self.args, "prior_f32", True),
                                         prior_arith=_get(self.args, "prior_arith", "auto") or "auto",
                                         ref_threads=_get(self.args, "ref_threads", None),
                                         ref_profile=_get(self.args, "ref_profile", None))
            self.vae_large = VAEDecoder(vae_sd, self.device, self.vae_attn_patch, up3_act=_get(self.args, "vae_up3_act", None))
        return self

    def load(self, checkpoint_path, vae_path=None):
        """inference.py:87-93 + the VAE file the reference fetches from the hub.  The ex-low-rate model
        (models/sd15_onedc_codec_z_only) uses the same two files with the same key layout: its IntraNoAR declares the
        same sub-modules as the stage-1 one (only `forward` differs), so its checkpoints load here unchanged."""
        ff = load_safetensor(os.path.join(checkpoint_path, "model.safetensors"))
        codec = load_safetensor(os.path.join(checkpoint_path, "model_1.safetensors"))
        vae_path = vae_path or os.path.join(checkpoint_path, "vae", "diffusion_pytorch_model.safetensors")
        vae = load_safetensor(vae_path)
        check_state_dicts(ff, codec, vae)
        self.load_state_dicts(ff, codec, vae)
        self._weight_source = lambda: (load_safetensor(os.path.join(checkpoint_path, "model.safetensors")),
                                       load_safetensor(os.path.join(checkpoint_path, "model_1.safetensors")),
                                       load_safetensor(vae_path))
        # encode side, when the files carry it (model_1.safetensors always does: enc.*, hyper_enc.*)
        if "enc.pix_emb.weight" in codec and "encoder.conv_in.weight" in vae:
            self.load_encoder_state_dicts(codec, codec, vae)
        return self

    def load_synthetic(self, seed=0, preset=None):
        """Deterministic synthetic weights of the real architecture (no checkpoints offline).  preset: "random_init"
        (default) or "trained_like" (residual branches damped so that x0 rms ~1 and the prior stays inside its table:
        the operating point of a trained checkpoint, onedc_amd/synth.py)."""
        self._weight_source = lambda: (synth.unet_state_dict(seed, preset=preset), synth.codec_state_dict(seed, preset=preset),
                                       synth.vae_state_dict(seed, preset=preset))
        return self.load_state_dicts(*self._weight_source())

    def bf16_twin(self):
        """This model with bf16 activations (built once, on first use): the decoder a batch goes to when its f16 decode left
        the half-precision range and `f16_range_fallback` is set."""
        if self._twin is None:
            if self._weight_source is None:
                raise RuntimeError("no weight source to build the bf16 twin from (load() / load_synthetic() record one)")
            args = dict(self.args) if isinstance(self.args, dict) else dict(vars(self.args))
            args["act_dtype"] = "bf16"
            args["f16_range_fallback"] = False
            twin = SD15_1step_codec_stage1(args, device=self.device).load_state_dicts(*self._weight_source())
            twin.codec_model.update(force=True)
            self._twin = twin
        return self._twin

    def load_encoder_state_dicts(self, codec_enc_sd, hyper_enc_sd, vae_enc_sd):
        with K.act_format(self.act):
            return self._load_encoder_state_dicts(codec_enc_sd, hyper_enc_sd, vae_enc_sd)

    def _load_encoder_state_dicts(self, codec_enc_sd, hyper_enc_sd, vae_enc_sd):
        """Encode side: `enc.*` and `hyper_enc.*` of model_1.safetensors, encoder half of the SD-2.1 VAE."""
        from .encoder import VAEEncoder
        with torch.cuda.device(self.device):
            self.codec_model.load_encoder(codec_enc_sd)
            self.codec_model.load_hyper_encoder(hyper_enc_sd)
            self.vae_encoder = VAEEncoder(vae_enc_sd, self.device, self.vae_attn_patch)
        return self

    def load_synthetic_encoder(self, seed=0):
        return self.load_encoder_state_dicts(synth.codec_encoder_state_dict(seed), synth.hyper_enc_state_dict(seed),
                                             synth.vae_encoder_state_dict(seed))

    def encode(self, image, fp=None, noise=None):
        with K.act_format(self.act):
            return self._encode(image, fp=fp, noise=noise)

    @torch.no_grad()
    def _encode(self, image, fp=None, noise=None):
        """Reference `encode` (model_sd15_with_codec_stage1.py:275-292): image f32 [B,3,H,W] in [-1,1] on this
        device (B = 1 in the reference) -> (stream bytes, bpp dict) (lists when B > 1).  Replicate padding up to a
        multiple of 64, SD-VAE encoder, analysis transform, hyper analysis, entropy coding.
        noise: f32 [B,Hp/8,Wp/8,4] standing in for `latent_dist.sample()`'s global-RNG draw (:193; seed 0 in
        inference.py:28); None encodes with the distribution's mode."""
        if getattr(self, "vae_encoder", None) is None:
            raise RuntimeError("encode needs the encoder weights: load_encoder_state_dicts / load_synthetic_encoder")
        B, _, H, W = image.shape
        ds = self.codec_model.ds
        Hp, Wp = (H + ds - 1) // ds * ds, (W + ds - 1) // ds * ds
        with torch.cuda.device(self.device):
            img = View(torch.zeros((B * Hp * Wp + K.SLACK_ROWS) * 32, dtype=torch.float32, device=self.device),
                       B, Hp, Wp, 32)
            K.nchw_to_nhwc(image.to(self.device, torch.float32).contiguous(), img.slice(0, 3))
            latent = self.vae_encoder.encode(img, noise)
            blobs, bpp = self.codec_model.encode(img, latent, W, H, fp=fp)
        return (blobs[0], bpp[0]) if B == 1 else (blobs, bpp)

    def forward_z_only(self, image, noise=None):
        with K.act_format(self.act):
            return self._forward_z_only(image, noise=noise)

    @torch.no_grad()
    def _forward_z_only(self, image, noise=None):
        """The ex-low-rate model's harness (models/sd15_onedc_codec_z_only/inference.py:100-131 around
        model_sd15_with_codec_stage1.py:198-218): reflect padding to a multiple of 64, analysis -> FSQ indices, the
        picture is reconstructed from the indices alone (four-part prior run with means only, no y stream).
        -> (dict with 'indices', 'stream' = the container with an empty y part: 16 B header + ceil(n*14/8) B of
        indices -- the reference never writes one --, 'bpp_z' as the harness computes it, pictures [B,3,H,W])."""
        if getattr(self, "vae_encoder", None) is None:
            raise RuntimeError("forward_z_only needs the encoder weights")
        from .entropy.utils import encode_i, pack_indices
        B, _, H, W = image.shape
        Hp, Wp = (H + 63) // 64 * 64, (W + 63) // 64 * 64
        with torch.cuda.device(self.device):
            img = View(torch.zeros((B * Hp * Wp + K.SLACK_ROWS) * 32, dtype=torch.float32, device=self.device),
                       B, Hp, Wp, 32)
            K.nchw_to_nhwc(image.to(self.device, torch.float32).contiguous(), img.slice(0, 3), reflect=True)
            latent = self.vae_encoder.encode(img, noise)
            cm = self.codec_model
            y, sem = cm.enc(img, latent)
            z = cm.hyper_enc(y, sem, cm.ws)
            indices = torch.empty(B * z.H * z.W, dtype=torch.int32, device=self.device)
            K.fsq_quantize(z.slice(0, 7), indices)
            indices = indices.view(B, z.H, z.W)
            recon = self.decode_z_only(indices)[:, :, :H, :W]
        idx_host = indices.cpu().numpy()
        streams = [encode_i(H, W, b"", pack_indices(idx_host[b].reshape(-1), cm.index_unit_length), b"", 0)
                   for b in range(B)]
        bits = (Hp // 64) * (Wp // 64) * 14
        # bpp_hard_y: the harness's second column (z_only/inference.py:126).  In eval mode the z-only codec sets y_q = 0 and
        # sigma = 0 (z_only/codec_module.py:252-254), so every symbol has probability 1 and the estimate is exactly 0
        # (compression_model.py:137-150: -log2(1 + 1e-5) lower-bounded at 0)
        return {"indices": indices, "stream": streams, "bpp_z": bits / (H * W), "bpp_hard_y": 0.0}, recon

    def prepare(self):
        return self

    def eval(self):
        return self

    # -- decode ---------------------------------------------------------------------------------
    def decode_latents(self, h_detail_hat, y_semantic, vae_gate=None):
        with K.act_format(self.act):
            return self._decode_latents(h_detail_hat, y_semantic, vae_gate)

    @torch.no_grad()
    def _decode_latents(self, h_detail_hat, y_semantic, vae_gate=None):
        """model...stage1.py:303-324 after the codec: UNet step, x0, VAE decode. Returns dict of Views.
        vae_gate: optional context manager entered around the VAE decoder's launches (DecodePipeline serialises the VAE
        stages of its streams with it)."""
        ctx = y_semantic                                            # 'b c h w -> b (h w) c' is the NHWC layout itself
        if self.feedforward_model.act != self.act:                  # (experiment: the UNet in the other 16-bit format)
            dt = torch.float16 if self.feedforward_model.act == "f16" else torch.bfloat16
            h_detail_hat, ctx = (View(v.t.to(dt), v.B, v.H, v.W, v.C, v.cs, v.co) for v in (h_detail_hat, ctx))
        eps, reduced = self.feedforward_model(h_detail_hat, self.conditioning_timestep, ctx)
        self._mark("unet")
        x0 = new_view(eps.B, eps.H, eps.W, 4, self.device, torch.float32)
        n = eps.B * eps.H * eps.W * 4
        K.x0_from_noise(reduced.t[:n], eps.t[:n], x0.t[:n], float(self.alphas_cumprod[self.conditioning_timestep].double()))
        if vae_gate is None:
            image = self.vae_large.decode(x0)
        else:
            with vae_gate:
                image = self.vae_large.decode(x0)
        return {"eps": eps, "reduced": reduced, "x0": x0, "image": image}

    def decode(self, fp=None, stream=None, unit_range=False, out="f32"):
        with K.act_format(self.act):
            return self._decode(fp=fp, stream=stream, unit_range=unit_range, out=out)

    @torch.no_grad()
    def _decode(self, fp=None, stream=None, unit_range=False, out="f32"):
        """-> f32 tensor [B,3,H,W] (B = 1 for a single path/bytes, as in the reference); out="u8": the 8-bit samples
        [B,H,W,3] that `save_image` would write (inference.py:106-107), converted on the device.
        Raises kernels.HalfRangeError when a picture holds NaN / Inf (f16 range exceeded) unless `check_finite` is off."""
        assert fp or stream
        with torch.cuda.device(self.device):
            entropy_out = self.codec_model.decode_entropy(fp=fp, stream=stream)
            res, flags = self._decode_after_entropy(entropy_out, unit_range, out=out)
            try:
                self._raise_if_nonfinite(flags)
            except K.HalfRangeError:
                if not (self.range_fallback and self.act == "f16"):
                    raise
                return self.bf16_twin().decode(fp=fp, stream=stream, unit_range=unit_range, out=out)
            return res

    check_finite = True      # model.decode() / DecodePipeline.map() raise HalfRangeError on NaN / Inf pictures

    def _raise_if_nonfinite(self, flags):
        """flags: int32 [B] written by the output kernel (device, or already on the host)."""
        if flags is None or not self.check_finite:
            return
        bad = torch.nonzero(flags.cpu()).flatten().tolist()         # (a device tensor synchronises its stream here)
        if bad:
            raise K.HalfRangeError(bad, self.act)

    @torch.no_grad()
    def _decode_after_entropy(self, entropy_out, unit_range=False, vae_gate=None, out="f32"):
        """Synthesis transform, UNet step, x0, VAE decoder, crop: everything after the rANS hops.
        -> (pictures, non-finite flags int32 [B] on the device)."""
        y_hat, z_semantic, orig_hw, pad_hw, pad_size = entropy_out
        self._mark("start")
        h_detail_hat, y_semantic = self.codec_model.synthesis(y_hat, z_semantic)
        self._mark("synthesis")
        image = self.decode_latents(h_detail_hat, y_semantic, vae_gate)["image"]
        self._mark("vae")
        H, W = orig_hw[0]
        if any(hw != (H, W) for hw in orig_hw):
            H, W = pad_hw                                            # mixed crops: return the padded frame
        res = self._image_out(image, H, W, unit_range, out)
        self._mark("out")
        return res

    stage_marks = None       # bench.py / tools: a list here collects (name, HIP event) at the stage boundaries of stage S

    def _mark(self, name):
        if self.stage_marks is not None:
            e = torch.cuda.Event(enable_timing=True)
            e.record()
            self.stage_marks.append((name, e))

    def _image_out(self, image, H, W, unit_range, out="f32"):
        flags = torch.zeros(image.B, dtype=torch.int32, device=self.device)
        if out == "u8":
            res = torch.empty(image.B, H, W, 3, dtype=torch.uint8, device=self.device)
            K.image_out_u8(image, res, H, W, True, flags)
        elif out == "f32":
            res = torch.empty(image.B, 3, H, W, dtype=torch.float32, device=self.device)
            K.image_out(image, res, H, W, unit_range, flags)
        else:
            raise ValueError(f"out must be 'f32' or 'u8', not {out!r}")
        return res, flags

    def pipeline(self, depth=2, s_streams=2, z_only=False, serialise_vae=True):
        """z_only: the batches are indices-only containers of the ex-low-rate model (`decode_z_only_stream`)."""
        return DecodePipeline(self, depth, s_streams, z_only, serialise_vae)

    def decode_z_only_stream(self, fp=None, stream=None, unit_range=False, out="f32"):
        """Decoder of the ex-low-rate model from files / byte strings: the 16 B + ceil(14 n / 8) B containers written by
        `forward_z_only` / `python -m onedc_amd.inference_z_only` (the reference has no such entry: its z-only harness
        runs `forward()` and never writes a stream, models/sd15_onedc_codec_z_only/inference.py:125)."""
        with K.act_format(self.act), torch.no_grad(), torch.cuda.device(self.device):
            res, flags = self._decode_after_entropy(self.codec_model.decode_entropy_z_only(fp=fp, stream=stream),
                                                    unit_range, out=out)
            try:
                self._raise_if_nonfinite(flags)
            except K.HalfRangeError:
                if not (self.range_fallback and self.act == "f16"):
                    raise
                return self.bf16_twin().decode_z_only_stream(fp=fp, stream=stream, unit_range=unit_range, out=out)
            return res

    def decode_z_only(self, indices):
        with K.act_format(self.act):
            return self._decode_z_only(indices)

    @torch.no_grad()
    def _decode_z_only(self, indices):
        """sd15_onedc_codec_z_only forward path from FSQ indices alone
        (models/sd15_onedc_codec_z_only/model_sd15_with_codec_stage1.py:198-218)."""
        with torch.cuda.device(self.device):
            y_hat, z_semantic = self.codec_model.recon_with_z(indices.to(self.device, torch.int32))
            h_detail_hat, y_semantic = self.codec_model.synthesis(y_hat, z_semantic)
            image = self.decode_latents(h_detail_hat, y_semantic)["image"]
            res, flags = self._image_out(image, image.H, image.W, False)
            self._raise_if_nonfinite(flags)
        return res


class DecodePipeline:
    """Throughput mode for a stream of batches: a software pipeline over whole batches.

    Stage E (entropy: container parse, prior networks, the four host rANS hops, all latency-bound and
    GPU-light) of batch n+1 runs on a worker thread with its own high-priority HIP stream while stage S
    (synthesis transform, UNet step, VAE decoder: 97 % of the device time) of earlier batches runs on
    `s_streams` further worker threads, each with its own stream.  Two S streams are the default: the ~700
    kernels of one stage S form a dependent chain, and between two dependent kernels the GPU idles for the
    dispatch latency (5-10 us, 13 % of a step; a captured HIP graph replays with exactly the same gaps).  A
    second, independent chain fills them: 152 -> 142 ms per batch of 16 (tools/two_stream.py).
    Hand-overs are HIP events; results are bit-identical to `decode` (same kernels, same order per batch;
    scratch buffers are per stream, layers.Workspace).  No inter-GPU traffic: one pipeline per process / GPU."""

    def __init__(self, model, depth=2, s_streams=2, z_only=False, serialise_vae=True):
        import queue
        import threading
        self.model = model
        self.z_only = z_only
        self.dev = model.device
        with torch.cuda.device(self.dev):
            self.stream_e = torch.cuda.Stream(device=self.dev, priority=-1)
            self.streams_s = [torch.cuda.Stream(device=self.dev) for _ in range(max(1, s_streams))]
        # The VAE decoders of the stage-S streams run one after the other ON THE GPU (each waits for the previous one's
        # end-of-VAE event): two VAE decoders side by side only time-slice the same CUs -- every one of their kernels
        # fills the chip on its own -- whereas a VAE next to the other batch's synthesis / UNet (hundreds of kernels of
        # 72..720 workgroups) fills the CUs those leave idle.  No host thread blocks on it.
        self.serialise_vae = serialise_vae and len(self.streams_s) > 1
        self.timing = False          # True: the end-of-batch events carry timestamps; map() leaves them in `last_done_events`
        self.last_done_events = []
        self._vae_lock = threading.Lock()
        self._vae_done = None
        self._in = queue.Queue()
        self._mid = queue.Queue(maxsize=max(1, depth, s_streams))
        self._worker = threading.Thread(target=self._entropy_loop, daemon=True)
        self._worker.start()
        self._s_workers = [threading.Thread(target=self._synthesis_loop, args=(st,), daemon=True)
                           for st in self.streams_s]
        for t in self._s_workers:
            t.start()

    def _entropy_loop(self):
        torch.cuda.set_device(self.dev)
        K.set_act(self.model.act)
        while True:
            item = self._in.get()
            if item is None:
                for _ in self._s_workers:
                    self._mid.put(None)
                return
            blobs, job, split = item
            try:
                with torch.cuda.stream(self.stream_e), torch.no_grad():
                    cm = self.model.codec_model
                    out = cm.decode_entropy_z_only(stream=blobs) if self.z_only else cm.decode_entropy(stream=blobs)
                    ev = torch.cuda.Event()
                    ev.record(self.stream_e)
                if split:
                    # the odd batch at the end of a call would run its stage S alone on one stream: its two halves go
                    # to the two streams instead (every kernel is batch-invariant: same bits as the whole batch)
                    y_hat, z_sem, orig_hw, pad_hw, pad_size = out
                    h = y_hat.B // 2
                    slot, results, opts, done = job
                    for part, (b0, b1) in enumerate(((0, h), (h, y_hat.B))):
                        sub = (y_hat.batch_slice(b0, b1), z_sem.batch_slice(b0, b1), orig_hw[b0:b1], pad_hw, pad_size)
                        self._mid.put((sub, ev, None, ((slot, part), results, opts, done)))
                else:
                    self._mid.put((out, ev, None, job))
            except BaseException as e:      # surfaced on the caller's thread
                self._mid.put((None, None, e, job))
                if split:
                    self._mid.put((None, None, e, job))

    def _synthesis_loop(self, stream):
        torch.cuda.set_device(self.dev)
        K.set_act(self.model.act)
        while True:
            item = self._mid.get()
            if item is None:
                return
            out, ev, err, (slot, results, opts, done) = item
            unit_range, fmt, to_host = opts
            try:
                if err is not None:
                    raise err
                with torch.cuda.stream(stream), torch.no_grad():
                    stream.wait_event(ev)
                    for v in (out[0], out[1]):      # y_hat / z_semantic were allocated on the entropy stream
                        v.t.record_stream(stream)
                    res, flags = self.model._decode_after_entropy(
                        out, unit_range, _VaeGate(self, stream) if self.serialise_vae else None, out=fmt)
                    # the non-finite flags ride to the host behind the pictures (pinned: the copy is asynchronous); with
                    # to_host the pictures themselves do too -- the hand-off a service does (inference.py:106-107,119-120)
                    flags_host = torch.empty(flags.shape, dtype=flags.dtype, pin_memory=True)
                    flags_host.copy_(flags, non_blocking=True)
                    if to_host:
                        host = torch.empty(res.shape, dtype=res.dtype, pin_memory=True)
                        host.copy_(res, non_blocking=True)
                        res = host
                    ev_done = torch.cuda.Event(enable_timing=self.timing)
                    ev_done.record(stream)
                entry = (res, ev_done, stream, None, flags_host)
            except BaseException as e:
                entry = (None, None, None, e, None)
            if isinstance(slot, tuple):                  # half of a split batch
                results[slot[0]][slot[1]] = entry
            else:
                results[slot] = entry
            done.release()

    def map(self, batches, unit_range=False, out="f32", to_host=False):
        """Decode a list of batches (each a list of container byte strings) -> list of [B,3,H,W] f32 tensors, or with
        out="u8" of [B,H,W,3] uint8 tensors (save_image's samples, converted on the device); to_host=True returns them in
        page-locked host memory (the copies run on the decode streams, overlapped with the next batches).
        Raises kernels.HalfRangeError for a batch with NaN / Inf pictures (model.check_finite)."""
        import threading
        results = [None] * len(batches)
        done = threading.Semaphore(0)
        n_jobs = 0
        opts = (unit_range, out, to_host)
        for i, b in enumerate(batches):
            split = (i == len(batches) - 1 and len(batches) % 2 == 1 and len(self.streams_s) == 2 and len(b) >= 2 and
                     len(b) % 2 == 0)
            if split:
                results[i] = [None, None]
            self._in.put((b, (i, results, opts, done), split))
            n_jobs += 2 if split else 1
        for _ in range(n_jobs):
            done.acquire()
        cur = torch.cuda.current_stream(self.dev)
        outs = []
        self.last_done_events = [[p[1] for p in (e if isinstance(e, list) else [e])] for e in results]
        fallback = self.model.range_fallback and self.model.act == "f16"
        for bi, entry in enumerate(results):
            parts = entry if isinstance(entry, list) else [entry]
            tensors, first, bad = [], 0, []
            for res, ev_done, stream, err, flags_host in parts:
                if err is not None:
                    raise err
                if self.model.check_finite or to_host:
                    ev_done.synchronize()                    # the flags (and host pictures) have landed
                    if self.model.check_finite:
                        bad += [first + b for b in torch.nonzero(flags_host).flatten().tolist()]
                if not to_host:
                    cur.wait_event(ev_done)
                    res.record_stream(cur)
                tensors.append(res)
                first += res.shape[0]
            if bad and fallback:
                # f16_range_fallback, as in model.decode(): THIS batch again on the bf16 twin (sequentially: the rare
                # path); the other batches of the call keep their pictures
                twin = self.model.bf16_twin()
                dec = twin.decode_z_only_stream if self.z_only else twin.decode
                again = dec(stream=batches[bi], unit_range=unit_range, out=out)
                if to_host:
                    host = torch.empty(again.shape, dtype=again.dtype, pin_memory=True)
                    host.copy_(again)
                    again = host
                outs.append(again)
                continue
            if bad:
                raise K.HalfRangeError(bad, self.model.act)
            outs.append(tensors[0] if len(tensors) == 1 else torch.cat(tensors, 0))
        return outs

    def close(self):
        self._in.put(None)
        self._worker.join(timeout=10)
        for t in self._s_workers:
            t.join(timeout=10)
        # the layers keep scratch tensors per HIP stream (layers.Workspace): this pipeline's streams are not used again
        m = self.model
        owners = [m.feedforward_model, m.vae_large, m.codec_model, getattr(m, "vae_encoder", None),
                  getattr(getattr(m.codec_model, "enc", None), "unet", None)]
        for st in [self.stream_e] + self.streams_s:
            st.synchronize()
            for o in owners:
                ws = getattr(o, "ws", None)
                if ws is not None:
                    ws.drop_stream(st.cuda_stream)


class _VaeGate:
    """Orders the VAE stages of a DecodePipeline's streams on the device: entering makes `stream` wait for the event the
    previous VAE stage recorded at its end, leaving records this stage's.  The host lock only covers the enqueueing."""

    def __init__(self, pipe, stream):
        self.pipe, self.stream = pipe, stream

    def __enter__(self):
        self.pipe._vae_lock.acquire()
        if self.pipe._vae_done is not None:
            self.stream.wait_event(self.pipe._vae_done)

    def __exit__(self, *exc):
        ev = torch.cuda.Event()
        ev.record(self.stream)
        self.pipe._vae_done = ev
        self.pipe._vae_lock.release()
        return False


def check_state_dicts(ff, codec, vae):
    """The strict-load contract of inference.py:87-93, against arch.py's key/shape tables."""
    from . import arch
    problems = []
    for name, sd, spec, ignore in (("model.safetensors", ff, arch.unet_spec(True), ()),
                                   ("model_1.safetensors", codec, arch.codec_decode_spec(), arch.CODEC_IGNORED_PREFIXES),
                                   ("vae", vae, arch.vae_decoder_spec(), ("encoder.", "quant_conv."))):
        for k, shape in spec.items():
            alt = None
            if name == "vae":
                for old, new in arch.VAE_LEGACY_ATTN.items():
                    if f".{new}." in k:
                        alt = k.replace(f".{new}.", f".{old}.")
            t = sd.get(k, sd.get(alt) if alt else None)
            if t is None:
                problems.append(f"{name}: missing {k}")
            elif tuple(t.shape) != tuple(shape) and t.numel() != int(torch.tensor(shape).prod()):
                problems.append(f"{name}: {k} has shape {tuple(t.shape)}, expected {tuple(shape)}")
        known = set(spec)
        for k in sd:
            if k not in known and not k.startswith(ignore) and name != "vae":
                problems.append(f"{name}: unexpected {k}")
    if problems:
        raise RuntimeError("Error(s) in loading state_dict:\n  " + "\n  ".join(problems[:20]))
