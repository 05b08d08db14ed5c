"""Parameter inventory of the decode path = the reference's strict ``load_state_dict`` contract.

Three weight files feed the decoder (reference src/inference.py:87-93 and
models/sd15_onedc_codec_stage1/model_sd15_with_codec_stage1.py:43-45):

* ``model.safetensors``    SD1.5 ``UNet2DConditionModel`` + peft LoRA (r=64, alpha=8) + the new
                           320->320 ``conv_in`` and the ``vae_reduction`` block
                           (decoder_unet.py:14-29, 331-408)
* ``model_1.safetensors``  ``IntraNoAR`` codec (codec_module.py:184-229); only the decode-side
                           prefixes are enumerated here, ``enc.*`` / ``hyper_enc.*`` / ``z_vq.*``
                           are accepted and ignored by the decode-only loader
* SD-2.1 VAE               ``AutoencoderKL`` decoder half (+ ``post_quant_conv``)

Each ``*_spec`` returns ``{key: shape}`` in checkpoint naming.  The synthetic-weight generator,
the loaders and the tests are all driven from these tables.
"""
from collections import OrderedDict

LORA_RANK = 64
LORA_ALPHA = 8.0


# ----------------------------------------------------------------------------- codec blocks
def _conv(d, name, cout, cin, k, bias=True, groups=1):
    d[f"{name}.weight"] = (cout, cin // groups, k, k)
    if bias:
        d[f"{name}.bias"] = (cout,)


def _norm(d, name, c):
    d[f"{name}.weight"] = (c,)
    d[f"{name}.bias"] = (c,)


def _linear(d, name, cout, cin, bias=True):
    d[f"{name}.weight"] = (cout, cin)
    if bias:
        d[f"{name}.bias"] = (cout,)


def dcb4(d, p, cin, cout):
    """DepthConvBlock4 — reference src/modules/dcvc.py:424-433 (DepthConv :242-265, ConvFFN3 :353-367)."""
    _conv(d, f"{p}.block.0.conv1.0", cin, cin, 1)
    _conv(d, f"{p}.block.0.depth_conv", cin, cin, 3, groups=cin)
    _conv(d, f"{p}.block.0.conv2", cout, cin, 1)
    if cin != cout:
        _conv(d, f"{p}.block.0.adaptor", cout, cin, 1)
    _conv(d, f"{p}.block.1.conv", cout * 4, cout, 1)
    _conv(d, f"{p}.block.1.conv_out", cout, cout * 2, 1)


def rbu(d, p, cin, cout):
    """ResidualBlockUpsample — dcvc.py:183-208."""
    _conv(d, f"{p}.subpel_conv.0", cout * 4, cin, 1)
    _conv(d, f"{p}.conv", cout, cout, 3)
    _conv(d, f"{p}.upsample.0", cout * 4, cin, 1)


def vq_res(d, p, c):
    """ResnetBlock — src/modules/vqgan/blocks.py:15-52 (bias-free convs; in == out everywhere)."""
    _norm(d, f"{p}.norm1", c)
    _conv(d, f"{p}.conv1", c, c, 3, bias=False)
    _norm(d, f"{p}.norm2", c)
    _conv(d, f"{p}.conv2", c, c, 3, bias=False)


def vq_attn(d, p, c):
    """AttnBlock — vqgan/blocks.py:55-106."""
    _norm(d, f"{p}.norm", c)
    for n in ("q", "k", "v", "proj_out"):
        _conv(d, f"{p}.{n}", c, c, 1)


def codec_decode_spec(N=128, internal_ch=512, semantic_ch=768, ctrl_ch=320, z_dim=7):
    d = OrderedDict()
    # HyperDecoder codec_module.py:145-166
    _conv(d, "hyper_dec.feat_in.0", N, z_dim, 1)
    dcb4(d, "hyper_dec.to_entropy.0", N, N)
    rbu(d, "hyper_dec.to_entropy.1", N, N)
    dcb4(d, "hyper_dec.to_entropy.2", N, N)
    rbu(d, "hyper_dec.to_entropy.3", N, N)
    dcb4(d, "hyper_dec.to_entropy.4", N, N)
    # prior nets codec_module.py:205-217
    dcb4(d, "y_prior_fusion.0", N, 2 * N)
    dcb4(d, "y_prior_fusion.1", 2 * N, 2 * N)
    _conv(d, "y_spatial_prior_reduction", N, 2 * N, 1)
    for i in (1, 2, 3):
        dcb4(d, f"y_spatial_prior_adaptor_{i}", 2 * N, 2 * N)
    for i in range(3):
        dcb4(d, f"y_spatial_prior.{i}", 2 * N, 2 * N)
    # SemanticAdaptor codec_module.py:169-181
    s = "semantic_adaptor.to_semantic"
    dcb4(d, f"{s}.0", N, semantic_ch)
    vq_res(d, f"{s}.1", semantic_ch)
    vq_attn(d, f"{s}.2", semantic_ch)
    vq_attn(d, f"{s}.3", semantic_ch)
    vq_res(d, f"{s}.4", semantic_ch)
    vq_attn(d, f"{s}.5", semantic_ch)
    vq_attn(d, f"{s}.6", semantic_ch)
    dcb4(d, f"{s}.7", semantic_ch, semantic_ch)
    # Decoder codec_module.py:88-116
    c16, c8 = internal_ch, internal_ch // 2
    dcb4(d, "dec.trans_coding.0", N, c16)
    dcb4(d, "dec.trans_coding.1", c16, c16)
    for i in range(3):
        vq_res(d, f"dec.blocks.{i}", c16)
    _conv(d, "dec.blocks.3", c16 * 4, c16, 1)      # get_upsample :32-40 (blocks.4 = PixelShuffle)
    _conv(d, "dec.blocks.5", c8, c16, 3)
    for i in (6, 7, 8):
        vq_res(d, f"dec.blocks.{i}", c8)
    rbu(d, "dec.sem_up.0", semantic_ch, c16)
    dcb4(d, "dec.sem_up.1", c16, c16)
    rbu(d, "dec.sem_up.2", c16, c8)
    dcb4(d, "dec.sem_up.3", c8, c8)
    rbu(d, "dec.sem_up.4", c8, c8)
    dcb4(d, "dec.conv_out", c8 * 2, ctrl_ch)
    return d


def hyper_enc_spec(N=128, internal_ch=512, semantic_ch=768, z_dim=7):
    """HyperEncoder — codec_module.py:119-142 (the hyper analysis of the encode path)."""
    d = OrderedDict()
    p = "hyper_enc"
    dcb4(d, f"{p}.y_trans_coding.0", N, N)
    _conv(d, f"{p}.y_trans_coding.1", N, N, 3)              # stride 2
    dcb4(d, f"{p}.y_trans_coding.2", N, N)
    _conv(d, f"{p}.y_trans_coding.3", N, N, 3)              # stride 2
    dcb4(d, f"{p}.fusion.0", N + semantic_ch, semantic_ch)
    vq_attn(d, f"{p}.fusion.1", semantic_ch)
    dcb4(d, f"{p}.fusion.2", semantic_ch, internal_ch)
    vq_attn(d, f"{p}.fusion.3", internal_ch)
    dcb4(d, f"{p}.fusion.4", internal_ch, internal_ch)
    _conv(d, f"{p}.fusion.5", z_dim, internal_ch, 1)
    return d


CODEC_IGNORED_PREFIXES = ("enc.", "hyper_enc.", "z_vq.")


# ----------------------------------------------------------------------------- SD1.5 UNet
UNET_CH = (320, 640, 1280, 1280)
UNET_TEMB = 1280
UNET_CTX = 768
UNET_HEADS = 8


def _lora_conv(d, name, cout, cin, k, lora):
    if lora:
        d[f"{name}.base_layer.weight"] = (cout, cin, k, k)
        d[f"{name}.base_layer.bias"] = (cout,)
        d[f"{name}.lora_A.default.weight"] = (LORA_RANK, cin, k, k)
        d[f"{name}.lora_B.default.weight"] = (cout, LORA_RANK, 1, 1)
    else:
        _conv(d, name, cout, cin, k)


def _lora_linear(d, name, cout, cin, bias, lora):
    if lora:
        d[f"{name}.base_layer.weight"] = (cout, cin)
        if bias:
            d[f"{name}.base_layer.bias"] = (cout,)
        d[f"{name}.lora_A.default.weight"] = (LORA_RANK, cin)
        d[f"{name}.lora_B.default.weight"] = (cout, LORA_RANK)
    else:
        _linear(d, name, cout, cin, bias)


def unet_resnet(d, p, cin, cout, lora):
    _norm(d, f"{p}.norm1", cin)
    _lora_conv(d, f"{p}.conv1", cout, cin, 3, lora)
    _lora_linear(d, f"{p}.time_emb_proj", cout, UNET_TEMB, True, lora)
    _norm(d, f"{p}.norm2", cout)
    _lora_conv(d, f"{p}.conv2", cout, cout, 3, lora)
    if cin != cout:
        _lora_conv(d, f"{p}.conv_shortcut", cout, cin, 1, lora)


def unet_transformer(d, p, c, lora):
    _norm(d, f"{p}.norm", c)
    _lora_conv(d, f"{p}.proj_in", c, c, 1, lora)
    t = f"{p}.transformer_blocks.0"
    for n in ("norm1", "norm2", "norm3"):
        _norm(d, f"{t}.{n}", c)
    for a, kv in (("attn1", c), ("attn2", UNET_CTX)):
        _lora_linear(d, f"{t}.{a}.to_q", c, c, False, lora)
        _lora_linear(d, f"{t}.{a}.to_k", c, kv, False, lora)
        _lora_linear(d, f"{t}.{a}.to_v", c, kv, False, lora)
        _lora_linear(d, f"{t}.{a}.to_out.0", c, c, True, lora)
    _lora_linear(d, f"{t}.ff.net.0.proj", 8 * c, c, True, lora)
    _lora_linear(d, f"{t}.ff.net.2", c, 4 * c, True, lora)
    _lora_conv(d, f"{p}.proj_out", c, c, 1, lora)


def unet_spec(lora=True, in_ch=320):
    """Key/shape table of the feed-forward UNet checkpoint (decoder_unet.py:371-408)."""
    d = OrderedDict()
    _conv(d, "conv_in", UNET_CH[0], in_ch, 3)                       # replaced layer, no LoRA
    _linear(d, "time_embedding.linear_1", UNET_TEMB, UNET_CH[0])
    _linear(d, "time_embedding.linear_2", UNET_TEMB, UNET_TEMB)
    prev = UNET_CH[0]
    for i, c in enumerate(UNET_CH):
        for j in range(2):
            unet_resnet(d, f"down_blocks.{i}.resnets.{j}", prev if j == 0 else c, c, lora)
            if i < 3:
                unet_transformer(d, f"down_blocks.{i}.attentions.{j}", c, lora)
        if i < 3:
            _lora_conv(d, f"down_blocks.{i}.downsamplers.0.conv", c, c, 3, lora)
        prev = c
    unet_resnet(d, "mid_block.resnets.0", 1280, 1280, lora)
    unet_transformer(d, "mid_block.attentions.0", 1280, lora)
    unet_resnet(d, "mid_block.resnets.1", 1280, 1280, lora)
    for i, (c, ins) in enumerate(UNET_UP_IN):
        for j, cin in enumerate(ins):
            unet_resnet(d, f"up_blocks.{i}.resnets.{j}", cin, c, lora)
            if i > 0:
                unet_transformer(d, f"up_blocks.{i}.attentions.{j}", c, lora)
        if i < 3:
            _lora_conv(d, f"up_blocks.{i}.upsamplers.0.conv", c, c, 3, lora)
    _norm(d, "conv_norm_out", UNET_CH[0])
    _conv(d, "conv_out", 4, UNET_CH[0], 3)
    # vae_reduction (decoder_unet.py:14-29): added after LoRA injection, so plain names
    _conv(d, "vae_reduction.short_cut", 4, in_ch, 1)
    _norm(d, "vae_reduction.blocks.0", in_ch)
    _conv(d, "vae_reduction.blocks.2", in_ch, in_ch, 3)
    _norm(d, "vae_reduction.blocks.3", in_ch)
    _conv(d, "vae_reduction.blocks.5", 4, in_ch, 3)
    return d


# (out channels, input channels of the three resnets = current + popped skip)
UNET_UP_IN = (
    (1280, (2560, 2560, 2560)),
    (1280, (2560, 2560, 1920)),
    (640, (1920, 1280, 960)),
    (320, (960, 640, 640)),
)


# ----------------------------------------------------------------------------- SD-2.1 VAE decoder
VAE_UP = ((512, 512), (512, 512), (512, 256), (256, 128))  # (in, out) per up block, 3 resnets each


def vae_resnet(d, p, cin, cout):
    _norm(d, f"{p}.norm1", cin)
    _conv(d, f"{p}.conv1", cout, cin, 3)
    _norm(d, f"{p}.norm2", cout)
    _conv(d, f"{p}.conv2", cout, cout, 3)
    if cin != cout:
        _conv(d, f"{p}.conv_shortcut", cout, cin, 1)


def vae_decoder_spec():
    d = OrderedDict()
    _conv(d, "post_quant_conv", 4, 4, 1)
    _conv(d, "decoder.conv_in", 512, 4, 3)
    vae_resnet(d, "decoder.mid_block.resnets.0", 512, 512)
    a = "decoder.mid_block.attentions.0"
    _norm(d, f"{a}.group_norm", 512)
    for n in ("to_q", "to_k", "to_v", "to_out.0"):
        _linear(d, f"{a}.{n}", 512, 512)
    vae_resnet(d, "decoder.mid_block.resnets.1", 512, 512)
    for i, (cin, cout) in enumerate(VAE_UP):
        for j in range(3):
            vae_resnet(d, f"decoder.up_blocks.{i}.resnets.{j}", cin if j == 0 else cout, cout)
        if i < 3:
            _conv(d, f"decoder.up_blocks.{i}.upsamplers.0.conv", cout, cout, 3)
    _norm(d, "decoder.conv_norm_out", 128)
    _conv(d, "decoder.conv_out", 3, 128, 3)
    return d


VAE_DOWN = ((128, 128), (128, 256), (256, 512), (512, 512))  # (in, out) per encoder down block, 2 resnets each


def vae_encoder_spec():
    """SD-2.1 `AutoencoderKL` encoder half + `quant_conv` (diffusers 0.32.1 `Encoder`, DownEncoderBlock2D x4,
    UNetMidBlock2D with one single-head attention; reference call: model_sd15_with_codec_stage1.py:191-195).
    Published size: 34 163 592 parameters (+ quant_conv 72)."""
    d = OrderedDict()
    _conv(d, "encoder.conv_in", 128, 3, 3)
    for i, (cin, cout) in enumerate(VAE_DOWN):
        for j in range(2):
            vae_resnet(d, f"encoder.down_blocks.{i}.resnets.{j}", cin if j == 0 else cout, cout)
        if i < 3:
            _conv(d, f"encoder.down_blocks.{i}.downsamplers.0.conv", cout, cout, 3)
    vae_resnet(d, "encoder.mid_block.resnets.0", 512, 512)
    a = "encoder.mid_block.attentions.0"
    _norm(d, f"{a}.group_norm", 512)
    for n in ("to_q", "to_k", "to_v", "to_out.0"):
        _linear(d, f"{a}.{n}", 512, 512)
    vae_resnet(d, "encoder.mid_block.resnets.1", 512, 512)
    _norm(d, "encoder.conv_norm_out", 512)
    _conv(d, "encoder.conv_out", 8, 512, 3)
    _conv(d, "quant_conv", 8, 8, 1)
    return d


# ----------------------------------------------------------------------------- analysis transform `enc`
ENC_UNET_CH = (512, 768, 768)          # unet_ch_config (codec_module.py:51, config_inference.yaml)
ENC_TEMB = 2048                        # 4 * block_out_channels[0]
ENC_HEAD_DIM = 8                       # UNet2DModel default attention_head_dim


def _enc_resnet(d, p, cin, cout):
    """diffusers ResnetBlock2D with time embedding (groups 32, eps 1e-5)."""
    _norm(d, f"{p}.norm1", cin)
    _conv(d, f"{p}.conv1", cout, cin, 3)
    _linear(d, f"{p}.time_emb_proj", cout, ENC_TEMB)
    _norm(d, f"{p}.norm2", cout)
    _conv(d, f"{p}.conv2", cout, cout, 3)
    if cin != cout:
        _conv(d, f"{p}.conv_shortcut", cout, cin, 1)


def _enc_attn(d, p, c):
    """diffusers Attention as used by AttnDownBlock2D / AttnUpBlock2D / UNetMidBlock2D (heads = c / 8)."""
    _norm(d, f"{p}.group_norm", c)
    for n in ("to_q", "to_k", "to_v", "to_out.0"):
        _linear(d, f"{p}.{n}", c, c)


def _enc_own_resnet(d, p, cin, cout):
    """The reference's own ResnetBlock (encoder_unet.py:14-52): bias-free convs, nin_shortcut on the OUTPUT."""
    _norm(d, f"{p}.norm1", cin)
    _conv(d, f"{p}.conv1", cout, cin, 3, bias=False)
    _norm(d, f"{p}.norm2", cout)
    _conv(d, f"{p}.conv2", cout, cout, 3, bias=False)
    if cin != cout:
        _conv(d, f"{p}.nin_shortcut", cout, cout, 1, bias=False)


def enc_unet_up_plan():
    """(block index, resnet index) -> (input channels, output channels) of the up path, following
    diffusers `UNet2DModel.__init__` / `AttnUpBlock2D.__init__` for block_out_channels (512, 768, 768)."""
    rev = list(reversed(ENC_UNET_CH))                     # (768, 768, 512)
    plan, out = {}, rev[0]
    for i in range(3):
        prev, out = out, rev[i]
        cin_blk = rev[min(i + 1, 2)]
        for j in range(3):
            skip = cin_blk if j == 2 else out
            rin = prev if j == 0 else out
            plan[(i, j)] = (rin + skip, out)
    return plan


def codec_encoder_spec(N=128, cond_ch=4):
    """`enc` of IntraNoAR: Encoder (codec_module.py:51-85) = pix_emb, pix_fusion, the patched
    UNet2DModel of encoder_unet.py:137-161 (forward_enc :54-134), trans_coding."""
    d = OrderedDict()
    _conv(d, "enc.pix_emb", 192, 3, 8)
    _conv(d, "enc.pix_fusion", 320, 192 + cond_ch, 1)
    u = "enc.unet"
    c0 = ENC_UNET_CH[0]
    _enc_own_resnet(d, f"{u}.conv_in.0", 320, c0)
    _enc_own_resnet(d, f"{u}.conv_in.1", c0, c0)
    _enc_own_resnet(d, f"{u}.conv_in.2", c0, c0)
    _conv(d, f"{u}.conv_in.3", c0, c0, 3)
    _linear(d, f"{u}.time_embedding.linear_1", ENC_TEMB, c0)
    _linear(d, f"{u}.time_embedding.linear_2", ENC_TEMB, ENC_TEMB)
    cin = c0
    for i, cout in enumerate(ENC_UNET_CH):
        for j in range(2):
            _enc_resnet(d, f"{u}.down_blocks.{i}.resnets.{j}", cin if j == 0 else cout, cout)
            if i < 2:
                _enc_attn(d, f"{u}.down_blocks.{i}.attentions.{j}", cout)
        if i < 2:
            _conv(d, f"{u}.down_blocks.{i}.downsamplers.0.conv", cout, cout, 3)
        cin = cout
    cm = ENC_UNET_CH[-1]
    _enc_resnet(d, f"{u}.mid_block.resnets.0", cm, cm)
    _enc_attn(d, f"{u}.mid_block.attentions.0", cm)
    _enc_resnet(d, f"{u}.mid_block.resnets.1", cm, cm)
    for (i, j), (rin, rout) in enc_unet_up_plan().items():
        _enc_resnet(d, f"{u}.up_blocks.{i}.resnets.{j}", rin, rout)
        if i < 2:
            _enc_attn(d, f"{u}.up_blocks.{i}.attentions.{j}", rout)
        if i < 2 and j == 2:
            _conv(d, f"{u}.up_blocks.{i}.upsamplers.0.conv", rout, rout, 3)
    _norm(d, f"{u}.conv_norm_out", c0)
    _conv(d, f"{u}.conv_out", c0, c0, 3)
    t = "enc.trans_coding"
    vq_res(d, f"{t}.0", c0)
    vq_attn(d, f"{t}.1", c0)
    vq_res(d, f"{t}.2", c0)
    dcb4(d, f"{t}.3", c0, c0)
    dcb4(d, f"{t}.4", c0, N)
    return d


# legacy attention names found in older SD VAE checkpoints (diffusers renames them on load)
VAE_LEGACY_ATTN = {"query": "to_q", "key": "to_k", "value": "to_v", "proj_attn": "to_out.0"}


def param_count(spec):
    n = 0
    for shape in spec.values():
        k = 1
        for s in shape:
            k *= s
        n += k
    return n
