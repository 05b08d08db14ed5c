"""Host-side weight preparation: checkpoint tensors -> the layouts the HIP kernels read.

* conv / linear weights -> bf16 [rows padded to 128][KH*KW*Cin], k-contiguous with k = (ky, kx, cin)
* GLU layers (ConvFFN3, GEGLU): rows interleaved in blocks of 16 (value block, gate block)
* sub-pixel convs: rows ordered (dy*2+dx)*C' + c so the pixel-shuffle store is channel-contiguous
* attention projections: heads zero-padded from d to DQ / DV (40->64/48, 80->128/96 as two half-heads, 160->160)
* peft LoRA: W += (alpha/r) * B @ A   (reference decoder_unet.py:360-367; config_inference.yaml:19-22)
"""
import torch

ROW_PAD = 128


def fold_lora(sd, rank=64, alpha=8.0):
    out = {}
    scale = alpha / rank
    for k, v in sd.items():
        if ".lora_A." in k or ".lora_B." in k:
            continue
        if ".base_layer." in k:
            name, leaf = k.split(".base_layer.")
            if leaf == "weight":
                A = sd[f"{name}.lora_A.default.weight"].double()
                B = sd[f"{name}.lora_B.default.weight"].double()
                delta = (B.reshape(B.shape[0], -1) @ A.reshape(A.shape[0], -1)).reshape(v.shape)
                v = (v.double() + scale * delta).float()
            out[f"{name}.{leaf}"] = v
        else:
            out[k] = v
    return out


def _pad_rows(m):
    rows = m.shape[0]
    pad = (-rows) % ROW_PAD
    if pad:
        m = torch.cat([m, m.new_zeros(pad, m.shape[1])], dim=0)
    return m


def pack_matrix(w2d, device, row_perm=None, dtype=None):
    """[rows][K] f32 -> padded 16-bit matrix in the current activation format (or f32 for the exact-f32 prior path) on
    `device`."""
    if dtype is None:
        from . import kernels
        dtype = kernels.a16()
    if row_perm is not None:
        w2d = w2d[row_perm]
    return _pad_rows(w2d.contiguous()).to(dtype).to(device).contiguous()


SPLIT16_SCALE = 2048.0          # 2^11: the low half of a split operand is stored times this (csrc/gemm_conv.hip, SPLIT)


def split16_slot_order():
    """Channel held by slot s (0..31) of a 32-channel K-step in the split form: the lane group fq = s // 8 of the matrix
    instruction reads the f32 chunks fq and 4 + fq of the activation row, i.e. channels 4 fq .. 4 fq + 3 and 16 + 4 fq .."""
    return torch.tensor([(4 * (s // 8) + s % 8) if s % 8 < 4 else (16 + 4 * (s // 8) + s % 8 - 4) for s in range(32)])


def pack_matrix_split16(w2d, device, row_perm=None):
    """[rows][K] f32 (K % 32 == 0) -> the split form of onedc_conv_params.in_f32 == 2: every weight as a pair of IEEE halves
    w = h + l * 2^-11 (h = half(w), l = half((w - h) * 2^11): 22 significant bits), per row and 32-channel block 32 halves h
    followed by 32 halves l, channels in the kernel's slot order -- the same 128 bytes per block an f32 row has.
    Returns a float16 tensor [rows padded to 128][2 K]."""
    if row_perm is not None:
        w2d = w2d[row_perm]
    w = _pad_rows(w2d.contiguous().float())
    rows, K = w.shape
    assert K % 32 == 0, K
    blocks = w.view(rows, K // 32, 32)[:, :, split16_slot_order()]
    h = blocks.to(torch.float16)
    l = ((blocks - h.float()) * SPLIT16_SCALE).to(torch.float16)
    return torch.cat([h, l], dim=2).reshape(rows, 2 * K).to(device).contiguous()


def conv_matrix(w):
    """(Cout, Cin, KH, KW) or (Cout, Cin) -> [Cout][KH*KW*Cin] with k = (ky, kx, cin)."""
    if w.dim() == 2:
        return w
    return w.permute(0, 2, 3, 1).reshape(w.shape[0], -1)


def upsample_phase_matrix(w):
    """(C, Cin, 3, 3) weights of a 3x3 conv that follows a nearest x2 upsample -> [4 C][4 Cin] matrix of the four
    2x2 phase convolutions on the low-resolution input (csrc/gemm_conv_ups.hip): row (py*2+px)*C + c, column
    (ty*2+tx)*Cin + cin, W'[py,px][ty,tx] = sum of W[ky,kx] over ky in S(py,ty), kx in S(px,tx) with
    S(0,0)={0}, S(0,1)={1,2}, S(1,0)={0,1}, S(1,1)={2}.  Sums in f32."""
    sets = {(0, 0): (0,), (0, 1): (1, 2), (1, 0): (0, 1), (1, 1): (2,)}
    w = w.float()
    C, Cin = w.shape[0], w.shape[1]
    out = w.new_zeros(4, C, 4, Cin)
    for py in range(2):
        for px in range(2):
            for ty in range(2):
                for tx in range(2):
                    acc = w.new_zeros(C, Cin)
                    for ky in sets[(py, ty)]:
                        for kx in sets[(px, tx)]:
                            acc += w[:, :, ky, kx]
                    out[py * 2 + px, :, ty * 2 + tx, :] = acc
    return out.reshape(4 * C, 4 * Cin)


def mxfp8_roundtrip(m, block=32):
    """[rows][K] f32 -> the same matrix with every block of 32 consecutive k quantised to OCP e4m3 under one shared
    power-of-two scale 2^(floor(log2 amax) - 8) (OCP MX v1.0) and dequantised again: the weight operand of an MXFP8 layer
    (csrc/elementwise.hip: mxfp8_roundtrip_kernel does the same to the activations)."""
    rows, Kk = m.shape
    assert Kk % block == 0
    b = m.float().reshape(rows, Kk // block, block)
    amax = b.abs().amax(-1, keepdim=True)
    e = torch.floor(torch.log2(amax.clamp_min(2.0 ** -126))) - 8
    scale = torch.where(amax > 0, torch.exp2(e), torch.ones_like(amax))
    q = (b / scale).clamp(-448.0, 448.0).to(torch.float8_e4m3fn).float()
    return (q * scale).reshape(rows, Kk)


def glu_perm(n_half):
    """Row order for GLU epilogues: [v0..v15, g0..g15, v16..v31, g16..g31, ...]."""
    assert n_half % 16 == 0
    idx = []
    for t in range(n_half // 16):
        idx += list(range(16 * t, 16 * t + 16)) + list(range(n_half + 16 * t, n_half + 16 * t + 16))
    return torch.tensor(idx, dtype=torch.long)


def shuffle_perm(c_out):
    """Row order for a conv followed by PixelShuffle(2): new row s*C' + c <- old row c*4 + s."""
    assert c_out % 4 == 0
    cq = c_out // 4
    return torch.tensor([c * 4 + s for s in range(4) for c in range(cq)], dtype=torch.long)


def head_positions(d, d_pad):
    """Where the d real dims of a head sit inside its d_pad padded dims.  d = 80 is laid out as two half-heads of 40
    (flash_attn80_kernel: [0..39 | pad | 40..79 | pad], halves of d_pad / 2); every other head dim is contiguous."""
    if d == 80:
        half = d_pad // 2
        return torch.cat([torch.arange(40), half + torch.arange(40)])
    return torch.arange(d)


def pad_heads_rows(w, heads, d, d_pad):
    """(heads*d, K) -> (heads*d_pad, K): the rows of each head at head_positions(d, d_pad), zero rows elsewhere."""
    K = w.shape[1]
    out = w.new_zeros(heads, d_pad, K)
    out[:, head_positions(d, d_pad).to(w.device)] = w.view(heads, d, K)
    return out.reshape(heads * d_pad, K)


def pad_heads_cols(w, heads, d, d_pad):
    """(N, heads*d) -> (N, heads*d_pad): the columns of each head at head_positions(d, d_pad), zero columns elsewhere."""
    N = w.shape[0]
    out = w.new_zeros(N, heads, d_pad)
    out[:, :, head_positions(d, d_pad).to(w.device)] = w.view(N, heads, d)
    return out.reshape(N, heads * d_pad)


def f32(t, device, perm=None):
    if t is None:
        return None
    if perm is not None:
        t = t[perm]
    return t.float().contiguous().to(device)


HEAD_PAD = {40: (64, 48), 80: (128, 96), 160: (160, 160), 8: (32, 16)}   # head dim -> (DQ, DV)
