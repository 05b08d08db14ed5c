"""Fit the summation structure of torch-CPU's f32 convolutions on THIS machine (no GPU needed) and write it as a writer's
profile for `IntraNoAR(prior_arith="torch_cpu", ref_profile=ref_arith.load_profile(path))`.

Why: a stream written by the reference on a CPU decodes on the GPU only in the writer's arithmetic.  Shapes
that go through oneDNN calibrate against any host's torch, but torch's native path (aten _slow_conv2d_forward -> the BLAS sgemm:
every 1x1 convolution of a one-thread writer, small inputs of any writer) sums in the order the BLAS picks for the CPU it runs
on.  Run this where the streams are WRITTEN, with the writer's thread count, and ship the JSON to the decoder:

    python tools/cpu_chain_probe.py --size 768x768 --threads 1 --out writer_profile.json

It is also how the tables of DESIGN.md section 5 / csrc/ref_arith.hip's header were found: candidate structures are evaluated with
real fused multiply-adds (a few lines of C compiled with `gcc -mfma -ffp-contract=off` into a temporary directory) and compared
with `F.conv2d` bit for bit.  The families are exactly the ones the GPU kernels implement:
    1x1   input channels in blocks of R; block sums added in order; the bias either starts the first chain (oneDNN jit_1x1) or
          is added to the first block's sum (native)
    3x3   "direct": bias + per 16 input channels one chain over (ky, kx, channel) (oneDNN jit avx512), or one chain from the
          bias over (8-channel block, ky, kx, channel) (oneDNN jit avx2), or
          "im2col": k = (channel, ky, kx), K in T parts, blocks inside a part, parts [0, s) and [s, T) summed in order, then added
"""
import argparse
import ctypes
import json
import math
import os
import subprocess
import sys
import tempfile

import numpy as np
import torch
import torch.nn.functional as F

C_SRC = r"""
#include <stdint.h>
// X [P][K], Wm [OC][K] -> out [P][OC][parts]: per part the sum, in order, of chains from 0 over blocks of blk
// (first_from_bias: the very first chain starts from the bias instead)
void gemm_parts(const float *X, const float *Wm, const float *bias, float *out, int P, int K, int OC, const int *bounds,
                int parts, int blk, int first_from_bias) {
    for (int p = 0; p < P; ++p)
        for (int oc = 0; oc < OC; ++oc)
            for (int pi = 0; pi < parts; ++pi) {
                float tot = 0.f;
                for (int k0 = bounds[pi]; k0 < bounds[pi + 1]; k0 += blk) {
                    float acc = (first_from_bias && pi == 0 && k0 == bounds[0]) ? bias[oc] : 0.f;
                    int k1 = k0 + blk < bounds[pi + 1] ? k0 + blk : bounds[pi + 1];
                    for (int k = k0; k < k1; ++k) acc = __builtin_fmaf(X[(int64_t)p * K + k], Wm[(int64_t)oc * K + k], acc);
                    tot = k0 == bounds[pi] ? acc : tot + acc;
                }
                out[((int64_t)p * OC + oc) * parts + pi] = tot;
            }
}
// depthwise 3x3, pad 1: one chain from the bias over the taps, row-major, padding taps skipped.  x [H][W][C], w [C][9]
void dw3x3(const float *x, const float *w, const float *bias, float *y, int H, int W, int C) {
    for (int yy = 0; yy < H; ++yy)
        for (int xx = 0; xx < W; ++xx)
            for (int c = 0; c < C; ++c) {
                float acc = bias[c];
                for (int ky = 0; ky < 3; ++ky)
                    for (int kx = 0; kx < 3; ++kx) {
                        int iy = yy + ky - 1, ix = xx + kx - 1;
                        if (iy < 0 || iy >= H || ix < 0 || ix >= W) continue;
                        acc = __builtin_fmaf(x[(iy * W + ix) * C + c], w[c * 9 + ky * 3 + kx], acc);
                    }
                y[(yy * W + xx) * C + c] = acc;
            }
}
"""


def build_helper(tmp):
    src = os.path.join(tmp, "chain.c")
    so = os.path.join(tmp, "chain.so")
    open(src, "w").write(C_SRC)
    subprocess.check_call(["gcc", "-O2", "-mfma", "-ffp-contract=off", "-shared", "-fPIC", "-o", so, src])
    return ctypes.CDLL(so)


def parts_of(lib, X, Wm, bn, bounds, blk, first_from_bias):
    P, K = X.shape
    OC = Wm.shape[0]
    T = len(bounds) - 1
    out = np.empty((P, OC, T), np.float32)
    bd = np.asarray(bounds, np.int32)
    ptr = lambda a: ctypes.c_void_p(a.ctypes.data)
    lib.gemm_parts(ptr(X), ptr(Wm), ptr(bn), ptr(out), P, K, OC, ptr(bd), T, blk, int(first_from_bias))
    return out


def f32(a):
    return a.astype(np.float32)


def fit(lib, C, OC, ks, H, W):
    """-> 1x1: [block, bias_first]; 3x3: ["direct", icb, one_chain] or ["im2col", parts, block, split]; None when nothing fits."""
    g = torch.Generator().manual_seed(C * 131 + OC * 7 + H)
    x = torch.randn(1, C, H, W, generator=g)
    w = torch.randn(OC, C, ks, ks, generator=g) / math.sqrt(C * ks * ks)
    b = torch.randn(OC, generator=g) * 0.05
    with torch.no_grad():
        ref = F.conv2d(x, w, b, padding=ks // 2)[0].permute(1, 2, 0).reshape(H * W, OC).contiguous().numpy()
        X = F.unfold(x, ks, padding=ks // 2)[0].t() if ks == 3 else x[0].permute(1, 2, 0).reshape(H * W, C)
    X = np.ascontiguousarray(X.contiguous().numpy())
    Wm = np.ascontiguousarray(w.reshape(OC, -1).numpy())
    bn = np.ascontiguousarray(b.numpy())
    K = X.shape[1]
    bb = np.broadcast_to(bn, ref.shape).astype(np.float32)
    if ks == 1:
        for bias_first in (1, 0):
            for blk in sorted(set([K] + list(range(16, K, 16))), reverse=True):
                po = parts_of(lib, X, Wm, bn, [0, K], blk, bias_first)[..., 0]
                # !bias_first: the bias joins the FIRST block's sum; with one part that sum is not separable here, so rebuild it
                if bias_first:
                    y = po
                else:
                    nb = -(-K // blk)
                    blocks = parts_of(lib, X, Wm, bn, [min(i * blk, K) for i in range(nb + 1)], blk, 0)
                    y = f32(blocks[..., 0] + bb)
                    for i in range(1, nb):
                        y = f32(y + blocks[..., i])
                if np.array_equal(y, ref):
                    return [blk, bias_first]
        return None
    if C % 16 == 0:                      # oneDNN direct: bias + per-16-channel chains over (ky, kx, channel in block)
        Xo = np.ascontiguousarray(X.reshape(-1, C // 16, 16, 9).transpose(0, 1, 3, 2).reshape(-1, K))
        Wo = np.ascontiguousarray(Wm.reshape(OC, C // 16, 16, 9).transpose(0, 1, 3, 2).reshape(OC, K))
        po = parts_of(lib, Xo, Wo, bn, [i * 144 for i in range(C // 16 + 1)], 144, 0)
        y = bb.copy()
        for i in range(C // 16):
            y = f32(y + po[..., i])
        if np.array_equal(y, ref):
            return ["direct", 16, 0]
    if C % 8 == 0:                       # oneDNN direct, avx2: ONE chain from the bias over (8-channel block, ky, kx, channel)
        Xo = np.ascontiguousarray(X.reshape(-1, C // 8, 8, 9).transpose(0, 1, 3, 2).reshape(-1, K))
        Wo = np.ascontiguousarray(Wm.reshape(OC, C // 8, 8, 9).transpose(0, 1, 3, 2).reshape(OC, K))
        if np.array_equal(parts_of(lib, Xo, Wo, bn, [0, K], K, 1)[..., 0], ref):
            return ["direct", 8, 1]
    for T in (1, 2, 3, 4, 6, 8):
        if K % T:
            continue
        per = K // T
        for blk in sorted({per, 384, 256, 192, 128, 96, 64} & set(range(1, per + 1)), reverse=True):
            nb = -(-per // blk)
            bounds = sorted({min(pi * per + i * blk, (pi + 1) * per) for pi in range(T) for i in range(nb + 1)})
            blocks = parts_of(lib, X, Wm, bn, bounds, blk, 0)           # every block on its own
            parts = []
            for pi in range(T):
                idx = [i for i in range(len(bounds) - 1) if pi * per <= bounds[i] < (pi + 1) * per]
                tot = f32(blocks[..., idx[0]] + bb) if pi == 0 else blocks[..., idx[0]]
                for i in idx[1:]:
                    tot = f32(tot + blocks[..., i])
                parts.append(tot)
            for split in range(T, 0, -1):
                left = parts[0]
                for pt in parts[1:split]:
                    left = f32(left + pt)
                y = left
                if split < T:
                    right = parts[split]
                    for pt in parts[split + 1:]:
                        right = f32(right + pt)
                    y = f32(left + right)
                if np.array_equal(y, ref):
                    return ["im2col", T, blk, split]
    return None


def fit_dw(lib, C, H, W):
    """Depthwise 3x3 (always oneDNN's jit_dw): is it the bias-first chain over the taps that csrc/elementwise.hip evaluates?"""
    g = torch.Generator().manual_seed(C * 131 + H)
    x = torch.randn(1, C, H, W, generator=g)
    w = torch.randn(C, 1, 3, 3, generator=g) / 3
    b = torch.randn(C, generator=g) * 0.05
    with torch.no_grad():
        ref = F.conv2d(x, w, b, padding=1, groups=C)[0].permute(1, 2, 0).contiguous().numpy()
    xn = np.ascontiguousarray(x[0].permute(1, 2, 0).numpy())
    wn = np.ascontiguousarray(w.reshape(C, 9).numpy())
    bn = np.ascontiguousarray(b.numpy())
    y = np.empty((H, W, C), np.float32)
    ptr = lambda a: ctypes.c_void_p(a.ctypes.data)
    lib.dw3x3(ptr(xn), ptr(wn), ptr(bn), ptr(y), H, W, C)
    return ["taps_from_bias"] if np.array_equal(y, ref) else None


def index_thresholds(scale_min=0.11, scale_max=64.0, levels=256):
    """thresholds[i] = the smallest f32 sigma whose table index under the reference's build_indexes (entropy_models.py:355-362) is
    >= i + 1, with THIS machine's torch.log (bisection over bit patterns; whole vectors only, as in onedc_amd/ref_arith.py)."""
    log_min = math.log(scale_min)
    log_step = (math.log(scale_max) - log_min) / (levels - 1)

    def index(s):
        n = s.numel()
        s = torch.cat([s, s.new_ones((-n) % 64)])
        s = torch.maximum(s, torch.zeros_like(s) + 1e-5)
        return ((torch.log(s) - log_min) / log_step).clamp_(0, levels - 1).int()[:n]

    want = torch.arange(1, levels, dtype=torch.int32)
    lo = torch.full((levels - 1,), int(np.float32(1e-6).view(np.int32)), dtype=torch.int32)
    hi = torch.full((levels - 1,), int(np.float32(128.0).view(np.int32)), dtype=torch.int32)
    with torch.no_grad():
        while int((hi - lo).max()) > 1:
            mid = lo + (hi - lo) // 2
            ok = index(mid.view(torch.float32)) >= want
            hi = torch.where(ok, mid, hi)
            lo = torch.where(ok, lo, mid)
    return [int(v) for v in hi]


def prior_shapes(h, w):
    """Every convolution shape (C, OC, k, H, W) of the prior networks (codec_module.py:145-166, 205-209; compression_model.py:
    369-407) for a z map of h x w -- dense convolutions only (the depthwise ones always run in oneDNN's jit_dw)."""
    out = [(7, 128, 1, h, w)]
    for lv in range(3):
        H, W = h << lv, w << lv
        out += [(128, 128, 1, H, W), (128, 512, 1, H, W), (256, 128, 1, H, W)]           # DCB4(128): conv1/conv2, ffn in, ffn out
        if lv < 2:
            out += [(128, 128, 3, 2 * H, 2 * W)]                                          # RBU: subpel/upsample are (128, 512) above
    H, W = 4 * h, 4 * w
    out += [(128, 256, 1, H, W), (256, 256, 1, H, W), (256, 1024, 1, H, W), (512, 256, 1, H, W)]
    return out


def main(argv=None):
    ap = argparse.ArgumentParser(description=__doc__, formatter_class=argparse.RawDescriptionHelpFormatter)
    ap.add_argument("--size", action="append", help="picture size HxW (repeatable; default 768x768)")
    ap.add_argument("--threads", type=int, default=torch.get_num_threads(), help="the writer's torch.set_num_threads")
    ap.add_argument("--out", default=None, help="write the profile as JSON (default: print only)")
    a = ap.parse_args(argv)
    torch.set_num_threads(a.threads)
    cpu = next((l.split(":", 1)[1].strip() for l in open("/proc/cpuinfo") if l.startswith("model name")), "?")
    prof = {"host": cpu, "torch": torch.__version__, "threads": a.threads, "onednn_isa_cap": os.environ.get("DNNL_MAX_CPU_ISA", ""),
            "1x1": {}, "3x3": {}, "dw3x3": {}}
    with tempfile.TemporaryDirectory() as tmp:
        lib = build_helper(tmp)
        for size in a.size or ["768x768"]:
            H, W = (int(v) for v in size.lower().split("x"))
            h, w = -(-H // 64), -(-W // 64)
            for C, OC, ks, hh, ww in prior_shapes(h, w):
                key = f"{C},{OC},{hh},{ww}"
                tab = prof["1x1" if ks == 1 else "3x3"]
                if key in tab:
                    continue
                tab[key] = fit(lib, C, OC, ks, hh, ww)
                print(f"[probe] {a.threads} thread(s) {ks}x{ks} {C:4d} -> {OC:4d} on {hh:3d}x{ww:<3d}: {tab[key]}", flush=True)
            for lv, Cd in ((0, 128), (1, 128), (2, 128), (2, 256)):      # the depthwise 3x3 of the DCB4 blocks at each level
                key = f"{Cd},{h << lv},{w << lv}"
                if key not in prof["dw3x3"]:
                    prof["dw3x3"][key] = fit_dw(lib, Cd, h << lv, w << lv)
                    print(f"[probe] {a.threads} thread(s) depthwise 3x3 {Cd:4d} on {h << lv:3d}x{w << lv:<3d}: {prof['dw3x3'][key]}", flush=True)
    prof["index_thresholds_bits"] = index_thresholds()       # the writer's logarithm, as 255 f32 bit patterns
    try:                                                     # the Gaussian CDF tables this machine's torch builds (Normal.cdf in fp32)
        import hashlib
        sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
        from onedc_amd.entropy.entropy_models import GaussianEncoder
        cdf = GaussianEncoder().build_tables()[0]
        prof["gaussian_tables_sha16"] = hashlib.sha256(np.asarray(cdf).astype(np.int32).tobytes()).hexdigest()[:16]
        print(f"[probe] Gaussian CDF tables: sha256[:16] {prof['gaussian_tables_sha16']}")
    except Exception as e:                                   # the coder library is not built here: the decoder cannot check
        print(f"[probe] Gaussian CDF tables not digested ({type(e).__name__}: {e})")
    if a.out:
        json.dump(prof, open(a.out, "w"), indent=1)
        print(f"[probe] profile of {cpu} at {a.threads} thread(s) -> {a.out}")
    return prof


if __name__ == "__main__":
    main()
