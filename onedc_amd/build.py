"""In-tree builds of the native pieces (no JIT cache: the .so files travel with the repo).

    build_coder()  g++     onedc_amd/csrc/coder.cpp          -> csrc/libonedc_coder.so
    build_hip()    hipcc   onedc_amd/csrc/*.hip (gfx950)     -> csrc/libonedc_hip.so
    build_oracle() make -C oracle  (the checkers: C restatement + reference build)
"""
import glob
import os
import shutil
import subprocess
import sys

from . import _native

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HIPCC = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"


def _stale(target, sources):
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.getmtime(s) > t for s in sources)


def _run(cmd):
    print("+", " ".join(cmd), flush=True)
    subprocess.run(cmd, check=True)


def build_coder(force=False):
    src = [os.path.join(_native.CSRC, "coder.cpp"), os.path.join(ROOT, "include", "onedc_coder.h")]
    if force or _stale(_native.CODER_SO, src):
        _run(["g++", "-O3", "-std=c++17", "-shared", "-fPIC", "-pthread", "-Wall",
              "-o", _native.CODER_SO, src[0]])
    return _native.CODER_SO


# attention.hip: finite inputs by construction; without this flag every fmaxf on an MFMA result gets a
# canonicalising v_max_f32 in front (100+ VALU instructions per key tile in a VALU-bound kernel)
# MFMA kernels are built without SLP vectorisation: the vectoriser packs epilogue / normalise-on-load / softmax arithmetic
# into v_pk_fma_f32 / v_pk_mul_f32 / v_pk_add_f32, which cost more issue time than the scalar pairs next to the MFMAs of
# the other wave on the SIMD (MI355X_MICROARCH.md, "packed f32 VALU").  Same results (IEEE fma either way).  Interleaved A/B
# on one box: window kernel with GroupNorm on load 128->128 at 768x768 3.47 -> 3.38 ms, 256->128 5.80 -> 5.57; GEGLU
# 320->2560 422 -> 410 us; d = 80 attention 0.951 -> 0.874 ms; everything else within noise.
_NO_SLP = ("-fno-slp-vectorize",)
FILE_FLAGS = {"attention.hip": ("-fno-honor-nans",) + _NO_SLP, "gemm_conv_win.hip": _NO_SLP, "gemm_conv_win4.hip": _NO_SLP, "gemm_conv.hip": _NO_SLP,
              "gemm_conv256.hip": _NO_SLP, "gemm_conv_ups.hip": _NO_SLP}


def hip_sources():
    return sorted(glob.glob(os.path.join(_native.CSRC, "*.hip")))


def build_hip(force=False, extra_flags=()):
    """Both builds of the kernel sources: bf16 activations (libonedc_hip.so) and IEEE-half activations
    (libonedc_hip_f16.so, -DONEDC_ACT_F16; csrc/common.h)."""
    srcs = hip_sources()
    # (this file is a dependency too: the per-file compiler flags live here)
    headers = glob.glob(os.path.join(_native.CSRC, "*.h")) + [os.path.join(ROOT, "include", "onedc_hip.h"), os.path.abspath(__file__)]
    if not srcs:
        raise RuntimeError("no HIP sources found")
    variants = ((_native.HIP_SO, ".o", ()), (_native.HIP_F16_SO, ".f16.o", ("-DONEDC_ACT_F16",)))
    jobs, links = [], []
    for so, suffix, defs in variants:
        if not (force or _stale(so, srcs + headers)):
            continue
        objs = []
        for s in srcs:
            o = s[:-4] + suffix
            if force or _stale(o, [s] + headers):
                jobs.append([HIPCC, "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-c", "-Wno-unused-result",
                             *FILE_FLAGS.get(os.path.basename(s), ()), *defs, *extra_flags, "-o", o, s])
            objs.append(o)
        links.append([HIPCC, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", so, *objs])
    if jobs:              # one hipcc per file, a few at a time (each is single-threaded and needs ~1-2 GB)
        from concurrent.futures import ThreadPoolExecutor
        with ThreadPoolExecutor(max_workers=min(len(jobs), max(1, min(6, (os.cpu_count() or 2) - 1)))) as pool:
            list(pool.map(_run, jobs))
    for cmd in links:
        _run(cmd)
    return _native.HIP_SO


def build_oracle():
    _run(["make", "-C", os.path.join(ROOT, "oracle"), "all"])


def build_all(force=False):
    build_coder(force)
    build_hip(force)
    build_oracle()


if __name__ == "__main__":
    build_all(force="--force" in sys.argv)
