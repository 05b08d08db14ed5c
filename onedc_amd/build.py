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


def _digest(paths, extra=""):
    """sha256 over (name, bytes) of the given files + `extra`: the identity of what a library is built from."""
    import hashlib
    h = hashlib.sha256(extra.encode())
    for p in sorted(paths):
        h.update(os.path.basename(p).encode() + b"\0")
        with open(p, "rb") as f:
            h.update(f.read())
    return h.hexdigest()[:16]


def coder_sources():
    return [os.path.join(_native.CSRC, "coder.cpp"), os.path.join(ROOT, "include", "onedc_coder.h")]


def coder_source_id():
    return _digest(coder_sources())


def hip_source_id(extra_flags=()):
    """Identity of the kernel sources, headers, per-file compiler flags and extra (experiment) flags; both activation formats
    are built from them.  The compiler's version is NOT part of it: the id must be reproducible on a box that only loads the
    prebuilt libraries (no hipcc call at import); a compiler change rebuilds through force=True / a clean tree."""
    return _digest(hip_sources() + glob.glob(os.path.join(_native.CSRC, "*.h")) + [os.path.join(ROOT, "include", "onedc_hip.h")],
                   repr(sorted(FILE_FLAGS.items())) + repr(tuple(extra_flags)))


BUILD_ID_MARKER = b"@(#)onedc-build-id:"


def _built_id(so, symbol=None):
    """The source id a built library carries (None: missing library or a build from before the id existed).  Read from the
    file's bytes, not through dlopen: a process that has loaded an older build of the same path would get that one back."""
    if not os.path.exists(so):
        return None
    with open(so, "rb") as f:
        data = f.read()
    i = data.find(BUILD_ID_MARKER)
    if i < 0:
        return None
    j = i + len(BUILD_ID_MARKER)
    return data[j:data.index(b"\0", j)].decode(errors="replace")


def build_coder(force=False):
    src = coder_sources()
    sid = coder_source_id()
    if force or _stale(_native.CODER_SO, src) or _built_id(_native.CODER_SO, "onedc_coder_build_id") != sid:
        _run(["g++", "-O3", "-std=c++17", "-shared", "-fPIC", "-pthread", "-Wall", f'-DONEDC_BUILD_ID="{sid}"',
              "-o", _native.CODER_SO, src[0]])
    print(f"[build] {os.path.relpath(_native.CODER_SO, ROOT)}: source id {_built_id(_native.CODER_SO, 'onedc_coder_build_id')}", flush=True)
    return _native.CODER_SO


# attention.hip: finite inputs by construction; without this flag every fmaxf on an MFMA result gets a
# canonicalising v_max_f32 in front (100+ VALU instructions per key tile in a VALU-bound kernel)
# MFMA kernels are built without SLP vectorisation: the vectoriser packs epilogue / normalise-on-load / softmax arithmetic
# into v_pk_fma_f32 / v_pk_mul_f32 / v_pk_add_f32, which cost more issue time than the scalar pairs next to the MFMAs of
# the other wave on the SIMD (MI355X_MICROARCH.md, "packed f32 VALU").  Same results (IEEE fma either way).  Interleaved A/B
# on one box: window kernel with GroupNorm on load 128->128 at 768x768 3.47 -> 3.38 ms, 256->128 5.80 -> 5.57; GEGLU
# 320->2560 422 -> 410 us; d = 80 attention 0.951 -> 0.874 ms; everything else within noise.
_NO_SLP = ("-fno-slp-vectorize",)
# ref_arith.hip reproduces torch-CPU's rounding sequence: a multiply followed by an add must stay two roundings (the chains
# themselves are explicit fmaf calls)
FILE_FLAGS = {"attention.hip": ("-fno-honor-nans",) + _NO_SLP, "xattn_block.hip": ("-fno-honor-nans",) + _NO_SLP, "gemm_conv_win.hip": _NO_SLP, "gemm_conv_win4.hip": _NO_SLP, "gemm_conv.hip": _NO_SLP,
              "gemm_conv256.hip": _NO_SLP, "gemm_conv_ups.hip": _NO_SLP, "ref_arith.hip": ("-ffp-contract=off",)}


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
    # extra_flags (ablation builds: results wrong by design) are part of the id: such a library is refused by the loader
    # (its id is not the sources' plain id) and the next plain build sees a wrong id and recompiles EVERY object
    sid = hip_source_id(extra_flags)
    jobs, links = [], []
    for so, suffix, defs in variants:
        # a library is current when it is newer than its sources AND carries the id of exactly these sources (mtimes do
        # not survive every way a tree is copied; the id does): onedc_hip_build_id(), checked again when it is loaded
        wrong_id = _built_id(so, "onedc_hip_build_id") != sid
        if not (force or wrong_id or _stale(so, srcs + headers)):
            continue
        objs = []
        for s in srcs:
            o = s[:-4] + suffix
            carries_id = os.path.basename(s) == "elementwise.hip"
            # a wrong id means the objects' provenance is unknown (a copied tree keeps no mtimes, an ablation build leaves
            # mtime-fresh objects behind): all of them are compiled again, not only the one that carries the id
            if force or wrong_id or _stale(o, [s] + headers):
                jobs.append([HIPCC, "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-c", "-Wno-unused-result",
                             *FILE_FLAGS.get(os.path.basename(s), ()), *defs, *extra_flags,
                             *((f'-DONEDC_BUILD_ID="{sid}"',) if carries_id else ()), "-o", o, s])
            objs.append(o)
        links.append([HIPCC, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", so, *objs])
    if jobs:              # one hipcc per file, a few at a time (each is single-threaded and needs ~1-2 GB)
        from concurrent.futures import ThreadPoolExecutor
        with ThreadPoolExecutor(max_workers=min(len(jobs), max(1, min(6, (os.cpu_count() or 2) - 1)))) as pool:
            list(pool.map(_run, jobs))
    for cmd in links:
        _run(cmd)
    for so, _, _ in variants:
        got = _built_id(so, "onedc_hip_build_id")
        print(f"[build] {os.path.relpath(so, ROOT)}: source id {got} ({'compiled now' if any(so in c for c in links) else 'up to date'})",
              flush=True)
        if got != sid:
            raise RuntimeError(f"{so} carries source id {got}, the sources are {sid}")
    return _native.HIP_SO


PROBE_F16_SO = os.path.join(_native.CSRC, "libonedc_probe_f16.so")


def build_probe(force=False):
    """The diagnostic twin of libonedc_hip_f16.so: the same objects with gemm_conv_win4.hip compiled -DEXP_W4_TIMELINE (every
    workgroup of the 4-wave window kernel stamps s_memtime / s_memrealtime around its K loop into the buffer passed as
    `stats`).  bench.py measures the shader clock the chip holds inside the dominant kernel with it (`roofline.clock_ghz`,
    MI355X_MICROARCH.md 'DVFS give-back' item 6: a separate build, no stamp executes in the product library).  Never loaded
    by the product path."""
    src = os.path.join(_native.CSRC, "gemm_conv_win4.hip")
    headers = glob.glob(os.path.join(_native.CSRC, "*.h")) + [os.path.join(ROOT, "include", "onedc_hip.h"), os.path.abspath(__file__)]
    obj = os.path.join(_native.CSRC, "gemm_conv_win4.probe.f16.o")
    others = [s[:-4] + ".f16.o" for s in hip_sources() if os.path.basename(s) != "gemm_conv_win4.hip"]
    if force or _stale(obj, [src] + headers):
        _run([HIPCC, "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-c", "-Wno-unused-result",
              *FILE_FLAGS.get("gemm_conv_win4.hip", ()), "-DONEDC_ACT_F16", "-DEXP_W4_TIMELINE", "-o", obj, src])
    if force or _stale(PROBE_F16_SO, [obj] + others):
        _run([HIPCC, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", PROBE_F16_SO, obj, *others])
    print(f"[build] {os.path.relpath(PROBE_F16_SO, ROOT)}: diagnostic build (clock probe)", flush=True)
    return PROBE_F16_SO


def build_oracle():
    _run(["make", "-C", os.path.join(ROOT, "oracle"), "all"])


def build_all(force=False):
    build_coder(force)
    build_hip(force)
    build_probe(force)
    build_oracle()


if __name__ == "__main__":
    build_all(force="--force" in sys.argv)
