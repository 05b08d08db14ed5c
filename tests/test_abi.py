"""The two C-ABI libraries load on a CPU-only box and export every symbol their headers declare
(no compute calls here: those need a GPU and live in the -m gpu tests)."""
import ctypes
import os
import re

import pytest

from onedc_amd import _native, build, kernels

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared(header):
    text = open(os.path.join(ROOT, "include", header)).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(onedc_[a-z0-9_]+)\s*\(", text)))


def test_coder_library_exports_header_symbols():
    build.build_coder()
    lib = ctypes.CDLL(_native.CODER_SO)
    names = declared("onedc_coder.h")
    assert len(names) >= 20
    for n in names:
        assert hasattr(lib, n), f"{n} declared in onedc_coder.h but not exported"
    assert set(names) == set(_native.CODER_SYMBOLS)
    assert _native.coder().onedc_coder_version().startswith(b"onedc-coder")


def test_hip_library_exports_header_symbols():
    """hipcc cross-compiles for gfx950 without a GPU; dlopen works because the HIP runtime
    initialises lazily."""
    build.build_hip()
    names = declared("onedc_hip.h")
    assert len(names) >= 20
    # the same sources are built twice: bf16 activations and IEEE-half activations (csrc/common.h)
    for path, fmt in ((_native.HIP_SO, b"bf16"), (_native.HIP_F16_SO, b"f16")):
        lib = ctypes.CDLL(path)
        for n in names:
            assert hasattr(lib, n), f"{n} declared in onedc_hip.h but not exported by {path}"
        lib.onedc_hip_version.restype = lib.onedc_hip_act_format.restype = ctypes.c_char_p
        assert fmt + b" activations" in lib.onedc_hip_version()
        assert lib.onedc_hip_act_format() == fmt
    assert set(names) == set(kernels.HIP_SYMBOLS)
    assert kernels.lib("bf16") is not kernels.lib("f16")
    # the ABI number of the header, of the binding and of both libraries is one number (a symbol-compatible break -- ADVICE
    # round 4: onedc_image_out's 0.3 -> 0.4 change -- shows up as a refused library, not as a stream pointer read from a flag slot)
    header = open(os.path.join(ROOT, "include", "onedc_hip.h")).read()
    abi = int(re.search(r"#define\s+ONEDC_HIP_ABI_VERSION\s+(\d+)", header).group(1))
    assert abi == kernels.ABI_VERSION == kernels.lib("bf16").onedc_hip_abi_version() == kernels.lib("f16").onedc_hip_abi_version()


def test_library_of_another_abi_is_refused(monkeypatch):
    build.build_hip()
    monkeypatch.setattr(kernels, "_libs", {})
    monkeypatch.setattr(kernels, "ABI_VERSION", kernels.ABI_VERSION + 1)
    with pytest.raises(_native.NativeLibraryError, match="ABI"):
        kernels.lib("f16")


def test_library_of_the_other_format_is_refused(monkeypatch):
    """b"f16" is a substring of b"bf16": the loader compares the exact token the library reports, so a bf16 build handed in
    as the f16 library (ONEDC_HIP_F16_SO pointing at an experiment build without -DONEDC_ACT_F16) cannot be used."""
    with pytest.raises(_native.NativeLibraryError):
        kernels.check_library_format("f16", b"bf16", b"onedc-hip 0.3 (gfx950, bf16 activations)")
    with pytest.raises(_native.NativeLibraryError):
        kernels.check_library_format("bf16", b"f16")
    kernels.check_library_format("f16", b"f16")
    build.build_hip()
    monkeypatch.setattr(kernels, "_libs", {})
    monkeypatch.setattr(_native, "_hip_f16", None)
    monkeypatch.setattr(_native, "HIP_F16_SO", _native.HIP_SO)            # the bf16 build under the f16 name
    with pytest.raises(_native.NativeLibraryError, match="built for 'bf16'"):
        kernels.lib("f16")
    monkeypatch.setattr(_native, "_hip_f16", None)


def test_libraries_carry_the_id_of_the_sources_beside_them(monkeypatch):
    """build() stamps every library with a digest of its sources, headers and per-file flags; the loaders compare it with the
    tree they sit in, so a stale .so (they travel untracked) is an error instead of a wrong answer -- and the build log says
    for each library whether it was compiled now or found current."""
    build.build_coder()
    build.build_hip()
    assert build._built_id(_native.CODER_SO) == build.coder_source_id() == _native.coder().onedc_coder_build_id().decode()
    for path, fmt in ((_native.HIP_SO, "bf16"), (_native.HIP_F16_SO, "f16")):
        assert build._built_id(path) == build.hip_source_id() == kernels.lib(fmt).onedc_hip_build_id().decode()
    # experiment flags (ablation builds compute wrong results by design) are part of the identity: such a build never
    # passes for the plain one, and the next plain build finds a wrong id and recompiles every object
    assert build.hip_source_id(("-DEXP_SILU_FREE",)) != build.hip_source_id()
    # a tree whose sources moved on: the loader refuses the library
    monkeypatch.setattr(kernels, "_libs", {})
    monkeypatch.setattr(_native, "_hip_f16", None)
    monkeypatch.setattr(build, "hip_source_id", lambda: "0123456789abcdef")
    with pytest.raises(_native.NativeLibraryError, match="built from other sources"):
        kernels.lib("f16")
    monkeypatch.setenv("ONEDC_ALLOW_STALE_LIB", "1")                   # experiment builds opt out
    monkeypatch.setattr(_native, "_hip_f16", None)
    assert kernels.lib("f16") is not None
    monkeypatch.delenv("ONEDC_ALLOW_STALE_LIB")
    monkeypatch.setattr(_native, "_coder", None)
    monkeypatch.setattr(build, "coder_source_id", lambda: "fedcba9876543210")
    with pytest.raises(_native.NativeLibraryError, match="built from other sources"):
        _native.coder()
    monkeypatch.setattr(_native, "_coder", None)
    monkeypatch.setattr(_native, "_hip_f16", None)


def test_missing_library_fails_loudly(monkeypatch):
    monkeypatch.setattr(_native, "_hip", None)
    monkeypatch.setattr(_native, "HIP_SO", "/nonexistent/libonedc_hip.so")
    with pytest.raises(_native.NativeLibraryError):
        _native.hip()
    monkeypatch.setattr(_native, "_hip_f16", None)
    monkeypatch.setattr(_native, "HIP_F16_SO", "/nonexistent/libonedc_hip_f16.so")
    with pytest.raises(_native.NativeLibraryError):
        _native.hip("f16")
    monkeypatch.setattr(kernels, "_libs", {})
    with pytest.raises(_native.NativeLibraryError):       # the default format's library is the one the product asks for
        kernels.lib()
    with pytest.raises(ValueError):
        _native.hip("fp8")


def test_activation_format_context_is_thread_local():
    """`kernels.act_format` switches the 16-bit format for the calling thread only (the pipeline's worker threads set
    their own), and `a16()` follows it."""
    import threading
    import torch
    seen = {}
    with kernels.act_format("bf16"):
        assert kernels.act() == "bf16" and kernels.a16() == torch.bfloat16

        def other():
            seen["act"] = kernels.act()
            kernels.set_act("f16")
            seen["after"] = kernels.a16()
        t = threading.Thread(target=other)
        t.start()
        t.join()
        assert kernels.act() == "bf16"
        with kernels.act_format("f16"):
            assert kernels.a16() == torch.float16
        assert kernels.act() == "bf16"
    assert seen["act"] == kernels.DEFAULT_ACT and seen["after"] == torch.float16
    assert kernels.act() == kernels.DEFAULT_ACT
    with pytest.raises(ValueError):
        kernels.set_act("fp8")


def test_only_tests_smoke_and_the_bench_baseline_touch_the_oracle():
    """oracle/ is test infrastructure: besides tests/, only __graft_entry__.smoke() and bench.py's cpu_baseline functions may
    import it -- no module of the package, no script under tools/."""
    pat = re.compile(r"^\s*(from|import)\s+oracle\b", flags=re.M)
    for sub in ("onedc_amd", "tools"):
        for dirpath, _, files in os.walk(os.path.join(ROOT, sub)):
            for f in files:
                if f.endswith((".py", ".sh")):
                    assert not pat.search(open(os.path.join(dirpath, f)).read()), f"{sub}/{f} imports the oracle"
    bench = open(os.path.join(ROOT, "bench.py")).read()
    for m in pat.finditer(bench):                       # every import sits inside a cpu_baseline* / *_oracle_* function
        head = bench[:m.start()]
        fn = re.findall(r"^def (\w+)\(", head, flags=re.M)[-1]
        assert fn.startswith(("cpu_baseline", "_time_oracle", "_oracle_weights", "host_rans_timing")), fn
    entry = open(os.path.join(ROOT, "__graft_entry__.py")).read()
    for m in pat.finditer(entry):
        assert re.findall(r"^def (\w+)\(", entry[:m.start()], flags=re.M)[-1] == "smoke"


def test_product_never_imports_the_oracle():
    bad = []
    for dirpath, _, files in os.walk(os.path.join(ROOT, "onedc_amd")):
        for f in files:
            if f.endswith(".py"):
                src = open(os.path.join(dirpath, f)).read()
                if re.search(r"^\s*(from|import)\s+oracle\b", src, flags=re.M):
                    bad.append(f)
    assert not bad, f"product modules import the oracle: {bad}"
