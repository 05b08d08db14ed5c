#!/usr/bin/env python3
"""Static instruction mix of a kernel's MFMA-carrying basic blocks, from the gfx950 assembly hipcc writes for one source file,
priced with the per-instruction ISSUE costs of /opt/skills/guides/MI355X_MICROARCH.md ("vector-instruction ISSUE cost": an MFMA
holds the SIMD's vector issue for 8 of its 16 cycles, v_exp / v_rcp / v_rsq 8, other VALU 4, v_cvt_pk 4; costs add).  No GPU needed.

    python tools/isa_mix.py onedc_amd/csrc/attention.hip flash_attn64_kernelILb1 [--min-mfma 20] [-D...]

What it is for: saying, for a loop that is neither at the matrix pipe's rate nor at HBM's, how far the instruction stream itself
allows it to go (the issue bound of ONE wave's stream; two waves of a SIMD share that issue port) -- round 6 used it to close
flash_attn64 (profiles/r6_isa_mix.txt)."""
import collections
import os
import re
import subprocess
import sys
import tempfile

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))

ISSUE = (("v_mfma", 8), ("v_exp", 8), ("v_log", 8), ("v_rcp", 8), ("v_rsq", 8), ("v_sqrt", 8), ("v_sin", 8), ("v_cos", 8))
MFMA_CYCLES = {"16x16x32": 16, "32x32x16": 32, "16x16x16": 8, "32x32x8": 16, "4x4x4": 8, "32x32x2": 64, "16x16x4": 32}


def issue_cost(op):
    if op.startswith("v_"):
        for k, c in ISSUE:
            if op.startswith(k):
                return c
        return 4
    if op.startswith(("ds_", "buffer_", "global_", "flat_")):
        return 4
    return 0                                   # scalar stream: its own port


def main():
    args = sys.argv[1:]
    min_mfma = 20
    if "--min-mfma" in args:
        i = args.index("--min-mfma")
        min_mfma = int(args[i + 1])
        del args[i:i + 2]
    src, sym = args[0], args[1]
    extra = args[2:]
    from onedc_amd.build import FILE_FLAGS
    flags = list(FILE_FLAGS.get(os.path.basename(src), ()))
    with tempfile.TemporaryDirectory() as d:
        out = os.path.join(d, "k.s")
        cmd = ["hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-DONEDC_ACT_F16", "-S", "--cuda-device-only",
               "-Wno-unused-result", *flags, *extra, "-o", out, src]
        subprocess.run(cmd, check=True, stderr=subprocess.DEVNULL)
        text = open(out).read().split("\n")
    start = end = None
    for i, l in enumerate(text):
        if start is None and re.match(r"^_Z\w*" + re.escape(sym) + r"\w*:", l):
            start = i
        elif start is not None and l.startswith(".Lfunc_end"):
            end = i
            break
    if start is None:
        raise SystemExit(f"no kernel matching {sym} in {src}")
    print(f"# {text[start].split(':')[0]}")
    print(f"# hipcc {' '.join(cmd[1:-3])}")
    meta = [l.strip() for l in text if sym in l and (".num_vgpr" in l or ".num_agpr" in l)]
    for m in meta[:2]:
        print("#", m.split(".set ")[-1])
    blocks, cur = [], ("entry", [])
    for l in text[start + 1:end]:
        m = re.match(r"^(\.LBB\d+_\d+):", l)
        if m:
            blocks.append(cur)
            cur = (m.group(1), [])
            continue
        t = l.strip()
        if t and not t.startswith((";", ".")):
            cur[1].append(t.split()[0])
    blocks.append(cur)
    for name, ins in blocks:
        c = collections.Counter(ins)
        n_mfma = sum(v for k, v in c.items() if k.startswith("v_mfma"))
        if n_mfma < min_mfma:
            continue
        pipe = 0
        for k, v in c.items():
            if k.startswith("v_mfma"):
                pipe += v * next((cy for shape, cy in MFMA_CYCLES.items() if shape in k), 16)
        issue = sum(issue_cost(k) * v for k, v in c.items())
        groups = collections.Counter()
        for k, v in c.items():
            key = ("MFMA" if k.startswith("v_mfma") else "transcendental" if issue_cost(k) == 8 else
                   "v_cvt_pk" if k.startswith("v_cvt_pk") else "other VALU" if k.startswith("v_") else
                   "LDS" if k.startswith("ds_") else "VMEM" if k.startswith(("buffer_", "global_", "flat_")) else
                   "s_waitcnt / s_nop / s_barrier" if k in ("s_waitcnt", "s_nop", "s_barrier") else "other scalar")
            groups[key] += v
        print(f"\n{name}: {len(ins)} instructions, {n_mfma} MFMAs = {pipe} matrix-pipe cycles of one wave; vector ISSUE cost "
              f"{issue} cycles -> the stream alone allows {min(1.0, pipe / issue):.2f} of the pipe "
              f"({'issue' if issue > pipe else 'pipe'}-bound)")
        print("   " + " · ".join(f"{k} {v}" for k, v in sorted(groups.items(), key=lambda x: -x[1])))
        top = sorted(((k, v) for k, v in c.items() if not k.startswith("s_")), key=lambda x: -x[1])[:12]
        print("   " + ", ".join(f"{k} x{v}" for k, v in top))


if __name__ == "__main__":
    main()
