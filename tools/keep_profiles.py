#!/usr/bin/env python3
"""Copy one measurement batch of tools/profile_round.sh from gpurun_out/ (scratch) into profiles/ (tracked):
    python tools/keep_profiles.py gpurun_out/r4final r4 ["remark for the CSV headers"]
The rocprofv3 kernel-stats CSVs get a comment header that says which command they are from and what the kernel time of one
batch is made of (sums by kernel family, computed here from the CSV itself)."""
import csv
import json
import re
import os
import shutil
import sys

FAMILIES = (
    ("window kernel", lambda s: "conv_win4_kernel" in s),
    ("  of it with conv_shortcut as a second source", lambda s: re.search(r"conv_win4_kernel<\d, \d, true", s) is not None),
    ("  of it multiplying bf16 operands (f16 storage)", lambda s: re.search(r"conv_win4_kernel<\d, \d, (?:true|false), true>", s) is not None),
    ("  of it GroupNorm on load", lambda s: "conv_win4_kernel<1" in s),
    ("  of it upsample phases", lambda s: "conv_win4_kernel<2" in s),
    ("8-wave window kernel", lambda s: "conv_win_kernel" in s),
    ("flash_attn64", lambda s: "flash_attn64" in s),
    ("flash_attn80", lambda s: "flash_attn80" in s),
    ("4-wave 16-bit GEMMs", lambda s: ("conv_gemm_kernel" in s and "false, false>" in s) or "conv_lin_kernel" in s),
    ("  of it the 128x256 ring tile (gated layers)", lambda s: "conv_lin_kernel" in s and "Tile<128, 256" in s),
    ("prior GEMMs (split-half operands)", lambda s: "conv_gemm_kernel" in s and "true, true>" in s),
    ("exact f32 GEMMs", lambda s: "conv_gemm_kernel" in s and "true, false>" in s),
    ("256x256 incl. split-K", lambda s: "conv_gemm256_kernel" in s or "splitk_reduce" in s),
    ("128x256", lambda s: "conv_gemm128x256" in s),
    ("scale/shift pass", lambda s: "scale_shift_act" in s),
    ("layernorm", lambda s: "layernorm" in s),
    ("GroupNorm statistics kernels", lambda s: "::gn_" in s),
    ("conv_norm_out + conv_out", lambda s: "few_out" in s),
)


def family_line(path, decodes):
    rows = list(csv.DictReader(open(path)))

    def ms(pred):
        return sum(float(r["TotalDurationNs"]) for r in rows if pred(r["Name"])) / 1e6 / decodes
    total = ms(lambda s: True)
    parts = "; ".join(f"{name.strip()} {ms(pred):.1f}" for name, pred in FAMILIES)
    return total, parts


def main():
    src, tag = sys.argv[1], sys.argv[2]
    remark = sys.argv[3] if len(sys.argv) > 3 else ""
    dst = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "profiles")
    plain = {"bench_line.json", "bench_line_bf16.json", "bench_line_sequential.json", "bench_line_zonly.json", "bench_line_encdec.json",
             "bench_line_under_profiler.json", "pmc_traffic.json", "pmc_mfma.json", "mfma_rate.txt", "unet_layers.txt",
             "vae_layers.txt", "ab_round4.txt", "bench_1x1.txt", "mfma_power.txt", "stage_times.txt", "bench_gn.txt", "bench_attn.txt",
             "bench_xattn.txt"}
    for name in sorted(plain):
        p = os.path.join(src, name)
        if os.path.exists(p) and os.path.getsize(p):
            shutil.copy(p, os.path.join(dst, f"{tag}_{name}"))
            print("kept", name)
    seq = os.path.join(src, "sequential_kernel_stats.csv")
    if os.path.exists(seq):
        line = json.load(open(os.path.join(src, "bench_line_seqprof.json")))
        # (round 6: the bench times `repeatability.regions` regions of `steps` steps back to back)
        decodes = line["steps"] * line.get("repeatability", {}).get("regions", 1) + line["warmup"]
        total, parts = family_line(seq, decodes)
        head = (f"# rocprofv3 --kernel-trace --stats of python3 bench.py --sequential --steps {line['steps']} --warmup {line['warmup']} "
                f"--no-cpu-baseline --no-extras (tools/profile_round.sh; {line.get('repeatability', {}).get('regions', 1)} timed region(s)):\n# {decodes} decodes of 16 x 768x768 one after the other, nothing "
                f"else on the GPU: {total:.1f} ms of kernel time per batch, by family (ms): {parts}.\n")
        if remark:
            head += f"# {remark}\n"
        with open(os.path.join(dst, f"{tag}_sequential_kernel_stats.csv"), "w") as f:
            f.write(head + open(seq).read())
        print(f"kept sequential_kernel_stats.csv ({total:.1f} ms per batch)")
    pipe = os.path.join(src, "kernel_stats.csv")
    if os.path.exists(pipe):
        line = json.load(open(os.path.join(src, "bench_line_under_profiler.json")))
        head = (f"# rocprofv3 --kernel-trace --stats of python3 bench.py --no-cpu-baseline --no-extras (tools/profile_round.sh): the pipelined default bench "
                f"({line['value']} images/s under the profiler); launches of different streams overlap, so durations here include the time a kernel "
                f"shares the GPU\n")
        if remark:
            head += f"# {remark}\n"
        with open(os.path.join(dst, f"{tag}_bench_default_kernel_stats.csv"), "w") as f:
            f.write(head + open(pipe).read())
        print("kept kernel_stats.csv")


if __name__ == "__main__":
    main()
