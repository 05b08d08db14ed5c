#!/usr/bin/env python3
"""Fold two rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE; counter_collection.csv each) into HBM bytes per launch
and kernel:  pmc_traffic.py FETCH.csv WRITE.csv OUT.json "note"
hbm_bytes = (2 * FETCH_SIZE + WRITE_SIZE) * 1024 (KiB counters; FETCH_SIZE doubled on gfx950 for wide coalesced /
LDS-DMA reads, as MI355X_MICROARCH.md's HBM section prescribes)."""
import csv
import json
import re
import sys


def short(name):
    # conv_gemm_kernel<Tile<TM, TN, ..>, F32, SPLIT>: 16-bit / exact f32 / split-half operands are three different kernels
    m = re.search(r"(conv_gemm_kernel)<.*Tile<(\d+), (\d+).*?>, (true|false)(?:, (true|false))?>", name)
    if m:
        return (f"conv_gemm_kernel<{m.group(2)}x{m.group(3)}>"
                + (" f32s" if m.group(5) == "true" else " f32" if m.group(4) == "true" else ""))
    m = re.search(r"(conv_lin_kernel)<.*Tile<(\d+), (\d+).*?>, (\d)>", name)
    if m:
        return f"conv_lin_kernel<{m.group(2)}x{m.group(3)}, {m.group(4)} stages>"
    # conv_win4_kernel<MODE, ST, X2>: MODE 2 = the upsample phases, their own row (4/9 of the taps, x4 outputs); MODE 0 / 1 with
    # or without a second source are the family bench.py's roofline object is about
    m = re.search(r"conv_win4_kernel<(\d), \d(?:, (?:true|false)){0,2}>", name)      # <MODE, ST[, X2[, BF]]>
    if m:
        return "conv_win4_kernel" + ("<ups>" if m.group(1) == "2" else "")
    m = re.search(r"(conv_win_kernel)<(true|false)(?:, (?:true|false))?>", name)
    if m:
        return "conv_win_kernel" + ("<up>" if m.group(2) == "true" else "")
    m = re.search(r"::(\w+?)(_kernel)?[<(]", name)
    return (m.group(1) + (m.group(2) or "")) if m else name[:60]


def fold(path, counter):
    out = {}
    for r in csv.DictReader(open(path)):
        if r["Counter_Name"] != counter:
            continue
        d = out.setdefault(short(r["Kernel_Name"]), [0, 0.0])
        d[0] += 1
        d[1] += float(r["Counter_Value"])
    return out


def main():
    fetch, write = fold(sys.argv[1], "FETCH_SIZE"), fold(sys.argv[2], "WRITE_SIZE")
    res = {"_note": sys.argv[4] if len(sys.argv) > 4 else ""}
    for k in sorted(fetch, key=lambda k: -(2 * fetch[k][1] + write.get(k, [0, 0.0])[1])):
        n, f = fetch[k]
        w = write.get(k, [n, 0.0])[1]
        if (2 * f + w) * 1024 / n < 32e6:
            continue
        res[k] = {"launches": n, "fetch_size_kib": round(f), "write_size_kib": round(w),
                  "hbm_bytes_per_launch": round((2 * f + w) * 1024 / n)}
    json.dump(res, open(sys.argv[3], "w"), indent=1)
    for k, v in res.items():
        if k != "_note":
            print(f"{k:40s} {v['launches']:5d} launches  {v['hbm_bytes_per_launch'] / 1e6:9.1f} MB/launch")


if __name__ == "__main__":
    main()
