#!/bin/bash
# round 5, VERDICT item 5: does the window kernel's input re-fetch on multi-tile layers cost time?  A/B of tile orders
# (channel tile slow inside groups of G patches) against the production order, same process, interleaved; then FETCH_SIZE
# of 512 -> 512 at 192x192 for the production order and G = 64.
out=${1:-gpurun_out/r5_nslow}; mkdir -p "$out"; root=$(pwd); export TMPDIR=/tmp
for g in 16 64 256; do
  timeout 300 python3 tools/ab_libs.py tools/exp/libonedc_hip_nslow$g.so > "$out/ab_nslow$g.txt" 2>&1 < /dev/null
done
for v in base nslow64; do
  for c in FETCH_SIZE WRITE_SIZE; do
    if [ $v = base ]; then unset ONEDC_HIP_F16_SO; else export ONEDC_HIP_F16_SO="$root/tools/exp/libonedc_hip_$v.so"; fi
    ( cd /tmp && timeout 300 rocprofv3 --kernel-trace --pmc $c -d "$root/$out/pmc_${v}_$c" -o pmc --output-format csv -- python3 "$root/tools/pmc_conv.py" 16 192 192 512 512 3 0 0 3 0 > /dev/null 2> "$root/$out/pmc_${v}_$c.err" < /dev/null )
    f=$(find "$out/pmc_${v}_$c" -name "*counter_collection.csv" < /dev/null | head -1)
    [ -n "$f" ] && python3 - "$f" $c $v >> "$out/pmc.txt" < /dev/null <<'PY'
import csv, sys
n = v = 0
for r in csv.DictReader(open(sys.argv[1])):
    if r["Counter_Name"] == sys.argv[2] and "conv_win4_kernel" in r["Kernel_Name"]:
        n += 1; v += float(r["Counter_Value"])
print(sys.argv[3], sys.argv[2], n, "launches", round(v / max(n, 1) * 1024 / 1e9, 4), "GB per launch (raw counter x 1024; FETCH_SIZE is doubled in the HBM figure)")
PY
    rm -rf "$out/pmc_${v}_$c"
  done
done
unset ONEDC_HIP_F16_SO
cat "$out"/ab_nslow*.txt "$out/pmc.txt"
