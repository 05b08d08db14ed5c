#!/bin/bash
out=gpurun_out/r2seq; mkdir -p $out; root=$(pwd); export TMPDIR=/tmp
( cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats -d "$root/$out/prof" -o seq --output-format csv -- python3 "$root/bench.py" --sequential --steps 3 --warmup 1 --no-cpu-baseline > "$root/$out/seq.json" 2> "$root/$out/seq.err" < /dev/null )
find "$out/prof" -name "*kernel_stats.csv" -exec cp {} "$out/seq_kernel_stats.csv" \; < /dev/null
rm -rf "$out/prof"
ls $out < /dev/null
