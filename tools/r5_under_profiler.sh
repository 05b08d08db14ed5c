out=gpurun_out/r5under; mkdir -p $out; root=$(pwd); export TMPDIR=/tmp
( cd /tmp && timeout 900 rocprofv3 --kernel-trace --stats -d "$root/$out/prof" -o r5 --output-format csv -- python3 "$root/bench.py" --no-cpu-baseline --no-extras > "$root/$out/bench_line_under_profiler.json" 2> "$root/$out/prof.err" < /dev/null; echo "prof rc=$?" )
find "$out" -name "*kernel_stats.csv" -exec cp {} "$out/kernel_stats.csv" \; < /dev/null
rm -rf "$out/prof"; ls -la $out
