#!/bin/bash
# Measurement batch of a round, on the GPU box:  tools/profile_round.sh OUTDIR
# bench lines of every mode, rocprofv3 kernel stats of the default bench command (without the extras: the clock probe's 1.5 s of
# back-to-back launches of the dominant kernel would sit in the same row of the statistics as the pipelined launches), PMC traffic passes (FETCH_SIZE and
# WRITE_SIZE separately, kernel trace only -- never with other trace domains).  Every step has its own timeout and
# no step reads stdin.
out=${1:-gpurun_out/round}
mkdir -p "$out"
root=$(pwd)
export TMPDIR=/tmp
run() { name=$1; shift; timeout 1000 "$@" > "$out/$name.json" 2> "$out/$name.err" < /dev/null; echo "$name rc=$?"; }
run bench_line python3 bench.py
run bench_line_bf16 env ONEDC_ACT=bf16 python3 bench.py --no-cpu-baseline --no-side-lines
run bench_line_sequential python3 bench.py --sequential --no-cpu-baseline --no-side-lines
run bench_line_zonly python3 bench.py --mode z_only --no-cpu-baseline
run bench_line_encdec python3 bench.py --mode encode_decode
( cd /tmp && timeout 900 rocprofv3 --kernel-trace --stats -d "$root/$out/prof" -o r6 --output-format csv -- python3 "$root/bench.py" --no-cpu-baseline --no-extras > "$root/$out/bench_line_under_profiler.json" 2> "$root/$out/prof.err" < /dev/null; echo "prof rc=$?" )
for c in FETCH_SIZE WRITE_SIZE; do
  ( cd /tmp && timeout 900 rocprofv3 --kernel-trace --pmc $c -d "$root/$out/pmc_$c" -o pmc --output-format csv -- python3 "$root/bench.py" --steps 1 --warmup 0 --no-cpu-baseline --no-extras --sequential > /dev/null 2> "$root/$out/pmc_$c.err" < /dev/null; echo "pmc $c rc=$?" )
done
find "$out" -name "*kernel_stats.csv" -exec cp {} "$out/kernel_stats.csv" \; < /dev/null
# the same decode with the GPU to itself: kernel time per batch without the pipeline's sharing (tools/seqprof.sh)
( cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats -d "$root/$out/seqprof" -o seq --output-format csv -- python3 "$root/bench.py" --sequential --steps 3 --warmup 1 --no-cpu-baseline --no-extras > "$root/$out/bench_line_seqprof.json" 2> "$root/$out/seqprof.err" < /dev/null; echo "seqprof rc=$?" )
find "$out/seqprof" -name "*kernel_stats.csv" -exec cp {} "$out/sequential_kernel_stats.csv" \; < /dev/null
rm -rf "$out/seqprof"
[ -x tools/micro/mfma_rate ] && timeout 120 tools/micro/mfma_rate > "$out/mfma_rate.txt" 2>&1 < /dev/null
[ -x tools/micro/mfma_power ] && timeout 120 tools/micro/mfma_power > "$out/mfma_power.txt" 2>&1 < /dev/null
( cd /tmp && timeout 900 rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE -d "$root/$out/pmc_mfma" -o pmc --output-format csv -- python3 "$root/bench.py" --steps 1 --warmup 0 --no-cpu-baseline --no-extras --sequential > /dev/null 2> "$root/$out/pmc_mfma.err" < /dev/null; echo "pmc mfma rc=$?" )
m=$(find "$out/pmc_mfma" -name "*counter_collection.csv" < /dev/null | head -1)
if [ -n "$m" ]; then ( cd tools && python3 pmc_mfma.py "$root/$m" "$root/$out/pmc_mfma.json" "rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE (one pass, tools/profile_round.sh) over python3 bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-extras --sequential; util = busy cycles of all SIMDs / (GRBM_GUI_ACTIVE / 8 XCDs x 1024 SIMDs). Folded by tools/pmc_mfma.py on the box." < /dev/null ); fi
rm -rf "$out/pmc_mfma"
f=$(find "$out/pmc_FETCH_SIZE" -name "*counter_collection.csv" < /dev/null | head -1)
w=$(find "$out/pmc_WRITE_SIZE" -name "*counter_collection.csv" < /dev/null | head -1)
if [ -n "$f" ] && [ -n "$w" ]; then python3 tools/pmc_traffic.py "$f" "$w" "$out/pmc_traffic.json" "rocprofv3 --kernel-trace --pmc FETCH_SIZE and --pmc WRITE_SIZE (two separate passes, tools/profile_round.sh) over python3 bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-extras --sequential (stream setup + one decode of 16 768x768 pictures), f16 activations. Counter unit KiB. hbm_bytes = (2*FETCH_SIZE + WRITE_SIZE)*1024: FETCH_SIZE doubled as MI355X_MICROARCH.md (HBM section) prescribes for wide coalesced / LDS-DMA reads on gfx950; WRITE_SIZE exact. Per launch = sum over the run / launches in the run; conv_win4_kernel = MODE 0 and 1 with or without a second source (the launches bench.py's roofline object is about), conv_win4_kernel<ups> = MODE 2. Folded by tools/pmc_traffic.py on the box." < /dev/null; fi
rm -rf "$out/prof" "$out/pmc_FETCH_SIZE" "$out/pmc_WRITE_SIZE"
ls -la "$out" < /dev/null
# round 4: per-layer tables, the A/B of the round's kernel forms, the tile sweep of the UNet's 1x1 shapes
timeout 400 python3 tools/unet_layers.py > "$out/unet_layers.txt" 2>&1 < /dev/null
timeout 400 python3 tools/vae_layers.py > "$out/vae_layers.txt" 2>&1 < /dev/null
timeout 300 python3 tools/bench_1x1.py 136 > "$out/bench_1x1.txt" 2>&1 < /dev/null
timeout 200 python3 tools/stage_times.py > "$out/stage_times.txt" 2>&1 < /dev/null
timeout 200 python3 tools/bench_gn.py > "$out/bench_gn.txt" 2>&1 < /dev/null
timeout 200 python3 tools/bench_attn.py > "$out/bench_attn.txt" 2>&1 < /dev/null
ls -la "$out" < /dev/null
# round 6: the fused cross-attention block against the four launches, the window kernel's instruction mix against LDS reads per MFMA
timeout 200 python3 tools/bench_xattn.py > "$out/bench_xattn.txt" 2>&1 < /dev/null
[ -x tools/micro/mfma_lds_mix ] && timeout 200 tools/micro/mfma_lds_mix > "$out/power_mix.txt" 2>&1 < /dev/null
ls -la "$out" < /dev/null
# round 6: the bench line once more, now that this batch's own PMC records exist (roofline.traffic / mfma_util_pmc cite them: traffic_from)
if [ -s "$out/pmc_traffic.json" ] && [ -s "$out/pmc_mfma.json" ]; then
  tag=$(python3 -c "import bench; print(bench.PROFILE_ROUNDS[0])")
  cp "$out/pmc_traffic.json" "profiles/${tag}_pmc_traffic.json"; cp "$out/pmc_mfma.json" "profiles/${tag}_pmc_mfma.json"
  cp "$out/bench_line.json" "$out/bench_line_first.json"
  run bench_line python3 bench.py
fi
ls -la "$out" < /dev/null
