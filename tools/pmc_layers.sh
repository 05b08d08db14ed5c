#!/bin/bash
# HBM traffic of the window kernel per layer shape, against the layer's algorithmic bytes:  tools/pmc_layers.sh OUTDIR
# One rocprofv3 --pmc pass per counter and shape (FETCH_SIZE and WRITE_SIZE separately, kernel trace only), three launches of
# tools/pmc_conv.py each; hbm_bytes = (2 * FETCH_SIZE + WRITE_SIZE) * 1024 as in tools/pmc_traffic.py.  Plain 3x3 launches, no
# GroupNorm on load: algorithmic bytes = input + output (+ residual) + weights.  The 1x1 shapes run on the 4-wave GEMM kernels.
out=${1:-gpurun_out/pmc_layers}
mkdir -p "$out"
root=$(pwd)
export TMPDIR=/tmp
# B H W Cin Cout k res
shapes=("16 768 768 128 128 3 0" "16 768 768 256 128 3 0" "16 384 384 256 256 3 0" "16 384 384 512 256 3 0" "16 192 192 512 512 3 0" "16 96 96 512 512 3 0" "16 96 96 320 320 3 0"
        "16 96 96 384 320 1 1" "16 96 96 1280 320 1 1" "16 96 96 320 320 1 1" "16 96 96 320 1024 1 0" "16 48 48 640 2048 1 0" "16 48 48 2560 640 1 1" "16 24 24 1280 1280 1 1")
: > "$out/pmc_by_layer.txt"
for s in "${shapes[@]}"; do
  set -- $s
  tag="${2}x${3}_${4}_${5}_k${6}"
  for c in FETCH_SIZE WRITE_SIZE; do
    ( cd /tmp && timeout 300 rocprofv3 --kernel-trace --pmc $c -d "$root/$out/$tag/$c" -o pmc --output-format csv -- python3 "$root/tools/pmc_conv.py" $1 $2 $3 $4 $5 $6 0 0 3 $7 > /dev/null 2> "$root/$out/$tag.$c.err" < /dev/null )
  done
  f=$(find "$out/$tag/FETCH_SIZE" -name "*counter_collection.csv" < /dev/null | head -1)
  w=$(find "$out/$tag/WRITE_SIZE" -name "*counter_collection.csv" < /dev/null | head -1)
  if [ -n "$f" ] && [ -n "$w" ]; then
    python3 - "$f" "$w" $1 $2 $3 $4 $5 $6 $7 >> "$out/pmc_by_layer.txt" < /dev/null <<'EOF'
import csv, sys
f, w = sys.argv[1:3]
B, H, W, Cin, Cout, k, res = map(int, sys.argv[3:10])
def fold(path, counter):
    n, v = 0, 0.0
    for r in csv.DictReader(open(path)):
        if r["Counter_Name"] == counter and ("conv_win4_kernel" in r["Kernel_Name"] or "conv_gemm" in r["Kernel_Name"] or "conv_lin_kernel" in r["Kernel_Name"]):
            n += 1
            v += float(r["Counter_Value"])
    return n, v
nf, fv = fold(f, "FETCH_SIZE")
nw, wv = fold(w, "WRITE_SIZE")
hbm = (2 * fv / max(nf, 1) + wv / max(nw, 1)) * 1024
algo = B * H * W * (Cin + Cout * (1 + res)) * 2 + k * k * Cin * Cout * 2
tiles = (Cout + 127) // 128
print(f"{k}x{k} {Cin:4d} -> {Cout:4d}{' + res' if res else '      '} at {H}x{W}, B = {B}: {nf} launches  fetched {2 * fv / max(nf, 1) * 1024 / 1e9:6.3f} GB  written {wv / max(nw, 1) * 1024 / 1e9:6.3f} GB  "
      f"algorithmic {algo / 1e9:6.3f} GB (input {B * H * W * Cin * 2 / 1e9:.3f})  ratio {hbm / algo:5.2f}   fetched / input {2 * fv / max(nf, 1) * 1024 / (B * H * W * Cin * 2):5.2f}   ({tiles} channel tile{'s' if tiles > 1 else ''})")
EOF
  fi
  rm -rf "$out/$tag"
done
cat "$out/pmc_by_layer.txt"
