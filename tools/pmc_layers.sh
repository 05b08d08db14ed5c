#!/bin/bash
# HBM traffic of the window kernel per layer shape, against the layer's algorithmic bytes:  tools/pmc_layers.sh OUTDIR
# One rocprofv3 --pmc pass per counter and shape (FETCH_SIZE and WRITE_SIZE separately, kernel trace only), three launches of
# tools/pmc_conv.py each; hbm_bytes = (2 * FETCH_SIZE + WRITE_SIZE) * 1024 as in tools/pmc_traffic.py.  Plain 3x3 launches, no
# residual, no GroupNorm on load: algorithmic bytes = input + output + weights.
out=${1:-gpurun_out/pmc_layers}
mkdir -p "$out"
root=$(pwd)
export TMPDIR=/tmp
# B H W Cin Cout
shapes=("16 768 768 128 128" "16 768 768 256 128" "16 384 384 256 256" "16 384 384 512 256" "16 192 192 512 512" "16 96 96 512 512" "16 96 96 320 320")
: > "$out/pmc_by_layer.txt"
for s in "${shapes[@]}"; do
  set -- $s
  tag="${2}x${3}_${4}_${5}"
  for c in FETCH_SIZE WRITE_SIZE; do
    ( cd /tmp && timeout 300 rocprofv3 --kernel-trace --pmc $c -d "$root/$out/$tag/$c" -o pmc --output-format csv -- python3 "$root/tools/pmc_conv.py" $1 $2 $3 $4 $5 3 0 0 3 > /dev/null 2> "$root/$out/$tag.$c.err" < /dev/null )
  done
  f=$(find "$out/$tag/FETCH_SIZE" -name "*counter_collection.csv" < /dev/null | head -1)
  w=$(find "$out/$tag/WRITE_SIZE" -name "*counter_collection.csv" < /dev/null | head -1)
  if [ -n "$f" ] && [ -n "$w" ]; then
    python3 - "$f" "$w" $1 $2 $3 $4 $5 >> "$out/pmc_by_layer.txt" < /dev/null <<'EOF'
import csv, sys
f, w = sys.argv[1:3]
B, H, W, Cin, Cout = map(int, sys.argv[3:8])
def fold(path, counter):
    n, v = 0, 0.0
    for r in csv.DictReader(open(path)):
        if r["Counter_Name"] == counter and "conv_win4_kernel" in r["Kernel_Name"]:
            n += 1
            v += float(r["Counter_Value"])
    return n, v
nf, fv = fold(f, "FETCH_SIZE")
nw, wv = fold(w, "WRITE_SIZE")
hbm = (2 * fv / max(nf, 1) + wv / max(nw, 1)) * 1024
algo = B * H * W * (Cin + Cout) * 2 + 9 * Cin * Cout * 2
tiles = (Cout + 127) // 128
print(f"3x3 {Cin:4d} -> {Cout:4d} at {H}x{W}, B = {B}: {nf} launches  fetched {2 * fv / max(nf, 1) * 1024 / 1e9:6.3f} GB  written {wv / max(nw, 1) * 1024 / 1e9:6.3f} GB  "
      f"algorithmic {algo / 1e9:6.3f} GB (input {B * H * W * Cin * 2 / 1e9:.3f})  ratio {hbm / algo:5.2f}   fetched / input {2 * fv / max(nf, 1) * 1024 / (B * H * W * Cin * 2):5.2f}   ({tiles} channel tile{'s' if tiles > 1 else ''})")
EOF
  fi
  rm -rf "$out/$tag"
done
cat "$out/pmc_by_layer.txt"
