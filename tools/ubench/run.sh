#!/bin/bash
# Build and run the scan micro-benchmarks on the GPU box; outputs go to gpurun_out/ubench/ (copy what is to be judged into profiles/).
#   gpurun -- bash tools/ubench/run.sh
set -e
cd "$(dirname "$0")"
OUT=../../gpurun_out/ubench
mkdir -p "$OUT"
for t in pair_rate block_scan scan_variants; do
    /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -o $t $t.hip
    ./$t > "$OUT/$t.txt" 2>&1
    tail -n 3 "$OUT/$t.txt"
done
grep '^JSON ' "$OUT/pair_rate.txt" | sed 's/^JSON //' > "$OUT/pair_rate.json"
cat "$OUT/pair_rate.json"
