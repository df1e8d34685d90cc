#!/bin/bash
# GPU box: per-kernel times of the CSR family (rocprofv3 --kernel-trace --stats), summary to gpurun_out/$1/
tag=${1:-sparse_stats}
out=/root/repo/gpurun_out/$tag
mkdir -p "$out"
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d "$out/trace" -- python3 /root/repo/tools/step_walls.py sparse 6 > "$out/walls.txt" 2> "$out/err.txt"
f=$(find "$out/trace" -name "*kernel_stats.csv" | head -1)
head -1 "$f" > "$out/kernel_stats.csv"; grep "fdx::" "$f" | head -40 >> "$out/kernel_stats.csv"
rm -rf "$out/trace"
cut -c1-200 "$out/kernel_stats.csv" | head -25
