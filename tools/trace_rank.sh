#!/bin/bash
# Runs on the GPU box: kernel trace of one rank's whole share of the sharded fit, then the launch-order timeline of the last repetition.
# Usage: tools/trace_rank.sh <tag> [W=8] [rank=3] [n=1000000] [config5=0]
tag=${1:-rank}; W=${2:-8}; r=${3:-3}; n=${4:-1000000}; big=${5:-0}
out=/root/repo/gpurun_out/${tag}
rm -rf "$out"; mkdir -p "$out"
cd /tmp && export TMPDIR=/tmp
FDX_RANK_TRACE=1 rocprofv3 --kernel-trace --output-format csv -d "$out/trace" -- python3 /root/repo/tools/rank_probe.py $W $r $n $big > "$out/probe.log" 2>&1
python3 /root/repo/tools/timeline.py "$out/trace" > "$out/timeline.txt" 2>&1
find "$out" -name "*kernel_trace.csv" -delete
find "$out" -name "*agent_info.csv" -delete
tail -4 "$out/probe.log"
