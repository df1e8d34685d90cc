#!/bin/bash
# Runs on the GPU box: kernel trace of one rank's share through the real driver (tools/driver_probe.py), timeline of the last repetition.
# Usage: tools/trace_driver.sh <tag> [W=8] [rank=3] [n=1000000]
tag=${1:-drv}; W=${2:-8}; r=${3:-3}; n=${4:-1000000}
out=/root/repo/gpurun_out/${tag}
rm -rf "$out"; mkdir -p "$out"
cd /tmp && export TMPDIR=/tmp
FDX_TRACE_DRIVER=1 rocprofv3 --kernel-trace --output-format csv -d "$out/trace" -- python3 /root/repo/tools/driver_probe.py $W $r $n > "$out/probe.log" 2>&1
python3 /root/repo/tools/timeline.py "$out/trace" > "$out/timeline.txt" 2>&1
find "$out" -name "*kernel_trace.csv" -delete
find "$out" -name "*agent_info.csv" -delete
tail -3 "$out/probe.log"
