#!/bin/bash
# Runs on the GPU box: kernel trace of a few gaussian/raw fits at the bench shape, then the launch-order timeline of the last one.
# Usage: tools/trace_one.sh <tag> [family=gaussian] [iters=100]
tag=${1:-trace}; fam=${2:-gaussian}; iters=${3:-100}
out=/root/repo/gpurun_out/${tag}
rm -rf "$out"; mkdir -p "$out"
cd /tmp && export TMPDIR=/tmp
PROBE_FAMILY=$fam PROBE_ITERS=$iters PROBE_REPS=4 rocprofv3 --kernel-trace --output-format csv -d "$out/trace" -- python3 /root/repo/tools/env_probe.py "base:" > "$out/probe.log" 2>&1
python3 /root/repo/tools/timeline.py "$out/trace" > "$out/timeline.txt" 2>&1
find "$out" -name "*kernel_trace.csv" -delete
find "$out" -name "*agent_info.csv" -delete
cat "$out/probe.log" | tail -2
