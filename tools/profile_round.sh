#!/bin/bash
# Runs on the GPU box (gpurun): the three rocprofv3 passes behind profiles/rNN_*.  Usage: tools/profile_round.sh r01 [bench.py args]
# (extra arguments select another workload for all three passes, e.g. `r02_config5 --config 5`).
# Counters are collected in their own passes (kernel-trace only alongside), as the pool requires.
tag=${1:-r01}; shift
if [ $# -gt 0 ]; then pmc_args="$*"; else pmc_args="--family all"; fi   # all: the CSR family's kernels too (FETCH / WRITE of the sparse path)
out=/root/repo/gpurun_out/${tag}_prof
rm -rf "$out"; mkdir -p "$out"
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d "$out/trace" -- python3 /root/repo/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-live-pmc --no-host-arrays "$@" > "$out/bench_trace.json" 2> "$out/bench_trace.err"
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d "$out/fetch" -- python3 /root/repo/bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-live-pmc --no-host-arrays $pmc_args > "$out/bench_fetch.json" 2> "$out/bench_fetch.err"
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d "$out/write" -- python3 /root/repo/bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-live-pmc --no-host-arrays $pmc_args > "$out/bench_write.json" 2> "$out/bench_write.err"
# keep what the summaries need: stats + counter csvs (the full kernel traces are large)
find "$out" -name "*kernel_trace.csv" -delete
find "$out" -name "*agent_info.csv" -delete
# counter csvs: the library's kernels only (the torch data generators dispatch thousands of kernels; gpurun copies at most 64 MB back)
for f in $(find "$out" -name "*counter_collection.csv"); do
  head -1 "$f" > "$f.tmp"; grep "fdx::" "$f" >> "$f.tmp"; mv "$f.tmp" "$f"
done
du -sh "$out"; find "$out" -type f | head -20
