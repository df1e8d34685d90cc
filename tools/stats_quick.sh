#!/bin/bash
# Runs on the GPU box: rocprofv3 kernel stats of one bench family, top kernels printed.  Usage: tools/stats_quick.sh <tag> <bench args...>
tag=$1; shift
out=/root/repo/gpurun_out/${tag}
rm -rf "$out"; mkdir -p "$out"
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d "$out/trace" -- python3 /root/repo/bench.py --no-cpu-baseline "$@" > "$out/bench.json" 2> "$out/bench.err"
f=$(find "$out" -name "*kernel_stats.csv" | head -1)
python3 - "$f" <<'PY' > "$out/top.txt"
import csv,sys
rows=list(csv.DictReader(open(sys.argv[1])))
rows=[r for r in rows if "fdx::" in r["Name"] or "rocprim" in r["Name"]]
rows.sort(key=lambda r:-float(r["TotalDurationNs"]))
for r in rows[:28]:
    print(f'{r["Name"][:90]:90s} calls {r["Calls"]:>5s} avg_us {float(r["AverageNs"])/1e3:9.1f} total_ms {float(r["TotalDurationNs"])/1e6:8.2f}')
PY
find "$out" -name "*kernel_trace.csv" -delete; find "$out" -name "*agent_info.csv" -delete
cat "$out/top.txt"
