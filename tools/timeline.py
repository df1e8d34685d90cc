"""Launch-order timeline of the last fit in a rocprofv3 --kernel-trace csv: start offset, duration and gap to the previous
kernel, so that host-side bubbles (read-backs, allocation, Python) show up between the kernels.

usage: python tools/timeline.py <dir with *_kernel_trace.csv> [marker kernel substring = normalize_export]"""
import csv, glob, os, sys

src = sys.argv[1]
marker = sys.argv[2] if len(sys.argv) > 2 else "normalize_export"
f = max(glob.glob(os.path.join(src, "**", "*kernel_trace.csv"), recursive=True), key=os.path.getmtime)
rows = [(int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]) for r in csv.DictReader(open(f))]
rows.sort()
ends = [i for i, r in enumerate(rows) if marker in r[2]]
lo = ends[-2] + 1 if len(ends) > 1 else 0
hi = ends[-1]
t0 = rows[lo][0]
prev_end = t0
print(f"{'start us':>9} {'dur us':>8} {'gap us':>8}  kernel")
for s, e, name in rows[lo:hi + 1]:
    print(f"{(s - t0) / 1e3:9.1f} {(e - s) / 1e3:8.1f} {(s - prev_end) / 1e3:8.1f}  {name[:100]}")
    prev_end = max(prev_end, e)
