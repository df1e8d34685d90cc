"""Turn the rocprofv3 output of tools/profile_round.sh (merged back under gpurun_out/<tag>_prof) into the committed
summaries: profiles/<tag>_kernel_stats.{md,csv}, profiles/<tag>_pmc_traffic.md, profiles/<tag>_traffic.json."""
import csv, glob, json, os, sys, shutil
from collections import defaultdict

tag = sys.argv[1] if len(sys.argv) > 1 else "r01"
extra = " ".join(sys.argv[2:])            # bench.py arguments the round was profiled with (tools/profile_round.sh <tag> <args>)
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
src = os.path.join(ROOT, "gpurun_out", f"{tag}_prof")
dst = os.path.join(ROOT, "profiles")


def one(pattern):
    f = glob.glob(os.path.join(src, pattern), recursive=True)
    if not f:
        raise SystemExit(f"missing {pattern} under {src}")
    return max(f, key=os.path.getmtime)          # gpurun merges into gpurun_out/: older runs' files may still be there


# ---- kernel stats
stats = one("trace/**/*kernel_stats.csv")
shutil.copy(stats, os.path.join(dst, f"{tag}_kernel_stats.csv"))
rows = list(csv.DictReader(open(stats)))
rows.sort(key=lambda r: -float(r["TotalDurationNs"]))
bench_line = open(os.path.join(src, "bench_trace.json")).read().strip().split("\n")[-1]
with open(os.path.join(dst, f"{tag}_kernel_stats.md"), "w") as f:
    f.write(f"# {tag} - rocprofv3 kernel stats of the bench command\n\n")
    f.write("Command on the MI355X box: `rocprofv3 --kernel-trace --stats --output-format csv -d ... -- python3 bench.py --steps 3 "
            f"--warmup 1 --no-cpu-baseline{' ' + extra if extra else ''}` (tools/profile_round.sh).\n\n")
    if extra:
        f.write(f"Workload: `bench.py {extra}` (see the bench line below).  torch kernels (at::, Cijk_) are the synthetic data "
                "generators, outside the timed region.\n\n")
    else:
        f.write("Workload: 1M spots x 2000 genes x 30 types, d = 512: gaussian/raw (1 warm-up + 3 timed fits), count-like/log_cpm "
                "(1 + 2 fits, 100 sweeps each) and the CSR family (1M x 20000, 1 + 2 fits) in one process.  torch kernels "
                "(at::, Cijk_) are the synthetic data generators, outside the timed region.\n\n")
    f.write("bench.py line of this profiled run:\n\n```\n" + bench_line + "\n```\n\n")
    f.write("| kernel | calls | total ms | avg us | % |\n|---|---|---|---|---|\n")
    for r in rows[:60]:
        f.write("| `%s` | %s | %.3f | %.1f | %.2f |\n" % (r["Name"][:110], r["Calls"], float(r["TotalDurationNs"]) / 1e6,
                                                          float(r["AverageNs"]) / 1e3, float(r["Percentage"])))


# ---- PMC traffic
def counter(pass_dir, name):
    f = one(f"{pass_dir}/**/*counter_collection.csv")
    acc = defaultdict(list)
    for r in csv.DictReader(open(f)):
        if r["Counter_Name"] == name:
            acc[r["Kernel_Name"]].append(float(r["Counter_Value"]))
    return acc


fetch, write = counter("fetch", "FETCH_SIZE"), counter("write", "WRITE_SIZE")
kern = {}
for k in fetch:
    if not k.startswith(("fdx::", "void fdx::")):
        continue
    fv, wv = fetch[k], write.get(k, [])
    # drop no-op launches (post-convergence sweeps exit at once): keep launches above 10 % of the largest
    big = max(fv) if fv else 0
    keep = [i for i, v in enumerate(fv) if v > 0.1 * big] if big else list(range(len(fv)))
    f_mean = sum(fv[i] for i in keep) / max(len(keep), 1)
    w_keep = [wv[i] for i in keep if i < len(wv)]
    w_mean = sum(w_keep) / max(len(w_keep), 1)
    name = k.replace("void ", "").replace("(anonymous namespace)::", "").split("(")[0]
    kern[name] = {"launches": len(keep), "fetch_kb": f_mean, "write_kb": w_mean,
                  "hbm_bytes_corrected": (2 * f_mean + w_mean) * 1024}
json.dump({"note": "per-launch means over non no-op launches; hbm_bytes_corrected = (2*FETCH_SIZE + WRITE_SIZE)*1024 (gfx950: "
                   "FETCH_SIZE counts half of a wide coalesced read, MI355X_MICROARCH.md HBM section)",
           "kernels": kern}, open(os.path.join(dst, f"{tag}_traffic.json"), "w"), indent=1)
with open(os.path.join(dst, f"{tag}_pmc_traffic.md"), "w") as f:
    f.write(f"# {tag} - HBM traffic per launch from PMC counters (FETCH_SIZE, WRITE_SIZE)\n\n")
    f.write("rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE in two separate passes of `python3 bench.py --steps 1 --warmup 0 "
            f"--no-cpu-baseline {extra if extra else '--family all'}` (tools/profile_round.sh); per-launch means over the real (non no-op) launches; "
            "hbm_bytes = (2*FETCH_SIZE + WRITE_SIZE)*1024: on gfx950 FETCH_SIZE reports half the bytes of a wide coalesced read "
            "(MI355X_MICROARCH.md, HBM section).\n\n")
    f.write("| kernel | launches | 2 x FETCH_SIZE (MB) | WRITE_SIZE (MB) | HBM bytes per launch (MB) |\n|---|---|---|---|---|\n")
    for name, v in sorted(kern.items(), key=lambda kv: -kv[1]["hbm_bytes_corrected"] * kv[1]["launches"]):
        f.write("| `%s` | %d | %.1f | %.1f | %.1f |\n" % (name[:100], v["launches"], 2 * v["fetch_kb"] * 1024 / 1e6,
                                                         v["write_kb"] * 1024 / 1e6, v["hbm_bytes_corrected"] / 1e6))
print("wrote", [p for p in os.listdir(dst) if p.startswith(tag)])
