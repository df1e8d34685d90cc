"""Summarise the counter passes of tools/pmc_round.sh (merged back under gpurun_out/<tag>_pmc) into
profiles/<tag>_pmc_counters.md: per kernel, per-launch averages of the SQ counters and the derived busy fractions."""
import csv, glob, os, sys
from collections import defaultdict

tag = sys.argv[1] if len(sys.argv) > 1 else "r02"
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
src = os.path.join(ROOT, "gpurun_out", f"{tag}_pmc")
vals = defaultdict(lambda: defaultdict(list))       # kernel -> counter -> per-launch values
dur = defaultdict(list)
for f in glob.glob(os.path.join(src, "pass*/**/*counter_collection.csv"), recursive=True):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"]
        if "fdx::" not in k:
            continue
        vals[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
for f in glob.glob(os.path.join(src, "pass1/**/*kernel_trace.csv"), recursive=True):
    for r in csv.DictReader(open(f)):
        if "fdx::" in r["Kernel_Name"]:
            dur[r["Kernel_Name"]].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)


def mean_big(xs):                                    # drop no-op launches (post-convergence sweeps)
    if not xs:
        return 0.0
    big = max(xs)
    keep = [x for x in xs if x > 0.1 * big] if big > 0 else xs
    return sum(keep) / len(keep)


want = [k for k in vals if any(s in k for s in ("tile_sketch", "sketch_contract", "xyt_split", "bcd_sweep_tiled", "sketch_rows_scatter", "sketch_csr", "csr_moments", "knn_kernel", "lev_eigen", "merge_rows", "normalize_export"))]
want.sort(key=lambda k: -mean_big(dur.get(k, [0])) * len(dur.get(k, [])))
out = os.path.join(ROOT, "profiles", f"{tag}_pmc_counters.md")
with open(out, "w") as f:
    f.write(f"# {tag} - SQ counters of the hot kernels (rocprofv3 --pmc, tools/pmc_round.sh)\n\n")
    f.write("Per-launch averages over the launches that did work.  SQ_* cycle counters are in quad-cycles summed over all waves "
            "(SQ_WAVE_CYCLES, SQ_WAIT_*, SQ_ACTIVE_INST_*) or over SIMDs/CUs (SQ_BUSY_*); fractions below are ratios of "
            "counters from the same pass, so the units cancel.\n\n")
    for k in want:
        v = {c: mean_big(x) for c, x in vals[k].items()}
        f.write(f"## `{k[:140]}`\n\n")
        if dur.get(k):
            f.write(f"launches {len(dur[k])}, average duration (pass 1, counters on) {mean_big(dur[k]):.1f} us\n\n")
        f.write("| counter | per launch |\n|---|---|\n")
        for c in sorted(v):
            f.write(f"| {c} | {v[c]:.4g} |\n")
        wc = v.get("SQ_WAVE_CYCLES", 0.0)
        f.write("\n| derived | value |\n|---|---|\n")
        if wc:
            for c, label in (("SQ_WAIT_ANY", "waves parked (s_waitcnt / barrier)"), ("SQ_WAIT_INST_ANY", "waves stalled at issue"),
                             ("SQ_ACTIVE_INST_ANY", "waves issuing"), ("SQ_WAIT_INST_LDS", "issue stalls on the LDS queue"),
                             ("SQ_ACTIVE_INST_LDS", "LDS instructions issuing"), ("SQ_ACTIVE_INST_VALU", "VALU instructions issuing")):
                if c in v:
                    f.write(f"| {label}: {c} / SQ_WAVE_CYCLES | {v[c] / wc:.3f} |\n")
        if v.get("SQ_LDS_IDX_ACTIVE"):
            f.write(f"| LDS bank-conflict share: SQ_LDS_BANK_CONFLICT / SQ_LDS_IDX_ACTIVE | {v.get('SQ_LDS_BANK_CONFLICT', 0) / v['SQ_LDS_IDX_ACTIVE']:.3f} |\n")
        if v.get("SQ_BUSY_CU_CYCLES"):
            b = v["SQ_BUSY_CU_CYCLES"]
            if "SQ_VALU_MFMA_BUSY_CYCLES" in v:
                f.write(f"| MFMA busy: SQ_VALU_MFMA_BUSY_CYCLES / (4 SIMDs x SQ_BUSY_CU_CYCLES) | {v['SQ_VALU_MFMA_BUSY_CYCLES'] / (4 * b):.3f} |\n")
        elif "SQ_VALU_MFMA_BUSY_CYCLES" in v and v.get("SQ_BUSY_CYCLES"):
            pass
        if v.get("SQ_LDS_IDX_ACTIVE") and v.get("GRBM_GUI_ACTIVE"):
            f.write(f"| LDS array busy: SQ_LDS_IDX_ACTIVE / (256 CUs x GRBM_GUI_ACTIVE / 8) | {v['SQ_LDS_IDX_ACTIVE'] / (256 * v['GRBM_GUI_ACTIVE'] / 8):.3f} |\n")
        if v.get("SQ_VALU_MFMA_BUSY_CYCLES") and v.get("GRBM_GUI_ACTIVE"):
            f.write(f"| MFMA pipe busy: SQ_VALU_MFMA_BUSY_CYCLES / (1024 SIMDs x GRBM_GUI_ACTIVE / 8) | {v['SQ_VALU_MFMA_BUSY_CYCLES'] / (1024 * v['GRBM_GUI_ACTIVE'] / 8):.3f} |\n")
        if v.get("TCC_HIT_sum") is not None and (v.get("TCC_HIT_sum", 0) + v.get("TCC_MISS_sum", 0)) > 0:
            f.write(f"| L2 hit rate: TCC_HIT_sum / (TCC_HIT_sum + TCC_MISS_sum) | {v['TCC_HIT_sum'] / (v['TCC_HIT_sum'] + v['TCC_MISS_sum']):.3f} |\n")
        f.write("\n")
print("wrote", out)
