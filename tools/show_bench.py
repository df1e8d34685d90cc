import json, sys
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print("headline", round(d["value"] / 1e6, 1), "M spots/s", round(d["ms_per_step"], 3), "ms; sketch", d["stage_ms"]["sketch_ms"], "frac", d["roofline"]["frac"])
c = d.get("count_like")
if c:
    print("count-like", round(c["ms_per_step"], 3), "ms; sketch", c["stage_ms"]["sketch_ms"], "sweeps", c["stage_ms"]["sweep_ms"])
for k in ("sparse_csr", "lattice", "y_f64"):
    if k in d:
        print(k, round(d[k]["ms_per_step"], 3), d[k].get("stage_ms", {}).get("sketch_ms"))
