"""Experiment harness: time the BCD sweep on the 1M x 30 count-like problem (tiled vs global-gather sweep)."""
import os, sys, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from flashdeconv_amd import FlashDeconv
n = int(os.environ.get("PROBE_N", 1_000_000)); K = int(os.environ.get("PROBE_K", 30))
dev = torch.device("cuda", 0)
Y, X, coords = bench.gen_counts(torch, n, 2000, K, dev, 0)
ref = None
for mode in ("tiled", "global"):
    if mode == "global": os.environ["FDX_NO_TILED"] = "1"
    else: os.environ.pop("FDX_NO_TILED", None)
    m = FlashDeconv(sketch_dim=512, preprocess="log_cpm", n_hvg=2000, max_iter=20, tol=1e-30)
    m.fit(Y, X, coords, output="torch"); m.fit(Y, X, coords, output="torch")
    t = m.timings_
    same = None if ref is None else bool(torch.equal(ref, m.beta_))
    ref = m.beta_.clone()
    print(json.dumps({"mode": mode, "halo_max": None, "sweep_us": round(t["sweep_ms"] / 20 * 1e3, 1), "bit_identical_to_prev": same,
                      **{k: round(v, 2) for k, v in t.items()}}))
