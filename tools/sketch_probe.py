"""Experiment harness: stage timings of the gaussian/raw 1M x 2000 x 30 fit under env-var variants."""
import os, sys, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from flashdeconv_amd import FlashDeconv
n = int(os.environ.get("PROBE_N", 1_000_000)); K = int(os.environ.get("PROBE_K", 30))
dev = torch.device("cuda", 0)
fam = os.environ.get("PROBE_FAMILY", "gaussian")
Y, X, coords = (bench.gen_gaussian if fam == "gaussian" else bench.gen_counts)(torch, n, 2000, K, dev, 0)
variants = [dict(e.split("=") for e in v.split(",") if e) for v in os.environ.get("PROBE_VARIANTS", "").split(";")]
for env in variants:
    for k in ("FDX_FIT_CHUNK", "FDX_SKETCH_NO_REG", "FDX_NO_TILED"): os.environ.pop(k, None)
    os.environ.update(env)
    m = FlashDeconv(sketch_dim=512, preprocess="raw" if fam == "gaussian" else "log_cpm", n_hvg=2000, max_iter=int(os.environ.get("PROBE_ITERS", 100)))
    m.fit(Y, X, coords, output="torch"); m.fit(Y, X, coords, output="torch")
    print(json.dumps({"env": env, **{k: round(v, 2) for k, v in m.timings_.items()}}))
