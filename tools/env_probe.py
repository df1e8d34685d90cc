#!/usr/bin/env python3
"""Times the stages of one fit under environment-variable variants and reports how far each variant's result is from the
first one's.  Usage:
    PROBE_FAMILY=counts|gaussian PROBE_N=1000000 PROBE_G=2000 PROBE_K=30 PROBE_D=512 PROBE_ITERS=3 \
    python tools/env_probe.py "base:" "logv1:FDX_TILE_LOGV=1" "noavl2:FDX_TILE_LOGV=2,FDX_TILE_NO_AVL2=1"
"""
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

import bench  # noqa: E402
from flashdeconv_amd import FlashDeconv  # noqa: E402


def main():
    n = int(os.environ.get("PROBE_N", 1_000_000))
    G = int(os.environ.get("PROBE_G", 2000))
    K = int(os.environ.get("PROBE_K", 30))
    d = int(os.environ.get("PROBE_D", 512))
    iters = int(os.environ.get("PROBE_ITERS", 3))
    fam = os.environ.get("PROBE_FAMILY", "counts")
    dev = torch.device("cuda:0")
    gen, pre = (bench.gen_gaussian, "raw") if fam == "gaussian" else (bench.gen_counts, "log_cpm")
    Y, X, coords = gen(torch, n, G, K, dev, 0)
    variants = []
    for spec in sys.argv[1:]:
        name, _, envs = spec.partition(":")
        variants.append((name, dict(e.split("=", 1) for e in envs.split(",") if e)))
    touched = sorted({k for _, env in variants for k in env})
    ref = None
    for name, env in variants:
        for k in touched:
            os.environ.pop(k, None)
        os.environ.update(env)
        import time
        best = None
        walls = []
        for _ in range(int(os.environ.get("PROBE_REPS", 4))):
            m = FlashDeconv(sketch_dim=d, preprocess=pre, n_hvg=G, max_iter=iters)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            m.fit(Y, X, coords, output="torch")
            torch.cuda.synchronize()
            walls.append((time.perf_counter() - t0) * 1e3)
            t = dict(m.timings_)
            if best is None or t["total_ms"] < best["total_ms"]:
                best = t
        walls.sort()
        best["wall_min_ms"], best["wall_med_ms"] = walls[0], walls[len(walls) // 2]
        best["n_iter"] = int(m.info_["n_iterations"])
        beta = m.beta_.double().cpu().numpy()
        if ref is None:
            ref = beta
        err = float(np.linalg.norm(beta - ref) / np.linalg.norm(ref))
        print(json.dumps({"variant": name, "rel_diff_vs_first": err, **{k: round(v, 3) for k, v in best.items()}}), flush=True)


if __name__ == "__main__":
    main()
