#!/usr/bin/env python3
"""A/B driver for experiment builds (make EXTRA="-DFDX_SWEEP_EXPERIMENT -DFDX_TILE_EXPERIMENT"): one configs[4]-shaped shard
(1.25M x 5000 x 50, d 1024; EXP_N / EXP_G / EXP_K / EXP_D / EXP_FAMILY override), the same fit under a list of environment
settings; prints the stage times of each and whether the proportions equal the first setting's bit for bit.
usage: python tools/exp_c5.py "" "FDX_SWEEP_EXP=4l" "FDX_TILE_DBG=1" ...   ('' = default build behaviour)"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

import bench  # noqa: E402
from flashdeconv_amd import FlashDeconv  # noqa: E402


def main():
    n, G, K, d = (int(os.environ.get("EXP_N", 1_250_000)), int(os.environ.get("EXP_G", 5000)), int(os.environ.get("EXP_K", 50)),
                  int(os.environ.get("EXP_D", 1024)))
    fam = os.environ.get("EXP_FAMILY", "gaussian")
    dev = torch.device("cuda:0")
    if fam == "gaussian":
        Y, X, coords = bench.gen_gaussian(torch, n, G, K, dev, 0)
        kw = dict(sketch_dim=d, preprocess="raw", n_hvg=G)
    else:
        Y, X, coords = bench.gen_counts(torch, n, G, K, dev, 0)
        kw = dict(sketch_dim=d, preprocess="log_cpm", n_hvg=G, max_iter=int(os.environ.get("EXP_ITERS", 100)))
    settings = sys.argv[1:] or [""]
    keys = set()
    for s in settings:
        for kv in s.split():
            keys.add(kv.split("=")[0])
    ref = None
    for s in settings:
        for k in keys:
            os.environ.pop(k, None)
        for kv in s.split():
            k, v = kv.split("=")
            os.environ[k] = v
        m = FlashDeconv(**kw)
        m.fit(Y, X, coords, output="torch")
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        reps = int(os.environ.get("EXP_REPS", 3))
        acc = {}
        for _ in range(reps):
            m.fit(Y, X, coords, output="torch")
            for k, v in m.timings_.items():
                acc[k] = acc.get(k, 0.0) + v / reps
        torch.cuda.synchronize()
        ms = (time.perf_counter() - t0) / reps * 1e3
        T = m.info_["n_iterations"]
        P = m.proportions_
        same = None
        if ref is None:
            ref = P.clone()
        else:
            same = bool(torch.equal(ref, P))
        print(f"[{s or 'default':28s}] step {ms:7.3f} ms  sketch {acc['sketch_ms']:6.3f}  sweeps {acc['sweep_ms']:6.3f} / {T} = "
              f"{acc['sweep_ms'] / max(T, 1):.4f}  solve {acc['solve_ms']:6.3f}  finish {acc['finish_ms']:.3f}  bits==first {same}", flush=True)


if __name__ == "__main__":
    main()
