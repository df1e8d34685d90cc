#!/usr/bin/env python3
"""One rank's share through the real driver (ShardedFlashDeconv + LoopbackComm), repeated: host timeline with FDX_TRACE_DRIVER=1.
usage: driver_probe.py [W = 8] [rank = W // 2] [n = 1_000_000]"""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))


def main():
    import torch
    import virtual_ranks as vr
    from flashdeconv_amd.distributed import LoopbackComm, ShardedFlashDeconv, shard_bounds
    W = int(sys.argv[1]) if len(sys.argv) > 1 else 8
    r = int(sys.argv[2]) if len(sys.argv) > 2 else W // 2
    n = int(sys.argv[3]) if len(sys.argv) > 3 else 1_000_000
    G, K, d = 2000, 30, 512
    dev = torch.device("cuda", 0)
    g = torch.Generator(device=dev)
    g.manual_seed(11)
    X32 = torch.randn(K, G, generator=g, device=dev, dtype=torch.float32)
    X = X32.double().cpu().numpy()
    raw = torch.rand(n, 2, generator=g, device=dev, dtype=torch.float64) * float(np.sqrt(n))
    coords = vr.morton_sorted_coords(torch, raw)
    bounds = shard_bounds(n, W)
    Y = vr.gaussian_rows(torch, X32, int(bounds[r]), int(bounds[r + 1]), 11)
    totals = {(3,): torch.tensor([7.06 * n, 0.0, 0.0], dtype=torch.float64, device=dev)}
    model = ShardedFlashDeconv(sketch_dim=d, preprocess="raw", n_hvg=G, max_iter=7, tol=1e-300, comm=LoopbackComm(r, W, totals),
                               knn_ties=os.environ.get("PROBE_TIES", "auto"))
    for rep in range(6):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        model.plan(coords, X)
        model.fit_transform(Y, X)
        torch.cuda.synchronize()
        print(f"rep {rep}: {1e3 * (time.perf_counter() - t0):.3f} ms", file=sys.stderr)


if __name__ == "__main__":
    main()
