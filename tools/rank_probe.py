#!/usr/bin/env python3
"""One rank's whole share of the sharded fit (tools/virtual_ranks.py: alone_pipelined), repeated, for a kernel trace:

    cd /tmp && rocprofv3 --kernel-trace --output-format csv -d out -- python3 /root/repo/tools/rank_probe.py 8 3
    python3 tools/timeline.py out            # launch-order timeline of the last repetition (marker: normalize_export)

usage: rank_probe.py [W = 8] [rank = W // 2] [n = 1_000_000] [config5 = 0]"""
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
    from flashdeconv_amd import _lib
    from flashdeconv_amd.distributed import diag_mean, shard_bounds
    W = int(sys.argv[1]) if len(sys.argv) > 1 else 8
    r = int(sys.argv[2]) if len(sys.argv) > 2 else W // 2
    n = int(sys.argv[3]) if len(sys.argv) > 3 else 1_000_000
    big = len(sys.argv) > 4 and sys.argv[4] == "1"
    G, K, d = (5000, 50, 1024) if big else (2000, 30, 512)
    torch.cuda.set_device(0)
    dev = torch.device("cuda", 0)
    g = torch.Generator(device=dev)
    g.manual_seed(11)
    X32 = torch.randn(K, G, generator=g, device=dev, dtype=torch.float32)
    X = X32.double().cpu().numpy()
    raw = torch.rand(n, 2, generator=g, device=dev, dtype=torch.float64) * float(np.sqrt(n))
    coords = vr.morton_sorted_coords(torch, raw)
    bounds = shard_bounds(n, W)
    times = {}
    ranks, nnz, ties, bounds = vr.virtual_plan(torch, coords, W, 6, times)
    S = ranks[r]
    S["H"] = torch.empty(1, device=dev)
    S["ld"] = ((S["n_own"] + S["n_halo"] + 1 + 63) // 64) * 64
    lam, rho_eff = 0.01, 0.3
    vr.alone_pipelined(torch, coords, [S] if False else ranks[r:r + 1] * 1, bounds, X, lambda lo, hi: vr.gaussian_rows(torch, X32, lo, hi, 11),
                       d, _lib.PRE_RAW, K, lam, rho_eff, 7, times, rank_ids=[r], world=W)
    print("pipelined ms:", times["pipelined_ms"])


if __name__ == "__main__":
    main()
