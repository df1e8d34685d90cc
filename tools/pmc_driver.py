#!/usr/bin/env python3
"""Workload for the counter passes of tools/pmc_round.sh: two complete fits per family (gaussian/raw, count-like/log_cpm)
at the bench shape, default kernel selection unless FDX_* switches are set by the caller.
Usage: [PMC_FAMILIES=gaussian,counts,csr] python3 tools/pmc_driver.py [n] [G] [K] [d]"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

import bench  # noqa: E402
from flashdeconv_amd import FlashDeconv  # noqa: E402


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000
    G = int(sys.argv[2]) if len(sys.argv) > 2 else 2000
    K = int(sys.argv[3]) if len(sys.argv) > 3 else 30
    d = int(sys.argv[4]) if len(sys.argv) > 4 else 512
    dev = torch.device("cuda:0")
    only = os.environ.get("PMC_FAMILIES", "gaussian,counts,csr").split(",")
    for fam, gen, pre, iters in (("gaussian/raw", bench.gen_gaussian, "raw", 100), ("counts/log_cpm", bench.gen_counts, "log_cpm", 12)):
        if fam.split("/")[0] not in only:
            continue
        Y, X, coords = gen(torch, n, G, K, dev, 0)
        for _ in range(2):
            m = FlashDeconv(sketch_dim=d, preprocess=pre, n_hvg=G, max_iter=iters)
            m.fit(Y, X, coords, output="torch")
        print(fam, {k: round(v, 3) for k, v in m.timings_.items()}, flush=True)
        del Y
    if "csr" in only and os.environ.get("PMC_WITH_CSR", "1") != "0" and n >= 100_000:     # the CSR family: gene statistics + fused sketch -> H
        Y, X, coords = bench.gen_sparse(torch, n, 20000, K, dev, seed=0)
        for _ in range(2):
            m = FlashDeconv(sketch_dim=d, preprocess="log_cpm", n_hvg=G, max_iter=5)
            m.fit(Y, X, coords, output="torch")
        print("sparse_csr", {k: round(v, 3) for k, v in m.timings_.items()}, flush=True)


if __name__ == "__main__":
    main()
