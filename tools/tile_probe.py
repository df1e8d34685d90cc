#!/usr/bin/env python3
"""Times the sketch -> H stage of one fit for the kernel variants (tile kernel with 16 / 8 waves, the atomic fused kernel,
the two-kernel path) on the bench's synthetic inputs.  Usage: python tools/tile_probe.py [n] [G] [K] [d]"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

import bench  # noqa: E402
from flashdeconv_amd import FlashDeconv  # noqa: E402

SWITCHES = ("FDX_TILE_CFG", "FDX_NO_TILE", "FDX_FUSED", "FDX_NO_FUSED", "FDX_ROWREG")


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000
    G = int(sys.argv[2]) if len(sys.argv) > 2 else 2000
    K = int(sys.argv[3]) if len(sys.argv) > 3 else 30
    d = int(sys.argv[4]) if len(sys.argv) > 4 else 512
    dev = torch.device("cuda:0")
    variants = [("tile", {}), ("rowreg", {"FDX_ROWREG": "1"}), ("tile12+4", {"FDX_TILE_CFG": "12"}), ("tile16+0", {"FDX_TILE_CFG": "16"}), ("tile8+2", {"FDX_TILE_CFG": "8"}),
                ("atomic", {"FDX_NO_TILE": "1", "FDX_FUSED": "1"}), ("two-kernel", {"FDX_NO_FUSED": "1"})]
    only = os.environ.get("FDX_PROBE_VARIANTS")
    if only:
        variants = [v for v in variants if v[0] in only.split(",")]
    for fam, gen, pre in (("gaussian/raw", bench.gen_gaussian, "raw"), ("counts/log_cpm", bench.gen_counts, "log_cpm")):
        Y, X, coords = gen(torch, n, G, K, dev, 0)
        ref = None
        for name, env in variants:
            for k in SWITCHES:
                os.environ.pop(k, None)
            os.environ.update(env)
            ts = []
            for _ in range(4):
                m = FlashDeconv(sketch_dim=d, preprocess=pre, n_hvg=G, max_iter=3)
                m.fit(Y, X, coords, output="torch")
                ts.append((m.timings_["sketch_ms"], m.timings_["gram_ms"]))
            beta = m.beta_.double().cpu().numpy()
            if ref is None:
                ref = beta
            err = float(np.linalg.norm(beta - ref) / np.linalg.norm(ref))
            s, g = min(ts)
            print(f"{fam:16s} {name:10s} sketch {s:7.3f} ms  gram {g:6.3f} ms  sum {s + g:7.3f}  rel diff vs first {err:.2e}",
                  flush=True)
        del Y
    for k in SWITCHES:
        os.environ.pop(k, None)


if __name__ == "__main__":
    main()
