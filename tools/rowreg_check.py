#!/usr/bin/env python3
"""Row-register kernel against the tile kernel (default; the row-register kernel is opt-in: FDX_ROWREG=1) on count data: odd spot counts, gene counts that are
not whole blocks, rows with negative / NaN entries (redo list).  Prints the relative difference of beta."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

import bench  # noqa: E402
from flashdeconv_amd import FlashDeconv  # noqa: E402


def fit(Y, X, coords, d, pre="log_cpm"):
    m = FlashDeconv(sketch_dim=d, preprocess=pre, n_hvg=Y.shape[1], max_iter=5)
    m.fit(Y, X, coords, output="torch")
    return m.beta_.double().cpu().numpy(), m.timings_["sketch_ms"]


def main():
    dev = torch.device("cuda:0")
    worst = 0.0
    for n, G, K, d, bad in ((4096, 2000, 30, 512, 0), (5001, 1996, 12, 512, 0), (777, 2048, 30, 500, 0), (3000, 520, 7, 64, 0),
                            (10000, 2000, 30, 512, 3), (200000, 2000, 30, 512, 0)):
        Y, X, coords = bench.gen_counts(torch, n, G, K, dev, 1)
        if bad:
            Y[17, 5] = -3.0
            Y[n - 1, G - 1] = float("nan")
            Y[4000, 100] = -0.5
        out = {}
        for name, env in (("rowreg", {"FDX_ROWREG": "1"}), ("tile", {})):
            os.environ.pop("FDX_ROWREG", None)
            os.environ.update(env)
            out[name] = fit(Y, X, coords, d)
        os.environ.pop("FDX_ROWREG", None)
        a, b = out["rowreg"][0], out["tile"][0]
        fin = np.isfinite(b)
        same_nan = np.array_equal(np.isfinite(a), fin)
        err = float(np.linalg.norm(a[fin] - b[fin]) / max(np.linalg.norm(b[fin]), 1e-300))
        worst = max(worst, err)
        print(f"n={n} G={G} K={K} d={d} bad={bad}: rel diff {err:.2e} same-nonfinite {same_nan} "
              f"sketch rowreg {out['rowreg'][1]:.3f} ms tile {out['tile'][1]:.3f} ms", flush=True)
    print("worst", worst)
    return 0 if worst < 1e-9 else 1


if __name__ == "__main__":
    sys.exit(main())
