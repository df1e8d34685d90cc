#!/usr/bin/env python3
"""Bit-for-bit repeatability of complete fits at the bench shape: SOAK_REPS fits per family, every result compared with the
first one's (beta_, proportions_, iteration count, objective)."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

import bench  # noqa: E402
from flashdeconv_amd import FlashDeconv  # noqa: E402


def main():
    reps = int(os.environ.get("SOAK_REPS", 30))
    n = int(os.environ.get("SOAK_N", 1_000_000))
    dev = torch.device("cuda:0")
    bad = 0
    for fam, gen, pre, iters in (("gaussian", bench.gen_gaussian, "raw", 100), ("counts", bench.gen_counts, "log_cpm", 10)):
        Y, X, coords = gen(torch, n, 2000, 30, dev, 0)
        ref = None
        for r in range(reps):
            m = FlashDeconv(sketch_dim=512, preprocess=pre, n_hvg=2000, max_iter=iters).fit(Y, X, coords, output="torch")
            cur = (m.beta_.clone(), m.proportions_.clone(), m.info_["n_iterations"], m.info_["final_objective"])
            if ref is None:
                ref = cur
            elif not (torch.equal(cur[0], ref[0]) and torch.equal(cur[1], ref[1]) and cur[2] == ref[2] and cur[3] == ref[3]):
                bad += 1
                print(fam, "fit", r, "differs: max |d beta|", float((cur[0] - ref[0]).abs().max()), cur[2], ref[2], flush=True)
        print(fam, reps, "fits, iterations", ref[2], "objective", ref[3], flush=True)
        del Y
    print("soak done; differing fits:", bad)
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
