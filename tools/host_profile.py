#!/usr/bin/env python3
"""cProfile of the host side of one fit (gaussian/raw 1M x 2000 x 30): where the Python / ctypes time goes."""
import cProfile
import io
import os
import pstats
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

import bench  # noqa: E402
from flashdeconv_amd import FlashDeconv  # noqa: E402

dev = torch.device("cuda", 0)
n = int(os.environ.get("PROBE_N", 1_000_000))
Y, X, coords = bench.gen_gaussian(torch, n, 2000, 30, dev, 0)
m = FlashDeconv(sketch_dim=512, preprocess="raw", n_hvg=2000)
for _ in range(3):
    m.fit(Y, X, coords, output="torch")
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(10):
    m.fit(Y, X, coords, output="torch")
torch.cuda.synchronize()
print("ms per fit (no profiler):", (time.perf_counter() - t0) * 100, {k: round(v, 3) for k, v in m.timings_.items()})
pr = cProfile.Profile()
pr.enable()
for _ in range(10):
    m.fit(Y, X, coords, output="torch")
torch.cuda.synchronize()
pr.disable()
s = io.StringIO()
pstats.Stats(pr, stream=s).sort_stats("cumulative").print_stats(35)
print(s.getvalue())
