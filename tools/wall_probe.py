"""Where does the wall time of one fit go? (gaussian/raw 1M x 2000 x 30)"""
import os, sys, time, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import bench
from flashdeconv_amd import FlashDeconv
from flashdeconv_amd.utils import genes
from flashdeconv_amd.core.sketching import countsketch_tables
dev = torch.device("cuda", 0)
Y, X, coords = bench.gen_gaussian(torch, 1_000_000, 2000, 30, dev, 0)
m = FlashDeconv(sketch_dim=512, preprocess="raw", n_hvg=2000)
for _ in range(2): m.fit(Y, X, coords, output="torch")
for _ in range(3):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    m.fit(Y, X, coords, output="torch")
    torch.cuda.synchronize(); t1 = time.perf_counter()
    print(json.dumps({"wall_ms": round((t1 - t0) * 1e3, 2), **{k: round(v, 2) for k, v in m.timings_.items()}}))
t0 = time.perf_counter(); lev = genes.compute_leverage_scores(X); t1 = time.perf_counter()
b, w = countsketch_tables(2000, 512, lev, 0); t2 = time.perf_counter()
print("leverage_ms", round((t1 - t0) * 1e3, 2), "tables_ms", round((t2 - t1) * 1e3, 2))
