"""Where the wall of a host-array fit goes (dense float32 1M x 2000 x 30)."""
import ctypes, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import bench
from flashdeconv_amd import FlashDeconv, _lib
from flashdeconv_amd.core import deconv as D
lib = _lib.load()
dev = torch.device("cuda", 0)
n, G, K, d = 1_000_000, 2000, 30, 512
Y, X, coords = bench.gen_gaussian(torch, n, G, K, dev, 0)
ch = _lib.tensor_to_host(coords); Yh = _lib.tensor_to_host(Y); del Y; torch.cuda.empty_cache()
m = FlashDeconv(sketch_dim=d, preprocess="raw", n_hvg=G)
for rep in range(3):
    t0 = time.perf_counter(); m.fit(Yh, X, ch); t1 = time.perf_counter()
    print("fit numpy->numpy ms", round((t1 - t0) * 1e3, 1), {k: round(v, 1) for k, v in m.timings_.items() if k in ("host_pre_ms", "span_ms", "host_post_ms")}, flush=True)
for rep in range(2):
    t0 = time.perf_counter(); ptr, code = _lib.upload_matrix(Yh); t1 = time.perf_counter()
    buf = D._DeviceBuffer.adopt(ptr, Yh.nbytes); buf.free(); t2 = time.perf_counter()
    print("upload ms", round((t1 - t0) * 1e3, 1), "free ms", round((t2 - t1) * 1e3, 2))
b = D._DeviceBuffer(n * K * 8)
for rep in range(2):
    t0 = time.perf_counter(); o = b.to_host((n, K)); t1 = time.perf_counter(); print("to_host 240 MB ms", round((t1 - t0) * 1e3, 1))
import cProfile, pstats
pr = cProfile.Profile(); pr.enable(); m.fit(Yh, X, ch); pr.disable()
pstats.Stats(pr).sort_stats("cumtime").print_stats(14)
