"""Does the single-workgroup leverage kernel overlap with the graph build?  (null stream vs an explicit stream)"""
import os, sys, time, ctypes
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from flashdeconv_amd import _lib
from flashdeconv_amd.utils.genes import LeverageJob, compute_leverage_scores
lib = _lib.load()
dev = torch.device("cuda", 0)
n = 1_000_000
coords = torch.rand(n, 2, device=dev, dtype=torch.float64) * 1000.0
X = np.random.default_rng(0).standard_normal((30, 2000))

def build(stream):
    h = ctypes.c_void_p()
    _lib.check(lib.fdx_graph_build_dev(ctypes.c_void_p(coords.data_ptr()), n, 2, _lib.GRAPH_KNN, 6, 0.0, ctypes.c_void_p(stream), ctypes.byref(h)))
    return _lib.Graph(h.value)

def run(tag, stream, overlap):
    for rep in range(4):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        job = LeverageJob(X) if overlap else None
        t1 = time.perf_counter()
        g = build(stream)
        if stream: torch.cuda.synchronize()
        t2 = time.perf_counter()
        lev = job.result() if overlap else compute_leverage_scores(X)
        t3 = time.perf_counter()
        g.close()
    print(tag, "begin %.2f graph %.2f leverage %.2f total %.2f ms" % ((t1 - t0) * 1e3, (t2 - t1) * 1e3, (t3 - t2) * 1e3, (t3 - t0) * 1e3))

run("null stream, sequential ", 0, False)
run("null stream, overlapped ", 0, True)
s = torch.cuda.Stream()
run("own stream,  sequential ", s.cuda_stream, False)
run("own stream,  overlapped ", s.cuda_stream, True)
