#!/usr/bin/env python3
"""Iteration loop of one rank's shard alone (loopback transport), for sweep-kernel experiments: prints ms per solve of n_iter sweeps.
usage: solve_probe.py [W = 8] [rank = 3] [n = 1_000_000] [n_iter = 7]"""
import ctypes
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
    from flashdeconv_amd.distributed import shard_bounds
    lib = _lib.load()
    W = int(sys.argv[1]) if len(sys.argv) > 1 else 8
    r = int(sys.argv[2]) if len(sys.argv) > 2 else W // 2
    n = int(sys.argv[3]) if len(sys.argv) > 3 else 1_000_000
    n_iter = int(sys.argv[4]) if len(sys.argv) > 4 else 7
    K = 30
    dev = torch.device("cuda", 0)
    g = torch.Generator(device=dev)
    g.manual_seed(11)
    raw = torch.rand(n, 2, generator=g, device=dev, dtype=torch.float64) * float(np.sqrt(n))
    coords = vr.morton_sorted_coords(torch, raw)
    bounds = shard_bounds(n, W)
    gr, own = vr.shard_graph(torch, coords, W, r, bounds)
    vr.shard_status(gr)
    nh = ctypes.c_int64(0)
    _lib.check(lib.fdx_graph_halo_info(gr.handle, ctypes.byref(nh), None, None))
    n_own = int(bounds[r + 1] - bounds[r])
    ld = ((n_own + int(nh.value) + 1 + 63) // 64) * 64
    H = torch.rand((K, ld), generator=g, device=dev, dtype=torch.float64)
    A = torch.rand((K, K), generator=g, device=dev, dtype=torch.float64)
    XtX = (A @ A.T + K * torch.eye(K, device=dev, dtype=torch.float64)).contiguous()
    bufs = [torch.empty((K, ld), dtype=torch.float64, device=dev) for _ in range(2)]
    comm = ctypes.c_void_p()
    _lib.check(lib.fdx_comm_init_loopback(r, W, ctypes.byref(comm)))
    info, which, rel = _lib.SolveInfo(), ctypes.c_int32(0), np.zeros(max(n_iter, 1))
    best = None
    for rep in range(8):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        _lib.check(lib.fdx_sharded_solve_dev(comm, gr.handle, ctypes.c_void_p(H.data_ptr()), ld, ctypes.c_void_p(XtX.data_ptr()), K, 0.1, 0.3,
                                             0.0, n_iter, ctypes.c_void_p(bufs[0].data_ptr()), ctypes.c_void_p(bufs[1].data_ptr()), ld,
                                             ctypes.byref(info), _lib.ptr_f64(rel), ctypes.byref(which),
                                             ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)))
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) * 1e3
        best = dt if best is None else min(best, dt)
    print(f"W {W} rank {r} n_own {n_own}: solve of {n_iter} sweeps {best:.3f} ms, loop (hipEvents) {info.sweep_ms:.3f} ms")


if __name__ == "__main__":
    main()
