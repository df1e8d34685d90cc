"""Device build of the restated cKDTree (csrc/kdtree_build_dev.cpp) on a square lattice: wall per build.  python tools/kd_dev_probe.py [side]"""
import ctypes
import sys
import time

import numpy as np
import torch

import os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from flashdeconv_amd import _lib  # noqa: E402

lib = _lib.load()
side = int(sys.argv[1]) if len(sys.argv) > 1 else 1000
coords = np.ascontiguousarray(np.stack(np.meshgrid(np.arange(float(side)), np.arange(float(side)), indexing="ij"), -1).reshape(-1, 2))
n = len(coords)
cd = torch.from_numpy(coords).cuda()
got = np.empty(n, dtype=np.int64)
info = np.zeros(3, dtype=np.int32)
for _ in range(4):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    _lib.check(lib.fdx_ckdtree_indices_dev(ctypes.c_void_p(cd.data_ptr()), n, 2, _lib.ptr_i64(got), _lib.ptr_i32(info), None))
    print("build + download %.2f ms" % ((time.perf_counter() - t0) * 1e3), info)
