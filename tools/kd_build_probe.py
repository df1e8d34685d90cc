"""Host time of the restated cKDTree's build (csrc/kdtree_order.cpp) on a square lattice, by thread share, by the node size from which
a node's passes are cut into pool tasks, and by the size up to which subtrees are built on a contiguous copy.
python tools/kd_build_probe.py [side]"""
import sys
import time

import numpy as np

sys.path.insert(0, ".")
from flashdeconv_amd import _lib  # noqa: E402

lib = _lib.load()
side = int(sys.argv[1]) if len(sys.argv) > 1 else 1000
coords = np.ascontiguousarray(np.stack(np.meshgrid(np.arange(float(side)), np.arange(float(side)), indexing="ij"), -1).reshape(-1, 2))
n = len(coords)
rows = np.zeros(1, dtype=np.int64)
out = np.empty((1, 7), dtype=np.int64)
for threads in (0, 8):
    for local_max in (65536,):
        for team_min in (1 << 40, 400000, 200000, 100000):
            lib.fdx_kdtree_set_threads(threads)
            lib.fdx_kdtree_tune(0, team_min)
            lib.fdx_kdtree_tune(1, local_max)
            ts = []
            for _ in range(9):
                t0 = time.perf_counter()
                _lib.check(lib.fdx_ckdtree_knn_rows(_lib.ptr_f64(coords), n, 2, 7, _lib.ptr_i64(rows), 1, _lib.ptr_i64(out)))
                ts.append((time.perf_counter() - t0) * 1e3)
            print(f"threads {threads:2d} local_max {local_max:7d} team_min {team_min:>14d}: min {min(ts):6.2f} median {sorted(ts)[4]:6.2f} max {max(ts):6.2f} ms",
                  flush=True)
