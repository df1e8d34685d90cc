#!/usr/bin/env python3
"""Host wall time per C entry point over a few fits (gaussian/raw 1M x 2000 x 30): every bound libfdx function is wrapped by
a timing shim, so the part of a fit's wall time that no stage timer covers shows up by name."""
import collections
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

import bench  # noqa: E402
from flashdeconv_amd import FlashDeconv, _lib  # noqa: E402

dev = torch.device("cuda", 0)
n = int(os.environ.get("PROBE_N", 1_000_000))
Y, X, coords = bench.gen_gaussian(torch, n, 2000, 30, dev, 0)
m = FlashDeconv(sketch_dim=512, preprocess="raw", n_hvg=2000)
for _ in range(3):
    m.fit(Y, X, coords, output="torch")
lib = _lib.load()
acc = collections.defaultdict(lambda: [0, 0.0])


class Shim:
    def __init__(self, lib):
        object.__setattr__(self, "_lib", lib)

    def __getattr__(self, name):
        fn = getattr(object.__getattribute__(self, "_lib"), name)

        def call(*a):
            t0 = time.perf_counter()
            r = fn(*a)
            e = acc[name]
            e[0] += 1
            e[1] += time.perf_counter() - t0
            return r
        return call


_lib._LIB = Shim(lib) if hasattr(_lib, "_LIB") else None
orig_load = _lib.load
shim = Shim(lib)
_lib.load = lambda: shim
import flashdeconv_amd.core.deconv as dc  # noqa: E402
import flashdeconv_amd.utils.genes as gn  # noqa: E402
for mod in (dc, gn):
    if hasattr(mod, "_lib"):
        mod._lib.load = _lib.load
reps = 10
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(reps):
    m.fit(Y, X, coords, output="torch")
torch.cuda.synchronize()
wall = (time.perf_counter() - t0) / reps * 1e3
print(f"wall per fit {wall:.3f} ms; stage total {m.timings_['total_ms']:.3f} ms")
tot = 0.0
for name, (cnt, t) in sorted(acc.items(), key=lambda kv: -kv[1][1]):
    print(f"  {name:34s} {cnt / reps:5.1f} calls/fit  {t / reps * 1e3:8.3f} ms/fit")
    tot += t / reps * 1e3
print(f"  sum of C calls {tot:.3f} ms/fit; python outside them {wall - tot:.3f} ms/fit")
