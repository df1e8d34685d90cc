"""Rates of the threaded host transfers (csrc/host_transfer.cpp) against one pinned copy, by thread count."""
import ctypes, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from flashdeconv_amd import _lib
lib = _lib.load()
_lib.require_gpu()
g = ctypes.c_double(0.0)
_lib.check(lib.fdx_pinned_copy_rate(1 << 30, 1, ctypes.byref(g))); print("pinned h2d GB/s", round(g.value, 1))
_lib.check(lib.fdx_pinned_copy_rate(1 << 30, 0, ctypes.byref(g))); print("pinned d2h GB/s", round(g.value, 1))
n = 2_000_000_000
a = np.ones(n, dtype=np.float32)
for T in (4, 8, 12, 16, 24, 32):
    os.environ["FDX_TRANSFER_THREADS"] = str(T)
    best = 0
    for rep in range(3):
        t0 = time.perf_counter(); ptr, code = _lib.upload_matrix(a); dt = time.perf_counter() - t0
        lib.fdx_free(ptr)
        best = max(best, a.nbytes / dt / 1e9)
    out = np.empty(120_000_000, dtype=np.float64)
    ptr = ctypes.c_void_p(); _lib.check(lib.fdx_malloc(ctypes.byref(ptr), out.nbytes))
    t0 = time.perf_counter(); _lib.download_bytes(out, ptr); d1 = time.perf_counter() - t0      # fresh pages
    t0 = time.perf_counter(); _lib.download_bytes(out, ptr); d2 = time.perf_counter() - t0      # touched pages
    lib.fdx_free(ptr)
    print(f"threads {T}: upload {best:.1f} GB/s; download 0.96 GB fresh {out.nbytes / d1 / 1e9:.1f} GB/s, touched {out.nbytes / d2 / 1e9:.1f} GB/s", flush=True)
t0 = time.perf_counter(); b = np.empty(120_000_000); b[:] = 0; print("first touch of 0.96 GB by one thread ms", round((time.perf_counter() - t0) * 1e3, 1))
