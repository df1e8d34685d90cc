"""ctypes binding of libfdx.so (the C ABI declared in include/fdx.h).

This module is the whole Python<->native boundary: plain pointers and sizes.  There is no CPU
fallback: if the shared library is missing or no MI355X is visible, the calls raise.
"""
import ctypes
import os
import threading

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libfdx.so")

c_int = ctypes.c_int
c_i32 = ctypes.c_int32
c_i64 = ctypes.c_int64
c_double = ctypes.c_double
c_void_p = ctypes.c_void_p
c_size_t = ctypes.c_size_t
p_double = ctypes.POINTER(ctypes.c_double)
p_i64 = ctypes.POINTER(ctypes.c_int64)
p_i32 = ctypes.POINTER(ctypes.c_int32)


class FdxError(RuntimeError):
    """A libfdx call failed (message from fdx_last_error())."""


class SolveInfo(ctypes.Structure):
    _fields_ = [
        ("converged", c_i32),
        ("n_iterations", c_i32),
        ("final_objective", c_double),
        ("final_change", c_double),
        ("n_objectives", c_i32),
        ("reserved", c_i32),
        ("sweep_ms", c_double),
        ("total_ms", c_double),
    ]


class FitParams(ctypes.Structure):
    _fields_ = [
        ("sketch_dim", c_i32), ("mode_y", c_i32), ("mode_x", c_i32), ("graph_method", c_i32), ("k_neighbors", c_i32),
        ("lambda_auto", c_i32), ("max_iter", c_i32), ("verbose", c_i32),
        ("radius", c_double), ("lambda_spatial", c_double), ("rho_sparsity", c_double), ("tol", c_double),
        ("stop_on_ties", c_i32), ("reserved", c_i32), ("carry", c_void_p),
    ]


class ShardFitParams(ctypes.Structure):
    """fdx_shard_fit_params (include/fdx.h)."""
    _fields_ = [("sketch_dim", c_i32), ("mode_y", c_i32), ("mode_x", c_i32), ("lambda_auto", c_i32), ("max_iter", c_i32),
                ("stop_on_ties", c_i32), ("lambda_spatial", c_double), ("rho_sparsity", c_double), ("tol", c_double),
                ("n_total_spots", c_i64), ("nnz_total", c_i64), ("X_dev", c_void_p)]


class ShardFitInfo(ctypes.Structure):
    """fdx_shard_fit_info (include/fdx.h)."""
    _fields_ = [("status", c_i32), ("reserved", c_i32), ("nnz_total", c_i64), ("knn_ties_total", c_i64), ("own_nnz", c_i64),
                ("n_halo", c_i64), ("lambda_used", c_double), ("rho_effective", c_double), ("YtY", c_double), ("solve", SolveInfo)]


SHARD_FAR, SHARD_OVERFLOW, SHARD_TIES = 1, 2, 3


class CsrView(ctypes.Structure):
    """fdx_csr_view: device pointers of a CSR matrix (include/fdx.h)."""
    _fields_ = [("indptr", c_void_p), ("indices", c_void_p), ("data", c_void_p), ("dtype", c_i32), ("n", c_i64),
                ("nnz", c_i64), ("G", c_i32), ("sorted_rows", c_i32)]


class FitInfo(ctypes.Structure):
    _fields_ = [
        ("solve", SolveInfo), ("lambda_used", c_double), ("rho_effective", c_double), ("YtY", c_double), ("nnz", c_i64),
        ("graph_ms", c_double), ("sketch_ms", c_double), ("gram_ms", c_double), ("solve_ms", c_double),
        ("finish_ms", c_double), ("total_ms", c_double), ("prologue_ms", c_double), ("span_ms", c_double),
        ("knn_ties", c_i64), ("status", c_i32), ("reserved", c_i32), ("carry", c_void_p),
    ]


FIT_TIES = 3


GRAPH_KNN, GRAPH_RADIUS, GRAPH_GIVEN = 0, 1, 2

# name -> (restype, argtypes); kept in one table so tests can check it against include/fdx.h
SIGNATURES = {
    "fdx_column_sums_dev": (c_int, [c_void_p, c_i32, c_i64, c_i32, c_i64, p_double, c_void_p]),
    "fdx_leverage_scores": (c_int, [p_double, c_i32, c_i32, c_double, p_double]),
    "fdx_csr_check_dev": (c_int, [ctypes.POINTER(CsrView), c_void_p]),
    "fdx_csr_gene_moments_dev": (c_int, [ctypes.POINTER(CsrView), p_double, p_double, p_double, c_void_p]),
    "fdx_fit_csr_dev": (c_int, [ctypes.POINTER(CsrView), p_i32, c_i32, p_double, c_i32, p_i32, p_double, p_double, c_void_p,
                                c_i32, ctypes.POINTER(FitParams), ctypes.POINTER(c_void_p), c_void_p, c_void_p, p_double,
                                p_double, ctypes.POINTER(FitInfo), c_void_p]),
    "fdx_type_sums_dev": (c_int, [c_void_p, c_i32, c_i64, c_i32, c_i64, c_void_p, c_void_p, c_i32, c_i32, c_void_p, c_void_p]),
    "fdx_type_sums_csr_dev": (c_int, [ctypes.POINTER(CsrView), c_void_p, c_void_p, c_i32, c_i32, c_void_p, c_void_p]),
    "fdx_graph_build_radius_rows_dev": (c_int, [c_void_p, c_i64, c_i32, c_double, c_i64, c_i64, c_void_p, ctypes.POINTER(c_void_p)]),
    "fdx_graph_knn_lists_dev": (c_int, [c_void_p, c_i64, c_i32, c_i32, c_i64, c_i64, c_void_p, c_void_p, c_void_p,
                                        ctypes.POINTER(c_void_p)]),
    "fdx_graph_knn_lists_band_dev": (c_int, [c_void_p, c_i64, c_i32, c_i32, c_i64, c_i64, c_void_p, c_void_p, c_void_p,
                                             ctypes.POINTER(c_void_p)]),
    "fdx_graph_from_knn_lists_dev": (c_int, [c_void_p, c_void_p, c_void_p, c_i64, c_i64, c_void_p, ctypes.POINTER(c_void_p)]),
    "fdx_leverage_begin": (c_int, [p_double, c_i32, c_i32, c_double, ctypes.POINTER(c_void_p)]),
    "fdx_leverage_begin_opt": (c_int, [p_double, c_i32, c_i32, c_double, c_i32, ctypes.POINTER(c_void_p)]),
    "fdx_leverage_end": (c_int, [c_void_p, p_double]),
    "fdx_leverage_end_keep": (c_int, [c_void_p, p_double, ctypes.POINTER(c_void_p)]),
    "fdx_fit_dev": (c_int, [c_void_p, c_i32, c_i64, c_i32, c_i64, p_double, c_i32, p_i32, p_double, p_double, c_void_p,
                            c_i32, ctypes.POINTER(FitParams), ctypes.POINTER(c_void_p), c_void_p, c_void_p, p_double,
                            p_double, ctypes.POINTER(FitInfo), c_void_p]),
    "fdx_fit_carry_free": (c_int, [c_void_p]),
    "fdx_graph_build_dev": (c_int, [c_void_p, c_i64, c_i32, c_i32, c_i32, c_double, c_void_p, ctypes.POINTER(c_void_p)]),
    "fdx_graph_perm_dev": (c_int, [c_void_p, c_void_p, c_void_p]),
    "fdx_side_stream": (c_int, [ctypes.POINTER(c_void_p)]),
    "fdx_stream_wait_stream": (c_int, [c_void_p, c_void_p]),
    "fdx_graph_localize": (c_int, [c_void_p, c_i32, p_i64, c_i32, c_void_p, ctypes.POINTER(c_void_p)]),
    "fdx_graph_shard_knn_dev": (c_int, [c_void_p, c_i64, c_i32, c_i32, c_i32, p_i64, c_i32, c_void_p, ctypes.POINTER(c_void_p)]),
    "fdx_graph_shard_status": (c_int, [c_void_p, p_i64, p_i64, p_i32, p_i32]),
    "fdx_graph_row_indices": (c_int, [c_void_p, c_i64, p_i32, c_i32, p_i32]),
    "fdx_graph_halo_info": (c_int, [c_void_p, p_i64, p_i32, p_i32]),
    "fdx_graph_send_indices_dev": (c_int, [c_void_p, c_void_p, c_void_p]),
    "fdx_prepare_dev": (c_int, [c_void_p, c_i32, c_i64, c_i32, c_i64, c_void_p, p_double, c_i32, p_i32, p_double, p_double,
                                c_i32, c_i32, c_i32, c_void_p, c_i64, c_void_p, p_double, p_double, c_void_p]),
    "fdx_init_beta_dev": (c_int, [c_void_p, c_i64, c_i64, c_i32, c_void_p]),
    "fdx_bcd_sweep_dev": (c_int, [c_void_p, c_void_p, c_i64, c_void_p, c_void_p, c_void_p, c_i64, c_i32, c_double, c_double,
                                  c_double, c_i32, c_void_p, c_void_p, c_void_p]),
    "fdx_bcd_fold_dev": (c_int, [c_void_p, c_void_p, c_i32, c_void_p]),
    "fdx_objective_partials_dev": (c_int, [c_void_p, c_void_p, c_i64, c_void_p, c_i64, c_void_p, c_i32, p_double, c_void_p]),
    "fdx_normalize_dev": (c_int, [c_void_p, c_i64, c_i64, c_i32, c_void_p, c_void_p, c_void_p]),
    "fdx_gene_moments_dev": (c_int, [c_void_p, c_i32, c_i64, c_i32, c_i64, p_double, p_double, c_void_p]),
    "fdx_gather_columns_dev": (c_int, [c_void_p, c_i32, c_i64, c_i32, c_i64, p_i32, c_i32, c_void_p, c_void_p]),
    "fdx_version": (c_int, []),
    "fdx_env_reload": (c_int, []),
    "fdx_env_switch": (ctypes.c_char_p, [c_i32, ctypes.POINTER(ctypes.c_char_p)]),
    "fdx_last_error": (ctypes.c_char_p, []),
    "fdx_device_count": (c_int, [ctypes.POINTER(c_int)]),
    "fdx_set_device": (c_int, [c_int]),
    "fdx_device_name": (c_int, [ctypes.c_char_p, c_int]),
    "fdx_malloc": (c_int, [ctypes.POINTER(c_void_p), c_size_t]),
    "fdx_free": (c_int, [c_void_p]),
    "fdx_trim": (c_int, []),
    "fdx_memcpy_h2d": (c_int, [c_void_p, c_void_p, c_size_t, c_void_p]),
    "fdx_memcpy_d2h": (c_int, [c_void_p, c_void_p, c_size_t, c_void_p]),
    "fdx_memset": (c_int, [c_void_p, c_int, c_size_t, c_void_p]),
    "fdx_upload_convert_dev": (c_int, [c_void_p, c_i32, c_void_p, c_i32, c_i64, p_double, c_void_p]),
    "fdx_download_dev": (c_int, [c_void_p, c_void_p, c_size_t, c_void_p]),
    "fdx_pinned_copy_rate": (c_int, [c_size_t, c_i32, p_double]),
    "fdx_stream_sync": (c_int, [c_void_p]),
    "fdx_sketch": (c_int, [c_void_p, c_i32, c_i64, c_i32, p_i64, p_i32, p_double, c_i32, c_i32, p_double]),
    "fdx_prepare_csr_dev": (c_int, [ctypes.POINTER(CsrView), p_i32, c_i32, p_double, c_i32, p_i32, p_double, p_double, c_i32, c_i32,
                                    c_i32, c_void_p, c_i64, c_void_p, p_double, ctypes.POINTER(c_double), c_void_p]),
    "fdx_gram_xty": (c_int, [p_double, p_double, c_i64, c_i32, c_i32, p_double, p_double]),
    "fdx_objective": (c_int, [c_void_p, p_double, p_double, p_double, c_i64, c_i32, c_double, c_double, c_double, p_double]),
    "fdx_comm_unique_id": (c_int, [c_void_p]),
    "fdx_comm_init": (c_int, [c_void_p, c_i32, c_i32, ctypes.POINTER(c_void_p)]),
    "fdx_local_world_create": (c_int, [c_i32, ctypes.POINTER(c_void_p)]),
    "fdx_local_world_destroy": (c_int, [c_void_p]),
    "fdx_comm_init_local": (c_int, [c_void_p, c_i32, ctypes.POINTER(c_void_p)]),
    "fdx_comm_init_loopback": (c_int, [c_i32, c_i32, ctypes.POINTER(c_void_p)]),
    "fdx_comm_destroy": (c_int, [c_void_p]),
    "fdx_comm_info": (c_int, [c_void_p, p_i32, p_i32]),
    "fdx_comm_rccl_count": (c_int, [c_void_p, p_i32]),
    "fdx_comm_allreduce_sum_dev": (c_int, [c_void_p, c_void_p, c_i32, c_void_p]),
    "fdx_sharded_solve_dev": (c_int, [c_void_p, c_void_p, c_void_p, c_i64, c_void_p, c_i32, c_double, c_double, c_double, c_i32,
                                      c_void_p, c_void_p, c_i64, ctypes.POINTER(SolveInfo), p_double, p_i32, c_void_p]),
    "fdx_solver_padded_k": (c_i32, [c_i32]),
    "fdx_shard_fit_dev": (c_int, [c_void_p, c_void_p, c_void_p, c_i32, c_i64, c_i32, c_i64, p_double, c_i32, p_i32, p_double, p_double,
                                  ctypes.POINTER(ShardFitParams), c_void_p, c_void_p, p_double, ctypes.POINTER(ShardFitInfo), c_void_p]),
    "fdx_sharded_solve_padded_dev": (c_int, [c_void_p, c_void_p, c_void_p, c_i64, c_void_p, c_i32, c_i32, c_double, c_double, c_double,
                                             c_i32, c_void_p, c_void_p, c_i64, ctypes.POINTER(SolveInfo), p_double, p_i32, c_void_p]),
    "fdx_tile_schedule": (c_int, [p_i32, p_double, c_i32, c_i32, c_i32, c_i32, c_i32, p_i32, p_i32, c_void_p, p_i32, p_double,
                                  c_void_p, c_i64]),
    "fdx_column_sums": (c_int, [c_void_p, c_i32, c_i64, c_i32, p_double]),
    "fdx_log1p_f32": (c_int, [c_void_p, ctypes.c_float, c_i64, c_void_p]),
    "fdx_graph_build_knn": (c_int, [p_double, c_i64, c_i32, c_i32, ctypes.POINTER(c_void_p)]),
    "fdx_graph_build_radius": (c_int, [p_double, c_i64, c_i32, c_double, ctypes.POINTER(c_void_p)]),
    "fdx_nearest_distance": (c_int, [p_double, c_i64, c_i32, p_double]),
    "fdx_graph_export_csr": (c_int, [c_void_p, p_i64, p_i32]),
    "fdx_graph_from_csr": (c_int, [p_i64, p_i64, c_i64, ctypes.POINTER(c_void_p)]),
    "fdx_graph_destroy": (c_int, [c_void_p]),
    "fdx_graph_info": (c_int, [c_void_p, p_i64, p_i64, p_i32]),
    "fdx_graph_knn_ties": (c_int, [c_void_p, p_i64]),
    "fdx_graph_knn_far": (c_int, [c_void_p, ctypes.POINTER(c_i32)]),
    "fdx_kdtree_set_threads": (c_int, [c_i32]),
    "fdx_kdtree_tune": (c_int, [c_i32, c_i64]),
    "fdx_ckdtree_indices_dev": (c_int, [c_void_p, c_i64, c_i32, p_i64, p_i32, c_void_p]),
    "fdx_ckdtree_prebuild": (c_int, [p_double, c_void_p, c_i64, c_i32]),
    "fdx_hvg_from_moments": (c_int, [p_double, p_double, c_i32, p_double, c_i32, c_i32, c_double, c_double, c_double, p_i64,
                                     ctypes.POINTER(c_i32), ctypes.POINTER(c_i32)]),
    "fdx_ckdtree_knn": (c_int, [p_double, c_i64, c_i32, c_i32, c_void_p, c_void_p]),
    "fdx_ckdtree_knn_rows": (c_int, [p_double, c_i64, c_i32, c_i32, p_i64, c_i64, c_void_p]),
    "fdx_graph_plan_set_ckdtree_lists_dev": (c_int, [c_void_p, p_double, c_void_p, c_i64, c_i32, c_void_p, c_i64, c_void_p, c_void_p, c_void_p]),
    "fdx_graph_plan_order_dev": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p]),
    "fdx_graph_plan_lists_replaced": (c_int, [c_void_p]),
    "fdx_graph_plan_set_lists_dev": (c_int, [c_void_p, c_void_p, c_void_p, c_i64, c_void_p, c_void_p, c_void_p]),
    "fdx_bcd_solve": (c_int, [c_void_p, p_double, p_double, c_i64, c_i32, c_i32, c_double, c_double, c_i32, c_double,
                              c_i32, p_double, p_double, p_double, ctypes.POINTER(SolveInfo)]),
}

_lock = threading.Lock()
_lib = None


def _preload_shared_hip_runtime():
    """PyTorch-ROCm wheels bundle their own libamdhip64.so (same soname as the system one).  Two HIP runtimes in one
    process do not coexist (the second one sees no GPUs), so when torch is installed its copy is loaded first and
    libfdx.so binds to it; whichever of torch / libfdx is imported first, there is exactly one runtime."""
    import importlib.util
    try:
        spec = importlib.util.find_spec("torch")
    except (ImportError, ValueError):
        spec = None
    if spec is None or not spec.submodule_search_locations:
        return
    libdir = os.path.join(list(spec.submodule_search_locations)[0], "lib")
    for name in ("libhsa-runtime64.so", "libamdhip64.so"):
        path = os.path.join(libdir, name)
        if os.path.exists(path):
            try:
                ctypes.CDLL(path, mode=ctypes.RTLD_GLOBAL)
            except OSError:
                pass


def load():
    """Load libfdx.so once and attach the signatures.  Raises FdxError if it has not been built."""
    global _lib
    with _lock:
        if _lib is None:
            if not os.path.exists(LIB_PATH):
                raise FdxError(
                    f"{LIB_PATH} not found: build it with `make -C flashdeconv_amd/csrc` "
                    "(or `python -c 'import __graft_entry__ as g; g.build()'`). There is no CPU fallback.")
            _preload_shared_hip_runtime()
            lib = ctypes.CDLL(LIB_PATH)
            for name, (res, args) in SIGNATURES.items():
                fn = getattr(lib, name)
                fn.restype = res
                fn.argtypes = args
            _lib = lib
    return _lib


def env_reload():
    """libfdx caches its FDX_* switches when it first reads them: after changing os.environ call this (no-op before the library
    is loaded)."""
    if _lib is not None:
        _lib.fdx_env_reload()


def runtime_switches():
    """[(name, description)] of the library's runtime switches (csrc/fdx_env.cpp)."""
    lib, out, i = load(), [], 0
    while True:
        what = ctypes.c_char_p()
        name = lib.fdx_env_switch(i, ctypes.byref(what))
        if not name:
            return out
        out.append((name.decode(), (what.value or b"").decode()))
        i += 1


def check(rc):
    if rc != 0:
        msg = load().fdx_last_error()
        raise FdxError(f"libfdx error {rc}: {msg.decode() if msg else 'unknown'}")


def require_gpu():
    """Fail loudly when no HIP device is visible (the product path has no CPU fallback)."""
    lib = load()
    n = c_int(0)
    rc = lib.fdx_device_count(ctypes.byref(n))
    if rc != 0 or n.value <= 0:
        raise FdxError("flashdeconv_amd needs an AMD Instinct GPU (gfx950); no HIP device is visible "
                       "and there is no CPU fallback.")
    return n.value


def host_cpu_budget():
    """CPUs this process may keep busy: its affinity mask cut to the control group's CPU bandwidth quota (csrc/kdtree_order.cpp has
    the same rule: threads beyond the quota only get the whole process throttled)."""
    cpus = float(len(os.sched_getaffinity(0))) if hasattr(os, "sched_getaffinity") else float(os.cpu_count() or 1)
    try:
        with open("/sys/fs/cgroup/cpu.max") as f:
            q, period = f.read().split()[:2]
        if q != "max" and float(period) > 0:
            cpus = min(cpus, float(q) / float(period))
    except (OSError, ValueError):
        try:
            with open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us") as f:
                q = float(f.read())
            with open("/sys/fs/cgroup/cpu/cpu.cfs_period_us") as f:
                period = float(f.read())
            if q > 0 and period > 0:
                cpus = min(cpus, q / period)
        except (OSError, ValueError):
            pass
    return max(1, int(round(cpus)))


def tensor_to_host(t):
    """A CUDA torch tensor as a numpy array, copied through the library's pinned staging (fdx_memcpy_d2h): ``t.cpu()`` hands the
    driver pageable memory to pin, and unmapping that memory later stalls the process's GPU queues (csrc/pool.cpp: copy_d2h)."""
    import torch
    if not getattr(t, "is_cuda", False):
        return t.detach().cpu().numpy()
    t = t.detach().contiguous()
    out = np.empty(tuple(t.shape), dtype=np.dtype(str(t.dtype).replace("torch.", "")))
    if out.nbytes:
        st = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
        fn = load().fdx_download_dev if out.nbytes >= (32 << 20) else load().fdx_memcpy_d2h      # (large: threaded pinned ring)
        check(fn(out.ctypes.data_as(ctypes.c_void_p), ctypes.c_void_p(t.data_ptr()), out.nbytes, st))
    return out


def as_f64(a):
    return np.ascontiguousarray(a, dtype=np.float64)


def ptr_f64(a):
    return a.ctypes.data_as(p_double)


def ptr_i64(a):
    return a.ctypes.data_as(p_i64)


def ptr_i32(a):
    return a.ctypes.data_as(p_i32)


FDX_F32, FDX_F64 = 0, 1
PRE_RAW, PRE_LOG_CPM, PRE_LOG_CPM_SPARSE = 0, 1, 2
PRE_F64_MATH = 0x100     # float32 storage of values the reference would transform in float64 (integer input)


SRC_CODES = {"float32": 0, "float64": 1, "int8": 2, "uint8": 3, "bool": 3, "int16": 4, "uint16": 5, "int32": 6, "uint32": 7,
             "int64": 8, "uint64": 9}
_BIG = 32 << 20     # bytes from which a transfer goes through the threaded pinned ring (csrc/host_transfer.cpp)


def upload_bytes(dev_ptr, arr):
    """The bytes of a C-contiguous host array into HBM at dev_ptr; large arrays through the threaded pinned ring."""
    if arr.nbytes >= _BIG and arr.nbytes % 4 == 0:
        check(load().fdx_upload_convert_dev(dev_ptr, FDX_F32, arr.ctypes.data_as(c_void_p), SRC_CODES["float32"], arr.nbytes // 4, None, None))
    elif arr.nbytes:
        check(load().fdx_memcpy_h2d(dev_ptr, arr.ctypes.data_as(c_void_p), arr.nbytes, None))


def download_bytes(arr, dev_ptr):
    """HBM at dev_ptr into the C-contiguous host array `arr`; large arrays through the threaded pinned ring."""
    if arr.nbytes >= _BIG:
        check(load().fdx_download_dev(arr.ctypes.data_as(c_void_p), dev_ptr, arr.nbytes, None))
    elif arr.nbytes:
        check(load().fdx_memcpy_d2h(arr.ctypes.data_as(c_void_p), dev_ptr, arr.nbytes, None))


def upload_matrix(Y):
    """A dense host spot-by-gene matrix of any numeric dtype into HBM as the float32 / float64 matrix the kernels stream:
    (device pointer - return it with fdx_free -, dtype code).  float32 / float64 pass through; integer counts (numpy promotes
    them to float64 in the reference, core/deconv.py:190-191, 229) are narrowed to float32 WHILE they are staged when every
    value is exactly representable (below 2**24: checked on the fly, no pass over the matrix on one host thread first), to
    float64 otherwise.  Arithmetic on the device is float64 either way."""
    lib = load()
    Y = np.asarray(Y)
    name = Y.dtype.name
    if name not in SRC_CODES:                           # float16, longdouble, object ...: the reference's astype on the host
        Y, name = Y.astype(np.float64), "float64"
    if not Y.flags.c_contiguous:
        Y = np.ascontiguousarray(Y)
    count = int(Y.size)
    integer = Y.dtype.kind in "iub"
    order = [FDX_F64] if name == "float64" else [FDX_F32, FDX_F64] if integer and Y.dtype.itemsize > 2 else [FDX_F32]
    for code in order:
        ptr = c_void_p()
        check(lib.fdx_malloc(ctypes.byref(ptr), max(count * (4 if code == FDX_F32 else 8), 8)))
        try:
            mx = c_double(0.0)
            check(lib.fdx_upload_convert_dev(ptr, code, Y.ctypes.data_as(c_void_p), SRC_CODES[name], count, ctypes.byref(mx), None))
        except Exception:
            lib.fdx_free(ptr)
            raise
        if code == FDX_F32 and integer and mx.value >= float(1 << 24):
            lib.fdx_free(ptr)                           # counts float32 cannot hold: once more, as float64
            continue
        return ptr, code
    raise FdxError("upload_matrix: no destination type fits")


def as_device_matrix(Y):
    """Dense spot-by-gene matrix -> (C-contiguous float32/float64 array, dtype code).

    float32/float64 inputs are passed through; integer counts become float32 when every value is exactly
    representable (< 2**24), float64 otherwise.  Arithmetic on the device is float64 either way."""
    Y = np.asarray(Y)
    if Y.dtype == np.float32:
        return np.ascontiguousarray(Y), FDX_F32
    if Y.dtype == np.float64:
        return np.ascontiguousarray(Y), FDX_F64
    if np.issubdtype(Y.dtype, np.integer) or Y.dtype == np.bool_:
        if Y.size == 0 or (Y.dtype.itemsize <= 2) or (np.abs(Y).max() < (1 << 24)):
            return np.ascontiguousarray(Y, dtype=np.float32), FDX_F32
        return np.ascontiguousarray(Y, dtype=np.float64), FDX_F64
    return np.ascontiguousarray(Y, dtype=np.float64), FDX_F64


class Graph:
    """Owns an fdx_graph handle (device-resident sliced-ELL graph)."""

    def __init__(self, handle):
        self._h = c_void_p(handle)

    @classmethod
    def from_csr(cls, indptr, indices, n):
        lib = load()
        indptr = np.ascontiguousarray(indptr, dtype=np.int64)
        indices = np.ascontiguousarray(indices, dtype=np.int64)
        h = c_void_p()
        check(lib.fdx_graph_from_csr(ptr_i64(indptr), ptr_i64(indices), int(n), ctypes.byref(h)))
        return cls(h.value)

    @classmethod
    def from_coords_knn(cls, coords, k):
        coords = as_f64(coords)
        h = c_void_p()
        check(load().fdx_graph_build_knn(ptr_f64(coords), coords.shape[0], coords.shape[1], int(k), ctypes.byref(h)))
        return cls(h.value)

    @classmethod
    def from_coords_radius(cls, coords, radius):
        coords = as_f64(coords)
        h = c_void_p()
        check(load().fdx_graph_build_radius(ptr_f64(coords), coords.shape[0], coords.shape[1], float(radius), ctypes.byref(h)))
        return cls(h.value)

    def to_csr_arrays(self):
        """(indptr int64, indices int32) in the caller's spot order, indices ascending per row."""
        n, nnz, _ = self.info()
        indptr = np.zeros(n + 1, dtype=np.int64)
        indices = np.zeros(max(nnz, 1), dtype=np.int32)
        check(load().fdx_graph_export_csr(self._h, ptr_i64(indptr), ptr_i32(indices)))
        return indptr, indices[:nnz]

    @property
    def handle(self):
        return self._h

    def info(self):
        n, nnz, md = c_i64(0), c_i64(0), c_i32(0)
        check(load().fdx_graph_info(self._h, ctypes.byref(n), ctypes.byref(nnz), ctypes.byref(md)))
        return n.value, nnz.value, md.value

    def knn_ties(self):
        """Spots whose k-th and (k+1)-th nearest neighbours are exactly equidistant (0 unless built by k-NN)."""
        t = c_i64(0)
        check(load().fdx_graph_knn_ties(self._h, ctypes.byref(t)))
        return int(t.value)

    def knn_far(self):
        """1 when a k-NN walk of the rows this graph was built for left the 3 x 3 block of grid cells (spot shards: fdx.h)."""
        t = c_i32(0)
        check(load().fdx_graph_knn_far(self._h, ctypes.byref(t)))
        return int(t.value)

    def close(self):
        if self._h is not None and self._h.value:
            load().fdx_graph_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class CsrOnDevice:
    """A CSR spot matrix resident in HBM (int64 indptr, int32 column indices, f32/f64 values) and its fdx_csr_view.

    Built from a scipy.sparse matrix (uploaded: bytes proportional to the stored entries) or from a CUDA
    ``torch.sparse_csr`` tensor (zero-copy when its index dtypes already match).  Structure is validated on the device
    once (fdx_csr_check_dev) so that no kernel ever indexes with an out-of-range column."""

    def __init__(self):
        self._owned, self._keep = [], []
        self.view = CsrView()

    @staticmethod
    def _value_dtype(data):
        if data.dtype == np.float32:
            return np.float32
        if data.dtype == np.float64:
            return np.float64
        if data.dtype.kind in "iub" and (data.size == 0 or float(np.abs(data).max()) < 2.0 ** 24):
            return np.float32                       # counts: exact in float32
        return np.float64

    @classmethod
    def from_scipy(cls, Y, sort=True):
        lib = load()
        if Y.format != "csr":
            Y = Y.tocsr()
        if sort and not Y.has_sorted_indices:       # canonical rows let the statistics kernel read the indices once
            Y = Y.sorted_indices()
        n, G = Y.shape
        if G >= 2 ** 31:
            raise ValueError("CSR matrix has too many columns")
        indptr = np.ascontiguousarray(Y.indptr, dtype=np.int64)
        indices = np.ascontiguousarray(Y.indices, dtype=np.int32)
        self = cls()
        ptrs = []
        for arr in (indptr, indices):
            p = c_void_p()
            check(lib.fdx_malloc(ctypes.byref(p), max(arr.nbytes, 8)))
            self._owned.append(p)
            upload_bytes(p, arr)
            ptrs.append(p)
        # the stored values like a dense matrix's: any numeric dtype, integer counts narrowed to float32 while they are staged
        if Y.data.size:
            pv, vcode = upload_matrix(Y.data)
        else:
            pv, vcode = c_void_p(), FDX_F32
            check(lib.fdx_malloc(ctypes.byref(pv), 8))
        self._owned.append(pv)
        ptrs.append(pv)
        self._fill(ptrs[0].value, ptrs[1].value, ptrs[2].value, vcode, n,
                   int(indptr[-1]) if len(indptr) else 0, G, sorted_rows=1 if Y.has_sorted_indices else 0)
        return self

    # Structure checks of torch CSR tensors already seen: a second fit on the SAME tensor object does not scan its 10^9
    # indices again.  An entry is bound to the live tensor object (a weak reference; the entry goes when the tensor is
    # collected, so neither its id nor the addresses of its index tensors - which the sparse tensor keeps alive - can be
    # taken over by another matrix while the entry exists) and to torch's in-place version counters of the index tensors.
    # Writes that bypass torch (dlpack, raw pointers) are not seen: whoever does that owns the consequences.
    _checked = {}

    @classmethod
    def _checked_lookup(cls, Y, crow0, col0, shape_key):
        ent = cls._checked.get(id(Y))
        if ent is None or ent[0]() is not Y or ent[1] != (crow0._version, col0._version) + shape_key:
            return None
        return ent[2]

    @classmethod
    def _checked_store(cls, Y, crow0, col0, shape_key, sorted_rows):
        import weakref
        key = id(Y)
        try:
            ref = weakref.ref(Y, lambda _r, k=key: cls._checked.pop(k, None))
        except TypeError:
            return
        if len(cls._checked) >= 8:
            cls._checked.pop(next(iter(cls._checked)))
        cls._checked[key] = (ref, (crow0._version, col0._version) + shape_key, int(sorted_rows))

    @classmethod
    def from_torch(cls, Y):
        import torch
        n, G = Y.shape
        crow0, col0 = Y.crow_indices(), Y.col_indices()
        crow = crow0.to(torch.int64).contiguous()
        col = col0.to(torch.int32).contiguous()
        val = Y.values()
        if val.dtype not in (torch.float32, torch.float64):
            val = val.to(torch.float32)
        val = val.contiguous()
        self = cls()
        self._keep = [crow, col, val]
        shape_key = (int(n), int(G), int(val.numel()), crow.data_ptr(), col.data_ptr())
        zero_copy = crow.data_ptr() == crow0.data_ptr() and col.data_ptr() == col0.data_ptr()
        known = cls._checked_lookup(Y, crow0, col0, shape_key) if zero_copy else None
        self._fill(crow.data_ptr(), col.data_ptr(), val.data_ptr(), FDX_F32 if val.dtype == torch.float32 else FDX_F64,
                   n, int(val.numel()), G, known_sorted=known)
        if zero_copy and known is None:
            cls._checked_store(Y, crow0, col0, shape_key, self.view.sorted_rows)
        return self

    def _fill(self, indptr, indices, data, dtype, n, nnz, G, sorted_rows=1, known_sorted=None):
        v = self.view
        v.indptr, v.indices, v.data, v.dtype, v.n, v.nnz, v.G = indptr, indices, data, dtype, int(n), int(nnz), int(G)
        v.sorted_rows = int(sorted_rows)            # claim, verified on the device
        if known_sorted is not None:                # this very tensor object, unmodified, passed the check before
            v.sorted_rows = int(known_sorted)
            return
        try:
            try:
                check(load().fdx_csr_check_dev(ctypes.byref(v), None))
            except FdxError as e:
                if "sorted_rows" not in str(e):
                    raise
                v.sorted_rows = 0                   # e.g. a torch CSR tensor built with unsorted columns: full-scan kernels
                check(load().fdx_csr_check_dev(ctypes.byref(v), None))
        except Exception:
            self.free()
            raise

    def gene_moments(self, want_colsum=False):
        """(mean, var, colsum) per column: utils/genes.py:52-83; colsum (raw column sums, for "pearson") on request."""
        G = self.view.G
        mean, var = np.empty(G), np.empty(G)
        colsum = np.empty(G) if want_colsum else None
        check(load().fdx_csr_gene_moments_dev(ctypes.byref(self.view), ptr_f64(mean), ptr_f64(var),
                                              ptr_f64(colsum) if want_colsum else None, None))
        return mean, var, colsum

    def free(self):
        lib = load()
        for p in self._owned:
            if p is not None and p.value:
                lib.fdx_free(p)
        self._owned, self._keep = [], []

    def __del__(self):
        try:
            self.free()
        except Exception:
            pass


def is_torch_sparse_csr(x):
    try:
        import torch
    except Exception:
        return False
    return isinstance(x, torch.Tensor) and x.layout == torch.sparse_csr and x.is_cuda
