"""Host side of the BCD solver: same names and argument meaning as the reference's
``flashdeconv/core/solver.py``; the arithmetic runs in libfdx.so on the GPU.

    soft_threshold          <- core/solver.py:18-26
    precompute_gram_matrix  <- core/solver.py:187-201   (fdx_gram_xty)
    precompute_XtY          <- core/solver.py:204-223   (fdx_gram_xty)
    compute_objective       <- core/solver.py:226-284   (fdx_objective)
    bcd_solve               <- core/solver.py:287-428   (fdx_bcd_solve)
    normalize_proportions   <- core/solver.py:431-452
"""
import ctypes

import numpy as np
from scipy import sparse

from .. import _lib


def soft_threshold(x, threshold):
    """sign(x) * max(|x| - threshold, 0) (core/solver.py:18-26).  A scalar helper of the reference's kernels, kept for its
    tests; on the device it is fused into the sweep (csrc/bcd_sweep_inst.cpp)."""
    x = np.asarray(x, dtype=np.float64)
    out = np.sign(x) * np.maximum(np.abs(x) - threshold, 0.0)
    return float(out) if out.ndim == 0 else out


def precompute_gram_matrix(X_sketch):
    """XtX = X_sketch X_sketch^T, (K, K) (core/solver.py:187-201)."""
    X_sketch = _lib.as_f64(X_sketch)
    K, d = X_sketch.shape
    _lib.require_gpu()
    out = np.empty((K, K), dtype=np.float64)
    _lib.check(_lib.load().fdx_gram_xty(_lib.ptr_f64(X_sketch), None, 0, d, K, _lib.ptr_f64(out), None))
    return out


def precompute_XtY(X_sketch, Y_sketch):
    """H = X_sketch Y_sketch^T, (K, N) C-order as in the reference (core/solver.py:204-223)."""
    X_sketch, Y_sketch = _lib.as_f64(X_sketch), _lib.as_f64(Y_sketch)
    K, d = X_sketch.shape
    n = Y_sketch.shape[0]
    if Y_sketch.shape[1] != d:
        raise ValueError(f"Sketch dimension mismatch: Y_sketch has {Y_sketch.shape[1]}, X_sketch has {d}")
    out = np.empty((K, n), dtype=np.float64)
    if n:
        _lib.require_gpu()
        _lib.check(_lib.load().fdx_gram_xty(_lib.ptr_f64(X_sketch), _lib.ptr_f64(Y_sketch), n, d, K, None, _lib.ptr_f64(out)))
    return out


def compute_objective(beta, H, XtX, YtY, L, lambda_, rho):
    """0.5 (YtY - 2 sum(beta * H^T) + sum(beta^T beta * XtX)) + 0.5 lambda sum(beta * (L beta)) + rho sum|beta|
    (core/solver.py:226-284).  L is the un-normalised Laplacian D - A (core/spatial.py:70-73); its off-diagonal structure
    is the graph the device kernel walks."""
    beta, H, XtX = _lib.as_f64(beta), _lib.as_f64(H), _lib.as_f64(XtX)
    n, K = beta.shape
    L = sparse.csr_matrix(L)
    A = L.copy().tolil()
    A.setdiag(0)
    A = A.tocsr()
    A.eliminate_zeros()
    _lib.require_gpu()
    graph = _lib.Graph.from_csr(A.indptr, A.indices, n)
    try:
        out = ctypes.c_double(0.0)
        _lib.check(_lib.load().fdx_objective(graph.handle, _lib.ptr_f64(beta), _lib.ptr_f64(H), _lib.ptr_f64(XtX), n, K,
                                             float(YtY), float(lambda_), float(rho), ctypes.byref(out)))
    finally:
        graph.close()
    return float(out.value)


def _graph_from_adjacency(A, n_spots):
    """The sweep uses the CSR *structure* of A only (reference: core/solver.py:157-159, 363-365)."""
    if isinstance(A, _lib.Graph):
        return A, False
    if not sparse.issparse(A):
        A = sparse.csr_matrix(np.asarray(A))
    A = A.tocsr()
    if A.shape[0] != n_spots or A.shape[1] != n_spots:
        raise ValueError(f"Adjacency shape {A.shape} does not match n_spots={n_spots}")
    return _lib.Graph.from_csr(A.indptr, A.indices, n_spots), True


def bcd_solve(Y_sketch, X_sketch, A, lambda_=0.1, rho=0.01, max_iter=100, tol=1e-4, verbose=False):
    """Solve  min 0.5||Y_s - beta X_s||_F^2 + 0.5 lambda Tr(beta^T L beta) + rho ||beta||_1, beta >= 0.

    Drop-in for the reference ``bcd_solve`` (core/solver.py:287-428): returns ``(beta, info)`` with
    ``info = {converged, n_iterations, final_objective, objectives, final_change}``.
    """
    Y_sketch = _lib.as_f64(Y_sketch)
    X_sketch = _lib.as_f64(X_sketch)
    n_spots = Y_sketch.shape[0]
    n_types = X_sketch.shape[0]
    if n_spots == 0 or n_types == 0:                                    # core/solver.py:334-343
        return (np.empty((n_spots, n_types), dtype=np.float64),
                {"converged": True, "n_iterations": 0, "final_objective": 0.0, "objectives": [], "final_change": 0.0})
    if Y_sketch.shape[1] != X_sketch.shape[1]:
        raise ValueError(f"Sketch dimension mismatch: Y_sketch has {Y_sketch.shape[1]}, X_sketch has {X_sketch.shape[1]}")
    _lib.require_gpu()
    lib = _lib.load()
    graph, owned = _graph_from_adjacency(A, n_spots)
    try:
        beta = np.empty((n_spots, n_types), dtype=np.float64)
        objs = np.zeros(max(int(max_iter), 1), dtype=np.float64)
        rels = np.zeros(max(int(max_iter), 1), dtype=np.float64)
        info = _lib.SolveInfo()
        _lib.check(lib.fdx_bcd_solve(graph.handle, _lib.ptr_f64(Y_sketch), _lib.ptr_f64(X_sketch), n_spots,
                                     Y_sketch.shape[1], n_types, float(lambda_), float(rho), int(max_iter), float(tol),
                                     1 if verbose else 0, _lib.ptr_f64(beta), _lib.ptr_f64(objs), _lib.ptr_f64(rels),
                                     ctypes.byref(info)))
    finally:
        if owned:
            graph.close()
    n_it = int(info.n_iterations)
    objectives = [float(v) for v in objs[:info.n_objectives]]
    if verbose:                                                         # core/solver.py:399-404, 411-412
        its = [t for t in range(n_it) if t % 10 == 0 or t == max_iter - 1]
        for t, obj in zip(its, objectives):
            print(f"Iteration {t}: objective = {obj:.6f}, rel_change = {rels[t]:.6e}")
        if info.converged:
            print(f"Converged at iteration {n_it - 1}")
    out = {
        "converged": bool(info.converged),
        "n_iterations": n_it,
        "final_objective": float(info.final_objective),
        "objectives": objectives if verbose else [],
        "final_change": float(info.final_change),
    }
    return beta, out


def normalize_proportions(beta):
    """Row-normalise abundances; all-zero rows become uniform (core/solver.py:431-452).

    Host helper for arrays already on the host (the fit path normalises on the device)."""
    beta = np.asarray(beta, dtype=np.float64)
    s = beta.sum(axis=1, keepdims=True)
    zero = (s == 0).ravel()
    out = beta / np.maximum(s, 1e-10)
    if zero.any():
        out[zero] = 1.0 / beta.shape[1]
    return out
