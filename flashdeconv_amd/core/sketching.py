"""CountSketch construction and projection: the reference's ``flashdeconv/core/sketching.py`` interface.

    build_countsketch_matrix <- core/sketching.py:18-84   (host: a G-entry table; hash/sign from numpy RandomState)
    project_to_sketch        <- core/sketching.py:160-206 (GPU: fdx_sketch)
    sketch_data              <- core/sketching.py:209-260

The sparse-Rademacher variant (core/sketching.py:87-157) is not reachable from ``FlashDeconv.fit`` and is not
provided; ``project_to_sketch`` itself accepts any sparse Omega.
"""
import numpy as np
from scipy import sparse

from .. import _lib
from ..utils.random import check_random_state


_TABLE_CACHE = {}       # (G, d, seed, leverage bytes) -> (bucket, weight); integer seeds only (a stateless draw)


def countsketch_tables(n_genes, sketch_dim, leverage_scores=None, random_state=None):
    """(bucket int64[G], weight float64[G]) with Omega[g, bucket[g]] = weight[g]  (core/sketching.py:48-82).
    Repeated calls with the same integer seed and leverage scores return the (read-only) tables of the first call."""
    key = None
    if isinstance(random_state, (int, np.integer)) and not isinstance(random_state, bool):
        lev_key = None if leverage_scores is None else np.ascontiguousarray(leverage_scores, dtype=np.float64).tobytes()
        key = (int(n_genes), int(sketch_dim), int(random_state), lev_key)
        hit = _TABLE_CACHE.get(key)
        if hit is not None:
            return hit
    bucket, weight = _countsketch_tables(n_genes, sketch_dim, leverage_scores, random_state)
    if key is not None:
        bucket.setflags(write=False)
        weight.setflags(write=False)
        if len(_TABLE_CACHE) >= 16:
            _TABLE_CACHE.pop(next(iter(_TABLE_CACHE)))
        _TABLE_CACHE[key] = (bucket, weight)
    return bucket, weight


def _countsketch_tables(n_genes, sketch_dim, leverage_scores, random_state):
    rng = check_random_state(random_state)
    if leverage_scores is None:
        prob = np.ones(n_genes) / n_genes
    else:
        prob = np.asarray(leverage_scores, dtype=np.float64)
        prob = prob / (np.sum(prob) + 1e-10)
    # the order of these two draws on one stream defines the hash: buckets first, then signs
    bucket = rng.randint(0, sketch_dim, size=n_genes)
    sign = rng.choice([-1, 1], size=n_genes)
    amp = np.clip(np.sqrt(prob * n_genes + 1e-10), 0.1, 10.0)
    val = sign * amp
    col_norm = np.sqrt(np.bincount(bucket, weights=val * val, minlength=sketch_dim))
    col_norm = np.maximum(col_norm, 1e-10)
    weight = val * (np.sqrt(n_genes / sketch_dim) / col_norm)[bucket]
    return bucket.astype(np.int64), weight


def build_countsketch_matrix(n_genes, sketch_dim, leverage_scores=None, random_state=None):
    bucket, weight = countsketch_tables(n_genes, sketch_dim, leverage_scores, random_state)
    return sparse.csr_matrix((weight, (np.arange(n_genes), bucket)), shape=(n_genes, sketch_dim), dtype=np.float64)


def _omega_csc(Omega, n_genes):
    Om = sparse.csc_matrix(Omega)
    if Om.shape[0] != n_genes:
        raise ValueError(f"Omega has {Om.shape[0]} rows but the data has {n_genes} genes")
    Om.sum_duplicates()
    Om.sort_indices()
    return (np.ascontiguousarray(Om.indptr, dtype=np.int64), np.ascontiguousarray(Om.indices, dtype=np.int32),
            np.ascontiguousarray(Om.data, dtype=np.float64), Om.shape[1])


def _project_dense(Y, col_ptr, gene_idx, weight, d, mode=_lib.PRE_RAW):
    Y, code = _lib.as_device_matrix(Y)
    n, G = Y.shape
    out = np.empty((n, d), dtype=np.float64)
    if n == 0:
        return out
    _lib.require_gpu()
    _lib.check(_lib.load().fdx_sketch(Y.ctypes.data, code, n, G, _lib.ptr_i64(col_ptr), _lib.ptr_i32(gene_idx),
                                      _lib.ptr_f64(weight), d, mode, _lib.ptr_f64(out)))
    return out


def project_to_sketch(Y_tilde, X_tilde, Omega):
    """Y_sketch = Y_tilde @ Omega, X_sketch = X_tilde @ Omega, both dense float64."""
    n_genes = Y_tilde.shape[1]
    col_ptr, gene_idx, weight, d = _omega_csc(Omega, n_genes)
    if sparse.issparse(Y_tilde):
        # reference core/sketching.py:194-199: Y_tilde @ Omega, densified.  The seam takes host arrays, so the row blocks are
        # densified on their way to the device kernel (the fit itself keeps CSR input sparse in HBM: csrc/csr_kernels.cpp).
        Yc = Y_tilde.tocsr()
        Ys = np.empty((Yc.shape[0], d), dtype=np.float64)
        step = max(1, (1 << 26) // max(n_genes, 1))
        for r0 in range(0, Yc.shape[0], step):
            Ys[r0:r0 + step] = _project_dense(np.asarray(Yc[r0:r0 + step].todense(), dtype=np.float64), col_ptr, gene_idx, weight, d)
    else:
        Ys = _project_dense(Y_tilde, col_ptr, gene_idx, weight, d)
    Xs = _project_dense(np.asarray(X_tilde, dtype=np.float64), col_ptr, gene_idx, weight, d)
    return Ys, Xs


def sketch_data(Y_tilde, X_tilde, sketch_dim=512, leverage_scores=None, method="countsketch", random_state=None):
    n_genes = Y_tilde.shape[1]
    if method == "countsketch":
        Omega = build_countsketch_matrix(n_genes, sketch_dim, leverage_scores, random_state)
    elif method == "rademacher":
        raise NotImplementedError("method='rademacher' is not on the FlashDeconv.fit path and is not provided")
    else:
        raise ValueError(f"Unknown sketching method: {method}")
    Ys, Xs = project_to_sketch(Y_tilde, X_tilde, Omega)
    return Ys, Xs, Omega
