"""Graph-Laplacian helpers with the reference's ``flashdeconv/core/spatial.py`` names.

``auto_tune_lambda`` (core/spatial.py:144-192) is a scalar formula on a K x K Gram matrix; the fit path evaluates it
inside fdx_fit_dev, this host version serves callers that hold host sketches.  ``compute_laplacian`` /
``compute_laplacian_quadratic`` (core/spatial.py:35-73, 118-141) are thin scipy expressions kept for API parity; the
solver's objective evaluates the Laplacian term on the device (csrc/finish_kernels.cpp).
"""
import numpy as np
from scipy import sparse


def compute_laplacian(A, normalized=False):
    A = sparse.csr_matrix(A)
    deg = np.asarray(A.sum(axis=1)).ravel()
    if not normalized:
        return (sparse.diags(deg) - A).tocsr()
    inv_sqrt = np.zeros_like(deg, dtype=np.float64)
    inv_sqrt[deg > 0] = 1.0 / np.sqrt(deg[deg > 0])
    D = sparse.diags(inv_sqrt)
    return (sparse.eye(A.shape[0]) - D @ A @ D).tocsr()


def compute_laplacian_quadratic(beta, L):
    return float(np.sum(beta * (L @ beta)))


def get_neighbor_indices(A):
    A = sparse.csr_matrix(A)
    return [A.indices[A.indptr[i]:A.indptr[i + 1]].copy() for i in range(A.shape[0])]


def get_neighbor_counts(A):
    return np.asarray(sparse.csr_matrix(A).sum(axis=1)).ravel().astype(np.int32)


def auto_tune_lambda(Y_sketch, X_sketch, A, alpha=0.005):
    X_sketch = np.asarray(X_sketch, dtype=np.float64)
    gram_diag = np.einsum("kd,kd->k", X_sketch, X_sketch)
    mean_deg = float(np.mean(np.asarray(A.sum(axis=1)).ravel())) if A.shape[0] else 0.0
    return float(alpha * gram_diag.mean() / max(mean_deg, 1.0))
