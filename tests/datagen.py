"""Seeded synthetic inputs for the parity tests and the golden-vector capture.

Everything here draws from legacy ``np.random.RandomState`` streams, which numpy
guarantees stable across versions, so the big cases (N=1000, G=2000) can be
regenerated from a seed instead of being committed; the golden files store the
SHA-256 of the regenerated inputs and the tests verify it before comparing.

Families:
  * ``gaussian_raw``  - "Gaussian/raw" inputs named by BASELINE.json configs:
    X ~ N(0,1), B ~ row-normalised U(0,1), Y = B X + 0.1 N(0,1), uniform coords.
    Used with preprocess="raw"; the solver converges in 5-7 iterations.
  * ``count_like``    - count data shaped like the reference's integration-test
    generator (reference tests/test_integration.py:10-84): log-normal signatures
    with 20 x5 markers per type, jittered grid coords, smooth true proportions,
    gamma depth, Poisson counts, drawn in the same RandomState call order.
  * ``sketched_problem`` - direct (Y_sketch, X_sketch, coords) problems shaped
    like reference tests/test_solver.py:66-89 and :298-310.
"""
import hashlib

import numpy as np


def sha256_arrays(*arrays):
    h = hashlib.sha256()
    for a in arrays:
        a = np.ascontiguousarray(a)
        h.update(str(a.dtype).encode())
        h.update(str(a.shape).encode())
        h.update(a.tobytes())
    return h.hexdigest()


def mix(B, X):
    """B @ X accumulated one cell type at a time with elementwise numpy ops.  BLAS matmul rounds differently on
    different CPUs; this form is bit-identical everywhere, so input SHA-256s stay valid on the GPU box."""
    out = np.zeros((B.shape[0], X.shape[1]))
    for k in range(B.shape[1]):
        out += B[:, k:k + 1] * X[k][None, :]
    return out


def gaussian_raw(n_spots, n_genes, n_types, seed=0, noise=0.1):
    rs = np.random.RandomState(seed)
    X = rs.randn(n_types, n_genes)
    B = rs.rand(n_spots, n_types)
    B /= B.sum(axis=1, keepdims=True)
    Y = mix(B, X) + noise * rs.randn(n_spots, n_genes)
    coords = rs.rand(n_spots, 2) * np.sqrt(n_spots)
    return Y, X, coords, B


def count_like(n_spots=100, n_genes=500, n_types=5, noise_level=0.1, seed=42):
    rs = np.random.RandomState(seed)
    X = np.exp(rs.randn(n_types, n_genes) * 0.5 + 1)
    for k in range(n_types):
        up = rs.choice(n_genes, size=20, replace=False)
        X[k, up] *= 5
    side = int(np.ceil(np.sqrt(n_spots)))
    gx = np.tile(np.arange(side), side)[:n_spots]
    gy = np.repeat(np.arange(side), side)[:n_spots]
    coords = np.column_stack([gx, gy]).astype(float)
    coords += rs.randn(n_spots, 2) * 0.1
    B = np.zeros((n_spots, n_types))
    for k in range(n_types):
        centre = rs.rand(2) * side
        dist = np.sqrt(np.sum((coords - centre) ** 2, axis=1))
        B[:, k] = np.exp(-dist / (side / 2))
    B = B / B.sum(axis=1, keepdims=True)
    expected = mix(B, X)
    depth = rs.gamma(shape=5, scale=1000, size=n_spots)
    expected = expected * depth[:, np.newaxis]
    Y = rs.poisson(expected * (1 + noise_level * rs.rand(*expected.shape)))
    return Y, X, coords, B


def sketched_problem(n_spots, n_types, sketch_dim, seed=42, noise=0.1):
    rs = np.random.RandomState(seed)
    Xs = rs.randn(n_types, sketch_dim)
    B = rs.rand(n_spots, n_types)
    B = B / B.sum(axis=1, keepdims=True)
    Ys = mix(B, Xs) + noise * rs.randn(n_spots, sketch_dim)
    coords = rs.rand(n_spots, 2)
    return Ys, Xs, coords, B
