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


class FakeAnnData:
    """Duck-typed AnnData: exactly the attributes the reference's io/loader.py and tl/_deconvolve.py touch (SURVEY.md
    section 8c: .X, .layers, .obsm, .obs, .var_names, .obs_names, .n_obs, .uns, .copy()) - anndata itself is not installed."""

    def __init__(self, X, var_names, obs_names, obs=None, obsm=None, layers=None):
        import pandas as pd
        self.X, self.var_names, self.obs_names = X, np.array(var_names), np.array(obs_names)
        self.obs = pd.DataFrame(obs or {}, index=self.obs_names)
        self.obsm, self.layers, self.uns = dict(obsm or {}), dict(layers or {}), {}
        self.n_obs = X.shape[0]

    def copy(self):
        X = self.X.copy() if hasattr(self.X, "copy") else self.X.clone()       # numpy / scipy, or a torch tensor
        c = FakeAnnData(X, self.var_names, self.obs_names, obsm=dict(self.obsm), layers=dict(self.layers))
        c.obs = self.obs.copy()
        return c


def anndata_case(seed=11):
    """The AnnData-surface problem behind tests/golden/anndata_*.npz: 120 spots x 520 spatial genes, a single-cell
    reference of 73 cells x 500 genes in REVERSED gene order with a partial overlap (470 shared genes), one duplicated gene
    name on each side (first occurrence wins, io/loader.py:181-188), five cell types with 9 / 3 / 21 / 1 / 39 cells in
    shuffled order, ~60 % zeros in the cells.  Returns plain arrays; `anndata_objects` wraps them."""
    Y, X, coords, _ = count_like(120, 520, 5, 0.1, seed)
    rs = np.random.RandomState(seed + 1)
    genes_st = np.array([f"g{i:04d}" for i in range(520)])
    genes_ref = np.array([f"g{i:04d}" for i in range(50, 550)])[::-1].copy()
    genes_st[7] = genes_st[3]                                   # duplicate names: the first occurrence is the one used
    genes_ref[11] = genes_ref[470]
    labels = rs.permutation(np.repeat(["T cell", "B cell", "myeloid", "rare", "stroma"], [9, 3, 21, 1, 39]))
    names = np.array(["T cell", "B cell", "myeloid", "rare", "stroma"])
    kidx = np.array([int(np.where(names == s)[0][0]) for s in labels])
    # signature of reference gene j = signature of the spatial gene with the same (original) number, where there is one
    num = np.array([int(g[1:]) for g in np.array([f"g{i:04d}" for i in range(50, 550)])[::-1]])
    sig = np.where(num[None, :] < 520, X[:, np.minimum(num, 519)], 2.0)
    cells = rs.poisson(sig[kidx] * 3.0).astype(np.float64)
    cells[rs.rand(*cells.shape) < 0.6] = 0.0
    return dict(Y=Y.astype(np.float64), coords=coords, genes_st=genes_st, genes_ref=genes_ref, labels=labels, cells=cells,
                obs_st=np.array([f"spot{i}" for i in range(120)]), obs_ref=np.array([f"cell{i}" for i in range(73)]))


def anndata_objects(case, wrap=lambda a: a):
    """(adata_st, adata_ref) of `anndata_case`; `wrap` turns the two matrices into the container under test (identity,
    scipy CSR, CUDA tensor ...)."""
    st = FakeAnnData(wrap(case["Y"]), case["genes_st"], case["obs_st"], obsm={"spatial": case["coords"]})
    ref = FakeAnnData(wrap(case["cells"]), case["genes_ref"], case["obs_ref"], obs={"celltype": case["labels"]})
    return st, ref
