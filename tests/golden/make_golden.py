#!/usr/bin/env python3
"""Capture golden vectors from the reference implementation (this container only).

The reference (``/root/reference``, pure Python) cannot travel to the GPU box and
imports ``numba``, which is absent here.  Its three ``@jit`` functions are plain
Python compiled without fastmath, so running them un-jitted has the same float64
semantics.  This script therefore puts a no-op stand-in module
named ``numba`` on ``sys.path`` (``jit`` = identity decorator, ``prange`` =
``range``) in a temporary directory, imports the reference, runs it on the
seeded inputs of ``tests/datagen.py`` and stores inputs' SHA-256 + outputs as
small ``.npz`` files next to this script.  Only data (arrays) is stored.

Run:  python tests/golden/make_golden.py        (takes a few minutes; pure-Python BCD)
"""
import os
import sys
import tempfile
import time

import numpy as np
from scipy import sparse

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
import datagen  # noqa: E402

REF = os.environ.get("FDX_REFERENCE", "/root/reference")


def _import_reference():
    shim_dir = tempfile.mkdtemp(prefix="numba_standin_")
    with open(os.path.join(shim_dir, "numba.py"), "w") as f:
        f.write(
            "def jit(*a, **k):\n"
            "    if len(a) == 1 and callable(a[0]) and not k:\n"
            "        return a[0]\n"
            "    return lambda fn: fn\n"
            "njit = jit\n"
            "prange = range\n"
        )
    sys.path.insert(0, shim_dir)
    sys.path.insert(0, REF)
    import flashdeconv  # noqa: F401
    return flashdeconv


fd = _import_reference()
from flashdeconv import FlashDeconv  # noqa: E402
from flashdeconv.core.sketching import build_countsketch_matrix, sketch_data  # noqa: E402
from flashdeconv.core.solver import (  # noqa: E402
    bcd_solve, compute_objective, normalize_proportions, precompute_gram_matrix, precompute_XtY,
)
from flashdeconv.core.spatial import auto_tune_lambda, compute_laplacian  # noqa: E402
from flashdeconv.utils.genes import (  # noqa: E402
    compute_leverage_scores, select_hvg, select_informative_genes, select_markers,
)
from flashdeconv.utils.graph import (  # noqa: E402
    build_grid_graph, build_knn_graph, build_radius_graph, coords_to_adjacency,
)


def save(name, **arrays):
    path = os.path.join(HERE, name)
    np.savez_compressed(path, **arrays)
    print(f"  wrote {name}  ({os.path.getsize(path) / 1024:.1f} KiB)")


def info_arrays(prefix, info):
    return {
        f"{prefix}converged": np.array(bool(info["converged"])),
        f"{prefix}n_iterations": np.array(int(info["n_iterations"])),
        f"{prefix}final_objective": np.array(float(info["final_objective"])),
        f"{prefix}final_change": np.array(float(info["final_change"])),
        f"{prefix}objectives": np.array(info["objectives"], dtype=np.float64),
    }


# --------------------------------------------------------------------------- G7
def golden_omega():
    print("G7: CountSketch hash/sign tables")
    out = {}
    cases = [(2000, 512, 0), (5000, 1024, 0), (2000, 512, 123), (100, 32, 42), (2000, 500, 0), (37, 1, 7), (64, 3, 5)]
    out["cases"] = np.array(cases, dtype=np.int64)
    for (G, d, seed) in cases:
        Om = build_countsketch_matrix(G, d, leverage_scores=None, random_state=seed).tocsr()
        assert np.all(np.diff(Om.indptr) == 1)
        tag = f"G{G}_d{d}_s{seed}"
        out[f"{tag}_bucket"] = Om.indices.astype(np.int64)
        out[f"{tag}_sign"] = np.sign(Om.data).astype(np.int64)
        out[f"{tag}_data"] = Om.data.astype(np.float64)
    # leverage-weighted table (amplitude path, core/sketching.py:51-82)
    rs = np.random.RandomState(3)
    lev = rs.rand(300) ** 4
    lev[:5] = 0.0
    lev[5:8] = 50.0
    Om = build_countsketch_matrix(300, 64, leverage_scores=lev, random_state=11).tocsr()
    out["lev_input"] = lev
    out["lev_bucket"] = Om.indices.astype(np.int64)
    out["lev_data"] = Om.data.astype(np.float64)
    save("omega_tables.npz", **out)


# ------------------------------------------------------------------- leverage
def golden_leverage():
    print("leverage scores")
    out = {}
    rs = np.random.RandomState(5)
    mats = {
        "randn_7x120": rs.randn(7, 120),
        "lognorm_30x400": np.exp(rs.randn(30, 400) * 0.5 + 1),
        "tiny_2x9": rs.rand(2, 9),
        "one_type_1x15": rs.rand(1, 15),
        "wide_K_gt_G_12x5": rs.randn(12, 5),
        "scaled_small_10x200": rs.rand(10, 200) * 1e-3,
    }
    X = np.exp(rs.randn(6, 50))
    X[3] = X[1] * 2.0  # rank-deficient beyond the centring
    mats["rankdef_6x50"] = X
    out["names"] = np.array(list(mats.keys()))
    for name, X in mats.items():
        out[f"{name}_X"] = X
        out[f"{name}_lev"] = compute_leverage_scores(X)
    save("leverage.npz", **out)


# --------------------------------------------------------------------- graphs
def csr_parts(A):
    A = A.tocsr()
    A.sort_indices()
    return A.indptr.astype(np.int64), A.indices.astype(np.int64), A.data.astype(np.float64)


def golden_graphs():
    print("graphs")
    out = {}
    rs = np.random.RandomState(9)
    sets = {
        "u2d_2000_k6": (rs.rand(2000, 2) * np.sqrt(2000), "knn", 6, None),
        "u2d_300_k4": (rs.rand(300, 2), "knn", 4, None),
        "u2d_50_k1": (rs.rand(50, 2), "knn", 1, None),
        "u2d_5_k6": (rs.rand(5, 2), "knn", 6, None),          # k clamped to N-1
        "u2d_1_k6": (rs.rand(1, 2), "knn", 6, None),          # single spot
        "u2d_40_k0": (rs.rand(40, 2), "knn", 0, None),        # no neighbours
        "clustered_800_k6": (np.concatenate([rs.randn(400, 2) * 0.05, rs.randn(400, 2) * 3 + 10]), "knn", 6, None),
        "aniso_600_k6": (rs.rand(600, 2) * np.array([1000.0, 1.0]), "knn", 6, None),
        "u3d_500_k6": (rs.rand(500, 3) * 8, "knn", 6, None),
        "u1d_100_k2": (rs.rand(100, 1) * 50, "knn", 2, None),
        "jgrid_400_k6": (datagen.count_like(400, 30, 2, seed=1)[2], "knn", 6, None),
        "u2d_700_r": (rs.rand(700, 2) * 20, "radius", 6, 1.3),
        "u2d_100_r_none": (rs.rand(100, 2) * 100, "radius", 6, 0.01),   # no pairs
        "jgrid_400_grid": (datagen.count_like(400, 30, 2, seed=2)[2], "grid", 6, None),
        "u3d_300_r": (rs.rand(300, 3) * 6, "radius", 6, 1.1),
    }
    out["names"] = np.array(list(sets.keys()))
    for name, (coords, method, k, radius) in sets.items():
        A = coords_to_adjacency(coords, method=method, k=k, radius=radius)
        ip, ix, dat = csr_parts(A)
        assert np.all(dat == 1.0) or len(dat) == 0
        out[f"{name}_coords"] = coords
        out[f"{name}_method"] = np.array(method)
        out[f"{name}_k"] = np.array(k)
        out[f"{name}_radius"] = np.array(-1.0 if radius is None else radius)
        out[f"{name}_indptr"] = ip
        out[f"{name}_indices"] = ix
    save("graphs.npz", **out)


# --------------------------------------------------------------- solver (G1)
def golden_solver():
    print("G1: direct bcd_solve problems")
    out = {}
    probs = {
        # reference tests/test_solver.py:298-313 (determinism problem)
        "det60": dict(n=60, K=7, d=48, seed=42, noise=0.05, k=4, lam=0.1, rho=0.01, max_iter=30, tol=1e-6),
        # reference tests/test_solver.py:66-89 (simple_problem)
        "simple50": dict(n=50, K=5, d=32, seed=42, noise=0.1, k=4, lam=0.1, rho=0.01, max_iter=100, tol=1e-4),
        "k30": dict(n=200, K=30, d=64, seed=7, noise=0.1, k=6, lam=0.3, rho=0.02, max_iter=40, tol=1e-5),
        "lam0": dict(n=80, K=6, d=40, seed=3, noise=0.1, k=5, lam=0.0, rho=0.0, max_iter=25, tol=1e-7),
        "bigrho": dict(n=70, K=9, d=40, seed=4, noise=0.1, k=3, lam=0.05, rho=0.6, max_iter=25, tol=1e-7),
        "k1": dict(n=40, K=1, d=16, seed=8, noise=0.1, k=3, lam=0.1, rho=0.01, max_iter=20, tol=1e-8),
        "k33": dict(n=90, K=33, d=48, seed=9, noise=0.1, k=6, lam=0.2, rho=0.01, max_iter=12, tol=1e-9),
    }
    out["names"] = np.array(list(probs.keys()))
    for name, p in probs.items():
        Ys, Xs, coords, _ = datagen.sketched_problem(p["n"], p["K"], p["d"], seed=p["seed"], noise=p["noise"])
        A = build_knn_graph(coords, k=p["k"])
        ip, ix, _ = csr_parts(A)
        out[f"{name}_Ys"] = Ys
        out[f"{name}_Xs"] = Xs
        out[f"{name}_coords"] = coords
        out[f"{name}_indptr"] = ip
        out[f"{name}_indices"] = ix
        out[f"{name}_params"] = np.array([p["lam"], p["rho"], p["max_iter"], p["tol"], p["k"]], dtype=np.float64)
        for t in (0, 1, 2, 10):
            b, inf = bcd_solve(Ys, Xs, A, lambda_=p["lam"], rho=p["rho"], max_iter=t, tol=1e-30)
            out[f"{name}_beta_it{t}"] = b
            out.update(info_arrays(f"{name}_it{t}_", inf))
        b, inf = bcd_solve(Ys, Xs, A, lambda_=p["lam"], rho=p["rho"], max_iter=p["max_iter"], tol=p["tol"])
        out[f"{name}_beta"] = b
        out.update(info_arrays(f"{name}_", inf))
        out[f"{name}_props"] = normalize_proportions(b)
        # verbose objective trace (core/solver.py:399-404)
        import contextlib, io
        with contextlib.redirect_stdout(io.StringIO()):
            b2, inf2 = bcd_solve(Ys, Xs, A, lambda_=p["lam"], rho=p["rho"], max_iter=p["max_iter"], tol=p["tol"], verbose=True)
        assert np.array_equal(b, b2)
        out[f"{name}_verbose_objectives"] = np.array(inf2["objectives"])
        out[f"{name}_XtX"] = precompute_gram_matrix(Xs)
        out[f"{name}_H"] = precompute_XtY(Xs, Ys)
    # empty graph / isolated spots: adjacency with empty rows
    Ys, Xs, coords, _ = datagen.sketched_problem(30, 4, 16, seed=12)
    A = sparse.csr_matrix((30, 30))
    b, inf = bcd_solve(Ys, Xs, A, lambda_=0.2, rho=0.01, max_iter=15, tol=1e-9)
    out["iso_Ys"], out["iso_Xs"], out["iso_beta"] = Ys, Xs, b
    out.update(info_arrays("iso_", inf))
    # normalize_proportions edge rows (core/solver.py:431-452)
    bn = np.array([[1.0, 3.0, 0.0], [0.0, 0.0, 0.0], [2.0, 2.0, 4.0], [1e-12, 0.0, 0.0], [5e-11, 0.0, 5e-11]])
    out["norm_in"], out["norm_out"] = bn, normalize_proportions(bn)
    save("solver_small.npz", **out)


# ------------------------------------------------------- full fits (G2..G6)
def run_fit_stages(Y, X, coords, *, sketch_dim, preprocess, n_hvg, n_markers=50, lambda_spatial="auto", rho=0.01,
                   k=6, max_iter=100, tol=1e-4, seed=0, snapshots=(1, 2, 10), keep_sketch_rows=8):
    """Run the reference stage by stage exactly as FlashDeconv.fit does (core/deconv.py:305-398)."""
    out = {}
    model = FlashDeconv(sketch_dim=sketch_dim, lambda_spatial=lambda_spatial, rho_sparsity=rho, n_hvg=n_hvg,
                        n_markers_per_type=n_markers, k_neighbors=k, max_iter=max_iter, tol=tol,
                        preprocess=preprocess, random_state=seed)
    t0 = time.time()
    with np.errstate(all="ignore"):
        props = model.fit_transform(Y, X, coords)
    out["fit_seconds"] = np.array(time.time() - t0)
    out["gene_idx"] = model.gene_idx_.astype(np.int64)
    out["lambda_used"] = np.array(model.lambda_used_)
    out["beta"] = model.beta_
    out["proportions"] = props
    out.update(info_arrays("", model.info_))
    ip, ix, _ = csr_parts(model.adjacency_)
    out["indptr"], out["indices"] = ip, ix
    # stage outputs
    with np.errstate(all="ignore"):
        gene_idx, lev = select_informative_genes(Y, X, n_hvg=n_hvg, n_markers_per_type=n_markers)
    assert np.array_equal(gene_idx, model.gene_idx_)
    out["leverage"] = lev
    Ysub = Y[:, gene_idx]
    if sparse.issparse(Ysub):
        Ysub = Ysub.tocsr()
    Yt, Xt = model._preprocess_data(Ysub, X[:, gene_idx], preprocess)
    Ysk, Xsk, Om = sketch_data(Yt, Xt, sketch_dim=sketch_dim, leverage_scores=lev, random_state=seed)
    Om = Om.tocsr()
    out["omega_bucket"] = Om.indices.astype(np.int64)
    out["omega_data"] = Om.data
    out["X_sketch"] = Xsk
    out["Y_sketch_head"] = Ysk[:keep_sketch_rows]
    out["Y_sketch_rowsum"] = Ysk.sum(axis=1)
    out["Y_sketch_rowsq"] = (Ysk ** 2).sum(axis=1)
    out["YtY"] = np.array(float(np.sum(Ysk ** 2)))
    XtX = precompute_gram_matrix(Xsk)
    H = precompute_XtY(Xsk, Ysk)
    out["XtX"] = XtX
    out["H"] = H
    A = model.adjacency_
    lam = auto_tune_lambda(Ysk, Xsk, A) if lambda_spatial == "auto" else float(lambda_spatial)
    assert lam == model.lambda_used_
    for t in snapshots:
        if t <= max_iter:
            b, _ = bcd_solve(Ysk, Xsk, A, lambda_=lam, rho=rho, max_iter=t, tol=1e-30)
            out[f"beta_it{t}"] = b
    return out


def golden_fits():
    # G2: the reference's own integration fixture 100x500x5 (tests/test_integration.py:92-100)
    print("G2: count-like 100x500x5 (dense, CSR; d=64/128)")
    Y, X, coords, B = datagen.count_like(100, 500, 5, 0.1, 42)
    for d in (64, 128):
        o = run_fit_stages(Y, X, coords, sketch_dim=d, preprocess="log_cpm", n_hvg=2000)
        o.update(Y=Y.astype(np.int32), X=X, coords=coords, B=B)
        save(f"fit_counts_100x500x5_d{d}.npz", **o)
    o = run_fit_stages(sparse.csr_matrix(Y.astype(np.float64)), X, coords, sketch_dim=64, preprocess="log_cpm", n_hvg=2000)
    o.update(Y=Y.astype(np.int32), X=X, coords=coords)
    save("fit_counts_100x500x5_d64_csr.npz", **o)
    # float32 dense input (numpy keeps log-CPM in float32: core/deconv.py:190-191)
    o = run_fit_stages(Y.astype(np.float32), X, coords, sketch_dim=64, preprocess="log_cpm", n_hvg=2000)
    o.update(Y=Y.astype(np.int32), X=X, coords=coords)
    save("fit_counts_100x500x5_d64_f32.npz", **o)

    # gene selection active (G > n_hvg): exercises select_hvg / select_markers
    print("gene selection: count-like 150x600x4, n_hvg=200, n_markers=10")
    Y, X, coords, B = datagen.count_like(150, 600, 4, 0.1, 5)
    o = run_fit_stages(Y, X, coords, sketch_dim=64, preprocess="log_cpm", n_hvg=200, n_markers=10, max_iter=30)
    with np.errstate(all="ignore"):
        o["hvg_idx"] = select_hvg(Y, n_top=200).astype(np.int64)
        o["hvg_idx_csr"] = select_hvg(sparse.csr_matrix(Y.astype(np.float64)), n_top=200).astype(np.int64)
        mi, ma = select_markers(X, n_markers=10)
    o["marker_idx"] = mi.astype(np.int64)
    o["marker_assign"] = ma.astype(np.int64)
    o.update(Y=Y.astype(np.int32), X=X, coords=coords)
    save("fit_genesel_150x600x4.npz", **o)

    # G6: pearson preprocess, dense + CSR
    print("G6: pearson 120x300x4")
    Y, X, coords, B = datagen.count_like(120, 300, 4, 0.1, 6)
    o = run_fit_stages(Y, X, coords, sketch_dim=48, preprocess="pearson", n_hvg=2000, max_iter=40)
    o.update(Y=Y.astype(np.int32), X=X, coords=coords)
    save("fit_pearson_120x300x4.npz", **o)
    o = run_fit_stages(sparse.csr_matrix(Y.astype(np.float64)), X, coords, sketch_dim=48, preprocess="pearson", n_hvg=2000, max_iter=40)
    o.update(Y=Y.astype(np.int32), X=X, coords=coords)
    save("fit_pearson_120x300x4_csr.npz", **o)

    # G3: Gaussian/raw miniature of config 1 (inputs regenerated from seed in the tests)
    print("G3: gaussian/raw 1000x2000x10 d=512")
    Y, X, coords, B = datagen.gaussian_raw(1000, 2000, 10, seed=0)
    o = run_fit_stages(Y, X, coords, sketch_dim=512, preprocess="raw", n_hvg=2000)
    o["input_sha256"] = np.array(datagen.sha256_arrays(Y, X, coords))
    o["gen"] = np.array([1000, 2000, 10, 0])
    save("fit_gauss_1000x2000x10.npz", **o)

    # G4: count-like log_cpm, runs the full 100 iterations without converging
    print("G4: count-like/log_cpm 1000x2000x10 d=512 (100 pure-Python iterations)")
    Y, X, coords, B = datagen.count_like(1000, 2000, 10, 0.1, 0)
    o = run_fit_stages(Y, X, coords, sketch_dim=512, preprocess="log_cpm", n_hvg=2000, snapshots=(1, 2, 10, 99))
    o["input_sha256"] = np.array(datagen.sha256_arrays(Y, X, coords))
    o["gen"] = np.array([1000, 2000, 10, 0])
    save("fit_counts_1000x2000x10.npz", **o)

    # G5: K=30
    print("G5: count-like/log_cpm 600x1000x30 d=256")
    Y, X, coords, B = datagen.count_like(600, 1000, 30, 0.1, 3)
    o = run_fit_stages(Y, X, coords, sketch_dim=256, preprocess="log_cpm", n_hvg=2000, max_iter=60)
    o["input_sha256"] = np.array(datagen.sha256_arrays(Y, X, coords))
    o["gen"] = np.array([600, 1000, 30, 3])
    save("fit_counts_600x1000x30.npz", **o)

    # raw + fixed lambda + K=20 gaussian (config-2-shaped miniature)
    print("gaussian/raw 800x2000x20 d=512, lambda fixed")
    Y, X, coords, B = datagen.gaussian_raw(800, 2000, 20, seed=1)
    o = run_fit_stages(Y, X, coords, sketch_dim=512, preprocess="raw", n_hvg=2000, lambda_spatial=0.5, rho=0.02)
    o["input_sha256"] = np.array(datagen.sha256_arrays(Y, X, coords))
    o["gen"] = np.array([800, 2000, 20, 1])
    save("fit_gauss_800x2000x20.npz", **o)


def golden_fits_sparse():
    """CSR input with gene selection active: the sparse branches of select_hvg (utils/genes.py:52-83), of log-CPM
    (core/deconv.py:181-188) and of project_to_sketch (core/sketching.py:194-199) in one fit.  ~94 % zeros."""
    print("sparse: count-like 300x900x5 CSR, n_hvg=250, n_markers=10, log_cpm and pearson")
    Y, X, coords, B = datagen.count_like(300, 900, 5, 0.1, 9)
    rs = np.random.RandomState(9)
    Y = (Y * (rs.rand(*Y.shape) < 0.08)).astype(np.int32)          # thin the counts to a realistically sparse matrix
    Y[7] = 0                                                        # an empty spot: library size 0 -> 1
    Ys = sparse.csr_matrix(Y.astype(np.float64))
    for pre in ("log_cpm", "pearson", "raw"):
        o = run_fit_stages(Ys, X, coords, sketch_dim=64, preprocess=pre, n_hvg=250, n_markers=10, max_iter=30)
        o.update(Y=Y, X=X, coords=coords)
        save(f"fit_sparse_{pre}_300x900x5.npz", **o)


def golden_objective():
    print("objective identity cases")
    out = {}
    rs = np.random.RandomState(21)
    n, K, d = 64, 6, 24
    Ys, Xs, coords, _ = datagen.sketched_problem(n, K, d, seed=21)
    A = build_knn_graph(coords, k=5)
    L = compute_laplacian(A)
    beta = np.abs(rs.randn(n, K))
    XtX = precompute_gram_matrix(Xs)
    H = precompute_XtY(Xs, Ys)
    YtY = float(np.sum(Ys ** 2))
    ip, ix, _ = csr_parts(A)
    out.update(Ys=Ys, Xs=Xs, beta=beta, indptr=ip, indices=ix)
    out["obj"] = np.array([compute_objective(beta, H, XtX, YtY, L, lam, rho) for lam, rho in [(0.0, 0.0), (0.3, 0.0), (0.0, 2.5), (0.7, 1.1)]])
    out["lam_rho"] = np.array([(0.0, 0.0), (0.3, 0.0), (0.0, 2.5), (0.7, 1.1)])
    save("objective.npz", **out)


def golden_lattice():
    """Regular lattices (Visium-HD / Stereo-seq bins are a square lattice, Visium a hexagonal one): on a square lattice with
    k = 6 EVERY spot has a tie at the k-th neighbour (4 at distance 1, two of the four at sqrt 2), which the reference
    resolves by cKDTree's traversal order (utils/graph.py:60-81).  Adjacency and a complete fit per lattice."""
    print("lattices: square 30x30, hex 28x28 (k=6 and spatial_method='grid')")
    out = {}
    sq = np.stack(np.meshgrid(np.arange(30.0), np.arange(30.0), indexing="ij"), axis=-1).reshape(-1, 2)
    hx = []
    for r in range(28):
        for c in range(28):
            hx.append((c + 0.5 * (r & 1), r * np.sqrt(3.0) / 2.0))
    hx = np.asarray(hx)
    sets = {"square_k6": (sq, "knn"), "square_grid": (sq, "grid"), "hex_k6": (hx, "knn"), "hex_grid": (hx, "grid"),
            "square100_k6": (sq * 100.0, "knn")}
    out["names"] = np.array(list(sets.keys()))
    for name, (coords, method) in sets.items():
        n = coords.shape[0]
        Y, X, _, _ = datagen.count_like(n, 400, 5, 0.1, 17)
        m = FlashDeconv(sketch_dim=64, preprocess="log_cpm", n_hvg=2000, spatial_method=method, k_neighbors=6, max_iter=30,
                        random_state=0, verbose=False)
        with np.errstate(all="ignore"):
            m.fit(Y, X, coords)
        ip, ix, _ = csr_parts(m.adjacency_)
        out[f"{name}_coords"] = coords
        out[f"{name}_method"] = np.array(method)
        out[f"{name}_indptr"] = ip
        out[f"{name}_indices"] = ix
        out[f"{name}_beta"] = m.beta_
        out[f"{name}_props"] = m.proportions_
        out[f"{name}_lambda"] = np.array(m.lambda_used_)
        out[f"{name}_n_iter"] = np.array(m.info_["n_iterations"])
        out[f"{name}_seed"] = np.array(17)
    save("lattice.npz", **out)


def golden_markers():
    """select_markers with its three specificity scores (utils/genes.py:148-235) on log-normal signatures of three shapes
    (case d has more cell types than genes, so some type is nowhere the largest one: the fallback branch :226-229)."""
    print("markers: diff / ratio / specificity")
    rs = np.random.RandomState(7)
    out = {}
    for tag, (K, G, nm) in dict(a=(6, 400, 12), b=(2, 90, 50), c=(9, 150, 7), d=(9, 6, 3)).items():   # d: more types than genes
        X = np.exp(rs.randn(K, G) * 0.8)
        out[f"{tag}_X"], out[f"{tag}_n_markers"] = X, np.array(nm)
        for method in ("diff", "ratio", "specificity"):
            mi, ma = select_markers(X, n_markers=nm, method=method)
            out[f"{tag}_{method}_idx"] = np.asarray(mi, dtype=np.int64)
            out[f"{tag}_{method}_assign"] = np.asarray(ma, dtype=np.int64)
    save("markers.npz", **out)


def golden_anndata():
    """The AnnData surface: the reference's io.load_reference (mean and sum), io.prepare_data /
    align_genes and tl.deconvolve run on a duck-typed AnnData (anndata is not installed; the loader only touches the
    attributes datagen.FakeAnnData has) with partial gene overlap, reversed gene order, duplicated gene names, unequal
    cells per type and, in the second file, scipy CSR matrices.  Stored: signatures, aligned gene lists and matrices, the
    obsm values, the dominant labels and the 15 uns keys."""
    import json
    from flashdeconv import tl
    from flashdeconv.io import align_genes, load_reference, load_spatial_data, prepare_data
    print("anndata surface: 120 spots x 520 genes, 73 reference cells x 500 genes, dense and CSR")
    case = datagen.anndata_case(11)
    for kind, wrap in (("dense", lambda a: a), ("csr", lambda a: sparse.csr_matrix(a))):
        st, ref = datagen.anndata_objects(case, wrap)
        out = {}
        for method in ("mean", "sum"):
            Xm, names, genes_ref = load_reference(ref, cell_type_key="celltype", method=method)
            out[f"X_{method}"] = Xm
        out["type_names"] = np.array([str(s) for s in names])
        Y0, coords0, genes_st = load_spatial_data(st)
        Ya, Xa, common = align_genes(Y0, out["X_mean"], genes_st, genes_ref)
        Yp, Xp, coords, names_p, genes_p = prepare_data(st, ref, cell_type_key="celltype")
        assert list(genes_p) == list(common) and np.array_equal(Xp, Xa)
        out["common_genes"] = np.array([str(s) for s in common])
        out["Y_aligned"] = np.asarray(Ya.todense()) if sparse.issparse(Ya) else np.asarray(Ya)
        out["X_aligned"] = Xa
        with np.errstate(all="ignore"):
            res = tl.deconvolve(st, ref, cell_type_key="celltype", sketch_dim=64, k_neighbors=4, n_hvg=300,
                                n_markers_per_type=20, copy=True)
            assert tl.deconvolve(st, ref, cell_type_key="celltype", sketch_dim=64, preprocess="pearson", spatial_method="radius",
                                 radius=1.6, key_added="alt") is None
        P = res.obsm["flashdeconv"]
        out["obsm"] = P.values
        out["obsm_columns"] = np.array([str(c) for c in P.columns])
        out["obsm_index"] = np.array([str(c) for c in P.index])
        dom = res.obs["flashdeconv_dominant"]
        out["dominant"] = np.array([str(c) for c in dom])
        out["dominant_categories"] = np.array([str(c) for c in dom.cat.categories])
        prm = dict(res.uns["flashdeconv_params"])
        prm["converged"] = bool(prm["converged"])
        prm["n_iterations"] = int(prm["n_iterations"])
        prm["cell_type_names"] = [str(s) for s in prm["cell_type_names"]]
        out["uns_json"] = np.array(json.dumps(prm, sort_keys=True))
        out["alt_obsm"] = st.obsm["alt"].values
        prm2 = dict(st.uns["alt_params"])
        prm2["converged"] = bool(prm2["converged"]); prm2["n_iterations"] = int(prm2["n_iterations"])
        prm2["cell_type_names"] = [str(s) for s in prm2["cell_type_names"]]
        out["alt_uns_json"] = np.array(json.dumps(prm2, sort_keys=True))
        out["alt_dominant"] = np.array([str(c) for c in st.obs["alt_dominant"]])
        out["input_sha256"] = np.array(datagen.sha256_arrays(case["Y"], case["cells"], case["coords"]))
        save(f"anndata_{kind}.npz", **out)


if __name__ == "__main__":
    only = sys.argv[1:]
    jobs = dict(omega=golden_omega, leverage=golden_leverage, graphs=golden_graphs, solver=golden_solver,
                objective=golden_objective, fits=golden_fits,
                fits_sparse=golden_fits_sparse, lattice=golden_lattice, anndata=golden_anndata, markers=golden_markers)
    for name, fn in jobs.items():
        if not only or name in only:
            fn()
