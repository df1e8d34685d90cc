"""Randomised small fits against the oracle (dense / CSR, all preprocess modes, odd sizes).  Not a test: a bug net."""
import os, sys, json
os.environ.setdefault("OMP_NUM_THREADS", "8"); os.environ.setdefault("OPENBLAS_NUM_THREADS", "8")
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "oracle"), os.path.join(ROOT, "tests")):
    sys.path.insert(0, p)
import numpy as np
from scipy import sparse
import fdx_oracle as orc
from flashdeconv_amd import FlashDeconv

def rel(a, b):
    return float(np.linalg.norm(a - b) / max(np.linalg.norm(b), 1e-300))

rs = np.random.RandomState(int(os.environ.get("SEED", 0)))
bad = 0
for trial in range(int(os.environ.get("TRIALS", 60))):
    n = int(rs.choice([2, 3, 7, 33, 64, 65, 200, 257, 513, 1000]))
    K = int(rs.choice([1, 2, 5, 8, 17, 31, 33, 48, 70, 100]))
    G = int(rs.choice([K + 1, 40, 130, 257, 700, 2100]))
    d = int(rs.choice([1, 7, 16, 64, 100, 512]))
    pre = str(rs.choice(["log_cpm", "pearson", "raw"]))
    kind = str(rs.choice(["dense64", "dense32", "int", "csr"]))
    k_nb = int(rs.choice([1, 3, 6, 12]))
    n_hvg = int(rs.choice([2000, max(5, G // 3)]))
    if n_hvg < G and (n < 7 or kind == "dense32"):
        # gene selection ranks genes by a binned z-score of their variance: with 2-3 spots most genes tie exactly (argsort's tie order
        # decides), and float32 rows move near-ties (the reference's own float32 and float64 gene sets differ by
        # the same genes) - neither is what this net is for
        n_hvg = max(2000, G)
    dim = int(rs.choice([1, 2, 3, 4, 6] if os.environ.get("HIGHDIM") else [1, 2, 3]))
    method = str(rs.choice(["knn", "knn", "radius", "grid"]))
    if dim > 3:
        method = "knn"                       # radius / grid graphs take 1-3 coordinate columns
    X = np.exp(rs.randn(K, G) * 0.6)
    B = rs.dirichlet(np.ones(K), size=n)
    Y = rs.poisson(B @ X * 2.0).astype(np.float64)
    if rs.rand() < 0.5:
        Y *= (rs.rand(n, G) < 0.4)
    coords = rs.rand(n, dim) * 10
    Yin = {"dense64": Y, "dense32": Y.astype(np.float32), "int": Y.astype(np.int64), "csr": sparse.csr_matrix(Y)}[kind]
    kw = dict(sketch_dim=d, preprocess=pre, n_hvg=n_hvg, n_markers_per_type=5, k_neighbors=k_nb, max_iter=12, tol=1e-9,
              spatial_method=method, radius=2.0 if method == "radius" else None)
    tag = dict(trial=trial, n=n, K=K, G=G, d=d, pre=pre, kind=kind, k=k_nb, n_hvg=n_hvg, dim=dim, method=method)
    try:
        with np.errstate(all="ignore"):
            want = orc.fit(Yin if kind != "int" else Y, X, coords, sketch_dim=d, preprocess_method=pre, n_hvg=n_hvg, n_markers_per_type=5,
                           k_neighbors=k_nb, max_iter=12, tol=1e-9, spatial_method=method, radius=kw["radius"],
                           graph="kdtree" if method == "knn" else "brute")
    except Exception as e:
        print("oracle-error", type(e).__name__, str(e)[:80], json.dumps(tag)); continue
    try:
        m = FlashDeconv(**kw).fit(Yin, X, coords)
    except Exception as e:
        bad += 1; print("FDX-ERROR", type(e).__name__, str(e)[:120], json.dumps(tag)); continue
    ok_genes = np.array_equal(m.gene_idx_, want["gene_idx"])
    tol = 1e-5 if kind == "dense32" else 1e-7
    eb = rel(m.beta_, want["beta"]) if ok_genes else float("nan")
    same_it = m.info_["n_iterations"] == want["info"]["n_iterations"]
    if not ok_genes or not (eb < tol) or not same_it:
        bad += 1; print("MISMATCH", "genes" if not ok_genes else "", eb, same_it, json.dumps(tag))
print("done; problems:", bad)
