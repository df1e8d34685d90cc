"""
oracle/fdx_oracle.py -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.

CPU restatement (NumPy float64; pure-Python loops for tiny cases; C via
oracle/bcd_ref.c for the BCD sweep and the MT19937 hash stream) of the reference's
sketched graph-regularised NNLS path, stage by stage.  Every function cites the
reference lines (under /root/reference/flashdeconv/) it follows.

Who may import this: tests/, __graft_entry__.smoke(), bench.py's cpu_baseline leg.
The product package (flashdeconv_amd/) never imports it and has no CPU fallback.

Parity status: PINNED.  tests/test_oracle.py checks every function here against
golden vectors captured by importing the reference in the build container
(tests/golden/make_golden.py; script and vectors are committed).

Third-party arithmetic restated here rather than called:
  * numpy legacy RandomState = MT19937 + masked-rejection randint (bcd_ref.c and
    `mt19937_*` below); pinned by the hash/sign goldens.
  * scipy.spatial.cKDTree k-NN: restated as exact brute-force k+1 nearest
    (`knn_graph`); `knn_graph_kdtree` calls scipy the way the reference does and is
    used only for the timed CPU baseline at sizes brute force cannot reach.
  * LAPACK gesdd via numpy.linalg.svd: called as the reference calls it (an SVD is
    defined up to signs, and the leverage formula squares U).
"""
import ctypes
import os

import numpy as np
from scipy import sparse

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None


def _lib():
    """Load oracle/_build/liboracle.so (built by oracle/Makefile or __graft_entry__.build())."""
    global _LIB
    if _LIB is None:
        path = os.path.join(_HERE, "_build", "liboracle.so")
        if not os.path.exists(path):
            import subprocess
            subprocess.check_call(["make", "-s", "-C", _HERE])
        lib = ctypes.CDLL(path)
        dp = ctypes.POINTER(ctypes.c_double)
        ip = ctypes.POINTER(ctypes.c_int64)
        lib.oracle_bcd_iteration.argtypes = [dp, dp, dp, dp, ip, ip, ctypes.c_int64, ctypes.c_int64,
                                             ctypes.c_double, ctypes.c_double, dp, dp]
        lib.oracle_bcd_iteration.restype = None
        lib.oracle_countsketch_draw.argtypes = [ctypes.c_uint32, ctypes.c_int64, ctypes.c_int64, ip, ip]
        lib.oracle_countsketch_draw.restype = ctypes.c_int64
        _LIB = lib
    return _LIB


def _dptr(a):
    return a.ctypes.data_as(ctypes.POINTER(ctypes.c_double))


def _iptr(a):
    return a.ctypes.data_as(ctypes.POINTER(ctypes.c_int64))


# =========================================================================== a5
# utils/random.py:64-65 -> np.random.RandomState(seed); core/sketching.py:58-59
class MT19937:
    """Pure-Python MT19937 with numpy's legacy integer seeding (init_genrand)."""

    def __init__(self, seed):
        mt = [0] * 624
        mt[0] = seed & 0xFFFFFFFF
        for i in range(1, 624):
            mt[i] = (1812433253 * (mt[i - 1] ^ (mt[i - 1] >> 30)) + i) & 0xFFFFFFFF
        self.mt, self.idx = mt, 624

    def next32(self):
        if self.idx >= 624:
            mt = self.mt
            for k in range(624):
                y = (mt[k] & 0x80000000) | (mt[(k + 1) % 624] & 0x7FFFFFFF)
                v = mt[(k + 397) % 624] ^ (y >> 1)
                if y & 1:
                    v ^= 0x9908B0DF
                mt[k] = v
            self.idx = 0
        y = self.mt[self.idx]
        self.idx += 1
        y ^= y >> 11
        y ^= (y << 7) & 0x9D2C5680
        y ^= (y << 15) & 0xEFC60000
        y ^= y >> 18
        return y & 0xFFFFFFFF


def countsketch_draw_py(seed, n_genes, sketch_dim):
    """bucket = randint(0, d, G) then sign = choice([-1, 1], G) on one stream (sketching.py:58-59)."""
    rng = MT19937(seed)
    top = sketch_dim - 1
    mask = top
    for s in (1, 2, 4, 8, 16):
        mask |= mask >> s
    bucket = np.zeros(n_genes, dtype=np.int64)
    for g in range(n_genes):
        if top == 0:
            continue
        while True:
            v = rng.next32() & mask
            if v <= top:
                break
        bucket[g] = v
    sign = np.array([1 if (rng.next32() & 1) else -1 for _ in range(n_genes)], dtype=np.int64)
    return bucket, sign


def countsketch_draw(seed, n_genes, sketch_dim):
    """Same stream through oracle/bcd_ref.c (fast)."""
    bucket = np.empty(n_genes, dtype=np.int64)
    sign = np.empty(n_genes, dtype=np.int64)
    _lib().oracle_countsketch_draw(ctypes.c_uint32(int(seed) & 0xFFFFFFFF), n_genes, sketch_dim, _iptr(bucket), _iptr(sign))
    return bucket, sign


def countsketch_omega(n_genes, sketch_dim, leverage=None, seed=0):
    """(bucket, weight): Omega[g, bucket[g]] = weight[g]; one entry per gene (sketching.py:48-84)."""
    if leverage is None:
        p = np.full(n_genes, 1.0 / n_genes)                       # :51-52
    else:
        p = leverage / (np.sum(leverage) + 1e-10)                 # :55
    bucket, sign = countsketch_draw(seed, n_genes, sketch_dim)    # :58-59
    amp = np.clip(np.sqrt(p * n_genes + 1e-10), 0.1, 10.0)        # :62-63
    val = sign * amp                                              # :68
    col_sq = np.bincount(bucket, weights=val * val, minlength=sketch_dim)
    col_norm = np.maximum(np.sqrt(col_sq), 1e-10)                 # :77-78
    weight = val * (np.sqrt(n_genes / sketch_dim) / col_norm)[bucket]   # :81-82
    return bucket, weight


# =========================================================================== a3
def leverage_scores(X, regularization=1e-6):
    """utils/genes.py:238-290."""
    Xc = X - X.mean(axis=0, keepdims=True)                        # :264
    U, s, _ = np.linalg.svd(Xc.T, full_matrices=False)            # :270
    k = min(X.shape[0], X.shape[1], len(s))                       # :278
    w = s[:k] ** 2 / (s[:k] ** 2 + regularization)                # :281
    lev = np.sum(U[:, :k] ** 2 * w, axis=1)                       # :285
    return lev / (lev.sum() + regularization)                     # :288


# =========================================================================== a2
def select_hvg(Y, n_top=2000, min_mean=0.0125, max_mean=3.0, min_disp=0.5):
    """utils/genes.py:18-145 (dense :85-102, sparse :52-83, binning :104-145)."""
    N, G = Y.shape
    if sparse.issparse(Y):
        Y = Y.tocsr()
        lib = np.maximum(np.asarray(Y.sum(axis=1)).ravel(), 1.0)           # :57-58
        Z = (sparse.diags(10000.0 / lib) @ Y).tocsr()                     # :59-60
        Z.data = np.log1p(Z.data)                                         # :64
        mean = np.asarray(Z.sum(axis=0)).ravel() / N                       # :68-69
        if N >= 2:
            sq = np.bincount(Z.indices, weights=Z.data ** 2, minlength=G) / N   # :74-76
            var = np.maximum(N / (N - 1) * (sq - mean ** 2), 0)           # :79-80
        else:
            var = np.zeros(G)
    else:
        Yd = np.asarray(Y)
        tot = np.maximum(Yd.sum(axis=1, keepdims=True), 1)                 # :90-91
        Z = np.log1p(Yd / tot * 10000)                                    # :92-95
        mean = Z.mean(axis=0)
        var = Z.var(axis=0, ddof=1) if N >= 2 else np.zeros(G)            # :98-102
    disp = np.zeros(G)
    pos = mean[mean > 0]
    if len(pos) >= 2:                                                      # :110
        edges = np.unique(np.percentile(pos, np.linspace(0, 100, 21)))     # :111-112
        if len(edges) >= 2:
            which = np.clip(np.digitize(mean, edges) - 1, 0, len(edges) - 2)   # :116-117
            for b in range(len(edges) - 1):
                m = which == b
                if m.sum() > 1:                                            # :122-126
                    v = var[m]
                    disp[m] = (v - v.mean()) / (v.std() + 1e-10)
    ok = np.where((mean >= min_mean) & (mean <= max_mean) & (disp >= min_disp))[0]   # :129-133
    if len(ok) < n_top:
        pick = np.argsort(disp)[::-1][:n_top]                              # :137-138
    else:
        pick = ok[np.argsort(disp[ok])[::-1][:n_top]]                      # :141-143
    return np.sort(pick)


def select_markers(X, n_markers=50):
    """utils/genes.py:148-235, method="diff" (the only one reachable from fit)."""
    K, G = X.shape
    if n_markers == 0 or K == 0:
        return np.array([], dtype=np.intp)
    Xn = X / (X.sum(axis=1, keepdims=True) + 1e-10)                        # :184
    if K == 1:
        return np.arange(min(n_markers, G))                                # :186-190
    srt = np.sort(Xn, axis=0)[::-1]
    spec = srt[0] - srt[1]                                                 # :194-195
    top = np.argmax(Xn, axis=0)                                            # :214
    picked = []
    for k in range(K):
        mine = np.where(top == k)[0]
        if len(mine) > 0:
            picked.extend(mine[np.argsort(spec[mine])[::-1][:n_markers]])  # :220-224
        else:
            picked.extend(np.argsort(Xn[k])[::-1][:n_markers])             # :227-228
    return np.unique(picked)


def select_informative_genes(Y, X, n_hvg=2000, n_markers_per_type=50):
    """utils/genes.py:293-341."""
    with np.errstate(all="ignore"):
        hvg = select_hvg(Y, n_top=n_hvg)
    mk = select_markers(X, n_markers=n_markers_per_type)
    idx = np.union1d(hvg, mk).astype(np.intp)                              # :330
    if len(idx) == 0:
        raise ValueError("No genes selected. Increase n_hvg or n_markers_per_type.")
    return idx, leverage_scores(X[:, idx])


# =========================================================================== a4
def preprocess(Y, X, method):
    """core/deconv.py:147-235.  Dense Y -> dense float64; sparse Y stays sparse."""
    if method == "log_cpm":
        if sparse.issparse(Y):
            lib = np.asarray(Y.sum(axis=1)).ravel().astype(np.float64)
            lib[lib == 0] = 1.0                                            # :183-184
            Yt = (sparse.diags(1e4 / lib) @ Y).tocsr()
            Yt.data = np.log1p(Yt.data)                                    # :185-188
        else:
            Yt = np.log1p(Y / (Y.sum(axis=1, keepdims=True) + 1e-10) * 1e4)   # :190-191
        Xt = np.log1p(X / (X.sum(axis=1, keepdims=True) + 1e-10) * 1e4)    # :194-195
        return Yt, Xt
    if method == "pearson":
        theta = 100.0
        if sparse.issparse(Y):
            mu = np.asarray(Y.mean(axis=0)).ravel() + 1e-6                 # :208
            Yt = Y.multiply(1.0 / np.sqrt(mu + mu ** 2 / theta)).tocsr()   # :209-212
        else:
            mu = Y.mean(axis=0, keepdims=True) + 1e-6                      # :214
            Yt = Y / np.sqrt(mu + mu ** 2 / theta)                         # :215-217
        mx = X.mean(axis=0, keepdims=True) + 1e-6
        Xt = X / np.sqrt(mx + mx ** 2 / theta)                             # :220-223
        return Yt, Xt
    if method == "raw":
        return Y.astype(np.float64, copy=False), X.astype(np.float64, copy=False)   # :229
    raise ValueError(f"Unknown preprocess method: {method}. Choose from 'log_cpm', 'pearson', or 'raw'.")


# =========================================================================== a6
def project(Y_tilde, X_tilde, bucket, weight, sketch_dim):
    """Y_tilde @ Omega and X_tilde @ Omega (core/sketching.py:160-206): each output bucket is the
    gene-ordered signed weighted sum of the genes hashed to it."""
    G = len(bucket)
    Om = sparse.csr_matrix((weight, (np.arange(G), bucket)), shape=(G, sketch_dim))
    Ys = Y_tilde @ Om
    if sparse.issparse(Ys):
        Ys = Ys.toarray()
    return np.asarray(Ys, dtype=np.float64), np.asarray(X_tilde @ Om, dtype=np.float64)


def project_loops(Y_tilde, bucket, weight, sketch_dim):
    """Same projection as explicit loops (small inputs): the definition the HIP kernel follows."""
    Yt = np.asarray(Y_tilde.todense()) if sparse.issparse(Y_tilde) else np.asarray(Y_tilde)
    out = np.zeros((Yt.shape[0], sketch_dim))
    for g in range(len(bucket)):
        out[:, bucket[g]] += Yt[:, g] * weight[g]
    return out


# =========================================================================== a7
def _symmetrise(rows, cols, n):
    A = sparse.csr_matrix((np.ones(len(rows)), (rows, cols)), shape=(n, n))
    A = A + A.T                                                            # utils/graph.py:80
    A.data[:] = 1.0                                                        # :81
    A.sort_indices()
    return A


def knn_graph(coords, k=6):
    """utils/graph.py:25-83 with cKDTree.query(k+1) restated as exact brute force (O(N^2))."""
    coords = np.asarray(coords, dtype=np.float64)
    if coords.ndim != 2 or coords.shape[1] == 0:
        raise ValueError(f"coords must be 2D with at least 1 coordinate dimension, got shape {coords.shape}")
    n = coords.shape[0]
    kk = min(k, n - 1)                                                     # :51
    if kk <= 0:
        return sparse.csr_matrix((n, n), dtype=np.float64)                 # :53-57
    d2 = ((coords[:, None, :] - coords[None, :, :]) ** 2).sum(-1)
    order = np.argsort(d2, axis=1, kind="stable")[:, :kk + 1]              # :63 (k+1 nearest incl. self)
    rows = np.repeat(np.arange(n), kk + 1)
    cols = order.ravel()
    keep = rows != cols                                                    # :70-74
    return _symmetrise(rows[keep], cols[keep], n)


def knn_graph_kdtree(coords, k=6):
    """Same, calling scipy's cKDTree exactly as the reference does (timed CPU baseline only)."""
    from scipy.spatial import cKDTree
    n = coords.shape[0]
    kk = min(k, n - 1)
    if kk <= 0:
        return sparse.csr_matrix((n, n), dtype=np.float64)
    _, idx = cKDTree(coords).query(coords, k=kk + 1)
    rows = np.repeat(np.arange(n), kk + 1)
    cols = idx.ravel()
    keep = rows != cols
    return _symmetrise(rows[keep], cols[keep], n)


def radius_graph(coords, radius):
    """utils/graph.py:86-133: all unordered pairs with distance <= radius (cKDTree.query_pairs)."""
    coords = np.asarray(coords, dtype=np.float64)
    n = coords.shape[0]
    d2 = ((coords[:, None, :] - coords[None, :, :]) ** 2).sum(-1)
    i, j = np.where(np.triu(np.sqrt(d2) <= radius, k=1))
    if len(i) == 0:
        return sparse.csr_matrix((n, n), dtype=np.float64)
    A = sparse.csr_matrix((np.ones(2 * len(i)), (np.concatenate([i, j]), np.concatenate([j, i]))), shape=(n, n))
    A.sort_indices()
    return A


def grid_graph(coords):
    """utils/graph.py:136-172: radius = 1.5 x median nearest-neighbour distance."""
    coords = np.asarray(coords, dtype=np.float64)
    n = coords.shape[0]
    if n <= 1:
        return sparse.csr_matrix((n, n), dtype=np.float64)
    d2 = ((coords[:, None, :] - coords[None, :, :]) ** 2).sum(-1)
    np.fill_diagonal(d2, np.inf)
    spacing = np.median(np.sqrt(d2.min(axis=1)))                           # :165-167
    return radius_graph(coords, spacing * 1.5)                             # :170-172


def coords_to_adjacency(coords, method="knn", k=6, radius=None):
    """utils/graph.py:175-212."""
    if method == "knn":
        return knn_graph(coords, k)
    if method == "radius":
        if radius is None:
            raise ValueError("radius must be specified for radius method")
        return radius_graph(coords, radius)
    if method == "grid":
        return grid_graph(coords)
    raise ValueError(f"Unknown method: {method}")


# =========================================================================== a8
def auto_tune_lambda(X_sketch, A, alpha=0.005):
    """core/spatial.py:144-192."""
    g = np.mean(np.diag(X_sketch @ X_sketch.T))                            # :181-182
    deg = np.mean(np.asarray(A.sum(axis=1)).ravel()) if A.shape[0] else 0.0    # :185
    return float(alpha * g / max(deg, 1.0))                                # :190


# ================================================================== a9..a14
def objective(beta, H, XtX, YtY, A, lam, rho_eff):
    """core/solver.py:226-284 with L = D - A (core/spatial.py:70-73)."""
    cross = np.sum(beta * H.T)                                             # :271
    quad = np.sum((beta.T @ beta) * XtX)                                   # :273-274
    deg = np.asarray(A.sum(axis=1)).ravel()
    Lb = deg[:, None] * beta - A @ beta                                    # :278
    return 0.5 * (YtY - 2.0 * cross + quad) + 0.5 * lam * np.sum(beta * Lb) + rho_eff * np.sum(np.abs(beta))


def normalize_proportions(beta):
    """core/solver.py:431-452."""
    s = beta.sum(axis=1, keepdims=True)
    zero = (s == 0).ravel()
    p = beta / np.maximum(s, 1e-10)
    if zero.any():
        p[zero] = 1.0 / beta.shape[1]
    return p


def bcd_iteration_py(H, XtX, b_in, b_out, indices, indptr, lam, rho_eff):
    """core/solver.py:104-184 + :29-101 as pure-Python loops (tiny cases only)."""
    N, K = b_in.shape
    diffs = np.zeros(N)
    absm = np.zeros(N)
    for i in range(N):
        b = b_in[i].copy()
        nbrs = indices[indptr[i]:indptr[i + 1]]
        deg = len(nbrs)
        nb = np.zeros(K)
        for j in nbrs:
            nb += b_in[j]
        r = np.array([sum(XtX[k, j] * b[j] for j in range(K)) for k in range(K)])
        for k in range(K):
            old = b[k]
            res = H[k, i] - r[k] + XtX[k, k] * old
            if deg > 0:
                res += lam * nb[k]
            den = XtX[k, k] + lam * deg
            if den > 1e-10:
                t = res - rho_eff if res > rho_eff else (res + rho_eff if res < -rho_eff else 0.0)
                new = max(0.0, t / den)
            else:
                new = 0.0
            b[k] = new
            delta = new - old
            if delta != 0.0:
                r = r + delta * XtX[:, k]
        b_out[i] = b
        diffs[i] = np.max(np.abs(b - b_in[i]))
        absm[i] = np.max(np.abs(b_in[i]))
    return diffs, absm


def bcd_iteration_c(H, XtX, b_in, b_out, indices, indptr, lam, rho_eff, n_rows=None):
    """One sweep over the first `n_rows` spots (default: all); neighbour indices may point at any row of b_in
    (rows >= n_rows are a read-only halo in the sharded tests).  H is (K, n_rows)."""
    K = b_in.shape[1]
    N = b_in.shape[0] if n_rows is None else int(n_rows)
    diffs = np.empty(N)
    absm = np.empty(N)
    _lib().oracle_bcd_iteration(_dptr(H), _dptr(XtX), _dptr(b_in), _dptr(b_out), _iptr(indices), _iptr(indptr),
                                N, K, float(lam), float(rho_eff), _dptr(diffs), _dptr(absm))
    return diffs, absm


def bcd_solve(Y_sketch, X_sketch, A, lambda_=0.1, rho=0.01, max_iter=100, tol=1e-4, verbose=False, engine="c",
              return_trace=False):
    """core/solver.py:287-428.  engine="c" uses oracle/bcd_ref.c, "py" the pure-Python sweep."""
    N, K = Y_sketch.shape[0], X_sketch.shape[0]
    if N == 0 or K == 0:                                                   # :334-343
        return np.empty((N, K)), dict(converged=True, n_iterations=0, final_objective=0.0, objectives=[], final_change=0.0)
    XtX = np.ascontiguousarray(X_sketch @ X_sketch.T)                      # :346
    H = np.ascontiguousarray(X_sketch @ Y_sketch.T)                        # :347
    YtY = float(np.sum(Y_sketch ** 2))                                     # :348
    rho_eff = rho * np.mean(np.diag(XtX))                                  # :359-360
    A = A.tocsr()
    indices = np.ascontiguousarray(A.indices.astype(np.int64))             # :364-365
    indptr = np.ascontiguousarray(A.indptr.astype(np.int64))
    a = np.ones((N, K)) / K                                                # :372
    b = np.empty_like(a)
    sweep = bcd_iteration_c if engine == "c" else bcd_iteration_py
    objectives, trace = [], []
    converged, it, rel = False, -1, 0.0
    for it in range(max_iter):                                             # :385
        diffs, absm = sweep(H, XtX, a, b, indices, indptr, lambda_, rho_eff)
        rel = diffs.max() / (absm.max() + 1e-10)                           # :395-397
        trace.append(rel)
        if verbose and (it % 10 == 0 or it == max_iter - 1):               # :399-404
            objectives.append(objective(b, H, XtX, YtY, A, lambda_, rho_eff))
        a, b = b, a                                                        # :407
        if rel < tol:                                                      # :409-413
            converged = True
            break
    info = dict(converged=converged, n_iterations=it + 1,
                final_objective=objective(a, H, XtX, YtY, A, lambda_, rho_eff),
                objectives=objectives if verbose else [], final_change=rel)
    if return_trace:
        info["rel_changes"] = trace
    return a, info


# ====================================================================== a1
def fit(Y, X, coords, *, sketch_dim=512, lambda_spatial="auto", rho_sparsity=0.01, n_hvg=2000,
        n_markers_per_type=50, spatial_method="knn", k_neighbors=6, radius=None, max_iter=100, tol=1e-4,
        preprocess_method="log_cpm", random_state=0, engine="c", graph="brute"):
    """core/deconv.py:237-405, steps 1-6.  Returns a dict of every fitted attribute and the
    intermediate stage outputs the parity tests compare against."""
    gene_idx, lev = select_informative_genes(Y, X, n_hvg, n_markers_per_type)     # :309
    Ysub = Y[:, gene_idx]                                                          # :321-324
    if sparse.issparse(Ysub):
        Ysub = Ysub.tocsr()
    Yt, Xt = preprocess(Ysub, X[:, gene_idx], preprocess_method)                   # :330
    bucket, weight = countsketch_omega(len(gene_idx), sketch_dim, lev, random_state)   # :344-349
    Ys, Xs = project(Yt, Xt, bucket, weight, sketch_dim)
    if graph == "kdtree" and spatial_method == "knn":
        A = knn_graph_kdtree(np.asarray(coords, dtype=np.float64), k_neighbors)
    else:
        A = coords_to_adjacency(coords, spatial_method, k_neighbors, radius)       # :358
    lam = auto_tune_lambda(Xs, A) if lambda_spatial == "auto" else float(lambda_spatial)   # :371-376
    beta, info = bcd_solve(Ys, Xs, A, lam, rho_sparsity, max_iter, tol, engine=engine)     # :386
    return dict(gene_idx=gene_idx, leverage=lev, bucket=bucket, weight=weight, Y_sketch=Ys, X_sketch=Xs,
                adjacency=A, lambda_used=lam, beta=beta, proportions=normalize_proportions(beta), info=info)
