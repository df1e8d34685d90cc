"""-m gpu parity tests of the whole fit (FlashDeconv.fit_transform -> fdx_fit_dev) against golden vectors captured
from the reference.  Tolerance of the contract: 1e-4 relative Frobenius on beta_ / proportions_ (BASELINE.json);
the float64 kernels land around 1e-12, asserted here at 1e-8 so regressions show."""
import numpy as np
import pytest
from scipy import sparse

import datagen
import fdx_oracle as orc
from conftest import load_golden, rel_fro

pytestmark = pytest.mark.gpu
TOL = 1e-8


def _check(model, g, tol=TOL):
    assert np.array_equal(model.gene_idx_, g["gene_idx"])
    A = model.adjacency_
    assert np.array_equal(A.indptr, g["indptr"]) and np.array_equal(A.indices, g["indices"])
    np.testing.assert_allclose(model.lambda_used_, float(g["lambda_used"]), rtol=1e-10)
    assert model.info_["n_iterations"] == int(g["n_iterations"])
    assert model.info_["converged"] == bool(g["converged"])
    assert rel_fro(model.beta_, g["beta"]) < tol
    assert rel_fro(model.proportions_, g["proportions"]) < tol
    np.testing.assert_allclose(model.info_["final_objective"], float(g["final_objective"]), rtol=max(tol, 1e-9))
    np.testing.assert_allclose(model.info_["final_change"], float(g["final_change"]), rtol=1e-5, atol=1e-12)
    np.testing.assert_allclose(model.proportions_.sum(axis=1), 1.0, rtol=1e-12)
    assert np.all(model.beta_ >= 0)


@pytest.mark.parametrize("d", [64, 128])
def test_fit_counts_small_dense(d):
    from flashdeconv_amd import FlashDeconv
    g = load_golden(f"fit_counts_100x500x5_d{d}.npz")
    m = FlashDeconv(sketch_dim=d)
    p = m.fit_transform(g["Y"].astype(np.int64), g["X"], g["coords"])
    assert p is m.proportions_ and p.shape == (100, 5)
    _check(m, g)
    assert m.summary()["n_genes_used"] == 500 and m.get_dominant_cell_type().shape == (100,)


def test_fit_counts_small_csr_and_f32():
    from flashdeconv_amd import FlashDeconv
    g = load_golden("fit_counts_100x500x5_d64_csr.npz")
    m = FlashDeconv(sketch_dim=64).fit(sparse.csr_matrix(g["Y"].astype(np.float64)), g["X"], g["coords"])
    _check(m, g)
    # float32 input: numpy keeps the reference's log-CPM in float32, we compute in float64 -> ~1e-7 apart
    g = load_golden("fit_counts_100x500x5_d64_f32.npz")
    m = FlashDeconv(sketch_dim=64).fit(g["Y"].astype(np.float32), g["X"], g["coords"])
    _check(m, g, tol=1e-5)


@pytest.mark.parametrize("suffix", ["", "_csr"])
def test_fit_pearson(suffix):
    from flashdeconv_amd import FlashDeconv
    g = load_golden(f"fit_pearson_120x300x4{suffix}.npz")
    Y = g["Y"].astype(np.int64)
    if suffix:
        Y = sparse.csr_matrix(Y.astype(np.float64))
    m = FlashDeconv(sketch_dim=48, preprocess="pearson", max_iter=40).fit(Y, g["X"], g["coords"])
    _check(m, g)


@pytest.mark.parametrize("pre", ["log_cpm", "pearson", "raw"])
@pytest.mark.parametrize("kind", ["scipy_f64", "scipy_i32", "csc", "torch_csr"])
def test_fit_sparse_input_stays_sparse(pre, kind):
    """CSR spot matrix, gene selection active, an empty spot: the device CSR kernels (csr_kernels.cpp) against the
    reference's sparse branches (utils/genes.py:52-83, core/deconv.py:181-188, core/sketching.py:194-199)."""
    import torch
    from flashdeconv_amd import FlashDeconv
    g = load_golden(f"fit_sparse_{pre}_300x900x5.npz")
    Yd = g["Y"]
    if kind == "scipy_f64":
        Y = sparse.csr_matrix(Yd.astype(np.float64))
    elif kind == "scipy_i32":
        Y = sparse.csr_matrix(Yd.astype(np.int32))
    elif kind == "csc":
        Y = sparse.csc_matrix(Yd.astype(np.float32))
    else:
        Y = torch.from_numpy(Yd.astype(np.float32)).cuda().to_sparse_csr()
    m = FlashDeconv(sketch_dim=64, preprocess=pre, n_hvg=250, n_markers_per_type=10, max_iter=30).fit(Y, g["X"], g["coords"])
    _check(m, g)


@pytest.mark.parametrize("n,G,K,d,n_hvg,pre", [(37, 90, 3, 16, 2000, "log_cpm"), (257, 700, 6, 100, 300, "log_cpm"),
                                               (130, 64, 2, 1, 2000, "raw"), (513, 1300, 9, 192, 500, "pearson"),
                                               (64, 40, 4, 33, 25, "log_cpm")])
def test_sparse_fit_vs_oracle_odd_shapes(n, G, K, d, n_hvg, pre):
    """CSR path on shapes the goldens do not cover (tiny n, sketch_dim 1 / odd, gene selection on and off): against the
    oracle's sparse branches, plus CSR == dense input when the reference's two log-CPM rules coincide (no empty spot)."""
    from flashdeconv_amd import FlashDeconv
    rs = np.random.RandomState(n + G)
    X = np.exp(rs.randn(K, G) * 0.7)
    B = rs.dirichlet(np.ones(K), size=n)
    Y = rs.poisson(B @ X * 3.0) * (rs.rand(n, G) < 0.35)
    Y[:, 0] += 1                                                   # no empty spot: sparse and dense log-CPM agree
    coords = rs.rand(n, 2) * 20
    Ys = sparse.csr_matrix(Y.astype(np.float64))
    kw = dict(sketch_dim=d, preprocess=pre, n_hvg=n_hvg, n_markers_per_type=5, max_iter=15, tol=1e-9)
    m = FlashDeconv(**kw).fit(Ys, X, coords)
    want = orc.fit(Ys, X, coords, sketch_dim=d, preprocess_method=pre, n_hvg=n_hvg, n_markers_per_type=5, max_iter=15,
                   tol=1e-9, graph="kdtree")
    assert np.array_equal(m.gene_idx_, want["gene_idx"])
    assert m.info_["n_iterations"] == want["info"]["n_iterations"]
    assert rel_fro(m.beta_, want["beta"]) < 1e-8 and rel_fro(m.proportions_, want["proportions"]) < 1e-8
    md = FlashDeconv(**kw).fit(Y.astype(np.float64), X, coords)
    assert np.array_equal(md.gene_idx_, m.gene_idx_) and rel_fro(md.beta_, m.beta_) < 1e-9


@pytest.mark.parametrize("dtype", [np.float32, np.float64])
def test_sparse_rows_longer_than_the_keep_buffer(dtype, monkeypatch):
    """Fused CSR sketch -> H, log-CPM (csr_kernels.cpp): the select pass compacts a row's selected entries into the wave's keep buffer
    in LDS.  At sketch_dim 1024 the buffer is small: rows with ~600 selected entries overflow it (walked again from memory), rows
    with ~60 fit, in the same launch - both against the oracle (core/deconv.py:181-188, core/sketching.py:194-199) and against
    a launch whose buffer holds 64 entries only."""
    from flashdeconv_amd import FlashDeconv
    n, G, K, d = 300, 1200, 8, 1024
    rs = np.random.RandomState(5)
    X = np.exp(rs.randn(K, G) * 0.7)
    B = rs.dirichlet(np.ones(K), size=n)
    dens = np.where(np.arange(n) % 2 == 0, 0.5, 0.05)[:, None]
    Y = rs.poisson(B @ X * 3.0) * (rs.rand(n, G) < dens)
    Y[:, 0] += 1
    coords = rs.rand(n, 2) * 20
    Ys = sparse.csr_matrix(Y.astype(dtype))
    assert (np.diff(Ys.indptr) > 400).sum() > 100 and (np.diff(Ys.indptr) < 100).sum() > 100
    kw = dict(sketch_dim=d, preprocess="log_cpm", n_hvg=2000, max_iter=10, tol=1e-9)
    m = FlashDeconv(**kw).fit(Ys, X, coords)
    want = orc.fit(sparse.csr_matrix(Y.astype(np.float64)), X, coords, sketch_dim=d, preprocess_method="log_cpm", n_hvg=2000, max_iter=10,
                   tol=1e-9, graph="kdtree")
    tol = 1e-8 if dtype == np.float64 else 1e-5
    assert m.info_["n_iterations"] == want["info"]["n_iterations"]
    assert rel_fro(m.beta_, want["beta"]) < tol and rel_fro(m.proportions_, want["proportions"]) < tol
    monkeypatch.setenv("FDX_CSR_KEEP_CAP", "64")                 # every long row overflows the keep buffer: walked again from memory
    m2 = FlashDeconv(**kw).fit(Ys, X, coords)
    assert rel_fro(m2.beta_, m.beta_) < 1e-10


def test_csr_gene_moments_and_validation():
    from flashdeconv_amd import _lib
    from flashdeconv_amd.utils import genes
    rs = np.random.RandomState(3)
    Yd = (rs.poisson(0.3, size=(500, 700)) * (rs.rand(500, 700) < 0.3)).astype(np.float64)
    Yd[11] = 0
    Y = sparse.csr_matrix(Yd)
    csr = _lib.CsrOnDevice.from_scipy(Y)
    assert csr.view.sorted_rows == 1                       # canonical scipy matrix: the cursor kernel is used
    mean, var, colsum = csr.gene_moments(want_colsum=True)
    mean2, var2, none = csr.gene_moments()
    assert none is None and np.allclose(mean2, mean, rtol=1e-14) and np.allclose(var2, var, rtol=1e-12)
    csr.free()
    lib_size = np.maximum(Yd.sum(axis=1, keepdims=True), 1.0)
    Z = np.log1p(Yd / lib_size * 1e4)
    np.testing.assert_allclose(mean, Z.mean(axis=0), rtol=1e-12, atol=1e-15)
    np.testing.assert_allclose(var, Z.var(axis=0, ddof=1), rtol=1e-9, atol=1e-13)
    np.testing.assert_allclose(colsum, Yd.sum(axis=0), rtol=1e-13)
    assert np.array_equal(genes.select_hvg(Y, 100), genes.select_hvg(Yd, 100))
    # unsorted rows: same statistics through the full-scan kernel (sorted_rows = 0), and a false claim is caught
    perm_cols = Y.copy()
    for r in range(0, 500, 3):
        a, b = perm_cols.indptr[r], perm_cols.indptr[r + 1]
        perm_cols.indices[a:b] = perm_cols.indices[a:b][::-1].copy()
        perm_cols.data[a:b] = perm_cols.data[a:b][::-1].copy()
    perm_cols.has_sorted_indices = False
    un = _lib.CsrOnDevice.from_scipy(perm_cols, sort=False)
    assert un.view.sorted_rows == 0
    m3, v3, c3 = un.gene_moments(want_colsum=True)
    np.testing.assert_allclose(m3, mean, rtol=1e-13)
    np.testing.assert_allclose(v3, var, rtol=1e-10, atol=1e-15)
    np.testing.assert_allclose(c3, colsum, rtol=1e-13)
    un.view.sorted_rows = 1
    with pytest.raises(_lib.FdxError, match="sorted_rows"):
        _lib.check(_lib.load().fdx_csr_check_dev(__import__("ctypes").byref(un.view), None))
    un.free()
    bad = sparse.csr_matrix(Yd)
    bad.indices = bad.indices.copy()
    bad.indices[5] = 700                                    # column out of range: rejected before any kernel uses it
    with pytest.raises(_lib.FdxError, match="malformed"):
        _lib.CsrOnDevice.from_scipy(bad)


def test_csr_structure_check_is_cached_per_tensor_object_only():
    """The cached verdict of fdx_csr_check_dev belongs to one live tensor object: a different CSR tensor of the same shape
    and entry count - even one that the caching allocator places at the freed tensor's addresses - is checked itself,
    and an in-place edit of the indices through torch invalidates the entry."""
    import gc
    import torch
    from flashdeconv_amd import _lib
    rs = np.random.RandomState(4)
    Yd = (rs.poisson(0.4, size=(300, 400)) * (rs.rand(300, 400) < 0.3)).astype(np.float32)
    t = torch.from_numpy(Yd).cuda().to_sparse_csr()
    # int32 columns: the layout the library uses in place (int64 columns are copied on every use and always checked)
    good = torch.sparse_csr_tensor(t.crow_indices(), t.col_indices().to(torch.int32), t.values(), size=(300, 400))
    del t
    a = _lib.CsrOnDevice.from_torch(good)
    assert id(good) in _lib.CsrOnDevice._checked
    b = _lib.CsrOnDevice.from_torch(good)                      # second use of the same object: served from the cache
    assert b.view.sorted_rows == a.view.sorted_rows == 1
    crow, col, val = good.crow_indices().clone(), good.col_indices().clone(), good.values().clone()
    addr = (good.crow_indices().data_ptr(), good.col_indices().data_ptr())
    del a, b, good
    gc.collect()
    assert not _lib.CsrOnDevice._checked                       # the entry died with the tensor
    col_bad = col.clone()
    col_bad[7] = 400                                           # column out of range
    bad = torch.sparse_csr_tensor(crow, col_bad, val, size=(300, 400))
    with pytest.raises(_lib.FdxError, match="malformed"):
        _lib.CsrOnDevice.from_torch(bad)
    del bad, col_bad
    # same-address reuse: allocate until the allocator hands the old index block out again (usually at once)
    for _ in range(4):
        c2 = col.clone()
        if c2.data_ptr() == addr[1]:
            break
    c2[3] = -1
    again = torch.sparse_csr_tensor(crow.clone(), c2, val, size=(300, 400))
    with pytest.raises(_lib.FdxError, match="malformed"):
        _lib.CsrOnDevice.from_torch(again)
    # in-place edit through torch: the version counter moves, the check runs again and fails
    ok = torch.sparse_csr_tensor(crow, col, val, size=(300, 400))
    _lib.CsrOnDevice.from_torch(ok)
    ok.col_indices()[11] = 4000
    with pytest.raises(_lib.FdxError, match="malformed"):
        _lib.CsrOnDevice.from_torch(ok)


@pytest.mark.parametrize("pre,dtype", [("raw", np.float32), ("log_cpm", np.float32), ("log_cpm", np.float64), ("pearson", np.float32)])
def test_fused_sketch_contraction_equals_two_kernel_path(pre, dtype, monkeypatch):
    """sketch_contract_kernel (rows -> LDS accumulators -> MFMA -> H, no Y_sketch) against the scatter-sketch + xyt_split
    pair it replaces: same per-row arithmetic, contraction index split over 16 waves instead of 8, so H agrees to
    rounding, not to the bit."""
    from flashdeconv_amd import FlashDeconv
    n, G, K = 5003, 1200, 17                      # n not a multiple of 16: the last group is partial
    Y, X, coords, _ = datagen.count_like(n, G, K, 0.1, 4)
    Y = Y.astype(dtype)
    kw = dict(sketch_dim=256, preprocess=pre, n_hvg=G, max_iter=25)
    monkeypatch.setenv("FDX_FUSED", "1")          # raw / pearson use the fused kernel by default, log-CPM on request
    a = FlashDeconv(**kw).fit(Y, X, coords)
    a2 = FlashDeconv(**kw).fit(Y, X, coords)
    monkeypatch.delenv("FDX_FUSED")
    monkeypatch.setenv("FDX_NO_FUSED", "1")
    b = FlashDeconv(**kw).fit(Y, X, coords)
    assert a.timings_["gram_ms"] == 0.0 and b.timings_["gram_ms"] > 0.0          # the two paths really ran
    assert np.array_equal(a.beta_, a2.beta_)                                      # deterministic
    # float32 rows under log-CPM: the tile kernel evaluates a float32-class log1p (as the reference does for float32 input),
    # the two-kernel path the float64 one - float32 rounding apart, not float64 rounding
    tol = 1e-5 if (pre == "log_cpm" and dtype == np.float32) else 1e-12
    assert a.info_["n_iterations"] == b.info_["n_iterations"] and rel_fro(a.beta_, b.beta_) < tol


def test_fit_gauss_1000_config1_miniature():
    from flashdeconv_amd import FlashDeconv
    g = load_golden("fit_gauss_1000x2000x10.npz")
    Y, X, coords, _ = datagen.gaussian_raw(1000, 2000, 10, seed=0)
    assert datagen.sha256_arrays(Y, X, coords) == str(g["input_sha256"])
    m = FlashDeconv(sketch_dim=512, preprocess="raw").fit(Y, X, coords)
    _check(m, g)


def test_fit_gauss_800_fixed_lambda():
    from flashdeconv_amd import FlashDeconv
    g = load_golden("fit_gauss_800x2000x20.npz")
    Y, X, coords, _ = datagen.gaussian_raw(800, 2000, 20, seed=1)
    m = FlashDeconv(sketch_dim=512, preprocess="raw", lambda_spatial=0.5, rho_sparsity=0.02).fit(Y, X, coords)
    _check(m, g)


def test_fit_counts_1000_full_100_iterations():
    from flashdeconv_amd import FlashDeconv
    g = load_golden("fit_counts_1000x2000x10.npz")
    Y, X, coords, _ = datagen.count_like(1000, 2000, 10, 0.1, 0)
    m = FlashDeconv(sketch_dim=512).fit(Y, X, coords)
    assert m.info_["n_iterations"] == 100 and not m.info_["converged"]
    _check(m, g)


def test_fit_counts_600_k30():
    from flashdeconv_amd import FlashDeconv
    g = load_golden("fit_counts_600x1000x30.npz")
    Y, X, coords, _ = datagen.count_like(600, 1000, 30, 0.1, 3)
    m = FlashDeconv(sketch_dim=256, max_iter=60).fit(Y, X, coords)
    _check(m, g)


@pytest.mark.parametrize("K,pre", [(70, "log_cpm"), (100, "raw"), (128, "log_cpm")])
def test_fit_with_more_than_64_cell_types(K, pre):
    """65-128 cell types: leverage scores on the Cholesky-QR route with the 128-row factor, sketch -> H by the two-kernel path, the
    sweeps on the next padded instantiation (pad types stay exactly zero and never reach beta_ / proportions_)."""
    from flashdeconv_amd import FlashDeconv
    Y, X, coords, _ = datagen.count_like(900, 700, K, seed=K)
    kw = dict(sketch_dim=256, preprocess=pre, n_hvg=2000, max_iter=15, random_state=2)
    want = orc.fit(Y, X, coords, sketch_dim=256, preprocess_method=pre, n_hvg=2000, max_iter=15, random_state=2)
    m = FlashDeconv(**kw).fit(Y, X, coords)
    assert m.beta_.shape == (900, K) and m.proportions_.shape == (900, K)
    assert m.info_["n_iterations"] == want["info"]["n_iterations"]
    assert rel_fro(m.beta_, want["beta"]) < 1e-8
    np.testing.assert_allclose(m.info_["final_objective"], want["info"]["final_objective"], rtol=1e-9)
    np.testing.assert_allclose(m.proportions_.sum(1), 1.0, atol=1e-12)
    if K == 70:                                    # the same through the CSR path (no fused contraction above 64 types) and float32 rows
        mc = FlashDeconv(**kw).fit(sparse.csr_matrix(Y), X, coords)
        assert rel_fro(mc.beta_, want["beta"]) < 1e-8
        m32 = FlashDeconv(**kw).fit(Y.astype(np.float32), X, coords)
        assert rel_fro(m32.beta_, want["beta"]) < 1e-5


def test_leverage_scores_vs_reference():
    from flashdeconv_amd.utils.genes import compute_leverage_scores
    g = load_golden("leverage.npz")
    for name in g["names"]:
        lev = compute_leverage_scores(g[f"{name}_X"])
        np.testing.assert_allclose(lev, g[f"{name}_lev"], rtol=1e-9, atol=1e-14, err_msg=str(name))


@pytest.mark.parametrize("K,G,cond", [(30, 2000, 1e2), (50, 5000, 1e5), (64, 3000, 1e6), (7, 90, 1e3), (2, 50, 1.0), (33, 1000, 1e4),
                                      (65, 700, 1e2), (100, 3000, 1e3), (128, 2100, 1e2)])
def test_leverage_multi_cu_path_vs_lapack_and_one_workgroup(K, G, cond, monkeypatch):
    """The streaming Gram/eigen/rotate passes against LAPACK (the reference's gesdd route, via the oracle) and against
    the one-workgroup one-sided Jacobi kernel, on signature matrices with a prescribed condition number."""
    from flashdeconv_amd.utils.genes import compute_leverage_scores
    rs = np.random.RandomState(K + G)
    U, _ = np.linalg.qr(rs.randn(G, K))
    Vt, _ = np.linalg.qr(rs.randn(K, K))
    s = np.geomspace(1.0, 1.0 / cond, K) * 50.0
    X = (U * s) @ Vt + rs.rand(G, 1) * 3.0          # (G, K) with a per-gene offset that centring removes
    X = np.ascontiguousarray(X.T)
    want = orc.leverage_scores(X)
    got = compute_leverage_scores(X)
    np.testing.assert_allclose(got, want, rtol=2e-8 * max(1.0, cond / 1e4), atol=1e-15)
    monkeypatch.setenv("FDX_LEV_ONE_WG", "1")
    one = compute_leverage_scores(X)
    np.testing.assert_allclose(got, one, rtol=2e-8 * max(1.0, cond / 1e4), atol=1e-15)


def test_leverage_cholesky_qr_route_refuses_rank_deficient_signatures(capfd, monkeypatch):
    """Two identical cell types (and a third that is their mean): the ridge-augmented Cholesky factorisation meets a pivot of
    ~reg against a diagonal of ~1e3, refuses, and fdx_leverage_end runs the Jacobi SVD passes; the well-conditioned matrix
    next to it stays on the fast route.  Both against LAPACK through the oracle."""
    from flashdeconv_amd.utils.genes import compute_leverage_scores
    rs = np.random.RandomState(4)
    X = rs.gamma(2.0, 3.0, size=(12, 700))
    monkeypatch.setenv("FDX_DEBUG", "1")
    got = compute_leverage_scores(X)
    assert "route=cholesky-qr" in capfd.readouterr().err
    np.testing.assert_allclose(got, orc.leverage_scores(X), rtol=1e-10, atol=1e-16)
    X[5] = X[2]
    X[7] = 0.5 * (X[2] + X[3])
    got = compute_leverage_scores(X)
    assert "route=jacobi-svd" in capfd.readouterr().err
    np.testing.assert_allclose(got, orc.leverage_scores(X), rtol=1e-8, atol=1e-15)


@pytest.mark.parametrize("K,G,cond", [(129, 700, 1e2), (200, 3000, 1e3), (150, 2000, 1e4), (272, 1500, 1e2), (140, 64, 5.0)])
def test_leverage_above_128_types_takes_the_blocked_cholesky_qr(K, G, cond, capfd, monkeypatch):
    """129 - 272 cell types (utils/genes.py:238-290 takes any K): CholeskyQR2 with the K x K part in global memory
    (leverage_kernels.cpp, lev_big_*) against LAPACK through the oracle; a few milliseconds where the Jacobi SVD passes took
    0.15 - 2 s; rank-deficient signatures are refused and fall back to those passes."""
    import time
    from flashdeconv_amd.utils.genes import compute_leverage_scores
    rs = np.random.RandomState(K + G)
    r = min(K, G)
    U, _ = np.linalg.qr(rs.randn(G, r))
    Vt, _ = np.linalg.qr(rs.randn(K, K))
    X = np.ascontiguousarray(((U * (np.geomspace(1.0, 1.0 / cond, r) * 50.0)) @ Vt[:r] + rs.rand(G, 1) * 3.0).T)
    want = orc.leverage_scores(X)
    monkeypatch.setenv("FDX_DEBUG", "1")
    got = compute_leverage_scores(X)
    assert "route=cholesky-qr" in capfd.readouterr().err
    np.testing.assert_allclose(got, want, rtol=2e-8 * max(1.0, cond / 1e4), atol=1e-15)
    if K == 200:
        monkeypatch.delenv("FDX_DEBUG")
        t0 = time.perf_counter()
        compute_leverage_scores(X)
        assert time.perf_counter() - t0 < 5e-3                      # round 4: ~650 ms
        monkeypatch.setenv("FDX_DEBUG", "1")
    if K == 129:
        X = rs.gamma(2.0, 3.0, size=(K, G))
        X[5] = X[2]
        X[7] = 0.5 * (X[2] + X[3])
        got = compute_leverage_scores(X)
        assert "route=jacobi-svd" in capfd.readouterr().err
        np.testing.assert_allclose(got, orc.leverage_scores(X), rtol=1e-8, atol=1e-15)


def test_sketch_dim_2048_dense_and_csr_against_the_oracle():
    """sketch_dim above the one-kernel forms' 1024 / 1056 (core/sketching.py takes any d): dense and CSR rows take the two-kernel
    sketch -> H path."""
    from flashdeconv_amd import FlashDeconv
    rs = np.random.RandomState(5)
    n, G, K, d = 700, 2600, 9, 2048
    X = np.exp(rs.randn(K, G) * 0.7)
    B = rs.dirichlet(np.ones(K), size=n)
    Y = rs.poisson(B @ X * 3.0) * (rs.rand(n, G) < 0.35)
    Y[:, 0] += 1
    coords = rs.rand(n, 2) * 30
    Ys = sparse.csr_matrix(Y.astype(np.float64))
    for pre in ("log_cpm", "raw"):
        kw = dict(sketch_dim=d, preprocess=pre, n_hvg=3000, n_markers_per_type=5, max_iter=15, tol=1e-9)
        want = orc.fit(Ys, X, coords, sketch_dim=d, preprocess_method=pre, n_hvg=3000, n_markers_per_type=5, max_iter=15, tol=1e-9,
                       graph="kdtree")
        for Yin in (Ys, Y.astype(np.float64)):
            m = FlashDeconv(**kw).fit(Yin, X, coords)
            assert m.info_["n_iterations"] == want["info"]["n_iterations"]
            assert rel_fro(m.beta_, want["beta"]) < 1e-8 and rel_fro(m.proportions_, want["proportions"]) < 1e-8


def test_seed_reproducibility_and_errors():
    from flashdeconv_amd import FlashDeconv
    g = load_golden("fit_counts_100x500x5_d64.npz")
    Y, X, coords = g["Y"].astype(np.int64), g["X"], g["coords"]
    a = FlashDeconv(sketch_dim=64, random_state=7).fit_transform(Y, X, coords)
    b = FlashDeconv(sketch_dim=64, random_state=7).fit_transform(Y, X, coords)
    assert np.array_equal(a, b)
    with pytest.raises(ValueError, match="Gene dimension mismatch"):
        FlashDeconv().fit(Y, X[:, :10], coords)
    with pytest.raises(ValueError, match="Spot count mismatch"):
        FlashDeconv().fit(Y, X, coords[:5])
    with pytest.raises(ValueError, match="at least one cell type"):
        FlashDeconv().fit(Y, X[:0], coords)
    with pytest.raises(RuntimeError, match="not been fitted"):
        FlashDeconv().get_cell_type_proportions()
    with pytest.raises(ValueError, match="radius must be specified"):
        FlashDeconv(spatial_method="radius")


def test_radius_and_grid_methods_run():
    # reference tests/test_integration.py:238-271
    from flashdeconv_amd import FlashDeconv
    g = load_golden("fit_counts_100x500x5_d64.npz")
    Y, X, coords = g["Y"].astype(np.int64), g["X"], g["coords"]
    for kw in (dict(spatial_method="radius", radius=1.5), dict(spatial_method="grid")):
        m = FlashDeconv(sketch_dim=64, **kw).fit(Y, X, coords)
        assert m.proportions_.shape == (100, 5) and np.allclose(m.proportions_.sum(axis=1), 1.0)
        assert m.adjacency_.nnz > 0 and (m.adjacency_ != m.adjacency_.T).nnz == 0


def test_fit_with_gene_selection_active():
    """G > n_hvg: HVG moments on the GPU + markers + gene subset (reference utils/genes.py:18-145, 293-341)."""
    from flashdeconv_amd import FlashDeconv
    from flashdeconv_amd.utils import genes
    g = load_golden("fit_genesel_150x600x4.npz")
    Y = g["Y"].astype(np.int64)
    assert np.array_equal(genes.select_hvg(Y, 200), g["hvg_idx"])
    assert np.array_equal(genes.select_hvg(sparse.csr_matrix(Y.astype(np.float64)), 200), g["hvg_idx_csr"])
    gi, lev = genes.select_informative_genes(Y, g["X"], 200, 10)
    assert np.array_equal(gi, g["gene_idx"])
    np.testing.assert_allclose(lev, g["leverage"], rtol=1e-9, atol=1e-15)
    m = FlashDeconv(sketch_dim=64, n_hvg=200, n_markers_per_type=10, max_iter=30).fit(Y, g["X"], g["coords"])
    _check(m, g)
    assert m.summary()["n_genes_used"] == len(g["gene_idx"]) < 600


_FakeAnnData = datagen.FakeAnnData


def test_tl_deconvolve_anndata_surface():
    # reference tests/test_integration.py:274-344: obsm / obs / uns layout and kwargs forwarding
    import flashdeconv_amd as fd
    g = load_golden("fit_counts_100x500x5_d64.npz")
    Y, X, coords = g["Y"].astype(np.float64), g["X"], g["coords"]
    rs = np.random.RandomState(0)
    genes = np.array([f"g{i}" for i in range(500)])
    labels = np.repeat([f"type{k}" for k in range(5)], 8)
    cells = np.vstack([rs.poisson(X[k] * 3.0, size=(8, 500)) for k in range(5)]).astype(np.float64)
    ref = _FakeAnnData(cells, genes[::-1], [f"c{i}" for i in range(40)], obs={"celltype": labels})
    ref.X = cells[:, ::-1]
    st = _FakeAnnData(Y, genes, [f"s{i}" for i in range(100)], obsm={"spatial": coords})
    out = fd.tl.deconvolve(st, ref, cell_type_key="celltype", sketch_dim=64, k_neighbors=4, copy=True)
    assert "flashdeconv" not in st.obsm and out is not st
    P = out.obsm["flashdeconv"]
    assert list(P.columns) == [f"type{k}" for k in range(5)] and list(P.index) == list(st.obs_names)
    np.testing.assert_allclose(P.values.sum(axis=1), 1.0, rtol=1e-12)
    assert str(out.obs["flashdeconv_dominant"].dtype) == "category"
    prm = out.uns["flashdeconv_params"]
    assert len(prm) == 15 and prm["k_neighbors"] == 4 and prm["n_genes_used"] == 500 and prm["n_cell_types"] == 5
    assert fd.tl.deconvolve(st, ref, cell_type_key="celltype", sketch_dim=64, key_added="fdx") is None
    assert "fdx" in st.obsm and "fdx_dominant" in st.obs and "fdx_params" in st.uns
    with pytest.raises(ValueError, match="not found in adata_ref.obs"):
        fd.tl.deconvolve(st, ref, cell_type_key="nope")


@pytest.mark.parametrize("ties", ["auto", "ckdtree"])
@pytest.mark.parametrize("name", ["square_k6", "hex_k6", "square100_k6"])
def test_lattice_k6_with_the_reference_tie_order_is_exact(name, ties):
    """On lattices where the k-th neighbour is tied (square: every spot) the neighbour lists come from the host restatement of
    scipy's cKDTree order (csrc/kdtree_order.cpp) - the adjacency is then the reference's index for index and the fit agrees
    at the usual 1e-8 (the device's own tie rule, knn_ties="index", leaves 4.5e-4 / 3.1e-4, see below).  knn_ties="auto" is
    the DEFAULT (the fit is repeated on the reference's choice once the device build has reported ties), "ckdtree" decides
    before the first solve.  No warning in either mode; the tie count stays in info_."""
    import warnings
    from flashdeconv_amd import FlashDeconv
    g = load_golden("lattice.npz")
    coords = g[f"{name}_coords"]
    Y, X, _, _ = datagen.count_like(coords.shape[0], 400, 5, 0.1, int(g[f"{name}_seed"]))
    with warnings.catch_warnings():
        warnings.simplefilter("error")
        kw = {} if ties == "auto" else dict(knn_ties=ties)            # "auto" through the constructor's default
        m = FlashDeconv(sketch_dim=64, preprocess="log_cpm", n_hvg=2000, spatial_method="knn", k_neighbors=6, max_iter=30,
                        **kw).fit(Y, X, coords)
        assert m.knn_ties == ties
    A = m.adjacency_
    assert np.array_equal(A.indptr, g[f"{name}_indptr"]) and np.array_equal(A.indices, g[f"{name}_indices"])
    assert m.info_["knn_ties"] > 0 and m.info_["n_iterations"] == int(g[f"{name}_n_iter"])
    np.testing.assert_allclose(m.lambda_used_, float(g[f"{name}_lambda"]), rtol=1e-10)
    assert rel_fro(m.beta_, g[f"{name}_beta"]) < 1e-8 and rel_fro(m.proportions_, g[f"{name}_props"]) < 1e-8
    # a tie-free input is untouched by the option: the device graph stays
    rs = np.random.RandomState(1)
    c2 = rs.rand(500, 2) * 22
    Y2, X2, _, _ = datagen.count_like(500, 300, 4, 0.1, 2)
    a = FlashDeconv(sketch_dim=32, max_iter=10, knn_ties="ckdtree").fit(Y2, X2, c2)
    b = FlashDeconv(sketch_dim=32, max_iter=10).fit(Y2, X2, c2)
    assert a.info_["knn_ties"] == 0 and np.array_equal(a.beta_, b.beta_)
    with pytest.raises(ValueError, match="knn_ties"):
        FlashDeconv(knn_ties="lapack")


@pytest.mark.parametrize("dim,shape", [(4, (6, 5, 6, 5)), (5, (4, 4, 4, 4, 3)), (3, (10, 9, 10))])
def test_lattice_in_three_to_five_dimensions_takes_the_reference_tie_order_by_default(dim, shape):
    """Integer lattices with 3, 4 and 5 coordinates (every k-th neighbour tied; above 3 dimensions the device search is exhaustive):
    the default knn_ties="auto" gives the reference's adjacency - scipy's cKDTree order, utils/graph.py:60-81 - index for index, in
    ONE solve, and the fit equals the oracle's on that graph.  (Round 4: the tie route stopped at 3 dimensions and raised.)"""
    import warnings
    from flashdeconv_amd import FlashDeconv
    from flashdeconv_amd.utils.graph import build_knn_graph
    grids = np.meshgrid(*[np.arange(s, dtype=np.float64) for s in shape], indexing="ij")
    coords = np.stack([g.ravel() for g in grids], axis=1)
    rs = np.random.RandomState(dim)
    coords = np.ascontiguousarray(coords[rs.permutation(len(coords))])
    n = len(coords)
    want = orc.knn_graph_kdtree(coords, 6)
    Y, X, _, _ = datagen.count_like(n, 300, 4, 0.1, 3)
    with warnings.catch_warnings():
        warnings.simplefilter("error")
        m = FlashDeconv(sketch_dim=32, max_iter=12).fit(Y, X, coords)
    A = m.adjacency_
    assert m.info_["knn_ties"] > 0
    assert np.array_equal(A.indptr, want.indptr) and np.array_equal(A.indices, want.indices)
    ref = orc.fit(Y, X, coords, sketch_dim=32, preprocess_method="log_cpm", n_hvg=2000, graph="kdtree", engine="c", max_iter=12)
    assert m.info_["n_iterations"] == ref["info"]["n_iterations"]
    assert rel_fro(m.proportions_, ref["proportions"]) < 1e-8
    B = build_knn_graph(coords, k=6, ties="ckdtree")
    assert np.array_equal(B.indptr, want.indptr) and np.array_equal(B.indices, want.indices)


def _lattice_case(name, **kw):
    from flashdeconv_amd import FlashDeconv
    g = load_golden("lattice.npz")
    coords = g[f"{name}_coords"]
    Y, X, _, _ = datagen.count_like(coords.shape[0], 400, 5, 0.1, int(g[f"{name}_seed"]))
    m = FlashDeconv(sketch_dim=64, preprocess="log_cpm", n_hvg=2000, spatial_method=str(g[f"{name}_method"]), k_neighbors=6,
                    max_iter=30, **kw).fit(Y, X, coords)
    A = m.adjacency_
    same_graph = np.array_equal(A.indptr, g[f"{name}_indptr"]) and np.array_equal(A.indices, g[f"{name}_indices"])
    return m, g, same_graph


@pytest.mark.parametrize("name", ["square_grid", "hex_grid"])
def test_lattice_without_ties_matches_reference(name):
    """Regular lattices where the neighbour sets are unambiguous: spatial_method='grid' (radius 1.5 x the nearest
    neighbour distance: the 8 / 6 surrounding bins) on square and hexagonal lattices.  Adjacency index-exact, abundances
    at the usual tolerance."""
    import warnings
    with warnings.catch_warnings():
        warnings.simplefilter("error")                               # spatial_method='grid': no ties, no warning
        m, g, same_graph = _lattice_case(name)
    assert m.info_["knn_ties"] == 0
    assert same_graph
    assert m.info_["n_iterations"] == int(g[f"{name}_n_iter"])
    np.testing.assert_allclose(m.lambda_used_, float(g[f"{name}_lambda"]), rtol=1e-10)
    assert rel_fro(m.beta_, g[f"{name}_beta"]) < 1e-8 and rel_fro(m.proportions_, g[f"{name}_props"]) < 1e-8


@pytest.mark.parametrize("name", ["square_k6", "square100_k6", "hex_k6"])
def test_square_lattice_k6_tie_deviation_is_bounded(name):
    """k = 6 on a square lattice: every spot has four neighbours at distance 1 and must pick two of the four at sqrt(2).
    The reference inherits cKDTree's traversal order there (an effectively arbitrary pair per spot, degrees 6-10 after the
    union symmetrisation); the device rule (knn_ties="index") takes the two lowest spot indices (degree 8 in the interior).  Both are
    k-NN graphs of the same points; they differ, and so do lambda (through the mean degree) and the abundances: measured
    4.5e-4 (square), 3.1e-4 (hexagonal: ties only along the border) relative Frobenius in the proportions.  That is the size
    of the reference's own dependence on the ORDER in which the same spots are listed (4.9e-4 - 5.2e-4 / 2.1e-4 - 2.6e-4:
    tests/test_oracle.py::test_lattice_ties_make_the_reference_depend_on_spot_order), which bounds what any tie rule other
    than a bit-for-bit cKDTree emulation can achieve.  DESIGN.md §4."""
    with pytest.warns(UserWarning, match="k-NN ties"):               # the deviation is announced, not silent
        m, g, same_graph = _lattice_case(name, knn_ties="index")
    assert m.info_["knn_ties"] > 0.5 * m.n_spots_ * (name != "hex_k6")  # square: nearly every spot; hexagonal: the border
    assert not same_graph                                            # if this ever holds, tighten the test above instead
    A = m.adjacency_
    assert (A != A.T).nnz == 0 and A.diagonal().sum() == 0
    deg = np.diff(A.indptr)
    assert deg.min() >= 6                                            # still a k-NN graph: every spot keeps >= k neighbours
    gap_p, gap_b = rel_fro(m.proportions_, g[f"{name}_props"]), rel_fro(m.beta_, g[f"{name}_beta"])
    print(f"lattice tie deviation {name}: proportions {gap_p:.3e} beta {gap_b:.3e} lambda {m.lambda_used_:.6g} vs {float(g[name + '_lambda']):.6g}")
    # measured (MI355X, this build) + 10 %: square 4.5e-4, hexagonal 3.1e-4; the contract is 1e-4 - this is the documented
    # deviation, asserted so that it cannot grow unnoticed
    bound = {"square_k6": 5.0e-4, "square100_k6": 5.0e-4, "hex_k6": 3.5e-4}[name]
    assert gap_p < bound and gap_b < bound, (name, gap_p, gap_b)


def test_dense_whole_transcriptome_float64_gene_subset():
    """A dense float64 matrix over ~35000 genes (a row is 280 KB, more than the LDS holds): the gene subset Y[:, gene_idx]
    (core/deconv.py:321) takes the direct-gather kernel; the fit must equal the same fit on the pre-selected columns."""
    from flashdeconv_amd import FlashDeconv
    rs = np.random.RandomState(5)
    n, G_all, K = 300, 35000, 4
    Y = np.zeros((n, G_all))
    live = np.sort(rs.choice(G_all, 1500, replace=False))
    Y[:, live] = rs.poisson(3.0, size=(n, len(live)))
    X = np.exp(rs.randn(K, G_all) * 0.3)
    X[:, live] *= np.exp(rs.randn(K, len(live)))
    coords = rs.rand(n, 2) * 20
    kw = dict(sketch_dim=64, n_hvg=600, n_markers_per_type=20, max_iter=15)
    a = FlashDeconv(**kw).fit(Y, X, coords)
    assert 600 <= len(a.gene_idx_) < G_all
    b = FlashDeconv(**dict(kw, n_hvg=len(a.gene_idx_))).fit(Y[:, a.gene_idx_], X[:, a.gene_idx_], coords)
    assert a.info_["n_iterations"] == b.info_["n_iterations"] and rel_fro(a.beta_, b.beta_) < 1e-12


@pytest.mark.parametrize("kind", ["dense", "csr"])
def test_tl_deconvolve_with_matrices_resident_on_the_gpu(kind):
    """an AnnData whose matrices are CUDA tensors (dense or sparse_csr) goes through tl.deconvolve
    without coming back to the host - per-type signatures by fdx_type_sums(_csr)_dev, gene alignment as a device-side column
    selection, the spot tensor straight into FlashDeconv.fit.  Same result as the host path on the same data."""
    import torch
    import flashdeconv_amd as fd
    from flashdeconv_amd.io import load_reference, prepare_data
    g = load_golden("fit_counts_100x500x5_d64.npz")
    Y, X, coords = g["Y"].astype(np.float64), g["X"], g["coords"]
    rs = np.random.RandomState(1)
    genes_st = np.array([f"g{i}" for i in range(500)])
    genes_ref = np.array([f"g{i}" for i in range(30, 530)])[::-1]            # partial overlap, other order
    labels = rs.permutation(np.repeat([f"type{k}" for k in range(5)], [9, 3, 14, 1, 13]))
    kidx = np.array([int(s[4:]) for s in labels])
    cells = rs.poisson(X[kidx][:, ::-1].repeat(1, axis=0)[:, :500] * 3.0).astype(np.float64)
    cells[rs.rand(*cells.shape) < 0.6] = 0.0
    dev = torch.device("cuda", 0)

    def on_device(a):
        t = torch.from_numpy(np.ascontiguousarray(a)).to(dev)
        return t.to_sparse_csr() if kind == "csr" else t

    host_in = (lambda a: sparse.csr_matrix(a)) if kind == "csr" else (lambda a: a)
    ref_h = _FakeAnnData(host_in(cells), genes_ref, [f"c{i}" for i in range(40)], obs={"celltype": labels})
    st_h = _FakeAnnData(host_in(Y), genes_st, [f"s{i}" for i in range(100)], obsm={"spatial": coords})
    ref_d = _FakeAnnData(on_device(cells), genes_ref, [f"c{i}" for i in range(40)], obs={"celltype": labels})
    st_d = _FakeAnnData(on_device(Y), genes_st, [f"s{i}" for i in range(100)], obsm={"spatial": coords})
    for method in ("mean", "sum"):
        Xh, nh, _ = load_reference(ref_h, cell_type_key="celltype", method=method)
        Xd, nd, _ = load_reference(ref_d, cell_type_key="celltype", method=method)
        assert list(nh) == list(nd)
        np.testing.assert_allclose(Xd, Xh, rtol=1e-13, atol=0)
    Yh, Xh, _, _, genes_h = prepare_data(st_h, ref_h, cell_type_key="celltype")
    Yd, Xd, _, _, genes_d = prepare_data(st_d, ref_d, cell_type_key="celltype")
    assert list(genes_h) == list(genes_d) and len(genes_d) == 470 and Yd.is_cuda
    Yd_dense = Yd.to_dense() if kind == "csr" else Yd
    assert np.array_equal(Yd_dense.cpu().numpy(), Yh.toarray() if kind == "csr" else Yh)
    assert fd.tl.deconvolve(st_h, ref_h, cell_type_key="celltype", sketch_dim=64, k_neighbors=4) is None
    assert fd.tl.deconvolve(st_d, ref_d, cell_type_key="celltype", sketch_dim=64, k_neighbors=4) is None
    Ph, Pd = st_h.obsm["flashdeconv"], st_d.obsm["flashdeconv"]
    assert list(Ph.columns) == list(Pd.columns)
    assert rel_fro(Pd.values, Ph.values) < 1e-9
    assert st_d.uns["flashdeconv_params"]["n_genes_used"] == st_h.uns["flashdeconv_params"]["n_genes_used"]


@pytest.mark.parametrize("kind", ["dense", "csr"])
@pytest.mark.parametrize("where", ["host", "device"])
def test_anndata_surface_against_the_reference_goldens(kind, where):
    """SURVEY.md section 8 f4, pinned to the reference: io.load_reference (mean / sum), io.prepare_data / align_genes and
    tl.deconvolve of the reference were run on the duck-typed AnnData of datagen.anndata_case (tests/golden/make_golden.py:
    golden_anndata) - partial gene overlap, reversed gene order, duplicated gene names, unequal cells per type, dense and
    scipy-CSR matrices.  Here the same objects go through flashdeconv_amd's host path (numpy / scipy matrices) and device
    path (CUDA tensors, dense / sparse_csr) and must give the reference's obsm / obs / uns
    (flashdeconv/io/loader.py:73-258, flashdeconv/tl/_deconvolve.py:116-174)."""
    import json
    import torch
    import flashdeconv_amd as fd
    from flashdeconv_amd.io import load_reference, prepare_data
    g = load_golden(f"anndata_{kind}.npz")
    case = datagen.anndata_case(11)
    assert datagen.sha256_arrays(case["Y"], case["cells"], case["coords"]) == str(g["input_sha256"])
    dev = torch.device("cuda", 0)
    if where == "host":
        wrap = (lambda a: sparse.csr_matrix(a)) if kind == "csr" else (lambda a: a)
    else:
        def wrap(a):
            t = torch.from_numpy(np.ascontiguousarray(a)).to(dev)
            return t.to_sparse_csr() if kind == "csr" else t
    st, ref = datagen.anndata_objects(case, wrap)
    for method in ("mean", "sum"):
        Xm, names, _ = load_reference(ref, cell_type_key="celltype", method=method)
        assert [str(s) for s in names] == list(g["type_names"])
        np.testing.assert_allclose(Xm, g[f"X_{method}"], rtol=1e-13, atol=0)
    Ya, Xa, _, _, genes = prepare_data(st, ref, cell_type_key="celltype")
    assert [str(s) for s in genes] == list(g["common_genes"]) and len(genes) == 470
    np.testing.assert_allclose(Xa, g["X_aligned"], rtol=1e-13, atol=0)
    if where == "device":
        Ya = (Ya.to_dense() if kind == "csr" else Ya).cpu().numpy()
    elif kind == "csr":
        Ya = Ya.toarray()
    assert np.array_equal(np.asarray(Ya), g["Y_aligned"])
    res = fd.tl.deconvolve(st, ref, cell_type_key="celltype", sketch_dim=64, k_neighbors=4, n_hvg=300, n_markers_per_type=20,
                           copy=True)
    P = res.obsm["flashdeconv"]
    assert [str(c) for c in P.columns] == list(g["obsm_columns"]) and [str(c) for c in P.index] == list(g["obsm_index"])
    assert rel_fro(P.values, g["obsm"]) < 1e-8
    dom = res.obs["flashdeconv_dominant"]
    assert [str(c) for c in dom] == list(g["dominant"]) and [str(c) for c in dom.cat.categories] == list(g["dominant_categories"])

    def same_params(got, want_json):
        want = json.loads(str(want_json))
        got = dict(got)
        assert sorted(got) == sorted(want) and len(got) == 15
        for k, w in want.items():
            if k == "lambda_spatial":
                np.testing.assert_allclose(float(got[k]), w, rtol=1e-10)
            elif k == "cell_type_names":
                assert [str(s) for s in got[k]] == w
            else:
                assert got[k] == w, (k, got[k], w)
    same_params(res.uns["flashdeconv_params"], g["uns_json"])
    assert fd.tl.deconvolve(st, ref, cell_type_key="celltype", sketch_dim=64, preprocess="pearson", spatial_method="radius",
                            radius=1.6, key_added="alt") is None
    assert rel_fro(st.obsm["alt"].values, g["alt_obsm"]) < 1e-8
    assert [str(c) for c in st.obs["alt_dominant"]] == list(g["alt_dominant"])
    same_params(st.uns["alt_params"], g["alt_uns_json"])


def test_more_than_63_neighbours_per_spot_against_the_oracle():
    """The reference takes any k (utils/graph.py:51: k_actual = min(k, N - 1)); the device's k-NN lists stop at 63.  Above that the
    lists come from the restated cKDTree on the host and the adjacency goes up as given: same graph, same fit as the oracle."""
    from flashdeconv_amd import FlashDeconv
    Y, X, coords, _ = datagen.gaussian_raw(700, 200, 6, seed=9)
    for k in (64, 100, 5000):                                  # 5000 > n - 1: every spot is every spot's neighbour
        m = FlashDeconv(sketch_dim=64, preprocess="raw", n_hvg=200, k_neighbors=k, max_iter=30).fit(Y, X, coords)
        want = orc.fit(Y, X, coords, sketch_dim=64, preprocess_method="raw", n_hvg=200, k_neighbors=k, max_iter=30, graph="kdtree")
        A, B = m.adjacency_, want["adjacency"].tocsr()
        assert np.array_equal(A.indptr, B.indptr) and np.array_equal(A.indices, B.indices), k
        assert m.info_["n_iterations"] == want["info"]["n_iterations"]
        np.testing.assert_allclose(m.lambda_used_, want["lambda_used"], rtol=1e-10)
        assert rel_fro(m.beta_, want["beta"]) < 1e-8, k


@pytest.mark.parametrize("dtype,K,d", [(np.float32, 8, 64), (np.float64, 8, 64), (np.float32, 40, 256), (np.float32, 60, 128)])
def test_sparse_rows_longer_than_the_register_resident_part(dtype, K, d):
    """Fused CSR sketch (csr_kernels.cpp): a wave holds 24 x 64 float32 (16 x 64 float64 / wide-type) entries of its row in
    registers; what lies beyond is streamed from memory.  Rows of ~2400 stored entries beside short and empty ones, selected-gene
    counts above the keep buffer, 8 / 40 / 60 cell types (one, three, four type tiles), against the oracle (core/deconv.py:181-188,
    core/sketching.py:194-199)."""
    from flashdeconv_amd import FlashDeconv
    n, G = 230, 4000
    rs = np.random.RandomState(8)
    X = np.exp(rs.randn(K, G) * 0.7)
    B = rs.dirichlet(np.ones(K), size=n)
    dens = np.choose(np.arange(n) % 3, [0.6, 0.05, 0.3])[:, None]
    Y = rs.poisson(B @ X * 2.0) * (rs.rand(n, G) < dens)
    Y[7] = 0                                                  # an empty spot
    Y[:, 1] += 1
    Y[7, 1] = 0
    coords = rs.rand(n, 2) * 20
    Ys = sparse.csr_matrix(Y.astype(dtype))
    assert (np.diff(Ys.indptr) > 1700).sum() > 40 and (np.diff(Ys.indptr) == 0).sum() == 1
    kw = dict(sketch_dim=d, preprocess="log_cpm", n_hvg=5000, max_iter=8, tol=1e-9)
    m = FlashDeconv(**kw).fit(Ys, X, coords)
    want = orc.fit(sparse.csr_matrix(Y.astype(np.float64)), X, coords, sketch_dim=d, preprocess_method="log_cpm", n_hvg=5000, max_iter=8,
                   tol=1e-9, graph="kdtree")
    assert m.info_["n_iterations"] == want["info"]["n_iterations"]
    tol = 1e-8 if dtype == np.float64 else 1e-5
    assert rel_fro(m.beta_, want["beta"]) < tol and rel_fro(m.proportions_, want["proportions"]) < tol


def test_csr_gene_moments_above_one_million_rows():
    """csr_moments_cursor_kernel walks a stripe of more than 4096 rows (256 stripes: above 1,048,576 spots) in parts whose partial
    sums add up in the stripe's slot - a path no smaller matrix takes.  1.2M very sparse rows against scipy's own sums
    (utils/genes.py:52-83)."""
    import torch
    from flashdeconv_amd import _lib
    n, G = 1_200_000, 700
    g = torch.Generator(device="cuda").manual_seed(3)
    nnz_row = 12
    cols = torch.randint(0, G, (n, nnz_row), generator=g, device="cuda", dtype=torch.int64)
    cols, _ = torch.sort(cols, dim=1)
    keep = torch.ones_like(cols, dtype=torch.bool)
    keep[:, 1:] = cols[:, 1:] != cols[:, :-1]                              # distinct columns per row
    vals = torch.randint(1, 9, (n, nnz_row), generator=g, device="cuda").to(torch.float32)
    crow = torch.zeros(n + 1, dtype=torch.int64, device="cuda")
    crow[1:] = torch.cumsum(keep.sum(dim=1), dim=0)
    Yt = torch.sparse_csr_tensor(crow, cols[keep].to(torch.int32), vals[keep], size=(n, G))
    csr = _lib.CsrOnDevice.from_torch(Yt)
    assert csr.view.sorted_rows == 1
    mean, var, _ = csr.gene_moments()
    Ys = sparse.csr_matrix((vals[keep].cpu().numpy().astype(np.float64), cols[keep].cpu().numpy(), crow.cpu().numpy()), shape=(n, G))
    lib = np.maximum(np.asarray(Ys.sum(axis=1)).ravel(), 1.0)
    Z = (sparse.diags(10000.0 / lib) @ Ys).tocsr()
    Z.data = np.log1p(Z.data)
    m_ref = np.asarray(Z.sum(axis=0)).ravel() / n
    sq = np.bincount(Z.indices, weights=Z.data ** 2, minlength=G) / n
    v_ref = np.maximum(n / (n - 1) * (sq - m_ref ** 2), 0)
    np.testing.assert_allclose(mean, m_ref, rtol=1e-11)
    np.testing.assert_allclose(var, v_ref, rtol=1e-9)
