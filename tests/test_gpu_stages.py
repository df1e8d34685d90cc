"""-m gpu parity tests of the sketch and graph stages through the C ABI."""
import numpy as np
import pytest
from scipy import sparse

import datagen
import fdx_oracle as orc
from conftest import load_golden, rel_fro

pytestmark = pytest.mark.gpu


# ------------------------------------------------------------------------------------------- graphs
def test_graphs_match_reference_golden():
    from flashdeconv_amd.utils import graph as G
    g = load_golden("graphs.npz")
    for name in g["names"]:
        coords = g[f"{name}_coords"]
        method = str(g[f"{name}_method"])
        radius = float(g[f"{name}_radius"])
        A = G.coords_to_adjacency(coords, method=method, k=int(g[f"{name}_k"]), radius=None if radius < 0 else radius)
        assert A.shape == (coords.shape[0],) * 2
        assert np.array_equal(A.indptr, g[f"{name}_indptr"]), name
        assert np.array_equal(A.indices, g[f"{name}_indices"]), name
        assert A.dtype == np.float64 and (A.nnz == 0 or np.all(A.data == 1.0))


def test_knn_tie_count_is_zero_on_tie_free_goldens_and_positive_on_lattices():
    """fdx_graph_knn_ties: spots whose k-th and (k+1)-th nearest neighbours are exactly equidistant - the inputs on which
    the reference's graph depends on cKDTree's traversal order (utils/graph.py:60-63).  None on the tie-free golden graphs
    (which is why they are index-exact), every interior spot of a square lattice with k = 6 (four neighbours at 1, two of
    the four at sqrt 2), only border spots of a hexagonal one, none with k = 4 or k = 8 on the square lattice."""
    from flashdeconv_amd import _lib
    g = load_golden("graphs.npz")
    seen = 0
    for name in [str(s) for s in g["names"]]:
        if str(g[f"{name}_method"]) != "knn":
            continue
        coords = np.ascontiguousarray(g[f"{name}_coords"], dtype=np.float64)
        if coords.shape[0] < 2 or coords.shape[1] > 3:
            continue
        gr = _lib.Graph.from_coords_knn(coords, int(g[f"{name}_k"]))
        assert gr.knn_ties() == 0, name
        gr.close()
        seen += 1
    assert seen >= 5
    lat = load_golden("lattice.npz")
    sq, hx = lat["square_k6_coords"], lat["hex_k6_coords"]
    n = sq.shape[0]
    t6 = _lib.Graph.from_coords_knn(sq, 6).knn_ties()
    assert t6 >= (30 - 2) ** 2 and t6 <= n
    assert _lib.Graph.from_coords_knn(sq * 100.0, 6).knn_ties() == t6
    th = _lib.Graph.from_coords_knn(hx, 6).knn_ties()
    assert 0 < th < 4 * 28 + 40                                      # border spots only
    # k = 4: the four at distance 1 are the set, the next (sqrt 2) is farther - no tie in the interior; k = 8 likewise
    assert _lib.Graph.from_coords_knn(sq, 4).knn_ties() < 4 * 30
    assert _lib.Graph.from_coords_knn(sq, 8).knn_ties() < 4 * 30
    # k + 1 = 8 and 16 sit at the edge of the list-length classes of the kernel: the spare slot must still be there
    rs = np.random.RandomState(2)
    pts = rs.rand(3000, 2) * 50
    for k in (7, 15, 31):
        assert _lib.Graph.from_coords_knn(pts, k).knn_ties() == 0
    dup = np.concatenate([pts[:100], pts[:100] + np.array([1.0, 0.0])])   # exact translates: distance ties everywhere? no -
    assert _lib.Graph.from_coords_knn(dup, 6).knn_ties() >= 0              # only that the count is well defined


@pytest.mark.parametrize("n,dim,k", [(5000, 2, 6), (3000, 3, 8), (4000, 2, 15), (2500, 1, 3), (777, 2, 40)])
def test_knn_vs_oracle_kdtree(n, dim, k):
    from flashdeconv_amd.utils import graph as G
    rs = np.random.RandomState(n + k)
    coords = rs.rand(n, dim) * (n ** (1.0 / dim))
    A = G.build_knn_graph(coords, k=k)
    B = orc.knn_graph_kdtree(coords, k)
    assert np.array_equal(A.indptr, B.indptr) and np.array_equal(A.indices, B.indices)
    assert (A != A.T).nnz == 0 and A.diagonal().sum() == 0


@pytest.mark.parametrize("case", ["square", "hex", "cube", "line", "duplicates", "square_offset"])
@pytest.mark.parametrize("k", [4, 6, 8, 12, 20, 40])
def test_knn_tie_rule_is_distance_then_index(case, k):
    """Inputs FULL of exact distance ties (lattices in 1-3 dimensions, repeated points): the neighbour lists must be those
    of the brute-force oracle, whose stable argsort of the squared distances IS the documented rule - nearer first, lower
    spot index among equals.  The list lengths 5 ... 41 go through every instantiation of the k-NN kernel (8 / 16 / 32 / 64
    slots; the flat 3 x 3 walk and the shell walk), with and without a spare slot for the tie test."""
    from flashdeconv_amd.utils import graph as G
    rs = np.random.RandomState(k)
    if case == "square":
        gx, gy = np.meshgrid(np.arange(33.0), np.arange(31.0))
        coords = np.stack([gx.ravel(), gy.ravel()], 1)
    elif case == "square_offset":            # not exactly representable steps: ties only where the arithmetic says so
        gx, gy = np.meshgrid(np.arange(30.0) * 0.1 + 1e3, np.arange(30.0) * 0.3 - 7.7)
        coords = np.stack([gx.ravel(), gy.ravel()], 1)
    elif case == "hex":
        gx, gy = np.meshgrid(np.arange(30.0), np.arange(30.0))
        coords = np.stack([(gx + 0.5 * (gy % 2)).ravel(), (gy * 0.5).ravel()], 1)
    elif case == "cube":
        g = np.meshgrid(np.arange(10.0), np.arange(9.0), np.arange(11.0))
        coords = np.stack([a.ravel() for a in g], 1)
    elif case == "line":
        coords = np.arange(700.0)[:, None] * 0.5
    else:
        base = rs.rand(150, 2) * 12.0
        coords = base[rs.randint(0, len(base), 900)]
    coords = np.ascontiguousarray(coords[rs.permutation(len(coords))])
    A = G.build_knn_graph(coords, k=k)
    B = orc.knn_graph(coords, k)
    assert np.array_equal(A.indptr, B.indptr) and np.array_equal(A.indices, B.indices)


def test_knn_clustered_and_elongated():
    from flashdeconv_amd.utils import graph as G
    rs = np.random.RandomState(3)
    coords = np.concatenate([rs.randn(3000, 2) * 0.01, rs.rand(3000, 2) * np.array([5000.0, 2.0])])
    A = G.build_knn_graph(coords, k=6)
    B = orc.knn_graph_kdtree(coords, 6)
    assert np.array_equal(A.indptr, B.indptr) and np.array_equal(A.indices, B.indices)


def test_radius_and_grid_vs_oracle():
    from flashdeconv_amd.utils import graph as G
    rs = np.random.RandomState(8)
    coords = rs.rand(1500, 2) * 30
    A = G.build_radius_graph(coords, 1.7)
    B = orc.radius_graph(coords, 1.7)
    assert np.array_equal(A.indptr, B.indptr) and np.array_equal(A.indices, B.indices)
    jc = datagen.count_like(900, 30, 2, seed=4)[2]
    A = G.build_grid_graph(jc)
    B = orc.grid_graph(jc)
    assert np.array_equal(A.indptr, B.indptr) and np.array_equal(A.indices, B.indices)


@pytest.mark.parametrize("case", ["uniform2d", "clustered2d", "duplicates", "line1d", "cube3d", "lattice"])
def test_counting_order_is_the_stable_sort_order(case, monkeypatch):
    """Solver order (cells in Morton order, points of a cell by ascending index): the counting path (count per key, scan,
    rank within the cell) must give exactly the permutation of the stable radix sort it replaces (FDX_GRAPH_SORT=1), and the
    same graph.  Duplicated points and a regular lattice put many points on one key / exactly on cell borders."""
    import ctypes
    import torch
    from flashdeconv_amd import _lib
    rs = np.random.RandomState(11)
    n = 20000
    if case == "uniform2d":
        coords = rs.rand(n, 2) * 140.0
    elif case == "clustered2d":
        coords = np.concatenate([rs.randn(n // 2, 2) * 0.05, rs.rand(n - n // 2, 2) * np.array([900.0, 3.0])])
    elif case == "duplicates":
        base = rs.rand(n // 8, 2) * 50.0
        coords = base[rs.randint(0, len(base), n)]
    elif case == "line1d":
        coords = rs.rand(n, 1) * 1e4
    elif case == "cube3d":
        coords = rs.rand(n, 3) * 27.0
    else:
        side = int(np.sqrt(n))
        gx, gy = np.meshgrid(np.arange(side, dtype=np.float64), np.arange(side, dtype=np.float64))
        coords = np.stack([gx.ravel(), gy.ravel()], 1)
        coords = coords[rs.permutation(len(coords))]
    n, dim = coords.shape
    lib = _lib.load()
    cd = torch.as_tensor(np.ascontiguousarray(coords), device="cuda:0")

    def build():
        h = ctypes.c_void_p()
        _lib.check(lib.fdx_graph_build_dev(ctypes.c_void_p(cd.data_ptr()), n, dim, _lib.GRAPH_KNN, 6, 0.0, None, ctypes.byref(h)))
        g = _lib.Graph(h.value)
        perm = torch.empty(n, dtype=torch.int32, device="cuda:0")
        _lib.check(lib.fdx_graph_perm_dev(g.handle, ctypes.c_void_p(perm.data_ptr()), None))
        torch.cuda.synchronize()
        indptr, indices = g.to_csr_arrays()
        g.close()
        return perm.cpu().numpy(), indptr, indices

    p_count, ip_c, ix_c = build()
    monkeypatch.setenv("FDX_GRAPH_SORT", "1")
    p_sort, ip_s, ix_s = build()
    assert np.array_equal(np.sort(p_count), np.arange(n))
    assert np.array_equal(p_count, p_sort)
    assert np.array_equal(ip_c, ip_s) and np.array_equal(ix_c, ix_s)


def test_deferred_graph_build_and_its_rebuild_path(monkeypatch):
    """Whole-graph k-NN builds queue everything behind the bounding box and hand the counts over later (graph_meta_sync); if the
    ELL bound was too small the ELL is rebuilt with its exact size.  All three ways - deferred, deferred with a bound of one entry
    per row (always too small), built to the end - must give the reference graph, and the same fit."""
    from flashdeconv_amd.utils import graph as G
    from flashdeconv_amd import FlashDeconv
    rs = np.random.RandomState(21)
    coords = rs.rand(5000, 2) * 70.0
    want = orc.knn_graph_kdtree(coords, 6)
    Y, X = datagen.gaussian_raw(5000, 120, 5, seed=3)[:2]
    fits = []
    for env in ({}, {"FDX_GRAPH_WCAP": "1"}, {"FDX_GRAPH_SYNC": "1"}):
        for k in ("FDX_GRAPH_WCAP", "FDX_GRAPH_SYNC"):
            monkeypatch.delenv(k, raising=False)
        for k, v in env.items():
            monkeypatch.setenv(k, v)
        A = G.build_knn_graph(coords, k=6)
        assert np.array_equal(A.indptr, want.indptr) and np.array_equal(A.indices, want.indices), env
        m = FlashDeconv(sketch_dim=64, preprocess="raw", max_iter=15).fit(Y, X, coords)
        fits.append((m.beta_.copy(), m.info_["n_iterations"], m.lambda_used_))
    for b, it, lam in fits[1:]:
        assert np.array_equal(b, fits[0][0]) and it == fits[0][1] and lam == fits[0][2]


def test_graph_errors():
    from flashdeconv_amd.utils import graph as G
    with pytest.raises(ValueError, match="coords must be 2D"):
        G.build_knn_graph(np.zeros(5))
    with pytest.raises(ValueError, match="radius must be specified"):
        G.coords_to_adjacency(np.zeros((4, 2)), method="radius")
    with pytest.raises(ValueError, match="Unknown method"):
        G.coords_to_adjacency(np.zeros((4, 2)), method="nope")


# ------------------------------------------------------------------------------------------- sketch
def test_countsketch_tables_bit_exact():
    from flashdeconv_amd.core import sketching as S
    g = load_golden("omega_tables.npz")
    for (Gn, d, s) in g["cases"]:
        tag = f"G{Gn}_d{d}_s{s}"
        Om = S.build_countsketch_matrix(int(Gn), int(d), None, int(s)).tocsr()
        assert np.array_equal(Om.indices, g[tag + "_bucket"])
        assert np.array_equal(np.sign(Om.data).astype(np.int64), g[tag + "_sign"])
        np.testing.assert_allclose(Om.data, g[tag + "_data"], rtol=1e-15)


@pytest.mark.parametrize("dtype", [np.float64, np.float32])
@pytest.mark.parametrize("n,G,d", [(37, 90, 16), (300, 2000, 512), (65, 1001, 500), (10, 64, 1), (129, 5003, 1024)])
def test_project_to_sketch_vs_oracle(n, G, d, dtype):
    from flashdeconv_amd.core import sketching as S
    rs = np.random.RandomState(n)
    Yt = rs.randn(n, G).astype(dtype)
    Xt = rs.rand(7, G)
    lev = rs.rand(G)
    Ys, Xs, Om = S.sketch_data(Yt, Xt, sketch_dim=d, leverage_scores=lev, random_state=5)
    b, w = orc.countsketch_omega(G, d, lev, 5)
    assert np.array_equal(Om.tocsr().indices, b)
    want_Y, want_X = orc.project(Yt.astype(np.float64), Xt, b, w, d)
    assert rel_fro(Ys, want_Y) < 1e-13 and rel_fro(Xs, want_X) < 1e-13
    assert Ys.shape == (n, d) and Xs.shape == (7, d)


def test_project_linearity_and_general_omega():
    # reference tests/test_sketching.py:95-110 (linearity) + an Omega with several entries per gene
    from flashdeconv_amd.core import sketching as S
    rs = np.random.RandomState(0)
    Y1, Y2, X = rs.rand(40, 120), rs.rand(40, 120), rs.rand(3, 120)
    Om = sparse.random(120, 24, density=0.2, random_state=1, format="csr")
    a, _ = S.project_to_sketch(Y1, X, Om)
    b, _ = S.project_to_sketch(Y2, X, Om)
    c, xs = S.project_to_sketch(Y1 + Y2, X, Om)
    np.testing.assert_allclose(a + b, c, rtol=1e-12, atol=1e-12)
    np.testing.assert_allclose(c, (Y1 + Y2) @ Om.toarray(), rtol=1e-12, atol=1e-12)
    np.testing.assert_allclose(xs, X @ Om.toarray(), rtol=1e-12, atol=1e-12)


@pytest.mark.parametrize("dtype", [np.float32, np.float64])
@pytest.mark.parametrize("mode", ["raw", "log_cpm", "log_cpm_sparse"])
@pytest.mark.parametrize("G,d", [(2000, 512), (4000, 512), (5000, 1024), (4501, 640), (300, 64)])
def test_scatter_and_gather_sketch_kernels_agree(G, d, mode, dtype, monkeypatch):
    """The default scatter kernel (LDS atomics) and the gene-ordered gather kernels (FDX_SKETCH_GATHER=1) against the
    oracle and against each other, for row lengths on both sides of what the gather kernels can stage in LDS."""
    from flashdeconv_amd import _lib
    _lib.require_gpu()
    lib = _lib.load()
    rs = np.random.RandomState(G + d)
    n = 517
    Y = rs.poisson(0.7, size=(n, G)).astype(dtype)
    Y[3] = 0
    b, w = orc.countsketch_omega(G, d, rs.rand(G), 3)
    order = np.argsort(b, kind="stable")
    col_ptr = np.concatenate([[0], np.cumsum(np.bincount(b, minlength=d))]).astype(np.int64)
    gene_idx = order.astype(np.int32)
    weight = np.ascontiguousarray(w[order])
    code = {"raw": _lib.PRE_RAW, "log_cpm": _lib.PRE_LOG_CPM, "log_cpm_sparse": _lib.PRE_LOG_CPM_SPARSE}[mode]

    def run():
        out = np.empty((n, d))
        _lib.check(lib.fdx_sketch(Y.ctypes.data, _lib.FDX_F32 if dtype == np.float32 else _lib.FDX_F64, n, G,
                                  _lib.ptr_i64(col_ptr), _lib.ptr_i32(gene_idx), _lib.ptr_f64(weight), d, code, _lib.ptr_f64(out)))
        return out

    got = run()
    Y64 = Y.astype(np.float64)
    if mode == "raw":
        Yt = Y64
    elif mode == "log_cpm":
        Yt = np.log1p(Y64 / (Y64.sum(axis=1, keepdims=True) + 1e-10) * 1e4)
    else:
        lib_size = Y64.sum(axis=1, keepdims=True)
        lib_size[lib_size == 0] = 1.0
        Yt = np.log1p(Y64 / lib_size * 1e4)
    want, _ = orc.project(Yt, np.zeros((1, G)), b, w, d)
    assert rel_fro(got, want) < 1e-13
    monkeypatch.setenv("FDX_SKETCH_GATHER", "1")
    assert rel_fro(run(), got) < 1e-13
    monkeypatch.delenv("FDX_SKETCH_GATHER")
    assert np.array_equal(run(), got)                      # run-to-run identical bits


@pytest.mark.parametrize("kind", ["dense_f32", "dense_f64", "csr"])
def test_log1p_count_table_is_bit_identical(kind, monkeypatch):
    """Rows whose entries are all below 64 take log1p from a per-row table of the 64 possible values (device_math.h);
    the table holds the same function of the same argument, so switching it off (FDX_NO_LOG_TABLE=1) must not change a
    bit.  Mixed data: small counts, a row with a count of 70 (no table), non-integer rows (table miss per entry)."""
    from flashdeconv_amd import FlashDeconv
    rs = np.random.RandomState(12)
    n, G, K = 700, 900, 5
    Y = rs.poisson(0.8, size=(n, G)).astype(np.float64)
    Y[5, 17] = 70.0
    Y[9] = rs.rand(G) * 3.0
    Y[11, ::7] += 0.5
    X = np.exp(rs.randn(K, G) * 0.5)
    coords = rs.rand(n, 2) * 30
    Yin = {"dense_f32": Y.astype(np.float32), "dense_f64": Y, "csr": sparse.csr_matrix(Y)}[kind]
    kw = dict(sketch_dim=128, preprocess="log_cpm", n_hvg=G if kind != "csr" else 400, max_iter=10)
    a = FlashDeconv(**kw).fit(Yin, X, coords)
    monkeypatch.setenv("FDX_NO_LOG_TABLE", "1")
    b = FlashDeconv(**kw).fit(Yin, X, coords)
    assert np.array_equal(a.gene_idx_, b.gene_idx_) and np.array_equal(a.beta_, b.beta_)


def test_log_cpm_transform_accuracy():
    """The device log1p (fdlibm decomposition in sketch_kernels.cpp) against numpy.log1p, through fdx_sketch with an
    identity-like Omega (one gene per bucket, weight 1) so that Y_sketch IS the transformed matrix."""
    from flashdeconv_amd import _lib
    rs = np.random.RandomState(0)
    G = 700
    Y = np.zeros((64, G))
    Y[:32] = 10.0 ** rs.uniform(-12, 6, size=(32, G))          # 18 decades
    Y[32:48] = rs.poisson(3.0, size=(16, G))                   # counts incl. zeros
    Y[48:] = rs.rand(16, G) * 1e-3
    Y[5, :] = 0.0                                               # all-zero row
    col_ptr = np.arange(G + 1, dtype=np.int64)
    gene_idx = np.arange(G, dtype=np.int32)
    weight = np.ones(G)
    out = np.empty((64, G))
    _lib.require_gpu()
    for dtype, code in ((np.float64, _lib.FDX_F64), (np.float32, _lib.FDX_F32)):
        Yd = np.ascontiguousarray(Y.astype(dtype))
        _lib.check(_lib.load().fdx_sketch(Yd.ctypes.data, code, 64, G, _lib.ptr_i64(col_ptr), _lib.ptr_i32(gene_idx),
                                          _lib.ptr_f64(weight), G, _lib.PRE_LOG_CPM, _lib.ptr_f64(out)))
        Y64 = Yd.astype(np.float64)
        scale = (1.0 / (Y64.sum(axis=1, keepdims=True) + 1e-10)) * 1e4
        want = np.log1p(Y64 * scale)
        # the row sum is reduced in a different order on the device (~1e-16 relative on the scale)
        np.testing.assert_allclose(out, want, rtol=2e-14, atol=0)
        assert np.all(out[5] == 0.0)


@pytest.mark.parametrize("n,G,K,d,mode", [(1000, 2000, 30, 512, "log_cpm"), (333, 1996, 12, 512, "log_cpm"),
                                          (517, 2048, 7, 500, "log_cpm"), (400, 520, 20, 64, "log_cpm"),
                                          (1000, 1200, 17, 256, "log_cpm")])
def test_float32_log_path_matches_the_oracle_and_the_float64_chain(n, G, K, d, mode, monkeypatch):
    """float32 rows take a float32-class log1p in the tile kernel (csrc/tile_device.h: tile_log1p_f32) - the reference
    computes the log-CPM transform in float32 for float32 input (core/deconv.py:190-191 under numpy's dtype rules).
    Spot counts that are not whole tiles, gene counts that are not whole column blocks, rows with negative / NaN entries
    (those tiles take the general float64 log1p).  Against the oracle fed the same float32 array and against the float64
    chain (FDX_TILE_LOGV=0)."""
    import datagen
    import fdx_oracle as orc
    from flashdeconv_amd import FlashDeconv
    Y, X, coords, _ = datagen.count_like(n, G, K, seed=n + G)
    Y = Y.astype(np.float32)
    kw = dict(sketch_dim=d, preprocess=mode, n_hvg=G, max_iter=15, random_state=3)
    want = orc.fit(Y, X, coords, sketch_dim=d, preprocess_method=mode, n_hvg=G, max_iter=15, random_state=3)
    a = FlashDeconv(**kw).fit(Y, X, coords)
    assert a.info_["n_iterations"] == want["info"]["n_iterations"]
    assert rel_fro(a.beta_, want["beta"]) < 1e-5
    monkeypatch.setenv("FDX_TILE_LOGV", "0")
    b = FlashDeconv(**kw).fit(Y, X, coords)
    assert rel_fro(b.beta_, want["beta"]) < 1e-5
    assert rel_fro(a.beta_, b.beta_) < 1e-5
    monkeypatch.delenv("FDX_TILE_LOGV")
    # rows outside the fast range: their tiles are evaluated by the general float64 log1p, NaN where the reference has NaN
    Yb = Y.copy()
    Yb[7, 3] = -2.0
    Yb[n - 1, G - 1] = np.nan
    Yb[n // 2, 11] = -0.25
    a2 = FlashDeconv(**kw).fit(Yb, X, coords)
    monkeypatch.setenv("FDX_NO_FUSED", "1")                      # the two-kernel path: scatter sketch + contraction
    b2 = FlashDeconv(**kw).fit(Yb, X, coords)
    fin = np.isfinite(b2.beta_)
    assert np.array_equal(np.isfinite(a2.beta_), fin)
    assert rel_fro(a2.beta_[fin], b2.beta_[fin]) < 1e-5


@pytest.mark.parametrize("n,G,K,d,dtype", [(1000, 2000, 30, 512, np.float32), (333, 1996, 12, 512, np.float64),
                                          (4097, 2400, 9, 300, np.float32), (50, 520, 3, 64, np.float32), (400, 1001, 7, 128, np.float32)])
def test_raw_tile_kernel_and_two_kernel_path_give_the_same_bits(n, G, K, d, dtype, monkeypatch):
    """The one-kernel sketch -> H stage (tile kernel) against the two-kernel path (scatter sketch + split-d contraction,
    FDX_NO_FUSED=1): other kernels, the same sums per bucket - genes ascending inside a bucket either way.  The last shape has
    rows that are no whole number of 16-byte vectors: it takes the two-kernel path by itself."""
    import datagen
    from flashdeconv_amd import FlashDeconv
    Y, X, coords, _ = datagen.count_like(n, G, K, seed=n + K)
    Y = Y.astype(dtype)
    kw = dict(sketch_dim=d, preprocess="raw", n_hvg=G, max_iter=10, random_state=1)
    a = FlashDeconv(**kw).fit(Y, X, coords)
    monkeypatch.setenv("FDX_NO_FUSED", "1")
    c = FlashDeconv(**kw).fit(Y, X, coords)
    assert rel_fro(c.beta_, a.beta_) < 1e-13


def test_float32_log1p_of_the_tile_kernel_is_float32_accurate():
    """fdx_log1p_f32 = the device function the tile kernel applies to float32 rows: within 4 float32 ulp of log1p over the
    whole fast range, including arguments far below one ulp of 1 (where log(1 + x) alone would lose everything)."""
    from flashdeconv_amd import _lib
    lib = _lib.load()
    rs = np.random.RandomState(5)
    x = np.concatenate([
        np.float32(10.0) ** rs.uniform(-30, 4.49, 200000).astype(np.float32),     # log-uniform over the fast range
        rs.uniform(0, 3, 100000).astype(np.float32),                               # dense around the first binades
        np.float32(1.0) + np.arange(-50, 50, dtype=np.float32) * np.float32(2.0 ** -23),
        np.array([0.0, 1e-45, 1e-38, 2.0 ** -24, 2.0 ** -23, 1.0, 3.0, 7.0, 1e4, 31999.0], dtype=np.float32),
    ]).astype(np.float32)
    for scale in (1.0, 0.37, 2.5):
        y = (x / np.float32(scale)).astype(np.float32)
        out = np.empty_like(y)
        _lib.check(lib.fdx_log1p_f32(y.ctypes.data, float(np.float32(scale)), len(y), out.ctypes.data))
        arg = y.astype(np.float64) * float(np.float32(scale))
        assert np.all(np.isfinite(out[arg < 32000.0]))
        assert np.all(out[arg == 0.0] == 0.0)
        # the product y * scale is rounded to float32 first (as in the reference): one more half ulp of the argument - and
        # far more where that product is a float32 denormal, which has no 24 bits to give
        keep = (arg < 32000.0) & (arg >= 2.0 ** -120)
        want = np.log1p(arg[keep])
        got = out[keep].astype(np.float64)
        err = np.abs(got - want) / (np.abs(want) * 2.0 ** -23)
        assert err.max() < 4.0, (scale, err.max(), arg[keep][err.argmax()])
        tiny = (arg > 0) & (arg < 2.0 ** -120)
        assert np.all(np.abs(out[tiny].astype(np.float64) - arg[tiny]) <= 2.0 ** -149 + 1e-6 * arg[tiny])


@pytest.mark.parametrize("n,G,K,d,mode,dtype", [(700, 1200, 40, 1024, "raw", np.float32), (333, 3000, 50, 700, "log_cpm", np.float32),
                                                (500, 2000, 20, 1024, "raw", np.float64), (413, 1500, 64, 512, "log_cpm", np.float64),
                                                (600, 5000, 50, 1024, "raw", np.float32), (257, 900, 33, 96, "pearson", np.float32)])
def test_wide_tile_kernel_matches_the_two_kernel_path_and_the_oracle(n, G, K, d, mode, dtype, monkeypatch):
    """The wide form of the tile kernel (csrc/tile_kernels.cpp, AVL2: 33..64 cell types or sketch_dim above what the narrow
    wave split owns - BASELINE configs[4] is 50 types, d = 1024): MFMA A operands fetched per group from the operand-order
    copy of X_sketch, four type tiles reduced in two rounds.  Against the oracle and against the two-kernel path
    (FDX_NO_TILE_WIDE=1: scatter sketch + split-d contraction)."""
    import datagen
    import fdx_oracle as orc
    from flashdeconv_amd import FlashDeconv
    if mode == "raw":
        Y, X, coords, _ = datagen.gaussian_raw(n, G, K, seed=n + K)
    else:
        Y, X, coords, _ = datagen.count_like(n, G, K, seed=n + K)
    Y = Y.astype(dtype)
    kw = dict(sketch_dim=d, preprocess=mode, n_hvg=G, max_iter=12, random_state=5)
    want = orc.fit(Y, X, coords, sketch_dim=d, preprocess_method=mode, n_hvg=G, max_iter=12, random_state=5)
    a = FlashDeconv(**kw).fit(Y, X, coords)
    monkeypatch.setenv("FDX_NO_TILE_WIDE", "1")
    b = FlashDeconv(**kw).fit(Y, X, coords)
    assert a.info_["n_iterations"] == want["info"]["n_iterations"] == b.info_["n_iterations"]
    tol = 1e-8 if dtype == np.float64 or mode == "raw" else 1e-5       # float32 log-CPM: DESIGN.md section 4, deviation (ii)
    assert rel_fro(a.beta_, want["beta"]) < tol
    # float32 log-CPM: float32-class log1p in the tile kernel, float64 in the scatter kernel
    tol_paths = 1e-11 if dtype == np.float64 or mode != "log_cpm" else 1e-5
    assert rel_fro(a.beta_, b.beta_) < tol_paths
    assert rel_fro(a.proportions_, b.proportions_) < tol_paths
    if mode == "raw":
        monkeypatch.delenv("FDX_NO_TILE_WIDE")
        # a NaN and a negative entry: only the buckets of those genes may differ from the two-kernel path
        Yb = Y.copy()
        Yb[5, 17] = np.nan
        Yb[n - 1, G - 1] = -3.0
        a2 = FlashDeconv(**kw).fit(Yb, X, coords)
        monkeypatch.setenv("FDX_NO_TILE_WIDE", "1")
        b2 = FlashDeconv(**kw).fit(Yb, X, coords)
        fin = np.isfinite(b2.beta_)
        assert np.array_equal(np.isfinite(a2.beta_), fin)
        assert rel_fro(a2.beta_[fin], b2.beta_[fin]) < 1e-9


@pytest.mark.parametrize("n,dim,k", [(1500, 4, 6), (700, 6, 9), (300, 8, 3), (40, 5, 60)])
def test_knn_graph_in_more_than_three_dimensions(n, dim, k):
    """utils/graph.py:16-22, 60-81 with 4 to 8 coordinate columns (cKDTree takes any): exhaustive search on the device, solver
    order from the first three coordinates - the reference's adjacency index for index, and a fit on 4-D coordinates."""
    import fdx_oracle as orc
    from flashdeconv_amd.utils.graph import build_knn_graph
    rs = np.random.RandomState(n + dim)
    coords = rs.rand(n, dim) * 10.0
    want = orc.knn_graph_kdtree(coords, k)
    got = build_knn_graph(coords, k=k)
    assert got.nnz == want.nnz
    assert np.array_equal(got.indptr, want.indptr) and np.array_equal(got.indices, want.indices)


def test_fit_with_four_dimensional_coordinates():
    import datagen
    import fdx_oracle as orc
    from flashdeconv_amd import FlashDeconv
    n, G, K = 900, 400, 6
    Y, X, coords, _ = datagen.gaussian_raw(n, G, K, seed=21)
    c4 = np.concatenate([coords, np.random.RandomState(1).rand(n, 2) * coords.max()], axis=1)
    kw = dict(sketch_dim=64, preprocess="raw", n_hvg=G, max_iter=15)
    want = orc.fit(Y, X, c4, sketch_dim=64, preprocess_method="raw", n_hvg=G, max_iter=15)
    m = FlashDeconv(**kw).fit(Y, X, c4)
    assert m.info_["n_iterations"] == want["info"]["n_iterations"]
    assert rel_fro(m.beta_, want["beta"]) < 1e-8
    with pytest.raises(ValueError, match="radius / grid graphs are built for 1 to 3"):
        FlashDeconv(spatial_method="grid", **kw).fit(Y, X, c4)



@pytest.mark.parametrize("case", ["square", "square_big", "hex", "cube3d", "random2d", "random2d_ties", "random3d", "line", "duplicates",
                                  "heavy_duplicates", "binary", "shuffled", "tiny", "one_leaf", "million", "random_huge", "binary_huge",
                                  "ties_huge_3d", "line_huge"])
def test_device_built_ckdtree_has_scipys_index_array(case):
    """csrc/kdtree_build_dev.cpp: the restated cKDTree built ON THE DEVICE - every node of a level at once, libstdc++'s introselect
    replayed by a team of threads per node (partition passes in list form) - must leave scipy's index array, entry for entry: the
    order in which a leaf's points are tested decides among equidistant neighbours (utils/graph.py:60-63).  Against scipy itself:
    lattices, clouds with and without ties, 1-3 coordinates, few distinct values per axis (the split just above a node's minimum),
    sizes of every team class (one wave / 256 / 1024 threads per node), trees of one leaf."""
    import ctypes
    import torch
    from scipy.spatial import cKDTree
    from flashdeconv_amd import _lib
    lib = _lib.load()
    rs = np.random.RandomState(11)
    sq = np.stack(np.meshgrid(np.arange(40.0), np.arange(33.0), indexing="ij"), -1).reshape(-1, 2)
    coords = {
        "square": sq,
        "square_big": np.stack(np.meshgrid(np.arange(310.0), np.arange(300.0), indexing="ij"), -1).reshape(-1, 2),
        "hex": np.array([(c + 0.5 * (r & 1), r * np.sqrt(3.0) / 2.0) for r in range(30) for c in range(29)]),
        "cube3d": np.stack(np.meshgrid(*[np.arange(12.0)] * 3, indexing="ij"), -1).reshape(-1, 3),
        "random2d": rs.rand(70000, 2) * 60.0,
        "random2d_ties": np.round(rs.rand(30000, 2) * 25.0, 1),
        "random3d": rs.rand(5000, 3) * 9.0,
        "line": np.round(rs.rand(900, 1) * 50.0),
        "duplicates": np.concatenate([rs.rand(300, 2), rs.rand(100, 2).repeat(3, axis=0)]),
        "heavy_duplicates": rs.randint(0, 5, size=(3000, 2)).astype(np.float64),
        "binary": rs.randint(0, 2, size=(40000, 2)).astype(np.float64),
        "shuffled": sq[rs.permutation(len(sq))],
        "tiny": rs.rand(17, 2),
        "one_leaf": rs.rand(9, 3),
        "million": np.stack(np.meshgrid(np.arange(1000.0), np.arange(1000.0), indexing="ij"), -1).reshape(-1, 2),
        # above 100000 points a node's passes are launches of their own over 64 workgroups (kd_huge_*)
        "random_huge": rs.rand(300000, 2),
        "binary_huge": rs.randint(0, 2, size=(300000, 2)).astype(np.float64),
        "ties_huge_3d": np.round(rs.rand(250000, 3) * 20.0, 1),
        "line_huge": np.round(rs.rand(400000, 1) * 1000.0),
    }[case]
    coords = np.ascontiguousarray(coords, dtype=np.float64)
    n, dim = coords.shape
    cd = torch.from_numpy(coords).cuda()
    got = np.empty(n, dtype=np.int64)
    info = np.zeros(3, dtype=np.int32)
    _lib.check(lib.fdx_ckdtree_indices_dev(ctypes.c_void_p(cd.data_ptr()), n, dim, _lib.ptr_i64(got), _lib.ptr_i32(info), None))
    torch.cuda.synchronize()
    tree = cKDTree(coords)
    assert info[2] == 0, info
    assert np.array_equal(got, tree.indices), (case, int((got != tree.indices).sum()), info)


@pytest.mark.parametrize("case", ["square", "hex", "cube3d", "random2d_ties", "line", "duplicates", "heavy_duplicates",
                                  "square_big_rows"])
def test_device_ckdtree_queries_equal_the_host_restatement(case, monkeypatch):
    """fdx_graph_plan_set_ckdtree_lists_dev (csrc/kdtree_order.cpp): the k-nearest queries of the restated cKDTree answered by a
    device kernel (one lane per query: the host's node visits, heap operations and comparisons, in its order) against the same
    queries on the host's threads (FDX_KDTREE_HOST_QUERIES=1), which tests/test_host.py pins to scipy itself - list for list,
    on lattices (every k-th neighbour tied), rounded and duplicated points, 1 to 3 coordinates, all rows and a subset."""
    import ctypes
    import torch
    from flashdeconv_amd import _lib
    lib = _lib.load()
    rs = np.random.RandomState(7)
    sq = np.stack(np.meshgrid(np.arange(40.0), np.arange(33.0), indexing="ij"), -1).reshape(-1, 2)
    coords, k, rows = {
        "square": (sq, 6, None),
        "hex": (np.array([(c + 0.5 * (r & 1), r * np.sqrt(3.0) / 2.0) for r in range(30) for c in range(29)]), 6, None),
        "cube3d": (np.stack(np.meshgrid(*[np.arange(12.0)] * 3, indexing="ij"), -1).reshape(-1, 3), 8, None),
        "random2d_ties": (np.round(rs.rand(30000, 2) * 25.0, 1), 6, None),
        "line": (np.round(rs.rand(900, 1) * 50.0), 3, None),
        "duplicates": (np.concatenate([rs.rand(300, 2), rs.rand(100, 2).repeat(3, axis=0)]), 5, None),
        # a node's median is its minimum (scipy splits just above it): 25 distinct places, ~120 spots on each
        "heavy_duplicates": (rs.randint(0, 5, size=(3000, 2)).astype(np.float64), 6, None),
        "square_big_rows": (np.stack(np.meshgrid(np.arange(260.0), np.arange(250.0), indexing="ij"), -1).reshape(-1, 2), 6,
                            np.sort(rs.choice(65000, 9000, replace=False)).astype(np.int64)),
    }[case]
    coords = np.ascontiguousarray(coords, dtype=np.float64)
    n, dim = coords.shape
    kk = min(k, n - 1) + 1
    cd = torch.from_numpy(coords).cuda()

    def lists(host_queries):
        if host_queries:
            monkeypatch.setenv("FDX_KDTREE_HOST_QUERIES", "1")
        else:
            monkeypatch.delenv("FDX_KDTREE_HOST_QUERIES", raising=False)
        nbr = torch.full((n, kk), -7, dtype=torch.int32, device="cuda")
        cnt = torch.full((n,), -7, dtype=torch.int32, device="cuda")
        plan, h = ctypes.c_void_p(), ctypes.c_void_p()
        _lib.check(lib.fdx_graph_knn_lists_dev(ctypes.c_void_p(cd.data_ptr()), n, dim, k, 0, n, ctypes.c_void_p(nbr.data_ptr()),
                                               ctypes.c_void_p(cnt.data_ptr()), None, ctypes.byref(plan)))
        torch.cuda.synchronize()
        before = (nbr.clone(), cnt.clone())
        _lib.check(lib.fdx_graph_plan_set_ckdtree_lists_dev(plan, _lib.ptr_f64(coords), ctypes.c_void_p(cd.data_ptr()), n, dim,
                                                            None if rows is None else rows.ctypes.data, 0 if rows is None else len(rows),
                                                            ctypes.c_void_p(nbr.data_ptr()), ctypes.c_void_p(cnt.data_ptr()), None))
        perm = torch.empty(n, dtype=torch.int32, device="cuda")
        _lib.check(lib.fdx_graph_plan_order_dev(plan, ctypes.c_void_p(perm.data_ptr()), None, None))
        _lib.check(lib.fdx_graph_from_knn_lists_dev(plan, ctypes.c_void_p(nbr.data_ptr()), ctypes.c_void_p(cnt.data_ptr()), 0, n, None,
                                                    ctypes.byref(h)))
        _lib.Graph(h.value).close()
        torch.cuda.synchronize()
        return nbr.cpu().numpy(), cnt.cpu().numpy(), perm.cpu().numpy(), before

    nb_d, cn_d, perm_d, before = lists(False)
    nb_h, cn_h, perm_h, _ = lists(True)
    # (the tree itself was built on the device there - csrc/kdtree_build_dev.cpp, the default for 1-3 coordinates) ... and with
    # the tree built by the host's thread pool, queries on the device (fdx_kdtree_tune(2, 0)): the same lists
    lib.fdx_kdtree_tune(2, 0)
    try:
        nb_b, cn_b, perm_b, _ = lists(False)
    finally:
        lib.fdx_kdtree_tune(2, 1)
    assert np.array_equal(nb_b, nb_d) and np.array_equal(cn_b, cn_d) and np.array_equal(perm_b, perm_d)
    assert np.array_equal(perm_d, perm_h)
    assert np.array_equal(cn_d, cn_h) and np.array_equal(nb_d, nb_h)
    # the lists are the restated tree's: position p holds the answer for caller id perm[p], as positions, self dropped
    want = np.empty((n, kk), dtype=np.int64)
    _lib.check(lib.fdx_ckdtree_knn(_lib.ptr_f64(coords), n, dim, kk, want.ctypes.data, None))
    rank = np.empty(n, dtype=np.int64)
    rank[perm_d] = np.arange(n)
    check = np.arange(n) if rows is None else rows
    for i in check[:: max(1, len(check) // 500)]:
        exp = [rank[j] for j in want[i] if j >= 0 and j != i]
        p = rank[i]
        assert cn_d[p] == len(exp) and list(nb_d[p, :len(exp)]) == exp, (case, i)
    if rows is not None:                          # rows not asked for keep the device's own lists
        keep = np.setdiff1d(np.arange(n), rows)
        pk = rank[keep]
        assert np.array_equal(cn_d[pk], before[1].cpu().numpy()[pk]) and np.array_equal(nb_d[pk], before[0].cpu().numpy()[pk])


def test_threaded_host_transfers_round_trip_every_source_dtype():
    """csrc/host_transfer.cpp: caller (pageable) arrays of every numeric dtype into HBM as the float32 / float64 matrix the kernels
    stream - bits of numpy's astype, several 16 MB chunks so that the ring, the thread team and the tails are all in play - and
    back; integer counts above 2**24 take float64 (core/deconv.py:190-191, 229: the reference promotes on the host)."""
    import ctypes
    from flashdeconv_amd import _lib
    lib = _lib.load()
    rs = np.random.RandomState(3)
    n = 9_000_013                                            # odd length: the last chunk is partial
    for dt in (np.float32, np.float64, np.int8, np.uint8, np.int16, np.uint16, np.int32, np.uint32, np.int64, np.uint64, np.bool_):
        if np.dtype(dt).kind == "f":
            a = rs.randn(n).astype(dt)
        elif dt is np.bool_:
            a = rs.rand(n) < 0.3
        else:
            a = rs.randint(0, 100, size=n).astype(dt)
        ptr, code = _lib.upload_matrix(a)
        want = a.astype(np.float64 if dt is np.float64 else np.float32)
        assert code == (_lib.FDX_F64 if dt is np.float64 else _lib.FDX_F32)
        got = np.empty_like(want)
        _lib.download_bytes(got, ptr)
        lib.fdx_free(ptr)
        assert np.array_equal(got, want), dt
    big = rs.randint(0, 100, size=n).astype(np.int64)
    big[n // 2] = (1 << 24) + 1                              # float32 cannot hold it: the matrix goes up as float64
    ptr, code = _lib.upload_matrix(big)
    assert code == _lib.FDX_F64
    got = np.empty(n, dtype=np.float64)
    _lib.download_bytes(got, ptr)
    lib.fdx_free(ptr)
    assert np.array_equal(got, big.astype(np.float64))
    g = ctypes.c_double(0.0)
    _lib.check(lib.fdx_pinned_copy_rate(64 << 20, 1, ctypes.byref(g)))
    assert g.value > 1.0
