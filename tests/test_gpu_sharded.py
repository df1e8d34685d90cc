"""-m gpu tests of the device side of the spot-sharded path on ONE GPU.

(1) `ShardedFlashDeconv` with a 1-rank nccl group must reproduce `FlashDeconv` bit for bit.
(2) "Virtual ranks": the full graph is cut into W shards with fdx_graph_localize; each shard gets its own buffers and
    runs fdx_bcd_sweep_dev on its local graph; halos are moved with the send / recv lists the C ABI reports (in-process
    copies standing in for RCCL).  The assembled result must equal the single-GPU solve bit for bit, for the same
    number of iterations.  Together with tests/test_distributed_cpu.py (the real exchange loop over gloo) this covers
    every piece of the N > 1 path without needing N GPUs."""
import ctypes
import os

import numpy as np
import pytest

import datagen
from conftest import rel_fro

pytestmark = pytest.mark.gpu


def _st(torch):
    return ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)


@pytest.mark.parametrize("build", ["replicated", "sharded", "band", "pipeline"])
@pytest.mark.parametrize("W", [2, 3, 5])
def test_virtual_ranks_equal_single_gpu(W, build):
    import torch
    from flashdeconv_amd import FlashDeconv, _lib
    from flashdeconv_amd.core.sketching import countsketch_tables
    from flashdeconv_amd.distributed import HipBackend, diag_mean, shard_bounds
    from flashdeconv_amd.utils.genes import compute_leverage_scores
    lib = _lib.load()
    dev = torch.device("cuda", 0)
    n, G, K, d = 6000, 300, 12, 64
    Y, X, coords, _ = datagen.count_like(n, G, K, 0.1, 11)
    coords = coords + np.random.RandomState(0).rand(n, 2) * 1e-3
    # float32 on both sides: the shards hold float32 rows, and the transform's class follows the input type (float32 rows ->
    # float32-class log1p, as in the reference; integer counts would be transformed in float64)
    ref = FlashDeconv(sketch_dim=d, max_iter=60, tol=1e-5).fit(Y.astype(np.float32), X, coords)
    T = ref.info_["n_iterations"]

    cd = torch.from_numpy(np.ascontiguousarray(coords)).to(dev)
    bounds = shard_bounds(n, W)
    locals_ = None
    if build == "replicated":
        h = ctypes.c_void_p()
        _lib.check(lib.fdx_graph_build_dev(ctypes.c_void_p(cd.data_ptr()), n, 2, _lib.GRAPH_KNN, 6, 0.0, _st(torch), ctypes.byref(h)))
        fulls = [_lib.Graph(h.value)] * W
    elif build == "pipeline":
        # the queued pipeline (fdx_graph_shard_knn_dev): every rank's LOCAL graph in one go, nothing read back before the counts
        locals_, nnz_sum = [], 0
        for r in range(W):
            hl = ctypes.c_void_p()
            _lib.check(lib.fdx_graph_shard_knn_dev(ctypes.c_void_p(cd.data_ptr()), n, 2, 6, W, _lib.ptr_i64(bounds), r, _st(torch),
                                                   ctypes.byref(hl)))
            locals_.append(_lib.Graph(hl.value))
        for g in locals_:
            nnz, ties, far, over = ctypes.c_int64(0), ctypes.c_int64(0), ctypes.c_int32(0), ctypes.c_int32(0)
            _lib.check(lib.fdx_graph_shard_status(g.handle, ctypes.byref(nnz), ctypes.byref(ties), ctypes.byref(far), ctypes.byref(over)))
            assert far.value == 0 and over.value == 0 and ties.value == 0
            nnz_sum += nnz.value
        assert nnz_sum == ref.adjacency_.nnz
    else:
        # sharded build: every rank finds the k-NN lists of its own rows, the rows are all-gathered (here: copied between
        # the virtual ranks' buffers), every rank symmetrises its own rows only
        kk = 7
        nbrs = [torch.full((n, kk), -7, dtype=torch.int32, device=dev) for _ in range(W)]
        cnts = [torch.full((n,), -7, dtype=torch.int32, device=dev) for _ in range(W)]
        plans = []
        lists = lib.fdx_graph_knn_lists_band_dev if build == "band" else lib.fdx_graph_knn_lists_dev
        for r in range(W):
            pl = ctypes.c_void_p()
            _lib.check(lists(ctypes.c_void_p(cd.data_ptr()), n, 2, 6, int(bounds[r]), int(bounds[r + 1]),
                             ctypes.c_void_p(nbrs[r].data_ptr()), ctypes.c_void_p(cnts[r].data_ptr()), _st(torch), ctypes.byref(pl)))
            plans.append(pl)
        if build == "band":
            # band recompute: nothing is exchanged - every rank found the lists of the rows that can point at its own rows itself
            for r in range(W):
                a, b = int(bounds[r]), int(bounds[r + 1])
                n_lists = int((cnts[r] > 0).sum())
                assert b - a <= n_lists < (b - a) + 0.35 * n, (r, n_lists)          # own rows + a band, not the whole graph
        for r in range(W):
            for q in range(W):
                if q != r and build != "band":
                    a, b = int(bounds[q]), int(bounds[q + 1])
                    nbrs[r][a:b] = nbrs[q][a:b]
                    cnts[r][a:b] = cnts[q][a:b]
        fulls, nnz_sum = [], 0
        for r in range(W):
            assert int(cnts[r].min()) >= 0 and (build == "band" or int(nbrs[r].min()) >= -1)    # band: rows without a list are never read
            h = ctypes.c_void_p()
            _lib.check(lib.fdx_graph_from_knn_lists_dev(plans[r], ctypes.c_void_p(nbrs[r].data_ptr()),
                                                        ctypes.c_void_p(cnts[r].data_ptr()), int(bounds[r]), int(bounds[r + 1]),
                                                        _st(torch), ctypes.byref(h)))
            fulls.append(_lib.Graph(h.value))
            nnz_sum += fulls[-1].info()[1]
            assert fulls[-1].knn_far() == 0                  # uniform density: no walk left its 3 x 3 block of cells
        assert nnz_sum == ref.adjacency_.nnz
    lev = compute_leverage_scores(X)
    bucket, weight = countsketch_tables(G, d, lev, 0)
    b32 = np.ascontiguousarray(bucket, dtype=np.int32)
    Yt = torch.from_numpy(Y.astype(np.float32)).to(dev)
    ranks = []
    yty = 0.0
    for r in range(W):
        if locals_ is not None:
            g = locals_[r]
        else:
            hl = ctypes.c_void_p()
            _lib.check(lib.fdx_graph_localize(fulls[r].handle, W, _lib.ptr_i64(bounds), r, _st(torch), ctypes.byref(hl)))
            g = _lib.Graph(hl.value)
        n_own = int(bounds[r + 1] - bounds[r])
        perm = torch.empty(max(n_own, 1), dtype=torch.int32, device=dev)
        _lib.check(lib.fdx_graph_perm_dev(g.handle, ctypes.c_void_p(perm.data_ptr()), _st(torch)))
        own = perm[:n_own].long()
        nh = ctypes.c_int64(0)
        sc, rc = np.zeros(W, dtype=np.int32), np.zeros(W, dtype=np.int32)
        _lib.check(lib.fdx_graph_halo_info(g.handle, ctypes.byref(nh), _lib.ptr_i32(sc), _lib.ptr_i32(rc)))
        sidx = torch.empty(max(int(sc.sum()), 1), dtype=torch.int32, device=dev)
        _lib.check(lib.fdx_graph_send_indices_dev(g.handle, ctypes.c_void_p(sidx.data_ptr()), _st(torch)))
        n_total = n_own + int(nh.value)
        ld = ((n_total + 1 + 63) // 64) * 64
        Hm = torch.zeros((K, ld), dtype=torch.float64, device=dev)
        XtX = torch.empty((K, K), dtype=torch.float64, device=dev)
        XtX_h = np.empty((K, K))
        part = ctypes.c_double(0.0)
        Yown = Yt[own].contiguous()
        _lib.check(lib.fdx_prepare_dev(ctypes.c_void_p(Yown.data_ptr()), _lib.FDX_F32, n_own, G, G, None, _lib.ptr_f64(np.ascontiguousarray(X)), K,
                                       _lib.ptr_i32(b32), _lib.ptr_f64(weight), _lib.ptr_f64(weight), d, _lib.PRE_LOG_CPM,
                                       _lib.PRE_LOG_CPM, ctypes.c_void_p(Hm.data_ptr()), ld, ctypes.c_void_p(XtX.data_ptr()),
                                       _lib.ptr_f64(XtX_h), ctypes.byref(part), _st(torch)))
        yty += part.value
        be = HipBackend(g, Hm, ld, XtX, K)
        beta = [torch.zeros((K, ld), dtype=torch.float64, device=dev) for _ in range(2)]
        be.init_beta(beta[0], n_total)
        ranks.append(dict(g=g, own=own, n_own=n_own, n_total=n_total, ld=ld, be=be, beta=beta, sc=sc, rc=rc,
                          sidx=sidx[:int(sc.sum())].long(), XtX_h=XtX_h,
                          stats=torch.zeros((60, 128), dtype=torch.float64, device=dev),
                          rel=torch.zeros(60, dtype=torch.float64, device=dev)))
        assert sum(rc) == nh.value
    # symmetric bookkeeping: what r sends to q is what q expects from r
    for r in range(W):
        for q in range(W):
            assert ranks[r]["sc"][q] == ranks[q]["rc"][r]
    lam, rho_eff = ref.lambda_used_, 0.01 * diag_mean(ranks[0]["XtX_h"])
    n_iter, conv = 0, False
    for it in range(60):
        for R in ranks:
            R["be"].sweep(it, R["beta"][it & 1], R["beta"][(it + 1) & 1], lam, rho_eff, 1e-5, R["stats"], R["rel"])
        for r, R in enumerate(ranks):                      # halo exchange (in-process stand-in for RCCL send/recv)
            soff = np.concatenate([[0], np.cumsum(R["sc"])])
            for q, Q in enumerate(ranks):
                if q == r or R["sc"][q] == 0:
                    continue
                rows = R["beta"][(it + 1) & 1].index_select(1, R["sidx"][soff[q]:soff[q + 1]])
                roff = np.concatenate([[0], np.cumsum(Q["rc"])])
                lo = Q["n_own"] + roff[r]
                Q["beta"][(it + 1) & 1][:, lo:lo + R["sc"][q]] = rows
        mx = torch.stack([R["stats"][it] for R in ranks]).max(dim=0).values      # all-reduce(MAX)
        for R in ranks:
            R["stats"][it] = mx
        ranks[0]["be"].fold(ranks[0]["stats"], ranks[0]["rel"], it)
        rc_it = float(ranks[0]["rel"][it].item())
        n_iter = it + 1
        if rc_it < 1e-5:
            conv = True
            break
    assert n_iter == T and conv == ref.info_["converged"]
    beta = torch.zeros((n, K), dtype=torch.float64, device=dev)
    for R in ranks:
        out = torch.empty((R["n_own"], K), dtype=torch.float64, device=dev)
        _lib.check(lib.fdx_normalize_dev(ctypes.c_void_p(R["beta"][n_iter & 1].data_ptr()), R["ld"], R["n_own"], K,
                                         ctypes.c_void_p(out.data_ptr()), None, _st(torch)))
        beta[R["own"]] = out
    torch.cuda.synchronize()
    assert np.array_equal(beta.cpu().numpy(), ref.beta_)


@pytest.mark.parametrize("dim,n,k,W", [(1, 3000, 4, 3), (2, 20000, 6, 4), (3, 9000, 6, 3), (2, 5000, 12, 5)])
def test_band_recompute_rows_are_the_references_rows(dim, n, k, W):
    """Every rank's own rows from the band recompute (no exchange; the shard's binning lays out its own neighbourhood only),
    exported in the caller's labels and put together, are the reference's adjacency (utils/graph.py:25-83) index for index -
    1-, 2- and 3-dimensional coordinates, k up to 12."""
    import torch
    import fdx_oracle as orc
    from scipy import sparse
    from flashdeconv_amd import _lib
    from flashdeconv_amd.distributed import shard_bounds
    lib = _lib.load()
    dev = torch.device("cuda", 0)
    rs = np.random.RandomState(dim * 100 + k)
    pts = rs.rand(n, dim) * (n ** (1.0 / dim))
    want = orc.knn_graph_kdtree(pts, k)
    cd = torch.from_numpy(np.ascontiguousarray(pts)).to(dev)
    bounds = shard_bounds(n, W)
    kk = min(k, n - 1) + 1
    rows, cols = [], []
    for r in range(W):
        nbr = torch.full((n, kk), -7, dtype=torch.int32, device=dev)
        cnt = torch.full((n,), -7, dtype=torch.int32, device=dev)
        pl, h = ctypes.c_void_p(), ctypes.c_void_p()
        _lib.check(lib.fdx_graph_knn_lists_band_dev(ctypes.c_void_p(cd.data_ptr()), n, dim, k, int(bounds[r]), int(bounds[r + 1]),
                                                    ctypes.c_void_p(nbr.data_ptr()), ctypes.c_void_p(cnt.data_ptr()), _st(torch), ctypes.byref(pl)))
        _lib.check(lib.fdx_graph_from_knn_lists_dev(pl, ctypes.c_void_p(nbr.data_ptr()), ctypes.c_void_p(cnt.data_ptr()),
                                                    int(bounds[r]), int(bounds[r + 1]), _st(torch), ctypes.byref(h)))
        g = _lib.Graph(h.value)
        assert g.knn_far() == 0
        indptr, indices = g.to_csr_arrays()                    # all n rows in the caller's labels; rows of other ranks are empty
        deg = np.diff(indptr)
        rows.append(np.repeat(np.arange(n), deg))
        cols.append(indices)
        g.close()
    got = sparse.csr_matrix((np.ones(sum(len(c) for c in cols)), (np.concatenate(rows), np.concatenate(cols))), shape=(n, n))
    got.sort_indices()
    assert got.nnz == want.nnz
    assert np.array_equal(got.indptr, want.indptr) and np.array_equal(got.indices, want.indices)


def test_band_recompute_reports_walks_that_leave_their_block():
    """Very uneven density (two tight clusters and a few stragglers far away): the stragglers' k-NN walks leave the 3 x 3 block of
    grid cells, the rank that owns them says so (fdx_graph_knn_far), and the driver then builds by exchanging the lists - whose
    result is the reference's graph."""
    import torch
    import fdx_oracle as orc
    from flashdeconv_amd import _lib
    from flashdeconv_amd.distributed import shard_bounds
    lib = _lib.load()
    dev = torch.device("cuda", 0)
    rs = np.random.RandomState(3)
    pts = np.concatenate([rs.rand(1500, 2) * 0.05, rs.rand(1500, 2) * 0.05 + 10.0, rs.rand(12, 2) * 10.0])
    n = len(pts)
    cd = torch.from_numpy(np.ascontiguousarray(pts)).to(dev)
    W, kk = 3, 7
    bounds = shard_bounds(n, W)
    far, graphs = 0, []
    nbrs = [torch.full((n, kk), -7, dtype=torch.int32, device=dev) for _ in range(W)]
    cnts = [torch.full((n,), -7, dtype=torch.int32, device=dev) for _ in range(W)]
    for r in range(W):
        pl, h = ctypes.c_void_p(), ctypes.c_void_p()
        _lib.check(lib.fdx_graph_knn_lists_band_dev(ctypes.c_void_p(cd.data_ptr()), n, 2, 6, int(bounds[r]), int(bounds[r + 1]),
                                                    ctypes.c_void_p(nbrs[r].data_ptr()), ctypes.c_void_p(cnts[r].data_ptr()), _st(torch),
                                                    ctypes.byref(pl)))
        _lib.check(lib.fdx_graph_from_knn_lists_dev(pl, ctypes.c_void_p(nbrs[r].data_ptr()), ctypes.c_void_p(cnts[r].data_ptr()),
                                                    int(bounds[r]), int(bounds[r + 1]), _st(torch), ctypes.byref(h)))
        g = _lib.Graph(h.value)
        far += g.knn_far()
        g.close()
    assert far > 0
    # the exchange route on the same points
    plans = []
    for r in range(W):
        pl = ctypes.c_void_p()
        _lib.check(lib.fdx_graph_knn_lists_dev(ctypes.c_void_p(cd.data_ptr()), n, 2, 6, int(bounds[r]), int(bounds[r + 1]),
                                               ctypes.c_void_p(nbrs[r].data_ptr()), ctypes.c_void_p(cnts[r].data_ptr()), _st(torch),
                                               ctypes.byref(pl)))
        plans.append(pl)
    for r in range(W):
        for q in range(W):
            if q != r:
                a, b = int(bounds[q]), int(bounds[q + 1])
                nbrs[r][a:b] = nbrs[q][a:b]
                cnts[r][a:b] = cnts[q][a:b]
    nnz = 0
    for r in range(W):
        h = ctypes.c_void_p()
        _lib.check(lib.fdx_graph_from_knn_lists_dev(plans[r], ctypes.c_void_p(nbrs[r].data_ptr()), ctypes.c_void_p(cnts[r].data_ptr()),
                                                    int(bounds[r]), int(bounds[r + 1]), _st(torch), ctypes.byref(h)))
        g = _lib.Graph(h.value)
        nnz += g.info()[1]
        g.close()
    assert nnz == orc.knn_graph_kdtree(pts, 6).nnz


def test_sharded_class_world1_equals_flashdeconv():
    import torch
    import torch.distributed as dist
    from flashdeconv_amd import FlashDeconv
    from flashdeconv_amd.distributed import ShardedFlashDeconv
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29731")
    dev = torch.device("cuda", 0)
    torch.cuda.set_device(dev)
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
    try:
        n, G, K = 5000, 400, 9
        Y, X, coords, _ = datagen.gaussian_raw(n, G, K, seed=2)
        ref = FlashDeconv(sketch_dim=128, preprocess="raw", n_hvg=G).fit(Y, X, coords)
        m = ShardedFlashDeconv(sketch_dim=128, preprocess="raw", n_hvg=G)
        own = m.plan(torch.from_numpy(coords).to(dev))
        P = m.fit_transform(torch.from_numpy(Y).to(dev)[own], X)
        got = np.zeros((n, K))
        got[own.cpu().numpy()] = P.cpu().numpy()
        assert m.info_["n_iterations"] == ref.info_["n_iterations"] and m.info_["converged"] == ref.info_["converged"]
        assert np.array_equal(got, ref.proportions_)
        np.testing.assert_allclose(m.info_["final_objective"], ref.info_["final_objective"], rtol=1e-12)
        np.testing.assert_allclose(m.lambda_used_, ref.lambda_used_, rtol=1e-14)
        # 65-96 cell types: the next instantiated sweep size with all-zero pad types, on both paths (fdx_solver_padded_k)
        for K7 in (70, 96):
            Y7, X7, c7, _ = datagen.gaussian_raw(3000, 500, K7, seed=K7)
            ref7 = FlashDeconv(sketch_dim=128, preprocess="raw", n_hvg=500, max_iter=12).fit(Y7, X7, c7)
            m7 = ShardedFlashDeconv(sketch_dim=128, preprocess="raw", n_hvg=500, max_iter=12)
            own7 = m7.plan(torch.from_numpy(c7).to(dev))
            P7 = m7.fit_transform(torch.from_numpy(Y7).to(dev)[own7], X7)
            assert P7.shape == (3000, K7) and m7.beta_.shape == (3000, K7)
            got7 = np.zeros((3000, K7))
            got7[own7.cpu().numpy()] = P7.cpu().numpy()
            assert m7.info_["n_iterations"] == ref7.info_["n_iterations"]
            np.testing.assert_allclose(got7, ref7.proportions_, rtol=1e-9, atol=1e-13)
            np.testing.assert_allclose(m7.info_["final_objective"], ref7.info_["final_objective"], rtol=1e-10)
        # above 96 types: the LDS-resident sweep over the whole shard per iteration (no boundary / interior split), as on one GPU
        for K9 in (100, 120):
            Y9, X9, c9, _ = datagen.gaussian_raw(1500, 400, K9, seed=K9)
            ref9 = FlashDeconv(sketch_dim=128, preprocess="raw", n_hvg=400, max_iter=8).fit(Y9, X9, c9)
            m9 = ShardedFlashDeconv(sketch_dim=128, preprocess="raw", n_hvg=400, max_iter=8)
            own9 = m9.plan(torch.from_numpy(c9).to(dev))
            P9 = m9.fit_transform(torch.from_numpy(Y9).to(dev)[own9], X9)
            got9 = np.zeros((1500, K9))
            got9[own9.cpu().numpy()] = P9.cpu().numpy()
            assert m9.info_["n_iterations"] == ref9.info_["n_iterations"]
            assert np.array_equal(got9, ref9.proportions_), K9
            np.testing.assert_allclose(m9.info_["final_objective"], ref9.info_["final_objective"], rtol=1e-10)
        # gene selection active (G > n_hvg): statistics reduced over the shards, same genes, same fit
        Yc, Xc, cc, _ = datagen.count_like(3000, 900, 6, 0.1, 8)
        kw = dict(sketch_dim=64, preprocess="log_cpm", n_hvg=250, n_markers_per_type=10, max_iter=20)
        ref2 = FlashDeconv(**kw).fit(Yc.astype(np.float32), Xc, cc)
        m2 = ShardedFlashDeconv(**kw)
        own2 = m2.plan(torch.from_numpy(cc).to(dev))
        P2 = m2.fit_transform(torch.from_numpy(Yc.astype(np.float32)).to(dev)[own2], Xc)
        assert np.array_equal(m2.gene_idx_, ref2.gene_idx_) and len(m2.gene_idx_) < 900
        got2 = np.zeros((3000, 6))
        got2[own2.cpu().numpy()] = P2.cpu().numpy()
        assert m2.info_["n_iterations"] == ref2.info_["n_iterations"]
        np.testing.assert_allclose(got2, ref2.proportions_, rtol=1e-9, atol=1e-12)
        # integer counts: both drivers store them as float32 and keep the float64 transform chain (numpy promotes integer
        # input, core/deconv.py:190-191) - same bits, and the oracle's float64 fit within 1e-8
        kw_i = dict(sketch_dim=64, preprocess="log_cpm", n_hvg=900, max_iter=15)
        Yi = torch.from_numpy(Yc.astype(np.int32)).to(dev)
        ref_i = FlashDeconv(**kw_i).fit(Yi, Xc, cc)
        ref_f = FlashDeconv(**kw_i).fit(Yc.astype(np.float64), Xc, cc)
        m_i = ShardedFlashDeconv(**kw_i)
        own_i = m_i.plan(torch.from_numpy(cc).to(dev))
        P_i = m_i.fit_transform(Yi[own_i], Xc)
        got_i = np.zeros((3000, 6))
        got_i[own_i.cpu().numpy()] = P_i.cpu().numpy()
        assert np.array_equal(got_i, ref_i.proportions_)
        assert np.linalg.norm(got_i - ref_f.proportions_) / np.linalg.norm(ref_f.proportions_) < 1e-8
        for dt, cap in ((torch.uint16, 60000), (torch.int64, 1 << 40)):    # uint16: torch has no max() for it -> float64 storage
            Yu = np.minimum(Yc, cap)
            want_u = FlashDeconv(**kw_i).fit(Yu.astype(np.float64), Xc, cc).proportions_
            Yt_u = torch.from_numpy(Yu.astype(np.int64)).to(dev)
            mu = ShardedFlashDeconv(**kw_i)
            own_u = mu.plan(torch.from_numpy(cc).to(dev))
            Pu = mu.fit_transform(Yt_u[own_u].to(dt), Xc)
            got_u = np.zeros((3000, 6))
            got_u[own_u.cpu().numpy()] = Pu.cpu().numpy()
            assert np.linalg.norm(got_u - want_u) / np.linalg.norm(want_u) < 1e-8, dt
            ref_u = FlashDeconv(**kw_i).fit(Yt_u.to(dt), Xc, cc)
            assert np.linalg.norm(ref_u.proportions_ - want_u) / np.linalg.norm(want_u) < 1e-8, dt
        # the same shard handed over as CSR rows: statistics, selection and the sketch all read the sparse rows
        import scipy.sparse as sp
        for pre in ("log_cpm", "pearson", "raw"):
            kw3 = dict(kw, preprocess=pre)
            ref3 = FlashDeconv(**kw3).fit(sp.csr_matrix(Yc), Xc, cc)
            m3 = ShardedFlashDeconv(**kw3)
            own3 = m3.plan(torch.from_numpy(cc).to(dev))
            S = sp.csr_matrix(Yc[own3.cpu().numpy()].astype(np.float32))
            Yt = torch.sparse_csr_tensor(torch.from_numpy(S.indptr.astype(np.int64)), torch.from_numpy(S.indices.astype(np.int64)),
                                         torch.from_numpy(S.data), size=S.shape).to(dev)
            P3 = m3.fit_transform(Yt, Xc)
            assert np.array_equal(m3.gene_idx_, ref3.gene_idx_), pre
            got3 = np.zeros((3000, 6))
            got3[own3.cpu().numpy()] = P3.cpu().numpy()
            assert m3.info_["n_iterations"] == ref3.info_["n_iterations"], pre
            np.testing.assert_allclose(got3, ref3.proportions_, rtol=1e-9, atol=1e-12, err_msg=pre)
            np.testing.assert_allclose(m3.info_["final_objective"], ref3.info_["final_objective"], rtol=1e-10, err_msg=pre)
    finally:
        dist.destroy_process_group()


def _pipeline_locals(torch, coords_dev, W, k=6):
    """Every rank's local graph by the queued pipeline (fdx_graph_shard_knn_dev); None when a rank owns nothing, a walk left its
    block or a bound was too small (the drivers then take the stepwise / exchange routes)."""
    from flashdeconv_amd import _lib
    from flashdeconv_amd.distributed import shard_bounds
    lib = _lib.load()
    n, dim = coords_dev.shape
    bounds = shard_bounds(n, W)
    if any(bounds[r + 1] <= bounds[r] for r in range(W)):
        return None
    out = []
    for r in range(W):
        hl = ctypes.c_void_p()
        _lib.check(lib.fdx_graph_shard_knn_dev(ctypes.c_void_p(coords_dev.data_ptr()), n, dim, k, W, _lib.ptr_i64(bounds), r, _st(torch),
                                               ctypes.byref(hl)))
        out.append(_lib.Graph(hl.value))
    bad = False
    for g in out:
        nnz, ties, far, over = ctypes.c_int64(0), ctypes.c_int64(0), ctypes.c_int32(0), ctypes.c_int32(0)
        _lib.check(lib.fdx_graph_shard_status(g.handle, ctypes.byref(nnz), ctypes.byref(ties), ctypes.byref(far), ctypes.byref(over)))
        bad = bad or far.value != 0 or over.value != 0
    if bad:
        for g in out:
            g.close()
        return None
    return out


def _native_shards(torch, coords_dev, Y_dev, X, W, d, K, mode, random_state=0, fulls=None, locals_=None):
    """Cut the problem into W shards (replicated graph build - or the ranks' own full-size graphs `fulls`, e.g. from the band
    recompute - + fdx_graph_localize; or the ranks' local graphs `locals_` of the queued pipeline), prepare H / XtX per shard."""
    from flashdeconv_amd import _lib
    from flashdeconv_amd.core.sketching import countsketch_tables
    from flashdeconv_amd.distributed import shard_bounds
    from flashdeconv_amd.utils.genes import compute_leverage_scores
    lib = _lib.load()
    dev = coords_dev.device
    n, G = Y_dev.shape
    bounds = shard_bounds(n, W)
    h = ctypes.c_void_p()
    _lib.check(lib.fdx_graph_build_dev(ctypes.c_void_p(coords_dev.data_ptr()), n, 2, _lib.GRAPH_KNN, 6, 0.0, _st(torch), ctypes.byref(h)))
    full = _lib.Graph(h.value)
    lev = compute_leverage_scores(X)
    bucket, weight = countsketch_tables(G, d, lev, random_state)
    b32 = np.ascontiguousarray(bucket, dtype=np.int32)
    Xc = np.ascontiguousarray(X, dtype=np.float64)
    shards = []
    for r in range(W):
        if locals_ is not None:
            g = locals_[r]
        else:
            hl = ctypes.c_void_p()
            _lib.check(lib.fdx_graph_localize((fulls[r] if fulls else full).handle, W, _lib.ptr_i64(bounds), r, _st(torch), ctypes.byref(hl)))
            g = _lib.Graph(hl.value)
        n_own = int(bounds[r + 1] - bounds[r])
        perm = torch.empty(max(n_own, 1), dtype=torch.int32, device=dev)
        _lib.check(lib.fdx_graph_perm_dev(g.handle, ctypes.c_void_p(perm.data_ptr()), _st(torch)))
        own = perm[:n_own].long()
        nh = ctypes.c_int64(0)
        _lib.check(lib.fdx_graph_halo_info(g.handle, ctypes.byref(nh), None, None))
        n_total = n_own + int(nh.value)
        ld = ((n_total + 1 + 63) // 64) * 64
        Hm = torch.zeros((K, ld), dtype=torch.float64, device=dev)
        XtX = torch.empty((K, K), dtype=torch.float64, device=dev)
        XtX_h = np.empty((K, K))
        part = ctypes.c_double(0.0)
        # rows gathered through the shard's row map: Y stays whole, as on a rank that was handed only its own rows
        own32 = own.to(torch.int32).contiguous()
        code = _lib.FDX_F32 if Y_dev.dtype == torch.float32 else _lib.FDX_F64
        _lib.check(lib.fdx_prepare_dev(ctypes.c_void_p(Y_dev.data_ptr()), code, n_own, G, G, ctypes.c_void_p(own32.data_ptr()),
                                       _lib.ptr_f64(Xc), K, _lib.ptr_i32(b32), _lib.ptr_f64(weight), _lib.ptr_f64(weight), d, mode,
                                       mode, ctypes.c_void_p(Hm.data_ptr()), ld, ctypes.c_void_p(XtX.data_ptr()),
                                       _lib.ptr_f64(XtX_h), ctypes.byref(part), _st(torch)))
        shards.append(dict(g=g, own=own, n_own=n_own, ld=ld, H=Hm, XtX=XtX, XtX_h=XtX_h,
                           beta=[torch.empty((K, ld), dtype=torch.float64, device=dev) for _ in range(2)]))
    return full, shards


def _run_native_threads(torch, shards, K, lam, rho_eff, tol, max_iter):
    """fdx_sharded_solve_dev on every shard at once: one host thread and one stream per rank."""
    import threading
    from flashdeconv_amd import _lib
    lib = _lib.load()
    W = len(shards)
    world = ctypes.c_void_p()
    _lib.check(lib.fdx_local_world_create(W, ctypes.byref(world)))
    torch.cuda.synchronize()
    results, errors = [None] * W, []

    def work(r):
        try:
            torch.cuda.set_device(0)
            S = shards[r]
            comm = ctypes.c_void_p()
            _lib.check(lib.fdx_comm_init_local(world, r, ctypes.byref(comm)))
            stream = torch.cuda.Stream()
            info = _lib.SolveInfo()
            which = ctypes.c_int32(0)
            rel = np.zeros(max(max_iter, 1))
            rc = lib.fdx_sharded_solve_dev(comm, S["g"].handle, ctypes.c_void_p(S["H"].data_ptr()), S["ld"],
                                           ctypes.c_void_p(S["XtX"].data_ptr()), K, lam, rho_eff, tol, max_iter,
                                           ctypes.c_void_p(S["beta"][0].data_ptr()), ctypes.c_void_p(S["beta"][1].data_ptr()), S["ld"],
                                           ctypes.byref(info), _lib.ptr_f64(rel), ctypes.byref(which),
                                           ctypes.c_void_p(stream.cuda_stream))
            if rc != 0:
                errors.append((r, lib.fdx_last_error()))
            results[r] = (int(info.n_iterations), bool(info.converged), float(info.final_change), which.value, rel)
            lib.fdx_comm_destroy(comm)
        except Exception as e:                                   # noqa: BLE001 - reported below, the other ranks would hang
            errors.append((r, repr(e)))

    threads = [threading.Thread(target=work, args=(r,)) for r in range(W)]
    for t in threads:
        t.start()
    for t in threads:
        t.join(timeout=90)
    assert not any(t.is_alive() for t in threads), "a rank thread hangs"
    lib.fdx_local_world_destroy(world)
    assert not errors, errors
    return results


def _assemble(torch, shards, results, n, K):
    from flashdeconv_amd import _lib
    lib = _lib.load()
    dev = shards[0]["H"].device
    beta = torch.zeros((n, K), dtype=torch.float64, device=dev)
    for S, res in zip(shards, results):
        out = torch.empty((S["n_own"], K), dtype=torch.float64, device=dev)
        _lib.check(lib.fdx_normalize_dev(ctypes.c_void_p(S["beta"][res[3]].data_ptr()), S["ld"], S["n_own"], K,
                                         ctypes.c_void_p(out.data_ptr()), None, _st(torch)))
        beta[S["own"]] = out
    torch.cuda.synchronize()
    return beta


@pytest.mark.parametrize("overlap", [True, False])
@pytest.mark.parametrize("W", [2, 3, 5])
def test_native_loop_thread_ranks_equal_single_gpu(W, overlap, monkeypatch):
    import torch
    from flashdeconv_amd import FlashDeconv, _lib
    from flashdeconv_amd.distributed import diag_mean
    if not overlap:
        monkeypatch.setenv("FDX_NO_OVERLAP", "1")               # one sweep launch per iteration, halo on the compute stream
    else:
        monkeypatch.setenv("FDX_SPLIT_MIN_TILES", "0")          # shards this small would not be split by themselves (comm.cpp)
    dev = torch.device("cuda", 0)
    n, G, K, d = 6000, 300, 12, 64
    Y, X, coords, _ = datagen.count_like(n, G, K, 0.1, 11)
    coords = coords + np.random.RandomState(0).rand(n, 2) * 1e-3
    # float32 on both sides: the shards hold float32 rows, and the transform's class follows the input type (float32 rows ->
    # float32-class log1p, as in the reference; integer counts would be transformed in float64)
    ref = FlashDeconv(sketch_dim=d, max_iter=60, tol=1e-5).fit(Y.astype(np.float32), X, coords)
    cd = torch.from_numpy(np.ascontiguousarray(coords)).to(dev)
    Yt = torch.from_numpy(Y.astype(np.float32)).to(dev)
    full, shards = _native_shards(torch, cd, Yt, X, W, d, K, _lib.PRE_LOG_CPM)
    lam, rho_eff = ref.lambda_used_, 0.01 * diag_mean(shards[0]["XtX_h"])
    results = _run_native_threads(torch, shards, K, lam, rho_eff, 1e-5, 60)
    for res in results:
        assert res[0] == ref.info_["n_iterations"] and res[1] == ref.info_["converged"]
        assert res[2] == results[0][2]                          # every rank saw the same global statistics
    np.testing.assert_allclose(results[0][2], ref.info_["final_change"], rtol=1e-12)
    assert np.array_equal(_assemble(torch, shards, results, n, K).cpu().numpy(), ref.beta_)


def test_sharded_class_three_processes_over_gloo_equal_single_gpu():
    """ShardedFlashDeconv itself with world = 3: three PROCESSES share this GPU and talk over gloo (tools/class_ranks_gloo.py) - plan
    by band recompute with its real all-reduce, gene statistics reduced over the shards, prepare, the Python exchange loop, the
    objective's all-reduce.  Rank 0's assembly must equal the single-GPU fit bit for bit in all eleven cases of the tool.  The only end-to-end run of the estimator's world > 1 code that one GPU allows."""
    import subprocess
    import sys
    tool = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools", "class_ranks_gloo.py")
    res = subprocess.run([sys.executable, tool, "3", "3000"], capture_output=True, text=True, timeout=600)
    assert res.returncode == 0, res.stderr[-2000:]
    assert "done; problems: 0" in res.stdout, res.stdout[-2000:]
    # raw, log-CPM with gene selection, pearson, CSR shards, radius / grid graphs, 70 and 100 cell types, integer counts, clustered
    # coordinates whose k-NN walks leave their block (band recompute falls back to the list exchange)
    # ... and a square lattice with k = 6: ties on every spot, both estimators then take the reference's (cKDTree) graph
    assert res.stdout.count("bits equal True") == 11, res.stdout[-3000:]
    assert "route band" in res.stdout and "route allgather" in res.stdout and "route ckdtree" in res.stdout, res.stdout[-3000:]


def test_bench_driver_with_two_ranks_rehearsed_over_gloo():
    """bench.py --gpus 2 as the driver starts it (it spawns its ranks), REHEARSED on this one GPU: FDX_BENCH_BACKEND=gloo puts both
    ranks on cuda:0 and the collectives on gloo.  The sharded driver's own code - plan + fit per step, barrier / max-over-ranks
    timing, the extra timed-sweeps fit, the result line - runs as it will on N GPUs; the numbers mean nothing here."""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, FDX_BENCH_BACKEND="gloo")
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE"):
        env.pop(k, None)
    res = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1", "--spots", "60000"],
                         capture_output=True, text=True, timeout=400, env=env)
    assert res.returncode == 0, res.stderr[-2000:]
    line = json.loads([ln for ln in res.stdout.splitlines() if ln.startswith("{")][-1])
    assert line["n_gpus"] == 2 and line["steps"] == 2 and line["spots_total"] == 60000 and line["scaling"] == "strong"
    assert line["value"] > 0 and line["ms_per_step"] > 0 and line["config"]["n_iterations"] >= 1
    assert line["roofline"] and line["roofline"]["ms_per_launch"] > 0 and "bcd_sweep" in line["roofline"]["kernel"]
    # what ran, per rank: over gloo there is no RCCL communicator of libfdx's own - the Python exchange loop, and the line says so
    assert line["loop"] == ["python"] and line["rccl_ranks"] is None and line["native_comm_error"] is None and line["ranks_agree"] is True
    assert [r["rank"] for r in line["ranks"]] == [0, 1] and all(r["plan_route"] == "band" and r["device"] == 0 for r in line["ranks"])


def test_native_loop_with_ranks_that_own_no_spot():
    """1000 spots over 6 ranks: shard boundaries sit on multiples of 256, so two ranks own nothing.  They launch no sweep, must still
    take part in every exchange and all-reduce and must see the same convergence trace (a sweep folds its predecessor's statistics -
    without rows nobody did, the empty ranks read 0.0, left after their first chunk and the others waited for ever)."""
    import torch
    from flashdeconv_amd import FlashDeconv, _lib
    from flashdeconv_amd.distributed import diag_mean, shard_bounds
    dev = torch.device("cuda", 0)
    n, G, K, d, W = 1000, 260, 20, 64, 6
    assert (np.diff(shard_bounds(n, W)) == 0).sum() == 2
    Y, X, coords, _ = datagen.count_like(n, G, K, 0.1, 5)
    coords = coords + np.random.RandomState(1).rand(n, 2) * 1e-3
    ref = FlashDeconv(sketch_dim=d, max_iter=9, tol=1e-9).fit(Y.astype(np.float32), X, coords)
    cd = torch.from_numpy(np.ascontiguousarray(coords)).to(dev)
    Yt = torch.from_numpy(Y.astype(np.float32)).to(dev)
    full, shards = _native_shards(torch, cd, Yt, X, W, d, K, _lib.PRE_LOG_CPM)
    lam, rho_eff = ref.lambda_used_, 0.01 * diag_mean(shards[0]["XtX_h"])
    results = _run_native_threads(torch, shards, K, lam, rho_eff, 1e-9, 9)
    assert all(res[0] == 9 and res[2] == results[0][2] for res in results)
    assert np.array_equal(_assemble(torch, shards, results, n, K).cpu().numpy(), ref.beta_)


@pytest.mark.parametrize("n,W,pre,dtype,build", [(3000, 4, "log_cpm", np.float64, "replicated"), (3000, 3, "log_cpm", np.float32, "pipeline"),
                                                 (4099, 5, "raw", np.float64, "pipeline"), (5000, 7, "log_cpm", np.float64, "pipeline"),
                                                 (1537, 3, "raw", np.float32, "replicated")])
def test_native_loop_thread_ranks_against_the_oracle(n, W, pre, dtype, build):
    """A sharded fit compared with the ORACLE directly (not with the single-GPU result): 3 to 7 thread ranks over the native halo
    exchange, float64 and float32 rows, log-CPM and raw, shards that are no multiple of 256 spots, the replicated graph build and the
    queued shard pipeline - abundances at the usual 1e-8 (float32 log-CPM rows: 1e-5, the float32-class transform), the reference's
    iteration count (core/solver.py:104-184 on the whole problem is what every rank's sweeps add up to)."""
    import torch
    import fdx_oracle as orc
    from flashdeconv_amd import _lib
    from flashdeconv_amd.distributed import diag_mean
    dev = torch.device("cuda", 0)
    G, K, d = 260, 7, 48
    Y, X, coords, _ = datagen.count_like(n, G, K, 0.1, 23) if pre == "log_cpm" else datagen.gaussian_raw(n, G, K, seed=23)
    coords = coords + np.random.RandomState(1).rand(n, 2) * 1e-3
    Yin = np.ascontiguousarray(Y, dtype=dtype)
    want = orc.fit(Yin, X, coords, sketch_dim=d, preprocess_method=pre, n_hvg=2000, max_iter=40, tol=1e-6, graph="kdtree")
    cd = torch.from_numpy(np.ascontiguousarray(coords)).to(dev)
    Yt = torch.from_numpy(Yin).to(dev)
    locs = _pipeline_locals(torch, cd, W) if build == "pipeline" else None
    assert build != "pipeline" or locs is not None
    full, shards = _native_shards(torch, cd, Yt, X, W, d, K, _lib.PRE_LOG_CPM if pre == "log_cpm" else _lib.PRE_RAW, locals_=locs)
    lam = float(want["lambda_used"]) if "lambda_used" in want else None
    if lam is None:
        lam = 0.005 * diag_mean(shards[0]["XtX_h"]) / max(full.info()[1] / n, 1.0)          # core/spatial.py:181-190
    rho_eff = 0.01 * diag_mean(shards[0]["XtX_h"])
    results = _run_native_threads(torch, shards, K, lam, rho_eff, 1e-6, 40)
    assert results[0][0] == want["info"]["n_iterations"]
    beta = _assemble(torch, shards, results, n, K).cpu().numpy()
    assert rel_fro(beta, want["beta"]) < (1e-5 if (dtype == np.float32 and pre == "log_cpm") else 1e-8)


def test_native_loop_8_ranks_at_1m_spots_config3():
    """BASELINE configs[3]'s workload on one GPU: the 1M x 2000 x 30 job cut into 8 shards, every shard driven by the
    native loop in its own thread; beta must equal the single-GPU fit bit for bit, in the same number of iterations."""
    import torch
    import bench
    from flashdeconv_amd import FlashDeconv, _lib
    from flashdeconv_amd.distributed import diag_mean
    free, _ = torch.cuda.mem_get_info()
    if free < 60 * 2 ** 30:
        pytest.skip("needs 60 GB of free HBM (1M x 2000 float32 plus eight shards' buffers)")
    dev = torch.device("cuda", 0)
    n, G, K, d, W = 1_000_000, 2000, 30, 512, 8
    Y, X, coords = bench.gen_gaussian(torch, n, G, K, dev, 0)
    ref = FlashDeconv(sketch_dim=d, preprocess="raw", n_hvg=G).fit(Y, X, coords, output="torch")
    full, shards = _native_shards(torch, coords, Y, X, W, d, K, _lib.PRE_RAW)
    lam, rho_eff = ref.lambda_used_, 0.01 * diag_mean(shards[0]["XtX_h"])
    results = _run_native_threads(torch, shards, K, lam, rho_eff, 1e-4, 100)
    assert all(res[0] == ref.info_["n_iterations"] and res[1] for res in results) and ref.info_["converged"]
    got = _assemble(torch, shards, results, n, K)
    assert torch.equal(got, ref.beta_)


class _ThreadWorld:
    """W thread ranks of one process: libfdx's in-process transport (fdx_local_world_create) for the native calls, a barrier and
    a slot list for the Python-side collectives of ShardedFlashDeconv."""

    def __init__(self, W):
        import threading
        from flashdeconv_amd import _lib
        self.W = W
        self.barrier = threading.Barrier(W, timeout=60)
        self.slots = [None] * W
        self.handle = ctypes.c_void_p()
        _lib.check(_lib.load().fdx_local_world_create(W, ctypes.byref(self.handle)))

    def close(self):
        from flashdeconv_amd import _lib
        _lib.load().fdx_local_world_destroy(self.handle)


class _ThreadComm:
    """The comm interface of flashdeconv_amd.distributed (TorchComm) for one thread rank; brings its own libfdx communicator, so
    ShardedFlashDeconv takes the native path (fdx_shard_fit_dev, the C++ iteration loop) exactly as over RCCL."""

    def __init__(self, world, rank):
        from flashdeconv_amd import _lib
        self.w, self.rank, self.world, self.group = world, rank, world.W, None
        self.native_handle = ctypes.c_void_p()
        _lib.check(_lib.load().fdx_comm_init_local(world.handle, rank, ctypes.byref(self.native_handle)))

    def _gather(self, item):
        import torch
        torch.cuda.current_stream().synchronize()
        self.w.slots[self.rank] = item
        self.w.barrier.wait()
        items = list(self.w.slots)
        self.w.barrier.wait()
        return items

    def all_reduce_sum(self, t):
        items = self._gather(t.clone())
        acc = items[0].clone()
        for x in items[1:]:
            acc += x
        t.copy_(acc)

    def all_reduce_max(self, t):
        import torch
        items = self._gather(t.clone())
        acc = items[0].clone()
        for x in items[1:]:
            acc = torch.maximum(acc, x)
        t.copy_(acc)

    def all_gather_rows(self, nbr, cnt, bounds):
        lo, hi = int(bounds[self.rank]), int(bounds[self.rank + 1])
        items = self._gather((nbr[lo:hi].clone(), cnt[lo:hi].clone()))
        for q, (a, b) in enumerate(items):
            if q != self.rank:
                nbr[int(bounds[q]):int(bounds[q + 1])] = a
                cnt[int(bounds[q]):int(bounds[q + 1])] = b

    def exchange(self, send_bufs, recv_bufs):
        items = self._gather({p: t.clone() for p, t in send_bufs.items()})
        for peer, t in recv_bufs.items():
            t.copy_(items[peer][self.rank])

    def close(self):
        from flashdeconv_amd import _lib
        _lib.load().fdx_comm_destroy(self.native_handle)


def _class_thread_ranks(torch, W, kw, Y, X, coords):
    """ShardedFlashDeconv on W thread ranks of this process: plan + fit_transform per rank, results put together in the caller's
    spot order.  Returns (proportions, [model of every rank])."""
    import threading
    from flashdeconv_amd.distributed import ShardedFlashDeconv
    dev = torch.device("cuda", 0)
    world = _ThreadWorld(W)
    cd = torch.from_numpy(np.ascontiguousarray(coords)).to(dev)
    Yt = torch.from_numpy(np.ascontiguousarray(Y)).to(dev)
    n, K = Y.shape[0], X.shape[0]
    got = np.zeros((n, K))
    models, errors = [None] * W, []

    def work(r):
        try:
            torch.cuda.set_device(0)
            comm = _ThreadComm(world, r)
            m = ShardedFlashDeconv(comm=comm, **kw)
            own = m.plan(cd)
            P = m.fit_transform(Yt[own], X)
            got[own.cpu().numpy()] = P.cpu().numpy()
            models[r] = m
        except Exception as e:                                   # noqa: BLE001 - reported below; the peers' barrier times out
            import traceback
            errors.append((r, repr(e), traceback.format_exc()))
            world.barrier.abort()

    threads = [threading.Thread(target=work, args=(r,)) for r in range(W)]
    for t in threads:
        t.start()
    for t in threads:
        t.join(timeout=120)
    assert not any(t.is_alive() for t in threads), "a rank thread hangs"
    assert not errors, errors
    return got, models


@pytest.mark.timeout(300)
@pytest.mark.parametrize("W,force_overflow_on", [(3, None), (4, 1), (3, "all")])
def test_sharded_class_on_thread_ranks_takes_the_native_fit(W, force_overflow_on, monkeypatch):
    """ShardedFlashDeconv END TO END with W > 1 ranks through the native path - fdx_shard_fit_dev (csrc/comm.cpp): the plan's counts
    all-reduced beside the sketch, lambda, the C++ iteration loop, objective sums - over libfdx's in-process transport: everything
    the RCCL job runs except the wire.  Bits of the single-GPU fit.  force_overflow_on: the "bound too small" remedy of the queued
    plan on ONE rank (the others keep their device-built graphs; the job-level all-reduce of the counts must still match on every
    rank: it hung when it was decided per graph) or on all of them."""
    import torch
    from flashdeconv_amd import FlashDeconv
    if force_overflow_on is not None:
        monkeypatch.setenv("FDX_GRAPH_WCAP", "1")
        if force_overflow_on != "all":
            monkeypatch.setenv("FDX_GRAPH_WCAP_RANK", str(force_overflow_on))
    n, G, K = 6000, 300, 9
    Y, X, coords, _ = datagen.gaussian_raw(n, G, K, seed=4)
    kw = dict(sketch_dim=64, preprocess="raw", n_hvg=G, max_iter=40)
    from flashdeconv_amd import _lib
    env_cap = os.environ.pop("FDX_GRAPH_WCAP", None)             # the single-GPU reference is built without the forced bound
    _lib.env_reload()
    try:
        ref = FlashDeconv(**kw).fit(Y, X, coords)
    finally:
        if env_cap is not None:
            os.environ["FDX_GRAPH_WCAP"] = env_cap
        _lib.env_reload()
    got, models = _class_thread_ranks(torch, W, kw, Y, X, coords)
    for m in models:
        assert m.comm_report()["loop"] == "native"
        assert m.info_["n_iterations"] == ref.info_["n_iterations"] and m.info_["converged"] == ref.info_["converged"]
        np.testing.assert_allclose(m.lambda_used_, ref.lambda_used_, rtol=1e-14)
        np.testing.assert_allclose(m.info_["final_objective"], ref.info_["final_objective"], rtol=1e-12)
    assert np.array_equal(got, ref.proportions_)
    rebuilt = [bool(getattr(m, "plan_rebuilt_stepwise_", False)) for m in models]
    want = [force_overflow_on == "all" or r == force_overflow_on for r in range(W)] if force_overflow_on is not None else [False] * W
    assert rebuilt == want, (rebuilt, want)                      # the remedy ran where it was forced, and only there


@pytest.mark.parametrize("W", [2, 3])
@pytest.mark.parametrize("method", ["radius", "grid"])
def test_sharded_radius_build_equals_the_replicated_one(W, method):
    """Radius / grid graphs for spot shards (fdx_graph_build_radius_rows_dev): every rank builds rows [lo, hi) only - a radius
    graph is symmetric by construction, nothing is exchanged.  The localized shard (own spots, halo, send lists, degrees)
    must be the one cut from the replicated graph, the own edge counts must add up to the whole graph, and a sweep on the
    shard must give the bits of a sweep on the shard of the replicated graph."""
    import torch
    from flashdeconv_amd import FlashDeconv, _lib
    from flashdeconv_amd.distributed import shard_bounds
    lib = _lib.load()
    dev = torch.device("cuda", 0)
    n = 5000
    rs = np.random.RandomState(W)
    side = int(np.ceil(np.sqrt(n)))
    coords = np.stack([np.arange(n) % side, np.arange(n) // side], axis=1).astype(np.float64) + rs.rand(n, 2) * 0.2
    proto = FlashDeconv(spatial_method=method, radius=1.3 if method == "radius" else None)
    cd = torch.from_numpy(np.ascontiguousarray(coords)).to(dev)
    gm, gk, gradius = proto._graph_request(cd, None)
    assert gm == _lib.GRAPH_RADIUS
    h = ctypes.c_void_p()
    _lib.check(lib.fdx_graph_build_dev(ctypes.c_void_p(cd.data_ptr()), n, 2, gm, gk, float(gradius), _st(torch), ctypes.byref(h)))
    full = _lib.Graph(h.value)
    bounds = shard_bounds(n, W)
    nnz_sum = 0
    for r in range(W):
        lo, hi = int(bounds[r]), int(bounds[r + 1])
        hp = ctypes.c_void_p()
        _lib.check(lib.fdx_graph_build_radius_rows_dev(ctypes.c_void_p(cd.data_ptr()), n, 2, float(gradius), lo, hi, _st(torch),
                                                       ctypes.byref(hp)))
        part = _lib.Graph(hp.value)
        nnz_sum += part.info()[1]
        locs = []
        for g in (full, part):
            hl = ctypes.c_void_p()
            _lib.check(lib.fdx_graph_localize(g.handle, W, _lib.ptr_i64(bounds), r, _st(torch), ctypes.byref(hl)))
            loc = _lib.Graph(hl.value)
            n_own = hi - lo
            perm = torch.empty(max(n_own, 1), dtype=torch.int32, device=dev)
            _lib.check(lib.fdx_graph_perm_dev(loc.handle, ctypes.c_void_p(perm.data_ptr()), _st(torch)))
            nh = ctypes.c_int64(0)
            sc, rc = np.zeros(W, dtype=np.int32), np.zeros(W, dtype=np.int32)
            _lib.check(lib.fdx_graph_halo_info(loc.handle, ctypes.byref(nh), _lib.ptr_i32(sc), _lib.ptr_i32(rc)))
            sidx = torch.empty(max(int(sc.sum()), 1), dtype=torch.int32, device=dev)
            _lib.check(lib.fdx_graph_send_indices_dev(loc.handle, ctypes.c_void_p(sidx.data_ptr()), _st(torch)))
            torch.cuda.synchronize()
            locs.append((loc.info(), perm.cpu().numpy()[:n_own], int(nh.value), sc, rc, sidx.cpu().numpy()[:int(sc.sum())], loc))
        a, b = locs
        assert a[0] == b[0] and a[2] == b[2]
        assert np.array_equal(a[1], b[1]) and np.array_equal(a[3], b[3]) and np.array_equal(a[4], b[4]) and np.array_equal(a[5], b[5])
        # one sweep on both shards: same bits
        K = 5
        n_total = (hi - lo) + a[2]
        ld = ((n_total + 1 + 63) // 64) * 64
        g1 = torch.Generator(device=dev).manual_seed(r)
        Hm = torch.rand((K, ld), dtype=torch.float64, device=dev, generator=g1)
        XtX = torch.eye(K, dtype=torch.float64, device=dev) * 2.0 + 0.1
        outs = []
        from flashdeconv_amd.distributed import HipBackend
        for loc in (a[6], b[6]):
            be = HipBackend(loc, Hm, ld, XtX, K)
            beta = [torch.zeros((K, ld), dtype=torch.float64, device=dev) for _ in range(2)]
            be.init_beta(beta[0], n_total)
            stats = torch.zeros((4, 128), dtype=torch.float64, device=dev)
            rel = torch.zeros(4, dtype=torch.float64, device=dev)
            be.sweep(0, beta[0], beta[1], 0.3, 0.01, 1e-9, stats, rel)
            torch.cuda.synchronize()
            outs.append(beta[1].cpu().numpy()[:, :hi - lo].copy())
        assert np.array_equal(outs[0], outs[1])
    assert nnz_sum == full.info()[1]



@pytest.mark.parametrize("tool,env,limit", [("fuzz_sharded.py", dict(SEED="5", TRIALS="40", BIGK="1"), 400),
                                            ("fuzz_fit.py", dict(SEED="3", TRIALS="48", HIGHDIM="1"), 400)])
def test_seeded_slices_of_the_randomised_bug_nets(tool, env, limit):
    """The two randomised nets that found the round-4 defects of the sharded path, as tests: a seeded slice of each - sharded fits
    with thread ranks (2-7 ranks, odd spot counts, clustered / strip-shaped coordinates, replicated / band / queued-pipeline graph
    builds, 3 to 100 cell types, raw / log-CPM, float32 / float64 rows) bit for bit against the single-GPU fit, and small single-GPU
    fits (all preprocess modes, dense / integer / CSR input, 1- to 6-dimensional coordinates, k-NN / radius / grid graphs) against the
    oracle."""
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    res = subprocess.run([sys.executable, os.path.join(root, "tools", tool)], capture_output=True, text=True, timeout=limit,
                         env=dict(os.environ, **env))
    assert res.returncode == 0, res.stderr[-3000:]
    assert "done; problems: 0" in res.stdout, res.stdout[-3000:]


def test_two_models_fitted_from_two_threads_at_once_equal_the_single_thread_bits():
    """SURVEY 8(b), threading: "two models may be fit from two threads" (the reference has no shared mutable state, core/deconv.py).
    Two FlashDeconv models with different problems are fitted concurrently from two Python threads (the C calls release the GIL;
    the library keeps per-thread error state, a locked buffer pool, per-device side stream); each must reproduce, bit for bit,
    what it gives alone.  Several rounds, alternating which thread starts."""
    import threading
    from flashdeconv_amd import FlashDeconv
    probs = []
    for seed, (n, G, K, d, pre) in enumerate([(3000, 400, 7, 64, "log_cpm"), (4500, 300, 12, 48, "raw")]):
        Y, X, coords, _ = (datagen.count_like(n, G, K, 0.1, 20 + seed) if pre == "log_cpm" else datagen.gaussian_raw(n, G, K, seed=20 + seed))
        probs.append((Y.astype(np.float32), X, coords, dict(sketch_dim=d, preprocess=pre, max_iter=25)))
    alone = [FlashDeconv(**kw).fit(Y, X, c) for Y, X, c, kw in probs]
    for rnd in range(4):
        out, errs = [None, None], []

        def work(j):
            try:
                Y, X, c, kw = probs[j]
                out[j] = FlashDeconv(**kw).fit(Y, X, c)
            except Exception as e:                           # noqa: BLE001
                errs.append((j, repr(e)))

        order = [0, 1] if rnd % 2 == 0 else [1, 0]
        ths = [threading.Thread(target=work, args=(j,)) for j in order]
        for t in ths:
            t.start()
        for t in ths:
            t.join(timeout=120)
        assert not errs, errs
        for j in range(2):
            assert out[j] is not None and out[j].info_["n_iterations"] == alone[j].info_["n_iterations"]
            assert np.array_equal(out[j].beta_, alone[j].beta_) and np.array_equal(out[j].proportions_, alone[j].proportions_), (rnd, j)


def test_sweep_variants_give_the_same_bits(monkeypatch):
    """Round 5's forms of the tiled sweep against the ones they replaced, bit for bit: the first sweep that takes the uniform start
    vector as a constant (INIT) vs the written one, the sweep that writes the send staging itself vs the separate pack kernel, and
    the one-pass ELL / tile-table kernel vs fill_ell + tile_halo."""
    import torch
    from flashdeconv_amd import FlashDeconv, _lib
    from flashdeconv_amd.distributed import diag_mean
    dev = torch.device("cuda", 0)
    n, G, K, d, W = 5000, 260, 7, 48, 4
    Y, X, coords, _ = datagen.gaussian_raw(n, G, K, seed=31)
    Yt = torch.from_numpy(np.ascontiguousarray(Y, dtype=np.float64)).to(dev)
    cd = torch.from_numpy(np.ascontiguousarray(coords)).to(dev)

    def single():
        m = FlashDeconv(sketch_dim=d, preprocess="raw", n_hvg=G, max_iter=25, tol=1e-7).fit(Yt, X, cd, output="torch")
        return m.beta_.clone(), m.info_["n_iterations"], m.lambda_used_

    def sharded(lam):
        full, shards = _native_shards(torch, cd, Yt, X, W, d, K, _lib.PRE_RAW)
        res = _run_native_threads(torch, shards, K, lam, 0.01 * diag_mean(shards[0]["XtX_h"]), 1e-7, 25)
        return _assemble(torch, shards, res, n, K), res[0][0]

    b0, it0, lam = single()
    s0, sit0 = sharded(lam)
    assert sit0 == it0 and torch.equal(s0, b0)
    # (FDX_NO_TILED: the global-gather sweep - graphs whose tiles have too large a halo take it by themselves;
    # FDX_NO_SIDE_STREAM: everything on the caller's stream; FDX_KDTREE_*: no effect on a tie-free fit, read by the registry)
    for var in ("FDX_NO_INIT_SWEEP", "FDX_GRAPH_TWO_ELL_KERNELS", "FDX_NO_TILED", "FDX_NO_SIDE_STREAM", "FDX_NO_PLAN_CACHE"):
        monkeypatch.setenv(var, "1")
        b1, it1, _ = single()
        monkeypatch.delenv(var)
        assert it1 == it0 and torch.equal(b1, b0), var
    for var in ("FDX_NO_INIT_SWEEP", "FDX_NO_FUSED_PACK"):
        monkeypatch.setenv(var, "1")
        s1, sit1 = sharded(lam)
        monkeypatch.delenv(var)
        assert sit1 == it0 and torch.equal(s1, b0), var
