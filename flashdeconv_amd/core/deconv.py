"""``FlashDeconv`` estimator: the reference's ``flashdeconv/core/deconv.py`` surface on the MI355X path.

Constructor arguments, validation messages, fitted attributes (``beta_``, ``proportions_``, ``gene_idx_``, ``info_``,
``adjacency_``, ``lambda_used_``, ``n_spots_`` ...), getters and ``summary()`` follow core/deconv.py:88-512.  ``fit``
keeps the six steps of core/deconv.py:305-398 but runs steps 2-6 in one device-resident call (``fdx_fit_dev``).

Inputs may be NumPy arrays, SciPy sparse matrices, or CUDA (HIP) ``torch`` tensors that already live in HBM; with
``output="torch"`` the fitted arrays stay on the device as well.
"""
import ctypes
import os
import sys
import time

import numpy as np
from scipy import sparse

from .. import _lib
from ..utils import genes as _genes
from .sketching import countsketch_tables

_PRE_MODES = ("log_cpm", "pearson", "raw")


def _is_torch_cuda(x):
    return hasattr(x, "data_ptr") and hasattr(x, "is_cuda") and bool(x.is_cuda)


def device_counts_as_float(Y):
    """A dense CUDA tensor as the float32 / float64 matrix the kernels stream, plus whether its transform must keep float64
    accuracy (``PRE_F64_MATH``).  Integer counts: numpy promotes them to float64 in the reference (core/deconv.py:190-191),
    so the log-CPM chain stays float64; float32 STORAGE is used when it holds every value exactly (below 2**24 - checked
    unless the dtype cannot exceed it), float64 otherwise.  Shared by FlashDeconv.fit and ShardedFlashDeconv.fit_transform."""
    import torch
    f64_math = False
    if Y.dtype not in (torch.float32, torch.float64):
        f64_math = not Y.dtype.is_floating_point
        small = Y.dtype in (torch.uint8, torch.int8, torch.int16, torch.bool, torch.float16, torch.bfloat16)
        exact = small or Y.numel() == 0
        if not exact:
            try:                                          # torch lacks min / max for some unsigned dtypes: float64 then
                hi = int(Y.max().item())
                lo = int(Y.min().item()) if Y.dtype.is_signed else 0
                exact = max(hi, -lo) < (1 << 24)
            except (RuntimeError, TypeError, NotImplementedError):
                exact = False
        Y = Y.to(torch.float32 if exact else torch.float64)
    return Y.contiguous(), f64_math


class _DeviceBuffer:
    """Plain device allocation through the C ABI (fdx_malloc / fdx_free)."""

    def __init__(self, nbytes):
        self.ptr = ctypes.c_void_p()
        self.nbytes = int(nbytes)
        _lib.check(_lib.load().fdx_malloc(ctypes.byref(self.ptr), self.nbytes))

    @classmethod
    def from_host(cls, arr):
        arr = np.ascontiguousarray(arr)
        buf = cls(arr.nbytes)
        _lib.upload_bytes(buf.ptr, arr)
        return buf

    @classmethod
    def adopt(cls, ptr, nbytes):
        """Takes over a block handed out by the library (fdx_malloc inside _lib.upload_matrix)."""
        buf = cls.__new__(cls)
        buf.ptr, buf.nbytes = ptr, int(nbytes)
        return buf

    def to_host(self, shape, dtype=np.float64, out=None):
        if out is None:
            out = np.empty(shape, dtype=dtype)
        _lib.download_bytes(out, self.ptr)
        return out

    def free(self):
        if self.ptr is not None and self.ptr.value:
            _lib.load().fdx_free(self.ptr)
            self.ptr = None

    def __del__(self):
        try:
            self.free()
        except Exception:
            pass


def reference_tie_graph(c_ptr, coords_host, n, dim, k):
    """The k-NN graph on the REFERENCE's choice among exactly equidistant neighbours (regular lattices tie on every spot): the
    lists come from the restated cKDTree (utils/graph.py: ckdtree_knn_lists - scipy's build and traversal order, on the host),
    everything else stays on the device - the lists are put, in solver positions, where the device's own lists would be and
    symmetrised / laid out by fdx_graph_from_knn_lists_dev.  The graph therefore keeps the Morton order and the sweep tiles of
    any other device-built graph (the round-4 route went through a host adjacency in the caller's order: A + A^T in scipy, an
    untiled upload).  Reference: utils/graph.py:60-81."""
    from ..utils.graph import _ckdtree_restatement_matches_scipy
    lib = _lib.load()
    tr = [("start", time.perf_counter())] if os.environ.get("FDX_TRACE_HOST") else None

    def mark(name):
        if tr is not None:
            tr.append((name, time.perf_counter()))
    kk = min(int(k), n - 1) + 1
    nbr, cnt = _DeviceBuffer(n * kk * 4), _DeviceBuffer(n * 4)
    plan, h = ctypes.c_void_p(), ctypes.c_void_p()
    # where the tree is built on the host (4-8 coordinates, fdx_kdtree_tune(2, 0)) that build starts NOW, in the library's thread
    # pool, beside the device's own lists below, which wait ~3 ms behind the stopped fit's sketch; for 1-3 coordinates the tree is
    # built on the device when the lists are asked for and this call returns at once
    _ckdtree_restatement_matches_scipy()
    ch_arr = None if coords_host is None else np.ascontiguousarray(coords_host, dtype=np.float64)
    ch = None if ch_arr is None else _lib.ptr_f64(ch_arr)
    if dim <= 8:
        _lib.check(lib.fdx_ckdtree_prebuild(ch, c_ptr, n, dim))
    try:
        _lib.check(lib.fdx_graph_knn_lists_dev(c_ptr, n, dim, int(k), 0, n, nbr.ptr, cnt.ptr, None, ctypes.byref(plan)))
        try:
            mark("device lists + order")
            # the restated tree is built on the host, its queries run on the device (1-3 coordinates), and the answers go - as solver
            # positions, self dropped - to the rows' positions without leaving the device
            # coordinates that only exist on the device are fetched by the library into pinned memory (a pageable copy here would be
            # pinned by the driver, and unmapping it afterwards stalls the process's GPU queues: DESIGN appendix)
            _lib.check(lib.fdx_graph_plan_set_ckdtree_lists_dev(plan, ch, c_ptr, n, dim, None, 0, nbr.ptr, cnt.ptr, None))
            mark("tree (device, 1-3 coordinates; else the host's pool) + device queries + lists to positions")
        except Exception:
            dead = ctypes.c_void_p()                                           # the plan owns device buffers: consume it
            lib.fdx_graph_from_knn_lists_dev(plan, nbr.ptr, cnt.ptr, 0, 0, None, ctypes.byref(dead))
            if dead.value:
                _lib.Graph(dead.value).close()
            raise
        _lib.check(lib.fdx_graph_from_knn_lists_dev(plan, nbr.ptr, cnt.ptr, 0, n, None, ctypes.byref(h)))
        mark("symmetrise + ELL + tiles (device)")
        if tr is not None:
            print("[fdx-host] reference_tie_graph: " + ", ".join(f"{b[0]} {1e3 * (b[1] - a[1]):.1f} ms" for a, b in zip(tr[:-1], tr[1:])),
                  file=sys.stderr)
        return _lib.Graph(h.value)
    finally:
        for b in (nbr, cnt):
            b.free()


class FlashDeconv:
    """Fast spatial transcriptomics deconvolution with spatial regularisation (MI355X implementation).

    Parameters are those of the reference estimator (core/deconv.py:27-68): ``sketch_dim``, ``lambda_spatial``
    (float or "auto"), ``rho_sparsity``, ``n_hvg``, ``n_markers_per_type``, ``spatial_method`` ("knn" | "radius" |
    "grid"), ``k_neighbors``, ``radius``, ``max_iter``, ``tol``, ``preprocess`` ("log_cpm" | "pearson" | "raw"),
    ``random_state``, ``verbose``.
    """

    def __init__(self, sketch_dim=512, lambda_spatial="auto", rho_sparsity=0.01, n_hvg=2000, n_markers_per_type=50,
                 spatial_method="knn", k_neighbors=6, radius=None, max_iter=100, tol=1e-4, preprocess="log_cpm",
                 random_state=0, verbose=False, knn_ties="auto"):
        if sketch_dim <= 0:
            raise ValueError(f"sketch_dim must be positive, got {sketch_dim}")
        if k_neighbors < 0:
            raise ValueError(f"k_neighbors must be non-negative, got {k_neighbors}")
        if max_iter < 0:
            raise ValueError(f"max_iter must be non-negative, got {max_iter}")
        if tol <= 0:
            raise ValueError(f"tol must be positive, got {tol}")
        if isinstance(lambda_spatial, (int, float)) and lambda_spatial < 0:
            raise ValueError(f"lambda_spatial must be non-negative, got {lambda_spatial}")
        if rho_sparsity < 0:
            raise ValueError(f"rho_sparsity must be non-negative, got {rho_sparsity}")
        if n_hvg < 0:
            raise ValueError(f"n_hvg must be non-negative, got {n_hvg}")
        if n_markers_per_type < 0:
            raise ValueError(f"n_markers_per_type must be non-negative, got {n_markers_per_type}")
        if spatial_method == "radius" and radius is None:
            raise ValueError("radius must be specified when spatial_method='radius'")
        if radius is not None and radius <= 0:
            raise ValueError(f"radius must be positive, got {radius}")
        self.sketch_dim = sketch_dim
        self.lambda_spatial = lambda_spatial
        self.rho_sparsity = rho_sparsity
        self.n_hvg = n_hvg
        self.n_markers_per_type = n_markers_per_type
        self.spatial_method = spatial_method
        self.k_neighbors = k_neighbors
        self.radius = radius
        self.max_iter = max_iter
        self.tol = tol
        self.preprocess = preprocess
        self.random_state = random_state
        self.verbose = verbose
        # additive (not in the reference): how exactly equidistant candidates for the k-th neighbour are taken (regular lattices
        # tie on every spot).  "auto" (default): the device build's graph when it met no tie - every tie-free input, at no cost -
        # and otherwise the fit is repeated on the reference's own choice (cKDTree's traversal order, restated on the host:
        # csrc/kdtree_order.cpp), so the result is the reference's on lattices too; "ckdtree": the same, decided before the
        # first solve (one fit, but the graph build is waited for); "index": ascending spot index, the device rule, always
        # (fastest on lattices; a warning reports the ties, proportions a few 1e-4 from the reference's)
        if knn_ties not in ("auto", "index", "ckdtree"):
            raise ValueError(f"knn_ties must be 'auto', 'index' or 'ckdtree', got {knn_ties}")
        self.knn_ties = knn_ties

        self.beta_ = None
        self.proportions_ = None
        self.gene_idx_ = None
        self.info_ = None
        self._fitted = False
        self._graph = None
        self._adjacency = None

    # ------------------------------------------------------------------ adjacency_ (materialised on first access)
    @property
    def adjacency_(self):
        if self._adjacency is None and self._graph is not None:
            n = self.n_spots_
            indptr, indices = self._graph.to_csr_arrays()
            self._adjacency = sparse.csr_matrix((np.ones(len(indices), dtype=np.float64), indices,
                                                 indptr.astype(np.int32) if len(indices) < 2**31 - 1 else indptr),
                                                shape=(n, n))
        return self._adjacency

    @adjacency_.setter
    def adjacency_(self, value):
        self._adjacency = value

    # ------------------------------------------------------------------ fit
    def fit(self, Y, X, coords, cell_type_names=None, output="numpy"):
        """Fit the model (core/deconv.py:237-405).  ``output="torch"`` keeps ``beta_``/``proportions_`` in HBM."""
        t_entry = time.perf_counter()
        if Y.shape[1] != X.shape[1]:
            raise ValueError(
                f"Gene dimension mismatch: Y has {Y.shape[1]} genes but X has {X.shape[1]} genes. They must share "
                f"the same gene space (align before calling fit).")
        if coords.shape[0] != Y.shape[0]:
            raise ValueError(
                f"Spot count mismatch: Y has {Y.shape[0]} spots but coords has {coords.shape[0]} rows. Each spot "
                f"needs exactly one coordinate.")
        if X.shape[0] == 0:
            raise ValueError(
                "Reference matrix X must contain at least one cell type (X.shape[0] > 0). Check your reference "
                "filtering and cell_type_key mapping.")
        if cell_type_names is not None and len(cell_type_names) != X.shape[0]:
            raise ValueError(
                f"cell_type_names length ({len(cell_type_names)}) does not match number of cell types in X "
                f"({X.shape[0]}).")
        if self.preprocess not in _PRE_MODES:
            raise ValueError(f"Unknown preprocess method: {self.preprocess}. Choose from 'log_cpm', 'pearson', or 'raw'.")
        if self.spatial_method not in ("knn", "radius", "grid"):
            raise ValueError(f"Unknown method: {self.spatial_method}")
        if len(coords.shape) != 2 or coords.shape[1] == 0:
            raise ValueError(f"coords must be 2D with at least 1 coordinate dimension, got shape {tuple(coords.shape)}")
        from ..utils.graph import check_coord_dims
        check_coord_dims(int(coords.shape[0]), int(coords.shape[1]), self.spatial_method == "knn")
        # the reference takes any k (utils/graph.py:51); the device's k-NN lists hold at most 64 entries per spot (self included):
        # above that the lists come from the restated cKDTree on the host (utils/graph.py: ckdtree_knn_adjacency - the reference's
        # own neighbours, ties included) and the adjacency is uploaded as given (caller's spot order, untiled sweep)
        big_k = self.spatial_method == "knn" and min(int(self.k_neighbors), int(coords.shape[0]) - 1) > 63
        _lib.require_gpu()
        lib = _lib.load()
        log = print if self.verbose else (lambda *a, **k: None)
        log("FlashDeconv: Starting deconvolution...")
        log(f"  Spatial data: {Y.shape[0]} spots x {Y.shape[1]} genes")
        log(f"  Reference: {X.shape[0]} cell types x {X.shape[1]} genes")

        n, G_all = int(Y.shape[0]), int(Y.shape[1])
        X = np.asarray(X.detach().cpu().numpy() if hasattr(X, "detach") else X, dtype=np.float64)
        K = X.shape[0]
        self.n_spots_, self.n_genes_, self.n_cell_types_ = n, G_all, K
        self.cell_type_names_ = cell_type_names
        if n == 0:
            raise ValueError("Y has no spots")

        # Y into HBM (full gene set), then Step 1: informative genes + leverage scores (core/deconv.py:305-318)
        owned = []
        try:
            if G_all == 0:
                raise ValueError("No genes selected. Increase n_hvg or n_markers_per_type.")
            csr = None
            y_f64_math = False
            if sparse.issparse(Y) or _lib.is_torch_sparse_csr(Y):
                # sparse stays sparse in HBM: gene statistics, log-CPM and the sketch read the stored entries only
                csr = _lib.CsrOnDevice.from_scipy(Y) if sparse.issparse(Y) else _lib.CsrOnDevice.from_torch(Y)
                owned.append(csr)
                y_ptr, y_code, y_sparse_rule = None, csr.view.dtype, True
            elif _is_torch_cuda(Y):
                import torch
                Y, y_f64_math = device_counts_as_float(Y)
                y_ptr, y_code = ctypes.c_void_p(Y.data_ptr()), (_lib.FDX_F32 if Y.dtype == torch.float32 else _lib.FDX_F64)
                y_sparse_rule = False
            else:
                y_sparse_rule = False
                Yh = np.asarray(Y)
                y_f64_math = Yh.dtype.kind in "iub"
                y_ptr, y_code = _lib.upload_matrix(Yh)         # threaded, staged, integer counts narrowed on the way
                owned.append(_DeviceBuffer.adopt(y_ptr, Yh.size * (4 if y_code == _lib.FDX_F32 else 8)))
            csr_colsum = None
            # coordinates into HBM; with gene selection active (G > n_hvg) the spatial graph - which does not depend on the genes -
            # is queued FIRST: its ~0.7 ms of kernels then run under the gene statistics and the host's ranking of the G-vector
            # instead of after them (the device idled ~1.5 ms there)
            if _is_torch_cuda(coords):
                import torch
                cd = coords.to(torch.float64).contiguous()
                c_ptr = ctypes.c_void_p(cd.data_ptr())
                coords_host = None
            else:
                coords_host = np.ascontiguousarray(np.asarray(coords), dtype=np.float64)
                cbuf = _DeviceBuffer.from_host(coords_host)
                owned.append(cbuf)
                c_ptr = cbuf.ptr
            dim = int(coords.shape[1])
            g_method, g_k, g_radius = self._graph_request(coords, coords_host)
            graph_early = G_all > self.n_hvg and not big_k
            t_graph = t_graph_done = None
            if self._graph is not None:
                self._graph.close()
                self._graph = None
            self._adjacency = None
            if graph_early:
                t_graph = time.perf_counter()
                gh = ctypes.c_void_p()
                side = ctypes.c_void_p()
                if os.environ.get("FDX_NO_SIDE_STREAM"):
                    side = None
                else:
                    # on the library's side stream, behind whatever produced the coordinates on the default stream so far
                    _lib.check(lib.fdx_side_stream(ctypes.byref(side)))
                    if _is_torch_cuda(coords):
                        import torch
                        _lib.check(lib.fdx_stream_wait_stream(side, ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)))
                _lib.check(lib.fdx_graph_build_dev(c_ptr, n, dim, g_method, g_k, g_radius, side, ctypes.byref(gh)))
                self._graph = _lib.Graph(gh.value)
                t_graph_done = time.perf_counter()
            log("Step 1: Selecting informative genes...")
            t_sel = time.perf_counter()
            if G_all <= self.n_hvg:
                # select_hvg returns every gene when the matrix has no more than n_hvg of them and the marker union is a
                # subset (utils/genes.py:135-145, 330) - no pass over Y needed.
                gene_idx = np.arange(G_all, dtype=np.intp)
            else:
                # the marker table depends on X only: a helper thread ranks it while the device reduces Y to its per-gene
                # moments (the C call releases the GIL)
                import concurrent.futures
                tr = [time.perf_counter()] if os.environ.get("FDX_TRACE_HOST") else None
                with concurrent.futures.ThreadPoolExecutor(max_workers=1) as pool:
                    fut = pool.submit(_genes.select_markers, X, self.n_markers_per_type)
                    if csr is not None:
                        mean, var, csr_colsum = csr.gene_moments(want_colsum=self.preprocess == "pearson")
                    else:
                        mean, var = _genes.gene_moments_device(y_ptr, y_code, n, G_all, G_all)
                    if tr is not None:
                        tr.append(time.perf_counter())
                    hvg = _genes._hvg_from_moments(mean, var, self.n_hvg, 0.0125, 3.0, 0.5)
                    if tr is not None:
                        tr.append(time.perf_counter())
                    markers, _ = fut.result()
                    if tr is not None:
                        tr.append(time.perf_counter())
                gene_idx = np.union1d(hvg, markers).astype(np.intp)                  # utils/genes.py:330
                if tr is not None:
                    tr.append(time.perf_counter())
                    print("[fdx-host] python select: moments (device + read-back) %.0f us, hvg ranking %.0f, wait for the marker table %.0f, "
                          "pool shutdown + union %.0f" % tuple(1e6 * (b - a) for a, b in zip(tr[:-1], tr[1:])), file=sys.stderr)
                if len(gene_idx) == 0:
                    raise ValueError("No genes selected. Increase n_hvg or n_markers_per_type.")
            self.gene_idx_ = gene_idx
            t_sel_end = time.perf_counter()
            t_sel = t_sel_end - t_sel
            G = len(gene_idx)
            log(f"  Selected {G} genes (HVG + markers)")
            Xsel = np.ascontiguousarray(X) if G == G_all else np.take(X, gene_idx, axis=1)          # (np.take: half the time of X[:, gene_idx] at 30 x 3300 of 20000)
            # The leverage SVD runs on the library's side stream beside the graph build.  Its job is set up FIRST: set up behind
            # the build call, its pooled buffers (last used on the caller's stream) order the side stream behind everything the
            # build has just queued - the SVD then starts when the graph is done (measured: the wait 0.6 -> 1.3 ms).
            t_x0 = time.perf_counter()
            lev_job = _genes.LeverageJob(Xsel, queue_async=not graph_early)   # graph already queued: the scores are collected next
            if os.environ.get("FDX_TRACE_HOST"):
                print(f"[fdx-host] python: select end -> Xsel {1e6 * (t_x0 - t_sel_end):.0f} us, leverage job set-up {1e6 * (time.perf_counter() - t_x0):.0f}",
                      file=sys.stderr)
            if G != G_all and csr is None:          # Y[:, gene_idx] (core/deconv.py:321) as a compact device matrix
                sub = _DeviceBuffer(n * G * (4 if y_code == _lib.FDX_F32 else 8))
                owned.append(sub)
                gi32 = np.ascontiguousarray(gene_idx, dtype=np.int32)
                _lib.check(lib.fdx_gather_columns_dev(y_ptr, y_code, n, G_all, G_all, _lib.ptr_i32(gi32), G, sub.ptr, None))
                y_ptr = sub.ptr
            ldy = G

            # Step 4 runs here, under the leverage SVD (no data dependence between core/deconv.py:318 and :358)
            if big_k:
                from ..utils.graph import ckdtree_knn_adjacency
                t_graph = time.perf_counter()
                A = ckdtree_knn_adjacency(coords_host if coords_host is not None else _lib.tensor_to_host(coords), int(self.k_neighbors))
                self._graph = _lib.Graph.from_csr(A.indptr, A.indices, n)
                self._adjacency = A                      # (adjacency_: the host matrix itself)
            elif not graph_early:
                t_graph = time.perf_counter()
                gh = ctypes.c_void_p()
                _lib.check(lib.fdx_graph_build_dev(c_ptr, n, dim, g_method, g_k, g_radius, None, ctypes.byref(gh)))
                self._graph = _lib.Graph(gh.value)
            n_ties = 0
            if self.spatial_method == "knn" and self.knn_ties == "ckdtree" and not big_k:
                # the reference's tie order, on request: only when the device build met ties (this question waits for the
                # build, which otherwise completes behind the sketch) the lists come from the host restatement of scipy's
                # tree (csrc/kdtree_order.cpp) and the graph is rebuilt from the reference's own adjacency
                n_ties = self._graph.knn_ties()
                if n_ties:
                    self._graph.close()
                    self._graph = None
                    self._graph = reference_tie_graph(c_ptr, coords_host, n, dim, int(self.k_neighbors))
            t_lev = time.perf_counter()
            leverage = lev_job.result()
            t_done = time.perf_counter()

            # Step 2+3 tables: preprocessing mode and CountSketch Omega (core/deconv.py:326-352)
            log(f"Step 2: Preprocessing with method='{self.preprocess}'...")
            bucket, weight = countsketch_tables(G, self.sketch_dim, leverage, self.random_state)
            weight_y = weight_x = weight
            mode_y = mode_x = _lib.PRE_RAW
            if self.preprocess == "log_cpm":
                mode_y = _lib.PRE_LOG_CPM_SPARSE if y_sparse_rule else _lib.PRE_LOG_CPM
                mode_x = _lib.PRE_LOG_CPM
                log("  Y and X normalized to log-CPM space")
            elif self.preprocess == "pearson":
                if csr is not None:
                    if csr_colsum is None:
                        _, _, csr_colsum = csr.gene_moments(want_colsum=True)
                    sums = csr_colsum[gene_idx]
                else:
                    sums = np.empty(G, dtype=np.float64)
                    _lib.check(lib.fdx_column_sums_dev(y_ptr, y_code, n, G, ldy, _lib.ptr_f64(sums), None))
                mu_y = sums / n + 1e-6                                        # core/deconv.py:208,214
                mu_x = Xsel.mean(axis=0) + 1e-6                               # core/deconv.py:220
                weight_y = weight / np.sqrt(mu_y + mu_y ** 2 / 100.0)         # sigma^2 = mu + mu^2/theta, theta = 100
                weight_x = weight / np.sqrt(mu_x + mu_x ** 2 / 100.0)
                log("  Y and X transformed with uncentered Pearson residuals")
            else:
                log("  No preprocessing applied (raw)")
            log(f"Step 3: Sketching to {self.sketch_dim} dimensions...")
            log(f"  Compressed {G} genes -> {self.sketch_dim} dims")

            prm = _lib.FitParams()
            prm.sketch_dim = int(self.sketch_dim)
            if y_f64_math and csr is None and y_code == _lib.FDX_F32:
                mode_y |= _lib.PRE_F64_MATH
            prm.mode_y, prm.mode_x = mode_y, mode_x
            prm.k_neighbors = int(self.k_neighbors)
            prm.max_iter, prm.tol, prm.verbose = int(self.max_iter), float(self.tol), 1 if self.verbose else 0
            prm.rho_sparsity = float(self.rho_sparsity)
            prm.lambda_auto = 1 if self.lambda_spatial == "auto" else 0
            prm.lambda_spatial = 0.0 if prm.lambda_auto else float(self.lambda_spatial)
            prm.radius = 0.0
            prm.graph_method = _lib.GRAPH_GIVEN
            # "auto": the question "is any k-th neighbour tied?" is answered inside the fit, where the graph's counts are taken
            # over anyway - tie-free inputs pay nothing; on ties the call returns before the solve and the graph is rebuilt on the
            # reference's choice
            prm.stop_on_ties = 1 if (self.spatial_method == "knn" and self.knn_ties == "auto" and not big_k) else 0
            log("Step 4: Building spatial graph...")

            if output == "torch":
                import torch
                dev = Y.device if _is_torch_cuda(Y) else torch.device("cuda", torch.cuda.current_device())
                beta_t = torch.empty((n, K), dtype=torch.float64, device=dev)
                prop_t = torch.empty((n, K), dtype=torch.float64, device=dev)
                b_ptr, p_ptr = ctypes.c_void_p(beta_t.data_ptr()), ctypes.c_void_p(prop_t.data_ptr())
            else:
                bbuf, pbuf = _DeviceBuffer(n * K * 8), _DeviceBuffer(n * K * 8)
                owned += [bbuf, pbuf]
                b_ptr, p_ptr = bbuf.ptr, pbuf.ptr

            objs = np.zeros(max(int(self.max_iter), 1), dtype=np.float64)
            rels = np.zeros(max(int(self.max_iter), 1), dtype=np.float64)
            info = _lib.FitInfo()
            gh = ctypes.c_void_p(self._graph.handle.value)
            bucket32 = np.ascontiguousarray(bucket, dtype=np.int32)
            wy, wx = _lib.as_f64(weight_y), _lib.as_f64(weight_x)
            t_call = time.perf_counter()
            gi32 = np.ascontiguousarray(gene_idx, dtype=np.int32)

            def run_fit():
                if csr is not None:
                    _lib.check(lib.fdx_fit_csr_dev(ctypes.byref(csr.view), _lib.ptr_i32(gi32), G, _lib.ptr_f64(Xsel), K,
                                                   _lib.ptr_i32(bucket32), _lib.ptr_f64(wy), _lib.ptr_f64(wx), c_ptr, dim,
                                                   ctypes.byref(prm), ctypes.byref(gh), b_ptr, p_ptr, _lib.ptr_f64(objs),
                                                   _lib.ptr_f64(rels), ctypes.byref(info), None))
                else:
                    _lib.check(lib.fdx_fit_dev(y_ptr, y_code, n, G, ldy, _lib.ptr_f64(Xsel), K, _lib.ptr_i32(bucket32),
                                               _lib.ptr_f64(wy), _lib.ptr_f64(wx), c_ptr, dim, ctypes.byref(prm),
                                               ctypes.byref(gh), b_ptr, p_ptr, _lib.ptr_f64(objs), _lib.ptr_f64(rels),
                                               ctypes.byref(info), None))

            run_fit()
            ties_remedy_ms = 0.0
            if info.status == _lib.FIT_TIES:
                # ties under "auto": the reference's neighbour choice, then the fit proper.  The stopped call's sketch -> H stage is
                # still running on the device while the host builds the tree; the second call takes it over (info.carry) - the
                # rebuilt graph keeps the spot order.  The first graph stays alive until then: the running kernel reads its order.
                n_ties = int(info.knn_ties)
                log(f"k-NN ties on {n_ties} of {n} spots: rebuilding the graph on the reference's (cKDTree) neighbour choice")
                stale, carry = self._graph, info.carry
                self._graph = None
                try:
                    self._graph = reference_tie_graph(c_ptr, coords_host, n, dim, int(self.k_neighbors))
                    gh = ctypes.c_void_p(self._graph.handle.value)
                    prm.stop_on_ties = 0
                    prm.carry, carry = carry, None             # consumed by the call
                    ties_resolved_here = True
                    ties_remedy_ms = (time.perf_counter() - t_call) * 1e3
                    t_call = time.perf_counter()
                    run_fit()
                finally:
                    if carry:
                        lib.fdx_fit_carry_free(carry)
                    stale.close()
            else:
                ties_resolved_here = False
            t_ret = time.perf_counter()
            if os.environ.get("FDX_TRACE_HOST"):
                print(f"[fdx-host] python: entry->build {1e6 * (t_graph - t_entry):.0f} us, build call {1e6 * (t_lev - t_graph):.0f}, leverage wait "
                      f"{1e6 * (t_done - t_lev):.0f}, tables + setup {1e6 * (t_call - t_done):.0f}, fit call {1e6 * (t_ret - t_call):.0f}", file=sys.stderr)
            if output == "torch":
                self.beta_, self.proportions_ = beta_t, prop_t
            else:
                # A model that is fitted again and whose previous result arrays nobody else holds writes into them: releasing
                # 240 MB arrays and faulting in fresh ones costs ~25 ms of munmap / page zeroing per fit at 1M x 30 - more than
                # the device spends on the whole fit (unobservable: no other reference to the old arrays exists)
                self.beta_ = bbuf.to_host((n, K), out=self._recyclable("beta_", (n, K)))
                self.proportions_ = pbuf.to_host((n, K), out=self._recyclable("proportions_", (n, K)))
        finally:
            for b in owned:
                b.free()

        log(f"  Average neighbors per spot: {info.nnz / max(n, 1):.1f}")
        self.lambda_used_ = float(info.lambda_used)
        log(f"Step 5: {'Auto-tuned' if prm.lambda_auto else 'Using'} lambda = {self.lambda_used_:.4f}")
        log("Step 6: Solving via Block Coordinate Descent...")
        n_it = int(info.solve.n_iterations)
        objectives = [float(v) for v in objs[:info.solve.n_objectives]]
        if self.verbose:
            its = [t for t in range(n_it) if t % 10 == 0 or t == self.max_iter - 1]
            for t, obj in zip(its, objectives):
                print(f"Iteration {t}: objective = {obj:.6f}, rel_change = {rels[t]:.6e}")
            if info.solve.converged:
                print(f"Converged at iteration {n_it - 1}")
        self.info_ = {
            "converged": bool(info.solve.converged),
            "n_iterations": n_it,
            "final_objective": float(info.solve.final_objective),
            "objectives": objectives if self.verbose else [],
            "final_change": float(info.solve.final_change),
        }
        # additive (not in the reference): spots whose k-NN set is a choice - the k-th and (k+1)-th neighbours exactly
        # equidistant.  The reference takes whichever cKDTree.query meets first (utils/graph.py:60-63), which changes
        # with the order the spots are listed in; here the lower spot index wins.  Regular lattices tie on every spot.
        resolved = self.spatial_method == "knn" and (self.knn_ties == "ckdtree" or ties_resolved_here)
        self.info_["knn_ties"] = (n_ties if resolved else int(info.knn_ties)) if self.spatial_method == "knn" else 0
        if self.info_["knn_ties"] and not resolved:
            import warnings
            warnings.warn(
                f"k-NN ties: {self.info_['knn_ties']} of {n} spots have their k-th and (k+1)-th nearest neighbours at exactly "
                "the same distance (regular lattice?), so the neighbour graph depends on how ties are broken - here by "
                "spot index, in the reference by cKDTree's traversal, i.e. by the order the spots are listed in.  "
                "Proportions can differ from the reference's by a few 1e-4 (relative); knn_ties='auto' (the default) or "
                "'ckdtree' reproduce the reference's choice (host-side, about a second per million spots), "
                "spatial_method='grid' builds a tie-free graph on lattices.", UserWarning, stacklevel=2)
        # additive diagnostics (not in the reference): per-stage GPU milliseconds
        self.timings_ = {k: float(getattr(info, k)) for k in ("graph_ms", "sketch_ms", "gram_ms", "solve_ms", "finish_ms", "total_ms",
                                                             "prologue_ms", "span_ms")}
        self.timings_["sweep_ms"] = float(info.solve.sweep_ms)
        # ties under "auto": the stopped first call + the reference's lists (host tree) + the rebuilt graph; ahead of the fit proper
        self.timings_["ties_remedy_ms"] = ties_remedy_ms
        # host wall of the graph build call, and of the wait for the leverage SVD that ran beside it
        self.timings_["graph_ms"] = ((t_graph_done if graph_early else t_lev) - t_graph) * 1e3
        self.timings_["select_ms"] = t_sel * 1e3      # gene statistics on the device + HVG/marker ranking on the host
        self.timings_["leverage_wait_ms"] = (t_done - t_lev) * 1e3
        # the graph is built by its own call ahead of fdx_fit_dev, whose total_ms starts after it: one figure for the fit
        self.timings_["device_ms"] = self.timings_["total_ms"]
        # Accounting that tiles the wall time of this call: host_pre_ms (entry -> the graph build call: conversions, gene selection,
        # leverage job set-up) + span_ms (device, hipEvents: first kernel of the graph build -> end of the export =
        # prologue_ms + sketch_ms + gram_ms + solve_ms + finish_ms) + host_post_ms (fit_dev's return -> here);
        # total_ms is their sum, and what the wall has beyond it is the host's last synchronisation
        self.timings_["host_pre_ms"] = (t_graph - t_entry) * 1e3
        if graph_early:
            # gene selection active: the graph was queued on the side stream FIRST and the device span starts with the fit call -
            # gene statistics (device) + ranking (host) + leverage + tables all lie before it (select_ms is the part of it that
            # selects)
            self.timings_["host_pre_ms"] = (t_call - t_entry) * 1e3
        if ties_resolved_here or (resolved and n_ties):
            # the graph was rebuilt on the reference's tie order: the device span starts with the fit call that used it, and
            # everything before that call (first build, the stopped call, the host tree, the rebuild) is host time
            self.timings_["host_pre_ms"] = (t_call - t_entry) * 1e3
        self.timings_["host_post_ms"] = (time.perf_counter() - t_ret) * 1e3
        self.timings_["total_ms"] = self.timings_["host_pre_ms"] + self.timings_["span_ms"] + self.timings_["host_post_ms"]
        self._fitted = True
        log(f"  Converged: {self.info_['converged']}")
        log(f"  Iterations: {self.info_['n_iterations']}")
        log("FlashDeconv: Done!")
        return self

    def _recyclable(self, name, shape):
        """The model's previous result array `name` when it can take the new result in place: same shape, float64, owns its
        memory, and referenced by nobody but the model (attribute + this frame's variable + getrefcount's argument)."""
        old = getattr(self, name, None)
        ok = (isinstance(old, np.ndarray) and old.shape == tuple(shape) and old.dtype == np.float64 and old.flags.c_contiguous
              and old.flags.owndata and old.flags.writeable and sys.getrefcount(old) <= 3)
        return old if ok else None

    def _graph_request(self, coords, coords_host):
        """(method, k, radius) of the graph build for this model's spatial_method (utils/graph.py:175-212)."""
        if self.spatial_method == "knn":
            return _lib.GRAPH_KNN, int(self.k_neighbors), 0.0
        if self.spatial_method == "radius":
            if self.radius is None:
                raise ValueError("radius must be specified for radius method")
            return _lib.GRAPH_RADIUS, 0, float(self.radius)
        if self.spatial_method == "grid":  # radius = 1.5 x median nearest-neighbour distance (utils/graph.py:163-170)
            if coords.shape[0] < 2:
                return _lib.GRAPH_KNN, 0, 0.0
            from ..utils.graph import grid_radius
            ch = coords_host if coords_host is not None else _lib.tensor_to_host(coords)
            return _lib.GRAPH_RADIUS, 0, grid_radius(ch)
        raise ValueError(f"Unknown method: {self.spatial_method}. Choose from 'knn', 'radius', 'grid'.")

    def fit_transform(self, Y, X, coords, **kwargs):
        self.fit(Y, X, coords, **kwargs)
        return self.proportions_

    # ------------------------------------------------------------------ getters (core/deconv.py:436-512)
    def _require_fitted(self):
        if not self._fitted:
            raise RuntimeError("Model has not been fitted. Call fit() first.")

    def get_cell_type_proportions(self):
        self._require_fitted()
        return self.proportions_

    def get_abundances(self):
        self._require_fitted()
        return self.beta_

    def get_dominant_cell_type(self):
        self._require_fitted()
        if hasattr(self.proportions_, "argmax") and not isinstance(self.proportions_, np.ndarray):
            return self.proportions_.argmax(dim=1)
        return np.argmax(self.proportions_, axis=1)

    def summary(self):
        if not self._fitted:
            return {"fitted": False}
        return {
            "fitted": True,
            "n_spots": self.n_spots_,
            "n_cell_types": self.n_cell_types_,
            "n_genes_used": len(self.gene_idx_),
            "sketch_dim": self.sketch_dim,
            "lambda_spatial": self.lambda_used_,
            "rho_sparsity": self.rho_sparsity,
            "preprocess_method": self.preprocess,
            "converged": self.info_["converged"],
            "n_iterations": self.info_["n_iterations"],
            "final_objective": self.info_["final_objective"],
        }

    def __repr__(self):
        status = "fitted" if self._fitted else "not fitted"
        return f"FlashDeconv(sketch_dim={self.sketch_dim}, lambda_spatial={self.lambda_spatial}, status={status})"
