"""Gene selection and leverage scores: the reference's ``flashdeconv/utils/genes.py`` interface.

    compute_leverage_scores   <- utils/genes.py:238-290  (GPU: one-sided Jacobi SVD, csrc/leverage_kernels.cpp)
    select_markers            <- utils/genes.py:148-235  (host: K x G table logic; "diff", "ratio", "specificity")
    select_hvg                <- utils/genes.py:18-145   (per-gene moments on the GPU, binning of the G-vector on the host)
    select_informative_genes  <- utils/genes.py:293-341
"""
import ctypes

import numpy as np
from scipy import sparse

from .. import _lib


def compute_leverage_scores(X, regularization=1e-6):
    X = _lib.as_f64(X)
    K, G = X.shape
    lev = np.empty(G, dtype=np.float64)
    _lib.require_gpu()
    _lib.check(_lib.load().fdx_leverage_scores(_lib.ptr_f64(X), K, G, float(regularization), _lib.ptr_f64(lev)))
    return lev


class LeverageJob:
    """compute_leverage_scores split in two: the constructor enqueues the single-workgroup SVD on a side stream,
    ``result()`` waits for it.  FlashDeconv.fit builds the spatial graph in between."""

    def __init__(self, X, regularization=1e-6, queue_async=True):
        X = _lib.as_f64(X)
        self.G = X.shape[1]
        _lib.require_gpu()
        self._job = ctypes.c_void_p()
        # queue_async: the library's helper thread queues the upload and the launches while this thread goes on (to a graph build);
        # False for a caller that collects the result at once
        _lib.check(_lib.load().fdx_leverage_begin_opt(_lib.ptr_f64(X), X.shape[0], X.shape[1], float(regularization),
                                                      1 if queue_async else 0, ctypes.byref(self._job)))

    def result(self, keep_x=False):
        """The scores.  keep_x: the job's device copy of X is kept (``self.x_dev``, until ``release_x()``) for a fit that needs
        the same matrix on the device."""
        if self._job is None:
            raise RuntimeError("leverage job already collected")
        lev = np.empty(self.G, dtype=np.float64)
        job, self._job = self._job, None
        if keep_x:
            xd = ctypes.c_void_p()
            _lib.check(_lib.load().fdx_leverage_end_keep(job, _lib.ptr_f64(lev), ctypes.byref(xd)))
            self.x_dev = xd if xd.value else None
        else:
            _lib.check(_lib.load().fdx_leverage_end(job, _lib.ptr_f64(lev)))
        return lev

    def release_x(self):
        xd, self.x_dev = getattr(self, "x_dev", None), None
        if xd is not None:
            _lib.load().fdx_free(xd)

    def __del__(self):
        if getattr(self, "_job", None) is not None:
            try:
                self.result()
            except Exception:
                pass
        try:
            self.release_x()
        except Exception:
            pass


def select_markers(X, n_markers=50, method="diff"):
    """Union over cell types of the ``n_markers`` most specific genes (max minus second max of the
    row-normalised signatures); returns (marker_idx, marker_assignments) like the reference."""
    X = np.asarray(X, dtype=np.float64)
    K, G = X.shape
    if n_markers < 0:
        raise ValueError(f"n_markers must be non-negative, got {n_markers}")
    if n_markers == 0 or K == 0:
        return np.array([], dtype=np.intp), np.array([], dtype=np.intp)
    if K == 1:
        idx = np.arange(min(n_markers, G))
        return idx, np.zeros(len(idx), dtype=np.intp)
    denom = X.sum(axis=1, keepdims=True) + 1e-10
    frac = None
    if method == "diff" and np.isfinite(X).all():     # utils/genes.py:197-200: largest minus second largest fraction per gene
        # One pass over the cell types with a running (largest, second largest, owner) per gene instead of the (K, G) fraction
        # matrix, its strided argmax and a masked second maximum (3.2 -> 1 ms at 30 x 20000; this runs beside the device's gene
        # statistics and must not outlast them).  The same divisions, the same selections: a strictly larger value takes over
        # (np.argmax keeps the FIRST maximum), a tie leaves the owner and makes the second largest equal the largest, as the
        # sorted column does.
        # (selections only - maximum / minimum of non-negative finite values - so the bits are those of the matrix form; masked
        # copies cost 20 x a plain pass in numpy, hence the arithmetic owner update)
        m1 = np.divide(X[0], denom[0])
        m2 = np.full(G, -np.inf)
        owner = np.zeros(G, dtype=np.int64)
        v = np.empty(G)
        lo = np.empty(G)
        bigger = np.empty(G, dtype=bool)
        step = np.empty(G, dtype=np.int64)
        for k in range(1, K):
            np.divide(X[k], denom[k], out=v)
            np.greater(v, m1, out=bigger)
            np.minimum(m1, v, out=lo)                 # what does not become the largest ...
            np.maximum(m2, lo, out=m2)                # ... competes for second place (a tie: the same value twice)
            np.maximum(m1, v, out=m1)
            np.subtract(k, owner, out=step)           # owner = k where v took over
            np.multiply(step, bigger, out=step)
            np.add(owner, step, out=owner)
        specificity = m1 - m2
    else:
        frac = X / denom
        owner = np.argmax(frac, axis=0)
        if method == "diff":                          # NaN / inf somewhere: the plain form (np.argmax's and max's NaN rules)
            cols = np.arange(G)
            largest = frac[owner, cols]
            rest = frac.copy()
            rest[owner, cols] = -np.inf
            specificity = largest - rest.max(axis=0)
    if method == "diff":
        pass
    elif method == "ratio":                           # utils/genes.py:202-206: largest fraction over the mean of the others
        largest = frac.max(axis=0)
        specificity = largest / ((frac.sum(axis=0) - largest) / (K - 1) + 1e-10)
    elif method == "specificity":                     # utils/genes.py:208-211: tau score
        largest = frac.max(axis=0)
        specificity = np.sum(1 - frac / (largest + 1e-10), axis=0) / (K - 1)
    else:
        raise ValueError(f"Unknown method: {method}")
    chosen, assign = [], []
    # the genes of every type in ascending order (= np.flatnonzero(owner == k)) from one stable sort
    by_owner = np.argsort(owner.astype(np.uint8 if K <= 256 else np.int64), kind="stable")   # (8-bit keys: a counting sort)
    ends = np.cumsum(np.bincount(owner, minlength=K))
    for k in range(K):
        mine = by_owner[(ends[k - 1] if k else 0):ends[k]]
        if mine.size:
            pick = mine[np.argsort(specificity[mine])[::-1][:n_markers]]
        else:
            row = frac[k] if frac is not None else X[k] / denom[k]
            pick = np.argsort(row)[::-1][:n_markers]
        chosen.extend(pick.tolist())
        assign.extend([k] * len(pick))
    return np.unique(chosen), np.array(assign)


def gene_moments_device(y_ptr, y_code, n, G, ldy):
    """Per-gene mean / ddof-1 variance of log1p(CPM-10k) for a matrix already in HBM (utils/genes.py:52-102)."""
    mean = np.empty(G, dtype=np.float64)
    var = np.empty(G, dtype=np.float64)
    _lib.check(_lib.load().fdx_gene_moments_dev(y_ptr, y_code, n, G, ldy, _lib.ptr_f64(mean), _lib.ptr_f64(var), None))
    return mean, var


def _gene_moments(Y):
    if sparse.issparse(Y) or _lib.is_torch_sparse_csr(Y):      # sparse branch (utils/genes.py:52-83) on the CSR arrays
        _lib.require_gpu()
        csr = _lib.CsrOnDevice.from_scipy(Y) if sparse.issparse(Y) else _lib.CsrOnDevice.from_torch(Y)
        try:
            mean, var, _ = csr.gene_moments()
        finally:
            csr.free()
        return mean, var
    _lib.require_gpu()
    lib = _lib.load()
    Yh = np.asarray(Y)
    ptr, code = _lib.upload_matrix(Yh)
    try:
        return gene_moments_device(ptr, code, Yh.shape[0], Yh.shape[1], Yh.shape[1])
    finally:
        lib.fdx_free(ptr)


def select_hvg(Y, n_top=2000, min_mean=0.0125, max_mean=3.0, min_disp=0.5):
    n_genes = Y.shape[1]
    if n_genes <= n_top:
        # Fewer genes than requested: every branch of the reference (utils/genes.py:135-145) returns all of them.
        return np.arange(n_genes)
    mean, var = _gene_moments(Y)
    return _hvg_from_moments(mean, var, n_top, min_mean, max_mean, min_disp)


def _hvg_from_moments(mean, var, n_top, min_mean, max_mean, min_disp):
    """Seurat-v3-style binned z-score of the variance, then top-n (utils/genes.py:104-145): libfdx's host restatement of the
    numpy arithmetic (csrc/hvg_rank.cpp: 0.15 ms instead of 0.9 at 20000 genes, between two device phases of a fit); numpy
    itself where exactly equal dispersions straddle the cut (the order its sort leaves them in decides) or one is NaN."""
    mean = np.ascontiguousarray(mean, dtype=np.float64)
    var = np.ascontiguousarray(var, dtype=np.float64)
    G = len(mean)
    idx = np.empty(max(min(int(n_top), G), 1), dtype=np.int64)
    n_out, amb = ctypes.c_int32(0), ctypes.c_int32(0)
    pos = np.sort(mean[mean > 0])                        # (numpy's vectorised sort: a tenth of std::sort's time)
    _lib.check(_lib.load().fdx_hvg_from_moments(_lib.ptr_f64(mean), _lib.ptr_f64(var), G, _lib.ptr_f64(pos), len(pos), min(int(n_top), G),
                                                float(min_mean), float(max_mean), float(min_disp), _lib.ptr_i64(idx), ctypes.byref(n_out),
                                                ctypes.byref(amb)))
    if not amb.value:
        return idx[:n_out.value].astype(np.intp)
    return _hvg_from_moments_numpy(mean, var, n_top, min_mean, max_mean, min_disp)


def _hvg_from_moments_numpy(mean, var, n_top, min_mean, max_mean, min_disp):
    """The same in numpy (the reference's own operations)."""
    G = len(mean)
    disp = np.zeros(G)
    pos = mean[mean > 0]
    if len(pos) >= 2:
        # np.percentile selects order statistics: on the sorted vector it returns the same bits and skips its 21-way partition
        edges = np.unique(np.percentile(np.sort(pos), np.linspace(0, 100, 21)))
        if len(edges) >= 2:
            # np.digitize(mean, edges) = number of edges <= mean (NaN sorts past the last edge): 21 vector compares
            below = np.zeros(G, dtype=np.int64)
            for e in edges:
                below += mean >= e
            below[np.isnan(mean)] = len(edges)
            which = np.clip(below - 1, 0, len(edges) - 2).astype(np.uint8)
            # one stable sort by bin instead of a boolean mask per bin: a bin's slice lists its genes in ascending index
            # order, i.e. var[order[s:e]] is the very array var[which == b] - same mean, same std, bit for bit
            order = np.argsort(which, kind="stable")
            ends = np.cumsum(np.bincount(which, minlength=len(edges) - 1))
            var_by_bin = var[order]
            s0 = 0
            for e0 in ends:
                if e0 - s0 > 1:
                    v = var_by_bin[s0:e0]
                    disp[order[s0:e0]] = (v - v.mean()) / (v.std() + 1e-10)
                s0 = e0
    ok = np.flatnonzero((mean >= min_mean) & (mean <= max_mean) & (disp >= min_disp))
    if len(ok) < n_top:
        pick = np.argsort(disp)[::-1][:n_top]
    else:
        pick = ok[np.argsort(disp[ok])[::-1][:n_top]]
    return np.sort(pick)


def select_informative_genes(Y, X, n_hvg=2000, n_markers_per_type=50):
    hvg = select_hvg(Y, n_top=n_hvg)
    markers, _ = select_markers(X, n_markers=n_markers_per_type)
    gene_idx = np.union1d(hvg, markers).astype(np.intp)
    if len(gene_idx) == 0:
        raise ValueError("No genes selected. Increase n_hvg or n_markers_per_type.")
    return gene_idx, compute_leverage_scores(np.asarray(X)[:, gene_idx])
