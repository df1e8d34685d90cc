"""Spatial neighbour graphs from spot coordinates: same functions as the reference's
``flashdeconv/utils/graph.py``, built on the GPU (csrc/graph_kernels.cpp) and returned as the same
``scipy.sparse.csr_matrix`` of ones (float64, int32 indices, sorted).

    build_knn_graph      <- utils/graph.py:25-83
    build_radius_graph   <- utils/graph.py:86-133
    build_grid_graph     <- utils/graph.py:136-172
    coords_to_adjacency  <- utils/graph.py:175-212

Coordinates with 1, 2 or 3 columns are binned on a grid; k-NN graphs also take 4 to 8 columns (exhaustive search, up to
262144 spots).  On exactly tied distances (regular lattices) the
k-th neighbour is chosen by the lower spot index, whereas the reference inherits cKDTree's traversal order;
``ties="ckdtree"`` reproduces that order (``ckdtree_knn_adjacency``: a host restatement of scipy's tree, used only when
the device build reports ties).
"""
import numpy as np
from scipy import sparse

from .. import _lib


def _validate_coords(coords, knn=False):
    if coords.ndim != 2 or coords.shape[1] == 0:                        # utils/graph.py:16-22
        raise ValueError(f"coords must be 2D with at least 1 coordinate dimension, got shape {coords.shape}")
    check_coord_dims(coords.shape[0], coords.shape[1], knn)


def check_coord_dims(n, dim, knn):
    """What the device builders take (csrc/graph_kernels.cpp): a grid over up to three axes; k-NN graphs of points with 4 to 8
    coordinates by exhaustive search (up to 262144 spots).  The reference's cKDTree takes any dimension (utils/graph.py:16-22)."""
    if dim > 3 and not knn:
        raise ValueError(f"coords has {dim} dimensions: radius / grid graphs are built for 1 to 3 coordinate dimensions "
                         "(k-NN graphs for up to 8)")
    if dim > 8:
        raise ValueError(f"coords has {dim} dimensions: k-NN graphs are built for 1 to 8 coordinate dimensions")
    if dim > 3 and n > (1 << 18):
        raise ValueError(f"coords has {dim} dimensions and {n} spots: above 3 dimensions the k-NN search is exhaustive, "
                         "at most 262144 spots")


def _to_csr(graph, n):
    indptr, indices = graph.to_csr_arrays()
    data = np.ones(len(indices), dtype=np.float64)
    return sparse.csr_matrix((data, indices, indptr.astype(np.int32) if len(indices) < 2**31 - 1 else indptr), shape=(n, n))


def knn_graph_handle(coords, k=6):
    """Device graph handle for the k-NN graph (used by FlashDeconv.fit to avoid a host round trip)."""
    coords = np.asarray(coords, dtype=np.float64)
    _validate_coords(coords, knn=True)
    _lib.require_gpu()
    return _lib.Graph.from_coords_knn(coords, k)


_ckdtree_probe = None


def _ckdtree_restatement_matches_scipy():
    """The restated traversal order was written against scipy 1.15's cKDTree (build: leafsize 16, sliding midpoint replaced by
    the median via introselect; query: near child first).  Another scipy / libstdc++ could order ties differently, so the
    first use replays a small all-ties case (a 19 x 17 lattice in shuffled order) against the installed scipy and warns on a
    mismatch instead of silently returning a different 'reference' order.  Cached per process."""
    global _ckdtree_probe
    if _ckdtree_probe is None:
        from scipy.spatial import cKDTree
        rs = np.random.RandomState(5)
        gx, gy = np.meshgrid(np.arange(19.0), np.arange(17.0))
        pts = np.ascontiguousarray(np.stack([gx.ravel(), gy.ravel()], axis=1)[rs.permutation(19 * 17)])
        idx = np.empty((len(pts), 7), dtype=np.int64)
        _lib.check(_lib.load().fdx_ckdtree_knn(_lib.ptr_f64(pts), len(pts), 2, 7, idx.ctypes.data, None))
        _ckdtree_probe = bool(np.array_equal(idx, cKDTree(pts).query(pts, k=7)[1]))
        if not _ckdtree_probe:
            import warnings
            warnings.warn("knn ties='ckdtree': libfdx's restatement of scipy.spatial.cKDTree's tie order does not reproduce the "
                          "installed scipy on a lattice probe (written against scipy 1.15); tied neighbours may be chosen "
                          "differently from what the reference would choose on this installation.", UserWarning, stacklevel=3)
    return _ckdtree_probe


def ckdtree_knn_lists(coords, k):
    """(n, k + 1) neighbour indices exactly as ``cKDTree(coords).query(coords, k=k+1)`` returns them (utils/graph.py:60-63),
    ties included: libfdx's host restatement of scipy's tree build and traversal order (csrc/kdtree_order.cpp), checked once
    per process against the installed scipy (``_ckdtree_restatement_matches_scipy``)."""
    _ckdtree_restatement_matches_scipy()
    coords = np.ascontiguousarray(coords, dtype=np.float64)
    n, dim = coords.shape
    kk = min(int(k), n - 1) + 1
    idx = np.empty((n, kk), dtype=np.int64)
    _lib.check(_lib.load().fdx_ckdtree_knn(_lib.ptr_f64(coords), n, dim, kk, idx.ctypes.data, None))
    return idx


def ckdtree_knn_lists_rows(coords, k, rows):
    """Rows `rows` of ``ckdtree_knn_lists``: the tree of ALL points, the queries of the listed ones only (a spot shard asks for
    its own rows and its band)."""
    _ckdtree_restatement_matches_scipy()
    coords = np.ascontiguousarray(coords, dtype=np.float64)
    rows = np.ascontiguousarray(rows, dtype=np.int64)
    n, dim = coords.shape
    kk = min(int(k), n - 1) + 1
    idx = np.empty((len(rows), kk), dtype=np.int64)
    _lib.check(_lib.load().fdx_ckdtree_knn_rows(_lib.ptr_f64(coords), n, dim, kk, _lib.ptr_i64(rows), len(rows), idx.ctypes.data))
    return idx


def ckdtree_knn_adjacency(coords, k):
    """The reference's k-NN adjacency from those lists: self dropped, ones, A + A^T, binary (utils/graph.py:66-81)."""
    idx = ckdtree_knn_lists(coords, k)
    n, kk = idx.shape
    rows = np.repeat(np.arange(n), kk)
    cols = idx.ravel()
    keep = rows != cols
    A = sparse.csr_matrix((np.ones(int(keep.sum())), (rows[keep], cols[keep])), shape=(n, n))
    A = (A + A.T).tocsr()
    A.data[:] = 1.0
    A.sort_indices()
    return A


def build_knn_graph(coords, k=6, include_self=False, ties="index"):
    """``ties``: "index" - equidistant candidates for the k-th place are taken by ascending spot index (device rule);
    "ckdtree" - as the reference's cKDTree query takes them (only looked at when the device build found ties)."""
    if ties not in ("index", "ckdtree"):
        raise ValueError(f"Unknown ties rule: {ties}. Choose from 'index', 'ckdtree'.")
    coords = np.asarray(coords, dtype=np.float64)
    _validate_coords(coords, knn=True)
    n = coords.shape[0]
    if min(k, n - 1) <= 0:                                               # utils/graph.py:51-57
        if include_self and n > 0:
            return sparse.eye(n, dtype=np.float64, format="csr")
        return sparse.csr_matrix((n, n), dtype=np.float64)
    g = knn_graph_handle(coords, k)
    try:
        if ties == "ckdtree" and g.knn_ties() > 0:
            A = ckdtree_knn_adjacency(coords, k)
        else:
            A = _to_csr(g, n)
    finally:
        g.close()
    if include_self:
        A = (A + sparse.eye(n, dtype=np.float64, format="csr")).tocsr()
        A.data[:] = 1.0
    return A


def radius_graph_handle(coords, radius):
    coords = np.asarray(coords, dtype=np.float64)
    _validate_coords(coords)
    _lib.require_gpu()
    return _lib.Graph.from_coords_radius(coords, radius)


def build_radius_graph(coords, radius, include_self=False):
    coords = np.asarray(coords, dtype=np.float64)
    _validate_coords(coords)
    n = coords.shape[0]
    g = radius_graph_handle(coords, radius)
    try:
        A = _to_csr(g, n)
    finally:
        g.close()
    if A.nnz == 0:                                                       # utils/graph.py:117-121
        if include_self and n > 0:
            return sparse.eye(n, dtype=np.float64, format="csr")
        return sparse.csr_matrix((n, n), dtype=np.float64)
    if include_self:
        A = (A + sparse.eye(n, dtype=np.float64)).tocsr()
    return A


def grid_radius(coords):
    """1.5 x the median nearest-neighbour distance (utils/graph.py:163-170)."""
    coords = _lib.as_f64(coords)
    dist = np.empty(coords.shape[0], dtype=np.float64)
    _lib.require_gpu()
    _lib.check(_lib.load().fdx_nearest_distance(_lib.ptr_f64(coords), coords.shape[0], coords.shape[1], _lib.ptr_f64(dist)))
    return float(np.median(dist)) * 1.5


def build_grid_graph(coords, grid_spacing=None):
    coords = np.asarray(coords, dtype=np.float64)
    _validate_coords(coords)
    n = coords.shape[0]
    if n <= 1:                                                           # utils/graph.py:160-161
        return sparse.csr_matrix((n, n), dtype=np.float64)
    radius = grid_radius(coords) if grid_spacing is None else grid_spacing * 1.5
    return build_radius_graph(coords, radius)


def coords_to_adjacency(coords, method="knn", k=6, radius=None):
    if method == "knn":
        return build_knn_graph(coords, k=k)
    if method == "radius":
        if radius is None:
            raise ValueError("radius must be specified for radius method")
        return build_radius_graph(coords, radius=radius)
    if method == "grid":
        return build_grid_graph(coords)
    raise ValueError(f"Unknown method: {method}")


def coords_to_graph_handle(coords, method="knn", k=6, radius=None):
    """Same dispatch as coords_to_adjacency but returns the device handle (no CSR materialised on the host)."""
    coords = np.asarray(coords, dtype=np.float64)
    _validate_coords(coords, knn=method == "knn")
    n = coords.shape[0]
    if method == "knn":
        return knn_graph_handle(coords, k)
    if method == "radius":
        if radius is None:
            raise ValueError("radius must be specified for radius method")
        return radius_graph_handle(coords, radius)
    if method == "grid":
        if n <= 1:
            return knn_graph_handle(coords, 0)
        return radius_graph_handle(coords, grid_radius(coords))
    raise ValueError(f"Unknown method: {method}")
