"""Duck-typed AnnData helpers with the reference's behaviour (``flashdeconv/io/loader.py:15-311``).

Nothing here is on the hot path: it pulls matrices out of AnnData-like objects (``.X``, ``.layers``, ``.obsm``, ``.obs``,
``.var_names``, ``.obs_names``, ``.n_obs``) and writes the result back.  Matrices that already live on the GPU (CUDA
``torch`` tensors, dense or ``sparse_csr``) stay there: the per-cell-type signatures are formed by
``fdx_type_sums_dev`` / ``fdx_type_sums_csr_dev``, the gene alignment is a device-side column selection, and the spot matrix
goes to ``FlashDeconv.fit`` as the tensor it is - only the K x G signature table and the result come to the host.
"""
import ctypes

import numpy as np
from scipy import sparse

from .. import _lib


def _is_cuda_tensor(x):
    return type(x).__module__.split(".")[0] == "torch" and getattr(x, "is_cuda", False)


def _type_sums_device(expr, codes, K, mean):
    """(K, G) float64 signatures of a CUDA tensor (dense or sparse_csr): rows added per type in ascending row order."""
    import torch
    lib = _lib.load()
    n, G = expr.shape
    order = np.argsort(codes, kind="stable").astype(np.int32)                 # cells by type, ascending row inside a type
    off = np.concatenate([[0], np.cumsum(np.bincount(codes, minlength=K))]).astype(np.int32)
    dev = expr.device
    rows_d = torch.from_numpy(order).to(dev)
    off_d = torch.from_numpy(off).to(dev)
    X = torch.empty((K, G), dtype=torch.float64, device=dev)
    st = ctypes.c_void_p(torch.cuda.current_stream(dev).cuda_stream)
    if _lib.is_torch_sparse_csr(expr):
        csr = _lib.CsrOnDevice.from_torch(expr)
        try:
            _lib.check(lib.fdx_type_sums_csr_dev(ctypes.byref(csr.view), ctypes.c_void_p(rows_d.data_ptr()),
                                                 ctypes.c_void_p(off_d.data_ptr()), K, 1 if mean else 0,
                                                 ctypes.c_void_p(X.data_ptr()), st))
            torch.cuda.synchronize(dev)
        finally:
            csr.free()
    else:
        Yd = expr if expr.dtype in (torch.float32, torch.float64) else expr.to(torch.float64)
        Yd = Yd.contiguous()
        _lib.check(lib.fdx_type_sums_dev(ctypes.c_void_p(Yd.data_ptr()), _lib.FDX_F32 if Yd.dtype == torch.float32 else _lib.FDX_F64,
                                         n, G, G, ctypes.c_void_p(rows_d.data_ptr()), ctypes.c_void_p(off_d.data_ptr()), K,
                                         1 if mean else 0, ctypes.c_void_p(X.data_ptr()), st))
    return X.cpu().numpy()


def _select_columns_device(Y, idx):
    """Y[:, idx] for a CUDA tensor, on the device.  sparse_csr: entries of other columns dropped, columns renumbered (the
    rows of the result are in general not column-sorted; the CSR kernels take that)."""
    import torch
    dev = Y.device
    if not _lib.is_torch_sparse_csr(Y):
        return Y.index_select(1, torch.as_tensor(np.asarray(idx, dtype=np.int64), device=dev))
    n, G = Y.shape
    lut = torch.full((G,), -1, dtype=torch.int64, device=dev)
    lut[torch.as_tensor(np.asarray(idx, dtype=np.int64), device=dev)] = torch.arange(len(idx), dtype=torch.int64, device=dev)
    crow, col, val = Y.crow_indices().to(torch.int64), Y.col_indices().to(torch.int64), Y.values()
    new_col = lut[col]
    keep = new_col >= 0
    kept_before = torch.zeros(col.numel() + 1, dtype=torch.int64, device=dev)
    torch.cumsum(keep.to(torch.int64), 0, out=kept_before[1:])
    return torch.sparse_csr_tensor(kept_before[crow], new_col[keep].to(torch.int32), val[keep], size=(n, len(idx)), device=dev)


def load_spatial_data(adata, layer=None, coord_key="spatial"):
    """(Y, coords, gene_names) from a spatial AnnData (io/loader.py:15-70)."""
    Y = adata.layers[layer] if layer is not None else adata.X
    if coord_key in adata.obsm:
        coords = np.array(adata.obsm[coord_key])
    elif "X_spatial" in adata.obsm:
        coords = np.array(adata.obsm["X_spatial"])
    elif "x" in adata.obs and "y" in adata.obs:
        coords = np.column_stack([adata.obs["x"], adata.obs["y"]])
    elif "array_row" in adata.obs and "array_col" in adata.obs:
        coords = np.column_stack([adata.obs["array_row"], adata.obs["array_col"]])
    else:
        raise ValueError(f"Could not find spatial coordinates. Expected key '{coord_key}' in adata.obsm or 'x'/'y' in adata.obs")
    return Y, coords, np.array(adata.var_names)


def load_reference(adata_ref, cell_type_key="cell_type", layer=None, method="mean"):
    """Per-cell-type mean (or sum) signatures (K, G), sorted unique type names, gene names (io/loader.py:73-140)."""
    expr = adata_ref.layers[layer] if layer is not None else adata_ref.X
    if cell_type_key not in adata_ref.obs:
        raise ValueError(f"Cell type key '{cell_type_key}' not found in adata_ref.obs")
    if method not in ("mean", "sum"):
        raise ValueError(f"Unknown aggregation method: {method}")
    labels = np.array(adata_ref.obs[cell_type_key])
    if _is_cuda_tensor(expr):                                               # the matrix stays on the GPU
        names, codes = np.unique(labels, return_inverse=True)
        return _type_sums_device(expr, codes.ravel(), len(names), method == "mean"), names, np.array(adata_ref.var_names)
    names = np.unique(labels)
    X = np.zeros((len(names), expr.shape[1]), dtype=np.float64)
    for i, name in enumerate(names):
        rows = expr[labels == name]
        agg = rows.mean(axis=0) if method == "mean" else rows.sum(axis=0)
        X[i] = np.asarray(agg).ravel()
    return X, names, np.array(adata_ref.var_names)


def align_genes(Y, X, genes_spatial, genes_ref):
    """Restrict both matrices to the shared genes, in sorted-name order, first occurrence wins (io/loader.py:143-194)."""
    common = np.intersect1d(genes_spatial, genes_ref)
    if len(common) == 0:
        raise ValueError("No common genes found between spatial data and reference")
    first_s, first_r = {}, {}
    for i, g in enumerate(genes_spatial):
        first_s.setdefault(g, i)
    for i, g in enumerate(genes_ref):
        first_r.setdefault(g, i)
    si = np.array([first_s[g] for g in common])
    ri = np.array([first_r[g] for g in common])
    if _is_cuda_tensor(Y):
        return _select_columns_device(Y, si), X[:, ri], common
    return Y[:, si], X[:, ri], common


def result_to_anndata(beta, adata, cell_type_names=None, key_added="flashdeconv"):
    """obsm[key] = DataFrame of proportions, obs[key_dominant] = categorical dominant type (io/loader.py:197-258)."""
    import pandas as pd
    if beta.ndim != 2:
        raise ValueError(f"beta must be 2D, got shape {beta.shape}")
    if beta.shape[0] != adata.n_obs:
        raise ValueError(f"beta rows must match adata.n_obs, got beta.shape[0]={beta.shape[0]} and adata.n_obs={adata.n_obs}")
    cols = np.asarray(cell_type_names) if cell_type_names is not None else np.array([f"CellType_{i}" for i in range(beta.shape[1])])
    if len(cols) != beta.shape[1]:
        raise ValueError(f"Length of cell_type_names ({len(cols)}) must match beta.shape[1] ({beta.shape[1]})")
    adata.obsm[key_added] = pd.DataFrame(beta, index=adata.obs_names, columns=cols)
    adata.obs[f"{key_added}_dominant"] = pd.Categorical(cols[np.argmax(beta, axis=1)], categories=cols)
    return adata


def prepare_data(adata_st, adata_ref, cell_type_key="cell_type", layer_st=None, layer_ref=None, spatial_coord_key="spatial"):
    """(Y, X, coords, cell_type_names, gene_names) aligned on the shared genes (io/loader.py:261-311)."""
    Y, coords, genes_st = load_spatial_data(adata_st, layer=layer_st, coord_key=spatial_coord_key)
    X, names, genes_ref = load_reference(adata_ref, cell_type_key=cell_type_key, layer=layer_ref)
    Y, X, genes = align_genes(Y, X, genes_st, genes_ref)
    if sparse.issparse(Y):
        Y = Y.tocsr()
    return Y, X, coords, names, genes
