"""Duck-typed AnnData helpers with the reference's behaviour (``flashdeconv/io/loader.py:15-311``).

Nothing here is on the hot path: it pulls matrices out of AnnData-like objects (``.X``, ``.layers``, ``.obsm``, ``.obs``,
``.var_names``, ``.obs_names``, ``.n_obs``) and writes the result back.  Out of scope to accelerate.
"""
import numpy as np
from scipy import sparse


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
