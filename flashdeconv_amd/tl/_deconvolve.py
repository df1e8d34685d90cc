"""``deconvolve(adata_st, adata_ref, ...)``: the reference's scanpy-style entry (``flashdeconv/tl/_deconvolve.py:6-174``)
forwarding the same keyword arguments to the MI355X ``FlashDeconv``."""


def deconvolve(adata_st, adata_ref, cell_type_key="cell_type", *, sketch_dim=512, lambda_spatial="auto", rho_sparsity=0.01,
               n_hvg=2000, n_markers_per_type=50, spatial_method="knn", k_neighbors=6, radius=None, preprocess="log_cpm",
               layer_st=None, layer_ref=None, spatial_key="spatial", key_added="flashdeconv", random_state=0, copy=False):
    """Writes ``.obsm[key_added]`` (proportions DataFrame), ``.obs[key_added + '_dominant']`` and
    ``.uns[key_added + '_params']``; returns the modified copy when ``copy=True``, else ``None``."""
    from ..core.deconv import FlashDeconv
    from ..io import prepare_data, result_to_anndata

    adata = adata_st.copy() if copy else adata_st
    Y, X, coords, names, _ = prepare_data(adata, adata_ref, cell_type_key=cell_type_key, layer_st=layer_st,
                                          layer_ref=layer_ref, spatial_coord_key=spatial_key)
    # like the reference, max_iter / tol / verbose are not exposed here (tl/_deconvolve.py:132-144)
    model = FlashDeconv(sketch_dim=sketch_dim, lambda_spatial=lambda_spatial, rho_sparsity=rho_sparsity, n_hvg=n_hvg,
                        n_markers_per_type=n_markers_per_type, spatial_method=spatial_method, k_neighbors=k_neighbors,
                        radius=radius, preprocess=preprocess, random_state=random_state, verbose=False)
    proportions = model.fit_transform(Y, X, coords, cell_type_names=names)
    result_to_anndata(proportions, adata, names, key_added=key_added)
    adata.uns[f"{key_added}_params"] = {
        "sketch_dim": sketch_dim,
        "lambda_spatial": float(model.lambda_used_),
        "rho_sparsity": rho_sparsity,
        "n_hvg": n_hvg,
        "n_markers_per_type": n_markers_per_type,
        "spatial_method": spatial_method,
        "k_neighbors": k_neighbors,
        "radius": radius,
        "preprocess": preprocess,
        "n_genes_used": len(model.gene_idx_),
        "n_cell_types": len(names),
        "cell_type_names": list(names),
        "random_state": random_state,
        "converged": model.info_.get("converged", False),
        "n_iterations": model.info_.get("n_iterations", 0),
    }
    return adata if copy else None
