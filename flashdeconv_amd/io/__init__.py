"""AnnData marshalling around the accelerated path (thin; the reference's ``flashdeconv/io`` surface)."""
from .loader import align_genes, load_reference, load_spatial_data, prepare_data, result_to_anndata  # noqa: F401
