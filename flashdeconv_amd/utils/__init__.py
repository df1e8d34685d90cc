"""Utilities of the path: the reference's ``flashdeconv/utils`` re-exports (utils/__init__.py:3-31) minus the evaluation
metrics (``compute_rmse`` / ``compute_correlation``: not on the fit path)."""
from .genes import select_hvg, select_markers, compute_leverage_scores  # noqa: F401
from .graph import build_knn_graph, build_radius_graph, coords_to_adjacency  # noqa: F401
from .random import check_random_state  # noqa: F401

__all__ = ["select_hvg", "select_markers", "compute_leverage_scores", "build_knn_graph", "build_radius_graph",
           "coords_to_adjacency", "check_random_state"]
