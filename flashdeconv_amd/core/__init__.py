"""Numerical core: same module layout and exports as the reference's ``flashdeconv/core`` (core/__init__.py:3-21)."""
from .deconv import FlashDeconv  # noqa: F401
from .sketching import build_countsketch_matrix, project_to_sketch  # noqa: F401
from .solver import bcd_solve  # noqa: F401
from .spatial import compute_laplacian, get_neighbor_indices  # noqa: F401

__all__ = ["FlashDeconv", "build_countsketch_matrix", "project_to_sketch", "compute_laplacian", "get_neighbor_indices",
           "bcd_solve"]
