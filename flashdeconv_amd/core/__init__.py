"""Numerical core: same module layout as the reference's ``flashdeconv/core``."""
from .solver import bcd_solve, normalize_proportions  # noqa: F401
