"""scanpy-style tools namespace (``fd.tl.deconvolve``), as in the reference's ``flashdeconv/tl``."""
from ._deconvolve import deconvolve  # noqa: F401

__all__ = ["deconvolve"]
