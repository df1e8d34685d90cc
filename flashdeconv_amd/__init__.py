"""flashdeconv_amd -- MI355X (gfx950) implementation of FlashDeconv's sketched graph-regularised NNLS path.

Same Python surface as the reference for that path (``FlashDeconv``, ``tl.deconvolve`` and the inner seams
``bcd_solve`` / ``sketch_data`` / ``coords_to_adjacency`` ...); the arithmetic runs in hand-written HIP kernels
behind the C ABI of ``libfdx.so`` (include/fdx.h).  No CPU fallback: without the built library and a GPU the
compute entry points raise.
"""
__version__ = "0.1.0"

from . import _lib  # noqa: F401
from .core.deconv import FlashDeconv  # noqa: F401
from . import tl  # noqa: F401

__all__ = ["FlashDeconv", "tl", "__version__"]
