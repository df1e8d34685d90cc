"""Random-state handling with the reference's semantics (flashdeconv/utils/random.py:16-71).

The CountSketch hash/sign stream is defined as draws from numpy's legacy ``RandomState`` (MT19937), so this
module returns exactly such an object: None -> numpy's global RandomState, int -> a fresh RandomState(seed),
RandomState -> itself.  The stream is pinned bit-for-bit by tests/test_oracle.py and tests/test_host.py.
"""
import numpy as np


def check_random_state(seed):
    if seed is None or seed is np.random:
        return np.random.mtrand._rand
    if isinstance(seed, (int, np.integer)):
        return np.random.RandomState(seed)
    if isinstance(seed, np.random.RandomState):
        return seed
    raise ValueError(
        f"'{seed}' cannot be used to seed a numpy.random.RandomState instance. "
        f"Expected None, int, or np.random.RandomState, got {type(seed)}.")
