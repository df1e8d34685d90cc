"""-m gpu: BASELINE.json's full single-GPU size (1M spots x 2000 genes x 30 types, d=512) through size-independent
properties - the oracle cannot run at this size.  Synthetic inputs are generated on the device (bench.py generators)."""
import hashlib
import os
import sys

import numpy as np
import pytest
from scipy import sparse

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def big():
    import torch
    sys.path.insert(0, ROOT)
    import bench
    free, _ = torch.cuda.mem_get_info()
    n = 1_000_000 if free > 40 * 2**30 else 200_000
    Y, X, coords = bench.gen_gaussian(torch, n, 2000, 30, torch.device("cuda", 0), seed=7)
    return n, Y, X, coords


def test_full_size_fit_properties(big):
    import torch
    from flashdeconv_amd import FlashDeconv
    n, Y, X, coords = big
    m = FlashDeconv(sketch_dim=512, preprocess="raw", n_hvg=2000)
    P = m.fit_transform(Y, X, coords, output="torch")
    assert P.shape == (n, 30) and m.beta_.shape == (n, 30)
    assert bool(torch.all(m.beta_ >= 0)) and bool(torch.isfinite(P).all())
    assert float((P.sum(dim=1) - 1).abs().max()) < 1e-12                      # rows of proportions sum to one
    assert m.info_["converged"] and 3 <= m.info_["n_iterations"] <= 15           # well-conditioned family: 5-7 sweeps
    assert np.isfinite(m.info_["final_objective"]) and m.lambda_used_ > 0
    h1 = hashlib.sha256(m.beta_.cpu().numpy().tobytes()).hexdigest()
    P2 = FlashDeconv(sketch_dim=512, preprocess="raw", n_hvg=2000).fit_transform(Y, X, coords, output="torch")
    assert torch.equal(P, P2)                                                    # run-to-run bit determinism
    # spot order independence: permuting the spots permutes the result (graph, hash and solve are order-free)
    perm = torch.randperm(n, device=Y.device, generator=torch.Generator(device=Y.device).manual_seed(1))
    P3 = FlashDeconv(sketch_dim=512, preprocess="raw", n_hvg=2000).fit_transform(Y[perm], X, coords[perm], output="torch")
    assert float((P3 - P[perm]).abs().max()) < 1e-9
    # adjacency: symmetric, binary, no self loops, k <= degree
    A = m.adjacency_
    assert A.shape == (n, n) and (A != A.T).nnz == 0 and A.diagonal().sum() == 0 and np.all(A.data == 1.0)
    deg = np.diff(A.indptr)
    assert deg.min() >= 6 and 6.5 < deg.mean() < 7.6
    assert h1 == hashlib.sha256(m.beta_.cpu().numpy().tobytes()).hexdigest()


def test_full_size_sketch_linearity_and_solver_fixed_point(big):
    import torch
    from flashdeconv_amd import FlashDeconv
    n, Y, X, coords = big
    # scaling Y and X by 2 (raw mode) scales the sketches by 2 and leaves the NNLS solution beta unchanged up to the
    # regularisers' scale: with rho = 0 and lambda = 0 the abundances are exactly scale invariant
    kw = dict(sketch_dim=512, preprocess="raw", n_hvg=2000, rho_sparsity=0.0, lambda_spatial=0.0, max_iter=8, tol=1e-30)
    b1 = FlashDeconv(**kw).fit(Y, X, coords, output="torch").beta_
    b2 = FlashDeconv(**kw).fit(Y * 2.0, X * 2.0, coords, output="torch").beta_
    assert float((b1 - b2).abs().max()) < 1e-9 * float(b1.abs().max())
