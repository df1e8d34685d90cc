"""N > 1 path on CPU: two gloo ranks run the product's ShardedSolver / HaloExchange / TorchComm loop
(flashdeconv_amd/distributed.py) with the oracle's C sweep standing in for the HIP kernel, and must reproduce the
single-process oracle solve: same iteration count, same abundances.  The device-side pieces of the sharded path
(graph localisation, sweep on a local graph) are covered on the GPU box by tests/test_gpu_sharded.py."""
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _localize(A, order, lo, hi):
    """Local CSR of rows order[lo:hi] in a local index space: own first, then halo (ascending sorted position)."""
    n = A.shape[0]
    rank_of = np.empty(n, dtype=np.int64)
    rank_of[order] = np.arange(n)
    rows = order[lo:hi]
    indptr = [0]
    ext = set()
    nbrs = []
    for i in rows:
        js = A.indices[A.indptr[i]:A.indptr[i + 1]]
        pos = rank_of[js]
        nbrs.append((js, pos))
        ext.update(int(p) for p in pos if p < lo or p >= hi)
    halo = np.array(sorted(ext), dtype=np.int64)
    hpos = {int(p): hi - lo + t for t, p in enumerate(halo)}
    indices = []
    for js, pos in nbrs:
        # keep the reference's summation order: ascending ORIGINAL index
        loc = [int(p - lo) if lo <= p < hi else hpos[int(p)] for p in pos]
        indices.extend(loc)
        indptr.append(len(indices))
    return np.asarray(indptr, dtype=np.int64), np.asarray(indices, dtype=np.int64), halo


class OracleBackend:
    """Test stand-in for HipBackend: same interface, oracle C sweep, same on-device stopping-rule semantics."""

    def __init__(self, orc, H_own, XtX, indptr, indices, n_own, n_total):
        self.orc, self.H, self.XtX = orc, np.ascontiguousarray(H_own), np.ascontiguousarray(XtX)
        self.indptr, self.indices, self.n_own, self.n_total = indptr, indices, n_own, n_total

    @staticmethod
    def _rc(row):
        r = row.numpy()
        return r[:64].max() / (r[64:].max() + 1e-10)

    def init_beta(self, beta, n_fill):
        beta.zero_()
        beta[:, :n_fill] = 1.0 / beta.shape[0]

    def sweep(self, it, b_in, b_out, lam, rho_eff, tol, stats, rel):
        if it > 0:
            rc = self._rc(stats[it - 1])
            rel[it - 1] = rc
            if rc < tol:
                return
        K = b_in.shape[0]
        bi = np.ascontiguousarray(b_in[:, :self.n_total].numpy().T)          # (n_total, K)
        bo = np.zeros((self.n_total, K))
        d, a = self.orc.bcd_iteration_c(self.H, self.XtX, bi, bo, self.indices, self.indptr, lam, rho_eff, n_rows=self.n_own)
        import torch
        b_out[:, :self.n_own] = torch.from_numpy(np.ascontiguousarray(bo[:self.n_own].T))
        stats[it, 0] = float(d.max()) if len(d) else 0.0
        stats[it, 64] = float(a.max()) if len(a) else 0.0

    def fold(self, stats, rel, it):
        rel[it] = self._rc(stats[it])


def _worker(rank, world, port, q):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    import torch
    import torch.distributed as dist
    import datagen
    import fdx_oracle as orc
    from flashdeconv_amd.distributed import HaloExchange, ShardedSolver, TorchComm, shard_bounds
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        out = {}
        for case, (n, K, d, max_iter, tol) in {"converges": (1500, 6, 32, 100, 1e-4), "max_iter": (900, 5, 24, 7, 1e-12), "empty_rank": (200, 4, 16, 9, 1e-12)}.items():
            Ys, Xs, coords, _ = datagen.sketched_problem(n, K, d, seed=3)
            A = orc.knn_graph_kdtree(coords * 40, 6).tocsr()
            order = np.lexsort((coords[:, 1], np.floor(coords[:, 0] * 8)))          # any locality-preserving order works
            bounds = shard_bounds(n, world)
            lo, hi = int(bounds[rank]), int(bounds[rank + 1])
            indptr, indices, halo = _localize(A, order, lo, hi)
            n_own, n_total = hi - lo, hi - lo + len(halo)
            # who owns my halo rows / what do the peers need from me (symmetric graph -> computed locally, like the device code)
            recv_counts = [int(((halo >= bounds[r]) & (halo < bounds[r + 1])).sum()) for r in range(world)]
            send_idx, send_counts = [], []
            for r in range(world):
                if r == rank:
                    send_counts.append(0)
                    continue
                _, _, halo_r = _localize(A, order, int(bounds[r]), int(bounds[r + 1]))
                mine = halo_r[(halo_r >= lo) & (halo_r < hi)] - lo
                send_idx.extend(mine.tolist())
                send_counts.append(len(mine))
            XtX = Xs @ Xs.T
            H_own = np.ascontiguousarray((Xs @ Ys[order[lo:hi]].T))                   # (K, n_own)
            rho_eff = 0.01 * np.mean(np.diag(XtX))
            comm = TorchComm()
            halo_x = HaloExchange(comm, n_own, torch.tensor(send_idx, dtype=torch.long), send_counts, recv_counts)
            be = OracleBackend(orc, H_own, XtX, indptr, indices, n_own, n_total)
            ld = ((n_total + 1 + 63) // 64) * 64
            solver = ShardedSolver(be, comm, halo_x, K, ld, n_own, n_total, max_iter=max_iter, tol=tol)
            beta, info = solver.run(lambda shape, dt: torch.zeros(shape, dtype=dt), 0.1, rho_eff)
            out[case] = (order[lo:hi], beta[:, :n_own].numpy().T.copy(), info)
        # the one exchange of the sharded graph build: every rank wrote its own rows of the k-NN list arrays; after
        # all_gather_rows all ranks hold all rows (uneven, 256-aligned shards; padded equal segments on the wire)
        n, kk = 1100, 7
        bounds = shard_bounds(n, world)
        full_nbr = (np.arange(n * kk, dtype=np.int32).reshape(n, kk) * 7919) % n
        full_cnt = (np.arange(n, dtype=np.int32) % kk)
        nbr = torch.full((n, kk), -5, dtype=torch.int32)
        cnt = torch.full((n,), -5, dtype=torch.int32)
        lo, hi = int(bounds[rank]), int(bounds[rank + 1])
        nbr[lo:hi] = torch.from_numpy(full_nbr[lo:hi])
        cnt[lo:hi] = torch.from_numpy(full_cnt[lo:hi])
        TorchComm().all_gather_rows(nbr, cnt, bounds)
        out["gather_ok"] = bool(np.array_equal(nbr.numpy(), full_nbr) and np.array_equal(cnt.numpy(), full_cnt))
        q.put((rank, out))
    finally:
        dist.destroy_process_group()


@pytest.mark.timeout(300)
def test_two_gloo_ranks_reproduce_single_process_solve():
    import torch.multiprocessing as mp
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import datagen
    import fdx_oracle as orc
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    import socket
    with socket.socket() as sk:                       # a free rendezvous port
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    got = dict(q.get(timeout=240) for _ in procs)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert got[0]["gather_ok"] and got[1]["gather_ok"]
    for case, (n, K, d, max_iter, tol) in {"converges": (1500, 6, 32, 100, 1e-4), "max_iter": (900, 5, 24, 7, 1e-12), "empty_rank": (200, 4, 16, 9, 1e-12)}.items():
        Ys, Xs, coords, _ = datagen.sketched_problem(n, K, d, seed=3)
        A = orc.knn_graph_kdtree(coords * 40, 6)
        want, winfo = orc.bcd_solve(Ys, Xs, A, 0.1, 0.01, max_iter=max_iter, tol=tol)
        beta = np.zeros_like(want)
        for r in range(2):
            ids, b, info = got[r][case]
            beta[ids] = b
            assert info["n_iterations"] == winfo["n_iterations"] and info["converged"] == winfo["converged"]
            np.testing.assert_allclose(info["final_change"], winfo["final_change"], rtol=1e-9, atol=1e-15)
        np.testing.assert_allclose(beta, want, rtol=1e-11, atol=1e-14)


def test_shard_bounds_are_tile_aligned_and_cover():
    sys.path.insert(0, ROOT)
    from flashdeconv_amd.distributed import shard_bounds
    for n in (0, 1, 255, 256, 257, 1000, 1_000_000, 999_937):
        for w in (1, 2, 3, 4, 8):
            b = shard_bounds(n, w)
            assert b[0] == 0 and b[-1] == n and np.all(np.diff(b) >= 0) and len(b) == w + 1
            assert np.all(b[:-1] % 256 == 0)
            if n >= 256 * w:
                assert np.diff(b).max() - np.diff(b).min() <= 256 + 255


def test_shard_moment_sums_combine_to_global_statistics():
    """Sharded gene selection: per-shard (mean, ddof-1 var) -> (sum z, sum z^2) -> all-reduce -> global (mean, var)."""
    sys.path.insert(0, ROOT)
    from flashdeconv_amd.distributed import combine_moment_sums, moments_from_sums
    rs = np.random.RandomState(0)
    Z = np.log1p(rs.poisson(1.3, size=(1000, 37)) * 2.5)
    cuts = [0, 1, 400, 401, 1000]                     # shards of 1, 399, 1 and 599 spots
    s1 = np.zeros(37)
    s2 = np.zeros(37)
    for a, b in zip(cuts[:-1], cuts[1:]):
        part = Z[a:b]
        var = part.var(axis=0, ddof=1) if b - a >= 2 else np.zeros(37)
        p1, p2 = combine_moment_sums(part.mean(axis=0), var, b - a)
        s1 += p1
        s2 += p2
    mean, var = moments_from_sums(s1, s2, 1000)
    np.testing.assert_allclose(mean, Z.mean(axis=0), rtol=1e-13)
    np.testing.assert_allclose(var, Z.var(axis=0, ddof=1), rtol=1e-11)

