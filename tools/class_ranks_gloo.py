#!/usr/bin/env python3
"""The ShardedFlashDeconv CLASS (plan by band recompute, fit_transform, Python exchange loop) with W real processes that share ONE
GPU and talk over gloo - the only way to run the estimator's world > 1 code paths end to end without W GPUs.  Rank 0 compares the
assembled proportions with the single-GPU fit.  usage: python tools/class_ranks_gloo.py [W] [n]"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "tests")):
    sys.path.insert(0, p)


def cases(n):
    """name -> dict(G, K, d, pre, n_hvg, family, + options); every rank and the checker build the same problem from it"""
    import numpy as np
    c = {"raw": dict(G=300, K=9, d=64, pre="raw", n_hvg=2000, family="gaussian"),
         "log_cpm_selected": dict(G=900, K=6, d=64, pre="log_cpm", n_hvg=250, family="counts"),
         "pearson": dict(G=300, K=5, d=48, pre="pearson", n_hvg=2000, family="counts"),
         "csr_log_cpm_selected": dict(G=900, K=6, d=64, pre="log_cpm", n_hvg=250, family="counts", csr=True, tol=1e-9),
         "radius": dict(G=300, K=9, d=64, pre="raw", n_hvg=2000, family="gaussian", method="radius"),
         "grid": dict(G=300, K=9, d=64, pre="raw", n_hvg=2000, family="gaussian", method="grid"),
         "seventy_types": dict(G=400, K=70, d=128, pre="raw", n_hvg=2000, family="gaussian", tol=1e-9),
         "hundred_types": dict(G=400, K=100, d=128, pre="raw", n_hvg=2000, family="gaussian"),
         "int_counts": dict(G=300, K=6, d=64, pre="log_cpm", n_hvg=2000, family="counts", dtype=np.int32),
         "clusters_far_walks": dict(G=300, K=9, d=64, pre="raw", n_hvg=2000, family="gaussian", clusters=True),
         "lattice_ties": dict(G=300, K=9, d=64, pre="raw", n_hvg=2000, family="gaussian", lattice=True)}
    only = os.environ.get("CASES")
    return {k: v for k, v in c.items() if not only or k in only.split(",")}


def make(case, kw, n):
    import numpy as np
    import datagen
    if kw["family"] == "gaussian":
        Y, X, coords, _ = datagen.gaussian_raw(n, kw["G"], kw["K"], seed=2)
    else:
        Y, X, coords, _ = datagen.count_like(n, kw["G"], kw["K"], 0.1, 8)
    if kw.get("clusters"):
        rs = np.random.RandomState(3)
        c = rs.rand(10, 2) * 100
        coords = c[rs.randint(10, size=n)] + rs.randn(n, 2) * rs.choice([0.3, 2.0, 8.0], size=(n, 1))
    if kw.get("lattice"):                  # square lattice, k = 6: every spot ties at its k-th neighbour (two of four at sqrt 2)
        coords = np.stack([np.arange(n) % 50, np.arange(n) // 50], axis=1).astype(np.float64)
    est = dict(sketch_dim=kw["d"], preprocess=kw["pre"], n_hvg=kw["n_hvg"], n_markers_per_type=10, max_iter=15)
    if kw.get("method") == "radius":
        est.update(spatial_method="radius", radius=float(np.sqrt(coords.var(axis=0).sum()) * 0.03))
    elif kw.get("method") == "grid":
        est.update(spatial_method="grid")
    return Y, X, coords, est


def worker(rank, W, n, port, q):
    import faulthandler
    faulthandler.dump_traceback_later(float(os.environ.get("FDX_WORKER_DEADLINE", "240")), exit=True)   # a hung rank ends with its stack, not silently
    import numpy as np
    import torch
    import torch.distributed as dist
    import datagen
    from flashdeconv_amd.distributed import ShardedFlashDeconv
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    dev = torch.device("cuda", 0)
    torch.cuda.set_device(dev)
    dist.init_process_group("gloo", rank=rank, world_size=W)
    out = {}
    try:
        for case, kw in cases(n).items():
            Y, X, coords, est = make(case, kw, n)
            m = ShardedFlashDeconv(**est)
            own = m.plan(torch.from_numpy(coords).to(dev), X)
            Yo = Y[own.cpu().numpy()]
            if kw.get("csr"):
                import scipy.sparse as sp
                S = sp.csr_matrix(Yo.astype(np.float32))
                Yt = torch.sparse_csr_tensor(torch.from_numpy(S.indptr.astype(np.int64)), torch.from_numpy(S.indices.astype(np.int64)),
                                             torch.from_numpy(S.data), size=S.shape).to(dev)
            else:
                Yt = torch.from_numpy(Yo.astype(kw.get("dtype", np.float32))).to(dev)
            P = m.fit_transform(Yt, X)
            out[case] = (own.cpu().numpy(), P.cpu().numpy(), m.info_, m.plan_route_, float(m.lambda_used_), np.asarray(m.gene_idx_), m.beta_.cpu().numpy())
        q.put((rank, out))
    finally:
        dist.destroy_process_group()


def main():
    import numpy as np
    import torch.multiprocessing as mp
    W = int(sys.argv[1]) if len(sys.argv) > 1 else 3
    n = int(sys.argv[2]) if len(sys.argv) > 2 else 5000
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    import socket
    with socket.socket() as sk:                       # a free rendezvous port (a fixed one can be taken: the ranks then wait for ever)
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    procs = [ctx.Process(target=worker, args=(r, W, n, port, q)) for r in range(W)]
    for p in procs:
        p.start()
    got = dict(q.get(timeout=300) for _ in procs)
    for p in procs:
        p.join(timeout=60)
    import datagen
    from flashdeconv_amd import FlashDeconv
    bad = 0
    for case, kw in cases(n).items():
        Y, X, coords, est = make(case, kw, n)
        K = X.shape[0]
        if kw.get("csr"):
            import scipy.sparse as sp
            Yin = sp.csr_matrix(Y.astype(np.float32))
        else:
            Yin = Y.astype(kw.get("dtype", np.float32))
        ref = FlashDeconv(**est).fit(Yin, X, coords)
        P = np.zeros((n, K))
        for r in range(W):
            own, Pr, info, route, lam, gidx, Br = got[r][case]
            P[own] = Pr
        rel = float(np.linalg.norm(P - ref.proportions_) / np.linalg.norm(ref.proportions_))
        info0 = got[0][case][2]
        ok = info0["n_iterations"] == ref.info_["n_iterations"] and rel < kw.get("tol", 1e-12)
        bad += 0 if ok else 1
        print(case, "lambda", got[0][case][4], ref.lambda_used_, "genes equal", bool(np.array_equal(got[0][case][5], ref.gene_idx_)), len(got[0][case][5]), len(ref.gene_idx_))
        print(case, "W", W, "route", got[0][case][3], "iterations", info0["n_iterations"], ref.info_["n_iterations"], "rel", rel, "bits equal",
              bool(np.array_equal(P, ref.proportions_)), "objective", info0["final_objective"], ref.info_["final_objective"], "OK" if ok else "MISMATCH")
    print("done; problems:", bad)


if __name__ == "__main__":
    main()
