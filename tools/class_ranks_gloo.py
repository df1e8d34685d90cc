#!/usr/bin/env python3
"""The ShardedFlashDeconv CLASS (plan by band recompute, fit_transform, Python exchange loop) with W real processes that share ONE
GPU and talk over gloo - the only way to run the estimator's world > 1 code paths end to end without W GPUs.  Rank 0 compares the
assembled proportions with the single-GPU fit.  usage: python tools/class_ranks_gloo.py [W] [n]"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "tests")):
    sys.path.insert(0, p)


def worker(rank, W, n, port, q):
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
        for case, (G, K, d, pre, n_hvg) in {"raw": (300, 9, 64, "raw", 2000), "log_cpm_selected": (900, 6, 64, "log_cpm", 250)}.items():
            if pre == "raw":
                Y, X, coords, _ = datagen.gaussian_raw(n, G, K, seed=2)
            else:
                Y, X, coords, _ = datagen.count_like(n, G, K, 0.1, 8)
            m = ShardedFlashDeconv(sketch_dim=d, preprocess=pre, n_hvg=n_hvg, n_markers_per_type=10, max_iter=25)
            own = m.plan(torch.from_numpy(coords).to(dev), X)
            P = m.fit_transform(torch.from_numpy(Y.astype(np.float32)).to(dev)[own], X)
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
    port = 29600 + os.getpid() % 1000
    procs = [ctx.Process(target=worker, args=(r, W, n, port, q)) for r in range(W)]
    for p in procs:
        p.start()
    got = dict(q.get(timeout=300) for _ in procs)
    for p in procs:
        p.join(timeout=60)
    import datagen
    from flashdeconv_amd import FlashDeconv
    bad = 0
    for case, (G, K, d, pre, n_hvg) in {"raw": (300, 9, 64, "raw", 2000), "log_cpm_selected": (900, 6, 64, "log_cpm", 250)}.items():
        if pre == "raw":
            Y, X, coords, _ = datagen.gaussian_raw(n, G, K, seed=2)
        else:
            Y, X, coords, _ = datagen.count_like(n, G, K, 0.1, 8)
        ref = FlashDeconv(sketch_dim=d, preprocess=pre, n_hvg=n_hvg, n_markers_per_type=10, max_iter=25).fit(Y.astype(np.float32), X, coords)
        P = np.zeros((n, K))
        for r in range(W):
            own, Pr, info, route, lam, gidx, Br = got[r][case]
            P[own] = Pr
        rel = float(np.linalg.norm(P - ref.proportions_) / np.linalg.norm(ref.proportions_))
        info0 = got[0][case][2]
        ok = info0["n_iterations"] == ref.info_["n_iterations"] and rel < 1e-9
        bad += 0 if ok else 1
        print(case, "lambda", got[0][case][4], ref.lambda_used_, "genes equal", bool(np.array_equal(got[0][case][5], ref.gene_idx_)), len(got[0][case][5]), len(ref.gene_idx_))
        print(case, "W", W, "route", got[0][case][3], "iterations", info0["n_iterations"], ref.info_["n_iterations"], "rel", rel, "bits equal",
              bool(np.array_equal(P, ref.proportions_)), "objective", info0["final_objective"], ref.info_["final_objective"], "OK" if ok else "MISMATCH")
    print("done; problems:", bad)


if __name__ == "__main__":
    main()
