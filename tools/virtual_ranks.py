#!/usr/bin/env python3
"""BASELINE configs[4] (10M spots x 5000 genes x 50 types, sketch_dim 1024, lambda auto, 8 ranks) driven by W "virtual
ranks" on ONE GPU: the sharded plan exactly as W processes would run it (every rank bins all coordinates and finds the
k-NN lists of its own rows, the list rows are all-gathered, every rank symmetrises and localises its own rows), one
shard of Y at a time (generated, sketched into H by fdx_prepare_dev, freed - only H survives, 0.5 GB per rank), and the
native iteration loop (fdx_sharded_solve_dev) with W host threads as ranks (device copies stand in for ncclSend/ncclRecv).

Used by tests/test_gpu_fullsize.py (properties, determinism, 8-rank bits == 4-rank bits) and by
`bench.py --config 5 --virtual-ranks W` (per-rank stage times: the fixed costs of the plan at 10M spots).
"""
import ctypes
import os
import sys
import threading
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

CHUNK = 1 << 16


def _st(torch):
    return ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)


def gaussian_rows(torch, X32, r0, r1, seed):
    """Rows [r0, r1) of the family-A matrix (Y = B X + 0.1 N(0, 1), B row-normalised U(0, 1)) of a job
    whose rows are drawn in fixed chunks of CHUNK rows, each from its own generator seeded by (seed, chunk): however the
    rows are cut into shards, a spot gets the same values.  Elementwise arithmetic only (a library GEMM may reduce in a
    run-dependent order): bit-reproducible."""
    K, G = X32.shape
    dev = X32.device
    out = torch.empty((r1 - r0, G), dtype=torch.float32, device=dev)
    for c in range(r0 // CHUNK, -(-r1 // CHUNK)):
        g = torch.Generator(device=dev)
        g.manual_seed(seed * 1_000_003 + c)
        B = torch.rand(CHUNK, K, generator=g, device=dev, dtype=torch.float32)
        B /= B.sum(dim=1, keepdim=True)
        Yc = torch.randn(CHUNK, G, generator=g, device=dev, dtype=torch.float32)
        Yc.mul_(0.1)
        for k in range(K):
            Yc.addcmul_(B[:, k:k + 1], X32[k][None, :])
        a = c * CHUNK
        lo, hi = max(r0, a), min(r1, a + CHUNK)
        out[lo - r0:hi - r0] = Yc[lo - a:hi - a]
        del Yc, B
    return out


def morton_sorted_coords(torch, coords, k=6):
    """The coordinates reordered into the solver's (Morton) order, so that a rank's own rows are a contiguous range of the
    caller's order too - what a real multi-GPU job gets by handing every rank the rows of its own region."""
    from flashdeconv_amd import _lib
    lib = _lib.load()
    n = coords.shape[0]
    kk = min(k, n - 1) + 1
    nbr = torch.empty((n, kk), dtype=torch.int32, device=coords.device)
    cnt = torch.empty((n,), dtype=torch.int32, device=coords.device)
    pl, h = ctypes.c_void_p(), ctypes.c_void_p()
    hi = min(n, 256)
    _lib.check(lib.fdx_graph_knn_lists_dev(ctypes.c_void_p(coords.data_ptr()), n, coords.shape[1], k, 0, hi,
                                           ctypes.c_void_p(nbr.data_ptr()), ctypes.c_void_p(cnt.data_ptr()), _st(torch), ctypes.byref(pl)))
    nbr[hi:] = -1
    cnt[hi:] = 0
    _lib.check(lib.fdx_graph_from_knn_lists_dev(pl, ctypes.c_void_p(nbr.data_ptr()), ctypes.c_void_p(cnt.data_ptr()), 0, hi,
                                                _st(torch), ctypes.byref(h)))
    g = _lib.Graph(h.value)
    perm = torch.empty(n, dtype=torch.int32, device=coords.device)
    _lib.check(lib.fdx_graph_perm_dev(g.handle, ctypes.c_void_p(perm.data_ptr()), _st(torch)))
    torch.cuda.synchronize()
    g.close()
    return coords[perm.long()].contiguous()


def virtual_plan(torch, coords, W, k=6, times=None, route=None):
    """The sharded k-NN plan of flashdeconv_amd/distributed.py with W ranks in one process.  Returns the W local graphs
    (with their bookkeeping) and the global structural nnz.  route "band" (default; FDX_PLAN_ALLGATHER=1 selects "allgather"):
    every rank finds the lists of its own rows and of its band and symmetrises - nothing is exchanged; "allgather": the lists
    of the own rows, all-gathered (here W^2 device copies)."""
    route = route or ("allgather" if os.environ.get("FDX_PLAN_ALLGATHER") else "band" if os.environ.get("FDX_PLAN_STEPWISE") else "pipeline")
    if route == "pipeline":
        return virtual_plan_pipeline(torch, coords, W, k, times)
    from flashdeconv_amd import _lib
    from flashdeconv_amd.distributed import shard_bounds
    lib = _lib.load()
    dev = coords.device
    n, dim = coords.shape
    bounds = shard_bounds(n, W)
    kk = min(k, n - 1) + 1
    times = times if times is not None else {}
    t = lambda: (torch.cuda.synchronize(), time.perf_counter())[1]
    nbrs = [torch.empty((n, kk), dtype=torch.int32, device=dev) for _ in range(W)]
    cnts = [torch.empty((n,), dtype=torch.int32, device=dev) for _ in range(W)]
    plans = []
    times["knn_lists_ms"] = []
    for r in range(W):                          # every rank: bin ALL points, lists of its own rows
        t0 = t()
        pl = ctypes.c_void_p()
        lists = lib.fdx_graph_knn_lists_band_dev if route == "band" else lib.fdx_graph_knn_lists_dev
        _lib.check(lists(ctypes.c_void_p(coords.data_ptr()), n, dim, k, int(bounds[r]), int(bounds[r + 1]),
                         ctypes.c_void_p(nbrs[r].data_ptr()), ctypes.c_void_p(cnts[r].data_ptr()), _st(torch), ctypes.byref(pl)))
        plans.append(pl)
        times["knn_lists_ms"].append((t() - t0) * 1e3)
    times["plan_route"] = route
    if route != "band":
        t0 = t()
        for r in range(W):                      # the one exchange step of the build: all-gather of the list rows
            for q in range(W):
                if q != r:
                    a, b = int(bounds[q]), int(bounds[q + 1])
                    nbrs[r][a:b] = nbrs[q][a:b]
                    cnts[r][a:b] = cnts[q][a:b]
        times["allgather_ms"] = (t() - t0) * 1e3
        times["allgather_bytes_per_rank"] = int(n * (kk + 1) * 4)
    ranks, nnz, ties, far = [], 0, 0, 0
    times["from_lists_ms"], times["localize_ms"] = [], []
    for r in range(W):
        t0 = t()
        h = ctypes.c_void_p()
        _lib.check(lib.fdx_graph_from_knn_lists_dev(plans[r], ctypes.c_void_p(nbrs[r].data_ptr()), ctypes.c_void_p(cnts[r].data_ptr()),
                                                    int(bounds[r]), int(bounds[r + 1]), _st(torch), ctypes.byref(h)))
        full = _lib.Graph(h.value)
        nnz += full.info()[1]
        ties += full.knn_ties()
        far += full.knn_far()
        t1 = t()
        nbrs[r] = cnts[r] = None
        hl = ctypes.c_void_p()
        _lib.check(lib.fdx_graph_localize(full.handle, W, _lib.ptr_i64(bounds), r, _st(torch), ctypes.byref(hl)))
        g = _lib.Graph(hl.value)
        n_own = int(bounds[r + 1] - bounds[r])
        perm = torch.empty(max(n_own, 1), dtype=torch.int32, device=dev)
        _lib.check(lib.fdx_graph_perm_dev(g.handle, ctypes.c_void_p(perm.data_ptr()), _st(torch)))
        nh = ctypes.c_int64(0)
        sc, rc = np.zeros(W, dtype=np.int32), np.zeros(W, dtype=np.int32)
        _lib.check(lib.fdx_graph_halo_info(g.handle, ctypes.byref(nh), _lib.ptr_i32(sc), _lib.ptr_i32(rc)))
        t2 = t()
        full.close()
        times["from_lists_ms"].append((t1 - t0) * 1e3)
        times["localize_ms"].append((t2 - t1) * 1e3)
        ranks.append(dict(g=g, own=perm[:n_own].long(), n_own=n_own, n_halo=int(nh.value), lo=int(bounds[r]), hi=int(bounds[r + 1]),
                          send=sc, recv=rc))
    if route == "band" and far:                 # a walk left its block / a band overflowed: the drivers rebuild by exchange
        for R in ranks:
            R["g"].close()
        return virtual_plan(torch, coords, W, k, times, route="allgather")
    for r in range(W):                          # what r sends to q is what q expects from r
        for q in range(W):
            assert ranks[r]["send"][q] == ranks[q]["recv"][r]
    return ranks, int(nnz), int(ties), bounds


def shard_graph(torch, coords, W, r, bounds, k=6):
    """Rank r's local graph by the queued pipeline (fdx_graph_shard_knn_dev), as ShardedFlashDeconv.plan queues it; returns
    (graph, own ids tensor) WITHOUT waiting for the counts."""
    from flashdeconv_amd import _lib
    lib = _lib.load()
    n, dim = coords.shape
    hl = ctypes.c_void_p()
    _lib.check(lib.fdx_graph_shard_knn_dev(ctypes.c_void_p(coords.data_ptr()), n, dim, k, W, _lib.ptr_i64(bounds), r, _st(torch),
                                           ctypes.byref(hl)))
    g = _lib.Graph(hl.value)
    n_own = int(bounds[r + 1] - bounds[r])
    perm = torch.empty(max(n_own, 1), dtype=torch.int32, device=coords.device)
    _lib.check(lib.fdx_graph_perm_dev(g.handle, ctypes.c_void_p(perm.data_ptr()), _st(torch)))
    return g, perm[:n_own].long()


def shard_status(g):
    from flashdeconv_amd import _lib
    lib = _lib.load()
    nnz, ties, far, over = ctypes.c_int64(0), ctypes.c_int64(0), ctypes.c_int32(0), ctypes.c_int32(0)
    _lib.check(lib.fdx_graph_shard_status(g.handle, ctypes.byref(nnz), ctypes.byref(ties), ctypes.byref(far), ctypes.byref(over)))
    return int(nnz.value), int(ties.value), int(far.value), int(over.value)


def virtual_plan_pipeline(torch, coords, W, k=6, times=None):
    """The sharded k-NN plan as ShardedFlashDeconv.plan runs it by default: one queued pipeline per rank, the counts read once."""
    from flashdeconv_amd import _lib
    from flashdeconv_amd.distributed import shard_bounds
    lib = _lib.load()
    n, dim = coords.shape
    bounds = shard_bounds(n, W)
    times = times if times is not None else {}
    t = lambda: (torch.cuda.synchronize(), time.perf_counter())[1]
    times["plan_route"] = "pipeline"
    times["plan_ms"] = []
    ranks, nnz, ties, far, over = [], 0, 0, 0, 0
    for r in range(W):
        if bounds[r + 1] <= bounds[r]:            # a rank without rows: the stepwise path (nothing to queue)
            return virtual_plan(torch, coords, W, k, times, route="band")
        t0 = t()
        g, own = shard_graph(torch, coords, W, r, bounds, k)
        a, b, c, d = shard_status(g)
        nh = ctypes.c_int64(0)
        sc, rc = np.zeros(W, dtype=np.int32), np.zeros(W, dtype=np.int32)
        _lib.check(lib.fdx_graph_halo_info(g.handle, ctypes.byref(nh), _lib.ptr_i32(sc), _lib.ptr_i32(rc)))
        times["plan_ms"].append((t() - t0) * 1e3)
        nnz, ties, far, over = nnz + a, ties + b, far + c, over + d
        n_own = int(bounds[r + 1] - bounds[r])
        ranks.append(dict(g=g, own=own, n_own=n_own, n_halo=int(nh.value), lo=int(bounds[r]), hi=int(bounds[r + 1]), send=sc, recv=rc))
    if far or over:                               # the drivers rebuild: far -> by exchange, a bound too small -> stepwise
        for R in ranks:
            R["g"].close()
        return virtual_plan(torch, coords, W, k, times, route="allgather" if far else "band")
    for r in range(W):
        for q in range(W):
            assert ranks[r]["send"][q] == ranks[q]["recv"][r]
    return ranks, int(nnz), int(ties), bounds


def virtual_prepare(torch, ranks, X, make_rows, d, mode, random_state=0, times=None):
    """Per rank: its rows of Y (make_rows(lo, hi) -> (n_own, G) float32 device tensor in the caller's order, which must be
    the solver's order: morton_sorted_coords), H and XtX by fdx_prepare_dev, Y freed."""
    from flashdeconv_amd import _lib
    from flashdeconv_amd.core.sketching import countsketch_tables
    from flashdeconv_amd.utils.genes import compute_leverage_scores
    lib = _lib.load()
    K, G = X.shape
    times = times if times is not None else {}
    t = lambda: (torch.cuda.synchronize(), time.perf_counter())[1]
    t0 = t()
    lev = compute_leverage_scores(X)
    times["leverage_ms"] = (t() - t0) * 1e3
    bucket, weight = countsketch_tables(G, d, lev, random_state)
    b32 = np.ascontiguousarray(bucket, dtype=np.int32)
    Xc = np.ascontiguousarray(X, dtype=np.float64)
    times["generate_ms"], times["prepare_ms"] = [], []
    yty = 0.0
    for R in ranks:
        dev = R["own"].device
        assert bool(torch.equal(R["own"], torch.arange(R["lo"], R["hi"], device=dev))), "coordinates are not in solver order"
        n_total = R["n_own"] + R["n_halo"]
        ld = ((n_total + 1 + 63) // 64) * 64
        H = torch.empty((K, ld), dtype=torch.float64, device=dev)          # the allocation outside the timed call (a job that fits twice has it pooled), the fill inside
        XtX = torch.empty((K, K), dtype=torch.float64, device=dev)
        XtX_h = np.empty((K, K))
        part = ctypes.c_double(0.0)
        t0 = t()
        Y = make_rows(R["lo"], R["hi"])
        t1 = t()
        H.zero_()
        _lib.check(lib.fdx_prepare_dev(ctypes.c_void_p(Y.data_ptr()), _lib.FDX_F32, R["n_own"], G, G, None, _lib.ptr_f64(Xc), K,
                                       _lib.ptr_i32(b32), _lib.ptr_f64(weight), _lib.ptr_f64(weight), d, mode, mode,
                                       ctypes.c_void_p(H.data_ptr()), ld, ctypes.c_void_p(XtX.data_ptr()), _lib.ptr_f64(XtX_h),
                                       ctypes.byref(part), _st(torch)))
        t2 = t()
        del Y
        yty += part.value
        R.update(ld=ld, H=H, XtX=XtX, XtX_h=XtX_h,
                 beta=[torch.empty((K, ld), dtype=torch.float64, device=dev) for _ in range(2)])
        times["generate_ms"].append((t1 - t0) * 1e3)
        times["prepare_ms"].append((t2 - t1) * 1e3)
    torch.cuda.empty_cache()
    return yty


def virtual_solve(torch, ranks, K, lam, rho_eff, tol, max_iter, times=None):
    """fdx_sharded_solve_dev on every shard at once: one host thread and one stream per rank."""
    from flashdeconv_amd import _lib
    lib = _lib.load()
    W = len(ranks)
    world = ctypes.c_void_p()
    _lib.check(lib.fdx_local_world_create(W, ctypes.byref(world)))
    torch.cuda.synchronize()
    results, errors = [None] * W, []
    dev_index = ranks[0]["H"].device.index or 0

    def work(r):
        try:
            torch.cuda.set_device(dev_index)
            S = ranks[r]
            comm = ctypes.c_void_p()
            _lib.check(lib.fdx_comm_init_local(world, r, ctypes.byref(comm)))
            stream = torch.cuda.Stream()
            info = _lib.SolveInfo()
            which = ctypes.c_int32(0)
            rel = np.zeros(max(max_iter, 1))
            rc = lib.fdx_sharded_solve_dev(comm, S["g"].handle, ctypes.c_void_p(S["H"].data_ptr()), S["ld"],
                                           ctypes.c_void_p(S["XtX"].data_ptr()), K, lam, rho_eff, tol, max_iter,
                                           ctypes.c_void_p(S["beta"][0].data_ptr()), ctypes.c_void_p(S["beta"][1].data_ptr()), S["ld"],
                                           ctypes.byref(info), _lib.ptr_f64(rel), ctypes.byref(which),
                                           ctypes.c_void_p(stream.cuda_stream))
            if rc != 0:
                errors.append((r, lib.fdx_last_error()))
            results[r] = dict(n_iterations=int(info.n_iterations), converged=bool(info.converged),
                              final_change=float(info.final_change), buffer=which.value, sweep_ms=float(info.sweep_ms), rel=rel)
            lib.fdx_comm_destroy(comm)
        except Exception as e:                                   # noqa: BLE001 - reported below, the other ranks would hang
            errors.append((r, repr(e)))

    t0 = time.perf_counter()
    threads = [threading.Thread(target=work, args=(r,)) for r in range(W)]
    for th in threads:
        th.start()
    for th in threads:
        th.join(timeout=900)
    alive = any(th.is_alive() for th in threads)
    torch.cuda.synchronize()
    if times is not None:
        times["solve_wall_ms"] = (time.perf_counter() - t0) * 1e3
        times["sweep_ms"] = [r["sweep_ms"] if r else None for r in results]
    assert not alive, "a rank thread hangs"
    lib.fdx_local_world_destroy(world)
    assert not errors, errors
    return results


def alone_solve_and_finish(torch, ranks, K, lam, rho_eff, n_iter, times):
    """Every rank's iteration loop and finish ALONE on the GPU (fdx_comm_init_loopback: the exchange is a copy of the rank's own
    staging, no all-reduce), n_iter iterations as in the real solve, on scratch abundance buffers: the kernel / launch / read-back
    time of a rank's critical path - no wire time, no waiting for peers.  Then the finish: objective partials + export."""
    from flashdeconv_amd import _lib
    lib = _lib.load()
    W = len(ranks)
    t = lambda: (torch.cuda.synchronize(), time.perf_counter())[1]
    times["solve_alone_ms"], times["finish_alone_ms"] = [], []
    for r, S in enumerate(ranks):
        dev = S["H"].device
        bufs = [torch.empty((K, S["ld"]), dtype=torch.float64, device=dev) for _ in range(2)]
        comm = ctypes.c_void_p()
        _lib.check(lib.fdx_comm_init_loopback(r, W, ctypes.byref(comm)))
        info, which, rel = _lib.SolveInfo(), ctypes.c_int32(0), np.zeros(max(n_iter, 1))
        best = None
        for rep in range(2):                                                  # second run: warm (tile lists built, pool filled)
            t0 = t()
            _lib.check(lib.fdx_sharded_solve_dev(comm, S["g"].handle, ctypes.c_void_p(S["H"].data_ptr()), S["ld"],
                                                 ctypes.c_void_p(S["XtX"].data_ptr()), K, lam, rho_eff, 0.0, n_iter,
                                                 ctypes.c_void_p(bufs[0].data_ptr()), ctypes.c_void_p(bufs[1].data_ptr()), S["ld"],
                                                 ctypes.byref(info), _lib.ptr_f64(rel), ctypes.byref(which), _st(torch)))
            best = (t() - t0) * 1e3
        lib.fdx_comm_destroy(comm)
        times["solve_alone_ms"].append(best)
        b = torch.empty((S["n_own"], K), dtype=torch.float64, device=dev)
        p = torch.empty((S["n_own"], K), dtype=torch.float64, device=dev)
        part = np.zeros(4)
        side = torch.cuda.Stream(device=dev, priority=-1)
        for rep in range(2):
            # as ShardedFlashDeconv._solve_shard does it: the export on a side stream BESIDE the objective pass (both only read the
            # final abundances)
            t0 = t()
            cur = torch.cuda.current_stream()
            side.wait_stream(cur)
            _lib.check(lib.fdx_normalize_dev(ctypes.c_void_p(S["beta"][0].data_ptr()), S["ld"], S["n_own"], K, ctypes.c_void_p(b.data_ptr()),
                                             ctypes.c_void_p(p.data_ptr()), ctypes.c_void_p(side.cuda_stream)))
            _lib.check(lib.fdx_objective_partials_dev(S["g"].handle, ctypes.c_void_p(S["beta"][0].data_ptr()), S["ld"],
                                                      ctypes.c_void_p(S["H"].data_ptr()), S["ld"], ctypes.c_void_p(S["XtX"].data_ptr()), K,
                                                      _lib.ptr_f64(part), _st(torch)))
            cur.wait_stream(side)
            best = (t() - t0) * 1e3
        times["finish_alone_ms"].append(best)
        del bufs, b, p


def alone_pipelined(torch, coords, ranks, bounds, X, make_rows, d, mode, K, lam, rho_eff, n_iter, times, k=6, random_state=0,
                    rank_ids=None, world=None):
    """Every rank's WHOLE share of the sharded fit, ALONE and as the driver queues it - plan (one queued pipeline), the sketch
    behind it without a host round trip in between, the counts, the iteration loop (loopback transport: no peers, no wire time),
    export beside the objective pass - timed as ONE interval, host clock, device idle before and after: the rank's critical path
    as it would run, not the sum of separately synchronised stages."""
    from flashdeconv_amd import _lib
    from flashdeconv_amd.core.sketching import countsketch_tables
    from flashdeconv_amd.utils.genes import compute_leverage_scores
    lib = _lib.load()
    W = world or len(ranks)
    rank_ids = rank_ids if rank_ids is not None else list(range(len(ranks)))
    trace = bool(os.environ.get("FDX_RANK_TRACE"))
    Kx, G = X.shape
    lev = compute_leverage_scores(X)
    bucket, weight = countsketch_tables(G, d, lev, random_state)
    b32 = np.ascontiguousarray(bucket, dtype=np.int32)
    Xc = np.ascontiguousarray(X, dtype=np.float64)
    t = lambda: (torch.cuda.synchronize(), time.perf_counter())[1]
    times["pipelined_ms"] = []
    torch.cuda.empty_cache()
    for r, S in zip(rank_ids, ranks):
        dev = S["H"].device
        Y = make_rows(S["lo"], S["hi"])
        n_own = S["n_own"]
        ldh = ((n_own + 1 + 63) // 64) * 64
        H = torch.empty((Kx, ldh), dtype=torch.float64, device=dev)
        XtX = torch.empty((Kx, Kx), dtype=torch.float64, device=dev)
        XtX_h = np.empty((Kx, Kx))
        bufs = [torch.empty((Kx, S["ld"]), dtype=torch.float64, device=dev) for _ in range(2)]     # ld from the real plan: same halo
        b = torch.empty((n_own, Kx), dtype=torch.float64, device=dev)
        p = torch.empty((n_own, Kx), dtype=torch.float64, device=dev)
        side = torch.cuda.Stream(device=dev, priority=-1)
        comm = ctypes.c_void_p()
        _lib.check(lib.fdx_comm_init_loopback(r, W, ctypes.byref(comm)))
        best = None
        for rep in range(int(os.environ.get("FDX_RANK_REPS", "6"))):     # the fastest of a few: a rank's share is ~1 ms, one stray host delay is 20 % of it
            part, yty = np.zeros(4), ctypes.c_double(0.0)
            info, which, rel = _lib.SolveInfo(), ctypes.c_int32(0), np.zeros(max(n_iter, 1))
            t0 = t()
            g, own = shard_graph(torch, coords, W, r, bounds, k)                                  # queued
            h1 = time.perf_counter()
            H.zero_()
            _lib.check(lib.fdx_prepare_dev(ctypes.c_void_p(Y.data_ptr()), _lib.FDX_F32, n_own, G, G, None, _lib.ptr_f64(Xc), Kx,
                                           _lib.ptr_i32(b32), _lib.ptr_f64(weight), _lib.ptr_f64(weight), d, mode, mode,
                                           ctypes.c_void_p(H.data_ptr()), ldh, ctypes.c_void_p(XtX.data_ptr()), _lib.ptr_f64(XtX_h),
                                           ctypes.byref(yty), _st(torch)))
            h2 = time.perf_counter()
            st_ = shard_status(g)                                                                # long there
            nh = ctypes.c_int64(0)
            _lib.check(lib.fdx_graph_halo_info(g.handle, ctypes.byref(nh), None, None))
            assert int(nh.value) == S["n_halo"] and not st_[2] and not st_[3], (nh.value, S["n_halo"], st_)
            _lib.check(lib.fdx_sharded_solve_dev(comm, g.handle, ctypes.c_void_p(H.data_ptr()), ldh, ctypes.c_void_p(XtX.data_ptr()), Kx,
                                                 lam, rho_eff, 0.0, n_iter, ctypes.c_void_p(bufs[0].data_ptr()),
                                                 ctypes.c_void_p(bufs[1].data_ptr()), S["ld"], ctypes.byref(info), _lib.ptr_f64(rel),
                                                 ctypes.byref(which), _st(torch)))
            h3 = time.perf_counter()
            cur = torch.cuda.current_stream()
            side.wait_stream(cur)
            _lib.check(lib.fdx_normalize_dev(ctypes.c_void_p(bufs[which.value].data_ptr()), S["ld"], n_own, Kx, ctypes.c_void_p(b.data_ptr()),
                                             ctypes.c_void_p(p.data_ptr()), ctypes.c_void_p(side.cuda_stream)))
            _lib.check(lib.fdx_objective_partials_dev(g.handle, ctypes.c_void_p(bufs[which.value].data_ptr()), S["ld"],
                                                      ctypes.c_void_p(H.data_ptr()), ldh, ctypes.c_void_p(XtX.data_ptr()), Kx,
                                                      _lib.ptr_f64(part), _st(torch)))
            cur.wait_stream(side)
            dt = (t() - t0) * 1e3
            if trace:
                print(f"[rank {r}] host: plan call {1e3 * (h1 - t0):.3f}, prepare (to its sync) {1e3 * (h2 - h1):.3f}, counts + loop "
                      f"{1e3 * (h3 - h2):.3f}, finish {dt - 1e3 * (h3 - t0):.3f}, total {dt:.3f} ms", file=sys.stderr)
            best = dt if best is None else min(best, dt)
            g.close()
        lib.fdx_comm_destroy(comm)
        times["pipelined_ms"].append(best)
        del Y, H, bufs, b, p
        torch.cuda.empty_cache()          # a 10M-spot job's shard is 25-100 GB: hand it back before the next rank's is made


def driver_alone(torch, coords, ranks, X, make_rows, d, K, nnz_total, n_iter, times, reps=None, knn_ties="auto", pre="raw"):
    """Every rank's share through the REAL driver - ShardedFlashDeconv.plan + fit_transform, Python included - alone on the GPU:
    a LoopbackComm stands in for the process group (its all-reduce of the plan counts returns the job's totals, the native loop
    runs over the loopback transport for the job's iteration count).  One interval per rank, host clock, device idle before and
    after; the fastest of a few repetitions."""
    from flashdeconv_amd.distributed import LoopbackComm, ShardedFlashDeconv
    W = len(ranks)
    G = X.shape[1]
    t = lambda: (torch.cuda.synchronize(), time.perf_counter())[1]
    reps = reps or int(os.environ.get("FDX_RANK_REPS", "6"))
    times["driver_ms"] = []
    for r, S in enumerate(ranks):
        dev = S["H"].device
        Y = make_rows(S["lo"], S["hi"])
        totals = {(3,): torch.tensor([float(nnz_total), 0.0, 0.0], dtype=torch.float64, device=dev)}
        model = ShardedFlashDeconv(sketch_dim=d, preprocess=pre, n_hvg=G, max_iter=n_iter, tol=1e-300, knn_ties=knn_ties,
                                   comm=LoopbackComm(r, W, totals))
        best = None
        for rep in range(reps):
            t0 = t()
            model.plan(coords, X)
            model.fit_transform(Y, X)
            dt = (t() - t0) * 1e3
            best = dt if best is None else min(best, dt)
        assert model.info_["n_iterations"] == n_iter and model.n_halo == S["n_halo"], (model.info_, model.n_halo, S["n_halo"])
        times["driver_ms"].append(best)
        model.close()
        del Y, model
        torch.cuda.empty_cache()


def assemble(torch, ranks, results, n, K, want_props=True):
    """(beta, proportions) of all spots, (n, K) row-major in the caller's order, through fdx_normalize_dev per rank."""
    from flashdeconv_amd import _lib
    lib = _lib.load()
    dev = ranks[0]["H"].device
    beta = torch.empty((n, K), dtype=torch.float64, device=dev)
    prop = torch.empty((n, K), dtype=torch.float64, device=dev) if want_props else None
    for S, res in zip(ranks, results):
        b = torch.empty((S["n_own"], K), dtype=torch.float64, device=dev)
        p = torch.empty((S["n_own"], K), dtype=torch.float64, device=dev) if want_props else None
        _lib.check(lib.fdx_normalize_dev(ctypes.c_void_p(S["beta"][res["buffer"]].data_ptr()), S["ld"], S["n_own"], K,
                                         ctypes.c_void_p(b.data_ptr()), ctypes.c_void_p(p.data_ptr()) if want_props else None,
                                         _st(torch)))
        beta[S["own"]] = b
        if want_props:
            prop[S["own"]] = p
    torch.cuda.synchronize()
    return beta, prop


def run_config5(torch, W, n=10_000_000, G=5000, K=50, d=1024, seed=11, max_iter=100, tol=1e-4, coords=None, keep=None, alone=False,
                pre="raw"):
    """The whole configs[4] job (or, with other n / G / K / d, configs[3]) with W virtual ranks; returns (beta, proportions, info
    dict with per-rank stage times).  alone=True: also every rank's iteration loop and finish timed alone (loopback transport) and
    `per_rank_critical_path_ms` = plan (lists + band + symmetrise) + localize + prepare + solve + finish of each rank."""
    from flashdeconv_amd import _lib
    from flashdeconv_amd.distributed import diag_mean
    dev = torch.device("cuda", torch.cuda.current_device())
    times = {}
    g = torch.Generator(device=dev)
    g.manual_seed(seed)
    X32 = torch.randn(K, G, generator=g, device=dev, dtype=torch.float32)
    X = X32.double().cpu().numpy()
    if coords is None:
        t0 = time.perf_counter()
        raw = torch.rand(n, 2, generator=g, device=dev, dtype=torch.float64) * float(np.sqrt(n))
        coords = morton_sorted_coords(torch, raw)
        del raw
        torch.cuda.synchronize()
        times["coords_ms"] = (time.perf_counter() - t0) * 1e3
    ranks, nnz, ties, bounds = virtual_plan(torch, coords, W, 6, times)
    # pre="log_cpm" (the count-like family's cost profile: the float32-class log1p sketch, every sweep of max_iter when tol is
    # tiny): the same rows made non-negative - what the kernels cost does not depend on the values
    mode = _lib.PRE_LOG_CPM if pre == "log_cpm" else _lib.PRE_RAW
    rows = (lambda lo, hi: gaussian_rows(torch, X32, lo, hi, seed).abs_()) if pre == "log_cpm" else (lambda lo, hi: gaussian_rows(torch, X32, lo, hi, seed))
    if pre == "log_cpm":
        X = np.abs(X)
    yty = virtual_prepare(torch, ranks, X, rows, d, mode, 0, times)
    gmean = diag_mean(ranks[0]["XtX_h"])
    lam = 0.005 * gmean / max(nnz / n, 1.0)                      # core/spatial.py:181-190 (lambda_spatial="auto")
    rho_eff = 0.01 * gmean                                       # core/solver.py:359-360
    results = virtual_solve(torch, ranks, K, lam, rho_eff, tol, max_iter, times)
    beta, prop = assemble(torch, ranks, results, n, K)
    if alone:
        alone_solve_and_finish(torch, ranks, K, lam, rho_eff, results[0]["n_iterations"], times)
        if times.get("plan_route") == "pipeline":
            plan_ms = times["plan_ms"]
        else:
            plan_ms = [times["knn_lists_ms"][r] + times["from_lists_ms"][r] + times["localize_ms"][r] for r in range(W)]
            times["plan_ms"] = plan_ms
        # sum of the separately synchronised stages (what rounds 3-4 reported) ...
        times["per_rank_stage_sum_ms"] = [plan_ms[r] + times["prepare_ms"][r] + times["solve_alone_ms"][r] + times["finish_alone_ms"][r]
                                          for r in range(W)]
        times["per_rank_critical_path_ms"] = times["per_rank_stage_sum_ms"]
        if times.get("plan_route") == "pipeline":
            # ... and the rank's share timed as ONE interval, queued as the driver queues it
            alone_pipelined(torch, coords, ranks, bounds, X, rows, d, mode, K,
                            lam, rho_eff, results[0]["n_iterations"], times)
            # ... and through the real driver class (Python included): the figure the projection uses
            driver_alone(torch, coords, ranks, X, rows, d, K, nnz,
                         results[0]["n_iterations"], times, pre=pre)
            times["per_rank_critical_path_ms"] = times["driver_ms"]
    info = dict(n=n, G=G, K=K, d=d, world=W, nnz=nnz, knn_ties=ties, lambda_used=lam, rho_eff=rho_eff, YtY=yty,
                n_iterations=[r["n_iterations"] for r in results], converged=[r["converged"] for r in results],
                final_change=[r["final_change"] for r in results], n_own=[R["n_own"] for R in ranks],
                n_halo=[R["n_halo"] for R in ranks], times={k: ([round(x, 2) for x in v] if isinstance(v, list) else
                                                              (round(v, 2) if isinstance(v, float) else v)) for k, v in times.items()})
    if keep is not None:
        keep.update(ranks=ranks, results=results, coords=coords, X=X, lam=lam, rho_eff=rho_eff)
    else:
        for R in ranks:
            R["g"].close()
    return beta, prop, info


if __name__ == "__main__":
    import json
    import torch
    W = int(sys.argv[1]) if len(sys.argv) > 1 else 8
    n = int(sys.argv[2]) if len(sys.argv) > 2 else 10_000_000
    b, p, info = run_config5(torch, W, n=n)
    print(json.dumps(info))
