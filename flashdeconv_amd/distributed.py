"""Spot-sharded multi-GPU fit: one process per GPU, torch.distributed (RCCL over xGMI) for the halo exchange.

The BCD update is Jacobi across spots (reference core/solver.py:161-166 reads only the previous iterate), so the result
does not depend on how spots are partitioned.  Spots are put in Morton order of a uniform grid and cut into `world`
contiguous ranges; rank r owns one range plus a read-only halo (the neighbours of its spots that live elsewhere).

Per iteration every rank
  1. runs one BCD sweep over its own spots (csrc/bcd_*.cpp through fdx_bcd_sweep_dev),
  2. sends the new abundances of its boundary spots to the ranks that hold them as halo (grouped point-to-point
     send/recv - xGMI is point-to-point, a few hundred KB per peer; no bulk all-reduce anywhere),
  3. all-reduces (MAX) the 128 convergence slots of that iteration, so the next sweep's on-device stopping test sees
     the global statistics and every rank takes the same decision.
One-off all-reduces (SUM): YtY and the four objective partials.  Everything else is local.

`ShardedSolver` is written against two small interfaces (`backend`: the per-rank compute, `comm`: the collectives) so
the same loop runs on GPUs (HipBackend + torch.distributed/nccl) and in the CPU tests (oracle sweep + gloo).
"""
import ctypes
import os
import time

import numpy as np

from . import _lib


def diag_mean(XtX):
    """mean(diag(XtX)) summed in index order, exactly as the single-GPU driver does (fit.cpp), so that lambda and the
    scaled rho - and with them every bit of the solve - are identical on the sharded and unsharded paths."""
    acc = 0.0
    for k in range(XtX.shape[0]):
        acc += float(XtX[k, k])
    return acc / XtX.shape[0]


def combine_moment_sums(mean, var, n, device=None):
    """(sum z, sum z^2) per gene from a shard's mean and ddof-1 variance over n spots (the statistics
    fdx_gene_moments_dev reports: var = n/(n-1) * (sum z^2 / n - mean^2), utils/genes.py:74-83)."""
    mean = np.asarray(mean, dtype=np.float64)
    var = np.asarray(var, dtype=np.float64)
    s1 = mean * n
    s2 = (var * ((n - 1) / n) + mean * mean) * n if n >= 2 else mean * mean * n
    if device is None:
        return s1, s2
    import torch
    return torch.from_numpy(s1).to(device), torch.from_numpy(s2).to(device)


def moments_from_sums(s1, s2, n):
    """Global mean and ddof-1 variance from the all-reduced sums (same formula as the device fold kernel)."""
    mean = s1 / n
    var = np.maximum((s2 / n - mean * mean) * (n / (n - 1)), 0.0) if n >= 2 else np.zeros_like(mean)
    return mean, var


def shard_bounds(n, world):
    """Range starts of `world` contiguous, 256-aligned, near-equal shards of n sorted spots (tiles of the sweep are
    256 spots, so shard boundaries coincide with tile boundaries)."""
    tiles = (n + 255) // 256
    b = [min(n, ((tiles * r) // world) * 256) for r in range(world)] + [n]
    return np.asarray(b, dtype=np.int64)


class TorchComm:
    """Collectives over a torch.distributed process group (nccl = RCCL on ROCm, gloo on CPU)."""

    def __init__(self, group=None):
        import torch.distributed as dist
        self.dist = dist
        self.group = group
        self.rank = dist.get_rank(group)
        self.world = dist.get_world_size(group)

    def all_reduce_max(self, t):
        if self.world > 1:
            self.dist.all_reduce(t, op=self.dist.ReduceOp.MAX, group=self.group)

    def all_reduce_sum(self, t):
        if self.world > 1:
            self.dist.all_reduce(t, op=self.dist.ReduceOp.SUM, group=self.group)

    def all_gather_rows(self, nbr, cnt, bounds):
        """Every rank has written rows [bounds[rank], bounds[rank+1]) of nbr (n, kk) / cnt (n); fill in the rows of the
        other ranks.  One all-gather of equal, padded segments (the shards differ by at most 256 rows)."""
        import torch
        W, r = self.world, self.rank
        if W == 1:
            return
        kk = nbr.shape[1]
        seg = int(max(int(bounds[i + 1] - bounds[i]) for i in range(W)))
        lo, hi = int(bounds[r]), int(bounds[r + 1])
        send = torch.zeros((seg, kk + 1), dtype=nbr.dtype, device=nbr.device)
        send[:hi - lo, :kk] = nbr[lo:hi]
        send[:hi - lo, kk] = cnt[lo:hi]
        out = torch.empty((W * seg, kk + 1), dtype=nbr.dtype, device=nbr.device)
        try:
            self.dist.all_gather_into_tensor(out, send, group=self.group)
        except (RuntimeError, NotImplementedError):                      # backend without the fused form
            parts = [out[q * seg:(q + 1) * seg] for q in range(W)]
            self.dist.all_gather(parts, send, group=self.group)
        for q in range(W):
            if q == r:
                continue
            a, b = int(bounds[q]), int(bounds[q + 1])
            nbr[a:b] = out[q * seg:q * seg + (b - a), :kk]
            cnt[a:b] = out[q * seg:q * seg + (b - a), kk]

    def exchange(self, send_bufs, recv_bufs):
        """send_bufs / recv_bufs: {peer: contiguous tensor}.  Grouped point-to-point."""
        if self.world > 1 and not getattr(self, "_stream_ordered", None):
            # nccl / RCCL operations are ordered on the device streams.  Any other backend (gloo in the tests, with W processes on one
            # GPU) reads device buffers from the HOST as soon as it is called - the kernels that pack them have to be finished
            # (measured without this: abundances 1e-9 .. 3e-5 off and different from run to run, tools/class_ranks_gloo.py)
            if getattr(self, "_stream_ordered", None) is None:
                self._stream_ordered = self.dist.get_backend(self.group) == "nccl"
            if not self._stream_ordered:
                for t in list(send_bufs.values()) + list(recv_bufs.values()):
                    if t.is_cuda:
                        import torch
                        torch.cuda.current_stream(t.device).synchronize()
                        break
        ops = []
        for peer, t in sorted(recv_bufs.items()):
            ops.append(self.dist.P2POp(self.dist.irecv, t, peer, self.group))
        for peer, t in sorted(send_bufs.items()):
            ops.append(self.dist.P2POp(self.dist.isend, t, peer, self.group))
        if ops:
            for req in self.dist.batch_isend_irecv(ops):
                req.wait()


class LoopbackComm:
    """MEASUREMENT ONLY: rank `rank` of a `world`-rank job ALONE on its GPU.  The collectives of the sharded driver return at
    once with what the real job would have produced - sums come from `totals` (tensor shape -> the all-reduced values, recorded
    from a run of the real job) when given, else the rank's own values - and the native iteration loop runs over the loopback
    transport (fdx_comm_init_loopback: the exchange is a device copy of the rank's own staging).  `bench.py --virtual-ranks` times
    every rank's whole share of the sharded fit through ShardedFlashDeconv with this comm: the driver's real code path - Python,
    launches, read-backs - without peers and without wire time.  Results are NOT those of the job (the halo values are the
    rank's own)."""

    loopback = True

    def __init__(self, rank, world, totals=None):
        self.rank, self.world = int(rank), int(world)
        self.totals = totals or {}
        self.group = None

    def all_reduce_max(self, t):
        pass

    def all_reduce_sum(self, t):
        tot = self.totals.get(tuple(t.shape))
        if tot is not None:
            t.copy_(tot.to(t.device))

    def all_gather_rows(self, nbr, cnt, bounds):
        raise RuntimeError("LoopbackComm has no peers to gather from")

    def exchange(self, send_bufs, recv_bufs):
        for peer, t in recv_bufs.items():
            src = send_bufs.get(peer)
            if src is not None and src.shape == t.shape:
                t.copy_(src)


class HaloExchange:
    """Moves boundary rows of a type-major (K, ld) abundance buffer into the halo columns of the peers."""

    def __init__(self, comm, n_own, send_idx, send_counts, recv_counts):
        import torch
        self.comm = comm
        self.n_own = int(n_own)
        self.send_idx = send_idx                      # int64 tensor, own local indices grouped by destination rank
        self.send_counts = [int(c) for c in send_counts]
        self.recv_counts = [int(c) for c in recv_counts]
        self.send_off = np.concatenate([[0], np.cumsum(self.send_counts)]).astype(int)
        self.recv_off = np.concatenate([[0], np.cumsum(self.recv_counts)]).astype(int)
        self._torch = torch

    def __call__(self, beta):
        torch = self._torch
        K = beta.shape[0]
        key = (K, beta.dtype, beta.device)
        if getattr(self, "_buf_key", None) != key:          # staging buffers are reused across iterations and fits
            self._send = {r: torch.empty((K, c), dtype=beta.dtype, device=beta.device)
                          for r, c in enumerate(self.send_counts) if c and r != self.comm.rank}
            self._recv = {r: torch.empty((K, c), dtype=beta.dtype, device=beta.device)
                          for r, c in enumerate(self.recv_counts) if c and r != self.comm.rank}
            self._idx = {r: self.send_idx[self.send_off[r]:self.send_off[r + 1]] for r in self._send}
            self._buf_key = key
        for r, buf in self._send.items():
            torch.index_select(beta, 1, self._idx[r], out=buf)
        self.comm.exchange(self._send, self._recv)
        for r, t in self._recv.items():
            lo = self.n_own + self.recv_off[r]
            beta[:, lo:lo + self.recv_counts[r]] = t


class ShardedSolver:
    """The reference's bcd_solve loop (core/solver.py:385-413) over a spot shard."""

    def __init__(self, backend, comm, halo, K, ld, n_own, n_total, max_iter=100, tol=1e-4):
        self.backend, self.comm, self.halo = backend, comm, halo
        self.K, self.ld, self.n_own, self.n_total = K, ld, n_own, n_total
        self.max_iter, self.tol = int(max_iter), float(tol)
        self.sweep_events = None          # set to [] to collect (start, end) torch.cuda.Event pairs around every sweep

    def run(self, new_buffer, lam, rho_eff):
        """new_buffer(shape, dtype) -> zeroed tensor on the right device.  Returns (beta_final, info)."""
        import torch
        be, K = self.backend, self.K
        beta = [new_buffer((K, self.ld), torch.float64), new_buffer((K, self.ld), torch.float64)]
        be.init_beta(beta[0], self.n_total)            # 1/K on own + halo columns, 0 on the pad (solver.py:372)
        stats = new_buffer((max(self.max_iter, 1), 128), torch.float64)   # bit patterns of doubles >= 0
        rel = new_buffer((max(self.max_iter, 1),), torch.float64)
        done, n_iter, converged, chunk = 0, 0, False, 4
        rc_all = []
        while done < self.max_iter and not converged:
            end = min(self.max_iter, done + chunk)
            for it in range(done, end):
                if self.sweep_events is not None:
                    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                    e0.record()
                be.sweep(it, beta[it & 1], beta[(it + 1) & 1], lam, rho_eff, self.tol, stats, rel)
                if self.sweep_events is not None:
                    e1.record()
                    self.sweep_events.append((e0, e1))
                self.halo(beta[(it + 1) & 1])
                self.comm.all_reduce_max(stats[it])
                if self.n_own == 0 and it + 1 < end:
                    be.fold(stats, rel, it)            # no rows, no sweep: nobody else folds sweep `it` into rel[it] (csrc/comm.cpp)
            be.fold(stats, rel, end - 1)
            rc = rel[done:end].cpu().numpy()
            for j, v in enumerate(rc):
                n_iter = done + j + 1
                rc_all.append(float(v))
                if v < self.tol:                        # solver.py:409-413
                    converged = True
                    break
            done = end
            chunk = min(chunk * 2, 32)
        final = beta[n_iter & 1]
        info = {"converged": converged, "n_iterations": n_iter,
                "final_change": rc_all[n_iter - 1] if n_iter else 0.0, "rel_changes": rc_all[:n_iter]}
        return final, info


# ------------------------------------------------------------------------------------------------ GPU side
class HipBackend:
    """Per-rank compute through the device-pointer C ABI (include/fdx.h, 'device-pointer building blocks')."""

    def __init__(self, graph, H, ldh, XtX, K, stream=None, K_real=None):
        self.lib = _lib.load()
        self.g, self.H, self.ldh, self.XtX, self.K = graph, H, int(ldh), XtX, int(K)
        self.K_real = int(K_real) if K_real else int(K)        # K > K_real: all-zero pad types (fdx_solver_padded_k)
        self.stream = stream

    def _st(self):
        import torch
        return ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)

    def init_beta(self, beta, n_fill):
        # the buffer arrives zeroed: the pad types' planes stay zero
        _lib.check(self.lib.fdx_init_beta_dev(ctypes.c_void_p(beta.data_ptr()), beta.shape[1], int(n_fill), self.K_real, self._st()))

    def sweep(self, it, b_in, b_out, lam, rho_eff, tol, stats, rel):
        _lib.check(self.lib.fdx_bcd_sweep_dev(self.g.handle, ctypes.c_void_p(self.H.data_ptr()), self.ldh,
                                              ctypes.c_void_p(self.XtX.data_ptr()), ctypes.c_void_p(b_in.data_ptr()),
                                              ctypes.c_void_p(b_out.data_ptr()), b_in.shape[1], self.K, float(lam),
                                              float(rho_eff), float(tol), int(it), ctypes.c_void_p(stats.data_ptr()),
                                              ctypes.c_void_p(rel.data_ptr()), self._st()))

    def fold(self, stats, rel, it):
        _lib.check(self.lib.fdx_bcd_fold_dev(ctypes.c_void_p(stats.data_ptr()), ctypes.c_void_p(rel.data_ptr()), int(it), self._st()))

    def objective_partials(self, beta):
        out = np.zeros(4)
        _lib.check(self.lib.fdx_objective_partials_dev(self.g.handle, ctypes.c_void_p(beta.data_ptr()), beta.shape[1],
                                                       ctypes.c_void_p(self.H.data_ptr()), self.ldh,
                                                       ctypes.c_void_p(self.XtX.data_ptr()), self.K, _lib.ptr_f64(out), self._st()))
        return out


class ShardedFlashDeconv:
    """Multi-GPU counterpart of FlashDeconv for one rank of a torch.distributed job.

        model = ShardedFlashDeconv(sketch_dim=512, ...)           # same hyper-parameters as FlashDeconv
        own_ids = model.plan(coords)                               # coords of ALL spots (replicated, on this GPU)
        props = model.fit_transform(Y_own, X)                      # rows of Y for own_ids, in that order

    Returns the proportions of the own spots (row i belongs to spot own_ids[i]); `beta_`, `info_`, `lambda_used_` as in
    the reference, `gene_idx_` when G > n_hvg (gene statistics are all-reduced over the shards).
    """

    def __init__(self, sketch_dim=512, lambda_spatial="auto", rho_sparsity=0.01, n_hvg=2000, k_neighbors=6,
                 spatial_method="knn", radius=None, max_iter=100, tol=1e-4, preprocess="log_cpm", random_state=0,
                 group=None, comm=None, n_markers_per_type=50, knn_ties="auto"):
        # the reference's constructor checks and messages (core/deconv.py:105-124), through the single-GPU estimator
        from .core.deconv import FlashDeconv
        self._proto = FlashDeconv(sketch_dim=sketch_dim, lambda_spatial=lambda_spatial, rho_sparsity=rho_sparsity, n_hvg=n_hvg,
                                  n_markers_per_type=n_markers_per_type, spatial_method=spatial_method, k_neighbors=k_neighbors,
                                  radius=radius, max_iter=max_iter, tol=tol, preprocess=preprocess, random_state=random_state)
        self.n_markers_per_type = n_markers_per_type
        # exactly tied k-th neighbour distances (regular lattices): "auto" / "ckdtree" - the graph is then the reference's own
        # (cKDTree's choice, restated on the host, csrc/kdtree_order.cpp; every rank builds it from the replicated coordinates and
        # cuts its own rows out, shards in the CALLER's spot order) as in FlashDeconv; "index" - the device rule, with a warning
        if knn_ties not in ("auto", "index", "ckdtree"):
            raise ValueError(f"knn_ties must be 'auto', 'index' or 'ckdtree', got {knn_ties}")
        self.knn_ties = knn_ties
        self.sketch_dim, self.lambda_spatial, self.rho_sparsity = sketch_dim, lambda_spatial, rho_sparsity
        self.n_hvg, self.k_neighbors, self.spatial_method, self.radius = n_hvg, k_neighbors, spatial_method, radius
        self.max_iter, self.tol, self.preprocess, self.random_state = max_iter, tol, preprocess, random_state
        self.comm = comm if comm is not None else TorchComm(group)
        self._native = None           # fdx_comm handle: the C++ / RCCL iteration loop (csrc/comm.cpp)
        self._full = self._local = None
        self.timings_ = {}
        self.knn_ties_ = 0
        self._profile = bool(os.environ.get("FDX_DIST_TIMING"))
        self._trace = [] if os.environ.get("FDX_TRACE_DRIVER") else None

    def native_comm(self):
        """libfdx's own RCCL communicator for the native iteration loop (fdx_sharded_solve_dev): rank 0 draws the
        ncclUniqueId, torch.distributed ships the 128 bytes, every rank calls ncclCommInitRank through fdx_comm_init.
        None when the process group is not an nccl one (the gloo tests use the Python loop) or FDX_PY_LOOP is set."""
        import torch
        if self._native is not None or os.environ.get("FDX_PY_LOOP"):
            return self._native
        lib = _lib.load()
        if getattr(self.comm, "native_handle", None) is not None:   # a comm that brings its own libfdx communicator (thread ranks
            self._native = self.comm.native_handle                  # over fdx_comm_init_local in the tests)
            self._native_borrowed = True
            return self._native
        if getattr(self.comm, "loopback", False):              # measurement: one rank alone (LoopbackComm)
            h = ctypes.c_void_p()
            _lib.check(lib.fdx_comm_init_loopback(self.comm.rank, self.comm.world, ctypes.byref(h)))
            self._native = h
            return self._native
        dist = getattr(self.comm, "dist", None)
        if dist is None or dist.get_backend(self.comm.group) != "nccl":
            return None
        ident = np.zeros(128, dtype=np.uint8)
        if self.comm.rank == 0:
            _lib.check(lib.fdx_comm_unique_id(ident.ctypes.data))
        t = torch.from_numpy(ident).cuda()
        dist.broadcast(t, src=dist.get_global_rank(self.comm.group, 0) if self.comm.group is not None else 0, group=self.comm.group)
        ident = t.cpu().numpy()
        h = ctypes.c_void_p()
        err = None
        try:
            _lib.check(lib.fdx_comm_init(ident.ctypes.data, self.comm.rank, self.comm.world, ctypes.byref(h)))
        except _lib.FdxError as e:
            err = str(e)
        # every rank must take the same loop: one rank without its communicator and the others inside the native exchange would
        # wait for each other for ever - the ranks agree on the outcome over torch's group first
        ok = torch.tensor([0.0 if err else 1.0], device="cuda")
        dist.all_reduce(ok, op=dist.ReduceOp.MIN, group=self.comm.group)
        if float(ok.item()) < 1.0:
            # LOUD fallback: the Python exchange loop over torch.distributed (same kernels, same bits, slower per iteration)
            import warnings
            if h.value:
                lib.fdx_comm_destroy(h)
            self.native_error_ = err or "libfdx's RCCL communicator could not be created on another rank"
            os.environ["FDX_PY_LOOP"] = "1"          # do not try again in this process
            warnings.warn(f"libfdx could not create its own RCCL communicator on every rank ({self.native_error_}); the sharded fit falls "
                          "back to the Python exchange loop over torch.distributed", RuntimeWarning, stacklevel=2)
            return None
        self._native = h
        return self._native

    def comm_report(self):
        """What ran: for bench.py's result line and for debugging a first multi-GPU contact."""
        import torch
        rep = {"rank": int(self.comm.rank), "device": int(torch.cuda.current_device()), "plan_route": self.plan_route_,
               "loop": "native" if self._native is not None else "python", "rccl_ranks": None,
               "native_comm_error": getattr(self, "native_error_", None)}
        if self._native is not None:
            c = ctypes.c_int32(0)
            _lib.check(_lib.load().fdx_comm_rccl_count(self._native, ctypes.byref(c)))
            rep["rccl_ranks"] = int(c.value)
        return rep

    def close(self):
        if self._native is not None:
            if not getattr(self, "_native_borrowed", False):
                _lib.load().fdx_comm_destroy(self._native)
            self._native = None
        for g in (self._local, self._full):
            if g is not None:
                g.close()
        self._local = self._full = None

    def _mark(self, name):
        """FDX_TRACE_DRIVER=1: host clock at the steps of plan / fit_transform (no synchronisation), printed after the fit."""
        if self._trace is not None:
            self._trace.append((name, time.perf_counter()))

    def _mark_dump(self):
        if self._trace:
            import sys
            t0 = self._trace[0][1]
            prev = t0
            out = []
            for name, t in self._trace:
                out.append(f"{name} +{1e6 * (t - prev):.0f}")
                prev = t
            print(f"[fdx-driver rank {self.comm.rank}] total {1e6 * (prev - t0):.0f} us: " + ", ".join(out), file=sys.stderr)
            self._trace = []

    def _tick(self, name, t0):
        """Stage timing for tools/dist_probe.py (FDX_DIST_TIMING=1: synchronises, so only for diagnosis)."""
        if not self._profile:
            return t0
        import torch
        torch.cuda.current_stream().synchronize()      # this stream only: the leverage side stream keeps running
        t1 = time.perf_counter()
        self.timings_[name] = self.timings_.get(name, 0.0) + (t1 - t0) * 1e3
        return t1

    def plan(self, coords, X=None):
        """Build the (replicated) spatial graph, cut it into shards, return the caller's ids of this rank's spots.
        With the signatures X given, their leverage SVD (one workgroup, side stream) runs under the graph build and
        the next fit_transform(Y_own, X) collects it."""
        import torch
        from .utils.genes import LeverageJob
        lib = _lib.load()
        self._lev_job = None
        self.knn_ties_ = 0
        self._mark("plan:entry")
        if X is not None:
            Xj = np.ascontiguousarray(X, dtype=np.float64)
            if Xj.shape[1] <= self.n_hvg:
                self._lev_job = (Xj, LeverageJob(Xj))
        self._mark("plan:leverage job")
        assert coords.is_cuda and coords.dtype == torch.float64
        coords = coords.contiguous()
        n, dim = coords.shape
        if self.spatial_method == "knn" and min(int(self.k_neighbors), int(n) - 1) > 63:
            # (FlashDeconv takes the host cKDTree route there; a shard plan has no such route: say so before any work)
            raise ValueError(f"k_neighbors = {self.k_neighbors}: the sharded plan builds k-NN lists of at most 63 neighbours per spot "
                             "(FlashDeconv on one GPU takes any k; spatial_method='radius' builds denser graphs on shards)")
        st = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
        for g in (self._local, self._full):
            if g is not None:
                g.close()
        self._local = self._full = None
        t0 = time.perf_counter()
        h = ctypes.c_void_p()
        self.bounds = shard_bounds(n, self.comm.world)
        k = int(self.k_neighbors)
        sharded_knn = self.spatial_method == "knn" and self.comm.world > 1 and n >= 2 and k >= 1
        self.plan_route_ = None
        self._pending_plan = None
        self._coords = coords
        self.n_total_spots = n
        self.n_own = int(self.bounds[self.comm.rank + 1] - self.bounds[self.comm.rank])
        if (sharded_knn and dim <= 3 and 2 <= self.comm.world <= 32 and min(k, n - 1) + 1 <= 64
                and not os.environ.get("FDX_PLAN_ALLGATHER") and not os.environ.get("FDX_PLAN_STEPWISE")):
            # ONE queued pipeline per rank (fdx_graph_shard_knn_dev): band lists, own rows of the symmetrised graph, halo, local
            # ELL, tile tables, send lists - nothing returns to the host after the bounding box.  The counts (and the one
            # all-reduce of the plan: edges, ties, "a walk left its block") are taken over by _finish_plan(): at once when the
            # tie rule may still change the graph, otherwise behind the sketch that fit_transform queues next.
            self.plan_route_ = "band"
            lo, hi = int(self.bounds[self.comm.rank]), int(self.bounds[self.comm.rank + 1])
            if hi > lo:
                hl = ctypes.c_void_p()
                _lib.check(lib.fdx_graph_shard_knn_dev(ctypes.c_void_p(coords.data_ptr()), n, dim, k, self.comm.world,
                                                       _lib.ptr_i64(self.bounds), self.comm.rank, st, ctypes.byref(hl)))
                self._local = _lib.Graph(hl.value)
            else:
                self._plan_stepwise_local(coords, lo, hi)          # a rank without rows: nothing to queue, no collective inside
            self._mark("plan:shard_knn queued")
            self._pending_plan = "shard"
            perm = torch.empty(max(self.n_own, 1), dtype=torch.int32, device=coords.device)
            _lib.check(lib.fdx_graph_perm_dev(self._local.handle, ctypes.c_void_p(perm.data_ptr()), st))
            self.own_ids = perm[:self.n_own].long()
            t0 = self._tick("plan_build", t0)
            self._mark("plan:perm")
            if self._profile:
                self._finish_plan()
            return self.own_ids
        if sharded_knn:
            lo, hi = int(self.bounds[self.comm.rank]), int(self.bounds[self.comm.rank + 1])
            kk = min(k, n - 1) + 1
            nbr = torch.empty((n, kk), dtype=torch.int32, device=coords.device)
            cnt = torch.empty((n,), dtype=torch.int32, device=coords.device)
            plan = ctypes.c_void_p()
            if not os.environ.get("FDX_PLAN_ALLGATHER") and dim <= 3:      # more than 3 coordinates: exhaustive search, no band
                # Band recompute ("recompute, don't communicate"): the lists of the own rows AND of the rows in the grid
                # cells next to an own cell, which are all the rows that can point at an own row while every k-NN walk stays inside
                # its 3 x 3 block of cells - own rows of the symmetrised graph without moving a list (include/fdx.h).  Each rank
                # checks the condition for its own rows; the flag rides in the all-reduce of the edge counts.
                self.plan_route_ = "band"
                _lib.check(lib.fdx_graph_knn_lists_band_dev(ctypes.c_void_p(coords.data_ptr()), n, dim, k, lo, hi,
                                                            ctypes.c_void_p(nbr.data_ptr()), ctypes.c_void_p(cnt.data_ptr()), st,
                                                            ctypes.byref(plan)))
                t0 = self._tick("plan_knn", t0)
                _lib.check(lib.fdx_graph_from_knn_lists_dev(plan, ctypes.c_void_p(nbr.data_ptr()), ctypes.c_void_p(cnt.data_ptr()), lo, hi,
                                                            st, ctypes.byref(h)))
                self._full = _lib.Graph(h.value)
                own = torch.tensor([float(self._full.info()[1]), float(self._full.knn_ties()), float(self._full.knn_far())],
                                   dtype=torch.float64, device=coords.device)
                self.comm.all_reduce_sum(own)
                tot = own.cpu().numpy()
                if tot[2] == 0:
                    self.nnz_total, self.knn_ties_ = int(round(float(tot[0]))), int(round(float(tot[1])))
                else:       # some rank's walk left its block (very uneven density) or a band list overflowed: exchange the lists
                    self._full.close()
                    self._full, h, plan = None, ctypes.c_void_p(), ctypes.c_void_p()
        if sharded_knn and self._full is None:
            # sharded build by exchange: own k-NN lists -> all-gather of the list rows -> own rows of the symmetrised graph
            self.plan_route_ = "allgather"
            _lib.check(lib.fdx_graph_knn_lists_dev(ctypes.c_void_p(coords.data_ptr()), n, dim, k, lo, hi,
                                                   ctypes.c_void_p(nbr.data_ptr()), ctypes.c_void_p(cnt.data_ptr()), st,
                                                   ctypes.byref(plan)))
            t0 = self._tick("plan_knn", t0)
            try:
                self.comm.all_gather_rows(nbr, cnt, self.bounds)
            except Exception:
                # a failed collective leaves the ranks out of step: release the plan (it owns device buffers; its rows of
                # the other ranks stay marked empty) without letting a second error mask the first, and re-raise
                nbr[:lo], nbr[hi:], cnt[:lo], cnt[hi:] = -1, -1, 0, 0
                dead = ctypes.c_void_p()
                lib.fdx_graph_from_knn_lists_dev(plan, ctypes.c_void_p(nbr.data_ptr()), ctypes.c_void_p(cnt.data_ptr()), lo, hi, st,
                                                 ctypes.byref(dead))
                if dead.value:
                    _lib.Graph(dead.value).close()
                raise
            t0 = self._tick("plan_gather", t0)
            _lib.check(lib.fdx_graph_from_knn_lists_dev(plan, ctypes.c_void_p(nbr.data_ptr()),
                                                        ctypes.c_void_p(cnt.data_ptr()), lo, hi, st, ctypes.byref(h)))
            self._full = _lib.Graph(h.value)
            # nnz of the whole graph (auto lambda) and, in the same all-reduce, the spots whose k-th neighbour is tied
            own_nnz = torch.tensor([float(self._full.info()[1]), float(self._full.knn_ties())], dtype=torch.float64,
                                   device=coords.device)
            self.comm.all_reduce_sum(own_nnz)
            tot = own_nnz.cpu().numpy()
            self.nnz_total, self.knn_ties_ = int(round(float(tot[0]))), int(round(float(tot[1])))
        elif not sharded_knn:
            # "radius" / "grid" resolve their radius exactly as FlashDeconv does (utils/graph.py:163-212)
            method, gk, gradius = self._proto._graph_request(coords, None)
            if method == _lib.GRAPH_RADIUS and self.comm.world > 1 and n >= 2:
                # sharded build: a radius graph is symmetric by construction, the own rows need no exchange
                lo, hi = int(self.bounds[self.comm.rank]), int(self.bounds[self.comm.rank + 1])
                _lib.check(lib.fdx_graph_build_radius_rows_dev(ctypes.c_void_p(coords.data_ptr()), n, dim, float(gradius), lo, hi, st,
                                                               ctypes.byref(h)))
                self._full = _lib.Graph(h.value)
                own_nnz = torch.tensor([float(self._full.info()[1])], dtype=torch.float64, device=coords.device)
                self.comm.all_reduce_sum(own_nnz)                           # nnz of the whole graph (auto lambda)
                self.nnz_total = int(round(float(own_nnz.item())))
            else:
                _lib.check(lib.fdx_graph_build_dev(ctypes.c_void_p(coords.data_ptr()), n, dim, method, gk, float(gradius), st,
                                                   ctypes.byref(h)))
                self._full = _lib.Graph(h.value)
                self.nnz_total = self._full.info()[1]
                self.knn_ties_ = self._full.knn_ties() if method == _lib.GRAPH_KNN else 0
        t0 = self._tick("plan_build", t0)
        self._resolve_ties_and_localize(coords, st, t0)
        return self.own_ids

    def _plan_stepwise_local(self, coords, lo, hi):
        """This rank's local graph by the three stepwise calls of the band route (lists of own rows + band, own rows of the
        symmetrised graph, localize) - no collective inside: the remedy when a bound of the queued pipeline was too small, and
        the path of a rank that owns no row."""
        import torch
        lib = _lib.load()
        n, dim = coords.shape
        k = int(self.k_neighbors)
        st = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
        kk = min(k, n - 1) + 1
        nbr = torch.empty((n, kk), dtype=torch.int32, device=coords.device)
        cnt = torch.empty((n,), dtype=torch.int32, device=coords.device)
        plan, h, hl = ctypes.c_void_p(), ctypes.c_void_p(), ctypes.c_void_p()
        _lib.check(lib.fdx_graph_knn_lists_band_dev(ctypes.c_void_p(coords.data_ptr()), n, dim, k, lo, hi,
                                                    ctypes.c_void_p(nbr.data_ptr()), ctypes.c_void_p(cnt.data_ptr()), st, ctypes.byref(plan)))
        _lib.check(lib.fdx_graph_from_knn_lists_dev(plan, ctypes.c_void_p(nbr.data_ptr()), ctypes.c_void_p(cnt.data_ptr()), lo, hi,
                                                    st, ctypes.byref(h)))
        full = _lib.Graph(h.value)
        self._step_counts = (float(full.info()[1]), float(full.knn_ties()), float(full.knn_far()))
        _lib.check(lib.fdx_graph_localize(full.handle, self.comm.world, _lib.ptr_i64(self.bounds), self.comm.rank, st, ctypes.byref(hl)))
        full.close()
        if self._local is not None:
            self._local.close()
        self._local = _lib.Graph(hl.value)
        self.plan_rebuilt_stepwise_ = True          # (diagnostic: this rank's graph came from the remedy / the no-row path)

    def _ties_remedy_lists(self, coords, st):
        """The reference's neighbour lists (cKDTree's choice among equidistant candidates) for this rank's own rows and band, in
        place of the device's index-rule lists; then the stepwise symmetrise + localize.  Needs the band to hold every row that
        can point at an own row - the same condition as the band recompute itself (no far walk: checked before this is called)."""
        import torch
        from .utils.graph import _ckdtree_restatement_matches_scipy
        lib = _lib.load()
        n, dim = coords.shape
        k = int(self.k_neighbors)
        kk = min(k, n - 1) + 1
        lo, hi = int(self.bounds[self.comm.rank]), int(self.bounds[self.comm.rank + 1])
        dev = coords.device
        nbr = torch.empty((n, kk), dtype=torch.int32, device=dev)
        cnt = torch.empty((n,), dtype=torch.int32, device=dev)
        plan, h, hl = ctypes.c_void_p(), ctypes.c_void_p(), ctypes.c_void_p()
        _lib.check(lib.fdx_graph_knn_lists_band_dev(ctypes.c_void_p(coords.data_ptr()), n, dim, k, lo, hi,
                                                    ctypes.c_void_p(nbr.data_ptr()), ctypes.c_void_p(cnt.data_ptr()), st, ctypes.byref(plan)))
        perm_t = torch.empty(n, dtype=torch.int32, device=dev)
        _lib.check(lib.fdx_graph_plan_order_dev(plan, ctypes.c_void_p(perm_t.data_ptr()), None, st))
        rows_pos = torch.nonzero(cnt > 0).flatten()                       # own rows + band: the rows that have lists
        if rows_pos.numel():
            ids = np.ascontiguousarray(_lib.tensor_to_host(perm_t[rows_pos].long()))
            cd = coords if (coords.dtype == torch.float64 and coords.is_contiguous()) else coords.double().contiguous()
            _ckdtree_restatement_matches_scipy()
            # every rank builds the tree of the replicated coordinates: the ranks of one host share its cores
            # (for the duration of the call; a caller's own setting wins)
            share = max(1, _lib.host_cpu_budget() // max(1, int(os.environ.get("LOCAL_WORLD_SIZE", self.comm.world))))
            lib.fdx_kdtree_set_threads(int(share))
            # the restated tree on the host, its queries for these rows on the device; the answers (caller ids, self included) go to
            # solver positions, self dropped (utils/graph.py:70-74), at the rows' positions
            _lib.check(lib.fdx_graph_plan_set_ckdtree_lists_dev(plan, None, ctypes.c_void_p(cd.data_ptr()), n, dim,
                                                                ids.ctypes.data, len(ids), ctypes.c_void_p(nbr.data_ptr()),
                                                                ctypes.c_void_p(cnt.data_ptr()), st))
            lib.fdx_kdtree_set_threads(0)
        _lib.check(lib.fdx_graph_from_knn_lists_dev(plan, ctypes.c_void_p(nbr.data_ptr()), ctypes.c_void_p(cnt.data_ptr()), lo, hi,
                                                    st, ctypes.byref(h)))
        full = _lib.Graph(h.value)
        own = torch.tensor([float(full.info()[1])], dtype=torch.float64, device=dev)
        self.comm.all_reduce_sum(own)                                        # edges of the reference's graph (auto lambda)
        self.nnz_total = int(round(float(own.item())))
        _lib.check(lib.fdx_graph_localize(full.handle, self.comm.world, _lib.ptr_i64(self.bounds), self.comm.rank, st, ctypes.byref(hl)))
        full.close()
        if self._local is not None:
            self._local.close()
        self._local = _lib.Graph(hl.value)
        perm = torch.empty(max(self.n_own, 1), dtype=torch.int32, device=dev)
        _lib.check(lib.fdx_graph_perm_dev(self._local.handle, ctypes.c_void_p(perm.data_ptr()), st))
        self.own_ids = perm[:self.n_own].long()                              # the same Morton range as before

    def _finish_plan(self, totals=None):
        """Second half of a queued plan (fdx_graph_shard_knn_dev): wait for the counts, all-reduce (edges, tied rows, far flag),
        apply the remedies (a bound too small: this rank rebuilds stepwise; a far walk anywhere: every rank rebuilds by the list
        exchange; ties under "auto" / "ckdtree": the reference's neighbour choice), halo bookkeeping.  totals: (edges, tied rows,
        far) already all-reduced by fdx_shard_fit_dev."""
        if getattr(self, "_pending_plan", None) is None:
            return
        import torch
        lib = _lib.load()
        self._pending_plan = None
        coords = self._coords
        n = coords.shape[0]
        st = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
        t0 = time.perf_counter()
        lo, hi = int(self.bounds[self.comm.rank]), int(self.bounds[self.comm.rank + 1])
        if hi > lo:
            nnz, ties, far, over = ctypes.c_int64(0), ctypes.c_int64(0), ctypes.c_int32(0), ctypes.c_int32(0)
            _lib.check(lib.fdx_graph_shard_status(self._local.handle, ctypes.byref(nnz), ctypes.byref(ties), ctypes.byref(far),
                                                  ctypes.byref(over)))
            counts = (float(nnz.value), float(ties.value), float(far.value))
            if over.value and not far.value and not (totals is not None and totals[2]):
                self._plan_stepwise_local(coords, lo, hi)          # same rows, same order, exact sizes
                counts = self._step_counts
        else:
            counts = self._step_counts
        self._mark("finish_plan:status")
        if totals is None:
            own = torch.tensor(counts, dtype=torch.float64, device=coords.device)
            self.comm.all_reduce_sum(own)
            tot = own.cpu().numpy()
        else:
            tot = np.asarray(totals, dtype=np.float64)
        t0 = self._tick("plan_counts", t0)
        self._mark("finish_plan:allreduce+cpu")
        if tot[2] != 0:
            # some rank's walk left its block (very uneven density) or a band list overflowed: every rank rebuilds by the exchange
            self._local.close()
            self._local = None
            prev = os.environ.get("FDX_PLAN_ALLGATHER")
            os.environ["FDX_PLAN_ALLGATHER"] = "1"
            try:
                lev = getattr(self, "_lev_job", None)
                self.plan(coords)
                self._lev_job = lev
            finally:
                if prev is None:
                    del os.environ["FDX_PLAN_ALLGATHER"]
                else:
                    os.environ["FDX_PLAN_ALLGATHER"] = prev
            return
        self.nnz_total, self.knn_ties_ = int(round(float(tot[0]))), int(round(float(tot[1])))
        self._full = None
        self._resolve_ties_and_localize(coords, st, t0, localized=True)

    def _resolve_ties_and_localize(self, coords, st, t0, localized=False):
        """Tail of plan(): the reference's tie order when asked for, the local graph of this rank, its halo bookkeeping."""
        import torch
        lib = _lib.load()
        n = coords.shape[0]
        self.knn_ties_resolved_ = False
        if (getattr(self, "knn_ties_", 0) and self.spatial_method == "knn" and self.knn_ties != "index" and self.comm.world > 1
                and coords.shape[1] <= 3 and not os.environ.get("FDX_TIES_FULL_GRAPH")):
            # The device builds chose among equidistant neighbours by spot index; the reference takes whichever cKDTree's query
            # meets first.  The shards stay what they are (Morton ranges: own_ids do not change): every rank asks the restated
            # tree (csrc/kdtree_order.cpp: the tree of all points - O(n log n) on the host - but only the queries of its own rows
            # and of its band) for the reference's lists, puts them in the place of the device's, and symmetrises / localises as
            # before - the reference's rows index for index.
            self._ties_remedy_lists(coords, st)
            self.plan_route_ = "ckdtree-lists"
            self.knn_ties_resolved_ = True
            localized = True
        elif getattr(self, "knn_ties_", 0) and self.spatial_method == "knn" and self.knn_ties != "index":
            # (one rank, or more than 3 coordinates) every rank builds the reference's whole graph on the host and takes its rows;
            # the graph is in the CALLER's order (no Morton sort), so a shard is a range of the caller's spot numbers.
            from .utils.graph import ckdtree_knn_adjacency
            A = ckdtree_knn_adjacency(_lib.tensor_to_host(coords).astype(np.float64), int(self.k_neighbors))
            if self._full is not None:
                self._full.close()
            self._full = _lib.Graph.from_csr(A.indptr, A.indices, n)
            self.nnz_total = int(A.nnz)
            self.plan_route_ = "ckdtree"
            self.knn_ties_resolved_ = True
            localized = False
        if getattr(self, "knn_ties_", 0) and not self.knn_ties_resolved_ and self.comm.rank == 0:
            import warnings
            warnings.warn(f"k-NN ties: {self.knn_ties_} of {n} spots have their k-th and (k+1)-th nearest neighbours at exactly the "
                          "same distance (regular lattice?): the neighbour graph depends on how ties are broken - here by spot "
                          "index, in the reference by cKDTree's traversal order.  spatial_method='grid' builds a tie-free graph "
                          "on lattices.", UserWarning, stacklevel=2)
        self.n_total_spots = n
        if not localized:
            hl = ctypes.c_void_p()
            _lib.check(lib.fdx_graph_localize(self._full.handle, self.comm.world, _lib.ptr_i64(self.bounds), self.comm.rank, st,
                                              ctypes.byref(hl)))
            if self._local is not None:
                self._local.close()
            self._local = _lib.Graph(hl.value)
            t0 = self._tick("plan_localize", t0)
            self.n_own = int(self.bounds[self.comm.rank + 1] - self.bounds[self.comm.rank])
            perm = torch.empty(max(self.n_own, 1), dtype=torch.int32, device=coords.device)
            _lib.check(lib.fdx_graph_perm_dev(self._local.handle, ctypes.c_void_p(perm.data_ptr()), st))
            self.own_ids = perm[:self.n_own].long()
        self.own_nnz_ = int(self._local.info()[1])
        n_halo = ctypes.c_int64(0)
        sc = np.zeros(self.comm.world, dtype=np.int32)
        rc = np.zeros(self.comm.world, dtype=np.int32)
        _lib.check(lib.fdx_graph_halo_info(self._local.handle, ctypes.byref(n_halo), _lib.ptr_i32(sc), _lib.ptr_i32(rc)))
        self.n_halo = int(n_halo.value)
        sidx = torch.empty(max(int(sc.sum()), 1), dtype=torch.int32, device=coords.device)
        _lib.check(lib.fdx_graph_send_indices_dev(self._local.handle, ctypes.c_void_p(sidx.data_ptr()), st))
        self._halo = HaloExchange(self.comm, self.n_own, sidx[:int(sc.sum())].long(), sc, rc)
        self._mark("plan:halo lists")
        self._tick("plan_lists", t0)

    def fit_transform(self, Y_own, X):
        import torch
        from .core.sketching import countsketch_tables
        # Every libfdx call of this driver is enqueued on torch's CURRENT stream, except the gene statistics and the column
        # gather (G > n_hvg), which use the legacy default stream: under a non-default, non-blocking torch stream those two
        # would not be ordered after the producer of Y_own, so that combination is refused rather than raced.
        if Y_own.shape[1] > self.n_hvg and torch.cuda.current_stream() != torch.cuda.default_stream():
            raise RuntimeError("ShardedFlashDeconv with gene selection active must run on torch's default stream")
        from .utils.genes import compute_leverage_scores
        lib = _lib.load()
        dev = Y_own.device
        st = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
        X = np.ascontiguousarray(X, dtype=np.float64)
        K, G = X.shape
        if _lib.is_torch_sparse_csr(Y_own):
            return self._fit_transform_csr(Y_own, X)
        self._mark("fit:entry")
        # integer counts keep the float64 transform chain, as in FlashDeconv.fit (numpy promotes them: core/deconv.py:190-191)
        from .core.deconv import device_counts_as_float
        Y_own, y_f64_math = device_counts_as_float(Y_own)
        assert Y_own.shape == (self.n_own, G)
        y_code = _lib.FDX_F32 if Y_own.dtype == torch.float32 else _lib.FDX_F64
        self.gene_idx_ = np.arange(G, dtype=np.intp)
        if G > self.n_hvg:
            # gene selection over ALL spots (utils/genes.py:293-341): every rank reduces its own rows to per-gene sums of
            # z = log1p(CPM-10k) and z^2 (fdx_gene_moments_dev), one all-reduce(SUM) of 2G doubles makes them global, the
            # ranking of the G-vector and the marker table are replicated host work
            from .utils import genes as _genes
            sums = torch.zeros((2, G), dtype=torch.float64, device=dev)
            if self.n_own:
                mean_r, var_r = _genes.gene_moments_device(ctypes.c_void_p(Y_own.data_ptr()), y_code, self.n_own, G, G)
                sums[0], sums[1] = combine_moment_sums(mean_r, var_r, self.n_own, dev)
            self.comm.all_reduce_sum(sums)
            mean, var = moments_from_sums(_lib.tensor_to_host(sums[0]), _lib.tensor_to_host(sums[1]), self.n_total_spots)
            hvg = _genes._hvg_from_moments(mean, var, self.n_hvg, 0.0125, 3.0, 0.5)
            markers, _ = _genes.select_markers(X, n_markers=self.n_markers_per_type)
            self.gene_idx_ = np.union1d(hvg, markers).astype(np.intp)
            if len(self.gene_idx_) == 0:
                raise ValueError("No genes selected. Increase n_hvg or n_markers_per_type.")
            gi32 = np.ascontiguousarray(self.gene_idx_, dtype=np.int32)
            Y_sel = torch.empty((self.n_own, len(gi32)), dtype=Y_own.dtype, device=dev)
            if self.n_own:
                _lib.check(lib.fdx_gather_columns_dev(ctypes.c_void_p(Y_own.data_ptr()), y_code, self.n_own, G, G,
                                                      _lib.ptr_i32(gi32), len(gi32), ctypes.c_void_p(Y_sel.data_ptr()), st))
            Y_own, X = Y_sel, np.ascontiguousarray(X[:, self.gene_idx_])
            G = len(gi32)
        t0 = time.perf_counter()
        job, self._lev_job = getattr(self, "_lev_job", None), None
        self._x_dev_job = None
        if job is not None and job[0].shape == X.shape and np.array_equal(job[0], X):
            lev = job[1].result(keep_x=True)       # the job's device copy of X serves the native fit below (no second upload)
            self._x_dev_job = job[1]
        else:
            lev = compute_leverage_scores(X)
        t0 = self._tick("leverage", t0)
        self._mark("fit:leverage")
        bucket, weight = countsketch_tables(G, self.sketch_dim, lev, self.random_state)
        weight_y = weight_x = weight
        mode_y = mode_x = _lib.PRE_RAW
        if self.preprocess == "log_cpm":
            mode_y = mode_x = _lib.PRE_LOG_CPM
            if y_f64_math and Y_own.dtype == torch.float32:
                mode_y |= _lib.PRE_F64_MATH
        elif self.preprocess == "pearson":
            sums = np.zeros(G)
            if self.n_own:
                _lib.check(lib.fdx_column_sums_dev(ctypes.c_void_p(Y_own.data_ptr()), _lib.FDX_F32 if Y_own.dtype == torch.float32 else _lib.FDX_F64,
                                                   self.n_own, G, G, _lib.ptr_f64(sums), st))
            tsum = torch.from_numpy(sums).to(dev)
            self.comm.all_reduce_sum(tsum)
            mu_y = _lib.tensor_to_host(tsum) / self.n_total_spots + 1e-6
            mu_x = X.mean(axis=0) + 1e-6
            weight_y = weight / np.sqrt(mu_y + mu_y ** 2 / 100.0)
            weight_x = weight / np.sqrt(mu_x + mu_x ** 2 / 100.0)
        elif self.preprocess != "raw":
            raise ValueError(f"Unknown preprocess method: {self.preprocess}. Choose from 'log_cpm', 'pearson', or 'raw'.")
        t0 = self._tick("tables", t0)
        self._mark("fit:tables")
        n_own = self.n_own
        b32 = np.ascontiguousarray(bucket, dtype=np.int32)
        wy, wx = _lib.as_f64(weight_y), _lib.as_f64(weight_x)
        try:
            for attempt in range(2):
                # the whole rest of the fit in ONE native call (csrc/comm.cpp: fdx_shard_fit_dev) when libfdx owns the communicator;
                # a status (far walk / bound too small / ties) sends the plan through its remedy and, with the final graph, back here
                done = self._fit_native(Y_own, y_code, X, K, G, b32, wy, wx, mode_y, mode_x)
                if done is not False:
                    break
        finally:
            if self._x_dev_job is not None:
                self._x_dev_job.release_x()
                self._x_dev_job = None
        if done:
            return self.proportions_
        ld = ((n_own + 1 + 63) // 64) * 64          # H is read for the own rows only: its stride does not wait for the halo count
        H = torch.zeros((K, ld), dtype=torch.float64, device=dev)
        XtX = torch.empty((K, K), dtype=torch.float64, device=dev)
        XtX_h = np.empty((K, K))
        yty = ctypes.c_double(0.0)
        _lib.check(lib.fdx_prepare_dev(ctypes.c_void_p(Y_own.data_ptr()), _lib.FDX_F32 if Y_own.dtype == torch.float32 else _lib.FDX_F64,
                                       n_own, G, G, None, _lib.ptr_f64(X), K, _lib.ptr_i32(b32), _lib.ptr_f64(wy), _lib.ptr_f64(wx),
                                       int(self.sketch_dim), mode_y, mode_x, ctypes.c_void_p(H.data_ptr()), ld,
                                       ctypes.c_void_p(XtX.data_ptr()), _lib.ptr_f64(XtX_h), ctypes.byref(yty), st))
        t0 = self._tick("prepare", t0)
        self._mark("fit:prepare")
        return self._solve_shard(H, XtX, XtX_h, yty.value, K, ld)

    def _fit_native(self, Y_own, y_code, X, K, G, b32, wy, wx, mode_y, mode_x):
        """fdx_shard_fit_dev: sketch -> H, the plan's counts all-reduced beside it, lambda, the loop, objective, export - one
        call.  True: done (results set); False: a remedy was applied to the plan, call again; None: not applicable here (the
        process group is not libfdx's own communicator, more than 96 cell types, a rank without rows, sweeps being timed)."""
        import torch
        # every rank must take the same route (the native call all-reduces on libfdx's communicator, the stepwise flow on
        # torch's): the conditions are properties of the JOB - a rank without rows anywhere sends all ranks the stepwise way
        if (os.environ.get("FDX_NO_SHARD_FIT") or getattr(self, "time_sweeps", False) or K > 96 or self._local is None
                or bool(np.any(np.diff(self.bounds) <= 0))):
            return None
        native = self.native_comm()
        if native is None:
            return None
        lib = _lib.load()
        dev = Y_own.device
        st = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
        pending = getattr(self, "_pending_plan", None) is not None
        prm = _lib.ShardFitParams()
        prm.sketch_dim, prm.mode_y, prm.mode_x = int(self.sketch_dim), int(mode_y), int(mode_x)
        prm.lambda_auto = 1 if self.lambda_spatial == "auto" else 0
        prm.lambda_spatial = 0.0 if prm.lambda_auto else float(self.lambda_spatial)
        prm.rho_sparsity, prm.tol, prm.max_iter = float(self.rho_sparsity), float(self.tol), int(self.max_iter)
        prm.stop_on_ties = 1 if (pending and self.spatial_method == "knn" and self.knn_ties != "index") else 0
        prm.n_total_spots = int(self.n_total_spots)
        prm.nnz_total = -1 if pending else int(self.nnz_total)
        xj = getattr(self, "_x_dev_job", None)
        prm.X_dev = getattr(xj, "x_dev", None) if xj is not None else None
        if getattr(self.comm, "loopback", False) and pending:               # measurement: the job's total, known to the stand-in
            tot = getattr(self.comm, "totals", {}).get((3,))
            if tot is not None:
                prm.nnz_total = int(round(float(tot[0])))
        info = _lib.ShardFitInfo()
        rel = np.zeros(max(int(self.max_iter), 1))
        beta_t = torch.empty((self.n_own, K), dtype=torch.float64, device=dev)
        prop_t = torch.empty((self.n_own, K), dtype=torch.float64, device=dev)
        _lib.check(lib.fdx_shard_fit_dev(native, self._local.handle, ctypes.c_void_p(Y_own.data_ptr()), y_code, self.n_own, G, G,
                                         _lib.ptr_f64(X), K, _lib.ptr_i32(b32), _lib.ptr_f64(wy), _lib.ptr_f64(wx), ctypes.byref(prm),
                                         ctypes.c_void_p(beta_t.data_ptr()), ctypes.c_void_p(prop_t.data_ptr()), _lib.ptr_f64(rel),
                                         ctypes.byref(info), st))
        self._mark("fit:native call")
        if info.status != 0:
            # (a bound too small on some rank: that rank rebuilds, and its exact edge count enters the job's total through a fresh
            # all-reduce - every rank takes this branch, the status is the job's)
            self._finish_plan(totals=None if info.status == _lib.SHARD_OVERFLOW else
                              (float(info.nnz_total), float(info.knn_ties_total), 1.0 if info.status == _lib.SHARD_FAR else 0.0))
            return False
        if pending:
            self._pending_plan = None
            self.nnz_total, self.knn_ties_ = int(info.nnz_total), int(info.knn_ties_total)
            self.knn_ties_resolved_ = False
            self.n_halo, self.own_nnz_ = int(info.n_halo), int(info.own_nnz)
            if self.knn_ties_ and self.comm.rank == 0:
                import warnings
                warnings.warn(f"k-NN ties: {self.knn_ties_} of {self.n_total_spots} spots have their k-th and (k+1)-th nearest "
                              "neighbours at exactly the same distance (regular lattice?): the neighbour graph depends on how ties "
                              "are broken - here by spot index, in the reference by cKDTree's traversal order.  "
                              "spatial_method='grid' builds a tie-free graph on lattices.", UserWarning, stacklevel=3)
        self.beta_, self.proportions_ = beta_t, prop_t
        self.lambda_used_ = float(info.lambda_used)
        n_it = int(info.solve.n_iterations)
        self.info_ = {"converged": bool(info.solve.converged), "n_iterations": n_it, "final_change": float(info.solve.final_change),
                      "rel_changes": [float(v) for v in rel[:n_it]], "final_objective": float(info.solve.final_objective),
                      "objectives": []}
        self.sweep_loop_ms_ = float(info.solve.sweep_ms)
        self._mark_dump()
        return True

    def _fit_transform_csr(self, Y_own, X):
        """CSR shard (the own rows as a CUDA torch.sparse_csr tensor; reference core/deconv.py:181-188, utils/genes.py:52-83):
        the rows stay sparse in HBM, gene statistics are all-reduced, selected columns are filtered inside the sketch kernel."""
        import torch
        from .core.sketching import countsketch_tables
        from .utils import genes as _genes
        from .utils.genes import compute_leverage_scores
        lib = _lib.load()
        dev = Y_own.device
        st = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
        K, G_all = X.shape
        assert tuple(Y_own.shape) == (self.n_own, G_all)
        if self.preprocess not in ("log_cpm", "raw", "pearson"):
            raise ValueError(f"Unknown preprocess method: {self.preprocess}. Choose from 'log_cpm', 'pearson', or 'raw'.")
        csr = _lib.CsrOnDevice.from_torch(Y_own)
        try:
            self.gene_idx_ = np.arange(G_all, dtype=np.intp)
            colsum = None
            if G_all > self.n_hvg or self.preprocess == "pearson":
                sums = torch.zeros((3, G_all), dtype=torch.float64, device=dev)
                if self.n_own:
                    mean_r, var_r, col_r = csr.gene_moments(want_colsum=True)
                    sums[0], sums[1] = combine_moment_sums(mean_r, var_r, self.n_own, dev)
                    sums[2] = torch.from_numpy(col_r).to(dev)
                self.comm.all_reduce_sum(sums)
                colsum = _lib.tensor_to_host(sums[2])
                if G_all > self.n_hvg:
                    mean, var = moments_from_sums(_lib.tensor_to_host(sums[0]), _lib.tensor_to_host(sums[1]), self.n_total_spots)
                    hvg = _genes._hvg_from_moments(mean, var, self.n_hvg, 0.0125, 3.0, 0.5)
                    markers, _ = _genes.select_markers(X, n_markers=self.n_markers_per_type)
                    self.gene_idx_ = np.union1d(hvg, markers).astype(np.intp)
                    if len(self.gene_idx_) == 0:
                        raise ValueError("No genes selected. Increase n_hvg or n_markers_per_type.")
            Xs = np.ascontiguousarray(X[:, self.gene_idx_])
            G = Xs.shape[1]
            job, self._lev_job = getattr(self, "_lev_job", None), None
            lev = job[1].result() if (job is not None and job[0].shape == Xs.shape and np.array_equal(job[0], Xs)) else compute_leverage_scores(Xs)
            bucket, weight = countsketch_tables(G, self.sketch_dim, lev, self.random_state)
            weight_y = weight_x = weight
            mode_y = mode_x = _lib.PRE_RAW
            if self.preprocess == "log_cpm":
                mode_y, mode_x = _lib.PRE_LOG_CPM_SPARSE, _lib.PRE_LOG_CPM
            elif self.preprocess == "pearson":
                mu_y = colsum[self.gene_idx_] / self.n_total_spots + 1e-6
                mu_x = Xs.mean(axis=0) + 1e-6
                weight_y = weight / np.sqrt(mu_y + mu_y ** 2 / 100.0)
                weight_x = weight / np.sqrt(mu_x + mu_x ** 2 / 100.0)
            n_own = self.n_own
            ld = ((n_own + 1 + 63) // 64) * 64
            H = torch.zeros((K, ld), dtype=torch.float64, device=dev)
            XtX = torch.empty((K, K), dtype=torch.float64, device=dev)
            XtX_h = np.empty((K, K))
            yty = ctypes.c_double(0.0)
            gi32 = np.ascontiguousarray(self.gene_idx_, dtype=np.int32)
            b32 = np.ascontiguousarray(bucket, dtype=np.int32)
            wy, wx = _lib.as_f64(weight_y), _lib.as_f64(weight_x)
            _lib.check(lib.fdx_prepare_csr_dev(ctypes.byref(csr.view), _lib.ptr_i32(gi32), G, _lib.ptr_f64(Xs), K, _lib.ptr_i32(b32),
                                               _lib.ptr_f64(wy), _lib.ptr_f64(wx), int(self.sketch_dim), mode_y, mode_x,
                                               ctypes.c_void_p(H.data_ptr()), ld, ctypes.c_void_p(XtX.data_ptr()), _lib.ptr_f64(XtX_h),
                                               ctypes.byref(yty), st))
        finally:
            csr.free()
        return self._solve_shard(H, XtX, XtX_h, yty.value, K, ld)

    def _solve_shard(self, H, XtX, XtX_h, yty_part, K, ldh):
        """Everything after H / XtX exist for the own rows: global YtY, lambda, the sharded BCD solve, the objective,
        normalisation (core/deconv.py:358-398)."""
        import torch
        lib = _lib.load()
        dev = H.device
        st = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
        self._finish_plan()                      # a queued plan: its counts have long arrived behind the sketch
        n_own, n_total = self.n_own, self.n_own + self.n_halo
        ld = ((n_total + 1 + 63) // 64) * 64     # abundance planes: own rows, halo, the all-zero pad row
        t0 = time.perf_counter()
        # YtY only enters the final objective: its sum over the ranks rides in the objective's all-reduce at the end
        dmean = diag_mean(XtX_h)
        if self.lambda_spatial == "auto":                                     # core/spatial.py:181-190
            lam = 0.005 * dmean / max(self.nnz_total / max(self.n_total_spots, 1), 1.0)
        else:
            lam = float(self.lambda_spatial)
        rho_eff = float(self.rho_sparsity) * dmean                            # core/solver.py:359-360
        t0 = self._tick("scalars", t0)
        self._mark("solve:finish_plan+scalars")
        K_real = K
        if K > 64:
            # 65-96 cell types: the next instantiated sweep size with all-zero pad types (include/fdx.h: fdx_solver_padded_k)
            K = int(lib.fdx_solver_padded_k(K_real))           # above 96: K itself (LDS-resident / generic sweep, one launch per iteration)
            if K != K_real:
                Hp = torch.zeros((K, ldh), dtype=torch.float64, device=dev)
                Hp[:K_real] = H
                Gp = torch.zeros((K, K), dtype=torch.float64, device=dev)
                Gp[:K_real, :K_real] = XtX
                H, XtX = Hp, Gp
        backend = HipBackend(self._local, H, ldh, XtX, K, K_real=K_real)
        native = self.native_comm()
        if native is not None and not getattr(self, "time_sweeps", False):
            # the whole iteration loop in C++ on RCCL: boundary tiles first, halo traffic beside the interior sweep
            bufs = [torch.empty((K, ld), dtype=torch.float64, device=dev) for _ in range(2)]
            sinfo = _lib.SolveInfo()
            rel = np.zeros(max(int(self.max_iter), 1))
            which = ctypes.c_int32(0)
            _lib.check(lib.fdx_sharded_solve_padded_dev(native, self._local.handle, ctypes.c_void_p(H.data_ptr()), ldh,
                                                 ctypes.c_void_p(XtX.data_ptr()), K, K_real, float(lam), float(rho_eff), float(self.tol),
                                                 int(self.max_iter), ctypes.c_void_p(bufs[0].data_ptr()),
                                                 ctypes.c_void_p(bufs[1].data_ptr()), ld, ctypes.byref(sinfo), _lib.ptr_f64(rel),
                                                 ctypes.byref(which), st))
            beta = bufs[which.value]
            info = {"converged": bool(sinfo.converged), "n_iterations": int(sinfo.n_iterations),
                    "final_change": float(sinfo.final_change), "rel_changes": [float(v) for v in rel[:sinfo.n_iterations]]}
            self.sweep_loop_ms_ = float(sinfo.sweep_ms)
        else:
            solver = ShardedSolver(backend, self.comm, self._halo, K, ld, n_own, n_total, self.max_iter, self.tol)
            if getattr(self, "time_sweeps", False):
                solver.sweep_events = []
            beta, info = solver.run(lambda shape, dt: torch.zeros(shape, dtype=dt, device=dev), lam, rho_eff)
            if solver.sweep_events:
                torch.cuda.current_stream().synchronize()
                real = solver.sweep_events[:info["n_iterations"]]              # later launches are post-convergence no-ops
                self.sweep_ms_ = [a.elapsed_time(b) for a, b in real]
        t0 = self._tick("solve", t0)
        self._mark("solve:loop")
        # the export (reads the final abundances, writes two (n_own, K) matrices) runs on a side stream BESIDE the objective pass
        # (reads the same abundances), as in the single-GPU fit; the objective's read-back then waits for both
        self.beta_ = torch.empty((n_own, K_real), dtype=torch.float64, device=dev)
        self.proportions_ = torch.empty((n_own, K_real), dtype=torch.float64, device=dev)
        cur = torch.cuda.current_stream()
        if getattr(self, "_side", None) is None:
            self._side = torch.cuda.Stream(device=dev, priority=-1)   # a distinct priority: same-priority streams share hardware queues and can serialise
        side = self._side
        side.wait_stream(cur)
        _lib.check(lib.fdx_normalize_dev(ctypes.c_void_p(beta.data_ptr()), ld, n_own, K_real, ctypes.c_void_p(self.beta_.data_ptr()),
                                         ctypes.c_void_p(self.proportions_.data_ptr()), ctypes.c_void_p(side.cuda_stream)))
        part_h = np.concatenate([backend.objective_partials(beta), [yty_part]])
        cur.wait_stream(side)
        if self.comm.world > 1:
            part = torch.from_numpy(part_h).to(dev)
            self.comm.all_reduce_sum(part)                                    # objective partials and YtY in one collective
            c = part.cpu().numpy()
        else:
            c = part_h
        YtY = float(c[4])
        info["final_objective"] = float(0.5 * (YtY - 2.0 * c[0] + c[1]) + 0.5 * lam * c[2] + rho_eff * c[3])
        info["objectives"] = []
        self._tick("finish", t0)
        self.lambda_used_, self.info_ = lam, info
        self._mark("solve:finish")
        self._mark_dump()
        return self.proportions_


def bench_main(a, rank, world, local_rank):
    """bench.py --gpus N (N > 1): ONE job sharded over the N ranks.  --scaling strong (default): the --spots job itself
    (BASELINE.json configs[3]: 1M spots over N GPUs); --scaling weak: N x --spots spots, --spots per rank.  Coordinates are
    replicated, each rank holds only its own rows of Y."""
    import json
    import time
    import torch
    import torch.distributed as dist
    import bench
    dev = torch.device("cuda", local_rank)
    weak = getattr(a, "scaling", "strong") == "weak"
    n, G, K, d = (a.spots * world if weak else a.spots), a.genes, a.types, a.sketch_dim
    g = torch.Generator(device=dev)
    g.manual_seed(12345)                                        # identical coordinates and signatures on every rank
    coords = torch.rand(n, 2, generator=g, device=dev, dtype=torch.float64) * float(np.sqrt(n))
    X = torch.randn(K, G, generator=g, device=dev, dtype=torch.float64)
    model = ShardedFlashDeconv(sketch_dim=d, preprocess="raw", n_hvg=G)
    own = model.plan(coords)
    g2 = torch.Generator(device=dev)
    g2.manual_seed(1000 + rank)                                 # this rank's rows of Y (gaussian/raw family)
    Y = torch.empty((model.n_own, G), device=dev, dtype=torch.float32)
    for r0 in range(0, model.n_own, 1 << 17):
        r1 = min(model.n_own, r0 + (1 << 17))
        B = torch.rand(r1 - r0, K, generator=g2, device=dev, dtype=torch.float64)
        B /= B.sum(dim=1, keepdim=True)
        Y[r0:r1] = (B @ X + 0.1 * torch.randn(r1 - r0, G, generator=g2, device=dev, dtype=torch.float64)).to(torch.float32)
    Xh = X.cpu().numpy()

    # A step that does not come back (a collective some rank never joined) must not sit until the launcher's own timeout: every
    # step re-arms a deadline after which this rank dumps its stack and exits non-zero (FDX_BENCH_STEP_DEADLINE seconds, default 240)
    import faulthandler
    deadline = float(os.environ.get("FDX_BENCH_STEP_DEADLINE", "240"))

    def step():
        faulthandler.dump_traceback_later(deadline, exit=True)
        model.plan(coords, Xh)
        model.fit_transform(Y, Xh)
        faulthandler.cancel_dump_traceback_later()

    for _ in range(a.warmup):
        step()
    dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(a.steps):
        step()
    torch.cuda.synchronize()
    dist.barrier()
    dt = torch.tensor([time.perf_counter() - t0], dtype=torch.float64, device=dev)
    dist.all_reduce(dt, op=dist.ReduceOp.MAX)
    dt = float(dt.item())
    n_it, conv = model.info_["n_iterations"], model.info_["converged"]
    # roofline of the sweep on rank 0's shard: hipEvents (torch.cuda.Event on the stream the kernels run on) around every
    # sweep of one extra, untimed fit; algorithmic bytes as in the single-GPU line
    model.time_sweeps = True
    step()
    model.time_sweeps = False
    sweep_ms = float(np.mean(model.sweep_ms_)) if getattr(model, "sweep_ms_", None) else None
    roof = None
    if sweep_ms:
        own_nnz = int(model.own_nnz_)
        alg = 3 * model.n_own * K * 8 + (own_nnz + model.n_own + 1) * 4
        ach = alg / (sweep_ms * 1e-3) / 1e9
        roof = {"bound": "hbm", "kernel": bench.sweep_kernel_name(K), "achieved": round(ach, 1),
                "peak": bench.HBM_PEAK_GBS, "unit": "GB/s", "frac": round(ach / bench.HBM_PEAK_GBS, 4), "traffic": None,
                "alg_bytes_per_launch": int(alg), "ms_per_launch": round(sweep_ms, 4), "rank": 0}
    reports = [None] * world
    rep = model.comm_report()
    # sanity of the run itself, per rank: the iteration count every rank stopped at (must agree), finite proportions whose rows sum to 1
    P = model.proportions_
    rep["n_iterations"] = int(n_it)
    rep["rows_sum_to_one"] = bool(P.numel() == 0 or (torch.isfinite(P).all() and ((P.sum(dim=1) - 1.0).abs().max() < 1e-9)))
    dist.all_gather_object(reports, rep)
    dist.destroy_process_group()
    ctypes.CDLL(None).fflush(None)          # RCCL's banner sits in the C stdio buffer: get it out BEFORE the result line
    if rank == 0:
        print(json.dumps({
            "metric": bench.METRIC if not weak else f"spots/sec to convergence ({world} x 1M x 2000 x 30, weak scaling)",
            "value": n * a.steps / dt, "unit": "spots/s", "spots_total": n, "spots_per_gpu": n // world,
            "n_gpus": world, "steps": a.steps, "warmup": a.warmup, "ms_per_step": dt / a.steps * 1e3,
            "higher_is_better": True, "scaling": "weak" if weak else "strong", "vs_baseline": None, "dtype": "f64",
            "data": "synthetic",
            "config": {"workload": f"{n} spots x {G} genes x {K} types, sketch_dim {d}, k_neighbors 6, gaussian/raw family, "
                                   f"Y float32 in HBM, spots sharded over {world} GPUs (Morton ranges, RCCL halo exchange)",
                       "n_iterations": n_it, "converged": conv},
            "roofline": roof, "cpu_baseline": None,
            # what actually ran on every rank: the ranks RCCL itself reports for libfdx's communicator (None: not created), the
            # device, the route of the plan, native or Python iteration loop
            "rccl_ranks": reports[0]["rccl_ranks"], "loop": sorted({r["loop"] for r in reports}),
            "native_comm_error": next((r["native_comm_error"] for r in reports if r["native_comm_error"]), None),
            "ranks_agree": len({r["n_iterations"] for r in reports}) == 1 and all(r["rows_sum_to_one"] for r in reports),
            "ranks": reports}), flush=True)
