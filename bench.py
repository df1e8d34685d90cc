#!/usr/bin/env python3
"""Benchmark of the hot path: spots/sec to convergence of FlashDeconv.fit_transform on synthetic N x G x K data.

    python bench.py --gpus N --steps K --warmup W [--scaling strong|weak]

One rank per GPU.  Under torch.distributed.run (RANK / LOCAL_RANK / WORLD_SIZE in the environment) this process is one of
the N ranks; started from a bare shell with --gpus N > 1 it spawns the N ranks itself (fresh child processes, before
anything has touched a GPU) and relays rank 0's line.

A "step" is one complete fit (graph build -> preprocess + CountSketch -> H -> BCD solve to the reference's stopping rule
-> proportions) from inputs resident in HBM to proportions_ resident in HBM.  Workload = BASELINE.json configs[2]:
1M spots x 2000 genes x 30 types, sketch_dim 512, k_neighbors 6, Gaussian/raw family (SURVEY.md §8d family A), Y stored
float32, all arithmetic float64.  --scaling strong (default, BASELINE.json configs[3]): the same 1M-spot job sharded over
the N ranks.  --scaling weak: N x 1M spots, 1M per rank.

Rank 0 prints ONE JSON line.  Besides the contract keys it carries
  roofline      - dominant kernel of the step vs the HBM roofline (algorithmic bytes / hipEvent-measured duration)
  cpu_baseline  - the oracle (CPU restatement of the reference, numpy/scipy + C/OpenMP BCD) timed on this box's host
                  cores on the headline's own rows (--cpu-sample, default all 1M: ~30 s of CPU; N=1 only)
  count_like    - the same measurement on the count-like / log_cpm family, which runs all 100 iterations
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

METRIC = "spots/sec to convergence (1M x 2000 x 30)"
HBM_PEAK_GBS = 8000.0   # MI355X HBM3E spec peak (MI355X_MICROARCH.md); ~6300 GB/s is the measured copy ceiling


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--spots", type=int, default=1_000_000)
    ap.add_argument("--genes", type=int, default=2000)
    ap.add_argument("--types", type=int, default=30)
    ap.add_argument("--sketch-dim", type=int, default=512)
    ap.add_argument("--family", choices=["gaussian", "counts", "both", "sparse", "lattice", "all"], default="all")
    ap.add_argument("--sparse-genes", type=int, default=20000, help="columns of the CSR family's matrix (HVG picks ~--genes of them)")
    ap.add_argument("--scaling", choices=["strong", "weak"], default="strong",
                    help="N > 1: strong = the --spots job sharded over N ranks (configs[3]); weak = N x --spots")
    ap.add_argument("--config", type=int, default=3, choices=[3, 5],
                    help="3: BASELINE configs[2]/[3] (1M x 2000 x 30, d 512); 5: one rank's shard of configs[4] "
                         "(1.25M x 5000 x 50, d 1024, lambda auto) on one GPU, gaussian/raw family only")
    ap.add_argument("--virtual-ranks", type=int, default=0,
                    help="the WHOLE configs[3] job (1M x 2000 x 30) or, with --config 5, configs[4] (10M x 5000 x 50; --spots overrides) "
                         "with this many virtual ranks on one GPU (tools/virtual_ranks.py): per-rank critical path, each rank timed "
                         "alone, the unsharded T1 and the projected speed-up (no RCCL wire time)")
    ap.add_argument("--vr-family", choices=["gaussian", "counts"], default="gaussian",
                    help="--virtual-ranks: gaussian = the metric's job (raw, converges in 7 sweeps); counts = the count-like family's cost "
                         "profile (log_cpm sketch, all 100 sweeps)")
    ap.add_argument("--rendezvous-timeout", type=float, default=None,
                    help="--gpus N > 1 started from a bare shell: seconds after which a job whose ranks have produced no result "
                         "is ended (default 900)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-live-pmc", action="store_true",
                    help="do not spawn the two rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE) that measure roofline.traffic in this run; "
                         "the committed profile's figure is reported instead")
    ap.add_argument("--no-host-arrays", action="store_true",
                    help="skip the host_arrays record (numpy / scipy in, numpy out: the literal drop-in, PCIe included)")
    ap.add_argument("--cpu-sample", type=int, default=1_000_000,
                    help="rows of the headline matrix the CPU baseline is timed on (default: all of configs[2]: ~30 s of CPU)")
    return ap.parse_args()


# ------------------------------------------------------------------------------------------------ synthetic inputs
def gen_gaussian(torch, n, G, K, device, seed):
    """Family A: X ~ N(0,1); B row-normalised U(0,1); Y = B X + 0.1 N(0,1); uniform coords."""
    g = torch.Generator(device=device)
    g.manual_seed(seed)
    X = torch.randn(K, G, generator=g, device=device, dtype=torch.float64)
    Y = torch.empty((n, G), device=device, dtype=torch.float32)
    step = 1 << 17
    for r0 in range(0, n, step):
        r1 = min(n, r0 + step)
        B = torch.rand(r1 - r0, K, generator=g, device=device, dtype=torch.float64)
        B /= B.sum(dim=1, keepdim=True)
        Y[r0:r1] = (B @ X + 0.1 * torch.randn(r1 - r0, G, generator=g, device=device, dtype=torch.float64)).to(torch.float32)
    coords = torch.rand(n, 2, generator=g, device=device, dtype=torch.float64) * float(np.sqrt(n))
    return Y, X.cpu().numpy(), coords


def gen_counts(torch, n, G, K, device, seed):
    """Family B: count-like data shaped like the reference's integration generator (tests/test_integration.py:10-84):
    log-normal signatures with 20 x5 markers per type, jittered grid, smooth proportions, gamma depth, Poisson counts."""
    g = torch.Generator(device=device)
    g.manual_seed(seed)
    X = torch.exp(torch.randn(K, G, generator=g, device=device, dtype=torch.float64) * 0.5 + 1)
    for k in range(K):
        idx = torch.randperm(G, generator=g, device=device)[:20]
        X[k, idx] *= 5
    side = int(np.ceil(np.sqrt(n)))
    ii = torch.arange(n, device=device)
    coords = torch.stack([(ii % side).double(), (ii // side).double()], dim=1)
    coords += torch.randn(n, 2, generator=g, device=device, dtype=torch.float64) * 0.1
    centres = torch.rand(K, 2, generator=g, device=device, dtype=torch.float64) * side
    Y = torch.empty((n, G), device=device, dtype=torch.float32)
    step = 1 << 16
    for r0 in range(0, n, step):
        r1 = min(n, r0 + step)
        dist = torch.cdist(coords[r0:r1], centres)
        B = torch.exp(-dist / (side / 2))
        B /= B.sum(dim=1, keepdim=True)
        depth = torch.distributions.Gamma(torch.tensor(5.0, device=device, dtype=torch.float64),
                                          torch.tensor(1.0 / 1000.0, device=device, dtype=torch.float64)).sample((r1 - r0,))
        lam = (B @ X) * depth[:, None]
        lam *= 1 + 0.1 * torch.rand(r1 - r0, G, generator=g, device=device, dtype=torch.float64)
        Y[r0:r1] = torch.poisson(lam, generator=g).to(torch.float32)
    return Y, X.cpu().numpy(), coords


def gen_sparse(torch, n, G_all, K, device, seed, depth=1500.0):
    """Family C: a CSR count matrix over the full transcriptome (G_all columns, ~7 % stored, ~`depth` counts per spot),
    the shape real Visium-HD / Stereo-seq input has; gene selection (HVG + markers) is active."""
    g = torch.Generator(device=device)
    g.manual_seed(seed)
    X = torch.exp(torch.randn(K, G_all, generator=g, device=device, dtype=torch.float64) * 1.2 - 1.0)
    for k in range(K):
        idx = torch.randperm(G_all, generator=g, device=device)[:40]
        X[k, idx] *= 8
    side = int(np.ceil(np.sqrt(n)))
    ii = torch.arange(n, device=device)
    coords = torch.stack([(ii % side).double(), (ii // side).double()], dim=1)
    coords += torch.randn(n, 2, generator=g, device=device, dtype=torch.float64) * 0.1
    centres = torch.rand(K, 2, generator=g, device=device, dtype=torch.float64) * side
    crow, col, val, nnz = [torch.zeros(1, dtype=torch.int64, device=device)], [], [], 0
    step = 1 << 15
    import warnings
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        for r0 in range(0, n, step):
            r1 = min(n, r0 + step)
            B = torch.exp(-torch.cdist(coords[r0:r1], centres) / (side / 2))
            B /= B.sum(dim=1, keepdim=True)
            lam = B @ X
            lam *= depth / lam.sum(dim=1, keepdim=True)
            Yc = torch.poisson(lam, generator=g).to(torch.float32).to_sparse_csr()
            crow.append(Yc.crow_indices()[1:] + nnz)
            col.append(Yc.col_indices().to(torch.int32))
            val.append(Yc.values())
            nnz += int(Yc.values().numel())
        Y = torch.sparse_csr_tensor(torch.cat(crow), torch.cat(col), torch.cat(val), size=(n, G_all), device=device)
    return Y, X.cpu().numpy(), coords


# ------------------------------------------------------------------------------------------------ measurement
def alg_bytes(n, G, K, s_y, nnz, n_slices_width_rows, T):
    """Algorithmic HBM bytes: one-off G*s_Y per spot; per sweep 3*N*K*8 + graph; finish 3*N*K*8."""
    sketch = n * G * s_y                           # read Y once
    sweep = 3 * n * K * 8 + (nnz + n + 1) * 4      # read H, read beta_in, write beta_out + CSR structure
    return sketch, sweep


TRAFFIC_PROFILE = "profiles/r06_traffic.json"
TRAFFIC_PROFILE_C5 = "profiles/r04_config5_traffic.json"     # the configs[4] shard (bench.py --config 5)


def pmc_traffic(kernel, shape):
    """HBM bytes per launch of `kernel` from the committed PMC pass (TRAFFIC_PROFILE: rocprofv3 --pmc FETCH_SIZE /
    WRITE_SIZE of this same command on an earlier run of the same build, gfx950 correction applied) - measured then, not
    in this run; None for other workloads or a kernel the profile does not hold."""
    prof = {(1_000_000, 2000, 30, 512): TRAFFIC_PROFILE, (1_250_000, 5000, 50, 1024): TRAFFIC_PROFILE_C5}.get(tuple(shape))
    if prof is None:
        return None
    try:
        with open(os.path.join(ROOT, prof)) as f:
            kernels = json.load(f)["kernels"]
        k = kernels.get(kernel) or next((v for name, v in kernels.items() if name.startswith(kernel)), None)
        return int(k["hbm_bytes_corrected"]) if k else None
    except (OSError, ValueError, KeyError):
        return None


def live_pmc_traffic(a, timeout_s=240):
    """HBM bytes per launch of every libfdx kernel of THIS build on THIS box: two child runs of this script under
    `rocprofv3 --pmc FETCH_SIZE|WRITE_SIZE --kernel-trace` (counters in passes of their own, as the pool requires; the program itself
    after `--`), one fit per family; bytes = (2 x FETCH_SIZE + WRITE_SIZE) x 1024 (gfx950: FETCH_SIZE counts half of a wide
    coalesced read - MI355X_MICROARCH.md, HBM section), mean over the launches that did work.  {} when rocprofv3 is not there
    or a pass fails (the committed profile's figure is reported then)."""
    import csv
    import glob
    import shutil
    import subprocess
    import tempfile
    from collections import defaultdict
    exe = shutil.which("rocprofv3")
    # not from inside a profiled run: the outer profiler's preloaded library would initialise the GPU in the child launcher, whose
    # exec of the program the box then refuses (tools/profile_round.sh passes --no-live-pmc as well)
    profiled = any(k.startswith(("ROCPROF", "ROCP_")) for k in os.environ) or "rocprof" in os.environ.get("LD_PRELOAD", "")
    if exe is None or profiled:
        return {}
    vals = {}
    work = tempfile.mkdtemp(prefix="fdx_pmc_", dir="/tmp")
    try:
        for ctr in ("FETCH_SIZE", "WRITE_SIZE"):
            out = os.path.join(work, ctr)
            cmd = [exe, "--pmc", ctr, "--kernel-trace", "--output-format", "csv", "-d", out, "--", sys.executable, os.path.abspath(__file__),
                   "--family", a.family, "--steps", "1", "--warmup", "0", "--no-cpu-baseline", "--no-host-arrays", "--no-live-pmc",
                   "--spots", str(a.spots), "--genes", str(a.genes), "--types", str(a.types), "--sketch-dim", str(a.sketch_dim),
                   "--sparse-genes", str(a.sparse_genes)]
            try:
                subprocess.run(cmd, cwd="/tmp", env=dict(os.environ, TMPDIR="/tmp"), stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL,
                               timeout=timeout_s, check=True)
            except (subprocess.SubprocessError, OSError):
                return {}
            acc = defaultdict(list)
            for f in glob.glob(os.path.join(out, "**", "*counter_collection.csv"), recursive=True):
                with open(f) as fh:
                    for r in csv.DictReader(fh):
                        if r.get("Counter_Name") == ctr and "fdx::" in r.get("Kernel_Name", ""):
                            acc[r["Kernel_Name"].replace("void ", "").split("(")[0]].append(float(r["Counter_Value"]))
            vals[ctr] = acc
    finally:
        shutil.rmtree(work, ignore_errors=True)
    res = {}
    for k, fv in vals.get("FETCH_SIZE", {}).items():
        wv = vals.get("WRITE_SIZE", {}).get(k, [])
        big = max(fv) if fv else 0.0
        keep = [i for i, v in enumerate(fv) if v > 0.1 * big] if big else list(range(len(fv)))    # no-op launches (sweeps behind convergence) out
        f_mean = sum(fv[i] for i in keep) / max(len(keep), 1)
        w_keep = [wv[i] for i in keep if i < len(wv)]
        res[k] = int((2.0 * f_mean + sum(w_keep) / max(len(w_keep), 1)) * 1024)
    return res


def apply_live_traffic(roof, live):
    """roofline.traffic of a record from this run's counter passes when they hold its kernel."""
    if not roof or not live:
        return
    k = roof.get("kernel", "")
    hit = live.get(k) or next((v for name, v in live.items() if name.startswith(k) or k.startswith(name)), None)
    if hit:
        roof["traffic"] = int(hit)
        roof["traffic_source"] = "this run: rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes spawned by bench.py, (2 x FETCH + WRITE) x 1024 per launch"


def sweep_chunk(K):
    """Cell types per chunk of the tiled sweep: mirror of sweep_chunk() in csrc/bcd_sweep_inst.cpp (second template argument
    of bcd_sweep_tiled_kernel; tests/test_host.py checks the two against each other)."""
    if K < 8:
        return K
    for bound, kc in ((29, 8), (33, 7), (37, 6), (41, 5), (45, 4), (49, 3), (51, 2), (64, 8), (72, 6), (88, 4), (96, 2)):
        if K <= bound:
            return kc
    return 8


def sweep_kernel_name(K):
    return "fdx::bcd_sweep_tiled_kernel<%d, %d, false, true, false>" % (K, sweep_chunk(K))    # <K, chunk, objective variant, quadratic term inside, constant start vector>


def sketch_kernel_name(mode, K, d=512, dtype="float"):
    """Kernel that serves the sketch -> H stage of the bench shapes (csrc/tile_kernels.cpp: tile_cfg): template arguments
    <input type, preprocess, consumer waves, loader waves, groups per wave, type tiles, A operands from L2, log1p class
    (2 = float32-class for float32 rows), weights by gene in the stage buffers (wide raw), flat schedule>."""
    cfg = os.environ.get("FDX_TILE_CFG")
    nwc, nwl, jw = {"12": (12, 4, 11), "16": (16, 0, 8), "8": (8, 2, 16)}.get(cfg, (12, 4, 11) if mode == 0 else (16, 0, 8))
    logv = 0 if (mode == 0 or os.environ.get("FDX_TILE_LOGV") == "0") else 2
    tt = -(-K // 16)
    avl2 = mode != 0 and nwc == 16 and (logv == 0 or bool(os.environ.get("FDX_TILE_AVL2")))
    if nwc != 16 and nwc != 12:
        logv = 0
    wg = False
    if K > 32 or d > 4 * nwc * jw:                 # wide form (raw: weights by gene in a ring of three stage buffers)
        (nwc, nwl, jw), tt, avl2 = ((12, 4, 22) if mode == 0 else (8, 0, 32)), 4, True
        wg = mode == 0 and not os.environ.get("FDX_TILE_NO_WG")
    return "fdx::tile_sketch_kernel<%s, %d, %d, %d, %d, %d, %s, %d, %s, %s>" % (
        dtype, mode, nwc, nwl, jw, tt, "true" if avl2 else "false", logv, "true" if wg else "false",
        "true" if (wg and os.environ.get("FDX_TILE_FLAT")) else "false")


def aux_steps(a):
    """Timed steps of the families beside the headline (count-like, CSR, lattice): the driver's --steps, at most 10; they get
    the same --warmup as the headline (a family's first two or three fits in a process still pay for pooled buffers changing
    hands: with two warm-up fits and two timed ones the driver's figure was the tail of the warm-up)."""
    return max(1, min(a.steps, 10))


def run_family(torch, model_kw, Y, X, coords, steps, warmup, barrier):
    from flashdeconv_amd import FlashDeconv
    model = FlashDeconv(**model_kw)
    for _ in range(warmup):
        model.fit(Y, X, coords, output="torch")
    barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    stage = {}
    t_prev, walls = t0, []
    for _ in range(steps):
        model.fit(Y, X, coords, output="torch")
        for k, v in model.timings_.items():
            stage[k] = stage.get(k, 0.0) + v
        t_now = time.perf_counter()             # (no synchronisation: fit returns with its results complete)
        walls.append((t_now - t_prev) * 1e3)
        t_prev = t_now
    torch.cuda.synchronize()
    barrier()
    dt = time.perf_counter() - t0
    for k in stage:
        stage[k] /= steps
    stage["step_ms_min_max"] = [round(min(walls), 3), round(max(walls), 3)]
    # cold fit: the same call with the library's content-keyed caches (sketch plans, tile schedules) bypassed - what a user who
    # calls fit_transform ONCE pays (the GPU context itself is warm); outside the timed region
    from flashdeconv_amd import _lib as _fdx_lib
    os.environ["FDX_NO_PLAN_CACHE"] = "1"
    _fdx_lib.env_reload()                          # (the library caches its switches: no fit is running here)
    try:
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        FlashDeconv(**model_kw).fit(Y, X, coords, output="torch")
        torch.cuda.synchronize()
        stage["cold_ms"] = (time.perf_counter() - t1) * 1e3
    finally:
        del os.environ["FDX_NO_PLAN_CACHE"]
        _fdx_lib.env_reload()
    return model, dt, stage


def stage_record(stage, ms_per_step):
    """Stage times of one step as they tile its wall time: host_pre (Python before the first kernel) + span (device, hipEvents:
    prologue + sketch + gram + solve + finish) + host_post; `unaccounted_ms` = wall - that sum."""
    out = {k: round(v, 3) for k, v in stage.items() if k not in ("cold_ms", "step_ms_min_max")}
    out["unaccounted_ms"] = round(ms_per_step - (stage["host_pre_ms"] + stage["span_ms"] + stage["host_post_ms"]), 3)
    return out


def host_cpu_budget():
    """CPUs this process may keep busy: its affinity mask, cut to the control group's CPU bandwidth quota (a GPU box of this pool
    shows 256 CPUs and grants 16: threads beyond the quota only get the process throttled)."""
    cpus = float(len(os.sched_getaffinity(0)))
    try:
        with open("/sys/fs/cgroup/cpu.max") as f:
            q, period = f.read().split()[:2]
        if q != "max" and float(period) > 0:
            cpus = min(cpus, float(q) / float(period))
    except (OSError, ValueError):
        try:
            with open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us") as f:
                q = float(f.read())
            with open("/sys/fs/cgroup/cpu/cpu.cfs_period_us") as f:
                period = float(f.read())
            if q > 0 and period > 0:
                cpus = min(cpus, q / period)
        except (OSError, ValueError):
            pass
    return max(1, int(round(cpus)))


def cpu_baseline(Yh32, X, coords_h, d):
    """Oracle (CPU restatement of the reference path, every stage as the reference computes it: dense HVG statistics, the
    fancy-index copy of Y, scipy's dense @ CSR sketch, cKDTree, BLAS Gram, then the C/OpenMP BCD sweep in place of numba's
    prange) on the HEADLINE's own rows: the float32 matrix the GPU steps read, as float64 (core/deconv.py:229)."""
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import fdx_oracle as orc
    n_cpu, G = Yh32.shape
    K = X.shape[0]
    Y = Yh32.astype(np.float64)
    cores = host_cpu_budget()
    try:                                       # the C sweep's OpenMP team: as many threads as the process may actually run
        import ctypes
        ctypes.CDLL("libgomp.so.1").omp_set_num_threads(int(cores))
    except OSError:
        pass
    t0 = time.perf_counter()
    out = orc.fit(Y, X, coords_h, sketch_dim=d, preprocess_method="raw", n_hvg=G, graph="kdtree", engine="c")
    dt = time.perf_counter() - t0
    return {"value": n_cpu / dt, "unit": "spots/s", "cores": cores, "kind": "port",
            "sample": f"{n_cpu} spots x {G} genes x {K} types (the headline's rows), gaussian/raw float64, {out['info']['n_iterations']} iterations, "
                      f"{dt:.1f} s wall (numpy/scipy stages single-threaded as in the reference, C/OpenMP BCD sweep on {cores} threads)"}


def host_arrays_record(torch, n, G, K, d, sparse_genes, device):
    """The literal drop-in (core/deconv.py:237-243, README.md:118-129: users hand numpy / scipy): host arrays in, numpy
    proportions_ / beta_ out, the same 1M-spot jobs.  Never the headline `value` - PCIe dominates: the yardstick is the time ONE
    pinned copy of the same bytes takes on this box's link (fdx_pinned_copy_rate); `ratio` = fit wall / that."""
    import ctypes
    from scipy import sparse as sp
    from flashdeconv_amd import FlashDeconv, _lib
    lib = _lib.load()
    rate = {}
    for name, to_dev in (("pinned_h2d_GBps", 1), ("pinned_d2h_GBps", 0)):
        g = ctypes.c_double(0.0)
        _lib.check(lib.fdx_pinned_copy_rate(1 << 30, to_dev, ctypes.byref(g)))
        rate[name] = round(g.value, 2)
    out = dict(rate)
    out["note"] = ("wall of FlashDeconv.fit from host arrays to host arrays (numpy in, numpy out), 2 timed fits after 1; ideal_ms = bytes in / "
                   "pinned H2D rate + bytes out / pinned D2H rate of this box; threaded staging through pinned buffers "
                   "(csrc/host_transfer.cpp), integer counts narrowed to float32 on the way")
    cases = {}

    def run(label, Yh, X, ch, kw, in_bytes):
        m = FlashDeconv(**kw)
        m.fit(Yh, X, ch)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        reps = 2
        for _ in range(reps):
            m.fit(Yh, X, ch)
        ms = (time.perf_counter() - t0) / reps * 1e3
        out_bytes = 2 * n * K * 8
        ideal = (in_bytes / (rate["pinned_h2d_GBps"] * 1e9) + out_bytes / (rate["pinned_d2h_GBps"] * 1e9)) * 1e3
        cases[label] = {"ms_per_fit": round(ms, 2), "in_GB": round(in_bytes / 1e9, 3), "out_GB": round(out_bytes / 1e9, 3),
                        "ideal_ms": round(ideal, 2), "ratio": round(ms / ideal, 3), "effective_in_GBps": round(in_bytes / (ms * 1e-3) / 1e9, 2),
                        "device_span_ms": round(m.timings_["span_ms"], 3), "n_iterations": m.info_["n_iterations"]}

    Y, X, coords = gen_gaussian(torch, n, G, K, device, seed=0)
    ch = _lib.tensor_to_host(coords)
    Yh = _lib.tensor_to_host(Y)
    del Y
    torch.cuda.empty_cache()
    run("dense_float32_raw", Yh, X, ch, dict(sketch_dim=d, preprocess="raw", n_hvg=G), Yh.nbytes)
    Yh64 = Yh.astype(np.float64)
    del Yh
    run("dense_float64_raw", Yh64, X, ch, dict(sketch_dim=d, preprocess="raw", n_hvg=G), Yh64.nbytes)
    del Yh64
    Y, X, coords = gen_counts(torch, n, G, K, device, seed=0)
    ch = _lib.tensor_to_host(coords)
    Yi = _lib.tensor_to_host(Y.to(torch.int32))
    del Y
    torch.cuda.empty_cache()
    run("dense_int32_log_cpm", Yi, X, ch, dict(sketch_dim=d, preprocess="log_cpm", n_hvg=G, max_iter=20), Yi.nbytes)
    del Yi
    Y, X, coords = gen_sparse(torch, n, sparse_genes, K, device, seed=0)
    ch = _lib.tensor_to_host(coords)
    Ys = sp.csr_matrix((_lib.tensor_to_host(Y.values()), _lib.tensor_to_host(Y.col_indices()), _lib.tensor_to_host(Y.crow_indices())),
                       shape=(n, sparse_genes))
    del Y
    torch.cuda.empty_cache()
    run("scipy_csr_float32_log_cpm", Ys, X, ch, dict(sketch_dim=d, preprocess="log_cpm", n_hvg=G, max_iter=20),
        Ys.data.nbytes + Ys.indices.nbytes + Ys.indptr.nbytes)
    out["cases"] = cases
    return out


def spawn_ranks(n_ranks, rendezvous_timeout_s=None):
    """--gpus N > 1 from a bare shell: start the N ranks as fresh child processes (this process never initialises a GPU), poll
    them all, relay rank 0's output and exit with the worst return code.  The first rank that exits non-zero ends the job: the
    others are terminated (they would sit in a collective until the driver's timeout) and the parent returns that code within
    seconds.  Every rank's stderr goes to a file named by rank (gpurun_out/bench_rank<r>.err, else the temp dir); a rank that
    produces no result within --rendezvous-timeout seconds of the start (default 900; FDX_BENCH_RDZV_TIMEOUT) ends the job too."""
    import socket
    import subprocess
    import tempfile
    with socket.socket() as sk:                       # a free rendezvous port
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    timeout_s = float(rendezvous_timeout_s or os.environ.get("FDX_BENCH_RDZV_TIMEOUT", "900"))
    logdir = os.path.join(ROOT, "gpurun_out")
    if not os.path.isdir(logdir) or not os.access(logdir, os.W_OK):
        logdir = tempfile.gettempdir()
    procs, errs = [], []
    out0 = tempfile.TemporaryFile()
    for r in range(n_ranks):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n_ranks), MASTER_ADDR="127.0.0.1",
                   MASTER_PORT=str(port))
        env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        err = open(os.path.join(logdir, f"bench_rank{r}.err"), "wb")
        errs.append(err)
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env=env,
                                      stdout=out0 if r == 0 else subprocess.DEVNULL, stderr=err))
    t0 = time.monotonic()
    rc, failed = 0, None
    while True:
        codes = [p.poll() for p in procs]
        bad = [(r, c) for r, c in enumerate(codes) if c not in (None, 0)]
        if bad:
            failed, rc = bad[0]
            break
        if all(c == 0 for c in codes):
            break
        if time.monotonic() - t0 > timeout_s:
            failed, rc = next(r for r, c in enumerate(codes) if c is None), 124
            break
        time.sleep(0.05)
    if failed is not None:
        for p in procs:                               # fresh children of this process only: their exact PIDs
            if p.poll() is None:
                p.terminate()
        for p in procs:
            try:
                p.wait(timeout=10)
            except subprocess.TimeoutExpired:
                p.kill()
                p.wait()
    for e in errs:
        e.close()
    out0.seek(0)
    sys.stdout.write(out0.read().decode(errors="replace"))
    sys.stdout.flush()
    if failed is not None:
        why = "produced no result in time" if rc == 124 else f"exited with code {rc}"
        tail = b""
        try:
            with open(os.path.join(logdir, f"bench_rank{failed}.err"), "rb") as f:
                tail = f.read()[-2000:]
        except OSError:
            pass
        print(f"bench.py: rank {failed} of {n_ranks} {why}; the other ranks were terminated.  Per-rank stderr: "
              f"{os.path.join(logdir, 'bench_rank<r>.err')}\n--- rank {failed} stderr (tail) ---\n{tail.decode(errors='replace')}",
              file=sys.stderr)
    else:                                             # relay rank 0's stderr (RCCL banners, warnings) as before
        try:
            with open(os.path.join(logdir, "bench_rank0.err"), "rb") as f:
                sys.stderr.write(f.read().decode(errors="replace"))
        except OSError:
            pass
    sys.exit(rc)


def main():
    a = parse()
    if a.virtual_ranks > 0:
        # The whole configs[3] (1M x 2000 x 30, d 512) or configs[4] (10M x 5000 x 50, d 1024, --config 5) job over virtual ranks
        # on ONE GPU: every rank's share of the sharded driver timed ALONE (plan = k-NN lists of own rows + band + symmetrise,
        # localize, prepare = sketch -> H, the native iteration loop over a loopback transport, finish), then the same job
        # unsharded on this GPU (T1) -> projected_speedup = T1 / slowest rank.  A projection: no RCCL wire time in it.
        import torch
        sys.path.insert(0, os.path.join(ROOT, "tools"))
        import virtual_ranks as vr
        from flashdeconv_amd import FlashDeconv
        torch.cuda.set_device(0)
        dev = torch.device("cuda", 0)
        big = a.config == 5
        n = a.spots if a.spots != 1_000_000 or not big else 10_000_000
        G, K, d = (5000, 50, 1024) if big else (a.genes, a.types, a.sketch_dim)
        W = a.virtual_ranks
        info = None
        reps = max(1, min(a.warmup, 1)) + max(1, min(a.steps, 2))
        for it in range(reps):                       # the first pass warms the schedule caches; the last one is reported
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            keep = {}
            vr_kw = dict(pre="log_cpm", tol=1e-300, max_iter=100) if a.vr_family == "counts" else {}
            _, _, info = vr.run_config5(torch, W, n=n, G=G, K=K, d=d, seed=11, alone=True, keep=keep, **vr_kw)
            torch.cuda.synchronize()
            wall = time.perf_counter() - t0
            coords, X = keep["coords"], keep["X"]
            for R in keep["ranks"]:
                R["g"].close()
            del keep
        torch.cuda.empty_cache()                     # only now: between the passes the freed blocks stay with torch's allocator, as in a job that fits twice
        from flashdeconv_amd import _lib as _fdx_lib
        _fdx_lib.load().fdx_trim()                   # ... and libfdx's pooled scratch (a 5M-spot rank's H and abundances): T1 needs the room at 10M spots
        t = info["times"]
        crit = t["per_rank_critical_path_ms"]
        # T1: the SAME job (same coordinates, same rows: generated chunk by chunk from the same seeds) unsharded on this GPU
        t1_ms, t1_iters = None, None
        if not os.environ.get("FDX_BENCH_NO_T1"):
            X32 = torch.from_numpy(X).to(dev).float()
            Y = torch.empty((n, G), dtype=torch.float32, device=dev)
            step = 1 << 20
            for r0 in range(0, n, step):
                Y[r0:min(n, r0 + step)] = vr.gaussian_rows(torch, X32, r0, min(n, r0 + step), 11)
            if a.vr_family == "counts":
                Y.abs_()
                model = FlashDeconv(sketch_dim=d, preprocess="log_cpm", n_hvg=G, max_iter=100, tol=1e-300)
            else:
                model = FlashDeconv(sketch_dim=d, preprocess="raw", n_hvg=G)
            model.fit(Y, X, coords, output="torch")
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            n_t1 = 12 if not big else 3          # (three fits of the 1M job differed by up to 10 % from run to run: averaged over more)
            for _ in range(n_t1):
                model.fit(Y, X, coords, output="torch")
            torch.cuda.synchronize()
            t1_ms = (time.perf_counter() - t0) / n_t1 * 1e3
            t1_iters = model.info_["n_iterations"]
            del Y, model
        line = {"family": a.vr_family,
                "metric": f"projected strong scaling of {'configs[4] (10M x 5000 x 50, d 1024)' if big else 'configs[3] (1M x 2000 x 30, d 512)'}"
                          f" over {W} ranks: per-rank critical path on one GPU, each rank timed alone",
                "n_gpus": 1, "virtual_ranks": W, "spots": n, "n_iterations": info["n_iterations"][0], "t1_n_iterations": t1_iters,
                "knn_ties": info["knn_ties"], "nnz": info["nnz"], "n_halo": info["n_halo"], "plan_route": t.get("plan_route"),
                "per_rank_critical_path_ms": [round(x, 3) for x in crit],
                "per_rank_stage_sum_ms": [round(x, 3) for x in t.get("per_rank_stage_sum_ms", crit)],
                "per_rank_c_calls_ms": [round(x, 3) for x in t["pipelined_ms"]] if "pipelined_ms" in t else None,
                "per_rank_stage_ms": {k: t[k] for k in ("plan_ms", "prepare_ms", "solve_alone_ms", "finish_alone_ms") if k in t},
                "t1_ms": None if t1_ms is None else round(t1_ms, 3),
                "projected_speedup": None if t1_ms is None else round(t1_ms / max(crit), 2),
                "projection_note": "projection, no RCCL wire time: every rank's whole share through the real driver "
                                   "(ShardedFlashDeconv.plan + fit_transform with a LoopbackComm: leverage, tables, plan, sketch -> H, the "
                                   "native iteration loop over the loopback transport for the job's iteration count, export + objective), "
                                   "timed ALONE on one GPU as ONE interval with warm caches (per_rank_c_calls_ms: the same C calls without "
                                   "the class; per_rank_stage_sum_ms: the stages synchronised one by one); T1 = the same job unsharded "
                                   "on the same GPU",
                "stage_ms": t, "wall_s_incl_generation": round(wall, 2), "data": "synthetic"}
        print(json.dumps(line))
        return
    if a.config == 5:      # the shape of one of the eight shards of BASELINE configs[4] (10M x 5000 x 50, d 1024)
        a.spots, a.genes, a.types, a.sketch_dim, a.family, a.no_cpu_baseline = 1_250_000, 5000, 50, 1024, "gaussian", True
    if a.gpus > 1 and "WORLD_SIZE" not in os.environ:
        spawn_ranks(a.gpus, a.rendezvous_timeout)
    if os.environ.get("FDX_BENCH_SPAWN_ECHO"):      # launcher self-test (tests/test_host.py): report the rank environment
        if os.environ.get("FDX_BENCH_SPAWN_FAIL_RANK") == os.environ.get("RANK"):
            print("rank asked to fail", file=sys.stderr)
            sys.exit(3)
        if os.environ.get("FDX_BENCH_SPAWN_HANG"):    # the others of a failing job: sit as in a collective
            time.sleep(float(os.environ["FDX_BENCH_SPAWN_HANG"]))
        print(json.dumps({k: os.environ.get(k) for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT")}
                         | {"gpus": a.gpus, "scaling": a.scaling}))
        return
    import torch
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    force_dist = bool(os.environ.get("FDX_BENCH_FORCE_DIST"))       # exercise the sharded driver with one rank
    if world > 1 or force_dist:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29533")
        if os.environ.get("FDX_BENCH_BACKEND", "nccl") == "gloo":
            # REHEARSAL of the N > 1 driver on a box with one GPU: every rank on cuda:0, collectives over gloo, the Python exchange
            # loop instead of the native RCCL one.  Checks the code path and the result line; its numbers mean nothing.
            local_rank = 0
            torch.cuda.set_device(0)
            dist.init_process_group("gloo", rank=rank, world_size=world)
        else:
            torch.cuda.set_device(local_rank)
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local_rank))
        barrier = dist.barrier
    else:
        barrier = lambda: None
    if world != a.gpus and rank == 0:
        print(f"warning: --gpus {a.gpus} but WORLD_SIZE={world}", file=sys.stderr)
    if world > 1 or force_dist:
        from flashdeconv_amd import distributed as fdist   # sharded driver
        return fdist.bench_main(a, rank, world, local_rank)

    device = torch.device("cuda", 0)
    torch.cuda.set_device(device)
    n, G, K, d = a.spots, a.genes, a.types, a.sketch_dim
    results = {}
    cpu_inputs = None
    y_f64 = None
    fams = {"both": ["gaussian", "counts"], "all": ["gaussian", "counts", "sparse", "lattice"]}.get(a.family, [a.family])
    for fam in fams:
        if fam == "lattice":
            # Visium-HD bins sit on a regular lattice: with k = 6 every spot's k-th neighbour is tied (four at distance 1, four at
            # sqrt 2) and the default knn_ties="auto" has to reproduce the reference's (cKDTree) choice - the restated tree (built on
            # the device since the end of round 6) plus its queries.  Count-like rows, log_cpm (runs max_iter).
            Y, X, coords = gen_counts(torch, n, G, K, device, seed=0)
            side = int(np.ceil(np.sqrt(n)))
            ii = torch.arange(n, device=device)
            coords = torch.stack([(ii % side).double(), (ii // side).double()], dim=1)
            kw = dict(sketch_dim=d, preprocess="log_cpm", n_hvg=G)
            lsteps = aux_steps(a)
            import warnings
            with warnings.catch_warnings():
                warnings.simplefilter("ignore")
                model, dt, stage = run_family(torch, kw, Y, X, coords, lsteps, a.warmup, barrier)
            ms_step = dt / lsteps * 1e3
            results[fam] = {
                "value": n * lsteps / dt, "unit": "spots/s", "ms_per_step": ms_step, "steps": lsteps, "warmup": a.warmup,
                "step_ms_min_max": stage.pop("step_ms_min_max"), "cold_ms": round(stage["cold_ms"], 3),
                "workload": f"{n} spots on a {side} x {side} square lattice (every k-th neighbour tied), {G} genes x {K} types, count-like / "
                            f"log_cpm, knn_ties='auto' (default): the reference's cKDTree tie order",
                "n_iterations": model.info_["n_iterations"], "converged": model.info_["converged"],
                "knn_ties": model.info_["knn_ties"], "ties_remedy_ms": round(stage.get("ties_remedy_ms", 0.0), 3),
                "stage_ms": stage_record(stage, ms_step),
            }
            del Y, coords, model
            torch.cuda.empty_cache()
            continue
        if fam == "sparse":
            Y, X, coords = gen_sparse(torch, n, a.sparse_genes, K, device, seed=0)
            kw = dict(sketch_dim=d, preprocess="log_cpm", n_hvg=G, max_iter=20)
            ssteps = aux_steps(a)
            model, dt, stage = run_family(torch, kw, Y, X, coords, ssteps, a.warmup, barrier)
            nnz_y = int(Y.values().numel())
            # fused CSR sketch -> H (gram_ms == 0): reads the stored entries (4 B value + 4 B column each) and the row extents, writes
            # H; the two-kernel path also writes and re-reads Y_sketch (n x d x 8 B)
            csr_bytes = nnz_y * 8 + n * 8 + n * K * 8 + (2 * n * d * 8 if stage["gram_ms"] > 0.0 else 0)
            csr_ms = stage["sketch_ms"] + stage["gram_ms"]
            csr_gbs = csr_bytes / (csr_ms * 1e-3) / 1e9
            ms_step = dt / ssteps * 1e3
            results[fam] = {
                "value": n * ssteps / dt, "unit": "spots/s", "ms_per_step": ms_step, "steps": ssteps, "warmup": a.warmup,
                "step_ms_min_max": stage.pop("step_ms_min_max"), "cold_ms": round(stage["cold_ms"], 3),
                "workload": f"CSR input: {n} spots x {a.sparse_genes} genes, {nnz_y} stored entries ({nnz_y / n / a.sparse_genes:.3f} dense, "
                            f"{nnz_y / n:.0f} per spot), float32 values + int32 columns in HBM; HVG+markers select "
                            f"{len(model.gene_idx_)} genes, log_cpm, max_iter 20 (the solve is not the subject here)",
                "n_iterations": model.info_["n_iterations"], "converged": model.info_["converged"],
                "stage_ms": stage_record(stage, ms_step),
                "sketch_GBps": round(csr_gbs, 1),
                "roofline": {"bound": "hbm", "kernel": "fdx::sketch_csr_contract_kernel<float, 2, 2, 2, 24, 12>" if stage["gram_ms"] == 0.0
                             else "fdx::sketch_csr_kernel<float, 2>", "achieved": round(csr_gbs, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                             "frac": round(csr_gbs / HBM_PEAK_GBS, 4),
                             "traffic": pmc_traffic("fdx::sketch_csr_contract_kernel", (n, G, K, d)) if stage["gram_ms"] == 0.0 else None,
                             "traffic_source": TRAFFIC_PROFILE + " (PMC pass of an earlier run of this build, not this run)",
                             "alg_bytes_per_launch": int(csr_bytes), "ms_per_launch": round(csr_ms, 4)},
            }
            del Y, coords, model
            torch.cuda.empty_cache()
            continue
        if fam == "gaussian":
            Y, X, coords = gen_gaussian(torch, n, G, K, device, seed=0)
            kw = dict(sketch_dim=d, preprocess="raw", n_hvg=G)
        else:
            Y, X, coords = gen_counts(torch, n, G, K, device, seed=0)
            kw = dict(sketch_dim=d, preprocess="log_cpm", n_hvg=G)
        steps = a.steps if fam == "gaussian" else aux_steps(a)
        model, dt, stage = run_family(torch, kw, Y, X, coords, steps, a.warmup, barrier)
        T = model.info_["n_iterations"]
        nnz = int(model._graph.info()[1])
        sk_bytes, sw_bytes = alg_bytes(n, G, K, 4, nnz, 0, T)
        sweep_ms = stage["sweep_ms"] / max(T, 1)
        sk_ms = stage["sketch_ms"]
        dom = "bcd_sweep" if stage["sweep_ms"] >= sk_ms else "sketch_rows"
        n_chunks = -(-n // (1 << 18))                 # the sketch kernel is launched once per 262144-row chunk (fit.cpp)
        if dom == "bcd_sweep":
            bytes_launch, ms_launch = sw_bytes, sweep_ms
            kname = sweep_kernel_name(K)
        else:
            if stage["gram_ms"] == 0.0:                # fused sketch -> H kernel: ONE launch reads all of Y, writes only H
                bytes_launch, ms_launch = sk_bytes, sk_ms
                kname = sketch_kernel_name(0 if fam == "gaussian" else 1, K, d)
            else:
                bytes_launch, ms_launch = sk_bytes / n_chunks, sk_ms / n_chunks
                kname = "fdx::sketch_rows_scatter_kernel<float, %d, true>" % (0 if fam == "gaussian" else 1)
        ach = bytes_launch / (ms_launch * 1e-3) / 1e9
        results[fam] = {
            "value": n * steps / dt, "ms_per_step": dt / steps * 1e3, "step_ms_min_max": stage.pop("step_ms_min_max"),
            "cold_ms": round(stage["cold_ms"], 3), "n_iterations": T,
            "converged": model.info_["converged"], "stage_ms": stage_record(stage, dt / steps * 1e3),
            "roofline": {"bound": "hbm", "kernel": kname, "achieved": round(ach, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": round(ach / HBM_PEAK_GBS, 4), "traffic": pmc_traffic(kname, (n, G, K, d)),
                         "traffic_source": (TRAFFIC_PROFILE_C5 if a.config == 5 else TRAFFIC_PROFILE) +
                                           " (PMC pass of an earlier run of this build, not this run)",
                         "alg_bytes_per_launch": int(bytes_launch), "ms_per_launch": round(ms_launch, 4)},
            "steps": steps,
        }
        if fam == "gaussian" and a.config == 3 and a.family == "all":
            # the same job with Y stored float64 in HBM - the reference's own dtype (core/deconv.py:229): twice the one-off read
            Y64 = Y.double()
            m64, dt64, st64 = run_family(torch, kw, Y64, X, coords, aux_steps(a), a.warmup, barrier)
            ms64 = dt64 / aux_steps(a) * 1e3
            g64 = n * G * 8 / (st64["sketch_ms"] * 1e-3) / 1e9
            y_f64 = {"value": n / (ms64 * 1e-3), "unit": "spots/s", "ms_per_step": ms64, "steps": aux_steps(a),
                     "step_ms_min_max": st64.pop("step_ms_min_max"), "cold_ms": round(st64["cold_ms"], 3),
                     "n_iterations": m64.info_["n_iterations"], "converged": m64.info_["converged"], "workload": "same job, Y float64 in HBM (16 GB)",
                     "roofline": {"bound": "hbm", "kernel": "fdx::tile_sketch_kernel<double, ...>", "achieved": round(g64, 1), "peak": HBM_PEAK_GBS,
                                  "unit": "GB/s", "frac": round(g64 / HBM_PEAK_GBS, 4), "traffic": None,
                                  "alg_bytes_per_launch": int(n * G * 8), "ms_per_launch": round(st64["sketch_ms"], 4)},
                     "stage_ms": stage_record(st64, ms64)}
            del Y64, m64
            torch.cuda.empty_cache()
        if fam == "gaussian" and not a.no_cpu_baseline and a.config == 3:
            # the CPU baseline runs on these very rows, at the end (outside every timed region): they are generated again there
            # from the same seed - an 8 GB host copy held through the other families' timed steps is host memory the kernel may
            # compact or migrate under them
            cpu_inputs = True
        del Y, coords, model
        torch.cuda.empty_cache()

    main_fam = "gaussian" if "gaussian" in results else ("counts" if "counts" in results else ("sparse" if "sparse" in results else "lattice"))
    if main_fam in ("sparse", "lattice"):
        print(json.dumps({"metric": f"spots/sec ({'CSR' if main_fam == 'sparse' else 'lattice'} family only)", **results[main_fam]}))
        return
    r = results[main_fam]
    line = {
        "metric": METRIC if a.config == 3 else "spots/sec to convergence (one 1.25M-spot shard of 10M x 5000 x 50, d 1024)",
        "value": r["value"], "unit": "spots/s", "n_gpus": 1,
        "steps": r["steps"], "warmup": a.warmup, "ms_per_step": r["ms_per_step"], "higher_is_better": True,
        "scaling": "strong", "vs_baseline": None, "dtype": "f64", "data": "synthetic",
        "config": {"workload": f"{n} spots x {G} genes x {K} types, sketch_dim {d}, k_neighbors 6, "
                               f"{'gaussian/raw' if main_fam == 'gaussian' else 'count-like/log_cpm'} family, Y float32 in HBM, "
                               f"tol 1e-4, max_iter 100", "n_iterations": r["n_iterations"], "converged": r["converged"]},
        "roofline": r["roofline"], "stage_ms": r["stage_ms"], "cold_ms": r["cold_ms"], "step_ms_min_max": r["step_ms_min_max"],
    }
    if "counts" in results and main_fam != "counts":
        c = results["counts"]
        line["count_like"] = {"value": c["value"], "unit": "spots/s", "ms_per_step": c["ms_per_step"], "steps": c["steps"],
                              "step_ms_min_max": c["step_ms_min_max"], "cold_ms": c["cold_ms"],
                              "n_iterations": c["n_iterations"], "converged": c["converged"], "roofline": c["roofline"],
                              "stage_ms": c["stage_ms"], "workload": "same shape, count-like / log_cpm family (runs max_iter)"}
    if y_f64 is not None:
        line["y_f64"] = y_f64
    if "sparse" in results:
        line["sparse_csr"] = results["sparse"]
    if "lattice" in results:
        line["lattice"] = results["lattice"]
    if not a.no_live_pmc and a.config == 3:
        live = live_pmc_traffic(a)
        apply_live_traffic(line.get("roofline"), live)
        for key in ("count_like", "sparse_csr"):
            if key in line:
                apply_live_traffic(line[key].get("roofline"), live)
    if not a.no_host_arrays and a.config == 3 and a.family == "all":
        line["host_arrays"] = host_arrays_record(torch, n, G, K, d, a.sparse_genes, device)
    if cpu_inputs is not None:
        from flashdeconv_amd import _lib as _fdx_lib
        Yc, Xc, cc = gen_gaussian(torch, n, G, K, device, seed=0)       # the headline family's own rows (same generator, same seed)
        n_cpu = min(a.cpu_sample, n)
        host = (_fdx_lib.tensor_to_host(Yc[:n_cpu]), Xc, _fdx_lib.tensor_to_host(cc[:n_cpu]))
        del Yc, cc
        torch.cuda.empty_cache()
        line["cpu_baseline"] = cpu_baseline(*host, d)
    print(json.dumps(line))


if __name__ == "__main__":
    main()
