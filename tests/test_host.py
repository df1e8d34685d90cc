"""CPU-only tests of the host side: the C ABI library loads and exports every symbol include/fdx.h declares, the Python
mirror of the reference interface validates like the reference, the hash/sign tables are bit-exact, and the product never
touches the oracle or a CPU fallback.  No kernel is launched here."""
import ctypes
import os
import re

import numpy as np
import pytest
from scipy import sparse

from conftest import ROOT, load_golden


def _declared_functions():
    text = open(os.path.join(ROOT, "include", "fdx.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(fdx_[a-z0-9_]+)\s*\(", text)))


def test_library_exports_every_declared_symbol():
    from flashdeconv_amd import _lib
    lib = _lib.load()
    names = _declared_functions()
    assert len(names) >= 30
    for name in names:
        assert hasattr(lib, name), f"{name} declared in include/fdx.h but not exported by libfdx.so"
        assert name in _lib.SIGNATURES, f"{name} has no ctypes signature in flashdeconv_amd/_lib.py"
    for name in _lib.SIGNATURES:
        assert name in names, f"{name} bound in _lib.py but not declared in include/fdx.h"
    assert lib.fdx_version() >= 100
    assert isinstance(lib.fdx_last_error(), bytes)


def test_no_gpu_fails_loudly_and_no_fallback():
    from flashdeconv_amd import _lib
    lib = _lib.load()
    n = ctypes.c_int(-1)
    rc = lib.fdx_device_count(ctypes.byref(n))
    if rc == 0 and n.value > 0:
        pytest.skip("a GPU is visible")
    with pytest.raises(_lib.FdxError, match="no CPU fallback"):
        _lib.require_gpu()
    from flashdeconv_amd import FlashDeconv
    from flashdeconv_amd.core.solver import bcd_solve
    with pytest.raises(_lib.FdxError):
        bcd_solve(np.zeros((4, 3)), np.ones((2, 3)), sparse.identity(4, format="csr"))
    with pytest.raises(_lib.FdxError):
        FlashDeconv().fit(np.ones((5, 7)), np.ones((2, 7)), np.random.rand(5, 2))


def test_product_never_imports_the_oracle():
    pkg = os.path.join(ROOT, "flashdeconv_amd")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".cpp", ".h")):
                src = open(os.path.join(dirpath, f)).read()
                assert "fdx_oracle" not in src and "liboracle" not in src and "oracle/" not in src, f


def test_constructor_and_fit_validation_messages():
    # reference core/deconv.py:105-124, 270-292 and tests/test_integration.py:384-422
    from flashdeconv_amd import FlashDeconv
    for kw, msg in [(dict(sketch_dim=0), "sketch_dim must be positive"), (dict(k_neighbors=-1), "k_neighbors must be non-negative"),
                    (dict(max_iter=-1), "max_iter must be non-negative"), (dict(tol=0), "tol must be positive"),
                    (dict(lambda_spatial=-1.0), "lambda_spatial must be non-negative"), (dict(rho_sparsity=-0.1), "rho_sparsity"),
                    (dict(n_hvg=-1), "n_hvg"), (dict(n_markers_per_type=-2), "n_markers_per_type"),
                    (dict(spatial_method="radius"), "radius must be specified"), (dict(radius=-1.0), "radius must be positive")]:
        with pytest.raises(ValueError, match=msg):
            FlashDeconv(**kw)
    m = FlashDeconv()
    assert repr(m) == "FlashDeconv(sketch_dim=512, lambda_spatial=auto, status=not fitted)"
    assert m.summary() == {"fitted": False}
    for getter in (m.get_cell_type_proportions, m.get_abundances, m.get_dominant_cell_type):
        with pytest.raises(RuntimeError, match="Model has not been fitted"):
            getter()
    Y, X, c = np.ones((6, 9)), np.ones((3, 9)), np.random.rand(6, 2)
    with pytest.raises(ValueError, match="Gene dimension mismatch"):
        m.fit(Y, X[:, :4], c)
    with pytest.raises(ValueError, match="Spot count mismatch"):
        m.fit(Y, X, c[:3])
    with pytest.raises(ValueError, match="at least one cell type"):
        m.fit(Y, X[:0], c)
    with pytest.raises(ValueError, match="cell_type_names length"):
        m.fit(Y, X, c, cell_type_names=np.array(["a"]))
    with pytest.raises(ValueError, match="Unknown preprocess method"):
        FlashDeconv(preprocess="zscore").fit(Y, X, c)


def test_countsketch_tables_bit_exact_on_host():
    from flashdeconv_amd.core.sketching import build_countsketch_matrix, countsketch_tables
    g = load_golden("omega_tables.npz")
    for (G, d, s) in g["cases"]:
        tag = f"G{G}_d{d}_s{s}"
        bucket, weight = countsketch_tables(int(G), int(d), None, int(s))
        assert np.array_equal(bucket, g[tag + "_bucket"])
        assert np.array_equal(np.sign(weight).astype(np.int64), g[tag + "_sign"])
        np.testing.assert_allclose(weight, g[tag + "_data"], rtol=1e-15)
    Om = build_countsketch_matrix(300, 64, g["lev_input"], 11).tocsr()
    assert np.array_equal(Om.indices, g["lev_bucket"]) and np.all(np.diff(Om.indptr) == 1)
    np.testing.assert_allclose(Om.data, g["lev_data"], rtol=1e-14)
    # seed reproducibility / RandomState instance / None (reference tests/test_sketching.py:36-41)
    a = countsketch_tables(50, 8, None, 3)
    b = countsketch_tables(50, 8, None, np.random.RandomState(3))
    assert np.array_equal(a[0], b[0]) and np.array_equal(a[1], b[1])
    np.random.seed(3)
    c = countsketch_tables(50, 8, None, None)
    assert np.array_equal(a[0], c[0])


def test_check_random_state_semantics():
    from flashdeconv_amd.utils.random import check_random_state
    assert check_random_state(None) is np.random.mtrand._rand
    rs = np.random.RandomState(1)
    assert check_random_state(rs) is rs
    assert check_random_state(5).randint(0, 100) == np.random.RandomState(5).randint(0, 100)
    with pytest.raises(ValueError, match="cannot be used to seed"):
        check_random_state("x")


def test_markers_and_hvg_binning_vs_reference_golden():
    from flashdeconv_amd.utils import genes
    g = load_golden("fit_genesel_150x600x4.npz")
    idx, assign = genes.select_markers(g["X"], n_markers=10)
    assert np.array_equal(idx, g["marker_idx"]) and np.array_equal(assign, g["marker_assign"])
    Y = g["Y"].astype(np.float64)
    Z = np.log1p(Y / np.maximum(Y.sum(axis=1, keepdims=True), 1) * 10000)
    hv = genes._hvg_from_moments(Z.mean(axis=0), Z.var(axis=0, ddof=1), 200, 0.0125, 3.0, 0.5)
    assert np.array_equal(hv, g["hvg_idx"])
    assert np.array_equal(genes.select_hvg(Y[:, :150], n_top=200), np.arange(150))        # identity short-circuit
    assert genes.select_markers(g["X"], 0)[0].size == 0
    with pytest.raises(ValueError, match="non-negative"):
        genes.select_markers(g["X"], -1)


def test_select_markers_three_methods_vs_reference_golden():
    """utils/genes.py:197-211: "diff", "ratio" and "specificity" scores, incl. the fallback for a type that tops no gene."""
    from flashdeconv_amd.utils import genes
    g = load_golden("markers.npz")
    for tag in "abcd":
        X, nm = g[f"{tag}_X"], int(g[f"{tag}_n_markers"])
        for method in ("diff", "ratio", "specificity"):
            idx, assign = genes.select_markers(X, n_markers=nm, method=method)
            assert np.array_equal(idx, g[f"{tag}_{method}_idx"]), (tag, method)
            assert np.array_equal(assign, g[f"{tag}_{method}_assign"]), (tag, method)
    with pytest.raises(ValueError, match="Unknown method"):
        genes.select_markers(g["a_X"], 5, method="nope")


def test_utils_package_exports_the_reference_names():
    """flashdeconv/utils/__init__.py:3-31 minus the evaluation metrics (out of scope)."""
    import flashdeconv_amd.utils as u
    for name in ("select_hvg", "select_markers", "compute_leverage_scores", "build_knn_graph", "build_radius_graph",
                 "coords_to_adjacency", "check_random_state"):
        assert callable(getattr(u, name)) and name in u.__all__


def test_coordinate_dimension_limits_are_value_errors_naming_the_limit():
    """utils/graph.py:16-22 takes any dimension; here k-NN takes 1-8 (4-8 by exhaustive search, up to 262144 spots), radius / grid 1-3."""
    from flashdeconv_amd.utils.graph import check_coord_dims
    check_coord_dims(1000, 3, False)
    check_coord_dims(1000, 8, True)
    with pytest.raises(ValueError, match="radius / grid graphs are built for 1 to 3"):
        check_coord_dims(1000, 4, False)
    with pytest.raises(ValueError, match="1 to 8 coordinate dimensions"):
        check_coord_dims(1000, 9, True)
    with pytest.raises(ValueError, match="at most 262144 spots"):
        check_coord_dims(300000, 4, True)


def test_spatial_helpers():
    from flashdeconv_amd.core.spatial import auto_tune_lambda, compute_laplacian, compute_laplacian_quadratic, get_neighbor_indices
    A = sparse.csr_matrix(np.array([[0, 1, 1, 0], [1, 0, 0, 0], [1, 0, 0, 1], [0, 0, 1, 0]], dtype=float))
    L = compute_laplacian(A)
    assert np.allclose(np.asarray(L.sum(axis=1)).ravel(), 0)                               # reference tests/test_spatial.py:122-139
    assert np.all(compute_laplacian(A, normalized=True).diagonal() <= 1 + 1e-12)
    assert compute_laplacian_quadratic(np.ones((4, 3)), L) == pytest.approx(0.0)
    assert [list(v) for v in get_neighbor_indices(A)] == [[1, 2], [0], [0, 3], [2]]
    Xs = np.arange(12.0).reshape(3, 4)
    assert auto_tune_lambda(None, Xs, A) == pytest.approx(0.005 * np.mean(np.diag(Xs @ Xs.T)) / 1.5)


def test_normalize_proportions_exact():
    from flashdeconv_amd.core.solver import normalize_proportions
    g = load_golden("solver_small.npz")
    np.testing.assert_array_equal(normalize_proportions(g["norm_in"]), g["norm_out"])


def test_public_header_is_plain_c99(tmp_path):
    """include/fdx.h is the drop-in boundary: it must compile as C (no C++-isms, no torch / HIP types)."""
    import shutil
    import subprocess
    gcc = shutil.which("gcc")
    if gcc is None:
        pytest.skip("no gcc")
    src = tmp_path / "t.c"
    src.write_text('#include "fdx.h"\nint main(void) { return fdx_version() ? 0 : 0; }\n')
    r = subprocess.run([gcc, "-std=c99", "-Wall", "-Wextra", "-pedantic", "-Werror", "-I", os.path.join(ROOT, "include"), "-c", str(src),
                        "-o", str(tmp_path / "t.o")], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr


@pytest.mark.parametrize("G,d,NW,JW,GB", [(2000, 512, 16, 8, 1024), (2000, 512, 8, 16, 768), (2000, 512, 16, 8, 2048),
                                          (3300, 512, 16, 8, 768), (1001, 500, 16, 8, 256), (300, 64, 16, 8, 512),
                                          (5000, 512, 8, 16, 512), (40, 7, 16, 8, 256)])
def test_tile_schedule_replays_to_the_countsketch(G, d, NW, JW, GB):
    """The tile kernel's static schedule (csrc/tile_plan.cpp), replayed on the host exactly as the kernel walks it,
    must reproduce f(Y) @ Omega: every gene visited once, in its bucket, with its weight, inside its column block."""
    import fdx_oracle as orc
    from flashdeconv_amd import _lib
    lib = _lib.load()
    rs = np.random.RandomState(G + d)
    bucket, weight = orc.countsketch_omega(G, d, rs.rand(G), 3)
    bucket = bucket.astype(np.int32)
    if G > 50:
        bucket[::37] = -1                                    # genes outside Omega (a CSR-style selected subset)
    dims = np.zeros(4, dtype=np.int32)
    _lib.check(lib.fdx_tile_schedule(_lib.ptr_i32(bucket), _lib.ptr_f64(weight), G, d, NW, JW, GB, _lib.ptr_i32(dims),
                                     None, None, None, None, None, 0))
    nblk, ne, steps, wave_max = (int(x) for x in dims)
    assert nblk == -(-G // GB)
    slot = np.zeros(NW * JW * 4, dtype=np.int32)
    ln = np.zeros(NW * nblk * JW, dtype=np.uint8)
    base = np.zeros(NW * (nblk + 1), dtype=np.int32)
    w = np.zeros(ne)
    off = np.zeros(ne, dtype=np.uint16)
    _lib.check(lib.fdx_tile_schedule(_lib.ptr_i32(bucket), _lib.ptr_f64(weight), G, d, NW, JW, GB, _lib.ptr_i32(dims),
                                     _lib.ptr_i32(slot), ln.ctypes.data, _lib.ptr_i32(base), _lib.ptr_f64(w), off.ctypes.data, ne))
    slot = slot.reshape(NW, JW, 4)
    ln = ln.reshape(NW, nblk, JW)
    base = base.reshape(NW, nblk + 1)
    used = slot[slot >= 0]
    assert np.array_equal(np.sort(used), np.arange(d))       # every bucket owned by exactly one slot
    y = rs.randn(G)
    sk = np.zeros(d)
    seen = np.zeros(G, dtype=int)
    for wv in range(NW):
        for c in range(nblk):
            p = base[wv, c]
            for j in range(JW):
                for _ in range(ln[wv, c, j]):
                    for q in range(4):
                        e = p + q
                        g = c * GB + int(off[e])
                        assert g < G and g < (c + 1) * GB
                        if w[e] != 0.0:
                            b = slot[wv, j, q]
                            assert b >= 0 and bucket[g] == b and w[e] == weight[g]
                            sk[b] += w[e] * y[g]
                            seen[g] += 1
                    p += 4
            assert p == base[wv, c + 1]
        assert base[wv, nblk] + 8 <= (base[wv + 1, 0] if wv + 1 < NW else ne)   # two padding steps per wave
    inside = bucket >= 0
    assert np.array_equal(seen[inside & (weight != 0)], np.ones(int((inside & (weight != 0)).sum()), dtype=int))
    assert not seen[~inside].any()
    want = np.zeros(d)
    np.add.at(want, bucket[inside], weight[inside] * y[inside])
    np.testing.assert_allclose(sk, want, rtol=1e-12, atol=1e-12)
    # lockstep padding: at most 35 % more lane-steps than genes for the shapes the fit uses (G >= 1000, up to 5 blocks)
    if G >= 1000 and nblk <= 5:
        assert steps * 4 <= 1.35 * inside.sum(), (steps * 4, inside.sum())
        assert wave_max * NW <= 1.15 * steps


def test_bench_spawns_its_own_ranks():
    """`python bench.py --gpus N` from a bare shell (no WORLD_SIZE) must start N ranks itself - one fresh process per GPU,
    before anything touches a GPU - and relay rank 0's line.  FDX_BENCH_SPAWN_ECHO makes the ranks report their
    environment instead of benchmarking."""
    import json
    import subprocess
    import sys
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT")}
    env["FDX_BENCH_SPAWN_ECHO"] = "1"
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "3", "--scaling", "weak"], env=env,
                         capture_output=True, text=True, timeout=120)
    assert out.returncode == 0, out.stderr
    line = json.loads(out.stdout.strip().splitlines()[-1])
    assert line["RANK"] == "0" and line["LOCAL_RANK"] == "0" and line["WORLD_SIZE"] == "3" and line["MASTER_ADDR"] == "127.0.0.1"
    assert line["gpus"] == 3 and line["scaling"] == "weak" and int(line["MASTER_PORT"]) > 0
    # one rank dies at start-up (exit code 3), the others sit as in a collective: the parent ends them and returns non-zero
    # within seconds, naming the rank and where its stderr went
    import time
    env3 = dict(env, FDX_BENCH_SPAWN_FAIL_RANK="1", FDX_BENCH_SPAWN_HANG="120")
    t0 = time.monotonic()
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "3"], env=env3, capture_output=True, text=True,
                         timeout=100)
    assert out.returncode == 3 and time.monotonic() - t0 < 30, (out.returncode, out.stderr)
    assert "rank 1 of 3 exited with code 3" in out.stderr and "rank asked to fail" in out.stderr
    # ... and a job whose ranks never produce a result ends at the rendezvous timeout
    env4 = dict(env, FDX_BENCH_SPAWN_HANG="120")
    t0 = time.monotonic()
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--rendezvous-timeout", "2"], env=env4,
                         capture_output=True, text=True, timeout=100)
    assert out.returncode == 124 and time.monotonic() - t0 < 30 and "produced no result in time" in out.stderr
    # under torch.distributed.run the environment is already there: no second level of processes
    env2 = dict(env, RANK="1", LOCAL_RANK="1", WORLD_SIZE="2", MASTER_ADDR="127.0.0.1", MASTER_PORT="29999")
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2"], env=env2, capture_output=True, text=True,
                         timeout=120)
    line = json.loads(out.stdout.strip().splitlines()[-1])
    assert line["RANK"] == "1" and line["WORLD_SIZE"] == "2" and line["scaling"] == "strong"


def test_schedule_builders_under_address_and_ub_sanitizers():
    """`make -C flashdeconv_amd/csrc asan-host`: the pure-host schedule builders (tile_plan.cpp) compiled with g++
    -fsanitize=address,undefined and replayed on a range of shapes (GPU AddressSanitizer is not available on the build pool)."""
    import shutil
    import subprocess
    if not shutil.which("g++") or not shutil.which("make"):
        pytest.skip("no g++ / make")
    r = subprocess.run(["make", "-s", "-C", os.path.join(ROOT, "flashdeconv_amd", "csrc"), "asan-host"], capture_output=True, text=True,
                       timeout=300)
    if r.returncode != 0 and ("cannot find -lasan" in r.stderr or "libasan" in r.stderr):
        pytest.skip("no sanitizer runtime")
    assert r.returncode == 0, r.stdout + r.stderr
    assert "ok under the sanitizers" in r.stdout


def test_bench_sweep_chunk_mirrors_the_kernel_table():
    """bench.py names the sweep kernel of the roofline line; its chunk table must be the one in csrc/bcd_sweep_inst.cpp."""
    import re
    import bench
    src = open(os.path.join(ROOT, "flashdeconv_amd", "csrc", "bcd_sweep_inst.cpp")).read()
    m = re.search(r"return K < 8 \? K : ([^;]+);", src)
    assert m, "sweep_chunk() not found"
    expr = m.group(1)

    def c_table(K):
        e = expr
        while True:
            mm = re.match(r"\s*K (<=|==) (\d+) \? (\d+) : (.*)", e)
            if not mm:
                return int(e.strip())
            op, b, v, rest = mm.group(1), int(mm.group(2)), int(mm.group(3)), mm.group(4)
            if (op == "<=" and K <= b) or (op == "==" and K == b):
                return v
            e = rest

    for K in list(range(1, 65)) + [72, 80, 88, 96]:          # every instantiated size (csrc/fdx_kernels.h: solver_padded_K)
        want = K if K < 8 else c_table(K)
        assert bench.sweep_chunk(K) == want, K


def test_ckdtree_order_restatement_on_heavily_duplicated_coordinates():
    """Many spots on few distinct places (integer coordinates out of 2 to 1000 values, 1 to 3 dimensions): the median of a node is
    often its MINIMUM along the split dimension, where scipy 1.15.3 splits just above it (split == nextafter(minimum, +inf),
    every point at the minimum in the lesser child).  Index array and query lists against scipy itself, with the passes of the
    nodes on the building thread and on the thread pool's tasks (fdx_kdtree_tune)."""
    from scipy.spatial import cKDTree
    from flashdeconv_amd import _lib
    lib = _lib.load()
    rs = np.random.RandomState(0)
    try:
        for it in range(90):
            n, dim, rng = int(rs.randint(70, 6000)), int(rs.randint(1, 4)), int(rs.choice([2, 5, 30, 1000]))
            coords = np.ascontiguousarray(rs.randint(0, rng, size=(n, dim)).astype(np.float64))
            kk = 5
            lib.fdx_kdtree_tune(0, 64 if it % 3 else 0)
            lib.fdx_kdtree_tune(1, (0, -1, 300)[it % 5 % 3])          # subtrees on a contiguous copy: never, from 65536, from 300 points
            lib.fdx_kdtree_set_threads(int(rs.randint(2, 12)))
            got, order = np.empty((n, kk), dtype=np.int64), np.empty(n, dtype=np.int64)
            _lib.check(lib.fdx_ckdtree_knn(_lib.ptr_f64(coords), n, dim, kk, got.ctypes.data, order.ctypes.data))
            tree = cKDTree(coords)
            assert np.array_equal(order, tree.indices), (it, n, dim, rng)
            assert np.array_equal(got, tree.query(coords, k=kk)[1]), (it, n, dim, rng)
    finally:
        lib.fdx_kdtree_tune(0, 0)
        lib.fdx_kdtree_tune(1, -1)
        lib.fdx_kdtree_set_threads(0)
    # the split the tree reports for the smallest such node
    c = np.array([3.0] * 12 + [4.0] * 7).reshape(-1, 1)
    t = cKDTree(c)
    assert t.tree.split == np.nextafter(3.0, np.inf) and t.tree.lesser.children == 12


def test_ckdtree_builds_from_several_threads_share_the_pool():
    """Thread ranks of one process each build the restated tree of their coordinates at the same time: the builds cut their work
    into tasks of ONE pool (csrc/kdtree_order.cpp: KdPool) and run each other's tasks while they wait.  Every tree against scipy."""
    import threading
    from scipy.spatial import cKDTree
    from flashdeconv_amd import _lib
    lib = _lib.load()
    lib.fdx_kdtree_tune(0, 64)
    bad = []

    def work(seed):
        rs = np.random.RandomState(seed)
        for it in range(6):
            n, dim, rng = int(rs.randint(3000, 90000)), int(rs.randint(1, 4)), int(rs.choice([5, 1000, 100000]))
            coords = np.ascontiguousarray(rs.randint(0, rng, size=(n, dim)).astype(np.float64))
            got, order = np.empty((n, 4), dtype=np.int64), np.empty(n, dtype=np.int64)
            _lib.check(lib.fdx_ckdtree_knn(_lib.ptr_f64(coords), n, dim, 4, _lib.ptr_i64(got), _lib.ptr_i64(order)))
            tree = cKDTree(coords)
            if not np.array_equal(order, tree.indices) or not np.array_equal(got, tree.query(coords, k=4)[1]):
                bad.append((seed, it, n, dim, rng))

    try:
        threads = [threading.Thread(target=work, args=(s,)) for s in range(3)]
        for t in threads:
            t.start()
        for t in threads:
            t.join()
    finally:
        lib.fdx_kdtree_tune(0, 0)
    assert not bad, bad


def test_ckdtree_thread_pool_survives_a_fork():
    """The thread pool of the tree build (csrc/kdtree_order.cpp: KdPool) lives in the process that started it: a fork()ed child
    (multiprocessing's default start method) has the pool's object without its threads - and its mutex / condition variable in
    whatever state the parent's workers, just going to sleep, had them in - and must start anew: same tree.  50000 points: above
    the size from which subtrees become pool tasks; the fork follows the parent's build at once."""
    import os
    from flashdeconv_amd import _lib
    lib = _lib.load()
    coords = np.ascontiguousarray(np.random.RandomState(0).rand(50000, 2))

    def lists():
        lib.fdx_kdtree_tune(0, 64)
        lib.fdx_kdtree_set_threads(4)
        got = np.empty((len(coords), 3), dtype=np.int64)
        try:
            _lib.check(lib.fdx_ckdtree_knn(_lib.ptr_f64(coords), len(coords), 2, 3, _lib.ptr_i64(got), None))
        finally:
            lib.fdx_kdtree_tune(0, 0)
            lib.fdx_kdtree_set_threads(0)
        return got

    for _ in range(3):
        want = lists()
        pid = os.fork()
        if pid == 0:
            code = 3
            try:
                import faulthandler
                faulthandler.dump_traceback_later(60, exit=True)         # a hang in the child fails the test instead of stalling it
                code = 0 if np.array_equal(lists(), want) else 4
            finally:
                os._exit(code)
        _, status = os.waitpid(pid, 0)
        assert status == 0, status


@pytest.mark.parametrize("case", ["square", "square_scaled", "hex", "cube3d", "random2d", "random3d", "line", "duplicates",
                                  "square_shuffled", "rect_large", "square_big", "random_big", "square_huge"])
def test_ckdtree_order_restatement_matches_scipy(case, monkeypatch):
    """fdx_ckdtree_knn (csrc/kdtree_order.cpp) is a host restatement of scipy.spatial.cKDTree's build and k-nearest query
    ORDER - what decides the reference's neighbour graph when distances tie exactly (flashdeconv/utils/graph.py:60-63).
    Against scipy itself: the tree's index array and the query result, index for index and in the same order, on lattices
    (every k-th neighbour tied), duplicated points, shuffled spot order and tie-free clouds.  Pure host code: no GPU."""
    from scipy.spatial import cKDTree
    from flashdeconv_amd import _lib
    lib = _lib.load()
    rs = np.random.RandomState(3)
    sq = np.stack(np.meshgrid(np.arange(30.0), np.arange(30.0), indexing="ij"), -1).reshape(-1, 2)
    coords, kk = {
        "square": (sq, 7),
        "square_scaled": (sq * 100.0 + 0.25, 7),
        "hex": (np.array([(c + 0.5 * (r & 1), r * np.sqrt(3.0) / 2.0) for r in range(28) for c in range(28)]), 7),
        "cube3d": (np.stack(np.meshgrid(*[np.arange(11.0)] * 3, indexing="ij"), -1).reshape(-1, 3), 9),
        "random2d": (rs.rand(4000, 2) * 60.0, 7),
        "random3d": (rs.rand(2500, 3) * 9.0, 16),
        "line": (rs.rand(400, 1), 3),
        "duplicates": (np.concatenate([rs.rand(300, 2), rs.rand(100, 2).repeat(3, axis=0)]), 5),
        "square_shuffled": (sq[rs.permutation(len(sq))], 7),
        "rect_large": (np.stack(np.meshgrid(np.arange(211.0), np.arange(97.0), indexing="ij"), -1).reshape(-1, 2), 7),
        # above 32768 points the subtrees are built on threads, above 4096 per node on contiguous (coordinate, index) pairs
        "square_big": (np.stack(np.meshgrid(np.arange(310.0), np.arange(300.0), indexing="ij"), -1).reshape(-1, 2), 7),
        "random_big": (np.round(rs.rand(70000, 2) * 40.0, 1), 7),          # many equal coordinates, a few coincident points
        "square_huge": (np.stack(np.meshgrid(np.arange(640.0), np.arange(500.0), indexing="ij"), -1).reshape(-1, 2), 7),
    }[case]
    coords = np.ascontiguousarray(coords, dtype=np.float64)
    n, dim = coords.shape
    got = np.empty((n, kk), dtype=np.int64)
    order = np.empty(n, dtype=np.int64)
    _lib.check(lib.fdx_ckdtree_knn(_lib.ptr_f64(coords), n, dim, kk, got.ctypes.data, order.ctypes.data))
    tree = cKDTree(coords)
    _, want = tree.query(coords, k=kk)
    assert np.array_equal(order, tree.indices)
    assert np.array_equal(got, want)
    if case == "square_big":          # another thread share / fork depth (FDX_KDTREE_THREADS, FDX_KDTREE_PAR_DEPTH, the setter): same tree
        for env in ({"FDX_KDTREE_THREADS": "3", "FDX_KDTREE_PAR_DEPTH": "2"}, {"FDX_KDTREE_PAR_DEPTH": "0"}, {}):
            for k, v in env.items():
                monkeypatch.setenv(k, v)
            lib.fdx_kdtree_set_threads(0 if env else 2)
            got2, order2 = np.empty_like(got), np.empty_like(order)
            _lib.check(lib.fdx_ckdtree_knn(_lib.ptr_f64(coords), n, dim, kk, got2.ctypes.data, order2.ctypes.data))
            for k in env:
                monkeypatch.delenv(k)
            lib.fdx_kdtree_set_threads(0)
            assert np.array_equal(order2, tree.indices) and np.array_equal(got2, want), env
    # the passes of the large nodes cut into tasks of the thread pool (from 400000 points; here from 64 and from 5000): the same tree
    # ... and with the subtrees built through the index array instead of on a contiguous copy of their points
    for team_min, threads, local_max in ((64, 5, 0), (5000, 0, 2000), (0, 0, 0)) if n > 700 else ():
        lib.fdx_kdtree_tune(0, team_min)
        lib.fdx_kdtree_tune(1, local_max)
        lib.fdx_kdtree_set_threads(threads)
        try:
            got2, order2 = np.empty_like(got), np.empty_like(order)
            _lib.check(lib.fdx_ckdtree_knn(_lib.ptr_f64(coords), n, dim, kk, got2.ctypes.data, order2.ctypes.data))
        finally:
            lib.fdx_kdtree_tune(0, 0)
            lib.fdx_kdtree_tune(1, -1)
            lib.fdx_kdtree_set_threads(0)
        assert np.array_equal(order2, tree.indices) and np.array_equal(got2, want), team_min
    # and the adjacency the reference builds from it (utils/graph.py:66-81)
    from flashdeconv_amd.utils.graph import ckdtree_knn_adjacency
    import fdx_oracle as orc
    A, B = ckdtree_knn_adjacency(coords, kk - 1), orc.knn_graph_kdtree(coords, kk - 1)
    assert np.array_equal(A.indptr, B.indptr) and np.array_equal(A.indices, B.indices)


def test_runtime_switch_registry_matches_the_sources_and_the_tests():
    """csrc/fdx_env.cpp: every runtime switch the sources read is in the registry (at most 25 of them), nothing else reads the
    environment with getenv, and every registered switch is exercised by a test or by bench.py / a tool (the verdict's rule:
    a switch nobody runs is a code path nobody tests).  Experiment switches (exp_env) exist only in -DFDX_EXPERIMENT builds."""
    import glob
    import re
    from flashdeconv_amd import _lib
    reg = dict(_lib.runtime_switches())
    assert 0 < len(reg) <= 25
    used, raw = set(), []
    for f in glob.glob(os.path.join(ROOT, "flashdeconv_amd", "csrc", "*.cpp")) + glob.glob(os.path.join(ROOT, "flashdeconv_amd", "csrc", "*.h")):
        src = open(f).read()
        if os.path.basename(f) not in ("fdx_env.cpp", "fdx_env.h"):
            raw += [(os.path.basename(f), m) for m in re.findall(r'(?<![A-Za-z_:])getenv\("(FDX_[A-Z0-9_]+)"\)', src)]
        if os.path.basename(f) != "fdx_env.h":                       # (its header comment spells the call)
            used |= set(re.findall(r'(?<!exp_)env\("(FDX_[A-Z0-9_]+)"\)', src))
    assert not raw, raw
    assert used <= set(reg), used - set(reg)
    assert set(reg) <= used, set(reg) - used                         # no dead entries either
    where = ""
    for f in glob.glob(os.path.join(ROOT, "tests", "*.py")) + glob.glob(os.path.join(ROOT, "tools", "*.py")) + [os.path.join(ROOT, "bench.py")] + \
            glob.glob(os.path.join(ROOT, "flashdeconv_amd", "*.py")) + glob.glob(os.path.join(ROOT, "flashdeconv_amd", "*", "*.py")):
        where += open(f).read()
    missing = [name for name in reg if name not in where]
    assert not missing, missing


def test_hvg_ranking_restatement_equals_numpy():
    """csrc/hvg_rank.cpp (the host tail of select_hvg, utils/genes.py:104-145, with numpy's arithmetic) against the numpy form of
    the same - the reference's own operations - on random moment vectors: sizes 1 ... 20000, zero means, NaN means, many exactly
    equal values (ties across the cut fall back to numpy by design), every n_top regime.  Pure host code: no GPU."""
    from flashdeconv_amd.utils import genes
    rs = np.random.RandomState(0)
    for trial in range(600):
        G = int(rs.choice([1, 2, 3, 5, 8, 17, 40, 130, 260, 1000, 5000, 20000], p=[.03, .03, .03, .05, .05, .08, .1, .15, .15, .15, .13, .05]))
        mean = np.abs(rs.randn(G)) * rs.choice([0.01, 0.3, 2.0])
        var = np.abs(rs.randn(G)) * 0.2
        mode = trial % 5
        if mode == 1:
            mean[rs.rand(G) < 0.3] = 0
        if mode == 2:
            mean, var = np.round(mean, 1), np.round(var, 1)
        if mode == 3 and G > 3:
            var[rs.randint(0, G, size=max(1, G // 10))] = var[0]
        if mode == 4 and G > 4:
            mean[rs.randint(0, G)] = np.nan
        n_top = int(rs.choice([0, 1, 2, G // 3 + 1, max(G - 1, 1), G, G + 5, 2000]))
        got = genes._hvg_from_moments(mean, var, n_top, 0.0125, 3.0, 0.5)
        want = genes._hvg_from_moments_numpy(mean, var, n_top, 0.0125, 3.0, 0.5)
        assert np.array_equal(got, want), (trial, G, n_top, mode)
