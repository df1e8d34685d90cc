// C-ABI layer of libfdx.so (see include/fdx.h for the contract and the reference lines each entry replaces).
#include "../../include/fdx.h"

#include <algorithm>
#include <cmath>
#include <cstring>
#include <vector>

#include "fdx_graph.h"
#include "graph_build.h"
#include "fdx_internal.h"
#include "fdx_kernels.h"
#include "sketch_plan.h"
#include "solver.h"

namespace fdx {
static thread_local std::string g_last_error;
void set_error(const std::string& msg) { g_last_error = msg; }
std::string get_error() { return g_last_error; }
int fail(int code, const std::string& msg) {
    g_last_error = msg;
    // An error return unwinds through DevBuf destructors, which hand their blocks back to the pool while kernels queued
    // by this call may still be running on the caller's or the library's side streams.  Errors are rare: drain the device
    // here, so that no block is ever recycled under a kernel in flight (argument errors are raised before any launch, but
    // a drain on an idle device costs microseconds).
    (void)hipDeviceSynchronize();
    (void)hipGetLastError();
    return code;
}
}  // namespace fdx

using namespace fdx;

extern "C" {

int fdx_version(void) { return 200; }  // 0.2.0

const char* fdx_last_error(void) { return g_last_error.c_str(); }

int fdx_device_count(int* count) {
    FDX_REQUIRE(count != nullptr, "fdx_device_count: null output");
    int c = 0;
    hipError_t e = hipGetDeviceCount(&c);
    if (e != hipSuccess) {
        *count = 0;
        return fail(FDX_ERR_HIP, std::string("hipGetDeviceCount: ") + hipGetErrorString(e));
    }
    *count = c;
    return 0;
}

int fdx_set_device(int device) {
    FDX_HIP(hipSetDevice(device));
    return 0;
}

int fdx_device_name(char* buf, int buflen) {
    FDX_REQUIRE(buf != nullptr && buflen > 0, "fdx_device_name: bad buffer");
    int dev = 0;
    FDX_HIP(hipGetDevice(&dev));
    hipDeviceProp_t prop;
    FDX_HIP(hipGetDeviceProperties(&prop, dev));
    std::snprintf(buf, (size_t)buflen, "%s (%s, %d CUs)", prop.name, prop.gcnArchName, prop.multiProcessorCount);
    return 0;
}

int fdx_malloc(void** dev_ptr, size_t bytes) {
    FDX_REQUIRE(dev_ptr != nullptr, "fdx_malloc: null output");
    size_t cap = 0;
    FDX_TRY(pool_alloc(bytes ? bytes : 8, dev_ptr, &cap));   // served from the caching pool (pool.cpp)
    return 0;
}

int fdx_free(void* dev_ptr) {
    if (dev_ptr) pool_free(dev_ptr, 0);
    return 0;
}

int fdx_trim(void) {
    pool_trim();
    return 0;
}

int fdx_memcpy_h2d(void* dev_dst, const void* host_src, size_t bytes, void* stream) {
    if (bytes == 0) return 0;
    FDX_TRY(copy_h2d(dev_dst, host_src, bytes, (hipStream_t)stream));
    FDX_HIP(hipStreamSynchronize((hipStream_t)stream));
    return 0;
}

int fdx_memcpy_d2h(void* host_dst, const void* dev_src, size_t bytes, void* stream) {
    if (bytes == 0) return 0;
    return copy_d2h(host_dst, dev_src, bytes, (hipStream_t)stream);
}

int fdx_memset(void* dev_dst, int value, size_t bytes, void* stream) {
    if (bytes == 0) return 0;
    FDX_HIP(hipMemsetAsync(dev_dst, value, bytes, (hipStream_t)stream));
    return 0;
}

int fdx_stream_sync(void* stream) {
    FDX_HIP(hipStreamSynchronize((hipStream_t)stream));
    return 0;
}

// ------------------------------------------------------------------------------------------------ graph
// Host CSR -> device sliced ELL.  The conversion runs on the host because its input is a host matrix handed
// over by the caller (the `A` of bcd_solve); graphs built from coordinates are produced on the device
// (graph_kernels.cpp) and never pass through here.
int fdx_graph_from_csr(const int64_t* indptr, const int64_t* indices, int64_t n, fdx_graph** out) {
    FDX_REQUIRE(out != nullptr, "fdx_graph_from_csr: null output");
    *out = nullptr;
    FDX_REQUIRE(n >= 0 && n < 0x7fffff00LL, "fdx_graph_from_csr: n out of range");
    FDX_REQUIRE(n == 0 || indptr != nullptr, "fdx_graph_from_csr: null indptr");
    const int64_t nnz = n ? indptr[n] : 0;
    FDX_REQUIRE(nnz >= 0 && nnz < 0x7fffff00LL, "fdx_graph_from_csr: nnz out of range");
    FDX_REQUIRE(nnz == 0 || indices != nullptr, "fdx_graph_from_csr: null indices");
    for (int64_t i = 0; i < n; ++i)
        FDX_REQUIRE(indptr[i + 1] >= indptr[i], "fdx_graph_from_csr: indptr must be non-decreasing");
    for (int64_t p = 0; p < nnz; ++p)
        FDX_REQUIRE(indices[p] >= 0 && indices[p] < n, "fdx_graph_from_csr: neighbour index out of range");

    fdx_graph* g = new fdx_graph();
    g->n = n;
    g->n_total = n;
    g->nnz = nnz;
    g->n_slices = (int)((n + 63) / 64);
    std::vector<int> slice_off((size_t)g->n_slices + 1, 0), deg((size_t)n, 0);
    int max_deg = 0;
    for (int s = 0; s < g->n_slices; ++s) {
        int w = 0;
        for (int64_t i = (int64_t)s * 64; i < std::min<int64_t>(n, (int64_t)s * 64 + 64); ++i) {
            const int dgi = (int)(indptr[i + 1] - indptr[i]);
            deg[(size_t)i] = dgi;
            w = std::max(w, dgi);
        }
        max_deg = std::max(max_deg, w);
        slice_off[(size_t)s + 1] = slice_off[(size_t)s] + w;
    }
    g->max_deg = max_deg;
    g->ell_rows = g->n_slices ? slice_off[(size_t)g->n_slices] : 0;
    std::vector<int> ell((size_t)g->ell_rows * 64, (int)n);  // pad = index of the all-zero row
    for (int64_t i = 0; i < n; ++i) {
        const int s = (int)(i >> 6), lane = (int)(i & 63);
        const int64_t b = indptr[i];
        for (int m = 0; m < deg[(size_t)i]; ++m)
            ell[((size_t)slice_off[(size_t)s] + m) * 64 + lane] = (int)indices[b + m];
    }
    int rc = 0;
    if ((rc = g->ell.alloc(ell.size() * sizeof(int))) || (rc = g->slice_off.alloc(slice_off.size() * sizeof(int))) ||
        (rc = g->deg.alloc(deg.size() * sizeof(int)))) {
        delete g;
        return rc;
    }
    // (through pinned staging: the vectors are unmapped when this returns - pool.cpp, copy_h2d)
    if (!ell.empty()) rc = copy_h2d(g->ell.p, ell.data(), ell.size() * sizeof(int), nullptr);
    if (!rc) rc = copy_h2d(g->slice_off.p, slice_off.data(), slice_off.size() * sizeof(int), nullptr);
    if (!rc && !deg.empty()) rc = copy_h2d(g->deg.p, deg.data(), deg.size() * sizeof(int), nullptr);
    if (!rc && hipStreamSynchronize(nullptr) != hipSuccess) rc = fail(FDX_ERR_HIP, "fdx_graph_from_csr upload: synchronisation failed");
    if (rc) {
        delete g;
        return rc;
    }
    *out = g;
    return 0;
}

static int upload_coords(const double* coords, int64_t n, int32_t dim, DevBuf* d) {
    FDX_REQUIRE(dim >= 1 && dim <= 8, "graph: coordinate dimension must be 1 to 8 (radius / grid graphs: 1 to 3)");
    FDX_REQUIRE(n >= 0, "graph: negative n");
    FDX_REQUIRE(n == 0 || coords != nullptr, "graph: null coords");
    FDX_TRY(d->alloc((size_t)n * dim * sizeof(double)));
    if (n) FDX_TRY(copy_h2d(d->p, coords, (size_t)n * dim * sizeof(double), nullptr)); FDX_HIP(hipStreamSynchronize(nullptr));
    return 0;
}

int fdx_graph_build_knn(const double* coords, int64_t n, int32_t dim, int32_t k, fdx_graph** out) {
    FDX_REQUIRE(out != nullptr, "fdx_graph_build_knn: null output");
    *out = nullptr;
    DevBuf dc;
    FDX_TRY(upload_coords(coords, n, dim, &dc));
    fdx_graph* g = new fdx_graph();
    const int rc = graph_build_knn(dc.as<double>(), n, dim, k, g, nullptr);
    if (rc) { delete g; return rc; }
    *out = g;
    return 0;
}

int fdx_graph_build_radius(const double* coords, int64_t n, int32_t dim, double radius, fdx_graph** out) {
    FDX_REQUIRE(out != nullptr, "fdx_graph_build_radius: null output");
    *out = nullptr;
    DevBuf dc;
    FDX_TRY(upload_coords(coords, n, dim, &dc));
    fdx_graph* g = new fdx_graph();
    const int rc = graph_build_radius(dc.as<double>(), n, dim, radius, 0, n, g, nullptr);
    if (rc) { delete g; return rc; }
    *out = g;
    return 0;
}

int fdx_nearest_distance(const double* coords, int64_t n, int32_t dim, double* dist_out) {
    FDX_REQUIRE(dist_out != nullptr, "fdx_nearest_distance: null output");
    DevBuf dc, dd;
    FDX_TRY(upload_coords(coords, n, dim, &dc));
    FDX_TRY(dd.alloc((size_t)n * sizeof(double)));
    FDX_TRY(graph_nearest_distance(dc.as<double>(), n, dim, dd.as<double>(), nullptr));
    FDX_TRY(copy_d2h(dist_out, dd.p, (size_t)n * sizeof(double), nullptr));
    return 0;
}

int fdx_graph_export_csr(const fdx_graph* g, int64_t* indptr, int32_t* indices) {
    FDX_TRY(fdx::graph_meta_sync(g));
    FDX_REQUIRE(g != nullptr && indptr != nullptr, "fdx_graph_export_csr: null argument");
    FDX_REQUIRE(g->nnz == 0 || indices != nullptr, "fdx_graph_export_csr: null indices");
    DevBuf dp, di;
    FDX_TRY(dp.alloc((size_t)(g->n + 1) * 8));
    FDX_TRY(di.alloc((size_t)std::max<long long>(g->nnz, 1) * 4));
    FDX_HIP(hipMemset(dp.p, 0, dp.bytes));
    FDX_TRY(graph_export_csr(g, dp.as<long long>(), di.as<int>(), nullptr));
    FDX_TRY(copy_d2h(indptr, dp.p, (size_t)(g->n + 1) * 8, nullptr));
    if (g->nnz) FDX_TRY(copy_d2h(indices, di.p, (size_t)g->nnz * 4, nullptr));
    return 0;
}

int fdx_graph_destroy(fdx_graph* g) {
    delete g;
    return 0;
}

int fdx_graph_info(const fdx_graph* g, int64_t* n, int64_t* nnz, int32_t* max_deg) {
    FDX_TRY(fdx::graph_meta_sync(g));
    FDX_REQUIRE(g != nullptr, "fdx_graph_info: null graph");
    if (n) *n = g->n;
    if (nnz) *nnz = g->nnz;
    if (max_deg) *max_deg = g->max_deg;
    return 0;
}

int fdx_graph_knn_ties(const fdx_graph* g, int64_t* ties) {
    FDX_TRY(fdx::graph_meta_sync(g));
    FDX_REQUIRE(g != nullptr && ties != nullptr, "fdx_graph_knn_ties: null argument");
    *ties = g->knn_ties;
    return 0;
}

// ----------------------------------------------------------------------------------------------- sketch
static size_t dtype_size(int dtype) { return dtype == FDX_F32 ? 4 : 8; }

int fdx_sketch(const void* Y, int32_t dtype, int64_t n, int32_t G, const int64_t* col_ptr, const int32_t* gene_idx,
               const double* weight, int32_t d, int32_t mode, double* Ys_out) {
    FDX_REQUIRE(dtype == FDX_F32 || dtype == FDX_F64, "fdx_sketch: dtype must be FDX_F32 or FDX_F64");
    FDX_REQUIRE(n >= 0 && G > 0 && d > 0, "fdx_sketch: bad shape");
    if (n == 0) return 0;
    FDX_REQUIRE(Y && Ys_out, "fdx_sketch: null array");
    hipStream_t st = nullptr;
    SketchPlan plan;
    FDX_TRY(plan.build((const long long*)col_ptr, gene_idx, weight, G, d, st));
    DevBuf dY, dYs;
    const size_t ybytes = (size_t)n * G * dtype_size(dtype);
    FDX_TRY(dY.alloc(ybytes));
    FDX_TRY(dYs.alloc((size_t)n * d * sizeof(double)));
    FDX_TRY(copy_h2d(dY.p, Y, ybytes, st));
    FDX_TRY(launch_sketch_rows(dY.p, dtype, G, nullptr, n, G, d, mode, plan.dev(), dYs.as<double>(), d, nullptr, st));
    FDX_TRY(copy_d2h(Ys_out, dYs.p, (size_t)n * d * sizeof(double), st));
    FDX_HIP(hipStreamSynchronize(st));
    return 0;
}

int fdx_column_sums(const void* Y, int32_t dtype, int64_t n, int32_t G, double* sums_out) {
    FDX_REQUIRE(dtype == FDX_F32 || dtype == FDX_F64, "fdx_column_sums: dtype must be FDX_F32 or FDX_F64");
    FDX_REQUIRE(n >= 0 && G > 0 && sums_out, "fdx_column_sums: bad arguments");
    hipStream_t st = nullptr;
    DevBuf dY, dPart, dOut;
    const size_t ybytes = (size_t)n * G * dtype_size(dtype);
    FDX_TRY(dY.alloc(ybytes));
    FDX_TRY(dPart.alloc((size_t)column_sums_parts(n) * G * sizeof(double)));
    FDX_TRY(dOut.alloc((size_t)G * sizeof(double)));
    if (n) FDX_TRY(copy_h2d(dY.p, Y, ybytes, st));
    FDX_TRY(launch_column_sums(dY.p, dtype, G, n, G, dPart.as<double>(), dOut.as<double>(), st));
    FDX_TRY(copy_d2h(sums_out, dOut.p, (size_t)G * sizeof(double), st));
    FDX_HIP(hipStreamSynchronize(st));
    return 0;
}

// ----------------------------------------------------------------------------------------------- solver
int fdx_bcd_solve(const fdx_graph* g, const double* Y_sketch, const double* X_sketch, int64_t n, int32_t d, int32_t K,
                  double lambda, double rho, int32_t max_iter, double tol, int32_t verbose, double* beta_out,
                  double* objectives_out, double* rel_changes_out, fdx_solve_info* info) {
    FDX_TRY(fdx::graph_meta_sync(g));
    FDX_REQUIRE(info != nullptr, "fdx_bcd_solve: null info");
    std::memset(info, 0, sizeof(*info));
    FDX_REQUIRE(n >= 0 && K >= 0 && d >= 0, "fdx_bcd_solve: negative size");
    FDX_REQUIRE(max_iter >= 0, "fdx_bcd_solve: max_iter must be non-negative");
    if (n == 0 || K == 0) {  // core/solver.py:334-343
        info->converged = 1;
        return 0;
    }
    FDX_REQUIRE(g != nullptr, "fdx_bcd_solve: null graph");
    FDX_REQUIRE(g->n == n, "fdx_bcd_solve: graph size does not match n");
    FDX_REQUIRE(d > 0, "fdx_bcd_solve: sketch_dim must be positive");
    FDX_REQUIRE(Y_sketch && X_sketch && beta_out, "fdx_bcd_solve: null array");
    hipStream_t st = nullptr;
    hipEvent_t e0 = nullptr, e1 = nullptr;
    FDX_HIP(hipEventCreate(&e0));
    FDX_HIP(hipEventCreate(&e1));
    FDX_HIP(hipEventRecord(e0, st));

    const long long ld = round_up(n + 1, 64);
    DevBuf dY, dX, dH, dG, dB0, dB1, dPart, dSum, dOut;
    FDX_TRY(dY.alloc((size_t)n * d * sizeof(double)));
    FDX_TRY(dX.alloc((size_t)K * d * sizeof(double)));
    const int KP = solver_padded_K(K);   // 65 - 128 cell types: planes of the next instantiated sweep, pad types all zero
    DevBuf dGp;
    FDX_TRY(dH.alloc((size_t)KP * ld * sizeof(double)));
    FDX_TRY(dG.alloc((size_t)K * K * sizeof(double)));
    FDX_TRY(dB0.alloc((size_t)KP * ld * sizeof(double)));
    FDX_TRY(dB1.alloc((size_t)KP * ld * sizeof(double)));
    FDX_TRY(dPart.alloc((size_t)xyt_partials_count(n) * sizeof(double)));
    FDX_TRY(dSum.alloc(sizeof(double)));
    FDX_TRY(dOut.alloc((size_t)n * K * sizeof(double)));
    FDX_TRY(copy_h2d(dY.p, Y_sketch, (size_t)n * d * sizeof(double), st));
    FDX_TRY(copy_h2d(dX.p, X_sketch, (size_t)K * d * sizeof(double), st));
    FDX_HIP(hipMemsetAsync(dH.p, 0, dH.bytes, st));
    // XtX = Xs Xs^T (solver.py:346), H = Xs Ys^T (:347), YtY = ||Ys||^2 (:348)
    FDX_TRY(launch_xyt(dX.as<double>(), dX.as<double>(), d, K, d, K, dG.as<double>(), K, nullptr, st));
    FDX_TRY(launch_xyt(dX.as<double>(), dY.as<double>(), d, n, d, K, dH.as<double>(), ld, dPart.as<double>(), st));
    FDX_TRY(launch_sum_partials(dPart.as<double>(), xyt_partials_count(n), dSum.as<double>(), 1, 1, st));
    std::vector<double> G((size_t)K * K);
    double YtY = 0.0;
    FDX_TRY(copy_d2h(G.data(), dG.p, G.size() * sizeof(double), st));
    FDX_TRY(copy_d2h(&YtY, dSum.p, sizeof(double), st));
    FDX_HIP(hipStreamSynchronize(st));
    double diag_mean = 0.0;  // rho <- rho * mean(diag XtX)   (solver.py:359-360)
    for (int k = 0; k < K; ++k) diag_mean += G[(size_t)k * K + k];
    diag_mean /= (double)K;

    if (KP != K) {
        FDX_TRY(dGp.alloc((size_t)KP * KP * sizeof(double)));
        FDX_TRY(solver_pad_square(dG.as<double>(), K, dGp.as<double>(), KP, st));
    }
    SolveProblem p;
    p.graph = g; p.H = dH.as<double>(); p.ldh = ld; p.XtX = KP != K ? dGp.as<double>() : dG.as<double>();
    p.beta[0] = dB0.as<double>(); p.beta[1] = dB1.as<double>(); p.ld = ld; p.K = KP; p.K_real = K; p.YtY = YtY;
    p.lambda = lambda; p.rho_eff = rho * diag_mean; p.max_iter = max_iter; p.tol = tol; p.verbose = verbose;
    SolveResult r;
    FDX_TRY(solver_run(p, &r, st));
    FDX_TRY(launch_normalize_export(p.beta[r.result_buffer], ld, g->identity_order ? nullptr : g->perm.as<int>(), (int)n,
                                    g->n_slices, K, dOut.as<double>(), nullptr, st));
    FDX_TRY(copy_d2h(beta_out, dOut.p, (size_t)n * K * sizeof(double), st));
    FDX_HIP(hipEventRecord(e1, st));
    FDX_HIP(hipStreamSynchronize(st));
    float ms = 0.f;
    FDX_HIP(hipEventElapsedTime(&ms, e0, e1));
    (void)hipEventDestroy(e0);
    (void)hipEventDestroy(e1);
    info->converged = r.converged;
    info->n_iterations = r.n_iterations;
    info->final_objective = r.final_objective;
    info->final_change = r.final_change;
    info->n_objectives = (int32_t)r.objectives.size();
    info->sweep_ms = r.sweep_ms;
    info->total_ms = ms;
    if (objectives_out)
        for (size_t t = 0; t < r.objectives.size(); ++t) objectives_out[t] = r.objectives[t];
    if (rel_changes_out)
        for (size_t t = 0; t < r.rel_changes.size(); ++t) rel_changes_out[t] = r.rel_changes[t];
    return 0;
}

}  // extern "C"

// ---- function-level seams of core/solver.py the reference's tests import (tests/test_solver.py:7-14) -----------------------
// precompute_gram_matrix (core/solver.py:187-201) and precompute_XtY (:204-223): XtX = Xs Xs^T, H = Xs Ys^T (K, n) row-major.
extern "C" int fdx_gram_xty(const double* X_sketch, const double* Y_sketch, int64_t n, int32_t d, int32_t K, double* XtX_out,
                            double* H_out) {
    FDX_REQUIRE(X_sketch && K > 0 && d > 0 && n >= 0, "fdx_gram_xty: bad arguments");
    FDX_REQUIRE(XtX_out || H_out, "fdx_gram_xty: nothing to compute");
    FDX_REQUIRE(!H_out || n == 0 || Y_sketch, "fdx_gram_xty: null Y_sketch");
    hipStream_t st = nullptr;
    DevBuf dX, dY, dG, dH;
    FDX_TRY(dX.alloc((size_t)K * d * sizeof(double)));
    FDX_TRY(copy_h2d(dX.p, X_sketch, (size_t)K * d * sizeof(double), st));
    if (XtX_out) {
        FDX_TRY(dG.alloc((size_t)K * K * sizeof(double)));
        FDX_TRY(launch_xyt(dX.as<double>(), dX.as<double>(), d, K, d, K, dG.as<double>(), K, nullptr, st));
        FDX_TRY(copy_d2h(XtX_out, dG.p, (size_t)K * K * sizeof(double), st));
    }
    if (H_out && n > 0) {
        const long long ld = round_up(n, 64);
        FDX_TRY(dY.alloc((size_t)n * d * sizeof(double)));
        FDX_TRY(dH.alloc((size_t)K * ld * sizeof(double)));
        FDX_TRY(copy_h2d(dY.p, Y_sketch, (size_t)n * d * sizeof(double), st));
        FDX_TRY(launch_xyt(dX.as<double>(), dY.as<double>(), d, n, d, K, dH.as<double>(), ld, nullptr, st));
        FDX_HIP(hipMemcpy2DAsync(H_out, (size_t)n * sizeof(double), dH.p, (size_t)ld * sizeof(double), (size_t)n * sizeof(double),
                                 (size_t)K, hipMemcpyDeviceToHost, st));
    }
    FDX_HIP(hipStreamSynchronize(st));
    return 0;
}

// compute_objective (core/solver.py:226-284): 0.5 (YtY - 2 <beta, H^T> + <beta^T beta, XtX>) + 0.5 lambda <beta, L beta>
// + rho |beta|_1 with L = D - A of the graph's structure.  beta (n, K) row-major, H (K, n) row-major, both on the host.
extern "C" int fdx_objective(const fdx_graph* g, const double* beta, const double* H, const double* XtX, int64_t n, int32_t K,
                             double YtY, double lambda, double rho, double* obj_out) {
    FDX_TRY(fdx::graph_meta_sync(g));
    FDX_REQUIRE(g && beta && H && XtX && obj_out && n > 0 && K > 0, "fdx_objective: bad arguments");
    FDX_REQUIRE(g->n == n && g->identity_order, "fdx_objective: needs a graph over the same n spots in the caller's order (fdx_graph_from_csr)");
    hipStream_t st = nullptr;
    const long long ld = round_up(n + 1, 64);
    std::vector<double> bt((size_t)K * ld, 0.0), ht((size_t)K * ld, 0.0);
    for (long long i = 0; i < n; ++i)
        for (int k = 0; k < K; ++k) {
            bt[(size_t)k * ld + i] = beta[(size_t)i * K + k];
            ht[(size_t)k * ld + i] = H[(size_t)k * n + i];
        }
    DevBuf dB, dH, dG, dPart, dOut;
    FDX_TRY(dB.alloc(bt.size() * sizeof(double)));
    FDX_TRY(dH.alloc(ht.size() * sizeof(double)));
    FDX_TRY(dG.alloc((size_t)K * K * sizeof(double)));
    FDX_TRY(dPart.alloc((size_t)std::max(objective_partials_count(g->n_slices), g->n_tiles) * 4 * sizeof(double)));
    FDX_TRY(dOut.alloc(4 * sizeof(double)));
    FDX_TRY(copy_h2d(dB.p, bt.data(), bt.size() * sizeof(double), st));
    FDX_TRY(copy_h2d(dH.p, ht.data(), ht.size() * sizeof(double), st));
    FDX_TRY(copy_h2d(dG.p, XtX, (size_t)K * K * sizeof(double), st));
    if (K <= FDX_MAX_K_FAST || true)
        FDX_TRY(solver_objective(*g, dB.as<double>(), ld, dH.as<double>(), ld, dG.as<double>(), K, YtY, lambda, rho, dPart.as<double>(),
                                 dOut.as<double>(), obj_out, st));
    return 0;
}
