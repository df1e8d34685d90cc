// Device-pointer C-ABI building blocks used by the spot-sharded multi-GPU driver (see include/fdx.h).
#include "fdx_env.h"
#include <algorithm>
#include <cmath>
#include <memory>
#include <vector>

#include "fdx_graph.h"
#include "fdx_internal.h"
#include "fdx_kernels.h"
#include "graph_build.h"
#include "prepare.h"
#include "sketch_plan.h"
#include "solver.h"

using namespace fdx;

extern "C" {

int fdx_graph_build_dev(const double* coords_dev, int64_t n, int32_t dim, int32_t method, int32_t k, double radius,
                        void* stream, fdx_graph** out) {
    PoolStream pool_stream((hipStream_t)stream);
    FDX_REQUIRE(out != nullptr, "fdx_graph_build_dev: null output");
    *out = nullptr;
    FDX_REQUIRE(n == 0 || coords_dev != nullptr, "fdx_graph_build_dev: null coords");
    fdx_graph* g = new fdx_graph();
    if (hipEventCreate(&g->begin_event) == hipSuccess && hipEventRecord(g->begin_event, (hipStream_t)stream) == hipSuccess)
        g->begin_stream = (hipStream_t)stream;                  // the fit's prologue timer (fdx_fit_info.prologue_ms) starts here
    int rc;
    if (method == FDX_GRAPH_KNN) rc = graph_build_knn(coords_dev, n, dim, k, g, (hipStream_t)stream);
    else if (method == FDX_GRAPH_RADIUS) rc = graph_build_radius(coords_dev, n, dim, radius, 0, n, g, (hipStream_t)stream);
    else rc = fail(FDX_ERR_INVALID, "fdx_graph_build_dev: unknown method");
    if (rc) { delete g; return rc; }
    *out = g;
    return 0;
}

int fdx_side_stream(void** stream_out) {
    FDX_REQUIRE(stream_out != nullptr, "fdx_side_stream: null output");
    *stream_out = (void*)library_side_stream();
    FDX_REQUIRE(*stream_out != nullptr, "fdx_side_stream: the side stream could not be created");
    return 0;
}

int fdx_stream_wait_stream(void* waiter, void* producer) {
    hipEvent_t ev = nullptr;
    FDX_HIP(hipEventCreateWithFlags(&ev, hipEventDisableTiming));
    hipError_t e = hipEventRecord(ev, (hipStream_t)producer);
    if (e == hipSuccess) e = hipStreamWaitEvent((hipStream_t)waiter, ev, 0);
    (void)hipEventDestroy(ev);                      // released when the wait has been satisfied
    if (e != hipSuccess) return fail(FDX_ERR_HIP, hipGetErrorString(e));
    return 0;
}

int fdx_graph_build_radius_rows_dev(const double* coords_dev, int64_t n, int32_t dim, double radius, int64_t lo, int64_t hi,
                                    void* stream, fdx_graph** out) {
    PoolStream pool_stream((hipStream_t)stream);
    FDX_REQUIRE(out != nullptr, "fdx_graph_build_radius_rows_dev: null output");
    *out = nullptr;
    FDX_REQUIRE(n == 0 || coords_dev != nullptr, "fdx_graph_build_radius_rows_dev: null coords");
    fdx_graph* g = new fdx_graph();
    const int rc = graph_build_radius(coords_dev, n, dim, radius, lo, hi, g, (hipStream_t)stream);
    if (rc) { delete g; return rc; }
    *out = g;
    return 0;
}

int fdx_graph_knn_lists_dev(const double* coords_dev, int64_t n, int32_t dim, int32_t k, int64_t lo, int64_t hi,
                            int32_t* nbr_dev, int32_t* cnt_dev, void* stream, fdx_graph_plan** plan) {
    PoolStream pool_stream((hipStream_t)stream);
    FDX_REQUIRE(plan != nullptr, "fdx_graph_knn_lists_dev: null output");
    *plan = nullptr;
    FDX_REQUIRE(coords_dev && nbr_dev && cnt_dev, "fdx_graph_knn_lists_dev: null argument");
    return graph_knn_lists(coords_dev, n, dim, k, lo, hi, nbr_dev, cnt_dev, plan, (hipStream_t)stream);
}

int fdx_graph_knn_lists_band_dev(const double* coords_dev, int64_t n, int32_t dim, int32_t k, int64_t lo, int64_t hi,
                                 int32_t* nbr_dev, int32_t* cnt_dev, void* stream, fdx_graph_plan** plan) {
    PoolStream pool_stream((hipStream_t)stream);
    FDX_REQUIRE(plan != nullptr, "fdx_graph_knn_lists_band_dev: null output");
    *plan = nullptr;
    FDX_REQUIRE(coords_dev && nbr_dev && cnt_dev, "fdx_graph_knn_lists_band_dev: null argument");
    return graph_knn_lists(coords_dev, n, dim, k, lo, hi, nbr_dev, cnt_dev, plan, (hipStream_t)stream, true);
}

int fdx_graph_plan_order_dev(const fdx_graph_plan* plan, int32_t* perm_out_dev, int32_t* rank_out_dev, void* stream) {
    FDX_REQUIRE(plan != nullptr, "fdx_graph_plan_order_dev: null plan");
    return graph_plan_order(plan, perm_out_dev, rank_out_dev, (hipStream_t)stream);
}

int fdx_graph_plan_set_lists_dev(fdx_graph_plan* plan, const int64_t* ids_host, const int64_t* rows_host, int64_t n_rows,
                                 int32_t* nbr_dev, int32_t* cnt_dev, void* stream) {
    PoolStream pool_stream((hipStream_t)stream);
    FDX_REQUIRE(plan && nbr_dev && cnt_dev && (n_rows == 0 || ids_host), "fdx_graph_plan_set_lists_dev: null argument");
    return graph_plan_set_lists(plan, (const long long*)ids_host, (const long long*)rows_host, n_rows, nbr_dev, cnt_dev, (hipStream_t)stream);
}

int fdx_graph_plan_set_ckdtree_lists_dev(fdx_graph_plan* plan, const double* coords_host, const double* coords_dev, int64_t n,
                                         int32_t dim, const int64_t* rows_host, int64_t n_rows, int32_t* nbr_dev, int32_t* cnt_dev,
                                         void* stream) {
    hipStream_t st = (hipStream_t)stream;
    PoolStream pool_stream(st);
    FDX_REQUIRE(plan && coords_dev && nbr_dev && cnt_dev, "fdx_graph_plan_set_ckdtree_lists_dev: null argument");   // coords_host NULL: fetched from coords_dev
    FDX_REQUIRE(n >= 1 && dim >= 1 && dim <= 8 && n_rows >= 0 && (n_rows == 0 || rows_host),
                "fdx_graph_plan_set_ckdtree_lists_dev: bad arguments (1 to 8 coordinates)");
    const int kk = graph_plan_kk(plan);
    const long long nq = rows_host ? n_rows : n;
    if (rows_host)
        for (int64_t j = 0; j < n_rows; ++j) FDX_REQUIRE(rows_host[j] >= 0 && rows_host[j] < n, "fdx_graph_plan_set_ckdtree_lists_dev: row index out of range");
    if (nq == 0) return graph_plan_lists_replaced(plan);
    DevBuf d_ids;
    FDX_TRY(d_ids.alloc((size_t)nq * kk * 8));
    try {
        FDX_TRY(ckdtree_lists_device(coords_host, coords_dev, n, dim, kk, (const long long*)rows_host, nq, d_ids.as<long long>(), st));
    } catch (...) {
        return fail(FDX_ERR_INVALID, "fdx_graph_plan_set_ckdtree_lists_dev: out of memory");
    }
    return graph_plan_set_lists_device(plan, d_ids.as<long long>(), (const long long*)rows_host, nq, nbr_dev, cnt_dev, st);
}

int fdx_graph_plan_lists_replaced(fdx_graph_plan* plan) {
    FDX_REQUIRE(plan != nullptr, "fdx_graph_plan_lists_replaced: null plan");
    return graph_plan_lists_replaced(plan);
}

int fdx_graph_knn_far(const fdx_graph* g, int32_t* far) {
    FDX_REQUIRE(g != nullptr && far != nullptr, "fdx_graph_knn_far: null argument");
    FDX_TRY(fdx::graph_meta_sync(g));
    *far = g->knn_far;
    return 0;
}

int fdx_graph_from_knn_lists_dev(fdx_graph_plan* plan, const int32_t* nbr_dev, const int32_t* cnt_dev, int64_t lo, int64_t hi,
                                 void* stream, fdx_graph** out) {
    PoolStream pool_stream((hipStream_t)stream);
    FDX_REQUIRE(plan != nullptr, "fdx_graph_from_knn_lists_dev: null plan");
    int rc = 0;
    fdx_graph* g = nullptr;
    if (!out || !nbr_dev || !cnt_dev) {
        rc = fail(FDX_ERR_INVALID, "fdx_graph_from_knn_lists_dev: null argument");
    } else {
        *out = nullptr;
        g = new fdx_graph();
        rc = graph_from_knn_lists(plan, nbr_dev, cnt_dev, lo, hi, g, (hipStream_t)stream);
    }
    graph_plan_destroy(plan);                      // consumed either way
    if (rc) { delete g; return rc; }
    *out = g;
    return 0;
}

int fdx_graph_perm_dev(const fdx_graph* g, int32_t* perm_out_dev, void* stream) {
    PoolStream pool_stream((hipStream_t)stream);
    if (!(g && g->shard_pending)) FDX_TRY(fdx::graph_meta_sync(g));   // a queued shard build: the copy below is ordered behind it on the stream
    FDX_REQUIRE(g && (g->n == 0 || perm_out_dev), "fdx_graph_perm_dev: null argument");
    return graph_copy_perm(g, perm_out_dev, (hipStream_t)stream);
}

int fdx_graph_localize(const fdx_graph* full, int32_t n_ranks, const int64_t* bounds, int32_t my_rank, void* stream,
                       fdx_graph** local) {
    PoolStream pool_stream((hipStream_t)stream);
    FDX_TRY(fdx::graph_meta_sync(full));
    FDX_REQUIRE(full && bounds && local, "fdx_graph_localize: null argument");
    FDX_REQUIRE(n_ranks >= 1 && my_rank >= 0 && my_rank < n_ranks, "fdx_graph_localize: bad rank");
    FDX_REQUIRE(bounds[0] == 0 && bounds[n_ranks] == full->n, "fdx_graph_localize: bounds must cover [0, n]");
    for (int r = 0; r < n_ranks; ++r) {
        FDX_REQUIRE(bounds[r + 1] >= bounds[r], "fdx_graph_localize: bounds must be non-decreasing");
        FDX_REQUIRE(bounds[r] % 256 == 0, "fdx_graph_localize: range starts must be multiples of 256");
    }
    *local = nullptr;
    fdx_graph* g = new fdx_graph();
    std::vector<long long> b(bounds, bounds + n_ranks + 1);
    const int rc = graph_localize(full, b[(size_t)my_rank], b[(size_t)my_rank + 1], n_ranks, b.data(), my_rank, g,
                                  (hipStream_t)stream);
    if (rc) { delete g; return rc; }
    *local = g;
    return 0;
}

int fdx_graph_shard_knn_dev(const double* coords_dev, int64_t n, int32_t dim, int32_t k, int32_t n_ranks, const int64_t* bounds,
                            int32_t my_rank, void* stream, fdx_graph** local) {
    PoolStream pool_stream((hipStream_t)stream);
    FDX_REQUIRE(coords_dev && bounds && local, "fdx_graph_shard_knn_dev: null argument");
    FDX_REQUIRE(n_ranks >= 1 && my_rank >= 0 && my_rank < n_ranks, "fdx_graph_shard_knn_dev: bad rank");
    *local = nullptr;
    fdx_graph* g = new fdx_graph();
    std::vector<long long> b(bounds, bounds + n_ranks + 1);
    const int rc = graph_shard_knn(coords_dev, n, dim, k, n_ranks, b.data(), my_rank, g, (hipStream_t)stream);
    if (rc) { delete g; return rc; }
    *local = g;
    return 0;
}

int fdx_graph_shard_status(const fdx_graph* local, int64_t* own_nnz, int64_t* knn_ties, int32_t* far, int32_t* overflow) {
    FDX_REQUIRE(local != nullptr, "fdx_graph_shard_status: null graph");
    FDX_TRY(fdx::graph_meta_sync(local));
    if (own_nnz) *own_nnz = local->nnz;
    if (knn_ties) *knn_ties = local->knn_ties;
    if (far) *far = local->knn_far;
    if (overflow) *overflow = local->shard_overflow;
    return 0;
}

// test hook: the stored neighbour indices of one row (positions of this graph's own order; for a local graph own rows are
// 0..n-1 and halo slots n..n_total-1), as the sweeps read them
int fdx_graph_row_indices(const fdx_graph* g, int64_t row, int32_t* idx_out, int32_t cap, int32_t* deg_out) {
    FDX_REQUIRE(g && idx_out && deg_out && row >= 0 && row < g->n, "fdx_graph_row_indices: bad arguments");
    int so = 0, dg = 0;
    FDX_HIP(hipMemcpy(&so, g->slice_off.as<int>() + (row >> 6), 4, hipMemcpyDeviceToHost));
    FDX_HIP(hipMemcpy(&dg, g->deg.as<int>() + row, 4, hipMemcpyDeviceToHost));
    *deg_out = dg;
    for (int m = 0; m < dg && m < cap; ++m)
        FDX_HIP(hipMemcpy(idx_out + m, g->ell.as<int>() + ((size_t)so + m) * 64 + (row & 63), 4, hipMemcpyDeviceToHost));
    return 0;
}

int fdx_graph_halo_info(const fdx_graph* local, int64_t* n_halo, int32_t* send_counts, int32_t* recv_counts) {
    FDX_REQUIRE(local != nullptr, "fdx_graph_halo_info: null graph");
    FDX_TRY(fdx::graph_meta_sync(local));
    if (n_halo) *n_halo = local->n_total - local->n;
    const size_t R = local->send_off.empty() ? 0 : local->send_off.size() - 1;
    for (size_t r = 0; r < R; ++r) {
        if (send_counts) send_counts[r] = local->send_off[r + 1] - local->send_off[r];
        if (recv_counts) recv_counts[r] = local->recv_off[r + 1] - local->recv_off[r];
    }
    return 0;
}

int fdx_graph_send_indices_dev(const fdx_graph* local, int32_t* idx_out_dev, void* stream) {
    FDX_REQUIRE(local != nullptr, "fdx_graph_send_indices_dev: null graph");
    FDX_TRY(fdx::graph_meta_sync(local));
    const int total = local->send_off.empty() ? 0 : local->send_off.back();
    if (total == 0) return 0;
    FDX_REQUIRE(idx_out_dev != nullptr, "fdx_graph_send_indices_dev: null output");
    FDX_HIP(hipMemcpyAsync(idx_out_dev, local->send_idx.p, (size_t)total * 4, hipMemcpyDeviceToDevice, (hipStream_t)stream));
    return 0;
}

int fdx_prepare_dev(const void* Y_dev, int32_t y_dtype, int64_t n, int32_t G, int64_t ldy, const int32_t* row_map_dev,
                    const double* X, int32_t K, const int32_t* bucket, const double* weight_y, const double* weight_x,
                    int32_t d, int32_t mode_y_in, int32_t mode_x, double* H_out_dev, int64_t ldh, double* XtX_out_dev,
                    double* XtX_out_host, double* YtY_partial_out, void* stream) {
    FDX_REQUIRE(XtX_out_dev != nullptr, "fdx_prepare_dev: null array");
    hipStream_t st = (hipStream_t)stream;
    PoolStream pool_stream(st);
    PrepareJob job;
    // on any return both streams are idle before the job's buffers go back to the pool
    struct Drain { PrepareJob* j; hipStream_t s; ~Drain() { if (j->side) (void)hipStreamSynchronize(j->side); (void)hipStreamSynchronize(s); } } drain{&job, st};
    FDX_TRY(prepare_queue(&job, Y_dev, y_dtype, n, G, ldy, row_map_dev, X, K, bucket, weight_y, weight_x, d, mode_y_in, mode_x,
                          H_out_dev, ldh, XtX_out_host, st));
    // the caller's XtX buffer belongs to the caller's stream: filled there (the stream already waits for the X side)
    FDX_HIP(hipMemcpyAsync(XtX_out_dev, job.dG.p, (size_t)K * K * sizeof(double), hipMemcpyDeviceToDevice, st));
    double yty = 0.0;
    if (n > 0 && job.evSum) FDX_HIP(hipStreamWaitEvent(st, job.evSum, 0));
    if (n > 0) FDX_HIP(hipMemcpyAsync(&yty, job.dSum.p, sizeof(double), hipMemcpyDeviceToHost, st));
    FDX_HIP(hipStreamSynchronize(st));
    if (job.side) FDX_HIP(hipStreamSynchronize(job.side));
    if (YtY_partial_out) *YtY_partial_out = yty;
    return 0;
}

}  // extern "C"

namespace fdx {
int prepare_queue(PrepareJob* job, const void* Y_dev, int y_dtype, long long n, int G, long long ldy, const int* row_map_dev,
                  const double* X, int K, const int* bucket, const double* weight_y, const double* weight_x, int d, int mode_y_in,
                  int mode_x, double* H_out_dev, long long ldh, double* XtX_host, hipStream_t st, const double* X_dev) {
    const int32_t mode_y = mode_y_in & 0xff;
    TileF64Math f64_math((mode_y_in & FDX_PRE_F64_MATH) != 0);
    FDX_REQUIRE(y_dtype == FDX_F32 || y_dtype == FDX_F64, "fdx_prepare_dev: Y dtype must be FDX_F32 or FDX_F64");
    FDX_REQUIRE(n >= 0 && G > 0 && K > 0 && d > 0, "fdx_prepare_dev: bad shape");
    FDX_REQUIRE(X && bucket && weight_y && weight_x && H_out_dev, "fdx_prepare_dev: null array");
    FDX_REQUIRE(ldh >= n && ldy >= G, "fdx_prepare_dev: leading dimension too small");
    // The X side (upload of the signatures - a pageable copy: the host waits for it -, X_sketch, XtX and its copy to the host) runs
    // on the library's side stream: queued on the caller's stream behind a shard plan that is still executing, the upload made
    // the host wait for the whole plan and the launches behind it arrived on an idle device (70 us of a 125k-spot rank's 1.6 ms).
    // The schedules of an Omega are built once per content and device (sketch_plan.cpp: the cache the single-GPU fit uses).
    hipStream_t side = fdx::env("FDX_NO_SIDE_STREAM") ? nullptr : library_side_stream();
    if (side == st) side = nullptr;
    job->side = side;
    const hipStream_t xs = side ? side : st;
    {
        PoolStream pool_xs(xs);
        // a new Omega's tables are uploaded on xs as well (the caller's stream waits for the event below before the sketch)
        FDX_TRY(sketch_plan_cached(bucket, weight_y, G, d, xs, &job->plan_y));
        if (weight_x == weight_y) job->plan_x = job->plan_y;
        else FDX_TRY(sketch_plan_cached(bucket, weight_x, G, d, xs, &job->plan_x));
        FDX_TRY(job->dXs.alloc((size_t)K * d * sizeof(double)));
        FDX_TRY(job->dG.alloc((size_t)K * K * sizeof(double)));
        const void* Xd = X_dev;                                            // already there (the leverage job's copy, complete)
        if (!Xd) {
            FDX_TRY(job->dX.alloc((size_t)K * G * sizeof(double)));
            FDX_TRY(copy_h2d(job->dX.p, X, (size_t)K * G * sizeof(double), xs));
            Xd = job->dX.p;
        }
        FDX_TRY(launch_sketch_rows(Xd, FDX_F64, G, nullptr, K, G, d, mode_x, job->plan_x->dev(), job->dXs.as<double>(), d, nullptr, xs));
        FDX_TRY(launch_xyt(job->dXs.as<double>(), job->dXs.as<double>(), d, K, d, K, job->dG.as<double>(), K, nullptr, xs));
        if (XtX_host)
            FDX_HIP(hipMemcpyAsync(XtX_host, job->dG.p, (size_t)K * K * sizeof(double), hipMemcpyDeviceToHost, xs));
        FDX_HIP(hipEventCreateWithFlags(&job->evX, hipEventDisableTiming));
        FDX_HIP(hipEventRecord(job->evX, xs));
        if (side) FDX_HIP(hipStreamWaitEvent(st, job->evX, 0));
    }
    SketchPlan& plan_y = *job->plan_y;
    if (n > 0) {
        FDX_REQUIRE(Y_dev != nullptr, "fdx_prepare_dev: null Y");
        const long long chunk = std::min<long long>(n, 1LL << 18);
        FDX_TRY(job->dRowSq.alloc((size_t)n * sizeof(double)));
        FDX_TRY(job->dSum.alloc(sizeof(double)));
        // same choice as the single-GPU fit (fit.cpp): shards start on multiples of 256, so the fused kernel's groups of 16
        // spots coincide with those of an unsharded run and H keeps the same bits
        const bool fused = fused_sketch_contract_ok(y_dtype, ldy, Y_dev, G, d, K, mode_y, plan_y.dev());
        if (fused)
            FDX_TRY(launch_sketch_contract(Y_dev, y_dtype, ldy, row_map_dev, n, G, d, mode_y, plan_y.dev(), job->dXs.as<double>(), K,
                                           H_out_dev, ldh, job->dRowSq.as<double>(), st));
        else
            FDX_TRY(job->dYs.alloc((size_t)chunk * d * sizeof(double)));
        for (long long r0 = 0; r0 < n && !fused; r0 += chunk) {
            const long long nr = std::min(chunk, n - r0);
            const unsigned char* ybase = static_cast<const unsigned char*>(Y_dev);
            if (!row_map_dev) ybase += (size_t)r0 * (size_t)ldy * (y_dtype == FDX_F32 ? 4 : 8);
            FDX_TRY(launch_sketch_rows(ybase, y_dtype, ldy, row_map_dev ? row_map_dev + r0 : nullptr, nr, G, d, mode_y,
                                       plan_y.dev(), job->dYs.as<double>(), d, job->dRowSq.as<double>() + r0, st));
            FDX_TRY(launch_xyt(job->dXs.as<double>(), job->dYs.as<double>(), d, nr, d, K, H_out_dev + r0, ldh, nullptr, st));
        }
        // the shard's partial YtY only enters the objective: its reduction goes to the side stream (behind the sketch, beside the
        // first sweep) instead of standing between the sketch and the sweeps
        hipStream_t ys = side ? side : st;
        if (ys != st) {
            hipEvent_t evSk = nullptr;
            FDX_HIP(hipEventCreateWithFlags(&evSk, hipEventDisableTiming));
            const hipError_t e1 = hipEventRecord(evSk, st);
            const hipError_t e2 = e1 == hipSuccess ? hipStreamWaitEvent(ys, evSk, 0) : e1;
            (void)hipEventDestroy(evSk);                              // released once the wait has consumed it
            FDX_HIP(e2);
        }
        FDX_TRY(launch_sum_partials(job->dRowSq.as<double>(), n, job->dSum.as<double>(), 1, 1, ys));
        if (ys != st) {
            FDX_HIP(hipEventCreateWithFlags(&job->evSum, hipEventDisableTiming));
            FDX_HIP(hipEventRecord(job->evSum, ys));
        }
    }
    return 0;
}
}  // namespace fdx

extern "C" {

// The same for a CSR shard (core/deconv.py:181-188 sparse log-CPM rule, core/sketching.py:194-199): the own rows stay
// sparse in HBM; gene_idx selects G of the matrix's columns (NULL = all, in order).
int fdx_prepare_csr_dev(const fdx_csr_view* Y, const int32_t* gene_idx, int32_t G, const double* X, int32_t K,
                        const int32_t* bucket, const double* weight_y, const double* weight_x, int32_t d, int32_t mode_y,
                        int32_t mode_x, double* H_out_dev, int64_t ldh, double* XtX_out_dev, double* XtX_out_host,
                        double* YtY_partial_out, void* stream) {
    FDX_REQUIRE(Y != nullptr && Y->n >= 0 && Y->G > 0, "fdx_prepare_csr_dev: bad matrix");
    FDX_REQUIRE(Y->dtype == FDX_F32 || Y->dtype == FDX_F64, "fdx_prepare_csr_dev: dtype must be FDX_F32 or FDX_F64");
    FDX_REQUIRE(G > 0 && K > 0 && d > 0 && (gene_idx || G == Y->G), "fdx_prepare_csr_dev: bad shape");
    FDX_REQUIRE(X && bucket && weight_y && weight_x && H_out_dev && XtX_out_dev, "fdx_prepare_csr_dev: null array");
    FDX_REQUIRE(mode_y == FDX_PRE_RAW || mode_y == FDX_PRE_LOG_CPM_SPARSE, "fdx_prepare_csr_dev: mode_y must be FDX_PRE_RAW or FDX_PRE_LOG_CPM_SPARSE");
    const long long n = Y->n;
    FDX_REQUIRE(ldh >= n, "fdx_prepare_csr_dev: leading dimension too small");
    hipStream_t st = (hipStream_t)stream;
    PoolStream pool_stream(st);
    const int G_all = Y->G;
    const bool csr_fused = csr_contract_ok(d, K, (G_all + 31) / 32);       // the same choice as fit_impl: sharded = unsharded bits
    CsrSelection sel;
    FDX_TRY(sel.build(gene_idx, G, G_all, bucket, weight_y, d, csr_fused, st, "fdx_prepare_csr_dev"));
    DevBuf dX, dXs, dYs, dRowSq, dSum;
    std::shared_ptr<SketchPlan> plan_x;
    FDX_TRY(sketch_plan_cached(bucket, weight_x, G, d, st, &plan_x));
    FDX_TRY(dX.alloc((size_t)K * G * sizeof(double)));
    FDX_TRY(dXs.alloc((size_t)K * d * sizeof(double)));
    FDX_TRY(copy_h2d(dX.p, X, (size_t)K * G * sizeof(double), st));
    FDX_TRY(launch_sketch_rows(dX.p, FDX_F64, G, nullptr, K, G, d, mode_x, plan_x->dev(), dXs.as<double>(), d, nullptr, st));
    FDX_TRY(launch_xyt(dXs.as<double>(), dXs.as<double>(), d, K, d, K, XtX_out_dev, K, nullptr, st));
    double yty = 0.0;
    if (n > 0) {
        const long long chunk = std::min<long long>(n, 1LL << 18);
        if (!csr_fused) FDX_TRY(dYs.alloc((size_t)chunk * d * sizeof(double)));
        FDX_TRY(dRowSq.alloc((size_t)n * sizeof(double)));
        FDX_TRY(dSum.alloc(sizeof(double)));
        if (csr_fused)
            FDX_TRY(launch_sketch_csr_contract((const long long*)Y->indptr, Y->indices, Y->data, Y->dtype, nullptr, n, d, mode_y, sel,
                                               dXs.as<double>(), K, H_out_dev, ldh, dRowSq.as<double>(), st));
        for (long long r0 = 0; r0 < n && !csr_fused; r0 += chunk) {
            const long long nr = std::min(chunk, n - r0);
            FDX_TRY(launch_sketch_csr((const long long*)Y->indptr, Y->indices, Y->data, Y->dtype, nullptr, r0, nr, d, mode_y, sel.slots.p,
                                      sel.bits.as<unsigned>(), sel.sel_words, dYs.as<double>(), d, dRowSq.as<double>() + r0, st));
            FDX_TRY(launch_xyt(dXs.as<double>(), dYs.as<double>(), d, nr, d, K, H_out_dev + r0, ldh, nullptr, st));
        }
        FDX_TRY(launch_sum_partials(dRowSq.as<double>(), n, dSum.as<double>(), 1, 1, st));
        FDX_HIP(hipMemcpyAsync(&yty, dSum.p, sizeof(double), hipMemcpyDeviceToHost, st));
    }
    if (XtX_out_host)
        FDX_HIP(hipMemcpyAsync(XtX_out_host, XtX_out_dev, (size_t)K * K * sizeof(double), hipMemcpyDeviceToHost, st));
    FDX_HIP(hipStreamSynchronize(st));      // the host tables above are stack objects
    if (YtY_partial_out) *YtY_partial_out = yty;
    return 0;
}

int fdx_init_beta_dev(double* beta_dev, int64_t ld, int64_t n_fill, int32_t K, void* stream) {
    FDX_REQUIRE(beta_dev && ld > 0 && K > 0 && n_fill <= ld, "fdx_init_beta_dev: bad arguments");
    return solver_init_beta(beta_dev, ld, n_fill, K, (hipStream_t)stream);
}

int fdx_bcd_sweep_dev(const fdx_graph* g, const double* H_dev, int64_t ldh, const double* XtX_dev, const double* beta_in,
                      double* beta_out, int64_t ld, int32_t K, double lambda, double rho_eff, double tol, int32_t it,
                      void* stats_dev, double* rel_change_dev, void* stream) {
    FDX_TRY(fdx::graph_meta_sync(g));
    FDX_REQUIRE(g && H_dev && XtX_dev && beta_in && beta_out && stats_dev && rel_change_dev, "fdx_bcd_sweep_dev: null argument");
    FDX_REQUIRE(ld >= g->n_total + 1, "fdx_bcd_sweep_dev: ld must cover own + halo + zero row");
    FDX_REQUIRE(K >= 1 && (sweep_instantiated(K) || K > FDX_MAX_K_PAD),
                "fdx_bcd_sweep_dev: K must be in 1..64, fdx_solver_padded_k of 65..96 cell types, or above 96");
    if (g->n == 0) return 0;
    // above 96 types: the LDS-resident sweep (XtX with its rows padded to 16, prepared per call) or the generic one (per-call
    // scratch) takes the whole shard in one launch, as in fdx_sharded_solve_padded_dev
    DevBuf scratch;
    size_t scratch_ld = 0;
    if (!sweep_instantiated(K)) {
        PoolStream pool_stream((hipStream_t)stream);
        if (sweep_uses_lds(K)) {
            FDX_TRY(scratch.alloc(sweep_lds_pad_doubles(K) * sizeof(double)));
            FDX_TRY(sweep_lds_prepare(XtX_dev, K, scratch.as<double>(), (hipStream_t)stream));
        } else {
            scratch_ld = (size_t)g->n_slices * 64;
            FDX_TRY(scratch.alloc(scratch_ld * 2 * K * sizeof(double)));
        }
    }
    BcdSweepArgs a{};
    a.H = H_dev; a.XtX = XtX_dev; a.beta_in = beta_in; a.beta_out = beta_out;
    a.ell = g->ell.as<int>(); a.slice_off = g->slice_off.as<int>(); a.deg = g->deg.as<int>();
    a.stats = (unsigned long long*)stats_dev; a.rel_change = rel_change_dev;
    a.lambda = lambda; a.rho = rho_eff; a.tol = tol; a.ldh = (int)ldh; a.ld = (int)ld; a.n = (int)g->n;
    a.n_slices = g->n_slices; a.K = K; a.it = it;
    if (g->tiled) {
        a.tiled = 1; a.ell_local = g->ell_local.as<unsigned short>(); a.tile_halo = g->tile_halo.as<int>();
        a.tile_hcnt = g->tile_hcnt.as<int>(); a.n_tiles = g->n_tiles; a.halo_max = g->halo_max;
    }
    return launch_bcd_sweep(a, scratch.as<double>(), scratch_ld, (hipStream_t)stream);
}

int fdx_bcd_fold_dev(void* stats_dev, double* rel_change_dev, int32_t it, void* stream) {
    FDX_REQUIRE(stats_dev && rel_change_dev && it >= 0, "fdx_bcd_fold_dev: bad arguments");
    return launch_bcd_fold_last((const unsigned long long*)stats_dev, rel_change_dev, it, (hipStream_t)stream);
}

int fdx_objective_partials_dev(const fdx_graph* g, const double* beta_dev, int64_t ld, const double* H_dev, int64_t ldh,
                               const double* XtX_dev, int32_t K, double* out4_host, void* stream) {
    FDX_TRY(fdx::graph_meta_sync(g));
    FDX_REQUIRE(g && beta_dev && H_dev && XtX_dev && out4_host, "fdx_objective_partials_dev: null argument");
    hipStream_t st = (hipStream_t)stream;
    PoolStream pool_stream(st);
    out4_host[0] = out4_host[1] = out4_host[2] = out4_host[3] = 0.0;
    if (g->n == 0) return 0;
    DevBuf part, out;
    const int nblk = objective_partials_count(g->n_slices);
    FDX_TRY(part.alloc((size_t)nblk * 4 * sizeof(double)));
    FDX_TRY(out.alloc(4 * sizeof(double)));
    FDX_TRY(solver_objective_partials(*g, beta_dev, ld, H_dev, ldh, XtX_dev, K, part.as<double>(), out.as<double>(), st));
    FDX_HIP(hipMemcpyAsync(out4_host, out.p, 4 * sizeof(double), hipMemcpyDeviceToHost, st));
    FDX_HIP(hipStreamSynchronize(st));
    return 0;
}

int fdx_normalize_dev(const double* beta_dev, int64_t ld, int64_t n, int32_t K, double* beta_out_dev, double* prop_out_dev,
                      void* stream) {
    FDX_REQUIRE(beta_dev && n >= 0 && K > 0, "fdx_normalize_dev: bad arguments");
    if (n == 0) return 0;
    return launch_normalize_export(beta_dev, ld, nullptr, (int)n, (int)((n + 63) / 64), K, beta_out_dev, prop_out_dev,
                                   (hipStream_t)stream);
}

}  // extern "C"

extern "C" int fdx_gene_moments_dev(const void* Y_dev, int32_t dtype, int64_t n, int32_t G, int64_t ldy, double* mean_out_host,
                                    double* var_out_host, void* stream) {
    FDX_REQUIRE(Y_dev && mean_out_host && var_out_host && n > 0 && G > 0, "fdx_gene_moments_dev: bad arguments");
    hipStream_t st = (hipStream_t)stream;
    PoolStream pool_stream(st);
    DevBuf scale, part, mean, var;
    FDX_TRY(scale.alloc((size_t)n * 8));
    FDX_TRY(part.alloc((size_t)column_sums_parts(n) * 2 * G * 8));
    FDX_TRY(mean.alloc((size_t)G * 8));
    FDX_TRY(var.alloc((size_t)G * 8));
    FDX_TRY(launch_gene_moments(Y_dev, dtype, ldy, n, G, scale.as<double>(), part.as<double>(), mean.as<double>(), var.as<double>(), st));
    FDX_TRY(copy_d2h(mean_out_host, mean.p, (size_t)G * 8, st));
    FDX_TRY(copy_d2h(var_out_host, var.p, (size_t)G * 8, st));
    return 0;
}

extern "C" int fdx_gather_columns_dev(const void* Y_dev, int32_t dtype, int64_t n, int32_t G, int64_t ldy, const int32_t* idx_host,
                                      int32_t G_sel, void* out_dev, void* stream) {
    FDX_REQUIRE(n >= 0 && G > 0 && G_sel > 0 && idx_host && (n == 0 || (Y_dev && out_dev)), "fdx_gather_columns_dev: bad arguments");
    for (int j = 0; j < G_sel; ++j) FDX_REQUIRE(idx_host[j] >= 0 && idx_host[j] < G, "fdx_gather_columns_dev: column index out of range");
    hipStream_t st = (hipStream_t)stream;
    PoolStream pool_stream(st);
    DevBuf di;
    FDX_TRY(di.alloc((size_t)G_sel * 4));
    FDX_TRY(copy_h2d(di.p, idx_host, (size_t)G_sel * 4, st));
    FDX_TRY(launch_gather_columns(Y_dev, dtype, ldy, n, G, di.as<int>(), G_sel, out_dev, st));
    FDX_HIP(hipStreamSynchronize(st));
    return 0;
}
