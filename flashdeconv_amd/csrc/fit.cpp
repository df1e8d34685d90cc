// Device-resident fit pipeline: steps 2-6 of FlashDeconv.fit (flashdeconv/core/deconv.py:326-398) behind one C call.
//   graph (utils/graph.py) -> X_sketch, XtX -> Y_sketch chunks -> H, YtY -> lambda (core/spatial.py:144-192)
//   -> BCD solve (core/solver.py:287-428) -> beta / proportions in the caller's spot order (core/solver.py:431-452).
// Everything between "Y, coords resident in HBM" and "beta_, proportions_ resident in HBM" stays on the device; the
// host only sees a handful of scalars (bounding box, XtX, rel_change trace).
#include "fdx_env.h"
#include <algorithm>
#include <chrono>
#include <cstdio>
#include <cmath>
#include <cstdlib>
#include <memory>
#include <vector>

#include "fdx_graph.h"
#include "fdx_internal.h"
#include "fdx_kernels.h"
#include "graph_build.h"
#include "sketch_plan.h"
#include "solver.h"

using namespace fdx;

namespace {

// FDX_TRACE_HOST=1: host time between the marked points of a fit (stderr), to see whether the host keeps ahead of the device
void fit_trace_host(const char* what) {
    static const bool on = fdx::env("FDX_TRACE_HOST") != nullptr;
    if (!on) return;
    static auto t_prev = std::chrono::steady_clock::now();
    const auto t = std::chrono::steady_clock::now();
    std::fprintf(stderr, "[fdx-host] +%7.1f us  fit: %s\n", std::chrono::duration<double, std::micro>(t - t_prev).count(), what);
    t_prev = t;
}

struct StageTimer {
    hipStream_t st;
    hipEvent_t ev[8];
    int n = 0;
    explicit StageTimer(hipStream_t s) : st(s) {
        for (auto& e : ev) (void)hipEventCreate(&e);
    }
    ~StageTimer() {
        for (auto& e : ev) (void)hipEventDestroy(e);
    }
    void mark() { if (n < 8) (void)hipEventRecord(ev[n++], st); }
    double ms(int a, int b) {
        float t = 0.f;
        if (a < n && b < n) (void)hipEventElapsedTime(&t, ev[a], ev[b]);
        return t;
    }
};

int build_csc_from_tables(const int32_t* bucket, const double* weight, int G, int d, std::vector<long long>* col_ptr,
                          std::vector<int>* gene_idx, std::vector<double>* w) {
    col_ptr->assign((size_t)d + 1, 0);
    for (int g = 0; g < G; ++g) {
        FDX_REQUIRE(bucket[g] >= 0 && bucket[g] < d, "fit: bucket index out of range");
        (*col_ptr)[(size_t)bucket[g] + 1]++;
    }
    for (int c = 0; c < d; ++c) (*col_ptr)[(size_t)c + 1] += (*col_ptr)[(size_t)c];
    gene_idx->assign((size_t)G, 0);
    w->assign((size_t)G, 0.0);
    std::vector<long long> cur(col_ptr->begin(), col_ptr->end() - 1);
    for (int g = 0; g < G; ++g) {   // ascending gene order inside every bucket
        const long long e = cur[(size_t)bucket[g]]++;
        (*gene_idx)[(size_t)e] = g;
        (*w)[(size_t)e] = weight[g];
    }
    return 0;
}

}  // namespace

// The Jacobi SVD runs in ONE workgroup, so it occupies one CU for ~1.6 ms while the other 255 idle.  begin/end let the
// caller put the spatial-graph build (many short, latency-bound kernels on the caller's stream) under it: the job runs
// on a library-owned non-blocking side stream.
struct fdx_leverage_job {
    // the scores and the status words travel to pinned memory right behind the kernels (queued by begin): end only waits for
    // the event - collecting a finished job cost 70 us of two pageable copies and a stream synchronisation
    double* pin = nullptr;          // G doubles, then 8 ints
    size_t pin_cap = 0;
    hipEvent_t done = nullptr;
    double* hX = nullptr;           // pinned copy of the signatures (the caller may drop X once begin returns; the upload does not stage)
    size_t hX_cap = 0;
    ~fdx_leverage_job() {
        if (done) (void)hipEventDestroy(done);
        if (pin) fdx::pinned_buffer_put(pin, pin_cap);
        if (hX) fdx::pinned_buffer_put(hX, hX_cap);
    }
    DevBuf dX, dW, dS, dL, dDbg, dScratch;
    int K = 0, G = 0, route = LEV_ROUTE_SVD;
    double reg = 0.0;
    hipStream_t st = nullptr;
    std::shared_ptr<fdx::HelperTicket> ticket;   // begin's launches were handed to the helper thread: end waits for them first
};

namespace fdx {
hipStream_t library_side_stream();
}
namespace {
hipStream_t leverage_side_stream() { return fdx::library_side_stream(); }
}
namespace fdx {
// the library's per-device side stream (leverage job, the X-side preamble of a fit / a shard's prepare, the export)
hipStream_t library_side_stream() {
    static hipStream_t streams[64] = {};
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return nullptr;
    if (!streams[dev]) {
        // A priority of its own keeps the side stream off the hardware queue the caller's streams share (HIP multiplexes
        // same-priority streams over a few queues; with RCCL's streams around, aliasing with the caller's stream
        // serialised the SVD with the graph build).
        int lo = 0, hi = 0;
        (void)hipDeviceGetStreamPriorityRange(&lo, &hi);
        if (hipStreamCreateWithPriority(&streams[dev], hipStreamNonBlocking, hi) != hipSuccess &&
            hipStreamCreateWithFlags(&streams[dev], hipStreamNonBlocking) != hipSuccess)
            streams[dev] = nullptr;
    }
    return streams[dev];
}
}  // namespace fdx

extern "C" int fdx_leverage_begin(const double* X, int32_t K, int32_t G, double regularization, fdx_leverage_job** out) {
    return fdx_leverage_begin_opt(X, K, G, regularization, 1, out);
}

// queue_async = 0: the launches are queued by the calling thread (a caller that collects the scores at once gains nothing from the
// helper thread - and in the gene-selection flow of FlashDeconv.fit the hand-over measurably cost ~3 ms of wait on some boxes)
extern "C" int fdx_leverage_begin_opt(const double* X, int32_t K, int32_t G, double regularization, int32_t queue_async,
                                      fdx_leverage_job** out) {
    FDX_REQUIRE(X && out && K > 0 && G > 0, "fdx_leverage_begin: bad arguments");
    *out = nullptr;
    auto* job = new fdx_leverage_job();
    job->K = K;
    job->G = G;
    job->st = leverage_side_stream();
    job->hX = (double*)pinned_buffer_get((size_t)K * G * sizeof(double), &job->hX_cap);
    if (!job->hX) { delete job; return fail(FDX_ERR_HIP, "fdx_leverage_begin: pinned host buffer"); }
    std::memcpy(job->hX, X, (size_t)K * G * sizeof(double));
    // the upload (a pageable copy: the host waits for it) and the launches cost ~75 us of host time that the caller - on its way to
    // a graph build, with the scores not needed before the sketch tables - has better uses for: the helper thread queues them
    const bool async = queue_async != 0 && !fdx::exp_env("FDX_NO_HELPER_THREAD");
    auto run = [job, K, G, regularization]() -> int {
        PoolStream pool_stream(job->st);
        FDX_TRY(job->dX.alloc((size_t)K * G * sizeof(double)));
        FDX_TRY(job->dW.alloc((size_t)K * G * sizeof(double)));
        FDX_TRY(job->dS.alloc((size_t)K * sizeof(double)));
        FDX_TRY(job->dL.alloc((size_t)G * sizeof(double)));
        FDX_TRY(job->dDbg.alloc(8 * sizeof(int)));
        FDX_HIP(hipMemcpyAsync(job->dX.p, job->hX, (size_t)K * G * sizeof(double), hipMemcpyHostToDevice, job->st));
        FDX_TRY(job->dScratch.alloc(leverage_scratch_doubles(K, G) * sizeof(double)));
        job->reg = regularization;
        job->route = leverage_qr_applies(K, G) ? LEV_ROUTE_QR : LEV_ROUTE_SVD;
        FDX_TRY(launch_leverage(job->dX.as<double>(), K, G, regularization, job->dW.as<double>(), job->dS.as<double>(),
                                job->dL.as<double>(), job->dDbg.as<int>(), job->dScratch.as<double>(), job->st, job->route));
        job->pin = (double*)pinned_buffer_get((size_t)G * sizeof(double) + 64, &job->pin_cap);
        FDX_REQUIRE(job->pin != nullptr, "fdx_leverage_begin: pinned host buffer");
        FDX_HIP(hipMemcpyAsync(job->pin, job->dL.p, (size_t)G * sizeof(double), hipMemcpyDeviceToHost, job->st));
        FDX_HIP(hipMemcpyAsync(job->pin + G, job->dDbg.p, 8 * sizeof(int), hipMemcpyDeviceToHost, job->st));
        FDX_HIP(hipEventCreateWithFlags(&job->done, hipEventDisableTiming));
        FDX_HIP(hipEventRecord(job->done, job->st));
        return 0;
    };
    if (async) {
        job->ticket = helper_submit(run);
        *out = job;
        return 0;
    }
    const int rc = run();
    if (rc) {
        (void)hipStreamSynchronize(job->st);
        delete job;
        return rc;
    }
    *out = job;
    return 0;
}

static int leverage_end_impl(fdx_leverage_job* job, double* lev_out, double** x_dev_out);
extern "C" int fdx_leverage_end(fdx_leverage_job* job, double* lev_out) { return leverage_end_impl(job, lev_out, nullptr); }
extern "C" int fdx_leverage_end_keep(fdx_leverage_job* job, double* lev_out, double** x_dev_out) {
    FDX_REQUIRE(x_dev_out != nullptr, "fdx_leverage_end_keep: null output");
    *x_dev_out = nullptr;
    return leverage_end_impl(job, lev_out, x_dev_out);
}
static int leverage_end_impl(fdx_leverage_job* job, double* lev_out, double** x_dev_out) {
    FDX_REQUIRE(job != nullptr, "fdx_leverage_end: null job");
    if (job->ticket) {
        const int qrc = helper_wait(job->ticket);
        job->ticket.reset();
        if (qrc) {
            (void)hipStreamSynchronize(job->st);
            delete job;
            return qrc;
        }
    }
    int rc = 0;
    int dbg[8] = {0};
    auto run = [&]() -> int {
        FDX_REQUIRE(lev_out != nullptr, "fdx_leverage_end: null output");
        FDX_HIP(hipMemcpyAsync(lev_out, job->dL.p, (size_t)job->G * sizeof(double), hipMemcpyDeviceToHost, job->st));
        FDX_HIP(hipMemcpyAsync(dbg, job->dDbg.p, sizeof(dbg), hipMemcpyDeviceToHost, job->st));
        return 0;
    };
    hipError_t e = hipSuccess;
    if (lev_out == nullptr) rc = fail(FDX_ERR_INVALID, "fdx_leverage_end: null output");
    if (job->done && job->pin) {
        // the job's kernels and its copies are the last work of the job on its stream: behind the event nothing of it is in flight
        e = hipEventSynchronize(job->done);
        if (!rc && e == hipSuccess) {
            std::memcpy(lev_out, job->pin, (size_t)job->G * sizeof(double));
            std::memcpy(dbg, job->pin + job->G, sizeof(dbg));
        }
    } else {
        if (!rc) rc = run();
        e = hipStreamSynchronize(job->st);          // always drain before the buffers go back to the pool
    }
    if (!rc && e == hipSuccess && job->route == LEV_ROUTE_QR && dbg[7] != 1) {
        // the Cholesky-QR route refused the matrix (rank-deficient or cond above ~3e4): the Jacobi SVD passes, as before
        PoolStream pool_stream(job->st);
        job->route = LEV_ROUTE_SVD;
        rc = launch_leverage(job->dX.as<double>(), job->K, job->G, job->reg, job->dW.as<double>(), job->dS.as<double>(),
                             job->dL.as<double>(), job->dDbg.as<int>(), job->dScratch.as<double>(), job->st, LEV_ROUTE_SVD);
        if (!rc) rc = run();
        e = hipStreamSynchronize(job->st);
        if (!rc) dbg[5] = 1;                                   // for the debug line: fell back
    }
    if (!rc && e != hipSuccess) rc = fail(FDX_ERR_HIP, hipGetErrorString(e));
    if (e == hipSuccess)              // nothing of the job is in flight any more: the blocks may follow any stream
        for (DevBuf* b : {&job->dX, &job->dW, &job->dS, &job->dL, &job->dDbg, &job->dScratch}) b->mark_idle();
    if (!rc && e == hipSuccess && x_dev_out) {      // the device copy of X changes hands (the caller returns it with fdx_free)
        *x_dev_out = job->dX.as<double>();
        job->dX.p = nullptr;
        job->dX.bytes = job->dX.cap = 0;
    }
    if (!rc && fdx::env("FDX_DEBUG"))   // phase stamps in 100 MHz ticks
        std::fprintf(stderr, "[fdx] leverage: K=%d G=%d route=%s passes/sweeps=%d converged=%d\n", job->K, job->G,
                     job->route == LEV_ROUTE_QR ? "cholesky-qr" : "jacobi-svd", dbg[0], dbg[6]);
    delete job;
    return rc;
}

extern "C" int fdx_leverage_scores(const double* X, int32_t K, int32_t G, double regularization, double* lev_out) {
    FDX_REQUIRE(X && lev_out && K > 0 && G > 0, "fdx_leverage_scores: bad arguments");
    fdx_leverage_job* job = nullptr;
    FDX_TRY(fdx_leverage_begin(X, K, G, regularization, &job));
    return fdx_leverage_end(job, lev_out);
}

extern "C" int fdx_column_sums_dev(const void* Y_dev, int32_t dtype, int64_t n, int32_t G, int64_t ldy,
                                   double* sums_out_host, void* stream) {
    FDX_REQUIRE(dtype == FDX_F32 || dtype == FDX_F64, "fdx_column_sums_dev: dtype must be FDX_F32 or FDX_F64");
    FDX_REQUIRE(n >= 0 && G > 0 && sums_out_host && (n == 0 || Y_dev), "fdx_column_sums_dev: bad arguments");
    hipStream_t st = (hipStream_t)stream;
    PoolStream pool_stream(st);
    DevBuf dPart, dOut;
    FDX_TRY(dPart.alloc((size_t)column_sums_parts(n) * G * sizeof(double)));
    FDX_TRY(dOut.alloc((size_t)G * sizeof(double)));
    FDX_TRY(launch_column_sums(Y_dev, dtype, ldy, n, G, dPart.as<double>(), dOut.as<double>(), st));
    FDX_TRY(copy_d2h(sums_out_host, dOut.p, (size_t)G * sizeof(double), st));
    FDX_HIP(hipStreamSynchronize(st));
    return 0;
}

// What a fit that stopped on k-NN ties leaves behind for the fit that follows on the rebuilt graph: the sketch -> H stage (the
// rebuilt graph keeps the Morton order, so H's columns are where the second fit wants them) with everything the still-running
// kernel reads, and an event behind it.  The stopped call returns at once instead of draining ~3 ms of sketch that the tie remedy's
// host work (a kd-tree build of ~20 ms) then runs beside.
struct fdx_fit_carry {
    fdx::DevBuf dH, dRowSq, dXs;
    fdx::CsrSelection csr_sel;
    std::shared_ptr<fdx::SketchPlan> plan_y;
    hipEvent_t done = nullptr;
    long long n = 0, ld = 0;
    int K = 0, KP = 0, d = 0, G = 0, mode_y = 0;
    const void* y_id = nullptr;
    double sketch_ms = 0.0, gram_ms = 0.0;
    ~fdx_fit_carry() {
        if (done) { (void)hipEventSynchronize(done); (void)hipEventDestroy(done); }
    }
};

extern "C" int fdx_fit_carry_free(void* carry) {
    delete static_cast<fdx_fit_carry*>(carry);
    return 0;
}

namespace {

// Where the spot rows come from: a dense (n, G) device matrix, or a CSR matrix over G_all columns of which gene_idx
// (host, G entries; NULL = all columns in order) are the selected genes.
struct YSource {
    const void* dense = nullptr;
    int32_t dtype = FDX_F32;
    int64_t ldy = 0;
    const fdx_csr_view* csr = nullptr;
    const int32_t* gene_idx = nullptr;
};

int fit_impl(const YSource& ysrc, int64_t n, int32_t G, const double* X, int32_t K, const int32_t* bucket,
             const double* weight_y, const double* weight_x, const double* coords_dev, int32_t dim,
             const fdx_fit_params* prm_in, fdx_graph** graph_inout, double* beta_out_dev, double* prop_out_dev,
             double* objectives_out, double* rel_changes_out, fdx_fit_info* info, void* stream) {
    FDX_REQUIRE(info != nullptr && prm_in != nullptr && graph_inout != nullptr, "fdx_fit_dev: null argument");
    fdx_fit_params prm_local = *prm_in;
    prm_local.mode_y &= 0xff;
    const fdx_fit_params* prm = &prm_local;
    TileF64Math f64_math((prm_in->mode_y & FDX_PRE_F64_MATH) != 0);
    std::memset(info, 0, sizeof(*info));
    const void* Y_dev = ysrc.dense;
    const int32_t y_dtype = ysrc.csr ? ysrc.csr->dtype : ysrc.dtype;
    const int64_t ldy = ysrc.ldy;
    FDX_REQUIRE(y_dtype == FDX_F32 || y_dtype == FDX_F64, "fdx_fit_dev: Y dtype must be FDX_F32 or FDX_F64");
    FDX_REQUIRE(n > 0 && G > 0 && K > 0, "fdx_fit_dev: empty problem");
    FDX_REQUIRE(n < 0x7fffff00LL, "fdx_fit_dev: n too large for one device");
    FDX_REQUIRE(ysrc.csr || ldy >= G, "fdx_fit_dev: ldy < G");
    FDX_REQUIRE((Y_dev || ysrc.csr) && X && bucket && weight_y && weight_x, "fdx_fit_dev: null array");
    const int d = prm->sketch_dim;
    FDX_REQUIRE(d > 0, "fdx_fit_dev: sketch_dim must be positive");
    FDX_REQUIRE(prm->max_iter >= 0, "fdx_fit_dev: max_iter must be non-negative");
    hipStream_t st = (hipStream_t)stream;
    PoolStream pool_stream(st);
    StageTimer tm(st);
    tm.mark();  // 0
    fit_trace_host("entry");

    // ---- spatial graph (core/deconv.py:358)
    fdx_graph* g = nullptr;
    if (prm->graph_method == FDX_GRAPH_GIVEN) {
        g = *graph_inout;
        FDX_REQUIRE(g != nullptr && g->n == n, "fdx_fit_dev: given graph does not match n");
    } else {
        FDX_REQUIRE(coords_dev != nullptr, "fdx_fit_dev: null coords");
        g = new fdx_graph();
        int rc = (prm->graph_method == FDX_GRAPH_KNN)
                     ? graph_build_knn(coords_dev, n, dim, prm->k_neighbors, g, st)
                     : graph_build_radius(coords_dev, n, dim, prm->radius, 0, n, g, st);
        if (rc) { delete g; return rc; }
        *graph_inout = g;
    }
    // a deferred build (graph_kernels.cpp) queued on ANOTHER stream: everything below reads the graph's arrays (perm first)
    if (g->meta_pending && g->meta_event && g->meta_stream != st) FDX_HIP(hipStreamWaitEvent(st, g->meta_event, 0));
    tm.mark();  // 1

    // ---- Everything that does not depend on the graph or on Y goes to the library's side stream and runs beside the graph build
    // queued on the caller's stream just before: beta0 = 1/K (core/solver.py:372) and the cleared pad rows, the sketch tables
    // (Omega comes from the host: hash/sign from numpy's RandomState, core/sketching.py:58-59; plans are shared through a
    // content-keyed cache, so a repeated fit builds nothing), X, X_sketch (K, d) and XtX (core/sketching.py:202-204,
    // core/solver.py:346).  The buffers are allocated with the side stream as their pool stream (a block last used elsewhere
    // orders it behind that work); the caller's stream waits for ONE event before the sketch.  (Queued on the caller's stream
    // these ~120 us of small launches and copies sat between the end of the graph build and the start of the sketch, and a new
    // plan's table upload made the host wait for the whole graph build.)
    const long long ld = round_up(n + 1, 64);
    const int KP = solver_padded_K(K);             // 65 - 128 cell types: planes of the next instantiated sweep, pad types all zero
    DevBuf dB0, dB1, dX, dXs, dG, dGp;
    CsrSelection csr_sel;           // CSR source: which columns are selected genes, their {weight, bucket}
    bool csr_fused = false;
    DevBuf dH, dYs, dRowSq, dSum;   // declared ABOVE the drains: on an early return both streams are drained before any of these goes back to the pool (dRowSq / dSum are read on the side stream)
    struct SideDrain { hipStream_t s = nullptr; ~SideDrain() { if (s) (void)hipStreamSynchronize(s); } } side_drain;   // before buffers are released
    // an early return leaves work on the caller's stream that uses the side-stream buffers above: wait for it before they go
    struct AbortDrain { hipStream_t s; bool armed = true; ~AbortDrain() { if (armed) (void)hipStreamSynchronize(s); } } abort_drain{st};
    hipEvent_t evInit = nullptr;
    struct EvGuard0 { hipEvent_t* e; ~EvGuard0() { if (*e) (void)hipEventDestroy(*e); } } evInit_guard{&evInit};
    hipStream_t side = fdx::env("FDX_NO_SIDE_STREAM") ? nullptr : leverage_side_stream();
    if (side == st) side = nullptr;
    const hipStream_t xs = side ? side : st;
    if (side) side_drain.s = side;
    std::shared_ptr<SketchPlan> plan_y_p, plan_x_p;
    for (int g_ = 0; g_ < G; ++g_) FDX_REQUIRE(bucket[g_] >= 0 && bucket[g_] < d, "fit: bucket index out of range");
    double* Gh = (double*)pinned_scratch(0, (size_t)K * K * sizeof(double));   // pinned: the copy below must not hold the host back
    FDX_REQUIRE(Gh != nullptr, "fit: pinned host buffer");
    hipEvent_t evG = nullptr;
    FDX_HIP(hipEventCreateWithFlags(&evG, hipEventDisableTiming));
    struct EvGuard { hipEvent_t e; ~EvGuard() { if (e) (void)hipEventDestroy(e); } } evG_guard{evG};
    const bool beta0_virtual = side && KP == K && K <= FDX_MAX_K_FAST && prm->max_iter > 0 && !prm->verbose && !fdx::env("FDX_NO_INIT_SWEEP");
    {
        PoolStream pool_xs(xs);
        FDX_TRY(dB0.alloc((size_t)KP * ld * sizeof(double)));
        FDX_TRY(dB1.alloc((size_t)KP * ld * sizeof(double)));
        if (side) {
            // no pad types: the uniform start vector is a constant of the first sweep and is not written (solver.cpp: beta0_virtual)
            if (beta0_virtual) FDX_TRY(solver_zero_pad(dB0.as<double>(), ld, g->n_total, KP, xs));
            else FDX_TRY(solver_init_beta(dB0.as<double>(), ld, g->n_total, K, xs, KP));
            FDX_TRY(solver_zero_pad(dB1.as<double>(), ld, g->n_total, KP, xs));
        }
        if (ysrc.csr) {
            csr_fused = csr_contract_ok(d, K, (ysrc.csr->G + 31) / 32);
            FDX_TRY(csr_sel.build(ysrc.gene_idx, G, ysrc.csr->G, bucket, weight_y, d, csr_fused, xs, "fdx_fit_csr_dev"));
        } else {
            FDX_TRY(sketch_plan_cached(bucket, weight_y, G, d, xs, &plan_y_p));
        }
        if (!ysrc.csr && weight_x == weight_y) plan_x_p = plan_y_p;
        else FDX_TRY(sketch_plan_cached(bucket, weight_x, G, d, xs, &plan_x_p));
        FDX_TRY(dX.alloc((size_t)K * G * sizeof(double)));
        FDX_TRY(dXs.alloc((size_t)K * d * sizeof(double)));
        FDX_TRY(dG.alloc((size_t)K * K * sizeof(double)));
        FDX_TRY(copy_h2d(dX.p, X, (size_t)K * G * sizeof(double), xs));
        FDX_TRY(launch_sketch_rows(dX.p, FDX_F64, G, nullptr, K, G, d, prm->mode_x, plan_x_p->dev(), dXs.as<double>(), d, nullptr, xs));
        FDX_TRY(launch_xyt(dXs.as<double>(), dXs.as<double>(), d, K, d, K, dG.as<double>(), K, nullptr, xs));
        if (KP != K) {
            FDX_TRY(dGp.alloc((size_t)KP * KP * sizeof(double)));
            FDX_TRY(solver_pad_square(dG.as<double>(), K, dGp.as<double>(), KP, xs));
        }
        // XtX goes to the host NOW: lambda and the scaled rho are host scalars of the sweeps, and with them known early the solve
        // is queued behind the sketch without the host waiting for it
        FDX_HIP(hipMemcpyAsync(Gh, dG.p, (size_t)K * K * sizeof(double), hipMemcpyDeviceToHost, xs));
        FDX_HIP(hipEventRecord(evG, xs));
        if (side) {
            FDX_HIP(hipEventCreateWithFlags(&evInit, hipEventDisableTiming));
            FDX_HIP(hipEventRecord(evInit, side));
            FDX_HIP(hipStreamWaitEvent(st, evInit, 0));          // tables, X_sketch, XtX, beta0: all behind this one
        }
    }
    SketchPlan plan_none;
    const SketchPlan& plan_y = plan_y_p ? *plan_y_p : plan_none;

    // ---- a carry: the sketch -> H stage of a call that stopped on ties for these inputs (the rebuilt graph keeps the spot order)
    std::unique_ptr<fdx_fit_carry> carry_in(static_cast<fdx_fit_carry*>(prm_in->carry));
    if (carry_in) {
        FDX_REQUIRE(carry_in->n == n && carry_in->ld == ld && carry_in->K == K && carry_in->KP == KP && carry_in->d == d &&
                        carry_in->G == G && carry_in->mode_y == prm->mode_y && carry_in->y_id == (ysrc.csr ? (const void*)ysrc.csr->data : Y_dev),
                    "fdx_fit_dev: the carry belongs to a different problem");
    }
    // ---- Y_sketch in solver order, chunked, contracted into H (K, ld) as it is produced
    if (!carry_in) {
        FDX_TRY(dH.alloc((size_t)KP * ld * sizeof(double)));
        if (KP != K) FDX_HIP(hipMemsetAsync(dH.as<double>() + (size_t)K * ld, 0, (size_t)(KP - K) * ld * sizeof(double), st));   // pad types
        FDX_TRY(dRowSq.alloc((size_t)n * sizeof(double)));
    }
    FDX_TRY(dSum.alloc(sizeof(double)));
    // Y_sketch is produced and consumed in chunks of 256k rows (1 GB at d = 512): measured on MI355X, smaller chunks
    // (down to Infinity-Cache size) under-fill the chip and are slower, larger ones gain nothing.
    long long chunk_rows = 1LL << 18;
    if (const char* e = fdx::exp_env("FDX_FIT_CHUNK")) chunk_rows = std::max<long long>(64, atoll(e));
    const long long chunk = std::min<long long>(n, chunk_rows);
    const bool fused = csr_fused || (!ysrc.csr && fused_sketch_contract_ok(y_dtype, ldy, Y_dev, G, d, K, prm->mode_y, plan_y.dev()));
    if (!fused && !carry_in) FDX_TRY(dYs.alloc((size_t)chunk * d * sizeof(double)));
    if (!carry_in) FDX_TRY(solver_zero_pad(dH.as<double>(), ld, n, K, st));   // columns of real spots are all written by the sketch -> H stage
    const int* row_map = g->identity_order ? nullptr : g->perm.as<int>();
    double sketch_ms = 0.0, gram_ms = 0.0;
    hipEvent_t eS0 = nullptr, eS1 = nullptr;     // around the sketch -> H stage; read at the end of the fit, no wait here
    struct EvGuard2 { hipEvent_t* a; hipEvent_t* b; ~EvGuard2() { if (*a) (void)hipEventDestroy(*a); if (*b) (void)hipEventDestroy(*b); } } eS_guard{&eS0, &eS1};
    FDX_HIP(hipEventCreate(&eS0));
    FDX_HIP(hipEventCreate(&eS1));
    fit_trace_host("side-stream preamble queued, buffers allocated");
    FDX_HIP(hipEventRecord(eS0, st));            // the prologue (graph chain, X-side preamble) ends here
    if (carry_in) {                               // H and the rows' squared norms as the stopped call left them, behind its event
        FDX_HIP(hipStreamWaitEvent(st, carry_in->done, 0));
        dH.take(carry_in->dH);
        dRowSq.take(carry_in->dRowSq);
        sketch_ms = carry_in->sketch_ms;
        gram_ms = carry_in->gram_ms;
        FDX_HIP(hipEventRecord(eS1, st));
    } else if (fused) {   // one kernel, no Y_sketch: rows -> LDS tile -> bucket sums -> MFMA contraction -> H  (tile_kernels.cpp)
        if (csr_fused)     // CSR rows -> LDS accumulators -> MFMA contraction -> H  (csr_kernels.cpp)
            FDX_TRY(launch_sketch_csr_contract((const long long*)ysrc.csr->indptr, ysrc.csr->indices, ysrc.csr->data, y_dtype,
                                               row_map, n, d, prm->mode_y, csr_sel, dXs.as<double>(), K, dH.as<double>(), ld,
                                               dRowSq.as<double>(), st));
        else
        FDX_TRY(launch_sketch_contract(Y_dev, y_dtype, ldy, row_map, n, G, d, prm->mode_y, plan_y.dev(), dXs.as<double>(), K,
                                       dH.as<double>(), ld, dRowSq.as<double>(), st));
        FDX_HIP(hipEventRecord(eS1, st));
    } else {
        const int n_chunks = (int)((n + chunk - 1) / chunk);
        const int n_timed = std::min(n_chunks, 64);             // stage timing from up to 64 chunks, scaled
        std::vector<hipEvent_t> ev((size_t)n_timed * 3);
        for (auto& e : ev) FDX_HIP(hipEventCreate(&e));
        int ci = 0;
        for (long long r0 = 0; r0 < n; r0 += chunk, ++ci) {
            const long long nr = std::min(chunk, n - r0);
            if (ci < n_timed) FDX_HIP(hipEventRecord(ev[(size_t)ci * 3], st));
            // with a row map the chunk gathers rows perm[r0..]; without one it reads rows r0.. of Y directly
            if (ysrc.csr) {
                FDX_TRY(launch_sketch_csr((const long long*)ysrc.csr->indptr, ysrc.csr->indices, ysrc.csr->data, y_dtype,
                                          row_map ? row_map + r0 : nullptr, r0, nr, d, prm->mode_y, csr_sel.slots.p,
                                          csr_sel.bits.as<unsigned>(), csr_sel.sel_words, dYs.as<double>(), d,
                                          dRowSq.as<double>() + r0, st));
            } else {
                const unsigned char* ybase = static_cast<const unsigned char*>(Y_dev);
                if (!row_map) ybase += (size_t)r0 * (size_t)ldy * (y_dtype == FDX_F32 ? 4 : 8);
                FDX_TRY(launch_sketch_rows(ybase, y_dtype, ldy, row_map ? row_map + r0 : nullptr, nr, G, d, prm->mode_y,
                                           plan_y.dev(), dYs.as<double>(), d, dRowSq.as<double>() + r0, st));
            }
            if (ci < n_timed) FDX_HIP(hipEventRecord(ev[(size_t)ci * 3 + 1], st));
            FDX_TRY(launch_xyt(dXs.as<double>(), dYs.as<double>(), d, nr, d, K, dH.as<double>() + r0, ld, nullptr, st));
            if (ci < n_timed) FDX_HIP(hipEventRecord(ev[(size_t)ci * 3 + 2], st));
        }
        FDX_HIP(hipStreamSynchronize(st));
        for (int c = 0; c < n_timed; ++c) {
            float t1 = 0.f, t2 = 0.f;
            (void)hipEventElapsedTime(&t1, ev[(size_t)c * 3], ev[(size_t)c * 3 + 1]);
            (void)hipEventElapsedTime(&t2, ev[(size_t)c * 3 + 1], ev[(size_t)c * 3 + 2]);
            sketch_ms += t1;
            gram_ms += t2;
        }
        const double scale = (double)n_chunks / (double)n_timed;
        sketch_ms *= scale;
        gram_ms *= scale;
        for (auto& e : ev) (void)hipEventDestroy(e);
        FDX_HIP(hipEventRecord(eS1, st));
    }
    // ---- the graph's counts (a build that was only queued has long finished behind the sketch launch).  Ties under stop_on_ties: the
    // caller wants the reference's choice among equidistant neighbours - nothing is solved on this graph, and the sketch -> H stage
    // that is already running goes to the caller as a carry for the fit on the rebuilt graph (tie-free inputs never pay for the
    // question; lattices no longer pay a second sketch and the wait for the first)
    if (!prm->verbose) FDX_HIP(hipEventSynchronize(evG));
    FDX_TRY(graph_meta_sync(g));
    info->knn_ties = g->knn_ties;
    info->nnz = g->nnz;
    if (prm->stop_on_ties && g->knn_ties > 0) {
        info->status = FDX_FIT_TIES;
        if (carry_in) return 0;                   // (a carry on a graph that still has ties: dropped, drained by the guards)
        auto carry = std::make_unique<fdx_fit_carry>();
        FDX_HIP(hipEventCreateWithFlags(&carry->done, hipEventDisableTiming));
        FDX_HIP(hipEventRecord(carry->done, st));
        carry->dH.take(dH);
        carry->dRowSq.take(dRowSq);
        carry->dXs.take(dXs);                     // (read by the fused kernels; the X side is cheap to redo)
        carry->csr_sel.slots.take(csr_sel.slots); carry->csr_sel.bits.take(csr_sel.bits); carry->csr_sel.words.take(csr_sel.words);
        carry->csr_sel.w.take(csr_sel.w); carry->csr_sel.b.take(csr_sel.b);
        carry->plan_y = plan_y_p;
        carry->n = n; carry->ld = ld; carry->K = K; carry->KP = KP; carry->d = d; carry->G = G; carry->mode_y = prm->mode_y;
        carry->y_id = ysrc.csr ? (const void*)ysrc.csr->data : Y_dev;
        float t_sk = 0.f;
        (void)t_sk;
        carry->sketch_ms = 0.0;                   // (the stage's events belong to the stopped call: the second fit reports the stage as carried)
        carry->gram_ms = gram_ms;
        if (!fused) FDX_HIP(hipStreamSynchronize(st));   // the chunked path's Y_sketch buffer goes back to the pool with this call
        info->carry = carry.release();
        abort_drain.armed = false;                // everything the running kernel reads lives in the carry (Y and the graph: the caller's)
        return 0;
    }
    // YtY (core/solver.py:348) only enters the objective: its two small reductions and the read-back go to the side stream (behind the
    // sketch, beside the first sweep) instead of standing between the sketch and the sweeps; the verbose trace needs it at once
    double* YtY_h = (double*)pinned_scratch(1, sizeof(double));   // pinned: the host runs ahead and queues the solve behind the sketch
    FDX_REQUIRE(YtY_h != nullptr, "fit: pinned host buffer");
    *YtY_h = 0.0;
    hipEvent_t evSk = nullptr;
    struct EvGuardSk { hipEvent_t* e; ~EvGuardSk() { if (*e) (void)hipEventDestroy(*e); } } evSk_guard{&evSk};
    const hipStream_t ys = (side && !prm->verbose) ? side : st;
    if (ys != st) {
        FDX_HIP(hipEventCreateWithFlags(&evSk, hipEventDisableTiming));
        FDX_HIP(hipEventRecord(evSk, st));
        FDX_HIP(hipStreamWaitEvent(ys, evSk, 0));
    }
    FDX_TRY(launch_sum_partials(dRowSq.as<double>(), n, dSum.as<double>(), 1, 1, ys));
    FDX_HIP(hipMemcpyAsync(YtY_h, dSum.p, sizeof(double), hipMemcpyDeviceToHost, ys));
    hipEvent_t evY = nullptr;                                  // YtY has arrived (the export is queued on the same stream later)
    struct EvGuardY { hipEvent_t* e; ~EvGuardY() { if (*e) (void)hipEventDestroy(*e); } } evY_guard{&evY};
    if (ys != st) {
        FDX_HIP(hipEventCreateWithFlags(&evY, hipEventDisableTiming));
        FDX_HIP(hipEventRecord(evY, ys));
    }
    if (prm->verbose) FDX_HIP(hipStreamSynchronize(st));
    fit_trace_host("sketch queued, XtX on the host");
    tm.mark();  // 2
    double diag_mean = 0.0;
    for (int k = 0; k < K; ++k) diag_mean += Gh[(size_t)k * K + k];
    diag_mean /= (double)K;
    // auto_tune_lambda (core/spatial.py:181-190): alpha * mean(diag XtX) / max(mean degree, 1), alpha = 0.005
    double lambda = prm->lambda_spatial;
    if (prm->lambda_auto) {
        const double mean_deg = (double)g->nnz / (double)n;
        lambda = 0.005 * diag_mean / std::max(mean_deg, 1.0);
    }

    // ---- solve
    SolveProblem p;
    p.graph = g; p.H = dH.as<double>(); p.ldh = ld; p.XtX = KP != K ? dGp.as<double>() : dG.as<double>();
    p.beta[0] = dB0.as<double>(); p.beta[1] = dB1.as<double>(); p.ld = ld; p.K = KP; p.K_real = K; p.YtY = prm->verbose ? *YtY_h : 0.0;   // verbose: the stream was synchronised above
    p.lambda = lambda; p.rho_eff = prm->rho_sparsity * diag_mean; p.max_iter = prm->max_iter; p.tol = prm->tol;
    p.verbose = prm->verbose;
    p.compute_objective = prm->verbose ? 1 : 0;
    SolveResult r;
    // The export of the result (type-major solver order -> row-major caller order, 0.25 ms at 1M x 30) and the objective pass
    // (0.2 ms) both only read the final abundances: the export goes to the library's side stream (idle here - the leverage
    // job was collected before this call) and runs beside the objective pass instead of after it.
    hipEvent_t evSolved = nullptr, evExported = nullptr;
    struct EvGuard3 { hipEvent_t* a; hipEvent_t* b; ~EvGuard3() { if (*a) (void)hipEventDestroy(*a); if (*b) (void)hipEventDestroy(*b); } } evX_guard{&evSolved, &evExported};
    if (evInit) {
        p.init_beta = 0;                     // done on the side stream at the top
        p.beta0_virtual = beta0_virtual ? 1 : 0;
        FDX_HIP(hipStreamWaitEvent(st, evInit, 0));
    }
    FDX_TRY(solver_run(p, &r, st));          // its chunked read-backs synchronise the stream: YtY has arrived after it
    bool exported = false;
    if ((beta_out_dev || prop_out_dev) && !prm->verbose && !fdx::exp_env("FDX_NO_EXPORT_OVERLAP")) {
        if (side) {
            FDX_HIP(hipEventCreateWithFlags(&evSolved, hipEventDisableTiming));
            FDX_HIP(hipEventCreateWithFlags(&evExported, hipEventDisableTiming));
            FDX_HIP(hipEventRecord(evSolved, st));
            FDX_HIP(hipStreamWaitEvent(side, evSolved, 0));
            FDX_TRY(launch_normalize_export(p.beta[r.result_buffer], ld, row_map, (int)n, g->n_slices, K, beta_out_dev,
                                            prop_out_dev, side));
            FDX_HIP(hipEventRecord(evExported, side));
            exported = true;
        }
    }
    if (!prm->verbose) {
        DevBuf objp, objo;
        FDX_TRY(objp.alloc((size_t)std::max(objective_partials_count(g->n_slices), g->n_tiles) * 4 * sizeof(double)));
        FDX_TRY(objo.alloc(4 * sizeof(double)));
        if (prm->max_iter == 0) FDX_HIP(hipStreamSynchronize(st));
        if (evY) FDX_HIP(hipEventSynchronize(evY));   // long there
        FDX_TRY(solver_objective(*g, p.beta[r.result_buffer], ld, p.H, ld, p.XtX, KP, *YtY_h, lambda, p.rho_eff, objp.as<double>(),
                                 objo.as<double>(), &r.final_objective, st));
    }
    tm.mark();  // 3
    if (exported) FDX_HIP(hipStreamWaitEvent(st, evExported, 0));
    else if (beta_out_dev || prop_out_dev)
        FDX_TRY(launch_normalize_export(p.beta[r.result_buffer], ld, row_map, (int)n, g->n_slices, K, beta_out_dev,
                                        prop_out_dev, st));
    tm.mark();  // 4
    FDX_HIP(hipStreamSynchronize(st));
    abort_drain.armed = false;

    info->solve.converged = r.converged;
    info->solve.n_iterations = r.n_iterations;
    info->solve.final_objective = r.final_objective;
    info->solve.final_change = r.final_change;
    info->solve.n_objectives = (int32_t)r.objectives.size();
    info->solve.sweep_ms = r.sweep_ms;
    info->lambda_used = lambda;
    info->rho_effective = p.rho_eff;
    info->YtY = *YtY_h;
    info->nnz = g->nnz;
    info->graph_ms = tm.ms(0, 1);
    float t_stage = 0.f;
    (void)hipEventElapsedTime(&t_stage, eS0, eS1);
    if (fused) sketch_ms = t_stage;              // the chunked path keeps its per-chunk split of the same interval
    info->sketch_ms = sketch_ms;
    info->gram_ms = gram_ms;
    // contiguous intervals on the caller's stream: [begin, eS0) prologue, [eS0, eS1) sketch -> H, [eS1, mark 3) solve + objective,
    // [mark 3, mark 4) what is left of the export
    float t_pro = 0.f, t_solve = 0.f, t_span = 0.f;
    const bool from_build = g->begin_event && g->begin_stream == st && prm->graph_method == FDX_GRAPH_GIVEN;
    (void)hipEventElapsedTime(&t_pro, from_build ? g->begin_event : tm.ev[0], eS0);
    (void)hipEventElapsedTime(&t_solve, eS1, tm.ev[3]);
    (void)hipEventElapsedTime(&t_span, from_build ? g->begin_event : tm.ev[0], tm.ev[4]);
    info->prologue_ms = t_pro;
    info->span_ms = t_span;
    if (from_build) {       // the build's start event has served its one fit: a later fit on the same handle starts its own clock
        (void)hipEventDestroy(g->begin_event);
        g->begin_event = nullptr;
        g->begin_stream = nullptr;
    }
    info->solve_ms = t_solve;
    info->finish_ms = tm.ms(3, 4);
    info->total_ms = tm.ms(0, 4);
    info->solve.total_ms = info->total_ms;
    if (objectives_out)
        for (size_t t = 0; t < r.objectives.size(); ++t) objectives_out[t] = r.objectives[t];
    if (rel_changes_out)
        for (size_t t = 0; t < r.rel_changes.size(); ++t) rel_changes_out[t] = r.rel_changes[t];
    return 0;
}

}  // namespace

extern "C" int fdx_fit_dev(const void* Y_dev, int32_t y_dtype, int64_t n, int32_t G, int64_t ldy, const double* X, int32_t K,
                           const int32_t* bucket, const double* weight_y, const double* weight_x, const double* coords_dev,
                           int32_t dim, const fdx_fit_params* prm, fdx_graph** graph_inout, double* beta_out_dev,
                           double* prop_out_dev, double* objectives_out, double* rel_changes_out, fdx_fit_info* info,
                           void* stream) {
    FDX_REQUIRE(Y_dev != nullptr, "fdx_fit_dev: null array");
    YSource ys;
    ys.dense = Y_dev;
    ys.dtype = y_dtype;
    ys.ldy = ldy;
    return fit_impl(ys, n, G, X, K, bucket, weight_y, weight_x, coords_dev, dim, prm, graph_inout, beta_out_dev, prop_out_dev,
                    objectives_out, rel_changes_out, info, stream);
}

static int csr_view_ok(const fdx_csr_view* Y, const char* who) {
    FDX_REQUIRE(Y != nullptr, std::string(who) + ": null matrix");
    FDX_REQUIRE(Y->n >= 0 && Y->nnz >= 0 && Y->G > 0, std::string(who) + ": bad CSR shape");
    FDX_REQUIRE(Y->indptr != nullptr && (Y->nnz == 0 || (Y->indices && Y->data)), std::string(who) + ": null CSR array");
    FDX_REQUIRE(Y->dtype == FDX_F32 || Y->dtype == FDX_F64, std::string(who) + ": dtype must be FDX_F32 or FDX_F64");
    return 0;
}

extern "C" int fdx_csr_check_dev(const fdx_csr_view* Y, void* stream) {
    FDX_TRY(csr_view_ok(Y, "fdx_csr_check_dev"));
    hipStream_t st = (hipStream_t)stream;
    PoolStream pool_stream(st);
    DevBuf flag;
    FDX_TRY(flag.alloc(sizeof(int)));
    FDX_TRY(launch_csr_check((const long long*)Y->indptr, Y->indices, Y->n, Y->nnz, Y->G, Y->sorted_rows, flag.as<int>(), st));
    int bad = 0;
    FDX_HIP(hipMemcpyAsync(&bad, flag.p, sizeof(int), hipMemcpyDeviceToHost, st));
    FDX_HIP(hipStreamSynchronize(st));
    FDX_REQUIRE((bad & 1) == 0, "CSR matrix is malformed (indptr not monotone from 0 to nnz, or a column index outside [0, G))");
    FDX_REQUIRE((bad & 2) == 0, "CSR matrix claims sorted_rows but a row's column indices are not ascending");
    return 0;
}

extern "C" int fdx_csr_gene_moments_dev(const fdx_csr_view* Y, double* mean_out_host, double* var_out_host,
                                        double* colsum_out_host, void* stream) {
    FDX_TRY(csr_view_ok(Y, "fdx_csr_gene_moments_dev"));
    FDX_REQUIRE(Y->n > 0, "fdx_csr_gene_moments_dev: empty matrix");
    hipStream_t st = (hipStream_t)stream;
    PoolStream pool_stream(st);
    const size_t G = (size_t)Y->G;
    DevBuf scale, part, out;
    const int ns = colsum_out_host ? 3 : 2;
    FDX_TRY(scale.alloc((size_t)Y->n * sizeof(double)));
    FDX_TRY(part.alloc((size_t)csr_moment_stripes(Y->n) * ns * G * sizeof(double)));
    FDX_TRY(out.alloc(3 * G * sizeof(double)));
    double* o = out.as<double>();
    FDX_TRY(launch_csr_moments((const long long*)Y->indptr, Y->indices, Y->data, Y->dtype, Y->n, Y->nnz, Y->G, scale.as<double>(),
                               part.as<double>(), o, o + G, colsum_out_host ? o + 2 * G : nullptr, Y->sorted_rows != 0, st));
    if (mean_out_host) FDX_TRY(copy_d2h(mean_out_host, o, G * sizeof(double), st));
    if (var_out_host) FDX_TRY(copy_d2h(var_out_host, o + G, G * sizeof(double), st));
    if (colsum_out_host) FDX_TRY(copy_d2h(colsum_out_host, o + 2 * G, G * sizeof(double), st));
    FDX_HIP(hipStreamSynchronize(st));
    return 0;
}

extern "C" int fdx_fit_csr_dev(const fdx_csr_view* Y, const int32_t* gene_idx, int32_t G, const double* X, int32_t K,
                               const int32_t* bucket, const double* weight_y, const double* weight_x,
                               const double* coords_dev, int32_t dim, const fdx_fit_params* prm, fdx_graph** graph_inout,
                               double* beta_out_dev, double* prop_out_dev, double* objectives_out, double* rel_changes_out,
                               fdx_fit_info* info, void* stream) {
    FDX_TRY(csr_view_ok(Y, "fdx_fit_csr_dev"));
    FDX_REQUIRE(gene_idx != nullptr || G == Y->G, "fdx_fit_csr_dev: gene_idx may be NULL only when G equals the matrix width");
    FDX_REQUIRE(prm != nullptr, "fdx_fit_csr_dev: null argument");
    FDX_REQUIRE(prm->mode_y == FDX_PRE_RAW || prm->mode_y == FDX_PRE_LOG_CPM_SPARSE,
                "fdx_fit_csr_dev: mode_y must be FDX_PRE_RAW or FDX_PRE_LOG_CPM_SPARSE");
    YSource ys;
    ys.csr = Y;
    ys.gene_idx = gene_idx;
    return fit_impl(ys, Y->n, G, X, K, bucket, weight_y, weight_x, coords_dev, dim, prm, graph_inout, beta_out_dev,
                    prop_out_dev, objectives_out, rel_changes_out, info, stream);
}
