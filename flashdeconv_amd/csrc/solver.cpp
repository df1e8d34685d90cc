// Host-side driver of the BCD solve (replaces the Python loop of flashdeconv/core/solver.py:287-428).
//
// All iteration state stays in HBM.  Sweeps are queued back to back in growing chunks; the stopping rule is
// evaluated on the device by the next sweep's prologue (bcd_kernels.cpp), so the host only reads back the
// rel_change trace once per chunk and sweeps queued past convergence retire as no-ops.
#include "fdx_env.h"
#include "solver.h"

#include <algorithm>
#include <cmath>
#include <cstdlib>
#include <vector>

namespace fdx {

__global__ void fill_beta_kernel(double* b, long long ld, long long n_fill, int K, double value, int K_planes) {
    const long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x;
    if (i >= ld) return;
    const double v = (i < n_fill) ? value : 0.0;  // pad rows (incl. the all-zero neighbour row) stay exactly 0
    for (int k = 0; k < K; ++k) b[(size_t)k * ld + i] = v;
    for (int k = K; k < K_planes; ++k) b[(size_t)k * ld + i] = 0.0;   // pad types (solver_padded_K)
}

__global__ void pad_square_kernel(const double* __restrict__ A, int K, double* __restrict__ B, int KP) {
    const int e = blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= KP * KP) return;
    const int i = e / KP, j = e - i * KP;
    B[e] = (i < K && j < K) ? A[i * K + j] : 0.0;
}

// zero the pad columns [n_used, ld) of a type-major (K, ld) array: everything below n_used is written by its producer
__global__ void zero_pad_kernel(double* b, long long ld, long long n_used, int K) {
    const long long i = n_used + blockIdx.x * (long long)blockDim.x + threadIdx.x;
    if (i >= ld) return;
    for (int k = 0; k < K; ++k) b[(size_t)k * ld + i] = 0.0;
}

int solver_zero_pad(double* b, long long ld, long long n_used, int K, hipStream_t st) {
    if (ld <= n_used || K <= 0) return 0;
    hipLaunchKernelGGL(zero_pad_kernel, dim3(ceil_div(ld - n_used, 256)), dim3(256), 0, st, b, ld, n_used, K);
    FDX_CHECK_LAUNCH();
    return 0;
}

int solver_init_beta(double* beta, long long ld, long long n_fill, int K, hipStream_t st, int K_planes) {
    if (ld <= 0 || K <= 0) return 0;
    hipLaunchKernelGGL(fill_beta_kernel, dim3(ceil_div(ld, 256)), dim3(256), 0, st, beta, ld, n_fill, K, 1.0 / (double)K, K_planes);
    FDX_CHECK_LAUNCH();
    return 0;
}

int solver_pad_square(const double* A, int K, double* B, int KP, hipStream_t st) {
    hipLaunchKernelGGL(pad_square_kernel, dim3(ceil_div((long long)KP * KP, 256)), dim3(256), 0, st, A, K, B, KP);
    FDX_CHECK_LAUNCH();
    return 0;
}

int solver_objective_partials(const fdx_graph& g, const double* beta, long long ld, const double* H, long long ldh,
                              const double* XtX, int K, double* scratch_partials, double* out4_dev, hipStream_t st) {
    // the four sums of compute_objective (core/solver.py:269-284): <H,beta>, beta' XtX beta, smoothness, |beta|_1
    int nblk = objective_partials_count(g.n_slices);
    int rc_t = 1;
    if (g.tiled && !fdx::env("FDX_NO_TILED")) {          // same LDS-tiled traversal as the sweep (n_tiles <= nblk partial rows)
        BcdSweepArgs a{};
        a.H = H; a.XtX = XtX; a.beta_in = beta; a.beta_out = nullptr; a.ell = g.ell.as<int>();
        a.slice_off = g.slice_off.as<int>(); a.deg = g.deg.as<int>(); a.stats = nullptr; a.rel_change = nullptr;
        a.lambda = 0.0; a.rho = 0.0; a.tol = 0.0; a.ldh = (int)ldh; a.ld = (int)ld; a.n = (int)g.n;
        a.n_slices = g.n_slices; a.K = K; a.tiled = 1; a.ell_local = g.ell_local.as<unsigned short>();
        a.tile_halo = g.tile_halo.as<int>(); a.tile_hcnt = g.tile_hcnt.as<int>(); a.n_tiles = g.n_tiles; a.halo_max = g.halo_max;
        // the padded sizes above 64 types: the traversal without the K^2 products, the quadratic term as a Gram matrix by MFMA
        a.skip_quad = (K > FDX_MAX_K_FAST && sweep_instantiated(K)) ? 1 : 0;
        rc_t = launch_bcd_objective_tiled(a, scratch_partials, st);
        if (rc_t < 0) return rc_t;
        if (rc_t == 0) {
            nblk = g.n_tiles;
            if (a.skip_quad) FDX_TRY(launch_beta_quad(beta, ld, g.n, K, XtX, scratch_partials, g.n_tiles, st));   // fills the zeros the traversal left in column 1
        }
    }
    if (rc_t != 0) {
        // no objective traversal for this K (97 types and more): the generic kernel without its K^2 reads per spot - the quadratic
        // term from the Gram matrix of the abundances (9 -> 3 ms at 100 types and 500k spots)
        const int skip_quad = K > FDX_MAX_K_FAST ? 1 : 0;
        FDX_TRY(launch_objective_partials(beta, ld, H, ldh, XtX, g.ell.as<int>(), g.slice_off.as<int>(), g.deg.as<int>(),
                                          (int)g.n, g.n_slices, K, scratch_partials, st, skip_quad));
        if (skip_quad) FDX_TRY(launch_beta_quad(beta, ld, g.n, K, XtX, scratch_partials, nblk, st));
    }
    return launch_sum_partials(scratch_partials, nblk, out4_dev, 4, 4, st);
}

int solver_objective(const fdx_graph& g, const double* beta, long long ld, const double* H, long long ldh,
                     const double* XtX, int K, double YtY, double lambda, double rho_eff, double* scratch_partials,
                     double* scratch_out4, double* obj_host, hipStream_t st) {
    FDX_TRY(solver_objective_partials(g, beta, ld, H, ldh, XtX, K, scratch_partials, scratch_out4, st));
    double r[4];
    FDX_HIP(hipMemcpyAsync(r, scratch_out4, sizeof(r), hipMemcpyDeviceToHost, st));
    FDX_HIP(hipStreamSynchronize(st));
    *obj_host = 0.5 * (YtY - 2.0 * r[0] + r[1]) + 0.5 * lambda * r[2] + rho_eff * r[3];
    return 0;
}

int solver_run(const SolveProblem& p, SolveResult* res, hipStream_t st) {
    const fdx_graph& g = *p.graph;
    const int K = p.K;
    res->result_buffer = 0;
    res->n_iterations = 0;
    res->converged = 0;
    res->final_change = 0.0;
    res->final_objective = 0.0;
    res->objective_iters.clear();
    res->objectives.clear();
    res->rel_changes.clear();
    res->sweep_ms = 0.0;
    if (g.n == 0 || K == 0) {  // core/solver.py:334-343
        res->converged = 1;
        return 0;
    }
    FDX_REQUIRE(p.ld >= g.n_total + 1, "solver: ld must cover owned + halo spots + the zero pad row");
    FDX_REQUIRE(p.max_iter >= 0, "solver: max_iter must be >= 0");

    const int max_iter = p.max_iter;
    DevBuf stats, relchg, obj_partials, obj_out, generic_scratch, obj_trace;
    // the max slots of every iteration and, behind them, the rel_change trace: one block, one fill
    const size_t stats_bytes = (size_t)std::max(max_iter, 1) * 128 * sizeof(unsigned long long);
    FDX_TRY(stats.alloc(stats_bytes + (size_t)std::max(max_iter, 1) * sizeof(double)));
    FDX_TRY(obj_partials.alloc((size_t)objective_partials_count(g.n_slices) * 4 * sizeof(double)));
    FDX_TRY(obj_out.alloc(4 * sizeof(double)));
    FDX_HIP(hipMemsetAsync(stats.p, 0, stats.bytes, st));
    double* const relchg_p = reinterpret_cast<double*>(static_cast<char*>(stats.p) + stats_bytes);
    size_t scratch_ld = 0;
    if (sweep_uses_lds(K)) {                          // the LDS-resident sweep reads XtX with its rows padded to 16
        FDX_TRY(generic_scratch.alloc(sweep_lds_pad_doubles(K) * sizeof(double)));
        FDX_TRY(sweep_lds_prepare(p.XtX, K, generic_scratch.as<double>(), st));
    } else if (!sweep_instantiated(K)) {
        scratch_ld = (size_t)g.n_slices * 64;
        FDX_TRY(generic_scratch.alloc(scratch_ld * 2 * K * sizeof(double)));
    }
    const int K_real = p.K_real > 0 ? p.K_real : K;
    if (p.init_beta) FDX_TRY(solver_init_beta(p.beta[0], p.ld, g.n_total, K_real, st, K));   // beta0 = 1/K (solver.py:372)
    // the second buffer's pad rows must also read as zero
    // (only the pad: every real row is written by the first sweep before anything reads it - a memset of the whole
    // buffer was 30 us of a 6 ms fit)
    if (p.init_beta) FDX_TRY(solver_zero_pad(p.beta[1], p.ld, g.n_total, K, st));

    BcdSweepArgs a{};
    a.H = p.H; a.XtX = p.XtX; a.ell = g.ell.as<int>(); a.slice_off = g.slice_off.as<int>(); a.deg = g.deg.as<int>();
    a.stats = stats.as<unsigned long long>(); a.rel_change = relchg_p;
    a.lambda = p.lambda; a.rho = p.rho_eff; a.tol = p.tol; a.ldh = (int)p.ldh; a.ld = (int)p.ld; a.n = (int)g.n;
    a.n_slices = g.n_slices; a.K = K;
    if (g.tiled && !fdx::env("FDX_NO_TILED")) {
        a.tiled = 1; a.ell_local = g.ell_local.as<unsigned short>(); a.tile_halo = g.tile_halo.as<int>();
        a.tile_hcnt = g.tile_hcnt.as<int>(); a.n_tiles = g.n_tiles; a.halo_max = g.halo_max;
    }

    // the start vector as a constant of the first sweep (bcd_sweep_inst.cpp, INIT) - or written now, where that kernel does not apply
    double init_uniform = 0.0;
    if (!p.init_beta && p.beta0_virtual) {
        BcdSweepArgs probe = a;
        probe.beta_in = p.beta[0];
        probe.beta_out = p.beta[1];
        if (max_iter > 0 && K_real == K && K <= FDX_MAX_K_FAST && g.n_total == g.n && bcd_sweep_uses_tiles(probe) && !fdx::env("FDX_NO_INIT_SWEEP"))
            init_uniform = 1.0 / (double)K;
        else
            FDX_TRY(solver_init_beta(p.beta[0], p.ld, g.n_total, K_real, st, K));
    }

    // per chunk: events around its sweeps (alternating pairs: the next chunk's first sweeps are queued before this chunk's
    // timing is read) and one after the trace copy the host waits on
    hipEvent_t ev0[2] = {nullptr, nullptr}, ev1[2] = {nullptr, nullptr}, evCopy = nullptr;
    struct EvGuard { hipEvent_t* e[5]; ~EvGuard() { for (auto* q : e) if (*q) (void)hipEventDestroy(*q); } } ev_guard{{&ev0[0], &ev0[1], &ev1[0], &ev1[1], &evCopy}};
    for (int j = 0; j < 2; ++j) {
        FDX_HIP(hipEventCreate(&ev0[j]));
        FDX_HIP(hipEventCreate(&ev1[j]));
    }
    FDX_HIP(hipEventCreateWithFlags(&evCopy, hipEventDisableTiming));
    double sweep_ms_acc = 0.0;   // GPU time of the queued sweeps only (host read-back gaps between chunks excluded)

    // the trace lands in pinned host memory: a copy into pageable memory returns only when it has been done, i.e. it would make
    // the host wait for the chunk before it can queue anything behind it
    const size_t rc_count = (size_t)std::max(max_iter, 1);
    double* rc_host = (double*)pinned_scratch(2, rc_count * sizeof(double));
    if (!rc_host) return fail(FDX_ERR_HIP, "solver: pinned host buffer");
    for (size_t j = 0; j < rc_count; ++j) rc_host[j] = 0.0;
    int done = 0;          // iterations whose rel_change is known on the host
    int n_iter = 0;
    bool converged = false;
    int chunk = p.first_chunk > 0 ? p.first_chunk : 4;
    // The host reads the rel_change trace once per chunk; so that the device does not idle during that round trip (~45 us,
    // a quarter of a sweep, per chunk) the first sweeps of the NEXT chunk are queued before the host waits: if this chunk
    // converged they are no-ops like every sweep past convergence (device-side stopping rule), otherwise they are simply early.
    const int n_ahead = (p.verbose || fdx::exp_env("FDX_NO_SWEEP_AHEAD")) ? 0 : 2;
    int queued_ahead = 0;  // sweeps of the current chunk that were queued during the previous chunk's read-back
    int ci = 0;            // chunk counter (event pair = ci & 1)
    // verbose objective trace: evaluated on the NEW buffer at it % 10 == 0 or it == max_iter-1 (solver.py:399-404)
    std::vector<std::pair<int, double>> trace;
    auto queue_sweep = [&](int it) -> int {
        a.it = it;
        a.beta_in = p.beta[it & 1];
        a.beta_out = p.beta[(it + 1) & 1];
        a.init_uniform = it == 0 ? init_uniform : 0.0;
        FDX_TRY(launch_bcd_sweep(a, generic_scratch.as<double>(), scratch_ld, st));
        if (p.verbose && (it % 10 == 0 || it == max_iter - 1)) {
            double obj = 0.0;  // synchronous; verbose mode trades speed for the trace, as the reference does
            FDX_TRY(solver_objective(g, a.beta_out, p.ld, p.H, p.ldh, p.XtX, K, p.YtY, p.lambda, p.rho_eff,
                                     obj_partials.as<double>(), obj_out.as<double>(), &obj, st));
            trace.emplace_back(it, obj);
        }
        return 0;
    };
    while (done < max_iter && !converged) {
        const int end = std::min(max_iter, done + chunk);
        const int pair = ci & 1;
        if (queued_ahead == 0) FDX_HIP(hipEventRecord(ev0[pair], st));          // otherwise recorded ahead of the early sweeps
        for (int it = done + queued_ahead; it < end; ++it) FDX_TRY(queue_sweep(it));
        FDX_TRY(launch_bcd_fold_last(a.stats, a.rel_change, end - 1, st));
        FDX_HIP(hipEventRecord(ev1[pair], st));
        FDX_HIP(hipMemcpyAsync(rc_host + done, relchg_p + done, (size_t)(end - done) * sizeof(double),
                               hipMemcpyDeviceToHost, st));
        FDX_HIP(hipEventRecord(evCopy, st));
        int ahead = 0;
        if (end < max_iter && n_ahead > 0) {
            ahead = std::min(n_ahead, max_iter - end);
            FDX_HIP(hipEventRecord(ev0[pair ^ 1], st));
            for (int it = end; it < end + ahead; ++it) FDX_TRY(queue_sweep(it));
        }
        FDX_HIP(hipEventSynchronize(evCopy));
        {
            float ms_chunk = 0.f;
            FDX_HIP(hipEventElapsedTime(&ms_chunk, ev0[pair], ev1[pair]));
            sweep_ms_acc += ms_chunk;
        }
        for (int it = done; it < end; ++it) {
            n_iter = it + 1;
            if (rc_host[it] < p.tol) { converged = true; break; }   // solver.py:409-413
        }
        done = end;
        queued_ahead = ahead;
        // 4, 4, 8, 16, 32, 32, ...: a solve that converges in 5-8 sweeps (the gaussian family: 7) retires one no-op sweep instead of five
        chunk = std::max(std::min(ci == 0 ? chunk : chunk * 2, 32), ahead);
        ++ci;
    }
    res->sweep_ms = sweep_ms_acc;

    res->n_iterations = n_iter;                        // iteration + 1 (solver.py:422); 0 when max_iter == 0
    res->converged = converged ? 1 : 0;
    res->final_change = n_iter > 0 ? rc_host[n_iter - 1] : 0.0;
    res->result_buffer = n_iter & 1;                   // sweep `it` writes buffer (it+1)&1
    res->rel_changes.assign(rc_host, rc_host + n_iter);
    for (auto& t : trace)
        if (t.first < n_iter) { res->objective_iters.push_back(t.first); res->objectives.push_back(t.second); }
    if (p.compute_objective)
        FDX_TRY(solver_objective(g, p.beta[res->result_buffer], p.ld, p.H, p.ldh, p.XtX, K, p.YtY, p.lambda, p.rho_eff,
                                 obj_partials.as<double>(), obj_out.as<double>(), &res->final_objective, st));
    return 0;
}

}  // namespace fdx
