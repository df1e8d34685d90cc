// Block-coordinate-descent sweep for the graph-regularised NNLS (the hot loop).
//
// Replaces the reference's numba kernels
//   flashdeconv/core/solver.py:104-184  _bcd_iteration_fused  (Jacobi over spots, prange)
//   flashdeconv/core/solver.py:29-101   update_spot_with_Xty  (Gauss-Seidel over cell types)
//   flashdeconv/core/solver.py:18-26    soft_threshold
// and the host-side convergence reduction of core/solver.py:395-397.
//
// Mapping (gfx950, wave64): ONE LANE = ONE SPOT, one wavefront = one 64-spot slice of the sliced-ELL
// graph.  The K abundances of the spot live in VGPRs for the whole sweep (the coordinate steps are
// sequential in k, so a lane-per-type mapping would leave 63/64 of the VALU idle - see DESIGN.md).
// beta and H are stored type-major ("SoA", (K, ld)) so that every own-row access of a wave is one
// fully coalesced 512-byte transaction; neighbour rows are gathered from the same planes and are
// served by L2 because the spots are in Morton order.  XtX (K x K) is wave-uniform and is read
// through the scalar cache (s_load) straight into the SGPR operand of v_fma_f64.
//
// Arithmetic per spot (float64, IEEE division, no fast-math):
//   nbr_k  = sum_{j in N(i)} beta_in[j,k]                      (CSR order, padded with exact +0.0)
//   r_k    = sum_j XtX[k,j] * b_j       with b_j already updated for j < k   (maintained residual of
//            solver.py:72,96-99 evaluated on demand: same value, K^2 instead of 1.5 K^2 FMAs, no r[] array)
//   res    = H[k,i] - r_k + XtX[k,k]*b_k (+ lambda*nbr_k if deg>0)          (solver.py:79-83)
//   b_k    = den > 1e-10 ? max(0, soft(res, rho)/den) : 0 ,  den = XtX[k,k] + lambda*deg  (solver.py:86-93)
// Convergence statistics max_i max_k|b_new-b_old| and max_i max_k|b_old| (solver.py:173-184) are reduced
// with wave shuffles and one integer atomicMax per wave into 64 slots (order-free, hence deterministic).
// The NEXT sweep (or the finishing kernel) folds the 64 slots and evaluates
//   rel_change = max_diff / (max_abs_old + 1e-10) < tol                       (solver.py:395-397,409)
// on the device, so a converged solve turns the already-queued sweeps into no-ops without a host round trip.
#include "bcd_device.h"

namespace fdx {

// Generic-K sweep (K > FDX_MAX_K_FAST): same arithmetic with the per-spot vectors in a per-lane slice of
// a global scratch buffer laid out (2K, n_pad) type-major so accesses stay coalesced.  Correct for any K;
// not tuned (cell-type panels beyond 64 types are rare).
__global__ __launch_bounds__(256) void bcd_sweep_generic_kernel(BcdSweepArgs a, double* scratch, size_t sld) {
    const int lane = threadIdx.x & 63;
    const int wib = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);  // wave-uniform, provably so
    const int K = a.K;
    if (a.it > 0) {
        const double rc = fold_rel_change(a.stats + (size_t)(a.it - 1) * 128, lane);
        if (blockIdx.x == 0 && threadIdx.x == 0) a.rel_change[a.it - 1] = rc;
        if (rc < a.tol) return;
    }
    const int slice = xcd_remap(blockIdx.x, gridDim.x) * 4 + wib;
    if (slice >= a.n_slices) return;
    const int i = slice * 64 + lane;
    const bool active = i < a.n;
    const int ii = active ? i : a.n - 1;
    const size_t ld = (size_t)a.ld;
    const int w0 = a.slice_off[slice];
    const int w = a.slice_off[slice + 1] - w0;
    double* bb = scratch + (size_t)(slice * 64 + lane);  // b_k at bb[k*sld], c_k at bb[(K+k)*sld]
    const int* ell = a.ell + (size_t)w0 * 64 + lane;
    const int dg = a.deg[ii];
    const double lam_deg = a.lambda * (double)dg;
    for (int k = 0; k < K; ++k) {
        double nb = 0.0;
        for (int m = 0; m < w; ++m) nb += a.beta_in[k * ld + ell[(size_t)m * 64]];
        const double h = a.H[k * (size_t)a.ldh + ii];
        bb[k * sld] = a.beta_in[k * ld + ii];
        bb[(K + k) * sld] = h + ((dg > 0) ? a.lambda * nb : 0.0);
    }
    double dmax = 0.0, amax = 0.0;
    for (int k = 0; k < K; ++k) {
        const double* g = a.XtX + (size_t)k * K;
        double r0 = 0.0, r1 = 0.0;
        int j = 0;
        for (; j + 1 < K; j += 2) {
            r0 = fma(g[j], bb[j * sld], r0);
            r1 = fma(g[j + 1], bb[(j + 1) * sld], r1);
        }
        if (j < K) r0 = fma(g[j], bb[j * sld], r0);
        const double gkk = g[k];
        const double old = bb[k * sld];
        const double res = bb[(K + k) * sld] - (r0 + r1) + gkk * old;
        const double den = gkk + lam_deg;
        double nw = 0.0;
        if (den > 1e-10) {
            const double st = res > a.rho ? res - a.rho : (res < -a.rho ? res + a.rho : 0.0);
            nw = fmax(0.0, st / den);
        }
        dmax = fmax(dmax, fabs(nw - old));
        amax = fmax(amax, fabs(old));
        bb[k * sld] = nw;
        if (active) a.beta_out[k * ld + i] = nw;
    }
    if (!active) { dmax = 0.0; amax = 0.0; }
    dmax = wave_max(dmax);
    amax = wave_max(amax);
    if (lane == 0) {
        unsigned long long* s = a.stats + (size_t)a.it * 128;
        const int slot = slice & 63;
        atomicMax(s + slot, (unsigned long long)__double_as_longlong(dmax));
        atomicMax(s + 64 + slot, (unsigned long long)__double_as_longlong(amax));
    }
}

// Evaluates the stopping rule for the last queued sweep (no later sweep exists to do it).
__global__ void bcd_fold_last_kernel(const unsigned long long* stats, double* rel_change, int it) {
    const double rc = fold_rel_change(stats + (size_t)it * 128, threadIdx.x & 63);
    if (threadIdx.x == 0) rel_change[it] = rc;
}

bool bcd_sweep_dispatch_part0(const BcdSweepArgs&, hipStream_t);
bool bcd_sweep_dispatch_part1(const BcdSweepArgs&, hipStream_t);
bool bcd_sweep_dispatch_part2(const BcdSweepArgs&, hipStream_t);
bool bcd_sweep_dispatch_part3(const BcdSweepArgs&, hipStream_t);
bool bcd_sweep_dispatch_part4(const BcdSweepArgs&, hipStream_t);
bool bcd_sweep_dispatch_part5(const BcdSweepArgs&, hipStream_t);
bool bcd_sweep_dispatch_part6(const BcdSweepArgs&, hipStream_t);
bool bcd_sweep_dispatch_part7(const BcdSweepArgs&, hipStream_t);

int launch_bcd_sweep(const BcdSweepArgs& a, double* generic_scratch, size_t scratch_ld, hipStream_t st) {
    if (a.n <= 0 || a.n_slices <= 0) return 0;
    if (sweep_instantiated(a.K)) {
        const bool hit = bcd_sweep_dispatch_part0(a, st) || bcd_sweep_dispatch_part1(a, st) ||
                         bcd_sweep_dispatch_part2(a, st) || bcd_sweep_dispatch_part3(a, st) ||
                         bcd_sweep_dispatch_part4(a, st) || bcd_sweep_dispatch_part5(a, st) ||
                         bcd_sweep_dispatch_part6(a, st) || bcd_sweep_dispatch_part7(a, st);
        if (!hit) return fail(FDX_ERR_INTERNAL, "bcd sweep dispatch failed");
    } else {
        if (!generic_scratch) return fail(FDX_ERR_INVALID, "generic BCD sweep needs a scratch buffer");
        const int nblk = ceil_div(a.n_slices, 4);
        hipLaunchKernelGGL(bcd_sweep_generic_kernel, dim3(nblk), dim3(256), 0, st, a, generic_scratch, scratch_ld);
    }
    FDX_CHECK_LAUNCH();
    return 0;
}

// Will launch_bcd_sweep run the LDS-tiled kernel for these arguments?  (The tile lists of the sharded solve are only
// honoured by that kernel: the global-gather fallback sweeps every spot.)  Mirrors launch_k in bcd_sweep_inst.cpp with the
// upper bound of sweep_chunk(K) - a conservative "no" costs the boundary / interior overlap, never correctness.
bool bcd_sweep_uses_tiles(const BcdSweepArgs& a) {
    if (!a.tiled || !sweep_instantiated(a.K)) return false;
    const int KC = a.K < 8 ? a.K : 8;
    return (size_t)KC * (256 + a.halo_max + 1) * sizeof(double) <= 64 * 1024 && (long long)a.ld * 8 < (1LL << 32) &&
           (long long)a.ldh * 8 < (1LL << 32);
}

int launch_bcd_objective_tiled(const BcdSweepArgs& a0, double* partials, hipStream_t st) {
    if (!a0.tiled || !sweep_instantiated(a0.K)) return 1;
    const int KC = a0.K < 8 ? a0.K : 8;   // upper bound of sweep_chunk(K)
    if ((size_t)KC * (256 + a0.halo_max + 1) * sizeof(double) > 64 * 1024) return 1;
    BcdSweepArgs a = a0;
    a.objective = 1;
    a.rel_change = partials;
    a.it = 0;
    const bool hit = bcd_sweep_dispatch_part0(a, st) || bcd_sweep_dispatch_part1(a, st) || bcd_sweep_dispatch_part2(a, st) ||
                     bcd_sweep_dispatch_part3(a, st) || bcd_sweep_dispatch_part4(a, st) || bcd_sweep_dispatch_part5(a, st) ||
                     bcd_sweep_dispatch_part6(a, st) || bcd_sweep_dispatch_part7(a, st);
    if (!hit) return 1;
    FDX_CHECK_LAUNCH();
    return 0;
}

int launch_bcd_fold_last(const unsigned long long* stats, double* rel_change, int it, hipStream_t st) {
    hipLaunchKernelGGL(bcd_fold_last_kernel, dim3(1), dim3(64), 0, st, stats, rel_change, it);
    FDX_CHECK_LAUNCH();
    return 0;
}

}  // namespace fdx
