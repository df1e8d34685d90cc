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
#include <algorithm>

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

// More than 64 cell types: the abundances of a 64-spot slice live in LDS (KP x 64 doubles, KP = K rounded up to 16; one workgroup =
// one wave), each lane reading and writing only its own column - LDS as indexed per-lane storage, no barrier anywhere - and the
// coordinate loop is ROLLED.  The register-resident kernels need every loop over the types unrolled (static register indices): K^2
// FMAs of straight-line code, which leaves the 64 KB instruction cache at 72 types (0.41 ms per sweep and 500k spots at 64 types, 0.74
// at 72), runs one wave per SIMD from 112 on (2 K registers for the abundances alone) and spills.  Here a sweep costs K^2
// ds_read_b64 per spot and the code is the same few hundred bytes for every K.  Per 16 types: one row piece of XtX by scalar loads
// (its rows are padded with zeros to KP, so nothing is clamped or predicated: a first version with per-element index clamps spent ten
// scalar instructions per product), 16 LDS reads with immediate offsets, 16 FMAs in the two chains.  The neighbour values and H of type
// k + 1 are requested before the products of type k.  Arithmetic and summation order are those of bcd_sweep_tiled_kernel (even / odd
// residual association; four chains over the types mod 4 where that kernel has two); the pad types' products are fma(0, 0, r) = r.
// (The arrays are separate __restrict__ parameters: only then does the compiler fetch the wave-uniform row of XtX with scalar loads.)
__global__ __launch_bounds__(64) void bcd_sweep_lds_kernel(
    const double* __restrict__ H, const double* __restrict__ Gp /* (K, KP) */, const double* __restrict__ beta_in,
    double* __restrict__ beta_out, const int* __restrict__ ell_base, const int* __restrict__ slice_off, const int* __restrict__ deg,
    unsigned long long* __restrict__ stats, double* __restrict__ rel_change, const double lambda, const double rho, const double tol,
    const int ldh_, const int ld_, const int n, const int K, const int KP, const int it) {
    extern __shared__ __attribute__((aligned(16))) double bl[];            // [KP][64]
    const int lane = threadIdx.x;
    if (it > 0) {
        const double rc = fold_rel_change(stats + (size_t)(it - 1) * 128, lane);
        if (blockIdx.x == 0 && lane == 0) rel_change[it - 1] = rc;
        if (rc < tol) return;
    }
    const int slice = xcd_remap(blockIdx.x, gridDim.x);
    const int i = min(slice * 64 + lane, n - 1);                            // lanes past the last spot mirror spot n-1
    const bool active = slice * 64 + lane < n;
    const size_t ld = (size_t)ld_, ldh = (size_t)ldh_;
    const int w0 = slice_off[slice];
    const int w = slice_off[slice + 1] - w0;                                // wave-uniform ELL width of this slice
    const int* ell = ell_base + (size_t)w0 * 64 + (i & 63);
    const int dg = deg[i];
    const double lam_deg = lambda * (double)dg;
    const double lam_eff = (dg > 0) ? lambda : 0.0;
    int nbi[16];
#pragma unroll
    for (int m = 0; m < 16; ++m) nbi[m] = (m < w) ? ell[(size_t)m * 64] : 0;
    double* bcol = bl + lane;
    for (int k = 0; k < K; ++k) bcol[k * 64] = beta_in[k * ld + i];
    for (int k = K; k < KP; ++k) bcol[k * 64] = 0.0;
    double nv[16], hn = H[i];
#pragma unroll
    for (int m = 0; m < 16; ++m) nv[m] = (m < w) ? beta_in[nbi[m]] : 0.0;
    double dmax = 0.0, amax = 0.0;
    for (int k = 0; k < K; ++k) {
        double c = 0.0;
#pragma unroll
        for (int m = 0; m < 16; ++m) if (m < w) c += nv[m];
        for (int m = 16; m < w; ++m) c += beta_in[k * ld + ell[(size_t)m * 64]];
        const double h = hn;
        if (k + 1 < K) {                                                    // type k + 1: requested now, used in the next trip
            const double* bn = beta_in + (size_t)(k + 1) * ld;
#pragma unroll
            for (int m = 0; m < 16; ++m) nv[m] = (m < w) ? bn[nbi[m]] : 0.0;
            hn = H[(size_t)(k + 1) * ldh + i];
        }
        const double* g = Gp + (size_t)k * KP;
        // four chains (types mod 4) and the next 16 types' loads issued ahead of the current 16 products: one wave per SIMD is all
        // the LDS leaves room for, so nothing else hides the LDS / scalar-load latency and the FMA chain's own
        double q0 = 0.0, q1 = 0.0, q2 = 0.0, q3 = 0.0;
        double gs[16], bv[16];
#pragma unroll
        for (int u = 0; u < 16; ++u) { gs[u] = g[u]; bv[u] = bcol[u * 64]; }
        for (int j0 = 16; j0 < KP; j0 += 16) {
            const double* gp = g + j0;
            const double* bp = bcol + j0 * 64;
            double gn[16], bn2[16];
#pragma unroll
            for (int u = 0; u < 16; ++u) gn[u] = gp[u];
#pragma unroll
            for (int u = 0; u < 16; ++u) bn2[u] = bp[u * 64];
#pragma unroll
            for (int u = 0; u < 16; u += 4) {
                q0 = fma(gs[u], bv[u], q0);
                q1 = fma(gs[u + 1], bv[u + 1], q1);
                q2 = fma(gs[u + 2], bv[u + 2], q2);
                q3 = fma(gs[u + 3], bv[u + 3], q3);
            }
#pragma unroll
            for (int u = 0; u < 16; ++u) { gs[u] = gn[u]; bv[u] = bn2[u]; }
        }
#pragma unroll
        for (int u = 0; u < 16; u += 4) {
            q0 = fma(gs[u], bv[u], q0);
            q1 = fma(gs[u + 1], bv[u + 1], q1);
            q2 = fma(gs[u + 2], bv[u + 2], q2);
            q3 = fma(gs[u + 3], bv[u + 3], q3);
        }
        const double r0 = q0 + q2, r1 = q1 + q3;
        const double gkk = g[k];
        const double old = bcol[k * 64];
        const double res = (h - (r0 + r1) + gkk * old) + lam_eff * c;
        const double den = gkk + lam_deg;
        const double st = res > rho ? res - rho : (res < -rho ? res + rho : 0.0);
        const double qv = fmax(0.0, st / den);
        const double nw = (den > 1e-10) ? qv : 0.0;
        dmax = fmax(dmax, fabs(nw - old));
        amax = fmax(amax, fabs(old));
        bcol[k * 64] = nw;
        if (active) beta_out[k * ld + i] = nw;
    }
    dmax = wave_max(dmax);
    amax = wave_max(amax);
    if (lane == 0) {
        unsigned long long* s = stats + (size_t)it * 128;
        const int slot = slice & 63;
        atomicMax(s + slot, (unsigned long long)__double_as_longlong(dmax));
        atomicMax(s + 64 + slot, (unsigned long long)__double_as_longlong(amax));
    }
}

__global__ void pad_rows_kernel(const double* __restrict__ A, int K, int KP, double* __restrict__ B) {
    const int e = blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= K * KP) return;
    const int i = e / KP, j = e - i * KP;
    B[e] = j < K ? A[i * K + j] : 0.0;
}

bool sweep_uses_lds(int K) {
    return !sweep_instantiated(K) && (size_t)round_up(K, 16) * 512 <= 150 * 1024 && !getenv("FDX_SWEEP_GENERIC");
}
size_t sweep_lds_pad_doubles(int K) { return (size_t)K * round_up(K, 16); }
int sweep_lds_prepare(const double* XtX, int K, double* padded, hipStream_t st) {
    const int KP = (int)round_up(K, 16);
    hipLaunchKernelGGL(pad_rows_kernel, dim3(ceil_div((long long)K * KP, 256)), dim3(256), 0, st, XtX, K, KP, padded);
    FDX_CHECK_LAUNCH();
    return 0;
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
    } else if (sweep_uses_lds(a.K)) {
        if (!generic_scratch || scratch_ld != 0) return fail(FDX_ERR_INVALID, "LDS-resident BCD sweep needs the padded copy of XtX (sweep_lds_prepare)");
        const int KP = (int)round_up(a.K, 16);
        const size_t lds = (size_t)KP * 512;
        if (lds > 64 * 1024) FDX_HIP(hipFuncSetAttribute((const void*)bcd_sweep_lds_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        hipLaunchKernelGGL(bcd_sweep_lds_kernel, dim3(a.n_slices), dim3(64), lds, st, a.H, generic_scratch, a.beta_in, a.beta_out, a.ell, a.slice_off,
                           a.deg, a.stats, a.rel_change, a.lambda, a.rho, a.tol, a.ldh, a.ld, a.n, a.K, KP, a.it);
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

// sum_i beta_i' G beta_i = sum_kj G[k][j] S[k][j] with S = B B' (B = beta, K x n): the quadratic term of the objective as ONE Gram
// matrix of the abundances instead of K^2 products per spot inside the objective traversal (above 64 types).
// v_mfma_f64_16x16x4_f64 with the SPOTS as contraction index: for a 16-type tile t the lane value
// beta[16 t + (lane & 15)][s + (lane >> 4)] is at once the A operand (types on rows) and the B operand (types on columns) of every
// tile pair t takes part in, so a step of 4 spots costs one load per type tile and one MFMA per tile pair a <= b.  The loads are
// 32-byte pieces of 16 planes; consecutive steps take the other half of the same lines.  256 workgroups x 4 waves, slices of 64 spots
// dealt round-robin - a fixed assignment, so the partial sums are reproducible; each workgroup contracts its S with G and leaves ONE
// number in column 1 of its row of the objective's partials (the traversal left zeros there).
template <int TT>
__global__ __launch_bounds__(256) void beta_quad_kernel(const double* __restrict__ beta, long long ld, long long n, int K,
                                                        const double* __restrict__ G, double* __restrict__ partials) {
    __shared__ double s_part[4];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int tr = lane & 15, kq = lane >> 4;
    constexpr int NP = TT * (TT + 1) / 2;
    typedef double double4_t __attribute__((ext_vector_type(4)));
    double4_t acc[NP];
#pragma unroll
    for (int p = 0; p < NP; ++p) acc[p] = double4_t{0.0, 0.0, 0.0, 0.0};
    const long long n_slices = (n + 63) / 64;
    for (long long sl = (long long)blockIdx.x * 4 + wave; sl < n_slices; sl += (long long)gridDim.x * 4) {
        const long long s0 = sl * 64;
#pragma unroll 2
        for (int ks = 0; ks < 16; ++ks) {
            const long long sp = s0 + ks * 4 + kq;
            double v[TT];
#pragma unroll
            for (int t = 0; t < TT; ++t) {
                const int type = t * 16 + tr;
                v[t] = (type < K && sp < n) ? beta[(size_t)type * ld + sp] : 0.0;
            }
            int p = 0;
#pragma unroll
            for (int a = 0; a < TT; ++a)
#pragma unroll
                for (int b = a; b < TT; ++b, ++p) acc[p] = __builtin_amdgcn_mfma_f64_16x16x4f64(v[a], v[b], acc[p], 0, 0, 0);
        }
    }
    // C layout of the instruction: register rr of lane l is S[m = (l >> 4) + 4 rr][n = l & 15] of its tile pair
    double q = 0.0;
    int p = 0;
#pragma unroll
    for (int a = 0; a < TT; ++a)
#pragma unroll
        for (int b = a; b < TT; ++b, ++p) {
            double t = 0.0;
#pragma unroll
            for (int rr = 0; rr < 4; ++rr) {
                const int row = a * 16 + kq + 4 * rr, col = b * 16 + tr;
                if (row < K && col < K) t = fma(G[(size_t)row * K + col], acc[p][rr], t);
            }
            q += (a == b) ? t : 2.0 * t;                                 // S and G are symmetric: the pair (b, a) is the same sum
        }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) q += __shfl_xor(q, off, 64);
    if (lane == 0) s_part[wave] = q;
    __syncthreads();
    if (threadIdx.x == 0) partials[(size_t)blockIdx.x * 4 + 1] = (s_part[0] + s_part[1]) + (s_part[2] + s_part[3]);
}

int launch_beta_quad(const double* beta, long long ld, long long n, int K, const double* XtX, double* partials, int rows,
                     hipStream_t st) {
    FDX_REQUIRE(K >= 1 && K <= 112 && rows >= 1, "beta quad: 1 <= K <= 112");
    if (n <= 0) return 0;
    const int grid = (int)std::min<long long>(std::min(256, rows), (n + 255) / 256);
    const int TT = (K + 15) / 16;
#define FDX_BQ(T) case T: hipLaunchKernelGGL(beta_quad_kernel<T>, dim3(grid), dim3(256), 0, st, beta, ld, n, K, XtX, partials); break;
    switch (TT) { FDX_BQ(1) FDX_BQ(2) FDX_BQ(3) FDX_BQ(4) FDX_BQ(5) FDX_BQ(6) FDX_BQ(7) default: return fail(FDX_ERR_INVALID, "beta quad: K"); }
#undef FDX_BQ
    FDX_CHECK_LAUNCH();
    return 0;
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
