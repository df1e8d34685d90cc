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
#include "fdx_env.h"
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

// More than 112 cell types (and every K without a register-resident instantiation): abundances in LDS, coordinate loop ROLLED.
// The register-resident kernels need every loop over the types unrolled (static register indices): K^2 FMAs of straight-line code,
// which leaves the 64 KB instruction cache at 72 types, the register file at 112 (2 K registers for the abundances alone) and spills.
// Mapping: a wave = 16 SPOTS x 4 lane groups.  Lane (s = lane & 15, q = lane >> 4) owns spot s and the types j = q mod 4: per
// coordinate it sums K / 4 products G[k][j] b[j][s] - b from the wave's LDS slice (KP x 16 doubles, row stride 17: the four groups of a
// read hit disjoint banks), G[k][j] from the row of XtX the wave stages in LDS for this coordinate (the 16 lanes of a group read one
// address) - and the four partial sums meet through two cross-row shuffles.  A first form with one spot per lane (K x 64 doubles of LDS
// per wave) left room for two waves per CU and ran at a tenth of its LDS bound (4.3 ms per sweep at 100 types and 500k spots, behind
// the padded 112-type register kernel's 3.9); a quarter of the LDS per wave is four times the waves.  The neighbour sum is split the same
// way (neighbour m by group m mod 4).  Every lane of a spot evaluates the update itself (same operands, same bits); group 0 stores.
// Summation order differs from the register kernels' (four chains over the types mod 4, neighbours in four partial sums).
constexpr int LDS_SW_SPOTS = 16;
constexpr int LDS_SW_STRIDE = 17;
__global__ __launch_bounds__(256) void bcd_sweep_lds_kernel(
    const double* __restrict__ H, const double* __restrict__ Gp /* (K, KP) */, const double* __restrict__ beta_in,
    double* __restrict__ beta_out, const int* __restrict__ ell_base, const int* __restrict__ slice_off, const int* __restrict__ deg,
    unsigned long long* __restrict__ stats, double* __restrict__ rel_change, const double lambda, const double rho, const double tol,
    const int ldh_, const int ld_, const int n, const int K, const int KP, const int it) {
    extern __shared__ __attribute__((aligned(16))) double lds_sw[];        // per wave: b [KP][17], then the current row of G [KP]
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    if (it > 0) {
        const double rc = fold_rel_change(stats + (size_t)(it - 1) * 128, lane);
        if (blockIdx.x == 0 && threadIdx.x == 0) rel_change[it - 1] = rc;
        if (rc < tol) return;
    }
    const int s = lane & 15, q = lane >> 4;
    double* bl = lds_sw + (size_t)wave * ((size_t)KP * LDS_SW_STRIDE + KP);
    double* gl = bl + (size_t)KP * LDS_SW_STRIDE;
    const long long g16 = ((long long)xcd_remap(blockIdx.x, gridDim.x) * 4 + wave);   // group of 16 spots
    const long long first = g16 * LDS_SW_SPOTS;
    if (first >= n) return;                                                 // whole wave (no barrier in this kernel)
    const int i = (int)min(first + s, (long long)n - 1);                    // lanes past the last spot mirror spot n-1
    const bool active = first + s < n;
    const size_t ld = (size_t)ld_, ldh = (size_t)ldh_;
    const int slice = i >> 6;
    const int w0 = slice_off[slice];
    const int w = slice_off[slice + 1] - w0;                                // ELL width of the spot's slice (uniform: 16 | 64)
    const int* ell = ell_base + (size_t)w0 * 64 + (i & 63);
    const int dg = deg[i];
    const double lam_deg = lambda * (double)dg;
    const double lam_eff = (dg > 0) ? lambda : 0.0;
    int nbi[4];                                                             // neighbours m = q, q + 4, q + 8, q + 12 of the spot
#pragma unroll
    for (int t = 0; t < 4; ++t) nbi[t] = (4 * t + q < w) ? ell[(size_t)(4 * t + q) * 64] : -1;
    for (int k = q; k < KP; k += 4) bl[k * LDS_SW_STRIDE + s] = (k < K) ? beta_in[k * ld + i] : 0.0;
    __builtin_amdgcn_wave_barrier();
    double dmax = 0.0, amax = 0.0;
    for (int k = 0; k < K; ++k) {
        // this coordinate's row of G into LDS (padded to KP with zeros by sweep_lds_prepare), neighbour values and H requested
        const double* g = Gp + (size_t)k * KP;
        for (int j = lane; j < KP; j += 64) gl[j] = g[j];
        __builtin_amdgcn_wave_barrier();                                    // the row is read by other lanes of this wave (LDS operations of a wave are in order)
        double c = 0.0;
        const double* bk = beta_in + (size_t)k * ld;
#pragma unroll
        for (int t = 0; t < 4; ++t) if (nbi[t] >= 0) c += bk[nbi[t]];
        for (int m = 16 + q; m < w; m += 4) c += bk[ell[(size_t)m * 64]];
        const double h = H[(size_t)k * ldh + i];
        // KP is a multiple of 16: every lane group has KP / 4 terms, taken four at a time with all eight LDS reads issued before the
        // products (one read pair per product waited ~250 cycles each time)
        double r = 0.0, r2 = 0.0;
        const double* glq = gl + q;
        const double* blq = bl + q * LDS_SW_STRIDE + s;
        for (int j = 0; j < KP; j += 16) {
            double gv[4], bv[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) { gv[u] = glq[j + 4 * u]; bv[u] = blq[(j + 4 * u) * LDS_SW_STRIDE]; }
            r = fma(gv[0], bv[0], r);
            r2 = fma(gv[1], bv[1], r2);
            r = fma(gv[2], bv[2], r);
            r2 = fma(gv[3], bv[3], r2);
        }
        r += r2;
        r += __shfl_xor(r, 16, 64);
        r += __shfl_xor(r, 32, 64);
        c += __shfl_xor(c, 16, 64);
        c += __shfl_xor(c, 32, 64);
        const double gkk = gl[k];
        const double old = bl[k * LDS_SW_STRIDE + s];
        const double res = (h - r + gkk * old) + lam_eff * c;
        const double den = gkk + lam_deg;
        const double st = res > rho ? res - rho : (res < -rho ? res + rho : 0.0);
        const double qv = fmax(0.0, st / den);
        const double nw = (den > 1e-10) ? qv : 0.0;
        dmax = fmax(dmax, fabs(nw - old));
        amax = fmax(amax, fabs(old));
        if (q == 0) {
            bl[k * LDS_SW_STRIDE + s] = nw;
            if (active) beta_out[k * ld + i] = nw;
        }
        __builtin_amdgcn_wave_barrier();
    }
    dmax = wave_max(dmax);
    amax = wave_max(amax);
    if (lane == 0) {
        unsigned long long* sp = stats + (size_t)it * 128;
        const int slot = (int)(g16 & 63);
        atomicMax(sp + slot, (unsigned long long)__double_as_longlong(dmax));
        atomicMax(sp + 64 + slot, (unsigned long long)__double_as_longlong(amax));
    }
}

__global__ void pad_rows_kernel(const double* __restrict__ A, int K, int KP, double* __restrict__ B) {
    const int e = blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= K * KP) return;
    const int i = e / KP, j = e - i * KP;
    B[e] = j < K ? A[i * K + j] : 0.0;
}

static size_t sweep_lds_bytes(int K) {       // per workgroup of four waves
    const size_t KP = (size_t)round_up(K, 16);
    return 4 * (KP * LDS_SW_STRIDE + KP) * sizeof(double);
}
bool sweep_uses_lds(int K) {
    return !sweep_instantiated(K) && sweep_lds_bytes(K) <= 160 * 1024 && !fdx::exp_env("FDX_SWEEP_GENERIC");
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
bool bcd_sweep_dispatch_part8(const BcdSweepArgs&, hipStream_t);
bool bcd_sweep_dispatch_part9(const BcdSweepArgs&, hipStream_t);

int launch_bcd_sweep(const BcdSweepArgs& a, double* generic_scratch, size_t scratch_ld, hipStream_t st) {
    if (a.n <= 0 || a.n_slices <= 0) return 0;
    if (sweep_instantiated(a.K)) {
        const bool hit = bcd_sweep_dispatch_part0(a, st) || bcd_sweep_dispatch_part1(a, st) ||
                         bcd_sweep_dispatch_part2(a, st) || bcd_sweep_dispatch_part3(a, st) ||
                         bcd_sweep_dispatch_part4(a, st) || bcd_sweep_dispatch_part5(a, st) ||
                         bcd_sweep_dispatch_part6(a, st) || bcd_sweep_dispatch_part7(a, st) ||
                         bcd_sweep_dispatch_part8(a, st) || bcd_sweep_dispatch_part9(a, st);
        if (!hit) return fail(FDX_ERR_INTERNAL, "bcd sweep dispatch failed");
    } else if (sweep_uses_lds(a.K)) {
        if (!generic_scratch || scratch_ld != 0) return fail(FDX_ERR_INVALID, "LDS-resident BCD sweep needs the padded copy of XtX (sweep_lds_prepare)");
        const int KP = (int)round_up(a.K, 16);
        const size_t lds = sweep_lds_bytes(a.K);
        if (lds > 64 * 1024) FDX_HIP(hipFuncSetAttribute((const void*)bcd_sweep_lds_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        const int grid = ceil_div(a.n, 4 * LDS_SW_SPOTS);
        hipLaunchKernelGGL(bcd_sweep_lds_kernel, dim3(grid), dim3(256), lds, st, a.H, generic_scratch, a.beta_in, a.beta_out, a.ell, a.slice_off,
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

// The same for more than 112 types, block by block: the tile pairs (a0 .. a0 + 3) x (b0 .. b0 + 3) of S in one launch (16 accumulator
// tiles), their contraction with G ADDED to column 1 (weight 2 for a block above the diagonal, which stands for its mirror image too;
// a diagonal block is computed as a full square).  Launches follow each other on one stream.
__global__ __launch_bounds__(256) void beta_quad_block_kernel(const double* __restrict__ beta, long long ld, long long n, int K,
                                                              const double* __restrict__ G, int a0, int b0, double weight,
                                                              double* __restrict__ partials) {
    __shared__ double s_part[4];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int tr = lane & 15, kq = lane >> 4;
    typedef double double4_t __attribute__((ext_vector_type(4)));
    double4_t acc[16];
#pragma unroll
    for (int p = 0; p < 16; ++p) acc[p] = double4_t{0.0, 0.0, 0.0, 0.0};
    const long long n_slices = (n + 63) / 64;
    for (long long sl = (long long)blockIdx.x * 4 + wave; sl < n_slices; sl += (long long)gridDim.x * 4) {
        const long long s0 = sl * 64;
#pragma unroll 2
        for (int ks = 0; ks < 16; ++ks) {
            const long long sp = s0 + ks * 4 + kq;
            double va[4], vb[4];
#pragma unroll
            for (int t = 0; t < 4; ++t) {
                const int ta = (a0 + t) * 16 + tr, tb = (b0 + t) * 16 + tr;
                va[t] = (ta < K && sp < n) ? beta[(size_t)ta * ld + sp] : 0.0;
                vb[t] = (tb < K && sp < n) ? beta[(size_t)tb * ld + sp] : 0.0;
            }
#pragma unroll
            for (int a = 0; a < 4; ++a)
#pragma unroll
                for (int b = 0; b < 4; ++b) acc[a * 4 + b] = __builtin_amdgcn_mfma_f64_16x16x4f64(va[a], vb[b], acc[a * 4 + b], 0, 0, 0);
        }
    }
    double q = 0.0;
#pragma unroll
    for (int a = 0; a < 4; ++a)
#pragma unroll
        for (int b = 0; b < 4; ++b) {
#pragma unroll
            for (int rr = 0; rr < 4; ++rr) {
                const int row = (a0 + a) * 16 + kq + 4 * rr, col = (b0 + b) * 16 + tr;
                if (row < K && col < K) q = fma(G[(size_t)row * K + col], acc[a * 4 + b][rr], q);
            }
        }
    q *= weight;
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) q += __shfl_xor(q, off, 64);
    if (lane == 0) s_part[wave] = q;
    __syncthreads();
    if (threadIdx.x == 0) partials[(size_t)blockIdx.x * 4 + 1] += (s_part[0] + s_part[1]) + (s_part[2] + s_part[3]);
}

int launch_beta_quad(const double* beta, long long ld, long long n, int K, const double* XtX, double* partials, int rows,
                     hipStream_t st) {
    FDX_REQUIRE(K >= 1 && rows >= 1, "beta quad: bad arguments");
    if (n <= 0) return 0;
    const int grid = (int)std::min<long long>(std::min(256, rows), (n + 255) / 256);
    const int TT = (K + 15) / 16;
    if (TT > 7) {                                   // more than 112 types: blocks of 4 x 4 type tiles, upper triangle of blocks
        const int nblk = (TT + 3) / 4;
        for (int A = 0; A < nblk; ++A)
            for (int B = A; B < nblk; ++B)
                hipLaunchKernelGGL(beta_quad_block_kernel, dim3(grid), dim3(256), 0, st, beta, ld, n, K, XtX, A * 4, B * 4, A == B ? 1.0 : 2.0,
                                   partials);
        FDX_CHECK_LAUNCH();
        return 0;
    }
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
