// End-of-solve kernels.
//   objective_partials  <- flashdeconv/core/solver.py:226-284 (compute_objective) with L = D - A
//                          (flashdeconv/core/spatial.py:70-73); includes the Laplacian SpMV.
//   normalize_export    <- flashdeconv/core/solver.py:431-452 (normalize_proportions) fused with the layout change
//                          from the solver's type-major / Morton-ordered beta to the reference's (n_spots, n_types)
//                          row-major arrays in the caller's spot order.
// Both map one lane to one spot (coalesced reads of the type planes) and stage the wave's 64 x K tile in LDS.
#include "fdx_internal.h"
#include "fdx_kernels.h"

namespace fdx {

// obj = 0.5*(YtY - 2*cross + quad) + 0.5*lambda*spat + rho*l1 ; this kernel emits per-block (cross, quad, spat, l1).
__global__ __launch_bounds__(256) void objective_partials_kernel(
    const double* __restrict__ beta, long long ld, const double* __restrict__ H, long long ldh,
    const double* __restrict__ XtX, const int* __restrict__ ell_base, const int* __restrict__ slice_off,
    const int* __restrict__ deg, int n, int n_slices, int K, int use_lds, int skip_quad, double* __restrict__ partials) {
    extern __shared__ __attribute__((aligned(16))) double smem[];
    __shared__ double red[4][4];
    const int lane = threadIdx.x & 63;
    const int wib = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int slice = blockIdx.x * 4 + wib;
    double cross = 0.0, quad = 0.0, spat = 0.0, l1 = 0.0;
    if (slice < n_slices) {
        const int i = slice * 64 + lane;
        const bool active = i < n;
        const int ii = active ? i : n - 1;
        double* tile = smem + (size_t)wib * K * 64;  // [k][lane]
        if (use_lds)
            for (int k = 0; k < K; ++k) tile[k * 64 + lane] = beta[(size_t)k * ld + ii];
        const int w0 = slice_off[slice];
        const int w = slice_off[slice + 1] - w0;
        const int* ell = ell_base + (size_t)w0 * 64 + lane;
        const double dg = (double)deg[ii];
        for (int k = 0; k < K; ++k) {
            const double bk = use_lds ? tile[k * 64 + lane] : beta[(size_t)k * ld + ii];
            double nb = 0.0;
            for (int m = 0; m < w; ++m) nb += beta[(size_t)k * ld + ell[(size_t)m * 64]];
            double gb = 0.0;
            const double* g = XtX + (size_t)k * K;
            if (!skip_quad)                                   // otherwise launch_beta_quad supplies the quadratic term
                for (int l = 0; l < K; ++l) gb = fma(g[l], use_lds ? tile[l * 64 + lane] : beta[(size_t)l * ld + ii], gb);
            cross = fma(bk, H[(size_t)k * ldh + ii], cross);
            quad = fma(bk, gb, quad);
            spat = fma(bk, dg * bk - nb, spat);
            l1 += fabs(bk);
        }
        if (!active) { cross = quad = spat = l1 = 0.0; }
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
        cross += __shfl_xor(cross, off, 64);
        quad += __shfl_xor(quad, off, 64);
        spat += __shfl_xor(spat, off, 64);
        l1 += __shfl_xor(l1, off, 64);
    }
    if (lane == 0) { red[wib][0] = cross; red[wib][1] = quad; red[wib][2] = spat; red[wib][3] = l1; }
    __syncthreads();
    if (threadIdx.x < 4) {
        const int o = threadIdx.x;
        partials[(size_t)blockIdx.x * 4 + o] = ((red[0][o] + red[1][o]) + red[2][o]) + red[3][o];
    }
}

// beta (K, ld) type-major in solver order -> beta_out / prop_out (n, K) row-major at row perm[i] (perm may be null).
__global__ __launch_bounds__(256) void normalize_export_kernel(const double* __restrict__ beta, long long ld,
                                                               const int* __restrict__ perm, int n, int n_slices, int K,
                                                               int use_lds, double* __restrict__ beta_out,
                                                               double* __restrict__ prop_out) {
    extern __shared__ __attribute__((aligned(16))) double smem[];
    __shared__ double inv_s[4][64];
    __shared__ int row_s[4][64];
    const int lane = threadIdx.x & 63;
    const int wib = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int slice = blockIdx.x * 4 + wib;
    if (slice >= n_slices) return;
    const int i = slice * 64 + lane;
    const bool active = i < n;
    const int ii = active ? i : n - 1;
    const int Ks = K | 1;                        // odd row stride: the 64 lanes of a column write hit 64 different banks
    double* tile = smem + (size_t)wib * Ks * 64; // [spot][k], row stride Ks
    double s = 0.0;
    for (int k = 0; k < K; ++k) {
        const double v = beta[(size_t)k * ld + ii];
        s += v;
        if (use_lds) tile[lane * Ks + k] = v;
    }
    // normalize_proportions: all-zero rows -> 1/K, otherwise beta / max(rowsum, 1e-10)   (solver.py:445-451)
    const double den = fmax(s, 1e-10);
    const int orow = perm ? perm[ii] : ii;
    if (!use_lds) {
        if (active)
            for (int k = 0; k < K; ++k) {
                const double v = beta[(size_t)k * ld + ii];
                if (beta_out) beta_out[(size_t)orow * K + k] = v;
                if (prop_out) prop_out[(size_t)orow * K + k] = (s == 0.0) ? 1.0 / (double)K : v / den;
            }
        return;
    }
    inv_s[wib][lane] = (s == 0.0) ? -1.0 : den;
    row_s[wib][lane] = active ? orow : -1;
    // same-wave LDS hand-off: the wave's own ds_writes are ordered before its later ds_reads
    __builtin_amdgcn_s_waitcnt(0xc07f);  // lgkmcnt(0)
    // element f = lane + 64 j of the wave's 64 x K block: (spot, type) advance by (64 / K, 64 % K) with a carry - one integer
    // division per lane instead of one per element; the results are written once and never read here: non-temporal stores
    const int total = 64 * K;
    const int ds = 64 / K, dk = 64 - ds * K;
    int sp = lane / K, k = lane - sp * K;
    const double inv_K = 1.0 / (double)K;
    for (int f = lane; f < total; f += 64) {
        const int row = row_s[wib][sp];
        if (row >= 0) {
            const double v = tile[sp * Ks + k];
            const double dd = inv_s[wib][sp];
            if (beta_out) __builtin_nontemporal_store(v, beta_out + (size_t)row * K + k);
            if (prop_out) __builtin_nontemporal_store((dd < 0.0) ? inv_K : v / dd, prop_out + (size_t)row * K + k);
        }
        sp += ds;
        k += dk;
        if (k >= K) { k -= K; ++sp; }
    }
}

static inline size_t tile_lds_bytes(int K) { return (size_t)(K | 1) * 256 * sizeof(double); }
static inline bool tile_fits(int K) { return tile_lds_bytes(K) <= 128 * 1024; }

int objective_partials_count(int n_slices) { return ceil_div(n_slices, 4); }

int launch_objective_partials(const double* beta, long long ld, const double* H, long long ldh, const double* XtX,
                              const int* ell, const int* slice_off, const int* deg, int n, int n_slices, int K,
                              double* partials, hipStream_t st, int skip_quad) {
    if (n <= 0) return 0;
    const int use_lds = tile_fits(K) ? 1 : 0;
    const size_t lds = use_lds ? tile_lds_bytes(K) : 0;
    if (lds > 64 * 1024)
        FDX_HIP(hipFuncSetAttribute((const void*)objective_partials_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    hipLaunchKernelGGL(objective_partials_kernel, dim3(ceil_div(n_slices, 4)), dim3(256), lds, st, beta, ld, H, ldh, XtX,
                       ell, slice_off, deg, n, n_slices, K, use_lds, skip_quad, partials);
    FDX_CHECK_LAUNCH();
    return 0;
}

int launch_normalize_export(const double* beta, long long ld, const int* perm, int n, int n_slices, int K,
                            double* beta_out, double* prop_out, hipStream_t st) {
    if (n <= 0) return 0;
    const int use_lds = tile_fits(K) ? 1 : 0;
    const size_t lds = use_lds ? tile_lds_bytes(K) : 0;
    if (lds > 64 * 1024)
        FDX_HIP(hipFuncSetAttribute((const void*)normalize_export_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    hipLaunchKernelGGL(normalize_export_kernel, dim3(ceil_div(n_slices, 4)), dim3(256), lds, st, beta, ld, perm, n,
                       n_slices, K, use_lds, beta_out, prop_out);
    FDX_CHECK_LAUNCH();
    return 0;
}

}  // namespace fdx
