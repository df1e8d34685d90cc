// CSR (sparse) spot matrix on the device: preprocess + CountSketch projection and gene statistics without densifying.
//
// Replaces, for scipy.sparse / CSR input,
//   flashdeconv/core/deconv.py:181-188      log-CPM on the stored values (library size 0 -> 1; zeros stay zeros)
//   flashdeconv/core/sketching.py:194-199   project_to_sketch(Y sparse) = Y @ Omega
//   flashdeconv/utils/genes.py:52-83        select_hvg sparse branch: per-gene mean / E[z^2]-mean^2 variance
//   flashdeconv/core/deconv.py:207-212      per-gene means of Y for "pearson"
//
// HBM traffic is proportional to the stored entries (8 bytes each for f32 data + int32 column), not to N x G, and the
// log1p work shrinks with it - on ~5-10 % dense count matrices that is the 10x the dense kernel cannot reach.
//
// Sketch mapping: one wavefront = one spot row.  The row's entries are read 64 at a time (coalesced), every lane looks
// its column up in a 16-byte per-gene table {weight, bucket} that covers ALL G_all columns (bucket -1 = gene not
// selected; the table is L2 resident: 16 B x 30k genes = 480 KB), applies the transform and adds weight * f(y) into
// the wave's private d-entry accumulator in LDS with ds_add_f64.  Several entries of one row can hit the same bucket,
// so the order of those additions is the hardware's; the reference's scipy product carries the same freedom, and
// the 1e-4 parity budget is 12 orders of magnitude above it.  The finished row is written as one coalesced d*8-byte row
// of Y_sketch exactly like the dense kernel, so the H contraction downstream is shared.
#include <algorithm>

#include "device_math.h"
#include "fdx_internal.h"
#include "fdx_kernels.h"

namespace fdx {

struct __attribute__((aligned(16))) GeneSlot {
    double w;
    int bucket;
    int pad;
};

__device__ __forceinline__ void lds_add(double* p, double v) {
    __hip_atomic_fetch_add(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
}

// MODE: FDX_PRE_RAW or FDX_PRE_LOG_CPM_SPARSE (the sparse rule is the only log-CPM rule for CSR input)
template <typename T, int MODE>
__global__ __launch_bounds__(256) void sketch_csr_kernel(const long long* __restrict__ indptr, const int* __restrict__ indices,
                                                         const T* __restrict__ data, const int* __restrict__ row_map,
                                                         long long row0, long long n, int d,
                                                         const GeneSlot* __restrict__ table, double* __restrict__ Ys,
                                                         long long ldys, double* __restrict__ row_sumsq) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int lane = threadIdx.x & 63;
    const int wib = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int waves_per_blk = blockDim.x >> 6;
    double* acc = reinterpret_cast<double*>(smem) + (size_t)wib * d;
    const long long wave0 = (long long)blockIdx.x * waves_per_blk + wib;
    const long long stride = (long long)gridDim.x * waves_per_blk;
    for (long long p = wave0; p < n; p += stride) {
        const long long row = row_map ? (long long)row_map[p] : row0 + p;
        const long long beg = indptr[row], end = indptr[row + 1];
        for (int c = lane; c < d; c += 64) acc[c] = 0.0;
        double scale = 1.0;
        if (MODE != FDX_PRE_RAW) {      // library size over the SELECTED genes (the subset is taken first, deconv.py:321)
            double s = 0.0;
            for (long long q = beg + lane; q < end; q += 64)
                if (table[indices[q]].bucket >= 0) s += (double)data[q];
            s = wave_sum(s);
            scale = 10000.0 / (s == 0.0 ? 1.0 : s);                    // deconv.py:183-185
        }
        __builtin_amdgcn_s_waitcnt(0xc07f);                            // zeroing done before the adds
        for (long long q = beg + lane; q < end; q += 64) {
            const GeneSlot e = table[indices[q]];
            if (e.bucket >= 0) {
                double y = (double)data[q];
                if (MODE != FDX_PRE_RAW) y = fast_log1p(y * scale);
                lds_add(acc + e.bucket, e.w * y);
            }
        }
        __builtin_amdgcn_s_waitcnt(0xc07f);
        double* dst = Ys + (size_t)p * ldys;
        double sq = 0.0;
        for (int c = lane; c < d; c += 64) {
            const double v = acc[c];
            dst[c] = v;
            sq = fma(v, v, sq);
        }
        if (row_sumsq) {
            sq = wave_sum(sq);
            if (lane == 0) row_sumsq[p] = sq;
        }
        __builtin_amdgcn_s_waitcnt(0xc07f);                            // reads of acc done before the next row zeroes it
    }
}

template <typename T>
static int launch_sketch_csr_t(const long long* indptr, const int* indices, const T* data, const int* row_map, long long row0,
                               long long n, int d, int mode, const void* table, double* Ys, long long ldys,
                               double* row_sumsq, hipStream_t st) {
    int waves = 4;
    while (waves > 1 && (size_t)d * 8 * waves > 64 * 1024) waves >>= 1;
    const size_t lds = (size_t)d * 8 * waves;
    if (lds > 160 * 1024) return fail(FDX_ERR_UNSUPPORTED, "sketch (CSR): sketch_dim does not fit in LDS");
    const int blocks = (int)std::min<long long>((n + waves - 1) / waves, 256LL * 16);
    auto launch = [&](auto kern) -> int {
        if (lds > 64 * 1024)
            FDX_HIP(hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        hipLaunchKernelGGL(kern, dim3(blocks), dim3(waves * 64), lds, st, indptr, indices, data, row_map, row0, n, d,
                           (const GeneSlot*)table, Ys, ldys, row_sumsq);
        FDX_CHECK_LAUNCH();
        return 0;
    };
    if (mode == FDX_PRE_RAW) return launch(sketch_csr_kernel<T, FDX_PRE_RAW>);
    if (mode == FDX_PRE_LOG_CPM_SPARSE || mode == FDX_PRE_LOG_CPM) return launch(sketch_csr_kernel<T, FDX_PRE_LOG_CPM_SPARSE>);
    return fail(FDX_ERR_INVALID, "sketch (CSR): unknown preprocess mode");
}

int launch_sketch_csr(const long long* indptr, const int* indices, const void* data, int dtype, const int* row_map,
                      long long row0, long long n, int d, int mode, const void* table, double* Ys, long long ldys,
                      double* row_sumsq, hipStream_t st) {
    if (n <= 0 || d <= 0) return 0;
    if (dtype == FDX_F32)
        return launch_sketch_csr_t<float>(indptr, indices, (const float*)data, row_map, row0, n, d, mode, table, Ys, ldys, row_sumsq, st);
    if (dtype == FDX_F64)
        return launch_sketch_csr_t<double>(indptr, indices, (const double*)data, row_map, row0, n, d, mode, table, Ys, ldys, row_sumsq, st);
    return fail(FDX_ERR_INVALID, "sketch (CSR): dtype must be FDX_F32 or FDX_F64");
}

size_t csr_gene_slot_bytes() { return sizeof(GeneSlot); }

// ------------------------------------------------------------------------------------------------ gene statistics
// One wave per row: library size over all genes, then z = log1p(y * 1e4 / max(lib, 1)) for every stored entry and three
// per-gene sums (z, z^2, y) with f64 atomics.  The sums live in `copies` replicas (block b adds into replica b % copies)
// so the genes almost every spot expresses do not serialise the L2 atomic units on one address; the replicas are folded
// in index order.  (Zeros contribute nothing to any of the three sums: genes.py:52-54.)
template <typename T>
__global__ __launch_bounds__(256) void csr_moments_kernel(const long long* __restrict__ indptr, const int* __restrict__ indices,
                                                          const T* __restrict__ data, long long n, int G, int copies,
                                                          double* __restrict__ sums /* (copies, 3, G) */) {
    const int lane = threadIdx.x & 63;
    const int wib = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int waves_per_blk = blockDim.x >> 6;
    double* mine = sums + (size_t)(blockIdx.x % copies) * 3 * G;
    const long long wave0 = (long long)blockIdx.x * waves_per_blk + wib;
    const long long stride = (long long)gridDim.x * waves_per_blk;
    for (long long row = wave0; row < n; row += stride) {
        const long long beg = indptr[row], end = indptr[row + 1];
        double s = 0.0;
        for (long long q = beg + lane; q < end; q += 64) s += (double)data[q];
        s = wave_sum(s);
        const double scale = 10000.0 / fmax(s, 1.0);                   // genes.py:57-59
        for (long long q = beg + lane; q < end; q += 64) {
            const int g = indices[q];
            const double y = (double)data[q];
            const double z = fast_log1p(y * scale);
            atomicAdd(mine + g, z);
            atomicAdd(mine + (size_t)G + g, z * z);
            atomicAdd(mine + 2 * (size_t)G + g, y);
        }
    }
}

__global__ __launch_bounds__(256) void csr_fold_moments_kernel(const double* __restrict__ sums, int copies, int G, long long n,
                                                               double* __restrict__ mean, double* __restrict__ var,
                                                               double* __restrict__ colsum) {
    const int g = blockIdx.x * 256 + threadIdx.x;
    if (g >= G) return;
    double s1 = 0.0, s2 = 0.0, s0 = 0.0;
    for (int c = 0; c < copies; ++c) {
        const double* p = sums + (size_t)c * 3 * G;
        s1 += p[g];
        s2 += p[(size_t)G + g];
        s0 += p[2 * (size_t)G + g];
    }
    const double m = s1 / (double)n;
    mean[g] = m;
    var[g] = (n >= 2) ? fmax(((s2 / (double)n) - m * m) * ((double)n / (double)(n - 1)), 0.0) : 0.0;   // genes.py:74-83
    colsum[g] = s0;
}

int csr_moment_copies() { return 32; }

int launch_csr_moments(const long long* indptr, const int* indices, const void* data, int dtype, long long n, int G,
                       double* sums, double* mean, double* var, double* colsum, hipStream_t st) {
    if (G <= 0 || n <= 0) return fail(FDX_ERR_INVALID, "gene moments (CSR): empty matrix");
    const int copies = csr_moment_copies();
    FDX_HIP(hipMemsetAsync(sums, 0, (size_t)copies * 3 * G * sizeof(double), st));
    const int blocks = (int)std::min<long long>((n + 3) / 4, 256LL * 8);
    if (dtype == FDX_F32)
        hipLaunchKernelGGL(csr_moments_kernel<float>, dim3(blocks), dim3(256), 0, st, indptr, indices, (const float*)data, n, G, copies, sums);
    else if (dtype == FDX_F64)
        hipLaunchKernelGGL(csr_moments_kernel<double>, dim3(blocks), dim3(256), 0, st, indptr, indices, (const double*)data, n, G, copies, sums);
    else
        return fail(FDX_ERR_INVALID, "gene moments (CSR): dtype must be FDX_F32 or FDX_F64");
    FDX_CHECK_LAUNCH();
    hipLaunchKernelGGL(csr_fold_moments_kernel, dim3(ceil_div(G, 256)), dim3(256), 0, st, sums, copies, G, n, mean, var, colsum);
    FDX_CHECK_LAUNCH();
    return 0;
}

// Structure check of an uploaded CSR matrix before any kernel indexes with it (an out-of-range column would fault the
// device): indptr non-decreasing from 0 to nnz, 0 <= column < G.  flag[0] != 0 on violation.
__global__ __launch_bounds__(256) void csr_check_kernel(const long long* __restrict__ indptr, const int* __restrict__ indices,
                                                        long long n, long long nnz, int G, int* __restrict__ flag) {
    const long long t = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    const long long stride = (long long)gridDim.x * blockDim.x;
    int bad = 0;
    if (t == 0 && (indptr[0] != 0 || indptr[n] != nnz)) bad = 1;
    for (long long r = t; r < n; r += stride)
        if (indptr[r + 1] < indptr[r]) bad = 1;
    for (long long q = t; q < nnz; q += stride) {
        const int c = indices[q];
        if (c < 0 || c >= G) bad = 1;
    }
    if (bad) atomicOr(flag, 1);
}

int launch_csr_check(const long long* indptr, const int* indices, long long n, long long nnz, int G, int* flag, hipStream_t st) {
    FDX_HIP(hipMemsetAsync(flag, 0, sizeof(int), st));
    const int blocks = (int)std::min<long long>(std::max<long long>(1, (std::max(n, nnz) + 255) / 256), 256LL * 8);
    hipLaunchKernelGGL(csr_check_kernel, dim3(blocks), dim3(256), 0, st, indptr, indices, n, nnz, G, flag);
    FDX_CHECK_LAUNCH();
    return 0;
}

}  // namespace fdx
