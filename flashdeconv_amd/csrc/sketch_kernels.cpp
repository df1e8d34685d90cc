// Preprocess + CountSketch projection of spot rows:  Y_sketch = f(Y) @ Omega   (N x G -> N x d)
//
// Replaces
//   flashdeconv/core/deconv.py:177-197,227-229   _preprocess_data ("log_cpm" dense / sparse rule, "raw")
//   flashdeconv/core/deconv.py:199-225           "pearson": a per-gene 1/sigma folded into Omega's weights, sigma from
//                                                the column means computed by column_sums_kernel below
//   flashdeconv/core/sketching.py:160-206        project_to_sketch (dense @ CSR with one entry per gene)
//
// Omega has exactly one non-zero per gene (bucket[g], weight[g]); output bucket c is the gene-ordered sum of
// weight[g]*f(y_g) over the ~G/d genes hashed to c.  The hash is the same for every spot, so the gather pattern is a
// STATIC SCHEDULE built once on the host (sketch_plan.cpp): buckets are sorted by list length and dealt to
// (group j, lane l) slots, so the 64 lanes of a wave walk lists of (almost) equal length and the trip count of every
// group is wave-uniform (short lists are padded with weight-0 entries).  Sums run in ascending gene order, without
// atomics: results are bit-reproducible and follow the reference's summation order.
//
// Mapping: one wavefront = one spot row at a time.  The row is streamed from HBM with 16-byte coalesced loads into
// the wave's private LDS slice (the only HBM traffic: G*sizeof(T) per spot), the library size for log-CPM is reduced
// on the way in, then lanes gather their genes from LDS.  No workgroup barrier is needed: a wave only reads LDS it
// wrote itself.  Results are un-permuted through LDS and written as one coalesced d*8-byte row.
#include "fdx_internal.h"
#include "fdx_kernels.h"

namespace fdx {

template <typename T> struct Vec4;
template <> struct Vec4<float> { typedef float type __attribute__((ext_vector_type(4))); };
template <> struct Vec4<double> { typedef double type __attribute__((ext_vector_type(2))); };  // 16 bytes

__device__ __forceinline__ double wave_sum(double v) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
    return v;
}

// MODE: FDX_PRE_RAW (0), FDX_PRE_LOG_CPM (1, dense rule), FDX_PRE_LOG_CPM_SPARSE (2, zero library size -> 1)
template <typename T, int MODE, bool VEC>
__global__ __launch_bounds__(256) void sketch_rows_kernel(const T* __restrict__ Y, long long ldy,
                                                          const int* __restrict__ row_map, long long n, int G, int d,
                                                          const int* __restrict__ sched_gene,
                                                          const double* __restrict__ sched_w,
                                                          const int* __restrict__ group_off, int n_groups,
                                                          const int* __restrict__ slot_bucket,
                                                          double* __restrict__ Ys, long long ldys,
                                                          double* __restrict__ row_sumsq) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int lane = threadIdx.x & 63;
    const int wib = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int waves_per_blk = blockDim.x >> 6;
    const size_t row_bytes = ((size_t)G * sizeof(T) + 15) & ~(size_t)15;
    const size_t per_wave = row_bytes + (size_t)d * sizeof(double);
    T* rowbuf = reinterpret_cast<T*>(smem + (size_t)wib * per_wave);
    double* outbuf = reinterpret_cast<double*>(smem + (size_t)wib * per_wave + row_bytes);

    const long long wave0 = (long long)blockIdx.x * waves_per_blk + wib;
    const long long wave_stride = (long long)gridDim.x * waves_per_blk;
    for (long long p = wave0; p < n; p += wave_stride) {
        const long long src_row = row_map ? (long long)row_map[p] : p;
        const T* yrow = Y + (size_t)src_row * ldy;
        // ---- stream the row into LDS, reducing the library size on the way
        double part = 0.0;
        if (VEC) {
            typedef typename Vec4<T>::type V;
            constexpr int PER = 16 / sizeof(T);
            const int nvec = G / PER;
            const V* src = reinterpret_cast<const V*>(yrow);
            V* dst = reinterpret_cast<V*>(rowbuf);
            for (int v = lane; v < nvec; v += 64) {
                const V x = src[v];
                dst[v] = x;
                if (MODE != FDX_PRE_RAW) {
#pragma unroll
                    for (int e = 0; e < PER; ++e) part += (double)x[e];
                }
            }
            for (int g = nvec * PER + lane; g < G; g += 64) {
                const T x = yrow[g];
                rowbuf[g] = x;
                if (MODE != FDX_PRE_RAW) part += (double)x;
            }
        } else {
            for (int g = lane; g < G; g += 64) {
                const T x = yrow[g];
                rowbuf[g] = x;
                if (MODE != FDX_PRE_RAW) part += (double)x;
            }
        }
        double scale = 1.0;
        if (MODE == FDX_PRE_LOG_CPM) {
            const double s = wave_sum(part);
            scale = (1.0 / (s + 1e-10)) * 1e4;             // y / (rowsum + 1e-10) * 1e4   (deconv.py:190)
        } else if (MODE == FDX_PRE_LOG_CPM_SPARSE) {
            double s = wave_sum(part);
            if (s == 0.0) s = 1.0;                         // lib_size[lib_size == 0] = 1  (deconv.py:183-185)
            scale = 1e4 / s;
        }
        __builtin_amdgcn_s_waitcnt(0xc07f);  // lgkmcnt(0): this wave's LDS writes have landed
        // ---- static schedule: group j, lane l owns bucket slot j*64+l
        double sq = 0.0;
        for (int j = 0; j < n_groups; ++j) {
            const int e0 = group_off[j], e1 = group_off[j + 1];   // wave-uniform trip count
            double acc = 0.0;
            for (int e = e0; e < e1; ++e) {
                const int g = sched_gene[(size_t)e * 64 + lane];
                const double w = sched_w[(size_t)e * 64 + lane];
                double y = (double)rowbuf[g];
                if (MODE != FDX_PRE_RAW) y = log1p(y * scale);
                acc = fma(w, y, acc);
            }
            const int slot = j * 64 + lane;
            const int bucket = slot_bucket[slot];            // -1 for the padding slots of the last group
            if (bucket >= 0) {
                outbuf[bucket] = acc;
                sq = fma(acc, acc, sq);
            }
        }
        __builtin_amdgcn_s_waitcnt(0xc07f);
        double* dst = Ys + (size_t)p * ldys;
        for (int c = lane; c < d; c += 64) dst[c] = outbuf[c];
        if (row_sumsq) {
            sq = wave_sum(sq);
            if (lane == 0) row_sumsq[p] = sq;
        }
        __builtin_amdgcn_s_waitcnt(0xc07f);  // reads of outbuf done before the next row overwrites it
    }
}

// Per-gene column sums of Y over all spots (pearson: mean_g = sum_g / N).  Block b sums a contiguous stripe of rows
// for every gene (lane = gene: coalesced); partials (n_blocks, G) are then folded in block order -> deterministic.
template <typename T>
__global__ __launch_bounds__(256) void column_sums_kernel(const T* __restrict__ Y, long long ldy, long long n, int G,
                                                          int rows_per_block, double* __restrict__ partials) {
    const long long r0 = (long long)blockIdx.y * rows_per_block;
    const long long r1 = min(n, r0 + rows_per_block);
    const int g = blockIdx.x * 256 + threadIdx.x;
    if (g >= G) return;
    double acc = 0.0;
    for (long long r = r0; r < r1; ++r) acc += (double)Y[(size_t)r * ldy + g];
    partials[(size_t)blockIdx.y * G + g] = acc;
}

__global__ __launch_bounds__(256) void fold_column_partials_kernel(const double* __restrict__ partials, int n_parts, int G,
                                                                   double* __restrict__ out) {
    const int g = blockIdx.x * 256 + threadIdx.x;
    if (g >= G) return;
    double acc = 0.0;
    for (int b = 0; b < n_parts; ++b) acc += partials[(size_t)b * G + g];
    out[g] = acc;
}

template <typename T, int MODE>
static int launch_sketch_mode(const T* Y, long long ldy, const int* row_map, long long n, int G, int d,
                              const SketchPlanDev& plan, double* Ys, long long ldys, double* row_sumsq, hipStream_t st) {
    const size_t row_bytes = ((size_t)G * sizeof(T) + 15) & ~(size_t)15;
    const size_t per_wave = row_bytes + (size_t)d * sizeof(double);
    int waves = 4;
    while (waves > 1 && per_wave * waves > 64 * 1024) waves >>= 1;   // keep >= 2 blocks per CU where possible
    const size_t lds = per_wave * waves;
    if (lds > 160 * 1024) return fail(FDX_ERR_UNSUPPORTED, "sketch: one gene row does not fit in LDS (G too large)");
    const bool vec = (ldy % (16 / (long long)sizeof(T)) == 0) && ((reinterpret_cast<uintptr_t>(Y) & 15) == 0);
    const long long want_blocks = (n + waves - 1) / waves;
    const int blocks = (int)std::min<long long>(want_blocks, 256LL * 8);
    auto kern = vec ? sketch_rows_kernel<T, MODE, true> : sketch_rows_kernel<T, MODE, false>;
    if (lds > 64 * 1024)
        FDX_HIP(hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    hipLaunchKernelGGL(kern, dim3(blocks), dim3(waves * 64), lds, st, Y, ldy, row_map, n, G, d, plan.sched_gene,
                       plan.sched_w, plan.group_off, plan.n_groups, plan.slot_bucket, Ys, ldys, row_sumsq);
    FDX_CHECK_LAUNCH();
    return 0;
}

template <typename T>
static int launch_sketch_t(const T* Y, long long ldy, const int* row_map, long long n, int G, int d, int mode,
                           const SketchPlanDev& plan, double* Ys, long long ldys, double* row_sumsq, hipStream_t st) {
    switch (mode) {
        case FDX_PRE_RAW: return launch_sketch_mode<T, FDX_PRE_RAW>(Y, ldy, row_map, n, G, d, plan, Ys, ldys, row_sumsq, st);
        case FDX_PRE_LOG_CPM: return launch_sketch_mode<T, FDX_PRE_LOG_CPM>(Y, ldy, row_map, n, G, d, plan, Ys, ldys, row_sumsq, st);
        case FDX_PRE_LOG_CPM_SPARSE: return launch_sketch_mode<T, FDX_PRE_LOG_CPM_SPARSE>(Y, ldy, row_map, n, G, d, plan, Ys, ldys, row_sumsq, st);
        default: return fail(FDX_ERR_INVALID, "sketch: unknown preprocess mode");
    }
}

int launch_sketch_rows(const void* Y, int dtype, long long ldy, const int* row_map, long long n, int G, int d, int mode,
                       const SketchPlanDev& plan, double* Ys, long long ldys, double* row_sumsq, hipStream_t st) {
    if (n <= 0 || d <= 0) return 0;
    if (dtype == FDX_F32) return launch_sketch_t<float>((const float*)Y, ldy, row_map, n, G, d, mode, plan, Ys, ldys, row_sumsq, st);
    if (dtype == FDX_F64) return launch_sketch_t<double>((const double*)Y, ldy, row_map, n, G, d, mode, plan, Ys, ldys, row_sumsq, st);
    return fail(FDX_ERR_INVALID, "sketch: dtype must be FDX_F32 or FDX_F64");
}

int column_sums_parts(long long n) { return (int)std::min<long long>(512, std::max<long long>(1, (n + 255) / 256)); }

int launch_column_sums(const void* Y, int dtype, long long ldy, long long n, int G, double* partials, double* out,
                       hipStream_t st) {
    if (G <= 0) return 0;
    const int parts = column_sums_parts(n);
    const int rows_per_block = (int)((n + parts - 1) / parts);
    dim3 grid(ceil_div(G, 256), parts);
    if (dtype == FDX_F32)
        hipLaunchKernelGGL(column_sums_kernel<float>, grid, dim3(256), 0, st, (const float*)Y, ldy, n, G, rows_per_block, partials);
    else if (dtype == FDX_F64)
        hipLaunchKernelGGL(column_sums_kernel<double>, grid, dim3(256), 0, st, (const double*)Y, ldy, n, G, rows_per_block, partials);
    else
        return fail(FDX_ERR_INVALID, "column_sums: dtype must be FDX_F32 or FDX_F64");
    FDX_CHECK_LAUNCH();
    hipLaunchKernelGGL(fold_column_partials_kernel, dim3(ceil_div(G, 256)), dim3(256), 0, st, partials, parts, G, out);
    FDX_CHECK_LAUNCH();
    return 0;
}

}  // namespace fdx
