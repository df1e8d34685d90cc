// Fused preprocess + CountSketch + H contraction:  H = X_sketch * f(Y) Omega ^T  without ever writing Y_sketch.
//
// Replaces the pair  sketch_rows_scatter_kernel -> xyt_split_kernel  (flashdeconv/core/deconv.py:177-197,
// core/sketching.py:160-206, core/solver.py:205-223) for the common shape (a CountSketch, sketch_dim <= 512 and a
// multiple of 16, K <= 32).  The two-kernel form writes the (N, d) float64 sketch and reads it back: 8.6 GB of HBM
// traffic per million spots at d = 512, more than the 8 GB of Y itself.  Here a workgroup of 8 waves works on GROUPS of
// 16 consecutive spots (solver order):
//   scatter phase   wave w streams the rows of spots 2w and 2w+1 of the group from HBM straight into registers and adds
//                   weight * f(y) into that spot's d-entry accumulator row in LDS (ds_add_f64; per-gene {weight, bucket}
//                   table in LDS) - exactly the arithmetic of sketch_rows_scatter_kernel; ||row||^2 goes to row_sumsq;
//   contract phase  the 16 x d block now sitting in LDS is the B operand of v_mfma_f64_16x16x4_f64; as in
//                   xyt_split_kernel the contraction index is split over the 8 waves, each holding its slice of X_sketch
//                   as register-resident A operands, and the 8 partial tiles are summed in wave order through LDS.
// The MFMA sequence and the reduction order are those of xyt_split_kernel, so H has the same bits as the two-kernel path
// (asserted in tests).  STATUS: correct but not faster yet - see fused_sketch_contract_ok() - hence opt-in.
#include <algorithm>
#include <cstdlib>

#include "device_math.h"
#include "fdx_internal.h"
#include "fdx_kernels.h"

namespace fdx {

typedef double double4_t __attribute__((ext_vector_type(4)));
template <typename T> struct FVec4;
template <> struct FVec4<float> { typedef float type __attribute__((ext_vector_type(4))); };
template <> struct FVec4<double> { typedef double type __attribute__((ext_vector_type(2))); };

constexpr int FUSED_ROWS = 16;   // spots per group = MFMA columns
constexpr int FUSED_PAD = 16;    // doubles of padding per accumulator row: row stride = 128 B mod 4 KB -> conflict-free reads

template <typename T, int MODE, bool VEC, int NB, int TT>
__global__ __launch_bounds__(512) void sketch_contract_kernel(const T* __restrict__ Y, long long ldy,
                                                              const int* __restrict__ row_map, long long n, int G, int d,
                                                              const double* __restrict__ gene_w,
                                                              const int* __restrict__ gene_bucket,
                                                              const double* __restrict__ Xs, int K,
                                                              double* __restrict__ Hout, long long ldh,
                                                              double* __restrict__ row_sumsq) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int Gp = (G + 7) & ~7;
    const int rs = d + FUSED_PAD;                                         // accumulator row stride (doubles)
    double* w_l = reinterpret_cast<double*>(smem);                        // [Gp]
    double* rows = w_l + Gp;                                              // [16][rs]
    double* red = rows + FUSED_ROWS * rs;                                 // [8][TT*4*64]
    unsigned short* b_l = reinterpret_cast<unsigned short*>(red + 8 * TT * 4 * 64);   // [Gp]
    for (int g = tid; g < Gp; g += 512) {
        const int b = (g < G) ? gene_bucket[g] : -1;
        w_l[g] = (g < G && b >= 0) ? gene_w[g] : 0.0;
        b_l[g] = (unsigned short)(b >= 0 ? b : 0xFFFF);                   // 0xFFFF: gene has no entry in Omega
    }
    // this wave's slice of X_sketch as MFMA A operands (xyt_split_kernel's layout)
    const int r = lane & 15, q = lane >> 4;
    double a[NB][TT][4];
#pragma unroll
    for (int b = 0; b < NB; ++b) {
        const int c0 = (wave * NB + b) * 16 + 4 * q;
#pragma unroll
        for (int t = 0; t < TT; ++t) {
            const int type = t * 16 + r;
            const bool ok = type < K && c0 < d;
            const double4_t v = ok ? *reinterpret_cast<const double4_t*>(Xs + (size_t)type * d + c0) : double4_t{0.0, 0.0, 0.0, 0.0};
            a[b][t][0] = v.x; a[b][t][1] = v.y; a[b][t][2] = v.z; a[b][t][3] = v.w;
        }
    }
    __syncthreads();
    typedef typename FVec4<T>::type V;
    constexpr int PER = 16 / sizeof(T);
    const int nvec = VEC ? G / PER : 0;
    const int nbat = (nvec + 511) / 512;                                  // batches of 8 x 64 sixteen-byte loads per row
    const long long n_groups = (n + FUSED_ROWS - 1) / FUSED_ROWS;
    // Software pipeline over this wave's rows: `cur` always holds batch 0 of the row about to be processed; the loads of
    // the following batch - of the same row, of the wave's other row, or of its first row in the NEXT group - are
    // issued before the current batch is consumed, so they are in flight across the barriers and the MFMA phase (8 waves
    // per CU cannot hide HBM latency by occupancy alone: the un-pipelined version of this kernel was slower than the
    // two kernels it replaces).
    V cur[8], nxt[8];
    auto row_src = [&](long long p) -> const T* {
        const long long row = row_map ? (long long)row_map[p] : p;
        return Y + (size_t)row * ldy;
    };
#define FDX_LOAD_BATCH(X_, YROW_, BI_)                                               \
    do {                                                                             \
        const V* src_ = reinterpret_cast<const V*>(YROW_);                           \
        _Pragma("unroll") for (int u_ = 0; u_ < 8; ++u_) {                           \
            const int v_ = (BI_) * 512 + u_ * 64 + lane;                             \
            if (v_ < nvec) X_[u_] = src_[v_];                                        \
        }                                                                            \
    } while (0)
    {
        const long long p0 = (long long)blockIdx.x * FUSED_ROWS + 2 * wave;
        if (nbat > 0 && blockIdx.x < n_groups && p0 < n) FDX_LOAD_BATCH(cur, row_src(p0), 0);
    }
    for (long long grp = blockIdx.x; grp < n_groups; grp += gridDim.x) {
        const long long s0 = grp * FUSED_ROWS;
        // ---- scatter phase: two spots per wave
#pragma unroll 1
        for (int half = 0; half < 2; ++half) {
            const int lr = 2 * wave + half;                               // row of the group
            const long long p = s0 + lr;
            double* acc = rows + (size_t)lr * rs;
            for (int c = lane; c < d; c += 64) acc[c] = 0.0;
            if (p >= n) continue;                                         // wave-uniform: spots past the end stay zero
            const T* yrow = row_src(p);
            // the row this wave streams after the current one (batch 0 is prefetched at the end of this row)
            long long pn = half == 0 ? p + 1 : (grp + gridDim.x) * (long long)FUSED_ROWS + 2 * wave;
            if (half == 1 && grp + gridDim.x >= n_groups) pn = n;
            const T* ynext = (pn < n) ? row_src(pn) : yrow;
            double scale = 1.0;
            if (MODE != FDX_PRE_RAW) {
                double part = 0.0;
                for (int bi = 0; bi < nbat; ++bi) {
                    if (bi + 1 < nbat) FDX_LOAD_BATCH(nxt, yrow, bi + 1);
                    else if (nbat > 1) FDX_LOAD_BATCH(nxt, yrow, 0);          // back to the start for the main pass (L2 hit)
#pragma unroll
                    for (int u = 0; u < 8; ++u) {
                        const int v = bi * 512 + u * 64 + lane;
                        if (v < nvec) {
#pragma unroll
                            for (int e = 0; e < PER; ++e) part += (double)cur[u][e];
                        }
                    }
                    if (nbat > 1) {
#pragma unroll
                        for (int u = 0; u < 8; ++u) cur[u] = nxt[u];
                    }
                }
                for (int g = nvec * PER + lane; g < G; g += 64) part += (double)yrow[g];
                double sum = wave_sum(part);
                if (MODE == FDX_PRE_LOG_CPM) {
                    scale = (1.0 / (sum + 1e-10)) * 1e4;                  // y / (rowsum + 1e-10) * 1e4   (deconv.py:190)
                } else {
                    if (sum == 0.0) sum = 1.0;                            // lib_size[lib_size == 0] = 1  (deconv.py:183-185)
                    scale = 1e4 / sum;
                }
            }
            __builtin_amdgcn_s_waitcnt(0xc07f);                           // zeroing done before the adds
            for (int bi = 0; bi < nbat; ++bi) {
                if (bi + 1 < nbat) FDX_LOAD_BATCH(nxt, yrow, bi + 1);
                else if (pn < n) FDX_LOAD_BATCH(nxt, ynext, 0);
#pragma unroll
                for (int u = 0; u < 8; ++u) {
                    const int v = bi * 512 + u * 64 + lane;
                    if (v < nvec) {
#pragma unroll
                        for (int e = 0; e < PER; ++e) {
                            const int g = v * PER + e;
                            double y = (double)cur[u][e];
                            if (MODE != FDX_PRE_RAW) y = fast_log1p(y * scale);
                            const unsigned b = b_l[g];
                            if (b != 0xFFFFu) __hip_atomic_fetch_add(acc + b, w_l[g] * y, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                        }
                    }
                    if (u & 1) __builtin_amdgcn_sched_barrier(0);         // keep the table reads of 32 elements from being
                                                                          // hoisted together (420 live registers otherwise)
                }
#pragma unroll
                for (int u = 0; u < 8; ++u) cur[u] = nxt[u];
            }
            for (int g = nvec * PER + lane; g < G; g += 64) {
                double y = (double)yrow[g];
                if (MODE != FDX_PRE_RAW) y = fast_log1p(y * scale);
                const unsigned b = b_l[g];
                if (b != 0xFFFFu) __hip_atomic_fetch_add(acc + b, w_l[g] * y, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            }
            __builtin_amdgcn_s_waitcnt(0xc07f);
            if (row_sumsq) {
                double sq = 0.0;
                for (int c = lane; c < d; c += 64) {
                    const double v = acc[c];
                    sq = fma(v, v, sq);
                }
                sq = wave_sum(sq);
                if (lane == 0) row_sumsq[p] = sq;
            }
        }
        __syncthreads();                                                  // the 16 x d block is complete
        // ---- contract phase (xyt_split_kernel's MFMA sequence; B operand from LDS)
        double4_t accm[TT];
#pragma unroll
        for (int t = 0; t < TT; ++t) accm[t] = double4_t{0.0, 0.0, 0.0, 0.0};
        const double* yrow_l = rows + (size_t)r * rs;
#pragma unroll
        for (int b = 0; b < NB; ++b) {
            const int c0 = (wave * NB + b) * 16 + 4 * q;
            const double4_t bv = (c0 < d) ? *reinterpret_cast<const double4_t*>(yrow_l + c0) : double4_t{0.0, 0.0, 0.0, 0.0};
            const double x[4] = {bv.x, bv.y, bv.z, bv.w};
#pragma unroll
            for (int t = 0; t < TT; ++t)
#pragma unroll
                for (int s = 0; s < 4; ++s) accm[t] = __builtin_amdgcn_mfma_f64_16x16x4f64(a[b][t][s], x[s], accm[t], 0, 0, 0);
        }
#pragma unroll
        for (int t = 0; t < TT; ++t)
#pragma unroll
            for (int rr = 0; rr < 4; ++rr) red[(size_t)wave * (TT * 4 * 64) + (t * 4 + rr) * 64 + lane] = accm[t][rr];
        __syncthreads();
        for (int o = tid; o < TT * 4 * 64; o += 512) {
            double sum = 0.0;
#pragma unroll
            for (int v = 0; v < 8; ++v) sum += red[(size_t)v * (TT * 4 * 64) + o];
            const int l = o & 63, tr = o >> 6;
            const int type = (tr >> 2) * 16 + (l >> 4) + 4 * (tr & 3);
            const long long sp = s0 + (l & 15);
            if (type < K && sp < n) Hout[(size_t)type * ldh + sp] = sum;
        }
        // next group: rows[] is rewritten only by waves that passed the barrier above (all MFMA reads done); red[] is
        // rewritten after the next group's first barrier, by which time every thread has finished the sums above
    }
}

size_t fused_lds_bytes(int G, int d, int K) {
    const size_t Gp = ((size_t)G + 7) & ~(size_t)7;
    const int TT = (K + 15) / 16;
    return Gp * 10 + (size_t)FUSED_ROWS * (d + FUSED_PAD) * 8 + (size_t)8 * TT * 4 * 64 * 8;
}

bool fused_sketch_contract_ok(int dtype, long long ldy, const void* Y, int G, int d, int K, const SketchPlanDev& plan) {
    // Opt-in (FDX_FUSED=1).  Measured on MI355X at 1M x 2000 -> 512, K = 30: 3.7-3.8 ms against 2.56 + 0.95 ms for the two
    // kernels it replaces, with or without the software pipeline.  The LDS footprint (table 20 KB + 16 accumulator rows
    // 68 KB + reduction 32 KB) allows one 8-wave workgroup per CU; a wave needs ~6.5 us per row (32 dependent
    // table-read -> atomic steps per lane, zeroing, norm) and two waves per SIMD cannot hide that.  A register-resident
    // gene table was tried on the scatter kernel and is no faster, so the LDS reads are not the bound.  Kept, with its
    // bit-equality test, as the starting point for a variant that holds >= 16 waves per CU.
    if (!getenv("FDX_FUSED") || getenv("FDX_NO_FUSED")) return false;
    if (!plan.scatter_ok || d % 16 != 0 || d > 512 || K > 32 || K <= 0 || G <= 0) return false;
    if (dtype != FDX_F32 && dtype != FDX_F64) return false;
    (void)ldy; (void)Y;
    return fused_lds_bytes(G, d, K) <= 150 * 1024;
}

template <typename T, int MODE, bool VEC>
static int launch_fused_nb(const T* Y, long long ldy, const int* row_map, long long n, int G, int d, const SketchPlanDev& plan,
                           const double* Xs, int K, double* H, long long ldh, double* row_sumsq, hipStream_t st) {
    const int nb = (d + 127) / 128, TT = (K + 15) / 16;
    const size_t lds = fused_lds_bytes(G, d, K);
    const long long groups = (n + FUSED_ROWS - 1) / FUSED_ROWS;
    const int grid = (int)std::min<long long>(groups, 256);
    const void* kern = nullptr;
#define FDX_FUSED(NB_, TT_) kern = (const void*)sketch_contract_kernel<T, MODE, VEC, NB_, TT_>
    if (nb <= 1) { if (TT == 1) FDX_FUSED(1, 1); else FDX_FUSED(1, 2); }
    else if (nb <= 2) { if (TT == 1) FDX_FUSED(2, 1); else FDX_FUSED(2, 2); }
    else { if (TT == 1) FDX_FUSED(4, 1); else FDX_FUSED(4, 2); }
#undef FDX_FUSED
    if (lds > 64 * 1024) FDX_HIP(hipFuncSetAttribute(kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    void* args[] = {(void*)&Y, (void*)&ldy, (void*)&row_map, (void*)&n, (void*)&G, (void*)&d, (void*)&plan.gene_w,
                    (void*)&plan.gene_bucket, (void*)&Xs, (void*)&K, (void*)&H, (void*)&ldh, (void*)&row_sumsq};
    FDX_HIP(hipLaunchKernel(kern, dim3(grid), dim3(512), args, lds, st));
    return 0;
}

template <typename T>
static int launch_fused_t(const T* Y, long long ldy, const int* row_map, long long n, int G, int d, int mode,
                          const SketchPlanDev& plan, const double* Xs, int K, double* H, long long ldh, double* row_sumsq,
                          hipStream_t st) {
    const bool vec = (ldy % (16 / (long long)sizeof(T)) == 0) && ((reinterpret_cast<uintptr_t>(Y) & 15) == 0);
#define FDX_FUSED_MODE(M_)                                                                                             \
    return vec ? launch_fused_nb<T, M_, true>(Y, ldy, row_map, n, G, d, plan, Xs, K, H, ldh, row_sumsq, st)             \
               : launch_fused_nb<T, M_, false>(Y, ldy, row_map, n, G, d, plan, Xs, K, H, ldh, row_sumsq, st)
    switch (mode) {
        case FDX_PRE_RAW: FDX_FUSED_MODE(FDX_PRE_RAW);
        case FDX_PRE_LOG_CPM: FDX_FUSED_MODE(FDX_PRE_LOG_CPM);
        case FDX_PRE_LOG_CPM_SPARSE: FDX_FUSED_MODE(FDX_PRE_LOG_CPM_SPARSE);
        default: return fail(FDX_ERR_INVALID, "fused sketch: unknown preprocess mode");
    }
#undef FDX_FUSED_MODE
}

// H[:, 0..n) (type-major, row stride ldh) and row_sumsq[0..n) for the n spots listed by row_map (NULL = rows 0..n-1 of Y).
// Call only when fused_sketch_contract_ok(...) holds; Xs must be 32-byte aligned.
int launch_sketch_contract(const void* Y, int dtype, long long ldy, const int* row_map, long long n, int G, int d, int mode,
                           const SketchPlanDev& plan, const double* Xs, int K, double* H, long long ldh, double* row_sumsq,
                           hipStream_t st) {
    if (n <= 0) return 0;
    if ((reinterpret_cast<uintptr_t>(Xs) & 31) != 0) return fail(FDX_ERR_INVALID, "fused sketch: X_sketch must be 32-byte aligned");
    if (dtype == FDX_F32)
        return launch_fused_t<float>((const float*)Y, ldy, row_map, n, G, d, mode, plan, Xs, K, H, ldh, row_sumsq, st);
    if (dtype == FDX_F64)
        return launch_fused_t<double>((const double*)Y, ldy, row_map, n, G, d, mode, plan, Xs, K, H, ldh, row_sumsq, st);
    return fail(FDX_ERR_INVALID, "fused sketch: dtype must be FDX_F32 or FDX_F64");
}

}  // namespace fdx
