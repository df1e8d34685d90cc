// Fused preprocess + CountSketch + H contraction with LDS atomics:  H = X_sketch * f(Y) Omega ^T  without writing Y_sketch.
//
// The first fused form of the stage (flashdeconv/core/deconv.py:177-197, core/sketching.py:160-206,
// core/solver.py:205-223), kept as the alternative behind FDX_NO_TILE=1 and for the shapes the tile kernel
// (tile_kernels.cpp: atomic-free, 1.9 ms against 2.9 ms at 1M x 2000 x 30) does not take: row lengths that are not a whole
// number of 16-byte vectors.  launch_sketch_contract() dispatches: tile kernel when its schedule exists, else this one.
// A 16-wave workgroup works on GROUPS of 16 consecutive spots (solver order):
//   scatter phase   wave w streams the row of spot w of the group from HBM straight into registers and adds
//                   weight * f(y) into that spot's d-entry accumulator row in LDS (ds_add_f64; per-gene {weight, bucket}
//                   table in LDS) - exactly the arithmetic of sketch_rows_scatter_kernel; ||row||^2 goes to row_sumsq;
//   contract phase  the 16 x d block now sitting in LDS is the B operand of v_mfma_f64_16x16x4_f64; as in
//                   xyt_split_kernel the contraction index is split over the waves, each holding its slice of X_sketch
//                   as register-resident A operands, and the partial tiles are summed in wave order through LDS.
// Bound by the LDS atomics (ds_add_f64 on random buckets: ~5-way bank conflicts, see profiles/r02_pmc_counters.md).
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

constexpr int FUSED_PAD = 16;    // doubles of padding per accumulator row: row stride = 128 B mod 4 KB -> conflict-free reads

// R: spots per group = waves per workgroup (one row per wave; the MFMA tile has 16 columns, columns >= R are zero);
// NB: 16-wide blocks of the contraction index per wave (d <= 16 * R * NB); TT: 16-type tiles (K <= 16 * TT).
// R = 16 (one 16-wave workgroup per CU, 96 KB of LDS) is the default; R = 8 (FDX_FUSED_ROWS=8: 58 KB, two workgroups per
// CU that drift apart, half-empty MFMA tiles) measures 10 % slower - the phases were not the problem, the bank conflicts
// of the gene-order table were (see the table layout below: 3.25 -> 2.91 ms).
template <typename T, int MODE, bool VEC, int R, int NB, int TT>
__global__ __launch_bounds__(R * 64, 4) void sketch_contract_kernel(const T* __restrict__ Y, long long ldy,
                                                               const int* __restrict__ row_map, long long n, int G, int d,
                                                               const double* __restrict__ gene_w,
                                                               const int* __restrict__ gene_bucket,
                                                               const double* __restrict__ Xs, int K,
                                                               double* __restrict__ Hout, long long ldh,
                                                               double* __restrict__ row_sumsq) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int Gp = (G + 256 + 7) & ~7;                                    // table capacity (lane-major layout, see below)
    const int rs = d + FUSED_PAD;                                         // accumulator row stride (doubles)
    double* w_l = reinterpret_cast<double*>(smem);                        // [Gp]
    double* rows = w_l + Gp;                                              // [16][rs]; re-used as red[16][TT*4*64]
    const int region = max(R * rs, R * TT * 4 * 64);
    double* tabs = rows + region;                                         // [16][64] per-wave log1p tables
    unsigned short* b_l = reinterpret_cast<unsigned short*>(tabs + R * 64);   // [Gp]
    double* red = rows;
    // The per-gene table is stored LANE-MAJOR: the entry of gene g = (v0 + u*64 + lane)*PER + e sits at
    // ((v0/64 + u)*PER + e)*64 + lane, so the 64 lanes of a table read touch 64 consecutive entries.  In gene order a lane's
    // PER genes are PER*8 bytes apart and a wave's ds_read_b64 hits every bank 8 times.  Genes past the last full
    // 16-byte vector (and all genes of the scalar path) follow in gene order.
    typedef typename FVec4<T>::type V;
    constexpr int PER = 16 / sizeof(T);
    const int nvec = VEC ? G / PER : 0;
    const int tail_base = ((nvec + 63) >> 6) * 64 * PER;
    for (int g = tid; g < G; g += R * 64) {
        const int b = gene_bucket[g];
        const int v = g / PER, e = g - v * PER;
        const int idx = (g < nvec * PER) ? ((((v >> 6) * PER + e) << 6) + (v & 63)) : (tail_base + (g - nvec * PER));
        w_l[idx] = (b >= 0) ? gene_w[g] : 0.0;
        b_l[idx] = (unsigned short)(b >= 0 ? b : 0xFFFF);                 // 0xFFFF: gene has no entry in Omega
    }
    // this wave's slice of X_sketch as MFMA A operands: the contraction index is split over the 16 waves
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
    double* acc = rows + (size_t)wave * rs;
    double* tab = tabs + (size_t)wave * 64;
    const long long n_groups = (n + R - 1) / R;
    for (long long grp = blockIdx.x; grp < n_groups; grp += gridDim.x) {
        const long long s0 = grp * R;
        const long long p = s0 + wave;
        // ---- scatter phase: one spot per wave (the arithmetic of sketch_rows_scatter_kernel)
        for (int c = lane; c < d; c += 64) acc[c] = 0.0;
        if (p < n) {                                                      // wave-uniform: spots past the end stay zero
            const long long row = row_map ? (long long)row_map[p] : p;
            const T* yrow = Y + (size_t)row * ldy;
            const V* src = reinterpret_cast<const V*>(yrow);
            double scale = 1.0;
            bool use_tab = false;
            if (MODE != FDX_PRE_RAW) {
                double part = 0.0;
                T mx = (T)0;
                for (int v0 = 0; v0 < nvec; v0 += 256) {   // 4 loads per batch: 16 waves share 512 VGPRs per SIMD lane
                    V x[4];
#pragma unroll
                    for (int u = 0; u < 4; ++u) {
                        const int v = v0 + u * 64 + lane;
                        if (v < nvec) x[u] = src[v];
                    }
#pragma unroll
                    for (int u = 0; u < 4; ++u) {
                        const int v = v0 + u * 64 + lane;
                        if (v < nvec) {
#pragma unroll
                            for (int e = 0; e < PER; ++e) { part += (double)x[u][e]; mx = x[u][e] > mx ? x[u][e] : mx; }
                        }
                    }
                }
                for (int g = nvec * PER + lane; g < G; g += 64) { part += (double)yrow[g]; mx = yrow[g] > mx ? yrow[g] : mx; }
                double sum = wave_sum(part);
                use_tab = wave_max((double)mx) < 64.0;
                if (MODE == FDX_PRE_LOG_CPM) {
                    scale = (1.0 / (sum + 1e-10)) * 1e4;                  // y / (rowsum + 1e-10) * 1e4   (deconv.py:190)
                } else {
                    if (sum == 0.0) sum = 1.0;                            // lib_size[lib_size == 0] = 1  (deconv.py:183-185)
                    scale = 1e4 / sum;
                }
                if (use_tab) log1p_table_fill(tab, scale, lane);
            }
            __builtin_amdgcn_s_waitcnt(0xc07f);                           // zeroing (and the table) done before the adds
            for (int v0 = 0; v0 < nvec; v0 += 256) {
                V x[4];
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    const int v = v0 + u * 64 + lane;
                    if (v < nvec) x[u] = src[v];
                }
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    const int v = v0 + u * 64 + lane;
                    if (v < nvec) {
#pragma unroll
                        for (int e = 0; e < PER; ++e) {
                            const int ti = ((((v0 >> 6) + u) * PER + e) << 6) + lane;      // lane-major table index of gene v*PER+e
                            double y = (double)x[u][e];
                            if (MODE != FDX_PRE_RAW) y = log1p_scaled(y, scale, tab, use_tab);
                            const unsigned b = b_l[ti];
                            if (b != 0xFFFFu) __hip_atomic_fetch_add(acc + b, w_l[ti] * y, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                            if (MODE != FDX_PRE_RAW) __builtin_amdgcn_sched_barrier(0);   // one log1p at a time: interleaving
                        }                                                                   // four of them spills at 128 VGPRs
                    }
                }
            }
            for (int g = nvec * PER + lane; g < G; g += 64) {
                double y = (double)yrow[g];
                if (MODE != FDX_PRE_RAW) y = log1p_scaled(y, scale, tab, use_tab);
                const int ti = tail_base + (g - nvec * PER);
                const unsigned b = b_l[ti];
                if (b != 0xFFFFu) __hip_atomic_fetch_add(acc + b, w_l[ti] * y, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
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
        // ---- contract phase: B operand from LDS, this wave's slice of the contraction index
        double4_t accm[TT];
#pragma unroll
        for (int t = 0; t < TT; ++t) accm[t] = double4_t{0.0, 0.0, 0.0, 0.0};
        const double* yrow_l = rows + (size_t)r * rs;
#pragma unroll
        for (int b = 0; b < NB; ++b) {
            const int c0 = (wave * NB + b) * 16 + 4 * q;
            const double4_t bv = (c0 < d && r < R) ? *reinterpret_cast<const double4_t*>(yrow_l + c0) : double4_t{0.0, 0.0, 0.0, 0.0};
            const double x[4] = {bv.x, bv.y, bv.z, bv.w};
#pragma unroll
            for (int t = 0; t < TT; ++t)
#pragma unroll
                for (int s = 0; s < 4; ++s) accm[t] = __builtin_amdgcn_mfma_f64_16x16x4f64(a[b][t][s], x[s], accm[t], 0, 0, 0);
        }
        __syncthreads();                                                  // every wave has read its B operands: rows -> red
#pragma unroll
        for (int t = 0; t < TT; ++t)
#pragma unroll
            for (int rr = 0; rr < 4; ++rr) red[(size_t)wave * (TT * 4 * 64) + (t * 4 + rr) * 64 + lane] = accm[t][rr];
        __syncthreads();
        for (int o = tid; o < TT * 4 * 64; o += R * 64) {
            double sum = 0.0;
#pragma unroll
            for (int v = 0; v < R; ++v) sum += red[(size_t)v * (TT * 4 * 64) + o];   // wave order: deterministic
            const int l = o & 63, tr = o >> 6;
            const int type = (tr >> 2) * 16 + (l >> 4) + 4 * (tr & 3);
            const long long sp = s0 + (l & 15);
            if (type < K && (l & 15) < R && sp < n) Hout[(size_t)type * ldh + sp] = sum;
        }
        __syncthreads();                                                  // red is rows again for the next group
    }
}

static int fused_rows() {             // spots per group (waves per workgroup): 16 unless FDX_FUSED_ROWS=8
    const char* e = getenv("FDX_FUSED_ROWS");
    return (e && atoi(e) == 8) ? 8 : 16;
}

size_t fused_lds_bytes(int G, int d, int K) {
    const size_t R = (size_t)fused_rows();
    const size_t Gp = ((size_t)G + 256 + 7) & ~(size_t)7;                   // lane-major table: up to 64 * PER slots of padding
    const size_t TT = (size_t)(K + 15) / 16;
    const size_t region = std::max(R * (d + FUSED_PAD), R * TT * 4 * 64);   // rows / reduction
    return Gp * 10 + region * 8 + R * 64 * 8;
}

bool fused_sketch_contract_ok(int dtype, long long ldy, const void* Y, int G, int d, int K, int mode, const SketchPlanDev& plan,
                              hipStream_t st) {
    if (getenv("FDX_NO_FUSED")) return false;
    if (tile_sketch_ok(dtype, ldy, Y, G, d, K, mode, plan, st)) return true;
    // History: an 8-wave version (two rows per wave, 120 KB of LDS) lost to the two-kernel path (3.8 vs 3.5 ms): all waves
    // move through scatter / barrier / MFMA / reduce together, so HBM idled outside the scatter phase and two waves per SIMD
    // could not hide a row's dependent LDS chain.  This version: 16 waves, one row each, reduction buffer aliased onto the
    // accumulator rows (96 KB of LDS): 3.2 ms against 2.53 + 0.95 ms.  Prefetching the next batch / the next group's row
    // across the barriers was tried twice and changes nothing: per group of 16 spots the CU spends ~5 us in its LDS pipe and
    // ~7 us waiting for HBM, and with all 16 waves in the same phase the two do not overlap the way 24 independent waves of
    // the scatter kernel do.  RAW (and pearson, which is RAW with scaled weights): default.  log-CPM: the 128-VGPR budget of 16 waves
    // spills around fast_log1p and the fused form loses (4.75 vs 4.27 ms on counts) - two kernels unless FDX_FUSED=1.
    if (getenv("FDX_NO_FUSED")) return false;
    if (mode != FDX_PRE_RAW && !getenv("FDX_FUSED")) return false;
    if (!plan.scatter_ok || d % 16 != 0 || d > 512 || K > 32 || K <= 0 || G <= 0) return false;
    if (dtype != FDX_F32 && dtype != FDX_F64) return false;
    (void)ldy; (void)Y;
    return fused_lds_bytes(G, d, K) <= 150 * 1024;
}

template <typename T, int MODE, bool VEC>
static int launch_fused_nb(const T* Y, long long ldy, const int* row_map, long long n, int G, int d, const SketchPlanDev& plan,
                           const double* Xs, int K, double* H, long long ldh, double* row_sumsq, hipStream_t st) {
    const int R = fused_rows();
    const int nb = (d + 16 * R - 1) / (16 * R), TT = (K + 15) / 16;   // R waves x NB blocks of 16 cover the contraction index
    const size_t lds = fused_lds_bytes(G, d, K);
    const long long groups = (n + R - 1) / R;
    const int per_cu = std::max<int>(1, (int)((160 * 1024) / lds));
    const int grid = (int)std::min<long long>(groups, 256LL * std::min(per_cu, R == 8 ? 2 : 1));
    const void* kern = nullptr;
#define FDX_FUSED(R_, NB_, TT_) kern = (const void*)sketch_contract_kernel<T, MODE, VEC, R_, NB_, TT_>
    if (R == 16) {
        if (nb <= 1) { if (TT == 1) FDX_FUSED(16, 1, 1); else FDX_FUSED(16, 1, 2); }
        else { if (TT == 1) FDX_FUSED(16, 2, 1); else FDX_FUSED(16, 2, 2); }
    } else {
        if (nb <= 1) { if (TT == 1) FDX_FUSED(8, 1, 1); else FDX_FUSED(8, 1, 2); }
        else if (nb <= 2) { if (TT == 1) FDX_FUSED(8, 2, 1); else FDX_FUSED(8, 2, 2); }
        else { if (TT == 1) FDX_FUSED(8, 4, 1); else FDX_FUSED(8, 4, 2); }
    }
#undef FDX_FUSED
    if (lds > 64 * 1024) FDX_HIP(hipFuncSetAttribute(kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    void* args[] = {(void*)&Y, (void*)&ldy, (void*)&row_map, (void*)&n, (void*)&G, (void*)&d, (void*)&plan.gene_w,
                    (void*)&plan.gene_bucket, (void*)&Xs, (void*)&K, (void*)&H, (void*)&ldh, (void*)&row_sumsq};
    FDX_HIP(hipLaunchKernel(kern, dim3(grid), dim3(R * 64), args, lds, st));
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
    if (tile_sketch_ok(dtype, ldy, Y, G, d, K, mode, plan, st))
        return launch_tile_sketch(Y, dtype, ldy, row_map, n, G, d, mode, plan, Xs, K, H, ldh, row_sumsq, st);
    if ((reinterpret_cast<uintptr_t>(Xs) & 31) != 0) return fail(FDX_ERR_INVALID, "fused sketch: X_sketch must be 32-byte aligned");
    if (dtype == FDX_F32)
        return launch_fused_t<float>((const float*)Y, ldy, row_map, n, G, d, mode, plan, Xs, K, H, ldh, row_sumsq, st);
    if (dtype == FDX_F64)
        return launch_fused_t<double>((const double*)Y, ldy, row_map, n, G, d, mode, plan, Xs, K, H, ldh, row_sumsq, st);
    return fail(FDX_ERR_INVALID, "fused sketch: dtype must be FDX_F32 or FDX_F64");
}

}  // namespace fdx
