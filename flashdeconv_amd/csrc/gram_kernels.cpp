// Dense contractions of the path on the f64 matrix cores (v_mfma_f64_16x16x4_f64):
//   H   = X_sketch @ Y_sketch^T     flashdeconv/core/solver.py:204-223 (precompute_XtY), stored type-major
//   XtX = X_sketch @ X_sketch^T     flashdeconv/core/solver.py:187-201 (precompute_gram_matrix) - same kernel, n := K
//   YtY = ||Y_sketch||_F^2          flashdeconv/core/solver.py:348
// from a materialised Y_sketch (n, d).  The fused preprocess->sketch->H kernel (sketch_kernels.cpp) shares the
// operand mapping below but feeds the B operand straight from registers.
//
// MFMA mapping (cdna guide §3, f64 form): D(16x16) += A(16x4) * B(4x16); lane l supplies A[i=l&15][k=l>>4] and
// B[k=l>>4][j=l&15] and receives D[row=(l>>4)+4r][col=l&15], r=0..3.  We put CELL TYPES on the rows and SPOTS on
// the columns, so a lane ends up holding 4 types of ONE spot and 16 lanes write 16 consecutive spots of a type
// plane (128-byte segments of the type-major H).  Within a 16-wide block of the contraction index, lane (r, q=l>>4)
// loads the 4 CONSECUTIVE doubles 4q..4q+3 of row r (full 128-byte lines per row per wave) and MFMA step s
// contracts {s, 4+s, 8+s, 12+s} - any order works as long as A and B agree.
#include "fdx_env.h"
#include <algorithm>
#include <cstdlib>

#include "fdx_internal.h"
#include "fdx_kernels.h"

namespace fdx {

typedef double double4_t __attribute__((ext_vector_type(4)));

__device__ __forceinline__ void load4_guarded(const double* row, int c0, int d, bool row_ok, double (&v)[4]) {
#pragma unroll
    for (int s = 0; s < 4; ++s) v[s] = (row_ok && (c0 + s) < d) ? row[c0 + s] : 0.0;
}

// One wave = 16 spots x up to 64 types (4 accumulator tiles); grid-stride over type groups if K > 64.
template <bool ALIGNED>
__global__ __launch_bounds__(256) void xyt_kernel(const double* __restrict__ Xs, const double* __restrict__ Ys,
                                                  long long ldy, int n, int d, int K, double* __restrict__ Hout,
                                                  long long ldh, double* __restrict__ sumsq_partials) {
    const int lane = threadIdx.x & 63;
    const int wave = (int)((blockIdx.x * (long long)blockDim.x + threadIdx.x) >> 6);
    const int s0 = wave * 16;
    if (s0 >= n) return;
    const int r = lane & 15, q = lane >> 4;
    const int spot = s0 + r;
    const bool spot_ok = spot < n;
    const double* yrow = Ys + (size_t)(spot_ok ? spot : (n - 1)) * ldy;
    double sq = 0.0;
    const int n_tt = (K + 15) / 16;
    for (int tg = 0; tg < n_tt; tg += 4) {
        double4_t acc[4];
#pragma unroll
        for (int t = 0; t < 4; ++t) acc[t] = double4_t{0.0, 0.0, 0.0, 0.0};
        for (int d0 = 0; d0 < d; d0 += 16) {
            const int c0 = d0 + 4 * q;
            double bv[4];
            if (ALIGNED) {
                const double4_t t4 = *reinterpret_cast<const double4_t*>(yrow + c0);
                bv[0] = t4.x; bv[1] = t4.y; bv[2] = t4.z; bv[3] = t4.w;
                if (!spot_ok) { bv[0] = bv[1] = bv[2] = bv[3] = 0.0; }
            } else {
                load4_guarded(yrow, c0, d, spot_ok, bv);
            }
            if (tg == 0) sq += bv[0] * bv[0] + bv[1] * bv[1] + bv[2] * bv[2] + bv[3] * bv[3];
#pragma unroll
            for (int t = 0; t < 4; ++t) {
                const int type = (tg + t) * 16 + r;
                if ((tg + t) < n_tt) {  // wave-uniform
                    double av[4];
                    const bool type_ok = type < K;
                    const double* xrow = Xs + (size_t)(type_ok ? type : 0) * d;
                    if (ALIGNED) {
                        const double4_t t4 = *reinterpret_cast<const double4_t*>(xrow + c0);
                        av[0] = t4.x; av[1] = t4.y; av[2] = t4.z; av[3] = t4.w;
                        if (!type_ok) { av[0] = av[1] = av[2] = av[3] = 0.0; }
                    } else {
                        load4_guarded(xrow, c0, d, type_ok, av);
                    }
#pragma unroll
                    for (int s = 0; s < 4; ++s)
                        acc[t] = __builtin_amdgcn_mfma_f64_16x16x4f64(av[s], bv[s], acc[t], 0, 0, 0);
                }
            }
        }
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            if ((tg + t) < n_tt) {
#pragma unroll
                for (int rr = 0; rr < 4; ++rr) {
                    const int type = (tg + t) * 16 + q + 4 * rr;
                    if (type < K && spot_ok) Hout[(size_t)type * ldh + spot] = acc[t][rr];
                }
            }
        }
    }
    if (sumsq_partials) {
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) sq += __shfl_xor(sq, off, 64);
        if (lane == 0) sumsq_partials[wave] = sq;
    }
}

// Split-d variant for the large contraction H = X_s Y_s^T (n >> K): a workgroup of 8 waves walks 16-spot tiles; wave v
// owns the slice [v*16*NB, (v+1)*16*NB) of the contraction index and keeps ITS slice of X_sketch (the MFMA A operands of
// all T type tiles) in registers for the whole kernel, so X_sketch is read once per workgroup instead of once per tile
// (the plain kernel above re-reads it from L2 for every 16 spots: 2x the Y_sketch bytes).  Per tile each wave streams its
// 16 x 16*NB block of Y_sketch (full 128-byte lines), issues NB*T*4 MFMAs, and the 8 partial accumulators are summed in
// a fixed order through LDS (deterministic).  Requires d % 16 == 0, d <= 128*NB, K <= 16*T, 32-byte aligned rows.
template <int NB, int T>
__global__ __launch_bounds__(512) void xyt_split_kernel(const double* __restrict__ Xs, const double* __restrict__ Ys,
                                                        long long ldy, int n, int d, int K, double* __restrict__ Hout,
                                                        long long ldh) {
    // T <= 2: two reduction buffers, one barrier per tile; T = 3, 4: one 64 KB buffer, a second barrier per tile
    constexpr int NBUF = (T <= 2) ? 2 : 1;
    __shared__ double red[NBUF][8][T * 4 * 64];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int r = lane & 15, q = lane >> 4;
    double a[NB][T][4];
#pragma unroll
    for (int b = 0; b < NB; ++b) {
        const int c0 = (wave * NB + b) * 16 + 4 * q;
#pragma unroll
        for (int t = 0; t < T; ++t) {
            const int type = t * 16 + r;
            const bool ok = type < K && c0 < d;
            const double4_t v = ok ? *reinterpret_cast<const double4_t*>(Xs + (size_t)type * d + c0) : double4_t{0.0, 0.0, 0.0, 0.0};
            a[b][t][0] = v.x; a[b][t][1] = v.y; a[b][t][2] = v.z; a[b][t][3] = v.w;
        }
    }
    const int n_tiles = (n + 15) / 16;
    int buf = 0;
    for (int tile = blockIdx.x; tile < n_tiles; tile += gridDim.x, buf = (NBUF == 2) ? (buf ^ 1) : 0) {
        const int s0 = tile * 16;
        const int spot = s0 + r;
        const bool spot_ok = spot < n;
        const double* yrow = Ys + (size_t)(spot_ok ? spot : (n - 1)) * ldy;
        double4_t acc[T];
#pragma unroll
        for (int t = 0; t < T; ++t) acc[t] = double4_t{0.0, 0.0, 0.0, 0.0};
        double4_t bv[NB];
#pragma unroll
        for (int b = 0; b < NB; ++b) {
            const int c0 = (wave * NB + b) * 16 + 4 * q;
            bv[b] = (c0 < d) ? *reinterpret_cast<const double4_t*>(yrow + c0) : double4_t{0.0, 0.0, 0.0, 0.0};
        }
#pragma unroll
        for (int b = 0; b < NB; ++b) {
            double x[4] = {bv[b].x, bv[b].y, bv[b].z, bv[b].w};
            if (!spot_ok) { x[0] = x[1] = x[2] = x[3] = 0.0; }
#pragma unroll
            for (int t = 0; t < T; ++t)
#pragma unroll
                for (int s = 0; s < 4; ++s) acc[t] = __builtin_amdgcn_mfma_f64_16x16x4f64(a[b][t][s], x[s], acc[t], 0, 0, 0);
        }
#pragma unroll
        for (int t = 0; t < T; ++t)
#pragma unroll
            for (int rr = 0; rr < 4; ++rr) red[buf][wave][(t * 4 + rr) * 64 + lane] = acc[t][rr];
        __syncthreads();
        for (int o = tid; o < T * 4 * 64; o += 512) {
            double sum = 0.0;
#pragma unroll
            for (int v = 0; v < 8; ++v) sum += red[buf][v][o];
            const int l = o & 63, tr = o >> 6;
            const int type = (tr >> 2) * 16 + (l >> 4) + 4 * (tr & 3);
            const int sp = s0 + (l & 15);
            if (type < K && sp < n) Hout[(size_t)type * ldh + sp] = sum;
        }
        // NBUF = 2: the other buffer is used by the next tile; a tile's buffer is reused two tiles later, after another barrier
        if (NBUF == 1) __syncthreads();
    }
}

// Deterministic sum of `count` doubles by one workgroup: fixed strided partial sums, fixed tree.
__global__ __launch_bounds__(1024) void sum_partials_kernel(const double* __restrict__ in, long long count,
                                                            double* __restrict__ out, int n_out, long long stride) {
    __shared__ double sh[1024];
    for (int o = 0; o < n_out; ++o) {
        double acc = 0.0;
        for (long long i = threadIdx.x; i < count; i += 1024) acc += in[i * stride + o];
        sh[threadIdx.x] = acc;
        __syncthreads();
        for (int s = 512; s > 0; s >>= 1) {
            if ((int)threadIdx.x < s) sh[threadIdx.x] += sh[threadIdx.x + s];
            __syncthreads();
        }
        if (threadIdx.x == 0) out[o] = sh[0];
        __syncthreads();
    }
}

// Stage 1 of the two-level sum for long inputs: block b reduces a contiguous range to mid[b*n_out + o].
__global__ __launch_bounds__(256) void sum_ranges_kernel(const double* __restrict__ in, long long count, int n_out,
                                                         long long stride, long long per_block, double* __restrict__ mid) {
    __shared__ double sh[256];
    const long long i0 = (long long)blockIdx.x * per_block, i1 = min(count, i0 + per_block);
    for (int o = 0; o < n_out; ++o) {
        double acc = 0.0;
        for (long long i = i0 + threadIdx.x; i < i1; i += 256) acc += in[i * stride + o];
        sh[threadIdx.x] = acc;
        __syncthreads();
        for (int s = 128; s > 0; s >>= 1) {
            if ((int)threadIdx.x < s) sh[threadIdx.x] += sh[threadIdx.x + s];
            __syncthreads();
        }
        if (threadIdx.x == 0) mid[(size_t)blockIdx.x * n_out + o] = sh[0];
        __syncthreads();
    }
}

int launch_sum_partials(const double* in, long long count, double* out, int n_out, long long stride, hipStream_t st) {
    if (count > 65536) {   // two fixed-shape levels: deterministic, and the long level uses the whole chip
        static thread_local DevBuf mid;
        const int nb = 512;
        const long long per_block = (count + nb - 1) / nb;
        if (mid.bytes < (size_t)nb * n_out * sizeof(double)) FDX_TRY(mid.alloc((size_t)nb * n_out * sizeof(double)));
        hipLaunchKernelGGL(sum_ranges_kernel, dim3(nb), dim3(256), 0, st, in, count, n_out, stride, per_block, mid.as<double>());
        FDX_CHECK_LAUNCH();
        hipLaunchKernelGGL(sum_partials_kernel, dim3(1), dim3(1024), 0, st, mid.as<double>(), (long long)nb, out, n_out, (long long)n_out);
        FDX_CHECK_LAUNCH();
        return 0;
    }
    hipLaunchKernelGGL(sum_partials_kernel, dim3(1), dim3(1024), 0, st, in, count, out, n_out, stride);
    FDX_CHECK_LAUNCH();
    return 0;
}

long long xyt_partials_count(long long n) { return (n + 15) / 16; }

int launch_xyt(const double* Xs, const double* Ys, long long ldy, long long n, int d, int K, double* Hout,
               long long ldh, double* sumsq_partials, hipStream_t st) {
    if (n <= 0 || K <= 0) return 0;
    if (n > 0x7fffff00LL) return fail(FDX_ERR_UNSUPPORTED, "launch_xyt: n too large");
    const long long waves = (n + 15) / 16;
    const int blocks = (int)((waves + 3) / 4);
    {   // large-n fast path: X_sketch register-resident, contraction split over 8 waves
        const bool ok = !sumsq_partials && (d % 16 == 0) && d <= 1024 && K <= 64 && (ldy % 4 == 0) &&
                        ((reinterpret_cast<uintptr_t>(Xs) & 31) == 0) && ((reinterpret_cast<uintptr_t>(Ys) & 31) == 0) &&
                        !fdx::exp_env("FDX_XYT_PLAIN");
        if (ok) {
            const int nb = (d + 127) / 128;             // 16-wide blocks per wave
            const int T = (K + 15) / 16;
            const int grid = (int)std::min<long long>(waves, 256LL * 2);
#define FDX_XYT_SPLIT(NB_, T_, XS_, H_, K_) hipLaunchKernelGGL((xyt_split_kernel<NB_, T_>), dim3(grid), dim3(512), 0, st, XS_, Ys, ldy, (int)n, d, K_, H_, ldh)
            if (T <= 2) {
                if (nb <= 1) { if (T == 1) FDX_XYT_SPLIT(1, 1, Xs, Hout, K); else FDX_XYT_SPLIT(1, 2, Xs, Hout, K); }
                else if (nb <= 2) { if (T == 1) FDX_XYT_SPLIT(2, 1, Xs, Hout, K); else FDX_XYT_SPLIT(2, 2, Xs, Hout, K); }
                else if (nb <= 4) { if (T == 1) FDX_XYT_SPLIT(4, 1, Xs, Hout, K); else FDX_XYT_SPLIT(4, 2, Xs, Hout, K); }
                else { if (T == 1) FDX_XYT_SPLIT(8, 1, Xs, Hout, K); else FDX_XYT_SPLIT(8, 2, Xs, Hout, K); }
            } else if (nb <= 4) {                       // 33..64 types, d <= 512: four type tiles in one pass
                if (nb <= 1) FDX_XYT_SPLIT(1, 4, Xs, Hout, K);
                else if (nb <= 2) FDX_XYT_SPLIT(2, 4, Xs, Hout, K);
                else FDX_XYT_SPLIT(4, 4, Xs, Hout, K);
            } else {                                    // 33..64 types, d > 512: A operands of four tiles do not fit in
                const int K2 = K - 32;                  // registers - two passes over Y_sketch, 32 types each
                FDX_XYT_SPLIT(8, 2, Xs, Hout, 32);
                FDX_CHECK_LAUNCH();
                if (K2 <= 16) FDX_XYT_SPLIT(8, 1, Xs + (size_t)32 * d, Hout + (size_t)32 * ldh, K2);
                else FDX_XYT_SPLIT(8, 2, Xs + (size_t)32 * d, Hout + (size_t)32 * ldh, K2);
            }
#undef FDX_XYT_SPLIT
            FDX_CHECK_LAUNCH();
            return 0;
        }
    }
    const bool aligned = (d % 16 == 0) && (ldy % 4 == 0) && ((reinterpret_cast<uintptr_t>(Xs) & 31) == 0) &&
                         ((reinterpret_cast<uintptr_t>(Ys) & 31) == 0);
    if (aligned)
        hipLaunchKernelGGL(xyt_kernel<true>, dim3(blocks), dim3(256), 0, st, Xs, Ys, ldy, (int)n, d, K, Hout, ldh, sumsq_partials);
    else
        hipLaunchKernelGGL(xyt_kernel<false>, dim3(blocks), dim3(256), 0, st, Xs, Ys, ldy, (int)n, d, K, Hout, ldh, sumsq_partials);
    FDX_CHECK_LAUNCH();
    return 0;
}

}  // namespace fdx
