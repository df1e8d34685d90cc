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
#include "fdx_env.h"
#include <algorithm>
#include <cstdlib>

#include "device_math.h"
#include "fdx_internal.h"
#include "fdx_kernels.h"

namespace fdx {

template <typename T> struct Vec4;
template <> struct Vec4<float> { typedef float type __attribute__((ext_vector_type(4))); };
template <> struct Vec4<double> { typedef double type __attribute__((ext_vector_type(2))); };  // 16 bytes

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
            // batches of 8 independent 16-byte loads per lane (8 KB per wave in flight) before the LDS writes: the row
            // comes straight from HBM, so a load-wait-store loop would pay the full memory latency once per 1 KB
            for (int v0 = 0; v0 < nvec; v0 += 512) {
                V x[8];
#pragma unroll
                for (int u = 0; u < 8; ++u) {
                    const int v = v0 + u * 64 + lane;
                    if (v < nvec) x[u] = src[v];
                }
#pragma unroll
                for (int u = 0; u < 8; ++u) {
                    const int v = v0 + u * 64 + lane;
                    if (v < nvec) {
                        dst[v] = x[u];
                        if (MODE != FDX_PRE_RAW) {
#pragma unroll
                            for (int e = 0; e < PER; ++e) part += (double)x[u][e];
                        }
                    }
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
                if (MODE != FDX_PRE_RAW) y = fast_log1p(y * scale);
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

// Register-resident schedule variant: when the whole schedule is at most NR rounds long (G/d up to ~7 genes per
// bucket) every lane keeps its NR (gene index, weight) pairs in VGPRs for the life of the wave, so a row costs one
// LDS gather + convert + fma per round and NO schedule traffic (the generic kernel above re-reads 12 bytes of
// schedule per round per lane from L2, ~3x the bytes of the row itself).  Same arithmetic, same summation order.
template <typename T, int MODE, bool VEC, int NR>
__global__ __launch_bounds__(256, 2) void sketch_rows_reg_kernel(const T* __restrict__ Y, long long ldy,
                                                              const int* __restrict__ row_map, long long n, int G, int d,
                                                              const unsigned int* __restrict__ sched_pack,
                                                              const double* __restrict__ sched_w, int total_len,
                                                              unsigned long long end_mask,
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

    // schedule -> registers (entry = gene | bucket_code << 20; see sketch_plan.cpp)
    unsigned int gi[NR];
    double wt[NR];
#pragma unroll
    for (int e = 0; e < NR; ++e) {
        gi[e] = (e < total_len) ? sched_pack[(size_t)e * 64 + lane] : 0xFFF00000u;
        wt[e] = (e < total_len) ? sched_w[(size_t)e * 64 + lane] : 0.0;
    }
    for (int c = lane; c < d; c += 64) outbuf[c] = 0.0;   // buckets no gene hashes to stay exactly 0

    const long long wave0 = (long long)blockIdx.x * waves_per_blk + wib;
    const long long wave_stride = (long long)gridDim.x * waves_per_blk;
    for (long long p = wave0; p < n; p += wave_stride) {
        const long long src_row = row_map ? (long long)row_map[p] : p;
        const T* yrow = Y + (size_t)src_row * ldy;
        double part = 0.0;
        if (VEC) {
            typedef typename Vec4<T>::type V;
            constexpr int PER = 16 / sizeof(T);
            const int nvec = G / PER;
            const V* src = reinterpret_cast<const V*>(yrow);
            V* dst = reinterpret_cast<V*>(rowbuf);
            // batches of 8 independent 16-byte loads per lane (8 KB per wave in flight) before the LDS writes: the row
            // comes straight from HBM, so a load-wait-store loop would pay the full memory latency once per 1 KB
            for (int v0 = 0; v0 < nvec; v0 += 512) {
                V x[8];
#pragma unroll
                for (int u = 0; u < 8; ++u) {
                    const int v = v0 + u * 64 + lane;
                    if (v < nvec) x[u] = src[v];
                }
#pragma unroll
                for (int u = 0; u < 8; ++u) {
                    const int v = v0 + u * 64 + lane;
                    if (v < nvec) {
                        dst[v] = x[u];
                        if (MODE != FDX_PRE_RAW) {
#pragma unroll
                            for (int e = 0; e < PER; ++e) part += (double)x[u][e];
                        }
                    }
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
            scale = (1.0 / (s + 1e-10)) * 1e4;
        } else if (MODE == FDX_PRE_LOG_CPM_SPARSE) {
            double s = wave_sum(part);
            if (s == 0.0) s = 1.0;
            scale = 1e4 / s;
        }
        __builtin_amdgcn_s_waitcnt(0xc07f);
        double sq = 0.0, acc = 0.0;
#pragma unroll
        for (int e = 0; e < NR; ++e) {
            if (e < total_len) {                       // wave-uniform
                double y = (double)rowbuf[gi[e] & 0xFFFFFu];
                if (MODE != FDX_PRE_RAW) y = fast_log1p(y * scale);
                acc = fma(wt[e], y, acc);
                if ((end_mask >> e) & 1ULL) {          // a group ends here (wave-uniform, scalar test)
                    const unsigned code = gi[e] >> 20;
                    if (code != 0xFFFu) {
                        outbuf[code] = acc;
                        sq = fma(acc, acc, sq);
                    }
                    acc = 0.0;
                }
            }
        }
        __builtin_amdgcn_s_waitcnt(0xc07f);
        double* dst = Ys + (size_t)p * ldys;
        for (int c = lane; c < d; c += 64) dst[c] = outbuf[c];
        if (row_sumsq) {
            sq = wave_sum(sq);
            if (lane == 0) row_sumsq[p] = sq;
        }
        __builtin_amdgcn_s_waitcnt(0xc07f);
    }
}

// Scatter kernel - the default for a CountSketch (exactly one bucket per gene).  Nothing of the ROW is staged: lanes stream
// it 16 bytes at a time straight into registers, look up each gene's {weight, bucket} in a per-gene table held in LDS (one
// copy per workgroup, 10 bytes per gene) and add weight * f(y) into the wave's d-entry accumulator with ds_add_f64.
// LDS per workgroup is G*10 + waves*d*8 bytes instead of waves*(G*sizeof(T) + d*8), so 16-24 waves per CU stay resident
// for any G (the gather kernels hold 8, and drop to 4 above ~3000 genes: 0.8 TB/s at 5000 genes -> 1024), and the per-row
// work is one LDS read pair and one LDS atomic per gene instead of a row write, a schedule walk and a gather.
// Measured on MI355X, 1M x 2000 f32 -> 512: 2.50 ms (8 GB read + 4.3 GB written = 4.9 TB/s at the memory system) against
// 3.38 ms for the register-schedule gather kernel; log-CPM 6.6 ms against 8.8 ms.
// The order in which the entries of one bucket are added is the hardware's lane order, not the gene order: a last-bit
// difference (the reference's own scipy product fixes no order either), far inside the 1e-4 parity budget, and
// repeatable run to run (asserted in tests).  log-CPM reads the row twice (library size first); the second read hits
// L2/MALL.  FDX_SKETCH_GATHER=1 selects the atomics-free, gene-ordered gather kernels instead.
template <typename T, int MODE, bool VEC>
__global__ __launch_bounds__(512, 4) void sketch_rows_scatter_kernel(const T* __restrict__ Y, long long ldy,
                                                                  const int* __restrict__ row_map, long long n, int G, int d,
                                                                  const double* __restrict__ gene_w,
                                                                  const int* __restrict__ gene_bucket,
                                                                  double* __restrict__ Ys, long long ldys,
                                                                  double* __restrict__ row_sumsq, int no_table) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int lane = threadIdx.x & 63;
    const int wib = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int waves_per_blk = blockDim.x >> 6;
    const int Gp = (G + 256 + 7) & ~7;                 // table capacity: lane-major layout pads up to 64 * PER slots
    double* w_l = reinterpret_cast<double*>(smem);                                   // [Gp]
    double* acc = w_l + Gp + (size_t)wib * (d + 64);                                  // [waves][d + 64]
    double* tab = acc + d;                                                            // this wave's log1p table (64)
    unsigned short* b_l = reinterpret_cast<unsigned short*>(w_l + Gp + (size_t)waves_per_blk * (d + 64));   // [Gp]
    // LANE-MAJOR table: the entry of gene (v0 + u*64 + lane)*PER + e sits at ((v0/64 + u)*PER + e)*64 + lane, so a wave's
    // table read touches 64 consecutive entries (in gene order a lane's genes are PER*8 bytes apart: 8-way bank conflicts);
    // genes past the last full 16-byte vector (all genes on the scalar path) follow in gene order.
    typedef typename Vec4<T>::type V;
    constexpr int PER = 16 / sizeof(T);
    const int nvec = VEC ? G / PER : 0;
    const int tail_base = ((nvec + 63) >> 6) * 64 * PER;
    for (int g = threadIdx.x; g < G; g += blockDim.x) {
        const int b = gene_bucket[g];
        const int v = g / PER, e = g - v * PER;
        const int idx = (g < nvec * PER) ? ((((v >> 6) * PER + e) << 6) + (v & 63)) : (tail_base + (g - nvec * PER));
        w_l[idx] = (b >= 0) ? gene_w[g] : 0.0;
        b_l[idx] = (unsigned short)(b >= 0 ? b : 0xFFFF);     // 0xFFFF: gene has no entry in Omega (never added)
    }
    __syncthreads();
    const long long wave0 = (long long)blockIdx.x * waves_per_blk + wib;
    const long long stride = (long long)gridDim.x * waves_per_blk;
    for (long long p = wave0; p < n; p += stride) {
        const long long row = row_map ? (long long)row_map[p] : p;
        const T* yrow = Y + (size_t)row * ldy;
        const V* src = reinterpret_cast<const V*>(yrow);
        for (int c = lane; c < d; c += 64) acc[c] = 0.0;
        double scale = 1.0;
        bool use_tab = false;
        if (MODE != FDX_PRE_RAW) {
            double part = 0.0;
            T mx = (T)0;                                       // row maximum in the input type (one cheap max per entry)
            for (int v0 = 0; v0 < nvec; v0 += 512) {
                V x[8];
#pragma unroll
                for (int u = 0; u < 8; ++u) {
                    const int v = v0 + u * 64 + lane;
                    if (v < nvec) x[u] = src[v];
                }
#pragma unroll
                for (int u = 0; u < 8; ++u) {
                    const int v = v0 + u * 64 + lane;
                    if (v < nvec) {
#pragma unroll
                        for (int e = 0; e < PER; ++e) { part += (double)x[u][e]; mx = x[u][e] > mx ? x[u][e] : mx; }
                    }
                }
            }
            for (int g = nvec * PER + lane; g < G; g += 64) { part += (double)yrow[g]; mx = yrow[g] > mx ? yrow[g] : mx; }
            double sum = wave_sum(part);
            use_tab = !no_table && wave_max((double)mx) < 64.0;        // a row of small counts: log1p by table (device_math.h)
            if (MODE == FDX_PRE_LOG_CPM) {
                scale = (1.0 / (sum + 1e-10)) * 1e4;           // y / (rowsum + 1e-10) * 1e4   (deconv.py:190)
            } else {
                if (sum == 0.0) sum = 1.0;                     // lib_size[lib_size == 0] = 1  (deconv.py:183-185)
                scale = 1e4 / sum;
            }
            if (use_tab) log1p_table_fill(tab, scale, lane);
        }
        __builtin_amdgcn_s_waitcnt(0xc07f);                    // zeroing (and the table) done before the adds
        for (int v0 = 0; v0 < nvec; v0 += 256) {               // 4 x 16-byte loads per lane in flight
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
                        const int ti = ((((v0 >> 6) + u) * PER + e) << 6) + lane;   // lane-major index of gene v*PER+e
                        double y = (double)x[u][e];
                        if (MODE != FDX_PRE_RAW) y = log1p_scaled(y, scale, tab, use_tab);
                        const unsigned b = b_l[ti];
                        if (b != 0xFFFFu) __hip_atomic_fetch_add(acc + b, w_l[ti] * y, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                    }
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
        double* dst = Ys + (size_t)p * ldys;
        double sq = 0.0;
        for (int c = lane; c < d; c += 64) {
            const double v = acc[c];
            __builtin_nontemporal_store(v, &dst[c]);           // written once, read once by the contraction: do not keep in L2
            sq = fma(v, v, sq);
        }
        if (row_sumsq) {
            sq = wave_sum(sq);
            if (lane == 0) row_sumsq[p] = sq;
        }
        __builtin_amdgcn_s_waitcnt(0xc07f);                    // reads of acc done before the next row zeroes it
    }
}

template <typename T, int MODE, bool VEC>
static const void* pick_reg_kernel(int total_len) {
    if (total_len <= 16) return (const void*)sketch_rows_reg_kernel<T, MODE, VEC, 16>;
    if (total_len <= 24) return (const void*)sketch_rows_reg_kernel<T, MODE, VEC, 24>;
    if (total_len <= 32) return (const void*)sketch_rows_reg_kernel<T, MODE, VEC, 32>;
    if (total_len <= 40) return (const void*)sketch_rows_reg_kernel<T, MODE, VEC, 40>;
    if (total_len <= 48) return (const void*)sketch_rows_reg_kernel<T, MODE, VEC, 48>;
    return nullptr;
}

template <typename T, int MODE>
static int launch_sketch_mode(const T* Y, long long ldy, const int* row_map, long long n, int G, int d,
                              const SketchPlanDev& plan, double* Ys, long long ldys, double* row_sumsq, hipStream_t st) {
    const size_t row_bytes = ((size_t)G * sizeof(T) + 15) & ~(size_t)15;
    const size_t per_wave = row_bytes + (size_t)d * sizeof(double);
    // A CountSketch (one entry per gene) takes the scatter kernel: measured 2.50 ms against 3.38 ms (raw) and 6.6 ms against
    // 8.8 ms (log-CPM) for 1M x 2000 -> 512 on MI355X.  The gather kernels below serve a general sparse Omega, and any Omega
    // under FDX_SKETCH_GATHER=1 (gene-ordered, atomics-free sums).
    const bool use_scatter = plan.scatter_ok && sketch_scatter_fits(G, d) && !fdx::env("FDX_SKETCH_GATHER") &&
                             !fdx::exp_env("FDX_SKETCH_NO_SCATTER");
    if (!use_scatter)
    {   // register-resident schedule when it is short enough and at least one gene is hashed to the first group
        const bool vec0 = (ldy % (16 / (long long)sizeof(T)) == 0) && ((reinterpret_cast<uintptr_t>(Y) & 15) == 0);
        const void* rk = (plan.pack_ok && !fdx::exp_env("FDX_SKETCH_NO_REG"))
                             ? (vec0 ? pick_reg_kernel<T, MODE, true>(plan.total_len) : pick_reg_kernel<T, MODE, false>(plan.total_len))
                             : nullptr;
        if (rk && per_wave * 4 <= 64 * 1024) {
            const size_t lds_r = per_wave * 4;
            const int blocks_r = (int)std::min<long long>((n + 3) / 4, 256LL * 8);
            void* args[] = {(void*)&Y, (void*)&ldy, (void*)&row_map, (void*)&n, (void*)&G, (void*)&d,
                            (void*)&plan.sched_pack, (void*)&plan.sched_w, (void*)&plan.total_len, (void*)&plan.end_mask,
                            (void*)&Ys, (void*)&ldys, (void*)&row_sumsq};
            FDX_HIP(hipLaunchKernel(rk, dim3(blocks_r), dim3(256), args, lds_r, st));
            return 0;
        }
    }
    if (use_scatter) {
        const size_t Gp = ((size_t)G + 256 + 7) & ~(size_t)7;
        int wv = 8;
        while (wv > 1 && Gp * 10 + (size_t)wv * (d + 64) * 8 > 150 * 1024) wv >>= 1;
        const size_t lds_s = Gp * 10 + (size_t)wv * (d + 64) * 8;
        if (lds_s <= 150 * 1024) {
            const bool vec_s = (ldy % (16 / (long long)sizeof(T)) == 0) && ((reinterpret_cast<uintptr_t>(Y) & 15) == 0);
            const int per_cu = std::max<int>(1, (int)((160 * 1024) / lds_s));
            const int blocks_s = (int)std::min<long long>((n + wv - 1) / wv, 256LL * per_cu);
            auto ks = vec_s ? sketch_rows_scatter_kernel<T, MODE, true> : sketch_rows_scatter_kernel<T, MODE, false>;
            if (lds_s > 64 * 1024)
                FDX_HIP(hipFuncSetAttribute((const void*)ks, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_s));
            const int no_table = fdx::env("FDX_NO_LOG_TABLE") ? 1 : 0;
            hipLaunchKernelGGL(ks, dim3(blocks_s), dim3(wv * 64), lds_s, st, Y, ldy, row_map, n, G, d, plan.gene_w,
                               plan.gene_bucket, Ys, ldys, row_sumsq, no_table);
            FDX_CHECK_LAUNCH();
            return 0;
        }
    }
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

// ------------------------------------------------------------------------------------------------ gene statistics
// Highly-variable-gene moments (flashdeconv/utils/genes.py:85-102 dense, :52-83 sparse; same arithmetic):
//   z = log1p(y / max(rowsum, 1) * 1e4);  per gene  mean = sum z / N,  var = (sum z^2 - (sum z)^2 / N) / (N - 1).
// Two passes over Y: row scales (one wave per row), then per-gene sums of z and z^2 (lane = gene, coalesced; a block sums
// a contiguous stripe of rows, stripes are folded in order -> deterministic).
namespace fdx {

template <typename T>
__global__ __launch_bounds__(256) void row_scale_kernel(const T* __restrict__ Y, long long ldy, long long n, int G,
                                                        double* __restrict__ scale) {
    const int lane = threadIdx.x & 63;
    const long long row = (blockIdx.x * (long long)blockDim.x + threadIdx.x) >> 6;
    if (row >= n) return;
    const T* y = Y + (size_t)row * ldy;
    double s = 0.0;
    for (int g0 = 0; g0 < G; g0 += 512) {
        double x[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) { const int g = g0 + u * 64 + lane; x[u] = (g < G) ? (double)y[g] : 0.0; }
#pragma unroll
        for (int u = 0; u < 8; ++u) s += x[u];
    }
    s = wave_sum(s);
    if (lane == 0) scale[row] = 10000.0 / fmax(s, 1.0);          // genes.py:90-92 / :57-59
}

template <typename T>
__global__ __launch_bounds__(256) void gene_moment_partials_kernel(const T* __restrict__ Y, long long ldy, long long n, int G,
                                                                   int rows_per_block, const double* __restrict__ scale,
                                                                   double* __restrict__ part /* (parts, 2, G) */) {
    const long long r0 = (long long)blockIdx.y * rows_per_block;
    const long long r1 = min(n, r0 + rows_per_block);
    const int g = blockIdx.x * 256 + threadIdx.x;
    if (g >= G) return;
    double s1 = 0.0, s2 = 0.0;
    for (long long r = r0; r < r1; ++r) {
        const double z = fast_log1p((double)Y[(size_t)r * ldy + g] * scale[r]);
        s1 += z;
        s2 = fma(z, z, s2);
    }
    part[((size_t)blockIdx.y * 2 + 0) * G + g] = s1;
    part[((size_t)blockIdx.y * 2 + 1) * G + g] = s2;
}

__global__ __launch_bounds__(256) void fold_moments_kernel(const double* __restrict__ part, int n_parts, int G, long long n,
                                                           double* __restrict__ mean, double* __restrict__ var) {
    const int g = blockIdx.x * 256 + threadIdx.x;
    if (g >= G) return;
    double s1 = 0.0, s2 = 0.0;
    for (int b = 0; b < n_parts; ++b) { s1 += part[((size_t)b * 2) * G + g]; s2 += part[((size_t)b * 2 + 1) * G + g]; }
    const double m = s1 / (double)n;
    mean[g] = m;
    var[g] = (n >= 2) ? fmax(((s2 / (double)n) - m * m) * ((double)n / (double)(n - 1)), 0.0) : 0.0;   // genes.py:74-83
}

// out[r, j] = Y[r, idx[j]]  (core/deconv.py:321 Y[:, gene_idx]); one wave per row, row staged through LDS
template <typename T>
__global__ __launch_bounds__(256) void gather_columns_kernel(const T* __restrict__ Y, long long ldy, long long n, int G,
                                                             const int* __restrict__ idx, int Gs, T* __restrict__ out) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int lane = threadIdx.x & 63;
    const int wib = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    T* rowbuf = reinterpret_cast<T*>(smem) + (size_t)wib * G;
    const long long wave0 = (long long)blockIdx.x * (blockDim.x >> 6) + wib;
    const long long stride = (long long)gridDim.x * (blockDim.x >> 6);
    for (long long r = wave0; r < n; r += stride) {
        const T* y = Y + (size_t)r * ldy;
        for (int g0 = 0; g0 < G; g0 += 512) {
            T x[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) { const int g = g0 + u * 64 + lane; if (g < G) x[u] = y[g]; }
#pragma unroll
            for (int u = 0; u < 8; ++u) { const int g = g0 + u * 64 + lane; if (g < G) rowbuf[g] = x[u]; }
        }
        __builtin_amdgcn_s_waitcnt(0xc07f);
        for (int j = lane; j < Gs; j += 64) out[(size_t)r * Gs + j] = rowbuf[idx[j]];
        __builtin_amdgcn_s_waitcnt(0xc07f);
    }
}

// The same gather for rows too long to stage (more than 160 KB: a dense float64 whole transcriptome): one wave per row reads
// Y[r, idx[j]] directly.  idx is ascending, so consecutive lanes touch the same or neighbouring cache lines.
template <typename T>
__global__ __launch_bounds__(256) void gather_columns_direct_kernel(const T* __restrict__ Y, long long ldy, long long n,
                                                                    const int* __restrict__ idx, int Gs, T* __restrict__ out) {
    const int lane = threadIdx.x & 63;
    const long long wave0 = (long long)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    const long long stride = (long long)gridDim.x * (blockDim.x >> 6);
    for (long long r = wave0; r < n; r += stride) {
        const T* y = Y + (size_t)r * ldy;
        for (int j = lane; j < Gs; j += 64) out[(size_t)r * Gs + j] = y[idx[j]];
    }
}

int launch_gene_moments(const void* Y, int dtype, long long ldy, long long n, int G, double* scale, double* partials,
                        double* mean, double* var, hipStream_t st) {
    if (G <= 0 || n <= 0) return fail(FDX_ERR_INVALID, "gene moments: empty matrix");
    const int parts = column_sums_parts(n);
    const int rows_per_block = (int)((n + parts - 1) / parts);
    const int rb = (int)((n * 64 + 255) / 256);
    dim3 grid(ceil_div(G, 256), parts);
    if (dtype == FDX_F32) {
        hipLaunchKernelGGL(row_scale_kernel<float>, dim3(rb), dim3(256), 0, st, (const float*)Y, ldy, n, G, scale);
        hipLaunchKernelGGL(gene_moment_partials_kernel<float>, grid, dim3(256), 0, st, (const float*)Y, ldy, n, G, rows_per_block, scale, partials);
    } else if (dtype == FDX_F64) {
        hipLaunchKernelGGL(row_scale_kernel<double>, dim3(rb), dim3(256), 0, st, (const double*)Y, ldy, n, G, scale);
        hipLaunchKernelGGL(gene_moment_partials_kernel<double>, grid, dim3(256), 0, st, (const double*)Y, ldy, n, G, rows_per_block, scale, partials);
    } else {
        return fail(FDX_ERR_INVALID, "gene moments: dtype must be FDX_F32 or FDX_F64");
    }
    FDX_CHECK_LAUNCH();
    hipLaunchKernelGGL(fold_moments_kernel, dim3(ceil_div(G, 256)), dim3(256), 0, st, partials, parts, G, n, mean, var);
    FDX_CHECK_LAUNCH();
    return 0;
}

int launch_gather_columns(const void* Y, int dtype, long long ldy, long long n, int G, const int* idx, int Gs, void* out,
                          hipStream_t st) {
    if (n <= 0 || Gs <= 0) return 0;
    const size_t esz = dtype == FDX_F32 ? 4 : 8;
    int waves = 4;
    while (waves > 1 && (size_t)G * esz * waves > 64 * 1024) waves >>= 1;
    const size_t lds = (size_t)G * esz * waves;
    if (lds > 160 * 1024) {   // one row does not fit in LDS: direct gather
        const int blocks = (int)std::min<long long>((n + 3) / 4, 256LL * 8);
        if (dtype == FDX_F32)
            hipLaunchKernelGGL(gather_columns_direct_kernel<float>, dim3(blocks), dim3(256), 0, st, (const float*)Y, ldy, n, idx, Gs, (float*)out);
        else if (dtype == FDX_F64)
            hipLaunchKernelGGL(gather_columns_direct_kernel<double>, dim3(blocks), dim3(256), 0, st, (const double*)Y, ldy, n, idx, Gs, (double*)out);
        else
            return fail(FDX_ERR_INVALID, "gather columns: dtype must be FDX_F32 or FDX_F64");
        FDX_CHECK_LAUNCH();
        return 0;
    }
    const int blocks = (int)std::min<long long>((n + waves - 1) / waves, 256LL * 8);
    if (dtype == FDX_F32) {
        if (lds > 64 * 1024) FDX_HIP(hipFuncSetAttribute((const void*)gather_columns_kernel<float>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        hipLaunchKernelGGL(gather_columns_kernel<float>, dim3(blocks), dim3(waves * 64), lds, st, (const float*)Y, ldy, n, G, idx, Gs, (float*)out);
    } else if (dtype == FDX_F64) {
        if (lds > 64 * 1024) FDX_HIP(hipFuncSetAttribute((const void*)gather_columns_kernel<double>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        hipLaunchKernelGGL(gather_columns_kernel<double>, dim3(blocks), dim3(waves * 64), lds, st, (const double*)Y, ldy, n, G, idx, Gs, (double*)out);
    } else {
        return fail(FDX_ERR_INVALID, "gather columns: dtype must be FDX_F32 or FDX_F64");
    }
    FDX_CHECK_LAUNCH();
    return 0;
}

}  // namespace fdx
