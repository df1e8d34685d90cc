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
// Kernels of this file:
//   sketch_csr_contract_kernel   DEFAULT sketch -> H for CSR rows (d <= 1024, K <= 64): a 16-wave workgroup takes 16 consecutive
//                                spots; wave w walks the row of spot w ONCE (non-temporal stream; a G_all-bit bitmap in LDS says
//                                which columns are selected; the library-size pass compacts the selected entries into a per-wave
//                                LDS buffer, the sketch pass reads them from there), gathers {weight, bucket} for selected entries
//                                only (16-byte slots of an L2-resident table) and adds weight * f(y) into the spot's accumulator
//                                row in LDS (ds_add_f64); the 16 x d block is then the B operand of v_mfma_f64_16x16x4_f64 against
//                                register-resident X_sketch slices - H is stored, Y_sketch never exists.
//   sketch_csr_kernel            the same walk, one wave = one row, writing Y_sketch (d * 8 bytes per row) for the shapes the fused
//                                kernel does not take; the contraction is then xyt_split_kernel's.
//   csr_row_scale_kernel, csr_moments_cursor_kernel, csr_fold_moments_kernel
//                                gene statistics of utils/genes.py:52-83: per-row 1e4 / library size, then per-gene sums of
//                                z = log1p(scaled) and z^2 in LDS tiles of 4096 genes (a row stripe per workgroup, a cursor per
//                                row through its sorted columns, 256-entry steps shrinking to 64 near a tile's end), folded in
//                                stripe order.  csr_moments_tiled_kernel: rows whose columns are not sorted.
// Several entries of one row can hit the same bucket, so the order of those additions is the hardware's; the reference's scipy
// product carries the same freedom, and the 1e-4 parity budget is 12 orders of magnitude above it.
#include <algorithm>
#include <cstdlib>

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
//
// A row is a chain of dependent hops (indptr -> indices -> table -> LDS), so the loop is software-pipelined: entries are
// taken 4 x 64 at a time, and the loads of the NEXT group (and the extents of the NEXT row) are issued after the table
// gathers of the current one - vector memory returns in order, so that placement lets the wave wait for its gathers
// while the next group's loads stay in flight.
template <typename T>
struct CsrGroup {
    int c[4];
    T y[4];
};

// NT: the row is read ONCE (the fused kernel's library-size pass keeps what it needs in LDS): non-temporal loads, so that the
// 11.5 GB stream does not push the 320 KB {weight, bucket} table out of L2 - its gathers (one 16-byte slot per selected entry,
// 3.8 GB of requests per fit) then stay L2 hits instead of going out to the fabric
template <typename T, bool NT = false>
__device__ __forceinline__ void csr_load_group(CsrGroup<T>& g, const int* __restrict__ indices, const T* __restrict__ data,
                                               long long q0, long long end, int lane) {
#pragma unroll
    for (int u = 0; u < 4; ++u) {
        const long long q = q0 + u * 64 + lane;
        const bool ok = q < end;
        if (NT) {
            g.c[u] = ok ? __builtin_nontemporal_load(indices + q) : -1;
            g.y[u] = ok ? __builtin_nontemporal_load(data + q) : (T)0;
        } else {
            g.c[u] = ok ? indices[q] : -1;       // no non-temporal hint: log-CPM reads the row a second time from L2/MALL
            g.y[u] = ok ? data[q] : (T)0;
        }
    }
}

template <typename T, int MODE>
__global__ __launch_bounds__(256) void sketch_csr_kernel(const long long* __restrict__ indptr, const int* __restrict__ indices,
                                                         const T* __restrict__ data, const int* __restrict__ row_map,
                                                         long long row0, long long n, int d,
                                                         const GeneSlot* __restrict__ table,
                                                         const unsigned* __restrict__ sel_bits, int sel_words,
                                                         double* __restrict__ Ys, long long ldys,
                                                         double* __restrict__ row_sumsq, int no_table) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int lane = threadIdx.x & 63;
    const int wib = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int waves_per_blk = blockDim.x >> 6;
    double* acc = reinterpret_cast<double*>(smem) + (size_t)wib * (d + 64);
    double* tab = acc + d;                                             // this wave's log1p table (device_math.h)
    // "is this column selected?" as a bitmap in LDS (G_all / 8 bytes): typically one stored entry in six belongs to a
    // selected gene, and only those go on to the 16-byte table gather - a gather per stored entry made the kernel
    // L2-request-bound (1.4e9 scattered 16-byte reads at 1M spots x 1438 entries).
    unsigned* bits = reinterpret_cast<unsigned*>(smem + (size_t)waves_per_blk * (d + 64) * sizeof(double));
    for (int j = threadIdx.x; j < sel_words; j += blockDim.x) bits[j] = sel_bits[j];
    __syncthreads();
    const long long wave0 = (long long)blockIdx.x * waves_per_blk + wib;
    const long long stride = (long long)gridDim.x * waves_per_blk;
    long long beg = 0, end = 0;
    if (wave0 < n) {
        const long long row = row_map ? (long long)row_map[wave0] : row0 + wave0;
        beg = indptr[row];
        end = indptr[row + 1];
    }
    for (long long p = wave0; p < n; p += stride) {
        long long nbeg = 0, nend = 0;                                  // extents of this wave's next row, fetched early
        if (p + stride < n) {
            const long long nrow = row_map ? (long long)row_map[p + stride] : row0 + p + stride;
            nbeg = indptr[nrow];
            nend = indptr[nrow + 1];
        }
        for (int c = lane; c < d; c += 64) acc[c] = 0.0;
        double scale = 1.0;
        bool use_tab = false;
        CsrGroup<T> cur, nxt;
        if (MODE != FDX_PRE_RAW) {      // library size over the SELECTED genes (the subset is taken first, deconv.py:321)
            double s = 0.0, mx = 0.0;
            csr_load_group(cur, indices, data, beg, end, lane);
            for (long long q0 = beg; q0 < end; q0 += 256) {
                csr_load_group(nxt, indices, data, q0 + 256, end, lane);
#pragma unroll
                for (int u = 0; u < 4; ++u)
                    if (cur.c[u] >= 0 && ((bits[cur.c[u] >> 5] >> (cur.c[u] & 31)) & 1u)) {
                        s += (double)cur.y[u];
                        mx = fmax(mx, (double)cur.y[u]);
                    }
                cur = nxt;
            }
            s = wave_sum(s);
            scale = 10000.0 / (s == 0.0 ? 1.0 : s);                    // deconv.py:183-185
            use_tab = !no_table && wave_max(mx) < 64.0;
            if (use_tab) log1p_table_fill(tab, scale, lane);
        }
        csr_load_group(cur, indices, data, beg, end, lane);
        __builtin_amdgcn_s_waitcnt(0xc07f);                            // zeroing done before the adds
        for (long long q0 = beg; q0 < end; q0 += 256) {
            GeneSlot e[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                e[u].bucket = -1;
                if (cur.c[u] >= 0 && ((bits[cur.c[u] >> 5] >> (cur.c[u] & 31)) & 1u)) e[u] = table[cur.c[u]];
            }
            csr_load_group(nxt, indices, data, q0 + 256, end, lane);
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                if (e[u].bucket >= 0) {
                    double v = (double)cur.y[u];
                    if (MODE != FDX_PRE_RAW) v = log1p_scaled(v, scale, tab, use_tab);
                    lds_add(acc + e[u].bucket, e[u].w * v);
                }
            }
            cur = nxt;
        }
        __builtin_amdgcn_s_waitcnt(0xc07f);
        double* dst = Ys + (size_t)p * ldys;
        double sq = 0.0;
        for (int c = lane; c < d; c += 64) {
            const double v = acc[c];
            __builtin_nontemporal_store(v, &dst[c]);
            sq = fma(v, v, sq);
        }
        if (row_sumsq) {
            sq = wave_sum(sq);
            if (lane == 0) row_sumsq[p] = sq;
        }
        __builtin_amdgcn_s_waitcnt(0xc07f);                            // reads of acc done before the next row zeroes it
        beg = nbeg;
        end = nend;
    }
}

template <typename T>
static int launch_sketch_csr_t(const long long* indptr, const int* indices, const T* data, const int* row_map, long long row0,
                               long long n, int d, int mode, const void* table, const unsigned* sel_bits, int sel_words,
                               double* Ys, long long ldys, double* row_sumsq, hipStream_t st) {
    int waves = 4;
    while (waves > 1 && ((size_t)d + 64) * 8 * waves > 64 * 1024) waves >>= 1;
    const size_t lds = ((size_t)d + 64) * 8 * waves + (size_t)sel_words * 4;
    if (lds > 160 * 1024) return fail(FDX_ERR_UNSUPPORTED, "sketch (CSR): sketch_dim and the gene bitmap do not fit in LDS");
    const int blocks = (int)std::min<long long>((n + waves - 1) / waves, 256LL * 16);
    auto launch = [&](auto kern) -> int {
        if (lds > 64 * 1024)
            FDX_HIP(hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        hipLaunchKernelGGL(kern, dim3(blocks), dim3(waves * 64), lds, st, indptr, indices, data, row_map, row0, n, d,
                           (const GeneSlot*)table, sel_bits, sel_words, Ys, ldys, row_sumsq, getenv("FDX_NO_LOG_TABLE") ? 1 : 0);
        FDX_CHECK_LAUNCH();
        return 0;
    };
    if (mode == FDX_PRE_RAW) return launch(sketch_csr_kernel<T, FDX_PRE_RAW>);
    if (mode == FDX_PRE_LOG_CPM_SPARSE || mode == FDX_PRE_LOG_CPM) return launch(sketch_csr_kernel<T, FDX_PRE_LOG_CPM_SPARSE>);
    return fail(FDX_ERR_INVALID, "sketch (CSR): unknown preprocess mode");
}

int launch_sketch_csr(const long long* indptr, const int* indices, const void* data, int dtype, const int* row_map,
                      long long row0, long long n, int d, int mode, const void* table, const unsigned* sel_bits,
                      int sel_words, double* Ys, long long ldys, double* row_sumsq, hipStream_t st) {
    if (n <= 0 || d <= 0) return 0;
    if (dtype == FDX_F32)
        return launch_sketch_csr_t<float>(indptr, indices, (const float*)data, row_map, row0, n, d, mode, table, sel_bits,
                                          sel_words, Ys, ldys, row_sumsq, st);
    if (dtype == FDX_F64)
        return launch_sketch_csr_t<double>(indptr, indices, (const double*)data, row_map, row0, n, d, mode, table, sel_bits,
                                           sel_words, Ys, ldys, row_sumsq, st);
    return fail(FDX_ERR_INVALID, "sketch (CSR): dtype must be FDX_F32 or FDX_F64");
}

size_t csr_gene_slot_bytes() { return sizeof(GeneSlot); }

// ------------------------------------------------------------------------------------------------ fused CSR sketch -> H
// H = X_sketch * (f(Y) Omega)^T for CSR rows without Y_sketch in HBM (core/deconv.py:181-188, core/sketching.py:194-199,
// core/solver.py:205-223 in one pass).  The two-kernel path writes every sketched row (d doubles = 4 KB at d = 512) and
// reads it back for the contraction - 8.2 GB beside 11.5 GB of input at 1M spots x 1438 stored entries.  Here a 16-wave
// workgroup takes GROUPS of 16 consecutive spots (solver order):
//   gather     wave w walks the CSR row of spot w exactly as sketch_csr_kernel does (bitmap filter, {weight, bucket} gather,
//              ds_add_f64 into the spot's d-entry accumulator row in LDS); ||row||^2 goes to row_sumsq;
//   contract   the 16 x d block in LDS is the B operand of v_mfma_f64_16x16x4_f64, the contraction index split over the
//              16 waves (X_sketch slices as register-resident A operands), partial 16 x 16 type tiles added in wave order
//              through LDS and stored to H - the contract phase of sketch_contract_kernel (fused_kernels.cpp).
// The rows of a group have different lengths and meet at a barrier: the group costs its longest row.
typedef double csr_double4_t __attribute__((ext_vector_type(4)));
constexpr int CSRF_PAD = 16;     // doubles of padding per accumulator row (conflict-free B-operand reads)

template <typename T, int MODE, int NB, int TT>
__global__ __launch_bounds__(1024, 4) void sketch_csr_contract_kernel(
    const long long* __restrict__ indptr, const int* __restrict__ indices, const T* __restrict__ data,
    const int* __restrict__ row_map, long long n, int d, const GeneSlot* __restrict__ table,
    const unsigned* __restrict__ sel_bits, int sel_words, const double* __restrict__ Xs, int K, double* __restrict__ Hout,
    long long ldh, double* __restrict__ row_sumsq, int no_table, int cap) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    constexpr int R = 16;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int rs = d + CSRF_PAD;
    double* rows = reinterpret_cast<double*>(smem);                       // [16][rs]; re-used as red[16][TT*4*64]
    const int region = max(R * rs, R * TT * 4 * 64);
    double* tabs = rows + region;                                         // [16][64] per-wave log1p tables
    unsigned* bits = reinterpret_cast<unsigned*>(tabs + R * 64);          // [sel_words]
    // log modes: the SELECTED entries of the wave's row (column, value; `cap` of them) - written by the library-size pass, so
    // that the sketch pass reads them from LDS and the row is fetched from HBM once (the second read of an 11.5 KB row did not
    // hit L2 with 16 rows per CU in flight: PMC 24.3 GB for 11.75 GB of rows)
    unsigned char* keep = reinterpret_cast<unsigned char*>(bits + ((sel_words + 3) & ~3)) + (size_t)wave * cap * (4 + sizeof(T));
    int* keep_c = reinterpret_cast<int*>(keep);
    T* keep_v = reinterpret_cast<T*>(keep_c + cap);
    double* red = rows;
    for (int j = tid; j < sel_words; j += R * 64) bits[j] = sel_bits[j];
    const int r = lane & 15, q = lane >> 4;
    double a[NB][TT][4];
#pragma unroll
    for (int b = 0; b < NB; ++b) {
        const int c0 = (wave * NB + b) * 16 + 4 * q;
#pragma unroll
        for (int t = 0; t < TT; ++t) {
            const int type = t * 16 + r;
            const bool ok = type < K && c0 < d;
            const csr_double4_t v = ok ? *reinterpret_cast<const csr_double4_t*>(Xs + (size_t)type * d + c0) : csr_double4_t{0.0, 0.0, 0.0, 0.0};
            a[b][t][0] = v.x; a[b][t][1] = v.y; a[b][t][2] = v.z; a[b][t][3] = v.w;
        }
    }
    __syncthreads();
    double* acc = rows + (size_t)wave * rs;
    double* tab = tabs + (size_t)wave * 64;
    const long long n_groups = (n + R - 1) / R;
    long long grp = blockIdx.x;
    long long beg = 0, end = 0;
    if (grp < n_groups && grp * R + wave < n) {
        const long long p0 = grp * R + wave;
        const long long row = row_map ? (long long)row_map[p0] : p0;
        beg = indptr[row];
        end = indptr[row + 1];
    }
    for (; grp < n_groups; grp += gridDim.x) {
        const long long s0 = grp * R;
        const long long p = s0 + wave;
        long long nbeg = 0, nend = 0;                                     // extents of this wave's row of the next group, early
        {
            const long long pn = (grp + gridDim.x) * R + wave;
            if (grp + gridDim.x < n_groups && pn < n) {
                const long long nrow = row_map ? (long long)row_map[pn] : pn;
                nbeg = indptr[nrow];
                nend = indptr[nrow + 1];
            }
        }
        for (int c = lane; c < d; c += 64) acc[c] = 0.0;
        if (p < n) {                                                      // wave-uniform: spots past the end stay zero
            double scale = 1.0;
            bool use_tab = false;
            int kept = 0;                                                 // selected entries of the row (wave-uniform)
            bool fits = true;                                             // ... all of them are in keep_c / keep_v
            CsrGroup<T> cur, nxt;
            if (MODE != FDX_PRE_RAW) {  // library size over the SELECTED genes (the subset is taken first, deconv.py:321)
                double s = 0.0, mx = 0.0;
                const bool stream = cap > 0 && (end - beg) <= 4LL * cap;     // wave-uniform guess: the row's selected entries will fit the keep buffer - it is read once
                if (stream) csr_load_group<T, true>(cur, indices, data, beg, end, lane);
                else csr_load_group(cur, indices, data, beg, end, lane);
                for (long long q0 = beg; q0 < end; q0 += 256) {
                    if (stream) csr_load_group<T, true>(nxt, indices, data, q0 + 256, end, lane);
                    else csr_load_group(nxt, indices, data, q0 + 256, end, lane);
#pragma unroll
                    for (int u = 0; u < 4; ++u) {
                        const bool sel = cur.c[u] >= 0 && ((bits[cur.c[u] >> 5] >> (cur.c[u] & 31)) & 1u);
                        if (sel) {
                            s += (double)cur.y[u];
                            mx = fmax(mx, (double)cur.y[u]);
                        }
                        if (cap > 0) {                                     // stream compaction in CSR order: ballot + prefix count
                            const unsigned long long m = __ballot(sel);
                            const int here = __popcll(m);
                            if (kept + here <= cap) {
                                if (sel) {
                                    const int pos = kept + __popcll(m & ((1ULL << lane) - 1ULL));
                                    keep_c[pos] = cur.c[u];
                                    keep_v[pos] = cur.y[u];
                                }
                            } else {
                                fits = false;
                            }
                            kept += here;
                        }
                    }
                    cur = nxt;
                }
                s = wave_sum(s);
                scale = 10000.0 / (s == 0.0 ? 1.0 : s);                    // deconv.py:183-185
                use_tab = !no_table && wave_max(mx) < 64.0;
                if (use_tab) log1p_table_fill(tab, scale, lane);
            }
            if (MODE != FDX_PRE_RAW && cap > 0 && fits) {
                // the row's selected entries from LDS: same entries, same order as the pass over the row below
                __builtin_amdgcn_s_waitcnt(0xc07f);                        // zeroing, table and kept entries are in LDS
                for (int i0 = 0; i0 < kept; i0 += 256) {
                    GeneSlot e[4];
                    T yv[4];
#pragma unroll
                    for (int u = 0; u < 4; ++u) {
                        const int i = i0 + u * 64 + lane;
                        e[u].bucket = -1;
                        yv[u] = (T)0;
                        if (i < kept) {
                            e[u] = table[keep_c[i]];
                            yv[u] = keep_v[i];
                        }
                    }
#pragma unroll
                    for (int u = 0; u < 4; ++u) {
                        if (e[u].bucket >= 0) {
                            const double v = log1p_scaled((double)yv[u], scale, tab, use_tab);
                            lds_add(acc + e[u].bucket, e[u].w * v);
                        }
                    }
                }
            } else {
            csr_load_group(cur, indices, data, beg, end, lane);
            __builtin_amdgcn_s_waitcnt(0xc07f);                            // zeroing done before the adds
            for (long long q0 = beg; q0 < end; q0 += 256) {
                GeneSlot e[4];
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    e[u].bucket = -1;
                    if (cur.c[u] >= 0 && ((bits[cur.c[u] >> 5] >> (cur.c[u] & 31)) & 1u)) e[u] = table[cur.c[u]];
                }
                csr_load_group(nxt, indices, data, q0 + 256, end, lane);
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    if (e[u].bucket >= 0) {
                        double v = (double)cur.y[u];
                        if (MODE != FDX_PRE_RAW) v = log1p_scaled(v, scale, tab, use_tab);
                        lds_add(acc + e[u].bucket, e[u].w * v);
                    }
                }
                cur = nxt;
            }
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
        csr_double4_t accm[TT];
#pragma unroll
        for (int t = 0; t < TT; ++t) accm[t] = csr_double4_t{0.0, 0.0, 0.0, 0.0};
        const double* yrow_l = rows + (size_t)r * rs;
#pragma unroll
        for (int b = 0; b < NB; ++b) {
            const int c0 = (wave * NB + b) * 16 + 4 * q;
            const csr_double4_t bv = (c0 < d) ? *reinterpret_cast<const csr_double4_t*>(yrow_l + c0) : csr_double4_t{0.0, 0.0, 0.0, 0.0};
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
            if (type < K && sp < n) Hout[(size_t)type * ldh + sp] = sum;
        }
        __syncthreads();                                                  // red is rows again for the next group
        beg = nbeg;
        end = nend;
    }
}

static size_t csr_contract_lds(int d, int TT, int sel_words) {
    const size_t region = std::max<size_t>(16 * ((size_t)d + CSRF_PAD), (size_t)16 * TT * 4 * 64);
    return region * 8 + 16 * 64 * 8 + (((size_t)sel_words + 3) & ~(size_t)3) * 4;
}

// log modes: entries per wave of the "selected entries of the row" buffer behind the bitmap - what is left of the 160 KB, in
// whole wave steps, at most 2048 (FDX_CSR_NO_KEEP=1: none, the row is read twice)
static int csr_contract_keep(int d, int TT, int sel_words, int value_bytes) {
    if (getenv("FDX_CSR_NO_KEEP")) return 0;
    const size_t base = csr_contract_lds(d, TT, sel_words);
    if (base >= 160 * 1024) return 0;
    const size_t per_wave = (160 * 1024 - base) / 16 / (size_t)(4 + value_bytes);
    return (int)std::min<size_t>(per_wave & ~(size_t)63, 2048);
}

// shapes the fused kernel takes: the A operands of a wave (NB x TT x 4 doubles) must fit beside the gather's registers
bool csr_contract_ok(int d, int K, int sel_words) {
    if (getenv("FDX_CSR_NO_FUSED")) return false;
    if (d <= 0 || K <= 0 || K > 64 || d % 4 != 0) return false;
    const int NB = (d + 255) / 256, TT = (K + 15) / 16;
    if (NB * TT > 4) return false;
    return csr_contract_lds(d, TT, sel_words) <= 160 * 1024;
}

template <typename T, int MODE>
static int launch_csr_contract_m(const long long* indptr, const int* indices, const T* data, const int* row_map, long long n, int d,
                                 const void* table, const unsigned* sel_bits, int sel_words, const double* Xs, int K, double* H,
                                 long long ldh, double* row_sumsq, hipStream_t st) {
    const int NB = (d + 255) / 256, TT = (K + 15) / 16;
    const int cap = MODE == FDX_PRE_RAW ? 0 : csr_contract_keep(d, TT, sel_words, (int)sizeof(T));
    const size_t lds = csr_contract_lds(d, TT, sel_words) + (size_t)16 * cap * (4 + sizeof(T));
    const int grid = (int)std::min<long long>((n + 15) / 16, 256);
    const int no_table = getenv("FDX_NO_LOG_TABLE") ? 1 : 0;
    auto launch = [&](auto kern) -> int {
        if (lds > 64 * 1024)
            FDX_HIP(hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        hipLaunchKernelGGL(kern, dim3(grid), dim3(1024), lds, st, indptr, indices, data, row_map, n, d, (const GeneSlot*)table,
                           sel_bits, sel_words, Xs, K, H, ldh, row_sumsq, no_table, cap);
        FDX_CHECK_LAUNCH();
        return 0;
    };
    if (NB == 1 && TT == 1) return launch(sketch_csr_contract_kernel<T, MODE, 1, 1>);
    if (NB == 1 && TT == 2) return launch(sketch_csr_contract_kernel<T, MODE, 1, 2>);
    if (NB == 1 && TT == 3) return launch(sketch_csr_contract_kernel<T, MODE, 1, 3>);
    if (NB == 1 && TT == 4) return launch(sketch_csr_contract_kernel<T, MODE, 1, 4>);
    if (NB == 2 && TT == 1) return launch(sketch_csr_contract_kernel<T, MODE, 2, 1>);
    if (NB == 2 && TT == 2) return launch(sketch_csr_contract_kernel<T, MODE, 2, 2>);
    if (NB == 3 && TT == 1) return launch(sketch_csr_contract_kernel<T, MODE, 3, 1>);
    if (NB == 4 && TT == 1) return launch(sketch_csr_contract_kernel<T, MODE, 4, 1>);
    return fail(FDX_ERR_UNSUPPORTED, "sketch (CSR, fused): shape not instantiated");
}

// H[:, 0..n) (type-major, row stride ldh) and row_sumsq[0..n) for the n spots listed by row_map (NULL = rows 0..n-1).
// Call only when csr_contract_ok(...) holds; Xs must be 32-byte aligned.
int launch_sketch_csr_contract(const long long* indptr, const int* indices, const void* data, int dtype, const int* row_map,
                               long long n, int d, int mode, const void* table, const unsigned* sel_bits, int sel_words,
                               const double* Xs, int K, double* H, long long ldh, double* row_sumsq, hipStream_t st) {
    if (n <= 0) return 0;
    if ((reinterpret_cast<uintptr_t>(Xs) & 31) != 0) return fail(FDX_ERR_INVALID, "sketch (CSR, fused): X_sketch must be 32-byte aligned");
    const bool raw = mode == FDX_PRE_RAW;
    if (!raw && mode != FDX_PRE_LOG_CPM_SPARSE && mode != FDX_PRE_LOG_CPM) return fail(FDX_ERR_INVALID, "sketch (CSR, fused): unknown preprocess mode");
    if (dtype == FDX_F32) {
        const float* y = (const float*)data;
        return raw ? launch_csr_contract_m<float, FDX_PRE_RAW>(indptr, indices, y, row_map, n, d, table, sel_bits, sel_words, Xs, K, H, ldh, row_sumsq, st)
                   : launch_csr_contract_m<float, FDX_PRE_LOG_CPM_SPARSE>(indptr, indices, y, row_map, n, d, table, sel_bits, sel_words, Xs, K, H, ldh, row_sumsq, st);
    }
    if (dtype == FDX_F64) {
        const double* y = (const double*)data;
        return raw ? launch_csr_contract_m<double, FDX_PRE_RAW>(indptr, indices, y, row_map, n, d, table, sel_bits, sel_words, Xs, K, H, ldh, row_sumsq, st)
                   : launch_csr_contract_m<double, FDX_PRE_LOG_CPM_SPARSE>(indptr, indices, y, row_map, n, d, table, sel_bits, sel_words, Xs, K, H, ldh, row_sumsq, st);
    }
    return fail(FDX_ERR_INVALID, "sketch (CSR, fused): dtype must be FDX_F32 or FDX_F64");
}

// ------------------------------------------------------------------------------------------------ gene statistics
// z = log1p(y * 1e4 / max(lib, 1)) for every stored entry, then per gene sum z, sum z^2 (and sum y for "pearson").
// Per-gene sums over a row-major sparse matrix are a scatter; global f64 atomics on ~30k addresses run at ~30 G adds/s
// on MI355X (measured: 143 ms for 1.4e9 entries), so the scatter goes to LDS instead: a block owns a TILE of genes
// (all of its sums fit in 64 KB of LDS) and a stripe of rows, scans the stripe's column indices and adds the entries
// that fall in its tile with ds_add_f64.  Entries of one row have distinct columns, so the 64 lanes of an add never
// collide.  Every tile re-reads the stripe's indices (G/TILE passes over 4 bytes per entry) - that is the price, and
// it is ~10x cheaper than the global atomics.  Stripe partials are folded in stripe order.
// (Zeros contribute nothing to any of the sums: genes.py:52-54.)
template <typename T>
__global__ __launch_bounds__(256) void csr_row_scale_kernel(const long long* __restrict__ indptr, const T* __restrict__ data,
                                                            long long n, double* __restrict__ scale, int no_table) {
    const int lane = threadIdx.x & 63;
    const long long row = ((long long)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    if (row >= n) return;
    const long long beg = indptr[row], end = indptr[row + 1];
    double s0 = 0.0, s1 = 0.0, s2 = 0.0, s3 = 0.0, mx = 0.0;
    long long q = beg + lane;
    for (; q + 192 < end; q += 256) {
        const double a = (double)data[q], b = (double)data[q + 64], c = (double)data[q + 128], e = (double)data[q + 192];
        s0 += a; s1 += b; s2 += c; s3 += e;
        mx = fmax(fmax(mx, a), fmax(fmax(b, c), e));
    }
    for (; q < end; q += 64) { s0 += (double)data[q]; mx = fmax(mx, (double)data[q]); }
    const double s = wave_sum((s0 + s1) + (s2 + s3));
    mx = wave_max(mx);
    // genes.py:57-59; the sign carries "every entry of the row is below 64" (log1p by table in the moments kernel)
    if (lane == 0) scale[row] = ((mx < 64.0 && !no_table) ? 1.0 : -1.0) * (10000.0 / fmax(s, 1.0));
}

// 8 waves per SIMD (<= 64 VGPRs): two 16-wave workgroups per CU - at 66 VGPRs only one fits and the kernel takes 17.6 ms
// instead of 10.7 (measured)
template <typename T, int NS>
__global__ __launch_bounds__(1024, 8) void csr_moments_tiled_kernel(const long long* __restrict__ indptr, const int* __restrict__ indices,
                                                                const T* __restrict__ data, const double* __restrict__ scale,
                                                                long long n, int G, int tile, int rows_per_stripe,
                                                                double* __restrict__ part /* (stripes, NS, G) */) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    double* acc = reinterpret_cast<double*>(smem);                       // [NS][tile]
    const int lane = threadIdx.x & 63;
    const int wib = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    double* tab = acc + (size_t)NS * tile + (size_t)wib * 64;           // per-wave log1p table (device_math.h)
    const int t0 = blockIdx.y * tile;
    const int tw = min(tile, G - t0);
    for (int j = threadIdx.x; j < NS * tile; j += 1024) acc[j] = 0.0;
    __syncthreads();
    const long long r0 = (long long)blockIdx.x * rows_per_stripe;
    const long long r1 = min(n, r0 + rows_per_stripe);
    for (long long row = r0 + wib; row < r1; row += 16) {     // 16 waves share the tile: 2 blocks = 32 waves per CU
        const long long beg = indptr[row], end = indptr[row + 1];
        const double sc_signed = scale[row];
        const double sc = fabs(sc_signed);
        const bool use_tab = sc_signed > 0.0;
        if (use_tab) log1p_table_fill(tab, sc, lane);
        __builtin_amdgcn_s_waitcnt(0xc07f);
        for (long long q0 = beg; q0 < end; q0 += 256) {
            int c[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const long long q = q0 + u * 64 + lane;
                c[u] = (q < end) ? indices[q] - t0 : -1;
            }
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                if (c[u] >= 0 && c[u] < tw) {
                    const double y = (double)data[q0 + u * 64 + lane];
                    const double z = log1p_scaled(y, sc, tab, use_tab);
                    lds_add(acc + c[u], z);
                    lds_add(acc + tile + c[u], z * z);
                    if (NS == 3) lds_add(acc + 2 * tile + c[u], y);
                }
            }
        }
    }
    __syncthreads();
    for (int j = threadIdx.x; j < NS * tile; j += 1024) {
        const int s = j / tile, g = j - s * tile;
        if (g < tw) part[((size_t)blockIdx.x * NS + s) * G + t0 + g] = acc[j];
    }
}

// Sorted rows (canonical CSR): the entries of a row that fall into gene tile t+1 start where those of tile t ended, so a
// workgroup keeps its stripe of rows and walks the tiles itself, resuming every row at a saved cursor - the column
// indices are then read once instead of once per tile (5x at 20000 genes: 29 GB -> 6 GB of the kernel's traffic).
template <typename T, int NS>
__global__ __launch_bounds__(1024, 8) void csr_moments_cursor_kernel(const long long* __restrict__ indptr, const int* __restrict__ indices,
                                                                     const T* __restrict__ data, const double* __restrict__ scale,
                                                                     long long n, int G, int tile, int rows_per_stripe,
                                                                     int* __restrict__ cursor /* (n) */,
                                                                     double* __restrict__ part /* (stripes, NS, G) */) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    double* acc = reinterpret_cast<double*>(smem);                       // [NS][tile]
    const int lane = threadIdx.x & 63;
    const int wib = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    double* tab = acc + (size_t)NS * tile + (size_t)wib * 64;
    const long long r0 = (long long)blockIdx.x * rows_per_stripe;
    const long long r1 = min(n, r0 + rows_per_stripe);
    for (int t0 = 0; t0 < G; t0 += tile) {
        const int t1 = min(G, t0 + tile);
        for (int j = threadIdx.x; j < NS * tile; j += 1024) acc[j] = 0.0;
        __syncthreads();
        for (long long row = r0 + wib; row < r1; row += 16) {
            const long long beg = indptr[row], end = indptr[row + 1];
            long long q0 = beg + (t0 == 0 ? 0 : (long long)cursor[row]);
            const double sc_signed = scale[row];
            const double sc = fabs(sc_signed);
            const bool use_tab = sc_signed > 0.0;
            if (use_tab) log1p_table_fill(tab, sc, lane);
            __builtin_amdgcn_s_waitcnt(0xc07f);
            // Steps of 256 entries while most of what the tile is expected to hold of this row (columns are roughly uniform: row
            // length x tile width / G) is still ahead, then steps of 64: a step that reaches past the tile's last entry reads the
            // rest of its 256 for nothing (and again on the next visit) - with ~290 entries per visit the 256-entry steps read
            // 512, i.e. 1.8 x the bytes (PMC 22.8 GB for 11.5 GB: the kernel ran at the copy ceiling on bytes it did not need)
            const long long est = ((end - beg) * (long long)(t1 - t0)) / (long long)G;
            long long consumed = 0;
            while (q0 < end && consumed + 256 <= est - (est >> 3)) {   // wave-uniform
                // 256 entries per step, FOUR CONSECUTIVE ones per lane: one 16-byte load for the columns and one for the values
                // instead of four 4-byte loads each (the kernel waits on its vector-memory instructions - 70 M of them per
                // pass over 1.44e9 entries, texture addresser half busy, 80 % of the wave time parked; the loads only need
                // 4-byte alignment)
                typedef int int4_u __attribute__((ext_vector_type(4), aligned(4)));
                typedef T val4_u __attribute__((ext_vector_type(4), aligned(4)));
                const long long q = q0 + 4 * lane;
                int c[4];
                double y[4];
                if (q + 3 < end) {
                    // the values travel with the columns (not after them): one round trip per step; what lies past the tile is
                    // read again on the next visit (~a fifth more value bytes)
                    const int4_u v = *reinterpret_cast<const int4_u*>(indices + q);
                    const val4_u w = *reinterpret_cast<const val4_u*>(data + q);
                    c[0] = v.x; c[1] = v.y; c[2] = v.z; c[3] = v.w;
                    y[0] = (double)w.x; y[1] = (double)w.y; y[2] = (double)w.z; y[3] = (double)w.w;
                } else {
#pragma unroll
                    for (int u = 0; u < 4; ++u) {
                        c[u] = (q + u < end) ? indices[q + u] : 0x7fffffff;
                        y[u] = (q + u < end) ? (double)data[q + u] : 0.0;
                    }
                }
                bool in[4];
                int taken = 0;
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    in[u] = c[u] < t1;                                    // sorted: the taken entries are a prefix
                    taken += __popcll(__ballot(in[u]));
                }
                if (in[0]) {
#pragma unroll
                    for (int u = 0; u < 4; ++u)
                        if (in[u]) {
                            const double z = log1p_scaled(y[u], sc, tab, use_tab);
                            lds_add(acc + (c[u] - t0), z);
                            lds_add(acc + tile + (c[u] - t0), z * z);
                            if (NS == 3) lds_add(acc + 2 * tile + (c[u] - t0), y[u]);
                        }
                }
                q0 += taken;
                consumed += taken;
                if (taken < 256) { consumed = -1; break; }                // reached the next tile (or the row's end)
            }
            while (consumed >= 0 && q0 < end) {                           // one entry per lane
                const long long q = q0 + lane;
                const bool ok = q < end;
                const int c = ok ? indices[q] : 0x7fffffff;
                const double y = ok ? (double)data[q] : 0.0;
                const bool in = c < t1;
                const int taken = __popcll(__ballot(in));
                if (in) {
                    const double z = log1p_scaled(y, sc, tab, use_tab);
                    lds_add(acc + (c - t0), z);
                    lds_add(acc + tile + (c - t0), z * z);
                    if (NS == 3) lds_add(acc + 2 * tile + (c - t0), y);
                }
                q0 += taken;
                if (taken < 64) break;
            }
            if (lane == 0) cursor[row] = (int)(q0 - beg);
        }
        __syncthreads();
        for (int j = threadIdx.x; j < NS * tile; j += 1024) {
            const int s = j / tile, g = j - s * tile;
            if (t0 + g < t1) part[((size_t)blockIdx.x * NS + s) * G + t0 + g] = acc[j];
        }
        __syncthreads();
    }
}

template <int NS>
__global__ __launch_bounds__(256) void csr_fold_moments_kernel(const double* __restrict__ part, int stripes, int G, long long n,
                                                               double* __restrict__ mean, double* __restrict__ var,
                                                               double* __restrict__ colsum) {
    const int g = blockIdx.x * 256 + threadIdx.x;
    if (g >= G) return;
    double s1 = 0.0, s2 = 0.0, s0 = 0.0;
    for (int b = 0; b < stripes; ++b) {
        const double* p = part + (size_t)b * NS * G;
        s1 += p[g];
        s2 += p[(size_t)G + g];
        if (NS == 3) s0 += p[2 * (size_t)G + g];
    }
    const double m = s1 / (double)n;
    mean[g] = m;
    var[g] = (n >= 2) ? fmax(((s2 / (double)n) - m * m) * ((double)n / (double)(n - 1)), 0.0) : 0.0;   // genes.py:74-83
    if (NS == 3) colsum[g] = s0;
}

int csr_moment_stripes(long long n) { return (int)std::min<long long>(512, std::max<long long>(1, (n + 255) / 256)); }

template <typename T, int NS>
static int launch_csr_moments_t(const long long* indptr, const int* indices, const T* data, long long n, int G, double* scale,
                                double* part, double* mean, double* var, double* colsum, int* cursor, hipStream_t st) {
    const int tile = std::min(G, (int)(64 * 1024 / (NS * sizeof(double))));   // + 16 waves x 512 B of log1p tables
    const int tiles = ceil_div(G, tile);
    const int stripes = csr_moment_stripes(n);
    const int rows_per_stripe = (int)((n + stripes - 1) / stripes);
    const int no_table = getenv("FDX_NO_LOG_TABLE") ? 1 : 0;
    hipLaunchKernelGGL(csr_row_scale_kernel<T>, dim3((unsigned)((n * 64 + 255) / 256)), dim3(256), 0, st, indptr, data, n, scale, no_table);
    FDX_CHECK_LAUNCH();
    const size_t lds_m = (size_t)NS * tile * sizeof(double) + 16 * 64 * sizeof(double);
    FDX_HIP(hipFuncSetAttribute((const void*)csr_moments_tiled_kernel<T, NS>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_m));
    if (cursor && !getenv("FDX_CSR_NO_CURSOR")) {     // rows sorted by column: one pass over the indices
        FDX_HIP(hipFuncSetAttribute((const void*)csr_moments_cursor_kernel<T, NS>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_m));
        hipLaunchKernelGGL((csr_moments_cursor_kernel<T, NS>), dim3(stripes), dim3(1024), lds_m, st, indptr, indices, data, scale,
                           n, G, tile, rows_per_stripe, cursor, part);
    } else {
        hipLaunchKernelGGL((csr_moments_tiled_kernel<T, NS>), dim3(stripes, tiles), dim3(1024), lds_m, st,
                           indptr, indices, data, scale, n, G, tile, rows_per_stripe, part);
    }
    FDX_CHECK_LAUNCH();
    hipLaunchKernelGGL(csr_fold_moments_kernel<NS>, dim3(ceil_div(G, 256)), dim3(256), 0, st, part, stripes, G, n, mean, var, colsum);
    FDX_CHECK_LAUNCH();
    return 0;
}

// scale: n doubles; part: csr_moment_stripes(n) * (colsum ? 3 : 2) * G doubles; colsum may be NULL
int launch_csr_moments(const long long* indptr, const int* indices, const void* data, int dtype, long long n, int G,
                       double* scale, double* part, double* mean, double* var, double* colsum, int* cursor, hipStream_t st) {
    if (G <= 0 || n <= 0) return fail(FDX_ERR_INVALID, "gene moments (CSR): empty matrix");
    if (dtype == FDX_F32)
        return colsum ? launch_csr_moments_t<float, 3>(indptr, indices, (const float*)data, n, G, scale, part, mean, var, colsum, cursor, st)
                      : launch_csr_moments_t<float, 2>(indptr, indices, (const float*)data, n, G, scale, part, mean, var, colsum, cursor, st);
    if (dtype == FDX_F64)
        return colsum ? launch_csr_moments_t<double, 3>(indptr, indices, (const double*)data, n, G, scale, part, mean, var, colsum, cursor, st)
                      : launch_csr_moments_t<double, 2>(indptr, indices, (const double*)data, n, G, scale, part, mean, var, colsum, cursor, st);
    return fail(FDX_ERR_INVALID, "gene moments (CSR): dtype must be FDX_F32 or FDX_F64");
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

// Row-wise variant used when the caller claims sorted rows: one wave per row checks the column range (bit 0) AND that the
// columns do not descend (bit 1) in a single pass over the indices.  Rows are clamped to [0, nnz] so that a broken indptr
// (reported through bit 0 by csr_indptr_kernel) cannot send the loads out of bounds.
__global__ __launch_bounds__(256) void csr_rows_check_kernel(const long long* __restrict__ indptr, const int* __restrict__ indices,
                                                             long long n, long long nnz, int G, int* __restrict__ flag) {
    const int lane = threadIdx.x & 63;
    const long long wave0 = ((long long)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    const long long stride = ((long long)gridDim.x * blockDim.x) >> 6;
    int bad = 0;
    for (long long row = wave0; row < n; row += stride) {
        const long long beg = min(max(indptr[row], 0LL), nnz), end = min(max(indptr[row + 1], 0LL), nnz);
        for (long long q = beg + lane; q < end; q += 64) {
            const int c = indices[q];
            if (c < 0 || c >= G) bad |= 1;
            if (q + 1 < end && c > indices[q + 1]) bad |= 2;
        }
    }
    if (bad) atomicOr(flag, bad);
}

__global__ __launch_bounds__(256) void csr_indptr_kernel(const long long* __restrict__ indptr, long long n, long long nnz,
                                                         int* __restrict__ flag) {
    const long long t = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    const long long stride = (long long)gridDim.x * blockDim.x;
    int bad = 0;
    if (t == 0 && (indptr[0] != 0 || indptr[n] != nnz)) bad = 1;
    for (long long r = t; r < n; r += stride)
        if (indptr[r + 1] < indptr[r]) bad = 1;
    if (bad) atomicOr(flag, 1);
}

int launch_csr_check(const long long* indptr, const int* indices, long long n, long long nnz, int G, int check_sorted, int* flag,
                     hipStream_t st) {
    FDX_HIP(hipMemsetAsync(flag, 0, sizeof(int), st));
    if (check_sorted && n > 0) {      // one row-wise pass over the indices does range + order
        const int ib = (int)std::min<long long>((n + 255) / 256, 256LL * 8);
        hipLaunchKernelGGL(csr_indptr_kernel, dim3(ib), dim3(256), 0, st, indptr, n, nnz, flag);
        FDX_CHECK_LAUNCH();
        const int sb = (int)std::min<long long>((n + 3) / 4, 256LL * 8);
        hipLaunchKernelGGL(csr_rows_check_kernel, dim3(sb), dim3(256), 0, st, indptr, indices, n, nnz, G, flag);
        FDX_CHECK_LAUNCH();
        return 0;
    }
    const int blocks = (int)std::min<long long>(std::max<long long>(1, (std::max(n, nnz) + 255) / 256), 256LL * 8);
    hipLaunchKernelGGL(csr_check_kernel, dim3(blocks), dim3(256), 0, st, indptr, indices, n, nnz, G, flag);
    FDX_CHECK_LAUNCH();
    return 0;
}

}  // namespace fdx
