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
//                                spots; wave w holds the row of spot w in REGISTERS (fetched once, a group ahead), looks every
//                                column up in an LDS table {selected?, rank}, takes {weight, bucket} by rank from LDS and adds
//                                weight * f(y) into the spot's accumulator row in LDS (ds_add_f64); the 16 x d block is then the
//                                B operand of v_mfma_f64_16x16x4_f64 against register-resident X_sketch slices - H is stored,
//                                Y_sketch never exists.
//   sketch_csr_kernel            the same walk, one wave = one row, writing Y_sketch (d * 8 bytes per row) for the shapes the fused
//                                kernel does not take; the contraction is then xyt_split_kernel's.
//   csr_row_scale_kernel, csr_moments_cursor_kernel, csr_fold_moments_kernel
//                                gene statistics of utils/genes.py:52-83: per-row 1e4 / library size, then per-gene sums of
//                                z = log1p(scaled) and z^2 in LDS tiles of 4096 genes (a row stripe per workgroup, a cursor per
//                                row through its sorted columns, 256-entry steps shrinking to 64 near a tile's end), folded in
//                                stripe order.  csr_moments_tiled_kernel: rows whose columns are not sorted.
// Several entries of one row can hit the same bucket, so the order of those additions is the hardware's; the reference's scipy
// product carries the same freedom, and the 1e-4 parity budget is 12 orders of magnitude above it.
#include "fdx_env.h"
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
                           (const GeneSlot*)table, sel_bits, sel_words, Ys, ldys, row_sumsq, fdx::env("FDX_NO_LOG_TABLE") ? 1 : 0);
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

// ------------------------------------------------------------------------------------------------ selection tables
// Which columns of the matrix are selected genes, and their {weight, bucket}: the device tables of the two sketch kernels.
//   two-kernel path   slots (per column GeneSlot), bits (u32 bitmap)
//   fused path        words: per 32 columns {bitmap word, number of selected columns before the word} - one 8-byte LDS read answers
//                     "selected?" and gives the gene's RANK among the selected ones; w / b: weight and bucket BY RANK (10 bytes per
//                     selected gene: the whole table sits in LDS beside the accumulators, no gather leaves the CU)
int CsrSelection::build(const int32_t* gene_idx, int G, int G_all, const int32_t* bucket, const double* weight, int d, bool fused,
                        hipStream_t st, const char* who) {
    std::vector<GeneSlot> sl((size_t)G_all, GeneSlot{0.0, -1, 0});
    for (int j = 0; j < G; ++j) {
        const int c = gene_idx ? gene_idx[j] : j;
        FDX_REQUIRE(c >= 0 && c < G_all, std::string(who) + ": gene index out of range");
        FDX_REQUIRE(sl[(size_t)c].bucket < 0, std::string(who) + ": duplicate gene index");
        FDX_REQUIRE(bucket[j] >= 0 && bucket[j] < d, std::string(who) + ": bucket index out of range");
        sl[(size_t)c] = GeneSlot{weight[j], bucket[j], 0};
    }
    sel_words = (G_all + 31) / 32;
    n_sel = G;
    if (!fused) {
        std::vector<unsigned> bt((size_t)sel_words, 0u);
        for (int c = 0; c < G_all; ++c)
            if (sl[(size_t)c].bucket >= 0) bt[(size_t)c >> 5] |= 1u << (c & 31);
        FDX_TRY(slots.alloc(sl.size() * sizeof(GeneSlot)));
        FDX_TRY(bits.alloc(bt.size() * sizeof(unsigned)));
        FDX_TRY(copy_h2d(slots.p, sl.data(), sl.size() * sizeof(GeneSlot), st));
        FDX_TRY(copy_h2d(bits.p, bt.data(), bt.size() * sizeof(unsigned), st));
    } else {
        std::vector<unsigned long long> wd((size_t)sel_words, 0ull);
        std::vector<double> wv((size_t)G);
        std::vector<unsigned short> bv((size_t)G);
        int r = 0;
        for (int c = 0; c < G_all; ++c) {
            if ((c & 31) == 0) wd[(size_t)c >> 5] = (unsigned long long)(unsigned)r << 32;
            if (sl[(size_t)c].bucket >= 0) {
                wd[(size_t)c >> 5] |= 1ull << (c & 31);
                wv[(size_t)r] = sl[(size_t)c].w;
                bv[(size_t)r] = (unsigned short)sl[(size_t)c].bucket;
                ++r;
            }
        }
        FDX_TRY(words.alloc(wd.size() * sizeof(unsigned long long)));
        FDX_TRY(w.alloc(wv.size() * sizeof(double)));
        FDX_TRY(b.alloc(bv.size() * sizeof(unsigned short)));
        FDX_TRY(copy_h2d(words.p, wd.data(), wd.size() * sizeof(unsigned long long), st));
        FDX_TRY(copy_h2d(w.p, wv.data(), wv.size() * sizeof(double), st));
        FDX_TRY(copy_h2d(b.p, bv.data(), bv.size() * sizeof(unsigned short), st));
    }
    FDX_HIP(hipStreamSynchronize(st));       // the host tables are locals (copies below the staging threshold read them directly)
    return 0;
}

// ------------------------------------------------------------------------------------------------ fused CSR sketch -> H
// H = X_sketch * (f(Y) Omega)^T for CSR rows without Y_sketch in HBM (core/deconv.py:181-188, core/sketching.py:194-199,
// core/solver.py:205-223 in one pass).  A 16-wave workgroup per CU takes GROUPS of 16 consecutive spots (solver order):
//   fetch      wave w holds the WHOLE row of spot w in registers (NE x 64 entries, a coalesced dword load per 64;
//              the few longer rows continue in a streamed loop): the loads of a row are all issued at once, and those of the
//              wave's row of the NEXT group before the barrier of this one - the contraction and the wait for the slowest wave
//              hide their latency (round 5: a chain of 256-entry hops per row, one in flight per wave: 4.9 ms, 0.29 of peak).
//   select     one 8-byte LDS read per entry: selected? / rank among the selected genes (CsrSelection); the selected entries are
//              compacted {rank, value} into the wave's keep buffer in LDS, the registers are free for the next fetch.  log modes:
//              library size over the selected entries (the subset is taken first, deconv.py:321).
//   scatter    over full waves of kept entries: weight and bucket by rank from LDS, log1p by the row's table, ds_add_f64 into the
//              spot's d accumulators.
//   contract   the 16 x d block in LDS is the B operand of v_mfma_f64_16x16x4_f64, the contraction index split over the
//              16 waves (X_sketch slices as register-resident A operands), partial 16 x 16 type tiles added in wave order
//              through LDS and stored to H.
// The rows of a group have different lengths and meet at a barrier: the group costs its longest row.
typedef double csr_double4_t __attribute__((ext_vector_type(4)));
constexpr int CSRF_PAD = 16;     // doubles of padding per accumulator row (conflict-free B-operand reads)

template <typename T, int NE>
struct CsrRowRegs {          // entry u * 64 + lane of the row in slot u
    int c[NE];
    T y[NE];
};

template <typename T, int NE, int U0 = 0, int U1 = NE>
__device__ __forceinline__ void csr_row_fetch(CsrRowRegs<T, NE>& r, const int* __restrict__ indices, const T* __restrict__ data,
                                              long long beg, long long end, int lane) {
    const int len = (int)min<long long>(end - beg, (long long)NE * 64);
    const int* ci = indices + beg + lane;
    const T* yi = data + beg + lane;
#pragma unroll
    for (int u = U0; u < U1; ++u) {
        const bool ok = u * 64 + lane < len;
        r.c[u] = ok ? __builtin_nontemporal_load(ci + u * 64) : -1;
        r.y[u] = ok ? __builtin_nontemporal_load(yi + u * 64) : (T)0;
    }
}

// the general log1p as a CALL: it is met on a rare path only (entries the row's table does not hold); inlined, its temporaries
// push the in-flight row of the next group out of the registers on the common path
static __device__ __attribute__((noinline)) double csr_log1p_any(double x) { return fast_log1p(x); }

// rank of column c among the selected genes, -1 when it is not selected (c < 0: no entry)
__device__ __forceinline__ int csr_sel_rank(const unsigned long long* words, int c) {
    if (c < 0) return -1;
    const unsigned long long wd = words[c >> 5];
    const unsigned lo = (unsigned)wd, bit = 1u << (c & 31);
    return (lo & bit) ? (int)(wd >> 32) + __popc(lo & (bit - 1u)) : -1;
}

template <typename T, int MODE, int NB, int TT, int NE, int NE1_>
__global__ __launch_bounds__(1024, 4) void sketch_csr_contract_kernel(
    const long long* __restrict__ indptr, const int* __restrict__ indices, const T* __restrict__ data,
    const int* __restrict__ row_map, long long n, int d, const unsigned long long* __restrict__ sel_words_g, int sel_words,
    const double* __restrict__ sel_w_g, const unsigned short* __restrict__ sel_b_g, int n_sel, int sel_in_lds,
    const double* __restrict__ Xs, int K, double* __restrict__ Hout, long long ldh, double* __restrict__ row_sumsq, int no_table,
    int cap) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    constexpr int R = 16;
    constexpr int NE1 = NE1_;                                             // slots of the next row fetched ahead of the contraction
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int rs = d + CSRF_PAD;
    double* rows = reinterpret_cast<double*>(smem);                       // [16][rs]; re-used as red[16][TT*4*64]
    const int region = max(R * rs, R * TT * 4 * 64);
    double* tabs = rows + region;                                         // [16][64] per-wave log1p tables
    unsigned long long* words = reinterpret_cast<unsigned long long*>(tabs + R * 64);   // [sel_words]
    double* sel_w_l = reinterpret_cast<double*>(words + sel_words);       // [n_sel] weight by rank (sel_in_lds)
    T* keep_y = reinterpret_cast<T*>(sel_w_l + (sel_in_lds ? n_sel : 0)) + (size_t)wave * cap;   // [16][cap] the wave's selected entries: value,
    unsigned short* keep_r = reinterpret_cast<unsigned short*>(reinterpret_cast<T*>(sel_w_l + (sel_in_lds ? n_sel : 0)) + (size_t)R * cap) + (size_t)wave * cap;   // rank
    unsigned short* sel_b_l = reinterpret_cast<unsigned short*>(reinterpret_cast<T*>(sel_w_l + (sel_in_lds ? n_sel : 0)) + (size_t)R * cap) + (size_t)R * cap;   // [n_sel] bucket by rank
    double* red = rows;
    for (int j = tid; j < sel_words; j += R * 64) words[j] = sel_words_g[j];
    if (sel_in_lds)
        for (int j = tid; j < n_sel; j += R * 64) {
            sel_w_l[j] = sel_w_g[j];
            sel_b_l[j] = sel_b_g[j];
        }
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
    auto extents = [&](long long g, long long& b0, long long& e0) {
        b0 = e0 = 0;
        const long long p0 = g * R + wave;
        if (g < n_groups && p0 < n) {
            const long long row = row_map ? (long long)row_map[p0] : p0;
            b0 = indptr[row];
            e0 = indptr[row + 1];
        }
    };
    long long grp = blockIdx.x;
    long long beg, end, nbeg, nend;
    extents(grp, beg, end);
    extents(grp + gridDim.x, nbeg, nend);
    CsrRowRegs<T, NE> row;
    csr_row_fetch(row, indices, data, beg, end, lane);
    for (; grp < n_groups; grp += gridDim.x) {
        const long long s0 = grp * R;
        const long long p = s0 + wave;
        long long nnbeg, nnend;                                           // extents two groups ahead: there when the next fetch is issued
        extents(grp + 2 * (long long)gridDim.x, nnbeg, nnend);
        for (int c = lane; c < d; c += 64) acc[c] = 0.0;
        // select: one LDS read per entry says whether its column is a selected gene and which (its rank); the selected entries go,
        // compacted in CSR order, to the wave's keep buffer {rank, value} (ballot + prefix count).  Typically one entry in four
        // to six is selected: everything after this loop - library size, log1p, weights, the adds - runs over full waves of
        // selected entries.  A row longer than the registers hold is continued from memory.
        int kept = 0;                                                     // wave-uniform
        auto keep_entry = [&](int c, T yv) {                               // c < 0: no entry
            const unsigned long long wd = words[max(c, 0) >> 5];
            const unsigned lo = (unsigned)wd, bit = 1u << (c & 31);
            const bool sel = c >= 0 && (lo & bit) != 0u;
            const unsigned long long m = __ballot(sel);
            const int pos = kept + __popcll(m & ((1ULL << lane) - 1ULL));
            if (sel && pos < cap) {
                keep_r[pos] = (unsigned short)((unsigned)(wd >> 32) + __popc(lo & (bit - 1u)));
                keep_y[pos] = yv;
            }
            kept += __popcll(m);
        };
#pragma unroll
        for (int u = 0; u < NE; ++u) keep_entry(row.c[u], row.y[u]);
        for (long long q0 = beg + (long long)NE * 64; q0 < end; q0 += 64) {       // (rows of more than NE x 64 entries)
            const long long qq = q0 + lane;
            const bool ok = qq < end;
            keep_entry(ok ? indices[qq] : -1, ok ? data[qq] : (T)0);
        }
        if (p < n) {                                                      // wave-uniform: spots past the end stay zero
            double scale = 1.0;
            bool table_ok = false;                                         // wave-uniform
            if (MODE != FDX_PRE_RAW) {
                // library size over the selected entries (the subset is taken first, deconv.py:321)
                double s = 0.0;
                __builtin_amdgcn_s_waitcnt(0xc07f);                        // the kept entries are in LDS
                if (kept <= cap) {
                    for (int i = lane; i < kept; i += 64) s += (double)keep_y[i];
                } else {
                    for (long long qq = beg + lane; qq < end; qq += 64)
                        if (csr_sel_rank(words, indices[qq]) >= 0) s += (double)data[qq];
                }
                s = wave_sum(s);
                scale = 10000.0 / (s == 0.0 ? 1.0 : s);                    // deconv.py:183-185
                // counts below 64 take log1p from the row's table ENTRY BY ENTRY (a row with a few large counts keeps the table for
                // the others).  The table is filled by the very function (and the very code: a call, not a second inlined copy
                // that the compiler may contract differently) that evaluates what the table does not hold - same bits either way
                table_ok = !no_table;
                if (table_ok) tab[lane] = csr_log1p_any(__dmul_rn((double)lane, scale));
            }
            __builtin_amdgcn_s_waitcnt(0xc07f);                            // zeroing, table and kept entries are in LDS
            auto add_entry = [&](int rk, double yv) {
                double val = yv;
                if (MODE != FDX_PRE_RAW) {
                    const int ci = (int)yv;
                    const bool hit = table_ok && (double)ci == yv && (unsigned)ci < 64u;
                    val = hit ? tab[ci] : 0.0;
                    if (__ballot(!hit) != 0ULL) {                           // (rare for counts: a call, its temporaries stay out of the loop)
                        if (!hit) val = csr_log1p_any(__dmul_rn(yv, scale));
                    }
                }
                const double wgt = sel_in_lds ? sel_w_l[rk] : sel_w_g[rk];
                const int bk = sel_in_lds ? (int)sel_b_l[rk] : (int)sel_b_g[rk];
                lds_add(acc + bk, wgt * val);
            };
            if (kept <= cap) {
                for (int i0 = 0; i0 < kept; i0 += 64) {
                    const int i = i0 + lane;
                    if (i < kept) add_entry((int)keep_r[i], (double)keep_y[i]);
                }
            } else {          // more selected entries than the keep buffer holds: the row is walked again from memory
                for (long long q0 = beg; q0 < end; q0 += 64) {
                    const long long qq = q0 + lane;
                    const int rk = qq < end ? csr_sel_rank(words, indices[qq]) : -1;
                    if (rk >= 0) add_entry(rk, (double)data[qq]);
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
        // The registers of the row are free since the select: the wave's row of the NEXT group is fetched into them - the first NE1
        // slots here (in flight under the barriers and the contraction), the others behind the MFMAs.  Not earlier, and not all at
        // once: vector memory returns in order, so ANY later vector-memory wait - a spill reload above all - waits for the whole
        // fetch.  Issued ahead of the scatter, the log1p table's spilled constants were reloaded behind it (the fetch was waited for
        // on the spot, as if not prefetched: 4.7 ms whatever else changed); whole, beside the A operands and the MFMA accumulators,
        // the allocator spilled the A operands.
        csr_row_fetch<T, NE, 0, NE1>(row, indices, data, nbeg, nend, lane);
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
        csr_row_fetch<T, NE, NE1, NE>(row, indices, data, nbeg, nend, lane);   // the rest of the next row
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
        beg = nbeg; end = nend;
        nbeg = nnbeg; nend = nnend;
    }
}

static size_t csr_contract_lds(int d, int TT, int sel_words) {
    const size_t region = std::max<size_t>(16 * ((size_t)d + CSRF_PAD), (size_t)16 * TT * 4 * 64);
    return region * 8 + 16 * 64 * 8 + (size_t)sel_words * 8;
}

// shapes the fused kernel takes: the A operands of a wave (NB x TT x 4 doubles) must fit beside the row's registers, and the keep
// buffers need room for a wave's worth of entries at least
bool csr_contract_ok(int d, int K, int sel_words) {
    if (fdx::exp_env("FDX_CSR_NO_FUSED")) return false;
    if (d <= 0 || K <= 0 || K > 64 || d % 4 != 0) return false;
    const int NB = (d + 255) / 256, TT = (K + 15) / 16;
    if (NB * TT > 4) return false;
    return csr_contract_lds(d, TT, sel_words) + 16 * 64 * 10 <= 160 * 1024;
}

template <typename T, int MODE>
static int launch_csr_contract_m(const long long* indptr, const int* indices, const T* data, const int* row_map, long long n, int d,
                                 const CsrSelection& sel, const double* Xs, int K, double* H, long long ldh, double* row_sumsq,
                                 hipStream_t st) {
    const int NB = (d + 255) / 256, TT = (K + 15) / 16;
    const size_t base = csr_contract_lds(d, TT, sel.sel_words);
    const size_t tables = (size_t)sel.n_sel * (sizeof(double) + sizeof(unsigned short)) + 16;
    const size_t per_entry = 16 * (sizeof(T) + sizeof(unsigned short));          // one keep-buffer entry of every wave
    // weight / bucket by rank in LDS when that leaves the keep buffers 256 entries per wave; else read by rank from global memory
    // (a few KB: cache-resident)
    const int sel_in_lds = (sel.n_sel < 65536 && base + tables + 256 * per_entry <= 160 * 1024) ? 1 : 0;
    FDX_REQUIRE(sel.n_sel < 65536, "sketch (CSR, fused): more than 65535 selected genes");
    const size_t used = base + (sel_in_lds ? tables : 16);
    int cap = (int)std::min<size_t>(((160 * 1024 - used) / per_entry) & ~(size_t)63, 4096);
    if (const char* e = env("FDX_CSR_KEEP_CAP")) cap = std::min(cap, std::max(64, atoi(e) & ~63));   // tests: rows that overflow the buffer
    const size_t lds = used + (size_t)cap * per_entry;
    const int grid = (int)std::min<long long>((n + 15) / 16, 256);
    const int no_table = fdx::env("FDX_NO_LOG_TABLE") ? 1 : 0;
    auto launch = [&](auto kern) -> int {
        if (lds > 64 * 1024)
            FDX_HIP(hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        hipLaunchKernelGGL(kern, dim3(grid), dim3(1024), lds, st, indptr, indices, data, row_map, n, d,
                           sel.words.as<unsigned long long>(), sel.sel_words, sel.w.as<double>(), sel.b.as<unsigned short>(), sel.n_sel,
                           sel_in_lds, Xs, K, H, ldh, row_sumsq, no_table, cap);
        FDX_CHECK_LAUNCH();
        return 0;
    };
    // rows in registers: 24 x 64 float32 entries (48 registers) beside two type tiles, 16 x 64 otherwise
    constexpr int NV2 = sizeof(T) == 4 ? 24 : 16;
    constexpr int NV1 = sizeof(T) == 4 ? 12 : 6;
    if (NB == 1 && TT == 1) return launch(sketch_csr_contract_kernel<T, MODE, 1, 1, NV2, NV1>);
    if (NB == 1 && TT == 2) return launch(sketch_csr_contract_kernel<T, MODE, 1, 2, NV2, NV1>);
    if (NB == 1 && TT == 3) return launch(sketch_csr_contract_kernel<T, MODE, 1, 3, 16, 8>);
    if (NB == 1 && TT == 4) return launch(sketch_csr_contract_kernel<T, MODE, 1, 4, 16, 8>);
    if (NB == 2 && TT == 1) return launch(sketch_csr_contract_kernel<T, MODE, 2, 1, NV2, NV1>);
    if (NB == 2 && TT == 2) return launch(sketch_csr_contract_kernel<T, MODE, 2, 2, NV2, NV1>);
    if (NB == 3 && TT == 1) return launch(sketch_csr_contract_kernel<T, MODE, 3, 1, NV2, NV1>);
    if (NB == 4 && TT == 1) return launch(sketch_csr_contract_kernel<T, MODE, 4, 1, NV2, NV1>);
    return fail(FDX_ERR_UNSUPPORTED, "sketch (CSR, fused): shape not instantiated");
}

// H[:, 0..n) (type-major, row stride ldh) and row_sumsq[0..n) for the n spots listed by row_map (NULL = rows 0..n-1).
// Call only when csr_contract_ok(...) holds and `sel` was built for the fused path; Xs must be 32-byte aligned.
int launch_sketch_csr_contract(const long long* indptr, const int* indices, const void* data, int dtype, const int* row_map,
                               long long n, int d, int mode, const CsrSelection& sel, const double* Xs, int K, double* H,
                               long long ldh, double* row_sumsq, hipStream_t st) {
    if (n <= 0) return 0;
    if ((reinterpret_cast<uintptr_t>(Xs) & 31) != 0) return fail(FDX_ERR_INVALID, "sketch (CSR, fused): X_sketch must be 32-byte aligned");
    FDX_REQUIRE(sel.words.p && sel.w.p && sel.b.p, "sketch (CSR, fused): selection tables were not built for the fused path");
    const bool raw = mode == FDX_PRE_RAW;
    if (!raw && mode != FDX_PRE_LOG_CPM_SPARSE && mode != FDX_PRE_LOG_CPM) return fail(FDX_ERR_INVALID, "sketch (CSR, fused): unknown preprocess mode");
    if (dtype == FDX_F32) {
        const float* y = (const float*)data;
        return raw ? launch_csr_contract_m<float, FDX_PRE_RAW>(indptr, indices, y, row_map, n, d, sel, Xs, K, H, ldh, row_sumsq, st)
                   : launch_csr_contract_m<float, FDX_PRE_LOG_CPM_SPARSE>(indptr, indices, y, row_map, n, d, sel, Xs, K, H, ldh, row_sumsq, st);
    }
    if (dtype == FDX_F64) {
        const double* y = (const double*)data;
        return raw ? launch_csr_contract_m<double, FDX_PRE_RAW>(indptr, indices, y, row_map, n, d, sel, Xs, K, H, ldh, row_sumsq, st)
                   : launch_csr_contract_m<double, FDX_PRE_LOG_CPM_SPARSE>(indptr, indices, y, row_map, n, d, sel, Xs, K, H, ldh, row_sumsq, st);
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
//
// A visit (row, tile) is a chain of dependent round trips - extents + cursor + scale, then the entries - and a wave that
// walks it in order spends its life waiting (round 5: 3 round trips per visit, 3.74 ms at 1M x 20000 whatever the bytes).
// Here the chain is a software pipeline over the wave's rows: while visit i is being added into the tile, the entries of
// visit i + 1 (a WINDOW of up to NW x 64 entries from that row's cursor, sized by what the wave's last finished visit took)
// are in flight.  What makes that real (vector memory returns IN ORDER, and the compiler can only wait for "all but the N
// youngest" when it can count them): the window is the ONLY vector-memory traffic of the loop and its loads are unconditional
// (clamped addresses, masked afterwards) - extents and scales come by scalar loads, the cursors live in LDS (a separate
// counter).  With a cursor in global memory, or loads under a branch, every wait was for everything outstanding and the
// "prefetched" window was waited for on the spot.  A window that ends before the tile does is continued by dependent windows.
template <typename T, int NW>
struct CsrWindow {
    int c[NW];
    T y[NW];
};

struct CsrVisit {          // wave-uniform
    long long beg, end, q0;
    double sc;
};

// nw steps of 64 entries from q0; entries past `end` read the matrix's last entry instead (nnz > 0) and are masked by the caller
template <typename T, int NW>
__device__ __forceinline__ void csr_window_load(CsrWindow<T, NW>& w, const int* __restrict__ indices, const T* __restrict__ data,
                                                long long q0, long long last, int nw, int lane) {
#pragma unroll
    for (int u = 0; u < NW; ++u) {
        const long long q = min(q0 + (min(u, nw - 1) * 64 + lane), last);   // (steps past nw repeat step nw - 1: same lines, no new traffic)
        w.c[u] = indices[q];
        w.y[u] = data[q];
    }
}

constexpr int CSR_MOM_SUB = 4096;     // rows of a stripe walked at a time: their cursors (16 KB) live in LDS

template <typename T, int NS, int NW, int WPE>
__global__ __launch_bounds__(1024, WPE) void csr_moments_cursor_kernel(const long long* __restrict__ indptr, const int* __restrict__ indices,
                                                                     const T* __restrict__ data, const double* __restrict__ scale,
                                                                     long long n, long long nnz, int G, int tile, int rows_per_stripe,
                                                                     double* __restrict__ part /* (stripes, NS, G) */) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    double* acc = reinterpret_cast<double*>(smem);                       // [NS][tile]
    const int lane = threadIdx.x & 63;
    const int wib = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    double* tab = acc + (size_t)NS * tile + (size_t)wib * 64;
    int* cursor = reinterpret_cast<int*>(acc + (size_t)NS * tile + 16 * 64);   // [CSR_MOM_SUB]
    const long long s0 = (long long)blockIdx.x * rows_per_stripe;
    const long long s1 = min(n, s0 + rows_per_stripe);
    const long long last = nnz - 1;
    for (long long r0 = s0; r0 < s1; r0 += CSR_MOM_SUB) {
        const long long r1 = min(s1, r0 + CSR_MOM_SUB);
        for (int t0 = 0; t0 < G; t0 += tile) {
            const int t1 = min(G, t0 + tile);
            for (int j = threadIdx.x; j < NS * tile; j += 1024) acc[j] = 0.0;
            __syncthreads();
            auto visit = [&](long long row) -> CsrVisit {
                CsrVisit v{0, 0, 0, 1.0};
                if (row < r1) {
                    v.beg = indptr[row];
                    v.end = indptr[row + 1];
                    v.q0 = v.beg + (t0 == 0 ? 0 : (long long)cursor[row - r0]);
                    v.sc = scale[row];
                }
                return v;
            };
            // entries a visit is expected to take: what this wave's last finished visit took, before that the share of the tile's
            // width in the row (columns roughly uniform); half a step of margin, whole steps of 64
            auto steps_for = [&](const CsrVisit& v, int pred) -> int {
                const long long left = v.end - v.q0;
                const long long want = min<long long>(left, (long long)pred + 32);
                return (int)max<long long>(1, min<long long>(NW, (want + 63) >> 6));
            };
            long long row = r0 + wib;
            CsrVisit v_cur = visit(row), v_nxt = visit(row + 16);
            int pred = (int)(((v_cur.end - v_cur.beg) * (long long)(t1 - t0)) / (long long)G);
            CsrWindow<T, NW> w_cur, w_nxt;
            int nw_cur = steps_for(v_cur, pred);
            csr_window_load(w_cur, indices, data, v_cur.q0, last, nw_cur, lane);
            for (; row < r1; row += 16) {
                const CsrVisit v_nn = visit(row + 32);
                const int nw_nxt = steps_for(v_nxt, pred);
                csr_window_load(w_nxt, indices, data, v_nxt.q0, last, nw_nxt, lane);
                const double sc = fabs(v_cur.sc);
                const bool use_tab = v_cur.sc > 0.0;
                if (use_tab) log1p_table_fill(tab, sc, lane);
                __builtin_amdgcn_s_waitcnt(0xc07f);
                long long q0 = v_cur.q0;
                int nw = nw_cur;
                for (;;) {
                    bool more = true;                                     // wave-uniform: every entry so far belonged to the tile
                    int taken = 0;
                    const int left = (int)min<long long>(v_cur.end - q0, 0x7fffffffLL);   // entries of the row from q0 on
#pragma unroll
                    for (int u = 0; u < NW; ++u) {
                        if (u < nw && more) {
                            const int c = w_cur.c[u];
                            const bool in = c < t1 && u * 64 + lane < left;   // sorted: the taken entries are a prefix
                            const int cnt = __popcll(__ballot(in));
                            if (in) {
                                const double y = (double)w_cur.y[u];
                                const double z = log1p_scaled(y, sc, tab, use_tab);
                                lds_add(acc + (c - t0), z);
                                lds_add(acc + tile + (c - t0), z * z);
                                if (NS == 3) lds_add(acc + 2 * tile + (c - t0), y);
                            }
                            taken += cnt;
                            more = cnt == 64;
                        }
                    }
                    q0 += taken;
                    if (!more || q0 >= v_cur.end) break;
                    nw = (int)min<long long>(NW, (v_cur.end - q0 + 63) >> 6);   // the window ended inside the tile: a dependent one
                    csr_window_load(w_cur, indices, data, q0, last, nw, lane);
                }
                if (lane == 0) cursor[row - r0] = (int)(q0 - v_cur.beg);
                pred = (int)(q0 - v_cur.q0);
                v_cur = v_nxt;
                v_nxt = v_nn;
                w_cur = w_nxt;
                nw_cur = nw_nxt;
            }
            __syncthreads();
            for (int j = threadIdx.x; j < NS * tile; j += 1024) {
                const int s = j / tile, g = j - s * tile;
                if (t0 + g < t1) {
                    double* dst = part + ((size_t)blockIdx.x * NS + s) * G + t0 + g;
                    *dst = (r0 == s0) ? acc[j] : *dst + acc[j];           // (a stripe of more than CSR_MOM_SUB rows: its parts add up)
                }
            }
            __syncthreads();
        }
    }
}

// mean / variance (/ column sums) from the stripes' partial sums, in stripe order: 64 genes per workgroup, the stripes dealt to
// four quarter-workgroups whose sums meet through LDS (256 stripes x 2 x 20000 doubles = 82 MB: one block per 256 genes read it
// at a tenth of the memory's rate)
template <int NS>
__global__ __launch_bounds__(256) void csr_fold_moments_kernel(const double* __restrict__ part, int stripes, int G, long long n,
                                                               double* __restrict__ mean, double* __restrict__ var,
                                                               double* __restrict__ colsum) {
    __shared__ double red[3][4][64];
    const int gl = threadIdx.x & 63, qtr = threadIdx.x >> 6;
    const int g = blockIdx.x * 64 + gl;
    double s1 = 0.0, s2 = 0.0, s0 = 0.0;
    const int per = (stripes + 3) / 4;
    if (g < G)
        for (int b = qtr * per; b < min(stripes, (qtr + 1) * per); ++b) {
            const double* p = part + (size_t)b * NS * G;
            s1 += p[g];
            s2 += p[(size_t)G + g];
            if (NS == 3) s0 += p[2 * (size_t)G + g];
        }
    red[0][qtr][gl] = s1;
    red[1][qtr][gl] = s2;
    red[2][qtr][gl] = s0;
    __syncthreads();
    if (qtr != 0 || g >= G) return;
    s1 = ((red[0][0][gl] + red[0][1][gl]) + red[0][2][gl]) + red[0][3][gl];
    s2 = ((red[1][0][gl] + red[1][1][gl]) + red[1][2][gl]) + red[1][3][gl];
    s0 = ((red[2][0][gl] + red[2][1][gl]) + red[2][2][gl]) + red[2][3][gl];
    const double m = s1 / (double)n;
    mean[g] = m;
    var[g] = (n >= 2) ? fmax(((s2 / (double)n) - m * m) * ((double)n / (double)(n - 1)), 0.0) : 0.0;   // genes.py:74-83
    if (NS == 3) colsum[g] = s0;
}

int csr_moment_stripes(long long n) { return (int)std::min<long long>(512, std::max<long long>(1, (n + 255) / 256)); }

template <typename T, int NS>
static int launch_csr_moments_t(const long long* indptr, const int* indices, const T* data, long long n, long long nnz, int G,
                                double* scale, double* part, double* mean, double* var, double* colsum, bool sorted_rows,
                                hipStream_t st) {
    // sorted rows: ONE 16-wave workgroup per CU at 128 registers with as much of the gene axis as 152 KB of LDS hold (fewer
    // visits per row, fewer windows that end beside a tile edge); FDX_CSR_MOM_CFG=2: two workgroups per CU, 64 KB tiles
    const bool cursor_path = sorted_rows && nnz > 0 && !fdx::exp_env("FDX_CSR_NO_CURSOR");
    const bool one_wg = cursor_path && !(fdx::exp_env("FDX_CSR_MOM_CFG") && atoi(fdx::exp_env("FDX_CSR_MOM_CFG")) == 2);
    // (+ 16 waves x 512 B of log1p tables, + 16 KB of row cursors on the sorted path)
    const int tile_max = (int)((one_wg ? 136 : (cursor_path ? 56 : 64)) * 1024 / (NS * sizeof(double)));
    const int tiles = ceil_div(G, tile_max);
    const int tile = one_wg ? std::min(G, (ceil_div(G, tiles) + 63) & ~63) : std::min(G, tile_max);
    const int stripes = one_wg ? (int)std::min<long long>(csr_moment_stripes(n), 256) : csr_moment_stripes(n);
    const int rows_per_stripe = (int)((n + stripes - 1) / stripes);
    const int no_table = fdx::env("FDX_NO_LOG_TABLE") ? 1 : 0;
    hipLaunchKernelGGL(csr_row_scale_kernel<T>, dim3((unsigned)((n * 64 + 255) / 256)), dim3(256), 0, st, indptr, data, n, scale, no_table);
    FDX_CHECK_LAUNCH();
    const size_t lds_m = (size_t)NS * tile * sizeof(double) + 16 * 64 * sizeof(double);
    if (cursor_path) {     // rows sorted by column: one pass over the indices
        auto launch = [&](auto kern) -> int {
            const size_t lds_c = lds_m + (size_t)CSR_MOM_SUB * sizeof(int);
            FDX_HIP(hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_c));
            hipLaunchKernelGGL(kern, dim3(stripes), dim3(1024), lds_c, st, indptr, indices, data, scale, n, nnz, G, tile,
                               rows_per_stripe, part);
            return 0;
        };
        if (one_wg) FDX_TRY(launch(csr_moments_cursor_kernel<T, NS, 8, 4>));
        else FDX_TRY(launch(csr_moments_cursor_kernel<T, NS, 6, 8>));
    } else {
        FDX_HIP(hipFuncSetAttribute((const void*)csr_moments_tiled_kernel<T, NS>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_m));
        hipLaunchKernelGGL((csr_moments_tiled_kernel<T, NS>), dim3(stripes, tiles), dim3(1024), lds_m, st,
                           indptr, indices, data, scale, n, G, tile, rows_per_stripe, part);
    }
    FDX_CHECK_LAUNCH();
    hipLaunchKernelGGL(csr_fold_moments_kernel<NS>, dim3(ceil_div(G, 64)), dim3(256), 0, st, part, stripes, G, n, mean, var, colsum);
    FDX_CHECK_LAUNCH();
    return 0;
}

// scale: n doubles; part: csr_moment_stripes(n) * (colsum ? 3 : 2) * G doubles; colsum may be NULL
int launch_csr_moments(const long long* indptr, const int* indices, const void* data, int dtype, long long n, long long nnz, int G,
                       double* scale, double* part, double* mean, double* var, double* colsum, bool sorted_rows, hipStream_t st) {
    if (G <= 0 || n <= 0) return fail(FDX_ERR_INVALID, "gene moments (CSR): empty matrix");
    if (dtype == FDX_F32)
        return colsum ? launch_csr_moments_t<float, 3>(indptr, indices, (const float*)data, n, nnz, G, scale, part, mean, var, colsum, sorted_rows, st)
                      : launch_csr_moments_t<float, 2>(indptr, indices, (const float*)data, n, nnz, G, scale, part, mean, var, colsum, sorted_rows, st);
    if (dtype == FDX_F64)
        return colsum ? launch_csr_moments_t<double, 3>(indptr, indices, (const double*)data, n, nnz, G, scale, part, mean, var, colsum, sorted_rows, st)
                      : launch_csr_moments_t<double, 2>(indptr, indices, (const double*)data, n, nnz, G, scale, part, mean, var, colsum, sorted_rows, st);
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
