// Internal launch interfaces between the kernel translation units and the C-ABI layer.
#pragma once
#include "fdx_env.h"
#include <cstdlib>
#include <hip/hip_runtime.h>

#include <cstddef>
#include <cstdint>
#include <utility>
#include <initializer_list>

namespace fdx {

constexpr int FDX_MAX_K_FAST = 64;  // register-resident sweep kernels are instantiated for every K = 1..64 ...
// ... and for 72, 80, 88, 96 (held to 256 registers: two waves per SIMD): a solve with 65-96 cell types runs the next of these with
// all-zero pad types (zero rows / columns of XtX, zero planes of H and beta: a pad type's update is max(0, soft(0) / den) = 0, it
// adds nothing to any sum) - solver_padded_K.  Above 96 types the LDS-resident sweep takes over (bcd_kernels.cpp: 16 spots x 4 lane
// groups per wave, abundances in LDS, rolled coordinate loop; K up to 272), beyond that the generic kernel (abundances in a per-lane
// slice of global memory).  Per sweep at 500k spots: 64 types 0.41 ms, 72: 0.74, 96: 1.30, 100: 3.3 (a 112-type register instantiation
// at one wave per SIMD: 3.9; generic ~10), 120: 4.0, 200: 16.
constexpr int FDX_MAX_K_PAD = 96;
inline int solver_padded_K(int K) {
    if (K <= FDX_MAX_K_FAST || K > FDX_MAX_K_PAD || fdx::exp_env("FDX_NO_K_PAD")) return K;
    for (int kp : {72, 80, 88, 96})
        if (K <= kp) return kp;
    return K;
}
inline bool sweep_instantiated(int K) { return K >= 1 && (K <= FDX_MAX_K_FAST || K == 72 || K == 80 || K == 88 || K == 96); }

struct BcdSweepArgs {
    const double* H;         // (K, ldh) type-major: H[k*ldh + i] = <X_sketch[k], Y_sketch[i]>
    const double* XtX;       // (K, K) row-major Gram matrix
    const double* beta_in;   // (K, ld) type-major, read only (Jacobi)
    double* beta_out;        // (K, ld) type-major
    const int* ell;          // sliced-ELL neighbour indices: ell[(slice_off[s] + m)*64 + lane]
    const int* slice_off;    // (n_slices+1) prefix sum of per-slice widths
    const int* deg;          // (n) structural neighbour count per spot
    // LDS-tiled variant (see fdx_graph.h): used when `tiled` is set
    const unsigned short* ell_local = nullptr;
    const int* tile_halo = nullptr;
    const int* tile_hcnt = nullptr;
    int n_tiles = 0;
    int halo_max = 0;
    int tiled = 0;
    int objective = 0;       // tiled kernel only: evaluate the objective partial sums instead of sweeping (see bcd_sweep_inst.cpp)
    int skip_quad = 0;       // objective above 64 types: the quadratic term is left to launch_beta_quad (Gram matrix of the abundances)
    const int* tile_list = nullptr;   // tiled kernel only: sweep just these n_list tiles (sharded solve: boundary / interior)
    int n_list = 0;
    // tiled kernel: rows a peer needs also go to the send staging (send_buf[K * e.x + k * e.y + e.z], see fdx_graph::send_ent); NULL: no
    const int* send_head = nullptr;
    const int4* send_ent = nullptr;
    double* send_buf = nullptr;
    double init_uniform = 0.0;        // tiled kernel, K <= 64, whole graph: != 0 - every old abundance IS this value, beta_in is not read (sweep_init_ok)
    unsigned long long* stats;  // (max_iter, 2, 64) per-iteration max slots (bit patterns of doubles >= 0)
    double* rel_change;      // (max_iter) rel_change per iteration, written by the following kernel
    double lambda;
    double rho;              // already scaled by mean(diag XtX)   (solver.py:359-360)
    double tol;
    int ldh;
    int ld;
    int n;                   // spots updated by this sweep (own spots; halo rows are read only)
    int n_slices;
    int K;
    int it;                  // iteration index (selects the statistics slot, enables the early-exit test)
};

// ---- sketch_kernels.cpp / sketch_plan.cpp
// Static gather schedule of a CountSketch (device pointers; owned by SketchPlan).
struct SketchPlanDev {
    const int* sched_gene = nullptr;    // (total_len, 64): gene index read by lane l at schedule row e
    const double* sched_w = nullptr;    // (total_len, 64): Omega weight of that gene (0 for padding entries)
    const int* group_off = nullptr;     // (n_groups+1): schedule rows of group j are [group_off[j], group_off[j+1])
    const int* slot_bucket = nullptr;   // (n_groups*64): output bucket of slot j*64+l, -1 for unused slots
    int n_groups = 0;
    int total_len = 0;                  // schedule rows in total (= group_off[n_groups])
    // packed schedule for the register-resident kernel (valid when pack_ok): gene | bucket_code << 20 per entry,
    // bit e of end_mask set when a group ends after round e
    const unsigned int* sched_pack = nullptr;
    unsigned long long end_mask = 0ULL;
    int pack_ok = 0;
    // per-gene form (valid when scatter_ok: every gene has at most one entry): weight and bucket (-1 = none) of gene g
    const double* gene_w = nullptr;
    const int* gene_bucket = nullptr;
    int scatter_ok = 0;
    const struct SketchPlan* owner = nullptr;   // host object (holds the tile kernel's schedules)
};
// true when the scatter kernel's LDS footprint (per-gene table + one accumulator row) fits: the kernel then serves the plan
inline bool sketch_scatter_fits(int G, int d) {
    return d < 65535 && (((size_t)G + 256 + 7) & ~(size_t)7) * 10 + ((size_t)d + 64) * 8 <= 150 * 1024;
}
int launch_sketch_rows(const void* Y, int dtype, long long ldy, const int* row_map, long long n, int G, int d, int mode,
                       const SketchPlanDev& plan, double* Ys, long long ldys, double* row_sumsq, hipStream_t st);
// fused sketch + H contraction (fused_kernels.cpp)
bool fused_sketch_contract_ok(int dtype, long long ldy, const void* Y, int G, int d, int K, int mode, const SketchPlanDev& plan,
                              hipStream_t st = nullptr);
// FDX_PRE_F64_MATH of the entry point running on this thread: float32 rows keep the float64 log1p chain (tile_kernels.cpp)
struct TileF64Math {
    bool prev;
    explicit TileF64Math(bool on);
    ~TileF64Math();
};
// tile kernel (tile_kernels.cpp): atomic-free form of the same contraction, preferred when its schedule fits
bool tile_sketch_ok(int dtype, long long ldy, const void* Y, int G, int d, int K, int mode, const SketchPlanDev& plan, hipStream_t st);
int launch_tile_sketch(const void* Y, int dtype, long long ldy, const int* row_map, long long n, int G, int d, int mode,
                       const SketchPlanDev& plan, const double* Xs, int K, double* H, long long ldh, double* row_sumsq,
                       hipStream_t st);
int launch_sketch_contract(const void* Y, int dtype, long long ldy, const int* row_map, long long n, int G, int d, int mode,
                           const SketchPlanDev& plan, const double* Xs, int K, double* H, long long ldh, double* row_sumsq,
                           hipStream_t st);
// units the next tile-kernel launches of this thread leave to other streams (0 = none); returns the previous setting
int tile_sketch_reserve_cus(int cus);
int column_sums_parts(long long n);
int launch_column_sums(const void* Y, int dtype, long long ldy, long long n, int G, double* partials, double* out,
                       hipStream_t st);

// ---- gram_kernels.cpp
int launch_xyt(const double* Xs, const double* Ys, long long ldy, long long n, int d, int K, double* Hout,
               long long ldh, double* sumsq_partials, hipStream_t st);
long long xyt_partials_count(long long n);
int launch_sum_partials(const double* in, long long count, double* out, int n_out, long long stride, hipStream_t st);

// ---- finish_kernels.cpp
int objective_partials_count(int n_slices);
int launch_objective_partials(const double* beta, long long ld, const double* H, long long ldh, const double* XtX,
                              const int* ell, const int* slice_off, const int* deg, int n, int n_slices, int K,
                              double* partials, hipStream_t st, int skip_quad = 0);
int launch_normalize_export(const double* beta, long long ld, const int* perm, int n, int n_slices, int K,
                            double* beta_out, double* prop_out, hipStream_t st);

// ---- bcd_kernels.cpp
int launch_bcd_sweep(const BcdSweepArgs& a, double* generic_scratch, size_t scratch_ld, hipStream_t st);
// More than 64 cell types, no register-resident instantiation: the LDS-resident sweep (bcd_kernels.cpp) while K x 64 doubles fit;
// it reads XtX from a copy whose rows are padded with zeros to a multiple of 16 (sweep_lds_pad_doubles(K) doubles, filled by
// sweep_lds_prepare) - launch_bcd_sweep then takes that copy as its scratch argument (scratch_ld = 0).
bool sweep_uses_lds(int K);
size_t sweep_lds_pad_doubles(int K);
int sweep_lds_prepare(const double* XtX, int K, double* padded, hipStream_t st);
bool bcd_sweep_uses_tiles(const BcdSweepArgs& a);   // tile lists are honoured only then
// objective through the tiled traversal; returns 1 if not applicable (caller falls back to the generic kernel)
int launch_bcd_objective_tiled(const BcdSweepArgs& a, double* partials /* (n_tiles, 4) */, hipStream_t st);
int launch_bcd_fold_last(const unsigned long long* stats, double* rel_change, int it, hipStream_t st);
// partials[4 * b + 1] = block b's share of sum_i beta_i' XtX beta_i over the own spots (blocks 0 .. min(256, rows) - 1; `rows` rows of
// four exist): beta beta' by MFMA, contracted with XtX at the end (above 112 types block by block, ADDING to the column).  For
// objective passes run with skip_quad, which leave zeros there.
int launch_beta_quad(const double* beta, long long ld, long long n, int K, const double* XtX, double* partials, int rows,
                     hipStream_t st);

}  // namespace fdx

namespace fdx {
// ---- leverage_kernels.cpp
size_t leverage_scratch_doubles(int K, int G);
// route: the Jacobi SVD passes, or the Cholesky-QR route (no SVD; sweeps[7] = 1 when it stands, 2 when its pivots refused
// the matrix and the caller must run the SVD route)
constexpr int LEV_ROUTE_SVD = 0, LEV_ROUTE_QR = 1;
bool leverage_qr_applies(int K, int G);
int launch_leverage(const double* X, int K, int G, double reg, double* work, double* sig2, double* lev, int* sweeps,
                    double* scratch, hipStream_t st, int route = LEV_ROUTE_SVD);
}  // namespace fdx

namespace fdx {
// ---- CSR spot matrix (csr_kernels.cpp)
int launch_sketch_csr(const long long* indptr, const int* indices, const void* data, int dtype, const int* row_map,
                      long long row0, long long n, int d, int mode, const void* table, const unsigned* sel_bits,
                      int sel_words, double* Ys, long long ldys, double* row_sumsq, hipStream_t st);
// fused form (csr_kernels.cpp): CSR rows -> LDS accumulators -> MFMA contraction -> H, no Y_sketch
bool csr_contract_ok(int d, int K, int sel_words);
struct CsrSelection;
int launch_sketch_csr_contract(const long long* indptr, const int* indices, const void* data, int dtype, const int* row_map,
                               long long n, int d, int mode, const CsrSelection& sel, const double* Xs, int K, double* H,
                               long long ldh, double* row_sumsq, hipStream_t st);
int csr_moment_stripes(long long n);
int launch_csr_moments(const long long* indptr, const int* indices, const void* data, int dtype, long long n, long long nnz, int G,
                       double* scale, double* part, double* mean, double* var, double* colsum, bool sorted_rows, hipStream_t st);
int launch_csr_check(const long long* indptr, const int* indices, long long n, long long nnz, int G, int check_sorted, int* flag,
                     hipStream_t st);
}  // namespace fdx

namespace fdx {
// ---- gene statistics / column gather (sketch_kernels.cpp)
int launch_gene_moments(const void* Y, int dtype, long long ldy, long long n, int G, double* scale, double* partials,
                        double* mean, double* var, hipStream_t st);
int launch_gather_columns(const void* Y, int dtype, long long ldy, long long n, int G, const int* idx, int Gs, void* out,
                          hipStream_t st);
}  // namespace fdx
