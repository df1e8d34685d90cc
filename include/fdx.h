/*
 * fdx.h -- C ABI of libfdx.so, the MI355X (gfx950) implementation of FlashDeconv's
 * sketched graph-regularised NNLS hot path.
 *
 * This is the drop-in boundary: plain pointers and sizes, no C++/torch types.  Each entry names the
 * reference function it replaces (paths under flashdeconv/ of cafferychen777/flashdeconv v0.1.6).
 * The reference is pure Python (numpy/scipy/numba); a maintainer binds these with ctypes (see
 * INTEGRATION.md), and flashdeconv_amd/_lib.py is exactly such a binding.
 *
 * Conventions
 *   - Every function returns 0 on success, <0 on failure (FDX_ERR_*); fdx_last_error() returns the
 *     message of the last failure on the calling thread.
 *   - "host" pointers are ordinary C-contiguous arrays owned by the caller; "dev" pointers are HIP device
 *     pointers on the current device (e.g. torch.Tensor.data_ptr()).  The library never frees caller memory.
 *   - `stream` is a hipStream_t passed as void* (NULL = the default stream).  Device-pointer entry points
 *     only enqueue work on that stream unless stated otherwise; host-pointer entry points are synchronous.
 *   - Device layout of abundances and of H is TYPE-MAJOR: element (type k, spot i) lives at [k*ld + i].
 *     Host-visible results (beta, proportions) are (n_spots, n_types) row-major like the reference.
 *   - Not thread-safe per handle; distinct handles may be used from distinct threads.
 */
#ifndef FDX_H
#define FDX_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define FDX_OK 0
#define FDX_ERR_INVALID (-1)
#define FDX_ERR_HIP (-2)
#define FDX_ERR_UNSUPPORTED (-3)
#define FDX_ERR_INTERNAL (-4)

/* dtype codes for the spot-by-gene matrix Y */
#define FDX_F32 0
#define FDX_F64 1

/* preprocess modes (core/deconv.py:147-235) */
#define FDX_PRE_RAW 0
#define FDX_PRE_LOG_CPM 1       /* dense rule: log1p(y / (rowsum + 1e-10) * 1e4)   (core/deconv.py:190-191) */
#define FDX_PRE_LOG_CPM_SPARSE 2 /* sparse rule: rowsum 0 -> 1, log1p on stored values (core/deconv.py:183-188) */
/* OR-ed into the mode of a dense FDX_F32 matrix whose values are float32 in STORAGE only (integer counts converted
 * exactly): numpy promotes integer input to float64 in core/deconv.py:190-191, so the transform must stay float64-accurate.
 * Without the flag float32 rows get a float32-class log1p, which is what the reference computes for float32 input. */
#define FDX_PRE_F64_MATH 0x100

int fdx_version(void);
/* Runtime switches (environment variables FDX_*: alternative tested kernel paths and diagnostics, csrc/fdx_env.cpp) are read once,
 * when the library first asks; fdx_env_reload re-reads them.  fdx_env_switch(i, &what): name and description of switch i, NULL
 * past the end of the registry. */
int fdx_env_reload(void);
const char* fdx_env_switch(int32_t i, const char** what_out);
const char* fdx_last_error(void);
int fdx_device_count(int* count);
int fdx_set_device(int device);
int fdx_device_name(char* buf, int buflen);

/* ---- plain device memory helpers (so a ctypes-only binding needs no other GPU runtime) ---------------- */
int fdx_malloc(void** dev_ptr, size_t bytes);
int fdx_free(void* dev_ptr);
/* Device scratch is recycled through a caching pool; fdx_trim() returns every cached block to the driver. */
int fdx_trim(void);
int fdx_memcpy_h2d(void* dev_dst, const void* host_src, size_t bytes, void* stream);
int fdx_memcpy_d2h(void* host_dst, const void* dev_src, size_t bytes, void* stream);
int fdx_memset(void* dev_dst, int value, size_t bytes, void* stream);
int fdx_stream_sync(void* stream);

/* ---- caller (pageable) memory <-> HBM for the matrices a fit is handed (core/deconv.py:237-243: Y is an ndarray or a scipy
 * matrix of any numeric dtype; core/deconv.py:190-191, 229: numpy promotes it to float64 on the host) --------------------------- */
/* element type of a host source array */
#define FDX_SRC_F32 0
#define FDX_SRC_F64 1
#define FDX_SRC_I8 2
#define FDX_SRC_U8 3
#define FDX_SRC_I16 4
#define FDX_SRC_U16 5
#define FDX_SRC_I32 6
#define FDX_SRC_U32 7
#define FDX_SRC_I64 8
#define FDX_SRC_U64 9
/* `count` elements of type src_code at src_host -> FDX_F32 / FDX_F64 elements at dst_dev: a team of host threads copies (integer
 * counts: converts - the reference's astype on the host) chunk by chunk into recycled pinned buffers, every chunk's DMA queued as
 * soon as it is filled.  max_abs_out (may be NULL): largest |value| of an integer source (is float32 exact?).  Returns with the
 * data in HBM; work queued on `stream` before the call is waited for first. */
int fdx_upload_convert_dev(void* dst_dev, int32_t dst_dtype, const void* src_host, int32_t src_code, int64_t count,
                           double* max_abs_out, void* stream);
/* bytes from HBM to caller memory through the same ring (beta_ / proportions_: core/deconv.py:395-398 are host arrays) */
int fdx_download_dev(void* dst_host, const void* src_dev, size_t bytes, void* stream);
/* GB/s of ONE pinned copy of `bytes` over the box's link (to_device != 0: host to device) - the yardstick for the two above */
int fdx_pinned_copy_rate(size_t bytes, int32_t to_device, double* gbps_out);

/* ---- preprocess + sketch (replaces core/deconv.py:147-235 and core/sketching.py:160-206) ------------- */
/* Y_sketch = f(Y) @ Omega for host arrays.  Y is (n, G) row-major of `dtype`; Omega (G x d) is given in CSC form
 * (col_ptr int64[d+1], gene_idx int32[nnz] ascending per column, weight f64[nnz]); `mode` is FDX_PRE_*.
 * For "pearson" pass FDX_PRE_RAW with the weights already divided by sigma_g (core/deconv.py:199-225).
 * Ys_out is (n, d) f64.  This is project_to_sketch(Y_tilde, ., Omega) when mode = FDX_PRE_RAW. */
int fdx_sketch(const void* Y, int32_t dtype, int64_t n, int32_t G, const int64_t* col_ptr, const int32_t* gene_idx,
               const double* weight, int32_t d, int32_t mode, double* Ys_out);
/* Host-only inspection of the gather schedule the tile kernel (csrc/tile_kernels.cpp) runs for a CountSketch
 * (core/sketching.py:58-74: gene g -> bucket gene_bucket[g] with weight gene_w[g]; -1 = gene not in Omega): buckets are
 * dealt to NW waves x JW groups x 4 lane classes, genes are cut into column blocks of GB.  No device call is made, so
 * tests can replay the schedule on the CPU.  dims_out[4] = {blocks, table entries, steps, steps of the busiest wave};
 * the tables are returned when slot_bucket_out is non-NULL: slot_bucket (NW*JW*4), len (NW*blocks*JW),
 * ent_base (NW*(blocks+1)), w / off (entries; cap_entries = capacity of w_out and off_out). */
int fdx_tile_schedule(const int32_t* gene_bucket, const double* gene_w, int32_t G, int32_t d, int32_t NW, int32_t JW,
                      int32_t GB, int32_t* dims_out, int32_t* slot_bucket_out, uint8_t* len_out, int32_t* ent_base_out,
                      double* w_out, uint16_t* off_out, int64_t cap_entries);
/* out[i] = log1p(y[i] * scale) exactly as the fused sketch kernel evaluates the log-CPM transform (core/deconv.py:190-191)
 * for FLOAT32 rows with every argument in [0, 32000): float32-class (the reference computes this step in float32 for
 * float32 input, numpy dtype rules), v_log_f32 plus a first-order correction of the rounding of 1 + x.  Host arrays;
 * accuracy tests call this, nothing else does. */
int fdx_log1p_f32(const float* y, float scale, int64_t n, float* out);
/* Per-gene column sums of a host (n, G) matrix (pearson's mean: core/deconv.py:207-214). */
int fdx_column_sums(const void* Y, int32_t dtype, int64_t n, int32_t G, double* sums_out);

/* Same column sums for a matrix already resident on the device (ld = row stride in elements). */
int fdx_column_sums_dev(const void* Y_dev, int32_t dtype, int64_t n, int32_t G, int64_t ldy, double* sums_out_host,
                        void* stream);

/* ---- gene selection support (utils/genes.py:18-145 select_hvg, core/deconv.py:321 gene subset) --------------- */
/* Per-gene mean and ddof-1 variance of log1p(y / max(rowsum,1) * 1e4) over the n spots of a device matrix (the
 * statistics select_hvg ranks genes by; utils/genes.py:85-102 and :52-83).  Host outputs of G doubles each. */
int fdx_gene_moments_dev(const void* Y_dev, int32_t dtype, int64_t n, int32_t G, int64_t ldy, double* mean_out_host,
                         double* var_out_host, void* stream);
/* out[r, j] = Y[r, idx[j]]: the gene subset Y[:, gene_idx] (core/deconv.py:321) as a dense (n, G_sel) device matrix of
 * the same dtype.  idx_host: G_sel int32 column indices. */
int fdx_gather_columns_dev(const void* Y_dev, int32_t dtype, int64_t n, int32_t G, int64_t ldy, const int32_t* idx_host,
                           int32_t G_sel, void* out_dev, void* stream);

/* ---- leverage scores (replaces utils/genes.py:238-290 compute_leverage_scores) ------------------------ */
/* X is the HOST (K, G) row-major reference signature matrix restricted to the selected genes; lev_out (G) receives
 * the normalised leverage scores.  One-sided Jacobi SVD of the centred G x K matrix on the device. */
int fdx_leverage_scores(const double* X, int32_t K, int32_t G, double regularization, double* lev_out);
/* Split form: begin enqueues the job on a library-owned side stream and returns; end waits, copies the scores to
 * lev_out and frees the job.  The kernel is a single workgroup, so a caller can build the spatial graph on its own
 * stream between the two calls (core/deconv.py:305-318 and :358 have no data dependence).  X may be released after
 * begin returns. */
typedef struct fdx_leverage_job fdx_leverage_job;
int fdx_leverage_begin(const double* X, int32_t K, int32_t G, double regularization, fdx_leverage_job** job);
int fdx_leverage_end(fdx_leverage_job* job, double* lev_out);
/* fdx_leverage_end that hands the job's device copy of X to the caller instead of releasing it (*x_dev_out: K x G float64, to be
 * returned with fdx_free): a fit that follows needs the same matrix on the device (fdx_shard_fit_params.X_dev). */
int fdx_leverage_end_keep(fdx_leverage_job* job, double* lev_out, double** x_dev_out);
/* fdx_leverage_begin with a choice of who queues the job's upload and launches: queue_async != 0 - the library's helper thread (the
 * caller returns at once and has ~75 us more for its own launches: fdx_leverage_begin does this); 0 - the calling thread (for a caller
 * that collects the scores right away). */
int fdx_leverage_begin_opt(const double* X, int32_t K, int32_t G, double regularization, int32_t queue_async, fdx_leverage_job** job);

/* ---- spatial graph (replaces utils/graph.py:25-212 and the CSR handling of core/solver.py:363-365) ---- */
typedef struct fdx_graph fdx_graph;

/* From an existing adjacency structure (the `A` argument of core/solver.py:287 bcd_solve): host CSR,
 * int64 indptr (n+1) / indices (nnz); values are ignored (structure only, core/solver.py:157-159).
 * Spots keep their order. */
int fdx_graph_from_csr(const int64_t* indptr, const int64_t* indices, int64_t n, fdx_graph** out);
/* From coordinates, entirely on the device: coords is a HOST (n, dim) row-major f64 array, dim in {1,2,3}.
 *   fdx_graph_build_knn    <- utils/graph.py:25-83  build_knn_graph(coords, k)   (k clamped to n-1, union symmetrised)
 *   fdx_graph_build_radius <- utils/graph.py:86-133 build_radius_graph(coords, radius)
 * Spots are reordered internally (grid-cell order) for locality; every host-visible result uses the caller's order. */
int fdx_graph_build_knn(const double* coords, int64_t n, int32_t dim, int32_t k, fdx_graph** out);
int fdx_graph_build_radius(const double* coords, int64_t n, int32_t dim, double radius, fdx_graph** out);
/* Distance from every spot to its nearest other spot (host in/out, n >= 2): the quantity whose median sets the
 * radius of build_grid_graph (utils/graph.py:163-170). */
int fdx_nearest_distance(const double* coords, int64_t n, int32_t dim, double* dist_out);
/* Adjacency as CSR in the caller's labels: indptr int64[n+1], indices int32[nnz] ascending per row (host outputs;
 * all values of the reference's matrix are 1.0).  Call fdx_graph_info first to size `indices`. */
int fdx_graph_export_csr(const fdx_graph* g, int64_t* indptr, int32_t* indices);
int fdx_graph_destroy(fdx_graph* g);
/* n spots, structural nnz, maximum degree */
int fdx_graph_info(const fdx_graph* g, int64_t* n, int64_t* nnz, int32_t* max_deg);
/* k-NN graphs: the number of spots (of the rows this graph was built for) whose k-th and (k+1)-th nearest neighbours
 * lie at EXACTLY the same distance.  The k-NN set of such a spot is not unique: the reference inherits whatever
 * scipy's cKDTree.query meets first (utils/graph.py:60-63), which depends on the order the spots are listed in; this
 * library keeps the lower spot index.  Regular lattices (Visium-HD bins, k = 6) tie on every spot.  0 for graphs
 * built from a radius or a given adjacency, and for k_neighbors = 63 (no spare slot). */
int fdx_graph_knn_ties(const fdx_graph* g, int64_t* ties);
/* The host tail of select_hvg (utils/genes.py:104-145) on a G-vector of per-gene moments, with numpy's arithmetic (percentile
 * interpolation, pairwise sums): 20 percentile bins of the positive means, z-score of the variance per bin, the mean / dispersion
 * filters, the n_top genes of largest dispersion - ascending indices in idx_out (room for n_top), their number in *n_out.
 * sorted_pos: the n_pos positive means in ascending order (the caller's np.sort).
 * *ambiguous = 1 (nothing written): exactly equal dispersions straddle the cut, or one is NaN - the order numpy's sort leaves them
 * in decides, so the caller runs numpy itself.  Pure host code. */
int fdx_hvg_from_moments(const double* mean, const double* var, int32_t G, const double* sorted_pos, int32_t n_pos, int32_t n_top,
                         double min_mean, double max_mean, double min_disp, int64_t* idx_out, int32_t* n_out, int32_t* ambiguous);
/* Starts the host build of the restated cKDTree of these coordinates (utils/graph.py:60) in a thread of the library's own and
 * returns: the next fdx_graph_plan_set_ckdtree_lists_dev on the SAME coordinates (same pointers, n, dim) takes the tree over
 * instead of building it.  coords_host may be NULL (fetched from coords_dev on the library's side stream).  One pending build per
 * process; the coordinates must stay valid until that call. */
int fdx_ckdtree_prebuild(const double* coords_host, const double* coords_dev, int64_t n, int32_t dim);
/* Host threads the restated cKDTree may use from now on (0: the process's budget): the ranks of one host, each building the tree
 * of the replicated coordinates (utils/graph.py:60), share its cores. */
int fdx_kdtree_set_threads(int32_t threads);
/* Two sizes of the restated tree's build, for the host tests and probes - the tree is the same whatever they are.
 * what 0: nodes of at least `points` points have their passes (bounds, median selection, partition) cut into tasks of the library's
 *         pool of host threads instead of made by one thread (0: the default, 400000);
 * what 1: subtrees of at most `points` points are built on a contiguous copy of their points (negative: the default, 65536;
 *         0: never);
 * what 2: 0 = fdx_graph_plan_set_ckdtree_lists_dev builds the tree on the host's threads even for 1-3 coordinates (default 1: ON
 *         THE DEVICE, csrc/kdtree_build_dev.cpp - the same tree, level by level, a team of threads per node; a tree deeper than
 *         128 levels falls back to the host build by itself). */
int fdx_kdtree_tune(int32_t what, int64_t points);
/* The index array of that tree as the DEVICE build makes it (1-3 coordinates; coords_dev: n x dim doubles on the device):
 * indices_out (host, n) must equal scipy.spatial.cKDTree(coords).indices.  info_out[3] (may be NULL) = {nodes, levels of split
 * nodes, 1 when the build gave up - a selection past introselect's depth budget or a tree deeper than the level cap; the product
 * then builds on the host}.  Tests call this; the fit uses the same build through fdx_graph_plan_set_ckdtree_lists_dev. */
int fdx_ckdtree_indices_dev(const double* coords_dev, int64_t n, int32_t dim, int64_t* indices_out, int32_t* info_out, void* stream);
/* The neighbour lists the REFERENCE gets on such inputs: a host restatement of scipy.spatial.cKDTree(coords) with the
 * constructor's defaults followed by tree.query(coords, k = kk) with p = 2 (utils/graph.py:60-63), reproducing the order in
 * which the library meets equidistant points - hence which of them it returns.  coords: HOST (n, dim) row-major f64,
 * dim 1..8; idx_out: HOST (n, kk) int64, row i = the kk nearest of point i (itself included), nearest first, -1 padded
 * when n < kk; tree_indices_out: optional HOST int64[n], the tree's index array (tests compare it with scipy's).  Pure
 * host code, serial, O(n log n): for the opt-in knn_ties="ckdtree" of the Python driver when fdx_graph_knn_ties() > 0. */
int fdx_ckdtree_knn(const double* coords, int64_t n, int32_t dim, int32_t kk, int64_t* idx_out, int64_t* tree_indices_out);
/* The same tree, queried for the listed points only (rows: HOST int64[n_rows], idx_out: HOST (n_rows, kk), row j = the answer for
 * point rows[j]) - a spot shard asks for its own rows and its band instead of all n.  dim 1..8 for both entries. */
int fdx_ckdtree_knn_rows(const double* coords, int64_t n, int32_t dim, int32_t kk, const int64_t* rows, int64_t n_rows,
                         int64_t* idx_out);

/* ---- solver (replaces core/solver.py:287-428 bcd_solve and everything it calls) ---------------------- */
typedef struct fdx_solve_info {
    int32_t converged;        /* info['converged']        */
    int32_t n_iterations;     /* info['n_iterations']     */
    double final_objective;   /* info['final_objective']  */
    double final_change;      /* info['final_change']     */
    int32_t n_objectives;     /* len(info['objectives']) (0 unless verbose) */
    int32_t reserved;
    double sweep_ms;          /* GPU time of the BCD sweeps (hipEvent), additive diagnostics */
    double total_ms;
} fdx_solve_info;

/* Host-pointer form of bcd_solve(Y_sketch, X_sketch, A, lambda_, rho, max_iter, tol, verbose):
 *   Y_sketch (n, d) row-major f64, X_sketch (K, d) row-major f64, graph built from A.
 *   beta_out (n, K) row-major f64.  rho is the user-facing fraction; it is scaled by mean(diag XtX)
 *   inside, as core/solver.py:359-360 does.  objectives_out (max_iter doubles, may be NULL) receives the
 *   verbose objective trace (core/solver.py:399-404); rel_changes_out (max_iter doubles, may be NULL)
 *   receives rel_change of every executed iteration. */
int fdx_bcd_solve(const fdx_graph* g, const double* Y_sketch, const double* X_sketch, int64_t n, int32_t d,
                  int32_t K, double lambda, double rho, int32_t max_iter, double tol, int32_t verbose,
                  double* beta_out, double* objectives_out, double* rel_changes_out, fdx_solve_info* info);

/* Function-level seams of the solver the reference's own tests import (tests/test_solver.py:7-14):
 *   fdx_gram_xty  <- precompute_gram_matrix (core/solver.py:187-201) and precompute_XtY (:204-223): XtX = Xs Xs^T (K, K),
 *                    H = Xs Ys^T (K, n) row-major; either output may be NULL.  Host arrays, f64 MFMA on the device.
 *   fdx_objective <- compute_objective (core/solver.py:226-284) for host beta (n, K) / H (K, n) and a graph built with
 *                    fdx_graph_from_csr from the structure of A (L = D - A); rho is passed as given (already scaled). */
int fdx_gram_xty(const double* X_sketch, const double* Y_sketch, int64_t n, int32_t d, int32_t K, double* XtX_out,
                 double* H_out);
int fdx_objective(const fdx_graph* g, const double* beta, const double* H, const double* XtX, int64_t n, int32_t K,
                  double YtY, double lambda, double rho, double* obj_out);

/* ---- whole fit (replaces steps 2-6 of FlashDeconv.fit, core/deconv.py:326-398) -------------------------- */
#define FDX_GRAPH_KNN 0
#define FDX_GRAPH_RADIUS 1
#define FDX_GRAPH_GIVEN 2   /* use the fdx_graph passed in */

typedef struct fdx_fit_params {
    int32_t sketch_dim;      /* d */
    int32_t mode_y;          /* FDX_PRE_* applied to Y rows */
    int32_t mode_x;          /* FDX_PRE_RAW or FDX_PRE_LOG_CPM applied to X rows (X always uses the dense rule) */
    int32_t graph_method;    /* FDX_GRAPH_* */
    int32_t k_neighbors;
    int32_t lambda_auto;     /* 1: lambda = 0.005*mean(diag XtX)/max(mean degree,1)  (core/spatial.py:144-192) */
    int32_t max_iter;
    int32_t verbose;
    double radius;
    double lambda_spatial;   /* used when lambda_auto == 0 */
    double rho_sparsity;     /* user-facing fraction, scaled by mean(diag XtX) inside */
    double tol;
    int32_t stop_on_ties;    /* k-NN graphs: 1 = return with info->status = FDX_FIT_TIES before the solve when some spot's k-th neighbour is
                                tied (the caller then rebuilds the graph on the reference's choice, utils/graph.py:60-63, and calls again) */
    int32_t reserved;
    void* carry;             /* NULL, or the info->carry of a call that stopped on ties for the SAME inputs: its sketch -> H stage is taken
                                over instead of being run again (the rebuilt graph keeps the spot order); consumed by the call */
} fdx_fit_params;
#define FDX_FIT_TIES 3

typedef struct fdx_fit_info {
    fdx_solve_info solve;
    double lambda_used;
    double rho_effective;
    double YtY;
    int64_t nnz;             /* structural non-zeros of the adjacency */
    double graph_ms, sketch_ms, gram_ms, solve_ms, finish_ms, total_ms;   /* hipEvent stage timings */
    /* contiguous device intervals: prologue_ms runs from the first kernel of the graph build (the event fdx_graph_build_dev
     * records at its top, when the graph was built on this stream; otherwise from the top of this call) to the start of the
     * sketch stage; span_ms from that same point to the end of the export, so that
     * prologue_ms + (sketch stage) + solve_ms + finish_ms = span_ms up to event granularity */
    double prologue_ms, span_ms;
    int64_t knn_ties;        /* spots whose k-th and (k+1)-th nearest neighbours are exactly equidistant (k-NN graphs built on the device) */
    int32_t status;          /* 0: fitted; FDX_FIT_TIES: stopped before the solve (stop_on_ties) - nothing but knn_ties / nnz / carry is valid */
    int32_t reserved;
    void* carry;             /* FDX_FIT_TIES: the stopped call's sketch -> H stage, still running on the device when the call returns - hand
                                it to the next fit of the same inputs (params->carry) or release it with fdx_fit_carry_free */
} fdx_fit_info;
/* Releases a carry that no fit consumed (waits for the work it holds). */
int fdx_fit_carry_free(void* carry);

/* Device-resident fit.  Y_dev: (n, G) matrix of `y_dtype` on the device, row stride ldy elements.  X: HOST (K, G)
 * f64 raw signatures (selected genes).  Omega as per-gene tables on the HOST: bucket int32[G] and the two weight
 * vectors weight_y / weight_x f64[G] (they differ only for "pearson", where each carries its own 1/sigma_g).
 * coords_dev: (n, dim) f64 on the device (ignored for FDX_GRAPH_GIVEN).  graph_inout: in for FDX_GRAPH_GIVEN,
 * otherwise receives the graph built here (caller destroys it).  beta_out_dev / prop_out_dev: (n, K) row-major f64
 * on the device in the caller's spot order (either may be NULL).  objectives_out / rel_changes_out: HOST arrays of
 * max_iter doubles (may be NULL).  Synchronous: results are complete on return. */
int fdx_fit_dev(const void* Y_dev, int32_t y_dtype, int64_t n, int32_t G, int64_t ldy, const double* X, int32_t K,
                const int32_t* bucket, const double* weight_y, const double* weight_x, const double* coords_dev,
                int32_t dim, const fdx_fit_params* params, fdx_graph** graph_inout, double* beta_out_dev,
                double* prop_out_dev, double* objectives_out, double* rel_changes_out, fdx_fit_info* info, void* stream);

/* ---- CSR (sparse) spot matrix on the device --------------------------------------------------------------- *
 * The reference accepts scipy.sparse Y and keeps it sparse through gene statistics, log-CPM and the sketch product
 * (utils/genes.py:52-83, core/deconv.py:181-188, core/sketching.py:194-199).  A view holds DEVICE pointers of a
 * canonical CSR matrix with G columns: indptr int64[n+1], indices int32[nnz], data dtype[nnz].                    */
typedef struct fdx_csr_view {
    const int64_t* indptr;
    const int32_t* indices;
    const void* data;
    int32_t dtype;           /* FDX_F32 or FDX_F64 */
    int64_t n;
    int64_t nnz;
    int32_t G;
    int32_t sorted_rows;     /* 1: column indices ascend within every row (canonical CSR) - verified by fdx_csr_check_dev;
                                lets the gene statistics read the indices once instead of once per gene tile */
} fdx_csr_view;

/* Structure check (monotone indptr from 0 to nnz, columns inside [0, G), and the sorted_rows claim if made); call once
 * after an upload, before any kernel indexes with the arrays.  FDX_ERR_INVALID with a message on violation. */
int fdx_csr_check_dev(const fdx_csr_view* Y, void* stream);
/* select_hvg statistics of the sparse branch (utils/genes.py:52-83): z = log1p(y * 1e4 / max(rowsum, 1)) on the stored
 * entries, mean_g = sum z / n, var_g = n/(n-1) * (sum z^2 / n - mean_g^2) clipped at 0; colsum_g = sum y (the "pearson"
 * means of core/deconv.py:207-212 are colsum / n).  HOST outputs of G doubles each; any may be NULL. */
int fdx_csr_gene_moments_dev(const fdx_csr_view* Y, double* mean_out_host, double* var_out_host, double* colsum_out_host,
                             void* stream);
/* fdx_fit_dev for a CSR matrix.  gene_idx: HOST int32[G] selected columns of Y (Y[:, gene_idx], core/deconv.py:321;
 * NULL = all G == Y->G columns); X, bucket, weight_y, weight_x are indexed by position in gene_idx as in fdx_fit_dev.
 * params->mode_y must be FDX_PRE_RAW or FDX_PRE_LOG_CPM_SPARSE (library size over the selected genes, 0 -> 1). */
int fdx_fit_csr_dev(const fdx_csr_view* Y, const int32_t* gene_idx, int32_t G, const double* X, int32_t K,
                    const int32_t* bucket, const double* weight_y, const double* weight_x, const double* coords_dev,
                    int32_t dim, const fdx_fit_params* params, fdx_graph** graph_inout, double* beta_out_dev,
                    double* prop_out_dev, double* objectives_out, double* rel_changes_out, fdx_fit_info* info, void* stream);

/* Per-cell-type signatures from a single-cell reference resident on the device (io/loader.py:114-135 load_reference):
 * X[k, :] = sum (mean = 0) or mean (mean = 1) over the cells of type k.  rows_dev: int32[n] the cell rows sorted by type and,
 * inside a type, ascending (a stable sort of the host-side labels); type_off_dev: int32[K + 1] the ranges of the types in
 * rows_dev.  Rows are added in that order (numpy's axis-0 order) in float64.  X_out_dev: (K, G) float64 on the device. */
int fdx_type_sums_dev(const void* Y_dev, int32_t dtype, int64_t n, int32_t G, int64_t ldy, const int32_t* rows_dev,
                      const int32_t* type_off_dev, int32_t K, int32_t mean, double* X_out_dev, void* stream);
int fdx_type_sums_csr_dev(const fdx_csr_view* Y, const int32_t* rows_dev, const int32_t* type_off_dev, int32_t K, int32_t mean,
                          double* X_out_dev, void* stream);

/* ---- device-pointer building blocks (spot-sharded multi-GPU driver, flashdeconv_amd/distributed.py) ------- *
 * One process per GPU; the host side (torch.distributed over RCCL) owns the buffers and the halo exchange, these
 * entry points only enqueue kernels on `stream`.  Same reference lines as the single-GPU entries above.           */

/* Graph from coordinates already on the device (method FDX_GRAPH_KNN / FDX_GRAPH_RADIUS).  A k-NN build may return with its
 * last kernels still queued on `stream`; every entry point that takes the graph waits for them (the fit entries order their
 * own stream behind the build when it differs). */
int fdx_graph_build_dev(const double* coords_dev, int64_t n, int32_t dim, int32_t method, int32_t k, double radius,
                        void* stream, fdx_graph** out);
/* The library's own non-blocking side stream of the current device (hipStream_t): a graph build queued there runs beside whatever
 * the caller's stream is doing (FlashDeconv.fit with gene selection: beside the gene statistics and the host's ranking). */
int fdx_side_stream(void** stream_out);
/* Work queued on `waiter` from now on starts after everything queued on `producer` so far (event record + stream wait). */
int fdx_stream_wait_stream(void* waiter, void* producer);
/* Spot shards, radius / grid graphs (utils/graph.py:84-212): rows [lo, hi) (solver positions, lo a multiple of 64) of the
 * radius graph, all other rows left empty - the shard's part of fdx_graph_build_dev(FDX_GRAPH_RADIUS).  A radius graph is
 * symmetric by construction (query_pairs, graph.py:115-121), so a shard's own rows need nothing from the other shards; the
 * result goes to fdx_graph_localize like the one of fdx_graph_from_knn_lists_dev. */
int fdx_graph_build_radius_rows_dev(const double* coords_dev, int64_t n, int32_t dim, double radius, int64_t lo, int64_t hi,
                                    void* stream, fdx_graph** out);
/* The same k-NN graph in two phases, so that a spot shard builds only its own rows (utils/graph.py:25-83):
 *   1. knn_lists: every rank bins ALL coordinates (replicated; the Morton order fixes the solver positions) and finds
 *      the k nearest neighbours of solver positions [lo, hi) only.  Rows [lo, hi) of nbr_dev (n x kk int32 solver
 *      positions, -1 padded, kk = min(k, n-1) + 1) and of cnt_dev (n int32) are written; needs n >= 2, k >= 1.
 *   2. the caller all-gathers the rows of the other ranks into the same two arrays (RCCL) - the symmetrisation
 *      A + A^T (graph.py:80-81) needs the lists of every spot that points at an own spot;
 *   3. from_knn_lists: symmetrise, sort and lay out rows [lo, hi) (lo a multiple of 64); rows outside keep degree 0.
 *      The result is a full-size graph that fdx_graph_localize can cut for this rank.  The plan is consumed.
 * With [lo, hi) = [0, n) and no exchange this is fdx_graph_build_dev(FDX_GRAPH_KNN). */
typedef struct fdx_graph_plan fdx_graph_plan;
int fdx_graph_knn_lists_dev(const double* coords_dev, int64_t n, int32_t dim, int32_t k, int64_t lo, int64_t hi,
                            int32_t* nbr_dev, int32_t* cnt_dev, void* stream, fdx_graph_plan** plan);
int fdx_graph_from_knn_lists_dev(fdx_graph_plan* plan, const int32_t* nbr_dev, const int32_t* cnt_dev, int64_t lo, int64_t hi,
                                 void* stream, fdx_graph** out);
/* Phase 1 WITHOUT the exchange ("recompute, don't communicate"): besides the lists of rows [lo, hi) the rank finds the lists of
 * its BAND - the rows outside [lo, hi) that live in grid cells within two cells of a cell holding an own row - and marks every
 * other row of cnt_dev empty.  A row whose k-NN walk stayed within two shells of its own cell (at ~4 points per cell: all but
 * freak rows) can only point at rows at most two cells away, so the band holds every row that can point at an own row, and
 * fdx_graph_from_knn_lists_dev (called directly, step 2 skipped) yields the same rows [lo, hi) bit for bit.  The condition is
 * checked where it can be: every rank reports through fdx_graph_knn_far() of the graph it built whether a walk of one of its
 * OWN rows went further (or its band list overflowed); if any rank reports it (one integer more in the all-reduce of the edge
 * counts), the ranks rebuild by the three steps above.  Cost: own rows + band, plus one 4-byte fill per spot. */
int fdx_graph_knn_lists_band_dev(const double* coords_dev, int64_t n, int32_t dim, int32_t k, int64_t lo, int64_t hi,
                                 int32_t* nbr_dev, int32_t* cnt_dev, void* stream, fdx_graph_plan** plan);
int fdx_graph_knn_far(const fdx_graph* g, int32_t* far);
/* Spot order of a plan of fdx_graph_knn_lists[_band]_dev: perm_out_dev[p] = caller's id at solver position p, rank_out_dev[id] = its
 * position (int32, n entries each, device; either may be NULL).  A band plan lays out only the neighbourhood of its own rows
 * (positions / ids elsewhere are not data). */
int fdx_graph_plan_order_dev(const fdx_graph_plan* plan, int32_t* perm_out_dev, int32_t* rank_out_dev, void* stream);
/* The caller has REPLACED rows of nbr_dev / cnt_dev (e.g. by the reference's choice among equidistant neighbours,
 * fdx_ckdtree_knn): call this before fdx_graph_from_knn_lists_dev so that the symmetrisation counts the lists it is given
 * (a whole-graph plan carries counts its k-NN kernel drew for its own lists). */
int fdx_graph_plan_lists_replaced(fdx_graph_plan* plan);
/* The usual replacement in one call: ids_host (n_rows, kk) int64 = the answers of a k-nearest query in CALLER ids (the point itself
 * usually among them, -1 padded: fdx_ckdtree_knn / fdx_ckdtree_knn_rows), row r for caller id rows_host[r] (NULL: r).  Uploaded and
 * written on the device where the symmetrisation expects them - the row's solver position, neighbours as solver positions, the
 * point itself dropped (utils/graph.py:70-74), -1 padded; includes fdx_graph_plan_lists_replaced.  Rows not listed keep their lists. */
int fdx_graph_plan_set_lists_dev(fdx_graph_plan* plan, const int64_t* ids_host, const int64_t* rows_host, int64_t n_rows,
                                 int32_t* nbr_dev, int32_t* cnt_dev, void* stream);
/* The reference's choice among equidistant neighbours for rows_host (n_rows caller ids; NULL: every spot) in one call: scipy
 * cKDTree's build restated on the host (coords_host), its k-nearest queries answered on the device for 1-3 coordinates (coords_dev:
 * the same (n, dim) float64 array; 4-8 coordinates: on the host's threads), the answers written where the symmetrisation expects
 * them as by fdx_graph_plan_set_lists_dev.  Replaces utils/graph.py:60-74 for those rows. */
int fdx_graph_plan_set_ckdtree_lists_dev(fdx_graph_plan* plan, const double* coords_host, const double* coords_dev, int64_t n,
                                         int32_t dim, const int64_t* rows_host, int64_t n_rows, int32_t* nbr_dev, int32_t* cnt_dev,
                                         void* stream);
/* perm_out_dev[p] = caller's spot id at solver position p (int32, n entries, device). */
int fdx_graph_perm_dev(const fdx_graph* g, int32_t* perm_out_dev, void* stream);
/* Shard of a full graph for rank `my_rank`: own spots are solver positions [bounds[my_rank], bounds[my_rank+1])
 * (range starts must be multiples of 256).  The local graph indexes own spots first, then the halo. */
int fdx_graph_localize(const fdx_graph* full, int32_t n_ranks, const int64_t* bounds, int32_t my_rank, void* stream,
                       fdx_graph** local);
/* The three steps above (band lists -> own rows of the symmetrised graph -> local graph of rank `my_rank`) as ONE queued
 * pipeline: after the bounding box of the coordinates nothing returns to the host - the halo, the local sliced ELL, the tile
 * tables of the sweep, the send lists of every peer and the boundary / interior tile lists are produced on the device, with
 * allocations sized by bounds the host knows.  The call returns with the kernels queued on `stream`; fdx_graph_perm_dev may be
 * queued behind it at once, every other consumer of the graph (and fdx_graph_shard_status) waits for its counts first.
 * 2 <= n_ranks <= 32, 1 to 3 coordinates, this rank must own at least one row; bounds as for fdx_graph_localize.
 * fdx_graph_shard_status: structural non-zeros of the own rows, own rows with a tied k-th neighbour, `far` (a k-NN walk of an own
 * row left its block or the band list overflowed: all-reduce it, and if any rank reports it every rank rebuilds by
 * fdx_graph_knn_lists_dev + exchange), `overflow` (a bound of this pipeline was too small, e.g. hub rows: rebuild this rank's graph
 * by the three stepwise calls - same rows, same order).  Reference: utils/graph.py:25-83 (the rows), core/solver.py:157-166
 * (what makes a shard's halo sufficient). */
int fdx_graph_shard_knn_dev(const double* coords_dev, int64_t n, int32_t dim, int32_t k, int32_t n_ranks, const int64_t* bounds,
                            int32_t my_rank, void* stream, fdx_graph** local);
int fdx_graph_shard_status(const fdx_graph* local, int64_t* own_nnz, int64_t* knn_ties, int32_t* far, int32_t* overflow);
/* Halo bookkeeping of a local graph: n_halo; send_counts[r] own rows rank r needs; recv_counts[r] halo rows owned by r. */
/* test hook: the stored neighbour indices of one row as the sweeps read them (positions in this graph's own order; local graph:
 * own rows 0..n-1, halo slots n..n_total-1); at most cap are written, *deg_out is the row's degree */
int fdx_graph_row_indices(const fdx_graph* g, int64_t row, int32_t* idx_out, int32_t cap, int32_t* deg_out);
int fdx_graph_halo_info(const fdx_graph* local, int64_t* n_halo, int32_t* send_counts, int32_t* recv_counts);
/* Own local indices to send, grouped by destination rank ascending (sum(send_counts) int32 entries, device). */
int fdx_graph_send_indices_dev(const fdx_graph* local, int32_t* idx_out_dev, void* stream);

/* X_sketch, XtX (device K*K and host copy), H (K, ldh) for `n` rows of Y (row_map_dev: int32 row ids or NULL),
 * and this shard's part of YtY.  core/sketching.py:160-206, core/solver.py:187-223,348. */
int fdx_prepare_dev(const void* Y_dev, int32_t y_dtype, int64_t n, int32_t G, int64_t ldy, const int32_t* row_map_dev,
                    const double* X, int32_t K, const int32_t* bucket, const double* weight_y, const double* weight_x,
                    int32_t d, int32_t mode_y, int32_t mode_x, double* H_out_dev, int64_t ldh, double* XtX_out_dev,
                    double* XtX_out_host, double* YtY_partial_out, void* stream);
/* The same for a shard kept sparse in HBM (scipy.sparse / torch CSR input: core/deconv.py:181-188, core/sketching.py:194-199):
 * Y = the own rows as an fdx_csr_view, gene_idx = the G selected columns (NULL = all); mode_y FDX_PRE_RAW or
 * FDX_PRE_LOG_CPM_SPARSE. */
int fdx_prepare_csr_dev(const fdx_csr_view* Y, const int32_t* gene_idx, int32_t G, const double* X, int32_t K,
                        const int32_t* bucket, const double* weight_y, const double* weight_x, int32_t d, int32_t mode_y,
                        int32_t mode_x, double* H_out_dev, int64_t ldh, double* XtX_out_dev, double* XtX_out_host,
                        double* YtY_partial_out, void* stream);
/* beta[k*ld + i] = 1/K for i < n_fill, 0 beyond (core/solver.py:372 plus the zero pad row). */
int fdx_init_beta_dev(double* beta_dev, int64_t ld, int64_t n_fill, int32_t K, void* stream);
/* One BCD sweep of the own spots of `g` (core/solver.py:104-184).  stats_dev: (max_iter, 128) uint64 slots zeroed by
 * the caller; rel_change_dev: (max_iter) doubles.  Sweep `it` tests sweep it-1's statistics on the device and
 * becomes a no-op once they are below tol, so the caller all-reduces (MAX) slot row it-1 across ranks first.  K: 1..64, the
 * fdx_solver_padded_k size of 65..96 cell types (pad planes all zero), or any number above 96.  A graph without own spots: no-op
 * (the caller folds such a rank's statistics itself: fdx_bcd_fold_dev after every all-reduce). */
int fdx_bcd_sweep_dev(const fdx_graph* g, const double* H_dev, int64_t ldh, const double* XtX_dev, const double* beta_in,
                      double* beta_out, int64_t ld, int32_t K, double lambda, double rho_eff, double tol, int32_t it,
                      void* stats_dev, double* rel_change_dev, void* stream);
int fdx_bcd_fold_dev(void* stats_dev, double* rel_change_dev, int32_t it, void* stream);
/* (cross, quad, spatial, l1) partial sums of the objective over the own spots (core/solver.py:269-284), to host. */
int fdx_objective_partials_dev(const fdx_graph* g, const double* beta_dev, int64_t ld, const double* H_dev, int64_t ldh,
                               const double* XtX_dev, int32_t K, double* out4_host, void* stream);
/* beta (K, ld) type-major -> beta_out / prop_out (n, K) row-major in solver order of the own spots. */
int fdx_normalize_dev(const double* beta_dev, int64_t ld, int64_t n, int32_t K, double* beta_out_dev, double* prop_out_dev,
                      void* stream);

/* ---- native sharded solve: the per-iteration loop on RCCL directly (csrc/comm.cpp) -------------------------- *
 * The reference has no distributed code; sharding spots is legal because the sweep is Jacobi across spots
 * (core/solver.py:157-166 reads only the previous iterate).  One process per GPU:
 *   rank 0:  fdx_comm_unique_id(id)  -> ship the 128 bytes to every rank (any side channel)
 *   all:     fdx_comm_init(id, rank, world, &comm)          (ncclCommInitRank; librccl is loaded with dlopen here)
 *            fdx_graph_localize(...) for `world` ranks, fdx_prepare_dev(...) for the own rows
 *            fdx_sharded_solve_dev(comm, local_graph, H, XtX, ...)
 * Per iteration: boundary tiles swept first, their rows packed (one kernel) and sent / received with grouped
 * ncclSend / ncclRecv on a communication stream while the interior tiles are swept, halo unpacked (one kernel),
 * ncclAllReduce(max) of the iteration's 128 convergence slots.  Same bits and iteration count as fdx_bcd_solve on the
 * unsharded problem.  fdx_local_world_* / fdx_comm_init_local: the same loop with host threads of ONE process as ranks
 * on one GPU (device copies through a shared mailbox) - for tests, no RCCL involved.  When one thread rank leaves a solve with
 * an error the world is marked aborted (the others return "another rank failed" instead of waiting for ever) and STAYS so:
 * destroy it and create a new one. */
typedef struct fdx_comm fdx_comm;
typedef struct fdx_local_world fdx_local_world;
int fdx_comm_unique_id(void* id_out_128);
int fdx_comm_init(const void* id_128, int32_t rank, int32_t world, fdx_comm** out);
int fdx_local_world_create(int32_t world, fdx_local_world** out);
/* One rank of a `world`-rank job ALONE: fdx_sharded_solve_dev runs its complete loop (boundary tiles, pack, interior tiles
 * beside the copy, unpack, stopping rule, chunked read-backs) with the exchange replaced by a device copy of the rank's own
 * staging and no all-reduce - the time of a rank's critical path without wire time (bench.py --virtual-ranks: the projected
 * speed-up of a job this build could not run on N GPUs).  The halo values are NOT those of the job: results are meaningless,
 * run it with tol = 0 and max_iter = the iteration count of the real solve. */
int fdx_comm_init_loopback(int32_t rank, int32_t world, fdx_comm** out);
int fdx_local_world_destroy(fdx_local_world* w);
int fdx_comm_init_local(fdx_local_world* w, int32_t rank, fdx_comm** out);
int fdx_comm_destroy(fdx_comm* comm);
int fdx_comm_info(const fdx_comm* comm, int32_t* rank, int32_t* world);
/* Ranks RCCL itself reports for this communicator (ncclCommCount); 0 for the in-process / loopback transports. */
int fdx_comm_rccl_count(const fdx_comm* comm, int32_t* count);
/* In-place sum over the ranks of `count` device doubles (YtY, objective partials, nnz, per-gene moment sums). */
int fdx_comm_allreduce_sum_dev(fdx_comm* comm, double* buf_dev, int32_t count, void* stream);
/* One rank's WHOLE fit behind a plan queued by fdx_graph_shard_knn_dev (or any local graph): X_sketch / XtX, sketch -> H of the
 * own rows (Y_dev: the own rows in the order of fdx_graph_perm_dev), the plan's counts all-reduced over the ranks while the sketch
 * runs, lambda (core/spatial.py:181-190), the iteration loop of fdx_sharded_solve_dev, the objective (its sums and YtY in one
 * all-reduce) and the export - core/deconv.py:326-398 for one shard, one call, no host round trip between the stages.
 * status != 0: nothing was solved and the caller applies the remedy - FDX_SHARD_FAR (some rank's k-NN walk left its block: every
 * rank rebuilds its graph by the list exchange), FDX_SHARD_OVERFLOW (a bound of some rank's queued plan was too small: the ranks
 * rebuild by the stepwise calls), FDX_SHARD_TIES (stop_on_ties set and some k-th neighbour is tied: the reference's order is the
 * caller's to establish).  nnz_total >= 0: the job's edge count (the graph was not built by the queued plan); 1 to 96 cell types. */
typedef struct {
    int32_t sketch_dim, mode_y, mode_x, lambda_auto, max_iter, stop_on_ties;
    double lambda_spatial, rho_sparsity, tol;
    int64_t n_total_spots, nnz_total;
    const double* X_dev;          /* optional: X already on the device (K x G row-major float64, e.g. from fdx_leverage_end_keep): not uploaded again */
} fdx_shard_fit_params;
typedef struct {
    int32_t status, reserved;
    int64_t nnz_total, knn_ties_total, own_nnz, n_halo;
    double lambda_used, rho_effective, YtY;
    fdx_solve_info solve;
} fdx_shard_fit_info;
#define FDX_SHARD_FAR 1
#define FDX_SHARD_OVERFLOW 2
#define FDX_SHARD_TIES 3
int fdx_shard_fit_dev(fdx_comm* comm, const fdx_graph* local, const void* Y_dev, int32_t y_dtype, int64_t n_own, int32_t G,
                      int64_t ldy, const double* X, int32_t K, const int32_t* bucket, const double* weight_y,
                      const double* weight_x, const fdx_shard_fit_params* prm, double* beta_out_dev, double* prop_out_dev,
                      double* rel_changes_out, fdx_shard_fit_info* info, void* stream);
/* The bcd_solve loop (core/solver.py:385-413) over this rank's shard.  beta0_dev / beta1_dev: (K, ld) type-major buffers
 * of the caller (initialised here: 1/K on own + halo, 0 on the pad); *result_buffer says which of the two holds the
 * final abundances.  info: n_iterations, converged, final_change, sweep_ms; rel_changes_out: max_iter doubles or NULL. */
int fdx_sharded_solve_dev(fdx_comm* comm, const fdx_graph* local, const double* H_dev, int64_t ldh, const double* XtX_dev,
                          int32_t K, double lambda, double rho_eff, double tol, int32_t max_iter, double* beta0_dev,
                          double* beta1_dev, int64_t ld, fdx_solve_info* info, double* rel_changes_out,
                          int32_t* result_buffer, void* stream);
/* More than 64 cell types on the sharded path (65..96): the solve runs the next instantiated sweep size fdx_solver_padded_k(K)
 * (72 / 80 / 88 / 96; = K up to 64) with all-zero pad types.  The caller provides H (zero planes K_real..K-1), XtX (K x K, zero
 * rows / columns for the pad types) and the two beta buffers with K = fdx_solver_padded_k(K_real) planes; the first K_real planes of
 * the result are the abundances.  fdx_sharded_solve_dev is this with K_real = K. */
int32_t fdx_solver_padded_k(int32_t K);
int fdx_sharded_solve_padded_dev(fdx_comm* comm, const fdx_graph* local, const double* H_dev, int64_t ldh, const double* XtX_dev,
                                 int32_t K, int32_t K_real, double lambda, double rho_eff, double tol, int32_t max_iter,
                                 double* beta0_dev, double* beta1_dev, int64_t ld, fdx_solve_info* info, double* rel_changes_out,
                                 int32_t* result_buffer, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* FDX_H */
