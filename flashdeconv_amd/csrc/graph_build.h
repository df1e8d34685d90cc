// Device graph builders (graph_kernels.cpp); internal.
#pragma once
#include "fdx_graph.h"

struct fdx_graph_plan;   // binned points of a two-phase k-NN build (graph_kernels.cpp)

namespace fdx {

// coords: device (n, dim) row-major float64, dim in {1,2,3}
int graph_build_knn(const double* d_coords, long long n, int dim, int k, fdx_graph* g, hipStream_t st);
// two-phase k-NN build for spot shards (see graph_kernels.cpp): lists of rows [lo, hi) -> caller all-gathers -> own rows
// band: also the lists of the rows outside [lo, hi) in cells next to a cell with an own row; every other row of cnt reads 0
int graph_knn_lists(const double* d_coords, long long n, int dim, int k, long long lo, long long hi, int* nbr, int* cnt,
                    fdx_graph_plan** out, hipStream_t st, bool band = false);
int graph_from_knn_lists(fdx_graph_plan* plan, const int* nbr, const int* cnt, long long lo, long long hi, fdx_graph* g,
                         hipStream_t st);
void graph_plan_destroy(fdx_graph_plan* plan);
int graph_plan_kk(const fdx_graph_plan* plan);
int graph_plan_lists_replaced(fdx_graph_plan* plan);   // nbr / cnt were overwritten by the caller: forget the counts drawn for the kernel's own lists
// replace list rows by the answers of a k-nearest query in caller ids (host arrays): mapped to solver positions on the device
int graph_plan_set_lists(fdx_graph_plan* plan, const long long* ids_host, const long long* rows_host, long long n_rows, int* nbr,
                         int* cnt, hipStream_t st);
int graph_plan_set_lists_device(fdx_graph_plan* plan, const long long* ids_dev, const long long* rows_host, long long n_rows, int* nbr,
                                int* cnt, hipStream_t st);
// kdtree_order.cpp: the reference's k-nearest lists (scipy cKDTree's tie order) for rows_host (NULL: all points) into ids_dev
// (n_rows x kk int64, caller ids, nearest first, self included, -1 padded); tree on the host, queries on the device
int ckdtree_lists_device(const double* coords_host, const double* coords_dev, long long n, int dim, int kk, const long long* rows_host,
                         long long n_rows, long long* ids_dev, hipStream_t st);
int graph_plan_order(const fdx_graph_plan* plan, int* d_perm_out, int* d_rank_out, hipStream_t st);   // device copies of perm / rank (either may be NULL)
// rows [lo, hi) (solver positions) of the radius graph, the others left empty; [0, n) = the whole graph
int graph_build_radius(const double* d_coords, long long n, int dim, double radius, long long lo, long long hi, fdx_graph* g,
                       hipStream_t st);
// distance of every point to its nearest other point (caller's order); used by the "grid" method (graph.py:163-167)
int graph_nearest_distance(const double* d_coords, long long n, int dim, double* d_out, hipStream_t st);
// CSR in the caller's labels; device outputs indptr (n+1) int64, indices (nnz) int32 ascending per row
int graph_export_csr(const fdx_graph* g, long long* d_indptr, int* d_indices, hipStream_t st);

// Shard [lo, hi) of a full coordinate-built graph for rank `my_rank` of `n_ranks` (bounds: n_ranks+1 range starts).
int graph_localize(const fdx_graph* full, long long lo, long long hi, int n_ranks, const long long* bounds, int my_rank,
                   fdx_graph* loc, hipStream_t st);
// The local graph of rank `my_rank` of a k-NN job in ONE queued pipeline (lists of own rows + band, symmetrise, halo, local ELL,
// tile tables, send lists, boundary / interior tile lists): nothing returns to the host after the bounding box; graph_meta_sync(loc)
// takes over the counts (loc->shard_overflow: a bound was too small - rebuild by knn_lists / from_knn_lists / localize).
int graph_shard_knn(const double* d_coords, long long n, int dim, int k, int n_ranks, const long long* bounds, int my_rank,
                    fdx_graph* loc, hipStream_t st);
// queue the second phase of a deferred shard build now (no-op when it has been queued); st NULL: the library's plan stream
int shard_queue_rest(fdx_graph* loc, hipStream_t st);
// a pending shard build's second phase is queued when this returns (waits for the helper thread, or queues it here)
int graph_shard_join(const fdx_graph* g);
// solver position -> caller's spot id, device int32 (n)
int graph_copy_perm(const fdx_graph* g, int* d_out, hipStream_t st);

}  // namespace fdx
