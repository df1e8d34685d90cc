// Device-resident spatial graph in the layout the BCD sweep consumes (internal).
#pragma once
#include <vector>

#include "fdx_internal.h"

#define FDX_TILE_HALO_CAP 1024

struct fdx_graph {
    long long n = 0;         // spots owned (updated) by this graph
    long long n_total = 0;   // owned + halo spots addressable by neighbour indices; the all-zero pad row is n_total
    long long nnz = 0;       // structural non-zeros of the owned rows
    int n_slices = 0;        // ceil(n / 64)
    long long ell_rows = 0;  // sum of per-slice widths
    int max_deg = 0;
    // Sliced ELL (slice = 64 consecutive spots = one wavefront): entry m of spot i in slice s lives at
    // ell[(slice_off[s] + m) * 64 + (i & 63)]; rows shorter than the slice width are padded with n_total,
    // the index of a spot whose abundances are identically zero.  Row entries keep CSR (ascending original
    // index) order so neighbour sums add in the reference's order.
    fdx::DevBuf ell, slice_off, deg;
    // Spot order used by the solver: position p holds original spot perm[p]; rank[perm[p]] = p.  Null = identity.
    fdx::DevBuf perm, rank;
    bool identity_order = true;
    // Row segments in solver space (neighbour positions ordered by ORIGINAL index), kept by the coordinate
    // builders for export / sharding: row p is rows[p*row_stride + row_extra[p] .. + deg[p]).
    fdx::DevBuf rows, row_extra;
    int row_stride = 0;
    // Workgroup tiles of the LDS-tiled BCD sweep (tile = 256 consecutive spots): tile_halo[t*FDX_TILE_HALO_CAP + h] is
    // the h-th neighbour position outside tile t (ascending, tile_hcnt[t] entries); ell_local mirrors `ell` with
    // tile-local slots (0..255 own, 256+h halo, 256+hcnt = all-zero pad).  `tiled` is false when some tile's halo does
    // not fit (graphs without spatial locality); the sweep then gathers from global memory.
    fdx::DevBuf tile_halo, tile_hcnt, ell_local;
    int n_tiles = 0;
    int halo_max = 0;
    bool tiled = false;
    // Sharded (local) graphs only: own spots are global sorted positions [global_lo, global_lo + n); local indices
    // n .. n_total-1 are the halo, halo_global[h] their global positions (ascending, hence grouped by owner rank:
    // rows recv_off[r] .. recv_off[r+1] come from rank r).  send_idx[send_off[r] .. send_off[r+1]) are the own local
    // indices rank r needs from this rank, in the order rank r stores them.
    long long global_lo = 0;
    long long world_n = 0;          // spots of the whole job (bounds[n_ranks]); 0: unknown
    fdx::DevBuf halo_global, send_idx;
    std::vector<int> send_off, recv_off;
    // tiles of the sweep that hold a row some peer needs (boundary) and the rest (interior), built on first use by the
    // native sharded solve (comm.cpp): boundary tiles are swept first, their rows packed and sent while the interior runs
    mutable fdx::DevBuf tiles_boundary, tiles_interior;
    // the send lists seen from the rows (built on first use by the native sharded solve): send_head[i] = 1 + first entry of own
    // row i in send_ent (0: no peer needs it), send_ent[e] = {send_off[r], rows to r, e - send_off[r], 1 + next entry of the row}
    // for entry e of send_idx - the tiled sweep writes a row's new abundances straight into the send staging through them
    mutable fdx::DevBuf send_head, send_ent;
    mutable int n_tiles_boundary = -1, n_tiles_interior = 0;
    // Deferred completion (whole-graph k-NN build, graph_kernels.cpp): every kernel of the build is queued without a host
    // round trip - the ELL is allocated for ell_cap_rows (an upper bound the kernels respect) - and the numbers only the device
    // knows (ell_rows, nnz, max_deg, halo_max, tiled) arrive in pinned memory behind meta_event.  graph_meta_sync() waits for
    // them (and rebuilds the ELL with its exact size if the bound was too small); every consumer of a graph calls it first.
    mutable bool meta_pending = false;
    mutable hipEvent_t meta_event = nullptr;
    mutable long long* meta_host = nullptr;      // pinned: [0] ell rows, [1] nnz, [2] max slice width, [3] max halo | bad-tile flag << 32
    mutable hipStream_t meta_stream = nullptr;
    long long ell_cap_rows = 0;
    mutable fdx::DevBuf keep_nbr, keep_cnt;      // inputs of the queued kernels, released by graph_meta_sync
    // k-NN builds: rows (of the range built) whose k-th and (k+1)-th nearest are exactly equidistant - the neighbour set of
    // such a spot is a choice (utils/graph.py:60-63 leaves it to cKDTree's traversal order)
    fdx::DevBuf ties_dev;
    mutable long long knn_ties = 0;
    // k-NN builds of a row range: some walk of the range left the 3 x 3 (x 3) block of cells around its point (or the band list
    // overflowed) - a band-recompute shard build is then not guaranteed to hold every reverse edge
    mutable int knn_far = 0;
    mutable struct fdx_graph_plan* keep_plan = nullptr;
    // Deferred SHARD build (graph_shard_knn, graph_kernels.cpp): the local graph of one rank - lists of own rows + band,
    // symmetrise, halo, local ELL, tile tables, send lists, boundary / interior tile lists - queued without a host round trip
    // after the bounding box.  What only the device knows (n_total, send_off / recv_off, nnz, ties, far, overflow flags)
    // arrives in the pinned block behind meta_event; graph_meta_sync() takes it over.  send_off_dev / recv_off_dev are the
    // per-peer offsets (n_ranks + 1 ints) the pack / unpack kernels of the native loop read.
    mutable bool shard_pending = false;
    mutable struct fdx_shard_build* keep_shard = nullptr;   // every buffer the queued kernels read
    int shard_failed = 0;                                    // the queued second phase of a shard build failed (its return code): graph_meta_sync keeps failing
    mutable int shard_overflow = 0;                          // a bound of the deferred build was too small: rebuild by the stepwise path
    int shard_world = 0;
    long long shard_ell_cap = 0, shard_send_cap = 0, shard_halo_cap = 0;
    fdx::DevBuf send_off_dev, recv_off_dev;
    fdx::DevBuf counts_dev;                                  // 4 doubles behind meta_event: own nnz, tied own rows, far, overflow (all-reduced by fdx_shard_fit_dev)
    // recorded by fdx_graph_build_dev ahead of the first kernel of the build, on begin_stream: the fit's prologue timer starts here
    hipEvent_t begin_event = nullptr;
    hipStream_t begin_stream = nullptr;
    ~fdx_graph();
};

namespace fdx {
int graph_meta_sync(const fdx_graph* g);         // no-op unless a deferred build is outstanding
}
