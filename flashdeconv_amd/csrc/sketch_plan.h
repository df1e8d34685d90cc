// Host-built static schedule for the CountSketch gather (internal).
#pragma once
#include <memory>
#include <mutex>
#include <vector>

#include "fdx_internal.h"
#include "fdx_kernels.h"

namespace fdx {

struct TilePlanDevice;   // tile_kernels.cpp

struct SketchPlan {
    int G = 0, d = 0;
    int n_groups = 0;
    long long total_len = 0;
    bool pack_ok = false;
    unsigned long long end_mask = 0ULL;
    bool scatter_ok = false;
    DevBuf sched_gene, sched_w, group_off, slot_bucket, sched_pack, gene_w, gene_bucket;
    // per-gene form on the host (valid when scatter_ok) and the tile kernel's schedules built from it on first use,
    // one per (input type, raw / log, type tiles, wave split) (tile_kernels.cpp)
    std::vector<int> host_bucket;
    std::vector<double> host_w;
    static constexpr int kTileKeys = 224;    // (input type, raw / log, type tiles, wave split) + the wide form (input type, raw / log), x ring of three, x wide raw without WG, x flat schedule
    mutable std::shared_ptr<TilePlanDevice> tile[kTileKeys];
    mutable bool tile_tried[kTileKeys] = {};
    mutable std::mutex tile_mu;              // plans are shared through the cache: schedules are built under this lock
    SketchPlanDev dev() const {
        SketchPlanDev p;
        p.sched_gene = sched_gene.as<int>();
        p.sched_w = sched_w.as<double>();
        p.group_off = group_off.as<int>();
        p.slot_bucket = slot_bucket.as<int>();
        p.sched_pack = sched_pack.as<unsigned int>();
        p.n_groups = n_groups;
        p.total_len = (int)total_len;
        p.pack_ok = pack_ok ? 1 : 0;
        p.end_mask = end_mask;
        p.gene_w = gene_w.as<double>();
        p.gene_bucket = gene_bucket.as<int>();
        p.scatter_ok = scatter_ok ? 1 : 0;
        p.owner = this;
        return p;
    }
    // Omega (G x d) in CSC form on the host: col_ptr (d+1), gene_idx / weight (nnz), genes ascending per column.
    // A CountSketch has exactly one entry per gene; any sparse Omega works (project_to_sketch accepts one).
    int build(const long long* col_ptr, const int* gene_idx, const double* weight, int G_, int d_, hipStream_t st);
};

// CountSketch plans by content: `bucket` (G int32) and `weight` (G doubles) of the per-gene tables (core/sketching.py:58-82).
// A fit rebuilds the same Omega every time it is called with the same genes, seed and leverage scores; the plans (CSC
// form, device tables, the tile kernel's schedule with its ~0.5 ms greedy packing) are kept in a small per-process cache
// and shared.  The returned plan stays valid while the caller holds the pointer, whatever the cache evicts.
int sketch_plan_cached(const int32_t* bucket, const double* weight, int G, int d, hipStream_t st, std::shared_ptr<SketchPlan>* out);

}  // namespace fdx
