// Host-built static schedule for the CountSketch gather (internal).
#pragma once
#include <memory>
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
    // one per (input type, raw / log) pair (tile_kernels.cpp)
    std::vector<int> host_bucket;
    std::vector<double> host_w;
    mutable std::shared_ptr<TilePlanDevice> tile[4];
    mutable bool tile_tried[4] = {false, false, false, false};
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

}  // namespace fdx
