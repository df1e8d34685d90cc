// Builds the static gather schedule of a CountSketch on the host and uploads it (a G-entry table; see
// sketch_kernels.cpp for how the kernel walks it).  The hash/sign tables themselves come from the caller:
// flashdeconv/core/sketching.py:48-84 is reproduced on the Python host from numpy's RandomState, so the
// bucket/sign indices are bit-exact by construction.
#include "sketch_plan.h"

#include <algorithm>
#include <cstdlib>
#include <numeric>
#include <vector>

namespace fdx {

int SketchPlan::build(const long long* col_ptr, const int* gene_idx, const double* weight, int G_, int d_,
                      hipStream_t st) {
    FDX_REQUIRE(G_ > 0 && d_ > 0, "sketch plan: G and d must be positive");
    FDX_REQUIRE(col_ptr && (col_ptr[d_] == 0 || (gene_idx && weight)), "sketch plan: null table");
    G = G_;
    d = d_;
    for (int c = 0; c < d; ++c) {
        FDX_REQUIRE(col_ptr[c + 1] >= col_ptr[c], "sketch plan: col_ptr must be non-decreasing");
        for (long long e = col_ptr[c]; e < col_ptr[c + 1]; ++e)
            FDX_REQUIRE(gene_idx[e] >= 0 && gene_idx[e] < G, "sketch plan: gene index out of range");
    }
    // per-gene form for the scatter kernel: usable when no gene appears in more than one bucket (a CountSketch)
    std::vector<double> gw((size_t)G, 0.0);
    std::vector<int> gb((size_t)G, -1);
    scatter_ok = true;
    for (int c = 0; c < d && scatter_ok; ++c)
        for (long long e = col_ptr[c]; e < col_ptr[c + 1]; ++e) {
            const int g = gene_idx[e];
            if (gb[(size_t)g] >= 0) { scatter_ok = false; break; }
            gb[(size_t)g] = c;
            gw[(size_t)g] = weight[e];
        }
    if (scatter_ok) {
        host_bucket = gb;
        host_w = gw;
    }
    for (int i = 0; i < 4; ++i) {
        tile[i].reset();
        tile_tried[i] = false;
    }
    FDX_TRY(gene_w.alloc(gw.size() * sizeof(double)));
    FDX_TRY(gene_bucket.alloc(gb.size() * sizeof(int)));
    FDX_HIP(hipMemcpyAsync(gene_w.p, gw.data(), gw.size() * sizeof(double), hipMemcpyHostToDevice, st));
    FDX_HIP(hipMemcpyAsync(gene_bucket.p, gb.data(), gb.size() * sizeof(int), hipMemcpyHostToDevice, st));
    if (scatter_ok && sketch_scatter_fits(G, d) && !getenv("FDX_SKETCH_GATHER") && !getenv("FDX_SKETCH_NO_SCATTER")) {
        // the scatter kernel will serve this plan: the gather schedule below is never read
        n_groups = 0;
        total_len = 0;
        pack_ok = false;
        end_mask = 0ULL;
        FDX_HIP(hipStreamSynchronize(st));   // the host vectors die at scope exit
        return 0;
    }
    // Omega in CSC form: column (bucket) c holds genes gene_idx[col_ptr[c] .. col_ptr[c+1]) in ascending order.
    std::vector<std::vector<int>> lists((size_t)d);      // entry positions, so a gene may sit in several buckets
    for (int c = 0; c < d; ++c) {
        FDX_REQUIRE(col_ptr[c + 1] >= col_ptr[c], "sketch plan: col_ptr must be non-decreasing");
        for (long long e = col_ptr[c]; e < col_ptr[c + 1]; ++e) {
            FDX_REQUIRE(gene_idx[e] >= 0 && gene_idx[e] < G, "sketch plan: gene index out of range");
            lists[(size_t)c].push_back((int)e);
        }
    }
    std::vector<int> order((size_t)d);
    std::iota(order.begin(), order.end(), 0);
    std::stable_sort(order.begin(), order.end(),
                     [&](int a, int b) { return lists[(size_t)a].size() > lists[(size_t)b].size(); });
    n_groups = (d + 63) / 64;
    std::vector<int> slot_b((size_t)n_groups * 64, -1), goff((size_t)n_groups + 1, 0);
    for (int s = 0; s < d; ++s) slot_b[(size_t)s] = order[(size_t)s];
    for (int j = 0; j < n_groups; ++j) {
        size_t len = 0;
        for (int l = 0; l < 64; ++l) {
            const int b = slot_b[(size_t)j * 64 + l];
            if (b >= 0) len = std::max(len, lists[(size_t)b].size());
        }
        goff[(size_t)j + 1] = goff[(size_t)j] + (int)len;
    }
    total_len = goff[(size_t)n_groups];
    std::vector<int> sg((size_t)std::max<long long>(total_len, 1) * 64, 0);
    std::vector<double> sw((size_t)std::max<long long>(total_len, 1) * 64, 0.0);
    for (int j = 0; j < n_groups; ++j)
        for (int l = 0; l < 64; ++l) {
            const int b = slot_b[(size_t)j * 64 + l];
            if (b < 0) continue;
            const auto& L = lists[(size_t)b];
            for (size_t t = 0; t < L.size(); ++t) {
                const size_t e = ((size_t)goff[(size_t)j] + t) * 64 + (size_t)l;
                sg[e] = gene_idx[L[t]];
                sw[e] = weight[L[t]];
            }
        }
    // Packed form for the register-resident kernel: entry = gene | bucket_code << 20, where bucket_code is the output
    // bucket of the lane's slot in the group the round belongs to (0xFFF = unused slot); end_mask bit e is set when a
    // group ends after round e.  Usable when G < 2^20, d < 4095 and the schedule has at most 64 rounds.
    pack_ok = (G < (1 << 20)) && (d < 4095) && (total_len >= 1) && (total_len <= 64);
    end_mask = 0ULL;
    std::vector<unsigned int> sp((size_t)std::max<long long>(total_len, 1) * 64, 0xFFF00000u);
    if (pack_ok) {
        for (int j = 0; j < n_groups; ++j) {
            if (goff[(size_t)j + 1] > goff[(size_t)j]) end_mask |= 1ULL << (goff[(size_t)j + 1] - 1);
            for (int l = 0; l < 64; ++l) {
                const int b = slot_b[(size_t)j * 64 + l];
                const unsigned code = (b < 0) ? 0xFFFu : (unsigned)b;
                for (int e = goff[(size_t)j]; e < goff[(size_t)j + 1]; ++e)
                    sp[(size_t)e * 64 + (size_t)l] = (unsigned)sg[(size_t)e * 64 + (size_t)l] | (code << 20);
            }
        }
    }
    FDX_TRY(sched_pack.alloc(sp.size() * sizeof(unsigned int)));
    FDX_HIP(hipMemcpyAsync(sched_pack.p, sp.data(), sp.size() * sizeof(unsigned int), hipMemcpyHostToDevice, st));
    FDX_TRY(sched_gene.alloc(sg.size() * sizeof(int)));
    FDX_TRY(sched_w.alloc(sw.size() * sizeof(double)));
    FDX_TRY(group_off.alloc(goff.size() * sizeof(int)));
    FDX_TRY(slot_bucket.alloc(slot_b.size() * sizeof(int)));
    FDX_HIP(hipMemcpyAsync(sched_gene.p, sg.data(), sg.size() * sizeof(int), hipMemcpyHostToDevice, st));
    FDX_HIP(hipMemcpyAsync(sched_w.p, sw.data(), sw.size() * sizeof(double), hipMemcpyHostToDevice, st));
    FDX_HIP(hipMemcpyAsync(group_off.p, goff.data(), goff.size() * sizeof(int), hipMemcpyHostToDevice, st));
    FDX_HIP(hipMemcpyAsync(slot_bucket.p, slot_b.data(), slot_b.size() * sizeof(int), hipMemcpyHostToDevice, st));
    FDX_HIP(hipStreamSynchronize(st));   // the host vectors die at scope exit
    return 0;
}

}  // namespace fdx
