// Builds the static gather schedule of a CountSketch on the host and uploads it (a G-entry table; see
// sketch_kernels.cpp for how the kernel walks it).  The hash/sign tables themselves come from the caller:
// flashdeconv/core/sketching.py:48-84 is reproduced on the Python host from numpy's RandomState, so the
// bucket/sign indices are bit-exact by construction.
#include "fdx_env.h"
#include "sketch_plan.h"

#include <algorithm>
#include <cstdlib>
#include <cstring>
#include <mutex>
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
    for (int i = 0; i < kTileKeys; ++i) {
        tile[i].reset();
        tile_tried[i] = false;
    }
    FDX_TRY(gene_w.alloc(gw.size() * sizeof(double)));
    FDX_TRY(gene_bucket.alloc(gb.size() * sizeof(int)));
    FDX_TRY(copy_h2d(gene_w.p, gw.data(), gw.size() * sizeof(double), st));
    FDX_TRY(copy_h2d(gene_bucket.p, gb.data(), gb.size() * sizeof(int), st));
    if (scatter_ok && sketch_scatter_fits(G, d) && !fdx::env("FDX_SKETCH_GATHER") && !fdx::exp_env("FDX_SKETCH_NO_SCATTER")) {
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
    FDX_TRY(copy_h2d(sched_pack.p, sp.data(), sp.size() * sizeof(unsigned int), st));
    FDX_TRY(sched_gene.alloc(sg.size() * sizeof(int)));
    FDX_TRY(sched_w.alloc(sw.size() * sizeof(double)));
    FDX_TRY(group_off.alloc(goff.size() * sizeof(int)));
    FDX_TRY(slot_bucket.alloc(slot_b.size() * sizeof(int)));
    FDX_TRY(copy_h2d(sched_gene.p, sg.data(), sg.size() * sizeof(int), st));
    FDX_TRY(copy_h2d(sched_w.p, sw.data(), sw.size() * sizeof(double), st));
    FDX_TRY(copy_h2d(group_off.p, goff.data(), goff.size() * sizeof(int), st));
    FDX_TRY(copy_h2d(slot_bucket.p, slot_b.data(), slot_b.size() * sizeof(int), st));
    FDX_HIP(hipStreamSynchronize(st));   // the host vectors die at scope exit
    return 0;
}

namespace {
struct CacheEntry {
    int dev = 0, G = 0, d = 0;
    unsigned long long hash = 0;
    std::vector<int32_t> bucket;
    std::vector<double> weight;
    std::shared_ptr<SketchPlan> plan;
};
// Both live on the heap and are never destroyed: the plans own pooled device buffers, and at process exit the pool's own
// statics (pool.cpp) may already be gone when this translation unit's would be torn down.
std::mutex& g_cache_mu = *new std::mutex();
std::vector<CacheEntry>& g_cache = *new std::vector<CacheEntry>();   // most recently used last
constexpr size_t kCacheCap = 12;

unsigned long long fnv1a(const void* p, size_t n, unsigned long long h) {
    const unsigned char* b = static_cast<const unsigned char*>(p);
    for (size_t i = 0; i < n; ++i) {
        h ^= b[i];
        h *= 1099511628211ULL;
    }
    return h;
}
}  // namespace

int sketch_plan_cached(const int32_t* bucket, const double* weight, int G, int d, hipStream_t st, std::shared_ptr<SketchPlan>* out) {
    FDX_REQUIRE(bucket && weight && G > 0 && d > 0 && out, "sketch plan: bad arguments");
    int dev = 0;
    FDX_HIP(hipGetDevice(&dev));
    unsigned long long h = fnv1a(bucket, (size_t)G * 4, 1469598103934665603ULL);
    h = fnv1a(weight, (size_t)G * 8, h);
    const bool use_cache = !fdx::env("FDX_NO_PLAN_CACHE");
    if (use_cache) {
        std::lock_guard<std::mutex> lk(g_cache_mu);
        for (size_t i = 0; i < g_cache.size(); ++i) {
            CacheEntry& e = g_cache[i];
            if (e.hash == h && e.dev == dev && e.G == G && e.d == d && std::memcmp(e.bucket.data(), bucket, (size_t)G * 4) == 0 &&
                std::memcmp(e.weight.data(), weight, (size_t)G * 8) == 0) {
                *out = e.plan;
                std::rotate(g_cache.begin() + (long)i, g_cache.begin() + (long)i + 1, g_cache.end());
                return 0;
            }
        }
    }
    // CSC form of Omega: genes ascending inside every bucket
    std::vector<long long> col_ptr((size_t)d + 1, 0);
    for (int g = 0; g < G; ++g) {
        FDX_REQUIRE(bucket[g] >= 0 && bucket[g] < d, "sketch plan: bucket index out of range");
        col_ptr[(size_t)bucket[g] + 1]++;
    }
    for (int c = 0; c < d; ++c) col_ptr[(size_t)c + 1] += col_ptr[(size_t)c];
    std::vector<int> gene_idx((size_t)G);
    std::vector<double> w((size_t)G);
    std::vector<long long> cur(col_ptr.begin(), col_ptr.end() - 1);
    for (int g = 0; g < G; ++g) {
        const long long e = cur[(size_t)bucket[g]]++;
        gene_idx[(size_t)e] = g;
        w[(size_t)e] = weight[g];
    }
    auto plan = std::make_shared<SketchPlan>();
    FDX_TRY(plan->build(col_ptr.data(), gene_idx.data(), w.data(), G, d, st));
    *out = plan;
    if (use_cache) {
        std::lock_guard<std::mutex> lk(g_cache_mu);
        if (g_cache.size() >= kCacheCap) g_cache.erase(g_cache.begin());
        CacheEntry e;
        e.dev = dev; e.G = G; e.d = d; e.hash = h;
        e.bucket.assign(bucket, bucket + G);
        e.weight.assign(weight, weight + G);
        e.plan = plan;
        g_cache.push_back(std::move(e));
    }
    return 0;
}

}  // namespace fdx
