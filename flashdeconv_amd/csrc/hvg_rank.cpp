// Highly-variable-gene ranking from the per-gene moments: the host tail of select_hvg (flashdeconv/utils/genes.py:104-145) as the
// reference's numpy computes it - 20 percentile bins of the positive means, z-score of the variance inside every bin, the mean /
// dispersion filters, the n_top genes of largest dispersion - for a G-vector that has just come back from the device's statistics
// kernels.  In numpy that tail cost 0.86 ms at 20000 genes (np.percentile's Python, 21 vector compares, two argsorts) between two
// device phases of a CSR fit; here ~0.15 ms.
//
// The arithmetic is numpy's, operation for operation, so that the dispersions carry numpy's bits: np.percentile's linear
// interpolation (_lerp), np.mean / np.std by PAIRWISE summation (numpy/core/src/umath/loops_utils.h.src: eight running sums over
// blocks of up to 128 elements, halves above that), no fused multiply-adds.  What is NOT restated is the order numpy's introsort
// leaves exactly equal dispersions in: the result is a SET, so ties only matter where they straddle the cut after the n_top-th
// gene - then, and when a dispersion is NaN (numpy sorts NaN last, i.e. first after the reversal), `ambiguous` is set and the
// caller takes numpy's own path.  tests/test_host.py compares the two on thousands of random vectors, bit for bit.
#include <algorithm>
#include <cmath>
#include <cstdint>
#include <vector>

#include "fdx_internal.h"

#pragma clang fp contract(off)

namespace {

double pairwise_sum(const double* a, long long n) {                  // numpy's DOUBLE_pairwise_sum, unit stride
    if (n < 8) {
        double res = 0.0;
        for (long long i = 0; i < n; ++i) res += a[i];
        return res;
    }
    if (n <= 128) {
        double r[8];
        for (int j = 0; j < 8; ++j) r[j] = a[j];
        long long i = 8;
        for (; i < n - (n % 8); i += 8)
            for (int j = 0; j < 8; ++j) r[j] += a[i + j];
        double res = ((r[0] + r[1]) + (r[2] + r[3])) + ((r[4] + r[5]) + (r[6] + r[7]));
        for (; i < n; ++i) res += a[i];
        return res;
    }
    long long n2 = n / 2;
    n2 -= n2 % 8;
    return pairwise_sum(a, n2) + pairwise_sum(a + n2, n - n2);
}

double lerp_np(double a, double b, double t) {                        // numpy/lib/_function_base_impl.py: _lerp
    const double diff = b - a;
    if (t >= 0.5) return b - diff * (1.0 - t);
    return a + diff * t;
}

}  // namespace

// mean, var: G doubles.  idx_out: room for n_top int64 (ascending gene indices on return), *n_out their number.
// *ambiguous = 1: nothing was written - take numpy's path (a tie across the cut, or a NaN dispersion).
// sorted_pos: the positive means in ascending order (np.sort(mean[mean > 0]): numpy's vectorised sort does that in a tenth of
// std::sort's time), n_pos of them.
extern "C" int fdx_hvg_from_moments(const double* mean, const double* var, int32_t G, const double* sorted_pos, int32_t n_pos,
                                    int32_t n_top, double min_mean, double max_mean, double min_disp, int64_t* idx_out,
                                    int32_t* n_out, int32_t* ambiguous) {
    FDX_REQUIRE(mean && var && idx_out && n_out && ambiguous && G >= 0 && n_top >= 0 && n_pos >= 0 && (n_pos == 0 || sorted_pos),
                "fdx_hvg_from_moments: bad arguments");
    *ambiguous = 0;
    *n_out = 0;
    std::vector<double> disp((size_t)G, 0.0);
    const double* pos = sorted_pos;
    if (n_pos >= 2) {
        const long long np_ = (long long)n_pos;
        std::vector<double> edges;
        for (int i = 0; i <= 20; ++i) {
            const double q = ((double)i * 5.0) / 100.0;               // linspace(0, 100, 21) / 100
            const double virt = (double)(np_ - 1) * q;
            double val;
            if (virt >= (double)(np_ - 1)) val = pos[(size_t)np_ - 1];
            else {
                const double prev = std::floor(virt);
                const long long ip = (long long)prev;
                val = lerp_np(pos[(size_t)ip], pos[(size_t)ip + 1], virt - prev);
            }
            edges.push_back(val);
        }
        std::sort(edges.begin(), edges.end());                        // np.unique: sorted, equal values once
        edges.erase(std::unique(edges.begin(), edges.end()), edges.end());
        const int ne = (int)edges.size();
        if (ne >= 2) {
            const int nb = ne - 1;
            std::vector<int> which((size_t)G);
            std::vector<int> count((size_t)nb + 1, 0);
            double ed[21];
            for (int e = 0; e < ne; ++e) ed[e] = edges[(size_t)e];
            for (int g = 0; g < G; ++g) {
                const double m = mean[g];
                int below = 0;                                        // np.digitize: number of edges <= m ...
                for (int e = 0; e < ne; ++e) below += m >= ed[e];     // (at most 21 compares, branch-free)
                if (m != m) below = ne;                               // ... NaN past the last edge
                const int b = std::min(std::max(below - 1, 0), nb - 1);
                which[(size_t)g] = b;
                ++count[(size_t)b + 1];
            }
            for (int b = 0; b < nb; ++b) count[(size_t)b + 1] += count[(size_t)b];
            std::vector<int> order((size_t)G);                        // a bin's genes in ascending index order (= var[which == b])
            {
                std::vector<int> cur(count.begin(), count.end() - 1);
                for (int g = 0; g < G; ++g) order[(size_t)cur[(size_t)which[(size_t)g]]++] = g;
            }
            std::vector<double> v, x;
            for (int b = 0; b < nb; ++b) {
                const int s0 = count[(size_t)b], e0 = count[(size_t)b + 1], len = e0 - s0;
                if (len <= 1) continue;
                v.resize((size_t)len);
                x.resize((size_t)len);
                for (int i = 0; i < len; ++i) v[(size_t)i] = var[order[(size_t)(s0 + i)]];
                const double mu = pairwise_sum(v.data(), len) / (double)len;                 // v.mean()
                for (int i = 0; i < len; ++i) {                                              // v.std(): _var with ddof 0
                    const double d = v[(size_t)i] - mu;
                    x[(size_t)i] = d * d;
                }
                const double sd = std::sqrt(pairwise_sum(x.data(), len) / (double)len);
                const double den = sd + 1e-10;
                for (int i = 0; i < len; ++i) disp[(size_t)order[(size_t)(s0 + i)]] = (v[(size_t)i] - mu) / den;
            }
        }
    }
    for (int g = 0; g < G; ++g)
        if (disp[(size_t)g] != disp[(size_t)g]) { *ambiguous = 1; return 0; }
    std::vector<int> cand;
    cand.reserve((size_t)G);
    for (int g = 0; g < G; ++g)
        if (mean[g] >= min_mean && mean[g] <= max_mean && disp[(size_t)g] >= min_disp) cand.push_back(g);
    if ((long long)cand.size() < (long long)n_top) {                  // not enough genes pass the filters: the top by dispersion of all
        cand.resize((size_t)G);
        for (int g = 0; g < G; ++g) cand[(size_t)g] = g;
    }
    std::vector<int> pick;
    if ((long long)cand.size() <= (long long)n_top) {
        pick = cand;
    } else if (n_top > 0) {
        std::vector<double> keys(cand.size());
        for (size_t i = 0; i < cand.size(); ++i) keys[i] = disp[(size_t)cand[i]];
        std::vector<double> sel(keys);
        std::nth_element(sel.begin(), sel.begin() + (n_top - 1), sel.end(), std::greater<double>());
        const double cut = sel[(size_t)n_top - 1];                    // the n_top-th largest dispersion
        long long above = 0, equal = 0;
        for (double k : keys) { above += k > cut; equal += k == cut; }
        if (above + equal != (long long)n_top) { *ambiguous = 1; return 0; }   // equal dispersions on both sides of the cut
        for (size_t i = 0; i < cand.size(); ++i)
            if (keys[i] >= cut) pick.push_back(cand[i]);
    }
    std::sort(pick.begin(), pick.end());
    for (size_t i = 0; i < pick.size(); ++i) idx_out[i] = pick[i];
    *n_out = (int32_t)pick.size();
    return 0;
}
