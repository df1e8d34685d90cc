// Host-side construction of the tile kernel's gather schedule (see tile_plan.h).  Pure C++: no device calls.
#include "tile_plan.h"

#include <algorithm>
#include <numeric>

namespace fdx {

bool build_tile_plan(const int* gene_bucket, const double* gene_w, int G, int d, int NW, int JW, int GB, TilePlanHost* out) {
    if (G <= 0 || d <= 0 || NW <= 0 || JW <= 0 || GB <= 0 || GB > 65536) return false;
    const int n_groups = NW * JW;
    if (d > 4 * n_groups) return false;
    const int NBLK = (G + GB - 1) / GB;
    TilePlanHost& p = *out;
    p = TilePlanHost();
    p.G = G; p.d = d; p.NW = NW; p.JW = JW; p.GB = GB; p.NBLK = NBLK;

    // genes of every bucket in ascending order (the reference's summation order inside a bucket), counts per block
    std::vector<std::vector<int>> genes((size_t)d);
    for (int g = 0; g < G; ++g) {
        const int b = gene_bucket[g];
        if (b < 0) continue;
        if (b >= d) return false;
        genes[(size_t)b].push_back(g);
    }
    std::vector<int> cnt((size_t)d * NBLK, 0);
    for (int b = 0; b < d; ++b)
        for (int g : genes[(size_t)b]) cnt[(size_t)b * NBLK + g / GB]++;

    // Four buckets share a group and advance in lockstep, so a group costs sum_c max_q cnt[b_q][c] steps.  Greedy
    // packing: seed a group with the largest unassigned bucket, then add three times the bucket that raises the group's
    // envelope (the per-block maximum) least, preferring the largest such bucket - companions slip under the envelope.
    struct Group { int b[4]; int cost; std::vector<int> len; };
    std::vector<Group> groups((size_t)n_groups);
    std::vector<int> total((size_t)d, 0);
    for (int b = 0; b < d; ++b) total[(size_t)b] = (int)genes[(size_t)b].size();
    std::vector<char> used((size_t)d, 0);
    int n_left = d;
    for (int gi = 0; gi < n_groups; ++gi) {
        Group& gr = groups[(size_t)gi];
        gr.len.assign((size_t)NBLK, 0);
        gr.cost = 0;
        for (int q = 0; q < 4; ++q) gr.b[q] = -1;
        for (int q = 0; q < 4 && n_left > 0; ++q) {
            int best = -1, best_inc = 0;
            for (int b = 0; b < d; ++b) {
                if (used[(size_t)b]) continue;
                int inc = 0;
                for (int c = 0; c < NBLK; ++c) inc += std::max(0, cnt[(size_t)b * NBLK + c] - gr.len[(size_t)c]);
                if (q == 0) inc = -total[(size_t)b];                        // seed: the largest bucket
                if (best < 0 || inc < best_inc || (inc == best_inc && total[(size_t)b] > total[(size_t)best])) {
                    best = b;
                    best_inc = inc;
                }
            }
            used[(size_t)best] = 1;
            --n_left;
            gr.b[q] = best;
            for (int c = 0; c < NBLK; ++c) gr.len[(size_t)c] = std::max(gr.len[(size_t)c], cnt[(size_t)best * NBLK + c]);
        }
        for (int c = 0; c < NBLK; ++c) {
            if (gr.len[(size_t)c] > 255) return false;
            gr.cost += gr.len[(size_t)c];
        }
    }
    // deal the groups to the waves longest first, always to the least loaded wave that still has a free group slot
    std::vector<int> gorder((size_t)n_groups);
    std::iota(gorder.begin(), gorder.end(), 0);
    std::stable_sort(gorder.begin(), gorder.end(), [&](int a, int b) { return groups[(size_t)a].cost > groups[(size_t)b].cost; });
    std::vector<std::vector<int>> of_wave((size_t)NW);
    std::vector<int> load((size_t)NW, 0);
    for (int gi : gorder) {
        int best = -1;
        for (int w = 0; w < NW; ++w)
            if ((int)of_wave[(size_t)w].size() < JW && (best < 0 || load[(size_t)w] < load[(size_t)best])) best = w;
        of_wave[(size_t)best].push_back(gi);
        load[(size_t)best] += groups[(size_t)gi].cost;
    }

    p.slot_bucket.assign((size_t)n_groups * 4, -1);
    p.len.assign((size_t)NW * NBLK * JW, 0);
    p.ent_base.assign((size_t)NW * (NBLK + 1), 0);
    p.w.clear();
    p.off.clear();
    p.gene.clear();
    for (int w = 0; w < NW; ++w) {
        int wave_steps = 0;
        for (int j = 0; j < JW; ++j)
            for (int q = 0; q < 4; ++q) {
                const int b = groups[(size_t)of_wave[(size_t)w][(size_t)j]].b[q];
                p.slot_bucket[((size_t)w * JW + j) * 4 + q] = b;
                if (b >= 0) p.jw_used = std::max(p.jw_used, j + 1);
            }
        std::vector<int> taken((size_t)JW * 4, 0);   // genes of slot (j, q) already scheduled
        for (int c = 0; c < NBLK; ++c) {
            p.ent_base[(size_t)w * (NBLK + 1) + c] = (int)p.w.size();
            for (int j = 0; j < JW; ++j) {
                const Group& gr = groups[(size_t)of_wave[(size_t)w][(size_t)j]];
                const int L = gr.len[(size_t)c];
                p.len[((size_t)w * NBLK + c) * JW + j] = (unsigned char)L;
                wave_steps += L;
                for (int t = 0; t < L; ++t)
                    for (int q = 0; q < 4; ++q) {
                        double wt = 0.0;
                        int o = 0;                               // padding: weight 0 on the block's first gene
                        int gid = -1;
                        const int b = gr.b[q];
                        if (b >= 0) {
                            int& k = taken[(size_t)j * 4 + q];
                            const std::vector<int>& L_b = genes[(size_t)b];
                            if (k < (int)L_b.size() && L_b[(size_t)k] / GB == c) {
                                wt = gene_w[L_b[(size_t)k]];
                                o = L_b[(size_t)k] - c * GB;
                                gid = L_b[(size_t)k];
                                ++k;
                            }
                        }
                        p.w.push_back(wt);
                        p.off.push_back((unsigned short)o);
                        p.gene.push_back(gid);
                    }
            }
        }
        p.ent_base[(size_t)w * (NBLK + 1) + NBLK] = (int)p.w.size();
        for (int q = 0; q < 8; ++q) {   // two padding steps: the kernel prefetches two steps past the end
            p.w.push_back(0.0);
            p.off.push_back(0);
            p.gene.push_back(-1);
        }
        p.steps += wave_steps;
        p.max_wave_steps = std::max(p.max_wave_steps, wave_steps);
        // every gene of every owned bucket must have been scheduled
        for (int j = 0; j < JW; ++j)
            for (int q = 0; q < 4; ++q) {
                const int b = groups[(size_t)of_wave[(size_t)w][(size_t)j]].b[q];
                if (b >= 0 && taken[(size_t)j * 4 + q] != (int)genes[(size_t)b].size()) return false;
            }
    }
    p.NE = (int)p.w.size();
    return true;
}

bool build_tile_flat(const TilePlanHost& p, int pad_off, TileFlatHost* out) {
    if (p.NW <= 0 || p.JW <= 0 || p.JW > 24 || p.NBLK <= 0 || pad_off < 0 || pad_off > 65535) return false;
    TileFlatHost& t = *out;
    t = TileFlatHost();
    const int NW = p.NW, JW = p.JW, NBLK = p.NBLK;
    // steps of a (wave, block): the groups below 16 first, padded to a multiple of 8 (the kernel serves 8 steps at a time and keeps
    // the sums of groups 0..15 and 16..23 in two register vectors, one of them indexed per phase), then the others, padded likewise
    auto phase_steps = [&](int w, int c, int j0, int j1) {
        int ns = 0;
        for (int j = j0; j < std::min(j1, JW); ++j) ns += p.len[((size_t)w * NBLK + c) * JW + j];
        return ns;
    };
    int ns_max = 0;
    for (int w = 0; w < NW; ++w)
        for (int c = 0; c < NBLK; ++c)
            ns_max = std::max(ns_max, ((phase_steps(w, c, 0, 16) + 7) & ~7) + ((phase_steps(w, c, 16, 24) + 7) & ~7));
    if (ns_max > 65535) return false;
    t.NSP = std::max(8, ns_max);
    t.GROW = 8 + t.NSP;
    t.off.assign((size_t)NW * NBLK * 4 * t.NSP, (unsigned short)pad_off);
    t.gid.assign((size_t)NW * NBLK * t.GROW, 0);
    for (int w = 0; w < NW; ++w)
        for (int c = 0; c < NBLK; ++c) {
            unsigned char* g = &t.gid[((size_t)w * NBLK + c) * t.GROW];
            const int e0 = p.ent_base[(size_t)w * (NBLK + 1) + c];
            int k = 0, e = 0, n_a = 0;
            for (int ph = 0; ph < 2; ++ph) {
                int last = ph * 16;
                for (int j = ph * 16; j < std::min(ph * 16 + (ph ? 8 : 16), JW); ++j) {
                    const int L = p.len[((size_t)w * NBLK + c) * JW + j];
                    for (int s = 0; s < L; ++s, ++k, ++e) {
                        for (int q = 0; q < 4; ++q) {
                            const size_t en = (size_t)e0 + (size_t)e * 4 + q;
                            t.off[(((size_t)w * NBLK + c) * 4 + q) * t.NSP + k] = p.gene[en] < 0 ? (unsigned short)pad_off : p.off[en];
                        }
                        g[8 + k] = (unsigned char)j;
                        last = j;
                    }
                }
                for (; k & 7; ++k) g[8 + k] = (unsigned char)last;      // padding steps: weight 0.0 into a group of this phase
                if (ph == 0) n_a = k;
            }
            if (e0 + e * 4 != p.ent_base[(size_t)w * (NBLK + 1) + c + 1]) return false;   // the entry stream is exactly these steps
            g[0] = (unsigned char)(n_a & 0xff); g[1] = (unsigned char)(n_a >> 8);          // steps of phase A (groups 0..15)
            g[2] = (unsigned char)(k & 0xff);   g[3] = (unsigned char)(k >> 8);            // all steps
        }
    return true;
}

}  // namespace fdx
