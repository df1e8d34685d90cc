// Host-only exerciser of the schedule builders (tile_plan.cpp) for the sanitizer build: `make -C flashdeconv_amd/csrc asan-host`
// compiles this file and tile_plan.cpp with g++ -fsanitize=address,undefined and runs it.  No HIP, no GPU.
// (GPU AddressSanitizer is not available on the build pool; the pure-host parts of the library are what can be sanitised.)
#include <cstdio>
#include <cstdlib>
#include <random>
#include <vector>

#include "../tile_plan.h"

using namespace fdx;

static int check_tile(int G, int d, int NW, int JW, int GB, unsigned seed) {
    std::mt19937 rng(seed);
    std::vector<int> bucket((size_t)G);
    std::vector<double> w((size_t)G);
    for (int g = 0; g < G; ++g) {
        bucket[(size_t)g] = (rng() % 41 == 0) ? -1 : (int)(rng() % (unsigned)d);
        w[(size_t)g] = (rng() & 1) ? 1.0 + (rng() % 100) * 0.01 : -1.0 - (rng() % 100) * 0.01;
    }
    TilePlanHost p;
    if (!build_tile_plan(bucket.data(), w.data(), G, d, NW, JW, GB, &p)) return d > 4 * NW * JW ? 0 : 1;
    // replay: every gene of Omega visited exactly once, in its bucket, inside its block
    std::vector<int> seen((size_t)G, 0);
    for (int wv = 0; wv < NW; ++wv)
        for (int c = 0; c < p.NBLK; ++c) {
            int e = p.ent_base[(size_t)wv * (p.NBLK + 1) + c];
            for (int j = 0; j < JW; ++j)
                for (int t = 0; t < p.len[((size_t)wv * p.NBLK + c) * JW + j]; ++t)
                    for (int q = 0; q < 4; ++q, ++e) {
                        const int g = p.gene[(size_t)e];
                        if (g < 0) continue;
                        if (g / GB != c || bucket[(size_t)g] != p.slot_bucket[((size_t)wv * JW + j) * 4 + q] || p.w[(size_t)e] != w[(size_t)g] ||
                            p.off[(size_t)e] != g - c * GB)
                            return 2;
                        ++seen[(size_t)g];
                    }
            if (e != p.ent_base[(size_t)wv * (p.NBLK + 1) + c + 1]) return 3;
        }
    for (int g = 0; g < G; ++g)
        if (seen[(size_t)g] != (bucket[(size_t)g] >= 0 ? 1 : 0)) return 4;
    return 0;
}

int main() {
    int bad = 0;
    const int shapes[][5] = {{2000, 512, 16, 8, 768}, {2000, 512, 12, 11, 1024}, {5000, 1024, 12, 22, 736}, {5003, 1024, 8, 32, 512},
                             {300, 64, 16, 8, 512}, {40, 7, 16, 8, 256}, {1001, 500, 8, 16, 256}, {2000, 600, 12, 11, 1024}};
    for (const auto& s : shapes)
        for (unsigned seed = 1; seed <= 3; ++seed) {
            const int rc = check_tile(s[0], s[1], s[2], s[3], s[4], seed);
            if (rc) { std::printf("tile plan G=%d d=%d NW=%d JW=%d GB=%d seed=%u: error %d\n", s[0], s[1], s[2], s[3], s[4], seed, rc); ++bad; }
        }
    std::printf(bad ? "FAILED\n" : "plan builders: ok under the sanitizers\n");
    return bad ? 1 : 0;
}
