// The registry of libfdx's runtime switches (fdx_env.h).  Each selects an alternative, TESTED kernel path or a diagnostic - none
// is a CPU fallback (the reference has no switches).
#include "fdx_env.h"

#include <cstring>
#include <mutex>
#include <string>

#include "fdx_internal.h"

namespace fdx {
namespace {

struct Switch {
    const char* name;
    const char* what;
    bool set;
    std::string value;
};

Switch g_switches[] = {
    {"FDX_TRACE_HOST", "host time between the marked points of a fit / a graph build (stderr)", false, {}},
    {"FDX_DEBUG", "one-line reports of the routes taken (leverage route, ELL rebuilds)", false, {}},
    {"FDX_NO_LOG_TABLE", "log1p without the per-row table of the 64 small counts (bit-identical: tests)", false, {}},
    {"FDX_GRAPH_WCAP", "ELL width bound of a graph build: forces the 'bound too small' rebuild / remedy (tests)", false, {}},
    {"FDX_GRAPH_WCAP_RANK", "... on this rank of a shard plan only", false, {}},
    {"FDX_NO_TILED", "global-gather sweep instead of the LDS-tiled one", false, {}},
    {"FDX_NO_INIT_SWEEP", "first sweep reads a written start vector instead of the constant 1/K", false, {}},
    {"FDX_SPLIT_MIN_TILES", "tiles from which a shard sweeps boundary and interior separately", false, {}},
    {"FDX_NO_OVERLAP", "sharded loop: one sweep launch per iteration, halo on the compute stream", false, {}},
    {"FDX_NO_FUSED", "two-kernel sketch -> H (scatter + contraction) instead of the tile kernel", false, {}},
    {"FDX_NO_TILE_WIDE", "no wide tile kernel (K > 32 / d > 704): the two-kernel path takes those shapes", false, {}},
    {"FDX_TILE_LOGV", "0: float64 log1p chain for float32 rows in the tile kernel", false, {}},
    {"FDX_TILE_CFG", "wave split of the tile kernel: 12 (12 + 4), 16 (16 + 0), 8 (8 + 2)", false, {}},
    {"FDX_SKETCH_GATHER", "gather form of the row sketch kernel", false, {}},
    {"FDX_GRAPH_SORT", "Morton order by a radix sort instead of by counting", false, {}},
    {"FDX_GRAPH_SYNC", "graph build completed inside the call (no deferred counts)", false, {}},
    {"FDX_GRAPH_TWO_ELL_KERNELS", "fill_ell + tile_halo as two kernels", false, {}},
    {"FDX_NO_FUSED_PACK", "sharded loop: halo_pack_kernel instead of the sweep writing the send staging", false, {}},
    {"FDX_LEV_ONE_WG", "leverage scores by the single-workgroup route", false, {}},
    {"FDX_KDTREE_HOST_QUERIES", "tie remedy: cKDTree queries on host threads instead of kd_query_kernel", false, {}},
    {"FDX_KDTREE_THREADS", "host threads of the restated cKDTree (build forks, host queries)", false, {}},
    {"FDX_KDTREE_PAR_DEPTH", "fork depth of the restated cKDTree build", false, {}},
    {"FDX_NO_PLAN_CACHE", "sketch plans / tile schedules rebuilt every fit (bench.py: cold_ms)", false, {}},
    {"FDX_NO_SIDE_STREAM", "everything on the caller's stream", false, {}},
    {"FDX_CSR_KEEP_CAP", "entries per wave of the fused CSR sketch's keep buffer (tests: rows that overflow it)", false, {}},
};
constexpr int kSwitches = sizeof(g_switches) / sizeof(g_switches[0]);
std::once_flag g_once;
std::mutex g_mu;

void load_all() {
    for (Switch& s : g_switches) {
        const char* v = getenv(s.name);
        s.set = v != nullptr;
        s.value = v ? v : "";
    }
}

}  // namespace

const char* env(const char* name) {
    std::call_once(g_once, load_all);
    for (const Switch& s : g_switches)
        if (std::strcmp(s.name, name) == 0) return s.set ? s.value.c_str() : nullptr;
    return nullptr;          // not a runtime switch (tests/test_host.py checks the sources against the registry)
}

void env_reload() {
    std::call_once(g_once, load_all);
    std::lock_guard<std::mutex> lk(g_mu);
    load_all();
}

}  // namespace fdx

extern "C" int fdx_env_reload(void) {
    fdx::env_reload();
    return 0;
}

// name / description of runtime switch i (NULL past the end): lets the tests and the docs list the registry
extern "C" const char* fdx_env_switch(int32_t i, const char** what_out) {
    if (i < 0 || i >= fdx::kSwitches) return nullptr;
    if (what_out) *what_out = fdx::g_switches[i].what;
    return fdx::g_switches[i].name;
}
