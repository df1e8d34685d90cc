// Native spot-sharded BCD solve: the per-iteration loop of the multi-GPU path in C++, on RCCL directly.
//
// The reference has no distributed code; what makes sharding legal is that the sweep is Jacobi across spots
// (flashdeconv/core/solver.py:157-166 reads only the previous iterate), so any partition gives the unsharded bits.
// Rank r owns one contiguous range of the Morton order plus a read-only halo (fdx_graph_localize).  Per iteration:
//   1. sweep the BOUNDARY tiles (those holding a row some peer needs),
//   2. pack their rows per peer (one kernel), send / receive them with grouped ncclSend / ncclRecv on the communication
//      stream (xGMI is point-to-point: a few hundred KB per peer, no bulk collective), unpack into the halo columns,
//   3. meanwhile sweep the INTERIOR tiles on the compute stream,
//   4. ncclAllReduce(max) of the iteration's 128 convergence slots, so the next sweep's on-device stopping test sees the
//      global statistics and every rank takes the same decision (core/solver.py:395-397, 407-413).
// RCCL is loaded with dlopen on first use: libfdx.so itself links nothing but the HIP runtime.
//
// Transports: RCCL (one process per GPU), and an in-process one (several host threads of one process act as ranks on ONE
// GPU, device copies through a shared mailbox) that lets the tests run this very loop without N GPUs.
#include "fdx_env.h"
#include <dlfcn.h>
#include <rccl/rccl.h>

#include <algorithm>
#include <chrono>
#include <cmath>
#include <condition_variable>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <vector>

#include "fdx_graph.h"
#include "fdx_internal.h"
#include "fdx_kernels.h"
#include "graph_build.h"
#include "prepare.h"
#include "solver.h"

using namespace fdx;

namespace {

// ---- RCCL entry points, resolved at run time ------------------------------------------------------------------------
struct RcclApi {
    decltype(&ncclGetUniqueId) GetUniqueId = nullptr;
    decltype(&ncclCommInitRank) CommInitRank = nullptr;
    decltype(&ncclCommDestroy) CommDestroy = nullptr;
    decltype(&ncclGroupStart) GroupStart = nullptr;
    decltype(&ncclGroupEnd) GroupEnd = nullptr;
    decltype(&ncclSend) Send = nullptr;
    decltype(&ncclRecv) Recv = nullptr;
    decltype(&ncclAllReduce) AllReduce = nullptr;
    decltype(&ncclGetErrorString) GetErrorString = nullptr;
    decltype(&ncclCommCount) CommCount = nullptr;
    bool ok = false;
};

const RcclApi* rccl() {
    static RcclApi api;
    static std::once_flag once;
    std::call_once(once, [] {
        void* h = nullptr;
        for (const char* name : {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"}) {
            h = dlopen(name, RTLD_NOW | RTLD_GLOBAL);
            if (h) break;
        }
        if (!h) return;
#define FDX_SYM(f) api.f = reinterpret_cast<decltype(api.f)>(dlsym(h, "nccl" #f))
        FDX_SYM(GetUniqueId); FDX_SYM(CommInitRank); FDX_SYM(CommDestroy); FDX_SYM(GroupStart); FDX_SYM(GroupEnd);
        FDX_SYM(Send); FDX_SYM(Recv); FDX_SYM(AllReduce); FDX_SYM(GetErrorString); FDX_SYM(CommCount);
#undef FDX_SYM
        api.ok = api.GetUniqueId && api.CommInitRank && api.CommDestroy && api.GroupStart && api.GroupEnd && api.Send && api.Recv &&
                 api.AllReduce && api.GetErrorString;
    });
    return api.ok ? &api : nullptr;
}

#define FDX_NCCL(expr)                                                                                       \
    do {                                                                                                     \
        ncclResult_t _r = (expr);                                                                            \
        if (_r != ncclSuccess) return ::fdx::fail(FDX_ERR_HIP, std::string(#expr) + ": " + rccl()->GetErrorString(_r)); \
    } while (0)

// ---- in-process world (tests): W host threads, one GPU ----------------------------------------------------------------
struct LocalWorld {
    int W = 0;
    std::mutex mu;
    std::condition_variable cv;
    int arrived = 0;
    long long generation = 0;
    std::vector<const void*> ptr;                   // mailbox: one pointer per rank
    std::vector<const void*> sptr;                  // mailbox: each rank's convergence slots of the iteration (or nullptr)
    std::vector<std::vector<long long>> off;        // mailbox: per rank, element offsets per destination (W + 1)
    std::vector<std::vector<unsigned long long>> host;   // mailbox for the reductions
    bool aborted = false;                           // a rank left its loop on an error: nobody waits for it any more
    // false when the world was aborted (by this or another rank): the caller returns an error instead of waiting for ever
    bool barrier() {
        std::unique_lock<std::mutex> lk(mu);
        if (aborted) return false;
        const long long gen = generation;
        if (++arrived == W) {
            arrived = 0;
            ++generation;
            cv.notify_all();
        } else {
            cv.wait(lk, [&] { return generation != gen || aborted; });
        }
        return !aborted;
    }
    void abort() {
        std::lock_guard<std::mutex> lk(mu);
        aborted = true;
        cv.notify_all();
    }
};

}  // namespace

struct fdx_local_world {
    LocalWorld w;
};

struct fdx_comm {
    int rank = 0, world = 1;
    ncclComm_t nccl = nullptr;          // RCCL transport
    LocalWorld* local = nullptr;        // in-process transport
    bool loopback = false;              // one rank of `world` ALONE (fdx_comm_init_loopback): the exchange is a device copy of its own
                                        // staging, the all-reduce a no-op - the loop's kernels, launches and read-backs without peers
    hipStream_t side = nullptr;         // communication stream (halo traffic beside the interior sweep)
    hipEvent_t ev_packed = nullptr, ev_halo = nullptr;
};

namespace {

// ---- halo pack / unpack ---------------------------------------------------------------------------------------------------
// Send staging: for peer r the (K, cnt_r) block starts at K * send_off[r]; recv staging likewise with recv_off.
// One thread per (row, type): a rank sends a few thousand rows - a thread per row walking its K planes (K dependent scattered
// loads) was 18 us of a 36 us sweep at 125k spots; this form is a handful of microseconds.
__global__ void halo_pack_kernel(const double* __restrict__ beta, long long ld, int K, const int* __restrict__ send_idx,
                                 const int* __restrict__ send_off, int world, int total, double* __restrict__ out) {
    const long long t = blockIdx.x * (long long)blockDim.x + threadIdx.x;
    if (t >= (long long)total * K) return;
    const int k = (int)(t / total), j = (int)(t - (long long)k * total);
    int r = 0;
    while (r + 1 < world && j >= send_off[r + 1]) ++r;
    const int base = send_off[r], cnt = send_off[r + 1] - base;
    out[(size_t)K * base + (size_t)k * cnt + (j - base)] = beta[(size_t)k * ld + send_idx[j]];
}

// the send lists seen from the rows (fdx_graph::send_head / send_ent): one thread per entry of send_idx
__global__ void send_entries_kernel(const int* __restrict__ send_idx, const int* __restrict__ send_off, int world, int total,
                                    int* __restrict__ head, int4* __restrict__ ent) {
    const int e = blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= total) return;
    int r = 0;
    while (r + 1 < world && e >= send_off[r + 1]) ++r;
    const int base = send_off[r];
    const int prev = atomicExch(&head[send_idx[e]], e + 1);           // a row goes to at most a few peers: a short chain, any order
    ent[e] = make_int4(base, send_off[r + 1] - base, e - base, prev);
}

__global__ void halo_unpack_kernel(double* __restrict__ beta, long long ld, int K, long long n_own, const int* __restrict__ recv_off,
                                   int world, int total, const double* __restrict__ in, unsigned long long* __restrict__ stat_local,
                                   const unsigned long long* __restrict__ stat_recv, int rank) {
    const long long t = blockIdx.x * (long long)blockDim.x + threadIdx.x;
    if (stat_local && t < 128) {        // the peers' convergence slots of this iteration, folded into ours (bit patterns of doubles >= 0)
        unsigned long long v = stat_local[t];
        for (int r = 0; r < world; ++r)
            if (r != rank) v = max(v, stat_recv[(size_t)r * 128 + t]);
        stat_local[t] = v;
    }
    if (t >= (long long)total * K) return;
    const int k = (int)(t / total), j = (int)(t - (long long)k * total);
    int r = 0;
    while (r + 1 < world && j >= recv_off[r + 1]) ++r;
    const int base = recv_off[r], cnt = recv_off[r + 1] - base;
    beta[(size_t)k * ld + n_own + j] = in[(size_t)K * base + (size_t)k * cnt + (j - base)];
}

// tiles (256 rows) that hold a row some peer needs, and the others
int build_tile_lists(const fdx_graph& g, hipStream_t st) {
    if (g.n_tiles_boundary >= 0) return 0;
    const int total = g.send_off.empty() ? 0 : g.send_off.back();
    std::vector<int> idx((size_t)total);
    if (total) {
        FDX_HIP(hipMemcpyAsync(idx.data(), g.send_idx.p, (size_t)total * 4, hipMemcpyDeviceToHost, st));
        FDX_HIP(hipStreamSynchronize(st));
    }
    std::vector<char> is_b((size_t)std::max(g.n_tiles, 1), 0);
    for (int i : idx)
        if (i >= 0 && i / 256 < g.n_tiles) is_b[(size_t)(i / 256)] = 1;
    std::vector<int> tb, ti;
    for (int t = 0; t < g.n_tiles; ++t) (is_b[(size_t)t] ? tb : ti).push_back(t);
    FDX_TRY(g.tiles_boundary.alloc(std::max<size_t>(tb.size(), 1) * 4));
    FDX_TRY(g.tiles_interior.alloc(std::max<size_t>(ti.size(), 1) * 4));
    if (!tb.empty()) FDX_HIP(hipMemcpyAsync(g.tiles_boundary.p, tb.data(), tb.size() * 4, hipMemcpyHostToDevice, st));
    if (!ti.empty()) FDX_HIP(hipMemcpyAsync(g.tiles_interior.p, ti.data(), ti.size() * 4, hipMemcpyHostToDevice, st));
    FDX_HIP(hipStreamSynchronize(st));
    g.n_tiles_boundary = (int)tb.size();
    g.n_tiles_interior = (int)ti.size();
    return 0;
}

// ---- transports ---------------------------------------------------------------------------------------------------------------
// send / recv staging blocks per peer, all on stream `st`
// stat_send != NULL: every rank also hands its 128 convergence slots of the iteration to EVERY peer inside the same group
// (stat_recv: W blocks of 128 words, block r from rank r) - the halo unpack kernel folds them, and the separate all-reduce(max) of
// the iteration is gone: one RCCL group per iteration instead of a group and a collective
int exchange(fdx_comm* c, const double* send, const std::vector<int>& send_off, double* recv, const std::vector<int>& recv_off, int K,
             hipStream_t st, const unsigned long long* stat_send = nullptr, unsigned long long* stat_recv = nullptr) {
    const int W = c->world;
    if (c->loopback) {       // what arrives is this rank's own staging (as many values as both sides have): timing only
        const long long ns = send_off.back(), nr = recv_off.back();
        const long long m = std::min(ns, nr);
        if (m > 0) FDX_HIP(hipMemcpyAsync(recv, send, (size_t)K * m * 8, hipMemcpyDeviceToDevice, st));
        return 0;
    }
    if (c->nccl) {
        const RcclApi* api = rccl();
        FDX_NCCL(api->GroupStart());
        for (int r = 0; r < W; ++r) {
            const int nr = recv_off[(size_t)r + 1] - recv_off[(size_t)r];
            if (r != c->rank && nr > 0)
                FDX_NCCL(api->Recv(recv + (size_t)K * recv_off[(size_t)r], (size_t)K * nr, ncclDouble, r, c->nccl, st));
        }
        for (int r = 0; r < W; ++r) {
            const int ns = send_off[(size_t)r + 1] - send_off[(size_t)r];
            if (r != c->rank && ns > 0)
                FDX_NCCL(api->Send(send + (size_t)K * send_off[(size_t)r], (size_t)K * ns, ncclDouble, r, c->nccl, st));
        }
        if (stat_send && stat_recv) {
            for (int r = 0; r < W; ++r)
                if (r != c->rank) FDX_NCCL(api->Recv(stat_recv + (size_t)r * 128, 128, ncclUint64, r, c->nccl, st));
            for (int r = 0; r < W; ++r)
                if (r != c->rank) FDX_NCCL(api->Send(stat_send, 128, ncclUint64, r, c->nccl, st));
        }
        FDX_NCCL(api->GroupEnd());
        return 0;
    }
    if (c->local) {
        LocalWorld& w = *c->local;
        FDX_HIP(hipStreamSynchronize(st));                        // my send staging is complete
        {
            std::lock_guard<std::mutex> lk(w.mu);
            w.ptr[(size_t)c->rank] = send;
            w.sptr[(size_t)c->rank] = stat_send;
            w.off[(size_t)c->rank].assign(send_off.begin(), send_off.end());
        }
        FDX_REQUIRE(w.barrier(), "sharded solve: another rank of this in-process world failed; the world stays aborted - create a new fdx_local_world");
        for (int q = 0; q < W; ++q) {
            const int nr = recv_off[(size_t)q + 1] - recv_off[(size_t)q];
            if (q == c->rank || nr == 0) continue;
            const double* src = static_cast<const double*>(w.ptr[(size_t)q]) + (size_t)K * w.off[(size_t)q][(size_t)c->rank];
            const long long ns = w.off[(size_t)q][(size_t)c->rank + 1] - w.off[(size_t)q][(size_t)c->rank];
            FDX_REQUIRE(ns == nr, "sharded solve: send and receive lists of two ranks disagree");
            FDX_HIP(hipMemcpyAsync(recv + (size_t)K * recv_off[(size_t)q], src, (size_t)K * nr * 8, hipMemcpyDeviceToDevice, st));
        }
        if (stat_send && stat_recv)
            for (int q = 0; q < W; ++q) {
                if (q == c->rank) continue;
                FDX_REQUIRE(w.sptr[(size_t)q] != nullptr, "sharded solve: the ranks disagree on how the convergence slots travel");
                FDX_HIP(hipMemcpyAsync(stat_recv + (size_t)q * 128, w.sptr[(size_t)q], 128 * 8, hipMemcpyDeviceToDevice, st));
            }
        FDX_HIP(hipStreamSynchronize(st));
        FDX_REQUIRE(w.barrier(), "sharded solve: another rank of this in-process world failed; the world stays aborted - create a new fdx_local_world");   // nobody overwrites its staging before all have copied
        return 0;
    }
    return 0;
}

// element-wise max / sum over the ranks of `count` 64-bit words on the device
int allreduce(fdx_comm* c, void* buf, int count, bool is_max, hipStream_t st) {
    if ((c->world == 1 && !c->nccl) || c->loopback) return 0;
    if (c->nccl) {
        const RcclApi* api = rccl();
        if (is_max) FDX_NCCL(api->AllReduce(buf, buf, (size_t)count, ncclUint64, ncclMax, c->nccl, st));
        else FDX_NCCL(api->AllReduce(buf, buf, (size_t)count, ncclDouble, ncclSum, c->nccl, st));
        return 0;
    }
    if (c->local) {
        LocalWorld& w = *c->local;
        std::vector<unsigned long long> mine((size_t)count);
        FDX_HIP(hipMemcpyAsync(mine.data(), buf, (size_t)count * 8, hipMemcpyDeviceToHost, st));
        FDX_HIP(hipStreamSynchronize(st));
        {
            std::lock_guard<std::mutex> lk(w.mu);
            w.host[(size_t)c->rank] = mine;
        }
        FDX_REQUIRE(w.barrier(), "sharded solve: another rank of this in-process world failed; the world stays aborted - create a new fdx_local_world");
        std::vector<unsigned long long> out((size_t)count, 0ULL);
        if (is_max) {
            for (int q = 0; q < w.W; ++q)
                for (int i = 0; i < count; ++i) out[(size_t)i] = std::max(out[(size_t)i], w.host[(size_t)q][(size_t)i]);
        } else {
            std::vector<double> acc((size_t)count, 0.0);
            for (int q = 0; q < w.W; ++q)                              // rank order: every rank gets the same bits
                for (int i = 0; i < count; ++i) {
                    double v;
                    std::memcpy(&v, &w.host[(size_t)q][(size_t)i], 8);
                    acc[(size_t)i] += v;
                }
            std::memcpy(out.data(), acc.data(), (size_t)count * 8);
        }
        FDX_REQUIRE(w.barrier(), "sharded solve: another rank of this in-process world failed; the world stays aborted - create a new fdx_local_world");   // everybody has read the mailbox
        FDX_HIP(hipMemcpyAsync(buf, out.data(), (size_t)count * 8, hipMemcpyHostToDevice, st));
        FDX_HIP(hipStreamSynchronize(st));
        return 0;
    }
    return 0;
}

int comm_streams(fdx_comm* c) {
    if (c->side) return 0;
    int lo = 0, hi = 0;
    (void)hipDeviceGetStreamPriorityRange(&lo, &hi);
    if (hipStreamCreateWithPriority(&c->side, hipStreamNonBlocking, hi) != hipSuccess)
        FDX_HIP(hipStreamCreateWithFlags(&c->side, hipStreamNonBlocking));
    FDX_HIP(hipEventCreateWithFlags(&c->ev_packed, hipEventDisableTiming));
    FDX_HIP(hipEventCreateWithFlags(&c->ev_halo, hipEventDisableTiming));
    return 0;
}

}  // namespace

extern "C" {

int fdx_comm_unique_id(void* id_out_128) {
    FDX_REQUIRE(id_out_128 != nullptr, "fdx_comm_unique_id: null output");
    const RcclApi* api = rccl();
    FDX_REQUIRE(api != nullptr, "fdx_comm_unique_id: librccl.so could not be loaded");
    static_assert(sizeof(ncclUniqueId) == 128, "ncclUniqueId is 128 bytes");
    ncclUniqueId id;
    FDX_NCCL(api->GetUniqueId(&id));
    std::memcpy(id_out_128, &id, sizeof(id));
    return 0;
}

int fdx_comm_init(const void* id_128, int32_t rank, int32_t world, fdx_comm** out) {
    FDX_REQUIRE(id_128 && out && world >= 1 && rank >= 0 && rank < world, "fdx_comm_init: bad arguments");
    const RcclApi* api = rccl();
    FDX_REQUIRE(api != nullptr, "fdx_comm_init: librccl.so could not be loaded");
    ncclUniqueId id;
    std::memcpy(&id, id_128, sizeof(id));
    auto* c = new fdx_comm();
    c->rank = rank;
    c->world = world;
    const ncclResult_t r = api->CommInitRank(&c->nccl, world, id, rank);
    if (r != ncclSuccess) {
        delete c;
        return fail(FDX_ERR_HIP, std::string("ncclCommInitRank: ") + api->GetErrorString(r));
    }
    *out = c;
    return 0;
}

int fdx_local_world_create(int32_t world, fdx_local_world** out) {
    FDX_REQUIRE(out && world >= 1, "fdx_local_world_create: bad arguments");
    auto* w = new fdx_local_world();
    w->w.W = world;
    w->w.ptr.assign((size_t)world, nullptr);
    w->w.sptr.assign((size_t)world, nullptr);
    w->w.off.assign((size_t)world, {});
    w->w.host.assign((size_t)world, {});
    *out = w;
    return 0;
}

int fdx_local_world_destroy(fdx_local_world* w) {
    delete w;
    return 0;
}

int fdx_comm_init_local(fdx_local_world* w, int32_t rank, fdx_comm** out) {
    FDX_REQUIRE(w && out && rank >= 0 && rank < w->w.W, "fdx_comm_init_local: bad arguments");
    auto* c = new fdx_comm();
    c->rank = rank;
    c->world = w->w.W;
    c->local = &w->w;
    *out = c;
    return 0;
}

int fdx_comm_init_loopback(int32_t rank, int32_t world, fdx_comm** out) {
    FDX_REQUIRE(out && world >= 1 && rank >= 0 && rank < world, "fdx_comm_init_loopback: bad arguments");
    auto* c = new fdx_comm();
    c->rank = rank;
    c->world = world;
    c->loopback = true;
    *out = c;
    return 0;
}

int fdx_comm_destroy(fdx_comm* c) {
    if (!c) return 0;
    if (c->side) (void)hipStreamSynchronize(c->side);
    if (c->nccl && rccl()) (void)rccl()->CommDestroy(c->nccl);
    if (c->ev_packed) (void)hipEventDestroy(c->ev_packed);
    if (c->ev_halo) (void)hipEventDestroy(c->ev_halo);
    if (c->side) (void)hipStreamDestroy(c->side);
    delete c;
    return 0;
}

int fdx_comm_info(const fdx_comm* c, int32_t* rank, int32_t* world) {
    FDX_REQUIRE(c != nullptr, "fdx_comm_info: null communicator");
    if (rank) *rank = c->rank;
    if (world) *world = c->world;
    return 0;
}

int fdx_comm_rccl_count(const fdx_comm* c, int32_t* count) {
    FDX_REQUIRE(c != nullptr && count != nullptr, "fdx_comm_rccl_count: null argument");
    *count = 0;                                        // not an RCCL communicator (in-process / loopback transport)
    if (c->nccl && rccl() && rccl()->CommCount) {
        int n = 0;
        FDX_NCCL(rccl()->CommCount(c->nccl, &n));
        *count = n;
    }
    return 0;
}

int fdx_comm_allreduce_sum_dev(fdx_comm* c, double* buf_dev, int32_t count, void* stream) {
    FDX_REQUIRE(c && buf_dev && count > 0, "fdx_comm_allreduce_sum_dev: bad arguments");
    return allreduce(c, buf_dev, count, false, (hipStream_t)stream);
}

int fdx_sharded_solve_dev(fdx_comm* c, const fdx_graph* g, const double* H_dev, int64_t ldh, const double* XtX_dev, int32_t K,
                          double lambda, double rho_eff, double tol, int32_t max_iter, double* beta0_dev, double* beta1_dev,
                          int64_t ld, fdx_solve_info* info, double* rel_changes_out, int32_t* result_buffer, void* stream) {
    return fdx_sharded_solve_padded_dev(c, g, H_dev, ldh, XtX_dev, K, K, lambda, rho_eff, tol, max_iter, beta0_dev, beta1_dev, ld, info,
                                        rel_changes_out, result_buffer, stream);
}

int32_t fdx_solver_padded_k(int32_t K) { return fdx::solver_padded_K(K); }

static int sharded_solve_impl(fdx_comm* c, const fdx_graph* g, const double* H_dev, int64_t ldh, const double* XtX_dev, int32_t K,
                              int32_t K_real, double lambda, double rho_eff, double tol, int32_t max_iter, double* beta0_dev,
                              double* beta1_dev, int64_t ld, fdx_solve_info* info, double* rel_changes_out, int32_t* result_buffer,
                              void* stream);

int fdx_sharded_solve_padded_dev(fdx_comm* c, const fdx_graph* g, const double* H_dev, int64_t ldh, const double* XtX_dev, int32_t K,
                                 int32_t K_real, double lambda, double rho_eff, double tol, int32_t max_iter, double* beta0_dev,
                                 double* beta1_dev, int64_t ld, fdx_solve_info* info, double* rel_changes_out, int32_t* result_buffer,
                                 void* stream) {
    return sharded_solve_impl(c, g, H_dev, ldh, XtX_dev, K, K_real, lambda, rho_eff, tol, max_iter, beta0_dev, beta1_dev, ld, info,
                              rel_changes_out, result_buffer, stream);
}

// One rank's whole fit behind a queued plan: sketch -> H of the own rows, the plan's counts all-reduced while the sketch runs,
// lambda, the iteration loop, objective and export - one call, the device never waiting for the host between its stages
// (the separate calls cost a 125k-spot rank ~0.35 ms of idle device: 1.25 -> 1.6 ms).  Reference: core/deconv.py:326-398 for the
// stages, core/solver.py:157-166 for the sharding.
// FDX_TRACE_HOST=1: host clock at the steps of a shard's fit (stderr)
static void shard_trace(const char* what) {
    static const bool on = fdx::env("FDX_TRACE_HOST") != nullptr;
    if (!on) return;
    static auto t_prev = std::chrono::steady_clock::now();
    const auto t = std::chrono::steady_clock::now();
    std::fprintf(stderr, "[fdx-host] +%7.1f us  shard fit: %s\n", std::chrono::duration<double, std::micro>(t - t_prev).count(), what);
    t_prev = t;
}

int fdx_shard_fit_dev(fdx_comm* c, const fdx_graph* g, const void* Y_dev, int32_t y_dtype, int64_t n_own, int32_t G, int64_t ldy,
                      const double* X, int32_t K, const int32_t* bucket, const double* weight_y, const double* weight_x,
                      const fdx_shard_fit_params* prm, double* beta_out_dev, double* prop_out_dev, double* rel_changes_out,
                      fdx_shard_fit_info* info, void* stream) {
    FDX_REQUIRE(c && g && prm && info && X && bucket && weight_y && weight_x, "fdx_shard_fit_dev: null argument");
    FDX_REQUIRE(n_own == g->n && n_own > 0, "fdx_shard_fit_dev: the matrix must hold the graph's own rows (at least one)");
    FDX_REQUIRE(K >= 1 && K <= FDX_MAX_K_PAD, "fdx_shard_fit_dev: 1 to 96 cell types (more: the stepwise calls)");
    FDX_REQUIRE(prm->sketch_dim > 0 && prm->max_iter >= 0 && prm->n_total_spots >= n_own, "fdx_shard_fit_dev: bad parameters");
    std::memset(info, 0, sizeof(*info));
    shard_trace("entry");
    hipStream_t st = (hipStream_t)stream;
    PoolStream pool_stream(st);
    FDX_TRY(comm_streams(c));
    const int W = c->world;
    const int KP = solver_padded_K(K);
    const long long ldh = round_up(n_own + 1, 64);
    PrepareJob job;
    DevBuf dH, dGp, dB0, dB1, dCnt, objp, objo, dFin;
    // on any return every stream this call used is idle before the buffers above go back to the pool
    struct Drain { PrepareJob* j; fdx_comm* c; hipStream_t s; ~Drain() { if (j->side) (void)hipStreamSynchronize(j->side); if (c->side) (void)hipStreamSynchronize(c->side); (void)hipStreamSynchronize(s); } } drain{&job, c, st};
    FDX_TRY(dH.alloc((size_t)KP * ldh * sizeof(double)));
    FDX_TRY(solver_zero_pad(dH.as<double>(), ldh, n_own, K, st));                       // the sketch writes columns [0, n_own) of the real types
    if (KP != K) FDX_HIP(hipMemsetAsync(dH.as<double>() + (size_t)K * ldh, 0, (size_t)(KP - K) * ldh * sizeof(double), st));
    double* Gh = (double*)pinned_scratch(4, (size_t)K * K * sizeof(double) + 64);
    FDX_REQUIRE(Gh != nullptr, "fdx_shard_fit_dev: pinned host buffer");
    // A shard build may still be on its way (graph_shard_knn: second phase queued by the helper thread on the plan stream): the
    // sketch of the own rows does not need the graph and runs beside it.
    // (FDX_SKETCH_RESERVE: compute units the sketch's persistent workgroups leave to that build; measured at 16 / 32 / 64 on a
    // 125k-spot shard it only slowed both down - the build is a chain of dependent launches, not a matter of free units.)
    int reserve = 0;
    if (g->shard_pending) if (const char* e = fdx::exp_env("FDX_SKETCH_RESERVE")) reserve = atoi(e);
    const int reserve_prev = tile_sketch_reserve_cus(reserve);
    const int prc = prepare_queue(&job, Y_dev, y_dtype, n_own, G, ldy, nullptr, X, K, bucket, weight_y, weight_x, prm->sketch_dim,
                                  prm->mode_y, prm->mode_x, dH.as<double>(), ldh, Gh, st, prm->X_dev);
    tile_sketch_reserve_cus(reserve_prev);
    FDX_TRY(prc);
    shard_trace("X side + sketch queued");
    FDX_TRY(graph_shard_join(g));
    shard_trace("rest of the plan queued");
    const double* XtX_dev = job.dG.as<double>();
    if (KP != K) {
        PoolStream pool_xs(job.side ? job.side : st);
        FDX_TRY(dGp.alloc((size_t)KP * KP * sizeof(double)));
        FDX_TRY(solver_pad_square(job.dG.as<double>(), K, dGp.as<double>(), KP, job.side ? job.side : st));
        if (job.side) {                                                     // behind the first event: the sweeps wait for this one too
            FDX_HIP(hipEventRecord(c->ev_packed, job.side));
            FDX_HIP(hipStreamWaitEvent(st, c->ev_packed, 0));
        }
        XtX_dev = dGp.as<double>();
    }

    // ---- the plan's counts: all-reduced on the communication stream as soon as the plan has produced them, beside the sketch
    double* cnt_h = (double*)pinned_scratch(5, 8 * sizeof(double));
    FDX_REQUIRE(cnt_h != nullptr, "fdx_shard_fit_dev: pinned host buffer");
    double tot[4] = {0, 0, 0, 0};
    // A JOB-level decision (the caller's nnz_total < 0: the plan's counts are still to be summed), identical on every rank - NOT
    // "does this rank's graph carry device counts": after an overflow remedy one rank's graph is a stepwise rebuild without them,
    // and a rank that skipped the all-reduce its peers issued left the communicator mismatched (hang).  A rank without device
    // counts contributes its host counts.
    const bool job_reduces = prm->nnz_total < 0 && W > 1 && !c->loopback;
    if (job_reduces) {
        FDX_TRY(dCnt.alloc(4 * sizeof(double)));
        if (g->counts_dev.p) {
            if (g->shard_pending) FDX_HIP(hipStreamWaitEvent(c->side, g->meta_event, 0));
            FDX_HIP(hipMemcpyAsync(dCnt.p, g->counts_dev.p, 4 * sizeof(double), hipMemcpyDeviceToDevice, c->side));
        } else {
            FDX_TRY(fdx::graph_meta_sync(g));
            cnt_h[4] = (double)g->nnz; cnt_h[5] = (double)g->knn_ties; cnt_h[6] = (double)g->knn_far; cnt_h[7] = (double)g->shard_overflow;
            FDX_HIP(hipMemcpyAsync(dCnt.p, cnt_h + 4, 4 * sizeof(double), hipMemcpyHostToDevice, c->side));
        }
        FDX_TRY(allreduce(c, dCnt.p, 4, false, c->side));
        FDX_HIP(hipMemcpyAsync(cnt_h, dCnt.p, 4 * sizeof(double), hipMemcpyDeviceToHost, c->side));
        FDX_HIP(hipStreamSynchronize(c->side));
        for (int j = 0; j < 4; ++j) tot[j] = cnt_h[j];
        FDX_TRY(fdx::graph_meta_sync(g));
    } else {
        FDX_TRY(fdx::graph_meta_sync(g));
        tot[0] = (double)g->nnz; tot[1] = (double)g->knn_ties; tot[2] = (double)g->knn_far; tot[3] = (double)g->shard_overflow;
    }
    shard_trace("plan's counts on the host");
    if (prm->nnz_total >= 0) tot[0] = (double)prm->nnz_total;              // the caller knows the job's total (a graph that was not built here)
    info->nnz_total = (int64_t)std::llround(tot[0]);
    info->knn_ties_total = (int64_t)std::llround(tot[1]);
    info->own_nnz = g->nnz;
    info->n_halo = g->n_total - g->n;
    // remedies are the caller's: a far walk anywhere (every rank rebuilds by the list exchange), a bound too small on some rank
    // (that rank rebuilds stepwise; the others wait for nobody - but lambda needs every rank's exact edge count, which an
    // overflowing rank cannot give), ties when the reference's order is asked for
    if (tot[2] != 0.0) info->status = FDX_SHARD_FAR;
    else if (tot[3] != 0.0) info->status = FDX_SHARD_OVERFLOW;
    else if (tot[1] != 0.0 && prm->stop_on_ties) info->status = FDX_SHARD_TIES;
    if (info->status != 0) return 0;                                        // the drain waits for the sketch; nothing was solved
    FDX_REQUIRE(g->send_off.size() == (size_t)W + 1, "fdx_shard_fit_dev: the graph was built for a different number of ranks");

    // ---- lambda, scaled rho (host scalars of the sweeps): XtX has long arrived
    FDX_HIP(hipEventSynchronize(job.evX));
    shard_trace("XtX on the host");
    double diag_mean = 0.0;
    for (int k = 0; k < K; ++k) diag_mean += Gh[(size_t)k * K + k];
    diag_mean /= (double)K;
    double lambda = prm->lambda_spatial;
    if (prm->lambda_auto) lambda = 0.005 * diag_mean / std::max(tot[0] / (double)prm->n_total_spots, 1.0);   // core/spatial.py:181-190
    const double rho_eff = prm->rho_sparsity * diag_mean;                                                       // core/solver.py:359-360
    info->lambda_used = lambda;
    info->rho_effective = rho_eff;

    // ---- the loop, queued behind the sketch
    const long long ld = round_up(g->n_total + 1, 64);
    FDX_TRY(dB0.alloc((size_t)KP * ld * sizeof(double)));
    FDX_TRY(dB1.alloc((size_t)KP * ld * sizeof(double)));
    int32_t which = 0;
    FDX_TRY(sharded_solve_impl(c, g, dH.as<double>(), ldh, XtX_dev, KP, K, lambda, rho_eff, prm->tol, prm->max_iter, dB0.as<double>(),
                               dB1.as<double>(), ld, &info->solve, rel_changes_out, &which, stream));
    const double* beta = which ? dB1.as<double>() : dB0.as<double>();
    shard_trace("loop returned");

    // ---- export beside the objective pass (both only read the final abundances), YtY and the objective's sums in one all-reduce
    FDX_TRY(objp.alloc((size_t)std::max(objective_partials_count(g->n_slices), g->n_tiles) * 4 * sizeof(double)));
    FDX_TRY(dFin.alloc(8 * sizeof(double)));
    if (beta_out_dev || prop_out_dev) {
        FDX_HIP(hipEventRecord(c->ev_packed, st));
        FDX_HIP(hipStreamWaitEvent(c->side, c->ev_packed, 0));
        FDX_TRY(launch_normalize_export(beta, ld, nullptr, (int)n_own, g->n_slices, K, beta_out_dev, prop_out_dev, c->side));
        FDX_HIP(hipEventRecord(c->ev_halo, c->side));
    }
    FDX_TRY(solver_objective_partials(*g, beta, ld, dH.as<double>(), ldh, XtX_dev, KP, objp.as<double>(), dFin.as<double>(), st));
    if (job.evSum) FDX_HIP(hipStreamWaitEvent(st, job.evSum, 0));
    FDX_HIP(hipMemcpyAsync(dFin.as<double>() + 4, job.dSum.p, sizeof(double), hipMemcpyDeviceToDevice, st));
    FDX_TRY(allreduce(c, dFin.p, 5, false, st));
    FDX_HIP(hipMemcpyAsync(cnt_h, dFin.p, 5 * sizeof(double), hipMemcpyDeviceToHost, st));
    if (beta_out_dev || prop_out_dev) FDX_HIP(hipStreamWaitEvent(st, c->ev_halo, 0));
    FDX_HIP(hipStreamSynchronize(st));
    shard_trace("objective on the host");
    info->YtY = cnt_h[4];
    info->solve.final_objective = 0.5 * (cnt_h[4] - 2.0 * cnt_h[0] + cnt_h[1]) + 0.5 * lambda * cnt_h[2] + rho_eff * cnt_h[3];   // core/solver.py:272-284
    return 0;
}

static int sharded_solve_impl(fdx_comm* c, const fdx_graph* g, const double* H_dev, int64_t ldh, const double* XtX_dev, int32_t K,
                              int32_t K_real, double lambda, double rho_eff, double tol, int32_t max_iter, double* beta0_dev,
                              double* beta1_dev, int64_t ld, fdx_solve_info* info, double* rel_changes_out, int32_t* result_buffer,
                              void* stream) {
    FDX_REQUIRE(c && g && H_dev && XtX_dev && beta0_dev && beta1_dev && info && result_buffer, "fdx_sharded_solve_dev: null argument");
    FDX_TRY(fdx::graph_meta_sync(g));
    // 1..64 types and the padded sizes 72 / 80 / 88 / 96 run the tiled register sweeps (boundary tiles first, halo traffic beside the
    // interior tiles); above that the LDS-resident sweep (to ~270 types) or the generic one takes the whole shard in one launch per
    // iteration, the exchange behind it - no tile lists there, nothing to overlap with
    FDX_REQUIRE(K >= 1 && K_real >= 1 && K_real <= K && (sweep_instantiated(K) || K > FDX_MAX_K_FAST),
                "fdx_sharded_solve_dev: K must be 1..64, fdx_solver_padded_k of 65..96 cell types, or above 96");
    FDX_REQUIRE(ld >= g->n_total + 1, "fdx_sharded_solve_dev: ld must cover own + halo + zero row");
    FDX_REQUIRE(max_iter >= 0, "fdx_sharded_solve_dev: max_iter must be >= 0");
    FDX_REQUIRE(g->send_off.size() == (size_t)c->world + 1 && g->recv_off.size() == (size_t)c->world + 1,
                "fdx_sharded_solve_dev: the graph was localized for a different number of ranks");
    hipStream_t st = (hipStream_t)stream;
    PoolStream pool_stream(st);
    std::memset(info, 0, sizeof(*info));
    *result_buffer = 0;
    const int W = c->world;
    const int total_send = g->send_off.back(), total_recv = g->recv_off.back();
    FDX_TRY(comm_streams(c));

    DevBuf stats, send_buf, recv_buf, soff, roff, sweep_scratch, stat_recv;
    size_t scratch_ld = 0;
    if (sweep_uses_lds(K)) {                          // as in solver_run: XtX with its rows padded to 16 for the LDS-resident sweep
        FDX_TRY(sweep_scratch.alloc(sweep_lds_pad_doubles(K) * sizeof(double)));
        FDX_TRY(sweep_lds_prepare(XtX_dev, K, sweep_scratch.as<double>(), (hipStream_t)stream));
    } else if (!sweep_instantiated(K)) {
        scratch_ld = (size_t)g->n_slices * 64;
        FDX_TRY(sweep_scratch.alloc(scratch_ld * 2 * K * sizeof(double)));
    }
    const int iters = std::max<int>(max_iter, 1);
    // the max slots of every iteration and, behind them, the rel_change trace: one block, one fill
    const size_t stats_bytes = (size_t)iters * 128 * 8;
    FDX_TRY(stats.alloc(stats_bytes + (size_t)iters * 8));
    double* const relchg_p = reinterpret_cast<double*>(static_cast<char*>(stats.p) + stats_bytes);
    FDX_TRY(send_buf.alloc((size_t)std::max(total_send, 1) * K * 8));
    FDX_TRY(recv_buf.alloc((size_t)std::max(total_recv, 1) * K * 8));
    FDX_TRY(stat_recv.alloc((size_t)W * 128 * 8));
    FDX_HIP(hipMemsetAsync(stats.p, 0, stats.bytes, st));
    if (c->loopback) FDX_HIP(hipMemsetAsync(recv_buf.p, 0, recv_buf.bytes, st));   // part of it is never written by the self-copy
    // per-peer offsets of the staging blocks: a graph from the queued shard build (fdx_graph_shard_knn_dev) has them on the device
    const int* soff_p = g->send_off_dev.as<int>();
    const int* roff_p = g->recv_off_dev.as<int>();
    const bool upload_offsets = !soff_p || !roff_p;
    if (upload_offsets) {
        FDX_TRY(soff.alloc((size_t)(W + 1) * 4));
        FDX_TRY(roff.alloc((size_t)(W + 1) * 4));
        FDX_HIP(hipMemcpyAsync(soff.p, g->send_off.data(), (size_t)(W + 1) * 4, hipMemcpyHostToDevice, st));
        FDX_HIP(hipMemcpyAsync(roff.p, g->recv_off.data(), (size_t)(W + 1) * 4, hipMemcpyHostToDevice, st));
        soff_p = soff.as<int>();
        roff_p = roff.as<int>();
    }
    // the second buffer: only its pad rows and halo columns must read as defined before the first sweep has written the own rows
    // and the first exchange the halo - a fill of the whole buffer is the simple form of that
    FDX_HIP(hipMemsetAsync(beta1_dev, 0, (size_t)K * ld * 8, st));
    if (upload_offsets) FDX_HIP(hipStreamSynchronize(st));                    // the offset vectors are host objects of g

    BcdSweepArgs a{};
    a.H = H_dev; a.XtX = XtX_dev; a.ell = g->ell.as<int>(); a.slice_off = g->slice_off.as<int>(); a.deg = g->deg.as<int>();
    a.stats = stats.as<unsigned long long>(); a.rel_change = relchg_p;
    a.lambda = lambda; a.rho = rho_eff; a.tol = tol; a.ldh = (int)ldh; a.ld = (int)ld; a.n = (int)g->n;
    a.n_slices = g->n_slices; a.K = K;
    const bool tiled = g->tiled && !fdx::env("FDX_NO_TILED");
    if (tiled) {
        a.tiled = 1; a.ell_local = g->ell_local.as<unsigned short>(); a.tile_halo = g->tile_halo.as<int>();
        a.tile_hcnt = g->tile_hcnt.as<int>(); a.n_tiles = g->n_tiles; a.halo_max = g->halo_max;
    }
    a.beta_in = beta0_dev;
    a.beta_out = beta1_dev;
    // boundary-first ordering needs the tiled sweep (tile lists) and somebody to talk to
    bool split = tiled && bcd_sweep_uses_tiles(a) && total_send > 0 && g->n > 0 && !fdx::env("FDX_NO_OVERLAP");
    // ... and a shard large enough for the split to pay: a sweep launch lasts at least one workgroup's life (~25-35 us) however few
    // tiles it has, so for shards of a few rounds of resident workgroups (256 CUs x 3) boundary + interior cost two such lives for the work of
    // one - 125k-spot shards: 36 + 34 us against 36 - and the interior sweep is too short to hide anything behind
    // (4096 tiles = 1M spots per rank: below that a rank's halo is a few hundred KB per peer - ~20 us of transfer and latency to
    // hide, against the ~36 us the extra launch costs; the 1.25M-spot ranks of configs[4], 1-2.6 MB per peer, are about even)
    const int split_min_tiles = fdx::env("FDX_SPLIT_MIN_TILES") ? atoi(fdx::env("FDX_SPLIT_MIN_TILES")) : 4096;
    if (split && g->n_tiles < split_min_tiles) split = false;
    // every rank must take the same route for the convergence slots: they ride with the halo only when NO rank of the job can
    // split (shards are equal to within one tile: the largest has at most ceil(tiles / W) + 1 of them)
    const long long tiles_max = g->world_n > 0 ? ((g->world_n + 255) / 256 + W - 1) / W + 1 : (1LL << 40);
    const bool piggy_ok = W > 1 && !c->loopback && (c->nccl || c->local) && !fdx::exp_env("FDX_STATS_ALLREDUCE") &&
                          (tiles_max < split_min_tiles || fdx::env("FDX_NO_OVERLAP") != nullptr);
    // (piggy_ok is the JOB's decision - world size and spot count only.  A rank whose own shard is larger than the balanced bound, from
    // bounds that fdx_graph_localize accepted but shard_bounds() would not have made, must not split on its own: it would all-reduce
    // while its peers piggy-back - a mismatched communicator.  It sweeps unsplit like everybody else.)
    if (piggy_ok) split = false;
    if (split) {
        FDX_TRY(build_tile_lists(*g, st));          // no-op for a graph of the queued shard build: its lists were made on the device
        split = g->n_tiles_boundary > 0 && g->n_tiles_interior > 0;
    }

    // beta0 = 1/K on own + halo (solver.py:372), pad types 0: a constant of the first sweep where the tiled kernel takes one
    // (bcd_sweep_inst.cpp, INIT: whole-shard launches, no pad types), written to the buffer otherwise
    double init_uniform = 0.0;
    if (!split && tiled && K_real == K && K <= FDX_MAX_K_FAST && max_iter > 0 && g->n > 0 && bcd_sweep_uses_tiles(a) && !fdx::env("FDX_NO_INIT_SWEEP")) {
        init_uniform = 1.0 / (double)K;
        FDX_TRY(solver_zero_pad(beta0_dev, ld, g->n_total, K, st));
    } else {
        FDX_TRY(solver_init_beta(beta0_dev, ld, g->n_total, K_real, st, K));
    }

    // rows a peer needs are written into the send staging by the sweep itself (tiled kernel; bcd_sweep_inst.cpp) - the lists seen
    // from the rows are made once per graph
    const bool fused_pack = tiled && total_send > 0 && g->n > 0 && bcd_sweep_uses_tiles(a) && !fdx::env("FDX_NO_FUSED_PACK");
    if (fused_pack) {
        if (!g->send_head.p) {
            FDX_TRY(g->send_head.alloc((size_t)g->n * 4));
            FDX_TRY(g->send_ent.alloc((size_t)total_send * sizeof(int4)));
            FDX_HIP(hipMemsetAsync(g->send_head.p, 0, (size_t)g->n * 4, st));
            hipLaunchKernelGGL(send_entries_kernel, dim3(ceil_div(total_send, 256)), dim3(256), 0, st, g->send_idx.as<int>(), soff_p, W, total_send,
                               g->send_head.as<int>(), g->send_ent.as<int4>());
            FDX_CHECK_LAUNCH();
        }
        a.send_head = g->send_head.as<int>();
        a.send_ent = g->send_ent.as<int4>();
        a.send_buf = send_buf.as<double>();
    }

    // the trace lands in pinned host memory behind an event; while the host waits for it the first iterations of the NEXT chunk
    // are already queued (no-ops if this chunk converged: device-side stopping rule), as in solver_run
    double* rc_host = (double*)pinned_scratch(2, (size_t)iters * sizeof(double));
    FDX_REQUIRE(rc_host != nullptr, "sharded solve: pinned host buffer");
    for (int j = 0; j < iters; ++j) rc_host[j] = 0.0;
    int done = 0, n_iter = 0, chunk = 4;
    bool converged = false;
    hipEvent_t ev0[2] = {nullptr, nullptr}, ev1[2] = {nullptr, nullptr}, evCopy = nullptr;
    struct EvGuard { hipEvent_t* e[5]; ~EvGuard() { for (auto* q : e) if (*q) (void)hipEventDestroy(*q); } } ev_guard{{&ev0[0], &ev0[1], &ev1[0], &ev1[1], &evCopy}};
    for (int j = 0; j < 2; ++j) {
        FDX_HIP(hipEventCreate(&ev0[j]));
        FDX_HIP(hipEventCreate(&ev1[j]));
    }
    FDX_HIP(hipEventCreateWithFlags(&evCopy, hipEventDisableTiming));
    double sweep_ms = 0.0;
    double* beta[2] = {beta0_dev, beta1_dev};
    // the in-process transport meets at host barriers: nothing is gained by queueing ahead there
    const int n_ahead = (c->local || fdx::exp_env("FDX_NO_SWEEP_AHEAD")) ? 0 : 2;
    auto iterate = [&](int it, bool last_of_chunk) -> int {
        a.it = it;
        a.beta_in = beta[it & 1];
        a.beta_out = beta[(it + 1) & 1];
        hipStream_t sx = st;                                          // stream of the halo traffic
        if (g->n > 0) {
            if (split) {
                a.tile_list = g->tiles_boundary.as<int>(); a.n_list = g->n_tiles_boundary;
                FDX_TRY(launch_bcd_sweep(a, nullptr, 0, st));
            } else {
                a.tile_list = nullptr; a.n_list = 0;
                a.init_uniform = it == 0 ? init_uniform : 0.0;
                FDX_TRY(launch_bcd_sweep(a, sweep_scratch.as<double>(), scratch_ld, st));
            }
        }
        if (total_send > 0 && !fused_pack) {
            hipLaunchKernelGGL(halo_pack_kernel, dim3(ceil_div((long long)total_send * K, 256)), dim3(256), 0, st, a.beta_out, (long long)ld, K,
                               g->send_idx.as<int>(), soff_p, W, total_send, send_buf.as<double>());
            FDX_CHECK_LAUNCH();
        }
        if (split) {
            FDX_HIP(hipEventRecord(c->ev_packed, st));
            a.tile_list = g->tiles_interior.as<int>(); a.n_list = g->n_tiles_interior;
            FDX_TRY(launch_bcd_sweep(a, nullptr, 0, st));             // beside the halo traffic
            sx = c->side;
            FDX_HIP(hipStreamWaitEvent(sx, c->ev_packed, 0));
        }
        // Without the boundary / interior split the iteration's convergence slots are complete when the rows are sent: they ride in
        // the same group to every peer and the unpack kernel folds them - no all-reduce.  (With the split the interior tiles are
        // still being swept then: large shards keep the all-reduce, whose latency is noise beside their sweeps.)
        const bool piggy = piggy_ok && !split;
        unsigned long long* st_it = a.stats + (size_t)it * 128;
        if (c->local || total_send > 0 || total_recv > 0 || piggy)      // the in-process transport meets at barriers
            FDX_TRY(exchange(c, send_buf.as<double>(), g->send_off, recv_buf.as<double>(), g->recv_off, K, sx, piggy ? st_it : nullptr,
                             piggy ? stat_recv.as<unsigned long long>() : nullptr));
        if (total_recv > 0 || piggy) {
            hipLaunchKernelGGL(halo_unpack_kernel, dim3(std::max(1, ceil_div((long long)total_recv * K, 256))), dim3(256), 0, sx, a.beta_out,
                               (long long)ld, K, (long long)g->n, roff_p, W, total_recv, recv_buf.as<double>(), piggy ? st_it : nullptr,
                               piggy ? stat_recv.as<unsigned long long>() : nullptr, c->rank);
            FDX_CHECK_LAUNCH();
        }
        if (split) {
            FDX_HIP(hipEventRecord(c->ev_halo, sx));
            FDX_HIP(hipStreamWaitEvent(st, c->ev_halo, 0));
        }
        if (!piggy) FDX_TRY(allreduce(c, a.stats + (size_t)it * 128, 128, true, st));
        // a rank without rows launches no sweep, and it is sweep it + 1 that folds the slots of sweep it into rel_change[it]:
        // fold here, or this rank reads 0.0, calls the solve converged after its first chunk and leaves the others waiting
        if (g->n == 0 && !last_of_chunk) FDX_TRY(launch_bcd_fold_last(a.stats, a.rel_change, it, st));
        return 0;
    };
    auto run = [&]() -> int {
        int queued_ahead = 0, ci = 0;
        while (done < max_iter && !converged) {
            const int end = std::min<int>(max_iter, done + chunk);
            const int pair = ci & 1;
            if (queued_ahead == 0) FDX_HIP(hipEventRecord(ev0[pair], st));
            for (int it = done + queued_ahead; it < end; ++it) FDX_TRY(iterate(it, it + 1 == end));
            FDX_TRY(launch_bcd_fold_last(a.stats, a.rel_change, end - 1, st));
            FDX_HIP(hipEventRecord(ev1[pair], st));
            FDX_HIP(hipMemcpyAsync(rc_host + done, relchg_p + done, (size_t)(end - done) * 8, hipMemcpyDeviceToHost, st));
            FDX_HIP(hipEventRecord(evCopy, st));
            int ahead = 0;
            if (end < max_iter && n_ahead > 0) {
                ahead = std::min(n_ahead, max_iter - end);
                FDX_HIP(hipEventRecord(ev0[pair ^ 1], st));
                // a rank without rows folds sweep it in iteration it itself unless it is the last of its chunk: these are not
                for (int it = end; it < end + ahead; ++it) FDX_TRY(iterate(it, false));
            }
            FDX_HIP(hipEventSynchronize(evCopy));
            float ms = 0.f;
            FDX_HIP(hipEventElapsedTime(&ms, ev0[pair], ev1[pair]));
            sweep_ms += ms;
            for (int it = done; it < end; ++it) {
                n_iter = it + 1;
                if (rc_host[(size_t)it] < tol) { converged = true; break; }   // solver.py:409-413
            }
            done = end;
            queued_ahead = ahead;
            chunk = std::max(std::min(ci == 0 ? chunk : chunk * 2, 32), ahead);
            ++ci;
        }
        return 0;
    };
    const int rc = run();
    if (rc && c->local) c->local->abort();     // the other thread ranks leave their barriers with an error instead of hanging
    (void)hipStreamSynchronize(st);
    if (c->side) (void)hipStreamSynchronize(c->side);
    if (rc) return rc;
    info->n_iterations = n_iter;
    info->converged = converged ? 1 : 0;
    info->final_change = n_iter > 0 ? rc_host[(size_t)n_iter - 1] : 0.0;
    info->sweep_ms = sweep_ms;
    *result_buffer = n_iter & 1;
    if (rel_changes_out)
        for (int i = 0; i < n_iter; ++i) rel_changes_out[i] = rc_host[(size_t)i];
    return 0;
}

}  // extern "C"
