// Which of several EXACTLY equidistant points does the reference's k-nearest-neighbour query return?
//
// flashdeconv/utils/graph.py:60-63 builds `cKDTree(coords)` and calls `tree.query(coords, k = k + 1)`; on a regular lattice
// (Visium-HD bins: four neighbours at distance 1, four at sqrt 2, k = 6) the k-th neighbour is one of several at the same
// distance, and which ones come back is decided by nothing but the order in which scipy's tree visits the points.  The
// device build (graph_kernels.cpp) breaks such ties by spot index; with knn_ties="ckdtree" the Python driver asks this file
// instead whenever fdx_graph_knn_ties() reports ties, and gets the reference's neighbour lists index for index.
//
// This is a host restatement of the published algorithm of scipy.spatial.cKDTree (third-party dependency of the reference,
// scipy >= 1.7 per its pyproject; checked here against scipy 1.15.3, the version of the build container) for exactly that
// call: tree construction with the constructor's defaults (leafsize 16, compact_nodes, balanced_tree) and the k-nearest
// query with p = 2, eps = 0, no distance bound.  What has to be reproduced is ORDER, so the restatement keeps every
// order-defining detail:
//   build   - recursive; bounds of a node recomputed from its points; split dimension = largest spread (first wins);
//             median by std::nth_element over the node's index range, compared by the coordinate alone, at
//             position n/2; split value = coordinate of that element; then a Hoare-style pass moving "< split" left and
//             ">= split" right (on a lattice many points equal the split: they all go right); when nothing is below the
//             split (the median is the node's minimum) the split moves to nextafter(minimum, +inf) and the pass is repeated;
//             children [start, p) and [p, end).  The index array after all of this is the order
//             in which a leaf's points are tested.  (std::nth_element is libstdc++'s introselect here as in scipy's
//             manylinux wheels.)
//   query   - best-first over nodes with scipy's own binary heap (strict comparisons: ties keep insertion structure),
//             near child followed directly, far child pushed when its distance <= the current k-th distance; a leaf's
//             points are tested in index-array order and accepted when d < bound (strict), the bound being the k-th
//             distance once k neighbours are known - so among equidistant candidates the ones met FIRST stay.
//             Squared distances accumulated coordinate by coordinate from zero, side distances updated incrementally
//             (min_distance += new - old) exactly as the library does, because equality decides here.
// tests/test_host.py compares this entry with scipy itself (index arrays of the tree and query results) on lattices,
// duplicated points, random clouds in 1-3 dimensions - no GPU involved.
#include "fdx_env.h"
#include <algorithm>
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstdint>
#include <cstdlib>
#include <cstring>
#include <system_error>
#include <atomic>
#include <condition_variable>
#include <deque>
#include <functional>
#include <memory>
#include <mutex>
#include <thread>
#include <unistd.h>
#include <vector>

#include "fdx_internal.h"
#include "kdtree_dev.h"

namespace fdx {
namespace {

struct KdNode {
    long long split_dim = -1;     // -1: leaf
    double split = 0.0;
    long long start = 0, end = 0;
    long long less = -1, greater = -1;
};

// std::vector that leaves new elements uninitialised (the index array is written in full right after its resize - by several
// threads, each touching its own pages first)
template <class T>
struct KdRawAlloc : std::allocator<T> {
    template <class U> struct rebind { using other = KdRawAlloc<U>; };
    template <class U> void construct(U* p) noexcept { ::new ((void*)p) U; }
    template <class U, class... A> void construct(U* p, A&&... a) { ::new ((void*)p) U(std::forward<A>(a)...); }
};

struct KdTree {
    const double* data = nullptr;
    long long n = 0;
    int m = 0;
    long long leafsize = 16;
    std::vector<long long, KdRawAlloc<long long>> indices;
    // all nodes, root first: raw storage that the threads of the build's last pass fill (and touch first) in parallel
    struct Nodes {
        KdNode* p = nullptr;
        size_t n = 0;
        Nodes() = default;
        Nodes(const Nodes&) = delete;
        Nodes& operator=(const Nodes&) = delete;
        ~Nodes() { ::operator delete(p); }
        void raw(size_t count) { ::operator delete(p); p = static_cast<KdNode*>(::operator new(count * sizeof(KdNode))); n = count; }
        size_t size() const { return n; }
        const KdNode& operator[](size_t i) const { return p[i]; }
    } nodes;
    std::vector<double> maxes, mins;   // of the whole data set
    // build only: the node vectors of the subtrees that were built on threads of their own (kd_children), until kd_build_tree
    // lays them out behind one another in `nodes`
    std::mutex seg_mu;
    std::deque<std::vector<KdNode>> segs;
    bool pooled = false;                 // the build may cut its work into tasks of the thread pool (more than one thread to its name)
    std::atomic<bool> failed{false};     // a task ran out of memory
    std::vector<int32_t, KdRawAlloc<int32_t>> lpos, rpos;   // the partition passes' position lists, by index-array position
};

// par_depth > 0: the `less` subtree of a large node is built as a task of the thread pool into a node vector of its own, which joins the
// tree's list of segments; the parent's link to it is the segment's tag until kd_build_tree lays all segments out in one
// vector (appending every subtree to its parent's vector on the way up copied the nodes once per level: 2 of 12 ms per million
// points).  The index array is partitioned in place - disjoint ranges; the NUMBERING of the nodes differs from a serial build,
// which nothing reads: queries follow the less / greater links.  The serial build was 130 of the 145 ms the tie remedy spent on
// the host for a million lattice points: its top levels are cache-missing passes over all points.
}  // namespace
// Host threads this process may keep busy: the hardware's count, cut to the CPU bandwidth quota of the control group the process
// runs in (containers: /sys/fs/cgroup/cpu.max or the v1 pair).  A burst on more threads than the quota covers gets the whole
// process - its main thread included - throttled for the rest of the scheduler period: on a 256-thread host with a 16-CPU quota
// the 128 query threads of one tie remedy cost the NEXT calls 10-30 ms stalls in unrelated places (measured; round 5).
unsigned host_cpu_budget() {
    static const unsigned budget = [] {
        unsigned hw = std::thread::hardware_concurrency();
        if (hw == 0) hw = 1;
        double cpus = (double)hw;
        if (FILE* f = std::fopen("/sys/fs/cgroup/cpu.max", "r")) {
            char q[64] = {0};
            long long period = 0;
            if (std::fscanf(f, "%63s %lld", q, &period) == 2 && period > 0 && std::strcmp(q, "max") != 0) cpus = std::min(cpus, atof(q) / (double)period);
            std::fclose(f);
        } else {
            long long quota = -1, period = 0;
            if (FILE* fq = std::fopen("/sys/fs/cgroup/cpu/cpu.cfs_quota_us", "r")) { if (std::fscanf(fq, "%lld", &quota) != 1) quota = -1; std::fclose(fq); }
            if (FILE* fp = std::fopen("/sys/fs/cgroup/cpu/cpu.cfs_period_us", "r")) { if (std::fscanf(fp, "%lld", &period) != 1) period = 0; std::fclose(fp); }
            if (quota > 0 && period > 0) cpus = std::min(cpus, (double)quota / (double)period);
        }
        return (unsigned)std::max(1.0, std::floor(cpus + 0.5));
    }();
    return budget;
}

// Host threads the restated cKDTree may use (build forks, host queries): the process's budget, or the share the caller set with
// fdx_kdtree_set_threads (the ranks of one host each build the tree of the replicated coordinates: they share its cores), or
// FDX_KDTREE_THREADS.
static std::atomic<int> g_kd_threads{0};
static std::atomic<long long> g_kd_team_min{0};
static std::atomic<long long> g_kd_local_max{-1};
static std::atomic<int> g_kd_device_build{1};   // fdx_kdtree_tune(2, 0): the tree of 1-3 coordinates is built on the host even where the device could (kdtree_build_dev.cpp)
static unsigned kd_thread_share() {
    if (const char* e = fdx::env("FDX_KDTREE_THREADS")) return (unsigned)std::max(1, atoi(e));
    const int v = g_kd_threads.load();
    return v > 0 ? (unsigned)v : host_cpu_budget();
}
namespace {
// ---- one standing pool of host threads for everything the build does in parallel ------------------------------------------------
// The build's critical path is the chain root -> child -> grandchild: bounds, selection and partition of a million, half a
// million, a quarter of a million points, each a serial pass in scipy (12 + 6 + 3 ms of the 21 ms a million lattice points cost
// with the subtrees already on threads of their own).  Starting threads per pass costs more than the pass (tried, round 5);
// threads that are already waiting do not.  Everything parallel is a TASK of this pool - the chunks of a top node's pass
// (`parallel`), the `less` subtree of a large node (`spawn`) - and whoever waits for tasks of its own runs other tasks meanwhile
// (`wait_help`), so several nodes have their passes under way at once on the same 16 threads, and the process never has more
// than its share of the host's CPUs busy: a burst on 70 threads (teams per level + a thread per subtree, the first form of this
// round) leaves a 5 ms scheduler slice stranded on every CPU it touched, and a 16-CPU quota then throttles a whole period
// (3 of 28 periods in a 20-fit run: 10 ms steps among 32 ms ones).
class KdPool {
public:
    static KdPool& get() { static KdPool p; return p; }
    // members (the caller included) this build may keep busy; creates what is missing
    void want(int members) {
        members = std::max(1, std::min(members, 32));
        if (pid_.load() != getpid()) {
            // A fork()ed child has the object but not the threads, and the mutex / condition variable in whatever state the
            // parent's threads had them in at that instant (a worker about to sleep holds the condition variable's internal
            // lock: the child's first notify would wait for it for ever).  Everything is made anew; the old thread objects are
            // left alone (they cannot be joined).  The first caller in the child does this before anything else runs here.
            new std::vector<std::thread>(std::move(th_));
            new (&th_) std::vector<std::thread>();
            new (&m_) std::mutex();
            new (&cv_) std::condition_variable();
            new (&q_) std::deque<Task>();
            n_queued_.store(0);
            active_.store(1);
            pid_.store(getpid());
        }
        std::lock_guard<std::mutex> lk(m_);
        while ((int)th_.size() + 1 < members) {
            const int wid = (int)th_.size();
            try { th_.emplace_back([this, wid] { loop(wid); }); } catch (const std::system_error&) { break; }
        }
        active_.store(std::min(members, (int)th_.size() + 1));
    }
    int size() const { return active_.load(); }
    // a task; *pending (may be NULL) is decremented when it has run
    void spawn(std::function<void()> fn, std::atomic<int>* pending) {
        if (pending) pending->fetch_add(1, std::memory_order_relaxed);
        {
            std::lock_guard<std::mutex> lk(m_);
            q_.push_back(Task{std::move(fn), pending});
            n_queued_.fetch_add(1, std::memory_order_release);
        }
        cv_.notify_one();
    }
    // runs tasks until *pending is zero
    // (a waiter with nothing to run spins briefly, then SLEEPS until a task is queued or a counter reaches zero: 32 members
    // spinning through a 10 ms build were 0.3 CPU-seconds per fit, and a 16-CPU quota throttled one step in five)
    void wait_help(std::atomic<int>& pending) {
        int idle = 0;
        while (pending.load(std::memory_order_acquire) != 0) {
            if (run_one()) { idle = 0; continue; }
            if (++idle < 400) { __builtin_ia32_pause(); continue; }
            std::unique_lock<std::mutex> lk(m_);
            cv_.wait(lk, [&] { return pending.load(std::memory_order_acquire) == 0 || !q_.empty(); });
            idle = 0;
        }
    }
    // f(tid, nt) for tid < nt, the caller taking tid 0 and whatever else nobody has started yet
    // (a chunk that throws - out of memory - sets *failed; the caller looks at it when the build is over)
    template <class F>
    void parallel(int nt, const F& f, std::atomic<bool>* failed) {
        if (nt <= 1) { f(0, 1); return; }
        std::atomic<int> pending{nt - 1};
        {                                                             // all chunks under one lock, one wake-up for everybody
            std::lock_guard<std::mutex> lk(m_);
            for (int tid = 1; tid < nt; ++tid)
                q_.push_back(Task{[&f, tid, nt, failed] { try { f(tid, nt); } catch (...) { failed->store(true); } }, &pending});
            n_queued_.fetch_add(nt - 1, std::memory_order_release);
        }
        cv_.notify_all();
        try { f(0, nt); } catch (...) { failed->store(true); }
        wait_help(pending);
    }
    ~KdPool() {
        if (pid_.load() != getpid()) { new std::vector<std::thread>(std::move(th_)); return; }
        { std::lock_guard<std::mutex> lk(m_); stop_ = true; }
        cv_.notify_all();
        for (auto& t : th_) if (t.joinable()) t.join();
    }
private:
    struct Task { std::function<void()> fn; std::atomic<int>* pending; };
    bool run_one() {
        if (n_queued_.load(std::memory_order_acquire) == 0) return false;
        Task t;
        {
            std::lock_guard<std::mutex> lk(m_);
            if (q_.empty()) return false;
            t = std::move(q_.back());                                 // newest first: the chunks of a pass that has just been cut run
            q_.pop_back();                                            // before subtrees queued earlier - the passes are the critical path
            n_queued_.fetch_sub(1, std::memory_order_release);
        }
        t.fn();                                                       // (tasks catch what they throw)
        if (t.pending && t.pending->fetch_sub(1, std::memory_order_acq_rel) == 1) {
            { std::lock_guard<std::mutex> lk(m_); }                   // (a waiter between its test and its sleep holds this)
            cv_.notify_all();
        }
        return true;
    }
    void loop(int wid) {
        for (;;) {
            if (wid + 1 < active_.load(std::memory_order_relaxed) && run_one()) continue;
            // nothing to do: a short spin (the next pass of a node usually follows within microseconds), then sleep
            bool got = false;
            for (int s = 0; s < 1500 && !got; ++s) {
                __builtin_ia32_pause();
                got = wid + 1 < active_.load(std::memory_order_relaxed) && n_queued_.load(std::memory_order_acquire) != 0;
            }
            if (got) continue;
            std::unique_lock<std::mutex> lk(m_);
            cv_.wait(lk, [&] { return stop_ || (wid + 1 < active_.load(std::memory_order_relaxed) && !q_.empty()); });
            if (stop_) return;
        }
    }
    std::mutex m_;
    std::condition_variable cv_;
    std::deque<Task> q_;
    std::atomic<int> n_queued_{0};
    std::vector<std::thread> th_;
    std::atomic<int> active_{1};
    bool stop_ = false;
    std::atomic<pid_t> pid_{getpid()};
};

// chunks a pass over `len` of the tree's n points is cut into: the node's share of the pool (the root all members, its children
// half each: the nodes of a level have their passes under way together), at least two, never chunks under 256 points
struct KdTree;
static int kd_pass_chunks(const KdTree& t, long long len);
// Members of the pool a build may keep busy: twice the caller's share of the host's CPUs where the hardware has the threads, 32
// at most.  The share is a CPU-TIME quota (host_cpu_budget), a build is a burst of ~10 ms: 32 threads for that long are a fifth
// of what 16 CPUs may spend per scheduler period, and the 32 subtrees of a million points run side by side instead of in two rounds.
static int kd_pool_members() {
    const unsigned share = kd_thread_share();
    if (share <= 1) return 1;
    const unsigned hw = std::max(1u, std::thread::hardware_concurrency());
    return (int)std::min(std::min(hw, 2u * share), 32u);
}
static long long kd_team_min() {                                     // fdx_kdtree_set_team_min: the tests send small nodes through the team
    const long long v = g_kd_team_min.load();
    return v > 0 ? v : 400000;                                        // (a million points: the root and its two children; 250000-point nodes do as well on one thread each)
}

static int kd_pass_chunks(const KdTree& t, long long len) {
    const long long members = KdPool::get().size();
    const long long share = (members * len + t.n - 1) / std::max<long long>(1, t.n);
    return (int)std::max<long long>(1, std::min<long long>(std::min<long long>(members, std::max<long long>(2, share)), len / 256));
}

// libstdc++'s __unguarded_partition(lo, hi, pivot) over the index range [lo, hi), keys read through the index array - by the whole
// team, with the result of the serial scan, swap for swap.  The serial scan stops its left pointer on every key >= pivot and its
// right pointer on every key <= pivot and swaps the two; between the pointers the array is still as it was, so the k-th swap
// is (k-th key >= pivot from the left, k-th key <= pivot from the right), for as long as the former lies left of the latter,
// and the returned cut is where the left pointer stops next: the next such key from the left, or the right partner of the
// last swap (which now holds a key >= pivot), whichever comes first.  (The pivot - moved to lo - 1 by the median step -
// and the median's other two samples guarantee both lists are non-empty inside the range.)
template <class Key>
long long team_unguarded_partition(KdTree& t, long long* idx, long long lo, long long hi, double pv, const Key& key) {
    KdPool& team = KdPool::get();
    const long long len = hi - lo;
    const int nt = kd_pass_chunks(t, len);
    int32_t* lp = t.lpos.data() + lo;                                // (the ranges of two nodes are disjoint: so are their lists)
    int32_t* rp = t.rpos.data() + lo;
    std::vector<long long> cb((size_t)nt + 1), cl((size_t)nt + 1, 0), cr((size_t)nt + 1, 0);
    for (int t = 0; t <= nt; ++t) cb[(size_t)t] = len * t / nt;
    team.parallel(nt, [&](int tid, int) {
        const long long b = cb[(size_t)tid], e = cb[(size_t)tid + 1];
        long long nl = 0, nr = 0;
        for (long long i = b; i < e; ++i) {
            const double v = key(idx[lo + i]);
            lp[b + nl] = (int32_t)i;
            nl += !(v < pv);
            rp[b + nr] = (int32_t)i;
            nr += !(pv < v);
        }
        cl[(size_t)tid + 1] = nl;
        cr[(size_t)tid + 1] = nr;
    }, &t.failed);
    // pl[t]: left-list entries before chunk t; sr[t]: right-list entries (counted from the right) after chunk t
    std::vector<long long> pl((size_t)nt + 1, 0), sr((size_t)nt + 1, 0);
    for (int t = 0; t < nt; ++t) pl[(size_t)t + 1] = pl[(size_t)t] + cl[(size_t)t + 1];
    for (int t = nt - 1; t >= 0; --t) sr[(size_t)t] = sr[(size_t)t + 1] + cr[(size_t)t + 1];
    const long long nL = pl[(size_t)nt], nR = sr[0];
    auto L = [&](long long k) {
        int t = (int)(std::upper_bound(pl.begin(), pl.end(), k) - pl.begin()) - 1;
        return (long long)lp[cb[(size_t)t] + (k - pl[(size_t)t])];
    };
    auto R = [&](long long k) {                                       // chunk t holds the k with sr[t + 1] <= k < sr[t]
        int t = nt - 1;
        while (sr[(size_t)t] <= k) --t;
        return (long long)rp[cb[(size_t)t] + (cr[(size_t)t + 1] - 1 - (k - sr[(size_t)t + 1]))];
    };
    long long a = 0, b = std::min(nL, nR);                            // K = number of k with L(k) < R(k): those come first
    while (a < b) {
        const long long mid = (a + b) / 2;
        if (L(mid) < R(mid)) a = mid + 1; else b = mid;
    }
    const long long K = a;
    if (K > 0)
        team.parallel(nt, [&](int tid, int n_t) {
            const long long k0 = K * tid / n_t, k1 = K * (tid + 1) / n_t;
            if (k0 >= k1) return;
            int tl = (int)(std::upper_bound(pl.begin(), pl.end(), k0) - pl.begin()) - 1;
            long long il = k0 - pl[(size_t)tl];
            int tr = nt - 1;
            while (sr[(size_t)tr] <= k0) --tr;
            long long ir = cr[(size_t)tr + 1] - 1 - (k0 - sr[(size_t)tr + 1]);
            for (long long k = k0; k < k1; ++k) {
                while (il >= cl[(size_t)tl + 1]) { ++tl; il = 0; }
                while (ir < 0) { --tr; ir = cr[(size_t)tr + 1] - 1; }
                std::swap(idx[lo + lp[cb[(size_t)tl] + il]], idx[lo + rp[cb[(size_t)tr] + ir]]);
                ++il;
                --ir;
            }
        }, &t.failed);
    long long cut = -1;
    if (K < nL) cut = L(K);
    if (K > 0) { const long long r = R(K - 1); cut = cut < 0 ? r : std::min(cut, r); }
    return lo + cut;
}

// std::nth_element(idx + first, idx + nth, idx + last, key(a) < key(b)) - libstdc++'s introselect with the partition passes of the
// long ranges done by the team; below `team_min` libstdc++'s own routine continues with the depth budget that is left.
template <class Key>
void team_nth_element(KdTree& t, long long* idx, long long first, long long nth, long long last, const Key& key, long long team_min) {
    auto cmp = [&key](long long a, long long b) { return key(a) < key(b); };
    auto icmp = __gnu_cxx::__ops::__iter_comp_iter(cmp);
    if (first == last || nth == last) return;
    long long depth = 2 * (long long)std::__lg(last - first);
    while (last - first > 3) {
        if (last - first < team_min || depth == 0) {
            std::__introselect(idx + first, idx + nth, idx + last, depth, icmp);
            return;
        }
        --depth;
        const long long mid = first + (last - first) / 2;
        std::__move_median_to_first(idx + first, idx + first + 1, idx + mid, idx + last - 1, icmp);
        const long long cut = team_unguarded_partition(t, idx, first + 1, last, key(idx[first]), key);
        if (cut <= nth) first = cut; else last = cut;
    }
    std::__insertion_sort(idx + first, idx + last, icmp);
}

static long long kd_fork_min() {
    static const long long v = fdx::exp_env("FDX_KDTREE_FORK_MIN") ? std::max(1024, atoi(fdx::exp_env("FDX_KDTREE_FORK_MIN"))) : 32768;
    return v;
}

// The two children of a node, the `less` one as a task of the pool when the node is large and forks are left (see the note at
// KdTree): build(nodes, lesser) appends the subtree of that child to `nodes` and returns its root's index there.  The parent
// builds the other child, then runs pool tasks until its own has finished (its stack holds what the task works on).
inline long long kd_seg_tag(long long seg) { return -(2 + seg); }     // a link to the root (node 0) of segment `seg`
template <class Build>
void kd_children(KdTree& t, std::vector<KdNode>& nodes, bool fork, const Build& build, long long& less, long long& greater) {
    if (fork && t.pooled) {
        std::vector<KdNode>* sub = nullptr;
        long long seg = -1;
        {
            std::lock_guard<std::mutex> lk(t.seg_mu);
            t.segs.emplace_back();                                    // (a deque: the element stays where it is)
            sub = &t.segs.back();
            seg = (long long)t.segs.size() - 1;
        }
        std::atomic<int> pending{0};
        KdPool& pool = KdPool::get();
        pool.spawn([&build, &t, sub] { try { (void)build(*sub, true); } catch (...) { t.failed.store(true); } }, &pending);   // (its root is node 0 of `sub`)
        less = kd_seg_tag(seg);
        try {
            greater = build(nodes, false);
        } catch (...) {
            pool.wait_help(pending);
            throw;
        }
        pool.wait_help(pending);
        return;
    }
    less = build(nodes, true);
    greater = build(nodes, false);
}

// A subtree small enough for a core's cache is built on a contiguous copy of its points - records {coordinates, index} in
// index-array order - instead of through the index array: the same comparisons, hence the same swaps on the same positions and
// the same index order when the records' indices are written back, but bounds, selection and partition stream through 24-byte
// records that sit in L2 instead of chasing 8-byte indices into the coordinate array (the build is throughput-bound once its
// subtrees have threads of their own: ~150 ms of core time per million points, 16 levels of three passes).
template <int M>
struct KdRec {
    double c[M];
    long long i;
};

static long long kd_local_max() {                                     // fdx_kdtree_tune(1, points); 0 switches the copy off
    const long long v = g_kd_local_max.load();
    return v >= 0 ? v : 65536;
}

// std::nth_element(r, r + nth, r + last, key d ascending) with the partition passes of libstdc++'s introselect done WITHOUT the
// data-dependent branches of its two-pointer scan (half of them mispredicted on a median pivot - the larger part of a
// subtree's build time): one pass lists the positions with key >= pivot and the positions with key <= pivot - the stops of the
// scan's left and right pointer -, the k-th swap is (k-th of the first list, k-th from the end of the second) while the former
// lies left of the latter, the cut is where the left pointer would stop next (team_unguarded_partition above has the argument).
// Short ranges go to libstdc++'s own routine with the depth budget that is left.
template <int M>
void rec_nth_element(KdRec<M>* r, long long nth, long long last, int d, int32_t* scratch) {
    auto cmp = [d](const KdRec<M>& a, const KdRec<M>& b) { return a.c[d] < b.c[d]; };
    auto icmp = __gnu_cxx::__ops::__iter_comp_iter(cmp);
    long long first = 0;
    if (first == last || nth == last) return;
    long long depth = 2 * (long long)std::__lg(last - first);
    while (last - first > 3) {
        if (last - first < 192 || depth == 0) {
            std::__introselect(r + first, r + nth, r + last, depth, icmp);
            return;
        }
        --depth;
        const long long mid = first + (last - first) / 2;
        std::__move_median_to_first(r + first, r + first + 1, r + mid, r + last - 1, icmp);
        const double pv = r[first].c[d];
        const long long lo = first + 1, len = last - lo;
        int32_t* lp = scratch;
        int32_t* rp = scratch + len;
        long long nl = 0, nr = 0;
        for (long long i = 0; i < len; ++i) {
            const double v = r[lo + i].c[d];
            lp[nl] = (int32_t)i;
            nl += !(v < pv);
            rp[nr] = (int32_t)i;
            nr += !(pv < v);
        }
        long long a = 0, b = std::min(nl, nr);
        while (a < b) {
            const long long m2 = (a + b) / 2;
            if (lp[m2] < rp[nr - 1 - m2]) a = m2 + 1; else b = m2;
        }
        const long long K = a;
        for (long long k = 0; k < K; ++k) std::swap(r[lo + lp[k]], r[lo + rp[nr - 1 - k]]);
        long long cut = K < nl ? lp[K] : len;
        if (K > 0) cut = std::min<long long>(cut, rp[nr - K]);
        cut += lo;
        if (cut <= nth) first = cut; else last = cut;
    }
    std::__insertion_sort(r + first, r + last, icmp);
}

template <int M>
long long kd_build_local(KdTree& t, std::vector<KdNode>& nodes, KdRec<M>* rec, long long base, long long start, long long end,
                         int32_t* scratch, int par_depth) {
    nodes.emplace_back();
    const long long node_index = (long long)nodes.size() - 1;
    nodes[(size_t)node_index].start = start;
    nodes[(size_t)node_index].end = end;
    const long long n = end - start;
    if (n <= t.leafsize) return node_index;
    KdRec<M>* r = rec + (start - base);
    double mx[M], mn[M];
    for (int i = 0; i < M; ++i) mx[i] = mn[i] = r[0].c[i];
    for (long long j = 1; j < n; ++j)
        for (int i = 0; i < M; ++i) {
            const double v = r[j].c[i];
            mx[i] = mx[i] > v ? mx[i] : v;
            mn[i] = mn[i] < v ? mn[i] : v;
        }
    int d = 0;
    double size = 0.0;
    for (int i = 0; i < M; ++i)
        if (mx[i] - mn[i] > size) { d = i; size = mx[i] - mn[i]; }
    if (mx[d] == mn[d]) return node_index;
    const long long half = n / 2;
    rec_nth_element<M>(r, half, n, d, scratch);
    double split = r[half].c[d];
    long long p = 0, q = half - 1;                                    // (see kd_build for the pass and its shortcut)
    while (p <= q) {
        if (r[p].c[d] < split) ++p;
        else if (r[q].c[d] >= split) --q;
        else { std::swap(r[p], r[q]); ++p; --q; }
    }
    if (p == 0) {
        split = std::nextafter(split, HUGE_VAL);
        q = n - 1;
        while (p <= q) {
            if (r[p].c[d] < split) ++p;
            else if (r[q].c[d] >= split) --q;
            else { std::swap(r[p], r[q]); ++p; --q; }
        }
    }
    long long less = -1, greater = -1;
    const bool fork = par_depth > 0 && n > kd_fork_min();
    kd_children(t, nodes, fork, [&](std::vector<KdNode>& into, bool lesser) {
        if (!lesser) return kd_build_local<M>(t, into, rec, base, start + p, end, scratch, par_depth - 1);
        if (!fork) return kd_build_local<M>(t, into, rec, base, start, start + p, scratch, par_depth - 1);
        std::vector<int32_t, KdRawAlloc<int32_t>> own((size_t)(2 * p));   // on another thread: list scratch of its own
        return kd_build_local<M>(t, into, rec, base, start, start + p, own.data(), par_depth - 1);
    }, less, greater);
    KdNode& nd = nodes[(size_t)node_index];
    nd.split_dim = d;
    nd.split = split;
    nd.less = less;
    nd.greater = greater;
    return node_index;
}

template <int M>
long long kd_build_on_copy(KdTree& t, std::vector<KdNode>& nodes, long long start, long long end, int par_depth) {
    const long long n = end - start;
    std::vector<KdRec<M>, KdRawAlloc<KdRec<M>>> rec((size_t)n);
    long long* indices = t.indices.data();
    for (long long j = 0; j < n; ++j) {
        const long long i = indices[start + j];
        for (int c = 0; c < M; ++c) rec[(size_t)j].c[c] = t.data[i * M + c];
        rec[(size_t)j].i = i;
    }
    std::vector<int32_t, KdRawAlloc<int32_t>> scratch((size_t)(2 * n));
    const long long root = kd_build_local<M>(t, nodes, rec.data(), start, start, end, scratch.data(), par_depth);
    for (long long j = 0; j < n; ++j) indices[start + j] = rec[(size_t)j].i;
    return root;
}

long long kd_build(KdTree& t, std::vector<KdNode>& nodes, long long start, long long end, double* maxes, double* mins, int par_depth,
                   int level) {
    const int m = t.m;
    if (end - start > t.leafsize && end - start <= kd_local_max()) {
        if (m == 1) return kd_build_on_copy<1>(t, nodes, start, end, par_depth);
        if (m == 2) return kd_build_on_copy<2>(t, nodes, start, end, par_depth);
        if (m == 3) return kd_build_on_copy<3>(t, nodes, start, end, par_depth);
    }
    const double* data = t.data;
    long long* indices = t.indices.data();
    nodes.emplace_back();
    const long long node_index = (long long)nodes.size() - 1;
    nodes[(size_t)node_index].start = start;
    nodes[(size_t)node_index].end = end;
    if (end - start <= t.leafsize) return node_index;                 // leaf
    // compact nodes: bounds from the node's own points
    // the large nodes' passes are cut into tasks of the pool (KdPool above): same bounds, same permutation
    const bool teamed = end - start >= kd_team_min() && end - start < 0x7fffffffLL && t.pooled && m <= 8;   // (32-bit positions in the lists)
    KdPool& team = KdPool::get();
    if (teamed) {
        const int nt = kd_pass_chunks(t, end - start);
        std::vector<double> bmx((size_t)nt * 8), bmn((size_t)nt * 8);
        team.parallel(nt, [&](int tid, int n_t) {
            const long long b = start + (end - start) * tid / n_t, e = start + (end - start) * (tid + 1) / n_t;
            double mx[8], mn[8];
            for (int i = 0; i < m; ++i) mx[i] = mn[i] = data[indices[b] * m + i];
            for (long long j = b + 1; j < e; ++j)
                for (int i = 0; i < m; ++i) {
                    const double v = data[indices[j] * m + i];
                    mx[i] = mx[i] > v ? mx[i] : v;
                    mn[i] = mn[i] < v ? mn[i] : v;
                }
            for (int i = 0; i < m; ++i) { bmx[(size_t)tid * 8 + i] = mx[i]; bmn[(size_t)tid * 8 + i] = mn[i]; }
        }, &t.failed);
        for (int i = 0; i < m; ++i) {
            maxes[i] = bmx[(size_t)i];
            mins[i] = bmn[(size_t)i];
            for (int t2 = 1; t2 < nt; ++t2) {
                const double a = bmx[(size_t)t2 * 8 + i], b2 = bmn[(size_t)t2 * 8 + i];
                maxes[i] = maxes[i] > a ? maxes[i] : a;
                mins[i] = mins[i] < b2 ? mins[i] : b2;
            }
        }
    } else {
        for (int i = 0; i < m; ++i) maxes[i] = mins[i] = data[indices[start] * m + i];
        for (long long j = start + 1; j < end; ++j)
            for (int i = 0; i < m; ++i) {
                const double v = data[indices[j] * m + i];
                maxes[i] = maxes[i] > v ? maxes[i] : v;
                mins[i] = mins[i] < v ? mins[i] : v;
            }
    }
    int d = 0;
    double size = 0.0;
    for (int i = 0; i < m; ++i)
        if (maxes[i] - mins[i] > size) { d = i; size = maxes[i] - mins[i]; }
    if (maxes[d] == mins[d]) return node_index;                       // all points identical: leaf
    // median split (balanced tree)
    const long long half = (end - start) / 2;
    // (Selection and partition on contiguous (coordinate, index) pairs instead of through the index array - the same comparison
    // results, hence the same permutation - was tried for the large nodes: no faster, 23 vs 21 ms per million points.)
    // the comparator of scipy 1.15.3 is the coordinate alone (no index tie-break: checked against the library's index array
    // on lattices - with one the leaves come out in another order), so equal coordinates fall where introselect leaves them
    auto cmp = [data, m, d](long long a, long long b) { return data[a * m + d] < data[b * m + d]; };
    auto key = [data, m, d](long long a) { return data[a * m + d]; };
    if (teamed) team_nth_element(t, indices, start, start + half, end, key, std::max<long long>(kd_team_min() / 4, 16));
    else std::nth_element(indices + start, indices + start + half, indices + end, cmp);
    double split = data[indices[start + half] * m + d];
    // scipy's two-pointer pass "< split | >= split" over the whole range.  After the selection everything from position `half` on
    // is >= split (std::nth_element's postcondition): the pass would walk q down through that half without a swap - it starts
    // where that walk ends.  Same swaps, same result, half the pass.
    long long p = start, q = start + half - 1;
    if (teamed) {
        // the same pass in tasks: its k-th swap is (k-th key >= split from the left, k-th key < split from the right) while the
        // former lies left of the latter, and it ends with p on the first key >= split - the number of keys below the split.
        // After a selection only keys EQUAL to the split stand left of position `half`, so the left list is short: the
        // team counts and lists, one thread pairs.
        const int nt = kd_pass_chunks(t, half);
        std::vector<std::vector<long long>> lefts((size_t)nt);
        std::vector<long long> below((size_t)nt, 0);
        team.parallel(nt, [&](int tid, int n_t) {
            const long long b = start + half * tid / n_t, e = start + half * (tid + 1) / n_t;
            long long nb = 0;
            for (long long i = b; i < e; ++i) {
                if (data[indices[i] * m + d] < split) ++nb;
                else lefts[(size_t)tid].push_back(i);
            }
            below[(size_t)tid] = nb;
        }, &t.failed);
        long long n_below = 0;
        for (int t2 = 0; t2 < nt; ++t2) n_below += below[(size_t)t2];
        long long r = q;                                              // walks down over the keys < split
        bool crossed = false;
        for (int t2 = 0; t2 < nt && !crossed; ++t2)
            for (long long l : lefts[(size_t)t2]) {
                while (r > l && !(data[indices[r] * m + d] < split)) --r;
                if (r <= l) { crossed = true; break; }
                std::swap(indices[l], indices[r]);
                --r;
            }
        p = start + n_below;
    } else
    while (p <= q) {
        if (data[indices[p] * m + d] < split) ++p;
        else if (data[indices[q] * m + d] >= split) --q;
        else { std::swap(indices[p], indices[q]); ++p; --q; }
    }
    if (p == start) {
        // Nothing below the split: the median IS the node's minimum along d (more than half of its points share it - duplicated
        // spots, or a ragged tissue edge).  scipy 1.15.3 then splits just ABOVE the minimum - the tree reports
        // split == nextafter(minimum, +inf), with every point at the minimum in the lesser child - by the same two-pointer
        // pass over the whole range.  (Rounds 1-5 restated the older "slide one point over" rule here; it never fired on a
        // full lattice, and did on heavily duplicated coordinates - caught by the duplicates sweep of tests/test_host.py.)
        split = std::nextafter(split, HUGE_VAL);
        q = end - 1;
        while (p <= q) {
            if (data[indices[p] * m + d] < split) ++p;
            else if (data[indices[q] * m + d] >= split) --q;
            else { std::swap(indices[p], indices[q]); ++p; --q; }
        }
    }
    long long less = -1, greater = -1;
    kd_children(t, nodes, par_depth > 0 && end - start > kd_fork_min(), [&](std::vector<KdNode>& into, bool lesser) {
        if (!lesser) return kd_build(t, into, p, end, maxes, mins, par_depth - 1, level + 1);
        std::vector<double> mx((size_t)m), mn((size_t)m);            // (possibly on another thread: bounds scratch of its own)
        return kd_build(t, into, start, p, mx.data(), mn.data(), par_depth - 1, level + 1);
    }, less, greater);
    KdNode& nd = nodes[(size_t)node_index];                           // (the vector may have moved)
    nd.split_dim = d;
    nd.split = split;
    nd.less = less;
    nd.greater = greater;
    return node_index;
}

// scipy's heap (ckdtree/src/ordered.h): binary min-heap on `priority`, strict comparisons
struct HeapItem {
    double priority;
    long long payload;
};
struct Heap {
    std::vector<HeapItem> h;
    long long n = 0;
    void push(const HeapItem& item) {
        ++n;
        if ((long long)h.size() < n) h.resize((size_t)(2 * h.size() + 1));
        long long i = n - 1;
        h[(size_t)i] = item;
        while (i > 0 && h[(size_t)i].priority < h[(size_t)((i - 1) / 2)].priority) {
            std::swap(h[(size_t)((i - 1) / 2)], h[(size_t)i]);
            i = (i - 1) / 2;
        }
    }
    const HeapItem& peek() const { return h[0]; }
    void remove() {
        h[0] = h[(size_t)(n - 1)];
        --n;
        const long long nn = n;
        long long i = 0, j = 1, k = 2;
        while ((j < nn && h[(size_t)i].priority > h[(size_t)j].priority) || (k < nn && h[(size_t)i].priority > h[(size_t)k].priority)) {
            const long long l = (k < nn && h[(size_t)j].priority > h[(size_t)k].priority) ? k : j;
            std::swap(h[(size_t)l], h[(size_t)i]);
            i = l;
            j = 2 * i + 1;
            k = 2 * i + 2;
        }
    }
    HeapItem pop() {
        const HeapItem it = h[0];
        remove();
        return it;
    }
};

struct NodeInfo {
    long long node;
    double min_distance;
    double side[8];
};

void kd_query_one(const KdTree& t, const double* x, int kmax, int64_t* out_idx, std::vector<NodeInfo>& pool, Heap& q,
                  Heap& neighbors) {
    const int m = t.m;
    pool.clear();
    q.n = 0;
    neighbors.n = 0;
    double upper = HUGE_VAL;
    pool.push_back(NodeInfo{0, 0.0, {0.0, 0.0, 0.0, 0.0, 0.0, 0.0, 0.0, 0.0}});
    {
        NodeInfo& r = pool[0];
        for (int i = 0; i < m; ++i) {
            // distance from x to the interval [min, max] of the whole data set, to the power p = 2
            double sd = std::max(0.0, std::max(t.mins[(size_t)i] - x[i], x[i] - t.maxes[(size_t)i]));
            sd = sd * sd;
            r.min_distance += sd - r.side[i];
            r.side[i] = sd;
        }
    }
    long long cur = 0;                                                // index into pool
    for (;;) {
        const KdNode& nd = t.nodes[(size_t)pool[(size_t)cur].node];
        if (nd.split_dim == -1) {
            for (long long i = nd.start; i < nd.end; ++i) {
                const double* y = t.data + t.indices[(size_t)i] * m;
                double d = 0.0;
                for (int a = 0; a < m; ++a) {
                    const double diff = y[a] - x[a];
                    d += diff * diff;
                }
                if (d < upper) {
                    if (neighbors.n == kmax) neighbors.remove();
                    neighbors.push(HeapItem{-d, t.indices[(size_t)i]});
                    if (neighbors.n == kmax) upper = -neighbors.peek().priority;
                }
            }
            if (q.n == 0) break;
            cur = q.pop().payload;
        } else {
            if (pool[(size_t)cur].min_distance > upper) break;         // the nearest remaining cell is too far: done
            pool.push_back(pool[(size_t)cur]);                         // ni2 = copy of ni1 (side distances included)
            long long far = (long long)pool.size() - 1;
            const int sd_dim = (int)nd.split_dim;
            double side_distance;
            if (x[sd_dim] < nd.split) {
                pool[(size_t)cur].node = nd.less;
                pool[(size_t)far].node = nd.greater;
                side_distance = nd.split - x[sd_dim];
            } else {
                pool[(size_t)cur].node = nd.greater;
                pool[(size_t)far].node = nd.less;
                side_distance = x[sd_dim] - nd.split;
            }
            side_distance = side_distance * side_distance;
            NodeInfo& f = pool[(size_t)far];
            f.min_distance += side_distance - f.side[sd_dim];
            f.side[sd_dim] = side_distance;
            if (pool[(size_t)cur].min_distance > pool[(size_t)far].min_distance) std::swap(cur, far);   // ni1 = the closer one
            if (pool[(size_t)far].min_distance <= upper) q.push(HeapItem{pool[(size_t)far].min_distance, far});
        }
    }
    // neighbours, nearest first (heap order decides among equal distances, as in the library)
    const long long nnb = neighbors.n;
    std::vector<HeapItem> sorted((size_t)kmax);
    for (long long i = neighbors.n - 1; i >= 0; --i) sorted[(size_t)i] = neighbors.pop();
    for (int i = 0; i < kmax; ++i) out_idx[i] = i < nnb ? sorted[(size_t)i].payload : -1;
}

}  // namespace
}  // namespace fdx

namespace fdx {
namespace {
// tree of all n points, then the queries of `rows` (NULL: every point, in order) - idx_out row j holds the answer for rows[j]
void kd_build_tree(KdTree& t, const double* coords, int64_t n, int32_t dim);
int ckdtree_query_host(const KdTree& t, int32_t kk, const int64_t* rows, int64_t n_rows, int64_t* idx_out, int64_t* tree_indices_out);

int ckdtree_knn_impl(const double* coords, int64_t n, int32_t dim, int32_t kk, const int64_t* rows, int64_t n_rows, int64_t* idx_out,
                     int64_t* tree_indices_out) {
    KdTree t;
    const bool trace = fdx::env("FDX_TRACE_HOST") != nullptr;
    const auto t0 = std::chrono::steady_clock::now();
    kd_build_tree(t, coords, n, dim);
    const auto t1 = std::chrono::steady_clock::now();
    const int rc = ckdtree_query_host(t, kk, rows, n_rows, idx_out, tree_indices_out);
    if (trace)
        std::fprintf(stderr, "[fdx-host] ckdtree: build %.1f ms (%lld nodes), %lld queries %.1f ms\n",
                     std::chrono::duration<double, std::milli>(t1 - t0).count(), (long long)t.nodes.size(), (long long)(rows ? n_rows : n),
                     std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t1).count());
    return rc;
}

void kd_build_tree(KdTree& t, const double* coords, int64_t n, int32_t dim) {
    t.data = coords;
    t.n = n;
    t.m = dim;
    t.indices.resize((size_t)n);
    t.maxes.assign((size_t)dim, 0.0);
    t.mins.assign((size_t)dim, 0.0);
    // identity index array and the bounds of the whole set: 3 ms of passes per million points when one thread makes them
    auto span = [&](long long b, long long e, double* mx, double* mn) {
        for (int a = 0; a < dim; ++a) mx[a] = mn[a] = coords[b * dim + a];
        for (long long i = b; i < e; ++i) {
            t.indices[(size_t)i] = i;
            for (int a = 0; a < dim; ++a) {
                const double v = coords[i * dim + a];
                mx[a] = std::max(mx[a], v);
                mn[a] = std::min(mn[a], v);
            }
        }
    };
    // the pool's threads are asked for where they can pay: from a few thousand points (the forks start at 32768, the passes'
    // tasks at 200000)
    KdPool& pool = KdPool::get();
    t.pooled = kd_thread_share() > 1 && n >= std::min<long long>(4096, kd_team_min()) && dim <= 8;
    if (t.pooled) pool.want(kd_pool_members());
    if (t.pooled && n >= kd_team_min()) {
        t.lpos.resize((size_t)n);                                     // (uninitialised: first touched by the tasks that fill them)
        t.rpos.resize((size_t)n);
        const int nt = pool.size();
        std::vector<double> bmx((size_t)nt * 8), bmn((size_t)nt * 8);
        pool.parallel(nt, [&](int tid, int n_t) { span(n * tid / n_t, n * (tid + 1) / n_t, &bmx[(size_t)tid * 8], &bmn[(size_t)tid * 8]); }, &t.failed);
        for (int a = 0; a < dim; ++a) {
            t.maxes[(size_t)a] = bmx[(size_t)a];
            t.mins[(size_t)a] = bmn[(size_t)a];
            for (int t2 = 1; t2 < nt; ++t2) {
                t.maxes[(size_t)a] = std::max(t.maxes[(size_t)a], bmx[(size_t)t2 * 8 + a]);
                t.mins[(size_t)a] = std::min(t.mins[(size_t)a], bmn[(size_t)t2 * 8 + a]);
            }
        }
    } else {
        span(0, n, t.maxes.data(), t.mins.data());
    }
    std::vector<double> mx(t.maxes), mn(t.mins);
    // the `less` child of every node above 32768 points becomes a task, five levels deep at most: 32 subtrees for a million points
    int par = t.pooled ? 5 : 0;
    if (const char* e = fdx::env("FDX_KDTREE_PAR_DEPTH")) par = t.pooled ? std::max(0, std::min(10, atoi(e))) : 0;
    if (fdx::exp_env("FDX_KDTREE_SERIAL_BUILD")) par = 0;
    std::vector<KdNode> top;
    top.reserve(1024);
    kd_build(t, top, 0, n, mx.data(), mn.data(), par, 0);
    // one vector: the top nodes, then the segments in the order they were handed in; tags become indices
    const size_t ns = t.segs.size();
    std::vector<long long> off(ns + 2, 0);
    off[1] = (long long)top.size();
    for (size_t i = 0; i < ns; ++i) off[i + 2] = off[i + 1] + (long long)t.segs[i].size();
    t.nodes.raw((size_t)off[ns + 1]);
    KdNode* flat = t.nodes.p;
    auto lay = [&](size_t si) {                                       // si = 0: the top nodes, else segment si - 1
        const std::vector<KdNode>& src = si == 0 ? top : t.segs[si - 1];
        const long long o = off[si];
        auto link = [&](long long l) { return l >= 0 ? l + o : l == -1 ? -1 : off[(size_t)(-l - 2) + 1]; };
        for (size_t i = 0; i < src.size(); ++i) {
            KdNode nd = src[i];
            nd.less = link(nd.less);
            nd.greater = link(nd.greater);
            ::new (flat + (size_t)o + i) KdNode(nd);
        }
    };
    if (t.failed.load()) throw std::bad_alloc();
    if (ns > 0 && t.pooled) pool.parallel(pool.size(), [&](int tid, int n_t) { for (size_t si = (size_t)tid; si < ns + 1; si += (size_t)n_t) lay(si); }, &t.failed);
    else for (size_t si = 0; si < ns + 1; ++si) lay(si);
    t.lpos = decltype(t.lpos)();
    t.rpos = decltype(t.rpos)();
    t.segs.clear();
}

int ckdtree_query_host(const KdTree& t, int32_t kk, const int64_t* rows, int64_t n_rows, int64_t* idx_out, int64_t* tree_indices_out) {
    const double* coords = t.data;
    const long long n = t.n;
    const int dim = t.m;
    if (tree_indices_out) std::memcpy(tree_indices_out, t.indices.data(), (size_t)n * sizeof(int64_t));
    // the queries are independent (the tree is read-only; every thread has its own node pool and heaps): contiguous ranges of a
    // few thousand points and more per host thread.  1000 x 1000 lattice points on 8 cores: 2.2 -> 0.8-1.0 s, of which 0.29 s the
    // serial build (its top levels are full passes over the points - bounds, introselect, partition; building the subtrees below
    // the second level on four threads was tried and returned nothing measurable).
    const long long nq = rows ? n_rows : n;
    auto run = [&](long long i0, long long i1) {
        std::vector<NodeInfo> pool;
        Heap q, nb;
        q.h.resize(12);
        nb.h.resize((size_t)kk);
        for (long long i = i0; i < i1; ++i) {
            const long long p = rows ? rows[i] : i;
            kd_query_one(t, coords + p * dim, kk, idx_out + i * kk, pool, q, nb);
        }
    };
    unsigned nt = kd_thread_share();
    nt = (unsigned)std::min<long long>(std::max(1u, std::min(nt, 128u)), std::max<long long>(1, nq / 4096));
    // a thread that cannot be started (process limits, W ranks x 32 threads) or a failed allocation inside a worker must not end
    // the process: what was started is joined, the rest of the range runs here
    std::vector<std::thread> th;
    std::vector<int> failed(nt, 0);
    const long long per = (nq + nt - 1) / std::max(1u, nt);
    long long next = 0;
    if (nt > 1) {
        for (unsigned k = 0; k < nt; ++k) {
            const long long i0 = (long long)k * per, i1 = std::min<long long>(nq, i0 + per);
            if (i0 >= i1) break;
            try {
                th.emplace_back([&, i0, i1, k] {
                    try { run(i0, i1); } catch (...) { failed[k] = 1; }
                });
            } catch (...) {
                break;                                   // this and the following ranges: serially below
            }
            next = i1;
        }
    }
    int rc = 0;
    try {
        if (next < nq) run(next, nq);
    } catch (...) {
        rc = fail(FDX_ERR_INVALID, "fdx_ckdtree_knn: out of memory in the query loop");
    }
    for (auto& x : th) x.join();
    for (unsigned k = 0; k < nt; ++k) {
        if (!failed[k]) continue;
        const long long i0 = (long long)k * per, i1 = std::min<long long>(nq, i0 + per);
        try { run(i0, i1); } catch (...) { rc = fail(FDX_ERR_INVALID, "fdx_ckdtree_knn: out of memory in the query loop"); }
    }
    return rc;
}
}  // namespace
}  // namespace fdx


// ================================================================================================================
// The QUERIES on the device.  The tree is built on the host (above: introselect's permutation has to be replayed step by step)
// and uploaded - 24 bytes per node, 4 per point; one lane answers one query with exactly the host's sequence of node visits,
// heap operations and comparisons (kd_query_one): best-first over the nodes with scipy's binary heap, the far node's record
// (node, distance, per-axis side distances) held IN the heap entry - the host keeps it in a pool and the index in the heap: the
// same priorities, the same sift sequences -, a leaf's points tested in index-array order against the strict bound.  Sums and
// products are rounded one by one (no contraction), as the host compiles them.  The two heaps of a lane live in a global work
// area laid out lane-interleaved (entry j of lane l at [j][l]): a wave's accesses to "its" entry j coalesce.  Entry counts are
// bounded (far-node heap: KD_QCAP; a deeper heap than that - never seen, the heap holds about one entry per tree level - raises
// a flag and the caller repeats the queries on the host).  1M lattice points, k = 6: 11-15 ms on the host's threads -> see DESIGN.
namespace fdx {
namespace {
constexpr int KD_QCAP = 48;
struct KdBounds { double mins[3], maxes[3]; };

// NBL: the k-nearest heap of a lane (kk <= 16 entries, by far the busiest of the two: a pop and a push per accepted candidate)
// lives in LDS, lane-interleaved like its global form - 12 bytes x kk x 256 lanes
template <int M, bool NBL>
__global__ __launch_bounds__(256) void kd_query_kernel(const double* __restrict__ coords, const int4* __restrict__ meta,
                                                       const double* __restrict__ split, const int* __restrict__ indices,
                                                       const KdBounds bnd, int kk, const long long* __restrict__ rows, long long n_rows,
                                                       long long* __restrict__ ids_out, double* __restrict__ q_pri, int* __restrict__ q_node,
                                                       double* __restrict__ q_side, double* __restrict__ nb_pri, int* __restrict__ nb_idx,
                                                       long long L, int* __restrict__ overflow) {
#pragma clang fp contract(off)
    const long long gid = blockIdx.x * 256LL + threadIdx.x;
    auto QP = [&](int j) -> double& { return q_pri[(size_t)j * L + gid]; };
    auto QN = [&](int j) -> int& { return q_node[(size_t)j * L + gid]; };
    auto QS = [&](int j, int a) -> double& { return q_side[((size_t)j * M + a) * L + gid]; };
    extern __shared__ __attribute__((aligned(16))) unsigned char kd_smem[];
    double* const l_pri = reinterpret_cast<double*>(kd_smem);
    int* const l_idx = reinterpret_cast<int*>(kd_smem + (size_t)kk * 256 * sizeof(double));
    auto NP = [&](int j) -> double& { return NBL ? l_pri[j * 256 + (int)threadIdx.x] : nb_pri[(size_t)j * L + gid]; };
    auto NI = [&](int j) -> int& { return NBL ? l_idx[j * 256 + (int)threadIdx.x] : nb_idx[(size_t)j * L + gid]; };
    for (long long r = gid; r < n_rows; r += L) {
        const long long self = rows ? rows[r] : r;
        double x[M];
#pragma unroll
        for (int a = 0; a < M; ++a) x[a] = coords[(size_t)self * M + a];
        int qn = 0, nn = 0;
        double upper = HUGE_VAL;
        bool over = false;
        // current node record
        int c_node = 0;
        double c_md = 0.0, c_side[M];
#pragma unroll
        for (int a = 0; a < M; ++a) {
            double sd = fmax(0.0, fmax(bnd.mins[a] - x[a], x[a] - bnd.maxes[a]));
            sd = sd * sd;
            c_md += sd - 0.0;
            c_side[a] = sd;
        }
        for (;;) {
            const int4 nd = meta[c_node];                          // x: split dim (-1 leaf), y / z: start / end or less / greater
            if (nd.x < 0) {
                for (int i = nd.y; i < nd.z; ++i) {
                    const int id = indices[i];
                    double d = 0.0;
#pragma unroll
                    for (int a = 0; a < M; ++a) {
                        const double diff = coords[(size_t)id * M + a] - x[a];
                        d += diff * diff;
                    }
                    if (d < upper) {
                        if (nn == kk) {                            // remove the root: the last entry sifts down from the top
                            const double mp = NP(nn - 1);
                            const int mi = NI(nn - 1);
                            --nn;
                            int h = 0;
                            for (;;) {
                                const int j = 2 * h + 1, k2 = 2 * h + 2;
                                const double pj = j < nn ? NP(j) : 0.0, pk = k2 < nn ? NP(k2) : 0.0;
                                if (!((j < nn && mp > pj) || (k2 < nn && mp > pk))) break;
                                const int l = (k2 < nn && pj > pk) ? k2 : j;
                                NP(h) = l == j ? pj : pk;
                                NI(h) = NI(l);
                                h = l;
                            }
                            if (nn > 0) { NP(h) = mp; NI(h) = mi; }
                        }
                        {                                          // push (-d, id): sift up from the end
                            const double np = -d;
                            int h = nn++;
                            while (h > 0) {
                                const int par = (h - 1) / 2;
                                const double pp = NP(par);
                                if (!(np < pp)) break;
                                NP(h) = pp;
                                NI(h) = NI(par);
                                h = par;
                            }
                            NP(h) = np;
                            NI(h) = id;
                        }
                        if (nn == kk) upper = -NP(0);
                    }
                }
                if (qn == 0) break;
                // pop the nearest pending node
                c_md = QP(0);
                c_node = QN(0);
#pragma unroll
                for (int a = 0; a < M; ++a) c_side[a] = QS(0, a);
                {
                    --qn;
                    const double mp = QP(qn);
                    const int mnode = QN(qn);
                    double ms[M];
#pragma unroll
                    for (int a = 0; a < M; ++a) ms[a] = QS(qn, a);
                    int h = 0;
                    for (;;) {
                        const int j = 2 * h + 1, k2 = 2 * h + 2;
                        const double pj = j < qn ? QP(j) : 0.0, pk = k2 < qn ? QP(k2) : 0.0;
                        if (!((j < qn && mp > pj) || (k2 < qn && mp > pk))) break;
                        const int l = (k2 < qn && pj > pk) ? k2 : j;
                        QP(h) = l == j ? pj : pk;
                        QN(h) = QN(l);
#pragma unroll
                        for (int a = 0; a < M; ++a) QS(h, a) = QS(l, a);
                        h = l;
                    }
                    if (qn > 0) {
                        QP(h) = mp;
                        QN(h) = mnode;
#pragma unroll
                        for (int a = 0; a < M; ++a) QS(h, a) = ms[a];
                    }
                }
            } else {
                if (c_md > upper) break;                           // the nearest remaining cell is too far: done
                const double sp = split[c_node];
                double xs = x[0];
#pragma unroll
                for (int a = 1; a < M; ++a) xs = nd.x == a ? x[a] : xs;
                int f_node;
                double side_distance;
                if (xs < sp) { c_node = nd.y; f_node = nd.z; side_distance = sp - xs; }
                else { c_node = nd.z; f_node = nd.y; side_distance = xs - sp; }
                side_distance = side_distance * side_distance;
                double f_side[M], f_md = c_md;
                double old = c_side[0];
#pragma unroll
                for (int a = 0; a < M; ++a) { f_side[a] = c_side[a]; old = nd.x == a ? c_side[a] : old; }
                f_md += side_distance - old;
#pragma unroll
                for (int a = 0; a < M; ++a) f_side[a] = nd.x == a ? side_distance : f_side[a];
                if (c_md > f_md) {                                 // the closer one is followed
                    const double t = c_md; c_md = f_md; f_md = t;
                    const int tn = c_node; c_node = f_node; f_node = tn;
#pragma unroll
                    for (int a = 0; a < M; ++a) { const double ts = c_side[a]; c_side[a] = f_side[a]; f_side[a] = ts; }
                }
                if (f_md <= upper) {
                    if (qn >= KD_QCAP) { over = true; break; }
                    int h = qn++;
                    while (h > 0) {
                        const int par = (h - 1) / 2;
                        const double pp = QP(par);
                        if (!(f_md < pp)) break;
                        QP(h) = pp;
                        QN(h) = QN(par);
#pragma unroll
                        for (int a = 0; a < M; ++a) QS(h, a) = QS(par, a);
                        h = par;
                    }
                    QP(h) = f_md;
                    QN(h) = f_node;
#pragma unroll
                    for (int a = 0; a < M; ++a) QS(h, a) = f_side[a];
                }
            }
        }
        if (over) {
            atomicOr(overflow, 1);
            for (int i = 0; i < kk; ++i) ids_out[(size_t)r * kk + i] = -1;
            continue;
        }
        // nearest first: the heap gives them farthest first
        const int found = nn;
        for (int i = found; i < kk; ++i) ids_out[(size_t)r * kk + i] = -1;
        for (int i = found - 1; i >= 0; --i) {
            ids_out[(size_t)r * kk + i] = (long long)NI(0);
            const double mp = NP(nn - 1);
            const int mi = NI(nn - 1);
            --nn;
            int h = 0;
            for (;;) {
                const int j = 2 * h + 1, k2 = 2 * h + 2;
                const double pj = j < nn ? NP(j) : 0.0, pk = k2 < nn ? NP(k2) : 0.0;
                if (!((j < nn && mp > pj) || (k2 < nn && mp > pk))) break;
                const int l = (k2 < nn && pj > pk) ? k2 : j;
                NP(h) = l == j ? pj : pk;
                NI(h) = NI(l);
                h = l;
            }
            if (nn > 0) { NP(h) = mp; NI(h) = mi; }
        }
    }
}
}  // namespace

// ids_dev (n_rows, kk) int64, device: the answers for rows_host (NULL: every point in order), nearest first, self included, -1 padded.
// coords_dev: the same (n, dim) float64 array as coords_host.  Queries on the device for 1-3 coordinates (FDX_KDTREE_HOST_QUERIES=1
// or a deeper far-node heap than the work area holds: on the host's threads, then uploaded).
// a recycled pinned buffer for the lifetime of one call (uploads / downloads from pageable memory make the driver pin the caller's
// pages, and unmapping such pages later - a freed vector, a collected numpy array - evicts the process's GPU queues for 10-25 ms)
struct PinnedScope {
    void* p = nullptr;
    size_t cap = 0;
    int get(size_t bytes) { p = pinned_buffer_get(bytes, &cap); return p ? 0 : fail(FDX_ERR_HIP, "ckdtree: pinned host buffer"); }
    ~PinnedScope() { if (p) pinned_buffer_put(p, cap); }
};

// A tree whose build was started ahead (fdx_ckdtree_prebuild): the tie remedy of a fit knows it will need the tree the moment the
// device reports ties, ~4 ms before it gets round to asking for the lists (the device's own lists come first, behind the still
// running sketch) - the 21-24 ms host build runs beside that.  One pending build per process, matched by its coordinates.
struct KdPrebuilt {
    const double* key_host = nullptr;
    const double* key_dev = nullptr;
    long long n = 0;
    int dim = 0;
    PinnedScope pin;                     // coordinates fetched from the device (key_host == NULL)
    KdTree t;
    std::thread th;
    bool failed = false;
};
static std::mutex g_pre_mu;
static std::unique_ptr<KdPrebuilt> g_pre;

static void kd_prebuilt_drop(std::unique_ptr<KdPrebuilt>& p) {
    if (p && p->th.joinable()) p->th.join();
    p.reset();
}

// the k-nearest queries of `rows_host` (NULL: every point) on a tree that lies on the device: ids_dev gets kk ids per query;
// *over = 1 when a lane's far-node heap was too deep (the caller repeats the queries on the host)
int kd_queries_device(const double* coords_dev, long long n, int dim, int kk, const int4* meta_d, const double* split_d, const int* idx_d,
                      const double* mins, const double* maxes, const long long* rows_host, long long nq, long long* ids_dev, int* over,
                      hipStream_t st) {
    (void)n;
    const long long L = std::min<long long>((nq + 255) / 256, 1024) * 256;   // (twice the lanes in flight - the kernel has the registers for it - measured no faster: 1.85 ms either way)
    DevBuf d_rows, d_over, qp, qnode, qs, np, ni;
    PinnedScope stage;
    FDX_TRY(d_over.alloc(4));
    FDX_TRY(qp.alloc((size_t)KD_QCAP * L * 8));
    FDX_TRY(qnode.alloc((size_t)KD_QCAP * L * 4));
    FDX_TRY(qs.alloc((size_t)KD_QCAP * dim * L * 8));
    FDX_TRY(np.alloc((size_t)kk * L * 8));
    FDX_TRY(ni.alloc((size_t)kk * L * 4));
    FDX_HIP(hipMemsetAsync(d_over.p, 0, 4, st));
    if (rows_host) {
        FDX_TRY(stage.get((size_t)nq * 8));
        std::memcpy(stage.p, rows_host, (size_t)nq * 8);
        FDX_TRY(d_rows.alloc((size_t)nq * 8));
        FDX_HIP(hipMemcpyAsync(d_rows.p, stage.p, (size_t)nq * 8, hipMemcpyHostToDevice, st));
    }
    KdBounds bnd{};
    for (int a = 0; a < dim; ++a) { bnd.mins[a] = mins[a]; bnd.maxes[a] = maxes[a]; }
    const dim3 grid((unsigned)(L / 256)), blk(256);
    const long long* rows_d = rows_host ? d_rows.as<long long>() : nullptr;
    const bool nbl = kk <= 16;
    const size_t nb_lds = nbl ? (size_t)kk * 256 * 12 : 0;
#define FDX_KD_LAUNCH(MM)                                                                                                              \
    do {                                                                                                                               \
        if (nbl)                                                                                                                       \
            hipLaunchKernelGGL((kd_query_kernel<MM, true>), grid, blk, nb_lds, st, coords_dev, meta_d, split_d, idx_d, bnd, kk, rows_d, nq,   \
                               ids_dev, qp.as<double>(), qnode.as<int>(), qs.as<double>(), np.as<double>(), ni.as<int>(), L, d_over.as<int>()); \
        else                                                                                                                           \
            hipLaunchKernelGGL((kd_query_kernel<MM, false>), grid, blk, 0, st, coords_dev, meta_d, split_d, idx_d, bnd, kk, rows_d, nq,       \
                               ids_dev, qp.as<double>(), qnode.as<int>(), qs.as<double>(), np.as<double>(), ni.as<int>(), L, d_over.as<int>()); \
    } while (0)
    if (dim == 1) FDX_KD_LAUNCH(1);
    else if (dim == 2) FDX_KD_LAUNCH(2);
    else FDX_KD_LAUNCH(3);
#undef FDX_KD_LAUNCH
    FDX_CHECK_LAUNCH();
    *over = 0;
    FDX_HIP(hipMemcpyAsync(over, d_over.p, 4, hipMemcpyDeviceToHost, st));
    FDX_HIP(hipStreamSynchronize(st));                     // the staging and the work areas are this function's; the flag decides the route
    return 0;
}

static bool kd_device_build_serves(const double* coords_dev, long long n, int dim) {
    return coords_dev && dim <= 3 && n < 0x3fffffffLL && g_kd_device_build.load() != 0 && !fdx::env("FDX_KDTREE_HOST_QUERIES");
}

int ckdtree_prebuild(const double* coords_host, const double* coords_dev, long long n, int dim) {
    if (kd_device_build_serves(coords_dev, n, dim)) return 0;          // the tree is built on the device when the lists are asked for
    std::unique_ptr<KdPrebuilt> old;
    {
        std::lock_guard<std::mutex> lk(g_pre_mu);
        old = std::move(g_pre);
    }
    kd_prebuilt_drop(old);
    auto p = std::make_unique<KdPrebuilt>();
    p->key_host = coords_host; p->key_dev = coords_dev; p->n = n; p->dim = dim;
    const double* src = coords_host;
    if (!src) {
        // on the library's side stream: the caller's stream may still be busy (the sketch of the stopped fit), the coordinates
        // are nobody's output
        hipStream_t ss = library_side_stream();
        const auto tf0 = std::chrono::steady_clock::now();
        FDX_TRY(p->pin.get((size_t)n * dim * sizeof(double)));
        FDX_HIP(hipMemcpyAsync(p->pin.p, coords_dev, (size_t)n * dim * sizeof(double), hipMemcpyDeviceToHost, ss));
        FDX_HIP(hipStreamSynchronize(ss));
        src = static_cast<const double*>(p->pin.p);
        if (fdx::env("FDX_TRACE_HOST"))
            std::fprintf(stderr, "[fdx-host] ckdtree prebuild: coordinates fetched in %.2f ms\n",
                         std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - tf0).count());
    }
    KdPrebuilt* raw = p.get();
    try {
        p->th = std::thread([raw, src, n, dim] {
            const auto tb0 = std::chrono::steady_clock::now();
            try { kd_build_tree(raw->t, src, n, dim); } catch (...) { raw->failed = true; }
            if (fdx::env("FDX_TRACE_HOST"))
                std::fprintf(stderr, "[fdx-host] ckdtree prebuild: tree of %lld points built in %.2f ms\n", n,
                             std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - tb0).count());
        });
    } catch (const std::system_error&) {
        return 0;                        // no thread: the lists call builds the tree itself
    }
    std::lock_guard<std::mutex> lk(g_pre_mu);
    g_pre = std::move(p);
    return 0;
}

int ckdtree_lists_device(const double* coords_host_in, const double* coords_dev, long long n, int dim, int kk, const long long* rows_host,
                         long long n_rows, long long* ids_dev, hipStream_t st) {
    const long long nq = rows_host ? n_rows : n;
    if (nq == 0) return 0;
    if (kd_device_build_serves(coords_dev, n, dim)) {
        // tree AND queries on the device (kdtree_build_dev.cpp); a selection that would leave introselect's partition loop, or a
        // tree deeper than the level cap, sends the call down the host build below
        const bool trace = fdx::env("FDX_TRACE_HOST") != nullptr;
        const auto t0 = std::chrono::steady_clock::now();
        KdDeviceTree dt;
        FDX_TRY(kd_build_device(coords_dev, n, dim, &dt, st));
        const auto t1 = std::chrono::steady_clock::now();
        if (!dt.overflow) {
            int over = 0;
            FDX_TRY(kd_queries_device(coords_dev, n, dim, kk, dt.meta.as<int4>(), dt.split.as<double>(), dt.idx.as<int>(), dt.mins, dt.maxes,
                                      rows_host, nq, ids_dev, &over, st));
            if (trace)
                std::fprintf(stderr, "[fdx-host] ckdtree: tree built on the device in %.2f ms (%d nodes, %d levels), %lld queries %.2f ms%s\n",
                             std::chrono::duration<double, std::milli>(t1 - t0).count(), dt.n_nodes, dt.levels, nq,
                             std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t1).count(),
                             over ? " (far-node heap too deep: repeated on the host)" : "");
            if (!over) return 0;
        } else if (trace) {
            std::fprintf(stderr, "[fdx-host] ckdtree: the device build gave up (selection depth / level cap): host build\n");
        }
    }
    std::unique_ptr<KdPrebuilt> pre;
    {
        std::lock_guard<std::mutex> lk(g_pre_mu);
        if (g_pre && g_pre->key_host == coords_host_in && g_pre->key_dev == coords_dev && g_pre->n == n && g_pre->dim == dim) pre = std::move(g_pre);
    }
    const auto tj0 = std::chrono::steady_clock::now();
    if (pre) {
        pre->th.join();
        if (pre->failed) pre.reset();
    }
    const double join_ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - tj0).count();
    // coords_host NULL: the coordinates are fetched here, into pinned memory
    PinnedScope pin_coords;
    const double* coords_host = coords_host_in;
    if (pre && !coords_host) coords_host = static_cast<const double*>(pre->pin.p);
    if (!coords_host) {
        FDX_TRY(pin_coords.get((size_t)n * dim * sizeof(double)));
        FDX_HIP(hipMemcpyAsync(pin_coords.p, coords_dev, (size_t)n * dim * sizeof(double), hipMemcpyDeviceToHost, st));
        FDX_HIP(hipStreamSynchronize(st));
        coords_host = static_cast<const double*>(pin_coords.p);
    }
    const bool trace = fdx::env("FDX_TRACE_HOST") != nullptr;
    const auto t0 = std::chrono::steady_clock::now();
    KdTree t_own;
    if (!pre) kd_build_tree(t_own, coords_host, n, dim);
    KdTree& t = pre ? pre->t : t_own;
    const auto t1 = std::chrono::steady_clock::now();
    bool on_device = dim <= 3 && n < 0x7fffff00LL && (long long)t.nodes.size() < 0x7fffff00LL && !fdx::env("FDX_KDTREE_HOST_QUERIES");
    if (on_device) {
        const size_t nn = t.nodes.size();
        // staging in pinned memory: [meta | split | indices | rows]
        const size_t o_split = nn * sizeof(int4), o_idx = o_split + nn * sizeof(double), o_rows = o_idx + (((size_t)n * 4 + 15) & ~(size_t)15);
        PinnedScope stage;
        FDX_TRY(stage.get(o_rows + (rows_host ? (size_t)nq * 8 : 0) + 64));
        char* sp = static_cast<char*>(stage.p);
        int4* meta = reinterpret_cast<int4*>(sp);
        double* split = reinterpret_cast<double*>(sp + o_split);
        int* idx32 = reinterpret_cast<int*>(sp + o_idx);
        auto stage_span = [&](int tid, int n_t) {                     // (1 ms per million points on one thread)
            for (size_t i = nn * (size_t)tid / (size_t)n_t, e = nn * ((size_t)tid + 1) / (size_t)n_t; i < e; ++i) {
                const KdNode& nd = t.nodes[i];
                const bool leaf = nd.split_dim == -1;
                meta[i] = make_int4(leaf ? -1 : (int)nd.split_dim, (int)(leaf ? nd.start : nd.less), (int)(leaf ? nd.end : nd.greater), 0);
                split[i] = nd.split;
            }
            for (long long i = n * tid / n_t, e = n * (tid + 1) / n_t; i < e; ++i) idx32[(size_t)i] = (int)t.indices[(size_t)i];
        };
        if (n >= kd_team_min() && kd_thread_share() > 1) {
            KdPool& pool = KdPool::get();
            pool.want(kd_pool_members());
            std::atomic<bool> stage_failed{false};                   // (nothing in the span allocates)
            pool.parallel(pool.size(), stage_span, &stage_failed);
        } else {
            stage_span(0, 1);
        }
        DevBuf d_meta, d_split, d_idx;
        FDX_TRY(d_meta.alloc(nn * sizeof(int4)));
        FDX_TRY(d_split.alloc(nn * sizeof(double)));
        FDX_TRY(d_idx.alloc((size_t)n * 4));
        FDX_HIP(hipMemcpyAsync(d_meta.p, meta, nn * sizeof(int4), hipMemcpyHostToDevice, st));
        FDX_HIP(hipMemcpyAsync(d_split.p, split, nn * sizeof(double), hipMemcpyHostToDevice, st));
        FDX_HIP(hipMemcpyAsync(d_idx.p, idx32, (size_t)n * 4, hipMemcpyHostToDevice, st));
        const auto ts1 = std::chrono::steady_clock::now();
        int over = 0;
        FDX_TRY(kd_queries_device(coords_dev, n, dim, kk, d_meta.as<int4>(), d_split.as<double>(), d_idx.as<int>(), t.mins.data(), t.maxes.data(),
                                  rows_host, nq, ids_dev, &over, st));
        if (trace)
            std::fprintf(stderr, "[fdx-host] ckdtree: waited %.1f ms for the tree, build here %.1f ms (%lld nodes), staging + uploads queued %.2f ms, %lld queries on the device %.2f ms%s\n",
                         join_ms, std::chrono::duration<double, std::milli>(t1 - t0).count(), (long long)nn,
                         std::chrono::duration<double, std::milli>(ts1 - t1).count(), nq,
                         std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - ts1).count(),
                         over ? " (far-node heap too deep: repeated on the host)" : "");
        if (!over) return 0;
    }
    std::vector<int64_t> ids((size_t)nq * kk);
    FDX_TRY(ckdtree_query_host(t, kk, (const int64_t*)rows_host, nq, ids.data(), nullptr));
    FDX_HIP(hipMemcpyAsync(ids_dev, ids.data(), ids.size() * 8, hipMemcpyHostToDevice, st));
    FDX_HIP(hipStreamSynchronize(st));
    return 0;
}
}  // namespace fdx

// include/fdx.h
extern "C" int fdx_ckdtree_prebuild(const double* coords_host, const double* coords_dev, int64_t n, int32_t dim) {
    using namespace fdx;
    FDX_REQUIRE((coords_host || coords_dev) && n >= 1 && dim >= 1 && dim <= 8, "fdx_ckdtree_prebuild: bad arguments (1 to 8 coordinates)");
    try {
        return fdx::ckdtree_prebuild(coords_host, coords_dev, n, dim);
    } catch (...) {
        return fail(FDX_ERR_INVALID, "fdx_ckdtree_prebuild: out of memory");
    }
}

extern "C" int fdx_kdtree_set_threads(int32_t threads) {
    fdx::g_kd_threads.store(threads > 0 ? threads : 0);
    return 0;
}

extern "C" int fdx_kdtree_tune(int32_t what, int64_t points) {
    FDX_REQUIRE(what >= 0 && what <= 2, "fdx_kdtree_tune: what is 0 (team_min), 1 (local_max) or 2 (device build)");
    if (what == 0) fdx::g_kd_team_min.store(points > 0 ? std::max<long long>(points, 64) : 0);
    else if (what == 1) fdx::g_kd_local_max.store(points);
    else fdx::g_kd_device_build.store(points != 0 ? 1 : 0);
    return 0;
}

extern "C" int fdx_ckdtree_indices_dev(const double* coords_dev, int64_t n, int32_t dim, int64_t* indices_out, int32_t* info_out, void* stream) {
    using namespace fdx;
    FDX_REQUIRE(coords_dev && indices_out && n >= 1 && dim >= 1 && dim <= 3, "fdx_ckdtree_indices_dev: bad arguments (1 to 3 coordinates)");
    hipStream_t st = (hipStream_t)stream;
    PoolStream pool_stream(st);
    KdDeviceTree dt;
    FDX_TRY(kd_build_device(coords_dev, n, dim, &dt, st));
    std::vector<int> h((size_t)n);
    FDX_HIP(hipMemcpyAsync(h.data(), dt.idx.p, (size_t)n * sizeof(int), hipMemcpyDeviceToHost, st));
    FDX_HIP(hipStreamSynchronize(st));
    for (int64_t i = 0; i < n; ++i) indices_out[i] = h[(size_t)i];
    if (info_out) { info_out[0] = dt.n_nodes; info_out[1] = dt.levels; info_out[2] = dt.overflow ? 1 : 0; }
    return 0;
}

extern "C" int fdx_ckdtree_knn(const double* coords, int64_t n, int32_t dim, int32_t kk, int64_t* idx_out, int64_t* tree_indices_out) {
    using namespace fdx;
    FDX_REQUIRE(coords && idx_out && n >= 1 && dim >= 1 && dim <= 8 && kk >= 1, "fdx_ckdtree_knn: bad arguments (1 to 8 coordinates)");
    try {
        return ckdtree_knn_impl(coords, n, dim, kk, nullptr, 0, idx_out, tree_indices_out);
    } catch (...) {
        return fail(FDX_ERR_INVALID, "fdx_ckdtree_knn: out of memory");
    }
}

extern "C" int fdx_ckdtree_knn_rows(const double* coords, int64_t n, int32_t dim, int32_t kk, const int64_t* rows, int64_t n_rows,
                                    int64_t* idx_out) {
    using namespace fdx;
    FDX_REQUIRE(coords && n >= 1 && dim >= 1 && dim <= 8 && kk >= 1 && n_rows >= 0 && (n_rows == 0 || (rows && idx_out)),
                "fdx_ckdtree_knn_rows: bad arguments (1 to 8 coordinates)");
    for (int64_t j = 0; j < n_rows; ++j) FDX_REQUIRE(rows[j] >= 0 && rows[j] < n, "fdx_ckdtree_knn_rows: row index out of range");
    if (n_rows == 0) return 0;
    try {
        return ckdtree_knn_impl(coords, n, dim, kk, rows, n_rows, idx_out, nullptr);
    } catch (...) {
        return fail(FDX_ERR_INVALID, "fdx_ckdtree_knn_rows: out of memory");
    }
}
