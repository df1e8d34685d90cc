// The restated cKDTree (kdtree_order.cpp: scipy.spatial.cKDTree's build, utils/graph.py:60) BUILT ON THE DEVICE for 1-3 coordinates.
//
// What has to come out is scipy's index array and node structure exactly, i.e. libstdc++'s introselect replayed swap for swap at
// every node (the order of equidistant neighbours hangs on it).  The host build does that with a pool of threads in 9-11 ms per
// million points; here every node of a level is processed at once, level by level, a TEAM of threads per node:
//   bounds       team reduction over the node's points (compact_nodes: the node's own points)
//   selection    std::nth_element's loop - median of three to the front (one thread), __unguarded_partition (the team), narrow to
//                the side holding nth, until three elements are left (insertion sort, one thread); the depth budget of 2 log2(n)
//                partitions is kept, and a node that would exhaust it (libstdc++ then switches to heap select) raises a flag: the
//                caller builds on the host instead
//   partition    in the list form of kdtree_order.cpp (team_unguarded_partition): the serial scan's k-th swap is (k-th position
//                from the left whose key stops the left pointer, k-th from the right whose key stops the right pointer) while the
//                former lies left of the latter.  Every thread lists the stops of its contiguous chunk, two scans give the ranks,
//                the number of swaps K is counted in parallel (a left stop at position x with rank k is swapped iff more than k
//                right stops lie beyond x - a prefix of the left stops), and each thread swaps its own left stops with the right
//                stops of equal rank (found by rank through the scan)
//   split        scipy's "< split | >= split" pass in the same form, the split just above the minimum when the median is the minimum
// Teams are workgroups of 1024 / 256 / 64 threads by node size (a 64-thread workgroup is one wave: its barriers cost nothing); all
// of a node's data stays in global memory - the bottom levels have thousands of nodes in flight and live out of L2, the top levels
// are streaming passes.  The level loop is queued without reading anything back: level L has at most 2^L nodes, a launch covers
// that many and the surplus workgroups leave at once; the queue lengths, the node count and the flags are read at the end.
//
// STATE (round 6): exact - scipy's index array on lattices, clouds, heavy duplicates, sorted input, 1-3 coordinates, up to a
// million points (tests/test_gpu_stages.py) - and the default for 1-3 coordinates (fdx_kdtree_tune(2, 0) selects the host's thread
// pool): a million lattice points in 5.3 ms against the pool's 9-11.  Nodes above 100000 points are not one workgroup's work
// (a million points through one compute unit: 2.5 ms for the root, 4.5 for its two children): each of their passes is three
// launches over 64 workgroups per node - classify / count / swap, the node's last workgroup through the swap doing what follows
// the pass (kd_huge_*) - ~70 launches and two read-backs per level, bound by the HOST's launch rate: 0.6-0.75 ms per level, 2.7 ms
// for levels 0-3; levels 4-7 (1024 threads per node) 1.2 ms, levels 8-9 (256 per node) 0.2 ms, levels 10-16 (a wave per node, the
// node staged in LDS) 0.7 ms - of which 0.4 are still the ONE queue counter 16384 nodes of the last-but-one level add to (the
// children's ids come without a counter, both children are pushed with one addition: 5.9 -> 5.3 ms).
// Next: the launches of a level as one hipGraph (cached per shape); per-wave aggregation of the queue pushes.
#include <algorithm>
#include <cmath>
#include <chrono>
#include <cstdint>
#include <cstdio>
#include <vector>

#include "fdx_env.h"
#include "fdx_internal.h"
#include "kdtree_dev.h"

namespace fdx {
namespace {

struct KdWork { int node, start, end; };
// team sizes by node size: up to 1024 points one wave, up to 4096 a workgroup of 256, above one of 1024
constexpr int KD_SMALL = 1024, KD_MID = 4096, KD_HUGE = 100000;
// classes: 0 = a wave per node, 1 = 256 threads, 2 = 1024 threads, 3 = huge (launches of their own).  (A lane per node for the
// nodes of at most 64 points - the serial algorithm itself, half of a tree's nodes - measured no faster than a wave per node: 64
// lanes chasing 64 nodes through L2 are as latency-bound as 20 passes of ballots.)
__host__ __device__ inline int kd_size_class(long long size) { return size <= KD_SMALL ? 0 : size <= KD_MID ? 1 : size <= KD_HUGE ? 2 : 3; }
constexpr int KD_NCLS = 4;
struct KdQueues { KdWork* q[KD_NCLS]; };
// both children of a node into the next level's queues: one atomic when they are of one class (the counters of a tree's last levels
// are what tens of thousands of nodes add to at once)
__device__ __forceinline__ void kd_push_children(const KdQueues& next, int* __restrict__ n_next, int4* __restrict__ meta, int c0, const int (&cs)[2],
                                                 const int (&ce)[2], int leafsize) {
    const bool leaf0 = ce[0] - cs[0] <= leafsize, leaf1 = ce[1] - cs[1] <= leafsize;
    if (leaf0) meta[c0] = make_int4(-1, cs[0], ce[0], 0);
    if (leaf1) meta[c0 + 1] = make_int4(-1, cs[1], ce[1], 0);
    const int cls0 = leaf0 ? -1 : kd_size_class(ce[0] - cs[0]), cls1 = leaf1 ? -2 : kd_size_class(ce[1] - cs[1]);
    if (cls0 == cls1) {
        const int slot = atomicAdd(n_next + cls0, 2);
        next.q[cls0][slot] = KdWork{c0, cs[0], ce[0]};
        next.q[cls0][slot + 1] = KdWork{c0 + 1, cs[1], ce[1]};
        return;
    }
    if (!leaf0) next.q[cls0][atomicAdd(n_next + cls0, 1)] = KdWork{c0, cs[0], ce[0]};
    if (!leaf1) next.q[cls1][atomicAdd(n_next + cls1, 1)] = KdWork{c0 + 1, cs[1], ce[1]};
}

// The ids of a node's two children without a shared counter (32768 nodes of a tree's last level adding to ONE address cost more than
// their selections): entry e of class c at a level owns the slots base + 2 (entries of the classes below c + e), base = 1 + twice
// the entries of all earlier levels (kd_level_base_kernel).  A node that turns out a leaf leaves its two slots unused.
__device__ __forceinline__ int kd_child_ids(const int* __restrict__ level_counts, int cls, int entry, const int* __restrict__ base) {
    int off = 0;
    for (int c = 0; c < cls; ++c) off += level_counts[c];
    return *base + 2 * (off + entry);
}
// Above KD_HUGE points a node is not one workgroup's work (a million points through one compute unit: 2.5 ms for the root, 4.5 ms
// for its children): every pass of such a node is cut into KD_CH chunks, a wave each, over 64 workgroups, and the phases of a pass
// that need everybody's results of the one before are separate launches - classify / count / swap / advance (kd_huge_*).
constexpr int KD_CH = 256;
struct KdHuge {
    int node, start, end;
    int first, last, nth, depth, d;
    int phase;                 // 1 selection, 2 the "< split | >= split" pass, 3 that pass with the split just above the minimum, 4 finished
    int lo, hi;                // the range of the pass under way
    int K, nL, nR, cutL, cutR;
    int pad;
    double pv;                 // the pivot of a selection pass, the split of a split pass
};

struct KdBuildState {
    int n_nodes;        // nodes allocated so far
    int overflow;       // a selection ran out of its depth budget (libstdc++ would switch to heap select): build on the host
    int pad0, pad1;
    double mins[3], maxes[3];   // of the whole set (the root's bounds)
};

template <int M>
__device__ __forceinline__ double kd_key(const double* __restrict__ coords, int point, int d) {
    return coords[(size_t)point * M + d];
}

// libstdc++'s way out of introselect when the depth budget (2 log2 n partitions) is spent - a few nodes in every ten thousand of a
// random cloud: std::__heap_select(first, nth + 1, last) and std::iter_swap(first, nth), transcribed (make_heap by __adjust_heap
// from the last parent down, then every later element smaller than the heap's top replaces it: __pop_heap).  One thread: the
// ranges that get here are what a selection has left over.
template <int M>
__device__ void kd_adjust_heap(const double* __restrict__ coords, int* __restrict__ a, int hole, int len, int value, int d) {
    const double kv = kd_key<M>(coords, value, d);
    const int top = hole;
    int child = hole;
    while (child < (len - 1) / 2) {
        child = 2 * (child + 1);
        if (kd_key<M>(coords, a[child], d) < kd_key<M>(coords, a[child - 1], d)) --child;
        a[hole] = a[child];
        hole = child;
    }
    if ((len & 1) == 0 && child == (len - 2) / 2) {
        child = 2 * (child + 1);
        a[hole] = a[child - 1];
        hole = child - 1;
    }
    int parent = (hole - 1) / 2;                                          // __push_heap
    while (hole > top && kd_key<M>(coords, a[parent], d) < kv) {
        a[hole] = a[parent];
        hole = parent;
        parent = (hole - 1) / 2;
    }
    a[hole] = value;
}

template <int M>
__device__ void kd_heap_select_and_swap(const double* __restrict__ coords, int* __restrict__ idx, int first, int nth, int last, int d) {
    int* a = idx + first;
    const int len = nth + 1 - first;                                      // the heap: [first, nth + 1)
    if (len >= 2)
        for (int parent = (len - 2) / 2;; --parent) {
            kd_adjust_heap<M>(coords, a, parent, len, a[parent], d);
            if (parent == 0) break;
        }
    for (int i = nth + 1; i < last; ++i)
        if (kd_key<M>(coords, idx[i], d) < kd_key<M>(coords, a[0], d)) {  // __pop_heap(first, middle, i)
            const int value = idx[i];
            idx[i] = a[0];
            kd_adjust_heap<M>(coords, a, 0, len, value, d);
        }
    const int t = idx[first];                                             // std::iter_swap(first, nth)
    idx[first] = idx[nth];
    idx[nth] = t;
}

// The list form of a two-pointer pass over positions [lo, hi) of idx: left stops = positions whose key satisfies SL, right stops =
// positions whose key satisfies SR; swaps the k-th left stop with the k-th right stop (from the right) while the former lies left
// of the latter.  The range is cut into one contiguous chunk per WAVE of the team; a wave walks its chunk 64 positions at a time
// (coalesced) and appends the stops to its lists in order (ballot + prefix count).  lp / rp: global scratch indexed by absolute
// position (a chunk's lists start at the chunk's first position).  Returns K (swaps), nL, nR and the positions L(K) (hi when there
// is none) and R(K - 1) (hi when there is none).
struct KdPassOut { int K, nL, nR, LK, RK1; };

template <int M, int T, class SL, class SR>
__device__ KdPassOut kd_pass(const double* __restrict__ coords, int* __restrict__ idx, int lo, int hi, int d, const SL& stop_left,
                             const SR& stop_right, int* __restrict__ lp, int* __restrict__ rp, int* s_nl, int* s_nr, int* s_cnt,
                             int* s_out, int tid) {
    constexpr int NW = T / 64;
    const int wave = tid >> 6, lane = tid & 63;
    const long long len = (long long)hi - lo;
    const int cb = lo + (int)(len * wave / NW), ce = lo + (int)(len * (wave + 1) / NW);
    const unsigned long long lt = lane == 0 ? 0ULL : (~0ULL >> (64 - lane));
    int nl = 0, nr = 0;
    for (int base = cb; base < ce; base += 256) {                          // four 64-position blocks at a time: their loads in flight together
        int pt[4];
        double v[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) { const int i = base + 64 * u + lane; pt[u] = i < ce ? idx[i] : -1; }
#pragma unroll
        for (int u = 0; u < 4; ++u) v[u] = pt[u] >= 0 ? kd_key<M>(coords, pt[u], d) : 0.0;
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int i = base + 64 * u + lane;
            const bool valid = pt[u] >= 0;
            const bool sl = valid && stop_left(v[u]), sr_ = valid && stop_right(v[u]);
            const unsigned long long ml = __ballot(sl), mr = __ballot(sr_);
            if (sl) lp[cb + nl + __popcll(ml & lt)] = i;
            if (sr_) rp[cb + nr + __popcll(mr & lt)] = i;
            nl += __popcll(ml);
            nr += __popcll(mr);
        }
    }
    if (lane == 0) { s_nl[wave] = nl; s_nr[wave] = nr; }
    __syncthreads();                                                      // (also: the lists are written)
    int pl = 0, nL = 0, sr = 0, nR = 0;
#pragma unroll
    for (int u = 0; u < NW; ++u) {
        const int a = s_nl[u], c = s_nr[u];
        pl += u < wave ? a : 0;
        nL += a;
        sr += u > wave ? c : 0;
        nR += c;
    }
    // this wave's left stops that are swapped: stop j (rank k = pl + j, position x) iff more than k right stops lie beyond x
    int cnt = 0;
    for (int j0 = 0; j0 < nl; j0 += 64) {
        const int j = j0 + lane;
        bool ok = false;
        if (j < nl) {
            const int x = lp[cb + j];
            int a = 0, c = nr;                                            // first local right stop with position > x
            while (a < c) {
                const int mid = (a + c) >> 1;
                if (rp[cb + mid] <= x) a = mid + 1; else c = mid;
            }
            ok = sr + (nr - a) > pl + j;
        }
        const int got = __popcll(__ballot(ok));
        cnt += got;
        if (got < 64) break;                                              // (a prefix of all left stops: nothing after the first failure)
    }
    if (lane == 0) s_cnt[wave] = cnt;
    __syncthreads();
    int K = 0;
#pragma unroll
    for (int u = 0; u < NW; ++u) K += s_cnt[u];
    // rank -> position of the k-th right stop from the right
    auto right_by_rank = [&](int k) -> int {
        int beyond = 0;
        for (int u = NW - 1; u >= 0; --u) {
            const int c = s_nr[u];
            if (k < beyond + c) {
                const int ub = lo + (int)(len * u / NW);
                return rp[ub + (c - 1 - (k - beyond))];
            }
            beyond += c;
        }
        return hi;
    };
    const int mine = min(nl, max(0, K - pl));
    for (int j = lane; j < mine; j += 64) {
        const int x = lp[cb + j];
        const int y = right_by_rank(pl + j);
        const int t = idx[x];
        idx[x] = idx[y];
        idx[y] = t;
    }
    if (tid == 0) { s_out[0] = hi; s_out[1] = K > 0 ? right_by_rank(K - 1) : hi; }
    __syncthreads();
    if (lane == 0 && K < nL && pl <= K && K < pl + nl) s_out[0] = lp[cb + (K - pl)];   // L(K): the next stop of the left pointer
    __syncthreads();
    KdPassOut o;
    o.K = K; o.nL = nL; o.nR = nR; o.LK = s_out[0]; o.RK1 = s_out[1];
    __syncthreads();                                                      // (the swaps are done, s_* may be written again)
    return o;
}

template <int M, int T>
__global__ __launch_bounds__(T) void kd_level_kernel(const double* __restrict__ coords, int* __restrict__ idx, int4* __restrict__ meta,
                                                     double* __restrict__ split_out, const KdWork* __restrict__ cur,
                                                     const int* __restrict__ n_cur, const KdQueues next, int* __restrict__ n_next,
                                                     KdBuildState* __restrict__ st, int* __restrict__ lp, int* __restrict__ rp, int leafsize,
                                                     int size_above, const int* __restrict__ level_counts, int cls, const int* __restrict__ base) {
#pragma clang fp contract(off)
    __shared__ int s_nl[T / 64], s_nr[T / 64], s_cnt[T / 64], s_out[8];
    __shared__ double s_red[2 * 3 * (T / 64)];
    __shared__ double s_bounds[6];
    __shared__ int s_ctl[4];
    const int tid = threadIdx.x;
    if ((int)blockIdx.x >= *n_cur) return;
    const KdWork w = cur[blockIdx.x];
    const int size = w.end - w.start;
    if (size <= size_above) return;                                     // (the wave class: such a node is kd_small_kernel's, out of LDS)
    // ---- bounds of the node's own points
    {
        double mx[M], mn[M];
#pragma unroll
        for (int a = 0; a < M; ++a) { mx[a] = -HUGE_VAL; mn[a] = HUGE_VAL; }
        for (int i = w.start + tid; i < w.end; i += T) {
            const int pt = idx[i];
#pragma unroll
            for (int a = 0; a < M; ++a) {
                const double v = coords[(size_t)pt * M + a];
                mx[a] = mx[a] > v ? mx[a] : v;
                mn[a] = mn[a] < v ? mn[a] : v;
            }
        }
#pragma unroll
        for (int a = 0; a < M; ++a) {
            for (int off = 32; off > 0; off >>= 1) {
                const double ox = __shfl_xor(mx[a], off), on = __shfl_xor(mn[a], off);
                mx[a] = mx[a] > ox ? mx[a] : ox;
                mn[a] = mn[a] < on ? mn[a] : on;
            }
            if ((tid & 63) == 0) { s_red[((tid >> 6) * 3 + a) * 2] = mx[a]; s_red[((tid >> 6) * 3 + a) * 2 + 1] = mn[a]; }
        }
        __syncthreads();
        if (tid == 0) {
#pragma unroll
            for (int a = 0; a < M; ++a) {
                double x = s_red[a * 2], m2 = s_red[a * 2 + 1];
                for (int wv = 1; wv < T / 64; ++wv) {
                    const double ox = s_red[(wv * 3 + a) * 2], on = s_red[(wv * 3 + a) * 2 + 1];
                    x = x > ox ? x : ox;
                    m2 = m2 < on ? m2 : on;
                }
                s_bounds[a] = x;
                s_bounds[3 + a] = m2;
            }
            int d = 0;
            double sz = 0.0;
#pragma unroll
            for (int a = 0; a < M; ++a)
                if (s_bounds[a] - s_bounds[3 + a] > sz) { d = a; sz = s_bounds[a] - s_bounds[3 + a]; }
            s_ctl[0] = d;
            s_ctl[1] = s_bounds[d] == s_bounds[3 + d] ? 1 : 0;          // all points identical: a leaf
            if (w.node == 0) {
#pragma unroll
                for (int a = 0; a < M; ++a) { st->maxes[a] = s_bounds[a]; st->mins[a] = s_bounds[3 + a]; }
            }
        }
        __syncthreads();
    }
    const int d = s_ctl[0];
    if (s_ctl[1]) {
        if (tid == 0) meta[w.node] = make_int4(-1, w.start, w.end, 0);
        return;
    }
    // ---- std::nth_element(idx + start, idx + start + half, idx + end) by the coordinate d
    const int half = size / 2;
    const int nth = w.start + half;
    {
        int first = w.start, last = w.end;
        int depth = 2 * (31 - __clz(size));
        while (last - first > 3) {
            if (depth == 0) {                                             // (uniform: every thread sees the same depth)
                if (tid == 0) kd_heap_select_and_swap<M>(coords, idx, first, nth, last, d);
                first = last;                                             // introselect returns here: no insertion sort
                break;
            }
            --depth;
            if (tid == 0) {                                               // __move_median_to_first(first, first + 1, mid, last - 1)
                const int ia = first + 1, ib = first + (last - first) / 2, ic = last - 1;
                const double ka = kd_key<M>(coords, idx[ia], d), kb = kd_key<M>(coords, idx[ib], d), kc = kd_key<M>(coords, idx[ic], d);
                int pick;
                if (ka < kb) pick = kb < kc ? ib : (ka < kc ? ic : ia);
                else pick = ka < kc ? ia : (kb < kc ? ic : ib);
                const int t0 = idx[first];
                idx[first] = idx[pick];
                idx[pick] = t0;
            }
            __syncthreads();
            const double pv = kd_key<M>(coords, idx[first], d);
            const KdPassOut o = kd_pass<M, T>(coords, idx, first + 1, last, d, [pv](double v) { return !(v < pv); },
                                              [pv](double v) { return !(pv < v); }, lp, rp, s_nl, s_nr, s_cnt, s_out, tid);
            const int cut = o.LK < o.RK1 ? o.LK : o.RK1;                  // (hi = last stands for "none")
            if (cut <= nth) first = cut; else last = cut;
        }
        if (tid == 0) {                                                   // __insertion_sort of the last (at most three) elements
            for (int i = first + 1; i < last; ++i) {
                const int val = idx[i];
                const double kv = kd_key<M>(coords, val, d);
                int j = i;
                while (j > first && kv < kd_key<M>(coords, idx[j - 1], d)) { idx[j] = idx[j - 1]; --j; }
                idx[j] = val;
            }
        }
        __syncthreads();
    }
    // ---- scipy's "< split | >= split" pass (everything from `half` on is >= split after the selection)
    double split = kd_key<M>(coords, idx[nth], d);
    int p;
    {
        const KdPassOut o = kd_pass<M, T>(coords, idx, w.start, nth, d, [split](double v) { return !(v < split); },
                                          [split](double v) { return v < split; }, lp, rp, s_nl, s_nr, s_cnt, s_out, tid);
        p = w.start + o.nR;                                               // the keys below the split
    }
    if (p == w.start) {                                                   // the median is the minimum: split just above it
        split = nextafter(split, HUGE_VAL);
        const KdPassOut o = kd_pass<M, T>(coords, idx, w.start, w.end, d, [split](double v) { return !(v < split); },
                                          [split](double v) { return v < split; }, lp, rp, s_nl, s_nr, s_cnt, s_out, tid);
        p = w.start + o.nR;
    }
    // ---- the children
    if (tid == 0) {
        const int c0 = kd_child_ids(level_counts, cls, (int)blockIdx.x, base);
        meta[w.node] = make_int4(d, c0, c0 + 1, 0);
        split_out[w.node] = split;
        const int cs[2] = {w.start, p}, ce[2] = {p, w.end};
        kd_push_children(next, n_next, meta, c0, cs, ce, leafsize);
    }
}

// ---- the wave class out of LDS: a node of at most `cap` points (cap x 16 bytes of LDS: the keys along the split dimension, the
// indices, the two position lists) is staged once, every pass of its selection and its split runs on the staged copy - the same
// passes as kd_pass, one chunk -, and the indices are written back.  The last six levels of a tree: 1.4 ms per million points
// through L2, where every one of a node's ~20 passes is a chain of global round trips.
template <int M>
__global__ __launch_bounds__(64) void kd_small_kernel(const double* __restrict__ coords, int* __restrict__ idx, int4* __restrict__ meta,
                                                      double* __restrict__ split_out, const KdWork* __restrict__ cur, const int* __restrict__ n_cur,
                                                      const KdQueues next, int* __restrict__ n_next, KdBuildState* __restrict__ st, int leafsize,
                                                      int cap, const int* __restrict__ level_counts, const int* __restrict__ base) {
#pragma clang fp contract(off)
    extern __shared__ __attribute__((aligned(16))) unsigned char kd_lds[];
    double* s_key = reinterpret_cast<double*>(kd_lds);
    int* s_idx = reinterpret_cast<int*>(s_key + cap);
    unsigned short* s_lp = reinterpret_cast<unsigned short*>(s_idx + cap);
    unsigned short* s_rp = s_lp + cap;
    if ((int)blockIdx.x >= *n_cur) return;
    const KdWork w = cur[blockIdx.x];
    const int n = w.end - w.start;
    if (n > cap) return;                                                  // (kd_level_kernel<M, 64>'s, through global memory)
    const int lane = threadIdx.x;
    const unsigned long long lt = lane == 0 ? 0ULL : (~0ULL >> (64 - lane));
    double mx[M], mn[M];
#pragma unroll
    for (int a = 0; a < M; ++a) { mx[a] = -HUGE_VAL; mn[a] = HUGE_VAL; }
    for (int i = lane; i < n; i += 64) {
        const int pt = idx[w.start + i];
        s_idx[i] = pt;
#pragma unroll
        for (int a = 0; a < M; ++a) {
            const double v = coords[(size_t)pt * M + a];
            mx[a] = mx[a] > v ? mx[a] : v;
            mn[a] = mn[a] < v ? mn[a] : v;
        }
    }
#pragma unroll
    for (int a = 0; a < M; ++a)
        for (int off = 32; off > 0; off >>= 1) {
            const double ox = __shfl_xor(mx[a], off), on = __shfl_xor(mn[a], off);
            mx[a] = mx[a] > ox ? mx[a] : ox;
            mn[a] = mn[a] < on ? mn[a] : on;
        }
    int d = 0;
    double sz = 0.0;
#pragma unroll
    for (int a = 0; a < M; ++a)
        if (mx[a] - mn[a] > sz) { d = a; sz = mx[a] - mn[a]; }
    if (w.node == 0 && lane == 0) {
#pragma unroll
        for (int a = 0; a < M; ++a) { st->maxes[a] = mx[a]; st->mins[a] = mn[a]; }
    }
    if (!(sz > 0.0)) {                                                    // maxes[d] == mins[d]: all points identical, a leaf
        if (lane == 0) meta[w.node] = make_int4(-1, w.start, w.end, 0);
        return;
    }
    __syncthreads();
    for (int i = lane; i < n; i += 64) s_key[i] = coords[(size_t)s_idx[i] * M + d];
    __syncthreads();
    auto swap_at = [&](int a, int b) {
        const int ti = s_idx[a]; s_idx[a] = s_idx[b]; s_idx[b] = ti;
        const double tk = s_key[a]; s_key[a] = s_key[b]; s_key[b] = tk;
    };
    // one pass in list form over positions [lo, hi): K swaps; nR; L(K) and R(K - 1) (hi when there is none)
    auto pass = [&](int lo, int hi, auto stop_left, auto stop_right, int& nR_out, int& LK, int& RK1) {
        int nl = 0, nr = 0;
        for (int base = lo; base < hi; base += 64) {
            const int i = base + lane;
            const bool valid = i < hi;
            const double v = valid ? s_key[i] : 0.0;
            const bool sl = valid && stop_left(v), sr_ = valid && stop_right(v);
            const unsigned long long ml = __ballot(sl), mr = __ballot(sr_);
            if (sl) s_lp[nl + __popcll(ml & lt)] = (unsigned short)i;
            if (sr_) s_rp[nr + __popcll(mr & lt)] = (unsigned short)i;
            nl += __popcll(ml);
            nr += __popcll(mr);
        }
        __syncthreads();
        int K = 0;
        for (int j0 = 0; j0 < nl; j0 += 64) {
            const int j = j0 + lane;
            bool ok = false;
            if (j < nl) {
                const int x = s_lp[j];
                int a = 0, e = nr;                                        // first right stop with position > x
                while (a < e) {
                    const int mid = (a + e) >> 1;
                    if ((int)s_rp[mid] <= x) a = mid + 1; else e = mid;
                }
                ok = nr - a > j;
            }
            const int got = __popcll(__ballot(ok));
            K += got;
            if (got < 64) break;
        }
        for (int j = lane; j < K; j += 64) swap_at(s_lp[j], s_rp[nr - 1 - j]);
        LK = K < nl ? (int)s_lp[K] : hi;
        RK1 = K > 0 ? (int)s_rp[nr - K] : hi;
        nR_out = nr;
        __syncthreads();
    };
    // ---- std::nth_element(0, n / 2, n) by the staged keys
    const int nth = n / 2;
    {
        int first = 0, last = n;
        int depth = 2 * (31 - __clz(n));
        while (last - first > 3) {
            if (depth == 0) {                                             // heap select: rare - through the global copy
                __syncthreads();
                for (int i = lane; i < n; i += 64) idx[w.start + i] = s_idx[i];
                __syncthreads();
                if (lane == 0) {
                    __threadfence_block();
                    kd_heap_select_and_swap<M>(coords, idx, w.start + first, w.start + nth, w.start + last, d);
                    __threadfence_block();
                }
                __syncthreads();
                for (int i = lane; i < n; i += 64) { const int pt = idx[w.start + i]; s_idx[i] = pt; s_key[i] = coords[(size_t)pt * M + d]; }
                __syncthreads();
                first = last;
                break;
            }
            --depth;
            if (lane == 0) {                                              // __move_median_to_first(first, first + 1, mid, last - 1)
                const int ia = first + 1, ib = first + (last - first) / 2, ic = last - 1;
                const double ka = s_key[ia], kb = s_key[ib], kc = s_key[ic];
                int pick;
                if (ka < kb) pick = kb < kc ? ib : (ka < kc ? ic : ia);
                else pick = ka < kc ? ia : (kb < kc ? ic : ib);
                swap_at(first, pick);
            }
            __syncthreads();
            const double pv = s_key[first];
            int nR, LK, RK1;
            pass(first + 1, last, [pv](double v) { return !(v < pv); }, [pv](double v) { return !(pv < v); }, nR, LK, RK1);
            const int cut = LK < RK1 ? LK : RK1;
            if (cut <= nth) first = cut; else last = cut;
        }
        if (lane == 0)
            for (int i = first + 1; i < last; ++i) {                      // __insertion_sort of the last (at most three) elements
                const int vi = s_idx[i];
                const double kv = s_key[i];
                int j = i;
                while (j > first && kv < s_key[j - 1]) { s_idx[j] = s_idx[j - 1]; s_key[j] = s_key[j - 1]; --j; }
                s_idx[j] = vi;
                s_key[j] = kv;
            }
        __syncthreads();
    }
    // ---- scipy's "< split | >= split" pass; the split just above the minimum when nothing is below it
    double split = s_key[nth];
    int p;
    {
        int nR, LK, RK1;
        pass(0, nth, [split](double v) { return !(v < split); }, [split](double v) { return v < split; }, nR, LK, RK1);
        p = nR;
        if (p == 0) {
            split = nextafter(split, HUGE_VAL);
            pass(0, n, [split](double v) { return !(v < split); }, [split](double v) { return v < split; }, nR, LK, RK1);
            p = nR;
        }
    }
    for (int i = lane; i < n; i += 64) idx[w.start + i] = s_idx[i];
    if (lane == 0) {
        const int c0 = kd_child_ids(level_counts, 0, (int)blockIdx.x, base);
        meta[w.node] = make_int4(d, c0, c0 + 1, 0);
        split_out[w.node] = split;
        const int cs[2] = {w.start, w.start + p}, ce[2] = {w.start + p, w.end};
        kd_push_children(next, n_next, meta, c0, cs, ce, leafsize);
    }
}

// ---- nodes above KD_HUGE points: a pass in four launches --------------------------------------------------------------------
template <int M>
__device__ void kd_huge_begin_select(const double* __restrict__ coords, int* __restrict__ idx, KdHuge& h, KdBuildState* st) {
    if (h.depth == 0) {                                                   // the budget is spent: heap select, and the selection is over
        kd_heap_select_and_swap<M>(coords, idx, h.first, h.nth, h.last, h.d);
        h.pv = kd_key<M>(coords, idx[h.nth], h.d);                        // the split
        h.lo = h.start;
        h.hi = h.nth;
        h.phase = 2;
        return;
    }
    --h.depth;
    const int ia = h.first + 1, ib = h.first + (h.last - h.first) / 2, ic = h.last - 1;   // __move_median_to_first(first, first + 1, mid, last - 1)
    const double ka = kd_key<M>(coords, idx[ia], h.d), kb = kd_key<M>(coords, idx[ib], h.d), kc = kd_key<M>(coords, idx[ic], h.d);
    int pick;
    if (ka < kb) pick = kb < kc ? ib : (ka < kc ? ic : ia);
    else pick = ka < kc ? ia : (kb < kc ? ic : ib);
    const int t0 = idx[h.first];
    idx[h.first] = idx[pick];
    idx[pick] = t0;
    h.pv = kd_key<M>(coords, idx[h.first], h.d);
    h.lo = h.first + 1;
    h.hi = h.last;
    h.phase = 1;
}

template <int M>
__device__ void kd_huge_bounds(const double* __restrict__ coords, const int* idx, const KdWork* __restrict__ cur, double* hb, int slot, int wg) {
    const KdWork w = cur[slot];
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int c = wg * 4 + wave;
    const long long len = (long long)w.end - w.start;
    const int cb = w.start + (int)(len * c / KD_CH), ce = w.start + (int)(len * (c + 1) / KD_CH);
    double mx[M], mn[M];
#pragma unroll
    for (int a = 0; a < M; ++a) { mx[a] = -HUGE_VAL; mn[a] = HUGE_VAL; }
    for (int i = cb + lane; i < ce; i += 64) {
        const int pt = idx[i];
#pragma unroll
        for (int a = 0; a < M; ++a) {
            const double v = coords[(size_t)pt * M + a];
            mx[a] = mx[a] > v ? mx[a] : v;
            mn[a] = mn[a] < v ? mn[a] : v;
        }
    }
#pragma unroll
    for (int a = 0; a < M; ++a) {
        for (int off = 32; off > 0; off >>= 1) {
            const double ox = __shfl_xor(mx[a], off), on = __shfl_xor(mn[a], off);
            mx[a] = mx[a] > ox ? mx[a] : ox;
            mn[a] = mn[a] < on ? mn[a] : on;
        }
        if (lane == 0) { hb[((size_t)slot * KD_CH + c) * 6 + a] = mx[a]; hb[((size_t)slot * KD_CH + c) * 6 + 3 + a] = mn[a]; }
    }
}

template <int M>
__device__ void kd_huge_setup(const double* __restrict__ coords, int* idx, const KdWork* __restrict__ cur, const double* hb, KdHuge* hs,
                              int4* __restrict__ meta, KdBuildState* st, int slot) {
#pragma clang fp contract(off)
    __shared__ double s_red[4 * 6];
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    double mx[M], mn[M];
#pragma unroll
    for (int a = 0; a < M; ++a) {
        mx[a] = hb[((size_t)slot * KD_CH + tid) * 6 + a];                 // (KD_CH == 256 == the workgroup: a chunk per thread)
        mn[a] = hb[((size_t)slot * KD_CH + tid) * 6 + 3 + a];
        for (int off = 32; off > 0; off >>= 1) {
            const double ox = __shfl_xor(mx[a], off), on = __shfl_xor(mn[a], off);
            mx[a] = mx[a] > ox ? mx[a] : ox;
            mn[a] = mn[a] < on ? mn[a] : on;
        }
        if (lane == 0) { s_red[wave * 6 + a] = mx[a]; s_red[wave * 6 + 3 + a] = mn[a]; }
    }
    __syncthreads();
    if (tid != 0) return;
    const KdWork w = cur[slot];
    double bx[M], bn[M];
#pragma unroll
    for (int a = 0; a < M; ++a) {
        bx[a] = s_red[a];
        bn[a] = s_red[3 + a];
        for (int wv = 1; wv < 4; ++wv) {
            bx[a] = bx[a] > s_red[wv * 6 + a] ? bx[a] : s_red[wv * 6 + a];
            bn[a] = bn[a] < s_red[wv * 6 + 3 + a] ? bn[a] : s_red[wv * 6 + 3 + a];
        }
    }
    if (w.node == 0) {
#pragma unroll
        for (int a = 0; a < M; ++a) { st->maxes[a] = bx[a]; st->mins[a] = bn[a]; }
    }
    int d = 0;
    double sz = 0.0;
#pragma unroll
    for (int a = 0; a < M; ++a)
        if (bx[a] - bn[a] > sz) { d = a; sz = bx[a] - bn[a]; }
    KdHuge h{};
    h.node = w.node; h.start = w.start; h.end = w.end;
    h.d = d;
    if (bx[d] == bn[d]) {                                                 // all points identical: a leaf
        meta[w.node] = make_int4(-1, w.start, w.end, 0);
        h.phase = 4;
        hs[slot] = h;
        return;
    }
    const int size = w.end - w.start;
    h.first = w.start; h.last = w.end; h.nth = w.start + size / 2;
    h.depth = 2 * (31 - __clz(size));
    kd_huge_begin_select<M>(coords, idx, h, st);                          // (size > 3)
    hs[slot] = h;
}

template <int M>
__device__ void kd_huge_classify(const double* __restrict__ coords, const int* idx, const KdHuge* hs, int* lp, int* rp, int* hc, int slot, int wg) {
    const int phase = hs[slot].phase;
    if (phase == 4) return;
    const int lo = hs[slot].lo, hi = hs[slot].hi, d = hs[slot].d;
    const double pv = hs[slot].pv;
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int c = wg * 4 + wave;
    const long long len = (long long)hi - lo;
    const int cb = lo + (int)(len * c / KD_CH), ce = lo + (int)(len * (c + 1) / KD_CH);
    const unsigned long long lt = lane == 0 ? 0ULL : (~0ULL >> (64 - lane));
    int nl = 0, nr = 0;
    for (int base = cb; base < ce; base += 256) {                          // four 64-position blocks at a time: their loads in flight together
        int pt[4];
        double v[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) { const int i = base + 64 * u + lane; pt[u] = i < ce ? idx[i] : -1; }
#pragma unroll
        for (int u = 0; u < 4; ++u) v[u] = pt[u] >= 0 ? kd_key<M>(coords, pt[u], d) : 0.0;
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int i = base + 64 * u + lane;
            const bool valid = pt[u] >= 0;
            const bool sl = valid && !(v[u] < pv);
            const bool sr_ = valid && (phase == 1 ? !(pv < v[u]) : v[u] < pv);
            const unsigned long long ml = __ballot(sl), mr = __ballot(sr_);
            if (sl) lp[cb + nl + __popcll(ml & lt)] = i;
            if (sr_) rp[cb + nr + __popcll(mr & lt)] = i;
            nl += __popcll(ml);
            nr += __popcll(mr);
        }
    }
    if (lane == 0) { hc[(slot * 5 + 0) * KD_CH + c] = nl; hc[(slot * 5 + 1) * KD_CH + c] = nr; }
}

__device__ void kd_huge_count(KdHuge* hs, const int* lp, const int* rp, int* hc, int slot, int wg) {
    if (hs[slot].phase == 4) return;
    const int lo = hs[slot].lo, hi = hs[slot].hi;
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int c = wg * 4 + wave;
    const long long len = (long long)hi - lo;
    const int cb = lo + (int)(len * c / KD_CH);
    const int* cl = hc + (slot * 5 + 0) * KD_CH;
    const int* cr = hc + (slot * 5 + 1) * KD_CH;
    int pl = 0, sr = 0, nL = 0, nR = 0;
    for (int u = lane; u < KD_CH; u += 64) {
        const int a = cl[u], b = cr[u];
        pl += u < c ? a : 0;
        sr += u > c ? b : 0;
        nL += a;
        nR += b;
    }
    for (int off = 32; off > 0; off >>= 1) {
        pl += __shfl_xor(pl, off);
        sr += __shfl_xor(sr, off);
        nL += __shfl_xor(nL, off);
        nR += __shfl_xor(nR, off);
    }
    const int nl = cl[c], nr = cr[c];
    int cnt = 0;
    for (int j0 = 0; j0 < nl; j0 += 64) {
        const int j = j0 + lane;
        bool ok = false;
        if (j < nl) {
            const int x = lp[cb + j];
            int a = 0, e = nr;                                            // first local right stop with position > x
            while (a < e) {
                const int mid = (a + e) >> 1;
                if (rp[cb + mid] <= x) a = mid + 1; else e = mid;
            }
            ok = sr + (nr - a) > pl + j;
        }
        const int got = __popcll(__ballot(ok));
        cnt += got;
        if (got < 64) break;
    }
    if (lane == 0) {
        hc[(slot * 5 + 2) * KD_CH + c] = pl;
        hc[(slot * 5 + 3) * KD_CH + c] = sr;
        hc[(slot * 5 + 4) * KD_CH + c] = cnt;
        if (c == 0) { hs[slot].nL = nL; hs[slot].nR = nR; hs[slot].cutL = hi; hs[slot].cutR = hi; }
    }
}

// What follows a pass, by ONE thread, once every swap of the pass is done and visible: narrow the selection and set up its next pass
// (median of three, pivot), or end it (insertion sort, the split pass), or make the node's children.
template <int M>
__device__ void kd_huge_advance(const double* __restrict__ coords, int* idx, KdHuge* hs, int slot, int4* __restrict__ meta,
                                double* __restrict__ split_out, const KdQueues& next, int* __restrict__ n_next, KdBuildState* st,
                                int leafsize, const int* __restrict__ level_counts, const int* __restrict__ base) {
#pragma clang fp contract(off)
    KdHuge h = hs[slot];
    // (written by other workgroups of the launch that calls this: real loads, not what this thread may hold in registers)
    h.K = __atomic_load_n(&hs[slot].K, __ATOMIC_RELAXED);
    h.nL = __atomic_load_n(&hs[slot].nL, __ATOMIC_RELAXED);
    h.nR = __atomic_load_n(&hs[slot].nR, __ATOMIC_RELAXED);
    h.cutL = __atomic_load_n(&hs[slot].cutL, __ATOMIC_RELAXED);
    h.cutR = __atomic_load_n(&hs[slot].cutR, __ATOMIC_RELAXED);
    if (h.phase == 4) return;
    int p = -1;
    if (h.phase == 1) {
        const int cut = h.cutL < h.cutR ? h.cutL : h.cutR;                // (hi stands for "none")
        if (cut <= h.nth) h.first = cut; else h.last = cut;
        if (h.last - h.first > 3) {
            kd_huge_begin_select<M>(coords, idx, h, st);
        } else {
            for (int i = h.first + 1; i < h.last; ++i) {                  // __insertion_sort of the last (at most three) elements
                const int val = idx[i];
                const double kv = kd_key<M>(coords, val, h.d);
                int j = i;
                while (j > h.first && kv < kd_key<M>(coords, idx[j - 1], h.d)) { idx[j] = idx[j - 1]; --j; }
                idx[j] = val;
            }
            h.pv = kd_key<M>(coords, idx[h.nth], h.d);                    // the split; everything from nth on is >= it
            h.lo = h.start;
            h.hi = h.nth;
            h.phase = 2;
        }
    } else if (h.phase == 2) {
        p = h.start + h.nR;                                               // the keys below the split
        if (p == h.start) {                                               // the median is the minimum: split just above it
            h.pv = nextafter(h.pv, HUGE_VAL);
            h.lo = h.start;
            h.hi = h.end;
            h.phase = 3;
            p = -1;
        }
    } else {
        p = h.start + h.nR;
    }
    if (p >= 0) {
        const int c0 = kd_child_ids(level_counts, 3, slot, base);
        meta[h.node] = make_int4(h.d, c0, c0 + 1, 0);
        split_out[h.node] = h.pv;
        const int cs[2] = {h.start, p}, ce[2] = {p, h.end};
        kd_push_children(next, n_next, meta, c0, cs, ce, leafsize);
        h.phase = 4;
    }
    hs[slot] = h;
}

// (returns false when the node is finished: nothing was done)
__device__ bool kd_huge_swap(int* idx, KdHuge* hs, const int* lp, const int* rp, const int* hc, int slot, int wg) {
    __shared__ int s_sr[KD_CH], s_nr[KD_CH], s_red[4];
    if (hs[slot].phase == 4) return false;
    const int lo = hs[slot].lo, hi = hs[slot].hi, nL = hs[slot].nL;
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    s_sr[tid] = hc[(slot * 5 + 3) * KD_CH + tid];                         // (KD_CH == 256 == the workgroup)
    s_nr[tid] = hc[(slot * 5 + 1) * KD_CH + tid];
    int k_part = hc[(slot * 5 + 4) * KD_CH + tid];
    for (int off = 32; off > 0; off >>= 1) k_part += __shfl_xor(k_part, off);
    if (lane == 0) s_red[wave] = k_part;
    __syncthreads();
    const int K = s_red[0] + s_red[1] + s_red[2] + s_red[3];
    const long long len = (long long)hi - lo;
    auto right_by_rank = [&](int k) -> int {                              // chunk u with s_sr[u] <= k < s_sr[u] + s_nr[u]: the smallest u with s_sr[u] <= k
        int a = 0, e = KD_CH - 1;
        while (a < e) {
            const int mid = (a + e) >> 1;
            if (s_sr[mid] <= k) e = mid; else a = mid + 1;
        }
        const int ub = lo + (int)(len * a / KD_CH);
        return rp[ub + (s_nr[a] - 1 - (k - s_sr[a]))];
    };
    const int c = wg * 4 + wave;
    const int cb = lo + (int)(len * c / KD_CH);
    const int pl = hc[(slot * 5 + 2) * KD_CH + c], nl = hc[(slot * 5 + 0) * KD_CH + c];
    const int mine = min(nl, max(0, K - pl));
    for (int j = lane; j < mine; j += 64) {
        const int x = lp[cb + j];
        const int y = right_by_rank(pl + j);
        const int t = idx[x];
        idx[x] = idx[y];
        idx[y] = t;
    }
    if (lane == 0) {
        if (K < nL && pl <= K && K < pl + nl) hs[slot].cutL = lp[cb + (K - pl)];   // L(K): the next stop of the left pointer
        if (c == 0) {
            hs[slot].K = K;
            if (K > 0) hs[slot].cutR = right_by_rank(K - 1);              // R(K - 1): where the last swap put a left-stopping key
        }
    }
    return true;
}

// (A one-launch form - all of a level's passes inside one launch, the 64 workgroups of a node meeting at a barrier of their own
// between the steps, every wait bounded - was built and measured: correct, and 12.1 ms per million points against 5.3.  The
// node's workgroups sit on all eight XCDs, whose L2s are not coherent with one another: every barrier is an agent-scope release
// and acquire, i.e. an L2 write-back and invalidate, ~30 us each - what a kernel boundary does once, for everybody.)
// ---- the general form: a launch per step
template <int M>
__global__ __launch_bounds__(256) void kd_huge_bounds_kernel(const double* __restrict__ coords, const int* __restrict__ idx,
                                                             const KdWork* __restrict__ cur, const int* __restrict__ n_cur,
                                                             double* __restrict__ hb) {
    if ((int)blockIdx.y >= *n_cur) return;
    kd_huge_bounds<M>(coords, idx, cur, hb, blockIdx.y, blockIdx.x);
}

template <int M>
__global__ __launch_bounds__(256) void kd_huge_setup_kernel(const double* __restrict__ coords, int* __restrict__ idx,
                                                            const KdWork* __restrict__ cur, const int* __restrict__ n_cur,
                                                            const double* __restrict__ hb, KdHuge* __restrict__ hs, int4* __restrict__ meta,
                                                            KdBuildState* __restrict__ st) {
    if ((int)blockIdx.x >= *n_cur) return;
    kd_huge_setup<M>(coords, idx, cur, hb, hs, meta, st, blockIdx.x);
}

template <int M>
__global__ __launch_bounds__(256) void kd_huge_classify_kernel(const double* __restrict__ coords, const int* __restrict__ idx,
                                                               const KdHuge* __restrict__ hs, const int* __restrict__ n_cur,
                                                               int* __restrict__ lp, int* __restrict__ rp, int* __restrict__ hc) {
    if ((int)blockIdx.y >= *n_cur) return;
    kd_huge_classify<M>(coords, idx, hs, lp, rp, hc, blockIdx.y, blockIdx.x);
}

__global__ __launch_bounds__(256) void kd_huge_count_kernel(KdHuge* __restrict__ hs, const int* __restrict__ n_cur, const int* __restrict__ lp,
                                                            const int* __restrict__ rp, int* __restrict__ hc) {
    if ((int)blockIdx.y >= *n_cur) return;
    kd_huge_count(hs, lp, rp, hc, blockIdx.y, blockIdx.x);
}

template <int M>
__global__ __launch_bounds__(256) void kd_huge_swap_kernel(const double* __restrict__ coords, int* idx, KdHuge* hs, const int* __restrict__ n_cur,
                                                           const int* __restrict__ lp, const int* __restrict__ rp, const int* __restrict__ hc,
                                                           int* __restrict__ tickets, int4* __restrict__ meta, double* __restrict__ split_out,
                                                           const KdQueues next, int* __restrict__ n_next, KdBuildState* st, int leafsize,
                                                           const int* __restrict__ level_counts, const int* __restrict__ base) {
    const int slot = blockIdx.y;
    if (slot >= *n_cur) return;
    if (!kd_huge_swap(idx, hs, lp, rp, hc, slot, blockIdx.x)) return;
    // the node's last workgroup through here does what follows the pass (a launch of its own otherwise): its swaps and everybody
    // else's are visible to it - each workgroup fences its writes before it draws its ticket, the last one fences again behind it
    __syncthreads();
    if (threadIdx.x == 0) {
        __threadfence();
        const int ticket = atomicAdd(&tickets[slot], 1);
        if (ticket == (int)gridDim.x - 1) {
            tickets[slot] = 0;
            __threadfence();
            kd_huge_advance<M>(coords, idx, hs, slot, meta, split_out, next, n_next, st, leafsize, level_counts, base);
        }
    }
}

__global__ void kd_level_base_kernel(const int* __restrict__ level_counts, int* __restrict__ base, KdBuildState* st) {
    int e = 0;
    for (int c = 0; c < KD_NCLS; ++c) e += level_counts[c];
    base[1] = base[0] + 2 * e;                                            // the first child slot of the next level's entries
    st->n_nodes = base[1];                                                // (slots handed out so far: the high-water mark)
}

__global__ void kd_init_kernel(int* __restrict__ idx, long long n, const KdQueues q, int* __restrict__ counts, int* __restrict__ bases,
                               KdBuildState* st, int4* __restrict__ meta, int leafsize) {
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) idx[i] = (int)i;
    if (i == 0) {
        st->n_nodes = 1;
        bases[0] = 1;                                                     // node 0 is the root
        st->overflow = 0;
        for (int a = 0; a < 3; ++a) { st->mins[a] = 0.0; st->maxes[a] = 0.0; }
        if (n <= leafsize) {
            meta[0] = make_int4(-1, 0, (int)n, 0);
        } else {
            const int cls = kd_size_class(n);
            q.q[cls][0] = KdWork{0, 0, (int)n};
            counts[cls] = 1;                                              // (level 0's four counters; the rest were cleared)
        }
    }
}

__global__ void kd_root_bounds_kernel(const double* __restrict__ coords, long long n, int dim, KdBuildState* st) {
    // a tree that is a single leaf: the whole-set bounds (the queries start from them) by one workgroup
    __shared__ double s[2 * 3 * 4];
    double mx[3] = {-HUGE_VAL, -HUGE_VAL, -HUGE_VAL}, mn[3] = {HUGE_VAL, HUGE_VAL, HUGE_VAL};
    for (long long i = threadIdx.x; i < n; i += blockDim.x)
        for (int a = 0; a < dim; ++a) {
            const double v = coords[i * dim + a];
            mx[a] = mx[a] > v ? mx[a] : v;
            mn[a] = mn[a] < v ? mn[a] : v;
        }
    for (int a = 0; a < 3; ++a) {
        for (int off = 32; off > 0; off >>= 1) {
            const double ox = __shfl_xor(mx[a], off), on = __shfl_xor(mn[a], off);
            mx[a] = mx[a] > ox ? mx[a] : ox;
            mn[a] = mn[a] < on ? mn[a] : on;
        }
        if ((threadIdx.x & 63) == 0) { s[((threadIdx.x >> 6) * 3 + a) * 2] = mx[a]; s[((threadIdx.x >> 6) * 3 + a) * 2 + 1] = mn[a]; }
    }
    __syncthreads();
    if (threadIdx.x == 0)
        for (int a = 0; a < dim; ++a) {
            double x = s[a * 2], m2 = s[a * 2 + 1];
            for (int wv = 1; wv < 4; ++wv) {
                x = fmax(x, s[(wv * 3 + a) * 2]);
                m2 = fmin(m2, s[(wv * 3 + a) * 2 + 1]);
            }
            st->maxes[a] = x;
            st->mins[a] = m2;
        }
}

template <int M>
int kd_launch_level(const double* coords, int* idx, int4* meta, double* split, KdWork* const (&cur)[KD_NCLS], const int* n_cur, const KdQueues& next,
                    int* n_next, KdBuildState* st, int* lp, int* rp, const long long (&grid)[3], int small_cap, int leafsize, const int* base,
                    hipStream_t s) {
    if (grid[2] > 0)
        hipLaunchKernelGGL((kd_level_kernel<M, 1024>), dim3((unsigned)grid[2]), dim3(1024), 0, s, coords, idx, meta, split, cur[2], n_cur + 2, next,
                           n_next, st, lp, rp, leafsize, 0, n_cur, 2, base);
    if (grid[1] > 0)
        hipLaunchKernelGGL((kd_level_kernel<M, 256>), dim3((unsigned)grid[1]), dim3(256), 0, s, coords, idx, meta, split, cur[1], n_cur + 1, next,
                           n_next, st, lp, rp, leafsize, 0, n_cur, 1, base);
    if (grid[0] > 0) {
        // the wave class: nodes of at most `small_cap` points (what a balanced tree holds at this level, with a factor two to spare)
        // out of LDS, the others through global memory
        hipLaunchKernelGGL((kd_small_kernel<M>), dim3((unsigned)grid[0]), dim3(64), (size_t)small_cap * 16, s, coords, idx, meta, split, cur[0], n_cur,
                           next, n_next, st, leafsize, small_cap, n_cur, base);
        if (small_cap < KD_SMALL)
            hipLaunchKernelGGL((kd_level_kernel<M, 64>), dim3((unsigned)grid[0]), dim3(64), 0, s, coords, idx, meta, split, cur[0], n_cur, next, n_next,
                               st, lp, rp, leafsize, small_cap, n_cur, 0, base);
    }
    FDX_CHECK_LAUNCH();
    return 0;
}

// the huge nodes of one level (count known: it was read back): bounds, then passes of four launches until every node has its children
template <int M>
int kd_run_huge(const double* coords, int* idx, int4* meta, double* split, const KdWork* cur3, const int* n_cur3, const int* base, int count, long long max_size,
                const KdQueues& next, int* n_next, KdBuildState* st, int* lp, int* rp, KdHuge* hs, int* hc, double* hb, int* tickets,
                int leafsize, bool* gave_up, hipStream_t s) {
    const dim3 wide(KD_CH / 4, (unsigned)count), one((unsigned)count);
    int lg = 0;
    while ((1LL << (lg + 1)) <= max_size) ++lg;
    const int budget = 2 * lg + 4;                                        // libstdc++'s depth budget + the last pass + the split passes
    hipLaunchKernelGGL((kd_huge_bounds_kernel<M>), wide, dim3(256), 0, s, coords, idx, cur3, n_cur3, hb);
    hipLaunchKernelGGL((kd_huge_setup_kernel<M>), one, dim3(256), 0, s, coords, idx, cur3, n_cur3, hb, hs, meta, st);
    FDX_CHECK_LAUNCH();
    auto round = [&]() -> int {
        hipLaunchKernelGGL((kd_huge_classify_kernel<M>), wide, dim3(256), 0, s, coords, idx, hs, n_cur3, lp, rp, hc);
        hipLaunchKernelGGL(kd_huge_count_kernel, wide, dim3(256), 0, s, hs, n_cur3, lp, rp, hc);
        hipLaunchKernelGGL((kd_huge_swap_kernel<M>), wide, dim3(256), 0, s, coords, idx, hs, n_cur3, lp, rp, hc, tickets, meta, split, next, n_next,
                           st, leafsize, n_cur3 - 3, base);
        FDX_CHECK_LAUNCH();
        return 0;
    };
    // a selection over `size` elements takes about log2(size) + a few partition passes (the budget is twice that), then one or two
    // split passes: that many rounds blind, then the phases are looked at
    int done_rounds = 0;
    for (int r = 0; r < std::min(budget, lg + 4); ++r, ++done_rounds) FDX_TRY(round());   // (sorted input: lg + 1 selection passes and a split pass)
    std::vector<KdHuge> h((size_t)count);
    for (;;) {
        FDX_HIP(hipMemcpyAsync(h.data(), hs, (size_t)count * sizeof(KdHuge), hipMemcpyDeviceToHost, s));
        FDX_HIP(hipStreamSynchronize(s));
        bool all = true;
        for (const KdHuge& x : h) all = all && x.phase == 4;
        if (all) break;
        if (done_rounds >= budget) { *gave_up = true; return 0; }
        for (int r = 0; r < 3 && done_rounds < budget; ++r, ++done_rounds) FDX_TRY(round());
    }
    return 0;
}

}  // namespace

// kdtree_dev.h
int kd_build_device(const double* coords_dev, long long n, int dim, KdDeviceTree* out, hipStream_t st) {
    if (dim < 1 || dim > 3 || n < 1 || n >= 0x3fffffffLL) return fail(FDX_ERR_INVALID, "kd_build_device: 1 to 3 coordinates, fewer than 2^30 points");
    const int leafsize = 16;
    const long long cap_nodes = 2 * n + 2;                               // every split makes two non-empty children
    FDX_TRY(out->meta.alloc((size_t)cap_nodes * sizeof(int4)));
    FDX_TRY(out->split.alloc((size_t)cap_nodes * sizeof(double)));
    FDX_TRY(out->idx.alloc((size_t)n * sizeof(int)));
    // queues of the nodes still to split, by level parity and team size: a level holds at most n / (size class's lower bound) of a class
    const long long qcap[KD_NCLS] = {n / (leafsize + 1) + 2, n / (KD_SMALL + 1) + 2, n / (KD_MID + 1) + 2, n / (KD_HUGE + 1) + 2};
    DevBuf q[2][KD_NCLS], counts, bases, state, lp, rp, hs, hc, hb, tickets;
    for (int par = 0; par < 2; ++par)
        for (int c = 0; c < KD_NCLS; ++c) FDX_TRY(q[par][c].alloc((size_t)qcap[c] * sizeof(KdWork)));
    const int max_levels = 128;
    FDX_TRY(counts.alloc((size_t)(max_levels + 2) * KD_NCLS * sizeof(int)));
    FDX_TRY(bases.alloc((size_t)(max_levels + 3) * sizeof(int)));
    FDX_TRY(state.alloc(sizeof(KdBuildState)));
    FDX_TRY(lp.alloc((size_t)n * sizeof(int)));
    FDX_TRY(rp.alloc((size_t)n * sizeof(int)));
    FDX_TRY(hs.alloc((size_t)qcap[3] * sizeof(KdHuge)));
    FDX_TRY(hc.alloc((size_t)qcap[3] * 5 * KD_CH * sizeof(int)));
    FDX_TRY(hb.alloc((size_t)qcap[3] * KD_CH * 6 * sizeof(double)));
    FDX_TRY(tickets.alloc((size_t)qcap[3] * sizeof(int)));
    FDX_HIP(hipMemsetAsync(tickets.p, 0, (size_t)qcap[3] * sizeof(int), st));
    FDX_HIP(hipMemsetAsync(counts.p, 0, (size_t)(max_levels + 2) * KD_NCLS * sizeof(int), st));
    KdQueues q0{{q[0][0].as<KdWork>(), q[0][1].as<KdWork>(), q[0][2].as<KdWork>(), q[0][3].as<KdWork>()}};
    hipLaunchKernelGGL(kd_init_kernel, dim3((unsigned)ceil_div(n, 256)), dim3(256), 0, st, out->idx.as<int>(), n, q0, counts.as<int>(),
                       bases.as<int>(), state.as<KdBuildState>(), out->meta.as<int4>(), leafsize);
    FDX_CHECK_LAUNCH();
    if (n <= leafsize) {
        hipLaunchKernelGGL(kd_root_bounds_kernel, dim3(1), dim3(256), 0, st, coords_dev, n, dim, state.as<KdBuildState>());
        FDX_CHECK_LAUNCH();
    }
    // A balanced tree of n points has ceil(log2(n / leafsize)) + 1 levels of split nodes; the split above the minimum (heavily
    // duplicated coordinates) can make it deeper: the expected levels are queued blind - level L has at most 2^L nodes, a launch
    // covers that many (and no more than the class can hold) and the surplus workgroups leave at once -, then the queue lengths
    // are looked at.  While a level can still hold huge nodes (a child is smaller than its parent: once a level has none, none
    // follows) their number is read first - their passes are launches of their own (kd_run_huge).
    int expected = 1;
    while (((long long)leafsize << (expected - 1)) < n) ++expected;
    bool huge_alive = n > KD_HUGE, gave_up = false;
    auto run_level = [&](int level) -> int {
        KdWork* cur[KD_NCLS];
        KdQueues nxt;
        for (int c = 0; c < KD_NCLS; ++c) { cur[c] = q[level & 1][c].as<KdWork>(); nxt.q[c] = q[(level + 1) & 1][c].as<KdWork>(); }
        const int* n_cur = counts.as<int>() + KD_NCLS * level;
        int* n_next = counts.as<int>() + KD_NCLS * (level + 1);
        int* base_l = bases.as<int>() + level;                            // bases[level]: first child slot of this level's entries
        hipLaunchKernelGGL(kd_level_base_kernel, dim3(1), dim3(1), 0, st, n_cur, base_l, state.as<KdBuildState>());
        int n_huge = 0;
        if (huge_alive) {
            FDX_HIP(hipMemcpyAsync(&n_huge, n_cur + 3, sizeof(int), hipMemcpyDeviceToHost, st));
            FDX_HIP(hipStreamSynchronize(st));
            if (n_huge == 0) huge_alive = false;
        }
        long long grid[3];
        for (int c = 0; c < 3; ++c) grid[c] = std::min<long long>(qcap[c] - 1, level < 40 ? (1LL << level) : qcap[c]);
        // a node of `level` forks below the root has shed at least `level` points (each split gives both sides one or more)
        if (n - level <= KD_MID) grid[2] = 0;
        if (n - level <= KD_SMALL) grid[1] = 0;
        long long bal = level < 40 ? (n >> level) + 1 : 1;                // a balanced tree's node at this level
        int small_cap = 64;
        while (small_cap < KD_SMALL && small_cap < 2 * bal) small_cap *= 2;
#define FDX_KDL(MM)                                                                                                              \
        do {                                                                                                                     \
            if (n_huge > 0)                                                                                                      \
                FDX_TRY(kd_run_huge<MM>(coords_dev, out->idx.as<int>(), out->meta.as<int4>(), out->split.as<double>(), cur[3], n_cur + 3, base_l, n_huge, \
                                        std::max<long long>(4, n - level), nxt, n_next, state.as<KdBuildState>(), lp.as<int>(), rp.as<int>(),     \
                                        hs.as<KdHuge>(), hc.as<int>(), hb.as<double>(), tickets.as<int>(), leafsize, &gave_up, st));                                  \
            return kd_launch_level<MM>(coords_dev, out->idx.as<int>(), out->meta.as<int4>(), out->split.as<double>(), cur, n_cur, nxt, n_next,     \
                                       state.as<KdBuildState>(), lp.as<int>(), rp.as<int>(), grid, small_cap, leafsize, base_l, st);              \
        } while (0)
        if (dim == 1) FDX_KDL(1);
        if (dim == 2) FDX_KDL(2);
        FDX_KDL(3);
#undef FDX_KDL
    };
    int level = 0;
    const bool trace = fdx::env("FDX_TRACE_HOST") != nullptr;
    auto now_ms = [] { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count(); };
    const double t_start = now_ms();
    if (n > leafsize) {
        for (; level < expected && level < max_levels && !gave_up; ++level) {
            const bool was_huge = huge_alive;
            const double t0 = now_ms();
            FDX_TRY(run_level(level));
            if (trace && was_huge) std::fprintf(stderr, "[fdx-host] kd device build: level %d queued at +%.2f ms, host time %.2f ms\n", level, t0 - t_start, now_ms() - t0);
        }
        while (!gave_up) {
            int left[KD_NCLS] = {0, 0, 0, 0};
            FDX_HIP(hipMemcpyAsync(left, counts.as<int>() + KD_NCLS * level, sizeof(left), hipMemcpyDeviceToHost, st));
            FDX_HIP(hipStreamSynchronize(st));
            if (left[0] + left[1] + left[2] + left[3] == 0) break;
            if (level >= max_levels) { gave_up = true; break; }
            FDX_TRY(run_level(level));
            ++level;
        }
    }
    KdBuildState hstate{};
    FDX_HIP(hipMemcpyAsync(&hstate, state.p, sizeof(hstate), hipMemcpyDeviceToHost, st));
    FDX_HIP(hipStreamSynchronize(st));
    out->n_nodes = hstate.n_nodes;
    out->overflow = gave_up || hstate.overflow != 0;
    out->levels = level;
    for (int a = 0; a < 3; ++a) { out->mins[a] = hstate.mins[a]; out->maxes[a] = hstate.maxes[a]; }
    return 0;
}

}  // namespace fdx
