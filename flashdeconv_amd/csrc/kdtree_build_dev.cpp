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
// STATE (round 6): exact - scipy's index array on lattices, clouds, heavy duplicates, 1-3 coordinates, a million points
// (tests/test_gpu_stages.py) - and NOT the default (fdx_kdtree_tune(2, 1) selects it): a million lattice points take 13.5 ms, of
// which 10.6 are levels 0-4, where a node is one workgroup on one compute unit (2.5 / 4.5 / 2.1 / 0.9 / 0.6 ms; levels 5-9, 256
// threads per node: 1.4 ms; levels 10-16, a wave per node: 1.4 ms), against 9-11 ms for the host's thread pool, which moreover
// runs BESIDE the device's own lists.  What it needs next: several workgroups per node for the top levels (each pass then is
// three or four launches - classify, rank, swap, advance - instead of a loop inside one workgroup).
#include <algorithm>
#include <cmath>
#include <cstdint>

#include "fdx_internal.h"
#include "kdtree_dev.h"

namespace fdx {
namespace {

struct KdWork { int node, start, end; };
// team sizes by node size: up to 1024 points one wave, up to 32768 a workgroup of 256, above one of 1024
constexpr int KD_SMALL = 1024, KD_MID = 32768;
__host__ __device__ inline int kd_size_class(long long size) { return size <= KD_SMALL ? 0 : size <= KD_MID ? 1 : 2; }

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
    for (int base = cb; base < ce; base += 64) {
        const int i = base + lane;
        const bool valid = i < ce;
        const double v = valid ? kd_key<M>(coords, idx[i], d) : 0.0;
        const bool sl = valid && stop_left(v), sr_ = valid && stop_right(v);
        const unsigned long long ml = __ballot(sl), mr = __ballot(sr_);
        if (sl) lp[cb + nl + __popcll(ml & lt)] = i;
        if (sr_) rp[cb + nr + __popcll(mr & lt)] = i;
        nl += __popcll(ml);
        nr += __popcll(mr);
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
                                                     const int* __restrict__ n_cur, KdWork* __restrict__ next0, KdWork* __restrict__ next1,
                                                     KdWork* __restrict__ next2, int* __restrict__ n_next,
                                                     KdBuildState* __restrict__ st, int* __restrict__ lp, int* __restrict__ rp, int leafsize) {
#pragma clang fp contract(off)
    __shared__ int s_nl[T / 64], s_nr[T / 64], s_cnt[T / 64], s_out[8];
    __shared__ double s_red[2 * 3 * (T / 64)];
    __shared__ double s_bounds[6];
    __shared__ int s_ctl[4];
    const int tid = threadIdx.x;
    if ((int)blockIdx.x >= *n_cur) return;
    const KdWork w = cur[blockIdx.x];
    const int size = w.end - w.start;
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
            if (depth == 0) {
                if (tid == 0) atomicExch(&st->overflow, 1);
                return;                                                   // (uniform: every thread sees the same depth)
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
        const int c0 = atomicAdd(&st->n_nodes, 2);
        meta[w.node] = make_int4(d, c0, c0 + 1, 0);
        split_out[w.node] = split;
        const int cs[2] = {w.start, p}, ce[2] = {p, w.end};
        for (int c = 0; c < 2; ++c) {
            if (ce[c] - cs[c] <= leafsize) {
                meta[c0 + c] = make_int4(-1, cs[c], ce[c], 0);
            } else {                                                       // the next level's queue of its team size
                const int cls = kd_size_class(ce[c] - cs[c]);
                const int slot = atomicAdd(n_next + cls, 1);
                (cls == 0 ? next0 : cls == 1 ? next1 : next2)[slot] = KdWork{c0 + c, cs[c], ce[c]};
            }
        }
    }
}

__global__ void kd_init_kernel(int* __restrict__ idx, long long n, KdWork* __restrict__ q0, KdWork* __restrict__ q1, KdWork* __restrict__ q2,
                               int* __restrict__ counts, KdBuildState* st, int4* __restrict__ meta, int leafsize) {
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) idx[i] = (int)i;
    if (i == 0) {
        st->n_nodes = 1;
        st->overflow = 0;
        for (int a = 0; a < 3; ++a) { st->mins[a] = 0.0; st->maxes[a] = 0.0; }
        if (n <= leafsize) {
            meta[0] = make_int4(-1, 0, (int)n, 0);
        } else {
            const int cls = kd_size_class(n);
            (cls == 0 ? q0 : cls == 1 ? q1 : q2)[0] = KdWork{0, 0, (int)n};
            counts[cls] = 1;                                              // (level 0's three counters; the rest were cleared)
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
int kd_launch_level(const double* coords, int* idx, int4* meta, double* split, KdWork* const (&cur)[3], const int* n_cur, KdWork* const (&next)[3],
                    int* n_next, KdBuildState* st, int* lp, int* rp, const long long (&grid)[3], int leafsize, hipStream_t s) {
    if (grid[2] > 0)
        hipLaunchKernelGGL((kd_level_kernel<M, 1024>), dim3((unsigned)grid[2]), dim3(1024), 0, s, coords, idx, meta, split, cur[2], n_cur + 2, next[0],
                           next[1], next[2], n_next, st, lp, rp, leafsize);
    if (grid[1] > 0)
        hipLaunchKernelGGL((kd_level_kernel<M, 256>), dim3((unsigned)grid[1]), dim3(256), 0, s, coords, idx, meta, split, cur[1], n_cur + 1, next[0],
                           next[1], next[2], n_next, st, lp, rp, leafsize);
    if (grid[0] > 0)
        hipLaunchKernelGGL((kd_level_kernel<M, 64>), dim3((unsigned)grid[0]), dim3(64), 0, s, coords, idx, meta, split, cur[0], n_cur, next[0], next[1],
                           next[2], n_next, st, lp, rp, leafsize);
    FDX_CHECK_LAUNCH();
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
    const long long qcap[3] = {n / (leafsize + 1) + 2, n / (KD_SMALL + 1) + 2, n / (KD_MID + 1) + 2};
    DevBuf q[2][3], counts, state, lp, rp;
    for (int par = 0; par < 2; ++par)
        for (int c = 0; c < 3; ++c) FDX_TRY(q[par][c].alloc((size_t)qcap[c] * sizeof(KdWork)));
    const int max_levels = 128;
    FDX_TRY(counts.alloc((size_t)(max_levels + 2) * 3 * sizeof(int)));
    FDX_TRY(state.alloc(sizeof(KdBuildState)));
    FDX_TRY(lp.alloc((size_t)n * sizeof(int)));
    FDX_TRY(rp.alloc((size_t)n * sizeof(int)));
    FDX_HIP(hipMemsetAsync(counts.p, 0, (size_t)(max_levels + 2) * 3 * sizeof(int), st));
    hipLaunchKernelGGL(kd_init_kernel, dim3((unsigned)ceil_div(n, 256)), dim3(256), 0, st, out->idx.as<int>(), n, q[0][0].as<KdWork>(),
                       q[0][1].as<KdWork>(), q[0][2].as<KdWork>(), counts.as<int>(), state.as<KdBuildState>(), out->meta.as<int4>(), leafsize);
    FDX_CHECK_LAUNCH();
    if (n <= leafsize) {
        hipLaunchKernelGGL(kd_root_bounds_kernel, dim3(1), dim3(256), 0, st, coords_dev, n, dim, state.as<KdBuildState>());
        FDX_CHECK_LAUNCH();
    }
    // A balanced tree of n points has ceil(log2(n / leafsize)) + 1 levels of split nodes; the split above the minimum (heavily
    // duplicated coordinates) can make it deeper: the expected levels are queued blind - level L has at most 2^L nodes, a launch
    // covers that many (and no more than the class can hold) and the surplus workgroups leave at once -, then the queue lengths
    // are looked at.
    int expected = 1;
    while (((long long)leafsize << (expected - 1)) < n) ++expected;
    auto run_level = [&](int level) -> int {
        long long grid[3];
        for (int c = 0; c < 3; ++c) grid[c] = std::min<long long>(qcap[c] - 1, level < 40 ? (1LL << level) : qcap[c]);
        // a node of `level` forks below the root has shed at least `level` points (each split gives both sides one or more)
        if (n - level <= KD_MID) grid[2] = 0;
        if (n - level <= KD_SMALL) grid[1] = 0;
        KdWork* cur[3] = {q[level & 1][0].as<KdWork>(), q[level & 1][1].as<KdWork>(), q[level & 1][2].as<KdWork>()};
        KdWork* nxt[3] = {q[(level + 1) & 1][0].as<KdWork>(), q[(level + 1) & 1][1].as<KdWork>(), q[(level + 1) & 1][2].as<KdWork>()};
        const int* n_cur = counts.as<int>() + 3 * level;
        int* n_next = counts.as<int>() + 3 * (level + 1);
#define FDX_KDL(MM)                                                                                                              \
        return kd_launch_level<MM>(coords_dev, out->idx.as<int>(), out->meta.as<int4>(), out->split.as<double>(), cur, n_cur, nxt, n_next, \
                                   state.as<KdBuildState>(), lp.as<int>(), rp.as<int>(), grid, leafsize, st)
        if (dim == 1) FDX_KDL(1);
        if (dim == 2) FDX_KDL(2);
        FDX_KDL(3);
#undef FDX_KDL
    };
    int level = 0;
    if (n > leafsize) {
        for (; level < expected && level < max_levels; ++level) FDX_TRY(run_level(level));
        for (;;) {
            int left[3] = {0, 0, 0};
            FDX_HIP(hipMemcpyAsync(left, counts.as<int>() + 3 * level, sizeof(left), hipMemcpyDeviceToHost, st));
            FDX_HIP(hipStreamSynchronize(st));
            if (left[0] + left[1] + left[2] == 0) break;
            if (level >= max_levels) { out->overflow = true; return 0; }
            FDX_TRY(run_level(level));
            ++level;
        }
    }
    KdBuildState hs{};
    FDX_HIP(hipMemcpyAsync(&hs, state.p, sizeof(hs), hipMemcpyDeviceToHost, st));
    FDX_HIP(hipStreamSynchronize(st));
    out->n_nodes = hs.n_nodes;
    out->overflow = hs.overflow != 0;
    out->levels = level;
    for (int a = 0; a < 3; ++a) { out->mins[a] = hs.mins[a]; out->maxes[a] = hs.maxes[a]; }
    return 0;
}

}  // namespace fdx
