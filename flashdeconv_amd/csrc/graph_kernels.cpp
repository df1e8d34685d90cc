// Spatial graph construction on the device.
//
// Replaces flashdeconv/utils/graph.py:
//   build_knn_graph    :25-83   (cKDTree build + query k+1 incl. self, drop self, A + A^T, binary)
//   build_radius_graph :86-133  (cKDTree.query_pairs(r), symmetric)
// and hands the result to the solver in the sliced-ELL layout of fdx_graph.h.
//
// Pipeline (all on the stream, one small D2H for the bounding box):
//   1. bounding box -> uniform grid with ~4 points per cell (k-NN; radius graphs: cell edge >= radius).
//   2. order by (Morton code of the cell, original index) -> perm / rank: count per key, scan, rank within the cell
//      (rocPRIM's stable radix sort above 4M keys; same order either way).  Points inside a cell keep caller order,
//      and any 256 consecutive sorted points form a compact patch (small tile halo in the BCD sweep).
//   3. exact k-NN: one lane per point scans the cells of growing Chebyshev shells until the k+1-th best squared distance
//      is provably inside the scanned block.  Squared distances are evaluated in float64 WITHOUT fma contraction
//      ((dx*dx + dy*dy) + dz*dz, each rounded) and ties are broken by the lower original index.
//   4. union symmetrisation: in-degree count (inside the k-NN kernel for whole-graph builds), reverse lists, per-row sort
//      by original index + unique.
//   5. sliced ELL (slice = 64 consecutive sorted points = one wavefront of the BCD sweep).
#include "fdx_env.h"
#include <chrono>
#include <functional>
#include <memory>
#include <mutex>
#include <cstring>

#include <rocprim/device/device_radix_sort.hpp>
#include <rocprim/device/device_reduce.hpp>
#include <rocprim/device/device_scan.hpp>
#include <rocprim/device/device_select.hpp>
#include <rocprim/iterator/transform_iterator.hpp>

#include <algorithm>
#include <cmath>
#include <vector>

#include "fdx_graph.h"
#include "fdx_internal.h"
#include "fdx_kernels.h"
#include "graph_build.h"

namespace fdx {

struct GridParams {
    double mn[3];
    double inv_h[3];
    double h[3];
    int nc[3];       // cells per axis (1 for unused / zero-extent axes)
    int stride[3];   // cell id = sum_a c_a * stride[a]
    int dim;
};

// ------------------------------------------------------------------------------------------------ bbox
__device__ __forceinline__ double wave_min_f64(double v) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v = fmin(v, __shfl_xor(v, off, 64));
    return v;
}
__device__ __forceinline__ double wave_max_f64(double v) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v = fmax(v, __shfl_xor(v, off, 64));
    return v;
}
// (the tree reduction of six 256-entry LDS arrays this kernel used to end with - eight barrier-separated steps - made a 16 MB
// read take 37-47 us; the loop itself is a few microseconds: wave shuffles, then four values per quantity through LDS)
__global__ __launch_bounds__(256) void bbox_partial_kernel(const double* __restrict__ coords, long long n, int dim,
                                                           double* __restrict__ part /* (nblk, 6) */) {
    __shared__ double s_v[6][4];
    double mn[3] = {INFINITY, INFINITY, INFINITY}, mx[3] = {-INFINITY, -INFINITY, -INFINITY};
    if (dim == 2 && ((unsigned long long)coords & 15ULL) == 0ULL) {      // one 16-byte load per point
        const double2* c2 = reinterpret_cast<const double2*>(coords);
        const long long stride = (long long)gridDim.x * 256;
        long long i = blockIdx.x * 256LL + threadIdx.x;
        for (; i + 3 * stride < n; i += 4 * stride) {          // four loads in flight per thread
            const double2 a = c2[i], b = c2[i + stride], c = c2[i + 2 * stride], d = c2[i + 3 * stride];
            mn[0] = fmin(fmin(mn[0], a.x), fmin(fmin(b.x, c.x), d.x)); mx[0] = fmax(fmax(mx[0], a.x), fmax(fmax(b.x, c.x), d.x));
            mn[1] = fmin(fmin(mn[1], a.y), fmin(fmin(b.y, c.y), d.y)); mx[1] = fmax(fmax(mx[1], a.y), fmax(fmax(b.y, c.y), d.y));
        }
        for (; i < n; i += stride) {
            const double2 v = c2[i];
            mn[0] = fmin(mn[0], v.x); mx[0] = fmax(mx[0], v.x);
            mn[1] = fmin(mn[1], v.y); mx[1] = fmax(mx[1], v.y);
        }
    } else {
        for (long long i = blockIdx.x * 256LL + threadIdx.x; i < n; i += (long long)gridDim.x * 256)
            for (int a = 0; a < dim; ++a) {
                const double v = coords[(size_t)i * dim + a];
                mn[a] = fmin(mn[a], v);
                mx[a] = fmax(mx[a], v);
            }
    }
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
#pragma unroll
    for (int a = 0; a < 3; ++a) {
        const double lo = wave_min_f64(mn[a]), hi = wave_max_f64(mx[a]);
        if (lane == 0) { s_v[a][wv] = lo; s_v[3 + a][wv] = hi; }
    }
    __syncthreads();
    if (threadIdx.x < 3) {
        const int a = threadIdx.x;
        part[(size_t)blockIdx.x * 6 + a] = fmin(fmin(s_v[a][0], s_v[a][1]), fmin(s_v[a][2], s_v[a][3]));
        part[(size_t)blockIdx.x * 6 + 3 + a] = fmax(fmax(s_v[3 + a][0], s_v[3 + a][1]), fmax(s_v[3 + a][2], s_v[3 + a][3]));
    }
}

// the blocks' boxes folded into one, written where `out` points (the host's pinned block): no copy to wait for
__global__ __launch_bounds__(256) void bbox_final_kernel(const double* __restrict__ part, int nblk, double* __restrict__ out) {
    __shared__ double smn[3][256], smx[3][256];
    double mn[3] = {INFINITY, INFINITY, INFINITY}, mx[3] = {-INFINITY, -INFINITY, -INFINITY};
    bool bad = false;
    for (int b = threadIdx.x; b < nblk; b += 256)
        for (int a = 0; a < 3; ++a) {
            const double lo = part[(size_t)b * 6 + a], hi = part[(size_t)b * 6 + 3 + a];
            bad = bad || lo != lo || hi != hi;           // fmin / fmax drop a NaN: carry it by hand
            mn[a] = fmin(mn[a], lo);
            mx[a] = fmax(mx[a], hi);
        }
    for (int a = 0; a < 3; ++a) { smn[a][threadIdx.x] = bad ? NAN : mn[a]; smx[a][threadIdx.x] = mx[a]; }
    __syncthreads();
    for (int s = 128; s > 0; s >>= 1) {
        if ((int)threadIdx.x < s)
            for (int a = 0; a < 3; ++a) {
                const double x = smn[a][threadIdx.x], y = smn[a][threadIdx.x + s];
                smn[a][threadIdx.x] = (x != x || y != y) ? NAN : fmin(x, y);
                smx[a][threadIdx.x] = fmax(smx[a][threadIdx.x], smx[a][threadIdx.x + s]);
            }
        __syncthreads();
    }
    if (threadIdx.x < 3) {
        out[threadIdx.x] = smn[threadIdx.x][0];
        out[3 + threadIdx.x] = smx[threadIdx.x][0];
    }
}

// ------------------------------------------------------------------------------------------------ binning
__device__ __forceinline__ int cell_coord(double x, double mn, double inv_h, int nc) {
    int c = (int)floor((x - mn) * inv_h);
    return max(0, min(nc - 1, c));
}

// Morton (Z-order) code of a cell: 256 consecutive points of the sorted order form a compact 2-D/3-D patch, which keeps
// the halo of a 256-spot workgroup tile of the BCD sweep small (perimeter instead of two full grid rows).
__host__ __device__ __forceinline__ unsigned long long spread_bits_2(unsigned long long v) {   // abcd -> 0a0b0c0d
    v &= 0xffffffffULL;
    v = (v | (v << 16)) & 0x0000ffff0000ffffULL;
    v = (v | (v << 8)) & 0x00ff00ff00ff00ffULL;
    v = (v | (v << 4)) & 0x0f0f0f0f0f0f0f0fULL;
    v = (v | (v << 2)) & 0x3333333333333333ULL;
    v = (v | (v << 1)) & 0x5555555555555555ULL;
    return v;
}
__host__ __device__ __forceinline__ unsigned long long spread_bits_3(unsigned long long v) {   // 21 bits -> every third bit
    v &= 0x1fffffULL;
    v = (v | (v << 32)) & 0x1f00000000ffffULL;
    v = (v | (v << 16)) & 0x1f0000ff0000ffULL;
    v = (v | (v << 8)) & 0x100f00f00f00f00fULL;
    v = (v | (v << 4)) & 0x10c30c30c30c30c3ULL;
    v = (v | (v << 2)) & 0x1249249249249249ULL;
    return v;
}
__device__ __forceinline__ unsigned long long morton_key(const int c[3], int dim) {
    if (dim == 1) return (unsigned long long)c[0];
    if (dim == 2) return spread_bits_2((unsigned)c[0]) | (spread_bits_2((unsigned)c[1]) << 1);
    return spread_bits_3((unsigned)c[0]) | (spread_bits_3((unsigned)c[1]) << 1) | (spread_bits_3((unsigned)c[2]) << 2);
}
__device__ __forceinline__ unsigned compact_bits_2(unsigned long long v) {                 // 0a0b0c0d -> abcd
    v &= 0x5555555555555555ULL;
    v = (v | (v >> 1)) & 0x3333333333333333ULL;
    v = (v | (v >> 2)) & 0x0f0f0f0f0f0f0f0fULL;
    v = (v | (v >> 4)) & 0x00ff00ff00ff00ffULL;
    v = (v | (v >> 8)) & 0x0000ffff0000ffffULL;
    v = (v | (v >> 16)) & 0x00000000ffffffffULL;
    return (unsigned)v;
}
__device__ __forceinline__ unsigned compact_bits_3(unsigned long long v) {                 // every third bit -> 21 bits
    v &= 0x1249249249249249ULL;
    v = (v | (v >> 2)) & 0x10c30c30c30c30c3ULL;
    v = (v | (v >> 4)) & 0x100f00f00f00f00fULL;
    v = (v | (v >> 8)) & 0x1f0000ff0000ffULL;
    v = (v | (v >> 16)) & 0x1f00000000ffffULL;
    v = (v | (v >> 32)) & 0x1fffffULL;
    return (unsigned)v;
}

__global__ __launch_bounds__(256) void cell_key_kernel(const double* __restrict__ coords, long long n, GridParams gp,
                                                       unsigned long long* __restrict__ keys, int* __restrict__ vals) {
    const long long i = blockIdx.x * 256LL + threadIdx.x;
    if (i >= n) return;
    int c[3] = {0, 0, 0};
    for (int a = 0; a < gp.dim; ++a) c[a] = cell_coord(coords[(size_t)i * gp.dim + a], gp.mn[a], gp.inv_h[a], gp.nc[a]);
    keys[i] = morton_key(c, gp.dim);
    vals[i] = (int)i;
}

// cell table (row-major cell id -> [start, end) in the sorted order); equal Morton key <=> same cell
__global__ __launch_bounds__(256) void cell_range_kernel(const unsigned long long* __restrict__ skeys,
                                                         const double* __restrict__ sc, long long n, GridParams gp,
                                                         int* __restrict__ cstart, int* __restrict__ cend) {
    const long long p = blockIdx.x * 256LL + threadIdx.x;
    if (p >= n) return;
    const unsigned long long k = skeys[p];
    const bool first = (p == 0 || skeys[p - 1] != k), last = (p == n - 1 || skeys[p + 1] != k);
    if (!first && !last) return;
    int id = 0;
    for (int a = 0; a < gp.dim; ++a) id += cell_coord(sc[(size_t)a * n + p], gp.mn[a], gp.inv_h[a], gp.nc[a]) * gp.stride[a];
    if (first) cstart[id] = (int)p;
    if (last) cend[id] = (int)p + 1;
}

// sorted coordinate planes sc[a*n + p] and rank[perm[p]] = p
__global__ __launch_bounds__(256) void gather_sorted_kernel(const double* __restrict__ coords, const int* __restrict__ perm,
                                                            long long n, int dim, double* __restrict__ sc,
                                                            int* __restrict__ rank, double2* __restrict__ sc2) {
    const long long p = blockIdx.x * 256LL + threadIdx.x;
    if (p >= n) return;
    const int o = perm[p];
    for (int a = 0; a < 3; ++a) sc[(size_t)a * n + p] = (a < dim) ? coords[(size_t)o * dim + a] : 0.0;
    if (sc2) sc2[p] = make_double2(coords[(size_t)o * dim], dim > 1 ? coords[(size_t)o * dim + 1] : 0.0);
    rank[o] = (int)p;
}

// ---- binning without a sort (Morton key space of at most a few million bins): count per key, scan, place.
// The order produced is the stable sort's: cells in Morton order, points of a cell by ascending index.
// pass 1: key of every point, and its arrival number among the points of the same key (any order)
__global__ __launch_bounds__(256) void cell_count_kernel(const double* __restrict__ coords, long long n, GridParams gp,
                                                         unsigned* __restrict__ key32, int* __restrict__ arrival,
                                                         int* __restrict__ count) {
    const long long i = blockIdx.x * 256LL + threadIdx.x;
    if (i >= n) return;
    int c[3] = {0, 0, 0};
    for (int a = 0; a < gp.dim; ++a) c[a] = cell_coord(coords[(size_t)i * gp.dim + a], gp.mn[a], gp.inv_h[a], gp.nc[a]);
    const unsigned k = (unsigned)morton_key(c, gp.dim);
    key32[i] = k;
    arrival[i] = atomicAdd(&count[k], 1);
}
// pass 2: the members of every key, contiguous, in arrival order
__global__ __launch_bounds__(256) void cell_place_kernel(const unsigned* __restrict__ key32, const int* __restrict__ arrival,
                                                         const int* __restrict__ start, long long n, int* __restrict__ members) {
    const long long i = blockIdx.x * 256LL + threadIdx.x;
    if (i >= n) return;
    members[start[key32[i]] + arrival[i]] = (int)i;
}
// pass 3: position of point i = start of its key + number of members with a smaller index (cells hold a handful of
// points); writes everything the sorted order defines: perm, rank, sorted coordinate planes
__global__ __launch_bounds__(256) void cell_rank_kernel(const double* __restrict__ coords, const unsigned* __restrict__ key32,
                                                        const int* __restrict__ start, const int* __restrict__ members,
                                                        long long n, int dim, int* __restrict__ perm, int* __restrict__ rank,
                                                        double* __restrict__ sc, double2* __restrict__ sc2,
                                                        const unsigned char* __restrict__ need) {
    // need != NULL (a spot shard's binning): only the cells the shard will look at are laid out - perm / rank / coordinates of
    // the others are never written, and never read (cell_need_kernel)
    // thread t takes the point that ARRIVED at slot t: its sorted position lies in the same cell's range as t, so the writes of
    // a wave (perm, the coordinate planes, the pairs) fall into one contiguous stretch - five scattered stores per point became
    // two gathers (key, coordinates) and one scattered store (rank)
    const long long t = blockIdx.x * 256LL + threadIdx.x;
    if (t >= n) return;
    const int i = members[t];
    const unsigned k = key32[i];
    if (need && !need[k]) return;
    const int s = start[k], e = start[k + 1];
    int below = 0;
    for (int q = s; q < e; ++q) below += (members[q] < i) ? 1 : 0;
    const int p = s + below;
    perm[p] = i;
    rank[i] = p;
    for (int a = 0; a < dim; ++a) sc[(size_t)a * n + p] = coords[(size_t)i * dim + a];   // planes past dim: zeroed by the caller (one contiguous fill)
    if (sc2) sc2[p] = make_double2(coords[(size_t)i * dim], dim > 1 ? coords[(size_t)i * dim + 1] : 0.0);   // (x, y) pairs: one 16-byte gather per k-NN candidate
}

// pass 4: the cell table (row-major cell id -> [start, end) of the sorted order) straight from the key starts: an occupied
// key IS a cell, its Morton code gives the cell coordinates back
__global__ __launch_bounds__(256) void cell_table_kernel(const int* __restrict__ start, long long bins, GridParams gp,
                                                         int* __restrict__ cstart, int* __restrict__ cend) {
    const long long k = blockIdx.x * 256LL + threadIdx.x;
    if (k >= bins) return;
    const int s0 = start[k], s1 = start[k + 1];
    if (s1 <= s0) return;
    int c[3] = {0, 0, 0};
    if (gp.dim == 1) c[0] = (int)k;
    else if (gp.dim == 2) { c[0] = (int)compact_bits_2((unsigned long long)k); c[1] = (int)compact_bits_2((unsigned long long)k >> 1); }
    else { c[0] = (int)compact_bits_3((unsigned long long)k); c[1] = (int)compact_bits_3((unsigned long long)k >> 1); c[2] = (int)compact_bits_3((unsigned long long)k >> 2); }
    const int id = c[0] * gp.stride[0] + c[1] * gp.stride[1] + c[2] * gp.stride[2];
    cstart[id] = s0;
    cend[id] = s1;
}

// A spot shard owns positions [lo, hi) of the sorted order.  Its build looks at: the own rows, the rows of cells within BAND_R cells
// of an own cell (the band, whose lists it finds itself), and - walking those lists - cells within BAND_R shells of a band cell.
// Only those cells need their points laid out (perm, rank, sorted coordinates): the ranking pass, the one pass of the binning
// that gathers coordinates and scatters, then costs the shard's share instead of all n points.  Two dilations by BAND_R of the
// set of own cells (keys whose range meets [lo, hi)); a walk that goes further reports itself (knn_kernel: far) and the build
// is redone by exchange with a full binning.
__device__ __forceinline__ void cell_of_key(long long k, int dim, int c[3]) {
    c[0] = c[1] = c[2] = 0;
    if (dim == 1) c[0] = (int)k;
    else if (dim == 2) { c[0] = (int)compact_bits_2((unsigned long long)k); c[1] = (int)compact_bits_2((unsigned long long)k >> 1); }
    else { c[0] = (int)compact_bits_3((unsigned long long)k); c[1] = (int)compact_bits_3((unsigned long long)k >> 1); c[2] = (int)compact_bits_3((unsigned long long)k >> 2); }
}
template <int PASS>
__global__ __launch_bounds__(256) void cell_need_kernel(const int* __restrict__ start, long long bins, GridParams gp, long long lo,
                                                        long long hi, int R, const unsigned char* __restrict__ in,
                                                        unsigned char* __restrict__ out) {
    const long long k = blockIdx.x * 256LL + threadIdx.x;
    if (k >= bins) return;
    const bool seed = PASS == 0 ? (start[k + 1] > start[k] && (long long)start[k] < hi && (long long)start[k + 1] > lo) : (in[k] != 0);
    if (!seed) return;
    int c[3];
    cell_of_key(k, gp.dim, c);
    const int r1 = gp.dim > 1 ? R : 0, r2 = gp.dim > 2 ? R : 0;
    for (int dz = -r2; dz <= r2; ++dz)
        for (int dy = -r1; dy <= r1; ++dy)
            for (int dx = -R; dx <= R; ++dx) {
                int cc[3] = {c[0] + dx, c[1] + dy, c[2] + dz};
                if (cc[0] < 0 || cc[0] >= gp.nc[0] || cc[1] < 0 || cc[1] >= gp.nc[1] || cc[2] < 0 || cc[2] >= gp.nc[2]) continue;
                const unsigned long long kk = morton_key(cc, gp.dim);
                if ((long long)kk < bins) out[kk] = 1;
            }
}

// ------------------------------------------------------------------------------------------------ k-NN
__device__ __forceinline__ double dist2_exact(double dx, double dy, double dz) {
    // sum of squares with every product and sum rounded, as a host float64 loop computes it: no fma contraction (the
    // compiler's default for device code, and __dmul_rn / __dadd_rn are plain operators to it)
#pragma clang fp contract(off)
    const double xx = dx * dx, yy = dy * dy, zz = dz * dz;
    return (xx + yy) + zz;
}

// one slot of an ascending list: (slot, carried) <- (min, max).  Distances are never NaN, and the plain instructions spare
// the canonicalising v_max_f64 x, x that fmin / fmax put in front of every slot value coming out of a loop.
__device__ __forceinline__ void minmax_f64(double& slot, double& carried) {
    double lo, hi;
    asm("v_min_f64 %0, %2, %3\n\tv_max_f64 %1, %2, %3" : "=&v"(lo), "=&v"(hi) : "v"(slot), "v"(carried));
    slot = lo;
    carried = hi;
}

// kk = k+1 nearest INCLUDING self (cKDTree.query(coords, k+1), graph.py:63); nbr_out has stride kk, -1 padded, entries in
// no particular order (the symmetrisation sorts rows by original index).
//
// What the kernel waits for is its gathers (SQ counters: waves parked on s_waitcnt 68 % of their life, VALU issuing 15 %): a
// wave's 64 lanes look into ~16 different cell neighbourhoods, so every load instruction is ~20 cache lines for the texture
// path.  Hence: ONE gather per candidate (the (x, y) pair from sc2; the caller index perm[q] is not loaded at all), and a
// per-lane list of (squared distance, position) ordered by DISTANCE ONLY - among equal distances the first met stays ahead.
// That list holds the right neighbour SET unless the kk-th and (kk+1)-th distances are equal; waves where some lane has such
// a tie walk the candidates a second time and give the places at the threshold distance to the lowest caller indices (the
// rule of the (distance, index) order).  A list slot costs v_min_f64 + v_max_f64 + one compare + two selects.
template <int KMAX, int BATCH>
__global__ __launch_bounds__(128) void knn_kernel(const double* __restrict__ sc, const double2* __restrict__ sc2,
                                                  const int* __restrict__ perm, const int* __restrict__ rank,
                                                  const int* __restrict__ cstart, const int* __restrict__ cend,
                                                  long long n, GridParams gp, int kk, int* __restrict__ nbr_out,
                                                  int* __restrict__ nbr_cnt, double* __restrict__ nn_dist,
                                                  long long lo, long long hi, int* __restrict__ indeg,
                                                  int* __restrict__ arrival, int* __restrict__ tie_count, int gp_no_batch,
                                                  const int* __restrict__ row_list, const int* __restrict__ row_count,
                                                  int* __restrict__ far_flag, int far_R, int far_drop, int n_direct, int list_cap) {
    // rows [lo, hi) of the sorted order, or (row_list != NULL: the band of a spot shard) the first min(*row_count, hi) listed rows,
    // or (n_direct >= 0: a spot shard's own rows AND its band in one launch - each of the two launches lasted one walk's latency,
    // ~50 us, whatever its size) rows lo .. lo + n_direct - 1 followed by the first min(*row_count, list_cap) listed rows; ties and
    // far walks are counted for the own rows only
    long long p = lo + blockIdx.x * (long long)blockDim.x + threadIdx.x;
    bool listed = false, ghost = false;
    if (n_direct >= 0) {
        const long long i = p - lo;
        if (i >= n_direct) {
            const long long j = i - n_direct;
            if (j >= list_cap || j >= (long long)*row_count) return;
            p = row_list[j];
            listed = true;
        }
    } else if (row_list) {
        if (p >= hi || p >= (long long)*row_count) return;
        p = row_list[p];
    } else if (p >= hi) {
        if (!indeg) return;
        ghost = true;                                     // whole-graph build: the block's in-degree counters meet at barriers below -
        p = hi - 1;                                       // a lane past the end walks as the last row and writes nothing
    }
    const double px = sc[p], py = sc[(size_t)n + p], pz = sc[2 * (size_t)n + p];
    const double pc[3] = {px, py, pz};
    int c[3];
    for (int a = 0; a < 3; ++a) c[a] = (a < gp.dim) ? cell_coord(pc[a], gp.mn[a], gp.inv_h[a], gp.nc[a]) : 0;
    double bd[KMAX];
    int bq[KMAX];
#pragma unroll
    for (int s = 0; s < KMAX; ++s) { bd[s] = INFINITY; bq[s] = -1; }
    const bool skip_self = nn_dist != nullptr;            // nearest OTHER point: self never enters the list

    // done when the kk-th best is provably inside the block of cells within R of c
    auto covered = [&](int R) -> bool {
        bool covers_all = true;
        double safe = INFINITY;
        for (int a = 0; a < gp.dim; ++a) {
            const double slack = 1e-12 * (fabs(pc[a]) + gp.h[a] * (double)gp.nc[a]);
            if (c[a] - R > 0) {
                covers_all = false;
                safe = fmin(safe, pc[a] - (gp.mn[a] + (double)(c[a] - R) * gp.h[a]) - slack);
            }
            if (c[a] + R < gp.nc[a] - 1) {
                covers_all = false;
                safe = fmin(safe, (gp.mn[a] + (double)(c[a] + R + 1) * gp.h[a]) - pc[a] - slack);
            }
        }
        if (covers_all) return true;
        double kth = INFINITY;                            // bd[kk-1] without dynamic register indexing
#pragma unroll
        for (int s = 0; s < KMAX; ++s) if (s == (skip_self ? 0 : kk - 1)) kth = bd[s];
        return safe > 0.0 && kth < safe * safe;
    };

    // Shells 0 and 1 together (one and two dimensions: the 3 x 3 block of cells, ~36 candidates at ~4 points per cell - with
    // k <= 8 that block always suffices), as a FLAT candidate list served in batches of BATCH gathers per round trip
    // (walking cell by cell and candidate by candidate is two dependent loads deep each time: ~45 round trips per point).
    // The nine cells' (first position, count) sit in LDS, one column per lane: the walk steps through them with a running
    // cell number, and a register array indexed by it is nine selects per candidate.  A lane only ever reads its own column -
    // no barrier anywhere.
    const bool use_block = KMAX <= 16 && gp.dim <= 2 && sc2 != nullptr && !gp_no_batch;
    __shared__ int s_cs[9][128], s_cn[9][128];
    const int tid = threadIdx.x;
    int total = 0;
    if (use_block) {
#pragma unroll
        for (int j = 0; j < 9; ++j) {
            const int x = c[0] + (j % 3) - 1, y = c[1] + (j / 3) - 1;
            const bool ok = x >= 0 && x < gp.nc[0] && y >= 0 && y < gp.nc[1];
            const int cell = ok ? x * gp.stride[0] + y * gp.stride[1] : 0;
            const int a = ok ? cstart[cell] : 0, b = ok ? cend[cell] : 0;
            s_cs[j][tid] = a;
            s_cn[j][tid] = b - a;
            total += b - a;
        }
    }
    auto walk_block = [&](auto&& f) {
        int j = -1, rem = 0, q = 0;
        for (int base = 0; base < total; base += BATCH) {
            int qv[BATCH];
            double2 xy[BATCH];
#pragma unroll
            for (int u = 0; u < BATCH; ++u) {
                qv[u] = -1;
                if (base + u < total) {
                    while (rem == 0) { ++j; q = s_cs[j][tid]; rem = s_cn[j][tid]; }   // base + u < total: a non-empty cell is ahead
                    qv[u] = q++;
                    --rem;
                    xy[u] = sc2[qv[u]];
                }
            }
#pragma unroll
            for (int u = 0; u < BATCH; ++u)
                if (qv[u] >= 0) f(qv[u], dist2_exact(xy[u].x - px, xy[u].y - py, 0.0));   // planes past dim are zero
        }
    };
    // the shell max_a |dc_a| == R of the cell block around c
    auto walk_shell = [&](int R, auto&& f) {
        const int lo0 = max(0, c[0] - R), hi0 = min(gp.nc[0] - 1, c[0] + R);
        const int lo1 = max(0, c[1] - R), hi1 = min(gp.nc[1] - 1, c[1] + R);
        const int lo2 = max(0, c[2] - R), hi2 = min(gp.nc[2] - 1, c[2] + R);
        for (int z = lo2; z <= hi2; ++z)
            for (int y = lo1; y <= hi1; ++y) {
                const bool edge_zy = (abs(z - c[2]) == R) || (abs(y - c[1]) == R);
                for (int x = lo0; x <= hi0; ++x) {
                    if (!edge_zy && abs(x - c[0]) != R) {      // interior of the shell: jump to the far face
                        if (x < c[0] + R) { x = c[0] + R - 1; }
                        continue;
                    }
                    const int cell = x * gp.stride[0] + y * gp.stride[1] + z * gp.stride[2];
                    const int s0 = cstart[cell], s1 = cend[cell];
                    for (int q = s0; q < s1; ++q)
                        f(q, dist2_exact(sc[q] - px, sc[(size_t)n + q] - py, sc[2 * (size_t)n + q] - pz));
                }
            }
    };

    // ---- the walk: the KMAX nearest by distance
    auto keep = [&](int q, double d2) {
        if (skip_self && q == (int)p) return;
#pragma unroll
        for (int s = 0; s < KMAX; ++s) {
            const bool ahead = d2 < bd[s];                // equal: the slot's occupant stays
            minmax_f64(bd[s], d2);
            const int t = ahead ? q : bq[s];
            q = ahead ? bq[s] : q;
            bq[s] = t;
        }
    };
    const int maxR = max(gp.nc[0], max(gp.nc[1], gp.nc[2]));
    int R_first = 0, R_end = 0;                           // shells [R_first, R_end) were walked one by one
    bool done = false;
    if (use_block) {
        walk_block(keep);
        done = covered(1);
        R_first = 2;
    }
    R_end = R_first;
    for (int R = R_first; R <= maxR && !done; ++R) {
        walk_shell(R, keep);
        done = covered(R);
        R_end = R + 1;
    }
    if (nn_dist) {
        nn_dist[perm[p]] = sqrt(bd[0]);
        return;
    }
    // a walk that went past shell far_R of its cell: a spot shard's band (the rows of the cells within far_R cells of an own one)
    // then does not hold every row that can point at an own row - the sharded build falls back to exchanging the lists
    const bool went_far = R_end > far_R + 1;
    {
        const unsigned long long mf = __ballot(went_far && !listed);
        if (far_flag && mf != 0ULL && (threadIdx.x & 63) == (unsigned)(__ffsll((long long)mf) - 1)) atomicOr(far_flag, 1);
    }
    // a spot shard with band recompute lays out only the cells within reach of such walks (bin_points): what a longer walk
    // met there is not data.  Its row gets an empty list - the build is redone by exchange anyway (far_flag; for a band row
    // the rank that owns it raises it).
    if (far_drop && went_far) {
        for (int s = 0; s < kk; ++s) nbr_out[(size_t)p * kk + s] = -1;
        nbr_cnt[p] = 0;
        return;
    }

    // ---- the threshold: the kk-th smallest distance, how many list entries lie below it, and whether the (kk+1)-th equals it
    double thr = INFINITY, next = INFINITY;
#pragma unroll
    for (int s = 0; s < KMAX; ++s) {
        if (s == kk - 1) thr = bd[s];
        if (s == kk) next = bd[s];
    }
    // Spots whose kk-th and (kk+1)-th nearest are at EXACTLY the same distance: their neighbour set is not unique (cKDTree
    // keeps whichever its traversal meets first, graph.py:60-63; here the lower spot index wins).  The (kk+1)-th best of the
    // walked block is the true one whenever it ties with the kk-th: a point at that distance lies inside the radius the
    // walk was proven to cover.  Needs a spare slot (KMAX > kk: the launch takes care of it); without one every lane takes
    // the index-ordered route.
    const bool tie = (KMAX > kk) ? (next == thr && thr < INFINITY) : true;
    if (tie_count) {
        const unsigned long long m = __ballot(tie && !listed && !ghost);
        if (m != 0ULL && (threadIdx.x & 63) == (unsigned)(__ffsll((long long)m) - 1)) atomicAdd(tie_count, (int)__popcll(m));
    }
    if (__ballot(tie) != 0ULL) {
        // second walk (every lane of the wave; a lane without a tie finds its own list again): entries below the threshold stay,
        // the places at the threshold go to the lowest caller indices among ALL candidates at that distance
        int below = 0;
#pragma unroll
        for (int s = 0; s < KMAX; ++s) below += (s < kk && bd[s] < thr) ? 1 : 0;
        const int at_thr = kk - below;
        int tl[KMAX];
#pragma unroll
        for (int s = 0; s < KMAX; ++s) tl[s] = 0x7fffffff;
        auto ties = [&](int q, double d2) {
            int v = 0x7fffffff;
            if (d2 == thr) v = perm[q];
#pragma unroll
            for (int s = 0; s < KMAX; ++s) {
                const int t = min(tl[s], v);
                v = max(tl[s], v);
                tl[s] = t;
            }
        };
        if (use_block) walk_block(ties);
        for (int R = R_first; R < R_end; ++R) walk_shell(R, ties);
#pragma unroll
        for (int s = 0; s < KMAX; ++s)
            if (s >= below && s < kk) {                   // slot s takes the (s - below)-th lowest index at the threshold
                int o = 0x7fffffff;
#pragma unroll
                for (int t = 0; t < KMAX; ++t) if (t == s - below) o = tl[t];
                bq[s] = (s - below < at_thr && o != 0x7fffffff) ? rank[o] : -1;
            }
    }

    // ---- the neighbours.  Self is dropped (graph.py:70-74); if self is not among the kk nearest (coincident points) all kk
    // stay, as in the reference.  indeg != NULL (whole graph in one piece): the symmetrisation's first pass rides along -
    // every list entry counts itself into its target's in-degree, and the number it draws is its place in the target's
    // reverse list (all counters asked at once: one round trip)
    // Most targets are rows of the same block (128 consecutive rows of the Morton order: an ~11 x 11 patch): those count in LDS
    // and the block adds each row's sum to the global counter once - 6 returning atomics per row on L2 became ~2.5.
    int arr[KMAX];
    if (indeg) {
        __shared__ int s_loc[128], s_base[128];
        const long long blk0 = lo + blockIdx.x * (long long)blockDim.x;
        s_loc[tid] = 0;
        __syncthreads();
        constexpr int LOCAL = 0x40000000;
#pragma unroll
        for (int s = 0; s < KMAX; ++s) {
            arr[s] = 0;
            if (s < kk && bq[s] >= 0 && bq[s] != (int)p && !ghost) {
                const long long t = (long long)bq[s] - blk0;
                arr[s] = (t >= 0 && t < 128) ? (atomicAdd(&s_loc[t], 1) | LOCAL) : atomicAdd(&indeg[bq[s]], 1);
            }
        }
        __syncthreads();
        {
            const int c = s_loc[tid];
            s_base[tid] = c > 0 ? atomicAdd(&indeg[blk0 + tid], c) : 0;
        }
        __syncthreads();
#pragma unroll
        for (int s = 0; s < KMAX; ++s)
            if (arr[s] & LOCAL) arr[s] = s_base[bq[s] - blk0] + (arr[s] & ~LOCAL);
        if (ghost) return;
    }
    int cnt = 0;
#pragma unroll
    for (int s = 0; s < KMAX; ++s)
        if (s < kk && bq[s] >= 0 && bq[s] != (int)p) {
            if (indeg) arrival[(size_t)p * kk + cnt] = arr[s];
            nbr_out[(size_t)p * kk + cnt++] = bq[s];
        }
    for (int s = cnt; s < kk; ++s) nbr_out[(size_t)p * kk + s] = -1;
    nbr_cnt[p] = cnt;
}

// ------------------------------------------------------------------------------------------------ more than three dimensions
// utils/graph.py:16-22 takes coordinates of any dimension (cKDTree does).  The grid of this file bins three axes; points with 4 to
// FDX_KNN_MAX_DIM coordinates are put in solver order by their FIRST THREE coordinates (locality of the sweep's tiles only - any
// order gives the same graph) and searched exhaustively: lane = row, candidates in ascending CALLER index staged through LDS 256 at
// a time, squared distances summed coordinate by coordinate without contraction, a candidate enters the list on strictly smaller
// distance - i.e. the (distance, index) rule of knn_kernel.  O(n^2 dim): for the tens of thousands of spots such data has.
constexpr int FDX_KNN_MAX_DIM = 8;
__global__ __launch_bounds__(256) void take3_kernel(const double* __restrict__ coords, long long n, int dim, double* __restrict__ out) {
    const long long i = blockIdx.x * 256LL + threadIdx.x;
    if (i >= n) return;
    for (int a = 0; a < 3; ++a) out[(size_t)i * 3 + a] = coords[(size_t)i * dim + a];
}

template <int KMAX>
__global__ __launch_bounds__(256) void knn_brute_kernel(const double* __restrict__ coords, const int* __restrict__ perm,
                                                        const int* __restrict__ rank, long long n, int dim, int kk,
                                                        int* __restrict__ nbr_out, int* __restrict__ nbr_cnt, long long lo, long long hi,
                                                        int* __restrict__ tie_count) {
#pragma clang fp contract(off)
    __shared__ double tile[256 * FDX_KNN_MAX_DIM];
    const long long p = lo + blockIdx.x * 256LL + threadIdx.x;
    const bool live = p < hi;
    const int op = live ? perm[p] : 0;
    double x[FDX_KNN_MAX_DIM];
    for (int a = 0; a < FDX_KNN_MAX_DIM; ++a) x[a] = (a < dim) ? coords[(size_t)op * dim + a] : 0.0;
    double bd[KMAX];
    int bq[KMAX];
#pragma unroll
    for (int s = 0; s < KMAX; ++s) { bd[s] = INFINITY; bq[s] = -1; }
    for (long long o0 = 0; o0 < n; o0 += 256) {
        const int cnt = (int)min(256LL, n - o0);
        __syncthreads();
        for (int t = threadIdx.x; t < cnt * dim; t += 256) tile[t] = coords[(size_t)o0 * dim + t];
        __syncthreads();
        if (!live) continue;
        for (int c = 0; c < cnt; ++c) {
            double d2 = 0.0;
            for (int a = 0; a < dim; ++a) { const double dx = tile[c * dim + a] - x[a]; d2 = d2 + dx * dx; }
            int q = (int)(o0 + c);                       // caller index; turned into a position when the list is written
#pragma unroll
            for (int s = 0; s < KMAX; ++s) {
                const bool ahead = d2 < bd[s];            // equal: the occupant (lower caller index) stays
                const double td = ahead ? bd[s] : d2;
                const int tq = ahead ? bq[s] : q;
                bd[s] = ahead ? d2 : bd[s];
                bq[s] = ahead ? q : bq[s];
                d2 = td;
                q = tq;
            }
        }
    }
    if (!live) return;
    double thr = INFINITY, next = INFINITY;
#pragma unroll
    for (int s = 0; s < KMAX; ++s) {
        if (s == kk - 1) thr = bd[s];
        if (s == kk) next = bd[s];
    }
    if (tie_count && KMAX > kk && next == thr && thr < INFINITY) atomicAdd(tie_count, 1);
    int cnt = 0;
#pragma unroll
    for (int s = 0; s < KMAX; ++s)
        if (s < kk && bq[s] >= 0 && bq[s] != op) nbr_out[(size_t)p * kk + cnt++] = rank[bq[s]];
    for (int s = cnt; s < kk; ++s) nbr_out[(size_t)p * kk + s] = -1;
    nbr_cnt[p] = cnt;
}

// ------------------------------------------------------------------------------------------------ band of a spot shard
// A shard owns rows [lo, hi) of the sorted order.  Row p of the symmetrised k-NN graph is out(p) U in(p): in(p) needs the list of
// every row q that points at p.  When q's walk stayed within BAND_R shells of its own cell (knn_kernel reports the rows for which
// it did not), q can only point at rows of cells at most BAND_R cells away - so the rows that can point at an OWN row all live in
// cells within BAND_R cells of a cell that holds an own row: the BAND.  The shard finds the lists of its band itself ("recompute,
// don't communicate") instead of receiving the lists of all n rows.  BAND_R = 2: at ~4 points per cell the 3 x 3 block serves the
// k <= 8 nearest of MOST points (the flat walk of knn_kernel), but on uniform random points 1-3 % of the walks need shell 2 (the
// 7-th nearest lies beyond the distance to the block's edge); shell 3 would need fewer than 7 points in a disc of 12 cells.
// counters: [0] cells listed, [1] band rows listed, [2] a walk left the block (knn_kernel), [3] the band list overflowed
constexpr int BAND_R = 2;
__global__ __launch_bounds__(256) void band_cells_kernel(const double* __restrict__ sc, long long n, GridParams gp, long long lo,
                                                         long long hi, int* __restrict__ cell_flag, int* __restrict__ cell_list,
                                                         int* __restrict__ counters) {
    const long long p = lo + blockIdx.x * 256LL + threadIdx.x;
    if (p >= hi) return;
    int c[3];
    for (int a = 0; a < 3; ++a) c[a] = (a < gp.dim) ? cell_coord(sc[(size_t)a * n + p], gp.mn[a], gp.inv_h[a], gp.nc[a]) : 0;
    const int r1 = gp.dim > 1 ? BAND_R : 0, r2 = gp.dim > 2 ? BAND_R : 0;
    for (int dz = -r2; dz <= r2; ++dz)
        for (int dy = -r1; dy <= r1; ++dy)
            for (int dx = -BAND_R; dx <= BAND_R; ++dx) {
                const int x = c[0] + dx, y = c[1] + dy, z = c[2] + dz;
                if (x < 0 || x >= gp.nc[0] || y < 0 || y >= gp.nc[1] || z < 0 || z >= gp.nc[2]) continue;
                const int cell = x * gp.stride[0] + y * gp.stride[1] + z * gp.stride[2];
                if (cell_flag[cell] == 0 && atomicExch(&cell_flag[cell], 1) == 0) cell_list[atomicAdd(&counters[0], 1)] = cell;
            }
}

__global__ __launch_bounds__(256) void band_rows_kernel(const int* __restrict__ cell_list, const int* __restrict__ cstart,
                                                        const int* __restrict__ cend, long long lo, long long hi, int cap,
                                                        int* __restrict__ band, int* __restrict__ counters) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= counters[0]) return;
    const int cell = cell_list[i];
    for (int q = cstart[cell]; q < cend[cell]; ++q) {
        if (q >= lo && q < hi) continue;
        const int at = atomicAdd(&counters[1], 1);
        if (at < cap) band[at] = q;
        else counters[3] = 1;
    }
}

// The same band from the need flags of the shard's binning (cell_need_kernel<0>: the keys within BAND_R cells of a key that holds
// an own row - exactly the band's cells): one thread per key, the rows of a flagged key outside [lo, hi) appended with one atomic
// per wave.  (band_cells_kernel + band_rows_kernel: 25 flag reads per OWN ROW and an atomic per band row - 55 + 17 us for a
// 125k-row shard whose whole k-NN search is 60.)
__global__ __launch_bounds__(256) void band_rows_need_kernel(const int* __restrict__ start, long long bins,
                                                             const unsigned char* __restrict__ need1, long long lo, long long hi,
                                                             int cap, int* __restrict__ band, int* __restrict__ counters) {
    const long long k = blockIdx.x * 256LL + threadIdx.x;
    const int lane = threadIdx.x & 63;
    int s0 = 0, s1 = 0;
    if (k < bins && need1[k]) { s0 = start[k]; s1 = start[k + 1]; }
    // rows of the key are positions [s0, s1); those inside [lo, hi) are own rows
    const int a0 = (int)min((long long)s1, max((long long)s0, lo)), a1 = (int)max((long long)a0, min((long long)s1, hi));   // own part [a0, a1)
    const int cnt = (s1 - s0) - (a1 - a0);
    int incl = cnt;
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
        const int v = __shfl_up(incl, off, 64);
        if (lane >= off) incl += v;
    }
    const int total = __shfl(incl, 63, 64);
    if (total == 0) return;
    int base = 0;
    if (lane == 63) base = atomicAdd(&counters[1], total);
    base = __shfl(base, 63, 64);
    int at = base + incl - cnt;
    for (int q = s0; q < s1; ++q) {
        if (q >= a0 && q < a1) continue;
        if (at < cap) band[at] = q;
        else counters[3] = 1;
        ++at;
    }
}

// indegree / reverse lists of a shard over the rows that HAVE lists: [lo, hi) and the band (every other row is skipped without a
// look at its count)
// n_direct >= 0: threads 0 .. n_direct - 1 take the own rows lo + i, the following ones the listed rows (one launch for both)
__global__ __launch_bounds__(256) void indegree_rows_kernel(const int* __restrict__ nbr, const int* __restrict__ nbr_cnt, int kk,
                                                            int* __restrict__ indeg, int lo, int hi, const int* __restrict__ rows,
                                                            const int* __restrict__ n_rows, int cap, int n_direct = -1) {
    long long i = blockIdx.x * 256LL + threadIdx.x;
    long long p;
    if (n_direct >= 0 && i < n_direct) p = lo + i;
    else if (rows) { if (n_direct >= 0) i -= n_direct; if (i >= cap || i >= *n_rows) return; p = rows[i]; }
    else { p = lo + i; if (p >= hi) return; }
    for (int m = 0; m < nbr_cnt[p]; ++m) {
        const int q = nbr[(size_t)p * kk + m];
        if (q >= lo && q < hi) atomicAdd(&indeg[q], 1);
    }
}

__global__ __launch_bounds__(256) void fill_reverse_rows_kernel(const int* __restrict__ nbr, const int* __restrict__ nbr_cnt, int kk,
                                                                const int* __restrict__ rev_off, int* __restrict__ cursor,
                                                                int* __restrict__ rev, int lo, int hi, const int* __restrict__ rows,
                                                                const int* __restrict__ n_rows, int cap, int n_direct = -1) {
    long long i = blockIdx.x * 256LL + threadIdx.x;
    long long p;
    if (n_direct >= 0 && i < n_direct) p = lo + i;
    else if (rows) { if (n_direct >= 0) i -= n_direct; if (i >= cap || i >= *n_rows) return; p = rows[i]; }
    else { p = lo + i; if (p >= hi) return; }
    for (int m = 0; m < nbr_cnt[p]; ++m) {
        const int q = nbr[(size_t)p * kk + m];
        if (q >= lo && q < hi) rev[rev_off[q] + atomicAdd(&cursor[q], 1)] = (int)p;
    }
}

// ------------------------------------------------------------------------------------------------ radius graph
// PASS 0 counts, PASS 1 fills nbr[off[p] + m] (sorted-space indices, unsorted order)
template <int PASS>
__global__ __launch_bounds__(128) void radius_kernel(const double* __restrict__ sc, const int* __restrict__ cstart,
                                                     const int* __restrict__ cend, long long n, GridParams gp,
                                                     double radius, int R, int* __restrict__ cnt,
                                                     const int* __restrict__ off, int* __restrict__ nbr, long long lo,
                                                     long long hi) {
    const long long p = blockIdx.x * (long long)blockDim.x + threadIdx.x;
    if (p >= n) return;
    if (p < lo || p >= hi) {                                               // a shard builds its own rows; the others stay empty
        if (!PASS) cnt[p] = 0;
        return;
    }
    const double px = sc[p], py = sc[(size_t)n + p], pz = sc[2 * (size_t)n + p];
    const double pc[3] = {px, py, pz};
    int c[3];
    for (int a = 0; a < 3; ++a) c[a] = (a < gp.dim) ? cell_coord(pc[a], gp.mn[a], gp.inv_h[a], gp.nc[a]) : 0;
    int m = 0;
    const int base = PASS ? off[p] : 0;
    for (int z = max(0, c[2] - R); z <= min(gp.nc[2] - 1, c[2] + R); ++z)
        for (int y = max(0, c[1] - R); y <= min(gp.nc[1] - 1, c[1] + R); ++y)
            for (int x = max(0, c[0] - R); x <= min(gp.nc[0] - 1, c[0] + R); ++x) {
                const int cell = x * gp.stride[0] + y * gp.stride[1] + z * gp.stride[2];
                for (int q = cstart[cell]; q < cend[cell]; ++q) {
                    if (q == (int)p) continue;
                    const double d2 = dist2_exact(sc[q] - px, sc[(size_t)n + q] - py, sc[2 * (size_t)n + q] - pz);
                    if (sqrt(d2) <= radius) {   // query_pairs(r): distance <= r   (graph.py:115)
                        if (PASS) nbr[base + m] = q;
                        ++m;
                    }
                }
            }
    if (!PASS) cnt[p] = m;
}

// ------------------------------------------------------------------------------------------------ symmetrise
__global__ __launch_bounds__(256) void indegree_kernel(const int* __restrict__ nbr, const int* __restrict__ nbr_cnt,
                                                       long long n, int kk, int* __restrict__ indeg, int lo, int hi) {
    const long long p = blockIdx.x * 256LL + threadIdx.x;
    if (p >= n) return;
    for (int m = 0; m < nbr_cnt[p]; ++m) {           // only edges INTO rows [lo, hi) (a shard builds its own rows)
        const int q = nbr[(size_t)p * kk + m];
        if (q >= lo && q < hi) atomicAdd(&indeg[q], 1);
    }
}

__global__ __launch_bounds__(256) void fill_reverse_kernel(const int* __restrict__ nbr, const int* __restrict__ nbr_cnt,
                                                           long long n, int kk, const int* __restrict__ rev_off,
                                                           int* __restrict__ cursor, int* __restrict__ rev, int lo, int hi) {
    const long long p = blockIdx.x * 256LL + threadIdx.x;
    if (p >= n) return;
    for (int m = 0; m < nbr_cnt[p]; ++m) {
        const int q = nbr[(size_t)p * kk + m];
        if (q >= lo && q < hi) rev[rev_off[q] + atomicAdd(&cursor[q], 1)] = (int)p;
    }
}

// the same with the places drawn by the k-NN kernel: plain stores
__global__ __launch_bounds__(256) void fill_reverse_placed_kernel(const int* __restrict__ nbr, const int* __restrict__ nbr_cnt,
                                                                  const int* __restrict__ arrival, long long n, int kk,
                                                                  const int* __restrict__ rev_off, int* __restrict__ rev) {
    const long long p = blockIdx.x * 256LL + threadIdx.x;
    if (p >= n) return;
    for (int m = 0; m < nbr_cnt[p]; ++m) rev[rev_off[nbr[(size_t)p * kk + m]] + arrival[(size_t)p * kk + m]] = (int)p;
}

// Row p: candidates = out(p) U in(p) -> sorted by ORIGINAL index, duplicates removed, stored at ws[seg_off(p) ...].
// seg_off(p) = p*kk + rev_off[p] (capacity kk + indeg[p]).  The arrival order of the reverse list is arbitrary (atomics);
// sorting makes the result deterministic.
__global__ __launch_bounds__(128) void merge_rows_kernel(const int* __restrict__ nbr, const int* __restrict__ nbr_cnt,
                                                         const int* __restrict__ rev, const int* __restrict__ rev_off,
                                                         const int* __restrict__ perm, const int* __restrict__ rank,
                                                         long long lo, long long hi, int kk,
                                                         int* __restrict__ ws, int* __restrict__ deg) {
    const long long p = lo + blockIdx.x * (long long)blockDim.x + threadIdx.x;
    if (p >= hi) return;
    int* seg = ws + (size_t)p * kk + rev_off[p];
    const int n_out = nbr_cnt[p], n_in = rev_off[p + 1] - rev_off[p];
    const int m = n_out + n_in;
    constexpr int CAP = 32;
    __shared__ int keys[CAP * 128];                  // keys[a * 128 + tid]: a thread's slots are 128 apart -> conflict-free
    if (m <= CAP) {
        // common case: sort the ORIGINAL indices (a bijection of the positions: rank[] leads back) in LDS and write the row
        // ONCE.  (An in-place insertion sort in global memory costs a 64-byte write per 4-byte move - measured 2 GB written
        // for 100 MB - and a private key[] array is dynamically indexed, i.e. scratch memory: 3.5 ms at 8M spots.  32-bit
        // keys: 16 KB per workgroup, twice the resident workgroups of the (original index, position) pairs used before.)
        int* key = keys + threadIdx.x;
        for (int t = 0; t < n_out; ++t) key[t * 128] = perm[nbr[(size_t)p * kk + t]];
        for (int t = 0; t < n_in; ++t) key[(n_out + t) * 128] = perm[rev[rev_off[p] + t]];
        for (int a = 1; a < m; ++a) {
            const int kv = key[a * 128];
            int b = a - 1;
            while (b >= 0 && key[b * 128] > kv) { key[(b + 1) * 128] = key[b * 128]; --b; }
            key[(b + 1) * 128] = kv;
        }
        int u = 0;
        int prev = -1;
        for (int a = 0; a < m; ++a) {
            const int kv = key[a * 128];
            if (a == 0 || kv != prev) seg[u++] = rank[kv];
            prev = kv;
        }
        deg[p] = u;
        return;
    }
    int mm = 0;
    for (int t = 0; t < n_out; ++t) seg[mm++] = nbr[(size_t)p * kk + t];
    for (int t = rev_off[p]; t < rev_off[p + 1]; ++t) seg[mm++] = rev[t];
    for (int a = 1; a < mm; ++a) {   // insertion sort by original index
        const int v = seg[a];
        const int kv = perm[v];
        int b = a - 1;
        while (b >= 0 && perm[seg[b]] > kv) { seg[b + 1] = seg[b]; --b; }
        seg[b + 1] = v;
    }
    int u = 0;
    for (int a = 0; a < mm; ++a)
        if (a == 0 || seg[a] != seg[a - 1]) seg[u++] = seg[a];
    deg[p] = u;
}

// Variant for already-symmetric neighbour lists with explicit offsets (radius graph): sort only.
__global__ __launch_bounds__(128) void sort_rows_kernel(int* __restrict__ nbr, const int* __restrict__ off,
                                                        const int* __restrict__ perm, long long n) {
    const long long p = blockIdx.x * (long long)blockDim.x + threadIdx.x;
    if (p >= n) return;
    int* seg = nbr + off[p];
    const int m = off[p + 1] - off[p];
    for (int a = 1; a < m; ++a) {
        const int v = seg[a];
        const int kv = perm[v];
        int b = a - 1;
        while (b >= 0 && perm[seg[b]] > kv) { seg[b + 1] = seg[b]; --b; }
        seg[b + 1] = v;
    }
}

// ------------------------------------------------------------------------------------------------ ELL
// width[s] = widest row of slice s; per block the sum of its degrees and its widest slice (part[2b], part[2b + 1]).
// (One atomic per slice on a single pair of counters was 370 us at 1M spots: 31k same-address atomics, ~12 ns each.)
constexpr int SLICE_WIDTH_BLOCKS = 1024;
// zero_tail: the closing entry of the scan's input (width[n_slices]) and the two summary words the tile kernel raises are cleared
// here instead of by a fill of their own in front of this launch
__global__ __launch_bounds__(256) void slice_width_kernel(const int* __restrict__ deg, long long n, int n_slices,
                                                          int* __restrict__ width, long long* __restrict__ part,
                                                          int* __restrict__ summary_zero = nullptr) {
    __shared__ long long s_sum[4];
    __shared__ int s_max[4];
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    if (summary_zero && blockIdx.x == 0 && threadIdx.x == 0) { width[n_slices] = 0; summary_zero[0] = 0; summary_zero[1] = 0; }
    long long tot = 0;
    int wmax = 0;
    for (int s = blockIdx.x * 4 + wv; s < n_slices; s += gridDim.x * 4) {
        const long long i = (long long)s * 64 + lane;
        const int d = (i < n) ? deg[i] : 0;
        int w = d, sum = d;
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) {
            w = max(w, __shfl_xor(w, off, 64));
            sum += __shfl_xor(sum, off, 64);
        }
        if (lane == 0) width[s] = w;
        tot += sum;
        wmax = max(wmax, w);
    }
    if (lane == 0) { s_sum[wv] = tot; s_max[wv] = wmax; }
    __syncthreads();
    if (threadIdx.x == 0 && part) {
        part[2 * blockIdx.x] = (s_sum[0] + s_sum[1]) + (s_sum[2] + s_sum[3]);
        part[2 * blockIdx.x + 1] = (long long)max(max(s_max[0], s_max[1]), max(s_max[2], s_max[3]));
    }
}

// red[0] = nnz, red[1] = widest slice, from the blocks' partials; with `meta` (a queued build): the numbers the host will ask
// for, written straight into its pinned block (one launch instead of a copy each)
__global__ __launch_bounds__(256) void graph_meta_kernel(const long long* __restrict__ part, int n_part, long long* __restrict__ red,
                                                         const int* __restrict__ slice_off, int n_slices,
                                                         const int* __restrict__ summary, const int* __restrict__ ties,
                                                         long long* __restrict__ meta) {
    __shared__ long long s_sum[256];
    __shared__ int s_max[256];
    long long tot = 0;
    int wmax = 0;
    for (int b = threadIdx.x; b < n_part; b += 256) { tot += part[2 * b]; wmax = max(wmax, (int)part[2 * b + 1]); }
    s_sum[threadIdx.x] = tot;
    s_max[threadIdx.x] = wmax;
    __syncthreads();
    for (int s = 128; s > 0; s >>= 1) {
        if ((int)threadIdx.x < s) {
            s_sum[threadIdx.x] += s_sum[threadIdx.x + s];
            s_max[threadIdx.x] = max(s_max[threadIdx.x], s_max[threadIdx.x + s]);
        }
        __syncthreads();
    }
    if (threadIdx.x != 0) return;
    red[0] = s_sum[0];
    red[1] = (long long)s_max[0];
    if (!meta) return;
    meta[0] = (long long)slice_off[n_slices];
    meta[1] = s_sum[0];
    meta[2] = (long long)s_max[0];
    meta[3] = (long long)(unsigned)summary[0] | ((long long)summary[1] << 32);
    meta[4] = ties ? (long long)ties[0] : 0;
}

// seg_off(p) = p*seg_stride + seg_extra[p]   (k-NN: seg_stride = kk, seg_extra = rev_off; radius: stride 0, extra = off)
__global__ __launch_bounds__(256) void fill_ell_kernel(const int* __restrict__ ws, int seg_stride,
                                                       const int* __restrict__ seg_extra, const int* __restrict__ deg,
                                                       const int* __restrict__ slice_off, long long n, int n_slices,
                                                       int pad, int* __restrict__ ell, long long cap_rows) {
    const int lane = threadIdx.x & 63;
    const int s = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (s >= n_slices) return;
    if ((long long)slice_off[n_slices] > cap_rows) return;      // deferred build with too small a bound: rebuilt later
    const long long i = (long long)s * 64 + lane;
    const int w0 = slice_off[s], w = slice_off[s + 1] - w0;
    const int dg = (i < n) ? deg[i] : 0;
    const int* seg = (i < n) ? ws + (size_t)i * seg_stride + seg_extra[i] : ws;
    for (int m = 0; m < w; ++m) ell[((size_t)w0 + m) * 64 + lane] = (m < dg) ? seg[m] : pad;
}

// export: CSR in the caller's labels.  deg_orig[perm[p]] = deg[p]; then indices[indptr[o] + m] = perm[seg_p[m]].
__global__ __launch_bounds__(256) void deg_to_orig_kernel(const int* __restrict__ deg, const int* __restrict__ perm,
                                                          long long n, int* __restrict__ deg_orig) {
    const long long p = blockIdx.x * 256LL + threadIdx.x;
    // rows without entries are skipped, not copied (deg_orig arrives zeroed): the full-size graph of a spot shard has its
    // permutation laid out only where the shard looks (bin_points, shard mode) - an empty row's perm[p] is not data there
    if (p < n && deg[p] > 0) deg_orig[perm[p]] = deg[p];
}

__global__ __launch_bounds__(256) void export_rows_kernel(const int* __restrict__ ws, int seg_stride,
                                                          const int* __restrict__ seg_extra, const int* __restrict__ deg,
                                                          const int* __restrict__ perm, const long long* __restrict__ indptr,
                                                          long long n, int* __restrict__ indices) {
    const long long p = blockIdx.x * 256LL + threadIdx.x;
    if (p >= n || deg[p] <= 0) return;              // empty rows: nothing to write, and (spot shards) no valid perm[p] to look up
    const int* seg = ws + (size_t)p * seg_stride + seg_extra[p];
    const long long base = indptr[perm[p]];
    for (int m = 0; m < deg[p]; ++m) indices[base + m] = perm[seg[m]];
}

__global__ __launch_bounds__(256) void iota_kernel(int* __restrict__ v, long long n) {
    const long long i = blockIdx.x * 256LL + threadIdx.x;
    if (i < n) v[i] = (int)i;
}

// ------------------------------------------------------------------------------------------------ sweep tiles
// A tile = 256 consecutive sorted spots = one workgroup of the tiled BCD sweep.  For every tile: the sorted, duplicate-
// free list of neighbour positions OUTSIDE the tile (its halo), and every ELL entry of its rows translated to a
// tile-local slot: 0..255 own spot, 256+h the h-th halo entry, 256+H the all-zero pad slot.
constexpr int TILE_HASH = 2048;
__global__ __launch_bounds__(256) void tile_halo_kernel(const int* __restrict__ ell, const int* __restrict__ deg,
                                                        const int* __restrict__ slice_off, long long n,
                                                        int* __restrict__ tile_halo, int* __restrict__ tile_hcnt,
                                                        unsigned short* __restrict__ ell_local, long long cap_rows,
                                                        int* __restrict__ summary /* [0] largest halo, [1] some tile failed */) {
    if ((long long)slice_off[(n + 63) >> 6] > cap_rows) return;
    __shared__ int tab[TILE_HASH];
    __shared__ int list[FDX_TILE_HALO_CAP];
    __shared__ int s_cnt, s_over;
    const int tid = threadIdx.x, tile = blockIdx.x;
    for (int s = tid; s < TILE_HASH; s += 256) tab[s] = -1;
    if (tid == 0) { s_cnt = 0; s_over = 0; }
    __syncthreads();
    const long long p = (long long)tile * 256 + tid;
    const int dg = (p < n) ? deg[p] : 0;
    // row p of the sliced ELL: entry m at ell[(slice_off[p/64] + m)*64 + p%64]
    const int* seg = (p < n) ? ell + (size_t)slice_off[p >> 6] * 64 + (p & 63) : ell;
    for (int m = 0; m < dg; ++m) {
        const int q = seg[(size_t)m * 64];
        if ((q >> 8) == tile && q < n) continue;          // a LOCAL graph's halo slots n..n_total-1 can carry the last tile's number: they are halo
        unsigned h = ((unsigned)q * 2654435761u) >> 21;   // 11 bits
        int probes = 0;
        while (true) {
            const int old = atomicCAS(&tab[h], -1, q);
            if (old == -1 || old == q) break;
            h = (h + 1) & (TILE_HASH - 1);
            if (++probes > TILE_HASH) { s_over = 1; break; }
        }
    }
    __syncthreads();
    for (int s = tid; s < TILE_HASH; s += 256)
        if (tab[s] != -1) {
            const int pos = atomicAdd(&s_cnt, 1);
            if (pos < FDX_TILE_HALO_CAP) list[pos] = tab[s];
        }
    __syncthreads();
    const int H = s_cnt;
    if (H > FDX_TILE_HALO_CAP || s_over) {      // irregular graph: this tile cannot use the LDS path
        if (tid == 0) { tile_hcnt[tile] = -1; if (summary) atomicOr(summary + 1, 1); }
        return;
    }
    if (tid == 0 && summary && H > __builtin_nontemporal_load(summary)) atomicMax(summary, H);   // a glance first: one address for 4000 tiles
    int P = 1;
    while (P < H) P <<= 1;
    for (int s = H + tid; s < P; s += 256) list[s] = 0x7fffffff;
    __syncthreads();
    for (int k = 2; k <= P; k <<= 1)
        for (int j = k >> 1; j > 0; j >>= 1) {
            for (int idx = tid; idx < P; idx += 256) {
                const int ixj = idx ^ j;
                if (ixj > idx) {
                    const int a = list[idx], b = list[ixj];
                    const bool up = ((idx & k) == 0);
                    if ((a > b) == up) { list[idx] = b; list[ixj] = a; }
                }
            }
            __syncthreads();
        }
    for (int h = tid; h < H; h += 256) tile_halo[(size_t)tile * FDX_TILE_HALO_CAP + h] = list[h];
    if (tid == 0) tile_hcnt[tile] = H;
    if (p < n) {
        const int s = (int)(p >> 6), lane = (int)(p & 63);
        const int w0 = slice_off[s], w = slice_off[s + 1] - w0;
        for (int m = 0; m < w; ++m) {
            int slot = 256 + H;   // pad -> zero slot
            if (m < dg) {
                const int q = seg[(size_t)m * 64];
                if ((q >> 8) == tile && q < n) slot = q & 255;
                else {
                    int lo = 0, hi = H;   // lower_bound in the sorted halo list
                    while (lo < hi) { const int mid = (lo + hi) >> 1; if (list[mid] < q) lo = mid + 1; else hi = mid; }
                    slot = 256 + lo;
                }
            }
            ell_local[((size_t)w0 + m) * 64 + lane] = (unsigned short)slot;
        }
    }
}

// fill_ell_kernel and tile_halo_kernel in one pass over the rows (deferred whole-graph build): a row is read once, from its
// segment, and goes to the global-index ELL and - through the tile's halo list - to the tile-local slots
__global__ __launch_bounds__(256) void tile_ell_kernel(const int* __restrict__ ws, int seg_stride, const int* __restrict__ seg_extra,
                                                       int pad, int* __restrict__ ell, const int* __restrict__ deg,
                                                        const int* __restrict__ slice_off, long long n,
                                                        int* __restrict__ tile_halo, int* __restrict__ tile_hcnt,
                                                        unsigned short* __restrict__ ell_local, long long cap_rows,
                                                        int* __restrict__ summary /* [0] largest halo, [1] some tile failed */) {
    if ((long long)slice_off[(n + 63) >> 6] > cap_rows) return;
    __shared__ int tab[TILE_HASH];
    __shared__ int list[FDX_TILE_HALO_CAP];
    __shared__ int s_cnt, s_over;
    const int tid = threadIdx.x, tile = blockIdx.x;
    for (int s = tid; s < TILE_HASH; s += 256) tab[s] = -1;
    if (tid == 0) { s_cnt = 0; s_over = 0; }
    __syncthreads();
    const long long p = (long long)tile * 256 + tid;
    const int dg = (p < n) ? deg[p] : 0;
    // row p of the sliced ELL: entry m at ell[(slice_off[p/64] + m)*64 + p%64]
    const int* seg = (p < n) ? ws + (size_t)p * seg_stride + seg_extra[p] : ws;
    for (int m = 0; m < dg; ++m) {
        const int q = seg[m];
        if ((q >> 8) == tile && q < n) continue;          // a LOCAL graph's halo slots n..n_total-1 can carry the last tile's number: they are halo
        unsigned h = ((unsigned)q * 2654435761u) >> 21;   // 11 bits
        int probes = 0;
        while (true) {
            const int old = atomicCAS(&tab[h], -1, q);
            if (old == -1 || old == q) break;
            h = (h + 1) & (TILE_HASH - 1);
            if (++probes > TILE_HASH) { s_over = 1; break; }
        }
    }
    __syncthreads();
    for (int s = tid; s < TILE_HASH; s += 256)
        if (tab[s] != -1) {
            const int pos = atomicAdd(&s_cnt, 1);
            if (pos < FDX_TILE_HALO_CAP) list[pos] = tab[s];
        }
    __syncthreads();
    const int H = s_cnt;
    if (H > FDX_TILE_HALO_CAP || s_over) {      // irregular graph: this tile cannot use the LDS path - the global-index ELL is still written
        if (tid == 0) { tile_hcnt[tile] = -1; if (summary) atomicOr(summary + 1, 1); }
        if ((p >> 6) < ((n + 63) >> 6)) {           // every lane of the last slice: lanes past n carry the pad index
            const int s = (int)(p >> 6), lane = (int)(p & 63);
            const int w0 = slice_off[s], w = slice_off[s + 1] - w0;
            for (int m = 0; m < w; ++m) ell[((size_t)w0 + m) * 64 + lane] = (m < dg) ? seg[m] : pad;
        }
        return;
    }
    if (tid == 0 && summary && H > __builtin_nontemporal_load(summary)) atomicMax(summary, H);   // a glance first: one address for 4000 tiles
    int P = 1;
    while (P < H) P <<= 1;
    for (int s = H + tid; s < P; s += 256) list[s] = 0x7fffffff;
    __syncthreads();
    for (int k = 2; k <= P; k <<= 1)
        for (int j = k >> 1; j > 0; j >>= 1) {
            for (int idx = tid; idx < P; idx += 256) {
                const int ixj = idx ^ j;
                if (ixj > idx) {
                    const int a = list[idx], b = list[ixj];
                    const bool up = ((idx & k) == 0);
                    if ((a > b) == up) { list[idx] = b; list[ixj] = a; }
                }
            }
            __syncthreads();
        }
    for (int h = tid; h < H; h += 256) tile_halo[(size_t)tile * FDX_TILE_HALO_CAP + h] = list[h];
    if (tid == 0) tile_hcnt[tile] = H;
    if ((p >> 6) < ((n + 63) >> 6)) {               // every lane of the last slice: lanes past n carry the pad index / the zero slot
        const int s = (int)(p >> 6), lane = (int)(p & 63);
        const int w0 = slice_off[s], w = slice_off[s + 1] - w0;
        for (int m = 0; m < w; ++m) {
            int slot = 256 + H;   // pad -> zero slot
            if (m >= dg) ell[((size_t)w0 + m) * 64 + lane] = pad;
            if (m < dg) {
                const int q = seg[m];
                ell[((size_t)w0 + m) * 64 + lane] = q;
                if ((q >> 8) == tile && q < n) slot = q & 255;
                else {
                    int lo = 0, hi = H;   // lower_bound in the sorted halo list
                    while (lo < hi) { const int mid = (lo + hi) >> 1; if (list[mid] < q) lo = mid + 1; else hi = mid; }
                    slot = 256 + lo;
                }
            }
            ell_local[((size_t)w0 + m) * 64 + lane] = (unsigned short)slot;
        }
    }
}

// ------------------------------------------------------------------------------------------------ host side
// FDX_TRACE_HOST=1: host clock at the steps of a graph build (stderr), to see which calls the host spends its time in
static void trace_host(const char* what) {
    static const bool on = fdx::env("FDX_TRACE_HOST") != nullptr;
    if (!on) return;
    static auto t_prev = std::chrono::steady_clock::now();
    const auto t = std::chrono::steady_clock::now();
    std::fprintf(stderr, "[fdx-host] +%7.1f us  %s\n", std::chrono::duration<double, std::micro>(t - t_prev).count(), what);
    t_prev = t;
}

static int exclusive_scan_int(const int* in, int* out, long long count, hipStream_t st, DevBuf& tmp) {
    size_t bytes = 0;
    FDX_HIP(rocprim::exclusive_scan(nullptr, bytes, in, out, 0, (size_t)count, rocprim::plus<int>(), st));
    if (tmp.bytes < bytes) FDX_TRY(tmp.alloc(bytes));
    FDX_HIP(rocprim::exclusive_scan(tmp.p, bytes, in, out, 0, (size_t)count, rocprim::plus<int>(), st));
    return 0;
}

static int exclusive_scan_i64(const int* in, long long* out, long long count, hipStream_t st, DevBuf& tmp) {
    size_t bytes = 0;
    auto in64 = rocprim::make_transform_iterator(in, [] __device__(int v) { return (long long)v; });
    FDX_HIP(rocprim::exclusive_scan(nullptr, bytes, in64, out, 0LL, (size_t)count, rocprim::plus<long long>(), st));
    if (tmp.bytes < bytes) FDX_TRY(tmp.alloc(bytes));
    FDX_HIP(rocprim::exclusive_scan(tmp.p, bytes, in64, out, 0LL, (size_t)count, rocprim::plus<long long>(), st));
    return 0;
}

// The bounding box is the one thing the host must read before it can queue the rest of a build (grid parameters size every
// launch).  bbox_begin queues it - block partials, one workgroup folds them and writes the six numbers into a pinned block -
// and make_grid waits for its event.  (Queueing it ahead of the fit's other host work - a hint entry the fit called before it
// set up the leverage job - was tried: the box arrives earlier, the leverage scores later, the wall time is the same.)
struct BboxJob {
    const double* coords = nullptr;
    long long n = 0;
    int dim = 0, dev = 0;
    hipEvent_t ev = nullptr;
    double* host = nullptr;      // pinned_block_get(): mn[3], mx[3]
    DevBuf part;
    ~BboxJob() {
        if (ev) { (void)hipEventSynchronize(ev); (void)hipEventDestroy(ev); }
        if (host) pinned_block_put(host);
    }
};

static int bbox_begin(const double* d_coords, long long n, int dim, hipStream_t st, BboxJob* job) {
    job->coords = d_coords; job->n = n; job->dim = dim;
    FDX_HIP(hipGetDevice(&job->dev));
    // (256 to 16384 blocks, one to sixteen points per thread, four loads in flight or one: 33-60 us for the 16 MB of a million 2-D
    // points whatever the shape - the kernel's time is not its loop; 512 blocks measured best)
    const int nblk = (int)std::min<long long>(fdx::exp_env("FDX_BBOX_BLOCKS") ? atoi(fdx::exp_env("FDX_BBOX_BLOCKS")) : 512, (n + 255) / 256);
    FDX_TRY(job->part.alloc((size_t)nblk * 6 * sizeof(double)));
    job->host = (double*)pinned_block_get();
    FDX_REQUIRE(job->host != nullptr, "graph: pinned host block");
    void* host_dev = nullptr;
    FDX_HIP(hipHostGetDevicePointer(&host_dev, job->host, 0));
    FDX_HIP(hipEventCreateWithFlags(&job->ev, hipEventDisableTiming));
    hipLaunchKernelGGL(bbox_partial_kernel, dim3(nblk), dim3(256), 0, st, d_coords, n, dim, job->part.as<double>());
    hipLaunchKernelGGL(bbox_final_kernel, dim3(1), dim3(256), 0, st, job->part.as<double>(), nblk, (double*)host_dev);
    FDX_CHECK_LAUNCH();
    FDX_HIP(hipEventRecord(job->ev, st));
    return 0;
}

// under_wait: queued behind the bounding-box kernels and ahead of the host's wait for them - fills whose sizes depend on n alone
// run on the device while the host takes the six numbers over (they used to sit in the chain of dependent launches after it)
static int make_grid(const double* d_coords, long long n, int dim, double target_per_cell, double min_h,
                     GridParams* gp, hipStream_t st, const std::function<int()>* under_wait = nullptr) {
    auto job = std::make_unique<BboxJob>();
    FDX_TRY(bbox_begin(d_coords, n, dim, st, job.get()));
    if (under_wait && *under_wait) FDX_TRY((*under_wait)());
    FDX_HIP(hipEventSynchronize(job->ev));
    double mn[3] = {0, 0, 0}, mx[3] = {0, 0, 0};
    for (int a = 0; a < dim; ++a) {
        mn[a] = job->host[a];
        mx[a] = job->host[3 + a];
        if (!(std::isfinite(mn[a]) && std::isfinite(mx[a])))
            return fail(FDX_ERR_INVALID, "graph: coordinates contain NaN or infinity");
    }
    job.reset();
    // cell edge from the occupied volume: ~target_per_cell points per cell over the axes with non-zero extent
    double vol = 1.0;
    int eff = 0;
    for (int a = 0; a < dim; ++a)
        if (mx[a] > mn[a]) { vol *= (mx[a] - mn[a]); ++eff; }
    double h = 1.0;
    if (eff > 0) h = std::pow(vol * target_per_cell / (double)std::max<long long>(n, 1), 1.0 / eff);
    if (min_h > 0.0) h = std::max(h, min_h);
    if (!(h > 0.0) || !std::isfinite(h)) h = 1.0;
    for (int attempt = 0; attempt < 64; ++attempt) {
        double cells = 1.0;
        for (int a = 0; a < 3; ++a) {
            gp->mn[a] = (a < dim) ? mn[a] : 0.0;
            gp->h[a] = h;
            gp->inv_h[a] = 1.0 / h;
            double nca = (a < dim && mx[a] > mn[a]) ? std::floor((mx[a] - mn[a]) / h) + 1.0 : 1.0;
            cells *= nca;
            gp->nc[a] = (int)std::min(nca, 2.0e9);
        }
        const int max_axis = std::max(gp->nc[0], std::max(gp->nc[1], gp->nc[2]));
        const bool morton_ok = (dim < 3) || max_axis < (1 << 21);   // 21 bits per axis in the 3-D Morton key
        if (cells <= std::max(4.0 * (double)n, 4096.0) && cells < 2.0e9 && morton_ok) break;
        h *= 1.5;   // very elongated / clustered inputs: coarsen until the table is O(N)
    }
    gp->dim = dim;
    // slowest-varying axis = most cells
    int order[3] = {0, 1, 2};
    std::sort(order, order + 3, [&](int a, int b) { return gp->nc[a] < gp->nc[b]; });
    int stride = 1;
    for (int t = 0; t < 3; ++t) { gp->stride[order[t]] = stride; stride *= gp->nc[order[t]]; }
    return 0;
}

struct BinnedPoints {
    GridParams gp;
    DevBuf perm, rank, sc, sc2, cstart;            // sc2: (x, y) pairs of the sorted points, dim <= 2 only; cstart also holds cend, the key counters and the need flags
    int* cend_p = nullptr;                         // cell -> end of its range (inside cstart's block)
    int* count_p = nullptr;                        // counting path: members per key
    unsigned char* need_p = nullptr;               // spot shards: [bins] keys within BAND_R cells of an own key, [bins] within 2 BAND_R
    DevBuf keys, vals, skeys, sort_tmp, start, scan_tmp;   // sort temporaries: kept until the struct dies so that binning needs no final sync
    long long n = 0;
    int n_cells = 0;
    long long bins = 0;            // counting path: size of the Morton key space (start has bins + 1 entries); 0 on the sorting path
};

// shard_lo < shard_hi: a spot shard's binning - the ranking pass lays out only the cells the shard's build looks at
// (cell_need_kernel; counting path only: the sorting path lays out everything)
static int bin_points(const double* d_coords, long long n, int dim, double target_per_cell, double min_h,
                      BinnedPoints* b, hipStream_t st, long long shard_lo = 0, long long shard_hi = 0, int shard_R = 0,
                      const std::function<int()>* extra_under_wait = nullptr) {
    b->n = n;
    FDX_TRY(b->sc.alloc((size_t)n * 3 * sizeof(double)));
    const std::function<int()> under_wait = [&]() -> int {
        // the coordinate planes past `dim` read as zero (8 MB at a million 2-D points: 16 us that used to sit between place and rank)
        if (dim < 3) FDX_HIP(hipMemsetAsync(b->sc.as<double>() + (size_t)dim * n, 0, (size_t)(3 - dim) * n * sizeof(double), st));
        if (extra_under_wait && *extra_under_wait) FDX_TRY((*extra_under_wait)());
        return 0;
    };
    FDX_TRY(make_grid(d_coords, n, dim, target_per_cell, min_h, &b->gp, st, &under_wait));
    trace_host("bin: make_grid (bbox kernel + read-back)");
    b->n_cells = b->gp.nc[0] * b->gp.nc[1] * b->gp.nc[2];
    DevBuf& keys = b->keys;
    DevBuf& vals = b->vals;
    DevBuf& skeys = b->skeys;
    DevBuf& tmp = b->sort_tmp;
    FDX_TRY(keys.alloc((size_t)n * 8));
    FDX_TRY(vals.alloc((size_t)n * 4));
    FDX_TRY(skeys.alloc((size_t)n * 8));
    FDX_TRY(b->perm.alloc((size_t)n * 4));
    FDX_TRY(b->rank.alloc((size_t)n * 4));
    if (dim <= 2) FDX_TRY(b->sc2.alloc((size_t)n * 2 * sizeof(double)));
    trace_host("bin: allocations");
    const int nb = ceil_div(n, 256);
    typedef unsigned long long u64;
    const int max_axis = std::max(b->gp.nc[0], std::max(b->gp.nc[1], b->gp.nc[2]));
    int axis_bits = 1;
    while ((1LL << axis_bits) < (long long)max_axis) ++axis_bits;
    const int bits = std::min(64, axis_bits * dim);      // significant bits of the Morton key
    const bool counting = bits <= 22 && (1LL << bits) <= 8 * n + 1024 && !fdx::env("FDX_GRAPH_SORT");
    const bool shard_need = counting && shard_hi > shard_lo && (shard_lo > 0 || shard_hi < n) && shard_R > 0 && !fdx::exp_env("FDX_BAND_FULL_BINNING");
    {
        // the cell table, the key counters and (spot shards) the need flags start as zero: one block, one fill
        auto up16 = [](size_t v) { return (v + 15) / 16 * 16; };
        const size_t cells_b = up16((size_t)b->n_cells * 4);
        const size_t count_b = counting ? up16((size_t)((1LL << bits) + 1) * 4) : 0;
        const size_t need_b = shard_need ? up16((size_t)(1LL << bits) * 2) : 0;
        FDX_TRY(b->cstart.alloc(2 * cells_b + count_b + need_b));
        FDX_HIP(hipMemsetAsync(b->cstart.p, 0, 2 * cells_b + count_b + need_b, st));
        b->cend_p = reinterpret_cast<int*>(static_cast<char*>(b->cstart.p) + cells_b);
        b->count_p = counting ? reinterpret_cast<int*>(static_cast<char*>(b->cstart.p) + 2 * cells_b) : nullptr;
        b->need_p = shard_need ? reinterpret_cast<unsigned char*>(static_cast<char*>(b->cstart.p) + 2 * cells_b + count_b) : nullptr;
    }
    trace_host("bin: 1 memset");
    // Up to 4M keys (and no more than 8 per point) the order comes from counting instead of sorting: 6 launches instead of
    // the ~30 of rocprim's sort at this size (1M points: 0.42 -> 0.1 ms); FDX_GRAPH_SORT=1 forces the sort.
    if (counting) {
        const long long bins = 1LL << bits;
        b->bins = bins;
        FDX_TRY(tmp.alloc((size_t)n * 4));                        // members in arrival order (keys: 32-bit keys, vals: arrival numbers)
        FDX_TRY(b->start.alloc((size_t)(bins + 1) * 4));
        trace_host("bin: start alloc");
        hipLaunchKernelGGL(cell_count_kernel, dim3(nb), dim3(256), 0, st, d_coords, n, b->gp, keys.as<unsigned>(), vals.as<int>(),
                           b->count_p);
        FDX_CHECK_LAUNCH();
        trace_host("bin: count kernel");
        FDX_TRY(exclusive_scan_int(b->count_p, b->start.as<int>(), bins + 1, st, b->scan_tmp));
        trace_host("bin: scan");
        hipLaunchKernelGGL(cell_place_kernel, dim3(nb), dim3(256), 0, st, keys.as<unsigned>(), vals.as<int>(), b->start.as<int>(), n,
                           tmp.as<int>());
        FDX_CHECK_LAUNCH();
        const unsigned char* need = nullptr;
        if (shard_need) {
            unsigned char* n1 = b->need_p;
            unsigned char* n2 = n1 + bins;
            hipLaunchKernelGGL(cell_need_kernel<0>, dim3(ceil_div(bins, 256)), dim3(256), 0, st, b->start.as<int>(), bins, b->gp, shard_lo,
                               shard_hi, shard_R, (const unsigned char*)nullptr, n1);
            hipLaunchKernelGGL(cell_need_kernel<1>, dim3(ceil_div(bins, 256)), dim3(256), 0, st, b->start.as<int>(), bins, b->gp, shard_lo,
                               shard_hi, shard_R, (const unsigned char*)n1, n2);
            FDX_CHECK_LAUNCH();
            need = n2;
        }
        hipLaunchKernelGGL(cell_rank_kernel, dim3(nb), dim3(256), 0, st, d_coords, keys.as<unsigned>(), b->start.as<int>(),
                           tmp.as<int>(), n, dim, b->perm.as<int>(), b->rank.as<int>(), b->sc.as<double>(), b->sc2.as<double2>(), need);
        FDX_CHECK_LAUNCH();
        hipLaunchKernelGGL(cell_table_kernel, dim3(ceil_div(bins, 256)), dim3(256), 0, st, b->start.as<int>(), bins, b->gp,
                           b->cstart.as<int>(), b->cend_p);
        FDX_CHECK_LAUNCH();
        trace_host("bin: place/rank/table kernels");
        return 0;                        // no sync: the temporaries live in *b, whose owners synchronise before dropping it
    } else {
        hipLaunchKernelGGL(cell_key_kernel, dim3(nb), dim3(256), 0, st, d_coords, n, b->gp, keys.as<u64>(), vals.as<int>());
        FDX_CHECK_LAUNCH();
        size_t bytes = 0;
        FDX_HIP(rocprim::radix_sort_pairs(nullptr, bytes, keys.as<u64>(), skeys.as<u64>(), vals.as<int>(), b->perm.as<int>(),
                                          (size_t)n, 0, (unsigned)bits, st));
        FDX_TRY(tmp.alloc(bytes));
        FDX_HIP(rocprim::radix_sort_pairs(tmp.p, bytes, keys.as<u64>(), skeys.as<u64>(), vals.as<int>(), b->perm.as<int>(),
                                          (size_t)n, 0, (unsigned)bits, st));
        hipLaunchKernelGGL(gather_sorted_kernel, dim3(nb), dim3(256), 0, st, d_coords, b->perm.as<int>(), n, dim, b->sc.as<double>(), b->rank.as<int>(), b->sc2.as<double2>());
        FDX_CHECK_LAUNCH();
    }
    hipLaunchKernelGGL(cell_range_kernel, dim3(nb), dim3(256), 0, st, skeys.as<u64>(), b->sc.as<double>(), n, b->gp,
                       b->cstart.as<int>(), b->cend_p);
    FDX_CHECK_LAUNCH();
    trace_host("bin: place/rank/range kernels");
    return 0;                            // no sync: the temporaries live in *b, whose owners synchronise before dropping it
}

// workgroup tiles of the LDS-tiled sweep (needs g->ell / deg / slice_off / ell_rows / n)
static int build_tiles(fdx_graph* g, hipStream_t st) {
    const long long n = g->n;
    g->n_tiles = (int)((n + 255) / 256);
    g->tiled = false;
    g->halo_max = 0;
    if (g->n_tiles > 0 && g->ell_rows > 0) {
        FDX_TRY(g->tile_halo.alloc((size_t)g->n_tiles * FDX_TILE_HALO_CAP * 4));
        FDX_TRY(g->tile_hcnt.alloc((size_t)g->n_tiles * 4));
        FDX_TRY(g->ell_local.alloc(((size_t)g->ell_rows + 16) * 64 * 2));   // + 16 rows: the tiled sweep loads 16 rows per slice unconditionally
        hipLaunchKernelGGL(tile_halo_kernel, dim3(g->n_tiles), dim3(256), 0, st, g->ell.as<int>(), g->deg.as<int>(),
                           g->slice_off.as<int>(), n, g->tile_halo.as<int>(), g->tile_hcnt.as<int>(),
                           g->ell_local.as<unsigned short>(), (long long)g->ell_rows, (int*)nullptr);
        FDX_CHECK_LAUNCH();
    trace_host("tiles: allocs + kernel");
        std::vector<int> hc((size_t)g->n_tiles);
        FDX_HIP(hipMemcpyAsync(hc.data(), g->tile_hcnt.p, hc.size() * 4, hipMemcpyDeviceToHost, st));
        FDX_HIP(hipStreamSynchronize(st));
    trace_host("tiles: read-back + sync");
        bool ok = true;
        int mx = 0;
        for (int v : hc) { if (v < 0) ok = false; mx = std::max(mx, v); }
        g->tiled = ok;
        g->halo_max = mx;
        if (fdx::env("FDX_TRACE_HOST")) std::fprintf(stderr, "[fdx-host] tiles: %d tiles, largest halo %d, tiled %d\n", g->n_tiles, mx, (int)ok);
    }
    return 0;
}

// deg + row segments -> sliced ELL inside g (pad index = n_total)
// defer: nothing is read back here - the ELL gets room for `W_CAP` entries per row on average (the kernels stop at that bound),
// the tile tables are built behind it and the counts travel to pinned memory behind g->meta_event (graph_meta_sync).
// no_tiles (the full-size graph of a spot shard: only rows [lo, hi) are filled): no tile tables - graph_localize builds the local
// graph's own, and a pass over all n / 256 tiles here would be the shard's only work proportional to the whole graph
static int finish_ell(fdx_graph* g, const int* ws, int seg_stride, const int* seg_extra, hipStream_t st, bool defer = false,
                      bool no_tiles = false) {
    const long long n = g->n;
    g->n_slices = (int)((n + 63) / 64);
    DevBuf width, tmp;
    // width (n_slices + 1 ints), then - 16-byte aligned - red: [0] nnz, [1] widest slice; summary (2 ints, zeroed); the blocks'
    // partials.  One block, one fill.
    const int wblocks = std::min(SLICE_WIDTH_BLOCKS, std::max(1, ceil_div(g->n_slices, 4)));
    const size_t red_at = ((size_t)(g->n_slices + 1) * 4 + 15) / 16 * 16;
    FDX_TRY(width.alloc(red_at + 32 + (size_t)wblocks * 16));
    FDX_TRY(g->slice_off.alloc((size_t)(g->n_slices + 1) * 4));
    long long* red = reinterpret_cast<long long*>(static_cast<char*>(width.p) + red_at);
    int* summary = reinterpret_cast<int*>(red + 2);
    long long* part = red + 4;
    hipLaunchKernelGGL(slice_width_kernel, dim3(wblocks), dim3(256), 0, st, g->deg.as<int>(), n, g->n_slices, width.as<int>(), part, summary);
    FDX_CHECK_LAUNCH();
    FDX_TRY(exclusive_scan_int(width.as<int>(), g->slice_off.as<int>(), g->n_slices + 1, st, tmp));
    trace_host("ell: width + sums + scan");
    if (defer) {
        // room per row: three times the list length, at least 24 (slice widths of a k = 6 graph are 9-12), at most 96; a graph
        // that needs more (hubs) is rebuilt with its exact size by graph_meta_sync
        const int w_cap = fdx::env("FDX_GRAPH_WCAP") ? std::max(1, atoi(fdx::env("FDX_GRAPH_WCAP")))        // tests: force the rebuild
                                                : std::min(96, std::max(24, 3 * std::max(seg_stride, 1) + 3));
        const long long cap = (long long)g->n_slices * w_cap;
        g->ell_cap_rows = cap;
        g->n_tiles = (int)((n + 255) / 256);
        FDX_TRY(g->ell.alloc((size_t)std::max<long long>(cap, 1) * 64 * 4));
        FDX_TRY(g->tile_halo.alloc((size_t)std::max(g->n_tiles, 1) * FDX_TILE_HALO_CAP * 4));
        FDX_TRY(g->tile_hcnt.alloc((size_t)std::max(g->n_tiles, 1) * 4));
        FDX_TRY(g->ell_local.alloc(((size_t)cap + 16) * 64 * 2));
        if (g->n_tiles > 0 && !fdx::env("FDX_GRAPH_TWO_ELL_KERNELS")) {
            hipLaunchKernelGGL(tile_ell_kernel, dim3(g->n_tiles), dim3(256), 0, st, ws, seg_stride, seg_extra, (int)g->n_total, g->ell.as<int>(),
                               g->deg.as<int>(), g->slice_off.as<int>(), n, g->tile_halo.as<int>(), g->tile_hcnt.as<int>(),
                               g->ell_local.as<unsigned short>(), cap, summary);
            FDX_CHECK_LAUNCH();
        } else {
            hipLaunchKernelGGL(fill_ell_kernel, dim3(ceil_div(g->n_slices, 4)), dim3(256), 0, st, ws, seg_stride, seg_extra,
                               g->deg.as<int>(), g->slice_off.as<int>(), n, g->n_slices, (int)g->n_total, g->ell.as<int>(), cap);
            FDX_CHECK_LAUNCH();
            if (g->n_tiles > 0) {
                hipLaunchKernelGGL(tile_halo_kernel, dim3(g->n_tiles), dim3(256), 0, st, g->ell.as<int>(), g->deg.as<int>(),
                                   g->slice_off.as<int>(), n, g->tile_halo.as<int>(), g->tile_hcnt.as<int>(),
                                   g->ell_local.as<unsigned short>(), cap, summary);
                FDX_CHECK_LAUNCH();
            }
        }
        if (!g->meta_host) g->meta_host = (long long*)pinned_block_get();
        FDX_REQUIRE(g->meta_host != nullptr, "graph: pinned host block");
        if (!g->meta_event) FDX_HIP(hipEventCreateWithFlags(&g->meta_event, hipEventDisableTiming));
        for (int j = 0; j < 8; ++j) g->meta_host[j] = 0;
        // [0] low word: ell rows; [1] nnz; [2] low word: max slice width; [3] low word: largest halo, high word: failed-tile flag
        void* meta_dev = nullptr;
        FDX_HIP(hipHostGetDevicePointer(&meta_dev, g->meta_host, 0));
        hipLaunchKernelGGL(graph_meta_kernel, dim3(1), dim3(256), 0, st, part, wblocks, red, g->slice_off.as<int>(), g->n_slices, summary,
                           g->ties_dev.as<int>(), (long long*)meta_dev);
        FDX_CHECK_LAUNCH();
        FDX_HIP(hipEventRecord(g->meta_event, st));
        g->meta_stream = st;
        g->meta_pending = true;
        g->ell_rows = 0; g->nnz = 0; g->max_deg = 0; g->tiled = false; g->halo_max = 0;   // until graph_meta_sync
        trace_host("ell: deferred build queued");
        return 0;
    }
    hipLaunchKernelGGL(graph_meta_kernel, dim3(1), dim3(256), 0, st, part, wblocks, red, nullptr, 0, nullptr, nullptr, nullptr);
    FDX_CHECK_LAUNCH();
    int total = 0;
    FDX_HIP(hipMemcpyAsync(&total, g->slice_off.as<int>() + g->n_slices, 4, hipMemcpyDeviceToHost, st));
    long long h_red[2] = {0, 0};
    int h_ties[2] = {0, 0};
    FDX_HIP(hipMemcpyAsync(h_red, red, 16, hipMemcpyDeviceToHost, st));
    if (g->ties_dev.p) FDX_HIP(hipMemcpyAsync(h_ties, g->ties_dev.p, std::min<size_t>(8, g->ties_dev.bytes), hipMemcpyDeviceToHost, st));
    FDX_HIP(hipStreamSynchronize(st));
    g->knn_ties = h_ties[0];
    g->knn_far = h_ties[1];
    trace_host("ell: read-back + sync");
    g->ell_rows = total;
    g->nnz = h_red[0];
    g->max_deg = (int)(h_red[1] & 0xffffffffLL);
    FDX_TRY(g->ell.alloc((size_t)std::max<long long>(g->ell_rows, 1) * 64 * 4));
    hipLaunchKernelGGL(fill_ell_kernel, dim3(ceil_div(g->n_slices, 4)), dim3(256), 0, st, ws, seg_stride, seg_extra,
                       g->deg.as<int>(), g->slice_off.as<int>(), n, g->n_slices, (int)g->n_total, g->ell.as<int>(),
                       (long long)g->ell_rows);
    FDX_CHECK_LAUNCH();
    trace_host("ell: alloc + fill_ell");
    if (no_tiles) { g->n_tiles = (int)((n + 255) / 256); g->tiled = false; g->halo_max = 0; return 0; }
    FDX_TRY(build_tiles(g, st));
    return 0;
}

static int empty_graph(long long n, fdx_graph* g, hipStream_t st) {
    g->n = n; g->n_total = n; g->nnz = 0; g->max_deg = 0; g->ell_rows = 0;
    g->n_slices = (int)((n + 63) / 64);
    g->identity_order = true;
    FDX_TRY(g->deg.alloc((size_t)std::max<long long>(n, 1) * 4));
    FDX_TRY(g->slice_off.alloc((size_t)(g->n_slices + 1) * 4));
    FDX_TRY(g->ell.alloc(256));
    FDX_HIP(hipMemsetAsync(g->deg.p, 0, g->deg.bytes, st));
    FDX_HIP(hipMemsetAsync(g->slice_off.p, 0, g->slice_off.bytes, st));
    FDX_HIP(hipStreamSynchronize(st));
    return 0;
}

template <int KMAX>
static void launch_knn_range(const BinnedPoints& b, const int* perm, int kk, int* nbr, int* cnt, double* nn_dist, long long lo,
                             long long hi, hipStream_t st, int* indeg = nullptr, int* arrival = nullptr, int* ties = nullptr,
                             const int* row_list = nullptr, const int* row_count = nullptr, int* far_flag = nullptr, int far_drop = 0,
                             int n_direct = -1, int list_cap = 0) {
    const int far_R = BAND_R;
    if (hi <= lo) return;
    const long long n_threads = n_direct >= 0 ? (long long)n_direct + list_cap : hi - lo;
    // candidates per round trip: 4 leaves the kernel 77 registers (6 waves per SIMD), 6: 87 (5 waves), 8: 97 (4 waves);
    // 1M spots, wall per fit: 4.69 / 4.84 / 4.88 ms
    const int batch_env = fdx::exp_env("FDX_KNN_BATCH") ? atoi(fdx::exp_env("FDX_KNN_BATCH")) : 0;
    // a launch of a few hundred thousand rows does not fill the chip anyway (a spot shard's own rows + band): what it takes is one
    // walk's chain of round trips, and 8 candidates per round trip halve that chain (97 registers, 4 waves per SIMD - no loss here)
    const int batch_auto = (KMAX <= 16 && n_threads <= 300000) ? 8 : 4;
    const int batch = (KMAX <= 16 && (batch_env == 4 || batch_env == 6 || batch_env == 8)) ? batch_env : (KMAX <= 16 ? batch_auto : 8);
    auto go = [&](auto kernel) {
        hipLaunchKernelGGL(kernel, dim3(ceil_div(n_threads, 128)), dim3(128), 0, st, b.sc.as<double>(), b.sc2.as<double2>(), perm,
                           b.rank.as<int>(), b.cstart.as<int>(), b.cend_p, b.n, b.gp, kk, nbr, cnt, nn_dist, lo, hi, indeg, arrival,
                           KMAX > kk ? ties : nullptr, fdx::exp_env("FDX_KNN_NO_BATCH") ? 1 : 0, row_list, row_count, far_flag, far_R, far_drop,
                           n_direct, list_cap);
    };
    if constexpr (KMAX <= 16) {
        if (batch == 4) go(knn_kernel<KMAX, 4>);
#ifdef FDX_EXPERIMENT
        else if (batch == 6) go(knn_kernel<KMAX, 6>);
#endif
        else go(knn_kernel<KMAX, 8>);
    } else {
        go(knn_kernel<KMAX, 8>);
    }
}

template <int KMAX>
static void launch_knn(const BinnedPoints& b, const int* perm, int kk, int* nbr, int* cnt, double* nn_dist, hipStream_t st) {
    launch_knn_range<KMAX>(b, perm, kk, nbr, cnt, nn_dist, 0, b.n, st);
}

int graph_nearest_distance(const double* d_coords, long long n, int dim, double* d_out, hipStream_t st) {
    FDX_REQUIRE(dim >= 1 && dim <= 3, "graph: coordinate dimension must be 1, 2 or 3");
    FDX_REQUIRE(n >= 2 && n < 0x7fffff00LL, "graph: nearest distance needs at least two points");
    BinnedPoints b;
    FDX_TRY(bin_points(d_coords, n, dim, 2.0, 0.0, &b, st));
    launch_knn<8>(b, b.perm.as<int>(), 2, nullptr, nullptr, d_out, st);
    FDX_CHECK_LAUNCH();
    FDX_HIP(hipStreamSynchronize(st));
    return 0;
}

// ---- k-NN graph in two phases, so that a spot shard can build only its own rows ------------------------------------
// Phase 1 (knn_lists): bin ALL points (replicated; the Morton order defines the solver positions on every rank) and find
// the k nearest neighbours of the rows [lo, hi) only.  Phase 2 (from_knn_lists): row p of the symmetrised graph is
// out(p) U in(p); in(p) needs the lists of every row that points at p, so between the phases the ranks all-gather
// their list rows (the one exchange step of the build).  Phase 2 then touches own rows only: rows outside [lo, hi) keep
// degree 0 (their slices have width 0), which is all graph_localize reads of a full graph anyway (symmetry).
// Single GPU: [lo, hi) = [0, n), no exchange - the same code.
}  // namespace fdx
struct fdx_graph_plan {
    fdx::BinnedPoints b;
    long long n = 0;
    int kk = 0;
    hipStream_t st = nullptr;      // stream the binning / k-NN kernels were queued on
    fdx::DevBuf indeg, arrival;    // whole graph in one piece: in-degrees and reverse-list places from the k-NN kernel
    fdx::DevBuf ties;              // [0] rows of [lo, hi) with a tie at the k-th neighbour, [1] some walk of [lo, hi) left the 3 x 3 block (knn_kernel)
    // spot shard with band recompute (graph_knn_lists, band = true): the rows outside [lo, hi) whose lists were found here,
    // counters = {cells listed, band rows, -, band list overflowed}
    fdx::DevBuf band_rows, band_counters;
    int band_cap = 0;
    bool kernels_done = false;     // set by graph_meta_sync: the graph's meta event (recorded behind every kernel that reads
                                   // the plan) has completed - no stream sync needed, which would also wait for whatever the
                                   // caller queued behind the build (the sketch kernel of the fit)
    ~fdx_graph_plan() { if (!kernels_done) (void)hipStreamSynchronize(st); }   // nothing may still read the buffers when they go back to the pool
};
struct fdx_shard_build;
static void shard_build_drop(fdx_shard_build* sb);
static void shard_build_join(fdx_shard_build* sb);
fdx_graph::~fdx_graph() {
    if (keep_shard) shard_build_join(keep_shard);             // the helper thread may still be queueing the build's second phase
    if ((meta_pending || shard_pending) && meta_event) (void)hipEventSynchronize(meta_event);   // queued kernels still write into the buffers below
    if (keep_shard) { shard_build_drop(keep_shard); keep_shard = nullptr; }
    if (keep_plan) { delete keep_plan; keep_plan = nullptr; }
    if (meta_event) (void)hipEventDestroy(meta_event);
    if (begin_event) (void)hipEventDestroy(begin_event);
    if (meta_host) fdx::pinned_block_put(meta_host);
}
namespace fdx {

int graph_knn_lists(const double* d_coords, long long n, int dim, int k, long long lo, long long hi, int* nbr, int* cnt,
                    fdx_graph_plan** out, hipStream_t st, bool band) {
    FDX_REQUIRE(dim >= 1 && dim <= FDX_KNN_MAX_DIM, "graph: k-NN graphs take coordinates of 1 to 8 dimensions");
    FDX_REQUIRE(dim <= 3 || n <= (1 << 18), "graph: coordinates of more than 3 dimensions are searched exhaustively: at most 262144 spots");
    FDX_REQUIRE(n >= 2 && n < 0x7fffff00LL, "graph: n out of range");
    FDX_REQUIRE(k >= 1, "graph: k must be positive");
    FDX_REQUIRE(0 <= lo && lo <= hi && hi <= n, "graph: bad row range");
    const int k_act = (int)std::min<long long>(k, n - 1);           // graph.py:51
    const int kk = k_act + 1;
    FDX_REQUIRE(kk <= 64, "graph: k_neighbors above 63 is not supported");
    FDX_REQUIRE((long long)n * kk < 0x7fffff00LL, "graph: n*k too large");
    auto* plan = new fdx_graph_plan();
    plan->n = n;
    plan->kk = kk;
    plan->st = st;
    // ~4 points per grid cell: the 3 x 3 block of cells then always holds the k <= 8 nearest (no second shell, no divergence),
    // and the 256-spot Morton tiles come out more compact (1M jittered-lattice spots: graph 1.11 -> 0.95 ms, sweep 0.192 -> 0.186 ms;
    // uniform random spots: unchanged); FDX_GRAPH_TPC overrides (experiments)
    const double tpc = fdx::exp_env("FDX_GRAPH_TPC") ? atof(fdx::exp_env("FDX_GRAPH_TPC")) : 4.0;
    int rc = 0;
    if (dim > 3) {
        // solver order from the first three coordinates, exhaustive search in all of them
        DevBuf c3;
        rc = c3.alloc((size_t)n * 3 * sizeof(double));
        if (!rc) {
            hipLaunchKernelGGL(take3_kernel, dim3(ceil_div(n, 256)), dim3(256), 0, st, d_coords, n, dim, c3.as<double>());
            rc = bin_points(c3.as<double>(), n, 3, tpc, 0.0, &plan->b, st);
        }
        if (!rc) rc = plan->ties.alloc(8);
        if (!rc && hipMemsetAsync(plan->ties.p, 0, 8, st) != hipSuccess) rc = fail(FDX_ERR_HIP, "graph: memset failed");
        if (rc) { delete plan; return rc; }
        const BinnedPoints& bb = plan->b;
        int* ties_hd = plan->ties.as<int>();
        if (band && (lo > 0 || hi < n)) {               // no band in this search: report it, the caller exchanges the lists
            const int one = 1;
            if (hipMemcpyAsync(ties_hd + 1, &one, 4, hipMemcpyHostToDevice, st) != hipSuccess) { delete plan; return fail(FDX_ERR_HIP, "graph: copy failed"); }
            if (hipMemsetAsync(cnt, 0, (size_t)n * 4, st) != hipSuccess) { delete plan; return fail(FDX_ERR_HIP, "graph: memset failed"); }
        }
        if (hi > lo) {
            const dim3 grid(ceil_div(hi - lo, 256)), blk(256);
            if (kk < 8) hipLaunchKernelGGL(knn_brute_kernel<8>, grid, blk, 0, st, d_coords, bb.perm.as<int>(), bb.rank.as<int>(), n, dim, kk, nbr, cnt, lo, hi, ties_hd);
            else if (kk < 16) hipLaunchKernelGGL(knn_brute_kernel<16>, grid, blk, 0, st, d_coords, bb.perm.as<int>(), bb.rank.as<int>(), n, dim, kk, nbr, cnt, lo, hi, ties_hd);
            else if (kk < 32) hipLaunchKernelGGL(knn_brute_kernel<32>, grid, blk, 0, st, d_coords, bb.perm.as<int>(), bb.rank.as<int>(), n, dim, kk, nbr, cnt, lo, hi, ties_hd);
            else hipLaunchKernelGGL(knn_brute_kernel<64>, grid, blk, 0, st, d_coords, bb.perm.as<int>(), bb.rank.as<int>(), n, dim, kk, nbr, cnt, lo, hi, ties_hd);
        }
        if (hipGetLastError() != hipSuccess) { delete plan; return fail(FDX_ERR_HIP, "graph: k-NN kernel launch failed"); }
        if (hipStreamSynchronize(st) != hipSuccess) { delete plan; return fail(FDX_ERR_HIP, "graph: sync failed"); }   // c3 and `one` die here
        *out = plan;
        return 0;
    }
    const bool shard_band = band && (lo > 0 || hi < n) && hi > lo;
    // the in-degree counters (whole graph) and the tie / far words start as zero: filled while the host waits for the bounding box
    const bool whole = lo == 0 && hi == n;
    const std::function<int()> fills = [&]() -> int {
        if (whole) {
            FDX_TRY(plan->indeg.alloc((size_t)(n + 1) * 4));
            FDX_TRY(plan->arrival.alloc((size_t)n * kk * 4));
            FDX_HIP(hipMemsetAsync(plan->indeg.p, 0, plan->indeg.bytes, st));
        }
        FDX_TRY(plan->ties.alloc(8));
        FDX_HIP(hipMemsetAsync(plan->ties.p, 0, 8, st));
        return 0;
    };
    rc = shard_band ? bin_points(d_coords, n, dim, tpc, 0.0, &plan->b, st, lo, hi, BAND_R, &fills)
                    : bin_points(d_coords, n, dim, tpc, 0.0, &plan->b, st, 0, 0, 0, &fills);
    if (rc) { delete plan; return rc; }
    const BinnedPoints& b = plan->b;
    const int* perm = b.perm.as<int>();
    int* indeg = whole ? plan->indeg.as<int>() : nullptr;
    int* arrival = whole ? plan->arrival.as<int>() : nullptr;
    int* ties = plan->ties.as<int>();
    band = band && (lo > 0 || hi < n);
    if (band) {
        // every row without a list must read as empty: one fill of 4 bytes per row (the only pass over all n rows left in a shard's
        // symmetrisation; the lists themselves are written for the own rows and the band only)
        if (hipMemsetAsync(cnt, 0, (size_t)n * 4, st) != hipSuccess) { delete plan; return fail(FDX_ERR_HIP, "graph: memset failed"); }
    }
    // one slot more than the list length, for the tie test (kk = 64 has none: no tie count there).
    // One launch (FDX_KNN_PIECES splits it: while the leverage scores still came from the Jacobi SVD passes on the library's
    // side stream, 2-4 pieces let their small workgroups in between; the Cholesky-QR route is over before this kernel starts).
    // Where the kernel's time goes at 1M spots (300 us): ~110 us are the 7M in-degree counters (returning atomics; measured
    // with the counters taken out), the rest the walk - waves parked on its gathers two thirds of their life.
    const long long rows = hi - lo;
    const int pieces_env = fdx::exp_env("FDX_KNN_PIECES") ? atoi(fdx::exp_env("FDX_KNN_PIECES")) : 0;
    const int pieces = pieces_env > 0 ? pieces_env : 1;
    const long long step = ((rows + pieces - 1) / pieces + 127) / 128 * 128;
    const bool merged = band && rows > 0 && pieces == 1 && rows < 0x3fffffffLL && !fdx::exp_env("FDX_KNN_TWO_LAUNCHES");
    if (band && rows > 0) {
        // the band: cells next to a cell with an own row -> their rows outside [lo, hi) -> the lists of those rows.  Room for as
        // many band rows as own rows (a band is a surface: thousands of rows beside a million); an overflow is reported and the
        // caller falls back to exchanging the lists.  The band is listed FIRST (it needs the binning only): own rows and band then
        // share one k-NN launch.
        const int n_cells = b.n_cells;
        DevBuf cell_flag, cell_list;
        plan->band_cap = (int)std::min<long long>(n - rows, std::max<long long>(rows, 4096));
        rc = cell_flag.alloc((size_t)std::max(n_cells, 1) * 4);
        if (!rc) rc = cell_list.alloc((size_t)std::max(n_cells, 1) * 4);
        if (!rc) rc = plan->band_rows.alloc((size_t)std::max(plan->band_cap, 1) * 4);
        if (!rc) rc = plan->band_counters.alloc(16);
        if (rc) { delete plan; return rc; }
        const bool from_need = b.need_p && b.bins > 0 && !fdx::exp_env("FDX_BAND_CELLS");
        if ((!from_need && hipMemsetAsync(cell_flag.p, 0, cell_flag.bytes, st) != hipSuccess) ||
            hipMemsetAsync(plan->band_counters.p, 0, 16, st) != hipSuccess) {
            delete plan;
            return fail(FDX_ERR_HIP, "graph: memset failed");
        }
        int* ctr = plan->band_counters.as<int>();
        if (from_need) {
            // the shard's binning has flagged the band's keys already (first dilation of cell_need_kernel)
            hipLaunchKernelGGL(band_rows_need_kernel, dim3(ceil_div(b.bins, 256)), dim3(256), 0, st, b.start.as<int>(), b.bins,
                               b.need_p, lo, hi, plan->band_cap, plan->band_rows.as<int>(), ctr);
        } else {
            hipLaunchKernelGGL(band_cells_kernel, dim3(ceil_div(rows, 256)), dim3(256), 0, st, b.sc.as<double>(), n, b.gp, lo, hi,
                               cell_flag.as<int>(), cell_list.as<int>(), ctr);
            // at most (2 BAND_R + 1)^dim cells per own row, and never more than there are cells
            const long long side = 2 * BAND_R + 1;
            const long long max_cells = std::min<long long>(n_cells, rows * (dim == 1 ? side : dim == 2 ? side * side : side * side * side));
            hipLaunchKernelGGL(band_rows_kernel, dim3(ceil_div(max_cells, 256)), dim3(256), 0, st, cell_list.as<int>(), b.cstart.as<int>(),
                               b.cend_p, lo, hi, plan->band_cap, plan->band_rows.as<int>(), ctr);
        }
        // cell_flag / cell_list go back to the pool here: the pool orders their next use on this stream behind these kernels
    }
    const int* bl = plan->band_rows.as<int>();
    const int* bctr = plan->band_counters.p ? plan->band_counters.as<int>() + 1 : nullptr;
    const int nd = merged ? (int)rows : -1, lc = merged ? plan->band_cap : 0;
    for (long long a = lo; a < hi; a += step) {
        const long long e = std::min(hi, a + step);
        const int fd = shard_band ? 1 : 0;
        const int* rl = merged ? bl : nullptr;
        const int* rcnt = merged ? bctr : nullptr;
        if (kk < 8) launch_knn_range<8>(b, perm, kk, nbr, cnt, nullptr, a, e, st, indeg, arrival, ties, rl, rcnt, ties + 1, fd, nd, lc);
        else if (kk < 16) launch_knn_range<16>(b, perm, kk, nbr, cnt, nullptr, a, e, st, indeg, arrival, ties, rl, rcnt, ties + 1, fd, nd, lc);
        else if (kk < 32) launch_knn_range<32>(b, perm, kk, nbr, cnt, nullptr, a, e, st, indeg, arrival, ties, rl, rcnt, ties + 1, fd, nd, lc);
        else launch_knn_range<64>(b, perm, kk, nbr, cnt, nullptr, a, e, st, indeg, arrival, ties, rl, rcnt, ties + 1, fd, nd, lc);
    }
    trace_host("knn: kernel launched");
    if (band && rows > 0 && !merged) {
        const long long cap = plan->band_cap;
        if (cap > 0) {
            if (kk < 8) launch_knn_range<8>(b, perm, kk, nbr, cnt, nullptr, 0, cap, st, nullptr, nullptr, nullptr, bl, bctr, nullptr, 1);
            else if (kk < 16) launch_knn_range<16>(b, perm, kk, nbr, cnt, nullptr, 0, cap, st, nullptr, nullptr, nullptr, bl, bctr, nullptr, 1);
            else if (kk < 32) launch_knn_range<32>(b, perm, kk, nbr, cnt, nullptr, 0, cap, st, nullptr, nullptr, nullptr, bl, bctr, nullptr, 1);
            else launch_knn_range<64>(b, perm, kk, nbr, cnt, nullptr, 0, cap, st, nullptr, nullptr, nullptr, bl, bctr, nullptr, 1);
        }
    }
    if (hipGetLastError() != hipSuccess) { delete plan; return fail(FDX_ERR_HIP, "graph: k-NN kernel launch failed"); }
    *out = plan;
    return 0;
}

void graph_plan_destroy(fdx_graph_plan* plan) { delete plan; }
int graph_plan_kk(const fdx_graph_plan* plan) { return plan->kk; }
// the caller has written other lists into nbr / cnt than the ones the k-NN kernel produced: the in-degrees and reverse-list places
// that kernel drew for ITS lists (whole-graph builds) no longer apply - the symmetrisation counts again
int graph_plan_lists_replaced(fdx_graph_plan* plan) {
    if (plan->indeg.p || plan->arrival.p) {
        FDX_HIP(hipStreamSynchronize(plan->st));        // the k-NN kernel may still be writing them
        plan->indeg.release();
        plan->arrival.release();
    }
    return 0;
}
// ids (n_rows, kk): caller ids as a k-nearest query returns them (the point itself usually among them, -1 padded); row r answers for
// caller id rows[r] (rows NULL: r itself).  Written where the symmetrisation expects a row's list: at the row's solver position,
// as solver positions, the point itself dropped (utils/graph.py:70-74), compacted, -1 padded.
__global__ __launch_bounds__(256) void lists_from_ids_kernel(const long long* __restrict__ ids, const long long* __restrict__ rows,
                                                            long long n_rows, int kk, const int* __restrict__ rank,
                                                            int* __restrict__ nbr, int* __restrict__ cnt) {
    const long long r = blockIdx.x * 256LL + threadIdx.x;
    if (r >= n_rows) return;
    const long long self = rows ? rows[r] : r;
    const int p = rank[self];
    int c = 0;
    for (int j = 0; j < kk; ++j) {
        const long long id = ids[(size_t)r * kk + j];
        if (id >= 0 && id != self) nbr[(size_t)p * kk + c++] = rank[id];
    }
    cnt[p] = c;
    for (; c < kk; ++c) nbr[(size_t)p * kk + c] = -1;
}

int graph_plan_set_lists(fdx_graph_plan* plan, const long long* ids_host, const long long* rows_host, long long n_rows, int* nbr,
                         int* cnt, hipStream_t st) {
    FDX_REQUIRE(n_rows >= 0 && n_rows <= plan->n, "graph: more list rows than spots");
    if (n_rows == 0) return graph_plan_lists_replaced(plan);
    DevBuf d_ids, d_rows;
    FDX_TRY(d_ids.alloc((size_t)n_rows * plan->kk * 8));
    FDX_TRY(copy_h2d(d_ids.p, ids_host, (size_t)n_rows * plan->kk * 8, st));
    if (rows_host) {
        for (long long r = 0; r < n_rows; ++r) FDX_REQUIRE(rows_host[r] >= 0 && rows_host[r] < plan->n, "graph: list row out of range");
        FDX_TRY(d_rows.alloc((size_t)n_rows * 8));
        FDX_TRY(copy_h2d(d_rows.p, rows_host, (size_t)n_rows * 8, st));
    }
    hipLaunchKernelGGL(lists_from_ids_kernel, dim3(ceil_div(n_rows, 256)), dim3(256), 0, st, d_ids.as<long long>(),
                       rows_host ? d_rows.as<long long>() : (const long long*)nullptr, n_rows, plan->kk, plan->b.rank.as<int>(), nbr, cnt);
    FDX_CHECK_LAUNCH();
    FDX_HIP(hipStreamSynchronize(st));            // the host arrays are the caller's
    return graph_plan_lists_replaced(plan);
}

// the same with the query answers already on the device (ids_dev: n_rows x kk int64; rows_host NULL: row r answers for caller id r)
int graph_plan_set_lists_device(fdx_graph_plan* plan, const long long* ids_dev, const long long* rows_host, long long n_rows, int* nbr,
                                int* cnt, hipStream_t st) {
    FDX_REQUIRE(n_rows >= 0 && n_rows <= plan->n, "graph: more list rows than spots");
    if (n_rows == 0) return graph_plan_lists_replaced(plan);
    DevBuf d_rows;
    if (rows_host) {
        for (long long r = 0; r < n_rows; ++r) FDX_REQUIRE(rows_host[r] >= 0 && rows_host[r] < plan->n, "graph: list row out of range");
        FDX_TRY(d_rows.alloc((size_t)n_rows * 8));
        FDX_TRY(copy_h2d(d_rows.p, rows_host, (size_t)n_rows * 8, st));
    }
    hipLaunchKernelGGL(lists_from_ids_kernel, dim3(ceil_div(n_rows, 256)), dim3(256), 0, st, ids_dev,
                       rows_host ? d_rows.as<long long>() : (const long long*)nullptr, n_rows, plan->kk, plan->b.rank.as<int>(), nbr, cnt);
    FDX_CHECK_LAUNCH();
    FDX_HIP(hipStreamSynchronize(st));            // rows_host is the caller's
    return graph_plan_lists_replaced(plan);
}

int graph_plan_order(const fdx_graph_plan* plan, int* d_perm_out, int* d_rank_out, hipStream_t st) {
    if (d_perm_out) FDX_HIP(hipMemcpyAsync(d_perm_out, plan->b.perm.p, (size_t)plan->n * 4, hipMemcpyDeviceToDevice, st));
    if (d_rank_out) FDX_HIP(hipMemcpyAsync(d_rank_out, plan->b.rank.p, (size_t)plan->n * 4, hipMemcpyDeviceToDevice, st));
    return 0;
}

static int graph_from_knn_lists_impl(fdx_graph_plan* plan, const int* nbr, const int* cnt, long long lo, long long hi, fdx_graph* g,
                                     hipStream_t st, bool defer) {
    const long long n = plan->n;
    const int kk = plan->kk;
    FDX_REQUIRE(0 <= lo && lo <= hi && hi <= n, "graph: bad row range");
    FDX_REQUIRE(lo % 64 == 0, "graph: a shard must start on a 64-row slice boundary");
    g->n = n; g->n_total = n; g->identity_order = false;
    g->perm.take(plan->b.perm);
    g->rank.take(plan->b.rank);
    g->ties_dev.take(plan->ties);
    DevBuf indeg, rev_off, cursor, rev, tmp;
    // symmetrise: A + A^T, binary   (graph.py:80-81)
    const int nb = ceil_div(n, 256);
    FDX_TRY(rev_off.alloc((size_t)(n + 1) * 4));
    const bool placed = lo == 0 && hi == n && plan->indeg.p && plan->arrival.p;   // the k-NN kernel counted and placed already
    if (placed) {
        indeg.take(plan->indeg);
    } else {
        FDX_TRY(indeg.alloc((size_t)(n + 1) * 4));
        FDX_TRY(cursor.alloc((size_t)n * 4));
        FDX_HIP(hipMemsetAsync(indeg.p, 0, indeg.bytes, st));
        FDX_HIP(hipMemsetAsync(cursor.p, 0, cursor.bytes, st));
        trace_host("sym: allocs + 2 memsets");
        if (plan->band_rows.p) {      // band recompute: only the own rows and the band have lists
            if (hi > lo)
                hipLaunchKernelGGL(indegree_rows_kernel, dim3(ceil_div(hi - lo, 256)), dim3(256), 0, st, nbr, cnt, kk, indeg.as<int>(), (int)lo,
                                   (int)hi, (const int*)nullptr, (const int*)nullptr, 0);
            if (plan->band_cap > 0)
                hipLaunchKernelGGL(indegree_rows_kernel, dim3(ceil_div(plan->band_cap, 256)), dim3(256), 0, st, nbr, cnt, kk, indeg.as<int>(),
                                   (int)lo, (int)hi, plan->band_rows.as<int>(), plan->band_counters.as<int>() + 1, plan->band_cap);
        } else {
            hipLaunchKernelGGL(indegree_kernel, dim3(nb), dim3(256), 0, st, nbr, cnt, n, kk, indeg.as<int>(), (int)lo, (int)hi);
        }
        FDX_CHECK_LAUNCH();
    }
    FDX_TRY(exclusive_scan_int(indeg.as<int>(), rev_off.as<int>(), n + 1, st, tmp));
    trace_host("sym: indegree + scan");
    if (lo == 0 && hi == n) {
        FDX_TRY(rev.alloc((size_t)n * kk * 4));          // whole graph: every list entry is a reverse edge - no read-back
    } else if (plan->band_rows.p) {
        // band recompute: only the own rows and the band rows have lists, so at most (own + band) * kk entries point into [lo, hi) -
        // a bound known on the host: no read-back of the count (a synchronisation per plan)
        FDX_TRY(rev.alloc(((size_t)(hi - lo) + (size_t)plan->band_cap) * kk * 4 + 4));
    } else {
        int total_in = 0;                                // edges into [lo, hi): known only now
        FDX_HIP(hipMemcpyAsync(&total_in, rev_off.as<int>() + n, 4, hipMemcpyDeviceToHost, st));
        FDX_HIP(hipStreamSynchronize(st));
        FDX_REQUIRE(total_in >= 0 && (long long)total_in <= n * (long long)kk, "graph: reverse edge count out of range");
        FDX_TRY(rev.alloc((size_t)std::max(total_in, 1) * 4));
    }
    if (placed)
        hipLaunchKernelGGL(fill_reverse_placed_kernel, dim3(nb), dim3(256), 0, st, nbr, cnt, plan->arrival.as<int>(), n, kk,
                           rev_off.as<int>(), rev.as<int>());
    else if (plan->band_rows.p) {
        if (hi > lo)
            hipLaunchKernelGGL(fill_reverse_rows_kernel, dim3(ceil_div(hi - lo, 256)), dim3(256), 0, st, nbr, cnt, kk, rev_off.as<int>(),
                               cursor.as<int>(), rev.as<int>(), (int)lo, (int)hi, (const int*)nullptr, (const int*)nullptr, 0);
        if (plan->band_cap > 0)
            hipLaunchKernelGGL(fill_reverse_rows_kernel, dim3(ceil_div(plan->band_cap, 256)), dim3(256), 0, st, nbr, cnt, kk, rev_off.as<int>(),
                               cursor.as<int>(), rev.as<int>(), (int)lo, (int)hi, plan->band_rows.as<int>(),
                               plan->band_counters.as<int>() + 1, plan->band_cap);
    } else
        hipLaunchKernelGGL(fill_reverse_kernel, dim3(nb), dim3(256), 0, st, nbr, cnt, n, kk, rev_off.as<int>(), cursor.as<int>(),
                           rev.as<int>(), (int)lo, (int)hi);
    FDX_CHECK_LAUNCH();
    FDX_TRY(g->rows.alloc((size_t)n * kk * 2 * 4));     // capacity sum_p (kk + indeg[p]) <= 2*n*kk
    FDX_TRY(g->deg.alloc((size_t)n * 4));
    if (lo > 0 || hi < n) FDX_HIP(hipMemsetAsync(g->deg.p, 0, g->deg.bytes, st));
    if (hi > lo) {
        hipLaunchKernelGGL(merge_rows_kernel, dim3(ceil_div(hi - lo, 128)), dim3(128), 0, st, nbr, cnt, rev.as<int>(),
                           rev_off.as<int>(), g->perm.as<int>(), g->rank.as<int>(), lo, hi, kk, g->rows.as<int>(), g->deg.as<int>());
        FDX_CHECK_LAUNCH();
    }
    trace_host("sym: fill_reverse, merge_rows launched");
    g->row_stride = kk;
    g->row_extra.take(rev_off);   // keep: segment offsets
    const bool shard = lo > 0 || hi < n;
    int band_over = 0;
    if (shard && plan->band_counters.p) FDX_HIP(hipMemcpyAsync(&band_over, plan->band_counters.as<int>() + 3, 4, hipMemcpyDeviceToHost, st));
    FDX_TRY(finish_ell(g, g->rows.as<int>(), g->row_stride, g->row_extra.as<int>(), st, defer, shard));
    if (!defer) FDX_HIP(hipStreamSynchronize(st));
    if (band_over) g->knn_far = 1;          // the band list overflowed: same remedy as a far walk (exchange the lists)
    return 0;
}

int graph_from_knn_lists(fdx_graph_plan* plan, const int* nbr, const int* cnt, long long lo, long long hi, fdx_graph* g,
                         hipStream_t st) {
    return graph_from_knn_lists_impl(plan, nbr, cnt, lo, hi, g, st, false);
}

// Waits for a deferred build (finish_ell) and takes over what only the device knew.  Cheap no-op otherwise.
static int shard_meta_sync(fdx_graph* g);
int graph_meta_sync(const fdx_graph* gc) {
    if (gc && gc->shard_pending) return shard_meta_sync(const_cast<fdx_graph*>(gc));
    if (!gc || !gc->meta_pending) return 0;
    fdx_graph* g = const_cast<fdx_graph*>(gc);
    FDX_HIP(hipEventSynchronize(g->meta_event));
    g->meta_pending = false;
    // the queued kernels are done: their inputs can go
    g->keep_nbr.release();
    g->keep_cnt.release();
    if (g->keep_plan) { g->keep_plan->kernels_done = true; graph_plan_destroy(g->keep_plan); g->keep_plan = nullptr; }
    const long long rows = g->meta_host[0] & 0xffffffffLL;
    g->nnz = g->meta_host[1];
    g->max_deg = (int)(g->meta_host[2] & 0xffffffffLL);
    g->knn_ties = g->meta_host[4] & 0xffffffffLL;
    if (rows > g->ell_cap_rows) {                     // the bound was too small (hubs): build the ELL again with its exact size
        trace_host("meta: ELL bound too small, rebuilding");
        return finish_ell(g, g->rows.as<int>(), g->row_stride, g->row_extra.as<int>(), g->meta_stream, false);
    }
    g->ell_rows = rows;
    g->halo_max = (int)(g->meta_host[3] & 0xffffffffLL);
    g->tiled = g->n_tiles > 0 && rows > 0 && (g->meta_host[3] >> 32) == 0;
    if (fdx::env("FDX_TRACE_HOST")) std::fprintf(stderr, "[fdx-host] meta: rows %lld of %lld, nnz %lld, largest halo %d, tiled %d\n", rows, g->ell_cap_rows, g->nnz, g->halo_max, (int)g->tiled);
    return 0;
}

int graph_build_knn(const double* d_coords, long long n, int dim, int k, fdx_graph* g, hipStream_t st) {
    FDX_REQUIRE(dim >= 1 && dim <= FDX_KNN_MAX_DIM, "graph: k-NN graphs take coordinates of 1 to 8 dimensions");
    FDX_REQUIRE(n >= 0 && n < 0x7fffff00LL, "graph: n out of range");
    FDX_REQUIRE(k >= 0, "graph: k must be non-negative");
    const int k_act = (int)std::min<long long>(k, n - 1);           // graph.py:51
    if (k_act <= 0) return empty_graph(n, g, st);                   // graph.py:53-57
    const int kk = k_act + 1;
    FDX_REQUIRE(kk <= 64, "graph: k_neighbors above 63 is not supported");
    FDX_REQUIRE((long long)n * kk < 0x7fffff00LL, "graph: n*k too large");
    DevBuf nbr, cnt;
    FDX_TRY(nbr.alloc((size_t)n * kk * 4));
    FDX_TRY(cnt.alloc((size_t)n * 4));
    fdx_graph_plan* plan = nullptr;
    FDX_TRY(graph_knn_lists(d_coords, n, dim, k, 0, n, nbr.as<int>(), cnt.as<int>(), &plan, st));
    // Whole graph in one piece: the rest is queued without a host round trip (FDX_GRAPH_SYNC=1: built to the end here); the
    // lists and the binned points stay with the graph until graph_meta_sync has seen the kernels finish.
    const bool defer = !fdx::env("FDX_GRAPH_SYNC");
    const int rc = graph_from_knn_lists_impl(plan, nbr.as<int>(), cnt.as<int>(), 0, n, g, st, defer);
    if (rc == 0 && defer && g->meta_pending) {
        g->keep_nbr.take(nbr);
        g->keep_cnt.take(cnt);
        g->keep_plan = plan;
        return 0;
    }
    delete plan;
    return rc;
}

// Rows [lo, hi) (solver positions) of the radius graph; the other rows are left empty.  A radius graph is symmetric by
// construction (graph.py:115-121), so a shard's own rows need nothing from the other shards.
int graph_build_radius(const double* d_coords, long long n, int dim, double radius, long long lo, long long hi, fdx_graph* g,
                       hipStream_t st) {
    FDX_REQUIRE(dim >= 1 && dim <= 3, "graph: coordinate dimension must be 1, 2 or 3");
    FDX_REQUIRE(n >= 0 && n < 0x7fffff00LL, "graph: n out of range");
    FDX_REQUIRE(radius > 0.0 && std::isfinite(radius), "graph: radius must be positive");
    FDX_REQUIRE(0 <= lo && lo <= hi && hi <= n, "graph: bad row range");
    FDX_REQUIRE(lo % 64 == 0, "graph: a shard must start on a 64-row slice boundary");
    if (n <= 1) return empty_graph(n, g, st);
    BinnedPoints b;
    FDX_TRY(bin_points(d_coords, n, dim, 1.0, radius, &b, st));   // cell edge >= radius: one shell suffices
    const int R = (int)std::ceil(radius / b.gp.h[0] * (1.0 + 1e-12));
    g->n = n; g->n_total = n; g->identity_order = false;
    g->perm.take(b.perm);
    g->rank.take(b.rank);
    DevBuf cnt, tmp;
    FDX_TRY(cnt.alloc((size_t)(n + 1) * 4));
    FDX_TRY(g->row_extra.alloc((size_t)(n + 1) * 4));
    FDX_HIP(hipMemsetAsync(cnt.p, 0, cnt.bytes, st));
    const int nb = ceil_div(n, 128);
    hipLaunchKernelGGL(radius_kernel<0>, dim3(nb), dim3(128), 0, st, b.sc.as<double>(), b.cstart.as<int>(), b.cend_p, n,
                       b.gp, radius, R, cnt.as<int>(), (const int*)nullptr, (int*)nullptr, lo, hi);
    FDX_CHECK_LAUNCH();
    FDX_TRY(exclusive_scan_int(cnt.as<int>(), g->row_extra.as<int>(), n + 1, st, tmp));
    int total = 0;
    FDX_HIP(hipMemcpyAsync(&total, g->row_extra.as<int>() + n, 4, hipMemcpyDeviceToHost, st));
    FDX_HIP(hipStreamSynchronize(st));
    FDX_REQUIRE(total >= 0, "graph: radius graph has too many edges");
    FDX_TRY(g->rows.alloc((size_t)std::max(total, 1) * 4));
    hipLaunchKernelGGL(radius_kernel<1>, dim3(nb), dim3(128), 0, st, b.sc.as<double>(), b.cstart.as<int>(), b.cend_p, n,
                       b.gp, radius, R, (int*)nullptr, g->row_extra.as<int>(), g->rows.as<int>(), lo, hi);
    FDX_CHECK_LAUNCH();
    hipLaunchKernelGGL(sort_rows_kernel, dim3(nb), dim3(128), 0, st, g->rows.as<int>(), g->row_extra.as<int>(), g->perm.as<int>(), n);
    FDX_CHECK_LAUNCH();
    g->deg.take(cnt);
    g->row_stride = 0;
    FDX_TRY(finish_ell(g, g->rows.as<int>(), 0, g->row_extra.as<int>(), st, false, lo > 0 || hi < n));
    FDX_HIP(hipStreamSynchronize(st));
    return 0;
}

// CSR in the caller's labels: indptr (n+1) int64, indices (nnz) int32 ascending per row.  Device outputs.
int graph_export_csr(const fdx_graph* g, long long* d_indptr, int* d_indices, hipStream_t st) {
    const long long n = g->n;
    if (n == 0) return 0;
    if (g->nnz == 0) {
        FDX_HIP(hipMemsetAsync(d_indptr, 0, (size_t)(n + 1) * 8, st));
        return 0;
    }
    FDX_REQUIRE(!g->identity_order && g->rows.p, "graph export: graph was not built from coordinates");
    DevBuf deg_o, tmp;
    FDX_TRY(deg_o.alloc((size_t)(n + 1) * 4));
    FDX_HIP(hipMemsetAsync(deg_o.p, 0, deg_o.bytes, st));
    const int nb = ceil_div(n, 256);
    hipLaunchKernelGGL(deg_to_orig_kernel, dim3(nb), dim3(256), 0, st, g->deg.as<int>(), g->perm.as<int>(), n, deg_o.as<int>());
    FDX_CHECK_LAUNCH();
    FDX_TRY(exclusive_scan_i64(deg_o.as<int>(), d_indptr, n + 1, st, tmp));
    hipLaunchKernelGGL(export_rows_kernel, dim3(nb), dim3(256), 0, st, g->rows.as<int>(), g->row_stride, g->row_extra.as<int>(),
                       g->deg.as<int>(), g->perm.as<int>(), d_indptr, n, d_indices);
    FDX_CHECK_LAUNCH();
    FDX_HIP(hipStreamSynchronize(st));
    return 0;
}

// ------------------------------------------------------------------------------------------------ sharding
// A rank owns the contiguous range [lo, hi) of the sorted order (lo a multiple of 256).  Its local graph indexes own
// spots 0..n_own-1, then the halo (neighbour positions outside the range, ascending global position), then the zero row.
// External neighbours of the own rows, with repetitions: PASS 0 counts them, PASS 1 appends them to `list` (order
// irrelevant: the list is sorted and made unique afterwards).  The halo is found from what the own rows reference, so the
// cost is proportional to the shard, not to the whole graph.
// PASS 1 also emits, for every such reference, the key (owner rank of q) << 32 | (own row - lo): by symmetry of the graph
// the owner of q needs this row in ITS halo, so the sorted unique keys are the send lists of all peers at once.
template <int PASS>
__global__ __launch_bounds__(256) void collect_halo_kernel(const int* __restrict__ ell, const int* __restrict__ slice_off,
                                                           const int* __restrict__ deg, long long lo, long long hi,
                                                           const long long* __restrict__ bounds, int n_ranks,
                                                           int* __restrict__ counter, int* __restrict__ list,
                                                           unsigned long long* __restrict__ send_keys) {
    const long long p = lo + blockIdx.x * 256LL + threadIdx.x;
    if (p >= hi) return;
    const int* seg = ell + (size_t)slice_off[p >> 6] * 64 + (p & 63);
    int local = 0;
    for (int m = 0; m < deg[p]; ++m) {
        const int q = seg[(size_t)m * 64];
        if (q < lo || q >= hi) {
            if (PASS) {
                const int at = atomicAdd(counter, 1);
                list[at] = q;
                int r = 0;
                while (r + 1 < n_ranks && (long long)q >= bounds[r + 1]) ++r;          // owner of q (a handful of ranks)
                send_keys[at] = ((unsigned long long)r << 32) | (unsigned long long)(p - lo);
            } else {
                ++local;
            }
        }
    }
    if (!PASS && local) atomicAdd(counter, local);
}

__global__ __launch_bounds__(256) void localize_ell_kernel(const int* __restrict__ ell_g, const int* __restrict__ slice_off_g,
                                                           const int* __restrict__ deg_g, const int* __restrict__ perm_g,
                                                           const int* __restrict__ halo, int n_halo, long long lo, long long hi,
                                                           int n_total_g, int* __restrict__ ell_l,
                                                           int* __restrict__ slice_off_l, int* __restrict__ deg_l,
                                                           int* __restrict__ perm_l) {
    const long long n_own = hi - lo;
    const int n_slices_l = (int)((n_own + 63) / 64);
    const int s0 = (int)(lo >> 6);
    const int base_rows = slice_off_g[s0];
    const long long t = blockIdx.x * 256LL + threadIdx.x;
    if (t <= n_slices_l) slice_off_l[t] = slice_off_g[s0 + t] - base_rows;
    if (t < n_own) {
        deg_l[t] = deg_g[lo + t];
        perm_l[t] = perm_g ? perm_g[lo + t] : (int)(lo + t);
    }
    const int s = (int)(t >> 6), lane = (int)(t & 63);
    if (s < n_slices_l) {
        const int w0 = slice_off_g[s0 + s], w = slice_off_g[s0 + s + 1] - w0;
        const int n_total_l = (int)n_own + n_halo;
        for (int m = 0; m < w; ++m) {
            const int q = ell_g[((size_t)w0 + m) * 64 + lane];
            int v;
            if (q == n_total_g) v = n_total_l;                 // pad -> local zero row
            else if (q >= lo && q < hi) v = (int)(q - lo);
            else {                                             // halo slot = rank of q in the sorted, unique halo list
                int a = 0, b = n_halo;
                while (a < b) { const int mid = (a + b) >> 1; if (halo[mid] < q) a = mid + 1; else b = mid; }
                v = (int)n_own + a;
            }
            ell_l[((size_t)(w0 - base_rows) + m) * 64 + lane] = v;
        }
    }
}

int graph_localize(const fdx_graph* full, long long lo, long long hi, int n_ranks, const long long* bounds, int my_rank,
                   fdx_graph* loc, hipStream_t st) {
    FDX_REQUIRE(full && loc && bounds, "graph_localize: null argument");
    FDX_REQUIRE(lo >= 0 && hi >= lo && hi <= full->n, "graph_localize: bad range");
    FDX_REQUIRE(lo % 256 == 0, "graph_localize: range start must be a multiple of 256");
    FDX_REQUIRE(full->n_total == full->n, "graph_localize: input must be a full (unsharded) graph");
    loc->knn_ties = full->knn_ties;   // a shard's full-size graph counted the ties of the rows it was built for
    const long long ng = full->n, n_own = hi - lo;
    loc->n = n_own;
    loc->identity_order = false;
    loc->global_lo = lo;
    loc->world_n = bounds[n_ranks];
    loc->n_slices = (int)((n_own + 63) / 64);
    // halo = sorted unique set of the neighbour positions outside [lo, hi) that the own rows reference
    DevBuf tmp, counter, ext, ext_sorted, n_uniq, d_bounds, skeys, skeys_sorted, skeys_uniq;
    int n_halo = 0, n_send = 0;
    const int s0 = (int)(lo >> 6);
    int so2[2] = {0, 0};
    bool have_so2 = false;
    FDX_TRY(counter.alloc(8));
    FDX_TRY(d_bounds.alloc((size_t)(n_ranks + 1) * 8));
    FDX_HIP(hipMemcpyAsync(d_bounds.p, bounds, (size_t)(n_ranks + 1) * 8, hipMemcpyHostToDevice, st));
    if (n_own > 0 && full->ell_rows > 0) {
        FDX_HIP(hipMemsetAsync(counter.p, 0, 8, st));
        hipLaunchKernelGGL(collect_halo_kernel<0>, dim3(ceil_div(n_own, 256)), dim3(256), 0, st, full->ell.as<int>(),
                           full->slice_off.as<int>(), full->deg.as<int>(), lo, hi, d_bounds.as<long long>(), n_ranks,
                           counter.as<int>(), (int*)nullptr, (unsigned long long*)nullptr);
        FDX_CHECK_LAUNCH();
        int n_ext = 0;
        FDX_HIP(hipMemcpyAsync(&n_ext, counter.p, 4, hipMemcpyDeviceToHost, st));
        if (loc->n_slices > 0) {                         // the two slice offsets that bound the own rows ride in the same round trip
            FDX_HIP(hipMemcpyAsync(&so2[0], full->slice_off.as<int>() + s0, 4, hipMemcpyDeviceToHost, st));
            FDX_HIP(hipMemcpyAsync(&so2[1], full->slice_off.as<int>() + s0 + loc->n_slices, 4, hipMemcpyDeviceToHost, st));
            have_so2 = true;
        }
        FDX_HIP(hipStreamSynchronize(st));
        if (n_ext > 0) {
            FDX_TRY(ext.alloc((size_t)n_ext * 4));
            FDX_TRY(ext_sorted.alloc((size_t)n_ext * 4));
            FDX_TRY(loc->halo_global.alloc((size_t)n_ext * 4));
            FDX_TRY(n_uniq.alloc(8));
            FDX_TRY(skeys.alloc((size_t)n_ext * 8));
            FDX_TRY(skeys_sorted.alloc((size_t)n_ext * 8));
            FDX_TRY(skeys_uniq.alloc((size_t)n_ext * 8));
            FDX_HIP(hipMemsetAsync(counter.p, 0, 8, st));
            hipLaunchKernelGGL(collect_halo_kernel<1>, dim3(ceil_div(n_own, 256)), dim3(256), 0, st, full->ell.as<int>(),
                               full->slice_off.as<int>(), full->deg.as<int>(), lo, hi, d_bounds.as<long long>(), n_ranks,
                               counter.as<int>(), ext.as<int>(), skeys.as<unsigned long long>());
            FDX_CHECK_LAUNCH();
            {   // send lists of all peers: sort + unique of the (peer, row) keys
                typedef unsigned long long u64;
                size_t b1 = 0, b2 = 0;
                FDX_HIP(rocprim::radix_sort_keys(nullptr, b1, skeys.as<u64>(), skeys_sorted.as<u64>(), (size_t)n_ext, 0, 64, st));
                FDX_HIP(rocprim::unique(nullptr, b2, skeys_sorted.as<u64>(), skeys_uniq.as<u64>(), n_uniq.as<int>() + 1, (size_t)n_ext,
                                        rocprim::equal_to<u64>(), st));
                FDX_TRY(tmp.alloc(std::max(b1, b2)));
                FDX_HIP(rocprim::radix_sort_keys(tmp.p, b1, skeys.as<u64>(), skeys_sorted.as<u64>(), (size_t)n_ext, 0, 64, st));
                FDX_HIP(rocprim::unique(tmp.p, b2, skeys_sorted.as<u64>(), skeys_uniq.as<u64>(), n_uniq.as<int>() + 1, (size_t)n_ext,
                                        rocprim::equal_to<u64>(), st));
                FDX_HIP(hipMemcpyAsync(&n_send, n_uniq.as<int>() + 1, 4, hipMemcpyDeviceToHost, st));
            }
            size_t sb = 0, ub = 0;
            FDX_HIP(rocprim::radix_sort_keys(nullptr, sb, ext.as<int>(), ext_sorted.as<int>(), (size_t)n_ext, 0, 32, st));
            FDX_HIP(rocprim::unique(nullptr, ub, ext_sorted.as<int>(), loc->halo_global.as<int>(), n_uniq.as<int>(), (size_t)n_ext,
                                    rocprim::equal_to<int>(), st));
            FDX_TRY(tmp.alloc(std::max(sb, ub)));
            FDX_HIP(rocprim::radix_sort_keys(tmp.p, sb, ext.as<int>(), ext_sorted.as<int>(), (size_t)n_ext, 0, 32, st));
            FDX_HIP(rocprim::unique(tmp.p, ub, ext_sorted.as<int>(), loc->halo_global.as<int>(), n_uniq.as<int>(), (size_t)n_ext,
                                    rocprim::equal_to<int>(), st));
            FDX_HIP(hipMemcpyAsync(&n_halo, n_uniq.p, 4, hipMemcpyDeviceToHost, st));
            FDX_HIP(hipStreamSynchronize(st));
        }
    }
    if (!loc->halo_global.p) FDX_TRY(loc->halo_global.alloc(4));
    loc->n_total = n_own + n_halo;
    // local ELL / deg / perm
    if (loc->n_slices > 0 && !have_so2) {
        FDX_HIP(hipMemcpyAsync(&so2[0], full->slice_off.as<int>() + s0, 4, hipMemcpyDeviceToHost, st));
        FDX_HIP(hipMemcpyAsync(&so2[1], full->slice_off.as<int>() + s0 + loc->n_slices, 4, hipMemcpyDeviceToHost, st));
        FDX_HIP(hipStreamSynchronize(st));
    }
    loc->ell_rows = so2[1] - so2[0];
    FDX_TRY(loc->ell.alloc((size_t)std::max<long long>(loc->ell_rows, 1) * 64 * 4));
    FDX_TRY(loc->slice_off.alloc((size_t)(loc->n_slices + 1) * 4));
    FDX_TRY(loc->deg.alloc((size_t)std::max<long long>(n_own, 1) * 4));
    FDX_TRY(loc->perm.alloc((size_t)std::max<long long>(n_own, 1) * 4));
    if (n_own > 0) {
        hipLaunchKernelGGL(localize_ell_kernel, dim3(ceil_div(loc->n_slices * 64LL + 1, 256)), dim3(256), 0, st,
                           full->ell.as<int>(), full->slice_off.as<int>(), full->deg.as<int>(),
                           full->identity_order ? (const int*)nullptr : full->perm.as<int>(), loc->halo_global.as<int>(), n_halo,
                           lo, hi, (int)ng,
                           loc->ell.as<int>(), loc->slice_off.as<int>(), loc->deg.as<int>(), loc->perm.as<int>());
        FDX_CHECK_LAUNCH();
    } else {
        FDX_HIP(hipMemsetAsync(loc->slice_off.p, 0, loc->slice_off.bytes, st));
    }
    // Everything else the host needs arrives in ONE round trip (each used to have its own - seven synchronisations, ~0.2 ms for
    // a strong-scaled rank whose whole sketch is 0.25 ms): nnz / widest row of the own rows, the tile tables' summary, the halo
    // positions (recv lists) and the (peer, row) send keys, all into pinned memory.
    loc->nnz = 0;
    loc->max_deg = 0;
    loc->n_tiles = (int)((n_own + 255) / 256);
    loc->tiled = false;
    loc->halo_max = 0;
    const bool tiles = loc->n_tiles > 0 && loc->ell_rows > 0;
    DevBuf red, rtmp;
    FDX_TRY(red.alloc(32));                               // [0] nnz, [1] low word: max degree; summary: 2 ints at byte 16
    FDX_HIP(hipMemsetAsync(red.p, 0, 32, st));
    int* summary = red.as<int>() + 4;
    if (n_own > 0) {
        size_t rb = 0, rb2 = 0;
        auto deg64 = rocprim::make_transform_iterator(loc->deg.as<int>(), [] __device__(int v) { return (long long)v; });
        FDX_HIP(rocprim::reduce(nullptr, rb, deg64, red.as<long long>(), 0LL, (size_t)n_own, rocprim::plus<long long>(), st));
        FDX_HIP(rocprim::reduce(nullptr, rb2, loc->deg.as<int>(), red.as<int>() + 2, 0, (size_t)n_own, rocprim::maximum<int>(), st));
        FDX_TRY(rtmp.alloc(std::max(rb, rb2)));
        FDX_HIP(rocprim::reduce(rtmp.p, rb, deg64, red.as<long long>(), 0LL, (size_t)n_own, rocprim::plus<long long>(), st));
        FDX_HIP(rocprim::reduce(rtmp.p, rb2, loc->deg.as<int>(), red.as<int>() + 2, 0, (size_t)n_own, rocprim::maximum<int>(), st));
    }
    if (tiles) {
        FDX_TRY(loc->tile_halo.alloc((size_t)loc->n_tiles * FDX_TILE_HALO_CAP * 4));
        FDX_TRY(loc->tile_hcnt.alloc((size_t)loc->n_tiles * 4));
        FDX_TRY(loc->ell_local.alloc(((size_t)loc->ell_rows + 16) * 64 * 2));   // + 16 rows: the tiled sweep loads 16 rows per slice unconditionally
        hipLaunchKernelGGL(tile_halo_kernel, dim3(loc->n_tiles), dim3(256), 0, st, loc->ell.as<int>(), loc->deg.as<int>(),
                           loc->slice_off.as<int>(), n_own, loc->tile_halo.as<int>(), loc->tile_hcnt.as<int>(),
                           loc->ell_local.as<unsigned short>(), (long long)loc->ell_rows, summary);
        FDX_CHECK_LAUNCH();
    }
    const size_t hg_at = 64, hk_at = hg_at + ((size_t)n_halo * 4 + 63) / 64 * 64;
    unsigned char* pin = (unsigned char*)pinned_scratch(3, hk_at + (size_t)n_send * 8 + 64);
    FDX_REQUIRE(pin != nullptr, "graph_localize: pinned host buffer");
    FDX_HIP(hipMemcpyAsync(pin, red.p, 32, hipMemcpyDeviceToHost, st));
    if (n_halo) FDX_HIP(hipMemcpyAsync(pin + hg_at, loc->halo_global.p, (size_t)n_halo * 4, hipMemcpyDeviceToHost, st));
    if (n_send) FDX_HIP(hipMemcpyAsync(pin + hk_at, skeys_uniq.p, (size_t)n_send * 8, hipMemcpyDeviceToHost, st));
    FDX_HIP(hipStreamSynchronize(st));
    {
        const long long* h_red = (const long long*)pin;
        const int* h_sum = (const int*)(pin + 16);
        loc->nnz = h_red[0];
        loc->max_deg = (int)(h_red[1] & 0xffffffffLL);
        if (tiles) {
            loc->tiled = h_sum[1] == 0;
            loc->halo_max = h_sum[0];
        }
    }
    // halo ownership (recv) and send lists per peer
    loc->recv_off.assign((size_t)n_ranks + 1, 0);
    loc->send_off.assign((size_t)n_ranks + 1, 0);
    const int* hg_p = (const int*)(pin + hg_at);
    const unsigned long long* hk = (const unsigned long long*)(pin + hk_at);
    struct { const int* p; const int* begin() const { return p; } } hg{hg_p};
    for (int r = 0; r < n_ranks; ++r) {
        const long long e = bounds[r + 1];
        loc->recv_off[(size_t)r + 1] = (int)(std::lower_bound(hg.begin(), hg.begin() + n_halo, (int)std::min<long long>(e, 0x7fffffff)) - hg.begin());
    }
    // send lists: the unique (peer, row) keys are already grouped by peer and ascending in the row
    std::vector<int> all_send((size_t)n_send);
    {
        int at = 0;
        for (int r = 0; r < n_ranks; ++r) {
            while (at < n_send && (int)(hk[(size_t)at] >> 32) == r) { all_send[(size_t)at] = (int)(hk[(size_t)at] & 0xffffffffULL); ++at; }
            loc->send_off[(size_t)r + 1] = at;
        }
    }
    FDX_TRY(loc->send_idx.alloc(std::max<size_t>(all_send.size(), 1) * 4));
    if (!all_send.empty())
        FDX_HIP(hipMemcpyAsync(loc->send_idx.p, all_send.data(), all_send.size() * 4, hipMemcpyHostToDevice, st));
    FDX_HIP(hipStreamSynchronize(st));
    return 0;
}

// ------------------------------------------------------------------------------------------------ deferred shard build
// One rank's LOCAL graph of a k-NN job in one queued pipeline (graph_shard_knn): after the bounding box nothing returns to the
// host.  Replaces, for the common case, the sequence knn_lists(band) -> from_knn_lists -> localize with its seven round trips
// (a strong-scaled rank of 125k spots spent 0.9 ms there for ~0.1 ms of kernels).  Every quantity the stepwise path read back
// to size an allocation is replaced by a bound the host knows (ELL rows: w_cap per slice as in the deferred whole-graph build;
// halo: the band capacity; send lists: 3 x own rows); a bound that turns out too small is reported (shard_overflow) and the
// caller rebuilds by the stepwise path.  Rows, order of the entries and tile tables are those of the stepwise path bit for bit.

// own row p: every neighbour position outside [lo, hi) is flagged (the halo is the set of flagged positions) and the owner
// ranks of those neighbours are collected in a bit mask (by symmetry the owner of q needs row p in ITS halo: the masks are the
// send lists); per 256-row tile and peer the number of rows to send, and whether the tile holds any such row
constexpr int SHARD_MAX_RANKS = 32;
struct ShardBounds { long long b[SHARD_MAX_RANKS + 1]; };      // range starts of the ranks, by value (no upload to wait for)
__global__ __launch_bounds__(256) void shard_mark_kernel(const int* __restrict__ ws, int kk, const int* __restrict__ seg_extra,
                                                         const int* __restrict__ deg, long long lo, long long hi,
                                                         const ShardBounds bounds_v, int n_ranks,
                                                         int* __restrict__ flag, unsigned* __restrict__ mask,
                                                         int* __restrict__ cnt_rb, int nblk, int* __restrict__ tileflag) {
    __shared__ int s_cnt[32];
    const int tid = threadIdx.x;
    if (tid < 32) s_cnt[tid] = 0;
    __syncthreads();
    const long long t = blockIdx.x * 256LL + tid;
    const long long p = lo + t;
    unsigned m = 0;
    if (p < hi) {
        const int* seg = ws + (size_t)t * kk + seg_extra[t];
        const int dg = deg[t];
        for (int e = 0; e < dg; ++e) {
            const int q = seg[e];
            if (q < lo || q >= hi) {
                flag[q] = 1;
                int r = 0;
#pragma unroll
                for (int j = 1; j < SHARD_MAX_RANKS; ++j) r += (j < n_ranks && (long long)q >= bounds_v.b[j]) ? 1 : 0;   // no dynamic index into the by-value array
                m |= 1u << r;
            }
        }
        mask[t] = m;
    }
    unsigned any = m;
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) any |= __shfl_xor(any, off, 64);
    if (any) {
        for (int r = 0; r < n_ranks; ++r)
            if ((any >> r) & 1u) {
                const int c = __popcll(__ballot((m >> r) & 1u));
                if ((tid & 63) == 0) atomicAdd(&s_cnt[r], c);
            }
    }
    __syncthreads();
    if (tid < n_ranks) cnt_rb[(size_t)tid * nblk + blockIdx.x] = s_cnt[tid];
    if (tid == 0) {
        int a = 0;
        for (int r = 0; r < n_ranks; ++r) a |= s_cnt[r];
        tileflag[blockIdx.x] = a ? 1 : 0;
    }
}

// halo_global[slot] = q for every flagged position (ascending: slot = number of flagged positions before q)
__global__ __launch_bounds__(256) void shard_halo_scatter_kernel(const int* __restrict__ flag, const int* __restrict__ hscan,
                                                                 long long n, int* __restrict__ halo_global, long long cap) {
    const long long q = blockIdx.x * 256LL + threadIdx.x;
    if (q >= n || !flag[q]) return;
    const int s = hscan[q];
    if (s < cap) halo_global[s] = (int)q;
}

// local sliced ELL straight from the row segments: own neighbour -> q - lo, outside -> n_own + its halo slot, pad -> n_total
__global__ __launch_bounds__(256) void fill_ell_local_kernel(const int* __restrict__ ws, int kk, const int* __restrict__ seg_extra,
                                                             const int* __restrict__ deg, const int* __restrict__ slice_off,
                                                             long long lo, long long hi, int n_slices,
                                                             const int* __restrict__ hscan, long long n_all,
                                                             int* __restrict__ ell, long long cap_rows,
                                                             const int* __restrict__ perm_g, int* __restrict__ perm_l) {
    const int lane = threadIdx.x & 63;
    const int s = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (s >= n_slices) return;
    const long long n_own = hi - lo;
    const long long t = (long long)s * 64 + lane;
    if (perm_l && t < n_own) perm_l[t] = perm_g[lo + t];         // caller's id of the own row
    if ((long long)slice_off[n_slices] > cap_rows) return;      // bound too small: reported, rebuilt by the stepwise path
    const int w0 = slice_off[s], w = slice_off[s + 1] - w0;
    const int dg = (t < n_own) ? deg[t] : 0;
    const int* seg = (t < n_own) ? ws + (size_t)t * kk + seg_extra[t] : ws;
    const int pad = (int)n_own + hscan[n_all];
    for (int m = 0; m < w; ++m) {
        int v = pad;
        if (m < dg) {
            const int q = seg[m];
            v = (q >= lo && q < hi) ? (int)(q - lo) : (int)n_own + hscan[q];
        }
        ell[((size_t)w0 + m) * 64 + lane] = v;
    }
}

// send_idx, grouped by peer, ascending row inside a peer: off_rb (exclusive scan of cnt_rb, peer-major) places every tile's rows
__global__ __launch_bounds__(256) void shard_send_fill_kernel(const unsigned* __restrict__ mask, const int* __restrict__ off_rb,
                                                              const int* __restrict__ tileflag, int nblk, long long n_own,
                                                              int n_ranks, int* __restrict__ send_idx, long long cap) {
    if (!tileflag[blockIdx.x]) return;
    __shared__ int s_w[4];
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const long long t = blockIdx.x * 256LL + tid;
    const unsigned m = (t < n_own) ? mask[t] : 0u;
    for (int r = 0; r < n_ranks; ++r) {
        const int base = off_rb[(size_t)r * nblk + blockIdx.x];
        const int cnt = off_rb[(size_t)r * nblk + blockIdx.x + 1] - base;       // peer-major: the next entry is the next tile (or the next peer's first)
        if (cnt == 0) continue;                                                // block-uniform
        const unsigned long long b = __ballot((m >> r) & 1u);
        if (lane == 0) s_w[wv] = __popcll(b);
        __syncthreads();
        int before = 0;
        for (int w2 = 0; w2 < wv; ++w2) before += s_w[w2];
        if ((m >> r) & 1u) {
            const long long at = (long long)base + before + __popcll(b & ((1ULL << lane) - 1ULL));
            if (at < cap) send_idx[at] = (int)t;
        }
        __syncthreads();
    }
}

// boundary tiles (hold a row some peer needs) and interior tiles, each ascending; one workgroup
__global__ __launch_bounds__(256) void shard_tile_lists_kernel(const int* __restrict__ tileflag, int n_tiles,
                                                               int* __restrict__ tiles_b, int* __restrict__ tiles_i,
                                                               int* __restrict__ counts) {
    __shared__ int s_w[4];
    __shared__ int s_base_b, s_base_i;
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    if (tid == 0) { s_base_b = 0; s_base_i = 0; }
    __syncthreads();
    for (int t0 = 0; t0 < n_tiles; t0 += 256) {
        const int t = t0 + tid;
        const bool in = t < n_tiles;
        const bool isb = in && tileflag[t] != 0;
        const unsigned long long b = __ballot(isb);
        if (lane == 0) s_w[wv] = __popcll(b);
        __syncthreads();
        int before = 0, total = 0;
        for (int w2 = 0; w2 < 4; ++w2) { if (w2 < wv) before += s_w[w2]; total += s_w[w2]; }
        const int rb = before + __popcll(b & ((1ULL << lane) - 1ULL));      // boundary tiles before t in this round
        const int bb = s_base_b, bi = s_base_i;
        if (in) {
            if (isb) tiles_b[bb + rb] = t;
            else tiles_i[bi + (tid - rb)] = t;
        }
        __syncthreads();
        if (tid == 0) { s_base_b = bb + total; s_base_i = bi + (min(256, n_tiles - t0) - total); }
        __syncthreads();
    }
    if (tid == 0) { counts[0] = s_base_b; counts[1] = s_base_i; }
}

// everything the host will ask for, in one pinned block (FDX_PINNED_BLOCK_BYTES): [0] ELL rows, [1] nnz of the own rows,
// [2] widest slice, [3] largest tile halo | failed-tile flag << 32, [4] tied rows, [5] far | band overflow << 1, [6] halo spots,
// [7] rows to send (all peers), [8] boundary tiles, [9] interior tiles; [16 + r] send_off[r], [56 + r] recv_off[r] (r = 0..n_ranks)
constexpr int SHARD_META_SEND = 16, SHARD_META_RECV = 56;
__global__ __launch_bounds__(256) void shard_meta_kernel(const long long* __restrict__ part, int n_part, const int* __restrict__ slice_off,
                                                         int n_slices, const int* __restrict__ summary, const int* __restrict__ ties,
                                                         const int* __restrict__ band_ctr, const int* __restrict__ hscan, long long n_all,
                                                         const int* __restrict__ off_rb, int nblk, const ShardBounds bounds_v, int n_ranks,
                                                         const int* __restrict__ tile_counts, int* __restrict__ send_off_dev,
                                                         int* __restrict__ recv_off_dev, long long* __restrict__ meta,
                                                         double* __restrict__ counts_dev, long long ell_cap, long long halo_cap,
                                                         long long send_cap) {
    __shared__ long long s_sum[256];
    __shared__ int s_max[256];
    __shared__ long long s_b[SHARD_MAX_RANKS + 1];
    const int r = threadIdx.x;
    if (r <= SHARD_MAX_RANKS) s_b[r] = bounds_v.b[r];
    long long tot = 0;                                            // nnz and widest slice from the blocks' partials (slice_width_kernel)
    int wmax = 0;
    for (int b = r; b < n_part; b += 256) { tot += part[2 * b]; wmax = max(wmax, (int)part[2 * b + 1]); }
    s_sum[r] = tot;
    s_max[r] = wmax;
    __syncthreads();
    for (int s = 128; s > 0; s >>= 1) {
        if (r < s) { s_sum[r] += s_sum[r + s]; s_max[r] = max(s_max[r], s_max[r + s]); }
        __syncthreads();
    }
    if (r <= n_ranks) {
        const int so = off_rb[(size_t)r * nblk];                      // r == n_ranks: the total (last entry of the scan)
        const int ro = hscan[s_b[r]];
        send_off_dev[r] = so;
        recv_off_dev[r] = ro;
        meta[SHARD_META_SEND + r] = so;
        meta[SHARD_META_RECV + r] = ro;
    }
    if (r != 0) return;
    meta[0] = (long long)slice_off[n_slices];
    meta[1] = s_sum[0];
    meta[2] = (long long)s_max[0];
    meta[3] = (long long)(unsigned)summary[0] | ((long long)summary[1] << 32);
    meta[4] = (long long)ties[0];
    meta[5] = (long long)(ties[1] != 0) | ((long long)(band_ctr[3] != 0) << 1);
    meta[6] = (long long)hscan[n_all];
    meta[7] = (long long)off_rb[(size_t)n_ranks * nblk];
    meta[8] = (long long)tile_counts[0];
    meta[9] = (long long)tile_counts[1];
    // the same counts where an all-reduce over the ranks can take them without the host: edges of the own rows, tied own rows,
    // "a walk left its block / the band list overflowed", "a bound of this pipeline was too small"
    counts_dev[0] = (double)s_sum[0];
    counts_dev[1] = (double)ties[0];
    counts_dev[2] = (ties[1] != 0 || band_ctr[3] != 0) ? 1.0 : 0.0;
    counts_dev[3] = ((long long)slice_off[n_slices] > ell_cap || (long long)hscan[n_all] > halo_cap ||
                     (long long)off_rb[(size_t)n_ranks * nblk] > send_cap) ? 1.0 : 0.0;
}

}  // namespace fdx
struct fdx_shard_build {
    fdx_graph_plan* plan = nullptr;
    fdx::DevBuf nbr, cnt, zeros, rev_off, rev, rows, hscan, mask, off_rb, tileflag, tile_counts, scan_tmp;
    // what the second phase (shard_queue_rest) needs; `queued` = it has run
    long long n = 0, lo = 0, hi = 0;
    int kk = 0, n_ranks = 0;
    long long bounds[fdx::SHARD_MAX_RANKS + 1] = {};
    hipStream_t st_first = nullptr;          // stream of the first phase (the caller's)
    hipEvent_t ev_first = nullptr;           // recorded there behind the k-NN lists and the copy of the own rows' ids
    bool queued = false;
    std::shared_ptr<fdx::HelperTicket> ticket;   // the second phase was handed to the helper thread: wait before touching anything it writes
    ~fdx_shard_build() {
        if (plan) fdx::graph_plan_destroy(plan);
        if (ev_first) (void)hipEventDestroy(ev_first);
    }
};
static void shard_build_join(fdx_shard_build* sb) {
    if (sb->ticket) { (void)fdx::helper_wait(sb->ticket); sb->ticket.reset(); }
}
static void shard_build_drop(fdx_shard_build* sb) {
    if (sb->ticket) { (void)fdx::helper_wait(sb->ticket); sb->ticket.reset(); }
    // second phase queued: the graph's meta event has completed, nothing reads the buffers any more; never queued: the plan's
    // destructor waits for the first phase's stream
    if (sb->plan && sb->queued) sb->plan->kernels_done = true;
    delete sb;
}
namespace fdx {

// the library's per-device stream for the second phase of a shard build (beside the sketch of the own rows on the caller's stream)
hipStream_t library_plan_stream() {
    static hipStream_t streams[64] = {};
    static std::mutex mu;
    if (fdx::exp_env("FDX_NO_PLAN_STREAM")) return nullptr;
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return nullptr;
    std::lock_guard<std::mutex> lk(mu);
    if (!streams[dev]) {
        int lo = 0, hi = 0;
        (void)hipDeviceGetStreamPriorityRange(&lo, &hi);
        if (hipStreamCreateWithPriority(&streams[dev], hipStreamNonBlocking, hi) != hipSuccess &&
            hipStreamCreateWithFlags(&streams[dev], hipStreamNonBlocking) != hipSuccess)
            streams[dev] = nullptr;
    }
    return streams[dev];
}

int shard_queue_rest(fdx_graph* loc, hipStream_t st);

int graph_shard_knn(const double* d_coords, long long n, int dim, int k, int n_ranks, const long long* bounds, int my_rank,
                    fdx_graph* loc, hipStream_t st) {
    FDX_REQUIRE(dim >= 1 && dim <= 3, "graph_shard_knn: the deferred shard build takes 1 to 3 coordinates");
    FDX_REQUIRE(n_ranks >= 2 && n_ranks <= SHARD_MAX_RANKS && my_rank >= 0 && my_rank < n_ranks, "graph_shard_knn: 2 to 32 ranks");
    static_assert(SHARD_META_RECV + SHARD_MAX_RANKS + 1 <= (int)(FDX_PINNED_BLOCK_BYTES / 8), "shard meta block");
    FDX_REQUIRE(n >= 2 && k >= 1, "graph_shard_knn: needs at least two spots and k >= 1");
    const long long lo = bounds[my_rank], hi = bounds[my_rank + 1];
    FDX_REQUIRE(bounds[0] == 0 && bounds[n_ranks] == n, "graph_shard_knn: bounds must cover [0, n]");
    for (int r = 0; r < n_ranks; ++r)
        FDX_REQUIRE(bounds[r + 1] >= bounds[r] && bounds[r] % 256 == 0, "graph_shard_knn: range starts must be non-decreasing multiples of 256");
    FDX_REQUIRE(hi > lo, "graph_shard_knn: this rank owns no row (use the stepwise path)");
    const long long n_own = hi - lo;
    auto sb = std::make_unique<fdx_shard_build>();
    const int kk = (int)std::min<long long>(k, n - 1) + 1;
    FDX_REQUIRE(kk <= 64, "graph: k_neighbors above 63 is not supported");
    FDX_REQUIRE((long long)n * kk < 0x7fffff00LL, "graph: n*k too large");
    FDX_TRY(sb->nbr.alloc((size_t)n * kk * 4));
    FDX_TRY(sb->cnt.alloc((size_t)n * 4));
    // lists of the own rows and of the band (bin_points reads the bounding box back: the one round trip of the build)
    FDX_TRY(graph_knn_lists(d_coords, n, dim, k, lo, hi, sb->nbr.as<int>(), sb->cnt.as<int>(), &sb->plan, st, true));
    fdx_graph_plan* plan = sb->plan;
    FDX_REQUIRE(plan->band_rows.p != nullptr, "graph_shard_knn: no band (one rank owns everything)");
    // the caller's ids of the own rows are final with the binning: copied now, so that the caller can lay its rows out while the
    // rest of the build is still to be queued
    FDX_TRY(loc->perm.alloc((size_t)n_own * 4));
    FDX_HIP(hipMemcpyAsync(loc->perm.p, plan->b.perm.as<int>() + lo, (size_t)n_own * 4, hipMemcpyDeviceToDevice, st));
    loc->n = n_own;
    loc->n_total = n_own;                        // until graph_meta_sync
    loc->identity_order = false;
    loc->global_lo = lo;
    loc->world_n = n;
    loc->n_tiles = ceil_div(n_own, 256);
    loc->n_slices = (int)((n_own + 63) / 64);
    loc->shard_world = n_ranks;
    sb->n = n; sb->lo = lo; sb->hi = hi; sb->kk = kk; sb->n_ranks = n_ranks;
    for (int r = 0; r <= SHARD_MAX_RANKS; ++r) sb->bounds[r] = r <= n_ranks ? bounds[r] : n;
    sb->st_first = st;
    FDX_HIP(hipEventCreateWithFlags(&sb->ev_first, hipEventDisableTiming));
    FDX_HIP(hipEventRecord(sb->ev_first, st));
    if (!loc->meta_event) FDX_HIP(hipEventCreateWithFlags(&loc->meta_event, hipEventDisableTiming));
    loc->shard_pending = true;
    loc->keep_shard = sb.release();
    // The second phase (~35 dependent launches, 0.12 ms of host time) is queued by the library's helper thread on the plan stream
    // while this thread returns to the caller: a rank's critical path is the host's way to the sketch of its own rows, which does
    // not need the graph.  Consumers join (graph_shard_join).  FDX_SHARD_EAGER=1: here and now, on the caller's stream (=2: on the
    // plan stream); FDX_NO_HELPER_THREAD: by the first consumer.
    if (const char* e = fdx::exp_env("FDX_SHARD_EAGER")) return shard_queue_rest(loc, atoi(e) == 2 ? nullptr : st);   // 2: on the plan stream
    if (!fdx::exp_env("FDX_NO_HELPER_THREAD")) loc->keep_shard->ticket = helper_submit([loc] { return shard_queue_rest(loc, nullptr); });
    return 0;
}

// the second phase of a pending shard build is queued when this returns (by the helper thread, or here)
int graph_shard_join(const fdx_graph* gc) {
    fdx_graph* g = const_cast<fdx_graph*>(gc);
    if (!g || !g->shard_pending || !g->keep_shard) return 0;
    fdx_shard_build* sb = g->keep_shard;
    if (sb->ticket) {
        const std::shared_ptr<HelperTicket> t = sb->ticket;
        sb->ticket.reset();
        FDX_TRY(helper_wait(t));
    }
    return shard_queue_rest(g, nullptr);         // no-op when queued
}

// Second phase of graph_shard_knn: symmetrisation of the own rows, halo, local ELL, tile tables, send lists, meta block - on `st`
// (the library's plan stream when NULL), behind the first phase.
int shard_queue_rest(fdx_graph* loc, hipStream_t st) {
    fdx_shard_build* sb = loc->keep_shard;
    if (!sb || sb->queued) return 0;
    if (!st) st = library_plan_stream();
    if (!st) st = sb->st_first;
    PoolStream pool_stream(st);
    if (st != sb->st_first) FDX_HIP(hipStreamWaitEvent(st, sb->ev_first, 0));
    sb->queued = true;                           // whatever happens below, kernels of this phase may be in flight on `st`
    fdx_graph_plan* plan = sb->plan;
    plan->st = st;
    const long long n = sb->n, lo = sb->lo, hi = sb->hi, n_own = hi - lo;
    const int kk = sb->kk, n_ranks = sb->n_ranks;
    const int* nbr = sb->nbr.as<int>();
    const int* cnt = sb->cnt.as<int>();

    // ---- everything that must start as zero, in one block with one fill: in-degrees and cursors of the own rows, the halo flags
    // (one per position of the whole order), per-tile / per-peer send counts (+ the closing entry of their scan), slice widths and
    // the reduction cells behind them
    const int nblk = ceil_div(n_own, 256);
    const int wblocks = std::min(SLICE_WIDTH_BLOCKS, std::max(1, ceil_div(loc->n_slices, 4)));
    auto up16 = [](size_t v) { return (v + 15) / 16 * 16; };
    const size_t z_indeg = 0, z_cursor = up16(z_indeg + (size_t)(n_own + 1) * 4), z_flag = up16(z_cursor + (size_t)n_own * 4),
                 z_cntrb = up16(z_flag + (size_t)(n + 1) * 4), z_width = up16(z_cntrb + ((size_t)n_ranks * nblk + 1) * 4),
                 z_red = up16(z_width + (size_t)(loc->n_slices + 1) * 4), z_end = z_red + 32 + (size_t)wblocks * 16;
    FDX_TRY(sb->zeros.alloc(z_end));
    FDX_HIP(hipMemsetAsync(sb->zeros.p, 0, z_red + 32, st));
    char* zb = static_cast<char*>(sb->zeros.p);
    int* indeg_l = reinterpret_cast<int*>(zb + z_indeg);
    int* cursor_l = reinterpret_cast<int*>(zb + z_cursor);
    int* flag = reinterpret_cast<int*>(zb + z_flag);
    int* cnt_rb = reinterpret_cast<int*>(zb + z_cntrb);
    int* width = reinterpret_cast<int*>(zb + z_width);
    long long* red = reinterpret_cast<long long*>(zb + z_red);
    int* summary = reinterpret_cast<int*>(red + 2);
    long long* part = red + 4;
    ShardBounds bv;
    for (int r = 0; r <= SHARD_MAX_RANKS; ++r) bv.b[r] = sb->bounds[r];

    // ---- symmetrise the own rows (graph.py:80-81): everything indexed by the LOCAL row t = p - lo (pointers shifted by lo);
    // own rows and band rows in one launch each
    FDX_TRY(sb->rev_off.alloc((size_t)(n_own + 1) * 4));
    int* indeg_s = indeg_l - lo;
    int* cursor_s = cursor_l - lo;
    int* rev_off_s = sb->rev_off.as<int>() - lo;
    const int bcap = plan->band_cap;
    hipLaunchKernelGGL(indegree_rows_kernel, dim3(ceil_div(n_own + bcap, 256)), dim3(256), 0, st, nbr, cnt, kk, indeg_s, (int)lo, (int)hi,
                       plan->band_rows.as<int>(), plan->band_counters.as<int>() + 1, bcap, (int)n_own);
    FDX_CHECK_LAUNCH();
    FDX_TRY(exclusive_scan_int(indeg_l, sb->rev_off.as<int>(), n_own + 1, st, sb->scan_tmp));
    FDX_TRY(sb->rev.alloc(((size_t)n_own + (size_t)bcap) * kk * 4 + 4));
    hipLaunchKernelGGL(fill_reverse_rows_kernel, dim3(ceil_div(n_own + bcap, 256)), dim3(256), 0, st, nbr, cnt, kk, rev_off_s, cursor_s,
                       sb->rev.as<int>(), (int)lo, (int)hi, plan->band_rows.as<int>(), plan->band_counters.as<int>() + 1, bcap,
                       (int)n_own);
    FDX_CHECK_LAUNCH();
    FDX_TRY(sb->rows.alloc((size_t)n_own * kk * 2 * 4));
    FDX_TRY(loc->deg.alloc((size_t)n_own * 4));
    int* rows_s = sb->rows.as<int>() - (size_t)lo * kk;
    int* deg_s = loc->deg.as<int>() - lo;
    hipLaunchKernelGGL(merge_rows_kernel, dim3(ceil_div(n_own, 128)), dim3(128), 0, st, nbr, cnt, sb->rev.as<int>(), rev_off_s,
                       plan->b.perm.as<int>(), plan->b.rank.as<int>(), lo, hi, kk, rows_s, deg_s);
    FDX_CHECK_LAUNCH();

    // ---- halo and send masks
    FDX_TRY(sb->hscan.alloc((size_t)(n + 1) * 4));
    FDX_TRY(sb->mask.alloc((size_t)n_own * 4));
    FDX_TRY(sb->off_rb.alloc(((size_t)n_ranks * nblk + 1) * 4));
    FDX_TRY(sb->tileflag.alloc((size_t)nblk * 4));
    FDX_TRY(sb->tile_counts.alloc(8));
    hipLaunchKernelGGL(shard_mark_kernel, dim3(nblk), dim3(256), 0, st, sb->rows.as<int>(), kk, sb->rev_off.as<int>(),
                       loc->deg.as<int>(), lo, hi, bv, n_ranks, flag, sb->mask.as<unsigned>(), cnt_rb, nblk, sb->tileflag.as<int>());
    FDX_CHECK_LAUNCH();
    FDX_TRY(exclusive_scan_int(flag, sb->hscan.as<int>(), n + 1, st, sb->scan_tmp));
    loc->shard_halo_cap = std::max<long long>(plan->band_cap, 1);          // halo rows are band rows
    FDX_TRY(loc->halo_global.alloc((size_t)loc->shard_halo_cap * 4));
    hipLaunchKernelGGL(shard_halo_scatter_kernel, dim3(ceil_div(n, 256)), dim3(256), 0, st, flag, sb->hscan.as<int>(), n,
                       loc->halo_global.as<int>(), loc->shard_halo_cap);
    FDX_CHECK_LAUNCH();

    // ---- local sliced ELL, tile tables
    FDX_TRY(loc->slice_off.alloc((size_t)(loc->n_slices + 1) * 4));
    hipLaunchKernelGGL(slice_width_kernel, dim3(wblocks), dim3(256), 0, st, loc->deg.as<int>(), n_own, loc->n_slices, width, part);
    FDX_CHECK_LAUNCH();
    FDX_TRY(exclusive_scan_int(width, loc->slice_off.as<int>(), loc->n_slices + 1, st, sb->scan_tmp));
    // (tests: FDX_GRAPH_WCAP forces the "bound too small" remedy, on every rank or - FDX_GRAPH_WCAP_RANK - on one)
    int rank = 0;
    while (rank + 1 < n_ranks && !(sb->bounds[rank] == lo && sb->bounds[rank + 1] == hi)) ++rank;
    const bool cap_forced = fdx::env("FDX_GRAPH_WCAP") && (!fdx::env("FDX_GRAPH_WCAP_RANK") || atoi(fdx::env("FDX_GRAPH_WCAP_RANK")) == rank);
    const int w_cap = cap_forced ? std::max(1, atoi(fdx::env("FDX_GRAPH_WCAP"))) : std::min(96, std::max(24, 3 * kk + 3));
    const long long cap = (long long)loc->n_slices * w_cap;
    loc->shard_ell_cap = cap;
    FDX_TRY(loc->ell.alloc((size_t)std::max<long long>(cap, 1) * 64 * 4));
    FDX_TRY(loc->tile_halo.alloc((size_t)nblk * FDX_TILE_HALO_CAP * 4));
    FDX_TRY(loc->tile_hcnt.alloc((size_t)nblk * 4));
    FDX_TRY(loc->ell_local.alloc(((size_t)cap + 16) * 64 * 2));
    hipLaunchKernelGGL(fill_ell_local_kernel, dim3(ceil_div(loc->n_slices, 4)), dim3(256), 0, st, sb->rows.as<int>(), kk,
                       sb->rev_off.as<int>(), loc->deg.as<int>(), loc->slice_off.as<int>(), lo, hi, loc->n_slices, sb->hscan.as<int>(), n,
                       loc->ell.as<int>(), cap, plan->b.perm.as<int>(), (int*)nullptr);
    FDX_CHECK_LAUNCH();
    hipLaunchKernelGGL(tile_halo_kernel, dim3(nblk), dim3(256), 0, st, loc->ell.as<int>(), loc->deg.as<int>(), loc->slice_off.as<int>(),
                       n_own, loc->tile_halo.as<int>(), loc->tile_hcnt.as<int>(), loc->ell_local.as<unsigned short>(), cap, summary);
    FDX_CHECK_LAUNCH();

    // ---- send lists, boundary / interior tiles
    FDX_TRY(exclusive_scan_int(cnt_rb, sb->off_rb.as<int>(), (long long)n_ranks * nblk + 1, st, sb->scan_tmp));
    loc->shard_send_cap = n_own * std::min(n_ranks - 1, 3) + 1024;
    FDX_TRY(loc->send_idx.alloc((size_t)loc->shard_send_cap * 4));
    hipLaunchKernelGGL(shard_send_fill_kernel, dim3(nblk), dim3(256), 0, st, sb->mask.as<unsigned>(), sb->off_rb.as<int>(),
                       sb->tileflag.as<int>(), nblk, n_own, n_ranks, loc->send_idx.as<int>(), loc->shard_send_cap);
    FDX_TRY(loc->tiles_boundary.alloc((size_t)nblk * 4));
    FDX_TRY(loc->tiles_interior.alloc((size_t)nblk * 4));
    hipLaunchKernelGGL(shard_tile_lists_kernel, dim3(1), dim3(256), 0, st, sb->tileflag.as<int>(), nblk, loc->tiles_boundary.as<int>(),
                       loc->tiles_interior.as<int>(), sb->tile_counts.as<int>());
    FDX_CHECK_LAUNCH();

    // ---- the numbers the host will ask for
    FDX_TRY(loc->send_off_dev.alloc((size_t)(n_ranks + 1) * 4));
    FDX_TRY(loc->recv_off_dev.alloc((size_t)(n_ranks + 1) * 4));
    FDX_TRY(loc->counts_dev.alloc(4 * sizeof(double)));
    if (!loc->meta_host) loc->meta_host = (long long*)pinned_block_get();
    FDX_REQUIRE(loc->meta_host != nullptr, "graph: pinned host block");
    std::memset(loc->meta_host, 0, FDX_PINNED_BLOCK_BYTES);
    void* meta_dev = nullptr;
    FDX_HIP(hipHostGetDevicePointer(&meta_dev, loc->meta_host, 0));
    hipLaunchKernelGGL(shard_meta_kernel, dim3(1), dim3(256), 0, st, part, wblocks, loc->slice_off.as<int>(), loc->n_slices, summary,
                       plan->ties.as<int>(), plan->band_counters.as<int>(), sb->hscan.as<int>(), n, sb->off_rb.as<int>(), nblk, bv, n_ranks,
                       sb->tile_counts.as<int>(), loc->send_off_dev.as<int>(), loc->recv_off_dev.as<int>(), (long long*)meta_dev,
                       loc->counts_dev.as<double>(), loc->shard_ell_cap, loc->shard_halo_cap, loc->shard_send_cap);
    FDX_CHECK_LAUNCH();
    FDX_HIP(hipEventRecord(loc->meta_event, st));
    loc->meta_stream = st;
    return 0;
}

// takes over what the queued shard build left in the pinned block
static int shard_meta_sync(fdx_graph* g) {
    // a failed second phase (helper-thread error, allocation failure in shard_queue_rest) stays failed: the join is a no-op the next
    // time round, the meta event was never recorded and the pinned block is empty or absent - every later call must fail again
    if (g->shard_failed) return fail(g->shard_failed, "graph: the queued shard build failed earlier; the graph is unusable");
    if (const int jrc = graph_shard_join(g)) { g->shard_failed = jrc; return jrc; }
    if (!g->meta_host || !g->meta_event) { g->shard_failed = FDX_ERR_INVALID; return fail(FDX_ERR_INVALID, "graph: the queued shard build left no counts"); }
    FDX_HIP(hipEventSynchronize(g->meta_event));
    g->shard_pending = false;
    if (g->keep_shard) { shard_build_drop(g->keep_shard); g->keep_shard = nullptr; }
    const long long* m = g->meta_host;
    const int W = g->shard_world;
    const long long rows = m[0], n_halo = m[6], n_send = m[7];
    g->nnz = m[1];
    g->max_deg = (int)(m[2] & 0xffffffffLL);
    g->knn_ties = m[4];
    g->knn_far = (int)(m[5] & 3) ? 1 : 0;
    g->shard_overflow = (rows > g->shard_ell_cap || n_halo > g->shard_halo_cap || n_send > g->shard_send_cap) ? 1 : 0;
    g->ell_rows = rows;
    g->n_total = g->n + n_halo;
    g->halo_max = (int)(m[3] & 0xffffffffLL);
    g->tiled = g->n_tiles > 0 && rows > 0 && (m[3] >> 32) == 0 && !g->shard_overflow;
    g->send_off.assign((size_t)W + 1, 0);
    g->recv_off.assign((size_t)W + 1, 0);
    for (int r = 0; r <= W; ++r) {
        g->send_off[(size_t)r] = (int)m[SHARD_META_SEND + r];
        g->recv_off[(size_t)r] = (int)m[SHARD_META_RECV + r];
    }
    g->n_tiles_boundary = (int)m[8];
    g->n_tiles_interior = (int)m[9];
    return 0;
}

int graph_copy_perm(const fdx_graph* g, int* d_out, hipStream_t st) {
    if (g->n == 0) return 0;
    if (g->identity_order || !g->perm.p) {
        hipLaunchKernelGGL(iota_kernel, dim3(ceil_div(g->n, 256)), dim3(256), 0, st, d_out, g->n);
        FDX_CHECK_LAUNCH();
    } else {
        // a shard build still pending: its first phase wrote the ids on the stream it was given
        if (g->shard_pending && g->keep_shard && g->keep_shard->ev_first && st != g->keep_shard->st_first)
            FDX_HIP(hipStreamWaitEvent(st, g->keep_shard->ev_first, 0));
        FDX_HIP(hipMemcpyAsync(d_out, g->perm.p, (size_t)g->n * 4, hipMemcpyDeviceToDevice, st));
    }
    return 0;
}

}  // namespace fdx
