// Tile kernel: preprocess + CountSketch + H contraction of 16 spots at a time, without atomics and without Y_sketch.
//
// Replaces, for the common shapes, both sketch_contract_kernel (fused_kernels.cpp: LDS atomics, 34 % of the HBM roofline,
// bound by ds_add_f64 bank conflicts) and the pair sketch_rows_scatter_kernel -> xyt_split_kernel
// (flashdeconv/core/deconv.py:177-197 _preprocess_data, core/sketching.py:160-206 project_to_sketch,
// core/solver.py:205-223 precompute_XtY).
//
//   staging     The rows of a tile (16 consecutive spots in solver order) are copied HBM -> LDS by LDS-DMA
//               (global_load_lds_dwordx4: 1 KB per wave instruction, no registers, no ds_write), one column block of GB
//               genes at a time, double buffered: block i+1 is in flight while block i is consumed.
//   gather      lane (r, q) of wave w owns spot r of the tile and the buckets of the slots (w, j, q), j < JW (tile_plan.h).
//               It walks the genes of those buckets through a static table in LDS {weight f64, offset u16} and adds
//               weight * f(y) into a register - no atomics, genes in ascending order inside every bucket (the
//               reference's summation order), bit-reproducible.
//   contraction the bucket sums sit exactly where v_mfma_f64_16x16x4_f64 wants its B operand (B[k = q][n = r]); the
//               wave's slice of X_sketch is register-resident as A operands, so the sums never leave the registers.
//               The NW partial 16 x 16 type tiles are added in wave order through LDS (deterministic) and stored to H.
//   log-CPM     needs the row sum before the first element can be transformed: each wave sums one or two rows of the
//               NEXT tile from registers (plain global loads, which also pull the rows into L2 / Infinity Cache ahead of
//               the DMA) while the current tile is consumed.  log1p is table driven: 1 + x is reduced by an 8-bit
//               reciprocal (v_rcp_f32) to 1 + r with |r| <= 2^-8, log1p(x) = T[reciprocal] + r - r^2/2 + ... - r^6/6
//               (~20 instructions instead of ~45; < 3 ulp).
#include <algorithm>
#include <cmath>
#include <cstdlib>
#include <memory>
#include <mutex>

#include "device_math.h"
#include "fdx_internal.h"
#include "fdx_kernels.h"
#include "sketch_plan.h"
#include "tile_plan.h"

namespace fdx {

typedef double double4_t __attribute__((ext_vector_type(4)));

constexpr int TILE_ROWS = 16;
constexpr int TILE_ROW_PAD = 16;          // bytes between staged rows: a 16-byte shift keeps the DMA destination aligned
constexpr int LOG_TAB_N = 1921;           // 15 binades x 128 + 1 reciprocals in [2^-15, 1]
constexpr int LOG_TAB_BASE = 14336;       // (bits of 2^-15) >> 16

// Scalars of a launch.  The arrays are separate __restrict__ kernel parameters: only then may the compiler fetch the
// wave-uniform ones (row_map, ent_base, len_tab) with scalar loads.  As vector loads they would sit in vmcnt behind the
// LDS-DMA pieces in flight, and every use would wait for the next block to land - no overlap left.
struct TileArgs {
    long long ldy, n, ldh;
    int G, d, K;
    int NE, GB, NBLK, RS, jw_used;
};

// log1p(x) for x in [0, 32000): see the header.  logt[i] = -log(c_i), c_i the reciprocal with bit pattern
// (LOG_TAB_BASE + i) << 16.  Anything else (negative, NaN, huge) takes the library path, as the reference would.
__device__ __forceinline__ double tile_log1p(double x, const double* logt) {
    if (__builtin_expect(!(x >= 0.0) || !(x < 32000.0), 0)) return log1p(x);
    const float uf = 1.0f + (float)x;
    unsigned bits = __float_as_uint(__builtin_amdgcn_rcpf(uf));
    bits = (bits + 0x8000u) & 0xFFFF0000u;                   // reciprocal rounded to 8 significant bits
    const float invf = __uint_as_float(bits);
    const double inv = (double)invf;
    const double im1 = (double)(invf - 1.0f);                // exact
    const double r = fma(x, inv, im1);                       // (1 + x) * inv - 1 with one rounding
    const double t = logt[(int)(bits >> 16) - LOG_TAB_BASE];
    double p = fma(r, -1.0 / 6.0, 0.2);
    p = fma(r, p, -0.25);
    p = fma(r, p, 1.0 / 3.0);
    p = fma(r, p, -0.5);
    p = fma(r, p, 1.0);
    return fma(r, p, t);
}

template <typename T> struct TileVec;
template <> struct TileVec<float> { typedef float type __attribute__((ext_vector_type(4))); };
template <> struct TileVec<double> { typedef double type __attribute__((ext_vector_type(2))); };

// One 1 KB piece of a staged row: lane l copies 16 bytes from src to lds_base + 16 * l.
__device__ __forceinline__ void dma16(const void* src, unsigned char* lds_base) {
    __builtin_amdgcn_global_load_lds((const void __attribute__((address_space(1)))*)src,
                                     (void __attribute__((address_space(3)))*)lds_base, 16, 0, 0);
}

// Workgroup barrier that orders LDS traffic only.  __syncthreads() also drains vmcnt, i.e. it would wait for the LDS-DMA
// pieces of the next block, which are meant to stay in flight across the reduction at the end of a tile.
__device__ __forceinline__ void lds_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

template <typename T, int MODE, int NW, int JW, int TT>
__global__ __launch_bounds__(NW * 64, NW / 4) void tile_sketch_kernel(
    const TileArgs a, const T* __restrict__ Yp, const int* __restrict__ row_map, const double* __restrict__ Xs,
    double* __restrict__ H, double* __restrict__ row_sumsq, const double* __restrict__ w_tab,
    const unsigned short* __restrict__ off_tab, const unsigned char* __restrict__ len_tab,
    const int* __restrict__ ent_base, const int* __restrict__ slot_bucket, const double* __restrict__ log_tab) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    typedef typename TileVec<T>::type V;
    constexpr int PER = 16 / sizeof(T);
    constexpr int NT = NW * 64;
    constexpr int RPW = TILE_ROWS / NW > 0 ? TILE_ROWS / NW : 1;   // rows of the next tile a wave sums (log modes)
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int r = lane & 15, q = lane >> 4;
    const int stage_bytes = TILE_ROWS * a.RS;
    const int NEp = (a.NE + 7) & ~7;
    double* w_l = reinterpret_cast<double*>(smem + 2 * (size_t)stage_bytes);
    unsigned short* off_l = reinterpret_cast<unsigned short*>(w_l + NEp);
    double* scales = reinterpret_cast<double*>(off_l + NEp);               // [2][16]
    double* logt = scales + 2 * TILE_ROWS;                                 // [LOG_TAB_N] (log modes)
    for (int i = tid; i < a.NE; i += NT) {
        w_l[i] = w_tab[i];
        off_l[i] = off_tab[i];
    }
    if (MODE != FDX_PRE_RAW)
        for (int i = tid; i < LOG_TAB_N; i += NT) logt[i] = log_tab[i];
    // this wave's slice of X_sketch as MFMA A operands: A[m = type r][k = q] = X_sketch[type, bucket of slot (w, j, q)]
    double av[JW][TT];
#pragma unroll
    for (int j = 0; j < JW; ++j) {
        const int b = slot_bucket[(wave * JW + j) * 4 + q];
#pragma unroll
        for (int t = 0; t < TT; ++t) {
            const int type = t * 16 + r;
            av[j][t] = (b >= 0 && type < a.K) ? Xs[(size_t)type * a.d + b] : 0.0;
        }
    }
    const long long n_tiles = (a.n + TILE_ROWS - 1) / TILE_ROWS;

    // The wave stages (and, for the log modes, sums) the rows rr = wave + NW * k of a tile, k < RPW.  Their addresses are
    // fetched with scalar loads one tile ahead, so no staging instruction waits for a row index.
    auto load_rows = [&](long long tile, const T* (&rp)[RPW]) {
#pragma unroll
        for (int k = 0; k < RPW; ++k) {
            const long long sp = tile * TILE_ROWS + wave + NW * k;
            rp[k] = nullptr;
            if (wave + NW * k < TILE_ROWS && sp < a.n) {
                const long long row = row_map ? (long long)row_map[sp] : sp;
                rp[k] = Yp + (size_t)row * (size_t)a.ldy;
            }
        }
    };
    auto issue_stage = [&](const T* const (&rp)[RPW], int c, int buf) {
        const int gene0 = c * a.GB;
        const int bytes = (min(a.GB, a.G - gene0)) * (int)sizeof(T);
        unsigned char* base = smem + (size_t)buf * stage_bytes;
#pragma unroll
        for (int k = 0; k < RPW; ++k) {
            if (!rp[k]) continue;                                            // row past the end: stale LDS, never stored
            const unsigned char* src = reinterpret_cast<const unsigned char*>(rp[k] + gene0) + lane * 16;
            unsigned char* dst = base + (wave + NW * k) * a.RS;
            for (int o = 0; o < bytes; o += 1024)
                if (o + lane * 16 < bytes) dma16(src + o, dst + o);
        }
    };
    // scale of one row for the log modes: sum in the scatter kernels' order (per-lane partials over ascending vectors,
    // butterfly over the wave), so every sketch path sees the same bits
    auto row_scale = [&](double sum) -> double {
        if (MODE == FDX_PRE_LOG_CPM) return (1.0 / (sum + 1e-10)) * 1e4;      // y / (rowsum + 1e-10) * 1e4   (deconv.py:190)
        if (sum == 0.0) sum = 1.0;                                           // lib_size[lib_size == 0] = 1  (deconv.py:183-185)
        return 1e4 / sum;
    };
    const int nvec = a.G / PER;                                              // launch requires G % PER == 0
    auto sum_row = [&](const T* rowp) -> double {
        const V* src = reinterpret_cast<const V*>(rowp);
        double part = 0.0;
        for (int v0 = 0; v0 < nvec; v0 += 512) {
            V x[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const int v = v0 + u * 64 + lane;
                if (v < nvec) x[u] = src[v];
            }
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const int v = v0 + u * 64 + lane;
                if (v < nvec) {
#pragma unroll
                    for (int e = 0; e < PER; ++e) part += (double)x[u][e];
                }
            }
        }
        return wave_sum(part);
    };

    long long tile = blockIdx.x;
    if (tile >= n_tiles) return;
    const T* rowp[RPW];
    const T* rown[RPW];
    load_rows(tile, rowp);
    issue_stage(rowp, 0, 0);
    if (MODE != FDX_PRE_RAW) {
#pragma unroll
        for (int k = 0; k < RPW; ++k)
            if (rowp[k]) {
                const double s = row_scale(sum_row(rowp[k]));
                if (lane == 0) scales[wave + NW * k] = s;
            }
    }
    int buf = 0, par = 0;
    for (; tile < n_tiles; tile += gridDim.x) {
        const long long next_tile = tile + gridDim.x;
        const bool has_next = next_tile < n_tiles;
        load_rows(has_next ? next_tile : tile, rown);
        double acc[JW];
#pragma unroll
        for (int j = 0; j < JW; ++j) acc[j] = 0.0;
        double scale = 1.0;
        for (int c = 0; c < a.NBLK; ++c) {
            __builtin_amdgcn_s_waitcnt(0x0f70);                              // vmcnt(0): this wave's pieces have landed
            __syncthreads();                                                 // everybody's have; the other buffer is free
            if (c + 1 < a.NBLK) issue_stage(rowp, c + 1, buf ^ 1);
            else if (has_next) issue_stage(rown, 0, buf ^ 1);
            if (MODE != FDX_PRE_RAW && c == 0) scale = scales[par * TILE_ROWS + r];
            // ---- gather this block's genes: software pipeline over the flat entry stream - weight and value of the
            // current step in registers, the offset of the step after next already fetched, so a step costs one LDS
            // round trip, not two
            const unsigned char* rowb = smem + (size_t)buf * stage_bytes + r * a.RS;
            int p = ent_base[wave * (a.NBLK + 1) + c] + q;
            const unsigned long long* lens = reinterpret_cast<const unsigned long long*>(len_tab + ((size_t)wave * a.NBLK + c) * JW);
            double wv = w_l[p];
            T yv = *reinterpret_cast<const T*>(rowb + (size_t)off_l[p] * sizeof(T));
            unsigned offn = off_l[p + 4];
#pragma unroll
            for (int j = 0; j < JW; ++j) {
                const int len = (int)((lens[j >> 3] >> ((j & 7) * 8)) & 0xffULL);
                for (int t = 0; t < len; ++t) {
                    p += 4;
                    const double wn = w_l[p];
                    const T yn = *reinterpret_cast<const T*>(rowb + (size_t)offn * sizeof(T));
                    offn = off_l[p + 4];
                    double y = (double)yv;
                    if (MODE != FDX_PRE_RAW) y = tile_log1p(y * scale, logt);
                    acc[j] = fma(wv, y, acc[j]);
                    wv = wn;
                    yv = yn;
                }
            }
            // ---- log modes: row sums of the next tile (this block's share of the wave's rows)
            if (MODE != FDX_PRE_RAW && has_next) {
#pragma unroll
                for (int k = 0; k < RPW; ++k) {
                    if (k % a.NBLK != c || !rown[k]) continue;
                    const double s = row_scale(sum_row(rown[k]));
                    if (lane == 0) scales[(par ^ 1) * TILE_ROWS + wave + NW * k] = s;
                }
            }
            buf ^= 1;
        }
        par ^= 1;
#pragma unroll
        for (int k = 0; k < RPW; ++k) rowp[k] = rown[k];
        // ---- contraction: D[type, spot] += sum_q A[type, q] * B[q, spot]
        double4_t accm[TT];
#pragma unroll
        for (int t = 0; t < TT; ++t) accm[t] = double4_t{0.0, 0.0, 0.0, 0.0};
        double sq = 0.0;
#pragma unroll
        for (int j = 0; j < JW; ++j) {
            if (j < a.jw_used) {
#pragma unroll
                for (int t = 0; t < TT; ++t) accm[t] = __builtin_amdgcn_mfma_f64_16x16x4f64(av[j][t], acc[j], accm[t], 0, 0, 0);
                sq = fma(acc[j], acc[j], sq);
            }
        }
        // ---- the NW partial tiles are added in a fixed order through LDS (the buffer of the block just consumed).  With
        // 16 waves the upper half first hands its tiles to the lower half (wave w + 8 -> wave w), which halves the footprint.
        constexpr int NR = NW == 16 ? 8 : NW;                                // partial tiles that reach the final sum
        constexpr int TS = TT * 4 * 64;
        double* red = reinterpret_cast<double*>(smem + (size_t)(buf ^ 1) * stage_bytes);   // [NR][TS] + [NR][64]
        double* red_sq = red + (size_t)NR * TS;
        lds_barrier();                                                       // the last block's buffer is free
        if (NW == 16) {
            if (wave >= 8) {
#pragma unroll
                for (int t = 0; t < TT; ++t)
#pragma unroll
                    for (int rr = 0; rr < 4; ++rr) red[(size_t)(wave - 8) * TS + (t * 4 + rr) * 64 + lane] = accm[t][rr];
                red_sq[(wave - 8) * 64 + lane] = sq;
            }
            lds_barrier();
            if (wave < 8) {
#pragma unroll
                for (int t = 0; t < TT; ++t)
#pragma unroll
                    for (int rr = 0; rr < 4; ++rr) red[(size_t)wave * TS + (t * 4 + rr) * 64 + lane] += accm[t][rr];
                red_sq[wave * 64 + lane] += sq;
            }
        } else {
#pragma unroll
            for (int t = 0; t < TT; ++t)
#pragma unroll
                for (int rr = 0; rr < 4; ++rr) red[(size_t)wave * TS + (t * 4 + rr) * 64 + lane] = accm[t][rr];
            red_sq[wave * 64 + lane] = sq;
        }
        lds_barrier();
        const long long s0 = tile * TILE_ROWS;
        for (int o = tid; o < TS; o += NT) {
            double sum = 0.0;
#pragma unroll
            for (int v = 0; v < NR; ++v) sum += red[(size_t)v * TS + o];                  // fixed order: deterministic
            const int l = o & 63, tr = o >> 6;
            const int type = (tr >> 2) * 16 + (l >> 4) + 4 * (tr & 3);
            const long long sp = s0 + (l & 15);
            if (type < a.K && sp < a.n) H[(size_t)type * a.ldh + sp] = sum;
        }
        if (row_sumsq && tid < TILE_ROWS && s0 + tid < a.n) {
            double sum = 0.0;
            for (int v = 0; v < NR; ++v)
#pragma unroll
                for (int qq = 0; qq < 4; ++qq) sum += red_sq[v * 64 + qq * 16 + tid];
            row_sumsq[s0 + tid] = sum;
        }
        // the top-of-block barrier of the next tile orders these reads before the next DMA into this buffer
    }
}

// ---- host side ------------------------------------------------------------------------------------------------------

struct TilePlanDevice {
    TilePlanHost h;
    DevBuf w, off, len, ent_base, slot_bucket;
    int NW = 0, JW = 0, RS = 0;
    size_t lds = 0;
};

static const double* log_table_dev(hipStream_t st) {   // -log of every 8-bit reciprocal in [2^-15, 1], one copy per device
    static std::mutex mu;
    static double* tabs[64] = {};
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return nullptr;
    std::lock_guard<std::mutex> lock(mu);
    if (!tabs[dev]) {
        std::vector<double> t((size_t)LOG_TAB_N);
        for (int i = 0; i < LOG_TAB_N; ++i) {
            const unsigned bits = (unsigned)(LOG_TAB_BASE + i) << 16;
            float c;
            std::memcpy(&c, &bits, 4);
            t[(size_t)i] = (double)(-logl((long double)c));
        }
        double* p = nullptr;
        if (hipMalloc(&p, t.size() * sizeof(double)) != hipSuccess) return nullptr;
        if (hipMemcpyAsync(p, t.data(), t.size() * sizeof(double), hipMemcpyHostToDevice, st) != hipSuccess ||
            hipStreamSynchronize(st) != hipSuccess) {
            (void)hipFree(p);
            return nullptr;
        }
        tabs[dev] = p;
    }
    return tabs[dev];
}

static int tile_waves() {   // waves per workgroup: 16 unless FDX_TILE_WAVES=8
    const char* e = getenv("FDX_TILE_WAVES");
    return (e && atoi(e) == 8) ? 8 : 16;
}

static size_t tile_lds_bytes(int RS, int NE, int mode) {
    const size_t NEp = ((size_t)NE + 7) & ~(size_t)7;
    return 2 * (size_t)TILE_ROWS * RS + NEp * 10 + 2 * TILE_ROWS * 8 + (mode != FDX_PRE_RAW ? (size_t)LOG_TAB_N * 8 : 0);
}

// Builds (once per SketchPlan and input type) the schedule for the largest column block that fits the LDS.
static const TilePlanDevice* tile_plan_for(const SketchPlan& sp, int dtype, int mode, int K, hipStream_t st) {
    const int sz = dtype == FDX_F32 ? 4 : 8;
    const int key = (dtype == FDX_F32 ? 0 : 1) * 2 + (mode != FDX_PRE_RAW ? 1 : 0);
    if (sp.tile_tried[key]) return sp.tile[key].get();
    sp.tile_tried[key] = true;
    const bool dbg = getenv("FDX_DEBUG") != nullptr;
    if (dbg) std::fprintf(stderr, "[fdx] tile plan: G=%d d=%d K=%d scatter_ok=%d host=%zu\n", sp.G, sp.d, K, (int)sp.scatter_ok, sp.host_bucket.size());
    if (!sp.scatter_ok || sp.host_bucket.empty()) return nullptr;
    const int NW = tile_waves();
    const int TT = (K + 15) / 16;
    const int JW = NW == 16 ? 8 : 16;
    if (sp.d > 4 * NW * JW || TT > 2 || TT < 1) return nullptr;
    const size_t red_bytes = (size_t)(NW == 16 ? 8 : NW) * (TT * 4 * 64 + 64) * 8;   // the kernel's reduction area
    const int unit = 1024 / sz;                                             // genes per 1 KB piece
    const int gb_max = (int)round_up(sp.G, unit);
    std::unique_ptr<TilePlanDevice> best;
    for (int GB = gb_max; GB >= unit; GB -= unit) {
        const int RS = GB * sz + TILE_ROW_PAD;
        if ((size_t)TILE_ROWS * RS < red_bytes) break;
        // cheap bound before building: the tables hold at least G entries
        if (tile_lds_bytes(RS, sp.G, mode) > 160 * 1024) continue;
        auto cand = std::make_unique<TilePlanDevice>();
        if (!build_tile_plan(sp.host_bucket.data(), sp.host_w.data(), sp.G, sp.d, NW, JW, GB, &cand->h)) return nullptr;
        cand->NW = NW; cand->JW = JW; cand->RS = RS;
        cand->lds = tile_lds_bytes(RS, cand->h.NE, mode);
        if (dbg) std::fprintf(stderr, "[fdx] tile plan: GB=%d blocks=%d NE=%d steps=%d lds=%zu\n", GB, cand->h.NBLK, cand->h.NE, cand->h.steps, cand->lds);
        if (cand->lds > 160 * 1024) continue;
        best = std::move(cand);
        break;
    }
    if (dbg) std::fprintf(stderr, "[fdx] tile plan: %s\n", best ? "ok" : "no block size fits");
    if (!best) return nullptr;
    TilePlanDevice& t = *best;
    auto up = [&](DevBuf& b, const void* src, size_t bytes) -> int {
        FDX_TRY(b.alloc(bytes));
        FDX_HIP(hipMemcpyAsync(b.p, src, bytes, hipMemcpyHostToDevice, st));
        return 0;
    };
    std::vector<unsigned char> len_pad(t.h.len);
    len_pad.resize(t.h.len.size() + 16, 0);                                 // the kernel reads lengths 8 at a time
    if (up(t.w, t.h.w.data(), t.h.w.size() * 8) || up(t.off, t.h.off.data(), t.h.off.size() * 2) ||
        up(t.len, len_pad.data(), len_pad.size()) || up(t.ent_base, t.h.ent_base.data(), t.h.ent_base.size() * 4) ||
        up(t.slot_bucket, t.h.slot_bucket.data(), t.h.slot_bucket.size() * 4))
        return nullptr;
    if (hipStreamSynchronize(st) != hipSuccess) return nullptr;            // the host vectors may die with the plan
    sp.tile[key] = std::move(best);
    return sp.tile[key].get();
}

bool tile_sketch_ok(int dtype, long long ldy, const void* Y, int G, int d, int K, int mode, const SketchPlanDev& plan,
                    hipStream_t st) {
    if (getenv("FDX_NO_TILE") || !plan.owner) return false;
    if (dtype != FDX_F32 && dtype != FDX_F64) return false;
    if (mode != FDX_PRE_RAW && mode != FDX_PRE_LOG_CPM && mode != FDX_PRE_LOG_CPM_SPARSE) return false;
    const int sz = dtype == FDX_F32 ? 4 : 8;
    if (K <= 0 || K > 32 || G <= 0 || d <= 0) return false;
    // whole 16-byte vectors only: row starts and row lengths multiples of 16 bytes
    if (((size_t)G * sz) % 16 != 0 || ((size_t)ldy * sz) % 16 != 0 || (reinterpret_cast<uintptr_t>(Y) & 15) != 0) return false;
    return tile_plan_for(*plan.owner, dtype, mode, K, st) != nullptr;
}

struct TileLaunch {
    TileArgs a;
    const void* Y;
    const int* row_map;
    const double* Xs;
    double* H;
    double* row_sumsq;
    const double* w_tab;
    const unsigned short* off_tab;
    const unsigned char* len_tab;
    const int* ent_base;
    const int* slot_bucket;
    const double* log_tab;
};

template <typename T, int MODE, int NW, int JW>
static int launch_tile_tt(const TileLaunch& L, int TT, size_t lds, int grid, hipStream_t st) {
    const void* kern = TT == 1 ? (const void*)tile_sketch_kernel<T, MODE, NW, JW, 1> : (const void*)tile_sketch_kernel<T, MODE, NW, JW, 2>;
    if (lds > 64 * 1024) FDX_HIP(hipFuncSetAttribute(kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    void* args[] = {(void*)&L.a, (void*)&L.Y, (void*)&L.row_map, (void*)&L.Xs, (void*)&L.H, (void*)&L.row_sumsq, (void*)&L.w_tab,
                    (void*)&L.off_tab, (void*)&L.len_tab, (void*)&L.ent_base, (void*)&L.slot_bucket, (void*)&L.log_tab};
    FDX_HIP(hipLaunchKernel(kern, dim3(grid), dim3(NW * 64), args, lds, st));
    return 0;
}

template <typename T, int MODE>
static int launch_tile_nw(const TileLaunch& L, int NW, int TT, size_t lds, int grid, hipStream_t st) {
    if (NW == 16) return launch_tile_tt<T, MODE, 16, 8>(L, TT, lds, grid, st);
    return launch_tile_tt<T, MODE, 8, 16>(L, TT, lds, grid, st);
}

template <typename T>
static int launch_tile_mode(const TileLaunch& L, int mode, int NW, int TT, size_t lds, int grid, hipStream_t st) {
    switch (mode) {
        case FDX_PRE_RAW: return launch_tile_nw<T, FDX_PRE_RAW>(L, NW, TT, lds, grid, st);
        case FDX_PRE_LOG_CPM: return launch_tile_nw<T, FDX_PRE_LOG_CPM>(L, NW, TT, lds, grid, st);
        case FDX_PRE_LOG_CPM_SPARSE: return launch_tile_nw<T, FDX_PRE_LOG_CPM_SPARSE>(L, NW, TT, lds, grid, st);
        default: return fail(FDX_ERR_INVALID, "tile sketch: unknown preprocess mode");
    }
}

// H[:, 0..n) (type-major, row stride ldh) and row_sumsq[0..n) for the n spots listed by row_map (NULL = rows 0..n-1).
// Call only when tile_sketch_ok(...) holds.
int launch_tile_sketch(const void* Y, int dtype, long long ldy, const int* row_map, long long n, int G, int d, int mode,
                       const SketchPlanDev& plan, const double* Xs, int K, double* H, long long ldh, double* row_sumsq,
                       hipStream_t st) {
    if (n <= 0) return 0;
    const TilePlanDevice* t = plan.owner ? tile_plan_for(*plan.owner, dtype, mode, K, st) : nullptr;
    if (!t) return fail(FDX_ERR_INVALID, "tile sketch: no schedule for this shape");
    TileLaunch L{};
    TileArgs& a = L.a;
    a.ldy = ldy; a.n = n; a.ldh = ldh; a.G = G; a.d = d; a.K = K;
    a.NE = t->h.NE; a.GB = t->h.GB; a.NBLK = t->h.NBLK; a.RS = t->RS; a.jw_used = t->h.jw_used;
    L.Y = Y; L.row_map = row_map; L.Xs = Xs; L.H = H; L.row_sumsq = row_sumsq;
    L.w_tab = t->w.as<double>(); L.off_tab = t->off.as<unsigned short>(); L.len_tab = t->len.as<unsigned char>();
    L.ent_base = t->ent_base.as<int>(); L.slot_bucket = t->slot_bucket.as<int>();
    L.log_tab = nullptr;
    if (mode != FDX_PRE_RAW) {
        L.log_tab = log_table_dev(st);
        if (!L.log_tab) return fail(FDX_ERR_HIP, "tile sketch: log table upload failed");
    }
    const long long n_tiles = (n + TILE_ROWS - 1) / TILE_ROWS;
    const int grid = (int)std::min<long long>(n_tiles, 256);
    const int TT = (K + 15) / 16;
    if (dtype == FDX_F32) return launch_tile_mode<float>(L, mode, t->NW, TT, t->lds, grid, st);
    return launch_tile_mode<double>(L, mode, t->NW, TT, t->lds, grid, st);
}

}  // namespace fdx

// include/fdx.h: the schedule the tile kernel would use, so tests can replay it on the host (no device call).
extern "C" int fdx_tile_schedule(const int32_t* gene_bucket, const double* gene_w, int32_t G, int32_t d, int32_t NW,
                                   int32_t JW, int32_t GB, int32_t* dims_out /* NBLK, NE, steps, max_wave_steps */,
                                   int32_t* slot_bucket_out, uint8_t* len_out, int32_t* ent_base_out, double* w_out,
                                   uint16_t* off_out, int64_t cap_entries) {
    using namespace fdx;
    FDX_REQUIRE(gene_bucket && gene_w && dims_out, "fdx_tile_schedule: null argument");
    TilePlanHost h;
    FDX_REQUIRE(build_tile_plan(gene_bucket, gene_w, G, d, NW, JW, GB, &h), "fdx_tile_schedule: shape cannot be scheduled");
    dims_out[0] = h.NBLK; dims_out[1] = h.NE; dims_out[2] = h.steps; dims_out[3] = h.max_wave_steps;
    if (!slot_bucket_out) return 0;
    FDX_REQUIRE(len_out && ent_base_out && w_out && off_out && cap_entries >= h.NE, "fdx_tile_schedule: output too small");
    std::copy(h.slot_bucket.begin(), h.slot_bucket.end(), slot_bucket_out);
    std::copy(h.len.begin(), h.len.end(), len_out);
    std::copy(h.ent_base.begin(), h.ent_base.end(), ent_base_out);
    std::copy(h.w.begin(), h.w.end(), w_out);
    std::copy(h.off.begin(), h.off.end(), off_out);
    return 0;
}
