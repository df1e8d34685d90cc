// Row-register kernel: log-CPM + CountSketch + H contraction of 16 spots at a time with every row of Y read from HBM ONCE.
//
// STATUS: opt-in (FDX_ROWREG=1).  Correct (tests/test_gpu_stages.py, tests/test_host.py replay its schedule) and it halves
// the HBM traffic of the log modes (PMC: 8.3 GB instead of 18.1 GB per 1M x 2000 float32), but at 1M x 2000 x 30 it takes
// 5.3 ms where the tile kernel takes 4.1 ms: the time goes to the LDS pipe (17 k cycles per tile: scattered 8-byte stores
// of the producers at ~3-way bank conflicts, the log-table gathers, the consumers' reads) and to nine barriers per tile
// with 16 waves whose work per block differs, not to HBM.  Kept as the starting point for a single-read log path; the
// measurements are in DESIGN.md section 3.
//
// The log modes need a row's sum before its first element can be transformed (flashdeconv/core/deconv.py:177-197), so
// the tile kernel (tile_kernels.cpp) reads every row twice: plain loads for the sum, LDS-DMA for the gather.  Here the
// rows of a tile live in REGISTERS between the two uses - a float32 row of up to 2048 genes is 8 x 16 bytes per lane of
// one wave, one wave per row - and the work is split by who has what:
//
//   producer (T)  the wave holding row r transforms it, 256 genes (one 16-byte vector per lane) at a time: four log1p
//                 chains per lane (table-driven as in the tile kernel), no gather, no loop carried state.  It stores
//                 weight * log1p(y * scale) of gene g where the consumer will look for it: the "slot image" of the block
//                 (tile_plan.h: RowregPlanHost), and zeroes its share of the lockstep padding.
//   consumer (G)  lane (r, q) of wave w owns, as in the tile kernel, spot r and the buckets of the slots (w, j, q).  A step
//                 is one conflict-free ds_read_b64 of the slot image and one add: no weight, no offset table, no address
//                 arithmetic beyond the running line pointer.  After the last block the bucket sums are the B operands of
//                 v_mfma_f64_16x16x4_f64 against the wave's slice of X_sketch (core/solver.py:205-223), fetched from a
//                 pre-arranged copy in L2 one type tile at a time.
//
// Two slot images alternate: between two barriers a wave gathers block c from one and transforms block c + 1 into the
// other, and half of the waves of a SIMD do the two in the opposite order, so that a latency-bound gather runs beside an
// ALU-bound transform.  The registers of block c are refilled with the NEXT tile's block c as soon as it has been
// transformed; the last refill is issued one block before the tile ends and has the rest of the tile, the MFMAs and the
// reduction to land before the next tile's row sum needs it.  Rows with an element outside the fast range of the
// table-driven log1p (negative, NaN, Inf) are left to the tile kernel through a redo list.
//
// Row sums: per-lane partials over ascending vectors as in the scatter kernels, then a DPP reduction (rr_wave_sum).
#include <algorithm>
#include <cstdlib>
#include <memory>
#include <mutex>
#include <vector>

#include "fdx_internal.h"
#include "fdx_kernels.h"
#include "sketch_plan.h"
#include "tile_device.h"
#include "tile_plan.h"

namespace fdx {

constexpr int RR_NW = 16;                 // waves per workgroup, one row of the tile each
constexpr int RR_JW = 8;                  // bucket groups per wave: 16 x 8 x 4 = 512 buckets
constexpr int RR_NQ = 8;                  // 16-byte vectors per lane and row: 8 x 64 x 4 = 2048 genes

// Wave reductions without the LDS crossbar: four butterfly stages inside each row of 16 lanes by DPP (xor 1, 2 as quad
// permutations; 4 and 8 as half-row / row mirrors, which pair the same partial sums once the quads are uniform), then the
// four row results are read out and added in a fixed order.  (Not the association order of device_math.h's wave_sum: the
// scale of a row differs from the scatter kernels' in the last bit or so; every path is deterministic by itself.)
template <int CTRL> __device__ __forceinline__ double rr_dpp(double v) {
    const int lo = __builtin_amdgcn_mov_dpp(__double2loint(v), CTRL, 0xF, 0xF, true);
    const int hi = __builtin_amdgcn_mov_dpp(__double2hiint(v), CTRL, 0xF, 0xF, true);
    return __hiloint2double(hi, lo);
}
template <int CTRL> __device__ __forceinline__ float rr_dpp(float v) {
    return __int_as_float(__builtin_amdgcn_mov_dpp(__float_as_int(v), CTRL, 0xF, 0xF, true));
}
__device__ __forceinline__ double rr_lane(double v, int l) {
    return __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(v), l), __builtin_amdgcn_readlane(__double2loint(v), l));
}
__device__ __forceinline__ float rr_lanef(float v, int l) { return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), l)); }
__device__ __forceinline__ double rr_wave_sum(double v) {
    v += rr_dpp<0xB1>(v);                                                   // quad_perm [1, 0, 3, 2]
    v += rr_dpp<0x4E>(v);                                                   // quad_perm [2, 3, 0, 1]
    v += rr_dpp<0x141>(v);                                                  // row_half_mirror
    v += rr_dpp<0x140>(v);                                                  // row_mirror
    return (rr_lane(v, 0) + rr_lane(v, 16)) + (rr_lane(v, 32) + rr_lane(v, 48));
}
__device__ __forceinline__ float rr_wave_max(float v) {
    v = fmaxf(v, rr_dpp<0xB1>(v));
    v = fmaxf(v, rr_dpp<0x4E>(v));
    v = fmaxf(v, rr_dpp<0x141>(v));
    v = fmaxf(v, rr_dpp<0x140>(v));
    return fmaxf(fmaxf(rr_lanef(v, 0), rr_lanef(v, 16)), fmaxf(rr_lanef(v, 32), rr_lanef(v, 48)));
}

typedef float float4_t __attribute__((ext_vector_type(4)));
typedef double double2_t __attribute__((ext_vector_type(2)));
typedef unsigned uint4_t __attribute__((ext_vector_type(4)));

struct RowregArgs {
    long long ldy, n, ldh;
    int G, d, K;
    int NBLK;                             // column blocks of 256 genes
    int SB;                               // bytes of one slot image (incl. its dump line), multiple of 128
    int n_pad;                            // entries of the pad list
};

template <int MODE, int TT>
__global__ __launch_bounds__(RR_NW * 64, RR_NW / 4) void rowreg_sketch_kernel(
    const RowregArgs a, const float* __restrict__ Yp, const int* __restrict__ row_map, const double* __restrict__ XA,
    double* __restrict__ H, double* __restrict__ row_sumsq, const double* __restrict__ gene_w,
    const unsigned* __restrict__ gene_ent, const unsigned* __restrict__ pad_line, const int* __restrict__ blk_tab,
    const int* __restrict__ slot_bucket, const double* __restrict__ log_tab, int* __restrict__ redo_flag,
    int* __restrict__ redo_list, int* __restrict__ redo_count) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    typedef float4_t V;
    constexpr int NW = RR_NW, JW = RR_JW, NQ = RR_NQ, NT = NW * 64;
    constexpr int NR = NW / 2;                                             // partial tiles that reach the final sum
    constexpr int TS = TT * 4 * 64;
    static_assert(NW == TILE_ROWS, "one row of the tile per wave");
    const int tid = threadIdx.x;
    // The lane number is "re-read" at the top of every tile (relane below): what the kernel derives from it - table and
    // image addresses of eight blocks - is loop invariant, and hoisted out of the tile loop it would occupy (and spill)
    // registers that the row pieces need; each is one add away from the lane number where it is used.
    int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    int r = lane & 15, q = lane >> 4;
    const int NG = a.NBLK * 256;
    // LDS: two slot images, the per-gene tables, the pad list; the log table at its fixed place
    double* wv_l = reinterpret_cast<double*>(smem + 2 * (size_t)a.SB);
    unsigned* ent_l = reinterpret_cast<unsigned*>(wv_l + NG);
    unsigned* pad_l = ent_l + NG;
    double* logt = reinterpret_cast<double*>(smem + LOG_TAB_LDS);
    for (int i = tid; i < NG; i += NT) {
        wv_l[i] = gene_w[i];
        ent_l[i] = gene_ent[i];
    }
    for (int i = tid; i < a.n_pad; i += NT) pad_l[i] = pad_line[i];
    for (int i = tid; i < LOG_TAB_N; i += NT) logt[i] = log_tab[i];
    const long long n_tiles = (a.n + TILE_ROWS - 1) / TILE_ROWS;
    long long tile = blockIdx.x;
    if (tile >= n_tiles) return;

    // This wave's rows of the block table (8 blocks x 8 ints) in ONE register, lane 8 c + k holding entry k of block c, read
    // with v_readlane: a scalar load per block would put its latency at the head of every gather and every transform.
    static_assert(NQ * 8 == 64, "the wave's block table is one register");
    const int bt_reg = (lane >> 3) < a.NBLK ? blk_tab[((size_t)(lane >> 3) * NW + wave) * 8 + (lane & 7)] : 0;
    auto bt_get = [&](int c, int k) -> int { return __builtin_amdgcn_readlane(bt_reg, c * 8 + k); };
    // consumer: byte position of this lane's row inside a line, by group class j & 3 (see RowregPlanHost)
    unsigned Aq[4];
    int lane16 = lane * 16;
    auto relane = [&]() {
        asm volatile("" : "+v"(lane));
        r = lane & 15;
        q = lane >> 4;
        lane16 = lane * 16;
#pragma unroll
        for (int i = 0; i < 4; ++i) Aq[i] = (unsigned)q * 128u + (unsigned)((r + 4 * i + q) & 15) * 8u;
    };
    relane();
    // producer: row `wave` of the tile
    const unsigned r8 = (unsigned)wave * 8u;
    const int nvec = a.G / 4;                                               // launch requires G % 4 == 0
    const bool stagger = ((wave >> 2) & 1) != 0;                            // waves w, w + 4, w + 8, w + 12 share a SIMD

    // The row pieces are separate variables, not an array: as an array they become one 32-register value to the compiler,
    // which then copies all of it around every refill.
    static_assert(NQ == 8, "the row pieces are spelled out");
    const V vzero = V{0.f, 0.f, 0.f, 0.f};
    V x0 = vzero, x1 = vzero, x2 = vzero, x3 = vzero, x4 = vzero, x5 = vzero, x6 = vzero, x7 = vzero;

    auto row_ptr = [&](long long t) -> const float* {
        const long long sp = t * TILE_ROWS + wave;
        if (t >= n_tiles || sp >= a.n) return nullptr;
        const long long row = row_map ? (long long)row_map[sp] : sp;
        return Yp + (size_t)row * (size_t)a.ldy;
    };
    auto load_quad = [&](const float* pr, int u, V& v) {                    // piece u: bytes [1024 u, 1024 u + 1024) of the row
        if (pr && lane < nvec - u * 64)
            v = __builtin_nontemporal_load(reinterpret_cast<const V*>(reinterpret_cast<const char*>(pr) + u * 1024 + lane16));
    };
    // sum, scale and fast-range flag of the row held in the pieces (lanes past the row hold zeros, which change nothing)
    auto row_scale = [&](V p0, V p1, V p2, V p3, V p4, V p5, V p6, V p7, double& scale, bool& ok) {
        double p = 0.0;
        float mx = 0.f;
        unsigned sg = 0u;
        auto piece = [&](V x) {
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                p += (double)x[e];
                mx = fmaxf(mx, x[e]);
                sg |= __float_as_uint(x[e]);
            }
        };
        // fenced: the conversions of all 32 elements hoisted above the chain of adds would take the registers of the pieces
#define FDX_RR_FENCE __builtin_amdgcn_sched_barrier(0);
        piece(p0); FDX_RR_FENCE piece(p1); FDX_RR_FENCE piece(p2); FDX_RR_FENCE piece(p3); FDX_RR_FENCE
        piece(p4); FDX_RR_FENCE piece(p5); FDX_RR_FENCE piece(p6); FDX_RR_FENCE piece(p7); FDX_RR_FENCE
#undef FDX_RR_FENCE
        const double s = tile_row_scale<MODE>(rr_wave_sum(p));
        // NaN / Inf anywhere in the row makes the sum, hence the scale, NaN or 0 * Inf: the test fails
        ok = (double)rr_wave_max(mx) * s < 32000.0 && s > 1e-14 && !__any((int)sg < 0);
        scale = s;
    };

    double scale = 1.0;
    bool ok = true;
    // A row with an element outside the fast range of the table-driven log1p (negative, NaN, Inf) is not transformed here:
    // the general function would cost every transform site its registers.  The tile is put on the redo list instead and
    // the tile kernel, launched behind this one, recomputes it (whatever this kernel stores for the tile is overwritten).
    auto flag_redo = [&](long long t, bool ok_) {
        if (!ok_ && lane == 0 && atomicExch(&redo_flag[t], 1) == 0) redo_list[atomicAdd(redo_count, 1)] = (int)t;
    };

    // ---- producer: block c of the row -> slot image at bufoff, plus this wave's share of the block's pad lines
    auto transform = [&](const V y, int c, unsigned bufoff) {
        const int v = c * 64 + lane;
        if (__builtin_expect(ok, 1)) {
            // log1p (tile_device.h: tile_log1p_core) of the four elements, written stage by stage over the four with the
            // stages fenced: left to itself the compiler finishes one element before it starts the next (fewest registers),
            // and every element then waits out its own table read and its own chain of dependent f64 operations.  The
            // gene tables (weight, place in the image) are read with the log table, behind the argument reduction.
            typedef const double __attribute__((address_space(3))) * lds_cdouble_p;
            const double scale_s = scale * FDX_LOG_DOWN;                      // tile_device.h: the reciprocal is formed of (1 + x) * 2^-65
            const float sf = (float)scale_s;
            float rc[4];
            double rr[4], tt[4], pp[4];
            unsigned ti[4];
#pragma unroll
            for (int e = 0; e < 4; ++e) rc[e] = __builtin_amdgcn_rcpf(fmaf(y[e], sf, FDX_LOG_DOWN_F));
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const unsigned bits = (__float_as_uint(rc[e]) + LOG_TAB_ROUND) & LOG_TAB_MASK;   // reciprocal of 1 + x, rounded to the table's bits
                const double inv = (double)__uint_as_float(bits);
                ti[e] = (bits >> (LOG_TAB_SHIFT - 3)) + (unsigned)(LOG_TAB_LDS - LOG_TAB_BASE * 8);
                rr[e] = fma((double)y[e] * scale_s, inv, fma(FDX_LOG_DOWN, inv, -1.0));      // (1 + x) * c - 1 with one rounding
            }
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int e = 0; e < 4; ++e) tt[e] = *(lds_cdouble_p)(size_t)ti[e];
            const double2_t w01 = reinterpret_cast<const double2_t*>(wv_l)[2 * v], w23 = reinterpret_cast<const double2_t*>(wv_l)[2 * v + 1];
            const uint4_t en = reinterpret_cast<const uint4_t*>(ent_l)[v];
#pragma unroll
            for (int e = 0; e < 4; ++e) pp[e] = LOG_TAB_MB == 7 ? fma3(rr[e], -1.0 / 6.0, 0.2) : fma3(rr[e], fma3(rr[e], 1.0 / 7.0, -1.0 / 6.0), 0.2);
#pragma unroll
            for (int e = 0; e < 4; ++e) pp[e] = fma(rr[e], pp[e], -0.25);
#pragma unroll
            for (int e = 0; e < 4; ++e) pp[e] = fma3(rr[e], pp[e], 1.0 / 3.0);
#pragma unroll
            for (int e = 0; e < 4; ++e) pp[e] = fma(rr[e], pp[e], -0.5);
#pragma unroll
            for (int e = 0; e < 4; ++e) pp[e] = fma(rr[e], pp[e], 1.0);
            __builtin_amdgcn_sched_barrier(0);
            const double w[4] = {w01[0], w01[1], w23[0], w23[1]};
#pragma unroll
            for (int e = 0; e < 4; ++e) pp[e] = fma(rr[e], pp[e], tt[e]) * w[e];
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const unsigned ad = ((en[e] + r8) & 127u) | (en[e] & ~127u);
                *reinterpret_cast<double*>(smem + bufoff + ad) = pp[e];
            }
        }
        const int p0 = bt_get(c, 1), rounds = bt_get(c, 2);
        for (int m = 0; m < rounds; ++m) {
            const unsigned line = pad_l[p0 + 4 * m + q];
            *reinterpret_cast<double*>(smem + bufoff + line + (unsigned)r * 8u) = 0.0;
        }
    };

    double acc[JW];
    // ---- consumer: block c from the slot image at bufoff.  The group lengths are scalars, so every group's place in the
    // image is known up front: the FIRST step of all groups is read at once (an empty group reads something it ignores),
    // one LDS round trip for the lot; the further steps of the longer groups follow one by one.  (Reading two steps of
    // every group ahead costs more in the LDS pipe, which is the busiest unit of this kernel, than it saves in latency.)
    auto gather = [&](int c, unsigned bufoff) {
        const unsigned p0 = bufoff + (unsigned)bt_get(c, 0);
        const unsigned lw[4] = {(unsigned)bt_get(c, 4), (unsigned)bt_get(c, 5), (unsigned)bt_get(c, 6), (unsigned)bt_get(c, 7)};
        int len[JW];
        unsigned pj[JW];
        unsigned p = p0;
#pragma unroll
        for (int j = 0; j < JW; ++j) {
            len[j] = (int)((lw[j >> 2] >> ((j & 3) * 8)) & 0xffu);
            pj[j] = p;
            p += 512u * (unsigned)len[j];
        }
        double f[JW];
#pragma unroll
        for (int j = 0; j < JW; ++j) f[j] = *reinterpret_cast<const double*>(smem + Aq[j & 3] + pj[j]);
#pragma unroll
        for (int j = 0; j < JW; ++j) {
            if (len[j] > 0) acc[j] += f[j];
#pragma unroll 1
            for (int t = 1; t < len[j]; ++t) acc[j] += *reinterpret_cast<const double*>(smem + Aq[j & 3] + pj[j] + 512u * (unsigned)t);
        }
    };

    // ---- prologue: the first tile's row, its scale, block 0
    const float* pr = row_ptr(tile);
#define FDX_RR_LOAD(U) load_quad(pr, U, x##U);
    FDX_RR_LOAD(0) FDX_RR_LOAD(1) FDX_RR_LOAD(2) FDX_RR_LOAD(3) FDX_RR_LOAD(4) FDX_RR_LOAD(5) FDX_RR_LOAD(6) FDX_RR_LOAD(7)
#undef FDX_RR_LOAD
    row_scale(x0, x1, x2, x3, x4, x5, x6, x7, scale, ok);
    flag_redo(tile, ok);
    __syncthreads();                                                        // the tables are in the LDS
    transform(x0, 0, 0u);

    for (;;) {
        relane();
        const float* pn = row_ptr(tile + gridDim.x);
        load_quad(pn, 0, x0);                                               // block 0 has been transformed: refill
#pragma unroll
        for (int j = 0; j < JW; ++j) acc[j] = 0.0;
        // block C is gathered and block U = C + 1 transformed between two barriers; spelled out: the row pieces are registers
#define FDX_RR_BLOCK(C, U)                                                                                                   \
        if (U < a.NBLK) {                                                                                                    \
            lds_barrier();   /* block C is complete in its image; the other image is free */                                 \
            const unsigned bo = (unsigned)(C & 1) * (unsigned)a.SB, bn = (unsigned)(U & 1) * (unsigned)a.SB;                 \
            if (stagger) gather(C, bo);                                                                      \
            transform(x##U, U, bn);                                                                        \
            load_quad(pn, U, x##U);   /* refill with the next tile's piece */                              \
            if (!stagger) gather(C, bo);                                                                     \
        }
        FDX_RR_BLOCK(0, 1) FDX_RR_BLOCK(1, 2) FDX_RR_BLOCK(2, 3) FDX_RR_BLOCK(3, 4) FDX_RR_BLOCK(4, 5) FDX_RR_BLOCK(5, 6)
        FDX_RR_BLOCK(6, 7)
#undef FDX_RR_BLOCK
        // this wave's slice of X_sketch as MFMA A operands, A[m = type r][k = q] = X_sketch[type, bucket of slot (w, j, q)],
        // from the launch's pre-arranged copy (one coalesced 512-byte read per operand): the first type tile's now, the
        // second's behind the first's MFMAs - 128 registers hold the row, the bucket sums and one tile's operands, not two
        const double* xw = XA + ((size_t)wave * JW * 2) * 64 + lane;
        double av[JW];
#pragma unroll
        for (int j = 0; j < JW; ++j) av[j] = xw[(size_t)(j * 2) * 64];
        lds_barrier();
        gather(a.NBLK - 1, (unsigned)((a.NBLK - 1) & 1) * (unsigned)a.SB);
        // The matrix pipe takes JW x TT x 64 cycles per wave for the MFMAs; the vector ALU work of the tile boundary - the
        // next tile's row sum - runs beside it.  Half of
        // the waves of a SIMD do the MFMAs first, the others last, so the two pipes work side by side.
        double nscale = 1.0;
        bool nok = true;
        double4_t accm[TT];                                                 // declared here: not live through the tile
        double sq = 0.0;
#pragma unroll
        for (int t = 0; t < TT; ++t) accm[t] = double4_t{0.0, 0.0, 0.0, 0.0};
#pragma unroll 1
        for (int ph = 0; ph < 2; ++ph) {
            if ((ph == 0) == stagger) {
                row_scale(x0, x1, x2, x3, x4, x5, x6, x7, nscale, nok);    // the next tile's row has had the tile to land
            } else {
#pragma unroll
                for (int t = 0; t < TT; ++t) {
                    double an[JW];
                    if (t + 1 < TT) {
#pragma unroll
                        for (int j = 0; j < JW; ++j) an[j] = xw[(size_t)(j * 2 + t + 1) * 64];
                    }
#pragma unroll
                    for (int j = 0; j < JW; ++j) accm[t] = __builtin_amdgcn_mfma_f64_16x16x4f64(av[j], acc[j], accm[t], 0, 0, 0);
                    if (t + 1 < TT) {
#pragma unroll
                        for (int j = 0; j < JW; ++j) av[j] = an[j];
                    }
                }
#pragma unroll
                for (int j = 0; j < JW; ++j) sq = fma(acc[j], acc[j], sq);
            }
        }
        // ---- the partial type tiles are added in a fixed order through LDS (second image: the upper half of the waves
        // hands to the lower half first) and stored
        double* red = reinterpret_cast<double*>(smem + (size_t)a.SB);       // [NR][TS] + [NW][16]
        double* red_sq = red + (size_t)NR * TS;
        sq += __shfl_xor(sq, 16, 64);
        sq += __shfl_xor(sq, 32, 64);                                       // the row's four lane classes
        lds_barrier();                                                      // every wave is done with the last block's image
        if (wave >= NR) {
#pragma unroll
            for (int t = 0; t < TT; ++t)
#pragma unroll
                for (int rr = 0; rr < 4; ++rr) red[(size_t)(wave - NR) * TS + (t * 4 + rr) * 64 + lane] = accm[t][rr];
        }
        if (lane < TILE_ROWS) red_sq[wave * TILE_ROWS + lane] = sq;
        lds_barrier();
        if (wave < NR) {
#pragma unroll
            for (int t = 0; t < TT; ++t)
#pragma unroll
                for (int rr = 0; rr < 4; ++rr) red[(size_t)wave * TS + (t * 4 + rr) * 64 + lane] += accm[t][rr];
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
#pragma unroll
            for (int v = 0; v < NW; ++v) sum += red_sq[v * TILE_ROWS + tid];
            row_sumsq[s0 + tid] = sum;
        }
        tile += gridDim.x;
        if (tile >= n_tiles) break;
        scale = nscale;
        ok = nok;
        flag_redo(tile, ok);
        // block 0 of the new tile goes to the first image, which nobody reads after the barrier that follows the last
        // gather; the reduction area (second image) is next written after the new tile's first barrier
        transform(x0, 0, 0u);
    }
}

// X_sketch rearranged into the MFMA A operands of the row-register kernel: XA[((w * JW + j) * 2 + t) * 64 + lane] =
// X_sketch[type = 16 t + (lane & 15), bucket of slot (w, j, lane >> 4)], 0 where there is no such type or bucket.
__global__ void rowreg_xa_kernel(const double* __restrict__ Xs, const int* __restrict__ slot_bucket, int K, int d,
                                 double* __restrict__ XA) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= RR_NW * RR_JW * 2 * 64) return;
    const int lane = i & 63, t = (i >> 6) & 1, wj = i >> 7;
    const int b = slot_bucket[wj * 4 + (lane >> 4)];
    const int type = t * 16 + (lane & 15);
    XA[i] = (b >= 0 && type < K) ? Xs[(size_t)type * d + b] : 0.0;
}

// ---- host side ------------------------------------------------------------------------------------------------------

struct RowregPlanDevice {
    RowregPlanHost h;
    DevBuf gene_w, gene_ent, pad_line, blk_tab, slot_bucket;
    int SB = 0;
    size_t lds = 0;
};

static bool rowreg_shape_ok(int dtype, long long ldy, const void* Y, int G, int d, int K, int mode) {
    if (!getenv("FDX_ROWREG")) return false;                                // opt-in: see the header
    if (dtype != FDX_F32) return false;
    if (mode != FDX_PRE_LOG_CPM && mode != FDX_PRE_LOG_CPM_SPARSE) return false;
    if (K <= 0 || K > 32 || d <= 0 || d > 4 * RR_NW * RR_JW || G <= 0 || G > RR_NQ * 256) return false;
    if (G % 4 != 0 || ldy % 4 != 0 || (reinterpret_cast<uintptr_t>(Y) & 15) != 0) return false;
    return true;
}

static const RowregPlanDevice* rowreg_plan_for(const SketchPlan& sp, int K, hipStream_t st) {
    std::lock_guard<std::mutex> lock(sp.tile_mu);
    if (sp.rowreg_tried) return sp.rowreg.get();
    sp.rowreg_tried = true;
    if (!sp.scatter_ok || sp.host_bucket.empty()) return nullptr;
    auto cand = std::make_unique<RowregPlanDevice>();
    if (!build_rowreg_plan(sp.host_bucket.data(), sp.host_w.data(), sp.G, sp.d, RR_NW, RR_JW, &cand->h)) return nullptr;
    RowregPlanDevice& t = *cand;
    const size_t red_bytes = ((size_t)(RR_NW / 2) * 2 * 4 * 64 + RR_NW * 16) * 8;   // reduction area of the widest instantiation
    t.SB = (int)round_up(std::max<size_t>((size_t)t.h.Smax * 512 + 128, red_bytes), 128);
    const size_t below = 2 * (size_t)t.SB + (size_t)t.h.NBLK * 256 * 12 + t.h.pad_line.size() * 4;
    if (getenv("FDX_DEBUG"))
        std::fprintf(stderr, "[fdx] rowreg plan: G=%d d=%d blocks=%d steps=%d Smax=%d image=%d B pads=%zu lds below table=%zu\n", sp.G,
                     sp.d, t.h.NBLK, t.h.steps, t.h.Smax, t.SB, t.h.pad_line.size(), below);
    if (below > (size_t)LOG_TAB_LDS) return nullptr;
    t.lds = (size_t)160 * 1024;
    auto up = [&](DevBuf& b, const void* src, size_t bytes) -> int {
        FDX_TRY(b.alloc(bytes));
        FDX_HIP(hipMemcpyAsync(b.p, src, bytes, hipMemcpyHostToDevice, st));
        return 0;
    };
    if (up(t.gene_w, t.h.gene_w.data(), t.h.gene_w.size() * 8) || up(t.gene_ent, t.h.gene_ent.data(), t.h.gene_ent.size() * 4) ||
        up(t.pad_line, t.h.pad_line.data(), t.h.pad_line.size() * 4) || up(t.blk_tab, t.h.blk_tab.data(), t.h.blk_tab.size() * 4) ||
        up(t.slot_bucket, t.h.slot_bucket.data(), t.h.slot_bucket.size() * 4))
        return nullptr;
    if (hipStreamSynchronize(st) != hipSuccess) return nullptr;            // the host vectors may die with the plan
    (void)K;
    sp.rowreg = std::move(cand);
    return sp.rowreg.get();
}

bool rowreg_sketch_ok(int dtype, long long ldy, const void* Y, int G, int d, int K, int mode, const SketchPlanDev& plan,
                      hipStream_t st) {
    if (!plan.owner || !rowreg_shape_ok(dtype, ldy, Y, G, d, K, mode)) return false;
    // the tile kernel must be there for the redo list
    if (!tile_sketch_ok(dtype, ldy, Y, G, d, K, mode, plan, st)) return false;
    return rowreg_plan_for(*plan.owner, K, st) != nullptr;
}

// H[:, 0..n) (type-major, row stride ldh) and row_sumsq[0..n) for the n spots listed by row_map (NULL = rows 0..n-1).
// Call only when rowreg_sketch_ok(...) holds.
int launch_rowreg_sketch(const void* Y, long long ldy, const int* row_map, long long n, int G, int d, int mode,
                         const SketchPlanDev& plan, const double* Xs, int K, double* H, long long ldh, double* row_sumsq,
                         hipStream_t st) {
    if (n <= 0) return 0;
    const RowregPlanDevice* t = plan.owner ? rowreg_plan_for(*plan.owner, K, st) : nullptr;
    if (!t) return fail(FDX_ERR_INVALID, "rowreg sketch: no schedule for this shape");
    RowregArgs a{};
    a.ldy = ldy; a.n = n; a.ldh = ldh; a.G = G; a.d = d; a.K = K;
    a.NBLK = t->h.NBLK; a.SB = t->SB; a.n_pad = (int)t->h.pad_line.size();
    const double* log_tab = log_table_dev(st);
    if (!log_tab) return fail(FDX_ERR_HIP, "rowreg sketch: log table upload failed");
    const float* Yf = (const float*)Y;
    const double* gw = t->gene_w.as<double>();
    const unsigned* ge = t->gene_ent.as<unsigned>();
    const unsigned* pl = t->pad_line.as<unsigned>();
    const int* bt = t->blk_tab.as<int>();
    const int* sb = t->slot_bucket.as<int>();
    const long long n_tiles = (n + TILE_ROWS - 1) / TILE_ROWS;
    const int grid = (int)std::min<long long>(n_tiles, 256);
    const int TT = (K + 15) / 16;
    const void* kern = nullptr;
    if (mode == FDX_PRE_LOG_CPM)
        kern = TT == 1 ? (const void*)rowreg_sketch_kernel<FDX_PRE_LOG_CPM, 1> : (const void*)rowreg_sketch_kernel<FDX_PRE_LOG_CPM, 2>;
    else
        kern = TT == 1 ? (const void*)rowreg_sketch_kernel<FDX_PRE_LOG_CPM_SPARSE, 1>
                       : (const void*)rowreg_sketch_kernel<FDX_PRE_LOG_CPM_SPARSE, 2>;
    // X_sketch as A operands
    DevBuf xa;
    FDX_TRY(xa.alloc((size_t)RR_NW * RR_JW * 2 * 64 * sizeof(double)));
    double* XA = xa.as<double>();
    hipLaunchKernelGGL(rowreg_xa_kernel, dim3(RR_NW * RR_JW * 2 * 64 / 256), dim3(256), 0, st, Xs, sb, K, d, XA);
    FDX_CHECK_LAUNCH();
    // redo list: [count, pad] [flag per tile] [list]
    DevBuf redo;
    FDX_TRY(redo.alloc((size_t)(2 + 2 * n_tiles) * sizeof(int)));
    int* redo_count = redo.as<int>();
    int* redo_flag = redo_count + 2;
    int* redo_list = redo_flag + n_tiles;
    FDX_HIP(hipMemsetAsync(redo_count, 0, (size_t)(2 + n_tiles) * sizeof(int), st));
    FDX_HIP(hipFuncSetAttribute(kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)t->lds));
    void* args[] = {(void*)&a, (void*)&Yf, (void*)&row_map, (void*)&XA, (void*)&H, (void*)&row_sumsq, (void*)&gw,
                    (void*)&ge, (void*)&pl, (void*)&bt, (void*)&sb, (void*)&log_tab, (void*)&redo_flag, (void*)&redo_list,
                    (void*)&redo_count};
    FDX_HIP(hipLaunchKernel(kern, dim3(grid), dim3(RR_NW * 64), args, t->lds, st));
    // the tiles left over (usually none: the workgroups of this launch look at the count and leave)
    return launch_tile_sketch(Y, FDX_F32, ldy, row_map, n, G, d, mode, plan, Xs, K, H, ldh, row_sumsq, st, redo_list, redo_count);
}

}  // namespace fdx

// include/fdx.h: the slot-image schedule of the row-register kernel, so tests can replay it on the host (no device call).
extern "C" int fdx_rowreg_schedule(const int32_t* gene_bucket, const double* gene_w, int32_t G, int32_t d,
                                     int32_t* dims_out /* NBLK, Smax, steps, n_pad */, int32_t* slot_bucket_out,
                                     int32_t* blk_tab_out, double* gene_w_out, uint32_t* gene_ent_out, uint32_t* pad_line_out,
                                     int64_t cap_pad) {
    using namespace fdx;
    FDX_REQUIRE(gene_bucket && gene_w && dims_out, "fdx_rowreg_schedule: null argument");
    RowregPlanHost h;
    FDX_REQUIRE(build_rowreg_plan(gene_bucket, gene_w, G, d, RR_NW, RR_JW, &h), "fdx_rowreg_schedule: shape cannot be scheduled");
    dims_out[0] = h.NBLK; dims_out[1] = h.Smax; dims_out[2] = h.steps; dims_out[3] = (int32_t)h.pad_line.size();
    if (!slot_bucket_out) return 0;
    FDX_REQUIRE(blk_tab_out && gene_w_out && gene_ent_out && pad_line_out && cap_pad >= (int64_t)h.pad_line.size(),
                "fdx_rowreg_schedule: output too small");
    std::copy(h.slot_bucket.begin(), h.slot_bucket.end(), slot_bucket_out);
    std::copy(h.blk_tab.begin(), h.blk_tab.end(), blk_tab_out);
    std::copy(h.gene_w.begin(), h.gene_w.end(), gene_w_out);
    std::copy(h.gene_ent.begin(), h.gene_ent.end(), gene_ent_out);
    std::copy(h.pad_line.begin(), h.pad_line.end(), pad_line_out);
    return 0;
}
