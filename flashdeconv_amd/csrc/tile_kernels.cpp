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
//               the DMA) while the current tile is consumed.  log1p is table driven: 1 + x is reduced by a 7-bit
//               reciprocal (v_rcp_f32) to 1 + r with |r| <= 2^-7, log1p(x) = T[reciprocal] + r - r^2/2 + ... + r^7/7
//               (~21 instructions instead of ~45; < 3 ulp; tile_device.h).
#include "fdx_env.h"
#include <algorithm>
#include <cmath>
#include <cstdlib>
#include <memory>
#include <mutex>
#include <type_traits>

#include "device_math.h"
#include "fdx_internal.h"
#include "fdx_kernels.h"
#include "sketch_plan.h"
#include "tile_device.h"
#include "tile_plan.h"

namespace fdx {

constexpr int TILE_ROW_PAD = 16;          // bytes between staged rows: a 16-byte shift keeps the DMA destination aligned

// Scalars of a launch.  The arrays are separate __restrict__ kernel parameters: only then may the compiler fetch the
// wave-uniform ones (row_map, ent_base, len_tab) with scalar loads.  As vector loads they would sit in vmcnt behind the
// LDS-DMA pieces in flight, and every use would wait for the next block to land - no overlap left.
struct TileArgs {
    long long ldy, n, ldh;
    int G, d, K;
    int NE, GB, NBLK, RS, jw_used;
    int NST;   // stage buffers: 2, or 3 (raw mode with loader waves: two blocks in flight while one is consumed)
    int WB;    // WG form: bytes of a block's weight table at the head of every stage buffer ((GB + 1) doubles, 16-byte rounded)
    int dbg;   // experiment builds (-DFDX_TILE_EXPERIMENT): phases to skip, timing only
    int NSP, GROW;   // flat form (FF, tile_plan.h: TileFlatHost): off_tab rows of NSP offsets per (wave, block, lane class), len_tab rows of GROW bytes
};

template <typename T> struct TileVec;
template <> struct TileVec<float> { typedef float type __attribute__((ext_vector_type(4))); };
template <> struct TileVec<double> { typedef double type __attribute__((ext_vector_type(2))); };

// One 1 KB piece of a staged row: lane l copies 16 bytes from src to lds_base + 16 * l.
__device__ __forceinline__ void dma16(const void* src, unsigned char* lds_base) {
    __builtin_amdgcn_global_load_lds((const void __attribute__((address_space(1)))*)src,
                                     (void __attribute__((address_space(3)))*)lds_base, 16, 0, 0);
}

// s_waitcnt vmcnt(n) for a wave-uniform n known only at run time (the instruction takes an immediate): waits until at most
// min(n, 24) of this wave's vector-memory operations are outstanding - the OLDER ones have completed (loads return in order)
__device__ __forceinline__ void wait_vmcnt_le(int n) {
#define FDX_VMCNT(N) case N: __builtin_amdgcn_s_waitcnt(0x0f70 | ((N) & 15) | (((N) >> 4) << 14)); break;
    switch (n) {
        FDX_VMCNT(0) FDX_VMCNT(1) FDX_VMCNT(2) FDX_VMCNT(3) FDX_VMCNT(4) FDX_VMCNT(5) FDX_VMCNT(6) FDX_VMCNT(7)
        FDX_VMCNT(8) FDX_VMCNT(9) FDX_VMCNT(10) FDX_VMCNT(11) FDX_VMCNT(12) FDX_VMCNT(13) FDX_VMCNT(14) FDX_VMCNT(15)
        FDX_VMCNT(16) FDX_VMCNT(17) FDX_VMCNT(18) FDX_VMCNT(19) FDX_VMCNT(20) FDX_VMCNT(21) FDX_VMCNT(22) FDX_VMCNT(23)
        default: __builtin_amdgcn_s_waitcnt(0x0f70 | (24 & 15) | ((24 >> 4) << 14)); break;
    }
#undef FDX_VMCNT
}

// group lengths are stored 8 to a 64-bit word: JW rounded up
__host__ __device__ constexpr int JW_PAD(int jw) { return (jw + 7) & ~7; }

// Waves of a workgroup: NWC consumers (they own the bucket slots: gather, MFMA, reduction) and NWL loaders (they only
// stage: a wave that issues vector-memory instructions sits at the issue port while the memory pipeline takes a CU's
// ~50 KB block over thousands of cycles, so staging from the consumers stalls them).  NWL = 0: the consumers stage
// themselves - better for the log modes, where the gather is bound by the vector ALU and every wave is needed for it.
// JW: groups per consumer wave, TT: 16-type tiles.  AVL2 (the wide form: up to 64 cell types, sketch_dim up to 1024): the
// wave's slice of X_sketch does not stay in registers as MFMA A operands (JW x TT of them would not fit beside the bucket
// sums) - each group's TT operands are fetched from a copy of X_sketch laid out in operand order (tile_xa_kernel; it
// stays in L2) when the group's gather starts, and have landed when its sums are final.
// LOGV (float32 input, log modes): 0 = the float64 table chain, 2 = the float32-class log1p (tile_device.h: tile_log1p_f32).
// WG (round 4, the wide raw form): the per-entry weight table (8 bytes per scheduled step and lane class: 55-65 KB at 5000 genes)
// is replaced by the weights BY GENE of the column block in flight - (GB + 1) doubles at the head of every stage buffer, copied
// by the loaders with the block's rows (w_tab then holds NBLK such tables back to back, WB bytes each; entry GB is 0.0 and is
// what the lockstep padding steps point at, together with the zeroed pad behind every staged row).  The 40 KB this frees make
// the column blocks larger: 5 of 1024 genes instead of 7 of 736 at 5000 genes - fewer barriers, fewer (group, block) loop
// entries (they average ~1.1 steps), 12 % less lockstep padding.
// FF (with WG): the FLAT schedule.  The dynamic form walks a (wave, block)'s steps group by group - a loop per group whose
// trip count is the group's length in that block, ~1.1 at 5000 genes x 1024 buckets: every step two dependent LDS round trips
// (offset -> weight, value) with nothing else in flight, which three waves per SIMD cannot hide (the ISA shows a
// `s_waitcnt lgkmcnt(0)` per step).  Here the same steps, in the same order, are one flat stream served eight at a time - one
// 16-byte read brings a lane's eight offsets, sixteen reads (weights, values) go out together, eight fused multiply-adds follow -
// and the group a step belongs to is DATA (a byte per step, scalar loads), not control flow: the bucket sums live in two register
// vectors that the steps index with a wave-uniform number (s_set_gpr_idx_on + v_mov: what the compiler emits for a dynamically
// indexed ext_vector_type in registers).  Same sums, bit for bit, as the dynamic form of the same schedule.
// MEASURED SLOWER (one 1.25M x 5000 x 50 shard, d = 1024: 9.85 ms against 8.69 ms for the group loops) and therefore behind
// FDX_TILE_FLAT=1 only: the round trips were not the limit - a wave64 vector instruction occupies its SIMD for four cycles, the
// group loops spend ~5 of them per step (two address forms, convert, multiply-add, pointer bump), the flat form ~10 (offset
// unpacking, two address forms, convert, multiply-add, four moves through the indexed register) - the gather of the wide form
// is bound by vector-instruction issue, with the 1056 MFMAs of a tile (7 us per SIMD) behind it.  Kept as a tested variant.
typedef double tile_acc16_t __attribute__((ext_vector_type(16)));
typedef double tile_acc8_t __attribute__((ext_vector_type(8)));

template <typename T, int MODE, int NWC, int NWL, int JW, int TT, bool AVL2, int LOGV = 0, bool WG = false, bool FF = false>
__global__ __launch_bounds__((NWC + NWL) * 64, (NWC + NWL) / 4) void tile_sketch_kernel(
    const TileArgs a, const T* __restrict__ Yp, const int* __restrict__ row_map, const double* __restrict__ Xs,
    double* __restrict__ H, double* __restrict__ row_sumsq, const double* __restrict__ w_tab,
    const unsigned short* __restrict__ off_tab, const unsigned char* __restrict__ len_tab,
    const int* __restrict__ ent_base, const int* __restrict__ slot_bucket, const double* __restrict__ log_tab,
    const double* __restrict__ XA) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    typedef typename TileVec<T>::type V;
    constexpr int PER = 16 / sizeof(T);
    constexpr int NT = (NWC + NWL) * 64;
    constexpr int NWS = NWL > 0 ? NWL : NWC;                                // waves that stage
    constexpr int RPL = (TILE_ROWS + NWS - 1) / NWS;                        // rows a staging wave handles
    constexpr bool PAIR = NWC > 8 || AVL2;                                 // wave w + NR hands its tile to wave w first
    constexpr int NR = PAIR ? NWC / 2 : NWC;                               // partial tiles that reach the final sum
    static_assert(!PAIR || NWC % 2 == 0, "paired reduction needs an even number of consumer waves");
    static_assert(TT == 1 || TT == 2 || TT == 4, "type tiles: 1, 2 or 4");
    constexpr int TH = TT > 2 ? 2 : TT;                                     // type tiles per round of the final reduction
    constexpr int ROUNDS = TT / TH;
    constexpr int TS = TH * 4 * 64;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int r = lane & 15, q = lane >> 4;
    const int WB = WG ? a.WB : 0;
    const int stage_bytes = WB + TILE_ROWS * a.RS;
    const int NEp = (a.NE + 7) & ~7;
    const int NST = (NWL > 0 && MODE == FDX_PRE_RAW) ? a.NST : 2;
    double* w_l = reinterpret_cast<double*>(smem + (size_t)NST * stage_bytes);
    unsigned short* off_l = reinterpret_cast<unsigned short*>(w_l + (WG ? 0 : NEp));
    double* scales = reinterpret_cast<double*>(off_l + NEp);               // [2][16] scale of a row (log modes)
    int* rowok = reinterpret_cast<int*>(scales + 2 * TILE_ROWS);           // [2][16] every log argument of the row in the fast range
    double* logt = reinterpret_cast<double*>(smem + LOG_TAB_LDS);          // [LOG_TAB_N] (log modes), fixed place: see LOG_TAB_LDS
    for (int i = tid; i < a.NE; i += NT) {          // FF: NE counts the flat form's offsets (NWC x NBLK x 4 x NSP)
        if (!WG) w_l[i] = w_tab[i];
        off_l[i] = off_tab[i];
    }
    if (WG) {       // the pad behind every staged row is what a padding step reads (times weight 0.0): finite, i.e. zero
        for (int i = tid; i < NST * TILE_ROWS * 4; i += NT)
            *reinterpret_cast<unsigned*>(smem + (size_t)(i / (TILE_ROWS * 4)) * stage_bytes + WB + ((i >> 2) % TILE_ROWS) * a.RS + a.RS - TILE_ROW_PAD + (i & 3) * 4) = 0u;
    }
    constexpr bool F32LOG = LOGV != 0 && sizeof(T) == 4 && MODE != FDX_PRE_RAW;   // float32-class log1p: no table, no float64 chain
    if (MODE != FDX_PRE_RAW && !F32LOG)
        for (int i = tid; i < LOG_TAB_N; i += NT) logt[i] = log_tab[i];
    const long long n_tiles = (a.n + TILE_ROWS - 1) / TILE_ROWS;
    long long tile = blockIdx.x;
    if (tile >= n_tiles) return;

    // ---- staging (loader waves, or every wave when NWL = 0): wave lw stages rows lw, lw + NWS, ...
    const int lw = NWL > 0 ? wave - NWC : wave;
    auto load_rows = [&](long long t, const T* (&rp)[RPL]) {                // row addresses by scalar loads
#pragma unroll
        for (int k = 0; k < RPL; ++k) {
            const int rr = lw + NWS * k;
            rp[k] = nullptr;
            if (rr >= TILE_ROWS || t >= n_tiles) continue;
            const long long sp = t * TILE_ROWS + rr;
            if (sp < a.n) {
                const long long row = row_map ? (long long)row_map[sp] : sp;
                rp[k] = Yp + (size_t)row * (size_t)a.ldy;
            }
        }
    };
    auto issue_stage = [&](const T* const (&rp)[RPL], int c, int buf) -> int {   // returns the instructions issued (wave-uniform)
#ifdef FDX_TILE_EXPERIMENT
        if (a.dbg & 1) return 0;
#endif
        const int gene0 = c * a.GB;
        const int bytes = (min(a.GB, a.G - gene0)) * (int)sizeof(T);
        unsigned char* base = smem + (size_t)buf * stage_bytes + WB;
        int issued = 0;
        if (WG) {                                                           // the block's weights by gene: pieces lw, lw + NWS, ...
            const unsigned char* wsrc = reinterpret_cast<const unsigned char*>(w_tab) + (size_t)c * WB + lane * 16;
            for (int o = lw * 1024; o < WB; o += NWS * 1024) {
                if (o + lane * 16 < WB) dma16(wsrc + o, base - WB + o);
                ++issued;
            }
        }
#pragma unroll
        for (int k = 0; k < RPL; ++k) {
            if (!rp[k]) continue;                                           // row past the end: stale LDS, never stored
            const unsigned char* src = reinterpret_cast<const unsigned char*>(rp[k] + gene0) + lane * 16;
            unsigned char* dst = base + (lw + NWS * k) * a.RS;
            for (int o = 0; o < bytes; o += 1024) {
                if (o + lane * 16 < bytes) dma16(src + o, dst + o);
                ++issued;
            }
        }
        return issued;
    };
    // Row sums (log modes) in the scatter kernels' order (per-lane partials over ascending vectors, butterfly over the
    // wave), so every sketch path sees the same bits; with them the row's extremes, which tell whether every log argument
    // of the row lies in the fast range.  Two rows at a time: the loads of both (up to 16 KB) are in flight before the
    // first is summed.
    const int nvec = a.G / PER;                                             // launch requires G % PER == 0
    constexpr bool TWO = RPL > 1;                                           // a wave with one row has no second one to overlap
    auto scale_two = [&](const T* r0, const T* r1_, double* out_scale, int* out_ok, int i0, int i1) {
        const T* r1 = TWO ? r1_ : nullptr;
        const V* src0 = reinterpret_cast<const V*>(r0);
        const V* src1 = reinterpret_cast<const V*>(r1);
        double p0 = 0.0, p1 = 0.0;
        // "some element negative" by OR-ing the raw bits (the sign bit survives); NaN / Inf show up in the sum.  With every
        // element >= 0 the log argument y * 1e4 / sum cannot exceed 1e4, inside the fast range: no maximum is needed.
        unsigned long long sg0 = 0ULL, sg1 = 0ULL;
        auto bits_of = [](T v) -> unsigned long long {
            if constexpr (sizeof(T) == 4) return (unsigned long long)__float_as_uint((float)v) << 32;
            else return (unsigned long long)__double_as_longlong((double)v);
        };
        for (int v0 = 0; v0 < nvec; v0 += 512) {
            V x0[8], x1[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const int v = v0 + u * 64 + lane;
                if (v < nvec) {
                    if (r0) x0[u] = src0[v];
                    if (TWO && r1) x1[u] = src1[v];
                }
            }
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const int v = v0 + u * 64 + lane;
                if (v < nvec) {
#pragma unroll
                    for (int e = 0; e < PER; ++e) {
                        if (r0) { p0 += (double)x0[u][e]; sg0 |= bits_of(x0[u][e]); }
                        if (TWO && r1) { p1 += (double)x1[u][e]; sg1 |= bits_of(x1[u][e]); }
                    }
                }
            }
        }
        const double sum0 = wave_sum(p0);
        const double s0 = tile_row_scale<MODE>(sum0);
        // false for a NaN / Inf sum; above 1e18 the scale * 2^-65 would leave the float range, below 1e-30 the scale itself
        auto sum_ok = [](double s) { return fabs(s) <= 1e18 && (s == 0.0 || fabs(s) >= 1e-30); };
        const bool ok0 = sum_ok(sum0) && !__any((long long)sg0 < 0);
        // rows past the end of the matrix: a defined scale and a set flag (a stale flag would send the whole tile down the
        // general path - same values in float64, but not in the float32 class)
        if (lane == 0 && i0 < TILE_ROWS) { out_scale[i0] = r0 ? s0 : 0.0; out_ok[i0] = (!r0 || ok0) ? 1 : 0; }
        if (TWO && !r1 && lane == 0 && i1 < TILE_ROWS) { out_scale[i1] = 0.0; out_ok[i1] = 1; }
        if (TWO && r1) {
            const double sum1 = wave_sum(p1);
            const double s1 = tile_row_scale<MODE>(sum1);
            const bool ok1 = sum_ok(sum1) && !__any((long long)sg1 < 0);
            if (lane == 0) { out_scale[i1] = s1; out_ok[i1] = ok1 ? 1 : 0; }
        }
    };
    // rows k = first, first + step, ... < RPL of `rp`, two at a time
    auto scale_rows = [&](const T* const (&rp)[RPL], int first, int step, int par) {
        for (int k = first; k < RPL; k += 2 * step) {
            const T* r0 = nullptr;
            const T* r1 = nullptr;
#pragma unroll
            for (int kk = 0; kk < RPL; ++kk) {                              // static indexing of rp[]
                if (kk == k) r0 = rp[kk];
                if (kk == k + step) r1 = rp[kk];
            }
            scale_two(r0, r1, scales + par * TILE_ROWS, rowok + par * TILE_ROWS, lw + NWS * k, lw + NWS * (k + step));
        }
    };
    const T* rowp[RPL];
    const T* rown[RPL];
    bool has_next = false;
    // one block step of a staging wave: block c of the current tile has landed; stage the next block, sum a share of the
    // next tile's rows
    auto stage_step = [&](int c, int buf, int par) {
        if (c + 1 < a.NBLK) (void)issue_stage(rowp, c + 1, buf ^ 1);
        else if (has_next) (void)issue_stage(rown, 0, buf ^ 1);
    };
    // Row sums of the next tile as LATE as possible (the wave's k-th pair of rows in block NBLK-1-k, counted from the end):
    // the sums read the rows from HBM, the DMA of the next tile re-reads them 0 - 1 tile periods later, and the closer the
    // two reads the more of the second one the XCD's 4 MB L2 still holds (32 CUs x 128 KB of rows per tile period).
    auto sums_step = [&](int c, int par) {
        if (MODE != FDX_PRE_RAW && has_next) scale_rows(rown, a.NBLK - 1 - c, a.NBLK, par ^ 1);
    };
    int young = 0;                                                          // pieces of the youngest block in flight (loader waves)
    if (NWL == 0 || wave >= NWC) {
        load_rows(tile, rowp);
        young = issue_stage(rowp, 0, 0);
        if (NST == 3) young = issue_stage(rowp, 1, 1);                     // the launcher gives three stages only to NBLK >= 2
        if (MODE != FDX_PRE_RAW) scale_rows(rowp, 0, 1, 0);
    }

    if (NWL > 0 && wave >= NWC) {
        // ================================================================================================ loader wave
        // Ring of NST stage buffers: block s is consumed from buffer s % NST.  At the barrier that opens block s (block s has
        // landed, block s - 1 is done with) the loaders request block s + NST - 1 into the buffer block s - 1 has left.  With
        // three stages a request goes out while the previous one is still in flight: two blocks' worth of bytes under way
        // instead of one (with two stages every block paid its full memory latency after the barrier).
        int buf = 0, par = 0;
        for (; tile < n_tiles; tile += gridDim.x) {
            has_next = tile + gridDim.x < n_tiles;
            load_rows(tile + gridDim.x, rown);
            for (int c = 0; c < a.NBLK; ++c) {
                if (NST == 3) wait_vmcnt_le(young);                          // everything OLDER than the youngest request: block c has landed
                else __builtin_amdgcn_s_waitcnt(0x0f70);                     // vmcnt(0): this wave's pieces of block c have landed
                lds_barrier();                                              // everybody's have; the buffer of block c - 1 is free
                if (NST == 3) {
                    const int cn = c + 2;
                    int into = buf + 2;
                    if (into >= 3) into -= 3;
                    if (cn < a.NBLK) young = issue_stage(rowp, cn, into);
                    else if (has_next) young = issue_stage(rown, cn - a.NBLK, into);
                    else young = 0;
                    buf = buf + 1 == 3 ? 0 : buf + 1;
                } else {
                    stage_step(c, buf, par);
                    sums_step(c, par);
                    buf ^= 1;
                }
            }
            par ^= 1;
#pragma unroll
            for (int k = 0; k < RPL; ++k) rowp[k] = rown[k];
#ifdef FDX_TILE_EXPERIMENT
            if (!(a.dbg & 8))
#endif
            for (int rd = 0; rd < ROUNDS; ++rd) {                            // the consumers' reduction
                lds_barrier();
                if (PAIR) lds_barrier();
                lds_barrier();
            }
            if (WG) lds_barrier();
        }
        return;
    }

    // ==================================================================================================== consumer wave
    // this wave's slice of X_sketch as MFMA A operands: A[m = type r][k = q] = X_sketch[type, bucket of slot (w, j, q)];
    // unconditional loads (index clamped, value selected) and one wait, so nothing of this is pending in the tile loop
    double av[AVL2 ? 1 : JW][TT];
    // AVL2: operand (j, t) of this wave at xu[(j * TT + t) * 64 + lane] - a uniform base per operand plus the lane, so that
    // the loads take scalar bases (128 per-lane 64-bit addresses would be hoisted out of the tile loop and spilled)
    const double* xu = XA + ((size_t)wave * JW * TT) * 64;
    unsigned lane8 = (unsigned)lane * 8u;
#pragma unroll
    for (int j = 0; j < (AVL2 ? 0 : JW); ++j) {
        const int b = slot_bucket[(wave * JW + j) * 4 + q];
#pragma unroll
        for (int t = 0; t < TT; ++t) {
            const int type = t * 16 + r;
            const bool ok = b >= 0 && type < a.K;
            const double v = Xs[(size_t)(ok ? type : 0) * a.d + (ok ? b : 0)];
            av[j][t] = ok ? v : 0.0;
        }
    }
    if (NWL > 0) __builtin_amdgcn_s_waitcnt(0x0f70);
    LogConsts lc{};
    if constexpr (!F32LOG) lc = log_consts();
    int buf = 0, par = 0;
    for (; tile < n_tiles; tile += gridDim.x) {
        if (AVL2) asm volatile("" : "+v"(lane8));                           // the operand addresses are formed where they are used
        if (NWL == 0) {
            has_next = tile + gridDim.x < n_tiles;
            load_rows(tile + gridDim.x, rown);
        }
        double acc[FF ? 1 : JW];
#pragma unroll
        for (int j = 0; j < (FF ? 1 : JW); ++j) acc[j] = 0.0;
        tile_acc16_t accA = {0.0, 0.0, 0.0, 0.0, 0.0, 0.0, 0.0, 0.0, 0.0, 0.0, 0.0, 0.0, 0.0, 0.0, 0.0, 0.0};   // FF: groups 0..15
        tile_acc8_t accB = {0.0, 0.0, 0.0, 0.0, 0.0, 0.0, 0.0, 0.0};                                              //     groups 16..23
        double4_t accm[TT];
#pragma unroll
        for (int t = 0; t < TT; ++t) accm[t] = double4_t{0.0, 0.0, 0.0, 0.0};
        double sq = 0.0;
        double scale = 1.0;
        double scale_s = FDX_LOG_DOWN;                                      // scale * 2^-65 (tile_device.h)
        float scale_sf = FDX_LOG_DOWN_F;
        float scale_f = 1.0f;
        bool fast = true;
        // One column block: software pipeline over the flat entry stream - weight and value of the current step in
        // registers, the offset of the step after next already fetched, so a step costs one LDS round trip, not two.
        // LAST: the group's sum is final when its loop ends, and its MFMAs go out at once - they run in the matrix pipe
        // beside the gather of the following groups.  (A run-time test per group instead of the template parameter would
        // make the accumulators merge points and serialise the MFMAs behind register copies.)  FAST: every log argument
        // of the tile is known to be in the fast range (rowok), no per-element test.
        auto consume = [&](int c, auto last_tag, auto fast_tag) {
            constexpr bool LAST = decltype(last_tag)::value;
            constexpr bool FAST = decltype(fast_tag)::value;
            const unsigned char* rowb = smem + (size_t)buf * stage_bytes + WB + r * a.RS;
            const double* wgl = reinterpret_cast<const double*>(smem + (size_t)buf * stage_bytes);   // WG: this block's weights by gene
            if constexpr (FF) {
                static_assert(!FF || (WG && MODE == FDX_PRE_RAW && JW <= 24), "flat form: raw mode, weights by gene, at most 24 groups per wave");
                typedef unsigned uint4_t __attribute__((ext_vector_type(4)));
                const unsigned short* so = off_l + ((size_t)(wave * a.NBLK + c) * 4 + q) * a.NSP;
                const unsigned long long* gw = reinterpret_cast<const unsigned long long*>(len_tab + (size_t)(wave * a.NBLK + c) * a.GROW);   // scalar loads
                const int ns_a = (int)(gw[0] & 0xffffULL), ns = (int)((gw[0] >> 16) & 0xffffULL);   // steps of groups 0..15, all steps
                auto batch = [&](int k0, auto first_tag) {
                    const uint4_t o = *reinterpret_cast<const uint4_t*>(so + k0);
                    const unsigned long long g8 = gw[1 + (k0 >> 3)];        // the eight steps' groups
                    double wv[8];
                    T yv[8];
#pragma unroll
                    for (int u = 0; u < 8; ++u) {
                        const unsigned wd = o[u >> 1];
                        const unsigned of = (u & 1) ? (wd >> 16) : (wd & 0xffffu);   // a padding step reads the padding offset: weight 0.0
                        wv[u] = wgl[of];
                        yv[u] = *reinterpret_cast<const T*>(rowb + (size_t)of * sizeof(T));
                    }
#pragma unroll
                    for (int u = 0; u < 8; ++u) {
                        const int j = (int)((g8 >> (8 * u)) & 0xffULL);
                        if (decltype(first_tag)::value) accA[j] = fma(wv[u], (double)yv[u], accA[j]);
                        else accB[j - 16] = fma(wv[u], (double)yv[u], accB[j - 16]);
                    }
                };
                for (int k0 = 0; k0 < ns_a; k0 += 8) batch(k0, std::true_type{});
                for (int k0 = ns_a; k0 < ns; k0 += 8) batch(k0, std::false_type{});
                if (LAST) {
#pragma unroll
                    for (int j = 0; j < JW; ++j) {
                        double an[TT];
                        if ((j & 3) == 0) __builtin_amdgcn_sched_barrier(0);    // at most four groups' operands in flight
#pragma unroll
                        for (int t = 0; t < TT; ++t) an[t] = AVL2 ? *reinterpret_cast<const double*>(reinterpret_cast<const char*>(xu + (size_t)(j * TT + t) * 64) + lane8) : av[AVL2 ? 0 : j][t];
                        const double sum = j < 16 ? accA[j < 16 ? j : 0] : accB[j < 16 ? 0 : j - 16];
#pragma unroll
                        for (int t = 0; t < TT; ++t) accm[t] = __builtin_amdgcn_mfma_f64_16x16x4f64(an[t], sum, accm[t], 0, 0, 0);
                        sq = fma(sum, sum, sq);
                    }
                }
            } else {
            int p = ent_base[wave * (a.NBLK + 1) + c] + q;
            const unsigned long long* lens = reinterpret_cast<const unsigned long long*>(len_tab + ((size_t)wave * a.NBLK + c) * JW_PAD(JW));
            const unsigned off0 = off_l[p];
            double wv = WG ? wgl[off0] : w_l[p];
            T yv = *reinterpret_cast<const T*>(rowb + (size_t)off0 * sizeof(T));
            unsigned offn = off_l[p + 4];
            auto f = [&](T yy) -> double {
                if (MODE == FDX_PRE_RAW) return (double)yy;
                if constexpr (F32LOG) {
                    if (FAST) return (double)tile_log1p_f32((float)yy, scale_f);
                    return tile_log1p_any((double)yy * scale);
                } else {
                    if (FAST) return tile_log1p_scaled(yy, scale_s, scale_sf, lc);
                    return tile_log1p((double)yy * scale, lc);
                }
            };
#pragma unroll
            for (int j = 0; j < JW; ++j) {
#ifdef FDX_TILE_EXPERIMENT
                const int len = (a.dbg & 2) ? 0 : (int)((lens[j >> 3] >> ((j & 7) * 8)) & 0xffULL);
#else
                const int len = (int)((lens[j >> 3] >> ((j & 7) * 8)) & 0xffULL);
#endif
                double an[TT];
                if (LAST && AVL2) {
                    __builtin_amdgcn_sched_barrier(0);                       // the operand loads of later groups stay with their groups
#pragma unroll
                    for (int t = 0; t < TT; ++t) an[t] = *reinterpret_cast<const double*>(reinterpret_cast<const char*>(xu + (size_t)(j * TT + t) * 64) + lane8);
                }
                int t = 0;
                for (; t + 2 <= len; t += 2) {                            // two steps per trip: the register sets swap roles
                    const double wb = WG ? wgl[offn] : w_l[p + 4];
                    const T yb = *reinterpret_cast<const T*>(rowb + (size_t)offn * sizeof(T));
                    const unsigned offb = off_l[p + 8];
                    acc[j] = fma(wv, f(yv), acc[j]);
                    p += 8;
                    wv = WG ? wgl[offb] : w_l[p];
                    yv = *reinterpret_cast<const T*>(rowb + (size_t)offb * sizeof(T));
                    offn = off_l[p + 4];
                    acc[j] = fma(wb, f(yb), acc[j]);
                }
                if (t < len) {
                    p += 4;
                    const double wn = WG ? wgl[offn] : w_l[p];
                    const T yn = *reinterpret_cast<const T*>(rowb + (size_t)offn * sizeof(T));
                    offn = off_l[p + 4];
                    acc[j] = fma(wv, f(yv), acc[j]);
                    wv = wn;
                    yv = yn;
                }
#ifdef FDX_TILE_EXPERIMENT
                if (LAST && !(a.dbg & 4)) {
#else
                if (LAST) {
#endif
#pragma unroll
                    for (int t = 0; t < TT; ++t)
                        accm[t] = __builtin_amdgcn_mfma_f64_16x16x4f64(AVL2 ? an[t] : av[AVL2 ? 0 : j][t], acc[j], accm[t], 0, 0, 0);
                    sq = fma(acc[j], acc[j], sq);
                }
            }
        }
        };
        auto block = [&](int c, auto last_tag) {
            if (NWL == 0) __builtin_amdgcn_s_waitcnt(0x0f70);                // vmcnt(0): this wave's pieces of block c have landed
            lds_barrier();                                                  // everybody's have (the loaders waited for theirs)
            if (NWL == 0) stage_step(c, buf, par);
            if (MODE != FDX_PRE_RAW && c == 0) {
                scale = scales[par * TILE_ROWS + r];
                if constexpr (F32LOG) {
                    scale_f = (float)scale;
                } else {
                    scale_s = scale * FDX_LOG_DOWN;
                    scale_sf = (float)scale_s;
                }
                fast = __all(rowok[par * TILE_ROWS + r] != 0);
            }
            if (MODE == FDX_PRE_RAW || fast) consume(c, last_tag, std::true_type{});
            else consume(c, last_tag, std::false_type{});
            if (NWL == 0) sums_step(c, par);
            buf = buf + 1 == NST ? 0 : buf + 1;
        };
        // raw: MFMAs interleaved with the last block's gather.  Log modes: afterwards - the gather is bound by the vector ALU
        // there, and the 16 accumulator registers held through it would spill.
        constexpr bool INTERLEAVE = MODE == FDX_PRE_RAW;
        for (int c = 0; c + 1 < a.NBLK; ++c) block(c, std::false_type{});
        block(a.NBLK - 1, std::integral_constant<bool, INTERLEAVE>{});
        if constexpr (!INTERLEAVE) {
#pragma unroll
            for (int j = 0; j < JW; ++j) {
                double an[TT];
                if (AVL2) {
                    if ((j & 3) == 0) __builtin_amdgcn_sched_barrier(0);    // at most four groups' operands in flight
#pragma unroll
                    for (int t = 0; t < TT; ++t) an[t] = *reinterpret_cast<const double*>(reinterpret_cast<const char*>(xu + (size_t)(j * TT + t) * 64) + lane8);
                }
#pragma unroll
                for (int t = 0; t < TT; ++t)
                    accm[t] = __builtin_amdgcn_mfma_f64_16x16x4f64(AVL2 ? an[t] : av[AVL2 ? 0 : j][t], acc[j], accm[t], 0, 0, 0);
                sq = fma(acc[j], acc[j], sq);
            }
        }
        par ^= 1;
        if (NWL == 0) {
#pragma unroll
            for (int k = 0; k < RPL; ++k) rowp[k] = rown[k];
        }
        // ---- the partial tiles are added in a fixed order through LDS (the buffer of the block just consumed) and stored,
        // TH type tiles per round (the area must fit a stage buffer)
        double* red = reinterpret_cast<double*>(smem + (size_t)(buf == 0 ? NST - 1 : buf - 1) * stage_bytes);   // the block just consumed: [NR][TS] + [NR][64]
        double* red_sq = red + (size_t)NR * TS;
        const long long s0 = tile * TILE_ROWS;
#ifdef FDX_TILE_EXPERIMENT
        if (!(a.dbg & 8))
#endif
#pragma unroll
        for (int rd = 0; rd < ROUNDS; ++rd) {
            lds_barrier();                                                  // the last block's buffer / the previous round's sums are free
            if (PAIR) {
                if (wave >= NR) {
#pragma unroll
                    for (int t = 0; t < TH; ++t)
#pragma unroll
                        for (int rr = 0; rr < 4; ++rr) red[(size_t)(wave - NR) * TS + (t * 4 + rr) * 64 + lane] = accm[rd * TH + t][rr];
                    if (rd == 0) red_sq[(wave - NR) * 64 + lane] = sq;
                }
                lds_barrier();
                if (wave < NR) {
#pragma unroll
                    for (int t = 0; t < TH; ++t)
#pragma unroll
                        for (int rr = 0; rr < 4; ++rr) red[(size_t)wave * TS + (t * 4 + rr) * 64 + lane] += accm[rd * TH + t][rr];
                    if (rd == 0) red_sq[wave * 64 + lane] += sq;
                }
            } else {
#pragma unroll
                for (int t = 0; t < TH; ++t)
#pragma unroll
                    for (int rr = 0; rr < 4; ++rr) red[(size_t)wave * TS + (t * 4 + rr) * 64 + lane] = accm[rd * TH + t][rr];
                if (rd == 0) red_sq[wave * 64 + lane] = sq;
            }
            lds_barrier();
            for (int o = tid; o < TS; o += NWC * 64) {
                double sum = 0.0;
#pragma unroll
                for (int v = 0; v < NR; ++v) sum += red[(size_t)v * TS + o];              // fixed order: deterministic
                const int l = o & 63, tr = o >> 6;
                const int type = (rd * TH + (tr >> 2)) * 16 + (l >> 4) + 4 * (tr & 3);
                const long long sp = s0 + (l & 15);
                if (type < a.K && sp < a.n) H[(size_t)type * a.ldh + sp] = sum;
            }
            if (rd == 0 && row_sumsq && tid < TILE_ROWS && s0 + tid < a.n) {
                double sum = 0.0;
                for (int v = 0; v < NR; ++v)
#pragma unroll
                    for (int qq = 0; qq < 4; ++qq) sum += red_sq[v * 64 + qq * 16 + tid];
                row_sumsq[s0 + tid] = sum;
            }
        }
        if (WG) {   // the sums were written over this buffer's row pads: zero them again before the next block lands here
            lds_barrier();
            if (tid < TILE_ROWS * 4)
                *reinterpret_cast<unsigned*>(reinterpret_cast<unsigned char*>(red) + WB + (tid >> 2) * a.RS + a.RS - TILE_ROW_PAD + (tid & 3) * 4) = 0u;
        }
        // the first barrier of the next tile orders these reads before the next DMA into this buffer
    }
}

// ---- host side ------------------------------------------------------------------------------------------------------

struct TilePlanDevice {
    TilePlanHost h;
    DevBuf w, off, len, ent_base, slot_bucket;
    int NWC = 0, NWL = 0, JW = 0, RS = 0, TT = 0, NST = 2;
    int WB = 0;          // WG form: bytes of a block's weights-by-gene table (w holds NBLK of them); 0 = per-entry weights
    TileFlatHost fh;     // flat form (fh.NSP > 0): `off` holds fh.off, `len` holds fh.gid
    bool wide = false;
    size_t lds = 0;
};

#if !FDX_TILE_PART
const double* log_table_dev(hipStream_t st) {   // -log of every table reciprocal in [2^-15, 1], one copy per device
    static std::mutex mu;
    static double* tabs[64] = {};
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return nullptr;
    std::lock_guard<std::mutex> lock(mu);
    if (!tabs[dev]) {
        std::vector<double> t((size_t)LOG_TAB_N);
        for (int i = 0; i < LOG_TAB_N; ++i) {
            const unsigned bits = (unsigned)(LOG_TAB_BASE + i) << LOG_TAB_SHIFT;
            float c;                                                        // 2^65 x the reciprocal
            std::memcpy(&c, &bits, 4);
            t[(size_t)i] = (double)(-logl((long double)c) + (long double)LOG_TAB_EXP_SHIFT * 0.693147180559945309417232121458176568L);
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
#endif

// Wave split of a workgroup: consumers x groups per consumer + loaders.  Raw: 12 + 4 (measured at 1M x 2000 x 30: 1.92 ms
// against 2.05 ms self-staged and 2.25 ms with 14 + 2).  Log modes: 16 self-staging waves - the table-driven log1p makes
// the gather ALU-bound and idle loader waves cost more than they save (4.3 ms against 6.4 ms with 12 + 4).
// FDX_TILE_CFG=12 / 16 / 8 forces 12 + 4 / 16 + 0 / 8 + 2 (tuning experiments).
// Wide form (33..64 cell types, or more buckets than the narrow split owns): twice the groups per wave, four type tiles,
// MFMA A operands from the L2-resident operand copy of X_sketch (AVL2).
#if !FDX_TILE_PART
struct TileCfg { int NWC, NWL, JW, TT; bool wide; };
static TileCfg tile_cfg(int mode, int K, int d) {
    const char* e = fdx::env("FDX_TILE_CFG");
    const int v = e ? atoi(e) : (mode == FDX_PRE_RAW ? 12 : 16);
    TileCfg c = v == 16 ? TileCfg{16, 0, 8, 0, false} : v == 8 ? TileCfg{8, 2, 16, 0, false} : TileCfg{12, 4, 11, 0, false};
    c.TT = (K + 15) / 16;
    if (K > 32 || d > 4 * c.NWC * c.JW) {
        // eight consumer waves: 256 registers each hold 32 bucket sums, four type tiles and the gather's pipeline
        c = mode == FDX_PRE_RAW ? TileCfg{12, 4, 22, 4, true} : TileCfg{8, 0, 32, 4, true};
    }
    return c;
}

static size_t tile_lds_bytes(int RS, int NE, int mode, int NST = 2, int WB = 0) {
    const size_t NEp = ((size_t)NE + 7) & ~(size_t)7;
    // WB > 0 (WG form): weights by gene inside every stage buffer, the entry table holds the 2-byte offsets only
    const size_t below = (size_t)NST * ((size_t)WB + (size_t)TILE_ROWS * RS) + NEp * (WB ? 2 : 10) + 2 * TILE_ROWS * (8 + 4);
    if (mode == FDX_PRE_RAW) return below;
    // log modes: the table has a fixed place at the top of the 160 KB (LOG_TAB_LDS); everything else must end below it
    return below <= (size_t)LOG_TAB_LDS ? (size_t)160 * 1024 : (size_t)161 * 1024;
}

// Builds (once per SketchPlan and input type) the schedule for the largest column block that fits the LDS.
static const TilePlanDevice* tile_plan_for(const SketchPlan& sp, int dtype, int mode, int K, hipStream_t st) {
    const int sz = dtype == FDX_F32 ? 4 : 8;
    const TileCfg cfg = tile_cfg(mode, K, sp.d);
    const int TT = cfg.TT;
    if (K < 1 || K > 64 || (!cfg.wide && TT > 2)) return nullptr;
    const int key0 = cfg.wide ? 24 + ((dtype == FDX_F32 ? 0 : 1) * 2 + (mode != FDX_PRE_RAW ? 1 : 0))
                             : ((((dtype == FDX_F32 ? 0 : 1) * 2 + (mode != FDX_PRE_RAW ? 1 : 0)) * 2 + (TT - 1)) * 3) +
                                   (cfg.NWC == 12 ? 0 : cfg.NWC == 16 ? 1 : 2);
    // Stage buffers: two.  FDX_TILE_NST=3 (raw mode with loader waves) makes it a ring of three - two column blocks in flight
    // while one is consumed, smaller blocks (2000 float32 genes: 3 x 704 instead of 2 x 1024).  Measured at 1M x 2000: 1.88-1.95
    // against 1.89-1.90 ms - the consumers' gather, not the bytes in flight, sets the block period; kept as a switch.
    // The wide raw form keeps its weights by gene in the stage buffers (WG, see the kernel): 5 column blocks of 1024 genes instead
    // of 7 of 736 at 5000 genes (one 1.25M x 5000 x 50 shard: 8.40 -> 8.05 ms).  FDX_TILE_NO_WG=1: the per-entry weight table.
    // Measured on the same shard and NOT the default: the ring of three stage buffers on top of it (FDX_TILE_NST=3: 8 blocks of
    // 672, 8.94 ms - more blocks, more lockstep padding, and the consumers, not the bytes in flight, set the block period) and the
    // flat schedule (FDX_TILE_FLAT=1, see the kernel: 9.85 ms against 8.69 in the same process).
    // (the three alternatives - FDX_TILE_NO_WG, FDX_TILE_NST=3, FDX_TILE_FLAT - were all measured slower and are compiled only
    // into experiment builds, `make EXTRA=-DFDX_TILE_EXPERIMENT`: the shipped library has one layout per shape)
#ifdef FDX_TILE_EXPERIMENT
    const bool wg = cfg.wide && mode == FDX_PRE_RAW && cfg.NWL > 0 && !fdx::exp_env("FDX_TILE_NO_WG");
    int NST = 2;
    if (mode == FDX_PRE_RAW && cfg.NWL > 0) {
        const char* e = fdx::exp_env("FDX_TILE_NST");
        NST = (e && atoi(e) == 3) ? 3 : 2;
    }
    const bool flat = wg && fdx::exp_env("FDX_TILE_FLAT") && cfg.JW <= 24;
#else
    const bool wg = cfg.wide && mode == FDX_PRE_RAW && cfg.NWL > 0;
    const int NST = 2;
    const bool flat = false;
#endif
    const int key = key0 + (NST == 3 ? 28 : 0) + (cfg.wide && mode == FDX_PRE_RAW && cfg.NWL > 0 && !wg ? 56 : 0) + (flat ? 112 : 0);
    static_assert(SketchPlan::kTileKeys == 224, "key space of the schedules");
    std::lock_guard<std::mutex> lock(sp.tile_mu);
    if (sp.tile_tried[key]) return sp.tile[key].get();
    sp.tile_tried[key] = true;
    const bool dbg = fdx::env("FDX_DEBUG") != nullptr;
    if (!sp.scatter_ok || sp.host_bucket.empty()) return nullptr;
    if (sp.d > 4 * cfg.NWC * cfg.JW) return nullptr;
    const size_t red_bytes = (size_t)(cfg.NWC > 8 || cfg.wide ? cfg.NWC / 2 : cfg.NWC) * (std::min(TT, 2) * 4 * 64 + 64) * 8;   // the kernel's reduction area
    // block sizes tried: whole 1 KB pieces; the wide form's tables leave less room, and an eighth of a piece more or less decides
    // whether 5000 genes take 7 blocks or 10 (21 % more lockstep padding)
    const int unit = (cfg.wide || NST == 3 ? 128 : 1024) / sz;
    std::unique_ptr<TilePlanDevice> best;
    for (int GB = (int)round_up(sp.G, unit); GB >= unit; GB -= unit) {
        const int RS = GB * sz + TILE_ROW_PAD;
        const int WB = wg ? (int)round_up((GB + 1) * 8, 16) : 0;
        if ((size_t)WB + (size_t)TILE_ROWS * RS < red_bytes) break;
        // cheap bound before building: the tables hold at least G entries
        const int nst = (NST == 3 && GB < sp.G) ? 3 : 2;                     // one block per tile: the ring's look-ahead needs two
        if (tile_lds_bytes(RS, sp.G, mode, nst, WB) > 160 * 1024) continue;
        auto cand = std::make_unique<TilePlanDevice>();
        if (!build_tile_plan(sp.host_bucket.data(), sp.host_w.data(), sp.G, sp.d, cfg.NWC, cfg.JW, GB, &cand->h)) return nullptr;
        cand->NWC = cfg.NWC; cand->NWL = cfg.NWL; cand->JW = cfg.JW; cand->RS = RS; cand->TT = TT; cand->wide = cfg.wide;
        cand->WB = WB;
        cand->NST = (nst == 3 && cand->h.NBLK >= 2) ? 3 : 2;
        int n_ent = cand->h.NE;
        if (flat) {
            if (!build_tile_flat(cand->h, GB, &cand->fh)) return nullptr;
            n_ent = (int)cand->fh.off.size();
        }
        cand->lds = tile_lds_bytes(RS, n_ent, mode, nst, WB);
        if (dbg) std::fprintf(stderr, "[fdx] tile plan: G=%d d=%d waves=%d+%d stages=%d GB=%d blocks=%d NE=%d steps=%d lds=%zu flat rows of %d\n", sp.G, sp.d, cfg.NWC,
                              cfg.NWL, cand->NST, GB, cand->h.NBLK, cand->h.NE, cand->h.steps, cand->lds, cand->fh.NSP);
        if (cand->lds > 160 * 1024) continue;
        best = std::move(cand);
        break;
    }
    if (!best) return nullptr;
    TilePlanDevice& t = *best;
    auto up = [&](DevBuf& b, const void* src, size_t bytes) -> int {
        FDX_TRY(b.alloc(bytes));
        FDX_TRY(copy_h2d(b.p, src, bytes, st));
        return 0;
    };
    // group lengths, 8 to a 64-bit word: rows of JW_PAD(JW) bytes
    const int jp = JW_PAD(t.JW);
    std::vector<unsigned char> len_pad((size_t)t.NWC * t.h.NBLK * jp + 16, 0);
    for (int wv = 0; wv < t.NWC; ++wv)
        for (int c = 0; c < t.h.NBLK; ++c)
            for (int j = 0; j < t.JW; ++j)
                len_pad[((size_t)wv * t.h.NBLK + c) * jp + j] = t.h.len[((size_t)wv * t.h.NBLK + c) * t.JW + j];
    std::vector<double> wg_tab;
    if (t.WB) {
        // weights by gene, one table of WB bytes per column block ((GB + 1) doubles: entry GB stays 0.0); the padding steps of the
        // schedule (no gene) point at entry GB - weight 0.0 times the zeroed pad behind the staged row
        const int per = t.WB / 8;
        wg_tab.assign((size_t)t.h.NBLK * per, 0.0);
        for (int g = 0; g < sp.G; ++g)
            if (sp.host_bucket[(size_t)g] >= 0) wg_tab[(size_t)(g / t.h.GB) * per + g % t.h.GB] = sp.host_w[(size_t)g];
        for (size_t i = 0; i < t.h.off.size(); ++i)
            if (t.h.gene[i] < 0) t.h.off[i] = (unsigned short)t.h.GB;
    }
    if ((t.WB ? up(t.w, wg_tab.data(), wg_tab.size() * 8) : up(t.w, t.h.w.data(), t.h.w.size() * 8)) ||
        (t.fh.NSP > 0 ? up(t.off, t.fh.off.data(), t.fh.off.size() * 2) : up(t.off, t.h.off.data(), t.h.off.size() * 2)) ||
        (t.fh.NSP > 0 ? up(t.len, t.fh.gid.data(), t.fh.gid.size()) : up(t.len, len_pad.data(), len_pad.size())) || up(t.ent_base, t.h.ent_base.data(), t.h.ent_base.size() * 4) ||
        up(t.slot_bucket, t.h.slot_bucket.data(), t.h.slot_bucket.size() * 4))
        return nullptr;
    if (hipStreamSynchronize(st) != hipSuccess) return nullptr;            // the host vectors may die with the plan
    sp.tile[key] = std::move(best);
    return sp.tile[key].get();
}

// X_sketch rearranged into MFMA A operands for the wide form: XA[((w * JW + j) * TT + t) * 64 + lane] =
// X_sketch[type = 16 t + (lane & 15), bucket of slot (w, j, lane >> 4)], 0 where there is no such type or bucket.
__global__ void tile_xa_kernel(const double* __restrict__ Xs, const int* __restrict__ slot_bucket, int K, int d, int n_groups,
                               int TT, double* __restrict__ XA) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n_groups * TT * 64) return;
    const int lane = i & 63, t = (i >> 6) % TT, wj = (i >> 6) / TT;
    const int b = slot_bucket[wj * 4 + (lane >> 4)];
    const int type = t * 16 + (lane & 15);
    XA[i] = (b >= 0 && type < K) ? Xs[(size_t)type * d + b] : 0.0;
}

bool tile_sketch_ok(int dtype, long long ldy, const void* Y, int G, int d, int K, int mode, const SketchPlanDev& plan,
                    hipStream_t st) {
    if (fdx::exp_env("FDX_NO_TILE") || !plan.owner) return false;
    if (dtype != FDX_F32 && dtype != FDX_F64) return false;
    if (mode != FDX_PRE_RAW && mode != FDX_PRE_LOG_CPM && mode != FDX_PRE_LOG_CPM_SPARSE) return false;
    const int sz = dtype == FDX_F32 ? 4 : 8;
    if (K <= 0 || K > 64 || G <= 0 || d <= 0) return false;
    if ((K > 32 || d > 512) && fdx::env("FDX_NO_TILE_WIDE")) return false;
    // whole 16-byte vectors only: row starts and row lengths multiples of 16 bytes
    if (((size_t)G * sz) % 16 != 0 || ((size_t)ldy * sz) % 16 != 0 || (reinterpret_cast<uintptr_t>(Y) & 15) != 0) return false;
    return tile_plan_for(*plan.owner, dtype, mode, K, st) != nullptr;
}

#endif  // !FDX_TILE_PART

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
    const double* XA;
    bool flat = false;   // the flat schedule (FF)
};

// This file is compiled twice (Makefile): FDX_TILE_PART 0 holds the schedule, the entry points and the float32 kernels,
// FDX_TILE_PART 1 the float64 kernels alone (launch_tile_mode<double>) - the kernel template's instantiations took 110 s in
// one translation unit, the longest of the build.
int tile_logv();
#if !FDX_TILE_PART
static thread_local bool t_f64_math = false;
TileF64Math::TileF64Math(bool on) : prev(t_f64_math) { t_f64_math = on; }
TileF64Math::~TileF64Math() { t_f64_math = prev; }

int tile_logv() {
    if (t_f64_math) return 0;
    const char* e = fdx::env("FDX_TILE_LOGV");
    return e ? atoi(e) : 2;
}
#endif

template <typename T, int MODE, int NWC, int NWL, int JW>
static int launch_tile_tt(const TileLaunch& L, int TT, size_t lds, int grid, hipStream_t st) {
    const void* kern = TT == 1 ? (const void*)tile_sketch_kernel<T, MODE, NWC, NWL, JW, 1, false>
                               : (const void*)tile_sketch_kernel<T, MODE, NWC, NWL, JW, 2, false>;
    // log modes, 16 self-staging waves.  float64 chain (float64 rows, integer counts, FDX_TILE_LOGV=0): operands from the L2
    // copy as in the wide form - the 32 registers they would occupy are what the 128-register budget lacks for the log1p
    // chains (22 spills with them; 4.23 -> 4.1 ms).  float32-class chain (float32 rows): no table, no polynomial constants -
    // the operands fit back into registers (127 VGPRs, no spills: 3.08 -> 2.87 ms); FDX_TILE_AVL2=1 fetches them all the
    // same.  Raw (12 + 4) keeps them in registers: 2.34 ms with the fetches against 1.98 ms.
    if constexpr (MODE != FDX_PRE_RAW && NWC == 16) {
        if (L.XA)
            kern = TT == 1 ? (const void*)tile_sketch_kernel<T, MODE, 16, 0, 8, 1, true> : (const void*)tile_sketch_kernel<T, MODE, 16, 0, 8, 2, true>;
        if constexpr (std::is_same<T, float>::value) {
            if (tile_logv() != 0)
                kern = L.XA ? (TT == 1 ? (const void*)tile_sketch_kernel<T, MODE, 16, 0, 8, 1, true, 2> : (const void*)tile_sketch_kernel<T, MODE, 16, 0, 8, 2, true, 2>)
                            : (TT == 1 ? (const void*)tile_sketch_kernel<T, MODE, 16, 0, 8, 1, false, 2> : (const void*)tile_sketch_kernel<T, MODE, 16, 0, 8, 2, false, 2>);
        }
    }
    if (lds > 64 * 1024) FDX_HIP(hipFuncSetAttribute(kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    void* args[] = {(void*)&L.a, (void*)&L.Y, (void*)&L.row_map, (void*)&L.Xs, (void*)&L.H, (void*)&L.row_sumsq, (void*)&L.w_tab,
                    (void*)&L.off_tab, (void*)&L.len_tab, (void*)&L.ent_base, (void*)&L.slot_bucket, (void*)&L.log_tab,
                    (void*)&L.XA};
    FDX_HIP(hipLaunchKernel(kern, dim3(grid), dim3((NWC + NWL) * 64), args, lds, st));
    return 0;
}

template <typename T, int MODE, int NWC, int NWL, int JW>
static int launch_tile_wide(const TileLaunch& L, size_t lds, int grid, hipStream_t st) {
    const void* kern = nullptr;
    if constexpr (MODE == FDX_PRE_RAW && NWL > 0) {
#ifdef FDX_TILE_EXPERIMENT
        kern = (const void*)tile_sketch_kernel<T, MODE, NWC, NWL, JW, 4, true>;
        if (L.a.WB) kern = L.flat ? (const void*)tile_sketch_kernel<T, MODE, NWC, NWL, JW, 4, true, 0, true, true>
                                  : (const void*)tile_sketch_kernel<T, MODE, NWC, NWL, JW, 4, true, 0, true>;
#else
        if (!L.a.WB || L.flat) return fail(FDX_ERR_INVALID, "tile sketch: this layout is compiled into experiment builds only");
        kern = (const void*)tile_sketch_kernel<T, MODE, NWC, NWL, JW, 4, true, 0, true>;
#endif
    } else {
        kern = (const void*)tile_sketch_kernel<T, MODE, NWC, NWL, JW, 4, true>;
    }
    if constexpr (MODE != FDX_PRE_RAW && std::is_same<T, float>::value) {
        if (tile_logv() != 0) kern = (const void*)tile_sketch_kernel<T, MODE, NWC, NWL, JW, 4, true, 2>;
    }
    if (lds > 64 * 1024) FDX_HIP(hipFuncSetAttribute(kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    void* args[] = {(void*)&L.a, (void*)&L.Y, (void*)&L.row_map, (void*)&L.Xs, (void*)&L.H, (void*)&L.row_sumsq, (void*)&L.w_tab,
                    (void*)&L.off_tab, (void*)&L.len_tab, (void*)&L.ent_base, (void*)&L.slot_bucket, (void*)&L.log_tab,
                    (void*)&L.XA};
    FDX_HIP(hipLaunchKernel(kern, dim3(grid), dim3((NWC + NWL) * 64), args, lds, st));
    return 0;
}

template <typename T, int MODE>
static int launch_tile_cfg(const TileLaunch& L, int NWC, int TT, size_t lds, int grid, hipStream_t st) {
    if (TT == 4) {
        if constexpr (MODE == FDX_PRE_RAW) return launch_tile_wide<T, MODE, 12, 4, 22>(L, lds, grid, st);
        else return launch_tile_wide<T, MODE, 8, 0, 32>(L, lds, grid, st);
    }
    if (NWC == 16) return launch_tile_tt<T, MODE, 16, 0, 8>(L, TT, lds, grid, st);
    if (NWC == 8) return launch_tile_tt<T, MODE, 8, 2, 16>(L, TT, lds, grid, st);
    return launch_tile_tt<T, MODE, 12, 4, 11>(L, TT, lds, grid, st);
}

template <typename T>
int launch_tile_mode(const TileLaunch& L, int mode, int NWC, int TT, size_t lds, int grid, hipStream_t st) {
    switch (mode) {
        case FDX_PRE_RAW: return launch_tile_cfg<T, FDX_PRE_RAW>(L, NWC, TT, lds, grid, st);
        case FDX_PRE_LOG_CPM: return launch_tile_cfg<T, FDX_PRE_LOG_CPM>(L, NWC, TT, lds, grid, st);
        case FDX_PRE_LOG_CPM_SPARSE: return launch_tile_cfg<T, FDX_PRE_LOG_CPM_SPARSE>(L, NWC, TT, lds, grid, st);
        default: return fail(FDX_ERR_INVALID, "tile sketch: unknown preprocess mode");
    }
}
#if FDX_TILE_PART
template int launch_tile_mode<double>(const TileLaunch&, int, int, int, size_t, int, hipStream_t);
#else
extern template int launch_tile_mode<double>(const TileLaunch&, int, int, int, size_t, int, hipStream_t);

// H[:, 0..n) (type-major, row stride ldh) and row_sumsq[0..n) for the n spots listed by row_map (NULL = rows 0..n-1).
// A persistent workgroup fills its compute unit (16 waves x 127 registers): a launch of 256 leaves nothing for kernels of another
// stream until it ends.  A caller with latency-bound work to run beside the sketch (the second phase of a shard build) asks for
// some units to be left alone; results do not depend on the grid (tiles are independent).
static thread_local int t_reserve_cus = 0;
int tile_sketch_reserve_cus(int cus) {
    const int prev = t_reserve_cus;
    t_reserve_cus = std::max(0, std::min(cus, 192));
    return prev;
}

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
    a.NE = t->h.NE; a.GB = t->h.GB; a.NBLK = t->h.NBLK; a.RS = t->RS; a.jw_used = t->h.jw_used; a.NST = t->NST; a.WB = t->WB;
#ifdef FDX_TILE_EXPERIMENT
    if (const char* e = fdx::exp_env("FDX_TILE_DBG")) a.dbg = atoi(e);
#endif
    if (t->fh.NSP > 0) {
        a.NE = (int)t->fh.off.size();
        a.NSP = t->fh.NSP; a.GROW = t->fh.GROW;
        L.flat = true;
    }
    L.Y = Y; L.row_map = row_map; L.Xs = Xs; L.H = H; L.row_sumsq = row_sumsq;
    L.w_tab = t->w.as<double>(); L.off_tab = t->off.as<unsigned short>(); L.len_tab = t->len.as<unsigned char>();
    L.ent_base = t->ent_base.as<int>(); L.slot_bucket = t->slot_bucket.as<int>();
    L.log_tab = nullptr;
    if (mode != FDX_PRE_RAW) {
        L.log_tab = log_table_dev(st);
        if (!L.log_tab) return fail(FDX_ERR_HIP, "tile sketch: log table upload failed");
    }
    const long long n_tiles = (n + TILE_ROWS - 1) / TILE_ROWS;
    const int grid = (int)std::min<long long>(n_tiles, 256 - t_reserve_cus);
    DevBuf xa;                                                              // wide form: X_sketch in operand order
    L.XA = nullptr;
    // narrow log modes: the float64 chain needs the registers the operands would take (operand copy in L2, fetched per
    // group); the float32-class chain leaves room for them (127 registers, no spills; the fetches cost 0.5 ms per 1M spots)
    const bool f32log = dtype == FDX_F32 && mode != FDX_PRE_RAW && tile_logv() != 0;
    const bool narrow_avl2 = mode != FDX_PRE_RAW && t->NWC == 16 && (f32log ? fdx::exp_env("FDX_TILE_AVL2") != nullptr : !fdx::exp_env("FDX_TILE_NO_AVL2"));
    if (t->wide || narrow_avl2) {
        const int n_groups = t->NWC * t->JW;
        FDX_TRY(xa.alloc((size_t)n_groups * t->TT * 64 * sizeof(double)));
        hipLaunchKernelGGL(tile_xa_kernel, dim3(ceil_div((long long)n_groups * t->TT * 64, 256)), dim3(256), 0, st, Xs,
                           t->slot_bucket.as<int>(), K, d, n_groups, t->TT, xa.as<double>());
        FDX_CHECK_LAUNCH();
        L.XA = xa.as<double>();
    }
    if (dtype == FDX_F32) return launch_tile_mode<float>(L, mode, t->NWC, t->TT, t->lds, grid, st);
    return launch_tile_mode<double>(L, mode, t->NWC, t->TT, t->lds, grid, st);
}

// The one-kernel sketch -> H stage: the tile kernel for every shape it takes (a CountSketch with d <= 1056, K <= 64, rows whole
// 16-byte vectors); everything else runs the two-kernel path (sketch_rows_* + xyt_split), also selected by FDX_NO_FUSED=1.
// (The round-1 atomic fused kernel that used to serve odd row lengths behind this pair was slower than the two-kernel path it
// replaced there and is gone.)
bool fused_sketch_contract_ok(int dtype, long long ldy, const void* Y, int G, int d, int K, int mode, const SketchPlanDev& plan,
                              hipStream_t st) {
    if (fdx::env("FDX_NO_FUSED")) return false;
    return tile_sketch_ok(dtype, ldy, Y, G, d, K, mode, plan, st);
}

int launch_sketch_contract(const void* Y, int dtype, long long ldy, const int* row_map, long long n, int G, int d, int mode,
                           const SketchPlanDev& plan, const double* Xs, int K, double* H, long long ldh, double* row_sumsq,
                           hipStream_t st) {
    if (n <= 0) return 0;
    if (!tile_sketch_ok(dtype, ldy, Y, G, d, K, mode, plan, st)) return fail(FDX_ERR_INVALID, "sketch -> H: no one-kernel form for this shape");
    return launch_tile_sketch(Y, dtype, ldy, row_map, n, G, d, mode, plan, Xs, K, H, ldh, row_sumsq, st);
}
#endif  // FDX_TILE_PART

}  // namespace fdx

#if !FDX_TILE_PART
// include/fdx.h: the float32-class log1p of the tile kernel on a host array (accuracy tests)
namespace fdx {
__global__ void log1p_f32_probe_kernel(const float* __restrict__ y, float scale, long long n, float* __restrict__ out) {
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) out[i] = tile_log1p_f32(y[i], scale);
}
}  // namespace fdx

extern "C" int fdx_log1p_f32(const float* y, float scale, int64_t n, float* out) {
    using namespace fdx;
    FDX_REQUIRE(n >= 0 && (n == 0 || (y && out)), "fdx_log1p_f32: null argument");
    if (n == 0) return 0;
    DevBuf in, res;
    FDX_TRY(in.alloc((size_t)n * 4));
    FDX_TRY(res.alloc((size_t)n * 4));
    FDX_HIP(hipMemcpy(in.p, y, (size_t)n * 4, hipMemcpyHostToDevice));
    hipLaunchKernelGGL(log1p_f32_probe_kernel, dim3((unsigned)ceil_div(n, 256)), dim3(256), 0, nullptr, in.as<float>(), scale,
                       (long long)n, res.as<float>());
    FDX_CHECK_LAUNCH();
    FDX_HIP(hipMemcpy(out, res.p, (size_t)n * 4, hipMemcpyDeviceToHost));
    return 0;
}

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
#endif  // !FDX_TILE_PART
