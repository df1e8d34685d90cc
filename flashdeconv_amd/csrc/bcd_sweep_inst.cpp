// Block-coordinate-descent sweep for the graph-regularised NNLS (the hot loop).
//
// Replaces the reference's numba kernels
//   flashdeconv/core/solver.py:104-184  _bcd_iteration_fused  (Jacobi over spots, prange)
//   flashdeconv/core/solver.py:29-101   update_spot_with_Xty  (Gauss-Seidel over cell types)
//   flashdeconv/core/solver.py:18-26    soft_threshold
// and the host-side convergence reduction of core/solver.py:395-397.
//
// Mapping (gfx950, wave64): ONE LANE = ONE SPOT, one wavefront = one 64-spot slice of the sliced-ELL
// graph.  The K abundances of the spot live in VGPRs for the whole sweep (the coordinate steps are
// sequential in k, so a lane-per-type mapping would leave 63/64 of the VALU idle - see DESIGN.md).
// beta and H are stored type-major ("SoA", (K, ld)) so that every own-row access of a wave is one
// fully coalesced 512-byte transaction; neighbour rows are gathered from the same planes and are
// served by L2 because the spots are in Morton order.  XtX (K x K) is wave-uniform and is read
// through the scalar cache (s_load) straight into the SGPR operand of v_fma_f64.
//
// Arithmetic per spot (float64, IEEE division, no fast-math):
//   nbr_k  = sum_{j in N(i)} beta_in[j,k]                      (CSR order, padded with exact +0.0)
//   r_k    = sum_j XtX[k,j] * b_j       with b_j already updated for j < k   (maintained residual of
//            solver.py:72,96-99 evaluated on demand: same value, K^2 instead of 1.5 K^2 FMAs, no r[] array)
//   res    = H[k,i] - r_k + XtX[k,k]*b_k (+ lambda*nbr_k if deg>0)          (solver.py:79-83)
//   b_k    = den > 1e-10 ? max(0, soft(res, rho)/den) : 0 ,  den = XtX[k,k] + lambda*deg  (solver.py:86-93)
// Convergence statistics max_i max_k|b_new-b_old| and max_i max_k|b_old| (solver.py:173-184) are reduced
// with wave shuffles and one integer atomicMax per wave into 64 slots (order-free, hence deterministic).
// The NEXT sweep (or the finishing kernel) folds the 64 slots and evaluates
//   rel_change = max_diff / (max_abs_old + 1e-10) < tol                       (solver.py:395-397,409)
// on the device, so a converged solve turns the already-queued sweeps into no-ops without a host round trip.
#include "fdx_env.h"
#include <algorithm>
#include <cstdlib>
#include "bcd_device.h"
#include "fdx_graph.h"

// This translation unit is compiled several times (csrc/Makefile) with -DFDX_PART=p -DFDX_K_LO=a -DFDX_K_HI=b so the
// 64 register-resident instantiations build in parallel; each build exports bcd_sweep_dispatch_part<p>().
#ifndef FDX_PART
#error "compile with -DFDX_PART=<n> -DFDX_K_LO=<lo> -DFDX_K_HI=<hi>"
#endif

namespace fdx {

// Cell types per chunk.  The K abundances of a spot are registers (2 K), every type of the chunk adds 6 more (neighbour sum,
// H value, halo value in flight), and 168 registers are the limit for three waves per SIMD, which is what decides the speed:
// K = 40 at 1M spots takes 417 us with chunks of 8 (171 registers, two waves) and 269 us with chunks of 5 (167).  The
// table is the largest chunk that stays within 168 registers, per K, from tools/sweep_regs.py (compile and count; timings
// of neighbouring chunk sizes on MI355X agree with that rule for K = 30 ... 38); from K = 52 on nothing fits and the
// chunk of 8 at two waves is kept.
constexpr int sweep_chunk(int K) {
#ifdef FDX_KC_OVERRIDE                       // tools/sweep_regs.py: register count per (K, chunk)
    return K < FDX_KC_OVERRIDE ? K : FDX_KC_OVERRIDE;
#endif
    // above 64 (the padded sizes of solver_padded_K) the kernel is held to 256 registers = two waves per SIMD up to 96 types: chunks
    // of 6 / 4 / 4 / 2 at 72 / 80 / 88 / 96 (239 / 239 / 255 / 256 registers, no spills: tools/sweep_regs.py); 112 types would run one wave
    // per SIMD with 150 spills - the LDS-resident sweep takes over from 97 (bcd_kernels.cpp)
    return K < 8 ? K : K <= 29 ? 8 : K <= 33 ? 7 : K <= 37 ? 6 : K <= 41 ? 5 : K <= 45 ? 4 : K <= 49 ? 3 : K <= 51 ? 2 : K <= 64 ? 8 : K <= 72 ? 6 : K <= 88 ? 4 : K <= 96 ? 2 : 8;
}

template <int K, int KC>
__global__ __launch_bounds__(256) void bcd_sweep_kernel(
    const double* __restrict__ H, const double* __restrict__ XtX, const double* __restrict__ beta_in,
    double* __restrict__ beta_out, const int* __restrict__ ell_base, const int* __restrict__ slice_off,
    const int* __restrict__ deg, unsigned long long* __restrict__ stats, double* __restrict__ rel_change,
    const double lambda, const double rho, const double tol, const int ldh, const int ld_, const int n,
    const int n_slices, const int it) {
    const int lane = threadIdx.x & 63;
    const int wib = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);  // wave-uniform, provably so

    if (it > 0) {  // device-side stopping rule on the previous sweep's statistics
        const double rc = fold_rel_change(stats + (size_t)(it - 1) * 128, lane);
        if (blockIdx.x == 0 && threadIdx.x == 0) rel_change[it - 1] = rc;
        if (rc < tol) return;
    }

    const int slice = xcd_remap(blockIdx.x, gridDim.x) * 4 + wib;
    if (slice >= n_slices) return;
    // Lanes past the last spot mirror spot n-1 (same loads, same value stored to the same address), so the
    // whole wave stays convergent and no lane needs predication.
    const int i = min(slice * 64 + lane, n - 1);
    const size_t ld = (size_t)ld_;

    const int w0 = slice_off[slice];
    const int w = slice_off[slice + 1] - w0;  // wave-uniform ELL width of this slice

    double b[K];
#pragma unroll
    for (int k = 0; k < K; ++k) b[k] = beta_in[k * ld + i];
    const int* ell = ell_base + (size_t)w0 * 64 + (i & 63);  // mirrored lanes read the mirrored spot's row
    const int dg = deg[i];
    const double lam_deg = lambda * (double)dg;
    const double lam_eff = (dg > 0) ? lambda : 0.0;  // spatial term only when the spot has neighbours

    double dmax = 0.0, amax = 0.0;
    // Cell types are processed in chunks of KC: gather the chunk's neighbour sums (KC independent
    // loads in flight per neighbour), then run the chunk's sequential coordinate steps.
#pragma unroll
    for (int kc = 0; kc < K; kc += KC) {
        double c[KC];
#pragma unroll
        for (int q = 0; q < KC; ++q) c[q] = 0.0;
#pragma unroll 2
        for (int m = 0; m < w; ++m) {
            const int j = ell[(size_t)m * 64];
#pragma unroll
            for (int q = 0; q < KC; ++q)
                if (kc + q < K) c[q] += beta_in[(kc + q) * ld + j];
        }
#pragma unroll
        for (int q = 0; q < KC; ++q) {
            const int k = kc + q;
            if (k < K) {
                const double h = H[k * (size_t)ldh + i];
                const double* g = XtX + k * K;
                double r0 = 0.0, r1 = 0.0;
#pragma unroll
                for (int j = 0; j + 1 < K; j += 2) {
                    r0 = fma(g[j], b[j], r0);
                    r1 = fma(g[j + 1], b[j + 1], r1);
                }
                if (K & 1) r0 = fma(g[K - 1], b[K - 1], r0);
                const double gkk = g[k];
                const double old = b[k];
                const double res = (h - (r0 + r1) + gkk * old) + lam_eff * c[q];
                const double den = gkk + lam_deg;
                const double st = res > rho ? res - rho : (res < -rho ? res + rho : 0.0);
                const double qv = fmax(0.0, st / den);
                const double nw = (den > 1e-10) ? qv : 0.0;
                dmax = fmax(dmax, fabs(nw - old));
                amax = fmax(amax, fabs(old));
                b[k] = nw;
                beta_out[k * ld + i] = nw;
            }
        }
    }
    dmax = wave_max(dmax);
    amax = wave_max(amax);
    if (lane == 0) {
        unsigned long long* s = stats + (size_t)it * 128;
        const int slot = slice & 63;
        atomicMax(s + slot, (unsigned long long)__double_as_longlong(dmax));
        atomicMax(s + 64 + slot, (unsigned long long)__double_as_longlong(amax));
    }
}

// A type plane's base address as a scalar the compiler cannot fold back into a per-lane 64-bit address (it would: base + lane
// offset first, plane stride added per access with vector adds): opaque in scalar registers, global address space kept.
__device__ __forceinline__ const char __attribute__((address_space(1)))* plane_base(const double* p) {
    unsigned long long v = (unsigned long long)p;
    asm("" : "+s"(v));
    return (const char __attribute__((address_space(1)))*)v;
}

// ... and the lane's 32-bit offset re-read where it is used: the global instructions take `scalar base + 32-bit lane offset`
// only if the zero extension is visible in the same basic block (hoisted, it is a 64-bit value like any other).
__device__ __forceinline__ unsigned lane_off(unsigned off) {
    asm("" : "+v"(off));
    return off;
}

// LDS-tiled variant (graphs built from coordinates): a workgroup owns a TILE of 256 consecutive Morton-ordered spots.
// Per chunk of KC cell types the tile's own old abundances (from registers) and its halo (the neighbour positions outside
// the tile, one coalesced-ish global gather per halo spot and type) are staged in LDS, and all neighbour sums are then
// served by ds_read_b64 from tile-local slots - ~0.5 global gathers per spot and type instead of ~11.  Arithmetic and
// summation order are identical to bcd_sweep_kernel, so both variants produce the same bits.
// QUAD = false (objective above 64 types): the quadratic term beta' XtX beta is left to launch_beta_quad - with the K products per type
// the objective variant of the padded sizes spills at 256 registers (88: 7 ... 112: hundreds) and cost 1.7 x a sweep.
// INIT = true (first sweep of a solve that starts from the uniform 1/K, core/solver.py:372): the old abundances are the constant
// init_v everywhere - own spots and halo alike - and are not read: the start vector is never written to HBM (240 MB at 1M x 30)
// and the first sweep moves two thirds of a sweep's bytes.  Same arithmetic on the same values: same bits.
#ifndef FDX_SWEEP_WPE
#define FDX_SWEEP_WPE 1
#endif
template <int K, int KC, bool OBJ, bool QUAD = true, bool INIT = false>
__global__ __launch_bounds__(256, ((OBJ && K > 40 && K <= 64) || (K > 64 && K <= 96)) ? 2 : (OBJ ? 1 : FDX_SWEEP_WPE)) void bcd_sweep_tiled_kernel(
    const double* __restrict__ H, const double* __restrict__ XtX, const double* __restrict__ beta_in,
    double* __restrict__ beta_out, const unsigned short* __restrict__ ell_local, const int* __restrict__ slice_off,
    const int* __restrict__ deg, const int* __restrict__ tile_halo, const int* __restrict__ tile_hcnt,
    unsigned long long* __restrict__ stats, double* __restrict__ rel_change, const double lambda, const double rho,
    const double tol, const int ldh, const int ld_, const int n, const int S, const int it,
    const int* __restrict__ tile_list, const double init_v, const int* __restrict__ send_head, const int4* __restrict__ send_ent,
    double* __restrict__ send_buf) {
    // tile_list != NULL: the grid covers the listed tiles only (sharded solve: boundary tiles first, interior tiles while
    // the halo is on the wire); NULL: all tiles, XCD-contiguous remap.
    // OBJ = true turns the same traversal into the objective evaluation (core/solver.py:269-284): no update, no store;
    // `rel_change` then receives the per-block partial sums (cross, quad, spatial, l1) at [4*block + o].
    extern __shared__ __attribute__((aligned(16))) double lds[];   // [KC][S]: 256 own | halo | zero slot
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    // The tile's small tables are requested together with the previous sweep's statistics, ahead of the stopping rule: the
    // addresses of everything else (slot table, halo values) depend on them, and behind the rule they would be a third
    // round trip before the first chunk can be staged (a workgroup lives ~25 us; each exposed round trip is ~1 us of it).
    const int tile = tile_list ? tile_list[xcd_remap(blockIdx.x, gridDim.x)] : xcd_remap(blockIdx.x, gridDim.x);
    const bool real = (tile * 256 + tid) < n;
    const int i = min(tile * 256 + tid, n - 1);   // lanes past the last spot mirror spot n-1
    const size_t ld = (size_t)ld_;
    const int slice = __builtin_amdgcn_readfirstlane(i >> 6);
    const int w0 = slice_off[slice];
    const int w = slice_off[slice + 1] - w0;
    const int Ht = tile_hcnt[tile];
    const int* halo = tile_halo + (size_t)tile * FDX_TILE_HALO_CAP;
    const int dg = deg[i];
    int hidx0 = halo[tid];                        // the table has FDX_TILE_HALO_CAP >= 256 entries per tile; past Ht: unused
    if (!OBJ && it > 0) {
        const double rc = fold_rel_change(stats + (size_t)(it - 1) * 128, lane);
        if (blockIdx.x == 0 && tid == 0) rel_change[it - 1] = rc;
        asm volatile("" :: "v"(dg), "v"(hidx0), "s"(w0), "s"(w), "s"(Ht));   // keeps the loads above on this side of the branch
        if (rc < tol) return;          // uniform over the whole grid
    }
    hidx0 = (tid < Ht) ? hidx0 : 0;

    // Addresses inside a type plane: uniform plane base (scalar registers, scalar arithmetic) + this lane's 32-bit byte
    // offset - the form the global instructions take directly; 64-bit per-lane addresses cost two vector adds per access
    // (six per coordinate step) and a register pair each.  The launcher guarantees 8 * ld < 4 GB.
    const unsigned ioff = (unsigned)i * 8u, hoff = (unsigned)hidx0 * 8u;
#define PLANE_AT(base, off) (*(const double __attribute__((address_space(1)))*)(plane_base(base) + lane_off(off)))
    double b[K];
#pragma unroll
    for (int k = 0; k < K; ++k) b[k] = INIT ? init_v : PLANE_AT(beta_in + k * ld, ioff);
    const unsigned short* ell = ell_local + (size_t)w0 * 64 + (i & 63);
    const double lam_deg = lambda * (double)dg;
    const double lam_eff = (dg > 0) ? lambda : 0.0;
    // tile-local neighbour slots of this spot: the first 16 live in registers for all chunks (two 16-bit slots per VGPR),
    // wider slices (rare) read the rest from memory
    unsigned slots[8];
    // all 16 loads go out together whatever the slice width (the table is padded by 16 rows, graph_kernels.cpp); entries
    // past the width belong to the next slice: they are replaced by the tile's ZERO slot, so that the neighbour sums below can
    // run over whole groups of four positions - a position past the width adds an exact +0.0 - instead of asking "m < w?" at
    // each of the 16 positions of every chunk (the compiler kept those 16 wave-uniform answers as lane masks, 32 scalar
    // registers it then spilled into vector lanes: two v_readlane, a select and a compare per position and chunk, ~400 of the
    // 2830 vector instructions a spot costs at 30 types)
    const unsigned zslot = (unsigned)(256 + Ht);
#pragma unroll
    for (int m2 = 0; m2 < 8; ++m2) {
        unsigned lo = (unsigned)__builtin_nontemporal_load(&ell[(size_t)(2 * m2) * 64]);
        unsigned hi = (unsigned)__builtin_nontemporal_load(&ell[(size_t)(2 * m2 + 1) * 64]);
        lo = (2 * m2 < w) ? lo : zslot;
        hi = (2 * m2 + 1 < w) ? hi : zslot;
        slots[m2] = lo | (hi << 16);
    }
    // Software pipeline over the chunks: what chunk c+1 needs from global memory - the old values of the first 256 halo
    // spots (hv, one per thread) and the spot's own H values (hreg) - is requested during the coordinate steps of chunk
    // c, two loads after each step, into the registers that step has just finished with (its neighbour sum and its H
    // value): halo values in the first half of the chunk (they are written to LDS right after it), H in the second half
    // (used after the next gather).  No load is waited for where it is issued.
    double hv[KC], hreg[KC];
#pragma unroll
    for (int q = 0; q < KC; ++q) {
        hv[q] = INIT ? init_v : PLANE_AT(beta_in + q * ld, hoff);                  // threads past Ht read spot 0: harmless
        hreg[q] = __builtin_nontemporal_load(&PLANE_AT(H + q * (size_t)ldh, ioff));             // streamed once per sweep: keep L2 for beta_in (halo re-use)
    }

    double dmax = 0.0, amax = 0.0;
    double o_cross = 0.0, o_quad = 0.0, o_spat = 0.0, o_l1 = 0.0;
#pragma unroll
    for (int kc = 0; kc < K; kc += KC) {
        // ---- stage old values of this chunk: own from registers, halo from global
#pragma unroll
        for (int q = 0; q < KC; ++q)
            if (kc + q < K) lds[q * S + tid] = b[kc + q];
        if (tid < Ht) {
#pragma unroll
            for (int q = 0; q < KC; ++q)
                if (kc + q < K) lds[q * S + 256 + tid] = hv[q];
        }
        for (int h = tid + 256; h < Ht; h += 256) {
            const int j = halo[h];
#pragma unroll
            for (int q = 0; q < KC; ++q)
                if (kc + q < K) lds[q * S + 256 + h] = INIT ? init_v : beta_in[(kc + q) * ld + j];
        }
        if (tid < KC) lds[tid * S + 256 + Ht] = 0.0;
        __syncthreads();
        double c[KC];
#pragma unroll
        for (int q = 0; q < KC; ++q) c[q] = 0.0;
#pragma unroll
        for (int m4 = 0; m4 < 16; m4 += 4) {
            if (m4 < w) {                                  // wave-uniform; positions past the width hold the zero slot
#pragma unroll
                for (int m = m4; m < m4 + 4; ++m) {
                    const int slot = (int)((slots[m >> 1] >> (16 * (m & 1))) & 0xffffu);
#pragma unroll
                    for (int q = 0; q < KC; ++q)
                        if (kc + q < K) c[q] += lds[q * S + slot];
                }
            }
        }
        for (int m = 16; m < w; ++m) {
            const int slot = ell[(size_t)m * 64];
#pragma unroll
            for (int q = 0; q < KC; ++q)
                if (kc + q < K) c[q] += lds[q * S + slot];
        }
        __syncthreads();
#pragma unroll
        for (int q = 0; q < KC; ++q) {
            const int k = kc + q;
            if (k < K) {
                const double h = hreg[q];
                const double* g = XtX + k * K;
                double r0 = 0.0, r1 = 0.0;
                if (!OBJ || QUAD) {
#pragma unroll
                    for (int j = 0; j + 1 < K; j += 2) {
                        r0 = fma(g[j], b[j], r0);
                        r1 = fma(g[j + 1], b[j + 1], r1);
                    }
                    if (K & 1) r0 = fma(g[K - 1], b[K - 1], r0);
                }
                const double gkk = (!OBJ || QUAD) ? g[k] : 0.0;
                const double old = b[k];
                if (OBJ) {
                    o_cross = fma(old, h, o_cross);
                    if (QUAD) o_quad = fma(old, r0 + r1, o_quad);
                    o_spat = fma(old, (double)dg * old - c[q], o_spat);
                    o_l1 += fabs(old);
                    asm volatile("" : "+v"(o_quad));   // pins this type's K products here: nothing else orders them in this variant, and sunk to the end they keep all of XtX live
                } else {
                    const double res = (h - (r0 + r1) + gkk * old) + lam_eff * c[q];
                    const double den = gkk + lam_deg;
                    const double st = res > rho ? res - rho : (res < -rho ? res + rho : 0.0);
                    const double qv = fmax(0.0, st / den);
                    const double nw = (den > 1e-10) ? qv : 0.0;
                    dmax = fmax(dmax, fabs(nw - old));
                    amax = fmax(amax, fabs(old));
                    b[k] = nw;
                    __builtin_nontemporal_store(nw, (double __attribute__((address_space(1)))*)(plane_base(beta_out + k * ld) + lane_off(ioff)));
                }
            }
            if (kc + KC < K) {                              // next chunk's loads 2q and 2q+1 (0..KC-1: halo, KC..2KC-1: H)
#pragma unroll
                for (int t = 2 * q; t < 2 * q + 2; ++t) {
                    const int j = t < KC ? t : t - KC;
                    if (kc + KC + j < K) {
                        if (t < KC) { if (!INIT) hv[j] = PLANE_AT(beta_in + (kc + KC + j) * ld, hoff); }
                        else hreg[j] = __builtin_nontemporal_load(&PLANE_AT(H + (kc + KC + j) * (size_t)ldh, ioff));
                    }
                }
                __builtin_amdgcn_sched_barrier(0);
            }
        }
    }
    if (OBJ) {
        if (!real) { o_cross = o_quad = o_spat = o_l1 = 0.0; }
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) {
            o_cross += __shfl_xor(o_cross, off, 64);
            o_quad += __shfl_xor(o_quad, off, 64);
            o_spat += __shfl_xor(o_spat, off, 64);
            o_l1 += __shfl_xor(o_l1, off, 64);
        }
        __syncthreads();                    // lds is free again after the last chunk
        if (lane == 0) { lds[(tid >> 6) * 4 + 0] = o_cross; lds[(tid >> 6) * 4 + 1] = o_quad; lds[(tid >> 6) * 4 + 2] = o_spat; lds[(tid >> 6) * 4 + 3] = o_l1; }
        __syncthreads();
        if (tid < 4) rel_change[(size_t)tile * 4 + tid] = ((lds[tid] + lds[4 + tid]) + lds[8 + tid]) + lds[12 + tid];
        return;
    }
    // sharded solve: a row some peer needs is written into the send staging here, where its new abundances are in registers (the
    // separate pack launch was one of four per iteration on a 125k-spot shard whose sweep lasts 36 us)
    if (!OBJ && send_head != nullptr) {
        int hd = real ? send_head[i] : 0;
        while (hd != 0) {
            const int4 e = send_ent[hd - 1];
            double* o = send_buf + (size_t)K * e.x + e.z;
#pragma unroll
            for (int k = 0; k < K; ++k) o[(size_t)k * e.y] = b[k];
            hd = e.w;
        }
    }
    dmax = wave_max(dmax);
    amax = wave_max(amax);
    if (lane == 0) {
        unsigned long long* s = stats + (size_t)it * 128;
        const int slot = slice & 63;
        atomicMax(s + slot, (unsigned long long)__double_as_longlong(dmax));
        atomicMax(s + 64 + slot, (unsigned long long)__double_as_longlong(amax));
    }
}

template <int K>
static void launch_k(const BcdSweepArgs& a, hipStream_t st) {
    if (a.tiled) {
        constexpr int KC = sweep_chunk(K);
        const int S = 256 + a.halo_max + 1;
        const size_t lds = (size_t)KC * S * sizeof(double);
        if (lds <= 64 * 1024 && (long long)a.ld * 8 < (1LL << 32) && (long long)a.ldh * 8 < (1LL << 32)) {
            if (a.objective) {
                if constexpr (K > 64) {                    // always with skip_quad (solver.cpp): only that variant is instantiated
                    hipLaunchKernelGGL((bcd_sweep_tiled_kernel<K, KC, true, false>), dim3(a.n_tiles), dim3(256), lds, st, a.H, a.XtX,
                                       a.beta_in, a.beta_out, a.ell_local, a.slice_off, a.deg, a.tile_halo, a.tile_hcnt,
                                       a.stats, a.rel_change, a.lambda, a.rho, a.tol, a.ldh, a.ld, a.n, S, a.it, nullptr, 0.0, a.send_head, a.send_ent, a.send_buf);
                } else {
                    hipLaunchKernelGGL((bcd_sweep_tiled_kernel<K, KC, true>), dim3(a.n_tiles), dim3(256), lds, st, a.H, a.XtX,
                                       a.beta_in, a.beta_out, a.ell_local, a.slice_off, a.deg, a.tile_halo, a.tile_hcnt,
                                       a.stats, a.rel_change, a.lambda, a.rho, a.tol, a.ldh, a.ld, a.n, S, a.it, nullptr, 0.0, a.send_head, a.send_ent, a.send_buf);
                }
            }
            else if (a.tile_list) {
                if (a.n_list > 0)
                    hipLaunchKernelGGL((bcd_sweep_tiled_kernel<K, KC, false>), dim3(a.n_list), dim3(256), lds, st, a.H, a.XtX,
                                       a.beta_in, a.beta_out, a.ell_local, a.slice_off, a.deg, a.tile_halo, a.tile_hcnt,
                                       a.stats, a.rel_change, a.lambda, a.rho, a.tol, a.ldh, a.ld, a.n, S, a.it, a.tile_list, 0.0, a.send_head, a.send_ent, a.send_buf);
            } else if (a.init_uniform != 0.0) {
                if constexpr (K <= 64) {
                    hipLaunchKernelGGL((bcd_sweep_tiled_kernel<K, KC, false, true, true>), dim3(a.n_tiles), dim3(256), lds, st, a.H, a.XtX,
                                       a.beta_in, a.beta_out, a.ell_local, a.slice_off, a.deg, a.tile_halo, a.tile_hcnt,
                                       a.stats, a.rel_change, a.lambda, a.rho, a.tol, a.ldh, a.ld, a.n, S, a.it, nullptr, a.init_uniform, a.send_head, a.send_ent, a.send_buf);
                }
            } else {
                // FDX_SWEEP_LDS_PAD_KB (diagnostic): unused dynamic LDS on top, to time this kernel at the occupancy a fused
                // two-sweep kernel would have (its intermediate iterate of tile + first ring lives in LDS: DESIGN section 7)
                static const int pad_kb = fdx::exp_env("FDX_SWEEP_LDS_PAD_KB") ? atoi(fdx::exp_env("FDX_SWEEP_LDS_PAD_KB")) : 0;
                const size_t lds_launch = lds + (size_t)std::max(0, pad_kb) * 1024;
                if (pad_kb > 0 && lds_launch > 64 * 1024)
                    (void)hipFuncSetAttribute((const void*)bcd_sweep_tiled_kernel<K, KC, false>, hipFuncAttributeMaxDynamicSharedMemorySize,
                                              (int)lds_launch);
                hipLaunchKernelGGL((bcd_sweep_tiled_kernel<K, KC, false>), dim3(a.n_tiles), dim3(256), lds_launch, st, a.H, a.XtX,
                                   a.beta_in, a.beta_out, a.ell_local, a.slice_off, a.deg, a.tile_halo, a.tile_hcnt,
                                   a.stats, a.rel_change, a.lambda, a.rho, a.tol, a.ldh, a.ld, a.n, S, a.it, nullptr, 0.0, a.send_head, a.send_ent, a.send_buf);
            }
            return;
        }
    }
    const int nblk = ceil_div(a.n_slices, 4);
    hipLaunchKernelGGL((bcd_sweep_kernel<K, sweep_chunk(K)>), dim3(nblk), dim3(256), 0, st, a.H, a.XtX, a.beta_in,
                       a.beta_out, a.ell, a.slice_off, a.deg, a.stats, a.rel_change, a.lambda, a.rho, a.tol, a.ldh,
                       a.ld, a.n, a.n_slices, a.it);
}

#ifndef FDX_K_STEP
#define FDX_K_STEP 1
#endif
template <int... Is>
static bool dispatch_range(const BcdSweepArgs& a, hipStream_t st, std::integer_sequence<int, Is...>) {
    bool hit = false;
    (void)std::initializer_list<int>{((a.K == FDX_K_LO + Is * FDX_K_STEP) ? (launch_k<FDX_K_LO + Is * FDX_K_STEP>(a, st), hit = true, 0) : 0)...};
    return hit;
}

#define FDX_CAT2(a, b) a##b
#define FDX_CAT(a, b) FDX_CAT2(a, b)
bool FDX_CAT(bcd_sweep_dispatch_part, FDX_PART)(const BcdSweepArgs& a, hipStream_t st) {
    if (a.K < FDX_K_LO || a.K > FDX_K_HI) return false;
    return dispatch_range(a, st, std::make_integer_sequence<int, (FDX_K_HI - FDX_K_LO) / FDX_K_STEP + 1>{});
}

}  // namespace fdx
