// Device helpers of the tile kernel (tile_kernels.cpp).
#pragma once
#include <hip/hip_runtime.h>

#include "device_math.h"
#include "fdx_internal.h"

namespace fdx {

typedef double double4_t __attribute__((ext_vector_type(4)));

constexpr int TILE_ROWS = 16;
// Table-driven log1p: 1 + x is reduced by a reciprocal rounded to LOG_TAB_MB explicit mantissa bits, taken from a table of
// -log(reciprocal) over 15 binades, to 1 + r with |r| <= 2^-(LOG_TAB_MB + 1); a polynomial of degree LOG_TAB_MB == 7 ? 6 : 7
// in r finishes it (truncation below 2^-56).  6 bits: 961 entries = 7.7 KB instead of 15.4 KB and one more fma per element -
// the 7.7 KB are what lets the tile kernel stage 2000 float32 genes in two column blocks instead of three (a third fewer
// group prologues and tails, 4 % less lockstep padding).
constexpr int LOG_TAB_MB = 6;
constexpr int LOG_TAB_SHIFT = 23 - LOG_TAB_MB;                    // bits of the float reciprocal below the table index
constexpr int LOG_TAB_N = 15 * (1 << LOG_TAB_MB) + 1;             // reciprocals in [2^-15, 1]
// The reciprocal is formed of (1 + x) * 2^-65, not of 1 + x: its exponent field then starts at 177 instead of 112, which puts
// (table address - index bytes of the first entry) inside the 16-bit offset field of ds_read_b64 with the table at the top of
// the LDS.  The scaling is free: it is folded into the row's scale (and undone inside the fma that forms r).
constexpr int LOG_TAB_EXP_SHIFT = 65;
constexpr int LOG_TAB_BASE = (112 + LOG_TAB_EXP_SHIFT) << LOG_TAB_MB;   // (bits of 2^(65 - 15)) >> LOG_TAB_SHIFT
constexpr unsigned LOG_TAB_ROUND = 1u << (LOG_TAB_SHIFT - 1);
constexpr unsigned LOG_TAB_MASK = ~((1u << LOG_TAB_SHIFT) - 1u);
// The table sits at a FIXED place, the top of the 160 KB: its address is then (rounded reciprocal bits >> (SHIFT - 3)) plus a
// compile-time constant that fits the 16-bit offset field of ds_read_b64 - no base add, no index mask per element.
constexpr int LOG_TAB_LDS = 160 * 1024 - ((LOG_TAB_N * 8 + 15) & ~15);
static_assert(LOG_TAB_LDS - LOG_TAB_BASE * 8 >= 0 && LOG_TAB_LDS - LOG_TAB_BASE * 8 <= 65535, "log table offset must fit ds_read's offset field");
#define FDX_LOG_DOWN 0x1p-65
#define FDX_LOG_DOWN_F 0x1p-65f

// d = a * b + c as one VOP3 instruction with the addend in its own register
__device__ __forceinline__ double fma3(double a, double b, double c) {
#if defined(__HIP_DEVICE_COMPILE__)
    double d;
    asm("v_fma_f64 %0, %1, %2, %3" : "=v"(d) : "v"(a), "v"(b), "v"(c));
    return d;
#else
    return fma(a, b, c);
#endif
}

// The polynomial's non-inline coefficients as register values the compiler cannot re-create: written as literals they are
// re-materialised (v_mov_b64 from scalar registers) at every use - four extra instructions per two elements in the gather loop.
struct LogConsts { double c7, c6, c5, c3; };
__device__ __forceinline__ LogConsts log_consts() {
    LogConsts c{1.0 / 7.0, -1.0 / 6.0, 0.2, 1.0 / 3.0};
#if defined(__HIP_DEVICE_COMPILE__)
    asm volatile("" : "+v"(c.c7), "+v"(c.c6), "+v"(c.c5), "+v"(c.c3));
#endif
    return c;
}

// log1p(x) for x in [0, 32000): see the header.  logt[i] = -log(c_i), c_i = 2^-65 x the value with bit pattern
// (LOG_TAB_BASE + i) << LOG_TAB_SHIFT.  xs = x * 2^-65; ufs = (1 + x) * 2^-65 to float accuracy (only the rounded
// reciprocal is taken from it).
__device__ __forceinline__ double tile_log1p_core(double xs, float ufs, const LogConsts& lc) {
    unsigned bits = __float_as_uint(__builtin_amdgcn_rcpf(ufs));
    bits = (bits + LOG_TAB_ROUND) & LOG_TAB_MASK;            // reciprocal rounded to LOG_TAB_MB + 1 significant bits
    const double inv = (double)__uint_as_float(bits);        // 2^65 / (1 + x), rounded
    const double r = fma(xs, inv, fma(FDX_LOG_DOWN, inv, -1.0));   // (1 + x) * c - 1 with one rounding (c - 1 is exact)
    // the dynamic LDS segment starts at LDS address 0 (no static __shared__ in these kernels): absolute address
    typedef const double __attribute__((address_space(3))) * lds_cdouble_p;
    const double t = *(lds_cdouble_p)(size_t)((bits >> (LOG_TAB_SHIFT - 3)) + (unsigned)(LOG_TAB_LDS - LOG_TAB_BASE * 8));
    // Horner steps whose addend is not an inline constant: three-operand v_fma_f64 on registers (left to the compiler they
    // become v_mov_b64 (constant -> destination) + v_fmac_f64)
    double p;
    if (LOG_TAB_MB == 7) {
        p = fma3(r, lc.c6, lc.c5);
    } else {
        p = fma3(r, lc.c7, lc.c6);
        p = fma3(r, p, lc.c5);
    }
    p = fma(r, p, -0.25);
    p = fma3(r, p, lc.c3);
    p = fma(r, p, -0.5);
    p = fma(r, p, 1.0);
    return fma(r, p, t);
}
__device__ __forceinline__ double tile_log1p_fast(double x, const LogConsts& lc) {
    const double xs = x * FDX_LOG_DOWN;
    return tile_log1p_core(xs, (float)xs + FDX_LOG_DOWN_F, lc);
}
// the same for y * scale with y already a float: (1 + x) * 2^-65 comes from one float fma.  scale_s = scale * 2^-65.
__device__ __forceinline__ double tile_log1p_scaled(float y, double scale_s, float scale_sf, const LogConsts& lc) {
    return tile_log1p_core((double)y * scale_s, fmaf(y, scale_sf, FDX_LOG_DOWN_F), lc);
}
__device__ __forceinline__ double tile_log1p_scaled(double y, double scale_s, float, const LogConsts& lc) {
    const double xs = y * scale_s;
    return tile_log1p_core(xs, (float)xs + FDX_LOG_DOWN_F, lc);
}
// float32 input: the reference itself evaluates log1p(Y / rowsum * 1e4) in float32 there (numpy dtype rules, core/deconv.py:
// 190-191), so a float32-class transform is all the path has to deliver - 7 vector instructions instead of ~20 f64 ones.
//   u = fl(1 + x);  log1p(x) = log(u) + log(1 + d / u),  d = x - (u - 1) exactly the rounding error of u (both
//   subtractions are exact);  log(u) = ln2 * v_log_f32(u) (1 ulp);  d / u to first order with 1 / u ~ the float whose bits
//   are 0x7F000000 - bits(u): exact at the powers of two - in particular at u = 1, where the term IS the result
//   (x < 2^-24) -, at most 12.5 % too large in between, on a term that is at most 2^-24 of u.
// Without the d term the error is 2^-24 ABSOLUTE, i.e. unbounded relative to log1p(x) ~ x for small x; with it the result
// is within ~3 ulp of float32 everywhere on [0, 32000).
__device__ __forceinline__ float tile_log1p_f32(float y, float s) {
#pragma clang fp contract(off)
    const float x = y * s;
    const float u = x + 1.0f;
    const float l = __builtin_amdgcn_logf(u);                      // v_log_f32: log2(u), u >= 1
    const float d = x - (u - 1.0f);
    const float g = __uint_as_float(0x7F000000u - __float_as_uint(u));   // ~ 1 / u
    return __builtin_fmaf(d, g, l * 0.693147180559945309f);
}

// Anything outside the fast range (negative, NaN, huge) takes the library function, as the reference would.  Kept out of
// line: inlined into every gather loop it costs registers on the path that matters.
static __device__ __attribute__((noinline)) double tile_log1p_slow(double x) { return log1p(x); }
__device__ __forceinline__ double tile_log1p(double x, const LogConsts& lc) {
    if (__builtin_expect(!(x >= 0.0) || !(x < 32000.0), 0)) return tile_log1p_slow(x);
    return tile_log1p_fast(x, lc);
}

// any argument, without the table (the general path of the float32 kernels): device_math.h's fdlibm-style log1p
static __device__ __attribute__((noinline)) double tile_log1p_any(double x) { return fast_log1p(x); }

// the same with the library function inlined (no call: a call site makes the caller spill its live registers around it)
__device__ __forceinline__ double tile_log1p_general(double x, const LogConsts& lc) {
    if (__builtin_expect(!(x >= 0.0) || !(x < 32000.0), 0)) return log1p(x);
    return tile_log1p_fast(x, lc);
}

// scale of one row for the log modes
template <int MODE> __device__ __forceinline__ double tile_row_scale(double sum) {
    if (MODE == FDX_PRE_LOG_CPM) return (1.0 / (sum + 1e-10)) * 1e4;          // y / (rowsum + 1e-10) * 1e4   (deconv.py:190)
    if (sum == 0.0) sum = 1.0;                                               // lib_size[lib_size == 0] = 1  (deconv.py:183-185)
    return 1e4 / sum;
}

// Workgroup barrier that orders LDS traffic only.  __syncthreads() also drains vmcnt, i.e. it would wait for the LDS-DMA
// pieces of the next block, which are meant to stay in flight across the reduction at the end of a tile.
__device__ __forceinline__ void lds_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

// -log of every table reciprocal in [2^-15, 1] (LOG_TAB_N doubles), one copy per device; NULL on failure
const double* log_table_dev(hipStream_t st);

}  // namespace fdx
