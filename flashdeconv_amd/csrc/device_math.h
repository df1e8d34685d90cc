// Device helpers shared by the dense and CSR sketch kernels.
#pragma once
#include <hip/hip_runtime.h>

namespace fdx {

__device__ __forceinline__ double wave_sum(double v) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
    return v;
}

// log1p for the log-CPM transform.  The library log1p costs ~200 issue slots per element, which makes the log-CPM
// sketch VALU-bound; this is the classic fdlibm decomposition specialised to x >= 0 (counts):
//   u = 1 + x = 2^k * m,  m in [sqrt(1/2), sqrt(2));   log1p(x) = k*ln2 + log(m) + c/u,   c = x - (u - 1) (rounding of 1+x)
//   log(m): f = m - 1, s = f/(2+f), log(m) = f - (f^2/2 - s*(f^2/2 + R(s^2))),  R = degree-7 minimax (fdlibm Lg1..Lg7)
// Error < 1 ulp (checked against numpy.log1p in tests/test_gpu_stages.py).  Negative / non-finite inputs take the
// library path (the reference yields NaN for x < -1 as well).
__device__ __forceinline__ double fast_log1p_core(double x) {        // 0 <= x <= 1e300 only
    const double u = 1.0 + x;
    const double c = (x >= 1.0) ? 1.0 - (u - x) : x - (u - 1.0);
    const double c_over_u = c * (double)__frcp_rn((float)u);          // |c| <= ulp(u)/2: 24-bit reciprocal is plenty
    long long bits = __double_as_longlong(u);
    int k = (int)((bits >> 52) & 0x7ff) - 1023;
    long long mant = bits & 0x000fffffffffffffLL;
    // m in [sqrt(1/2), sqrt(2)): mantissas above sqrt(2) move down one binade
    const int up = (mant >= 0x0006a09e667f3bcdLL) ? 1 : 0;
    k += up;
    const double m = __longlong_as_double(mant | ((long long)(1023 - up) << 52));
    const double f = m - 1.0;
    const double den = 2.0 + f;
    double r = __drcp_rn(den);                                          // den in [1.70, 2.42]
    double s = f * r;
    s = fma(r, fma(-s, den, f), s);                                     // one correction step: s = f/den to ~0.5 ulp
    const double z = s * s, w = z * z;
    const double t1 = w * fma(w, fma(w, 1.531383769920937332e-01, 2.222219843214978396e-01), 3.999999999940941908e-01);
    const double t2 = z * fma(w, fma(w, fma(w, 1.479819860511658591e-01, 1.818357216161805012e-01),
                                     2.857142874366239149e-01), 6.666666666666735130e-01);
    const double R = t2 + t1;
    const double hfsq = 0.5 * f * f;
    const double dk = (double)k;
    const double ln2_hi = 6.93147180369123816490e-01, ln2_lo = 1.90821492927058770002e-10;
    return dk * ln2_hi - ((hfsq - (s * (hfsq + R) + (dk * ln2_lo + c_over_u))) - f);
}

__device__ __forceinline__ double fast_log1p(double x) {
    if (!(x >= 0.0) || x > 1e300) return log1p(x);
    return fast_log1p_core(x);
}

// log1p(y * scale) for one row of count data.  Counts are small non-negative integers, and all entries of a row share
// `scale`, so the wave evaluates tab[c] = fast_log1p(c * scale) for c = 0..63 once per row (one value per lane, kept in
// LDS) and every entry that is such an integer takes a table read instead of ~45 f64 instructions; anything else - larger
// counts, normalised or non-integer input, NaN - is computed directly.  tab[c] is the same function of the same argument,
// so results are bit-identical with and without the table.
__device__ __forceinline__ void log1p_table_fill(double* tab, double scale, int lane) {
    tab[lane] = fast_log1p(__dmul_rn((double)lane, scale));      // (rounded product: never fused into the 1 + x of the log1p)
}

// use_tab is wave-uniform (row maximum below 64): rows of large counts or normalised values skip the table and its
// per-entry test altogether.
__device__ __forceinline__ double log1p_scaled(double y, double scale, const double* tab, bool use_tab) {
    if (use_tab) {
        const int c = (int)y;
        if ((double)c == y && (unsigned)c < 64u) return tab[c];
    }
    return fast_log1p(__dmul_rn(y, scale));
}

__device__ __forceinline__ double wave_max(double v) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v = fmax(v, __shfl_xor(v, off, 64));
    return v;
}

}  // namespace fdx
