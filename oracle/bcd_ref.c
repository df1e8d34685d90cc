/*
 * oracle/bcd_ref.c -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.
 *
 * CPU restatement (plain C, float64, no fast-math, no FMA contraction) of the
 * two pieces of the reference hot path that are too slow to check in pure
 * Python at useful sizes:
 *
 *   oracle_bcd_iteration   <- flashdeconv/core/solver.py:104-184 (_bcd_iteration_fused)
 *                             + :29-101 (update_spot_with_Xty) + :18-26 (soft_threshold)
 *   oracle_countsketch_draw<- numpy legacy RandomState (MT19937) as driven by
 *                             flashdeconv/core/sketching.py:58-59 via utils/random.py:64-65
 *
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load
 * this library; the product (flashdeconv_amd/) never does.  Parity status: pinned
 * against golden vectors captured from the reference (tests/golden/, see
 * tests/test_oracle.py).
 *
 * Build: see oracle/Makefile (gcc -O2 -fopenmp -ffp-contract=off).
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

/* ---- soft threshold: core/solver.py:18-26 -------------------------------- */
static inline double soft_thr(double x, double t)
{
    if (x > t) return x - t;
    if (x < -t) return x + t;
    return 0.0;
}

/*
 * One Jacobi-over-spots / Gauss-Seidel-over-types sweep.
 *   H        (K, N) row-major   : H[k*N + i]            (core/solver.py:150 reads the strided column)
 *   XtX      (K, K) row-major
 *   beta_in  (N, K) row-major, read only ; beta_out (N, K) written
 *   indptr   (N+1) int64, indices (nnz) int64          (core/solver.py:363-365)
 *   spot_diffs / spot_abs (N)                           (core/solver.py:173-184)
 */
void oracle_bcd_iteration(const double *H, const double *XtX, const double *beta_in, double *beta_out,
                          const int64_t *indices, const int64_t *indptr, int64_t N, int64_t K,
                          double lambda, double rho, double *spot_diffs, double *spot_abs)
{
#pragma omp parallel
    {
        double *r = (double *)malloc(sizeof(double) * (size_t)K);
        double *nb = (double *)malloc(sizeof(double) * (size_t)K);
#pragma omp for schedule(static)
        for (int64_t i = 0; i < N; ++i) {
            double *b = beta_out + i * K;
            const double *bi = beta_in + i * K;
            for (int64_t k = 0; k < K; ++k) b[k] = bi[k];                 /* :153-154 */
            const int64_t s = indptr[i], e = indptr[i + 1];
            const int64_t deg = e - s;                                     /* :157-159 */
            for (int64_t k = 0; k < K; ++k) nb[k] = 0.0;
            for (int64_t p = s; p < e; ++p) {                              /* :163-166 */
                const double *bj = beta_in + indices[p] * K;
                for (int64_t k = 0; k < K; ++k) nb[k] += bj[k];
            }
            for (int64_t k = 0; k < K; ++k) {                              /* :72  r = XtX @ beta_i */
                double acc = 0.0;
                for (int64_t j = 0; j < K; ++j) acc += XtX[k * K + j] * b[j];
                r[k] = acc;
            }
            for (int64_t k = 0; k < K; ++k) {                              /* :75-99 */
                const double old = b[k];
                const double gkk = XtX[k * K + k];
                double res = H[k * N + i] - r[k] + gkk * old;              /* :79 */
                if (deg > 0) res += lambda * nb[k];                        /* :82-83 */
                const double den = gkk + lambda * (double)deg;             /* :86 */
                double nw;
                if (den > 1e-10) {                                         /* :89-93 */
                    nw = soft_thr(res, rho) / den;
                    nw = nw > 0.0 ? nw : 0.0;
                } else {
                    nw = 0.0;
                }
                b[k] = nw;
                const double delta = nw - old;                             /* :96-99 */
                if (delta != 0.0)
                    for (int64_t kk = 0; kk < K; ++kk) r[kk] += delta * XtX[kk * K + k];
            }
            double dmax = 0.0, amax = 0.0;                                 /* :174-184 */
            for (int64_t k = 0; k < K; ++k) {
                const double d = fabs(b[k] - bi[k]);
                if (d > dmax) dmax = d;
                const double a = fabs(bi[k]);
                if (a > amax) amax = a;
            }
            spot_diffs[i] = dmax;
            spot_abs[i] = amax;
        }
        free(r);
        free(nb);
    }
}

/* ---- MT19937 (Matsumoto & Nishimura 1998), numpy legacy seeding ---------- */
typedef struct { uint32_t mt[624]; int idx; } mt_state;

static void mt_seed(mt_state *s, uint32_t seed)
{   /* init_genrand: what RandomState(int) does for a 32-bit integer seed */
    s->mt[0] = seed;
    for (int i = 1; i < 624; ++i)
        s->mt[i] = 1812433253u * (s->mt[i - 1] ^ (s->mt[i - 1] >> 30)) + (uint32_t)i;
    s->idx = 624;
}

static uint32_t mt_next(mt_state *s)
{
    if (s->idx >= 624) {
        uint32_t *mt = s->mt;
        for (int k = 0; k < 624; ++k) {
            uint32_t y = (mt[k] & 0x80000000u) | (mt[(k + 1) % 624] & 0x7fffffffu);
            uint32_t v = mt[(k + 397) % 624] ^ (y >> 1);
            if (y & 1u) v ^= 0x9908b0dfu;
            mt[k] = v;
        }
        s->idx = 0;
    }
    uint32_t y = s->mt[s->idx++];
    y ^= y >> 11;
    y ^= (y << 7) & 0x9d2c5680u;
    y ^= (y << 15) & 0xefc60000u;
    y ^= y >> 18;
    return y;
}

/*
 * bucket = rng.randint(0, d, size=G); sign = rng.choice([-1, 1], size=G)   (sketching.py:58-59)
 * Legacy randint on a range that fits 32 bits draws one 32-bit output per
 * attempt, masks it with the smallest all-ones mask >= d-1 and rejects values
 * > d-1; a range of a single value (d == 1) consumes no draws.  choice() over
 * a 2-element population is randint(0, 2): next output & 1, never rejected.
 * Returns the number of 32-bit outputs consumed.
 */
int64_t oracle_countsketch_draw(uint32_t seed, int64_t G, int64_t d, int64_t *bucket, int64_t *sign)
{
    mt_state st;
    mt_seed(&st, seed);
    int64_t draws = 0;
    const uint32_t top = (uint32_t)(d - 1);
    uint32_t mask = top;
    mask |= mask >> 1; mask |= mask >> 2; mask |= mask >> 4; mask |= mask >> 8; mask |= mask >> 16;
    for (int64_t g = 0; g < G; ++g) {
        if (top == 0) { bucket[g] = 0; continue; }
        uint32_t v;
        do { v = mt_next(&st) & mask; ++draws; } while (v > top);
        bucket[g] = (int64_t)v;
    }
    for (int64_t g = 0; g < G; ++g) {
        uint32_t v = mt_next(&st) & 1u; ++draws;
        sign[g] = v ? 1 : -1;
    }
    return draws;
}
