// Leverage scores of the genes from the SVD of the centred reference signatures.
//
// Replaces flashdeconv/utils/genes.py:238-290 (compute_leverage_scores):
//   Xc = X - mean over cell types (:264);  Xc^T = U diag(s) V^T  (:270, LAPACK gesdd in the reference);
//   w_j = s_j^2 / (s_j^2 + reg) (:281);  lev_g = sum_j U[g,j]^2 w_j (:285);  lev /= (sum lev + reg) (:288).
//
// The SVD is a one-sided (Hestenes) Jacobi on the tall G x K matrix A = Xc^T: pairs of columns are rotated until
// all are mutually orthogonal; then s_j = ||a_j|| and U[:,j] = a_j / s_j, so
//   lev_g = sum_j a_gj^2 / (s_j^2 + reg)
// needs no division by a (possibly zero) singular value: the direction that centring annihilates contributes
// exactly like LAPACK's ~1e-16 singular value does, i.e. nothing.  Working on the tall matrix (not on the K x K
// Gram matrix) keeps small singular values accurate to machine precision.
//
// Mapping: one workgroup of 8 wavefronts (2 per SIMD, so a wave can hold both of its columns in 256 VGPRs).  Columns of A are ROWS of Xc (contiguous, G doubles).  A round-robin
// tournament gives K/2 disjoint column pairs per round; each wave owns one pair, streams both columns from L2 with
// coalesced loads, reduces the three inner products with wave shuffles and applies the rotation.  Rounds are
// separated by workgroup barriers.  The reference matrix is tiny (K x G doubles), so this kernel is latency-, not
// bandwidth-bound; it runs once per fit.
#include "fdx_env.h"
#include <algorithm>
#include <cstdlib>

#include "fdx_internal.h"
#include "fdx_kernels.h"

namespace fdx {

__device__ __forceinline__ double wsum(double v) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
    return v;
}

template <int NCH>
__global__ __launch_bounds__(512) void leverage_jacobi_kernel(const double* __restrict__ X, int K, int G, double reg,
                                                               double* __restrict__ A /* (K, G) work */,
                                                               double* __restrict__ sig2 /* (K) */,
                                                               double* __restrict__ lev /* (G) */,
                                                               int* __restrict__ sweeps_out) {
    extern __shared__ __attribute__((aligned(16))) double s_gram[];   // 2 x 64 x 65 doubles (Gram matrix, rotations)
    __shared__ int s_rot;
    __shared__ double s_red[16];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    // centre over cell types (genes.py:264)
    for (int g = tid; g < G; g += 512) {
        double m = 0.0;
        for (int k = 0; k < K; ++k) m += X[(size_t)k * G + g];
        m /= (double)K;
        for (int k = 0; k < K; ++k) A[(size_t)k * G + g] = X[(size_t)k * G + g] - m;
    }
    long long t_dbg0 = wall_clock64();
    __syncthreads();
    // ---- preconditioning (K <= 64): rotate the columns by the eigenvectors of the K x K Gram matrix, A <- A V.
    // V is a product of plane rotations (orthogonal to machine precision), so A V has exactly A's singular values and
    // left vectors, but nearly orthogonal columns: the one-sided sweeps below then converge in 2-3 instead of ~13.
    // The Gram matrix only steers; every inner product that decides the result is recomputed from the columns.
    if (K <= 64 && K >= 2) {
        double (*C)[65] = reinterpret_cast<double (*)[65]>(s_gram);
        double (*V)[65] = reinterpret_cast<double (*)[65]>(s_gram + 64 * 65);
        for (int p = wave; p < K; p += 8)                 // C = A A^T (upper triangle by row owner, mirrored)
            for (int q = p; q < K; ++q) {
                double acc = 0.0;
                for (int g0 = 0; g0 < G; g0 += 512) {     // 16 independent loads in flight per step
                    double x[8], y[8];
#pragma unroll
                    for (int u = 0; u < 8; ++u) {
                        const int g = g0 + u * 64 + lane;
                        x[u] = (g < G) ? A[(size_t)p * G + g] : 0.0;
                        y[u] = (g < G) ? A[(size_t)q * G + g] : 0.0;
                    }
#pragma unroll
                    for (int u = 0; u < 8; ++u) acc = fma(x[u], y[u], acc);
                }
                acc = wsum(acc);
                if (lane == 0) { C[p][q] = acc; C[q][p] = acc; }
            }
        for (int e = tid; e < K * K; e += 512) V[e / K][e % K] = (e / K == e % K) ? 1.0 : 0.0;
        __syncthreads();
        if (tid == 0 && sweeps_out) sweeps_out[1] = (int)(wall_clock64() - t_dbg0);
        // parallel-order two-sided Jacobi on C: the K/2 disjoint pairs of a round-robin round are rotated together
        // (rotation parameters -> column update of C and V -> row update of C, workgroup barriers in between)
        {
            double* cs = s_gram + 2 * 64 * 65;            // (c, s) per pair of the current round
            const int Kq = (K + 1) & ~1, npair = Kq / 2;
            for (int sw = 0; sw < 12; ++sw) {
                if (tid == 0) s_rot = 0;
                __syncthreads();
                for (int r = 0; r < Kq - 1; ++r) {
                    if (tid < npair) {
                        int p, q;
                        if (tid == 0) { p = Kq - 1; q = r; } else { p = (r + tid) % (Kq - 1); q = (r - tid + (Kq - 1)) % (Kq - 1); }
                        if (p > q) { const int t = p; p = q; q = t; }
                        double c = 1.0, sn = 0.0;
                        if (q < K) {
                            const double cpq = C[p][q], cpp = C[p][p], cqq = C[q][q];
                            if (fabs(cpq) > 1e-15 * sqrt(fabs(cpp * cqq)) && cpq != 0.0) {
                                const double theta = (cqq - cpp) / (2.0 * cpq);
                                const double t = (theta >= 0.0 ? 1.0 : -1.0) / (fabs(theta) + sqrt(1.0 + theta * theta));
                                c = 1.0 / sqrt(1.0 + t * t);
                                sn = c * t;
                                s_rot = 1;
                            }
                        }
                        cs[2 * tid] = c; cs[2 * tid + 1] = sn;
                    }
                    __syncthreads();
                    for (int e = tid; e < npair * K; e += 512) {          // columns p,q of C and V
                        const int m = e / K, k = e - m * K;
                        int p, q;
                        if (m == 0) { p = Kq - 1; q = r; } else { p = (r + m) % (Kq - 1); q = (r - m + (Kq - 1)) % (Kq - 1); }
                        if (p > q) { const int t = p; p = q; q = t; }
                        if (q >= K) continue;
                        const double c = cs[2 * m], sn = cs[2 * m + 1];
                        const double ckp = C[k][p], ckq = C[k][q];
                        C[k][p] = c * ckp - sn * ckq;
                        C[k][q] = sn * ckp + c * ckq;
                        const double vkp = V[k][p], vkq = V[k][q];
                        V[k][p] = c * vkp - sn * vkq;
                        V[k][q] = sn * vkp + c * vkq;
                    }
                    __syncthreads();
                    for (int e = tid; e < npair * K; e += 512) {          // rows p,q of C
                        const int m = e / K, k = e - m * K;
                        int p, q;
                        if (m == 0) { p = Kq - 1; q = r; } else { p = (r + m) % (Kq - 1); q = (r - m + (Kq - 1)) % (Kq - 1); }
                        if (p > q) { const int t = p; p = q; q = t; }
                        if (q >= K) continue;
                        const double c = cs[2 * m], sn = cs[2 * m + 1];
                        const double cpk = C[p][k], cqk = C[q][k];
                        C[p][k] = c * cpk - sn * cqk;
                        C[q][k] = sn * cpk + c * cqk;
                    }
                    __syncthreads();
                }
                const int any = s_rot;
                __syncthreads();
                if (!any) break;
            }
        }
        if (tid == 0 && sweeps_out) sweeps_out[2] = (int)(wall_clock64() - t_dbg0);
        // A[:, g] <- V^T A[:, g]: lane = gene, the gene's K values in registers, V broadcast from LDS
        for (int g0 = wave * 64; g0 < G; g0 += 512) {
            const int g = g0 + lane;
            double av[64];
#pragma unroll
            for (int i = 0; i < 64; ++i) av[i] = (i < K && g < G) ? A[(size_t)i * G + g] : 0.0;
            for (int j = 0; j < K; ++j) {
                double acc = 0.0;
#pragma unroll
                for (int i = 0; i < 64; ++i)
                    if (i < K) acc = fma(V[i][j], av[i], acc);
                if (g < G) A[(size_t)j * G + g] = acc;
            }
        }
        __syncthreads();
    }
    if (tid == 0 && sweeps_out) sweeps_out[3] = (int)(wall_clock64() - t_dbg0);
    const int Kp = (K + 1) & ~1;          // even number of players (one dummy if K is odd)
    const int n_pairs = Kp / 2, n_rounds = Kp - 1;
    int sweep = 0;
    for (; sweep < 60; ++sweep) {
        if (tid == 0) s_rot = 0;
        __syncthreads();
        for (int r = 0; r < n_rounds; ++r) {
            for (int m = wave; m < n_pairs; m += 8) {
                int p, q;
                if (m == 0) { p = Kp - 1; q = r; }
                else { p = (r + m) % (Kp - 1); q = (r - m + (Kp - 1)) % (Kp - 1); }
                if (p >= K || q >= K) continue;     // pair with the dummy player
                if (p > q) { const int t = p; p = q; q = t; }
                double* ap = A + (size_t)p * G;
                double* aq = A + (size_t)q * G;
                double alpha = 0.0, beta = 0.0, gamma = 0.0;
                if (NCH > 0) {
                    // register-resident round (G <= NCH*512): both columns are loaded ONCE (all loads in flight together, the
                    // matrix lives in L2), reduced, rotated in registers and stored - one L2 round trip per round
                    constexpr int NE = NCH > 0 ? NCH * 8 : 1;
                    double x[NE], y[NE];
#pragma unroll
                    for (int u = 0; u < NE; ++u) {
                        const int g = u * 64 + lane;
                        x[u] = (g < G) ? ap[g] : 0.0;
                        y[u] = (g < G) ? aq[g] : 0.0;
                    }
#pragma unroll
                    for (int u = 0; u < NE; ++u) {
                        alpha = fma(x[u], x[u], alpha);
                        beta = fma(y[u], y[u], beta);
                        gamma = fma(x[u], y[u], gamma);
                    }
                    alpha = wsum(alpha); beta = wsum(beta); gamma = wsum(gamma);
                    if (fabs(gamma) > 1e-15 * sqrt(alpha * beta) && gamma != 0.0) {
                        const double zeta = (beta - alpha) / (2.0 * gamma);
                        const double t = (zeta >= 0.0 ? 1.0 : -1.0) / (fabs(zeta) + sqrt(1.0 + zeta * zeta));
                        const double c = 1.0 / sqrt(1.0 + t * t), s = c * t;
#pragma unroll
                        for (int u = 0; u < NE; ++u) {
                            const int g = u * 64 + lane;
                            if (g < G) {
                                ap[g] = c * x[u] - s * y[u];
                                aq[g] = s * x[u] + c * y[u];
                            }
                        }
                        if (lane == 0) s_rot = 1;
                    }
                    continue;
                }
                // streaming round (any G): 8 independent element pairs per lane per step keep 16 loads in flight
                for (int g0 = 0; g0 < G; g0 += 512) {
                    double x[8], y[8];
#pragma unroll
                    for (int u = 0; u < 8; ++u) {
                        const int g = g0 + u * 64 + lane;
                        x[u] = (g < G) ? ap[g] : 0.0;
                        y[u] = (g < G) ? aq[g] : 0.0;
                    }
#pragma unroll
                    for (int u = 0; u < 8; ++u) {
                        alpha = fma(x[u], x[u], alpha);
                        beta = fma(y[u], y[u], beta);
                        gamma = fma(x[u], y[u], gamma);
                    }
                }
                alpha = wsum(alpha); beta = wsum(beta); gamma = wsum(gamma);
                if (fabs(gamma) > 1e-15 * sqrt(alpha * beta) && gamma != 0.0) {
                    const double zeta = (beta - alpha) / (2.0 * gamma);
                    const double t = (zeta >= 0.0 ? 1.0 : -1.0) / (fabs(zeta) + sqrt(1.0 + zeta * zeta));
                    const double c = 1.0 / sqrt(1.0 + t * t), s = c * t;
                    for (int g0 = 0; g0 < G; g0 += 512) {
                        double x[8], y[8];
#pragma unroll
                        for (int u = 0; u < 8; ++u) {
                            const int g = g0 + u * 64 + lane;
                            x[u] = (g < G) ? ap[g] : 0.0;
                            y[u] = (g < G) ? aq[g] : 0.0;
                        }
#pragma unroll
                        for (int u = 0; u < 8; ++u) {
                            const int g = g0 + u * 64 + lane;
                            if (g < G) {
                                ap[g] = c * x[u] - s * y[u];
                                aq[g] = s * x[u] + c * y[u];
                            }
                        }
                    }
                    if (lane == 0) s_rot = 1;   // benign race: every writer stores 1
                }
            }
            __syncthreads();
        }
        const int any = s_rot;
        __syncthreads();
        if (!any) break;
    }
    if (tid == 0 && sweeps_out) sweeps_out[4] = (int)(wall_clock64() - t_dbg0);
    // squared singular values
    for (int j = wave; j < K; j += 8) {
        double a2 = 0.0;
        for (int g = lane; g < G; g += 64) { const double x = A[(size_t)j * G + g]; a2 = fma(x, x, a2); }
        a2 = wsum(a2);
        if (lane == 0) sig2[j] = a2;
    }
    __syncthreads();
    // lev_g = sum_j a_gj^2 / (s_j^2 + reg)   (genes.py:281-285), then normalise (genes.py:288)
    double part = 0.0;
    for (int g = tid; g < G; g += 512) {
        double l = 0.0;
        for (int j = 0; j < K; ++j) { const double x = A[(size_t)j * G + g]; l += x * x / (sig2[j] + reg); }
        lev[g] = l;
        part += l;
    }
    part = wsum(part);
    if (lane == 0) s_red[wave] = part;
    __syncthreads();
    double total = 0.0;
    for (int w = 0; w < 8; ++w) total += s_red[w];
    for (int g = tid; g < G; g += 512) lev[g] = lev[g] / (total + reg);
    if (tid == 0 && sweeps_out) { *sweeps_out = sweep; sweeps_out[5] = (int)(wall_clock64() - t_dbg0); }
}

// ================================================================================================================
// Multi-CU path (K <= 64): the same orthogonalisation as a short sequence of streaming kernels.
//
// One-sided Jacobi keeps the whole G x K matrix inside one workgroup: 1.6 ms at 30 x 2000, 9 ms at 50 x 5000, on one of
// 256 CUs.  Here the tall matrix is only ever touched by grid-wide streaming kernels, and the workgroup-sized part is the
// K x K problem:
//   pass:  C = A A^T            (lev_gram_kernel: every block reduces a stripe of genes, partials summed in block order)
//          C = V L V^T          (lev_eigen_kernel: one workgroup, parallel-order two-sided Jacobi in LDS; sets `done`
//                                and sig2 = diag(C) when no off-diagonal entry is above the rotation threshold)
//          A <- V^T A           (lev_rotate_kernel: lane = gene, V broadcast from LDS)
// V is a product of plane rotations, so every pass preserves A's singular values and left vectors to machine precision
// and only improves the orthogonality of its columns: after pass 1 they are orthogonal to ~eps*cond^2, after pass 2 to
// ~eps (the Gram matrix of nearly orthogonal columns is nearly diagonal, where Jacobi is accurate in the RELATIVE sense),
// pass 3 finds nothing to rotate and records sig2_j = ||a_j||^2 from the columns themselves.  Passes after `done` return
// at once, so the sequence is launched unconditionally (no host round trip).  Finally
//   lev_g = sum_j a_gj^2 / (sig2_j + reg)   (genes.py:281-285),   lev /= (sum lev + reg)   (genes.py:288).
constexpr int LEV_STRIPE = 64;     // genes per block of the Gram kernel (64 x 65 doubles of LDS)
constexpr int LEV_PASSES = 4;

__global__ __launch_bounds__(256) void lev_centre_kernel(const double* __restrict__ X, int K, int G, double* __restrict__ A) {
    const int g = blockIdx.x * 256 + threadIdx.x;
    if (g >= G) return;
    double m = 0.0;
    for (int k = 0; k < K; ++k) m += X[(size_t)k * G + g];
    m /= (double)K;                                                       // genes.py:264
    for (int k = 0; k < K; ++k) A[(size_t)k * G + g] = X[(size_t)k * G + g] - m;
}

// part[b][p][q] = sum over the block's genes of A[p][g] A[q][g]
__global__ __launch_bounds__(256) void lev_gram_kernel(const double* __restrict__ A, int K, int G,
                                                       double* __restrict__ part, const int* __restrict__ done) {
    if (*done) return;
    __shared__ double tile[64][LEV_STRIPE + 1];
    const int g0 = blockIdx.x * LEV_STRIPE;
    for (int e = threadIdx.x; e < K * LEV_STRIPE; e += 256) {
        const int k = e / LEV_STRIPE, j = e - k * LEV_STRIPE;
        tile[k][j] = (g0 + j < G) ? A[(size_t)k * G + g0 + j] : 0.0;
    }
    __syncthreads();
    double* out = part + (size_t)blockIdx.x * K * K;
    for (int e = threadIdx.x; e < K * K; e += 256) {
        const int p = e / K, q = e - p * K;
        if (q < p) continue;                                              // upper triangle, mirrored
        double a0 = 0.0, a1 = 0.0;
#pragma unroll 8
        for (int j = 0; j < LEV_STRIPE; j += 2) {
            a0 = fma(tile[p][j], tile[q][j], a0);
            a1 = fma(tile[p][j + 1], tile[q][j + 1], a1);
        }
        const double v = a0 + a1;
        out[p * K + q] = v;
        out[q * K + p] = v;
    }
}

// C = sum_b part[b] (block order); two-sided Jacobi; V -> global.  `done` and sig2 when C is already diagonal.
__global__ __launch_bounds__(512) void lev_eigen_kernel(const double* __restrict__ part, int n_blocks, int K,
                                                        double* __restrict__ Vout, double* __restrict__ sig2,
                                                        int* __restrict__ done, int* __restrict__ passes) {
    if (*done) return;
    extern __shared__ __attribute__((aligned(16))) double s_gram[];
    __shared__ int s_rot, s_first;
    __shared__ int s_partner[64];
    __shared__ double s_own[64], s_oth[64];
    // C and V twice: a round reads one copy and writes the other, so the two-sided update C <- J^T C J and V <- V J of all
    // disjoint pairs is ONE phase (element (i, j) from the four entries C[{i, r_i}][{j, r_j}]) instead of a column phase and a
    // row phase with a barrier between them - the kernel is nothing but barrier latency (K = 50: 49 rounds x ~7 sweeps)
    typedef double (*Mat)[65];
    auto Cbuf = [&](int b) { return reinterpret_cast<Mat>(s_gram + b * 64 * 65); };
    auto Vbuf = [&](int b) { return reinterpret_cast<Mat>(s_gram + (2 + b) * 64 * 65); };
    double* cs = s_gram + 4 * 64 * 65;
    const int tid = threadIdx.x;
    for (int e = tid; e < K * K; e += 512) {
        double acc = 0.0;
        for (int b = 0; b < n_blocks; ++b) acc += part[(size_t)b * K * K + e];
        Cbuf(0)[e / K][e % K] = acc;
        Vbuf(0)[e / K][e % K] = (e / K == e % K) ? 1.0 : 0.0;
    }
    if (tid == 0) s_first = 0;
    __syncthreads();
    if (tid < K) cs[128 + tid] = Cbuf(0)[tid][tid];                         // ||a_j||^2 of the incoming columns
    const int Kq = (K + 1) & ~1, npair = Kq / 2, M = Kq - 1;
    int cur = 0;
    const int j = tid & 63, w8 = tid >> 6;
    for (int sw = 0; sw < 12; ++sw) {
        if (tid == 0) s_rot = 0;
        __syncthreads();
        for (int r = 0; r < M; ++r) {
            Mat C = Cbuf(cur), V = Vbuf(cur), Cn = Cbuf(cur ^ 1), Vn = Vbuf(cur ^ 1);
            if (tid < npair) {
                // round-robin pairing without divisions: both sums stay below 2 M
                int p, q;
                if (tid == 0) { p = M; q = r; }
                else {
                    p = r + tid; if (p >= M) p -= M;
                    q = r - tid + M; if (q >= M) q -= M;
                }
                if (p > q) { const int t = p; p = q; q = t; }
                double c = 1.0, sn = 0.0;
                if (q < K) {
                    const double cpq = C[p][q], cpp = C[p][p], cqq = C[q][q];
                    // Rotate when |c_pq| > 1e-14 sqrt(c_pp c_qq) (compared as squares) - the criterion that gives the small
                    // singular values their relative accuracy - AND |c_pq| > 1e-15 max(c_pp, c_qq): the updates of this
                    // kernel round at eps x the LARGER diagonal entry, so below that an off-diagonal entry is noise that no
                    // rotation of this pass can remove (a pair with c_qq < 1e-4 c_pp - always present: centring makes one
                    // direction null - never met the first test alone, and every pass ran all 12 sweeps).  What is left is
                    // taken up by the next pass, which forms C afresh from the rotated columns.
                    // First sweep of a pass: C is fresh from the columns, accurate to a few eps - floor 1e-15.  Later sweeps
                    // carry the rounding of ~K updates per entry - floor 1e-13.
                    const double big = fmax(fabs(cpp), fabs(cqq));
                    if (cpq * cpq > 1e-28 * fabs(cpp * cqq) && fabs(cpq) > (sw == 0 ? 1e-15 : 1e-13) * big) {
                        const double theta = (cqq - cpp) / (2.0 * cpq);
                        const double t = (theta >= 0.0 ? 1.0 : -1.0) / (fabs(theta) + sqrt(1.0 + theta * theta));
                        c = 1.0 / sqrt(1.0 + t * t);
                        sn = c * t;
                        s_rot = 1;
                        s_first = 1;
                    }
                    // x'_p = c x_p - sn x_q ,  x'_q = sn x_p + c x_q : own coefficient, partner, partner's coefficient
                    s_partner[p] = q; s_own[p] = c; s_oth[p] = -sn;
                    s_partner[q] = p; s_own[q] = c; s_oth[q] = sn;
                } else if (p < K) {                                       // odd K: p sits this round out
                    s_partner[p] = p; s_own[p] = 1.0; s_oth[p] = 0.0;
                }
            }
            __syncthreads();
            // lane = column j (its pair data read once), wave w takes rows w, w + 8, ... (row data is wave-uniform); the rows
            // of a wave are independent: their reads are issued together (up to eight rows: K <= 64)
            if (j < K) {
                const int rj = s_partner[j];
                const double cj = s_own[j], tj = s_oth[j];
                constexpr int RW = 8;
                int ri[RW];
                double ci[RW], ti[RW], a0[RW], a1[RW], a2[RW], a3[RW], v0[RW], v1[RW];
#pragma unroll
                for (int u = 0; u < RW; ++u) {
                    const int i = w8 + 8 * u;
                    if (i < K) { ri[u] = s_partner[i]; ci[u] = s_own[i]; ti[u] = s_oth[i]; }
                }
#pragma unroll
                for (int u = 0; u < RW; ++u) {
                    const int i = w8 + 8 * u;
                    if (i < K) {
                        a0[u] = C[i][j]; a1[u] = C[i][rj]; a2[u] = C[ri[u]][j]; a3[u] = C[ri[u]][rj];
                        v0[u] = V[i][j]; v1[u] = V[i][rj];
                    }
                }
#pragma unroll
                for (int u = 0; u < RW; ++u) {
                    const int i = w8 + 8 * u;
                    if (i < K) {
                        const double t_i = cj * a0[u] + tj * a1[u];           // column step, rows i and r_i
                        const double t_r = cj * a2[u] + tj * a3[u];
                        Cn[i][j] = ci[u] * t_i + ti[u] * t_r;                 // row step
                        Vn[i][j] = cj * v0[u] + tj * v1[u];
                    }
                }
            }
            __syncthreads();
            cur ^= 1;
        }
        const int any = s_rot;
        __syncthreads();
        if (!any) break;
    }
    Mat V = Vbuf(cur);
    if (!s_first) {                      // nothing to rotate: the columns are orthogonal, their norms are the answer
        if (tid < K) sig2[tid] = cs[128 + tid];
        if (tid == 0) *done = 1;
    } else {
        for (int e = tid; e < K * K; e += 512) Vout[e] = V[e / K][e % K];
        if (tid == 0) *passes += 1;
    }
}

// A[:, g] <- V^T A[:, g]
__global__ __launch_bounds__(256) void lev_rotate_kernel(double* __restrict__ A, int K, int G, const double* __restrict__ Vin,
                                                         const int* __restrict__ done) {
    if (*done) return;
    __shared__ double V[64][65];
    for (int e = threadIdx.x; e < K * K; e += 256) V[e / K][e % K] = Vin[e];
    __syncthreads();
    const int g = blockIdx.x * 256 + threadIdx.x;
    double av[64];
#pragma unroll
    for (int i = 0; i < 64; ++i) av[i] = (i < K && g < G) ? A[(size_t)i * G + g] : 0.0;
    for (int j = 0; j < K; ++j) {
        double acc = 0.0;
#pragma unroll
        for (int i = 0; i < 64; ++i)
            if (i < K) acc = fma(V[i][j], av[i], acc);
        if (g < G) A[(size_t)j * G + g] = acc;
    }
}

// not converged within LEV_PASSES (never seen): take the column norms of the last pass's Gram diagonal anyway
__global__ __launch_bounds__(64) void lev_diag_kernel(const double* __restrict__ part, int n_blocks, int K,
                                                      double* __restrict__ sig2, const int* __restrict__ done) {
    if (*done) return;
    const int j = threadIdx.x;
    if (j >= K) return;
    double acc = 0.0;
    for (int b = 0; b < n_blocks; ++b) acc += part[(size_t)b * K * K + j * K + j];
    sig2[j] = acc;
}

__global__ __launch_bounds__(256) void lev_scores_kernel(const double* __restrict__ A, int K, int G, const double* __restrict__ sig2,
                                                         double reg, double* __restrict__ lev, double* __restrict__ bsum) {
    __shared__ double sh[256];
    const int g = blockIdx.x * 256 + threadIdx.x;
    double l = 0.0;
    if (g < G) {
        for (int j = 0; j < K; ++j) { const double x = A[(size_t)j * G + g]; l += x * x / (sig2[j] + reg); }
        lev[g] = l;
    }
    sh[threadIdx.x] = l;
    __syncthreads();
    for (int s = 128; s > 0; s >>= 1) {
        if ((int)threadIdx.x < s) sh[threadIdx.x] += sh[threadIdx.x + s];
        __syncthreads();
    }
    if (threadIdx.x == 0) bsum[blockIdx.x] = sh[0];
}

__global__ __launch_bounds__(256) void lev_normalise_kernel(double* __restrict__ lev, int G, const double* __restrict__ bsum,
                                                            int n_blocks, double reg) {
    double total = 0.0;
    for (int b = 0; b < n_blocks; ++b) total += bsum[b];                  // same order in every block
    const int g = blockIdx.x * 256 + threadIdx.x;
    if (g < G) lev[g] = lev[g] / (total + reg);
}

// ---------------------------------------------------------------------------------------------------------------------
// Fast route for well-conditioned signature matrices: no SVD at all.
//
// With Xc^T = U S V^T, genes.py:281-285 is  lev_g = sum_j (xc_g . v_j)^2 / (s_j^2 + reg)  =  xc_g^T (Xc Xc^T + reg I)^-1 xc_g
// over the directions with s_j > 0.  Centring makes 1/sqrt(K) an exact null vector of Xc; a Householder reflection that maps it
// to e_K turns every centred gene column into z_g (K - 1 entries, the K-th is the rounding of a zero sum and is dropped: the
// reference weights that direction with s^2 / (s^2 + reg) ~ 1e-20).  Then
//   lev_g = || row g of Q ||^2,   A' = [Z^T ; sqrt(reg) I] = Q R   (G + m rows, m = K - 1 columns),
// the leverage of the rows of a ridge-augmented tall matrix.  Q comes from CholeskyQR2: R1 = chol(A'^T A'), Q1 = A' R1^-1,
// R2 = chol(Q1^T Q1), Q = Q1 R2^-1 - orthogonal to machine precision while cond(A')^2 eps << 1 (Yamamoto et al. 2015), i.e. the
// small singular values are NOT squared away as they would be in a plain normal-equations solve.  The pivots of both
// factorisations are watched: a pivot below 1e-9 of the largest diagonal entry (cond(A') above ~3e4: rank-deficient or badly
// scaled signatures) sets status 2 and fdx_leverage_end runs the Jacobi SVD passes above instead.
//
// Three launches of (G / 64 + 1) workgroups: a stripe of 64 genes per workgroup plus one workgroup for the m ridge rows; the
// last workgroup to finish a stage adds the per-stripe Gram matrices in stripe order (deterministic) and factorises in LDS.
constexpr double LEV_PIVOT_TOL = 1e-9;

// MMAX: 64 (up to 64 cell types) or 128 (65 - 128).  The symmetric matrix / its factor is kept as a packed lower triangle: at 128 the
// full square would be 132 KB beside the 66 KB stripe.
template <int MMAX>
struct LevQrShared {
    double tile[MMAX][LEV_STRIPE + 1];
    double Cp[MMAX * (MMAX + 1) / 2];
    double dpiv[MMAX];
    double thr;
    int bad, last;
    __device__ __forceinline__ double& C(int i, int j) { return Cp[i * (i + 1) / 2 + j]; }   // j <= i
};

// lower Cholesky factor of the symmetric C (lower triangle used) in place; 256 threads; *bad when a pivot is too small
template <int MMAX>
__device__ void lev_chol_lds(LevQrShared<MMAX>& sh, int m) {
    const int tid = threadIdx.x;
    if (tid == 0) {
        double mx = 0.0;
        for (int i = 0; i < m; ++i) mx = fmax(mx, sh.C(i, i));
        sh.thr = mx * LEV_PIVOT_TOL;
        sh.bad = !(mx > 0.0) || !(mx < 1e300);
    }
    __syncthreads();
    for (int j = 0; j < m; ++j) {
        double d = sh.C(j, j);
        if (!(d > sh.thr)) {                                              // also NaN
            if (tid == 0) sh.bad = 1;
            d = sh.thr > 0.0 ? sh.thr : 1.0;
        }
        if (tid == 0) sh.dpiv[j] = d;
        const double inv = 1.0 / d;
        const int r = m - 1 - j;
        for (int e = tid; e < r * r; e += 256) {
            const int i = j + 1 + e / r, k = j + 1 + e % r;
            if (k <= i) sh.C(i, k) -= sh.C(i, j) * sh.C(k, j) * inv;     // column j itself is not written in step j
        }
        __syncthreads();
    }
    for (int e = tid; e < m * m; e += 256) {
        const int i = e / m, j = e - i * m;
        if (j > i) continue;
        const double sd = sqrt(sh.dpiv[j]);
        sh.C(i, j) = (i == j) ? sd : sh.C(i, j) / sd;
    }
    __syncthreads();
}

// tile[:, j] <- L^-1 tile[:, j] for the 64 columns of the stripe (L in sh.C, lane = column)
template <int MMAX>
__device__ void lev_forward_lds(LevQrShared<MMAX>& sh, int m) {
    const int j = threadIdx.x;
    if (j < LEV_STRIPE) {
        for (int i = 0; i < m; ++i) {
            double s0 = sh.tile[i][j], s1 = 0.0;
            int k = 0;
            for (; k + 1 < i; k += 2) {
                s0 = fma(-sh.C(i, k), sh.tile[k][j], s0);
                s1 = fma(-sh.C(i, k + 1), sh.tile[k + 1][j], s1);
            }
            if (k < i) s0 = fma(-sh.C(i, k), sh.tile[k][j], s0);
            sh.tile[i][j] = (s0 + s1) / sh.C(i, i);
        }
    }
    __syncthreads();
}

// STAGE 0: X -> Z (work), Gram of the stripes, L1.   STAGE 1: Z -> Q1 = L1^-1 Z (work, in place), Gram, L2.
// STAGE 2: lev_g = |L2^-1 q1_g|^2 and the stripe sums.
template <int STAGE, int MMAX>
__global__ __launch_bounds__(256) void lev_qr_kernel(const double* __restrict__ X, int K, int G, double reg, double* work,
                                                     double* part, const double* Lin, double* Lout, double* __restrict__ lev,
                                                     double* __restrict__ bsum, int* counter, int* status) {
    __shared__ LevQrShared<MMAX> sh;
    const int tid = threadIdx.x, m = K - 1;
    const int nb = (G + LEV_STRIPE - 1) / LEV_STRIPE;
    const int nr = (m + LEV_STRIPE - 1) / LEV_STRIPE;                     // workgroups for the ridge rows sqrt(reg) I (64 rows each)
    const int b = blockIdx.x;                                             // b >= nb: ridge rows
    const int g0 = b * LEV_STRIPE;
    if (STAGE == 0) {
        if (b < nb) {
            const int g = g0 + tid;
            if (tid < LEV_STRIPE) {
                if (g < G) {
                    double mean = 0.0;
                    for (int k = 0; k < K; ++k) mean += X[(size_t)k * G + g];
                    mean /= (double)K;                                    // genes.py:264
                    double s = 0.0;
                    for (int k = 0; k < K; ++k) s += X[(size_t)k * G + g] - mean;
                    const double rk = sqrt((double)K);
                    const double xl = X[(size_t)m * G + g] - mean;
                    const double shift = (s / rk + xl) / (rk + 1.0);      // Householder v = 1/sqrt(K) + e_K applied to xc
                    for (int k = 0; k < m; ++k) {
                        const double z = (X[(size_t)k * G + g] - mean) - shift;
                        sh.tile[k][tid] = z;
                        work[(size_t)k * G + g] = z;
                    }
                } else {
                    for (int k = 0; k < m; ++k) sh.tile[k][tid] = 0.0;
                }
            }
        } else {
            for (int e = tid; e < m * LEV_STRIPE; e += 256) {
                const int k = e / LEV_STRIPE, j = e - k * LEV_STRIPE;
                sh.tile[k][j] = (k == (b - nb) * LEV_STRIPE + j) ? sqrt(reg) : 0.0;   // ridge rows 64 (b - nb) ...
            }
        }
        __syncthreads();
    } else {
        for (int e = tid; e < m * m; e += 256) {
            const int p = e / m, q = e - p * m;
            if (q <= p) sh.C(p, q) = Lin[e];
        }
        if (b < nb) {
            for (int e = tid; e < m * LEV_STRIPE; e += 256) {
                const int k = e / LEV_STRIPE, j = e - k * LEV_STRIPE;
                sh.tile[k][j] = (g0 + j < G) ? work[(size_t)k * G + g0 + j] : 0.0;
            }
        } else {
            for (int e = tid; e < m * LEV_STRIPE; e += 256) {
                const int k = e / LEV_STRIPE, j = e - k * LEV_STRIPE;
                sh.tile[k][j] = (k == (b - nb) * LEV_STRIPE + j) ? sqrt(reg) : 0.0;   // ridge rows 64 (b - nb) ...
            }
        }
        __syncthreads();
        lev_forward_lds(sh, m);
    }
    if (STAGE == 2) {
        double l = 0.0;
        if (tid < LEV_STRIPE && g0 + tid < G) {
            for (int k = 0; k < m; ++k) l = fma(sh.tile[k][tid], sh.tile[k][tid], l);
            lev[g0 + tid] = l;
        }
        if (tid < 64) {                                                   // LEV_STRIPE == 64: one wavefront holds the stripe
            l = wsum(l);
            if (tid == 0) bsum[b] = l;
        }
        return;
    }
    if (STAGE == 1 && b < nb) {
        for (int e = tid; e < m * LEV_STRIPE; e += 256) {
            const int k = e / LEV_STRIPE, j = e - k * LEV_STRIPE;
            if (g0 + j < G) work[(size_t)k * G + g0 + j] = sh.tile[k][j];
        }
    }
    double* out = part + (size_t)b * m * m;
    for (int e = tid; e < m * m; e += 256) {
        const int p = e / m, q = e - p * m;
        if (q > p) continue;                                              // lower triangle
        double a0 = 0.0, a1 = 0.0;
#pragma unroll 8
        for (int j = 0; j < LEV_STRIPE; j += 2) {
            a0 = fma(sh.tile[p][j], sh.tile[q][j], a0);
            a1 = fma(sh.tile[p][j + 1], sh.tile[q][j + 1], a1);
        }
        out[p * m + q] = a0 + a1;
    }
    __threadfence();
    __syncthreads();
    if (tid == 0) sh.last = (__hip_atomic_fetch_add(counter, 1, __ATOMIC_ACQ_REL, __HIP_MEMORY_SCOPE_AGENT) == (int)gridDim.x - 1);
    __syncthreads();
    if (!sh.last) return;
    __threadfence();
    const double* pin = part;                                             // written by other workgroups of this launch
    for (int e = tid; e < m * m; e += 256) {
        const int p = e / m, q = e - p * m;
        if (q > p) continue;
        double acc = 0.0;
#pragma unroll 16
        for (int bb = 0; bb < nb + nr; ++bb)                               // stripe order: deterministic
            acc += __hip_atomic_load(&pin[(size_t)bb * m * m + e], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        sh.C(p, q) = acc;
    }
    __syncthreads();
    lev_chol_lds(sh, m);
    for (int e = tid; e < m * m; e += 256) {
        const int p = e / m, q = e - p * m;
        Lout[e] = (q <= p) ? sh.C(p, q) : 0.0;
    }
    if (tid == 0) {
        if (sh.bad) *status = 2;
        else if (STAGE == 1 && *status == 0) *status = 1;
    }
}

static int launch_leverage_qr(const double* X, int K, int G, double reg, double* work, double* lev, int* sweeps,
                              double* scratch, hipStream_t st) {
    const int nb = (G + LEV_STRIPE - 1) / LEV_STRIPE, gb = (G + 255) / 256, m = K - 1;
    double* part = scratch;
    const int nr = (m + LEV_STRIPE - 1) / LEV_STRIPE;
    double* L1 = part + (size_t)(nb + 2) * K * K;
    double* L2 = L1 + (size_t)K * K;
    double* bsum = L2 + (size_t)K * K;
    FDX_HIP(hipMemsetAsync(sweeps, 0, 8 * sizeof(int), st));             // [4], [5]: arrival counters; [7]: 1 = done, 2 = refused
    if (m <= 63) {
        hipLaunchKernelGGL((lev_qr_kernel<0, 64>), dim3(nb + nr), dim3(256), 0, st, X, K, G, reg, work, part, nullptr, L1, lev, bsum, sweeps + 5, sweeps + 7);
        hipLaunchKernelGGL((lev_qr_kernel<1, 64>), dim3(nb + nr), dim3(256), 0, st, X, K, G, reg, work, part, L1, L2, lev, bsum, sweeps + 4, sweeps + 7);
        hipLaunchKernelGGL((lev_qr_kernel<2, 64>), dim3(nb), dim3(256), 0, st, X, K, G, reg, work, part, L2, nullptr, lev, bsum, nullptr, sweeps + 7);
    } else {                                                              // 65 - 128 cell types: the same with 128 rows (132 KB of LDS)
        hipLaunchKernelGGL((lev_qr_kernel<0, 128>), dim3(nb + nr), dim3(256), 0, st, X, K, G, reg, work, part, nullptr, L1, lev, bsum, sweeps + 5, sweeps + 7);
        hipLaunchKernelGGL((lev_qr_kernel<1, 128>), dim3(nb + nr), dim3(256), 0, st, X, K, G, reg, work, part, L1, L2, lev, bsum, sweeps + 4, sweeps + 7);
        hipLaunchKernelGGL((lev_qr_kernel<2, 128>), dim3(nb), dim3(256), 0, st, X, K, G, reg, work, part, L2, nullptr, lev, bsum, nullptr, sweeps + 7);
    }
    hipLaunchKernelGGL(lev_normalise_kernel, dim3(gb), dim3(256), 0, st, lev, G, bsum, nb, reg);
    FDX_CHECK_LAUNCH();
    return 0;
}

// ---------------------------------------------------------------------------------------------------------------------
// The same CholeskyQR2 for 129 - 272 cell types: the (K - 1)^2 matrices no longer fit in LDS beside a stripe, so the K x K part lives
// in global memory (L2-resident: 0.6 MB at 272 types) and every step is a kernel of its own -
//   Z (Householder-deflated, centred genes) and the ridge rows sqrt(reg) I as extra columns      lev_big_z_kernel
//   C = Z Z^T over all columns, 32 x 32 output blocks, columns in order (deterministic)           lev_big_gram_kernel
//   C = L L^T in place, one workgroup of 1024, left-looking by block columns of 32, pivots watched  lev_big_chol_kernel
//   Z <- L^-1 Z by stripes of 64 columns (stripe in LDS, L by scalar loads, blocked by 32 rows)   lev_big_forward_kernel<1>
//   again Gram, Cholesky, then lev_g = |L2^-1 q1_g|^2                                             lev_big_forward_kernel<2>
// (The Jacobi SVD passes take 0.65 s at 200 types; this takes a few milliseconds.)
constexpr int LEV_BIG_MAX_K = 272;

__global__ __launch_bounds__(256) void lev_big_z_kernel(const double* __restrict__ X, int K, int G, double reg, double* __restrict__ Z,
                                                        int ldz, int nb) {
    __shared__ double s_mean[LEV_STRIPE], s_shift[LEV_STRIPE];
    const int tid = threadIdx.x, lane = tid & 63, q = tid >> 6, m = K - 1;
    const int b = blockIdx.x, c0 = b * LEV_STRIPE;
    if (b >= nb) {                                                        // ridge rows 64 (b - nb) ... as columns nb * 64 + r
        for (int k = q; k < m; k += 4) Z[(size_t)k * ldz + c0 + lane] = (k == (b - nb) * LEV_STRIPE + lane) ? sqrt(reg) : 0.0;
        return;
    }
    const int g = c0 + lane;
    if (tid < LEV_STRIPE) {
        double mean = 0.0, shift = 0.0;
        if (g < G) {
            for (int k = 0; k < K; ++k) mean += X[(size_t)k * G + g];
            mean /= (double)K;                                            // genes.py:264
            double sc = 0.0;
            for (int k = 0; k < K; ++k) sc += X[(size_t)k * G + g] - mean;
            const double rk = sqrt((double)K);
            const double xl = X[(size_t)m * G + g] - mean;
            shift = (sc / rk + xl) / (rk + 1.0);                          // Householder v = 1/sqrt(K) + e_K applied to xc
        }
        s_mean[tid] = mean;
        s_shift[tid] = shift;
    }
    __syncthreads();
    const double mean = s_mean[lane], shift = s_shift[lane];
    for (int k = q; k < m; k += 4) Z[(size_t)k * ldz + g] = (g < G) ? (X[(size_t)k * G + g] - mean) - shift : 0.0;
}

// C (m x m, lower triangle written) = Z Z^T over all ldz columns
__global__ __launch_bounds__(256) void lev_big_gram_kernel(const double* __restrict__ Z, int ldz, int m, double* __restrict__ C) {
    __shared__ double P[32][LEV_STRIPE + 1], Q[32][LEV_STRIPE + 1];
    int bi = 0, rem = blockIdx.x;
    while (rem > bi) { rem -= bi + 1; ++bi; }
    const int bj = rem;                                                   // bj <= bi
    const int tid = threadIdx.x, ty = tid >> 4, tx = tid & 15;
    double a00 = 0.0, a01 = 0.0, a10 = 0.0, a11 = 0.0;
    for (int col0 = 0; col0 < ldz; col0 += LEV_STRIPE) {
        for (int e = tid; e < 32 * LEV_STRIPE; e += 256) {
            const int r = e >> 6, j = e & 63;
            const int pi = bi * 32 + r, qi = bj * 32 + r;
            P[r][j] = pi < m ? Z[(size_t)pi * ldz + col0 + j] : 0.0;
            Q[r][j] = qi < m ? Z[(size_t)qi * ldz + col0 + j] : 0.0;
        }
        __syncthreads();
#pragma unroll 8
        for (int j = 0; j < LEV_STRIPE; ++j) {
            const double p0 = P[2 * ty][j], p1 = P[2 * ty + 1][j], q0 = Q[2 * tx][j], q1 = Q[2 * tx + 1][j];
            a00 = fma(p0, q0, a00);
            a01 = fma(p0, q1, a01);
            a10 = fma(p1, q0, a10);
            a11 = fma(p1, q1, a11);
        }
        __syncthreads();
    }
    const int r0 = bi * 32 + 2 * ty, c0 = bj * 32 + 2 * tx;
    if (r0 < m && c0 <= r0) C[(size_t)r0 * m + c0] = a00;
    if (r0 < m && c0 + 1 <= r0) C[(size_t)r0 * m + c0 + 1] = a01;
    if (r0 + 1 < m && c0 <= r0 + 1 && c0 < m) C[(size_t)(r0 + 1) * m + c0] = a10;
    if (r0 + 1 < m && c0 + 1 <= r0 + 1 && c0 + 1 < m) C[(size_t)(r0 + 1) * m + c0 + 1] = a11;
}

// lower Cholesky factor of C in place (global memory), one workgroup; Lout = the factor with zeros above the diagonal
__global__ __launch_bounds__(1024) void lev_big_chol_kernel(double* C, int m, double* __restrict__ Lout, int* status, int final_stage) {
    __shared__ double D[32][33];
    __shared__ double s_thr;
    __shared__ int s_bad;
    const int tid = threadIdx.x;
    if (tid < 64) {
        double mx = 0.0;
        for (int i = tid; i < m; i += 64) mx = fmax(mx, C[(size_t)i * m + i]);
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) mx = fmax(mx, __shfl_xor(mx, off, 64));
        if (tid == 0) {
            s_thr = mx * LEV_PIVOT_TOL;
            s_bad = !(mx > 0.0) || !(mx < 1e300);
        }
    }
    __syncthreads();
    for (int J0 = 0; J0 < m; J0 += 32) {
        const int w = min(32, m - J0), rows = m - J0;
        // (a) the block column minus the contributions of the finished columns (k < J0: read only here)
        for (int e = tid; e < rows * w; e += 1024) {
            const int i = J0 + e / w, j = J0 + e % w;
            if (j > i) continue;
            const double* ri = C + (size_t)i * m;
            const double* rj = C + (size_t)j * m;
            double a0 = 0.0, a1 = 0.0;
            for (int k = 0; k < J0; k += 2) {                            // J0 is a multiple of 32
                a0 = fma(ri[k], rj[k], a0);
                a1 = fma(ri[k + 1], rj[k + 1], a1);
            }
            C[(size_t)i * m + j] -= a0 + a1;
        }
        __threadfence();
        __syncthreads();
        // (b) the diagonal block, in LDS
        if (tid < w * w) {
            const int i = tid / w, j = tid % w;
            D[i][j] = (j <= i) ? C[(size_t)(J0 + i) * m + J0 + j] : 0.0;
        }
        __syncthreads();
        for (int j = 0; j < w; ++j) {
            double d = D[j][j];
            const bool small = !(d > s_thr);                              // also NaN
            if (small) d = s_thr > 0.0 ? s_thr : 1.0;
            const double sd = sqrt(d);
            __syncthreads();                                              // everyone has read the pivot
            if (tid == 0) {
                D[j][j] = sd;
                if (small) s_bad = 1;
            }
            if (tid > j && tid < w) D[tid][j] /= sd;
            __syncthreads();
            const int i = tid >> 5, k = tid & 31;
            if (i < w && k <= i && k > j) D[i][k] -= D[i][j] * D[k][j];
            __syncthreads();
        }
        if (tid < w * w) {
            const int i = tid / w, j = tid % w;
            if (j <= i) C[(size_t)(J0 + i) * m + J0 + j] = D[i][j];
        }
        // (c) the rows below the block: x L_JJ^T = c, one thread per row (its own entries only)
        for (int i = J0 + w + tid; i < m; i += 1024) {
            double* r = C + (size_t)i * m + J0;
            for (int j = 0; j < w; ++j) {
                double sv = r[j];
                for (int k = 0; k < j; ++k) sv = fma(-r[k], D[j][k], sv);
                r[j] = sv / D[j][j];
            }
        }
        __threadfence();
        __syncthreads();
    }
    for (int e = tid; e < m * m; e += 1024) {
        const int i = e / m, j = e - i * m;
        Lout[e] = (j <= i) ? C[e] : 0.0;
    }
    if (tid == 0) {
        if (s_bad) *status = 2;
        else if (final_stage && *status == 0) *status = 1;
    }
}

// STAGE 1: columns [64 b, 64 b + 64) of Z <- L^-1 (those columns).  STAGE 2: lev_g = |L^-1 z_g|^2 and the stripe sums.
template <int STAGE>
__global__ __launch_bounds__(256) void lev_big_forward_kernel(double* __restrict__ Z, int ldz, int m, int G, const double* __restrict__ L,
                                                              double* __restrict__ lev, double* __restrict__ bsum) {
    extern __shared__ __attribute__((aligned(16))) double lev_big_lds[];
    double (*tile)[LEV_STRIPE + 1] = reinterpret_cast<double (*)[LEV_STRIPE + 1]>(lev_big_lds);
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6), c0 = blockIdx.x * LEV_STRIPE;
    for (int e = tid; e < m * LEV_STRIPE; e += 256) {
        const int k = e >> 6, j = e & 63;
        tile[k][j] = Z[(size_t)k * ldz + c0 + j];
    }
    __syncthreads();
    for (int I0 = 0; I0 < m; I0 += 32) {
        const int w = min(32, m - I0);
        if (I0 > 0) {                                                     // rows of the block minus the finished rows, 8 rows per wave
            for (int i = I0 + wave; i < I0 + w; i += 4) {
                const double* Li = L + (size_t)i * m;
                double s0 = tile[i][lane], s1 = 0.0;
                for (int k = 0; k < I0; k += 2) {
                    s0 = fma(-Li[k], tile[k][lane], s0);
                    s1 = fma(-Li[k + 1], tile[k + 1][lane], s1);
                }
                tile[i][lane] = s0 + s1;
            }
        }
        __syncthreads();
        if (wave == 0) {                                                  // inside the block: a lane's own column only
            for (int i = I0; i < I0 + w; ++i) {
                const double* Li = L + (size_t)i * m;
                double sv = tile[i][lane];
                for (int k = I0; k < i; ++k) sv = fma(-Li[k], tile[k][lane], sv);
                tile[i][lane] = sv / Li[i];
            }
        }
        __syncthreads();
    }
    if (STAGE == 1) {
        for (int e = tid; e < m * LEV_STRIPE; e += 256) {
            const int k = e >> 6, j = e & 63;
            Z[(size_t)k * ldz + c0 + j] = tile[k][j];
        }
    } else if (wave == 0) {
        double l = 0.0;
        if (c0 + lane < G) {
            for (int k = 0; k < m; ++k) l = fma(tile[k][lane], tile[k][lane], l);
            lev[c0 + lane] = l;
        }
        l = wsum(l);
        if (lane == 0) bsum[blockIdx.x] = l;
    }
}

static int launch_leverage_qr_big(const double* X, int K, int G, double reg, double* lev, int* sweeps, double* scratch, hipStream_t st) {
    const int m = K - 1, nb = (G + LEV_STRIPE - 1) / LEV_STRIPE, nr = (m + LEV_STRIPE - 1) / LEV_STRIPE, gb = (G + 255) / 256;
    const int ldz = (nb + nr) * LEV_STRIPE;
    double* Z = scratch;
    double* C = Z + (size_t)m * ldz;
    double* L1 = C + (size_t)m * m;
    double* L2 = L1 + (size_t)m * m;
    double* bsum = L2 + (size_t)m * m;
    const int tb = (m + 31) / 32, gram_blocks = tb * (tb + 1) / 2;
    const size_t lds = (size_t)m * (LEV_STRIPE + 1) * sizeof(double);
    FDX_HIP(hipFuncSetAttribute((const void*)lev_big_forward_kernel<1>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    FDX_HIP(hipFuncSetAttribute((const void*)lev_big_forward_kernel<2>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    FDX_HIP(hipMemsetAsync(sweeps, 0, 8 * sizeof(int), st));             // [7]: 1 = done, 2 = refused
    hipLaunchKernelGGL(lev_big_z_kernel, dim3(nb + nr), dim3(256), 0, st, X, K, G, reg, Z, ldz, nb);
    hipLaunchKernelGGL(lev_big_gram_kernel, dim3(gram_blocks), dim3(256), 0, st, Z, ldz, m, C);
    hipLaunchKernelGGL(lev_big_chol_kernel, dim3(1), dim3(1024), 0, st, C, m, L1, sweeps + 7, 0);
    hipLaunchKernelGGL(lev_big_forward_kernel<1>, dim3(nb + nr), dim3(256), lds, st, Z, ldz, m, G, L1, lev, bsum);
    hipLaunchKernelGGL(lev_big_gram_kernel, dim3(gram_blocks), dim3(256), 0, st, Z, ldz, m, C);
    hipLaunchKernelGGL(lev_big_chol_kernel, dim3(1), dim3(1024), 0, st, C, m, L2, sweeps + 7, 1);
    hipLaunchKernelGGL(lev_big_forward_kernel<2>, dim3(nb), dim3(256), lds, st, Z, ldz, m, G, L2, lev, bsum);
    hipLaunchKernelGGL(lev_normalise_kernel, dim3(gb), dim3(256), 0, st, lev, G, bsum, nb, reg);
    FDX_CHECK_LAUNCH();
    return 0;
}

bool leverage_qr_applies(int K, int G) {
    return K >= 2 && K <= LEV_BIG_MAX_K && G >= 1 && !fdx::exp_env("FDX_LEV_NO_QR") && !fdx::env("FDX_LEV_ONE_WG") &&
           (K <= 128 || !fdx::exp_env("FDX_LEV_NO_QR_BIG"));
}

size_t leverage_scratch_doubles(int K, int G) {
    const size_t nb = (size_t)(G + LEV_STRIPE - 1) / LEV_STRIPE;
    const size_t small = (nb + 2) * K * K + 2 * (size_t)K * K + nb + 16;   // the larger of the Jacobi / Cholesky-QR layouts up to 128 types
    const size_t nr = (size_t)(K + LEV_STRIPE - 1) / LEV_STRIPE;
    const size_t big = (size_t)K * (nb + nr) * LEV_STRIPE + 3 * (size_t)K * K + nb + 16;
    return std::max(small, big);
}

static int launch_leverage_multi(const double* X, int K, int G, double reg, double* work, double* sig2, double* lev,
                                 int* sweeps, double* scratch, hipStream_t st) {
    const int nb = (G + LEV_STRIPE - 1) / LEV_STRIPE, gb = (G + 255) / 256;
    double* part = scratch;
    double* V = part + (size_t)nb * K * K;
    double* bsum = V + (size_t)K * K;
    int* done = sweeps + 6;                                               // sweeps: 8 ints, [0] = passes, [6] = done flag
    FDX_HIP(hipMemsetAsync(sweeps, 0, 8 * sizeof(int), st));
    constexpr size_t kLds = (4 * 64 * 65 + 64 + 128 + 64) * sizeof(double);   // C and V twice (lev_eigen_kernel), rotation scratch
    FDX_HIP(hipFuncSetAttribute((const void*)lev_eigen_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)kLds));
    hipLaunchKernelGGL(lev_centre_kernel, dim3(gb), dim3(256), 0, st, X, K, G, work);
    for (int pass = 0; pass < LEV_PASSES; ++pass) {
        hipLaunchKernelGGL(lev_gram_kernel, dim3(nb), dim3(256), 0, st, work, K, G, part, done);
        hipLaunchKernelGGL(lev_eigen_kernel, dim3(1), dim3(512), kLds, st, part, nb, K, V, sig2, done, sweeps);
        hipLaunchKernelGGL(lev_rotate_kernel, dim3(gb), dim3(256), 0, st, work, K, G, V, done);
    }
    hipLaunchKernelGGL(lev_gram_kernel, dim3(nb), dim3(256), 0, st, work, K, G, part, done);
    hipLaunchKernelGGL(lev_diag_kernel, dim3(1), dim3(64), 0, st, part, nb, K, sig2, done);
    hipLaunchKernelGGL(lev_scores_kernel, dim3(gb), dim3(256), 0, st, work, K, G, sig2, reg, lev, bsum);
    hipLaunchKernelGGL(lev_normalise_kernel, dim3(gb), dim3(256), 0, st, lev, G, bsum, gb, reg);
    FDX_CHECK_LAUNCH();
    return 0;
}

// X: device (K, G) row-major; work: device K*G doubles; sig2: device K doubles; lev: device G doubles; sweeps: 8 ints;
// scratch: leverage_scratch_doubles(K, G) doubles
int launch_leverage(const double* X, int K, int G, double reg, double* work, double* sig2, double* lev, int* sweeps,
                    double* scratch, hipStream_t st, int route) {
    if (K <= 0 || G <= 0) return fail(FDX_ERR_INVALID, "leverage: empty reference matrix");
    if (route == LEV_ROUTE_QR) {
        if (!leverage_qr_applies(K, G) || !scratch) return fail(FDX_ERR_INVALID, "leverage: the Cholesky-QR route needs 2 <= K <= 272");
        if (K > 128) return launch_leverage_qr_big(X, K, G, reg, lev, sweeps, scratch, st);
        return launch_leverage_qr(X, K, G, reg, work, lev, sweeps, scratch, st);
    }
    if (K <= 64 && K >= 2 && scratch && !fdx::env("FDX_LEV_ONE_WG"))
        return launch_leverage_multi(X, K, G, reg, work, sig2, lev, sweeps, scratch, st);
    constexpr size_t kLevLds = (2 * 64 * 65 + 64) * sizeof(double);   // Gram matrix, rotations, per-round (c, s) pairs
    FDX_HIP(hipFuncSetAttribute((const void*)leverage_jacobi_kernel<0>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)kLevLds));
    FDX_HIP(hipFuncSetAttribute((const void*)leverage_jacobi_kernel<1>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)kLevLds));
    FDX_HIP(hipFuncSetAttribute((const void*)leverage_jacobi_kernel<2>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)kLevLds));
    FDX_HIP(hipFuncSetAttribute((const void*)leverage_jacobi_kernel<4>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)kLevLds));
    const int nch = (G + 511) / 512;
    if (nch == 1) hipLaunchKernelGGL(leverage_jacobi_kernel<1>, dim3(1), dim3(512), kLevLds, st, X, K, G, reg, work, sig2, lev, sweeps);
    else if (nch == 2) hipLaunchKernelGGL(leverage_jacobi_kernel<2>, dim3(1), dim3(512), kLevLds, st, X, K, G, reg, work, sig2, lev, sweeps);
    else if (nch <= 4) hipLaunchKernelGGL(leverage_jacobi_kernel<4>, dim3(1), dim3(512), kLevLds, st, X, K, G, reg, work, sig2, lev, sweeps);
    else hipLaunchKernelGGL(leverage_jacobi_kernel<0>, dim3(1), dim3(512), kLevLds, st, X, K, G, reg, work, sig2, lev, sweeps);
    FDX_CHECK_LAUNCH();
    return 0;
}

}  // namespace fdx
