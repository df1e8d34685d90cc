// Internal interface of the BCD solve driver (solver.cpp).
#pragma once
#include <vector>

#include "fdx_graph.h"
#include "fdx_internal.h"
#include "fdx_kernels.h"

namespace fdx {

struct SolveProblem {
    const fdx_graph* graph = nullptr;
    const double* H = nullptr;     // device (K, ldh) type-major
    long long ldh = 0;
    const double* XtX = nullptr;   // device (K, K)
    double* beta[2] = {nullptr, nullptr};  // device (K, ld) type-major double buffer
    long long ld = 0;
    int K = 0;                     // planes of H / beta, order of XtX (solver_padded_K of the cell types: pad types are all zero)
    int K_real = 0;                // cell types (0: = K); beta0 = 1 / K_real on the first K_real planes
    double YtY = 0.0;
    double lambda = 0.0;
    double rho_eff = 0.0;          // rho * mean(diag XtX)
    int max_iter = 100;
    double tol = 1e-4;
    int verbose = 0;
    int init_beta = 1;             // fill beta[0] with 1/K and clear the pad rows
    int beta0_virtual = 0;         // init_beta == 0 only: the caller cleared the pad rows of BOTH buffers but did not write 1/K into
                                   // beta[0] - the first sweep takes the constant instead of reading it (tiled kernel, no pad types);
                                   // where that kernel does not apply, solver_run writes the start vector itself
    int compute_objective = 1;
    int first_chunk = 4;
};

struct SolveResult {
    int result_buffer = 0;         // index into SolveProblem::beta holding the final abundances
    int n_iterations = 0;
    int converged = 0;
    double final_change = 0.0;
    double final_objective = 0.0;
    double sweep_ms = 0.0;
    std::vector<int> objective_iters;
    std::vector<double> objectives;
    std::vector<double> rel_changes;
};

int solver_init_beta(double* beta, long long ld, long long n_fill, int K, hipStream_t st, int K_planes = 0);   // K_planes > K: the rest zero
int solver_pad_square(const double* A, int K, double* B, int KP, hipStream_t st);   // B (KP, KP) = A (K, K) bordered with zeros
// zero the pad columns [n_used, ld) of a type-major (K, ld) array
int solver_zero_pad(double* b, long long ld, long long n_used, int K, hipStream_t st);
// The four sums of the objective on the device (out4_dev); the sharded driver all-reduces them across ranks.
int solver_objective_partials(const fdx_graph& g, const double* beta, long long ld, const double* H, long long ldh,
                              const double* XtX, int K, double* scratch_partials, double* out4_dev, hipStream_t st);
int solver_objective(const fdx_graph& g, const double* beta, long long ld, const double* H, long long ldh,
                     const double* XtX, int K, double YtY, double lambda, double rho_eff, double* scratch_partials,
                     double* scratch_out4, double* obj_host, hipStream_t st);
int solver_run(const SolveProblem& p, SolveResult* res, hipStream_t st);

}  // namespace fdx
