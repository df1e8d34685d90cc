// Per-cell-type signatures from a single-cell reference already in HBM: X[k, :] = mean (or sum) over the cells of type k
// (flashdeconv/io/loader.py:114-135 load_reference).  One-off, not on the fit path; here so that an AnnData whose matrices
// live on the device never has to come back to the host.
//
// The cells of a type are given as a list sorted by type and, inside a type, by ascending row (a stable sort of the labels
// on the host: the labels are host metadata).  Both kernels add the rows of a type in that order - the order of numpy's
// axis-0 reduction - in float64, so the result does not depend on the launch geometry.
#include "fdx_internal.h"
#include "fdx_kernels.h"

namespace fdx {

// dense: block = (256-gene tile, type); thread = gene
template <typename T>
__global__ __launch_bounds__(256) void type_sums_dense_kernel(const T* __restrict__ Y, long long ldy, const int* __restrict__ rows,
                                                              const int* __restrict__ type_off, int G, int mean,
                                                              double* __restrict__ X) {
    const int k = blockIdx.y;
    const int g = blockIdx.x * 256 + threadIdx.x;
    if (g >= G) return;
    const int a = type_off[k], b = type_off[k + 1];
    double acc = 0.0;
    for (int i = a; i < b; ++i) acc += (double)Y[(size_t)rows[i] * (size_t)ldy + g];
    if (mean) acc /= (double)(b - a);              // an empty type gives 0 / 0 = NaN, as numpy's mean of an empty slice
    X[(size_t)k * G + g] = acc;
}

// CSR: one workgroup per type walks the type's rows one after the other; the lanes share a row's entries (the columns of a
// canonical row are distinct; atomicAdd only makes a row with repeated columns well defined)
template <typename T>
__global__ __launch_bounds__(256) void type_sums_csr_kernel(const long long* __restrict__ indptr, const int* __restrict__ indices,
                                                            const T* __restrict__ values, const int* __restrict__ rows,
                                                            const int* __restrict__ type_off, int G, int mean,
                                                            double* __restrict__ X) {
    const int k = blockIdx.x;
    double* out = X + (size_t)k * G;
    for (int g = threadIdx.x; g < G; g += 256) out[g] = 0.0;
    __syncthreads();
    const int a = type_off[k], b = type_off[k + 1];
    for (int i = a; i < b; ++i) {
        const long long r = rows[i];
        const long long e0 = indptr[r], e1 = indptr[r + 1];
        for (long long e = e0 + threadIdx.x; e < e1; e += 256) atomicAdd(&out[indices[e]], (double)values[e]);
        __threadfence_block();
        __syncthreads();                            // the next row adds after this one: fixed order per (type, gene)
    }
    if (mean) {
        const double cnt = (double)(b - a);
        for (int g = threadIdx.x; g < G; g += 256) out[g] /= cnt;
    }
}

}  // namespace fdx

using namespace fdx;

extern "C" {

int fdx_type_sums_dev(const void* Y_dev, int32_t dtype, int64_t n, int32_t G, int64_t ldy, const int32_t* rows_dev,
                      const int32_t* type_off_dev, int32_t K, int32_t mean, double* X_out_dev, void* stream) {
    FDX_REQUIRE(Y_dev && rows_dev && type_off_dev && X_out_dev, "fdx_type_sums_dev: null argument");
    FDX_REQUIRE(n >= 0 && G > 0 && K > 0 && ldy >= G, "fdx_type_sums_dev: bad shape");
    FDX_REQUIRE(dtype == FDX_F32 || dtype == FDX_F64, "fdx_type_sums_dev: dtype must be FDX_F32 or FDX_F64");
    hipStream_t st = (hipStream_t)stream;
    PoolStream pool_stream(st);
    const dim3 grid((unsigned)ceil_div(G, 256), (unsigned)K);
    if (dtype == FDX_F32)
        hipLaunchKernelGGL(type_sums_dense_kernel<float>, grid, dim3(256), 0, st, (const float*)Y_dev, (long long)ldy, rows_dev,
                           type_off_dev, G, mean, X_out_dev);
    else
        hipLaunchKernelGGL(type_sums_dense_kernel<double>, grid, dim3(256), 0, st, (const double*)Y_dev, (long long)ldy, rows_dev,
                           type_off_dev, G, mean, X_out_dev);
    FDX_CHECK_LAUNCH();
    return 0;
}

int fdx_type_sums_csr_dev(const fdx_csr_view* Y, const int32_t* rows_dev, const int32_t* type_off_dev, int32_t K, int32_t mean,
                          double* X_out_dev, void* stream) {
    FDX_REQUIRE(Y && rows_dev && type_off_dev && X_out_dev, "fdx_type_sums_csr_dev: null argument");
    FDX_REQUIRE(Y->G > 0 && K > 0, "fdx_type_sums_csr_dev: bad shape");
    FDX_REQUIRE(Y->dtype == FDX_F32 || Y->dtype == FDX_F64, "fdx_type_sums_csr_dev: dtype must be FDX_F32 or FDX_F64");
    hipStream_t st = (hipStream_t)stream;
    PoolStream pool_stream(st);
    if (Y->dtype == FDX_F32)
        hipLaunchKernelGGL(type_sums_csr_kernel<float>, dim3((unsigned)K), dim3(256), 0, st, (const long long*)Y->indptr, Y->indices,
                           (const float*)Y->data, rows_dev, type_off_dev, Y->G, mean, X_out_dev);
    else
        hipLaunchKernelGGL(type_sums_csr_kernel<double>, dim3((unsigned)K), dim3(256), 0, st, (const long long*)Y->indptr, Y->indices,
                           (const double*)Y->data, rows_dev, type_off_dev, Y->G, mean, X_out_dev);
    FDX_CHECK_LAUNCH();
    return 0;
}

}  // extern "C"
