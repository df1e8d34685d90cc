// Microbenchmark: throughput of LDS atomics by type (MI355X).  16 waves per CU, one workgroup per CU, every lane adds to a
// pseudo-random slot of a 64 KB LDS array (distinct slots inside a wave-instruction, as the entries of one CSR row are).
//   hipcc --offload-arch=gfx950 -O3 tools/micro/lds_atomic_rate.cpp -o gpurun_out/lds_atomic_rate && gpurun_out/lds_atomic_rate
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>

template <typename T> __device__ __forceinline__ void lds_add(T* p, T v) { __hip_atomic_fetch_add(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP); }

template <typename T, int MODE>   // MODE 0: atomic add, 1: plain read + write (racy), 2: read only
__global__ __launch_bounds__(1024) void k(int iters, T* out) {
    __shared__ T acc[8192];
    for (int i = threadIdx.x; i < 8192; i += 1024) acc[i] = (T)0;
    __syncthreads();
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    unsigned s = wave * 7919u + 13u;
    T sum = (T)0;
    for (int it = 0; it < iters; ++it) {
        s = s * 1664525u + 1013904223u;
        const int idx = (int)(((s >> 8) * 64u + (unsigned)lane * 127u) & 8191u);   // distinct across the lanes of a wave (127 odd), scattered
        if (MODE == 0) lds_add(&acc[idx], (T)1);
        else if (MODE == 1) acc[idx] = acc[idx] + (T)1;
        else sum += acc[idx];
    }
    __syncthreads();
    if (threadIdx.x == 0) out[blockIdx.x] = acc[5] + sum;
}

template <typename T, int MODE> void run(const char* name) {
    T* out; hipMalloc(&out, 256 * sizeof(T));
    const int iters = 20000;
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    k<T, MODE><<<256, 1024>>>(100, out);
    hipEventRecord(a);
    k<T, MODE><<<256, 1024>>>(iters, out);
    hipEventRecord(b); hipEventSynchronize(b);
    float ms; hipEventElapsedTime(&ms, a, b);
    const double lane_ops = 1024.0 * iters;                 // per CU
    const double cycles = ms * 1e-3 * 2.4e9;
    printf("%-28s %8.3f ms  %6.2f lanes/clk/CU  (%5.1f clk per wave-instruction)\n", name, ms, lane_ops / cycles, cycles / (16.0 * iters));
    hipFree(out);
}

int main() {
    run<double, 0>("ds_add_f64");
    run<float, 0>("ds_add_f32");
    run<unsigned long long, 0>("ds_add_u64");
    run<unsigned, 0>("ds_add_u32");
    run<double, 1>("read+write b64 (racy)");
    run<double, 2>("ds_read_b64");
    return 0;
}
