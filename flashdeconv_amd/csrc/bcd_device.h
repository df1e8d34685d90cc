// Device helpers shared by the BCD sweep translation units (internal).
#pragma once
#include "fdx_internal.h"
#include "fdx_kernels.h"

namespace fdx {

static __device__ __forceinline__ double wave_max(double v) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v = fmax(v, __shfl_xor(v, off, 64));
    return v;
}

// Fold the 64+64 slots of one iteration's statistics into rel_change (every lane gets the same value).
static __device__ __forceinline__ double fold_rel_change(const unsigned long long* slots, int lane) {
    const double d = wave_max(__longlong_as_double((long long)slots[lane]));
    const double a = wave_max(__longlong_as_double((long long)slots[64 + lane]));
    return d / (a + 1e-10);
}


}  // namespace fdx
