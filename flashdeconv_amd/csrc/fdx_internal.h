// Internal helpers shared by the libfdx translation units (not part of the C ABI).
#pragma once
#include <hip/hip_runtime.h>

#include "../../include/fdx.h"

#include <cstdint>
#include <cstdio>
#include <cstring>
#include <string>

namespace fdx {

// ---- error plumbing --------------------------------------------------------------------------
// Every C-ABI entry returns 0 on success or a negative code; the message is kept per thread and
// read back with fdx_last_error().
// (codes: FDX_OK / FDX_ERR_* from include/fdx.h)

void set_error(const std::string& msg);
int fail(int code, const std::string& msg);

#define FDX_HIP(expr)                                                                             \
    do {                                                                                          \
        hipError_t _e = (expr);                                                                   \
        if (_e != hipSuccess) {                                                                   \
            return ::fdx::fail(FDX_ERR_HIP, std::string(#expr) + ": " + hipGetErrorString(_e)); \
        }                                                                                         \
    } while (0)

#define FDX_REQUIRE(cond, msg)                                               \
    do {                                                                     \
        if (!(cond)) return ::fdx::fail(FDX_ERR_INVALID, (msg));      \
    } while (0)

#define FDX_TRY(expr)              \
    do {                           \
        int _rc = (expr);          \
        if (_rc != 0) return _rc;  \
    } while (0)

inline hipError_t last_launch_error() { return hipGetLastError(); }

#define FDX_CHECK_LAUNCH() FDX_HIP(::fdx::last_launch_error())

// ---- device scratch buffer (RAII) ---------------------------------------------------------------
struct DevBuf {
    void* p = nullptr;
    size_t bytes = 0;
    DevBuf() = default;
    DevBuf(const DevBuf&) = delete;
    DevBuf& operator=(const DevBuf&) = delete;
    ~DevBuf() { release(); }
    int alloc(size_t n) {
        release();
        if (n == 0) n = 8;
        hipError_t e = hipMalloc(&p, n);
        if (e != hipSuccess) {
            p = nullptr;
            return fail(FDX_ERR_HIP, std::string("hipMalloc(") + std::to_string(n) + "): " + hipGetErrorString(e));
        }
        bytes = n;
        return 0;
    }
    void release() {
        if (p) (void)hipFree(p);
        p = nullptr;
        bytes = 0;
    }
    template <class T> T* as() const { return reinterpret_cast<T*>(p); }
    void take(DevBuf& o) {   // move ownership
        release();
        p = o.p;
        bytes = o.bytes;
        o.p = nullptr;
        o.bytes = 0;
    }
};

inline int ceil_div(long long a, long long b) { return (int)((a + b - 1) / b); }
inline long long round_up(long long a, long long b) { return (a + b - 1) / b * b; }

// Contiguous-per-XCD block remap: blocks that the dispatcher places on one XCD (same blockIdx % 8
// label) get a contiguous range of logical tiles, so neighbouring tiles share that XCD's L2.
// Bijective for any grid size (cdna guide, "XCD swizzle must be bijective").
__device__ __forceinline__ int xcd_remap(int bid, int nblk) {
    const int q = nblk >> 3, r = nblk & 7, x = bid & 7;
    const int base = (x < r) ? x * (q + 1) : r * (q + 1) + (x - r) * q;
    return base + (bid >> 3);
}

}  // namespace fdx
