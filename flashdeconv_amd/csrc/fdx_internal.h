// Internal helpers shared by the libfdx translation units (not part of the C ABI).
#pragma once
#include <hip/hip_runtime.h>

#include "../../include/fdx.h"

#include <cstdint>
#include <cstdio>
#include <condition_variable>
#include <cstring>
#include <functional>
#include <memory>
#include <mutex>
#include <string>

namespace fdx {

// ---- error plumbing --------------------------------------------------------------------------
// Every C-ABI entry returns 0 on success or a negative code; the message is kept per thread and
// read back with fdx_last_error().
// (codes: FDX_OK / FDX_ERR_* from include/fdx.h)

void set_error(const std::string& msg);
std::string get_error();
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
// Caching device allocator (pool.cpp): hipMalloc/hipFree cost 0.1-several ms and hipFree synchronises the device, so
// scratch blocks are recycled through per-device free lists keyed by a rounded capacity.
// Stream ordering: every entry point that queues work on a stream names it (PoolStream); a block remembers the stream of
// the entry point that allocated it, and a later owner on a different stream is ordered behind the work queued there
// (pool.cpp).  Buffers handed to a library side stream inside an entry point are covered by that entry's own events /
// drains before they are released; error returns drain the device (capi.cpp: fail()).
int pool_alloc(size_t bytes, void** p, size_t* cap);
hipStream_t pool_set_stream(hipStream_t s);      // returns the previous one
struct PoolStream {
    hipStream_t prev;
    explicit PoolStream(hipStream_t s) : prev(pool_set_stream(s)) {}
    ~PoolStream() { pool_set_stream(prev); }
    PoolStream(const PoolStream&) = delete;
    PoolStream& operator=(const PoolStream&) = delete;
};
void pool_free(void* p, size_t cap);
void pool_mark_idle(void* p);                    // the owner has waited for all work on the block: no ordering on reuse
void pool_trim();   // return every cached block to the driver
// Thread-local pinned host scratch, a few grow-only slots: the target of asynchronous device-to-host copies (into pageable
// memory hipMemcpyAsync returns only when the copy has been done, i.e. the host waits for everything queued before it).
void* pinned_scratch(int slot, size_t bytes);   // nullptr on failure; slot 0..7
constexpr size_t FDX_PINNED_BLOCK_BYTES = 1024;
void* pinned_block_get();                        // a recycled pinned block of FDX_PINNED_BLOCK_BYTES (nullptr on failure) ...
void pinned_block_put(void* p);                  // ... and back
void* pinned_buffer_get(size_t bytes, size_t* cap_out);   // a recycled pinned buffer of at least `bytes` (capacity class in *cap_out)
void pinned_buffer_put(void* p, size_t cap);

// Helper thread (helper_thread.cpp): fn runs there with the caller's device current; helper_wait returns fn's code (and makes its
// message this thread's last error).  fn must only queue device work, never wait for the device.
struct HelperTicket { std::mutex mu; std::condition_variable cv; bool done = false; int rc = 0; std::string err; };
std::shared_ptr<HelperTicket> helper_submit(std::function<int()> fn);
int helper_wait(const std::shared_ptr<HelperTicket>& ticket);

// Copies between the device and CALLER (possibly pageable) memory, through recycled pinned buffers for 64 KB - 32 MB (pool.cpp: the
// driver must never pin pages the caller may unmap).  copy_h2d: asynchronous like hipMemcpyAsync (src may be reused at once);
// copy_d2h: complete on return (it synchronises the stream).
int copy_h2d(void* dst_dev, const void* src_host, size_t bytes, hipStream_t st);
int copy_d2h(void* dst_host, const void* src_dev, size_t bytes, hipStream_t st);

// the library's per-device non-blocking side stream (its own priority: fit.cpp); nullptr if it cannot be made
hipStream_t library_side_stream();

struct DevBuf {
    void* p = nullptr;
    size_t bytes = 0;   // requested size
    size_t cap = 0;     // capacity class of the pooled block
    DevBuf() = default;
    DevBuf(const DevBuf&) = delete;
    DevBuf& operator=(const DevBuf&) = delete;
    ~DevBuf() { release(); }
    int alloc(size_t n) {
        release();
        if (n == 0) n = 8;
        FDX_TRY(pool_alloc(n, &p, &cap));
        bytes = n;
        return 0;
    }
    void release() {
        if (p) pool_free(p, cap);
        p = nullptr;
        bytes = 0;
        cap = 0;
    }
    template <class T> T* as() const { return reinterpret_cast<T*>(p); }
    void mark_idle() { pool_mark_idle(p); }
    void take(DevBuf& o) {   // move ownership
        release();
        p = o.p;
        bytes = o.bytes;
        cap = o.cap;
        o.p = nullptr;
        o.bytes = 0;
        o.cap = 0;
    }
};

// Which columns of a CSR spot matrix are selected genes, and their CountSketch {weight, bucket}: device tables of the CSR sketch
// kernels (csr_kernels.cpp), built from the host-side gene list of a fit / a shard's prepare.
struct CsrSelection {
    DevBuf slots, bits;        // two-kernel path: per column {weight, bucket}, u32 bitmap
    DevBuf words, w, b;        // fused path: per 32 columns {bitmap word, rank of its first column}; weight / bucket by rank
    int sel_words = 0, n_sel = 0;
    // gene_idx: G selected columns of G_all (NULL = all, in order); fused: the tables of the fused kernel, else of the two-kernel path
    int build(const int32_t* gene_idx, int G, int G_all, const int32_t* bucket, const double* weight, int d, bool fused,
              hipStream_t st, const char* who);
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
