// Large transfers between CALLER memory (pageable: a numpy array, a scipy matrix's arrays) and HBM.
//
// The reference's users hand FlashDeconv.fit host arrays (flashdeconv/core/deconv.py:237-243, README.md:118-129): for the literal
// drop-in the spot matrix has to cross PCIe first - 8 GB at 1M x 2000 float32, ~150 ms at the link's rate, forty times the fit.
// hipMemcpy on pageable memory stages through the driver on ONE thread (10-15 GB/s) and pins the caller's pages on the way (the MMU
// notifier stalls of pool.cpp).  Here a team of host threads copies - and, for integer counts, converts: the reference's
// astype(float64) on the host becomes a narrowing to the float32 / float64 the kernels stream, done while the bytes are moved anyway -
// chunk by chunk into a ring of recycled pinned buffers, each chunk's DMA queued as soon as it is filled: the CPU copies of later
// chunks run under the DMA of earlier ones, and the caller's memory is only ever touched by plain loads.
#include "fdx_env.h"
#include <algorithm>
#include <atomic>
#include <cmath>
#include <cstdio>
#include <cstring>
#include <thread>
#include <type_traits>
#include <vector>

#include "fdx_internal.h"

namespace fdx {
unsigned host_cpu_budget();          // kdtree_order.cpp: hardware threads cut to the control group's CPU quota
}

namespace {

using namespace fdx;

constexpr size_t kChunkBytes = 16u << 20;       // destination bytes per chunk (one pinned ring slot)
constexpr int kSlotsPerThread = 2;

size_t src_itemsize(int code) {
    switch (code) {
        case FDX_SRC_F32: case FDX_SRC_I32: case FDX_SRC_U32: return 4;
        case FDX_SRC_F64: case FDX_SRC_I64: case FDX_SRC_U64: return 8;
        case FDX_SRC_I16: case FDX_SRC_U16: return 2;
        case FDX_SRC_I8: case FDX_SRC_U8: return 1;
        default: return 0;
    }
}

template <typename S, typename D>
double convert_span(const void* src, void* dst, size_t count) {
    // the largest |value| in the source's own integer domain (unsigned magnitude): a loop the compiler vectorises - with the
    // maximum taken in double the converting copy ran at 38-40 GB/s on 16 threads, below the link's 56
    typedef typename std::make_unsigned<S>::type U;
    const S* s = static_cast<const S*>(src);
    D* d = static_cast<D*>(dst);
    U mx = 0;
    for (size_t i = 0; i < count; ++i) {
        const S v = s[i];
        d[i] = (D)v;
        const U a = v < 0 ? (U)((U)0 - (U)v) : (U)v;
        mx = a > mx ? a : mx;
    }
    return (double)mx;
}

// count elements of the source type `code` at src -> count elements of float32 / float64 at dst; returns max |value| of integer
// sources (0 for floating-point ones: not looked at)
template <typename D>
double convert_to(int code, const void* src, void* dst, size_t count) {
    switch (code) {
        case FDX_SRC_F32: { const float* s = (const float*)src; D* d = (D*)dst; for (size_t i = 0; i < count; ++i) d[i] = (D)s[i]; return 0.0; }
        case FDX_SRC_F64: { const double* s = (const double*)src; D* d = (D*)dst; for (size_t i = 0; i < count; ++i) d[i] = (D)s[i]; return 0.0; }
        case FDX_SRC_I8: return convert_span<int8_t, D>(src, dst, count);
        case FDX_SRC_U8: return convert_span<uint8_t, D>(src, dst, count);
        case FDX_SRC_I16: return convert_span<int16_t, D>(src, dst, count);
        case FDX_SRC_U16: return convert_span<uint16_t, D>(src, dst, count);
        case FDX_SRC_I32: return convert_span<int32_t, D>(src, dst, count);
        case FDX_SRC_U32: return convert_span<uint32_t, D>(src, dst, count);
        case FDX_SRC_I64: return convert_span<int64_t, D>(src, dst, count);
        case FDX_SRC_U64: return convert_span<uint64_t, D>(src, dst, count);
        default: return 0.0;
    }
}

// measured on the MI355X boxes (16-CPU quota, PCIe 56.5 GB/s pinned): uploads 54.9 / 54.4 / 52.7 / 39 GB/s on 4 / 8 / 16 / 32 threads,
// downloads into fresh pages 37.5 / 38.8 / 29.9 on 4 / 8 / 16 - a few threads saturate the link, more only contend
int team_size(size_t bytes, int want) {
    int t = (int)std::min<unsigned>(host_cpu_budget(), (unsigned)want);
    if (const char* e = fdx::exp_env("FDX_TRANSFER_THREADS")) t = std::max(1, atoi(e));
    const size_t chunks = (bytes + kChunkBytes - 1) / kChunkBytes;
    return (int)std::max<size_t>(1, std::min<size_t>((size_t)t, chunks));
}

struct Slot {
    void* pin = nullptr;
    size_t cap = 0;
    hipEvent_t ev = nullptr;
    bool busy = false;
};

}  // namespace

// count elements (src_code at src_host) -> dst_dtype (FDX_F32 / FDX_F64) elements at dst_dev.  max_abs_out (may be NULL): the
// largest |value| of an integer source (the caller's check that float32 holds every count exactly).  Returns when the data is in HBM.
extern "C" int fdx_upload_convert_dev(void* dst_dev, int32_t dst_dtype, const void* src_host, int32_t src_code, int64_t count,
                                      double* max_abs_out, void* stream) {
    FDX_REQUIRE(dst_dtype == FDX_F32 || dst_dtype == FDX_F64, "fdx_upload_convert_dev: dst_dtype must be FDX_F32 or FDX_F64");
    const size_t ssz = src_itemsize(src_code);
    FDX_REQUIRE(ssz != 0, "fdx_upload_convert_dev: unknown source type");
    FDX_REQUIRE(count >= 0 && (count == 0 || (dst_dev && src_host)), "fdx_upload_convert_dev: bad arguments");
    if (max_abs_out) *max_abs_out = 0.0;
    if (count == 0) return 0;
    const size_t dsz = dst_dtype == FDX_F32 ? 4 : 8;
    const size_t total = (size_t)count * dsz;
    const size_t per_chunk = kChunkBytes / dsz;                      // elements
    const size_t n_chunks = ((size_t)count + per_chunk - 1) / per_chunk;
    const bool same = (src_code == FDX_SRC_F32 && dst_dtype == FDX_F32) || (src_code == FDX_SRC_F64 && dst_dtype == FDX_F64);
    hipStream_t st = (hipStream_t)stream;
    const int T = team_size(total, same ? 8 : 16);               // (a converting copy is CPU work: the whole budget)
    int dev = 0;
    FDX_HIP(hipGetDevice(&dev));
    std::atomic<size_t> next{0};
    std::atomic<int> rc{0};
    std::vector<double> mx((size_t)T, 0.0);
    auto work = [&](int t) {
        if (hipSetDevice(dev) != hipSuccess) { rc = FDX_ERR_HIP; return; }
        hipStream_t cs = nullptr;                                    // a copy stream of the thread's own: its DMAs queue behind each other only
        if (hipStreamCreateWithFlags(&cs, hipStreamNonBlocking) != hipSuccess) { rc = FDX_ERR_HIP; return; }
        Slot slots[kSlotsPerThread];
        for (Slot& s : slots) {
            s.pin = pinned_buffer_get(kChunkBytes, &s.cap);
            if (!s.pin || hipEventCreateWithFlags(&s.ev, hipEventDisableTiming) != hipSuccess) rc = FDX_ERR_HIP;
        }
        int k = 0;
        while (rc == 0) {
            const size_t c = next.fetch_add(1);
            if (c >= n_chunks) break;
            Slot& s = slots[k];
            k = (k + 1) % kSlotsPerThread;
            if (s.busy && hipEventSynchronize(s.ev) != hipSuccess) { rc = FDX_ERR_HIP; break; }
            const size_t e0 = c * per_chunk, ne = std::min(per_chunk, (size_t)count - e0);
            const char* src = static_cast<const char*>(src_host) + e0 * ssz;
            if (same) std::memcpy(s.pin, src, ne * dsz);
            else {
                const double m = dst_dtype == FDX_F32 ? convert_to<float>(src_code, src, s.pin, ne) : convert_to<double>(src_code, src, s.pin, ne);
                mx[(size_t)t] = std::max(mx[(size_t)t], m);
            }
            if (hipMemcpyAsync(static_cast<char*>(dst_dev) + e0 * dsz, s.pin, ne * dsz, hipMemcpyHostToDevice, cs) != hipSuccess ||
                hipEventRecord(s.ev, cs) != hipSuccess) { rc = FDX_ERR_HIP; break; }
            s.busy = true;
        }
        (void)hipStreamSynchronize(cs);
        for (Slot& s : slots) {
            if (s.ev) (void)hipEventDestroy(s.ev);
            if (s.pin) pinned_buffer_put(s.pin, s.cap);
        }
        (void)hipStreamDestroy(cs);
    };
    // the destination may still be in use by work queued on the caller's stream (a recycled block): drain it first
    FDX_HIP(hipStreamSynchronize(st));
    std::vector<std::thread> team;
    for (int t = 1; t < T; ++t) team.emplace_back(work, t);
    work(0);
    for (auto& th : team) th.join();
    if (rc != 0) { (void)hipGetLastError(); return fail(FDX_ERR_HIP, "fdx_upload_convert_dev: staged upload failed"); }
    if (max_abs_out) *max_abs_out = *std::max_element(mx.begin(), mx.end());
    return 0;
}

// bytes from HBM to caller memory through the same ring (results: beta_ / proportions_ are 240 MB each at 1M x 30)
extern "C" int fdx_download_dev(void* dst_host, const void* src_dev, size_t bytes, void* stream) {
    if (bytes == 0) return 0;
    FDX_REQUIRE(dst_host && src_dev, "fdx_download_dev: null pointer");
    hipStream_t st = (hipStream_t)stream;
    FDX_HIP(hipStreamSynchronize(st));                               // the producer of src_dev
    const size_t n_chunks = (bytes + kChunkBytes - 1) / kChunkBytes;
    const int T = team_size(bytes, 4);
    int dev = 0;
    FDX_HIP(hipGetDevice(&dev));
    std::atomic<size_t> next{0};
    std::atomic<int> rc{0};
    auto work = [&]() {
        if (hipSetDevice(dev) != hipSuccess) { rc = FDX_ERR_HIP; return; }
        hipStream_t cs = nullptr;
        if (hipStreamCreateWithFlags(&cs, hipStreamNonBlocking) != hipSuccess) { rc = FDX_ERR_HIP; return; }
        Slot slots[kSlotsPerThread];
        size_t chunk_of[kSlotsPerThread] = {};
        for (Slot& s : slots) {
            s.pin = pinned_buffer_get(kChunkBytes, &s.cap);
            if (!s.pin || hipEventCreateWithFlags(&s.ev, hipEventDisableTiming) != hipSuccess) rc = FDX_ERR_HIP;
        }
        auto drain = [&](int k) {                                     // the slot's DMA has landed: copy it out
            Slot& s = slots[k];
            if (!s.busy) return;
            if (hipEventSynchronize(s.ev) != hipSuccess) { rc = FDX_ERR_HIP; return; }
            const size_t o = chunk_of[k] * kChunkBytes, nb = std::min(kChunkBytes, bytes - o);
            std::memcpy(static_cast<char*>(dst_host) + o, s.pin, nb);
            s.busy = false;
        };
        int k = 0;
        while (rc == 0) {
            const size_t c = next.fetch_add(1);
            if (c >= n_chunks) break;
            drain(k);                                                 // (the other slot's DMA runs meanwhile)
            Slot& s = slots[k];
            const size_t o = c * kChunkBytes, nb = std::min(kChunkBytes, bytes - o);
            if (hipMemcpyAsync(s.pin, static_cast<const char*>(src_dev) + o, nb, hipMemcpyDeviceToHost, cs) != hipSuccess ||
                hipEventRecord(s.ev, cs) != hipSuccess) { rc = FDX_ERR_HIP; break; }
            chunk_of[k] = c;
            s.busy = true;
            k = (k + 1) % kSlotsPerThread;
        }
        for (int j = 0; j < kSlotsPerThread; ++j) drain(j);
        (void)hipStreamSynchronize(cs);
        for (Slot& s : slots) {
            if (s.ev) (void)hipEventDestroy(s.ev);
            if (s.pin) pinned_buffer_put(s.pin, s.cap);
        }
        (void)hipStreamDestroy(cs);
    };
    std::vector<std::thread> team;
    for (int t = 1; t < T; ++t) team.emplace_back(work);
    work();
    for (auto& th : team) th.join();
    if (rc != 0) { (void)hipGetLastError(); return fail(FDX_ERR_HIP, "fdx_download_dev: staged download failed"); }
    return 0;
}

// The rate the box's link gives a PINNED buffer (one hipMemcpyAsync of `bytes`, H2D when to_device): the yardstick bench.py holds
// the host-array fit against.  Returns GB/s in *gbps_out.
extern "C" int fdx_pinned_copy_rate(size_t bytes, int32_t to_device, double* gbps_out) {
    FDX_REQUIRE(gbps_out != nullptr && bytes > 0, "fdx_pinned_copy_rate: bad arguments");
    void* pin = nullptr;
    void* devp = nullptr;
    FDX_HIP(hipHostMalloc(&pin, bytes, hipHostMallocDefault));
    if (hipMalloc(&devp, bytes) != hipSuccess) { (void)hipHostFree(pin); return fail(FDX_ERR_HIP, "fdx_pinned_copy_rate: hipMalloc"); }
    std::memset(pin, 1, bytes);
    hipEvent_t a = nullptr, b = nullptr;
    (void)hipEventCreate(&a);
    (void)hipEventCreate(&b);
    double best = 0.0;
    for (int rep = 0; rep < 3; ++rep) {
        (void)hipEventRecord(a, nullptr);
        if (to_device) (void)hipMemcpyAsync(devp, pin, bytes, hipMemcpyHostToDevice, nullptr);
        else (void)hipMemcpyAsync(pin, devp, bytes, hipMemcpyDeviceToHost, nullptr);
        (void)hipEventRecord(b, nullptr);
        (void)hipEventSynchronize(b);
        float ms = 0.f;
        (void)hipEventElapsedTime(&ms, a, b);
        if (ms > 0.f) best = std::max(best, (double)bytes / (ms * 1e-3) / 1e9);
    }
    (void)hipEventDestroy(a);
    (void)hipEventDestroy(b);
    (void)hipFree(devp);
    (void)hipHostFree(pin);
    *gbps_out = best;
    return 0;
}
