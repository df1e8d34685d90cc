// Caching device allocator behind DevBuf (see fdx_internal.h).
#include "fdx_env.h"
#include <algorithm>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <map>
#include <mutex>
#include <vector>

#include "fdx_internal.h"

namespace fdx {

namespace {
std::mutex g_mu;
// A cached block remembers the stream its last owner queued work on (the stream of the entry point that allocated it,
// PoolStream below).  Handing it to work on the SAME stream needs nothing - stream order puts the new user behind the old
// one.  Handing it to ANOTHER stream makes that stream wait for everything queued on the old one so far (an event recorded
// there at that moment: later than the free, so conservative, and free of charge on the common same-stream path).
struct Cached { void* p; hipStream_t last; };
struct Live { int dev; size_t cap; hipStream_t stream; };
std::map<std::pair<int, size_t>, std::vector<Cached>> g_free;   // (device, capacity) -> cached blocks
std::map<void*, Live> g_live;                                   // block -> owner
thread_local hipStream_t t_stream = nullptr;                    // stream of the entry point running on this thread
const hipStream_t kIdleStream = reinterpret_cast<hipStream_t>(~(uintptr_t)0);   // owner's work known complete: anyone may follow

size_t capacity_class(size_t n) {
    if (n <= (1u << 20)) {                 // small: next power of two, at least 256 B
        size_t c = 256;
        while (c < n) c <<= 1;
        return c;
    }
    const size_t gran = n <= (64u << 20) ? (1u << 20) : (16u << 20);   // large: 1 MiB / 16 MiB granules
    return (n + gran - 1) / gran * gran;
}

// `to` waits for the work queued on `from` up to now; the device is drained instead when `from` cannot take an event any
// more (a caller's stream that has been destroyed since)
void order_after(hipStream_t from, hipStream_t to) {
    thread_local hipEvent_t ev[64] = {};
    int dev = 0;
    bool ok = hipGetDevice(&dev) == hipSuccess && dev >= 0 && dev < 64;
    if (ok && !ev[dev]) ok = hipEventCreateWithFlags(&ev[dev], hipEventDisableTiming) == hipSuccess;
    if (ok) ok = hipEventRecord(ev[dev], from) == hipSuccess && hipStreamWaitEvent(to, ev[dev], 0) == hipSuccess;
    if (!ok) {
        (void)hipGetLastError();
        (void)hipDeviceSynchronize();
    }
}
}  // namespace

hipStream_t pool_set_stream(hipStream_t s) {
    const hipStream_t prev = t_stream;
    t_stream = s;
    return prev;
}

int pool_alloc(size_t bytes, void** p, size_t* cap) {
    int dev = 0;
    FDX_HIP(hipGetDevice(&dev));
    const size_t c = capacity_class(bytes);
    const hipStream_t mine = t_stream;
    {
        std::unique_lock<std::mutex> lk(g_mu);
        auto it = g_free.find({dev, c});
        if (it != g_free.end() && !it->second.empty()) {
            std::vector<Cached>& v = it->second;
            size_t pick = v.size() - 1;
            for (size_t i = v.size(); i-- > 0;)                  // a block last used on this stream (or known idle), if there is one
                if (v[i].last == mine || v[i].last == kIdleStream) { pick = i; break; }
            const Cached b = v[pick];
            v.erase(v.begin() + (long)pick);
            *p = b.p;
            *cap = c;
            g_live[*p] = Live{dev, c, mine};
            lk.unlock();
            if (b.last != mine && b.last != kIdleStream && !fdx::exp_env("FDX_POOL_NO_ORDER")) order_after(b.last, mine);
            return 0;
        }
    }
    static const bool trace = fdx::exp_env("FDX_POOL_TRACE") != nullptr;   // diagnostic: every miss of the cache, with what the driver took for it
    const auto t_miss = std::chrono::steady_clock::now();
    hipError_t e = hipMalloc(p, c);
    if (trace)
        std::fprintf(stderr, "[fdx-pool] hipMalloc(%zu MB) %.2f ms\n", c >> 20,
                     std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t_miss).count());
    if (e != hipSuccess) {     // out of memory: drop the cache and retry once
        (void)hipGetLastError();
        pool_trim();
        e = hipMalloc(p, c);
    }
    if (e != hipSuccess) {
        *p = nullptr;
        return fail(FDX_ERR_HIP, std::string("hipMalloc(") + std::to_string(c) + "): " + hipGetErrorString(e));
    }
    *cap = c;
    std::lock_guard<std::mutex> lk(g_mu);
    g_live[*p] = Live{dev, c, mine};
    return 0;
}

void pool_mark_idle(void* p) {
    if (!p) return;
    std::lock_guard<std::mutex> lk(g_mu);
    auto it = g_live.find(p);
    if (it != g_live.end()) it->second.stream = kIdleStream;
}

void pool_free(void* p, size_t /*cap*/) {
    if (!p) return;
    std::lock_guard<std::mutex> lk(g_mu);
    auto it = g_live.find(p);
    if (it == g_live.end()) return;    // not ours (or already returned)
    g_free[{it->second.dev, it->second.cap}].push_back(Cached{p, it->second.stream});
    g_live.erase(it);
}

// 64-byte pinned blocks for objects that outlive a call (a graph's deferred counts): hipHostMalloc / hipHostFree cost hundreds of
// microseconds each, so the blocks are recycled through a process-wide list and never returned.
static std::mutex g_pin_mu;
static std::vector<void*> g_pin_free;
void* pinned_block_get() {
    {
        std::lock_guard<std::mutex> lk(g_pin_mu);
        if (!g_pin_free.empty()) { void* p = g_pin_free.back(); g_pin_free.pop_back(); return p; }
    }
    void* p = nullptr;
    if (hipHostMalloc(&p, FDX_PINNED_BLOCK_BYTES, hipHostMallocDefault) != hipSuccess) { (void)hipGetLastError(); return nullptr; }
    return p;
}
void pinned_block_put(void* p) {
    if (!p) return;
    std::lock_guard<std::mutex> lk(g_pin_mu);
    g_pin_free.push_back(p);
}

// Recycled pinned buffers of any size (capacity classes: powers of two from 4 KB) for objects that outlive a call and read back
// more than a block - a leverage job's scores.  Never returned to the driver.
static std::map<size_t, std::vector<void*>> g_pinbuf_free;
void* pinned_buffer_get(size_t bytes, size_t* cap_out) {
    size_t cap = 4096;
    while (cap < bytes) cap <<= 1;
    if (cap_out) *cap_out = cap;
    {
        std::lock_guard<std::mutex> lk(g_pin_mu);
        auto& v = g_pinbuf_free[cap];
        if (!v.empty()) { void* p = v.back(); v.pop_back(); return p; }
    }
    void* p = nullptr;
    if (hipHostMalloc(&p, cap, hipHostMallocDefault) != hipSuccess) { (void)hipGetLastError(); return nullptr; }
    return p;
}
void pinned_buffer_put(void* p, size_t cap) {
    if (!p) return;
    std::lock_guard<std::mutex> lk(g_pin_mu);
    g_pinbuf_free[cap].push_back(p);
}

// ---- transfers between the device and CALLER (pageable) memory --------------------------------------------------------------------
// hipMemcpy on pageable memory lets the driver pin the caller's pages for the DMA; when such pages are unmapped later (a collected
// numpy array, a freed std::vector above malloc's mmap threshold) the kernel's MMU notifier evicts the process's GPU queues and the
// next launch waits 10-25 ms for their restore (measured, round 5: the lattice and CSR families' "stalls after a host phase").
// Copies of 64 KB to 32 MB therefore go through recycled pinned buffers: the caller's memory is only ever touched by memcpy.
// (Smaller ones live on the heap, which is not unmapped; larger ones - a result matrix - keep the direct path: a staging copy of
// hundreds of megabytes costs more than the stall it avoids.)
namespace {
constexpr size_t kStageMin = 64 * 1024, kStageMax = 32u << 20;
struct PendingPin { void* p; size_t cap; hipEvent_t ev; };
std::vector<PendingPin> g_pin_pending;        // uploads still in flight: released when their event has completed (g_pin_mu)
void reap_pending_pins() {
    std::vector<PendingPin> done;
    {
        std::lock_guard<std::mutex> lk(g_pin_mu);
        for (size_t i = 0; i < g_pin_pending.size();) {
            if (hipEventQuery(g_pin_pending[i].ev) != hipErrorNotReady) {
                done.push_back(g_pin_pending[i]);
                g_pin_pending[i] = g_pin_pending.back();
                g_pin_pending.pop_back();
            } else {
                ++i;
            }
        }
    }
    for (const PendingPin& d : done) {
        (void)hipEventDestroy(d.ev);
        pinned_buffer_put(d.p, d.cap);
    }
}
}  // namespace

int copy_h2d(void* dst_dev, const void* src_host, size_t bytes, hipStream_t st) {
    if (bytes == 0) return 0;
    if (bytes < kStageMin || bytes > kStageMax || fdx::exp_env("FDX_NO_STAGED_COPIES")) {
        FDX_HIP(hipMemcpyAsync(dst_dev, src_host, bytes, hipMemcpyHostToDevice, st));
        return 0;
    }
    reap_pending_pins();
    size_t cap = 0;
    void* pin = pinned_buffer_get(bytes, &cap);
    if (!pin) return fail(FDX_ERR_HIP, "copy_h2d: pinned host buffer");
    std::memcpy(pin, src_host, bytes);
    hipEvent_t ev = nullptr;
    hipError_t e = hipMemcpyAsync(dst_dev, pin, bytes, hipMemcpyHostToDevice, st);
    if (e == hipSuccess) e = hipEventCreateWithFlags(&ev, hipEventDisableTiming);
    if (e == hipSuccess) e = hipEventRecord(ev, st);
    if (e != hipSuccess) {                                   // nothing may still read the buffer when it goes back
        (void)hipStreamSynchronize(st);
        if (ev) (void)hipEventDestroy(ev);
        pinned_buffer_put(pin, cap);
        return fail(FDX_ERR_HIP, std::string("copy_h2d: ") + hipGetErrorString(e));
    }
    std::lock_guard<std::mutex> lk(g_pin_mu);
    g_pin_pending.push_back(PendingPin{pin, cap, ev});
    return 0;
}

int copy_d2h(void* dst_host, const void* src_dev, size_t bytes, hipStream_t st) {
    if (bytes == 0) return 0;
    if (bytes < kStageMin || bytes > kStageMax || fdx::exp_env("FDX_NO_STAGED_COPIES")) {
        FDX_HIP(hipMemcpyAsync(dst_host, src_dev, bytes, hipMemcpyDeviceToHost, st));
        FDX_HIP(hipStreamSynchronize(st));
        return 0;
    }
    size_t cap = 0;
    void* pin = pinned_buffer_get(bytes, &cap);
    if (!pin) return fail(FDX_ERR_HIP, "copy_d2h: pinned host buffer");
    hipError_t e = hipMemcpyAsync(pin, src_dev, bytes, hipMemcpyDeviceToHost, st);
    if (e == hipSuccess) e = hipStreamSynchronize(st);
    if (e == hipSuccess) std::memcpy(dst_host, pin, bytes);
    else (void)hipStreamSynchronize(st);
    pinned_buffer_put(pin, cap);
    if (e != hipSuccess) return fail(FDX_ERR_HIP, std::string("copy_d2h: ") + hipGetErrorString(e));
    return 0;
}

void* pinned_scratch(int slot, size_t bytes) {
    static const bool pageable = fdx::exp_env("FDX_PAGEABLE_READBACK") != nullptr;   // diagnostic: what the copies cost without pinning
    struct Slot {
        void* p = nullptr; size_t cap = 0; bool pinned = true;
        void drop() { if (p) { if (pinned) (void)hipHostFree(p); else free(p); } p = nullptr; cap = 0; }
        ~Slot() { drop(); }
    };
    static thread_local Slot slots[8];
    if (slot < 0 || slot >= 8) return nullptr;
    Slot& s = slots[slot];
    if (s.cap < bytes) {
        s.drop();
        const size_t want = std::max<size_t>(bytes, 4096);
        s.pinned = !pageable;
        if (pageable) s.p = malloc(want);
        else if (hipHostMalloc(&s.p, want, hipHostMallocDefault) != hipSuccess) { (void)hipGetLastError(); s.p = nullptr; }
        if (!s.p) return nullptr;
        s.cap = want;
    }
    return s.p;
}

void pool_trim() {
    std::lock_guard<std::mutex> lk(g_mu);
    for (auto& kv : g_free) {
        for (const Cached& q : kv.second) (void)hipFree(q.p);      // hipFree waits for the device: nothing is in flight after it
        kv.second.clear();
    }
}

}  // namespace fdx
