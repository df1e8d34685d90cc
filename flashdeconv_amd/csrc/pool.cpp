// Caching device allocator behind DevBuf (see fdx_internal.h).
#include <algorithm>
#include <cstdlib>
#include <map>
#include <mutex>
#include <vector>

#include "fdx_internal.h"

namespace fdx {

namespace {
std::mutex g_mu;
std::map<std::pair<int, size_t>, std::vector<void*>> g_free;   // (device, capacity) -> cached blocks
std::map<void*, std::pair<int, size_t>> g_live;                 // block -> (device, capacity)

size_t capacity_class(size_t n) {
    if (n <= (1u << 20)) {                 // small: next power of two, at least 256 B
        size_t c = 256;
        while (c < n) c <<= 1;
        return c;
    }
    const size_t gran = n <= (64u << 20) ? (1u << 20) : (16u << 20);   // large: 1 MiB / 16 MiB granules
    return (n + gran - 1) / gran * gran;
}
}  // namespace

int pool_alloc(size_t bytes, void** p, size_t* cap) {
    int dev = 0;
    FDX_HIP(hipGetDevice(&dev));
    const size_t c = capacity_class(bytes);
    {
        std::lock_guard<std::mutex> lk(g_mu);
        auto it = g_free.find({dev, c});
        if (it != g_free.end() && !it->second.empty()) {
            *p = it->second.back();
            it->second.pop_back();
            *cap = c;
            g_live[*p] = {dev, c};
            return 0;
        }
    }
    hipError_t e = hipMalloc(p, c);
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
    g_live[*p] = {dev, c};
    return 0;
}

void pool_free(void* p, size_t /*cap*/) {
    if (!p) return;
    std::lock_guard<std::mutex> lk(g_mu);
    auto it = g_live.find(p);
    if (it == g_live.end()) return;    // not ours (or already returned)
    g_free[it->second].push_back(p);
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
    if (hipHostMalloc(&p, 64, hipHostMallocDefault) != hipSuccess) { (void)hipGetLastError(); return nullptr; }
    return p;
}
void pinned_block_put(void* p) {
    if (!p) return;
    std::lock_guard<std::mutex> lk(g_pin_mu);
    g_pin_free.push_back(p);
}

void* pinned_scratch(int slot, size_t bytes) {
    static const bool pageable = getenv("FDX_PAGEABLE_READBACK") != nullptr;   // diagnostic: what the copies cost without pinning
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
        for (void* q : kv.second) (void)hipFree(q);
        kv.second.clear();
    }
}

}  // namespace fdx
