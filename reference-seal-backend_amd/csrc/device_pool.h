// device_pool.h — per-context device (HBM) memory pool.
//
// Why: the timed region of a HEBench run is operate() (/root/reference/src/benchmarks/ckks/seal_ckks_element_wise_benchmark.cpp:306-366;
// Latency = one op per call, :138-141).  The reference's operate() takes its result and temporaries from SEAL's thread-local memory
// pools (MemoryPoolHandle::ThreadLocal(), :343) -- no system allocation per call.  A hipMalloc / hipFree pair costs tens to hundreds of
// microseconds and hipFree drains the device; at batch 1 (0.3 ms of kernels) that is the same order as the work.  So result slabs and
// temporaries come from size-class free lists over hipMalloc'd blocks, and a release is a push onto a list: no HIP call, no
// synchronisation.
//
// Why a release needs no synchronisation: every kernel and copy of a context is issued on its stream_ (or on stream2_ between a
// fork event recorded on stream_ and a join event stream_ waits for), so work that touches a re-issued block is stream-ordered after
// everything that touched it before it was released.  Host transfers (he355_upload / he355_download) run on the same stream and
// wait for it.  Blocks never move between contexts.
//
// Every raw hipMalloc / hipFree of a DeviceContext goes through raw_malloc / raw_free so that he355_alloc_stats can prove a
// steady-state operate() performs none (tests/test_api_bridge_gpu.py::test_steady_state_operate_does_not_allocate).
#pragma once
#include <hip/hip_runtime.h>

#include <atomic>
#include <cstdint>
#include <map>
#include <mutex>
#include <stdexcept>
#include <string>
#include <unordered_map>
#include <unordered_set>
#include <vector>

namespace he355 {

struct DeviceError : std::runtime_error { // any HIP failure: reported as HE355_E_DEVICE
    using std::runtime_error::runtime_error;
};
struct OutOfDeviceMemory : DeviceError {
    using DeviceError::DeviceError;
};

// process-wide totals over every pool (he355_alloc_stats with a null context: what a harness-level test reads, since the
// benchmark objects behind the API-Bridge handles own their contexts)
struct PoolTotals {
    std::atomic<uint64_t> raw_mallocs{0}, raw_frees{0}, pool_hits{0}, pool_misses{0};
};
inline PoolTotals g_pool_totals;

class DevicePool {
public:
    struct Stats {
        uint64_t raw_mallocs = 0, raw_frees = 0; // hipMalloc / hipFree calls since the context was created
        uint64_t pool_hits = 0, pool_misses = 0;  // allocations served from a free list / by a new block
        uint64_t cached_bytes = 0, live_bytes = 0; // bytes parked in free lists / handed out
    };
    DevicePool() = default;
    DevicePool(const DevicePool &) = delete;
    DevicePool &operator=(const DevicePool &) = delete;

    // size classes: multiples of 256 B below 64 KiB, of 64 KiB below 16 MiB, of 2 MiB above (waste < 1/8 from 16 MiB up, and the
    // slabs a benchmark asks for repeat exactly from call to call, which is what makes the lists hit)
    static size_t size_class(size_t bytes)
    {
        if (bytes == 0) bytes = 8;
        const size_t g = bytes < (64u << 10) ? 256 : bytes < (16u << 20) ? (64u << 10) : (2u << 20);
        return (bytes + g - 1) / g * g;
    }
    // counted hipMalloc; when the device is out of memory the cached blocks are given back and the call is tried once more
    void *raw_malloc(size_t bytes)
    {
        void *p = nullptr;
        hipError_t e = hipMalloc(&p, bytes ? bytes : 8);
        if (e == hipErrorOutOfMemory || e == hipErrorMemoryAllocation) {
            (void)hipGetLastError();
            if (trim() > 0) e = hipMalloc(&p, bytes ? bytes : 8);
        }
        if (e == hipErrorOutOfMemory || e == hipErrorMemoryAllocation) {
            (void)hipGetLastError();
            throw OutOfDeviceMemory("HIP error: out of device memory allocating " + std::to_string(bytes) + " bytes");
        }
        if (e != hipSuccess) throw DeviceError(std::string("HIP error: ") + hipGetErrorString(e) + " in hipMalloc");
        std::lock_guard<std::mutex> g(mu_);
        ++st_.raw_mallocs;
        ++g_pool_totals.raw_mallocs;
        return p;
    }
    void raw_free(void *p)
    {
        if (!p) return;
        (void)hipFree(p);
        std::lock_guard<std::mutex> g(mu_);
        ++st_.raw_frees;
        ++g_pool_totals.raw_frees;
    }
    void *alloc(size_t bytes)
    {
        const size_t cls = size_class(bytes);
        {
            std::lock_guard<std::mutex> g(mu_);
            auto it = free_.find(cls);
            if (it != free_.end() && !it->second.empty()) {
                void *p = it->second.back();
                it->second.pop_back();
                cached_.erase(p);
                st_.cached_bytes -= cls;
                st_.live_bytes += cls;
                ++st_.pool_hits;
                ++g_pool_totals.pool_hits;
                live_[p] = cls;
                return p;
            }
        }
        void *p = raw_malloc(cls);
        std::lock_guard<std::mutex> g(mu_);
        ++st_.pool_misses;
        ++g_pool_totals.pool_misses;
        st_.live_bytes += cls;
        live_[p] = cls;
        return p;
    }
    // kReleased: the block was handed out by this pool and is back on its list.  kCached: it already IS on a list -- a second free of
    // the same pointer (it must not be hipFree'd: the list still holds it and would hand freed memory to the next allocation).
    // kUnknown: never issued by this pool (another context's block, or not a device allocation of this library at all).
    enum Release { kReleased, kCached, kUnknown };
    Release release(void *p)
    {
        if (!p) return kReleased;
        std::lock_guard<std::mutex> g(mu_);
        auto it = live_.find(p);
        if (it == live_.end()) return cached_.count(p) ? kCached : kUnknown;
        const size_t cls = it->second;
        live_.erase(it);
        st_.live_bytes -= cls;
        st_.cached_bytes += cls;
        free_[cls].push_back(p);
        cached_.insert(p);
        return kReleased;
    }
    // hipFree every cached block; returns the bytes given back.  hipFree waits for the device, so work in flight on a cached block
    // (released with kernels still queued behind it) has drained before the memory goes.
    size_t trim()
    {
        std::vector<void *> blocks;
        size_t bytes = 0;
        {
            std::lock_guard<std::mutex> g(mu_);
            for (auto &kv : free_) {
                for (void *p : kv.second) blocks.push_back(p);
                kv.second.clear();
            }
            cached_.clear();
            bytes = st_.cached_bytes;
            st_.cached_bytes = 0;
            st_.raw_frees += blocks.size();
            g_pool_totals.raw_frees += blocks.size();
        }
        for (void *p : blocks) (void)hipFree(p);
        return bytes;
    }
    // context teardown: cached and still-live blocks alike
    void destroy()
    {
        trim();
        std::vector<void *> blocks;
        {
            std::lock_guard<std::mutex> g(mu_);
            for (auto &kv : live_) blocks.push_back(kv.first);
            live_.clear();
            st_.live_bytes = 0;
            st_.raw_frees += blocks.size();
            g_pool_totals.raw_frees += blocks.size();
        }
        for (void *p : blocks) (void)hipFree(p);
    }
    Stats stats() const
    {
        std::lock_guard<std::mutex> g(mu_);
        return st_;
    }
    size_t cached_bytes() const { return stats().cached_bytes; }

private:
    mutable std::mutex mu_;
    std::map<size_t, std::vector<void *>> free_;
    std::unordered_map<void *, size_t> live_;
    std::unordered_set<void *> cached_; // the blocks on the free lists (release() tells a second free from a foreign pointer)
    Stats st_;
};

} // namespace he355
