// Caching device allocator on kf_malloc (hipMalloc).
// Same contract as the reference's DeviceAllocator (src/core/include/device_allocator.h:42-82,
// device_allocator.cpp:37-72): singleton, 1 KiB granularity, eight size classes bounded at
// 4K/64K/256K/1M/4M/64M/256M/inf, best-fit reuse of a cached block of the same class, no splitting,
// memory is never returned to the driver. Rebuilt for multi-GPU MI355X nodes: pools are keyed by
// DEVICE (the reference's free lists ignore it — a block freed on device 0 can be handed to device 1,
// device_allocator.h:30-40), all state sits behind a mutex, and zero-initialised scratch is available
// for kernels that need it.
#pragma once

#include <cstddef>
#include <cstdint>
#include <map>
#include <memory>
#include <mutex>
#include <set>
#include <unordered_map>
#include <vector>

namespace utils {
namespace memory {

// owning handle to a cached block; releasing it returns the block to the pool
class DataPtr {
public:
    DataPtr() = default;
    DataPtr(void *p, size_t bytes, int device) : p_(p), bytes_(bytes), device_(device) {}
    DataPtr(const DataPtr &) = delete;
    DataPtr &operator=(const DataPtr &) = delete;
    DataPtr(DataPtr &&o) noexcept { steal(o); }
    DataPtr &operator=(DataPtr &&o) noexcept {
        if (this != &o) { clear(); steal(o); }
        return *this;
    }
    ~DataPtr() { clear(); }
    void *get() const { return p_; }
    size_t capacity() const { return bytes_; }
    int device() const { return device_; }
    explicit operator bool() const { return p_ != nullptr; }
    void clear();

private:
    void steal(DataPtr &o) { p_ = o.p_; bytes_ = o.bytes_; device_ = o.device_; o.p_ = nullptr; o.bytes_ = 0; }
    void *p_ = nullptr;
    size_t bytes_ = 0;
    int device_ = -1;
};

class DeviceAllocator {
public:
    static constexpr size_t kAlignment = 1024;
    static constexpr int kNumPools = 8;
    static DeviceAllocator *GetInstance();

    DataPtr allocate(size_t size_in_bytes, int device);
    void free(void *ptr);
    void print();

    // Capture mode (HIP graphs): an instantiated graph keeps reading and writing the addresses its kernels were recorded
    // with, on every replay. While a capture is open on `device`, every block freed there - and, afterwards, every block
    // that was allocated during the capture - goes to a pool PRIVATE to that graph instead of the shared cache, so no later
    // allocation can be handed memory a live graph still uses; allocations inside the capture may reuse that private pool
    // (stream order inside the graph is the recording order). release_graph() returns the pool to the shared cache.
    uint64_t begin_capture(int device);
    void end_capture(uint64_t graph_id);
    void release_graph(uint64_t graph_id);

    // Hands every idle cached block of `device` (owned by no live graph) back to the driver; returns the bytes released. allocate()
    // does this by itself, once, when the driver reports out-of-memory (KF_ERR_OOM), before it gives up with utils::OutOfMemory.
    size_t release_cached(int device);
    uint64_t oom_retries();               // how often allocate() took that path
    void debug_fail_above(size_t bytes);  // test hook (tests/test_gpu_host_api.py): driver allocations larger than this fail as out-of-memory; 0 = off

    struct Stats { size_t active_blocks, cached_blocks, active_bytes, cached_bytes, driver_allocs, graph_blocks, graph_bytes; };
    Stats stats(int device = -1);
    static int pool_index(size_t size);

private:
    DeviceAllocator() = default;
    struct Block { void *ptr; size_t size; int device; uint32_t id; bool in_use; uint64_t graph; };
    struct BySizeThenPtr {
        bool operator()(const Block *a, const Block *b) const {
            if (a->size != b->size) return a->size < b->size;
            return reinterpret_cast<uintptr_t>(a->ptr) < reinterpret_cast<uintptr_t>(b->ptr);
        }
    };
    using Pool = std::set<Block *, BySizeThenPtr>;
    std::mutex mu_;
    std::map<int, std::vector<Pool>> free_;            // device -> size class -> cached blocks
    std::unordered_map<void *, Block *> by_ptr_;       // every block ever allocated
    std::map<uint64_t, std::vector<Pool>> graph_free_; // live graph -> size class -> blocks only that graph may reuse
    uint64_t capturing_ = 0;                           // id of the open capture (0: none) ...
    int capture_device_ = -1;                          // ... and its device
    uint64_t next_graph_ = 0;
    uint32_t next_id_ = 0;
    size_t driver_allocs_ = 0;
    size_t fail_above_ = 0;
    uint64_t oom_retries_ = 0;
    size_t release_cached_locked(int device);
};

} // namespace memory
} // namespace utils
