#include "allocator.h"

#include <algorithm>
#include <iostream>
#include <limits>

#include "device_api.h"

namespace utils {
namespace memory {

namespace {
const size_t kBounds[DeviceAllocator::kNumPools] = {4u << 10, 64u << 10, 256u << 10, 1u << 20, 4u << 20,
                                                    64u << 20, 256u << 20, std::numeric_limits<size_t>::max()};
}

void DataPtr::clear() {
    if (p_) DeviceAllocator::GetInstance()->free(p_);
    p_ = nullptr;
    bytes_ = 0;
}

DeviceAllocator *DeviceAllocator::GetInstance() {
    static DeviceAllocator *inst = new DeviceAllocator(); // leaked on purpose: tensors may outlive static teardown
    return inst;
}

int DeviceAllocator::pool_index(size_t size) {
    return static_cast<int>(std::lower_bound(kBounds, kBounds + kNumPools, size) - kBounds);
}

DataPtr DeviceAllocator::allocate(size_t size, int device) {
    dev::set_device(device);
    const size_t rounded = std::max<size_t>(kAlignment, (size + kAlignment - 1) / kAlignment * kAlignment);
    std::lock_guard<std::mutex> lk(mu_);
    const bool capturing = capturing_ != 0 && capture_device_ == device;
    Block key{nullptr, size, device, 0, false, 0};
    Block *b = nullptr;
    if (capturing) { // first the blocks this capture has already retired
        auto &gp = graph_free_[capturing_];
        if (gp.empty()) gp.resize(kNumPools);
        Pool &pool = gp[pool_index(size)];
        auto it = pool.lower_bound(&key);
        if (it != pool.end()) { b = *it; pool.erase(it); }
    }
    if (!b) {
        auto &pools = free_[device];
        if (pools.empty()) pools.resize(kNumPools);
        Pool &pool = pools[pool_index(size)];
        auto it = pool.lower_bound(&key); // smallest cached block of this class that fits
        if (it != pool.end()) {
            b = *it;
            pool.erase(it);
        } else {
            void *p = nullptr;
            int st = (fail_above_ && rounded > fail_above_) ? (int)KF_ERR_OOM : kf_malloc(&p, rounded);
            if (st == KF_ERR_OOM) {
                // the driver is out of room while this cache may sit on gigabytes nobody uses: hand every idle block of the device back
                // (the reference never returns memory: device_allocator.cpp:37-72, `dfree` is never called) and try once more
                ++oom_retries_;
                if (capturing_ == 0) release_cached_locked(device); // (hipFree is not a capturable operation)
                st = (fail_above_ && rounded > fail_above_) ? (int)KF_ERR_OOM : kf_malloc(&p, rounded);
            }
            if (st == KF_ERR_OOM)
                throw utils::OutOfMemory(utils::concat("[device error in kf_malloc, status ", st, "] "),
                                         utils::concat("out of device memory: ", rounded, " bytes requested on device ", device));
            dev::check(st, "kf_malloc(&p, rounded)");
            b = new Block{p, rounded, device, next_id_++, false, 0};
            by_ptr_[p] = b;
            ++driver_allocs_;
        }
    }
    b->in_use = true;
    if (capturing) b->graph = capturing_; // recorded kernels use this address: it stays with the graph until release_graph
    return DataPtr(b->ptr, b->size, device);
}

void DeviceAllocator::free(void *ptr) {
    std::lock_guard<std::mutex> lk(mu_);
    auto it = by_ptr_.find(ptr);
    if (it == by_ptr_.end() || !it->second->in_use) return;
    Block *b = it->second;
    b->in_use = false;
    // freed while recording: the graph being captured may still touch it. A block that already belongs to ANOTHER live graph (say an
    // output that graph's replays still write) keeps its owner: re-tagging it would hand it back to the shared cache when the
    // capturing graph is destroyed, while the first graph can still be replayed.
    if (capturing_ != 0 && capture_device_ == b->device && !(b->graph != 0 && b->graph != capturing_ && graph_free_.count(b->graph)))
        b->graph = capturing_;
    if (b->graph != 0) {
        auto g = graph_free_.find(b->graph);
        if (g != graph_free_.end()) {
            if (g->second.empty()) g->second.resize(kNumPools);
            g->second[pool_index(b->size)].insert(b);
            return;
        }
        b->graph = 0; // that graph is gone
    }
    auto &pools = free_[b->device];
    if (pools.empty()) pools.resize(kNumPools);
    pools[pool_index(b->size)].insert(b);
}

size_t DeviceAllocator::release_cached_locked(int device) {
    size_t freed = 0;
    auto dv = free_.find(device);
    if (dv == free_.end()) return 0;
    for (Pool &pool : dv->second) {
        for (auto it = pool.begin(); it != pool.end();) { // idle, owned by no graph (graph-held blocks live in graph_free_)
            Block *b = *it;
            if (kf_free(b->ptr) != KF_OK) { // not freed: it stays a cached block (still reusable, still known by its pointer)
                ++it;
                continue;
            }
            freed += b->size;
            by_ptr_.erase(b->ptr);
            it = pool.erase(it);
            delete b;
        }
    }
    return freed;
}

size_t DeviceAllocator::release_cached(int device) {
    {
        // a device synchronise + hipFree inside a stream capture would invalidate the capture (allocate() never gets here while capturing)
        std::lock_guard<std::mutex> lk(mu_);
        if (capturing_ != 0) return 0;
    }
    dev::set_device(device);
    dev::synchronize(device); // nothing queued may still touch a block that was freed to the cache a moment ago
    std::lock_guard<std::mutex> lk(mu_);
    return release_cached_locked(device);
}

void DeviceAllocator::debug_fail_above(size_t bytes) {
    std::lock_guard<std::mutex> lk(mu_);
    fail_above_ = bytes;
}

uint64_t DeviceAllocator::oom_retries() {
    std::lock_guard<std::mutex> lk(mu_);
    return oom_retries_;
}

uint64_t DeviceAllocator::begin_capture(int device) {
    std::lock_guard<std::mutex> lk(mu_);
    CHECK_FAIL(capturing_ == 0, "a graph capture is already open on device ", capture_device_);
    capturing_ = ++next_graph_;
    capture_device_ = device;
    graph_free_[capturing_].resize(kNumPools);
    return capturing_;
}

void DeviceAllocator::end_capture(uint64_t graph_id) {
    std::lock_guard<std::mutex> lk(mu_);
    if (capturing_ == graph_id) { capturing_ = 0; capture_device_ = -1; }
}

void DeviceAllocator::release_graph(uint64_t graph_id) {
    std::lock_guard<std::mutex> lk(mu_);
    if (capturing_ == graph_id) { capturing_ = 0; capture_device_ = -1; }
    auto g = graph_free_.find(graph_id);
    if (g == graph_free_.end()) return;
    for (Pool &pool : g->second)
        for (Block *b : pool) {
            b->graph = 0;
            auto &pools = free_[b->device];
            if (pools.empty()) pools.resize(kNumPools);
            pools[pool_index(b->size)].insert(b);
        }
    graph_free_.erase(g);
    for (auto &kv : by_ptr_)
        if (kv.second->graph == graph_id) kv.second->graph = 0; // still alive: back to the shared cache when freed
}

DeviceAllocator::Stats DeviceAllocator::stats(int device) {
    std::lock_guard<std::mutex> lk(mu_);
    Stats s{0, 0, 0, 0, driver_allocs_, 0, 0};
    for (auto &kv : by_ptr_) {
        const Block *b = kv.second;
        if (device >= 0 && b->device != device) continue;
        if (b->in_use) { ++s.active_blocks; s.active_bytes += b->size; }
        else if (b->graph != 0 && graph_free_.count(b->graph)) { ++s.graph_blocks; s.graph_bytes += b->size; } // held for a live graph
        else { ++s.cached_blocks; s.cached_bytes += b->size; }
    }
    return s;
}

void DeviceAllocator::print() { // kfunca.memstat() (reference register.cpp:61-63)
    std::lock_guard<std::mutex> lk(mu_);
    for (auto &dv : free_) {
        std::cout << "Device " << dv.first << " unused blocks:\n";
        size_t lo = 0;
        for (int p = 0; p < kNumPools; ++p) {
            std::cout << "[" << lo << ", " << kBounds[p] << "):";
            for (const Block *b : dv.second[p]) std::cout << b->id << ":" << b->size << ":" << b->ptr << ", ";
            std::cout << "\n";
            lo = kBounds[p];
        }
    }
    std::cout << "Active blocks:\n";
    for (auto &kv : by_ptr_)
        if (kv.second->in_use) std::cout << kv.second->id << ":" << kv.second->size << ":" << kv.second->ptr << "@" << kv.second->device << ", ";
    std::cout << std::endl;
}

} // namespace memory
} // namespace utils
