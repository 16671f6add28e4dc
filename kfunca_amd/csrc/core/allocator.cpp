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
    auto &pools = free_[device];
    if (pools.empty()) pools.resize(kNumPools);
    Pool &pool = pools[pool_index(size)];
    Block key{nullptr, size, device, 0, false};
    auto it = pool.lower_bound(&key); // smallest cached block of this class that fits
    Block *b;
    if (it != pool.end()) {
        b = *it;
        pool.erase(it);
    } else {
        void *p = nullptr;
        DEV_CALL(kf_malloc(&p, rounded));
        b = new Block{p, rounded, device, next_id_++, false};
        by_ptr_[p] = b;
        ++driver_allocs_;
    }
    b->in_use = true;
    return DataPtr(b->ptr, b->size, device);
}

void DeviceAllocator::free(void *ptr) {
    std::lock_guard<std::mutex> lk(mu_);
    auto it = by_ptr_.find(ptr);
    if (it == by_ptr_.end() || !it->second->in_use) return;
    Block *b = it->second;
    b->in_use = false;
    auto &pools = free_[b->device];
    if (pools.empty()) pools.resize(kNumPools);
    pools[pool_index(b->size)].insert(b);
}

DeviceAllocator::Stats DeviceAllocator::stats(int device) {
    std::lock_guard<std::mutex> lk(mu_);
    Stats s{0, 0, 0, 0, driver_allocs_};
    for (auto &kv : by_ptr_) {
        const Block *b = kv.second;
        if (device >= 0 && b->device != device) continue;
        if (b->in_use) { ++s.active_blocks; s.active_bytes += b->size; }
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
