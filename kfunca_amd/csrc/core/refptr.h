// Intrusive reference counting for TensorImpl / TensorStorage / GradFunction.
// Observable behaviour follows the reference (src/core/utils/memory/intrusive_ptr.h): ref_count()
// is what Python sees as storage_ref_count()/impl_ref_count(), unsafe_set_ptr() adopts a raw pointer
// with an increment, move-construction steals, and move-ASSIGNMENT shares like a copy (asserted by
// the reference's test/core/test_intrusive_ptr.cpp:60-77).
#pragma once

#include <atomic>
#include <cstddef>

namespace utils {
namespace memory {

class intrusive_ptr_target {
public:
    intrusive_ptr_target() noexcept : refs_(0) {}
    intrusive_ptr_target(const intrusive_ptr_target &) noexcept : refs_(0) {}
    intrusive_ptr_target &operator=(const intrusive_ptr_target &) noexcept { return *this; }
    virtual ~intrusive_ptr_target() = default;
    size_t use_count() const noexcept { return refs_.load(std::memory_order_acquire); }
    void retain() const noexcept { refs_.fetch_add(1, std::memory_order_acq_rel); }
    bool release() const noexcept { return refs_.fetch_sub(1, std::memory_order_acq_rel) == 1; } // true: last owner

private:
    mutable std::atomic<size_t> refs_;
};

template <typename T>
class intrusive_ptr {
public:
    intrusive_ptr() noexcept : p_(nullptr) {}
    intrusive_ptr(T *p) noexcept : p_(p) { if (p_) p_->retain(); }
    intrusive_ptr(const intrusive_ptr &o) noexcept : p_(o.p_) { if (p_) p_->retain(); }
    intrusive_ptr(intrusive_ptr &&o) noexcept : p_(o.p_) { o.p_ = nullptr; }
    ~intrusive_ptr() { drop(); }
    intrusive_ptr &operator=(const intrusive_ptr &o) noexcept { share(o.p_); return *this; }
    intrusive_ptr &operator=(intrusive_ptr &&o) noexcept { share(o.p_); return *this; } // shares: see header note
    T *get() const noexcept { return p_; }
    T *operator->() const noexcept { return p_; }
    explicit operator bool() const noexcept { return p_ != nullptr; }
    size_t ref_count() const noexcept { return p_ ? p_->use_count() : 0; }
    void unsafe_set_ptr(T *p) noexcept { share(p); }
    void reset() noexcept { drop(); }

private:
    void share(T *p) noexcept {
        if (p == p_) return;
        if (p) p->retain();
        drop();
        p_ = p;
    }
    void drop() noexcept {
        if (p_ && p_->release()) delete p_;
        p_ = nullptr;
    }
    T *p_;
};

} // namespace memory
} // namespace utils
