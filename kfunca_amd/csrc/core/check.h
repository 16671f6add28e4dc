// Error convention of the operator API: CHECK_FAIL(cond, msg...) throws utils::Error whose text keeps
// the reference's shape — "[enforce fail at file:line:func] Expected <cond> to be true, but got false. <msg>"
// (reference: src/core/utils/exception.h:36-56,123-131). pybind11 turns it into RuntimeError.
#pragma once

#include <cstdint>
#include <exception>
#include <sstream>
#include <string>

namespace utils {

class Error : public std::exception {
public:
    Error(std::string where, std::string what) : where_(std::move(where)), msg_(std::move(what)) {
        text_ = where_ + msg_ + "\n";
    }
    const char *what() const noexcept override { return text_.c_str(); }
    const std::string &msg() const { return msg_; }

private:
    std::string where_, msg_, text_;
};

// the device has no room for an allocation (KF_ERR_OOM): the one device error a caller can sensibly recover from
class OutOfMemory : public Error {
public:
    using Error::Error;
};

template <typename... Args>
inline std::string concat(const Args &...args) {
    std::ostringstream os;
    (void)std::initializer_list<int>{((os << args), 0)...};
    return os.str();
}

[[noreturn]] inline void raise_check(const char *func, const char *file, uint32_t line, const char *cond, const std::string &msg) {
    throw Error(concat("[enforce fail at ", file, ":", line, ":", func, "] "),
                concat("Expected ", cond, " to be true, but got false. ", msg));
}

} // namespace utils

#define CHECK_FAIL(cond, ...)                                                                        \
    do {                                                                                             \
        if (__builtin_expect(!(cond), 0))                                                            \
            ::utils::raise_check(__func__, __FILE__, static_cast<uint32_t>(__LINE__), #cond,        \
                                 ::utils::concat(__VA_ARGS__));                                      \
    } while (0)
