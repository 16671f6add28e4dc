// ScalarType and its rules. Enum order and names are the reference's (src/core/include/scalar_type.h:9-27)
// because they are API: the Python `dtype` enum exports them and the C ABI's KF_* codes equal them.
#pragma once

#include <cmath>
#include <cstdint>
#include <cstring>
#include <ostream>

#include "check.h"

enum class ScalarType : int8_t { Bool = 0, Byte, Char, Short, Int, Long, Half, BFloat16, Float, Double, Undefined, NumOptions };

inline const char *to_string(ScalarType t) {
    static const char *names[] = {"Bool", "Byte", "Char", "Short", "Int", "Long", "Half", "BFloat16", "Float", "Double"};
    const int i = static_cast<int>(t);
    return (i >= 0 && i < 10) ? names[i] : "UNKNOWN_SCALAR";
}
inline std::ostream &operator<<(std::ostream &os, ScalarType t) { return os << to_string(t); }

inline size_t element_size(ScalarType t) {
    switch (t) {
    case ScalarType::Bool: case ScalarType::Byte: case ScalarType::Char: return 1;
    case ScalarType::Short: case ScalarType::Half: case ScalarType::BFloat16: return 2;
    case ScalarType::Int: case ScalarType::Float: return 4;
    case ScalarType::Long: case ScalarType::Double: return 8;
    default: CHECK_FAIL(false, "Unknown ScalarType");
    }
    return 0;
}

inline bool is_floating_type(ScalarType t) {
    return t == ScalarType::Double || t == ScalarType::Float || t == ScalarType::Half || t == ScalarType::BFloat16;
}
inline bool is_unsigned_int_type(ScalarType t) { return t == ScalarType::Byte || t == ScalarType::Bool; }

// dtype promotion of a binary op (reference: tensor_iterator.cpp:32-44): floats beat ints, the wider
// float wins (Half + BFloat16 -> BFloat16 by enum order), signed beats unsigned, else the wider.
inline ScalarType promote_types(ScalarType a, ScalarType b) {
    const bool fa = is_floating_type(a), fb = is_floating_type(b);
    if (fa != fb) return fa ? a : b;
    if (!fa) {
        const bool ua = is_unsigned_int_type(a), ub = is_unsigned_int_type(b);
        if (ua != ub) return ua ? b : a;
    }
    return a >= b ? a : b;
}

// accumulate dtype of the floating family (reference accumulate_type.h:29-42)
inline ScalarType accumulate_type(ScalarType t) {
    switch (t) {
    case ScalarType::Half: case ScalarType::BFloat16: case ScalarType::Float: return ScalarType::Float;
    case ScalarType::Double: return ScalarType::Double;
    default: return ScalarType::Undefined;
    }
}

namespace dtype {
// host-side 16-bit float conversions (item(), printing): bf16 is the upper half of an f32 with
// round-to-nearest-even, NaN -> 0x7FC0 (reference half.h:195-208); f16 is IEEE binary16.
inline float bf16_bits_to_float(uint16_t b) {
    const uint32_t u = static_cast<uint32_t>(b) << 16;
    float f;
    std::memcpy(&f, &u, 4);
    return f;
}
inline uint16_t float_to_bf16_bits(float f) {
    if (std::isnan(f)) return 0x7FC0;
    uint32_t u;
    std::memcpy(&u, &f, 4);
    return static_cast<uint16_t>((u + 0x7FFFu + ((u >> 16) & 1u)) >> 16);
}
inline float f16_bits_to_float(uint16_t h) {
    const uint32_t sign = static_cast<uint32_t>(h & 0x8000u) << 16;
    uint32_t e = (h >> 10) & 0x1F, m = h & 0x3FFu, u;
    if (e == 0) {
        if (m == 0) {
            u = sign;
        } else {
            int sh = 0;
            while (!(m & 0x400u)) { m <<= 1; ++sh; }
            u = sign | (static_cast<uint32_t>(113 - sh) << 23) | ((m & 0x3FFu) << 13);
        }
    } else if (e == 31) {
        u = sign | 0x7F800000u | (m << 13);
    } else {
        u = sign | ((e + 112) << 23) | (m << 13);
    }
    float f;
    std::memcpy(&f, &u, 4);
    return f;
}
struct Half { uint16_t x; operator float() const { return f16_bits_to_float(x); } };
struct BFloat16 { uint16_t x; operator float() const { return bf16_bits_to_float(x); } };
} // namespace dtype
