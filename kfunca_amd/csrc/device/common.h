// Shared helpers of the gfx950 device layer (not part of the C ABI).
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <string.h>

#include "kfunca_hip.h"

namespace kf {

void set_error(const char *fmt, ...);

#define KF_HIP_TRY(expr)                                                                              \
    do {                                                                                              \
        hipError_t e_ = (expr);                                                                       \
        if (e_ != hipSuccess) {                                                                       \
            ::kf::set_error("%s failed: %s (%s:%d)", #expr, hipGetErrorString(e_), __FILE__, __LINE__); \
            (void)hipGetLastError(); /* since ROCm 7.0 the last error is STICKY: reported here, it must not surface again at the next launch check */ \
            return KF_ERR_HIP;                                                                        \
        }                                                                                             \
    } while (0)

#define KF_REQUIRE(cond, code, ...)        \
    do {                                   \
        if (!(cond)) {                     \
            ::kf::set_error(__VA_ARGS__);  \
            return (code);                 \
        }                                  \
    } while (0)

// Checked after every launch: the reference never checks launch errors.
#define KF_LAUNCH_CHECK()                                                                    \
    do {                                                                                     \
        hipError_t e_ = hipGetLastError();                                                   \
        if (e_ != hipSuccess) {                                                              \
            ::kf::set_error("kernel launch failed: %s (%s:%d)", hipGetErrorString(e_), __FILE__, __LINE__); \
            return KF_ERR_HIP;                                                               \
        }                                                                                    \
    } while (0)

static inline int dtype_size(int dt) {
    switch (dt) {
    case KF_BOOL: case KF_U8: case KF_I8: return 1;
    case KF_I16: case KF_F16: case KF_BF16: return 2;
    case KF_I32: case KF_F32: return 4;
    case KF_I64: case KF_F64: return 8;
    default: return 0;
    }
}

// ---- 16-bit float storage types (bit-exact RNE, matching reference half.h:150-208) -----------
struct alignas(2) bf16_t { uint16_t x; };
struct alignas(2) f16_t { uint16_t x; };

__device__ __forceinline__ float bf16_to_f32(bf16_t v) { return __uint_as_float(((uint32_t)v.x) << 16); }
__device__ __forceinline__ bf16_t f32_to_bf16(float f) {
    // round-to-nearest-even, NaN -> 0x7FC0 (reference half.h:195-208)
    bf16_t r;
    uint32_t u = __float_as_uint(f);
    if ((u & 0x7fffffffu) > 0x7f800000u) { r.x = 0x7FC0; return r; }
    r.x = (uint16_t)((u + 0x7FFFu + ((u >> 16) & 1u)) >> 16);
    return r;
}
// two floats -> two bf16 in one word (lo in the low half) with the hardware's converter: round-to-nearest-even like f32_to_bf16; a NaN
// keeps its payload's top bits instead of becoming 0x7FC0, so the bit-exact conversion paths (copy / convert kernels) do not use it -
// the floating-point kernels, whose results are held to tolerances, do (one instruction instead of ~16 for the pair)
__device__ __forceinline__ uint32_t f32x2_to_bf16x2_hw(float lo, float hi) {
    uint32_t r;
    asm("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(r) : "v"(lo), "v"(hi));
    return r;
}
__device__ __forceinline__ float f16_to_f32(f16_t v) {
    _Float16 h;
    __builtin_memcpy(&h, &v.x, 2);
    return (float)h;
}
__device__ __forceinline__ f16_t f32_to_f16(float f) {
    _Float16 h = (_Float16)f; // v_cvt_f16_f32: RNE
    f16_t r;
    __builtin_memcpy(&r.x, &h, 2);
    return r;
}

constexpr int kWave = 64;

// per-launch timing (kf_profile_*): an event pair around the launches made inside the scope
struct ProfScope {
    const char *name;
    hipStream_t st;
    void *rec;
    ProfScope(const char *name, hipStream_t st);
    ~ProfScope();
};
#define KF_PROF(name, st) ::kf::ProfScope kf_prof_scope_((name), (st))

static inline hipStream_t as_stream(void *s) { return reinterpret_cast<hipStream_t>(s); }

// A/B switches (KF_* environment variables): read ONCE per process, not per launch (getenv walks the whole environment);
// kf_knobs_reload() re-reads them (tests and tools that flip a switch between two calls).
enum Knob {
    KNOB_ATTN_NO_XCD, KNOB_ATTN_NO_DEFER, KNOB_ATTN_NO_PAIR, KNOB_ATTN_F32_GENERIC, KNOB_ATTN_SPLIT_BWD, KNOB_GEMM_128, KNOB_GEMM_W4,
    KNOB_GEMM_W8, KNOB_GEMM_GROUP_M, KNOB_GEMM_F64_GENERIC, KNOB_REDUCE_NO_TALL, KNOB_GEMM_NO_SPLITK, KNOB_GEMM_NO_GROUP, KNOB_ATTN_DS_CAP_MB, KNOB_NORM_BWD_TPR, KNOB_ATTN_FWD_V3, KNOB_ATTN_DKV_V4, KNOB_ATTN_SCALED_OPERANDS, KNOB_ATTN_GRID_WGS, KNOB_GEMM_NO_PAD, KNOB_GEMM_H256_MIN, KNOB_EW_ALIGNED_ONLY, KNOB_ATTN_DS_TRI, KNOB_COUNT
};
bool knob(Knob k);              // the variable is set
long knob_int(Knob k, long dflt); // its integer value, dflt when unset

// hipFuncSetAttribute(MaxDynamicSharedMemorySize) once per (kernel, device), not once per launch
int ensure_dynamic_lds(const void *kernel, int bytes);
#define KF_ENSURE_LDS(kernel, bytes)                                                      \
    do {                                                                                  \
        int rc_ = ::kf::ensure_dynamic_lds((const void *)(kernel), (int)(bytes));         \
        if (rc_ != KF_OK) return rc_;                                                     \
    } while (0)

// kf_sort with a promise about the keys (sort.hip)
int sort_with_key_bits(int dtype, const void *keys_in, void *keys_out, int64_t *pos_out, int64_t nseg, int64_t n, int descending, void *workspace,
                       size_t workspace_bytes, void *stream, int key_bits);

} // namespace kf
