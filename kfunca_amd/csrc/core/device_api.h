// The only door from the host core to the device layer: the C ABI of include/kfunca_hip.h.
// Every non-zero status becomes the operator API's utils::Error; there is no other path to compute,
// so a missing GPU or a failed launch is loud by construction.
#pragma once

#include <cstddef>
#include <cstdint>

#include "check.h"
#include "kfunca_hip.h"

namespace dev {

inline void check(int status, const char *what) {
    if (status == KF_ERR_OOM) throw utils::OutOfMemory(utils::concat("[device error in ", what, ", status ", status, "] "), kf_last_error());
    if (status != KF_OK) {
        throw utils::Error(utils::concat("[device error in ", what, ", status ", status, "] "), kf_last_error());
    }
}
#define DEV_CALL(expr) ::dev::check((expr), #expr)

// per-device execution context: one explicit non-blocking stream per device (the reference runs
// kernels on the legacy default stream and copies on throw-away streams, launcher_cuda.h:170-202,315-353)
void set_device(int device);
int current_device();
void *stream(int device);
void synchronize(int device);
int device_count();

} // namespace dev
