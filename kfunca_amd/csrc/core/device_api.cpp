#include "device_api.h"

#include <mutex>
#include <vector>

namespace dev {

namespace {
std::mutex g_mu;
std::vector<void *> g_streams;
// HIP's current device is per THREAD, and the ctypes path (kfunca_amd/hip_abi.py, parallel.py) sets it without going through
// here: no shadow copy decides whether hipSetDevice is needed - it is always called (~100 ns); this one only answers
// current_device() for the calling thread.
thread_local int g_current = -1;
int g_count = -1;
} // namespace

int device_count() {
    int n = 0;
    if (kf_device_count(&n) != KF_OK) return 0;
    return n;
}

void set_device(int device) {
    int n;
    {
        std::lock_guard<std::mutex> lk(g_mu);
        if (g_count < 0) g_count = device_count();
        n = g_count;
    }
    CHECK_FAIL(n > 0, "no HIP device is visible: the kfunca_amd operator API has no CPU execution path");
    CHECK_FAIL(device >= 0 && device < n, "device ", device, " out of range (", n, " visible)");
    DEV_CALL(kf_set_device(device));
    g_current = device;
}

int current_device() { return g_current; }

void *stream(int device) {
    set_device(device);
    std::lock_guard<std::mutex> lk(g_mu);
    if ((int)g_streams.size() <= device) g_streams.resize(device + 1, nullptr);
    if (!g_streams[device]) DEV_CALL(kf_stream_create(&g_streams[device]));
    return g_streams[device];
}

void synchronize(int device) {
    void *s = stream(device);
    DEV_CALL(kf_stream_sync(s));
}

} // namespace dev
