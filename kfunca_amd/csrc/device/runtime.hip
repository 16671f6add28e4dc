// Runtime half of the C ABI: device selection, raw memory, copies, streams, events.
// Replaces the reference's memory_engine (src/device/memory_engine.cu:6-28) and the runtime
// half of its Launcher singleton (src/device/launcher_cuda.h:105-291) with plain HIP calls —
// no singleton, no per-copy stream churn (the reference creates+syncs+destroys a stream for
// every memcpy/memset, launcher_cuda.h:170-202).
#include <stdarg.h>

#include "common.h"

namespace kf {
static thread_local char g_err[1024] = "";
void set_error(const char *fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}
} // namespace kf

// ---- per-launch timing registry -------------------------------------------------------------
#include <algorithm>
#include <mutex>
#include <string>
#include <vector>
namespace kf {
struct ProfRec { std::string name; hipEvent_t e0, e1; };
static bool g_prof_on = false;
static int g_capturing = 0; // open stream captures: event records would be captured into the graph, so profiling stands aside
static std::mutex g_prof_mu;
static std::vector<ProfRec *> g_prof_recs;
ProfScope::ProfScope(const char *n, hipStream_t s) : name(n), st(s), rec(nullptr) {
    if (!g_prof_on || g_capturing) return;
    ProfRec *r = new ProfRec();
    r->name = n;
    if (hipEventCreate(&r->e0) != hipSuccess || hipEventCreate(&r->e1) != hipSuccess) { delete r; return; }
    hipEventRecord(r->e0, s);
    rec = r;
}
ProfScope::~ProfScope() {
    if (!rec) return;
    ProfRec *r = (ProfRec *)rec;
    hipEventRecord(r->e1, st);
    std::lock_guard<std::mutex> lk(g_prof_mu);
    g_prof_recs.push_back(r);
}
struct ProfSum { std::string name; double ms; int64_t n; std::vector<float> each; }; // each: the launches' own durations (for percentiles)
static std::vector<ProfSum> g_prof_sums;
static void prof_collect() { // folds finished records into per-name sums
    std::lock_guard<std::mutex> lk(g_prof_mu);
    for (ProfRec *r : g_prof_recs) {
        float ms = 0.f;
        hipEventSynchronize(r->e1);
        hipEventElapsedTime(&ms, r->e0, r->e1);
        hipEventDestroy(r->e0);
        hipEventDestroy(r->e1);
        bool found = false;
        for (auto &s : g_prof_sums)
            if (s.name == r->name) { s.ms += ms; s.n += 1; if (s.each.size() < 65536) s.each.push_back(ms); found = true; break; }
        if (!found) g_prof_sums.push_back({r->name, (double)ms, 1, {ms}});
        delete r;
    }
    g_prof_recs.clear();
}
} // namespace kf

// ---- A/B switches and per-kernel attributes, cached ----------------------------------------------
#include <stdlib.h>
namespace kf {
static const char *const g_knob_names[KNOB_COUNT] = {
    "KF_ATTN_NO_XCD", "KF_ATTN_NO_DEFER", "KF_ATTN_NO_PAIR", "KF_ATTN_F32_GENERIC", "KF_ATTN_SPLIT_BWD", "KF_GEMM_128", "KF_GEMM_W4",
    "KF_GEMM_W8", "KF_GEMM_GROUP_M", "KF_GEMM_F64_GENERIC", "KF_REDUCE_NO_TALL", "KF_GEMM_NO_SPLITK", "KF_GEMM_NO_GROUP", "KF_ATTN_DS_CAP_MB", "KF_NORM_BWD_TPR", "KF_ATTN_FWD_V3", "KF_ATTN_DKV_V4", "KF_ATTN_SCALED_OPERANDS", "KF_ATTN_GRID_WGS", "KF_GEMM_NO_PAD", "KF_GEMM_H256_MIN", "KF_EW_ALIGNED_ONLY", "KF_ATTN_DS_TRI"};
static std::mutex g_knob_mu;
static bool g_knob_loaded = false;
static bool g_knob_set[KNOB_COUNT];
static long g_knob_val[KNOB_COUNT];
static void knobs_load_locked() {
    for (int i = 0; i < KNOB_COUNT; ++i) {
        const char *e = getenv(g_knob_names[i]);
        g_knob_set[i] = e != nullptr;
        g_knob_val[i] = e ? strtol(e, nullptr, 10) : 0;
    }
    g_knob_loaded = true;
}
bool knob(Knob k) {
    if (!g_knob_loaded) {
        std::lock_guard<std::mutex> lk(g_knob_mu);
        if (!g_knob_loaded) knobs_load_locked();
    }
    return g_knob_set[k];
}
long knob_int(Knob k, long dflt) { return knob(k) ? g_knob_val[k] : dflt; }

struct LdsAttr { const void *fn; int dev; int bytes; };
static std::mutex g_lds_mu;
static std::vector<LdsAttr> g_lds_attrs;
int ensure_dynamic_lds(const void *kernel, int bytes) {
    int dev = 0;
    KF_HIP_TRY(hipGetDevice(&dev));
    std::lock_guard<std::mutex> lk(g_lds_mu);
    for (auto &a : g_lds_attrs)
        if (a.fn == kernel && a.dev == dev) {
            if (a.bytes >= bytes) return KF_OK;
            KF_HIP_TRY(hipFuncSetAttribute(kernel, hipFuncAttributeMaxDynamicSharedMemorySize, bytes));
            a.bytes = bytes;
            return KF_OK;
        }
    KF_HIP_TRY(hipFuncSetAttribute(kernel, hipFuncAttributeMaxDynamicSharedMemorySize, bytes));
    g_lds_attrs.push_back({kernel, dev, bytes});
    return KF_OK;
}
} // namespace kf

using namespace kf;

extern "C" {

int kf_knobs_reload(void) {
    std::lock_guard<std::mutex> lk(g_knob_mu);
    knobs_load_locked();
    return KF_OK;
}

int kf_profile_enable(int on) {
    KF_REQUIRE(!(on && g_capturing), KF_ERR_INVALID, "kf_profile_enable: a stream capture is open (timing events would be recorded into the graph)");
    g_prof_on = on != 0;
    return KF_OK;
}
int kf_profile_reset(void) {
    prof_collect();
    std::lock_guard<std::mutex> lk(g_prof_mu);
    g_prof_sums.clear();
    return KF_OK;
}
int kf_profile_count(int *n) {
    KF_REQUIRE(n, KF_ERR_INVALID, "kf_profile_count: null out pointer");
    prof_collect();
    *n = (int)g_prof_sums.size();
    return KF_OK;
}
int kf_profile_get(int i, char name[64], double *total_ms, int64_t *launches) {
    KF_REQUIRE(name && total_ms && launches, KF_ERR_INVALID, "kf_profile_get: null out pointer");
    prof_collect();
    KF_REQUIRE(i >= 0 && i < (int)g_prof_sums.size(), KF_ERR_INVALID, "kf_profile_get: index %d out of range", i);
    snprintf(name, 64, "%s", g_prof_sums[i].name.c_str());
    *total_ms = g_prof_sums[i].ms;
    *launches = g_prof_sums[i].n;
    return KF_OK;
}

int kf_profile_samples(int i, float *ms, int capacity, int *written) {
    KF_REQUIRE(ms && written && capacity >= 0, KF_ERR_INVALID, "kf_profile_samples: null out pointer");
    prof_collect();
    KF_REQUIRE(i >= 0 && i < (int)g_prof_sums.size(), KF_ERR_INVALID, "kf_profile_samples: index %d out of range", i);
    const int n = std::min<int>(capacity, (int)g_prof_sums[i].each.size());
    for (int j = 0; j < n; ++j) ms[j] = g_prof_sums[i].each[j];
    *written = n;
    return KF_OK;
}

const char *kf_last_error(void) { return g_err; }
int kf_abi_version(void) { return KF_ABI_VERSION; }

int kf_device_count(int *count) {
    KF_REQUIRE(count, KF_ERR_INVALID, "kf_device_count: null out pointer");
    *count = 0;
    KF_HIP_TRY(hipGetDeviceCount(count));
    return KF_OK;
}

int kf_set_device(int device) {
    KF_HIP_TRY(hipSetDevice(device));
    return KF_OK;
}

int kf_get_device(int *device) {
    KF_REQUIRE(device, KF_ERR_INVALID, "kf_get_device: null out pointer");
    KF_HIP_TRY(hipGetDevice(device));
    return KF_OK;
}

int kf_malloc(void **ptr, size_t bytes) {
    KF_REQUIRE(ptr, KF_ERR_INVALID, "kf_malloc: null out pointer");
    *ptr = nullptr;
    if (bytes == 0) return KF_OK;
    const hipError_t e = hipMalloc(ptr, bytes);
    if (e == hipErrorOutOfMemory) {
        // a recoverable condition with a status of its own: the caller may release cached memory and retry, or ask for less.
        // hipGetLastError() keeps the last REAL error until read (ROCm >= 7.0; successful calls no longer reset it): unread, the next
        // KF_LAUNCH_CHECK would report "out of memory" for a kernel that launched fine.
        (void)hipGetLastError();
        *ptr = nullptr;
        set_error("kf_malloc: out of device memory (%zu bytes requested)", bytes);
        return KF_ERR_OOM;
    }
    KF_HIP_TRY(e);
    return KF_OK;
}

int kf_free(void *ptr) {
    if (!ptr) return KF_OK;
    KF_HIP_TRY(hipFree(ptr));
    return KF_OK;
}

int kf_memcpy_h2d(void *dst, const void *src, size_t bytes, void *stream) {
    if (bytes == 0) return KF_OK;
    KF_REQUIRE(dst && src, KF_ERR_INVALID, "kf_memcpy_h2d: null pointer");
    KF_HIP_TRY(hipMemcpyAsync(dst, src, bytes, hipMemcpyHostToDevice, as_stream(stream)));
    KF_HIP_TRY(hipStreamSynchronize(as_stream(stream)));
    return KF_OK;
}

int kf_memcpy_d2h(void *dst, const void *src, size_t bytes, void *stream) {
    if (bytes == 0) return KF_OK;
    KF_REQUIRE(dst && src, KF_ERR_INVALID, "kf_memcpy_d2h: null pointer");
    KF_HIP_TRY(hipMemcpyAsync(dst, src, bytes, hipMemcpyDeviceToHost, as_stream(stream)));
    KF_HIP_TRY(hipStreamSynchronize(as_stream(stream)));
    return KF_OK;
}

int kf_memcpy_d2d(void *dst, const void *src, size_t bytes, void *stream) {
    if (bytes == 0) return KF_OK;
    KF_REQUIRE(dst && src, KF_ERR_INVALID, "kf_memcpy_d2d: null pointer");
    KF_HIP_TRY(hipMemcpyAsync(dst, src, bytes, hipMemcpyDeviceToDevice, as_stream(stream)));
    return KF_OK;
}

int kf_memset_zero(void *ptr, size_t bytes, void *stream) {
    if (bytes == 0) return KF_OK;
    KF_REQUIRE(ptr, KF_ERR_INVALID, "kf_memset_zero: null pointer");
    KF_HIP_TRY(hipMemsetAsync(ptr, 0, bytes, as_stream(stream)));
    return KF_OK;
}

int kf_stream_create(void **stream) {
    KF_REQUIRE(stream, KF_ERR_INVALID, "kf_stream_create: null out pointer");
    hipStream_t s;
    KF_HIP_TRY(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
    *stream = s;
    return KF_OK;
}

int kf_graph_begin_capture(void *stream) {
    KF_REQUIRE(stream, KF_ERR_INVALID, "kf_graph_begin_capture: capture needs a stream from kf_stream_create, not the null stream");
    KF_HIP_TRY(hipStreamBeginCapture(as_stream(stream), hipStreamCaptureModeRelaxed));
    ++g_capturing;
    return KF_OK;
}

int kf_graph_end_capture(void *stream, void **graph_exec) {
    KF_REQUIRE(stream && graph_exec, KF_ERR_INVALID, "kf_graph_end_capture: null argument");
    hipGraph_t g = nullptr;
    if (g_capturing > 0) --g_capturing;
    KF_HIP_TRY(hipStreamEndCapture(as_stream(stream), &g));
    KF_REQUIRE(g, KF_ERR_HIP, "kf_graph_end_capture: the capture was invalidated");
    hipGraphExec_t ge = nullptr;
    hipError_t e = hipGraphInstantiate(&ge, g, nullptr, nullptr, 0);
    (void)hipGraphDestroy(g);
    KF_HIP_TRY(e);
    *graph_exec = ge;
    return KF_OK;
}

int kf_graph_launch(void *graph_exec, void *stream) {
    KF_REQUIRE(graph_exec, KF_ERR_INVALID, "kf_graph_launch: null graph");
    KF_HIP_TRY(hipGraphLaunch((hipGraphExec_t)graph_exec, as_stream(stream)));
    return KF_OK;
}

int kf_graph_destroy(void *graph_exec) {
    if (!graph_exec) return KF_OK;
    KF_HIP_TRY(hipGraphExecDestroy((hipGraphExec_t)graph_exec));
    return KF_OK;
}

int kf_stream_destroy(void *stream) {
    if (!stream) return KF_OK;
    KF_HIP_TRY(hipStreamDestroy(as_stream(stream)));
    return KF_OK;
}

int kf_stream_sync(void *stream) {
    KF_HIP_TRY(hipStreamSynchronize(as_stream(stream)));
    return KF_OK;
}

int kf_stream_wait_event(void *stream, void *event) {
    KF_REQUIRE(event, KF_ERR_INVALID, "kf_stream_wait_event: null event");
    KF_HIP_TRY(hipStreamWaitEvent(as_stream(stream), reinterpret_cast<hipEvent_t>(event), 0));
    return KF_OK;
}

int kf_device_sync(void) {
    KF_HIP_TRY(hipDeviceSynchronize());
    return KF_OK;
}

int kf_event_create(void **event) {
    KF_REQUIRE(event, KF_ERR_INVALID, "kf_event_create: null out pointer");
    hipEvent_t e;
    KF_HIP_TRY(hipEventCreate(&e));
    *event = e;
    return KF_OK;
}

int kf_event_destroy(void *event) {
    if (!event) return KF_OK;
    KF_HIP_TRY(hipEventDestroy(reinterpret_cast<hipEvent_t>(event)));
    return KF_OK;
}

int kf_event_record(void *event, void *stream) {
    KF_HIP_TRY(hipEventRecord(reinterpret_cast<hipEvent_t>(event), as_stream(stream)));
    return KF_OK;
}

int kf_event_sync(void *event) {
    KF_HIP_TRY(hipEventSynchronize(reinterpret_cast<hipEvent_t>(event)));
    return KF_OK;
}

int kf_event_elapsed_ms(void *start, void *stop, float *ms) {
    KF_REQUIRE(ms, KF_ERR_INVALID, "kf_event_elapsed_ms: null out pointer");
    KF_HIP_TRY(hipEventElapsedTime(ms, reinterpret_cast<hipEvent_t>(start), reinterpret_cast<hipEvent_t>(stop)));
    return KF_OK;
}

int kf_device_props_get(int device, kf_device_props *out) {
    KF_REQUIRE(out, KF_ERR_INVALID, "kf_device_props_get: null out pointer");
    hipDeviceProp_t p;
    KF_HIP_TRY(hipGetDeviceProperties(&p, device));
    memset(out, 0, sizeof(*out));
    snprintf(out->name, sizeof(out->name), "%s", p.name);
    snprintf(out->arch, sizeof(out->arch), "%s", p.gcnArchName);
    out->compute_units = p.multiProcessorCount;
    out->wavefront_size = p.warpSize;
    out->max_threads_per_block = p.maxThreadsPerBlock;
    out->clock_khz = p.clockRate;
    out->memory_clock_khz = p.memoryClockRate;
    out->memory_bus_bits = p.memoryBusWidth;
    out->lds_per_block = (int64_t)p.sharedMemPerBlock;
    out->l2_bytes = (int64_t)p.l2CacheSize;
    out->total_mem = p.totalGlobalMem;
    int cur = 0;
    KF_HIP_TRY(hipGetDevice(&cur));
    if (cur != device) KF_HIP_TRY(hipSetDevice(device));
    size_t fr = 0, tot = 0;
    hipError_t e = hipMemGetInfo(&fr, &tot);
    if (cur != device) KF_HIP_TRY(hipSetDevice(cur));
    out->free_mem = (e == hipSuccess) ? fr : 0;
    return KF_OK;
}

} // extern "C"
