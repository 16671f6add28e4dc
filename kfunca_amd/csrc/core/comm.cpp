#include "comm.h"

#include <algorithm>
#include <cstring>
#include <mutex>

#include "device_api.h"
#include "ops.h"

namespace gpu {

namespace {
struct CommState {
    void *comm = nullptr;
    int rank = 0, world = 1, device = 0;
};
CommState g_comm;
std::mutex g_comm_mu;

int code(ScalarType t) { return static_cast<int>(t); } // ScalarType order == KF_* codes (scalar_type.h)

void check_dense(const Tensor &t, const char *who) {
    CHECK_FAIL(t.defined() && t.is_dense(), who, ": the tensor must be dense (contiguous)");
}
} // namespace

std::string comm_unique_id() {
    char id[KF_COMM_ID_BYTES];
    DEV_CALL(kf_comm_unique_id(id));
    return std::string(id, KF_COMM_ID_BYTES);
}

void comm_init(const std::string &unique_id, int rank, int world_size, int device) {
    std::lock_guard<std::mutex> lk(g_comm_mu);
    CHECK_FAIL(g_comm.comm == nullptr, "comm_init: a communicator already exists (one per process: comm_destroy first)");
    CHECK_FAIL(unique_id.size() == KF_COMM_ID_BYTES, "comm_init: the unique id must be ", KF_COMM_ID_BYTES, " bytes, got ", unique_id.size());
    dev::set_device(device);
    void *c = nullptr;
    DEV_CALL(kf_comm_init(&c, unique_id.data(), rank, world_size));
    g_comm = {c, rank, world_size, device};
}

void comm_destroy() {
    std::lock_guard<std::mutex> lk(g_comm_mu);
    if (!g_comm.comm) return;
    dev::synchronize(g_comm.device);
    DEV_CALL(kf_comm_destroy(g_comm.comm));
    g_comm = CommState{};
}

namespace {
CommState comm_snapshot() { // the communicator as of now, read under the lock that comm_init / comm_destroy write it under
    std::lock_guard<std::mutex> lk(g_comm_mu);
    return g_comm;
}
} // namespace
bool comm_initialized() { return comm_snapshot().comm != nullptr; }
int comm_rank() { return comm_snapshot().rank; }
int comm_world_size() { return comm_snapshot().world; }

Tensor &all_reduce_(Tensor &t) {
    check_dense(t, "all_reduce_");
    const CommState cs = comm_snapshot();
    CHECK_FAIL(cs.comm != nullptr, "all_reduce_: no communicator (comm_init)");
    DEV_CALL(kf_allreduce_sum(cs.comm, t.data_ptr(), (size_t)t.numel(), code(t.dtype()), dev::stream(t.device())));
    return t;
}

void all_reduce_(std::vector<Tensor> &ts) {
    if (ts.empty()) return;
    const CommState cs = comm_snapshot();
    CHECK_FAIL(cs.comm != nullptr, "all_reduce_: no communicator (comm_init)");
    std::vector<void *> bufs;
    std::vector<size_t> counts;
    for (auto &t : ts) {
        check_dense(t, "all_reduce_");
        CHECK_FAIL(t.dtype() == ts[0].dtype() && t.device() == ts[0].device(), "all_reduce_: the tensors of one call share dtype and device");
        bufs.push_back(t.data_ptr());
        counts.push_back((size_t)t.numel());
    }
    DEV_CALL(kf_allreduce_sum_multi(cs.comm, (int)ts.size(), bufs.data(), counts.data(), code(ts[0].dtype()), dev::stream(ts[0].device())));
}

// ---- GradBucket ---------------------------------------------------------------------------------------------------------------
std::vector<int64_t> GradBucket::slot_offsets(const std::vector<int64_t> &numels, int64_t *total) {
    std::vector<int64_t> off(numels.size());
    int64_t o = 0;
    for (size_t i = 0; i < numels.size(); ++i) {
        off[i] = o;
        o += (numels[i] + 63) / 64 * 64; // 64 elements keep every slot 16-byte aligned for any dtype (and 128-byte for 16-bit ones)
    }
    if (total) *total = o;
    return off;
}

std::vector<GradBucket::Chunk> GradBucket::plan(const std::vector<int64_t> &numels, int64_t cap_elements) {
    int64_t total = 0;
    const std::vector<int64_t> off = slot_offsets(numels, &total);
    std::vector<Chunk> chunks; // cut from the END: chunk 0 holds the last parameters (their gradients arrive first)
    int last = (int)numels.size() - 1;
    while (last >= 0) {
        int first = last;
        auto span = [&](int f) { return (last + 1 < (int)numels.size() ? off[last + 1] : total) - off[f]; };
        while (first > 0 && span(first - 1) <= cap_elements) --first;
        chunks.push_back({first, last, off[first], span(first)});
        last = first - 1;
    }
    return chunks;
}

GradBucket::Tracker::Tracker(const std::vector<Chunk> &chunks, int nparams) : chunks_(chunks) {
    chunk_of_.assign(nparams, 0);
    for (size_t c = 0; c < chunks_.size(); ++c)
        for (int i = chunks_[c].first; i <= chunks_[c].last; ++i) chunk_of_[i] = (int)c;
    have_.assign(nparams, 0);
    fired_.assign(chunks_.size(), 0);
    missing_.resize(chunks_.size());
    reset();
}

void GradBucket::Tracker::reset() {
    pass_open_ = false;
    std::fill(have_.begin(), have_.end(), 0);
    std::fill(fired_.begin(), fired_.end(), 0);
    for (size_t c = 0; c < chunks_.size(); ++c) missing_[c] = chunks_[c].last - chunks_[c].first + 1;
}

int GradBucket::Tracker::arrive(int i) {
    if (!pass_open_) { fired_order_.clear(); pass_open_ = true; }
    if (have_[i]) return -1; // a second backward pass before wait(): accumulated in place, the chunk has already left (caller's protocol)
    have_[i] = 1;
    const int c = chunk_of_[i];
    if (--missing_[c] != 0) return -1;
    fired_[c] = 1;
    fired_order_.push_back(c);
    return c;
}

std::vector<int> GradBucket::Tracker::finish() {
    std::vector<int> late;
    if (!pass_open_) fired_order_.clear();
    for (size_t c = 0; c < chunks_.size(); ++c)
        if (!fired_[c]) { late.push_back((int)c); fired_order_.push_back((int)c); }
    const std::vector<int> order = fired_order_;
    reset();
    fired_order_ = order; // the finished pass's order stays readable until the next pass opens
    return late;
}

std::vector<int> GradBucket::simulate_fired_order(const std::vector<int64_t> &numels, int64_t cap_elements, const std::vector<int> &arrivals) {
    Tracker t(plan(numels, cap_elements), (int)numels.size());
    for (int i : arrivals) {
        CHECK_FAIL(i >= 0 && i < (int)numels.size(), "simulate_fired_order: parameter index ", i, " out of range");
        t.arrive(i);
    }
    t.finish();
    return t.fired_order();
}

std::shared_ptr<GradBucket> GradBucket::create(const std::vector<Tensor> &params, int64_t cap_bytes, bool accum_f32) {
    CHECK_FAIL(!params.empty(), "GradBucket: no parameters");
    std::shared_ptr<GradBucket> b(new GradBucket());
    b->params_ = params;
    b->device_ = params[0].device();
    std::vector<int64_t> numels;
    for (size_t i = 0; i < params.size(); ++i) {
        const Tensor &p = params[i];
        CHECK_FAIL(p.defined() && p.requires_grad() && !p.has_grad_fn(), "GradBucket: parameter ", i, " is not a leaf that requires a gradient");
        CHECK_FAIL(p.dtype() == params[0].dtype() && p.device() == b->device_, "GradBucket: the parameters of one bucket share dtype and device");
        CHECK_FAIL(b->index_.emplace(p.impl(), (int)i).second, "GradBucket: parameter ", i, " appears twice");
        numels.push_back(p.numel());
    }
    int64_t total = 0;
    b->offsets_ = slot_offsets(numels, &total);
    const ScalarType flat_dtype = accum_f32 ? ScalarType::Float : params[0].dtype();
    const int64_t es = (int64_t)element_size(flat_dtype);
    b->chunks_ = plan(numels, std::max<int64_t>(1, cap_bytes / es));
    b->flat_ = zeros({total}, flat_dtype, b->device_);
    b->tracker_ = Tracker(b->chunks_, (int)params.size());
    for (size_t i = 0; i < params.size(); ++i) b->slots_.push_back(b->flat_.narrow(0, b->offsets_[i], numels[i]).view(params[i].sizes()));
    b->taken_.assign(params.size(), 0);
    dev::set_device(b->device_);
    DEV_CALL(kf_stream_create(&b->comm_stream_));
    b->ev_ready_.resize(b->chunks_.size(), nullptr);
    b->ev_done_.resize(b->chunks_.size(), nullptr);
    b->ev_start_.resize(b->chunks_.size(), nullptr);
    b->timed_.assign(b->chunks_.size(), 0);
    for (size_t c = 0; c < b->chunks_.size(); ++c) {
        DEV_CALL(kf_event_create(&b->ev_ready_[c]));
        DEV_CALL(kf_event_create(&b->ev_done_[c]));
        DEV_CALL(kf_event_create(&b->ev_start_[c]));
    }
    return b;
}

GradBucket::~GradBucket() {
    // nothing may still be queued behind these events when they go: drain the communication stream first. The parameters hold the
    // bucket weakly (TensorImpl::sink_), so dropping the last handle lands here without a detach().
    if (comm_stream_) kf_stream_sync(comm_stream_);
    for (void *e : ev_ready_) if (e) kf_event_destroy(e);
    for (void *e : ev_done_) if (e) kf_event_destroy(e);
    for (void *e : ev_start_) if (e) kf_event_destroy(e);
    if (comm_stream_) kf_stream_destroy(comm_stream_);
}

void GradBucket::attach() {
    auto self = shared_from_this();
    for (auto &p : params_) {
        p.impl()->grad_.reset();
        p.impl()->sink_ = self;
    }
    attached_ = true;
}

void GradBucket::detach() {
    for (auto &p : params_)
        if (p.impl()->sink_.lock().get() == this) p.impl()->sink_.reset();
    attached_ = false;
}

Tensor GradBucket::slot(TensorImpl *leaf) {
    auto it = index_.find(leaf);
    CHECK_FAIL(it != index_.end(), "GradBucket: this tensor is not one of the bucket's parameters");
    return slots_[it->second];
}

Tensor GradBucket::take_slot(TensorImpl *leaf) {
    auto it = index_.find(leaf);
    CHECK_FAIL(it != index_.end(), "GradBucket: this tensor is not one of the bucket's parameters");
    if (taken_[it->second]) return Tensor(); // a second producer in this pass: it allocates its own and the engine adds
    taken_[it->second] = 1;
    return slots_[it->second];
}

int64_t GradBucket::reduced_bytes() const { return flat_.numel() * flat_.element_size_in_bytes(); }

void GradBucket::arrived(TensorImpl *leaf) {
    const int c = tracker_.arrive(index_.at(leaf));
    if (c >= 0) fire(c);
}

void GradBucket::fire(int c) {
    timed_[c] = 0;
    const CommState cs = comm_snapshot();
    if (!cs.comm || !collectives_) return; // a single process without a communicator: the sum over one rank is the gradient itself
    void *compute = dev::stream(device_);
    // the chunk's last gradient has been ENQUEUED on the compute stream: the collective waits for it there, not on the host
    DEV_CALL(kf_event_record(ev_ready_[c], compute));
    DEV_CALL(kf_stream_wait_event(comm_stream_, ev_ready_[c]));
    DEV_CALL(kf_event_record(ev_start_[c], comm_stream_));
    timed_[c] = 1;
    char *base = static_cast<char *>(flat_.data_ptr()) + chunks_[c].offset * flat_.element_size_in_bytes();
    DEV_CALL(kf_allreduce_sum(cs.comm, base, (size_t)chunks_[c].numel, code(flat_.dtype()), comm_stream_));
    DEV_CALL(kf_event_record(ev_done_[c], comm_stream_));
}

void GradBucket::wait() {
    void *compute = dev::stream(device_);
    // a chunk whose parameters got no gradient this pass (unused in the graph) is still reduced: every rank must issue the same collectives
    for (int c : tracker_.finish()) fire(c);
    for (size_t c = 0; c < chunks_.size(); ++c)
        if (timed_[c]) DEV_CALL(kf_stream_wait_event(compute, ev_done_[c]));
    std::fill(taken_.begin(), taken_.end(), 0);
}

std::vector<double> GradBucket::chunk_ms() {
    std::vector<double> out(chunks_.size(), 0.0);
    if (comm_stream_) DEV_CALL(kf_stream_sync(comm_stream_));
    for (size_t c = 0; c < chunks_.size(); ++c) {
        if (!timed_[c]) continue;
        float ms = 0.f;
        DEV_CALL(kf_event_elapsed_ms(ev_start_[c], ev_done_[c], &ms));
        out[c] = ms;
    }
    return out;
}

} // namespace gpu
