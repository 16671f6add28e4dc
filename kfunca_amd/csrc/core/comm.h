// namespace gpu, the collective half of the operator API: the ONE exchange step of the batch-sharded path - a
// sum all-reduce of weight gradients over RCCL / xGMI, through the C ABI (kf_comm_*, kf_allreduce_sum*). The reference has no
// distributed code at all (src/device/launcher_cuda.h:139-147 is its whole multi-device surface: set_device), so there is nothing to
// mirror; the shape follows the path: one process per GPU, one communicator per process, gradients reduced in place.
#pragma once

#include <cstdint>
#include <memory>
#include <string>
#include <unordered_map>
#include <vector>

#include "tensor.h"

namespace gpu {

// ---- the process's communicator ----------------------------------------------------------------------------------------------
std::string comm_unique_id();                                                      // rank 0 makes it (128 bytes), every rank gets it out of band
void comm_init(const std::string &unique_id, int rank, int world_size, int device);  // collective: every rank calls it
void comm_destroy();
bool comm_initialized();
int comm_rank();
int comm_world_size();

// ---- eager collectives, ordered on the device's compute stream -------------------------------------------------------------------
Tensor &all_reduce_(Tensor &t);              // in-place sum over ranks; t dense (contiguous)
void all_reduce_(std::vector<Tensor> &ts);   // the same for several tensors as ONE collective launch (same dtype and device, all dense)

// ---- flat gradient bucket ------------------------------------------------------------------------------------------------------
// The weight gradients of a model in ONE flat buffer, reduced in CHUNKS of at most `cap_bytes`, each chunk as soon as the backward
// pass has produced the last gradient it holds - on a communication stream of its own, ordered by events, so the collective runs
// under the rest of the backward (ring all-reduce over 7 point-to-point xGMI links is bandwidth-bound per link: few, large
// messages, started early). Gradients arrive in roughly the reverse of the order the parameters are used, so chunks are cut from
// the END of the parameter list: the last parameters' chunk completes - and leaves - first.
//   GradBucket b(params, cap);  b.attach();          once
//   per step:  zero_grad of the params;  forward;  loss.backward(g);  b.wait();   then read p.grad() (views into b.flat())
class GradBucket : public GradSink, public std::enable_shared_from_this<GradBucket> {
public:
    struct Chunk { int first, last; int64_t offset, numel; };  // parameters first..last, elements [offset, offset + numel) of the flat buffer
    // pure layout arithmetic (also what the CPU tests check): slot offsets (64-element aligned) and the chunks cut from the end
    static std::vector<int64_t> slot_offsets(const std::vector<int64_t> &numels, int64_t *total);
    static std::vector<Chunk> plan(const std::vector<int64_t> &numels, int64_t cap_elements);
    // Which collective leaves when - the bookkeeping of one backward pass, device-free so that the CPU tests run the SAME code the GPU
    // path runs (every rank must issue the same collectives in the same order: RCCL matches them by sequence). arrive(i): parameter i's
    // gradient is complete; returns the chunk that is now complete (to be fired) or -1. finish(): the chunks no gradient completed this
    // pass, in index order (still reduced: every rank issues every collective); then the tracker is ready for the next pass.
    class Tracker {
    public:
        Tracker() = default;
        Tracker(const std::vector<Chunk> &chunks, int nparams);
        int arrive(int param);
        std::vector<int> finish();
        const std::vector<int> &fired_order() const { return fired_order_; }
        int chunk_of(int param) const { return chunk_of_[param]; }
    private:
        void reset();
        std::vector<Chunk> chunks_;
        std::vector<int> chunk_of_, missing_, fired_order_;
        std::vector<char> have_, fired_;
        bool pass_open_ = false;
    };
    // the order in which the chunks of plan(numels, cap_elements) fire when the parameters' gradients arrive in `arrivals` (indices; repeats
    // and absentees allowed), finish() included: pure arithmetic, what tests/test_parallel_gloo.py compares across 8 ranks
    static std::vector<int> simulate_fired_order(const std::vector<int64_t> &numels, int64_t cap_elements, const std::vector<int> &arrivals);

    // accum_f32: the flat buffer (and every parameter's .grad view of it) is FLOAT whatever the parameters' dtype: the dW GEMMs of 16-bit
    // layers write their f32 accumulators into the slots unrounded (kf_gemm_epilogue.c_f32), RCCL sums floats, and the error of the
    // reduced gradient no longer grows with the number of ranks (a 16-bit bucket rounds once per addition: N 2^-8 sum |dW_r| at worst).
    // Twice the bytes on the wire; cap_bytes still bounds a chunk's message.
    static std::shared_ptr<GradBucket> create(const std::vector<Tensor> &params, int64_t cap_bytes, bool accum_f32 = false);
    ~GradBucket() override;
    void attach();   // the parameters' gradients now live in (and are written straight into) the flat buffer
    void detach();
    void wait();     // the compute stream waits for every chunk's collective; the bucket is ready for the next backward
    Tensor flat() const { return flat_; }
    const std::vector<Chunk> &chunks() const { return chunks_; }
    const std::vector<int> &fired_order() const { return tracker_.fired_order(); }  // chunk indices in the order their collectives were issued (last pass)
    int64_t reduced_bytes() const;
    // attribution (tools/block_bench.py): with collectives off the bucket still collects the gradients but issues nothing (the "off" arm
    // of exposed-communication timing); chunk_ms() = what each chunk's collective of the LAST pass took on the communication stream,
    // from the moment its gradients were ready to its end (0 for a chunk that issued none). Synchronises the communication stream.
    void set_collectives(bool on) { collectives_ = on; }
    std::vector<double> chunk_ms();

    Tensor slot(TensorImpl *leaf) override;
    Tensor take_slot(TensorImpl *leaf) override;
    void arrived(TensorImpl *leaf) override;

private:
    GradBucket() = default;
    void fire(int chunk);
    std::vector<Tensor> params_, slots_;
    std::unordered_map<TensorImpl *, int> index_;
    std::vector<int64_t> offsets_;
    std::vector<Chunk> chunks_;
    Tracker tracker_;
    std::vector<char> taken_;
    std::vector<void *> ev_ready_, ev_done_, ev_start_;
    std::vector<char> timed_;
    bool collectives_ = true;
    void *comm_stream_ = nullptr;
    Tensor flat_;
    int device_ = 0;
    bool attached_ = false;
};

} // namespace gpu
