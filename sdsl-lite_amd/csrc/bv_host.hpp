// bv_host.hpp — host-side owner of a device bit vector (rank lines + select directories) and the
// launch helpers other translation units (wavelet tree, FM-index) call.
#pragma once
#include "bv_device.hpp"
#include <mutex>

#include "common.hpp"

namespace sdslhip {

// Working memory of the bucketed batch paths (bv_swc.hip / bv_sorted.hip / rrr_sorted.hip): ONE pool per device, shared by every
// handle on it (12 bytes per query of the largest pass so far + tables: 13 GB for 10^9 queries — round 2 kept one per handle).
// `ev` is recorded behind the last user, `m` orders the host threads that enqueue users; sdsl_hip_*_release_scratch frees it.
struct DeviceScratch
{
    std::mutex m;
    DevBuf buf;
    hipEvent_t ev = nullptr;
};
DeviceScratch & device_scratch(int device); // (common.cpp; never destroyed: the runtime may be gone by the time statics are)

// One large batch's hold on its working memory.
//  * Outside a stream capture: the device's pool.  The lease holds the pool's mutex while the batch is being enqueued, makes the
//    stream wait for the previous user's event and RECORDS the event when it ends — on every path, also when a launch failed
//    half-way (kernels already queued behind the stream may still use the pool).
//  * While the stream is being captured into a graph: the HANDLE's own capture scratch, reserved beforehand
//    (sdsl_hip_*_reserve_capture_scratch).  A graph bakes the address in, and its replays are ordered with nothing the library
//    can see: it must not hold the pool (which moves when it grows and is shared with every other handle on the device), and an
//    event recorded inside a capture is not one later calls could wait on.  Without a reservation of sufficient size the
//    lease stays empty and the caller takes the direct kernel (no allocation is legal during a capture).
struct ScratchLease
{
    void * p = nullptr;
    size_t bytes = 0;
    bool capturing = false;
    // SDSL_HIP_OK with p == nullptr: no room (the caller falls back to the direct kernel)
    sdsl_hip_status acquire(int device, DevBuf & capture_buf, size_t need, hipStream_t s);
    ~ScratchLease();
    ScratchLease() = default;
    ScratchLease(const ScratchLease &) = delete;
    ScratchLease & operator=(const ScratchLease &) = delete;

private:
    DeviceScratch * pool = nullptr;
    std::unique_lock<std::mutex> lock;
    hipStream_t stream = nullptr;
};
void device_scratch_quiesce(int device);            // waits for the pool's last user (before a handle's memory is freed)
sdsl_hip_status device_scratch_release(int device); // the same, then frees the pool

struct BvHost
{
    int device = 0;
    BvView view{};
    DevBuf lines;  // n_lines * 64 B
    DevBuf cnts;   // u32 per line, build-time only
    DevBuf sel[2]; // select sample directories
    DevBuf lmask[2], lidx[2], lpos[2]; // sparse stretches of the select directories (BvView::lmask ...)
    DevBuf spread_probe; // two words the spread sample of the automatic dispatch writes (bv_sorted.hip)
    DevBuf capture_scratch; // working memory of large batches enqueued while the stream is being captured (ScratchLease)
    struct SelPlan // buckets of the bucketed batch select (bv_sorted.hip), built on first use
    {
        bool ready = false, ok = false;
        DevBuf bnd;
        unsigned bm = 8, bs = 3, nf = 0; // buckets of bm << bs argument ranks
        double wide_frac = 0;
    } sel_plan[2];
    std::mutex scratch_mutex; // the handle's own lazily built state (select plans, added directories, the spread probe)
    size_t device_bytes() const
    { // (without the device's scratch pool: sdsl_hip_device_scratch_bytes)
        size_t b = lines.bytes + sel[0].bytes + sel[1].bytes + spread_probe.bytes + capture_scratch.bytes;
        for (int i = 0; i < 2; ++i)
            b += lmask[i].bytes + lidx[i].bytes + lpos[i].bytes + sel_plan[i].bnd.bytes;
        return b;
    }
};

uint32_t default_sel_shift();
sdsl_hip_status device_exclusive_scan_u32(const uint32_t * d_in, uint64_t n, uint64_t * d_out, uint64_t stride,
                                          uint64_t * total);
sdsl_hip_status bv_build_from_device_words(BvHost & bv, const uint64_t * d_words, uint64_t n_bits, uint32_t flags,
                                           uint32_t sel_shift);
sdsl_hip_status compressed_stream_to_device_words(const void * bytes, size_t len, int kind, DevBuf & d_words, uint64_t & n_bits);
sdsl_hip_status bv_export_words_device(const BvView & v, uint64_t * d_words, uint64_t n_words, hipStream_t s);
sdsl_hip_status bv_launch_rank(const BvView & v, int bit, const uint64_t * d_idx, uint64_t n, uint64_t * d_out,
                               hipStream_t s);
// the layout behind a handle, and a handle on another device with buffers of the same sizes (contents: the caller's
// broadcast) — group.cpp
BvHost & bv_host_of(sdsl_hip_bv_t bv);
sdsl_hip_status bv_new_replica(const BvHost & src, int device, sdsl_hip_bv_t * out);
// large batches, bucketed by index region (bv_sorted.hip)
std::string bv_sorted_last_phases();
bool bv_sorted_rank_possible(const BvView & v);
bool bv_sorted_rank_applicable(const BvView & v, uint64_t n);
sdsl_hip_status bv_sorted_rank_is_spread(const BvView & v, const uint64_t * d_idx, uint64_t n, hipStream_t s, void * scratch,
                                         bool & spread);
size_t bv_sorted_rank_scratch_bytes(const BvView & v, uint64_t n);
// `go`: nullptr = run; else a device word the passes look at and return at once when it is zero (automatic dispatch)
sdsl_hip_status bv_launch_rank_sorted(const BvView & v, int bit, const uint64_t * d_idx, uint64_t n, uint64_t * d_out,
                                      hipStream_t s, void * scratch, size_t scratch_bytes, const uint32_t * go = nullptr);
bool bv_sorted_device_verdict();
sdsl_hip_status bv_sorted_rank_sample(const BvView & v, const uint64_t * d_idx, uint64_t n, hipStream_t s, uint32_t * out3);
sdsl_hip_status bv_sorted_select_sample(const BvHost & h, int bit, const uint64_t * d_idx, uint64_t n, hipStream_t s, uint32_t * out3);
sdsl_hip_status bv_select_sorted_prepare(BvHost & h, int bit);
bool bv_sorted_select_applicable(const BvHost & h, int bit, uint64_t n);
sdsl_hip_status bv_sorted_select_is_spread(const BvHost & h, int bit, const uint64_t * d_idx, uint64_t n, hipStream_t s,
                                           void * scratch, bool & spread);
sdsl_hip_status bv_launch_select_sorted(BvHost & h, int bit, const uint64_t * d_i, uint64_t n, uint64_t * d_out, hipStream_t s,
                                        void * scratch, size_t scratch_bytes, const uint32_t * go = nullptr);
sdsl_hip_status bv_launch_select(const BvView & v, int bit, const uint64_t * d_i, uint64_t n, uint64_t * d_out,
                                 hipStream_t s);

} // namespace sdslhip
