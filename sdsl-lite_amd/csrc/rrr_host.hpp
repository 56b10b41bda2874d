// rrr_host.hpp — host-side owner of a device rrr_vector<63> and its builders.
#pragma once
#include <mutex>
#include "bv_host.hpp"
#include "rrr_device.hpp"
#include "sdsl_stream.hpp"

namespace sdslhip {

struct RrrHost
{
    int device = 0;
    RrrView view{};
    DevBuf rec, stream, tables, sel[2];
    unsigned sparse_max = 10; // classes sparse_max + 1 .. 62 - sparse_max are stored raw (rrr_device.hpp)
    unsigned fmt = 0;         // record format (rrr_device.hpp: RrrFmtW / RrrFmtS); the vectors of a wavelet tree are always wide
    bool allow_slim = false;  // set by the stand-alone handle before the vector is built
    DevBuf spread_probe; // the verdict of the spread sample (rrr_sorted.hip); the passes' working memory is the device's pool (bv_host.hpp)
    DevBuf capture_scratch; // working memory of large batches enqueued while the stream is being captured (ScratchLease)
    struct SelPlan // buckets of the bucketed batch select (rrr_sorted.hip), built on first use
    {
        bool ready = false, ok = false;
        DevBuf bnd;
        unsigned bm = 8, bs = 3, nf = 0, rlog = 7; // buckets of bm << bs argument ranks; 2^rlog records per slice
        double wide_frac = 0;
    } sel_plan[2];
    std::mutex scratch_mutex;
    size_t device_bytes() const
    {
        return rec.bytes + stream.bytes + tables.bytes + sel[0].bytes + sel[1].bytes + spread_probe.bytes + capture_scratch.bytes;
    }
};
// large batches, bucketed by slice of the record array (rrr_sorted.hip)
bool rrr_sorted_rank_possible(const RrrView & v);
bool rrr_sorted_rank_applicable(const RrrView & v, uint64_t n);
void rrr_sorted_rank_sample(const RrrView & v, const uint64_t * d_idx, uint64_t n, hipStream_t s, uint32_t * out3);
sdsl_hip_status rrr_select_sorted_prepare(RrrHost & h, int bit);
bool rrr_sorted_select_applicable(const RrrHost & h, int bit, uint64_t n);
void rrr_sorted_select_sample(const RrrHost & h, int bit, const uint64_t * d_idx, uint64_t n, hipStream_t s, uint32_t * out3);
sdsl_hip_status rrr_launch_select_sorted(RrrHost & h, int bit, const uint64_t * d_i, uint64_t n, uint64_t * d_out, hipStream_t s, void * scratch,
                                         size_t scratch_bytes, const uint32_t * go);
// the direct select kernel on device arrays (rrr.hip)
sdsl_hip_status rrr_launch_select(const RrrView & v, int bit, const uint64_t * d_i, uint64_t n, uint64_t * d_out, hipStream_t s);
size_t bv_swc_scratch_bytes(const BvView & v, uint64_t n); // (bv_swc.hip: the scratch of a pass depends on the batch only)
sdsl_hip_status rrr_launch_rank_sorted(const RrrView & v, int bit, const uint64_t * d_idx, uint64_t n, uint64_t * d_out, hipStream_t s,
                                       void * scratch, size_t scratch_bytes, const uint32_t * go);

// rrr_vector<63>(bit_vector const&) on the device: words (device memory) -> records, stream, directories
sdsl_hip_status rrr_build_device(RrrHost & h, const uint64_t * d_words, uint64_t n_bits, int device);
// from rrr_vector<63>::serialize bytes (rrr_vector.hpp:366-378); advances the reader
sdsl_hip_status rrr_build_from_stream(RrrHost & h, StreamReader & rd, int device);
// same, optionally also returning the plain bits (host decode) for structural validation by the caller
sdsl_hip_status rrr_parse_and_upload(RrrHost & h, StreamReader & rd, int device, std::vector<uint64_t> * words_out);
// writes rrr_vector<63>::serialize bytes for the device structure
sdsl_hip_status rrr_serialize_host(const RrrHost & h, StreamWriter & w);

} // namespace sdslhip
