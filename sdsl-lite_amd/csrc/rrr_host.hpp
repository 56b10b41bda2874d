// rrr_host.hpp — host-side owner of a device rrr_vector<63> and its builders.
#pragma once
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
    size_t device_bytes() const
    {
        return rec.bytes + stream.bytes + tables.bytes + sel[0].bytes + sel[1].bytes;
    }
};

// rrr_vector<63>(bit_vector const&) on the device: words (device memory) -> records, stream, directories
sdsl_hip_status rrr_build_device(RrrHost & h, const uint64_t * d_words, uint64_t n_bits, int device);
// from rrr_vector<63>::serialize bytes (rrr_vector.hpp:366-378); advances the reader
sdsl_hip_status rrr_build_from_stream(RrrHost & h, StreamReader & rd, int device);
// same, optionally also returning the plain bits (host decode) for structural validation by the caller
sdsl_hip_status rrr_parse_and_upload(RrrHost & h, StreamReader & rd, int device, std::vector<uint64_t> * words_out);
// writes rrr_vector<63>::serialize bytes for the device structure
sdsl_hip_status rrr_serialize_host(const RrrHost & h, StreamWriter & w);

} // namespace sdslhip
