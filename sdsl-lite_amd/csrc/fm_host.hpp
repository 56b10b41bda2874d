// fm_host.hpp — host-side owner of a device FM-index (shared by fm.hip: count path, and locate.hip: SA / ISA /
// LF / psi / extract / locate).
#pragma once
#include <vector>

#include "fm_device.hpp"
#include "wt_host.hpp"

struct sdsl_hip_wt_s;
sdsl_hip_wt_s * sdsl_hip_wt_alloc();
sdslhip::WtHost & sdsl_hip_wt_host(sdsl_hip_wt_s * w);
sdsl_hip_status sdsl_hip_wt_serialize_into(sdsl_hip_wt_s * wt, int32_t layout, sdslhip::StreamWriter & w);
const uint64_t * sdsl_hip_wt_device_occ(sdsl_hip_wt_s * w);
sdsl_hip_status sdsl_hip_wt_finish(sdsl_hip_wt_s * w);

struct sdsl_hip_fm_s
{
    int device = 0;
    uint64_t size = 0;
    uint32_t sigma = 0;
    sdsl_hip_wt_s * wt = nullptr;
    uint64_t uid = sdslhip::next_handle_uid(); // key of the serialiser's size-query cache
    sdslhip::FmTables tab;
    sdslhip::DevBuf d_tab;
    sdslhip::DevBuf d_sa; // suffix array (u32 per suffix) of an index created from text; empty otherwise
    sdslhip::DevBuf d_sa64; // the same as u64 per suffix, for an index of 2^32 symbols and more created from text (d_sa is empty then)
    sdslhip::DevBuf d_text; // the text itself, kept beside the whole suffix array: count() verifies a pattern whose interval has
                            // shrunk to ONE suffix against the text instead of walking its remaining characters (fm.hip)
    // SA-order SA samples SA[k*sa_dens] and text-order ISA samples ISA[k*isa_dens] (csa_sampling_strategy.hpp:72-135,
    // 735-806), u64 each; density 0 = not present
    sdslhip::DevBuf d_sa_s, d_isa_s;
    uint32_t sa_dens = 0, isa_dens = 0;
    uint64_t n_sa_s = 0, n_isa_s = 0;
    bool samples32 = false; // the samples are u32 each (sdsl_hip_fm_set_footprint packs them; fewer than 2^32 symbols)
    sdslhip::DevBuf d_jump; // jump-start table (fm_device.hpp FmJump), sigma^jump_k (l, r) pairs
    uint32_t jump_k = 0;
    // count() of large batches (fm_count2.hip): the per-byte step tables of the flat kernel and the k-mer table
    sdslhip::DevBuf d_ctab;
    bool ctab_ok = false;
    sdslhip::DevBuf d_deep;
    uint32_t deep_k = 0, deep_buckets = 0;
    uint64_t deep_kmers = 0;
    sdslhip::FmDeep deep() const
    {
        sdslhip::FmDeep d;
        d.tab = deep_k ? d_deep.as<ulonglong2>() : nullptr;
        d.k = deep_k;
        d.n_buckets = deep_buckets;
        return d;
    }
    sdslhip::FmJump jump() const
    {
        sdslhip::FmJump j;
        j.tab = jump_k ? d_jump.as<uint64_t>() : nullptr;
        j.k = jump_k;
        j.sigma = sigma;
        return j;
    }
};

namespace sdslhip {

// what the locate kernels need besides the wavelet tree and the alphabet
struct FmLocView
{
    const uint32_t * sa_full; // or null
    const uint64_t * sa_s;    // or null
    const uint64_t * isa_s;   // or null
    uint64_t sa_dens, isa_dens, n_isa_s;
    uint64_t size;
    uint32_t s32;             // samples are 32 bits wide (read through loc_sample)
};

// sample i of a sample array that is u64 or, packed (sdsl_hip_fm_set_footprint), u32 per entry
__device__ __forceinline__ uint64_t loc_sample(const uint64_t * p, uint32_t s32, uint64_t i)
{
    return s32 ? (uint64_t)reinterpret_cast<const uint32_t *>(p)[i] : p[i];
}

sdsl_hip_status sa_build_bwt_device(const uint8_t * text, uint64_t n_text, int device, DevBuf & d_bwt, DevBuf & d_sa);
// texts of 2^32 - 2 bytes and more: 64-bit suffixes (d_sa: u64 per suffix)
sdsl_hip_status sa_build_bwt_device64(const uint8_t * text, uint64_t n_text, int device, DevBuf & d_bwt, DevBuf & d_sa);
sdsl_hip_status sa_samples_device64(const uint64_t * d_sa, uint64_t n, uint64_t sa_dens, uint64_t isa_dens, DevBuf * sa_s,
                                    DevBuf * isa_s);
sdsl_hip_status sa_samples_to_host(const uint32_t * d_sa, uint64_t n, uint64_t sa_dens, uint64_t isa_dens,
                                   std::vector<uint64_t> & sa_s, std::vector<uint64_t> & isa_s);
// samples from the full suffix array, left on the device (either output may be null)
sdsl_hip_status sa_samples_device(const uint32_t * d_sa, uint64_t n, uint64_t sa_dens, uint64_t isa_dens, DevBuf * sa_s,
                                  DevBuf * isa_s);
// fm_count2.hip
sdsl_hip_status fm_build_count_tab(sdsl_hip_fm_s * f);
sdsl_hip_status fm_build_deep(sdsl_hip_fm_s * f, uint32_t k_max, uint64_t budget_bytes);
sdsl_hip_status fm_build_deep_default(sdsl_hip_fm_s * f);
bool fm_fast_applies(const sdsl_hip_fm_s * f, uint32_t m, uint64_t n_pat);
sdsl_hip_status fm_count_fast(sdsl_hip_fm_s * f, const uint8_t * d_pats, uint32_t m, uint64_t n_pat, uint64_t * d_out, bool verify,
                              hipStream_t s);
size_t sort_pairs_u64_u32_temp_bytes(uint64_t n, unsigned end_bit);
sdsl_hip_status sort_pairs_u64_u32(uint64_t * keys_in, uint64_t * keys_out, uint32_t * vals_in, uint32_t * vals_out,
                                   uint64_t n, unsigned end_bit, hipStream_t s, void * tmp = nullptr, size_t tmp_bytes = 0);
// out[i] = in[0] + ... + in[i-1] (n entries).  With working memory of the caller's (>= exclusive_scan_u64_temp_bytes(n)): stream-ordered,
// nothing allocated, nothing waited for; without: allocates, and synchronises the stream before the memory is returned
size_t exclusive_scan_u64_temp_bytes(uint64_t n);
sdsl_hip_status exclusive_scan_u64(const uint64_t * in, uint64_t * out, uint64_t n, hipStream_t s, void * tmp = nullptr, size_t tmp_bytes = 0);

} // namespace sdslhip
