// sdsl_ref_r15.cpp — the rrr_vector<15> SPECIALISATION of the reference (rrr_vector_15.hpp), which a translation unit only
// gets when it includes that header explicitly: without it rrr_vector<15> is the generic template (what sdsl_ref.cpp sees).
// Own translation unit for that reason.  TEST INFRASTRUCTURE ONLY, like sdsl_ref.cpp.
#include <sdsl/bit_vectors.hpp>
#include <sdsl/rrr_vector_15.hpp>

#include <cstdlib>
#include <cstring>
#include <sstream>

extern "C" __attribute__((visibility("default"))) void ref_rrr15_spec_serialize(const uint64_t * words, uint64_t n_bits, uint8_t ** out, uint64_t * len)
{
    sdsl::bit_vector bv(n_bits, 0);
    if (n_bits)
        memcpy(bv.data(), words, ((n_bits + 63) >> 6) * 8);
    if (n_bits & 63)
        bv.data()[n_bits >> 6] &= sdsl::bits::lo_set[n_bits & 63];
    sdsl::rrr_vector<15> v(bv);
    std::ostringstream os;
    v.serialize(os);
    std::string s = os.str();
    *len = s.size();
    *out = (uint8_t *)malloc(s.size() ? s.size() : 1);
    memcpy(*out, s.data(), s.size());
}
