// bv_serialize.hpp — host-side writers of SDSL's serialised rank / select supports over exported bit-vector words
// (bv_serialize.cpp).  `words` must hold ceil(n_bits / 64) words; bit = 0 / 1 is the supported bit value.
#pragma once
#include "sdsl_stream.hpp"

namespace sdslhip {

void rank_v5_serialize_host(const uint64_t * words, uint64_t n_bits, int bit, StreamWriter & out);
void rank_v_serialize_host(const uint64_t * words, uint64_t n_bits, int bit, StreamWriter & out);
void select_mcl_serialize_host(const uint64_t * words, uint64_t n_bits, int bit, StreamWriter & out);

} // namespace sdslhip
