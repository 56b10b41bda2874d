/* oracle.h — CPU restatement of the SDSL algorithms on the rank/select/wt/count hot path.
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing under oracle/ is part of the product: only tests/,
 * __graft_entry__.smoke() and the cpu_baseline leg of bench.py may load this library, and only
 * as the checker / as the reported CPU baseline.  The product (sdsl-lite_amd/lib/libsdsl_hip.so)
 * never links, loads or calls it and has no CPU path of its own.
 *
 * Every function restates one reference routine (file:line cited at the definition) with the
 * reference's own data layout (rank_support_v5 directory, select_support_mcl samples, rrr arrays,
 * byte tree, byte alphabet) — NOT the device layout — so that it checks the HIP path
 * independently.  Pinning: `serialize` functions emit SDSL's byte format and are compared
 * byte-for-byte with files written by the real library (tests/golden/, oracle/_ref); the
 * query functions are compared with the real library's answers on the same inputs.
 */
#ifndef SDSL_ORACLE_H
#define SDSL_ORACLE_H
#include <stddef.h>
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

/* ---- bits.hpp ------------------------------------------------------------------------- */
uint32_t orc_cnt(uint64_t x);             /* bits::cnt  bits.hpp:486-502 */
uint32_t orc_sel(uint64_t x, uint32_t i); /* bits::sel  bits.hpp:586-612 (1-based i) */
uint32_t orc_hi(uint64_t x);              /* bits::hi   bits.hpp:653-684 (hi(0)=0) */
uint64_t orc_read_int(const uint64_t * d, uint64_t pos, uint8_t len); /* bits::read_int bits.hpp:777-790 */

/* std::mt19937_64 (the generator behind util::set_random_bits, util.hpp:467-485) */
void orc_set_random_bits(uint64_t * words, uint64_t n_bits, uint64_t seed);
void orc_mt19937_64_fill(uint64_t * out, uint64_t n, uint64_t seed);

/* growable byte sink used by all serialisers */
typedef struct
{
    uint8_t * p;
    size_t len, cap;
} orc_buf;
void orc_buf_free(orc_buf * b);

/* ---- rank_support_v5<b,1> --------------------------------------------------------------- */
typedef struct orc_rank_v5 orc_rank_v5;
orc_rank_v5 * orc_rank_v5_build(const uint64_t * words, uint64_t n_bits, int bit); /* rank_support_v5.hpp:68-124 */
void orc_rank_v5_free(orc_rank_v5 *);
uint64_t orc_rank_v5_rank(const orc_rank_v5 *, uint64_t idx); /* rank_support_v5.hpp:131-149 */
void orc_rank_v5_batch(const orc_rank_v5 *, const uint64_t * idx, uint64_t n, uint64_t * out);
void orc_rank_v5_batch_mt(const orc_rank_v5 *, const uint64_t * idx, uint64_t n, uint64_t * out, int threads);
size_t orc_rank_v5_serialize(const orc_rank_v5 *, orc_buf * out); /* rank_support_v5.hpp:160-167 */

/* ---- select_support_mcl<b,1> ------------------------------------------------------------ */
typedef struct orc_select_mcl orc_select_mcl;
orc_select_mcl * orc_select_mcl_build(const uint64_t * words, uint64_t n_bits, int bit); /* select_support_mcl.hpp:121-128 */
void orc_select_mcl_free(orc_select_mcl *);
uint64_t orc_select_mcl_arg_cnt(const orc_select_mcl *);
uint64_t orc_select_mcl_select(const orc_select_mcl *, uint64_t i); /* select_support_mcl.hpp:384-439 */
void orc_select_mcl_batch(const orc_select_mcl *, const uint64_t * i, uint64_t n, uint64_t * out);
size_t orc_select_mcl_serialize(const orc_select_mcl *, orc_buf * out); /* select_support_mcl.hpp:474-518 */

/* ---- rrr_vector<63, int_vector<>, 32> --------------------------------------------------- */
/* ---- sd_vector<> ------------------------------------------------------------------------- */
typedef struct orc_sd orc_sd;
orc_sd * orc_sd_build(const uint64_t * words, uint64_t n_bits);              /* sd_vector.hpp:217-257 */
orc_sd * orc_sd_build_from_positions(const uint64_t * pos, uint64_t m);      /* sd_vector.hpp:259-305 */
void orc_sd_free(orc_sd *);
uint64_t orc_sd_size(const orc_sd *);
uint64_t orc_sd_ones(const orc_sd *);
uint32_t orc_sd_wl(const orc_sd *);
int orc_sd_access(const orc_sd *, uint64_t i);                /* sd_vector.hpp:328-349 */
uint64_t orc_sd_rank(const orc_sd *, uint64_t i, int bit);    /* sd_vector.hpp:553-575 */
uint64_t orc_sd_select(const orc_sd *, uint64_t i, int bit);  /* sd_vector.hpp:621-664 */
size_t orc_sd_serialize(const orc_sd *, orc_buf * out);       /* sd_vector.hpp:435-445 */

typedef struct orc_rrr orc_rrr;
orc_rrr * orc_rrr_build(const uint64_t * words, uint64_t n_bits); /* rrr_vector.hpp:158-270 */
void orc_rrr_free(orc_rrr *);
uint64_t orc_rrr_size(const orc_rrr *);
uint64_t orc_rrr_rank(const orc_rrr *, uint64_t i, int bit);    /* rrr_vector.hpp:503-544 */
uint64_t orc_rrr_select(const orc_rrr *, uint64_t i, int bit);  /* rrr_vector.hpp:639-726 */
int orc_rrr_access(const orc_rrr *, uint64_t i);                /* rrr_vector.hpp:276-298 */
void orc_rrr_rank_batch(const orc_rrr *, int bit, const uint64_t * i, uint64_t n, uint64_t * out);
void orc_rrr_select_batch(const orc_rrr *, int bit, const uint64_t * i, uint64_t n, uint64_t * out);
size_t orc_rrr_serialize(const orc_rrr *, orc_buf * out); /* rrr_vector.hpp:366-378 */

/* ---- wt_huff<bit_vector, rank_support_v5<>, select_support_mcl<1>, select_support_mcl<0>> */
typedef struct orc_wt orc_wt;
orc_wt * orc_wt_build(const uint8_t * text, uint64_t n); /* wt_pc.hpp:194-248, wt_huff.hpp:83-115, wt_helper.hpp:230-327 */
void orc_wt_free(orc_wt *);
uint64_t orc_wt_size(const orc_wt *);
uint64_t orc_wt_sigma(const orc_wt *);
uint64_t orc_wt_bv_size(const orc_wt *);
const uint64_t * orc_wt_bv_words(const orc_wt *);
uint64_t orc_wt_rank(const orc_wt *, uint64_t i, uint8_t c);   /* wt_pc.hpp:371-399 */
uint8_t orc_wt_access(const orc_wt *, uint64_t i);             /* wt_pc.hpp:336-357 */
uint64_t orc_wt_inverse_select(const orc_wt *, uint64_t i, uint8_t * c_out); /* wt_pc.hpp:411-430 */
uint64_t orc_wt_select(const orc_wt *, uint64_t i, uint8_t c); /* wt_pc.hpp:443-474 */
void orc_wt_rank_batch(const orc_wt *, const uint64_t * i, const uint8_t * c, uint64_t n, uint64_t * out);
/* select_is_mcl: serialise bv_select1/0 as select_support_mcl (1) or select_support_scan (0, zero bytes) */
size_t orc_wt_serialize(const orc_wt *, int select_is_mcl, orc_buf * out); /* wt_pc.hpp:713-726 */
void orc_wt_code_lengths(const orc_wt *, uint8_t len_out[256]);

/* ---- csa_wt<wt_huff<...>> restricted to backward_search / count --------------------------- */
typedef struct orc_csa orc_csa;
/* text must not contain 0 bytes (construct.hpp:41); builds SA of text+'\0' by prefix doubling */
orc_csa * orc_csa_build(const uint8_t * text, uint64_t n_text);
orc_csa * orc_csa_build_ex(const uint8_t * text, uint64_t n_text, uint64_t sa_dens, uint64_t isa_dens);
orc_csa * orc_csa_build_from_bwt(const uint8_t * bwt, uint64_t n); /* csa_wt.hpp:323-355 (no samples: count only) */
/* the rest of the csa_wt API; needs a csa built from text (samples) */
uint64_t orc_csa_lf(const orc_csa *, uint64_t i);  /* suffix_array_helper.hpp:346-360 */
uint64_t orc_csa_psi(const orc_csa *, uint64_t i); /* suffix_array_helper.hpp:330-342 */
uint64_t orc_csa_sa(const orc_csa *, uint64_t i);  /* csa_wt.hpp:363-381 */
uint64_t orc_csa_isa(const orc_csa *, uint64_t i); /* suffix_array_helper.hpp:519-537 */
uint64_t orc_csa_extract(const orc_csa *, uint64_t begin, uint64_t end, uint8_t * text); /* suffix_array_algorithm.hpp:578-600 */
uint64_t orc_csa_locate(const orc_csa *, const uint8_t * pat, uint64_t m, uint64_t * out, uint64_t cap); /* :505-523 */
void orc_csa_free(orc_csa *);
uint64_t orc_csa_size(const orc_csa *);
uint64_t orc_csa_sigma(const orc_csa *);
const uint8_t * orc_csa_bwt(const orc_csa *);
const orc_wt * orc_csa_wt(const orc_csa *);
void orc_csa_alphabet(const orc_csa *, uint8_t char2comp[256], uint64_t C[257]);
/* backward_search(csa,l,r,c) suffix_array_algorithm.hpp:167-201; returns interval size */
uint64_t orc_csa_backward_search_char(const orc_csa *, uint64_t l, uint64_t r, uint8_t c, uint64_t * l_res,
                                      uint64_t * r_res);
/* backward_search(csa,0,size-1,begin,end) :228-248 + count :464-471 */
uint64_t orc_csa_count(const orc_csa *, const uint8_t * pat, uint64_t m);
uint64_t orc_csa_interval(const orc_csa *, const uint8_t * pat, uint64_t m, uint64_t * l_res, uint64_t * r_res);
void orc_csa_count_batch(const orc_csa *, const uint8_t * pats, uint32_t m, uint64_t n_pat, uint64_t * out);
/* serialises wt + EMPTY sa/isa samples + alphabet is not SDSL-compatible; instead the pieces: */
size_t orc_csa_serialize_alphabet(const orc_csa *, orc_buf * out); /* csa_alphabet_strategy.hpp:258-268 */

#ifdef __cplusplus
}
#endif
#endif
