/* sdsl_hip.h — C ABI of the MI355X (gfx950) batched rank/select + wavelet-tree query engine.
 *
 * This is the drop-in boundary for SDSL's rank/select/wt/count hot path.
 * SDSL has no FFI layer: its boundary is a set of C++ concepts (`rank(i)`, `select(i)`,
 * `wt.rank(i,c)`, `count(csa,begin,end)`), all scalar.  Every entry point below is the
 * *batched* form of one of those members; the header-only adaptors in
 * include/sdsl_hip/adaptors.hpp put them back behind SDSL's operator()/rank/select names.
 *
 * Conventions
 *  - plain C, opaque handles, status codes (no exceptions cross the ABI);
 *  - every array argument may be a HOST or a DEVICE pointer (detected with
 *    hipPointerGetAttributes).  Device pointers: the call is asynchronous on `stream`
 *    (a hipStream_t passed as void*; NULL = the null stream).  Host pointers: the library
 *    stages through device memory and returns after the results are back on the host;
 *  - all integers are unsigned 64-bit, little endian, exactly SDSL's size_type;
 *  - out-of-range arguments are undefined behaviour in SDSL (asserts only,
 *    rank_support_v5.hpp:133-134, select_support_mcl.hpp:386, wt_pc.hpp:373).  Here they
 *    are *defined*: the result slot is set to SDSL_HIP_NPOS (all ones).  The one overflow
 *    SDSL does define — select_support_rrr returns size() (rrr_vector.hpp:641-642,686-689)
 *    — is reproduced exactly;
 *  - large batches with ALL arrays in host memory (>= 2^23 queries: rank / select of the bit-vector family,
 *    sdsl_hip_wt_rank_batch; >= 2^22 patterns: sdsl_hip_fm_count_batch) are cut
 *    into chunks that travel on two internal streams, so uploads, kernels and downloads overlap; `stream` is then
 *    only a placeholder and the call returns when all results are in place;
 *  - threading: like SDSL's, query calls are const on the handle and may run concurrently from
 *    several host threads (the error text of sdsl_hip_last_error() is per thread).  Calls that change a handle —
 *    *_destroy, sdsl_hip_fm_drop_sa, sdsl_hip_fm_set_jump_depth, and the first ISA / extract call on an index that
 *    still holds its whole suffix array (it materialises the ISA samples) — must not overlap other calls on that handle;
 *  - serialisers follow a size-query / fill protocol (buf == NULL returns the size in *written): the size query keeps
 *    the stream it built, per calling thread, and the fill call that follows with the same handle and arguments takes
 *    it — the stream of a large index is built once, not twice;
 *  - there is NO CPU fallback: without a usable gfx950 device every create call fails with
 *    SDSL_HIP_ERR_NO_DEVICE.
 */
#ifndef SDSL_HIP_H
#define SDSL_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define SDSL_HIP_VERSION_MAJOR 0
#define SDSL_HIP_VERSION_MINOR 1

typedef int32_t sdsl_hip_status;
enum {
    SDSL_HIP_OK = 0,
    SDSL_HIP_ERR_INVALID = -1,   /* bad argument (null handle, bit not in {0,1}, ...) */
    SDSL_HIP_ERR_NOMEM = -2,     /* host or device allocation failed */
    SDSL_HIP_ERR_HIP = -3,       /* HIP runtime error, see sdsl_hip_last_error() */
    SDSL_HIP_ERR_FORMAT = -4,    /* malformed / truncated SDSL serialised stream */
    SDSL_HIP_ERR_NO_DEVICE = -5, /* no gfx950 device visible: the engine has no CPU path */
    SDSL_HIP_ERR_UNSUPPORTED = -6
};

#define SDSL_HIP_NPOS UINT64_C(0xFFFFFFFFFFFFFFFF)

/* build flags for sdsl_hip_bv_create */
#define SDSL_HIP_BV_SELECT1 1u /* build the select_1 sample directory */
#define SDSL_HIP_BV_SELECT0 2u /* build the select_0 sample directory */

/* build flags for sdsl_hip_wt_create_ex / sdsl_hip_fm_create_from_{text,bwt}_ex */
#define SDSL_HIP_WT_RRR63 1u /* store the wavelet tree's bit vector as rrr_vector<63>: wt_huff<rrr_vector<63>> */
#define SDSL_HIP_WT_BLCD 2u  /* balanced tree shape (wt_blcd, wt_blcd.hpp:50-127) instead of the Huffman shape */
#define SDSL_HIP_WT_HUTU 4u  /* Hu-Tucker shape (wt_hutu, wt_hutu.hpp:355-676): the optimal alphabetic code */

/* `layout` of a serialised wt_pc / csa_wt stream handed to *_create_from_sdsl */
#define SDSL_HIP_LAYOUT_BV_SCAN 0 /* wt_huff<bit_vector, rank_support_v5<>, select_support_scan<>, select_support_scan<0>> */
#define SDSL_HIP_LAYOUT_BV_MCL 1  /* wt_huff<bit_vector, rank_support_v5<>> (select_support_mcl<1>, <0>) */
#define SDSL_HIP_LAYOUT_RRR63 2   /* wt_huff<rrr_vector<63>> (rank/select_support_rrr serialise to nothing) */
#define SDSL_HIP_LAYOUT_BV_DEFAULT 3 /* wt_huff<> with its default arguments: rank_support_v<1>, select_support_mcl<1>, <0> */

typedef struct sdsl_hip_bv_s * sdsl_hip_bv_t;   /* bit_vector + rank_support_v5 + select_support_mcl */
typedef struct sdsl_hip_rrr_s * sdsl_hip_rrr_t; /* rrr_vector<63> + rank_support_rrr + select_support_rrr */
typedef struct sdsl_hip_wt_s * sdsl_hip_wt_t;   /* wt_huff<bit_vector, rank_support_v5<>> */
typedef struct sdsl_hip_fm_s * sdsl_hip_fm_t;   /* csa_wt<wt_huff<...>> restricted to count() */

/* ---- library ------------------------------------------------------------------------- */
const char * sdsl_hip_last_error(void); /* thread-local message of the last failing call */
/* device memory this process currently holds through the library: all live handles plus the staging of calls in flight
 * (the sum over handles is what the *_device_bytes calls report) */
uint64_t sdsl_hip_allocated_bytes(void);
const char * sdsl_hip_version(void);
int32_t sdsl_hip_device_count(void);    /* number of visible gfx950 devices (0 if none) */

/* ---- synthetic input helpers (host) ----------------------------------------------------
 * util::set_random_bits (util.hpp:467-485): words = successive std::mt19937_64(seed) outputs.
 * Only the host container is touched; like SDSL the last word keeps its stray high bits. */
sdsl_hip_status sdsl_hip_util_set_random_bits(uint64_t * words, uint64_t n_bits, uint64_t seed);
/* The benchmark inputs of SURVEY.md 8(d), integer-only and seeded, so that the build container (where the real
 * sdsl-lite answers them once: tests/golden/make_golden_large.py) and the GPU box hold identical bytes.
 *  rnd_positions: out[q] = add + (q-th output of std::mt19937_64(seed)) % mod   (mod 0: the raw outputs) — the way
 *                 util::rnd_positions draws benchmark arguments (util.hpp:438-448)
 *  density_bits:  bit i = (i-th output of mt19937_64(seed) % 100 < percent), drawn sequentially (configs[2]); with a
 *                 table of generator states taken every `stride` draws (313 words each: sdsl_hip_util_mt_checkpoints;
 *                 committed as tests/golden/mt9_checkpoints.bin) every host thread produces its own stretch
 *  english_text:  English-class stand-in for Pizza&Chili english (not available offline,
 *                 benchmark/indexing_count/test_case.config:6): sigma > 200, H0 about 4.6 bits, no zero byte */
sdsl_hip_status sdsl_hip_util_rnd_positions(uint64_t seed, uint64_t count, uint64_t mod, uint64_t add, uint64_t * out);
sdsl_hip_status sdsl_hip_util_mt_checkpoints(uint64_t seed, uint64_t stride, uint64_t n, uint64_t * out);
sdsl_hip_status sdsl_hip_util_density_bits(uint64_t * words, uint64_t n_bits, uint64_t seed, uint32_t percent,
                                           const uint64_t * checkpoints, uint64_t n_checkpoints, uint64_t stride);
sdsl_hip_status sdsl_hip_util_english_text(uint8_t * out, uint64_t n_bytes, uint64_t seed);
/* The same text with `percent` (<= 95) of its 64 KiB blocks replaced by rotated copies of earlier blocks: the duplicated passages of a
 * real collection (english.1GB concatenates Gutenberg books) that keep the suffix-array interval of a pattern drawn from the text
 * wide for many characters (benchmark/indexing_count/src/genpatterns.c:183-203 draws its patterns that way).  percent 0 = english_text. */
sdsl_hip_status sdsl_hip_util_english_text_repetitive(uint8_t * out, uint64_t n_bytes, uint64_t seed, uint32_t percent);
/* The same stream as rnd_positions, written straight into DEVICE memory: `checkpoints` (host) = the generator's state before
 * every `stride`-th draw (sdsl_hip_util_mt_checkpoints, 313 words each), one block per checkpoint regenerates its stretch.
 * For processes that must not hold the stream on the host (one rank per GPU: eight times 8 GB).  Synchronises `stream`. */
sdsl_hip_status sdsl_hip_util_rnd_positions_device(const uint64_t * checkpoints, uint64_t n_checkpoints, uint64_t stride,
                                                   uint64_t count, uint64_t mod, uint64_t add, uint64_t * d_out, int32_t device,
                                                   void * stream);

/* ---- plain bit vector: rank_support_v5 / select_support_mcl ---------------------------
 * Replaces: rank_support_v5<b>::rank / operator() (rank_support_v5.hpp:131-154),
 *           select_support_mcl<b>::select / operator() (select_support_mcl.hpp:384-445),
 *           their constructors from `bit_vector const*` (rank_support_v5.hpp:68-124,
 *           select_support_mcl.hpp:121-128).
 * `words` = bit_vector::data() (int_vector.hpp:619-630): ceil(n_bits/64) u64, bit i is
 * (words[i>>6]>>(i&63))&1.  Bits at positions >= n_bits in the last word are ignored.
 * The data is COPIED into the device layout; the caller keeps ownership of `words`. */
sdsl_hip_status sdsl_hip_bv_create(const uint64_t * words, uint64_t n_bits, int32_t device, uint32_t flags,
                                   sdsl_hip_bv_t * out);
/* Two-bit patterns: rank_support_v / rank_support_v5 <10,2> <01,2> <00,2> <11,2> (rank_support.hpp:160-284) and
 * select_support_mcl for the same patterns (select_support.hpp:206-409).  The handle holds the pattern's occurrence
 * vector (bit i set iff the pattern ends at position i, SDSL's carry convention in front of position 0), so
 *   rank_support_v5<10,2>::rank(idx)      == sdsl_hip_bv_rank_batch(h, 1, ...)
 *   select_support_mcl<10,2>::select(i)   == sdsl_hip_bv_select_batch(h, 1, ...)   (flags: SDSL_HIP_BV_SELECT1)
 * (t_b, t_pat_len) exactly as SDSL's template arguments: (10,2) (01 = 1,2) (00 = 0,2) (11,2); (0,1) and (1,1) fall
 * through to sdsl_hip_bv_create. */
sdsl_hip_status sdsl_hip_bv_create_pattern(const uint64_t * words, uint64_t n_bits, int32_t device, uint32_t t_b,
                                           uint32_t t_pat_len, uint32_t flags, sdsl_hip_bv_t * out);
/* Builds the select directories named in `flags` (SDSL_HIP_BV_SELECT1 / SELECT0) if the handle does not have them yet: what
 * lets rank and select supports of one bit_vector share one device replica (select_support_mcl.hpp:95-117 holds no copy of
 * the vector either; rank_support.hpp:33, select_support.hpp:35). */
sdsl_hip_status sdsl_hip_bv_add_select(sdsl_hip_bv_t bv, uint32_t flags);
/* The bytes SDSL's own serialize() writes for the vector and its supports (bit_vector: int_vector.hpp:1978-2004;
 * rank_support_v5: rank_support_v5.hpp:160-167; rank_support_v: rank_support_v.hpp:156-163; select_support_mcl:
 * select_support_mcl.hpp:474-518, including what its two construction paths leave behind).  For a pattern handle
 * (sdsl_hip_bv_create_pattern) they describe the occurrence vector.  buf == NULL queries the size. */
#define SDSL_HIP_SER_BIT_VECTOR 0
#define SDSL_HIP_SER_RANK_V5_1 1
#define SDSL_HIP_SER_RANK_V5_0 2
#define SDSL_HIP_SER_SELECT_MCL_1 3
#define SDSL_HIP_SER_SELECT_MCL_0 4
#define SDSL_HIP_SER_RANK_V_1 5
#define SDSL_HIP_SER_RANK_V_0 6
/* Sibling representations of bit_vectors.hpp, handed over as the bytes of their own serialize() and decoded to plain
 * bits on the device (their rank / select answers are those of the plain vector): kind SDSL_HIP_SIBLING_IL =
 * bit_vector_il<t_bs> (bit_vector_il.hpp:212-224; any block size), SDSL_HIP_SIBLING_RRR15 / SDSL_HIP_SIBLING_RRR(t_bs, t_k)
 * = rrr_vector<...> (rrr_vector_15.hpp:409-420; rrr_vector.hpp:366-378).  flags as sdsl_hip_bv_create. */
#define SDSL_HIP_SIBLING_IL 0
#define SDSL_HIP_SIBLING_RRR15 1 /* the rrr_vector<15> SPECIALISATION (only with #include <sdsl/rrr_vector_15.hpp>) */
/* the generic rrr_vector<t_bs, int_vector<>, t_k>, 2 <= t_bs <= 63 (what rrr_vector<15> is without that include) */
#define SDSL_HIP_SIBLING_RRR(t_bs, t_k) (2 | ((int32_t)(t_bs) << 8) | ((int32_t)(t_k) << 16))
sdsl_hip_status sdsl_hip_bv_create_from_sdsl(const void * bytes, size_t len, int32_t kind, int32_t device, uint32_t flags,
                                             sdsl_hip_bv_t * out);
sdsl_hip_status sdsl_hip_bv_serialize(sdsl_hip_bv_t bv, int32_t what, void * buf, size_t cap, size_t * written);
sdsl_hip_status sdsl_hip_bv_destroy(sdsl_hip_bv_t bv);
/* ONE query, value in, value out: what = 0 rank_<bit>(arg), 1 select_<bit>(arg).  The argument and the answer travel through a
 * mapped pinned mailbox, so the call costs one kernel launch and one stream synchronisation (about 14 microseconds) — correct, but
 * never the right thing in a loop.  The C++ adaptors do NOT answer their scalar operator() with it (they forward to the caller's own
 * SDSL object); it is what their rank_on_device / select_on_device members call, for checking the device image. */
sdsl_hip_status sdsl_hip_bv_query_one(sdsl_hip_bv_t bv, int32_t what, int32_t bit, uint64_t arg, uint64_t * out);
/* The bucketed batch paths (large rank / select batches on plain and rrr vectors) work in ONE scratch pool per device, shared by
 * every handle on it and kept between calls: 12 bytes per query of the largest pass so far (at most 2^30 queries' worth) plus
 * about 70 MB of tables.  sdsl_hip_bv_release_scratch frees the pool of the handle's device (after its last user has finished);
 * the next large batch allocates it again.  sdsl_hip_device_scratch_bytes reports its size (the *_device_bytes calls do not
 * include it). */
sdsl_hip_status sdsl_hip_bv_release_scratch(sdsl_hip_bv_t bv);
uint64_t sdsl_hip_device_scratch_bytes(int32_t device);
/* Stream capture.  A batch call with a hipStream_t only enqueues, so it can be captured into a HIP graph — but a graph bakes in the
 * addresses it was captured with and its replays are ordered with nothing the library sees.  A large batch enqueued WHILE ITS STREAM
 * IS BEING CAPTURED therefore never touches the device's shared pool: it works in a scratch area owned by the handle, which
 * sdsl_hip_bv_reserve_capture_scratch / sdsl_hip_rrr_reserve_capture_scratch allocate beforehand for batches of up to max_queries
 * (they also build the bucket plans of the handle's select directories and the spread sample's verdict word — nothing may be
 * allocated or built during a capture).  Without a reservation of sufficient size a captured batch takes the direct kernel:
 * same answers, the speed of small batches.  Replays of graphs captured from ONE handle must not overlap each other (they share
 * that handle's area); they may overlap anything else, including large batches of other handles and un-captured batches of the
 * same handle.  max_queries = 0 releases the area; so does destroying the handle — graphs captured from it are invalid
 * afterwards (as they are after sdsl_hip_bv_release_scratch, which frees the verdict word with the pool). */
sdsl_hip_status sdsl_hip_bv_reserve_capture_scratch(sdsl_hip_bv_t bv, uint64_t max_queries);
uint64_t sdsl_hip_bv_size(sdsl_hip_bv_t bv);         /* bit_vector::size() */
uint64_t sdsl_hip_bv_ones(sdsl_hip_bv_t bv);         /* == rank_1(size()) */
uint64_t sdsl_hip_bv_device_bytes(sdsl_hip_bv_t bv); /* HBM footprint of the device layout */
/* out[q] = number of `bit`-bits in [0, idx[q]), idx[q] in [0, size()] */
sdsl_hip_status sdsl_hip_bv_rank_batch(sdsl_hip_bv_t bv, int32_t bit, const uint64_t * idx, uint64_t n,
                                       uint64_t * out, void * stream);
/* Measurement aid, not a query: the memory-access skeleton of sdsl_hip_bv_rank_batch with the arithmetic removed (for
 * every position: fetch its 64-byte rank line, write one word).  Device arrays only.  Its rate on a given vector is the
 * ceiling the memory system sets for batched rank there; bench.py reports it next to the real kernel's rate. */
sdsl_hip_status sdsl_hip_bv_gather_probe(sdsl_hip_bv_t bv, const uint64_t * d_idx, uint64_t n, uint64_t * d_out,
                                         void * stream);
/* out[q] = position of the i[q]-th `bit`-bit, i[q] in [1, #bit-bits] (1-based like SDSL) */
sdsl_hip_status sdsl_hip_bv_select_batch(sdsl_hip_bv_t bv, int32_t bit, const uint64_t * i, uint64_t n,
                                         uint64_t * out, void * stream);
/* out[q] = bit idx[q]  (bit_vector::operator[], int_vector.hpp:1900-1904), as 0/1 bytes */
sdsl_hip_status sdsl_hip_bv_access_batch(sdsl_hip_bv_t bv, const uint64_t * idx, uint64_t n, uint8_t * out,
                                         void * stream);
/* Writes the bit vector back as SDSL words (inverse of create; used by round-trip tests). */
sdsl_hip_status sdsl_hip_bv_export_words(sdsl_hip_bv_t bv, uint64_t * words_out, void * stream);

/* ---- rrr_vector<63, int_vector<>, 32> --------------------------------------------------
 * Replaces: rrr_vector(bit_vector const&) (rrr_vector.hpp:158-270),
 *           rank_support_rrr<b,63>::rank (rrr_vector.hpp:503-544),
 *           select_support_rrr<b,63>::select (rrr_vector.hpp:639-726),
 *           rrr_vector::operator[] (rrr_vector.hpp:276-298). */
sdsl_hip_status sdsl_hip_rrr_create(const uint64_t * words, uint64_t n_bits, int32_t device, sdsl_hip_rrr_t * out);
/* from the bytes written by rrr_vector<63>::serialize (rrr_vector.hpp:366-378) */
sdsl_hip_status sdsl_hip_rrr_create_from_sdsl(const void * bytes, size_t len, int32_t device, sdsl_hip_rrr_t * out);
/* A sibling representation kept COMPRESSED on the device: the serialize() bytes of bit_vector_il<t_bs> (bit_vector_il.hpp:212-224),
 * of the rrr_vector<15> specialisation (rrr_vector_15.hpp:409-420) or of any rrr_vector<t_bs, int_vector<>, t_k>
 * (rrr_vector.hpp:366-378) — kind as for sdsl_hip_bv_create_from_sdsl — are decoded on the device and re-encoded at once as this
 * library's rrr_vector<63> records (sdsl_hip_bv_create_from_sdsl keeps such a vector as plain rank lines instead: 1.14 + 0.13 bits
 * per bit whatever its density).  rank / select / access answers are those of the source vector (rank_support_rrr / select_support_rrr:
 * rrr_vector.hpp:503-544, 639-726; rrr_vector_15.hpp:135-300 — the same numbers for the same bits); sdsl_hip_rrr_serialize writes
 * rrr_vector<63>'s stream, not the source type's. */
sdsl_hip_status sdsl_hip_rrr_create_from_sibling(const void * bytes, size_t len, int32_t kind, int32_t device, sdsl_hip_rrr_t * out);
/* Writes exactly the bytes rrr_vector<63>::serialize would write for the same bit vector (rrr_vector.hpp:366-378):
 * size, bt (with SDSL's superblock inversion, :203-228), btnr, btnrp, rank samples, invert.  buf == NULL queries the
 * size.  An index encoded on the GPU can thus be handed to unmodified SDSL code (load / load_from_file). */
sdsl_hip_status sdsl_hip_rrr_serialize(sdsl_hip_rrr_t v, void * buf, size_t cap, size_t * written);
sdsl_hip_status sdsl_hip_rrr_destroy(sdsl_hip_rrr_t v);
sdsl_hip_status sdsl_hip_rrr_reserve_capture_scratch(sdsl_hip_rrr_t v, uint64_t max_queries); /* see sdsl_hip_bv_reserve_capture_scratch */
uint64_t sdsl_hip_rrr_size(sdsl_hip_rrr_t v);
uint64_t sdsl_hip_rrr_ones(sdsl_hip_rrr_t v);
uint64_t sdsl_hip_rrr_device_bytes(sdsl_hip_rrr_t v);
sdsl_hip_status sdsl_hip_rrr_rank_batch(sdsl_hip_rrr_t v, int32_t bit, const uint64_t * idx, uint64_t n,
                                        uint64_t * out, void * stream);
/* i[q] > #bit-bits  ->  out[q] = size()   (SDSL's defined overflow, rrr_vector.hpp:641-642) */
sdsl_hip_status sdsl_hip_rrr_select_batch(sdsl_hip_rrr_t v, int32_t bit, const uint64_t * i, uint64_t n,
                                          uint64_t * out, void * stream);
sdsl_hip_status sdsl_hip_rrr_access_batch(sdsl_hip_rrr_t v, const uint64_t * idx, uint64_t n, uint8_t * out,
                                          void * stream);
/* rrr_vector::get_int(idx, len) (rrr_vector.hpp:308-356): out[q] = the len <= 64 bits [idx[q], idx[q] + len), bit idx[q]
 * lowest; a window reaching beyond size() answers SDSL_HIP_NPOS (the reference asserts) */
sdsl_hip_status sdsl_hip_rrr_get_int_batch(sdsl_hip_rrr_t v, const uint64_t * idx, uint32_t len, uint64_t n, uint64_t * out,
                                           void * stream);

/* ---- wt_huff<bit_vector, rank_support_v5<>> over bytes ----------------------------------
 * Replaces: wt_pc(begin,end) (wt_pc.hpp:194-248) with the Huffman shape (wt_huff.hpp:83-115)
 *           and byte tree (wt_helper.hpp:230-327); wt_pc::rank (wt_pc.hpp:371-399);
 *           wt_pc::operator[] (wt_pc.hpp:336-357); wt_pc::inverse_select (wt_pc.hpp:411-430);
 *           wt_pc::select (wt_pc.hpp:443-474). */
sdsl_hip_status sdsl_hip_wt_create(const uint8_t * text, uint64_t n, int32_t device, sdsl_hip_wt_t * out);
/* flags: SDSL_HIP_WT_RRR63 -> wt_huff<rrr_vector<63>>;  SDSL_HIP_WT_BLCD -> wt_blcd<...>, SDSL_HIP_WT_HUTU -> wt_hutu<...>
 * (a shape flag may be combined with SDSL_HIP_WT_RRR63).
 * Trees of any wt_pc byte shape (wt_huff, wt_blcd, wt_hutu) LOAD through sdsl_hip_wt_create_from_sdsl: the stream
 * carries the node table. */
sdsl_hip_status sdsl_hip_wt_create_ex(const uint8_t * text, uint64_t n, int32_t device, uint32_t flags,
                                      sdsl_hip_wt_t * out);
/* from wt_pc::serialize bytes (wt_pc.hpp:713-726).  `layout` names the serialised type (SDSL_HIP_LAYOUT_*): plain
 * bit_vector with select_support_scan (zero bytes; benchmark/indexing_count/index.config:8) or select_support_mcl
 * (the wt_huff<bit_vector, rank_support_v5<>> default), or rrr_vector<63>. */
sdsl_hip_status sdsl_hip_wt_create_from_sdsl(const void * bytes, size_t len, int32_t layout, int32_t device,
                                             sdsl_hip_wt_t * out, size_t * consumed);
/* Writes the bytes of wt_huff<bit_vector, rank_support_v5<>, select_support_scan<>, select_support_scan<0>>::serialize
 * (wt_pc.hpp:713-726: size, sigma, bv, rank_support_v5 directory, tree; scan supports serialise to nothing) — or, for a
 * tree created with SDSL_HIP_WT_RRR63, of wt_huff<rrr_vector<63>>::serialize — for a wavelet tree that was built on the
 * GPU.  buf == NULL queries the size. */
sdsl_hip_status sdsl_hip_wt_serialize(sdsl_hip_wt_t wt, void * buf, size_t cap, size_t * written);
/* the same with the flavour chosen: SDSL_HIP_LAYOUT_BV_SCAN, SDSL_HIP_LAYOUT_BV_MCL (rank_support_v5 + the two
 * select_support_mcl) or SDSL_HIP_LAYOUT_BV_DEFAULT (wt_huff<> / wt_blcd<> with SDSL's default arguments) */
sdsl_hip_status sdsl_hip_wt_serialize_ex(sdsl_hip_wt_t wt, int32_t layout, void * buf, size_t cap, size_t * written);
sdsl_hip_status sdsl_hip_wt_destroy(sdsl_hip_wt_t wt);
uint64_t sdsl_hip_wt_size(sdsl_hip_wt_t wt);     /* wt.size()  */
uint64_t sdsl_hip_wt_sigma(sdsl_hip_wt_t wt);    /* wt.sigma   */
uint64_t sdsl_hip_wt_bv_size(sdsl_hip_wt_t wt);  /* wt.bv.size() */
/* HBM held by the handle.  A plain (non-rrr) tree with fewer than 2^36 symbols keeps TWO layouts: SDSL's binary levels
 * (select, writers) and a fused four-levels-per-fetch layout that rank / operator[] / inverse_select — and through
 * them backward_search, count, csa[i], isa, extract, locate — and select walk (5.6 bits per symbol per
 * fused level, the levels being those of a 16-ary Huffman tree of its own).  SDSL_HIP_WT_FUSED=0 in the environment at
 * creation time leaves it out; answers are the same. */
uint64_t sdsl_hip_wt_device_bytes(sdsl_hip_wt_t wt);
/* Releases SDSL's binary tree levels (rank lines + both select directories: 1.27 bits per tree bit) of a plain tree that has its
 * fused layout: rank / access / inverse_select / select keep walking the fused lines (wt_huff<> of a 1 GiB English text: 1.65 GB ->
 * 0.94 GB resident, 1.17 x the reference's stream); sdsl_hip_wt_serialize derives the tree's bits from the fused lines into a buffer of the call
 * (same bytes; the handle is not touched), select on a tree without the fused directory rebuilds the levels for good.  Nothing may be in flight on
 * the handle.  (sdsl_hip_fm_set_footprint does this to an FM-index's tree.) */
sdsl_hip_status sdsl_hip_wt_release_binary_levels(sdsl_hip_wt_t wt);
/* sum over c of count(c) * code_length(c) / size() is what bench.py needs for the roofline */
sdsl_hip_status sdsl_hip_wt_code_lengths(sdsl_hip_wt_t wt, uint8_t len_out[256]);
/* fetches a rank / access / select of symbol c costs on the fused layout (its depth in the layout's own 16-ary tree);
 * all zero when the handle has no fused layout */
sdsl_hip_status sdsl_hip_wt_fused_steps(sdsl_hip_wt_t wt, uint8_t steps_out[256]);
/* the form of the fused lines this library was built with: tree levels per fetch (4; 3 in a SDSL_HIP_FUSED_K=3 build), positions per
 * 128-byte line (184; 256), lines per superblock (1024; 0 = none).  Diagnostic, for tests and the bench's traffic model. */
void sdsl_hip_wt_fused_geometry(uint32_t * levels, uint32_t * positions_per_line, uint32_t * lines_per_superblock);
/* out[q] = occurrences of c[q] in [0, i[q]),  i[q] in [0, size()]   (wt.rank(i,c)) */
sdsl_hip_status sdsl_hip_wt_rank_batch(sdsl_hip_wt_t wt, const uint64_t * i, const uint8_t * c, uint64_t n,
                                       uint64_t * out, void * stream);
/* out_c[q] = wt[i[q]]  (operator[]) */
sdsl_hip_status sdsl_hip_wt_access_batch(sdsl_hip_wt_t wt, const uint64_t * i, uint64_t n, uint8_t * out_c,
                                         void * stream);
/* (out_rank[q], out_c[q]) = wt.inverse_select(i[q]) */
sdsl_hip_status sdsl_hip_wt_inverse_select_batch(sdsl_hip_wt_t wt, const uint64_t * i, uint64_t n,
                                                 uint64_t * out_rank, uint8_t * out_c, void * stream);
/* out[q] = wt.select(i[q], c[q]),  i[q] in [1, rank(size(), c[q])] */
sdsl_hip_status sdsl_hip_wt_select_batch(sdsl_hip_wt_t wt, const uint64_t * i, const uint8_t * c, uint64_t n,
                                         uint64_t * out, void * stream);

/* ---- csa_wt<wt_huff<...>>: backward_search / count --------------------------------------
 * Replaces: backward_search(csa,l,r,c,..) (suffix_array_algorithm.hpp:167-201),
 *           backward_search(csa,l,r,begin,end,..) (:228-248), count (:464-471),
 *           csa_wt::rank_bwt (csa_wt.hpp:286-289), byte_alphabet C/char2comp
 *           (csa_alphabet_strategy.hpp:175-212).
 * create_from_bwt: `bwt` is the BWT of text+'\0' (n = text length + 1, exactly one 0 byte),
 * i.e. what construct_bwt (construct_bwt.hpp:38-80) hands to csa_wt (csa_wt.hpp:323-355).
 * create_from_text: builds the suffix array of text+'\0' on the device (prefix doubling) and
 * derives the BWT; `text` must not contain 0 bytes (construct.hpp:41 throws in that case).
 * Texts of fewer than 2^32 - 2 bytes keep the whole suffix array (4 bytes per symbol), the text and the k-mer table of
 * count() on the device (sdsl_hip_fm_drop_sa releases them).  Longer texts (below 2^40 bytes; the reference switches to its
 * 64-bit sorter the same way, construct_sa.hpp:120-153) are sorted with 64-bit suffixes — 40 bytes of working memory per
 * symbol — and keep the suffix array as 8 bytes per suffix, the text and a k-mer table with 40-bit intervals (k <= 6), plus SA /
 * ISA samples at csa_wt's default densities 32 / 64 (csa_wt.hpp:56), which are what sdsl_hip_fm_drop_sa leaves and what
 * sdsl_hip_fm_serialize writes; every query is answered (rank, LF, count and select on the fused 16-ary lines up to 2^36 symbols —
 * 64-bit superblock counts and select-directory entries from 2^32 symbols on, line arithmetic exact for every position — count of
 * large batches through the 40-bit variants of its kernels; from 2^36 symbols on the binary levels answer.  sdsl_hip_limit names
 * every gate, INTEGRATION.md 3b tabulates them). */
sdsl_hip_status sdsl_hip_fm_create_from_bwt(const uint8_t * bwt, uint64_t n, int32_t device, sdsl_hip_fm_t * out);
sdsl_hip_status sdsl_hip_fm_create_from_text(const uint8_t * text, uint64_t n_text, int32_t device,
                                             sdsl_hip_fm_t * out);
/* flags: SDSL_HIP_WT_RRR63 -> csa_wt<wt_huff<rrr_vector<63>>> (the compressed FM-index of SDSL's README) */
sdsl_hip_status sdsl_hip_fm_create_from_bwt_ex(const uint8_t * bwt, uint64_t n, int32_t device, uint32_t flags,
                                               sdsl_hip_fm_t * out);
sdsl_hip_status sdsl_hip_fm_create_from_text_ex(const uint8_t * text, uint64_t n_text, int32_t device, uint32_t flags,
                                                sdsl_hip_fm_t * out);
/* from csa_wt::serialize bytes (csa_wt.hpp:389-402); SA/ISA samples are skipped; `layout` as for the wavelet tree */
sdsl_hip_status sdsl_hip_fm_create_from_sdsl(const void * bytes, size_t len, int32_t layout, int32_t device,
                                             sdsl_hip_fm_t * out);
/* The same with the type's sampling densities (csa_wt's template arguments t_dens and t_inv_dens, csa_wt.hpp:56-64 —
 * they are not part of the stream): the SA samples (sa_order_sa_sampling, csa_sampling_strategy.hpp:72-135) and ISA
 * samples (isa_sampling, :735-806) are kept, so that SA / ISA access, locate and extract below work on a loaded index.
 * sa_dens == 0 or isa_dens == 0: as sdsl_hip_fm_create_from_sdsl. */
sdsl_hip_status sdsl_hip_fm_create_from_sdsl_ex(const void * bytes, size_t len, int32_t layout, uint32_t sa_dens,
                                                uint32_t isa_dens, int32_t device, sdsl_hip_fm_t * out);
/* An index created from text keeps its suffix array in HBM (4 bytes per suffix) so that it can be written out as a
 * complete SDSL csa_wt: sdsl_hip_fm_serialize produces the bytes of
 *   csa_wt<wt_huff<bit_vector, rank_support_v5<>, select_support_scan<>, select_support_scan<0>>, sa_dens, isa_dens>
 *   (or csa_wt<wt_huff<rrr_vector<63>>, sa_dens, isa_dens> for an index created with SDSL_HIP_WT_RRR63)
 * ::serialize (csa_wt.hpp:389-402: wavelet tree, SA samples every sa_dens-th suffix, ISA samples every isa_dens-th text
 * position — csa_sampling_strategy.hpp:97-114,755-777 — and the byte alphabet), i.e. the index type of the reference's
 * count benchmark (benchmark/indexing_count/index.config:8), loadable by unmodified SDSL for locate/extract.
 * sdsl_hip_fm_drop_sa releases the suffix array and keeps SDSL's default samples (SA every 32nd suffix, ISA every 64th
 * text position) unless the index already has samples.  buf == NULL queries the size. */
sdsl_hip_status sdsl_hip_fm_serialize(sdsl_hip_fm_t fm, uint32_t sa_dens, uint32_t isa_dens, void * buf, size_t cap,
                                      size_t * written);
/* the same with the flavour of the wavelet tree chosen: SDSL_HIP_LAYOUT_BV_SCAN (as above), SDSL_HIP_LAYOUT_BV_MCL
 * (csa_wt<wt_huff<bit_vector, rank_support_v5<>>, ...>) or SDSL_HIP_LAYOUT_BV_DEFAULT (csa_wt<> with SDSL's default
 * wavelet tree: rank_support_v, select_support_mcl); ignored for an index created with SDSL_HIP_WT_RRR63 */
sdsl_hip_status sdsl_hip_fm_serialize_ex(sdsl_hip_fm_t fm, int32_t layout, uint32_t sa_dens, uint32_t isa_dens, void * buf,
                                         size_t cap, size_t * written);
sdsl_hip_status sdsl_hip_fm_drop_sa(sdsl_hip_fm_t fm);
/* ... with the sampling densities of the caller's csa_wt type (template parameters t_dens / t_inv_dens, csa_wt.hpp:51-57): SA-order
 * SA samples every sa_dens-th suffix, text-order ISA samples every isa_dens-th position; 0 = what the index holds, else 32 / 64.  The
 * walks behind csa[i] / locate take sa_dens - 1 LF steps on average, extract isa_dens / 2 before its first byte: csa[i] runs 0.87 G/s
 * at 32 and 3.4 G/s at 8 on a 1 GiB text (both at the fabric's request ceiling), for 4 / sa_dens bytes per symbol.
 * sdsl_hip_fm_serialize(fm, sa_dens, isa_dens) then writes csa_wt<..., sa_dens, isa_dens> from the samples. */
sdsl_hip_status sdsl_hip_fm_drop_sa_ex(sdsl_hip_fm_t fm, uint32_t sa_dens, uint32_t isa_dens);
/* The opposite, for an index loaded from an SDSL stream WITH its sampling densities: the text is read back through the ISA
 * samples (suffix_array_algorithm.hpp:578-600, extract), suffix-sorted on the device, checked against the stream's own SA
 * samples and kept together with the whole suffix array — and the k-mer table of count() is built: from then on the index
 * answers like one created from text (count of large batches: the k-mer table + text comparison road).  About 1.5 s per GiB. */
sdsl_hip_status sdsl_hip_fm_restore_suffix_array(sdsl_hip_fm_t fm);
/* Jump-start table of count / interval / locate: the SA interval of every k-mer over the index's alphabet (16 bytes
 * each), read with the last k characters of a pattern instead of walking their k LF steps.  The entries are computed by
 * the search itself, so answers do not change.  Every index gets a default depth at creation (table <= half the wavelet
 * tree, >= 64 MiB allowed); k = 0 drops the table, a larger k trades HBM for speed (sigma^k entries). */
sdsl_hip_status sdsl_hip_fm_set_jump_depth(sdsl_hip_fm_t fm, uint32_t k);
uint32_t sdsl_hip_fm_jump_depth(sdsl_hip_fm_t fm);
/* The k-mer table of count(): the SA interval of every k-mer (k <= 8 bytes) that occurs in the text, in a hash table of 128-byte
 * buckets — ONE fetch replaces the first k backward-search steps (suffix_array_algorithm.hpp:228-248) of a pattern, and a pattern
 * whose last k bytes are not in the table cannot occur.  Built by default for an index created from text (which still holds its
 * suffix array and the text), as deep as fits into the wavelet tree's own size; sdsl_hip_fm_set_kmer_table rebuilds it with the
 * deepest k <= k_max whose table (32 bytes per distinct k-mer) stays within budget_bytes (k_max 0: release it).  The table
 * survives sdsl_hip_fm_drop_sa.  Answers never depend on it. */
sdsl_hip_status sdsl_hip_fm_set_kmer_table(sdsl_hip_fm_t fm, uint32_t k_max, uint64_t budget_bytes);
uint32_t sdsl_hip_fm_kmer_table_depth(sdsl_hip_fm_t fm);
uint64_t sdsl_hip_fm_kmer_table_bytes(sdsl_hip_fm_t fm);
/* The index at a chosen footprint.  The reference's csa_wt holds a wavelet tree, SA samples, ISA samples and the alphabet
 * (csa_wt.hpp:389-402; 0.93 bytes per symbol for csa_wt<wt_huff<>, 32, 64> on English text); an index created from text here
 * holds the whole suffix array, the text, both tree layouts and a k-mer table on top (8.3 bytes per symbol) because that is what
 * makes count() fastest.  sdsl_hip_fm_set_footprint(fm, max_bytes) gives HBM back until sdsl_hip_fm_device_bytes(fm) <= max_bytes,
 * in the order that costs count() least per byte: (1) SDSL's binary tree levels with their select directories (a serialize call derives
 * the tree's bits from the fused lines into a buffer of its own; select keeps the fused directory); (2) suffix array and text -> SDSL's default
 * samples SA 32 / ISA 64 (csa_wt.hpp:56), 32 bits each, the dense jump table cut to <= 4 MiB, the k-mer table rebuilt as deep as
 * the remaining budget allows.  Answers never change; csa[i] / locate / extract walk LF steps from the samples as the reference
 * does (csa_wt.hpp:363-381).  Floor: fused tree lines + samples + alphabet (about 0.9 bytes per symbol of English text, i.e.
 * the reference's own footprint); a budget below it is SDSL_HIP_ERR_INVALID and the message names the floor.  Plain tree with
 * the fused layout (an index of 2^32 symbols and more keeps 64-bit samples); nothing may be in flight on the handle.  One-way (restore_suffix_array brings
 * suffix array, text and the default table back). */
sdsl_hip_status sdsl_hip_fm_set_footprint(sdsl_hip_fm_t fm, uint64_t max_bytes);
/* resident bytes by part: [0] binary tree levels + select directories (or the rrr vector), [1] fused tree lines + directory,
 * [2] whole suffix array, [3] text, [4] SA + ISA samples, [5] k-mer table, [6] dense jump table, [7] alphabet / node tables */
void sdsl_hip_fm_footprint_parts(sdsl_hip_fm_t fm, uint64_t parts[8]);
sdsl_hip_status sdsl_hip_fm_destroy(sdsl_hip_fm_t fm);
uint64_t sdsl_hip_fm_size(sdsl_hip_fm_t fm);  /* csa.size() = text length + 1 */
uint64_t sdsl_hip_fm_sigma(sdsl_hip_fm_t fm); /* csa.sigma */
uint64_t sdsl_hip_fm_device_bytes(sdsl_hip_fm_t fm);
sdsl_hip_wt_t sdsl_hip_fm_wavelet_tree(sdsl_hip_fm_t fm); /* csa.wavelet_tree (borrowed handle) */
sdsl_hip_status sdsl_hip_fm_alphabet(sdsl_hip_fm_t fm, uint8_t char2comp_out[256], uint64_t C_out[257]);
/* one LF step per element: (l_out,r_out) = backward_search(csa, l, r, c); an empty result is
 * (l_out > r_out) exactly as SDSL leaves it */
sdsl_hip_status sdsl_hip_fm_backward_search_batch(sdsl_hip_fm_t fm, const uint64_t * l, const uint64_t * r,
                                                  const uint8_t * c, uint64_t n, uint64_t * l_out,
                                                  uint64_t * r_out, void * stream);
/* n_patterns fixed-length patterns, pattern p = patterns[p*m .. p*m+m);  out[p] = count(csa, pattern p) */
sdsl_hip_status sdsl_hip_fm_count_batch(sdsl_hip_fm_t fm, const uint8_t * patterns, uint32_t m,
                                        uint64_t n_patterns, uint64_t * out, void * stream);
/* ragged patterns: pattern p = bytes[offsets[p] .. offsets[p+1]);  offsets has n_patterns+1 entries */
sdsl_hip_status sdsl_hip_fm_count_ragged(sdsl_hip_fm_t fm, const uint8_t * bytes, const uint64_t * offsets,
                                         uint64_t n_patterns, uint64_t * out, void * stream);
/* same as count_batch but also returns the SA interval [l_out[p], r_out[p]] */
sdsl_hip_status sdsl_hip_fm_interval_batch(sdsl_hip_fm_t fm, const uint8_t * patterns, uint32_t m,
                                           uint64_t n_patterns, uint64_t * l_out, uint64_t * r_out,
                                           void * stream);

/* ---- csa_wt<wt_huff<...>>: SA / ISA / LF / psi access, extract, locate --------------------
 * Replaces: csa_wt::operator[] (csa_wt.hpp:363-381), csa.isa[i] (suffix_array_helper.hpp:519-537), csa.lf[i] (:346-360),
 *           csa.psi[i] (:330-342), extract(csa, begin, end, text) (suffix_array_algorithm.hpp:578-600),
 *           locate(csa, begin, end) (:505-523).
 * They need the whole suffix array (index created from text) or SA / ISA samples (create_from_sdsl_ex, or drop_sa);
 * otherwise SDSL_HIP_ERR_UNSUPPORTED.  Arguments outside [0, size()) give SDSL_HIP_NPOS. */
sdsl_hip_status sdsl_hip_fm_sampling(sdsl_hip_fm_t fm, uint32_t * sa_dens, uint32_t * isa_dens, int32_t * has_full_sa);
/* out[q] = csa[idx[q]] */
sdsl_hip_status sdsl_hip_fm_sa_batch(sdsl_hip_fm_t fm, const uint64_t * idx, uint64_t n, uint64_t * out, void * stream);
/* out[q] = csa.isa[idx[q]] */
sdsl_hip_status sdsl_hip_fm_isa_batch(sdsl_hip_fm_t fm, const uint64_t * idx, uint64_t n, uint64_t * out, void * stream);
/* out[q] = csa.lf[idx[q]] */
sdsl_hip_status sdsl_hip_fm_lf_batch(sdsl_hip_fm_t fm, const uint64_t * idx, uint64_t n, uint64_t * out, void * stream);
/* out[q] = csa.psi[idx[q]] */
sdsl_hip_status sdsl_hip_fm_psi_batch(sdsl_hip_fm_t fm, const uint64_t * idx, uint64_t n, uint64_t * out, void * stream);
/* Ragged answers follow one protocol: *total receives the number of output elements; out_offsets (optional, n+1
 * entries) receives where each query's answer starts; with out == NULL the call only sizes the answer, otherwise cap
 * (elements available in out) must be >= *total.  These calls synchronise the stream.
 * extract: text[begin[q] .. end[q]] inclusive (begin <= end < size(); other queries yield nothing) */
sdsl_hip_status sdsl_hip_fm_extract_batch(sdsl_hip_fm_t fm, const uint64_t * begin, const uint64_t * end, uint64_t n,
                                          uint64_t * out_offsets, uint8_t * out_text, uint64_t cap, uint64_t * total,
                                          void * stream);
/* csa[l[q]], ..., csa[r[q]] for every SA interval (an empty interval l > r yields nothing) */
sdsl_hip_status sdsl_hip_fm_sa_range_batch(sdsl_hip_fm_t fm, const uint64_t * l, const uint64_t * r, uint64_t n,
                                           uint64_t * out_offsets, uint64_t * out_pos, uint64_t cap, uint64_t * total,
                                           void * stream);
/* locate: all occurrences of every pattern, in SA order like the reference (= interval_batch + sa_range_batch) */
sdsl_hip_status sdsl_hip_fm_locate_batch(sdsl_hip_fm_t fm, const uint8_t * patterns, uint32_t m, uint64_t n_patterns,
                                         uint64_t * out_offsets, uint64_t * out_pos, uint64_t cap, uint64_t * total,
                                         void * stream);

/* ---- sd_vector<>: Elias-Fano coded sparse bit vector ---------------------------------------
 * Replaces: sd_vector<>::operator[] (sd_vector.hpp:328-349), rank_support_sd<b>::rank (:553-575),
 *           select_support_sd<b>::select (:621-664).
 * create: the ones of a plain bit vector (sd_vector(bit_vector const&), :217-257); create_from_positions: a strictly
 * increasing position list with an explicit size (the builder / iterator constructors, :259-324);
 * create_from_sdsl: sd_vector<>::serialize bytes (:435-445).  Out-of-domain arguments give SDSL_HIP_NPOS / 0xFF.
 * select_0 is served by an interpolated search over buckets over select_1 (O(log m) probes per query). */
typedef struct sdsl_hip_sd_s * sdsl_hip_sd_t;
sdsl_hip_status sdsl_hip_sd_create(const uint64_t * words, uint64_t n_bits, int32_t device, sdsl_hip_sd_t * out);
sdsl_hip_status sdsl_hip_sd_create_from_positions(const uint64_t * positions, uint64_t m, uint64_t n_bits, int32_t device,
                                                  sdsl_hip_sd_t * out);
sdsl_hip_status sdsl_hip_sd_create_from_sdsl(const void * bytes, size_t len, int32_t device, sdsl_hip_sd_t * out,
                                             size_t * consumed);
/* the bytes of sd_vector<>::serialize (size, wl, low, high and the two select_support_mcl of high): a vector built on
 * the device loads into unmodified SDSL.  buf == NULL queries the size. */
sdsl_hip_status sdsl_hip_sd_serialize(sdsl_hip_sd_t v, void * buf, size_t cap, size_t * written);
sdsl_hip_status sdsl_hip_sd_destroy(sdsl_hip_sd_t v);
uint64_t sdsl_hip_sd_size(sdsl_hip_sd_t v);
uint64_t sdsl_hip_sd_ones(sdsl_hip_sd_t v);
uint32_t sdsl_hip_sd_low_width(sdsl_hip_sd_t v); /* sd_vector::wl */
/* which of the vector's queries the one-lane-per-query kernels answer (decided when it is built, from a probe batch: bit 0 rank /
 * access, bit 1 select_0; 0: the quad kernels — small, empty or clustered vectors, or SDSL_HIP_SD_NO_LANES).  Answers do not depend
 * on it. */
uint32_t sdsl_hip_sd_lane_kernels(sdsl_hip_sd_t v);
uint64_t sdsl_hip_sd_device_bytes(sdsl_hip_sd_t v);
sdsl_hip_status sdsl_hip_sd_rank_batch(sdsl_hip_sd_t v, int32_t bit, const uint64_t * idx, uint64_t n, uint64_t * out,
                                       void * stream);
sdsl_hip_status sdsl_hip_sd_select_batch(sdsl_hip_sd_t v, int32_t bit, const uint64_t * i, uint64_t n, uint64_t * out,
                                         void * stream);
sdsl_hip_status sdsl_hip_sd_access_batch(sdsl_hip_sd_t v, const uint64_t * idx, uint64_t n, uint8_t * out, void * stream);

/* ---- several GPUs of one node, one process ---------------------------------
 * The reference has no multi-device layer; this is the batch ABI above over a GROUP of devices.  The index is
 * replicated (load time: one ncclBroadcast per device buffer), a batch owned by the root device devices[0] — its arrays
 * in that device's memory, or in host memory — is cut into one contiguous shard per device: scatter (grouped
 * ncclSend / ncclRecv, every peer on its own xGMI link), the single-GPU kernels, gather; `chunks` pieces are pipelined
 * over three streams per device (chunks <= 1: one piece).  Answers are those of the single-GPU calls, in the caller's
 * order.  RCCL (librccl.so) is loaded at sdsl_hip_group_create; SDSL_HIP_ERR_NO_DEVICE if it is missing. */
typedef struct sdsl_hip_group_s * sdsl_hip_group_t;
sdsl_hip_status sdsl_hip_group_create(const int32_t * devices, int32_t n, sdsl_hip_group_t * out);
sdsl_hip_status sdsl_hip_group_destroy(sdsl_hip_group_t g);
int32_t sdsl_hip_group_size(sdsl_hip_group_t g);
int32_t sdsl_hip_group_device(sdsl_hip_group_t g, int32_t r);
/* link check: every device sends `bytes` to the next one of the group (to itself in a group of one) through both
 * communicators; the data is verified; *ms_out (optional) = duration of the first round */
sdsl_hip_status sdsl_hip_group_loopback(sdsl_hip_group_t g, uint64_t bytes, float * ms_out);
/* replicas[0] = root (a handle on devices[0]); replicas[1..n) are new handles on the other devices, to be released with
 * sdsl_hip_bv_destroy */
sdsl_hip_status sdsl_hip_group_bv_replicate(sdsl_hip_group_t g, sdsl_hip_bv_t root, sdsl_hip_bv_t * replicas);
sdsl_hip_status sdsl_hip_group_bv_rank_batch(sdsl_hip_group_t g, const sdsl_hip_bv_t * replicas, int32_t bit, const uint64_t * idx,
                                             uint64_t n, uint64_t * out, int32_t chunks);
sdsl_hip_status sdsl_hip_group_bv_select_batch(sdsl_hip_group_t g, const sdsl_hip_bv_t * replicas, int32_t bit, const uint64_t * i,
                                               uint64_t n, uint64_t * out, int32_t chunks);
/* configs[4]: the text (host memory or devices[0]) reaches every device with one broadcast and every device lays out its
 * own csa_wt; flags as sdsl_hip_fm_create_from_text_ex; all n handles are new (sdsl_hip_fm_destroy) */
sdsl_hip_status sdsl_hip_group_fm_create_from_text(sdsl_hip_group_t g, const uint8_t * text, uint64_t n_text, uint32_t flags,
                                                   sdsl_hip_fm_t * replicas);
sdsl_hip_status sdsl_hip_group_fm_count_batch(sdsl_hip_group_t g, const sdsl_hip_fm_t * replicas, const uint8_t * patterns,
                                              uint32_t m, uint64_t n_patterns, uint64_t * out, int32_t chunks);

/* ---- measurement hooks -------------------------------------------------------------------
 * Duration (ms) of the most recent kernel launched by a *_batch call on this handle's device,
 * measured with hipEvents recorded on the launch stream.  Timing is off by default because
 * the event pair adds a few microseconds; bench.py turns it on. */
sdsl_hip_status sdsl_hip_set_timing(int32_t enabled);
/* Process-wide knobs.  "rank_sorted": how sdsl_hip_bv_rank_batch answers a batch in device memory: 0 = always the direct
 * kernel (one rank line per query), 1 = the bucketed path whenever the vector allows it (bv_sorted.hip: the batch is
 * partitioned by index slice, slices are staged in LDS; 12 bytes of device scratch per query, kept with the handle),
 * -1 = automatic (bucketed when the batch addresses every line of a vector larger than the caches several times AND a
 * sample of the batch is spread over the vector — a windowed or sorted batch stays on the direct kernel; reading that
 * sample back synchronises the stream once per call, so use 0 or 1 where a call must stay asynchronous, e.g. under
 * graph capture).
 * Answers are identical in every mode.  Initial value: environment variable SDSL_HIP_RANK_SORTED, else -1.
 * "select_sorted": the same for sdsl_hip_bv_select_batch (buckets of consecutive argument ranks, their lines staged in LDS;
 * automatic only for vectors without long sparse stretches: those are answered by a slow fix-up pass).  Initial value:
 * SDSL_HIP_SELECT_SORTED, else -1.
 * "rrr_raw_budget" (permille, default 20): how much of its compressed size an rrr_vector<63> built or loaded AFTER the call
 * may spend on storing blocks raw instead of as enumerative offsets.  Blocks of the classes 11..52 are always raw on the
 * device (decoding them costs 63 dependent steps); within the budget the classes next to them follow, largest first, so
 * that fewer and fewer decoder steps remain per block (0: none beyond 11..52; a wavelet tree over text: 20 -> classes 9..54
 * raw, count 1.2x; 60 -> 4..59 raw, count 1.4x at +0.6 % of the index).  Answers and the serialised SDSL bytes do not
 * depend on it.
 * "rrr_sparse_limit" (0..20, default 10): the largest class an rrr_vector<63> built or loaded AFTER the call may keep as an
 * enumerative offset (and, through the complement, the smallest: 63 - limit); "rrr_raw_budget" then works downwards from
 * there.  The default is about speed.  A vector of 10-30 % density consists of the classes 6..25: with the limit at 20 it
 * takes 1.1-1.2 times the space of SDSL's rrr_vector<63> instead of 1.3-1.4 times, a block costs up to eighteen bisections
 * to decode, and large batches stay on the direct kernels.  Answers and the serialised SDSL bytes do not depend on it
 * (rrr_vector.hpp:158-270 stores every class as an offset). */
sdsl_hip_status sdsl_hip_set_option(const char * name, int64_t value);
/* The library's size gates by name (INTEGRATION.md 5 tabulates them; the reference itself is 64-bit throughout and bounded by host
 * memory alone, int_vector.hpp / wt_pc.hpp:366-474).  Sizes from the returned value ON are refused or take the other road named in the
 * table: "bv_bits", "bv_bucketed_bits", "rrr_bits", "rrr_bucketed_bits", "wt_fused_symbols", "wt_select_bucketed_symbols",
 * "fm_fast_symbols", "sorter32_symbols", "sorter64_symbols", "step_table_lines".  0 = unknown name. */
uint64_t sdsl_hip_limit(const char * what);
/* "trace_phases" (0/1): the bucketed batch rank times each of its passes with HIP events on the launch stream (one host
 * synchronisation per call) and sdsl_hip_last_phases returns them as "select=0|1;hist1=ms;offs1=ms;part1=ms;..." for the
 * most recent bucketed batch since the option was last set (empty: no batch took that path) */
sdsl_hip_status sdsl_hip_last_phases(char * buf, size_t cap);
/* where a vector's device layout lives: out = { address of the rank lines, their size in bytes, address of the select_1
 * directory, address of the batch-rank scratch } — for allocation / alignment experiments (tools/alloc_sensitivity.py) */
sdsl_hip_status sdsl_hip_bv_layout_info(sdsl_hip_bv_t bv, uint64_t out[4]);
sdsl_hip_status sdsl_hip_last_kernel_ms(float * ms_out); /* synchronises on the stop event */

#ifdef __cplusplus
}
#endif
#endif /* SDSL_HIP_H */
