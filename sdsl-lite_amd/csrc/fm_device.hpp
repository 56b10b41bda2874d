// fm_device.hpp — byte alphabet tables of the FM-index (csa_alphabet_strategy.hpp:136-308: C, char2comp) as staged in
// LDS by the count kernels, shared by fm.hip (plain wavelet tree) and wt_rrr.hip (rrr-compressed wavelet tree).
#pragma once
#include "common.hpp"
#include "wt_device.hpp"

namespace sdslhip {

struct FmTables
{
    uint64_t C[257];        // C[cc] = number of symbols smaller than comp2char[cc]; C[sigma] = size
    uint8_t char2comp[256]; // 0 for absent bytes (and for the sentinel itself)
};

__device__ __forceinline__ void fm_stage_tables(FmTables * lds, const FmTables * g)
{
    const uint64_t * src = reinterpret_cast<const uint64_t *>(g);
    uint64_t * dst = reinterpret_cast<uint64_t *>(lds);
    for (unsigned i = threadIdx.x; i < sizeof(FmTables) / 8; i += blockDim.x)
        dst[i] = src[i];
    // callers follow with wt_stage_tables(), which ends in __syncthreads()
}

// Jump-start table of backward search: the SA interval after the LAST k characters of a pattern, for every k-mer over
// the compact alphabet (key = digits char2comp[p[m-1]], char2comp[p[m-2]], ... in base sigma, first digit most
// significant).  One 16-byte read replaces the first k LF steps — the expensive ones, whose two cascades still walk
// different lines.  The entries are what the search kernel itself computes for the k-mer, so every answer (including
// the (l, r) of an empty interval) is unchanged.
struct FmJump
{
    const uint64_t * tab; // (l, r) pairs, sigma^k of them; null = no table
    uint32_t k, sigma;
};

// the interval after the last J.k characters of pattern [begin, end), if the table applies; returns the new `it`
__device__ __forceinline__ uint64_t fm_jump_start(const FmJump & J, const FmTables & F, const uint8_t * __restrict__ pats,
                                                  uint64_t begin, uint64_t end, uint64_t & l, uint64_t & r)
{
    if (!J.tab || end - begin < J.k)
        return end;
    uint64_t key = 0;
    bool ok = true;
    for (uint32_t t = 0; t < J.k; ++t)
    {
        const unsigned c = pats[end - 1 - t];
        const unsigned cc = F.char2comp[c];
        ok = ok && !(cc == 0 && c > 0); // a character that does not occur: leave it to the search (it ends there)
        key = key * J.sigma + cc;
    }
    if (!ok)
        return end;
    const ulonglong2 e = *reinterpret_cast<const ulonglong2 *>(J.tab + 2 * key);
    l = e.x;
    r = e.y;
    return end - J.k;
}

// Per-byte tables of the flat count kernel (fm_count2.hip), 7 KiB of LDS: the path of every symbol through the fused wavelet
// tree, spelled out as the steps the search takes — no node-table descent, no 64-bit path word in the loop.
// count() hands a search over to the TEXT when its interval is down to a few suffixes: the remaining characters stand in front of each
// of them or not (fm_count2.hip: verify_pays; fm.hip: k_fm_verify; DESIGN: 04_wavelet_tree_fm_index.md, "count stops at a FEW suffixes").
// The pending word a search leaves: [1 | suffixes - 1 : 3 | characters left : 60 - LB | first suffix : LB], LB = 32 or 40.
#ifndef SDSL_HIP_FM_VERIFY_MAX
#define SDSL_HIP_FM_VERIFY_MAX 8
#endif
constexpr uint32_t kFmVerifyMax = SDSL_HIP_FM_VERIFY_MAX; // 1: round 3's form — one suffix (A/B)
static_assert(kFmVerifyMax >= 1 && kFmVerifyMax <= 8, "three bits hold suffixes - 1");
// The index is the BWT of text + sentinel, i.e. of a CYCLIC string: backward search lets a pattern run through the sentinel (count("b\\0a")
// is 1 on the text "ab": suffix_array_algorithm.hpp:167-201 does not care), the text buffer does not hold it.  A search may only be
// handed to the text if none of the characters it still has to match is the byte 0 (round 3's single-suffix shortcut missed this).  The
// test runs on the pattern's next sixteen bytes, which the flat kernels hold in registers; a search with more than sixteen characters to
// go walks on until it has that few.
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));

// the 16 bytes in front of byte offset `end` of the pattern array (as a little-endian 128-bit number: .w's top byte is the
// byte at end - 1); what lies in front of the array's first byte reads as 0
__device__ __forceinline__ u32x4 load_tail16(const uint8_t * __restrict__ pats, uint64_t end)
{
    u32x4 v;
    if (end >= 16)
        __builtin_memcpy(&v, pats + end - 16, 16);
    else
    {
        uint8_t b[16];
#pragma unroll
        for (int j = 0; j < 16; ++j)
            b[j] = (uint64_t)(15 - j) < end ? pats[end - 16 + j] : 0; // (only reached for the first pattern of a batch)
        __builtin_memcpy(&v, b, 16);
    }
    return v;
}

// does any of the TOP `valid` (<= 16) bytes of the 128-bit number w — the next `valid` characters a search will process — equal zero?
// (bytes below them belong to the pattern in front, or lie in front of the array: filled with ones so that no borrow reaches the test)
__device__ __forceinline__ bool fm_tail_has_zero(u32x4 w, uint32_t valid)
{
    uint32_t x[4] = {w.w, w.z, w.y, w.x}; // most significant word first
    bool z = false;
#pragma unroll
    for (int j = 0; j < 4; ++j)
    {
        const int v = (int)valid - 4 * j; // valid top bytes of word j
        const uint32_t fill = v >= 4 ? 0u : (v <= 0 ? 0xFFFFFFFFu : 0xFFFFFFFFu >> (8 * v));
        const uint32_t y = x[j] | fill;
        z = z || ((y - 0x01010101u) & ~y & 0x80808080u) != 0;
    }
    return z;
}
SH_HD uint64_t fm_pending_word(unsigned LB, uint64_t l, uint64_t suffixes, uint64_t rem)
{
    return (UINT64_C(1) << 63) | ((suffixes - 1) << 60) | (rem << LB) | l;
}
constexpr unsigned kFmMaxSteps = 1280;
// (an index of 2^32 symbols and more uses FmCountTabW, which begins with these fields)
struct FmCountTab
{
    uint32_t cb[256];            // by byte c: C[char2comp[c]] (32 bits: the fused layout is for fewer than 2^32 symbols)
    uint32_t meta[256];          // by byte c: first step | number of steps << 16; 0 = the byte does not occur
    uint32_t steps[kFmMaxSteps]; // first line of the fused node (28 bits) | slot << 28
};

// the same for 2^32 .. 2^39 symbols: the high words of C[], the fused node of every step and the layout's list of places where a count
// passes a multiple of 2^32 (wt_device.hpp: WtFusedTables::cross_*) — 11 KiB of LDS
struct FmCountTabW : FmCountTab
{
    uint32_t cbh[256];            // by byte c: C[char2comp[c]] >> 32
    uint16_t snode[kFmMaxSteps];  // fused node of step i
    uint32_t n_cross;
    uint16_t cross_key[kFusedMaxCross];
    uint64_t cross_pos[kFusedMaxCross];
};

// The k-mer table ("deep jump"): the SA interval [l, e) of every k-mer (k <= 8 bytes) that occurs in the text, in an open hash
// table of 128-byte buckets of eight 16-byte entries [k-mer as a little-endian number | l | e << 32]; key 0 = free (a k-mer
// of the text holds no 0 byte).  One bucket fetch replaces the first k LF steps of a search.  An index of 2^32 symbols and more keeps
// k <= 6 and the bits 32..39 of l and e in the two top bytes of the key word: [k-mer : 48 | l >> 32 : 8 | e >> 32 : 8], [l | e << 32].
struct FmDeep
{
    const ulonglong2 * tab; // null = no table
    uint32_t k, n_buckets;
};

// ---- helpers of the k-mer table lookup, shared by fm_count2.hip (a quad per pattern) and wt_rrr.hip (a lane per pattern) ----
__device__ __forceinline__ uint64_t load_tail8(const uint8_t * __restrict__ pats, uint64_t end)
{
    uint64_t v;
    if (end >= 8)
        __builtin_memcpy(&v, pats + end - 8, 8);
    else
    {
        v = 0;
        for (uint64_t j = 0; j < end; ++j)
            v |= (uint64_t)pats[j] << (8 * (8 - end + j));
    }
    return v;
}

constexpr uint64_t kDeepMul = UINT64_C(0x9E3779B97F4A7C15);
__device__ __forceinline__ uint32_t deep_bucket(uint64_t key, uint32_t n_buckets)
{
    return (uint32_t)__umul64hi(key * kDeepMul, (uint64_t)n_buckets);
}
// does any of the low k bytes of x equal zero?
__device__ __forceinline__ bool has_zero_byte(uint64_t x, uint32_t k)
{
    const uint64_t m = k >= 8 ? ~UINT64_C(0) : ((UINT64_C(1) << (8 * k)) - 1);
    return (((x - UINT64_C(0x0101010101010101)) & ~x & UINT64_C(0x8080808080808080)) & m) != 0;
}

// one LANE looks `key` up (narrow entries: fewer than 2^32 symbols): true and [l, e) if the k-mer occurs in the text
__device__ __forceinline__ bool lane_deep_find(const FmDeep & D, uint64_t key, uint64_t & l, uint64_t & e)
{
    uint32_t b = deep_bucket(key, D.n_buckets);
    for (uint32_t tries = 0; tries < D.n_buckets; ++tries)
    {
        const ulonglong2 * bp = D.tab + (uint64_t)b * 8;
        unsigned used = 0;
#pragma unroll
        for (int t = 0; t < 8; ++t)
        {
            const ulonglong2 en = bp[t];
            if (en.x == key)
            {
                l = (uint32_t)en.y;
                e = en.y >> 32;
                return true;
            }
            used += en.x != 0 ? 1u : 0u;
        }
        if (used < 8)
            return false; // an insertion would have used the free slot
        b = b + 1 == D.n_buckets ? 0 : b + 1;
    }
    return false;
}

struct WtHost;
sdsl_hip_status fm_rrr_launch_count(const WtHost & wt, const FmTables * d_tab, FmJump jump, FmDeep deep, uint64_t csa_size,
                                    const uint8_t * d_pats, uint32_t m, const uint64_t * d_offsets, const uint32_t * d_order,
                                    uint64_t n_pat, uint64_t * d_cnt, uint64_t * d_l, uint64_t * d_r, hipStream_t s, bool verify = false);
sdsl_hip_status fm_rrr_launch_backward_step(const WtHost & wt, const FmTables * d_tab, uint64_t csa_size,
                                            const uint64_t * d_l, const uint64_t * d_r, const uint8_t * d_c, uint64_t n,
                                            uint64_t * d_lo, uint64_t * d_ro, hipStream_t s);

} // namespace sdslhip
