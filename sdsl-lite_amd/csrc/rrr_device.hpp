// rrr_device.hpp — device layout of rrr_vector<63, int_vector<>, 32> and the per-query device functions shared by
// rrr.hip (rank / select / access kernels) and the wavelet-tree kernels that run on an rrr-compressed bit vector
// (wt_huff<rrr_vector<63>>).  Layout and rationale: rrr.hip header comment, DESIGN.md §5.
#pragma once
#include "bv_device.hpp"

namespace sdslhip {

constexpr unsigned kRrrBS = 63;
constexpr unsigned kRrrK = 32; // SDSL's t_k: blocks per sample of the SERIALISED form (m_rank, m_btnrp, m_invert)
constexpr uint64_t kRrrSB = (uint64_t)kRrrBS * kRrrK; // 2016 bits per SDSL superblock (parser / serialiser only)
// The device groups 34 blocks per record, not 32: a 128-byte record per 2016 bits costs 0.51 bits/bit before a single
// offset is stored; per 2142 bits it costs 0.478, and at 5 % density the ~552 offset bits of 34 blocks fit the 576 inline
// bits for all but 1.7 % of the queries (36 blocks: 0.457 bits/bit, but 585 offset bits: four waves in five then wait for a
// lane with a second fetch, profiles/rrr_record_size_r02.txt).  Groups of nine classes, 7 bits each (9 x 7 = 63 bits of a
// word); the fourth class word holds seven.
constexpr unsigned kRecK = 34;
constexpr unsigned kGrp = 9;   // blocks per class word
constexpr unsigned kClsW = 7;  // bits per class field
constexpr uint64_t kRecSB = (uint64_t)kRrrBS * kRecK; // 2142 bits per record
constexpr unsigned kRecWords = 16;
constexpr unsigned kRecClasses = 3;  // words 3..6: the 34 block classes, nine 7-bit fields per word (seven in the last)
constexpr unsigned kRecInline = 7;   // words 7..15: the first 576 bits of the record's offsets (the rest: RrrView::stream)
constexpr unsigned kInlineWords = kRecWords - kRecInline;
constexpr unsigned kInlineBits = 64 * kInlineWords; // 576
constexpr unsigned kRrrBlock = 512; // threads per block (LDS holds the 32 KiB binomial table)

struct RrrTables
{
    // binom[m][k] = C(m, k), m,k in [0,63].  Rows are padded to 65 entries: with a stride of 64 the LDS bank of an
    // entry would depend on k alone, and the lanes of a wave decoding blocks of the same (small) class at different
    // rows m — the normal case of the sparse decoder — would all collide (77 % of the LDS cycles were bank conflicts).
    uint64_t binom[64][65];
    // bits of the field a block of class k occupies on the DEVICE: hi(C(63,k))+1 for the classes the sparse decoder handles
    // (0 if C == 1), 63 — the block itself — for the classes in between (see RAW classes below)
    uint8_t space[64];
    uint8_t sdsl_space[64]; // SDSL's width of the offset field, hi(C(63,k))+1 for every class (parser / serialiser)
    uint8_t space2[256];    // space[b & 15] + space[b >> 4]: two 4-bit class fields of the slim record format at once
    SH_HD uint64_t C(unsigned m, unsigned k) const { return binom[m][k]; }
    SH_HD uint64_t C63(unsigned k) const { return binom[63][k]; }
};

// What the kernels over the vectors INSIDE a wavelet tree stage in LDS instead (wt_rrr.hip; round 6): those vectors keep at most the classes
// 0..10 and 53..63 enumerative (rrr.hip, choose_sparse_max: limit 10 when not stand-alone), so the decoder reads the binomial columns 0..10
// and C(63, k) for k >= 53 only — 5.8 KiB instead of 33: seven blocks of 256 threads per CU instead of three of 512 (by LDS), i.e. the 70
// VGPRs of those kernels decide the occupancy (7 waves per SIMD) and not the table.  Wide records only (no space2).
constexpr unsigned kWtCols = 11;
struct RrrTablesWt
{
    uint64_t binom[64][kWtCols]; // C(m, k), k <= 10 (a row stride of 11 words: the lanes of a wave decoding one class at different rows spread over the banks)
    uint64_t top[kWtCols];       // C(63, 53 + j)
    uint8_t space[64];
    SH_HD uint64_t C(unsigned m, unsigned k) const { return binom[m][k]; }
    SH_HD uint64_t C63(unsigned k) const { return top[k - 53]; }
};

// RAW classes.  Decoding a block from its offset costs one bisection per set (or, via the complement, unset) bit, and a WAVE
// pays for the lane with the most of them; a block of the middle classes (63 dependent steps) bounded every kernel that met
// one (count on csa_wt<wt_huff<rrr_vector<63>>>: 95 % VALU).  So per vector a threshold t <= 10 is chosen when it is built
// (rrr.hip, choose_sparse_max): classes k <= t and k >= 63 - t stay enumerative, everything in between is stored as the 63
// bits themselves (space[k] == 63) and needs no decoder.  t is the smallest value that keeps the extra space within 2 % of
// the vector's compressed size — a 5 %-dense vector gets t = 7 (+0.6 % of its length), a wavelet tree over text t = 3 or so.
SH_HD bool rrr_raw_width(unsigned len)
{
    return len == kRrrBS;
}

struct RrrView
{
    const uint64_t * rec;    // n_sb * 16 words (n_sb = number of RECORDS, 34 blocks each)
    const uint64_t * stream; // per record the offset bits beyond its inline area, in whole words; padded by two words
    const RrrTables * tables;
    const uint32_t * sel[2]; // select directories: (position of the j<<shift-th argument) >> pshift, + sentinel
    uint64_t n_bits, n_blocks, n_sb, ones;
    uint32_t sel_shift[2]; // log2 of the select sampling rate, per bit value (the rarer value gets the denser samples)
    uint32_t sel_pshift; // position quantisation of the samples (0 for n_bits < 2^32)
    // automatic dispatch of large batches (rrr.hip): the direct rank kernel returns at once when this word is non-zero (the batch
    // is then answered by the bucketed path enqueued beside it, rrr_sorted.hip); nullptr everywhere else
    const uint32_t * skip_if;
    uint32_t fmt; // record format: 0 = RrrFmtW (34 blocks, 7-bit classes), 1 = RrrFmtS (42 blocks, 4-bit classes)
    uint32_t sparse_max; // classes 0..sparse_max and 63-sparse_max..63 are enumerative offsets, the rest raw (<= 20: rrr.hip, choose_sparse_max)
};

// ---- device: block decoder -----------------------------------------------------------------------
// Sparse blocks (k <= 10 ones, or <= 10 zeros via the complement) are decoded from their offset: one bisection over the
// binomial column per set bit (the idea of the reference's k <= 10 path).  Everything in between is stored raw.
// `stop`: only the positions below it are wanted (rank needs the bits in front of its position): set bits come out in
// increasing position, so the loop ends at the first one at or behind `stop` — a wave then runs as many rounds as its lane
// with the most set bits IN FRONT of its position, not in its whole block
template <class TT>
__device__ __forceinline__ uint64_t rrr_decode_sparse(const TT * T, unsigned k, uint64_t nr, unsigned stop = kRrrBS)
{
    uint64_t bits = 0;
    int hi = 62; // candidate rows m = 62 - position
    while (k > 2)
    {
        // largest m in [k-1, hi] with C(m, k) <= nr  (C(k-1,k) = 0 always qualifies)
        int lo = (int)k - 1, h = hi;
        while (lo < h)
        {
            int mid = (lo + h + 1) >> 1;
            if (T->C(mid, k) <= nr)
                lo = mid;
            else
                h = mid - 1;
        }
        if (62 - lo >= (int)stop)
            return bits;
        bits |= UINT64_C(1) << (62 - lo);
        nr -= T->C(lo, k);
        --k;
        hi = lo - 1;
    }
    // the last two set bits in closed form, without table reads: nr = C(m2, 2) + m3 with m2 > m3 — m2 from a square root,
    // checked in integers (a block of class k costs max(0, k - 2) bisections instead of k)
    if (k == 2)
    {
        const unsigned x = (unsigned)nr; // < C(63, 2)
        unsigned m = (unsigned)((1.0f + __builtin_sqrtf(1.0f + 8.0f * (float)x)) * 0.5f);
        if (m * (m - 1) / 2 > x)
            --m;
        else if ((m + 1) * m / 2 <= x)
            ++m;
        if (62 - m >= stop)
            return bits;
        bits |= UINT64_C(1) << (62 - m);
        nr = x - m * (m - 1) / 2;
        k = 1;
    }
    if (k == 1 && 62 - (unsigned)nr < stop)
        bits |= UINT64_C(1) << (62 - (unsigned)nr);
    return bits;
}

// the 63-bit block of class k whose device field holds f (rrr_helper.hpp:480-534 for the enumerative classes: blocks of a
// class are numbered in lexicographic order of (b0, b1, ...), 0 < 1)
// (stop < 63: only the bits at positions below `stop` are valid in the result)
template <class TT>
__device__ __forceinline__ uint64_t rrr_decode_block(const TT * T, unsigned k, uint64_t f, unsigned stop = kRrrBS)
{
    if (rrr_raw_width(T->space[k]))
        return f;
    const bool flip = k > 31; // the complement of a block with k ones is the block with 63-k ones and offset C(63,k)-1-nr
    uint64_t bits = rrr_decode_sparse(T, flip ? kRrrBS - k : k, flip ? T->C63(k) - 1 - f : f, stop);
    if (flip)
        bits = ~bits & lo_set(kRrrBS);
    return bits;
}

__device__ __forceinline__ void rrr_stage_tables(RrrTables * lds, const RrrTables * g)
{
    const uint64_t * src = reinterpret_cast<const uint64_t *>(g);
    uint64_t * dst = reinterpret_cast<uint64_t *>(lds);
    for (unsigned i = threadIdx.x; i < sizeof(RrrTables) / 8; i += blockDim.x)
        dst[i] = src[i];
    __syncthreads();
}

__device__ __forceinline__ void rrr_stage_tables(RrrTablesWt * lds, const RrrTables * g)
{
    for (unsigned i = threadIdx.x; i < 64 * kWtCols; i += blockDim.x)
        lds->binom[i / kWtCols][i % kWtCols] = g->binom[i / kWtCols][i % kWtCols];
    for (unsigned i = threadIdx.x; i < 64; i += blockDim.x)
    {
        lds->space[i] = g->space[i];
        if (i >= 53)
            lds->top[i - 53] = g->binom[63][i];
    }
    __syncthreads();
}

// ---- record accessors ------------------------------------------------------------------------------
// word 2 of a record: for g = 1..3 the offset bits and the ones in blocks [0, 9g) of the record, packed as
// [bits | ones << w] with field width w = 10 at bit 0 (g = 1: <= 9 * 60, 9 * 63), w = 11 at bit 20 (g = 2) and at bit 42 (g = 3).
SH_HD uint64_t rrr_pack_prefix(const unsigned ones[3], const unsigned bits[3])
{
    return (uint64_t)bits[0] | ((uint64_t)ones[0] << 10) | ((uint64_t)bits[1] << 20) | ((uint64_t)ones[1] << 31)
           | ((uint64_t)bits[2] << 42) | ((uint64_t)ones[2] << 53);
}
SH_HD void rrr_prefix(uint64_t P, unsigned g, unsigned & ones, unsigned & bits)
{ // g in [0,3]
    const unsigned w = g <= 1 ? 10u : 11u, m = (1u << w) - 1;
    const uint64_t x = g == 0 ? 0 : (g == 1 ? P : P >> (g == 2 ? 20 : 42));
    bits = (unsigned)x & m;
    ones = (unsigned)(x >> w) & m;
}

// class fields of a class word
SH_HD unsigned rrr_cls(uint64_t cw, unsigned u)
{
    return (unsigned)(cw >> (kClsW * u)) & 0x7Fu;
}
SH_HD uint64_t rrr_cls_below(uint64_t cw, unsigned u)
{ // the fields of blocks 0..u-1
    return cw & ((UINT64_C(1) << (kClsW * u)) - 1);
}
constexpr uint64_t kCls63 = UINT64_C(0x3F) * ((UINT64_C(1) << 0) | (UINT64_C(1) << 7) | (UINT64_C(1) << 14) | (UINT64_C(1) << 21) | (UINT64_C(1) << 28)
                                               | (UINT64_C(1) << 35) | (UINT64_C(1) << 42) | (UINT64_C(1) << 49) | (UINT64_C(1) << 56)); // 63 in every field

// offset field of `len` bits at relative position `rel` of record `r`: the record's offsets are the words of its inline
// area followed by the words of its stretch of the stream (from word `ptr` on: a stretch starts at a word boundary,
// so the seam needs no special case — a field simply takes its two words from wherever they live)
template <unsigned INL0, unsigned INLW>
__device__ __forceinline__ uint64_t rrr_field_t(const RrrView & v, const uint64_t * r, uint64_t ptr, unsigned rel, unsigned len)
{
    if (len == 0)
        return 0;
    const unsigned w = rel >> 6, o = rel & 63;
    const uint64_t * far = v.stream + ptr - INLW; // word w of the record's offsets, w >= INLW
    const uint64_t * p0 = w < INLW ? r + INL0 + w : far + w;
    uint64_t x = *p0 >> o;
    if (o + len > 64)
    {
        const uint64_t * p1 = w + 1 < INLW ? r + INL0 + w + 1 : far + w + 1;
        x |= *p1 << (64 - o);
    }
    return x & lo_set(len);
}
__device__ __forceinline__ uint64_t rrr_field(const RrrView & v, const uint64_t * r, uint64_t ptr, unsigned rel,
                                              unsigned len)
{
    return rrr_field_t<kRecInline, kInlineWords>(v, r, ptr, rel, len);
}

// sum of the nine 7-bit fields of x (each <= 63)
SH_HD unsigned sum_fields9(uint64_t x)
{
    constexpr uint64_t M = UINT64_C(0x7F) * ((UINT64_C(1) << 0) | (UINT64_C(1) << 14) | (UINT64_C(1) << 28) | (UINT64_C(1) << 42) | (UINT64_C(1) << 56));
    const uint64_t t = (x & M) + ((x >> 7) & M); // five 14-bit lanes, each <= 126
    const unsigned lo = (unsigned)t, hi = (unsigned)(t >> 28); // lanes 0, 1 | lanes 2, 3
    return (lo & 0x3FFFu) + ((lo >> 14) & 0x3FFFu) + (hi & 0x3FFFu) + ((hi >> 14) & 0x3FFFu) + (unsigned)(t >> 56);
}

// sum of space[] over the nine class fields of m (class 0 adds nothing: space[0] == 0)
template <class TT>
__device__ __forceinline__ unsigned rrr_space_sum9(const TT * T, uint64_t m)
{
    const unsigned lo = (unsigned)m, hi = (unsigned)(m >> 28); // fields 0..3 | fields 4..8
    unsigned bits = 0;
#pragma unroll
    for (int t = 0; t < 4; ++t)
        bits += T->space[(lo >> (7 * t)) & 0x7Fu];
#pragma unroll
    for (int t = 0; t < 5; ++t)
        bits += T->space[(t < 4 ? (hi >> (7 * t)) : (unsigned)(m >> 56)) & 0x7Fu];
    return bits;
}

// ---- the two record formats -------------------------------------------------------------------------
// W ("wide", everything above): 34 blocks, 7-bit classes, three header words.  Right for the vectors of a wavelet tree (classes
// around 31, raw blocks) and for anything dense; 0.478 bits per bit before the first overflow word.
// S ("slim"): for SPARSE vectors, where the classes are most of the overhead.  42 blocks per 128-byte record (2646 bits):
//   word 0      bits 0..39 ones before the record | 40..49 ones in blocks [0,16) | 50..60 ones in blocks [0,32)
//   word 1      bits 0..32 WORD pointer into the overflow stream | 33..42 offset bits of blocks [0,16) | 43..53 of [0,32)
//               | 54..63 ones in blocks [32,42)
//   words 2..4  the 42 classes as 4-bit fields, sixteen per word (ten in the last).  Field 15 is an ESCAPE: the block has 15 or
//               more ones and its field in the offsets is the block itself (63 bits); its class is the popcount of that.  (In a
//               vector of density 5 % one block in 2.5 million has 15 ones; the slow paths below are correct for any number.)
//   words 5..15 the first 704 bits of the record's offsets
// 2^34 bits at 5 %: 682 +- 51 offset bits per record: 0.40 bits per bit all in (W: 0.488, SDSL: 0.37), and as with W-34 fewer than
// 2 % of the queries need a second fetch.  Chosen per vector when it is built (rrr.hip, choose_format).
struct RrrFmtW
{
    static constexpr unsigned id = 0, K = kRecK, GRP = kGrp, NCW = 4, CLS0 = kRecClasses, INL0 = kRecInline, INLW = kInlineWords;
    static constexpr unsigned INLB = kInlineBits;
    static constexpr uint64_t SB = kRecSB;
    static SH_HD uint64_t ones_before(uint64_t r0) { return r0; }
    static SH_HD uint64_t ptr(uint64_t r1) { return r1 & ((UINT64_C(1) << 48) - 1); }
};
struct RrrFmtS
{
    static constexpr unsigned id = 1, K = 42, GRP = 16, NCW = 3, CLS0 = 2, INL0 = 5, INLW = 11;
    static constexpr unsigned INLB = 64 * INLW; // 704
    static constexpr uint64_t SB = (uint64_t)kRrrBS * K; // 2646
    static SH_HD uint64_t ones_before(uint64_t r0) { return r0 & ((UINT64_C(1) << 40) - 1); }
    static SH_HD uint64_t ptr(uint64_t r1) { return r1 & ((UINT64_C(1) << 33) - 1); }
};
constexpr unsigned kEsc = 15;            // class field of an escaped block (format S)
constexpr uint64_t kSlimMaxStream = UINT64_C(1) << 33; // words of overflow stream a slim vector may have

SH_HD uint64_t rrs_pack0(uint64_t ones_before, unsigned o16, unsigned o32)
{
    return ones_before | ((uint64_t)o16 << 40) | ((uint64_t)o32 << 50);
}
SH_HD uint64_t rrs_pack1(uint64_t ptr, unsigned b16, unsigned b32, unsigned ones3)
{
    return ptr | ((uint64_t)b16 << 33) | ((uint64_t)b32 << 43) | ((uint64_t)ones3 << 54);
}
SH_HD void rrs_prefix(uint64_t r0, uint64_t r1, unsigned g, unsigned & ones, unsigned & bits)
{ // g in [0,2]: ones and offset bits in blocks [0, 16 g)
    ones = g == 0 ? 0u : (g == 1 ? (unsigned)(r0 >> 40) & 0x3FFu : (unsigned)(r0 >> 50) & 0x7FFu);
    bits = g == 0 ? 0u : (g == 1 ? (unsigned)(r1 >> 33) & 0x3FFu : (unsigned)(r1 >> 43) & 0x7FFu);
}
SH_HD unsigned rrs_ones_in(uint64_t r0, uint64_t r1)
{
    return ((unsigned)(r0 >> 50) & 0x7FFu) + (unsigned)(r1 >> 54);
}
SH_HD unsigned rrs_cls(uint64_t cw, unsigned u)
{
    return (unsigned)(cw >> (4 * u)) & 15u;
}
SH_HD uint64_t rrs_below(uint64_t cw, unsigned u)
{ // the fields of blocks 0..u-1, u in [0,16]
    return u >= 16 ? cw : cw & ((UINT64_C(1) << (4 * u)) - 1);
}
SH_HD unsigned rrs_sum16(uint64_t x)
{ // sum of the sixteen 4-bit fields
    const uint64_t t = (x & UINT64_C(0x0F0F0F0F0F0F0F0F)) + ((x >> 4) & UINT64_C(0x0F0F0F0F0F0F0F0F)); // bytes <= 30
    return (unsigned)((t * UINT64_C(0x0101010101010101)) >> 56);
}
SH_HD uint64_t rrs_esc(uint64_t x)
{ // bit 4u set for every field u that holds the escape
    return x & (x >> 1) & (x >> 2) & (x >> 3) & UINT64_C(0x1111111111111111);
}
__device__ __forceinline__ unsigned rrs_space_sum(const RrrTables * T, uint64_t m)
{ // offset bits of the fields of m (an escape: 63; a zero field: 0)
    const unsigned lo = (unsigned)m, hi = (unsigned)(m >> 32);
    unsigned bits = 0;
#pragma unroll
    for (int t = 0; t < 4; ++t)
        bits += T->space2[(lo >> (8 * t)) & 0xFFu] + T->space2[(hi >> (8 * t)) & 0xFFu];
    return bits;
}

// ---- rank / access ---------------------------------------------------------------------------------
// One query per lane, no cooperation: the record is laid out so that a lane needs 32 bytes of it (two header words,
// prefix word, ONE class word) plus the offset field.
struct RankTail
{
    uint64_t rank; // ones before the block
    uint64_t nr;   // offset of the block
    unsigned k, off;
};

template <class TT>
__device__ __forceinline__ RankTail rrr_rank_head(const RrrView & v, const TT * T, uint64_t i)
{
    RankTail t;
    const uint64_t sb = i / kRecSB; // one 64-bit division; everything below it is 32-bit
    const unsigned in_sb = (unsigned)(i - sb * kRecSB), j = in_sb / kRrrBS, g = j / kGrp, u = j - g * kGrp;
    t.off = in_sb - j * kRrrBS;
    const uint64_t * r = (const uint64_t *)__builtin_assume_aligned(v.rec + sb * kRecWords, 128);
    const uint64_t r0 = r[0], r1 = r[1], P = r[2];
    const uint64_t cw = r[kRecClasses + g];
    unsigned ones, bits;
    rrr_prefix(P, g, ones, bits);
    const uint64_t below = rrr_cls_below(cw, u);
    ones += sum_fields9(below);
    bits += rrr_space_sum9(T, below);
    t.rank = r0 + ones;
    t.k = rrr_cls(cw, u);
    t.nr = rrr_field(v, r, r1 & ((UINT64_C(1) << 48) - 1), bits, T->space[t.k]);
    return t;
}

// the same on a slim record
__device__ __forceinline__ RankTail rrs_rank_head(const RrrView & v, const RrrTables * T, uint64_t i)
{
    using F = RrrFmtS;
    RankTail t;
    const uint64_t sb = i / F::SB;
    const unsigned in_sb = (unsigned)(i - sb * F::SB), j = in_sb / kRrrBS, g = j / F::GRP, u = j - g * F::GRP;
    t.off = in_sb - j * kRrrBS;
    const uint64_t * r = (const uint64_t *)__builtin_assume_aligned(v.rec + sb * kRecWords, 128);
    const uint64_t r0 = r[0], r1 = r[1];
    const uint64_t cw = r[F::CLS0 + g];
    unsigned ones, bits;
    rrs_prefix(r0, r1, g, ones, bits);
    const uint64_t below = rrs_below(cw, u), ptr = F::ptr(r1);
    uint64_t esc = rrs_esc(below);
    if (esc)
    { // an escaped block in front: its field counted 15, the block says how many it really has
        do
        {
            const unsigned e = (unsigned)__builtin_ctzll(esc) >> 2;
            esc &= esc - 1;
            const uint64_t raw = rrr_field_t<F::INL0, F::INLW>(v, r, ptr, bits + rrs_space_sum(T, rrs_below(cw, e)), kRrrBS);
            ones += popc64(raw) - kEsc;
        } while (esc);
    }
    ones += rrs_sum16(below);
    bits += rrs_space_sum(T, below);
    t.rank = F::ones_before(r0) + ones;
    t.k = rrs_cls(cw, u); // (an escape decodes as class 15: raw, like every class from 11 to 52)
    t.nr = rrr_field_t<F::INL0, F::INLW>(v, r, ptr, bits, T->space[t.k]);
    return t;
}
template <class F, class TT>
__device__ __forceinline__ RankTail rrr_rank_head_f(const RrrView & v, const TT * T, uint64_t i)
{
    if constexpr (F::id == 0)
        return rrr_rank_head(v, T, i);
    else
        return rrs_rank_head(v, T, i);
}

// rank_1(pos); optionally the bit at pos (pos < n_bits then)
template <class F = RrrFmtW, class TT>
__device__ __forceinline__ uint64_t rrr_rank1(const RrrView & v, const TT * T, uint64_t pos,
                                              unsigned * bit_out = nullptr)
{
    const RankTail t = rrr_rank_head_f<F>(v, T, pos);
    uint64_t bits = 0;
    if (bit_out || t.off != 0) // rank at a block boundary needs no decode
        bits = rrr_decode_block(T, t.k, t.nr, t.off + (bit_out ? 1u : 0u));
    if (bit_out)
        *bit_out = (unsigned)(bits >> t.off) & 1u;
    return t.rank + popc64(bits & lo_set(t.off));
}

// rank_1(pa) and rank_1(pb), pa <= pb: the two cascades of an LF step (suffix_array_algorithm.hpp:195-196).  Once
// the SA interval is narrow both positions usually fall into the same 63-bit block: then b reuses a's head and
// decoded block; and a b that starts a block needs no decode at all — for an interval of size one (b == a + 1) one
// of the two always holds.
template <class TT>
__device__ __forceinline__ void rrr_rank2(const RrrView & v, const TT * T, uint64_t pa, uint64_t pb, uint64_t & ra,
                                          uint64_t & rb)
{
    const uint64_t blk_a = pa / kRrrBS, blk_b = pb / kRrrBS;
    const bool same = blk_a == blk_b;
    const RankTail ta = rrr_rank_head(v, T, pa);
    RankTail tb;
    tb.rank = ta.rank;
    tb.nr = 0;
    tb.k = 0;
    tb.off = (unsigned)(pb - blk_b * kRrrBS);
    if (!same)
        tb = rrr_rank_head(v, T, pb);
    uint64_t bits_a = 0;
    if (ta.off != 0 || (same && tb.off != 0))
        bits_a = rrr_decode_block(T, ta.k, ta.nr, same ? tb.off : ta.off); // (same block: pa <= pb)
    ra = ta.rank + popc64(bits_a & lo_set(ta.off));
    uint64_t bits_b = bits_a;
    if (!same && tb.off != 0)
        bits_b = rrr_decode_block(T, tb.k, tb.nr, tb.off);
    rb = tb.rank + popc64(bits_b & lo_set(tb.off));
}

// ---- select ----------------------------------------------------------------------------------------
// position of the (k0+1)-th BIT-valued bit; 0 <= k0 < #BIT-valued bits.  Superblock search over the record headers
// (directory of argument POSITIONS, interpolated probe, exact counts from the probed header, bisection every second
// late probe — the scheme of bv_device.hpp), then the prefix word picks the group of 8 blocks, byte arithmetic on
// ONE class word the block, and the decoded block the bit.
// smp0 / smp1: the directory samples sel[BIT][k0 >> sel_shift] and the next one (callers may load them ahead of time)
// The search is split in two so that a kernel can keep every lane busy: rrr_sel_probe does ONE probe of the
// superblock search, rrr_sel_finish everything after the hit.
struct RrrSelState
{ // invariant: lo_pos <= position(k0) < hi_pos, lo_cnt <= k0 < hi_cnt
    uint64_t k0, lo_pos, lo_cnt, hi_pos, hi_cnt;
    int tries;
};
struct RrrSelHit
{
    const uint64_t * r; // the record of superblock g
    uint64_t g, before, r1, P, c0, c1, c2, c3;
};

template <int BIT>
__device__ __forceinline__ void rrr_sel_init(const RrrView & v, RrrSelState & st, uint64_t k0, uint32_t smp0, uint32_t smp1)
{
    const uint64_t total = BIT ? v.ones : v.n_bits - v.ones;
    const uint32_t sh = v.sel_shift[BIT], ps = v.sel_pshift;
    const uint64_t js = k0 >> sh;
    st.k0 = k0;
    st.lo_pos = (uint64_t)smp0 << ps;
    st.lo_cnt = js << sh;
    st.hi_pos = ((uint64_t)smp1 + 1) << ps;
    st.hi_cnt = (js + 1) << sh;
    if (st.hi_cnt > total)
        st.hi_cnt = total;
    st.tries = 0;
}

// one probe; true when superblock h.g holds the argument
template <int BIT, class F = RrrFmtW>
__device__ __forceinline__ bool rrr_sel_probe(const RrrView & v, RrrSelState & st, RrrSelHit & h)
{
    const uint64_t span = st.hi_pos - st.lo_pos;
    uint64_t p;
    if (st.tries >= 3 && (st.tries & 1))
        p = st.lo_pos + (span >> 1);
    else
        p = sel_interpolate(st.lo_pos, span, st.k0 - st.lo_cnt, st.hi_cnt - st.lo_cnt, v.sel_shift[BIT]);
    ++st.tries;
    uint64_t g = p / F::SB;
    if (g >= v.n_sb)
        g = v.n_sb - 1;
    const uint64_t * r = (const uint64_t *)__builtin_assume_aligned(v.rec + g * kRecWords, 128);
    const uint64_t r0 = r[0];
    h.r = r;
    h.g = g;
    h.r1 = r[1];
    // prefix and class words ride along (same line, no extra latency): needed as soon as the probe hits
    uint64_t ones_in;
    if constexpr (F::id == 0)
    {
        h.P = r[2];
        h.c0 = r[kRecClasses];
        h.c1 = r[kRecClasses + 1];
        h.c2 = r[kRecClasses + 2];
        h.c3 = r[kRecClasses + 3];
        ones_in = (h.r1 >> 48) & 0xFFF;
    }
    else
    {
        h.P = r0; // (the slim format keeps its prefix counts in the two header words)
        h.c0 = r[F::CLS0];
        h.c1 = r[F::CLS0 + 1];
        h.c2 = r[F::CLS0 + 2];
        h.c3 = 0;
        ones_in = rrs_ones_in(r0, h.r1);
    }
    const uint64_t before1 = F::ones_before(r0);
    const uint64_t start = g * F::SB;
    const uint64_t len_in = v.n_bits - start < F::SB ? v.n_bits - start : F::SB;
    h.before = BIT ? before1 : start - before1;
    const uint64_t c = BIT ? ones_in : len_in - ones_in;
    if (st.k0 < h.before)
    {
        st.hi_pos = start;
        st.hi_cnt = h.before;
        return false;
    }
    if (st.k0 >= h.before + c)
    {
        st.lo_pos = start + F::SB;
        st.lo_cnt = h.before + c;
        return false;
    }
    return true;
}

// after the hit: the block that holds the argument and where its offset field lies
struct RrrSelLoc
{
    uint64_t bstart; // first position of the block
    uint64_t ptr;    // the record's stretch of the overflow stream
    unsigned k, rel, want; // class, position of the offset field among the record's offsets, rank of the argument inside the block
};

template <int BIT, class TT>
__device__ __forceinline__ RrrSelLoc rrr_sel_locate_w(const RrrView & v, const TT * T, uint64_t k0, const RrrSelHit & h)
{
    const uint64_t g = h.g, before = h.before, P = h.P, c0 = h.c0, c1 = h.c1, c2 = h.c2, c3 = h.c3;
    // inside record g: the group of 9 blocks.  Zeros before block 9q are 567q - ones (every block in front of
    // the one that holds an existing argument is a complete 63-bit block).
    unsigned want = (unsigned)(k0 - before);
    unsigned o[4], b[4];
    o[0] = b[0] = 0;
    rrr_prefix(P, 1, o[1], b[1]);
    rrr_prefix(P, 2, o[2], b[2]);
    rrr_prefix(P, 3, o[3], b[3]);
    unsigned q = 0;
#pragma unroll
    for (unsigned t = 1; t < 4; ++t)
        q += want >= (BIT ? o[t] : kGrp * kRrrBS * t - o[t]) ? 1u : 0u;
    unsigned rel = q == 0 ? 0u : (q == 1 ? b[1] : (q == 2 ? b[2] : b[3]));
    const unsigned oq = q == 0 ? 0u : (q == 1 ? o[1] : (q == 2 ? o[2] : o[3]));
    want -= BIT ? oq : kGrp * kRrrBS * q - oq;
    const uint64_t cw = q == 0 ? c0 : (q == 1 ? c1 : (q == 2 ? c2 : c3));
    // arguments per block of the group as 7-bit fields; the block holding the argument is complete or the vector's last
    // block, and blocks behind the end of the vector must not offer zeros
    uint64_t args = cw;
    if (!BIT)
    {
        const uint64_t b0 = g * kRecK + kGrp * (uint64_t)q;
        if ((b0 + kGrp) * kRrrBS <= v.n_bits)
            args = kCls63 - cw;
        else
        {
            args = 0;
            for (unsigned t = 0; t < kGrp; ++t)
            {
                const uint64_t bstart = (b0 + t) * kRrrBS;
                const unsigned blen =
                    bstart >= v.n_bits ? 0u : (unsigned)(v.n_bits - bstart < kRrrBS ? v.n_bits - bstart : kRrrBS);
                args |= (uint64_t)(blen - rrr_cls(cw, t)) << (kClsW * t);
            }
        }
    }
    // first block u with want < a_0 + ... + a_u: the sum of the first four fields picks a half, three running sums the
    // block inside it (the ninth field can never be passed: the argument lies in this group)
    unsigned u = 0;
    {
        const unsigned lo = (unsigned)args & 0x0FFFFFFFu, hi = (unsigned)(args >> 28); // fields 0..3 | 4..7 (+ 4 bits of 8)
        const unsigned pair = (lo & 0x001FC07Fu) + ((lo >> 7) & 0x001FC07Fu); // a0 + a1 | a2 + a3 in 14-bit lanes
        const unsigned sum4 = (pair & 0x3FFFu) + (pair >> 14);
        unsigned half = lo;
        if (want >= sum4)
        {
            want -= sum4;
            u = 4;
            half = hi;
        }
        unsigned acc = 0, sub = 0, t4 = 0;
#pragma unroll
        for (unsigned t = 0; t < 4; ++t)
        { // (in the upper half the fourth step compares with a4 + .. + a7: passing it means block 8)
            acc += (half >> (7 * t)) & 0x7Fu;
            const bool ge = want >= acc;
            t4 += ge ? 1u : 0u;
            sub = ge ? acc : sub;
        }
        want -= sub;
        u += t4;
    }
    RrrSelLoc L;
    L.rel = rel + rrr_space_sum9(T, rrr_cls_below(cw, u));
    L.k = rrr_cls(cw, u);
    L.bstart = (g * kRecK + kGrp * (uint64_t)q + u) * kRrrBS;
    L.ptr = h.r1 & ((UINT64_C(1) << 48) - 1);
    L.want = want;
    return L;
}

// the same on a slim record: two prefix counts pick the group of sixteen blocks, a bisection over running sums of its 4-bit
// fields the block
template <int BIT>
__device__ __forceinline__ RrrSelLoc rrr_sel_locate_s(const RrrView & v, const RrrTables * T, uint64_t k0, const RrrSelHit & hit)
{
    using F = RrrFmtS;
    const RrrSelHit h = hit; // (a copy: taken by reference the hit stayed in scratch memory, 80 bytes per lane)
    unsigned want = (unsigned)(k0 - h.before);
    unsigned o[3], b[3];
    rrs_prefix(h.P, h.r1, 0, o[0], b[0]);
    rrs_prefix(h.P, h.r1, 1, o[1], b[1]);
    rrs_prefix(h.P, h.r1, 2, o[2], b[2]);
    unsigned q = 0;
#pragma unroll
    for (unsigned t = 1; t < 3; ++t)
        q += want >= (BIT ? o[t] : F::GRP * kRrrBS * t - o[t]) ? 1u : 0u;
    unsigned rel = q == 0 ? 0u : (q == 1 ? b[1] : b[2]);
    const unsigned oq = q == 0 ? 0u : (q == 1 ? o[1] : o[2]);
    want -= BIT ? oq : F::GRP * kRrrBS * q - oq;
    const uint64_t cw = q == 0 ? h.c0 : (q == 1 ? h.c1 : h.c2);
    const uint64_t ptr = F::ptr(h.r1);
    unsigned u = 0;
    if (!rrs_esc(cw))
    { // largest u with (arguments in blocks [0, u) of the group) <= want.  Blocks in front of the one that holds an existing
      // argument are complete, and phantom blocks behind the end of the vector lie behind it: 63 u - ones is exact where it matters
#pragma unroll
        for (unsigned step = 8; step; step >>= 1)
        {
            const unsigned c = u + step, ob = rrs_sum16(rrs_below(cw, c));
            if ((BIT ? ob : kRrrBS * c - ob) <= want)
                u = c;
        }
        const unsigned ob = rrs_sum16(rrs_below(cw, u));
        want -= BIT ? ob : kRrrBS * u - ob;
        rel += rrs_space_sum(T, rrs_below(cw, u));
    }
    else
    { // an escaped block in the group: walk it, asking the blocks themselves
        for (;; ++u)
        {
            const unsigned c = rrs_cls(cw, u);
            unsigned k = c;
            if (c == kEsc)
                k = popc64(rrr_field_t<F::INL0, F::INLW>(v, h.r, ptr, rel, kRrrBS));
            const unsigned a = BIT ? k : kRrrBS - k;
            if (want < a || u == F::GRP - 1)
                break;
            want -= a;
            rel += T->space[c];
        }
    }
    RrrSelLoc L;
    L.rel = rel;
    L.k = rrs_cls(cw, u);
    L.bstart = (h.g * F::K + F::GRP * (uint64_t)q + u) * kRrrBS;
    L.ptr = ptr;
    L.want = want;
    return L;
}
template <int BIT, class F = RrrFmtW, class TT>
__device__ __forceinline__ RrrSelLoc rrr_sel_locate(const RrrView & v, const TT * T, uint64_t k0, const RrrSelHit & h)
{
    if constexpr (F::id == 0)
        return rrr_sel_locate_w<BIT>(v, T, k0, h);
    else
        return rrr_sel_locate_s<BIT>(v, T, k0, h); // (slim records: full tables only)
}

// does the offset field of the located block reach into the overflow stream (a second, random fetch)?
template <class F = RrrFmtW, class TT>
__device__ __forceinline__ bool rrr_sel_in_stream(const TT * T, const RrrSelLoc & L)
{
    return L.rel + T->space[L.k] > F::INLB;
}
// the field of a block for which rrr_sel_in_stream is false
template <class F = RrrFmtW>
__device__ __forceinline__ uint64_t rrr_field_inline(const uint64_t * r, unsigned rel, unsigned len)
{
    return read_bits(r + F::INL0, rel, len);
}

template <int BIT, class TT>
__device__ __forceinline__ uint64_t rrr_sel_decode(const RrrView & v, const TT * T, const RrrSelLoc & L, uint64_t nr)
{
    uint64_t bits = rrr_decode_block(T, L.k, nr);
    if (!BIT)
    {
        const unsigned blen = (unsigned)(v.n_bits - L.bstart < kRrrBS ? v.n_bits - L.bstart : kRrrBS);
        bits = ~bits & lo_set(blen);
    }
    return L.bstart + sel64(bits, L.want + 1);
}

template <int BIT, class F = RrrFmtW, class TT>
__device__ __forceinline__ uint64_t rrr_sel_finish(const RrrView & v, const TT * T, uint64_t k0, const RrrSelHit & h)
{
    const RrrSelLoc L = rrr_sel_locate<BIT, F>(v, T, k0, h);
    return rrr_sel_decode<BIT>(v, T, L, rrr_field_t<F::INL0, F::INLW>(v, h.r, L.ptr, L.rel, T->space[L.k]));
}

template <int BIT, class F = RrrFmtW, class TT>
__device__ __forceinline__ uint64_t rrr_select(const RrrView & v, const TT * T, uint64_t k0, uint32_t smp0,
                                               uint32_t smp1)
{
    RrrSelState st;
    RrrSelHit h;
    rrr_sel_init<BIT>(v, st, k0, smp0, smp1);
    while (!rrr_sel_probe<BIT, F>(v, st, h))
    {
    }
    return rrr_sel_finish<BIT, F>(v, T, k0, h);
}

template <int BIT, class F = RrrFmtW, class TT>
__device__ __forceinline__ uint64_t rrr_select(const RrrView & v, const TT * T, uint64_t k0)
{
    const uint64_t js = k0 >> v.sel_shift[BIT];
    return rrr_select<BIT, F>(v, T, k0, v.sel[BIT][js], v.sel[BIT][js + 1]);
}

} // namespace sdslhip
