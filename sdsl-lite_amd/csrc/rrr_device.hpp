// rrr_device.hpp — device layout of rrr_vector<63, int_vector<>, 32> and the per-query device functions shared by
// rrr.hip (rank / select / access kernels) and the wavelet-tree kernels that run on an rrr-compressed bit vector
// (wt_huff<rrr_vector<63>>).  Layout and rationale: rrr.hip header comment, DESIGN.md §5.
#pragma once
#include "bv_device.hpp"

namespace sdslhip {

constexpr unsigned kRrrBS = 63;
constexpr unsigned kRrrK = 32;
constexpr uint64_t kRrrSB = (uint64_t)kRrrBS * kRrrK; // 2016 bits per superblock
constexpr unsigned kRecWords = 16;
constexpr unsigned kInlineBits = 640;
constexpr unsigned kRrrBlock = 512; // threads per block (LDS holds the 32 KiB binomial table)

struct RrrTables
{
    uint64_t binom[64][64]; // binom[m][k] = C(m, k), m,k in [0,63]
    uint8_t space[64];      // bits of an offset field for class k: hi(C(63,k))+1, 0 if C == 1
};

struct RrrView
{
    const uint64_t * rec;    // n_sb * 16 words
    const uint64_t * stream; // offset stream (SDSL's m_btnr), padded by one word
    const RrrTables * tables;
    const uint32_t * sel[2]; // select directories: (position of the j<<shift-th argument) >> pshift, + sentinel
    uint64_t n_bits, n_blocks, n_sb, ones;
    uint32_t sel_shift;  // log2 of the select sampling rate
    uint32_t sel_pshift; // position quantisation of the samples (0 for n_bits < 2^32)
};

// ---- device: block decoder -----------------------------------------------------------------------
// Decodes the 63-bit block with k ones and offset nr (rrr_helper.hpp:480-534: blocks of a class are numbered in
// lexicographic order of (b0, b1, ...), 0 < 1).
//
// Two instruction streams, chosen PER WAVE so that a wave never pays for both:
//  * sparse: one bisection over the binomial column per set bit (the idea of the reference's k <= 10 path).  The
//    complement of a block with k ones is the block with 63-k ones and offset C(63,k)-1-nr, so classes >= 53 take
//    this path too;
//  * dense: 63 unrolled compare/subtract steps without branches or exec-mask traffic; correct for every class,
//    so a wave with mixed classes runs only this one.
__device__ __forceinline__ uint64_t rrr_decode_sparse(const RrrTables * T, unsigned k, uint64_t nr)
{
    uint64_t bits = 0;
    int hi = 62; // candidate rows m = 62 - position
    while (k > 0)
    {
        // largest m in [k-1, hi] with C(m, k) <= nr  (C(k-1,k) = 0 always qualifies)
        int lo = (int)k - 1, h = hi;
        while (lo < h)
        {
            int mid = (lo + h + 1) >> 1;
            if (T->binom[mid][k] <= nr)
                lo = mid;
            else
                h = mid - 1;
        }
        bits |= UINT64_C(1) << (62 - lo);
        nr -= T->binom[lo][k];
        --k;
        hi = lo - 1;
    }
    return bits;
}

__device__ __forceinline__ uint64_t rrr_decode_dense(const RrrTables * T, unsigned k, uint64_t nr)
{
    // acc collects the decisions MSB-first (one shift-or per step); bit-reversed at the end
    unsigned acc0 = 0, acc1 = 0;
#pragma unroll
    for (int p = 0; p < 32; ++p)
    {
        const uint64_t c = T->binom[62 - p][k];
        const bool one = nr >= c;
        nr -= one ? c : 0;
        k -= one ? 1u : 0u;
        acc0 = (acc0 << 1) | (one ? 1u : 0u);
    }
#pragma unroll
    for (int p = 32; p < 63; ++p)
    {
        const uint64_t c = T->binom[62 - p][k];
        const bool one = nr >= c;
        nr -= one ? c : 0;
        k -= one ? 1u : 0u;
        acc1 = (acc1 << 1) | (one ? 1u : 0u);
    }
    return (uint64_t)__brev(acc0) | ((uint64_t)(__brev(acc1) >> 1) << 32);
}

__device__ __forceinline__ uint64_t rrr_decode_block(const RrrTables * T, unsigned k, uint64_t nr)
{
    const bool flip = k > 31;
    const unsigned ks = flip ? kRrrBS - k : k;
    uint64_t bits;
    if (__builtin_amdgcn_ballot_w64(ks > 10) == 0)
    { // wave-uniform: every active lane has a sparse block (or a sparse complement)
        if (flip)
            nr = T->binom[63][k] - 1 - nr;
        bits = rrr_decode_sparse(T, ks, nr);
        if (flip)
            bits = ~bits & lo_set(kRrrBS);
    }
    else
        bits = rrr_decode_dense(T, k, nr);
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

// offset field of `len` bits at relative position `rel` inside superblock record `r`
__device__ __forceinline__ uint64_t rrr_field(const RrrView & v, const uint64_t * r, uint64_t ptr, unsigned rel,
                                              unsigned len)
{
    if (rel + len <= kInlineBits)
        return read_bits(r + 6, rel, len); // same 128-byte line as the header
    return read_bits(v.stream, ptr + rel, len);
}

// sum of the eight bytes of x (each <= 63)
__device__ __forceinline__ unsigned sum_bytes8(uint64_t x)
{
    uint64_t t = (x & UINT64_C(0x00FF00FF00FF00FF)) + ((x >> 8) & UINT64_C(0x00FF00FF00FF00FF));
    return (unsigned)((t * UINT64_C(0x0001000100010001)) >> 48);
}

// Sum of (class, space[class]) over this lane's 8 classes with in-superblock index < j; the quad sum
// gives ones and offset bits before block j.  packed = ones | bits << 16
__device__ __forceinline__ unsigned rrr_lane_prefix(const RrrTables * T, uint64_t cls8, int s, unsigned j)
{
    // keep the classes with in-lane index < j - 8s; the others become class 0, which adds nothing (space[0] == 0)
    int cnt = (int)j - 8 * s;
    cnt = cnt < 0 ? 0 : (cnt > 8 ? 8 : cnt);
    const uint64_t keep = cnt == 8 ? ~UINT64_C(0) : ((UINT64_C(1) << (8 * cnt)) - 1);
    const uint64_t m = cls8 & keep;
    unsigned bits = 0;
#pragma unroll
    for (int t = 0; t < 8; ++t)
        bits += T->space[(unsigned)(m >> (8 * t)) & 0xFF];
    return sum_bytes8(m) | (bits << 16);
}

// value of quad lane U in all four lanes (U is a compile-time constant: DPP quad_perm:[U,U,U,U])
template <int U>
__device__ __forceinline__ unsigned quad_bcast_lane(unsigned v)
{
    return (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, U * 0x55, 0xF, 0xF, true);
}
template <int U>
__device__ __forceinline__ uint64_t quad_bcast_lane_u64(uint64_t v)
{
    unsigned lo = quad_bcast_lane<U>((unsigned)v), hi = quad_bcast_lane<U>((unsigned)(v >> 32));
    return ((uint64_t)hi << 32) | lo;
}

// What the lane-parallel phase needs to finish one rank/access query.
struct RankTail
{
    uint64_t rank; // ones before the block
    uint64_t nr;   // the block's offset
    unsigned k, off;
};

// Cooperative half of rank(i): the quad fetches the superblock record of i (one 128-byte line), sums the
// class bytes below the block (8 per lane, DPP reduction) and fetches the block's offset field.
__device__ __forceinline__ RankTail rrr_rank_head(const RrrView & v, const RrrTables * T, int s, uint64_t i)
{
    uint64_t blk = i / kRrrBS;
    RankTail t;
    t.off = (unsigned)(i - blk * kRrrBS);
    uint64_t sb = blk / kRrrK;
    unsigned j = (unsigned)(blk % kRrrK);
    const uint64_t * r = v.rec + sb * kRecWords;
    uint64_t r0 = r[0], r1 = r[1];
    uint64_t cls8 = r[2 + s];
    uint64_t clsj = r[2 + (j >> 3)];
    unsigned tot = quad_sum(rrr_lane_prefix(T, cls8, s, j));
    t.rank = r0 + (tot & 0xFFFF);
    t.k = (unsigned)(clsj >> (8 * (j & 7))) & 0xFF;
    t.nr = rrr_field(v, r, r1 & ((UINT64_C(1) << 48) - 1), tot >> 16, T->space[t.k]);
    return t;
}

// What the lane-parallel phase needs to finish one select query.
struct SelTail
{
    const uint64_t * r; // superblock record
    uint64_t bstart;    // first bit of the block
    unsigned k, blen, rel, want; // class, valid bits, offset position in the superblock's stream, 0-based rank in block
};

// Cooperative half of select: superblock search over the record headers (directory of argument POSITIONS,
// interpolated probe, exact counts from the probed header, bisection every second late probe — the scheme of
// bv_device.hpp) and block location from the class bytes.  All four lanes return the same tail.
template <int BIT>
__device__ __forceinline__ SelTail rrr_select_head(const RrrView & v, const RrrTables * T, int s, uint64_t k0)
{
    const uint64_t total = BIT ? v.ones : v.n_bits - v.ones;
    const uint32_t sh = v.sel_shift, ps = v.sel_pshift;
    const uint64_t j = k0 >> sh;
    uint64_t lo_pos = (uint64_t)v.sel[BIT][j] << ps, lo_cnt = j << sh;
    uint64_t hi_pos = ((uint64_t)v.sel[BIT][j + 1] + 1) << ps, hi_cnt = (j + 1) << sh;
    if (hi_cnt > total)
        hi_cnt = total;
    const uint64_t * r;
    uint64_t g, before, cls8, r1;
    for (int tries = 0;; ++tries)
    { // invariant: lo_pos <= position(k0) < hi_pos, lo_cnt <= k0 < hi_cnt
        uint64_t span = hi_pos - lo_pos, p;
        if (tries >= 3 && (tries & 1))
            p = lo_pos + (span >> 1);
        else
            p = sel_interpolate(lo_pos, span, k0 - lo_cnt, hi_cnt - lo_cnt, sh);
        g = p / kRrrSB;
        if (g >= v.n_sb)
            g = v.n_sb - 1;
        r = v.rec + g * kRecWords;
        uint64_t r0 = r[0];
        r1 = r[1];
        cls8 = r[2 + s]; // same line: free, and needed as soon as the probe hits
        uint64_t ones_in = (r1 >> 48) & 0xFFF;
        uint64_t start = g * kRrrSB;
        uint64_t len_in = v.n_bits - start < kRrrSB ? v.n_bits - start : kRrrSB;
        before = BIT ? r0 : start - r0;
        uint64_t c = BIT ? ones_in : len_in - ones_in;
        if (k0 < before)
        {
            hi_pos = start;
            hi_cnt = before;
        }
        else if (k0 >= before + c)
        {
            lo_pos = start + kRrrSB;
            lo_cnt = before + c;
        }
        else
            break;
    }
    // inside superblock g: lane s owns classes [8s, 8s+8)
    const unsigned want = (unsigned)(k0 - before);
    const uint64_t b0 = g * kRrrK + 8 * (uint64_t)s;
    const bool full = (b0 + 8) * kRrrBS <= v.n_bits; // all eight blocks are complete 63-bit blocks
    unsigned my_args;
    if (BIT)
        my_args = sum_bytes8(cls8);
    else if (full)
        my_args = sum_bytes8(UINT64_C(0x3F3F3F3F3F3F3F3F) - cls8);
    else
    {
        my_args = 0;
        for (int t = 0; t < 8; ++t)
        {
            uint64_t bstart = (b0 + t) * kRrrBS;
            unsigned blen = bstart >= v.n_bits ? 0u : (unsigned)(v.n_bits - bstart < kRrrBS ? v.n_bits - bstart : kRrrBS);
            my_args += blen - ((unsigned)(cls8 >> (8 * t)) & 0xFF);
        }
    }
    unsigned my_bits = 0;
#pragma unroll
    for (int t = 0; t < 8; ++t)
        my_bits += T->space[(unsigned)(cls8 >> (8 * t)) & 0xFF];
    unsigned ex = quad_excl(my_args | (my_bits << 16), s);
    unsigned acc = ex & 0xFFFF, rel = ex >> 16;
    const bool owner = want >= acc && want < acc + my_args;
    // the owner lane walks its eight classes; packed = block-in-lane | k<<8 | blen<<16 | (want-acc)<<24, rel
    unsigned kk = 0, bl = 0, tt = 0;
    bool done = !owner;
#pragma unroll
    for (int t = 0; t < 8; ++t)
    {
        unsigned k = (unsigned)(cls8 >> (8 * t)) & 0xFF;
        uint64_t bstart = (b0 + t) * kRrrBS;
        unsigned blen =
            full ? kRrrBS : (bstart >= v.n_bits ? 0u : (unsigned)(v.n_bits - bstart < kRrrBS ? v.n_bits - bstart : kRrrBS));
        unsigned a = BIT ? k : blen - k;
        bool here = !done && want < acc + a;
        bool skip = !done && !here;
        if (here)
        {
            kk = k;
            bl = blen;
            tt = (unsigned)t + 8u * (unsigned)s;
            done = true;
        }
        if (skip)
        {
            acc += a;
            rel += T->space[k];
        }
    }
    // hand the owner's findings to the whole quad (exactly one lane contributes non-zero values)
    unsigned p0 = owner ? (tt | (kk << 8) | (bl << 16) | ((want - acc) << 24)) : 0u;
    unsigned p1 = owner ? rel : 0u;
    p0 = quad_sum(p0);
    p1 = quad_sum(p1);
    SelTail t;
    t.r = r;
    t.bstart = (g * kRrrK + (p0 & 0xFF)) * kRrrBS;
    t.k = (p0 >> 8) & 0xFF;
    t.blen = (p0 >> 16) & 0xFF;
    t.want = p0 >> 24;
    t.rel = p1;
    return t;
}


} // namespace sdslhip
