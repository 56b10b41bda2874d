// bv_device.hpp — device layout of a plain bit vector ("rank lines") and the per-query
// device functions every kernel in this library builds on.
//
// Layout: the bit vector is re-laid into 64-byte lines
//     line L = [ u64 ones_before(L*448) | 7 x u64 data words = bits [L*448, (L+1)*448) ]
// so that a rank touches EXACTLY ONE aligned 64-byte line (SDSL's rank_support_v5 touches a
// 16-byte directory entry plus up to two data lines, rank_support_v5.hpp:131-149; the
// reference's own precedent for interleaving is bit_vector_il, bit_vector_il.hpp:120-163).
// A query is served by a group of G = 4 adjacent lanes: lane s loads the 16-byte quarter
// [2s, 2s+1] of the line with one global_load_dwordx4 (the four lanes coalesce into one
// 64-byte request), popcounts its share under the query mask, and the quad is reduced with
// two DPP quad_perm adds — no LDS round trip, no divergence inside the quad.
#pragma once
#include "bits.hpp"

namespace sdslhip {

constexpr int kLW = 8;                    // u64 words per line (64 B)
constexpr int kG = kLW / 2;               // lanes cooperating on one query
constexpr int kDW = kLW - 1;              // data words per line
constexpr uint64_t kDB = 64ull * kDW;     // data bits per line (448)
constexpr unsigned kWave = 64;
constexpr unsigned kBlock = 256;          // threads per block in every query kernel
constexpr unsigned kQPB = kBlock / kG;    // queries per block per round

struct BvView
{
    const uint64_t * lines; // n_lines * kLW words
    uint64_t n_bits;
    uint64_t n_lines;       // n_bits / kDB + 1 rounded up to even (>= 2); trailing lines hold 0 valid bits
    uint64_t ones;
    const uint32_t * sel[2]; // sel[b][j] = (position of the b-bit of 0-based rank j<<sel_shift) >> sel_pshift; + sentinel
    uint32_t sel_shift;      // log2 of the sampling rate
    uint32_t sel_pshift;     // position quantisation so that samples fit 32 bits (0 for n_bits < 2^32)
    // Sparse stretches (the counterpart of select_support_mcl's "long" blocks, select_support_mcl.hpp:242-252): a sample
    // interval whose arguments lie more than kSelLongGap bits apart on average keeps the position of EVERY argument
    // (S + 1 entries, quantised like the samples).  lmask: one bit per interval; lidx: which long interval it is.
    // nullptr when the vector has no such interval (then the kernels do not even look).
    const uint32_t * lmask[2];
    const uint32_t * lidx[2];
    const uint32_t * lpos[2];
    // automatic dispatch of large batches (bv.hip): the direct batch kernels return at once when this word is non-zero — the
    // batch is then answered by the bucketed path, which was enqueued beside them.  nullptr everywhere else.
    const uint32_t * skip_if;
};
constexpr uint64_t kSelLongGap = 512;

// samples of the interval of argument k, refined to the argument itself where the interval is a long one
struct SelSamples
{
    uint32_t s0, s1;
    bool fine; // s0 / s1 bracket the k-th argument itself
};
template <int BIT>
__device__ __forceinline__ SelSamples sel_samples(const BvView & bv, uint64_t k)
{
    const uint64_t j = k >> bv.sel_shift;
    SelSamples r;
    r.fine = false;
    if (bv.lmask[BIT] && ((bv.lmask[BIT][j >> 5] >> (j & 31)) & 1u))
    {
        const uint64_t base = (uint64_t)bv.lidx[BIT][j] * ((UINT64_C(1) << bv.sel_shift) + 1) + (k & ((UINT64_C(1) << bv.sel_shift) - 1));
        r.s0 = bv.lpos[BIT][base];
        r.s1 = bv.lpos[BIT][base + 1];
        r.fine = true;
        return r;
    }
    r.s0 = bv.sel[BIT][j];
    r.s1 = bv.sel[BIT][j + 1];
    return r;
}

struct Pair
{
    uint64_t a, b; // line words 2s and 2s+1 of sub-lane s
};

// ---- quad primitives (G == 4): DPP quad_perm, all four lanes of the quad must be active ----
__device__ __forceinline__ unsigned quad_xor1(unsigned v)
{
    return (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, 0xB1, 0xF, 0xF, true); // quad_perm:[1,0,3,2]
}
__device__ __forceinline__ unsigned quad_xor2(unsigned v)
{
    return (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, 0x4E, 0xF, 0xF, true); // quad_perm:[2,3,0,1]
}
__device__ __forceinline__ unsigned quad_shr1(unsigned v)
{
    return (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, 0x90, 0xF, 0xF, true); // quad_perm:[0,0,1,2]
}
__device__ __forceinline__ unsigned quad_shr2(unsigned v)
{
    return (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, 0x44, 0xF, 0xF, true); // quad_perm:[0,1,0,1]
}
__device__ __forceinline__ unsigned quad_bcast0(unsigned v)
{
    return (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, 0x00, 0xF, 0xF, true); // quad_perm:[0,0,0,0]
}
__device__ __forceinline__ uint64_t quad_bcast0_u64(uint64_t v)
{
    unsigned lo = quad_bcast0((unsigned)v), hi = quad_bcast0((unsigned)(v >> 32));
    return ((uint64_t)hi << 32) | lo;
}
// sum over the quad, result in all four lanes
__device__ __forceinline__ unsigned quad_sum(unsigned v)
{
    v += quad_xor1(v);
    v += quad_xor2(v);
    return v;
}
// exclusive prefix sum over the quad (lane s gets v[0]+..+v[s-1])
__device__ __forceinline__ unsigned quad_excl(unsigned v, int s)
{
    unsigned t = v;
    unsigned u = quad_shr1(t);
    if (s >= 1)
        t += u;
    u = quad_shr2(t);
    if (s >= 2)
        t += u;
    return t - v;
}

// ---- line access ------------------------------------------------------------------------
template <bool NT>
__device__ __forceinline__ Pair load_pair(const uint64_t * lines, uint64_t L, int s)
{
    typedef unsigned long long v2u64 __attribute__((ext_vector_type(2)));
    const v2u64 * p = reinterpret_cast<const v2u64 *>(lines + L * kLW) + s;
    v2u64 v;
    if (NT)
        v = __builtin_nontemporal_load(p);
    else
        v = *p;
    Pair r;
    r.a = v.x;
    r.b = v.y;
    return r;
}

// Number of ones among this lane's share of the line strictly below in-line bit offset `off`
// (off in [0, kDB)).  Lane s owns data words 2s-1 (in .a, s>0) and 2s (in .b); .a of lane 0 is
// the header and never contributes.
__device__ __forceinline__ unsigned lane_ones_below(Pair w, int s, unsigned off)
{
    // bits of the lane's 128-bit share below the offset: t = off - 64 * (2s - 1), clamped to [0, 64] per word; the
    // count of the low t bits of x is popc(x << (64 - t)) for t >= 1
    const int t = (int)off - 64 * (2 * s - 1);
    const int ta = t < 0 ? 0 : (t > 64 ? 64 : t), tb = t < 64 ? 0 : (t > 128 ? 64 : t - 64);
    const uint64_t xa = s == 0 ? UINT64_C(0) : w.a; // lane 0 holds the header there
    const unsigned ca = ta ? popc64(xa << (64 - ta)) : 0u;
    const unsigned cb = tb ? popc64(w.b << (64 - tb)) : 0u;
    return ca + cb;
}

// line index and in-line offset of a bit position.  Positions below 2^38 (every wavelet tree over < 2^35 symbols, every
// vector below 32 GiB) take a 32-bit division by 7 instead of the 64-bit division by 448; `small` must be uniform.
__device__ __forceinline__ void line_of(uint64_t pos, bool small, uint64_t & L, unsigned & off)
{
    if (small)
    {
        const uint32_t w = (uint32_t)(pos >> 6), l = w / 7u;
        L = l;
        off = ((w - 7u * l) << 6) | ((uint32_t)pos & 63u);
    }
    else
    {
        L = pos / kDB;
        off = (unsigned)(pos - L * kDB);
    }
}

// quad_rank1 with the line index / offset already known
__device__ __forceinline__ uint64_t quad_rank1_at(Pair w, int s, unsigned off)
{
    unsigned part = lane_ones_below(w, s, off);
    unsigned tot = quad_sum(part);
    uint64_t hdr = quad_bcast0_u64(w.a);
    return hdr + tot;
}

// rank_1(idx) for idx in [0, n_bits]; all four lanes of the quad call it with the same idx and
// all four get the result.  `w` is the quad's line for idx (load_pair(..., idx / kDB, s)).
__device__ __forceinline__ uint64_t quad_rank1(Pair w, int s, uint64_t idx, uint64_t L)
{
    unsigned off = (unsigned)(idx - L * kDB);
    unsigned part = lane_ones_below(w, s, off);
    unsigned tot = quad_sum(part);
    uint64_t hdr = quad_bcast0_u64(w.a);
    return hdr + tot;
}

// rank_1(pos) by ONE lane (construction passes that walk many neighbouring positions; the query kernels use the quad
// forms above).  pos in [0, n_bits]; *bit receives the bit at pos when asked for (then pos < n_bits).
__device__ __forceinline__ uint64_t lane_rank1(const uint64_t * lines, uint64_t pos, unsigned * bit)
{
    const uint64_t L = pos / kDB;
    const unsigned off = (unsigned)(pos - L * kDB), wi = off >> 6;
    const uint64_t * ln = lines + L * kLW;
    uint64_t r = ln[0];
    for (unsigned j = 0; j < wi; ++j)
        r += popc64(ln[1 + j]);
    const uint64_t w = ln[1 + wi];
    if (bit)
        *bit = (unsigned)((w >> (off & 63)) & 1);
    return r + popc64(w & lo_set(off & 63));
}

// Mask of the valid data bits of word `d` (0..6) of line L given the vector length.
__device__ __forceinline__ uint64_t valid_mask(uint64_t n_bits, uint64_t L, int d)
{
    uint64_t start = L * kDB + 64ull * (uint64_t)d;
    if (start >= n_bits)
        return 0;
    uint64_t rem = n_bits - start;
    return rem >= 64 ? ~UINT64_C(0) : lo_set((unsigned)rem);
}

// Turn a lane's pair into "argument words" for select on bit BIT: for BIT==1 the data as is,
// for BIT==0 the complement restricted to valid positions.  The header slot (.a of lane 0) is
// cleared so that popcounts over the pair count only data.
template <int BIT>
__device__ __forceinline__ Pair arg_words(Pair w, int s, uint64_t n_bits, uint64_t L)
{
    Pair r;
    if (BIT)
    {
        r.a = s == 0 ? 0 : w.a;
        r.b = w.b;
    }
    else if ((L + 1) * kDB <= n_bits)
    { // every position of the line exists (all lines but the last one or two): no masking (quad-uniform branch)
        r.a = s == 0 ? 0 : ~w.a;
        r.b = ~w.b;
    }
    else
    {
        r.a = s == 0 ? 0 : (~w.a & valid_mask(n_bits, L, 2 * s - 1));
        r.b = ~w.b & valid_mask(n_bits, L, 2 * s);
    }
    return r;
}

// Select inside one line: the quad holds the argument words of line L; r0 is the 0-based rank
// of the wanted argument inside the line (0 <= r0 < quad_sum(popc)).  Exactly one lane gets
// `mine = true` and the absolute bit position.
__device__ __forceinline__ uint64_t quad_select_in_line(Pair aw, int s, uint64_t L, unsigned r0, bool & mine)
{
    unsigned ca = popc64(aw.a), cb = popc64(aw.b);
    unsigned cl = ca + cb;
    unsigned ex = quad_excl(cl, s);
    mine = (r0 >= ex) && (r0 < ex + cl);
    unsigned rr = r0 - ex; // only meaningful if mine
    uint64_t pos = 0;
    if (mine)
    {
        if (rr < ca)
            pos = L * kDB + 64ull * (uint64_t)(2 * s - 1) + sel64(aw.a, rr + 1);
        else
            pos = L * kDB + 64ull * (uint64_t)(2 * s) + sel64(aw.b, rr - ca + 1);
    }
    return pos;
}

// ---- select ------------------------------------------------------------------------------------
// Directory: sel[BIT][j] = (position of the argument of rank j << sel_shift) >> sel_pshift, plus a
// sentinel n_bits >> sel_pshift.  The probe position is interpolated between the two surrounding
// (position, count) pairs and ONE aligned 128-byte window (two rank lines, the unit a fabric request
// moves anyway) is fetched; the window's two headers give exact counts, so a miss tightens the bracket
// to (window edge, exact count) and the next guess is interpolated again; every second late probe
// bisects, which bounds the worst case at O(log n) probes.  The stages are separate functions so that
// a kernel can keep several queries per quad in flight (the chain idx -> samples -> window is
// latency-bound, not request-bound).
struct SelBracket
{ // invariant: lo_pos <= position(k) < hi_pos,  lo_cnt <= k < hi_cnt
    uint64_t lo_pos, lo_cnt, hi_pos, hi_cnt;
};

template <int BIT>
__device__ __forceinline__ SelBracket sel_bracket(const BvView & bv, uint64_t k, uint32_t s0, uint32_t s1, bool fine = false)
{
    const uint32_t sh = bv.sel_shift, ps = bv.sel_pshift;
    const uint64_t j = k >> sh;
    const uint64_t total = BIT ? bv.ones : bv.n_bits - bv.ones;
    SelBracket b;
    b.lo_pos = (uint64_t)s0 << ps;
    b.lo_cnt = fine ? k : j << sh;
    b.hi_pos = ((uint64_t)s1 + 1) << ps;
    b.hi_cnt = fine ? k + 1 : (j + 1) << sh;
    if (b.hi_cnt > total)
        b.hi_cnt = total;
    return b;
}
template <int BIT>
__device__ __forceinline__ SelBracket sel_bracket(const BvView & bv, uint64_t k, const SelSamples & sm)
{
    return sel_bracket<BIT>(bv, k, sm.s0, sm.s1, sm.fine);
}

// position estimate inside a bracket: lo_pos + (k - lo_cnt) / (hi_cnt - lo_cnt) * span.  Only a probe hint
// (exact header counts decide), so a float quotient replaces the 64-bit division; the first probe of a query
// has a power-of-two denominator and stays exact.
__device__ __forceinline__ uint64_t sel_interpolate(uint64_t lo_pos, uint64_t span, uint64_t num, uint64_t den,
                                                    uint32_t first_shift)
{
    if (den == (UINT64_C(1) << first_shift))
        return lo_pos + ((num * span) >> first_shift); // num < 2^20, span < 2^40: no overflow
    const float f = (float)num * __builtin_amdgcn_rcpf((float)(den ? den : 1)); // 1-ulp reciprocal: a hint needs no more
    uint64_t off = (uint64_t)(f * (float)span);
    return lo_pos + (off >= span ? span - 1 : off);
}

// window (pair of lines) to probe next
__device__ __forceinline__ uint64_t sel_guess(const BvView & bv, const SelBracket & b, uint64_t k, int tries)
{
    const uint64_t span = b.hi_pos - b.lo_pos;
    uint64_t p;
    if (tries >= 3 && (tries & 1))
        p = b.lo_pos + (span >> 1);
    else
        p = sel_interpolate(b.lo_pos, span, k - b.lo_cnt, b.hi_cnt - b.lo_cnt, bv.sel_shift);
    // window index = p / 896 = (p >> 7) / 7: positions are below 2^40 (select directory limit), so (p >> 8) fits 32 bits
    // and the division by 7 runs on 32-bit multiplies: q = (2 * (p >> 8) + bit 7 of p) / 7
    const uint32_t hi = (uint32_t)(p >> 8);
    uint64_t W = ((uint64_t)(hi / 7u) << 1) + (((hi % 7u) * 2u + ((uint32_t)(p >> 7) & 1u)) / 7u);
    const uint64_t last_win = (bv.n_lines >> 1) - 1; // n_lines is even
    return W > last_win ? last_win : W;
}

// Evaluate window W given its two lines.  Returns true when the argument lies inside (then exactly one
// lane has mine == true and pos); otherwise tightens the bracket.
template <int BIT>
__device__ __forceinline__ bool sel_eval(const BvView & bv, int s, uint64_t k, uint64_t W, Pair wa, Pair wb,
                                         SelBracket & b, bool & mine, uint64_t & pos)
{
    const uint64_t LA = 2 * W, LB = LA + 1;
    uint64_t h1a = quad_bcast0_u64(wa.a), h1b = quad_bcast0_u64(wb.a);
    uint64_t ha = BIT ? h1a : LA * kDB - h1a; // arguments before line A
    uint64_t hb = BIT ? h1b : LB * kDB - h1b; // ... before line B (== ha + count(A))
    Pair awa = arg_words<BIT>(wa, s, bv.n_bits, LA);
    Pair awb = arg_words<BIT>(wb, s, bv.n_bits, LB);
    unsigned cb = quad_sum(popc64(awb.a) + popc64(awb.b));
    mine = false;
    if (k < ha)
    {
        b.hi_pos = LA * kDB;
        b.hi_cnt = ha;
        return false;
    }
    if (k >= hb + cb)
    {
        b.lo_pos = (LB + 1) * kDB;
        b.lo_cnt = hb + cb;
        return false;
    }
    const bool inB = k >= hb; // quad-uniform
    Pair aw = inB ? awb : awa;
    pos = quad_select_in_line(aw, s, inB ? LB : LA, (unsigned)(k - (inB ? hb : ha)), mine);
    return true;
}

// sel_eval with the bit value chosen at run time (quad-uniform): the wavelet-tree select alternates between ones and
// zeros from level to level, and a wave whose quads disagree would otherwise execute both instantiations
__device__ __forceinline__ bool sel_eval_rt(const BvView & bv, int s, bool one, uint64_t k, uint64_t W, Pair wa, Pair wb,
                                            SelBracket & b, bool & mine, uint64_t & pos)
{
    const uint64_t LA = 2 * W, LB = LA + 1;
    const uint64_t h1a = quad_bcast0_u64(wa.a), h1b = quad_bcast0_u64(wb.a);
    const uint64_t ha = one ? h1a : LA * kDB - h1a;
    const uint64_t hb = one ? h1b : LB * kDB - h1b;
    Pair awa, awb;
    if (one || (LB + 1) * kDB <= bv.n_bits)
    { // all positions of both lines exist (or ones are wanted: padding bits are zero): complement by mask
        const uint64_t cm = one ? UINT64_C(0) : ~UINT64_C(0);
        awa.a = s == 0 ? 0 : wa.a ^ cm;
        awa.b = wa.b ^ cm;
        awb.a = s == 0 ? 0 : wb.a ^ cm;
        awb.b = wb.b ^ cm;
    }
    else
    {
        awa = arg_words<0>(wa, s, bv.n_bits, LA);
        awb = arg_words<0>(wb, s, bv.n_bits, LB);
    }
    const unsigned cb = quad_sum(popc64(awb.a) + popc64(awb.b));
    mine = false;
    if (k < ha)
    {
        b.hi_pos = LA * kDB;
        b.hi_cnt = ha;
        return false;
    }
    if (k >= hb + cb)
    {
        b.lo_pos = (LB + 1) * kDB;
        b.lo_cnt = hb + cb;
        return false;
    }
    const bool inB = k >= hb; // quad-uniform
    const Pair aw = inB ? awb : awa;
    pos = quad_select_in_line(aw, s, inB ? LB : LA, (unsigned)(k - (inB ? hb : ha)), mine);
    return true;
}

// exactly one lane of the quad has mine == true: give its value to all four
__device__ __forceinline__ uint64_t quad_gather_u64(uint64_t v, bool mine)
{
    uint64_t x = mine ? v : 0;
    unsigned lo = quad_sum((unsigned)x), hi = quad_sum((unsigned)(x >> 32));
    return ((uint64_t)hi << 32) | lo;
}

// One query per quad, probes issued one after the other (used by the wavelet-tree select cascade).
template <int BIT, bool NT>
__device__ __forceinline__ uint64_t quad_select(const BvView & bv, int s, uint64_t k, bool & mine)
{
    SelBracket b = sel_bracket<BIT>(bv, k, sel_samples<BIT>(bv, k));
    uint64_t pos = 0;
    for (int tries = 0;; ++tries)
    {
        uint64_t W = sel_guess(bv, b, k, tries);
        Pair wa = load_pair<NT>(bv.lines, 2 * W, s);
        Pair wb = load_pair<NT>(bv.lines, 2 * W + 1, s);
        if (sel_eval<BIT>(bv, s, k, W, wa, wb, b, mine, pos))
            return pos;
    }
}

} // namespace sdslhip
