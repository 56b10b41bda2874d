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
    uint64_t n_lines;       // n_bits / kDB + 1  (always >= 1; the last line may hold 0 valid bits)
    uint64_t ones;
    const uint32_t * sel[2]; // sel[b][j] = line holding the b-bit of 0-based rank j<<sel_shift; +1 sentinel
    uint32_t sel_shift;      // log2 of the sampling rate
};

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
    const int wi = (int)(off >> 6);
    const unsigned bi = off & 63;
    const int db = 2 * s, da = 2 * s - 1;
    uint64_t mb = db < wi ? ~UINT64_C(0) : (db == wi ? lo_set(bi) : UINT64_C(0));
    uint64_t ma = (s == 0) ? UINT64_C(0) : (da < wi ? ~UINT64_C(0) : (da == wi ? lo_set(bi) : UINT64_C(0)));
    return popc64(w.b & mb) + popc64(w.a & ma);
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

// Find the line holding the BIT-argument of 0-based rank k (k < total args) and select inside it.
// All four lanes pass identical (k).  Returns the position in the lane with mine==true.
// Search = one interpolated probe between the two surrounding samples, then capacity-bounded
// neighbour steps, then bisection; every probe is one 64-byte line.
template <int BIT, bool NT>
__device__ __forceinline__ uint64_t quad_select(const BvView & bv, int s, uint64_t k, bool & mine)
{
    const uint32_t sh = bv.sel_shift;
    const uint64_t j = k >> sh;
    uint64_t lo = bv.sel[BIT][j], hi = bv.sel[BIT][j + 1];
    uint64_t g = lo + (((hi - lo) * (k - (j << sh))) >> sh);
    int tries = 0;
    for (;;)
    {
        Pair w = load_pair<NT>(bv.lines, g, s);
        uint64_t h1 = quad_bcast0_u64(w.a);
        uint64_t h = BIT ? h1 : g * kDB - h1; // arguments before line g
        Pair aw = arg_words<BIT>(w, s, bv.n_bits, g);
        unsigned c = quad_sum(popc64(aw.a) + popc64(aw.b));
        if (k < h)
        { // target is in an earlier line; d arguments lie in [target line, g) so it is >= ceil(d/448) lines back
            uint64_t d = h - k;
            hi = g - (d + kDB - 1) / kDB;
        }
        else if (k >= h + c)
        {
            uint64_t d = k - (h + c); // arguments strictly between line g and the target's
            lo = g + 1 + d / kDB;
        }
        else
        {
            return quad_select_in_line(aw, s, g, (unsigned)(k - h), mine);
        }
        ++tries;
        if (lo >= hi)
            g = lo;
        else if (tries <= 2)
            g = (k < h) ? hi : lo;
        else
            g = lo + ((hi - lo) >> 1);
    }
}

} // namespace sdslhip
