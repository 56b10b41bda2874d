// wt_device.hpp — device view of a byte wavelet tree (wt_huff<bit_vector, rank_support_v5<>>)
// and the quad-cooperative traversals shared by wt.hip (rank/access/inverse_select/select) and
// fm.hip (backward search / count).
//
// The node table of SDSL's _byte_tree (wt_helper.hpp:200-327: BFS node array with bv_pos,
// bv_pos_rank, parent, child[2]; m_c_to_leaf[256]; m_path[256]) is tiny (<= 511 nodes), so each
// workgroup stages it in LDS once; the concatenated bit vector (wt_pc.hpp:90) lives in HBM as
// rank lines (bv_device.hpp).  One rank cascade level = ONE 64-byte line fetch.
#pragma once
#ifndef SDSL_HIP_FSEL_LOG
#define SDSL_HIP_FSEL_LOG 8
#endif
#ifndef SDSL_HIP_FUSED_K
#define SDSL_HIP_FUSED_K 4 // tree levels per fused step: 4 (16 slots, 192 - 4 * spare positions per line; the default) or 3 (8 slots, 256 positions)
#endif
#ifndef SDSL_HIP_FUSED_SPARE
#define SDSL_HIP_FUSED_SPARE 2 // 16-ary lines: width of the line's relative counts = 16 + spare bits (0, 2 or 4)
#endif
#include "bv_device.hpp"
#include "rrr_device.hpp"

namespace sdslhip {

constexpr int kWtMaxNodes = 512;
constexpr uint16_t kWtUndef = 0xFFFF;

struct WtTables // global-memory image, identical layout in LDS
{
    uint64_t bv_pos[kWtMaxNodes];      // start of the node's slice in the WT bit vector
    uint64_t bv_pos_rank[kWtMaxNodes]; // inner: rank_1(bv_pos); leaf: the symbol (wt_pc.hpp:355-356)
    uint64_t path[256];                // bits 0..55 path LSB-first from the root, bits 56..63 length
    uint16_t child[kWtMaxNodes][2];
    uint16_t parent[kWtMaxNodes];
    uint16_t c_to_leaf[256];
};

// Sequences of 2^32 symbols and more: a line's header keeps the LOW 32 bits of its counts and the walk adds them up modulo 2^32,
// exactly as for a small sequence; the PLACES where a count reaches a multiple of 2^32 are listed here — entry e: the count of slot
// (cross_key & 7) in fused node (cross_key >> 3), taken at place cross_pos[e] = (absolute line << 8) + offset inside the line, is
// the first to be j * 2^32 or more.  Counts grow with the place, so the high part of a count is the number of its (node, slot)'s
// entries at or in front of its place: a handful for a text of a few 2^32 symbols.  Sequences below 2^32 symbols have no entries and
// never look (`wt.size >> 32` is kernel-uniform).
constexpr unsigned kFusedMaxCross = 64;
struct WtFusedTables // node tables of the fused layout (below); staged in LDS by the kernels that walk it
{
    uint32_t fline[kWtMaxNodes]; // first line of the node's sequence (nodes at depth 0, 3, 6, ...)
    uint32_t n_cross;
    uint16_t cross_key[kFusedMaxCross];
    uint64_t cross_pos[kFusedMaxCross];
};

// select on the fused layout: for every fused node u and slot t the directory lists the position (inside u's sequence)
// of every 256th occurrence of t and ends with u's size.  off[root_id[u]][t] is where the list of (u, t) starts in
// WtView::f_sel, cnt[..][t] the number of occurrences.
constexpr unsigned kFselLog = SDSL_HIP_FSEL_LOG; // the directory lists every 2^kFselLog-th occurrence
constexpr int kFselMaxRoots = SDSL_HIP_FUSED_K == 3 ? 80 : 40; // a balanced tree over 256 symbols has 1 + 8 + 64 = 73 (16-ary: 1 + 16)
constexpr uint32_t kFselNone = 0xFFFFFFFFu;
struct WtFusedSelTables
{
    uint32_t off[kFselMaxRoots][1u << SDSL_HIP_FUSED_K];
    uint32_t cnt[kFselMaxRoots][1u << SDSL_HIP_FUSED_K];
    uint16_t root_id[kWtMaxNodes];
    uint8_t cnt_hi[kFselMaxRoots][1u << SDSL_HIP_FUSED_K]; // bits 32..39 of cnt (sequences of 2^32 symbols and more: 64-bit directory entries)
};

// The fused layout BY FUSED NODE, for the walks that only ever stand on one (inverse_select: the LF walks): first line and, per slot,
// what comes next — the fused node below, or the symbol where the slot ends at a leaf.  One LDS read per step instead of the binary
// node table's descent (kFK dependent reads), and 1.5 KiB of LDS instead of the node table's 13.5.
constexpr uint16_t kFWalkLeaf = 0x8000u, kFWalkNone = 0xFFFFu;
struct WtFusedWalk
{
    uint32_t n_roots, pad_;
    uint32_t rline[kFselMaxRoots];
    uint16_t succ[kFselMaxRoots][1u << SDSL_HIP_FUSED_K]; // next fused node | kFWalkLeaf + symbol | kFWalkNone
};

// The fused layout BY SYMBOL, for rank(i, c): the steps of c's path spelled out as (first line of the fused node, slot) — what the flat
// count kernel keeps per byte (fm_device.hpp: FmCountTab), without the FM-index's C[].  meta[c] = first step | steps << 16 (0: c does
// not occur), steps[] = first line (28 bits) | slot << 28.
constexpr unsigned kWtMaxSteps = 1280;
struct WtStepTab
{
    uint32_t meta[256];
    uint32_t steps[kWtMaxSteps];
};

struct WtView
{
    BvView bv;   // backend 0: the bit vector as rank lines
    RrrView rrr; // backend 1: the bit vector as an rrr_vector<63> (wt_huff<rrr_vector<63>>)
    uint32_t backend;
    const WtTables * tables;
    uint64_t size;  // number of symbols
    uint64_t sigma; // effective alphabet size
    uint32_t n_nodes;
    const uint64_t * f_lines; // fused (8-ary / 16-ary) layout of the same tree, nullptr if not built
    const WtFusedTables * f_tables;
    const uint32_t * f_super;    // 16-ary lines: the superblocks' counts, [superblock][slot]: 32-bit, or — sequences of 2^32 symbols and
                                 // more — 64-bit (one record = one 128-byte line either way it is read: a word per step), else nullptr
    const struct WtStepTab * f_steps;  // the layout by symbol (nullptr: paths longer than the table, or 8-ary lines of 2^32 symbols and more)
    const struct WtFusedWalk * f_walk; // the layout by fused node (nullptr: more fused nodes than the table holds, or 8-ary lines of 2^32 symbols and more)
    const uint32_t * f_sel;                      // select directory of the fused layout (below; 64-bit entries for 2^32 symbols and more), nullptr if not built
    const struct WtFusedSelTables * f_sel_tables;
};

// cooperative copy of the tables into LDS (all threads of the block)
__device__ __forceinline__ void wt_stage_tables(WtTables * lds, const WtTables * g)
{
    const uint64_t * src = reinterpret_cast<const uint64_t *>(g);
    uint64_t * dst = reinterpret_cast<uint64_t *>(lds);
    constexpr unsigned n = sizeof(WtTables) / 8;
    for (unsigned i = threadIdx.x; i < n; i += blockDim.x)
        dst[i] = src[i];
    __syncthreads();
}

__device__ __forceinline__ void wt_stage_fused(WtFusedTables * lds, const WtView & wt)
{
    if (wt.f_lines) // kernel-uniform
    {
        const uint64_t * src = reinterpret_cast<const uint64_t *>(wt.f_tables);
        uint64_t * dst = reinterpret_cast<uint64_t *>(lds);
        constexpr unsigned n = sizeof(WtFusedTables) / 8;
        for (unsigned i = threadIdx.x; i < n; i += blockDim.x)
            dst[i] = src[i];
    }
    __syncthreads();
}

// wt_pc::rank(i, c) (wt_pc.hpp:371-399) by one quad; i in [0, size].
template <bool NT>
__device__ __forceinline__ uint64_t quad_wt_rank(const WtView & wt, const WtTables * T, int s, uint64_t i, unsigned c)
{
    if (T->c_to_leaf[c] == kWtUndef)
        return 0; // c does not occur (:374-377)
    if (wt.sigma == 1)
        return i; // (:378-381)
    uint64_t p = T->path[c];
    unsigned len = (unsigned)(p >> 56);
    uint64_t result = i;
    unsigned v = 0;
    const bool small = wt.bv.n_bits < (UINT64_C(1) << 38);
    for (unsigned l = 0; l < len && result; ++l, p >>= 1)
    {
        uint64_t pos = T->bv_pos[v] + result;
        uint64_t L;
        unsigned off;
        line_of(pos, small, L, off);
        Pair w = load_pair<NT>(wt.bv.lines, L, s);
        uint64_t r = quad_rank1_at(w, s, off) - T->bv_pos_rank[v];
        unsigned bit = (unsigned)(p & 1);
        result = bit ? r : result - r;
        v = T->child[v][bit];
    }
    return result;
}

// Two rank cascades for the same symbol, walked level by level with both line loads in flight:
// a = rank(ia, c), b = rank(ib, c).  This is the inner step of backward_search
// (suffix_array_algorithm.hpp:195-196).  When both positions fall into the same line (the usual
// case once the SA interval is narrow) the line is fetched once.
// one level of both cascades: node v, path bit `bit`; a <= b are offsets inside v's slice
template <bool NT>
__device__ __forceinline__ void quad_wt_rank2_level(const WtView & wt, const WtTables * T, int s, unsigned & v,
                                                    unsigned bit, uint64_t & a, uint64_t & b)
{
    const uint64_t base = T->bv_pos[v], brank = T->bv_pos_rank[v];
    const bool small = wt.bv.n_bits < (UINT64_C(1) << 38);
    uint64_t pa = base + a, pb = base + b;
    uint64_t La, Lb;
    unsigned oa, ob;
    line_of(pa, small, La, oa);
    line_of(pb, small, Lb, ob);
    Pair wb = load_pair<NT>(wt.bv.lines, Lb, s);
    Pair wa = wb;
    if (La != Lb) // quad-uniform
        wa = load_pair<NT>(wt.bv.lines, La, s);
    // rank at bv_pos+0 equals bv_pos_rank, so a == 0 stays 0 without a special case
    uint64_t ra = quad_rank1_at(wa, s, oa) - brank;
    uint64_t rb = quad_rank1_at(wb, s, ob) - brank;
    a = bit ? ra : a - ra;
    b = bit ? rb : b - rb;
    v = T->child[v][bit];
}

// ---- fused layout: several tree levels per memory access ----------------------------------------
// A rank cascade is a chain of DEPENDENT random line fetches, one per tree level, and the fetch rate is what bounds it
// (DESIGN.md §3.1).  The fused layout stores the same tree a second time with kFK binary levels collapsed into one
// 2^kFK-ary level: every node u at depth 0, kFK, 2 kFK, ... owns the sequence of the symbols routed through it (the order of its
// slice of the binary tree), each symbol reduced to a SLOT = its next kFK path bits (a leaf reached earlier pads with zeros).
// One fetch answers "how many of the first i symbols of u continue along slot t" = the offset inside the node kFK levels down;
// the answers are those of the binary cascade, level for level.
//
// The 16-ary form (SDSL_HIP_FUSED_K = 4, the default): 16 slots, 128-byte line = 4 sections of 32 bytes, one per lane of the quad;
// section s = [four 16-bit counts of slots 4s .. 4s+3 | three words, word j = positions 16j .. 16j+15 of the section: bit k of their
// slots in bits 16k .. 16k+15].  The counts are relative to the line's SUPERBLOCK (2^kFSuperLog lines), whose 16 absolute counts live
// in WtView::f_super — a table of 0.05 % of the lines that the caches hold, read beside the line.  16-ary Huffman: 1.23 steps per symbol
// of the bench text against the 8-ary tree's 1.63, at 5.6 bits per position and step against 4.
//
// The 8-ary form (SDSL_HIP_FUSED_K = 3; rounds 2-5): section s = [count[2s] | count[2s+1] << 32, plane 0, plane 1, plane 2] for
// positions 64s .. 64s+63 of the line's 256: count[t] = occurrences of slot t in the node's sequence before the line (its low 32
// bits: WtFusedTables lists where a count passes 2^32), plane k = bit k of the slots.
constexpr unsigned kFK = SDSL_HIP_FUSED_K; // tree levels per fused step
static_assert(kFK == 3 || kFK == 4, "SDSL_HIP_FUSED_K is 3 or 4");
constexpr unsigned kFSlots = 1u << kFK;
// 16-ary lines: a section's relative counts are 16 + kFSpare bits wide — the low 16 in the header word, the rest in the top bits of
// the third word's four fields, which then hold 16 - kFSpare positions.  Wider counts reach further, so the superblocks are longer and
// their table smaller: 1.7 MB for the 1 GiB index at 16 bits, where a fifth of its reads missed the L2 (the lines stream through it)
// and went to the fabric after all; 0.45 MB at 18 bits.
constexpr unsigned kFSpare = kFK == 3 ? 0u : (unsigned)SDSL_HIP_FUSED_SPARE;
static_assert(kFSpare == 0 || kFSpare == 2 || kFSpare == 4, "SDSL_HIP_FUSED_SPARE is 0, 2 or 4");
constexpr unsigned kFLane = kFK == 3 ? 64u : 48u - kFSpare; // positions per section (lane of the quad)
constexpr unsigned kFusedPos = 4 * kFLane;                  // positions per line
constexpr unsigned kFusedWords = 16;
constexpr unsigned kFSuperLog = 8 + kFSpare; // 16-ary: lines per superblock, 2^kFSuperLog * kFusedPos < 2^(16 + kFSpare)
static_assert(kFK == 3 || (UINT64_C(1) << kFSuperLog) * kFusedPos < (UINT64_C(1) << (16 + kFSpare)), "relative counts must fit");
constexpr uint32_t kFWord2Mask = 0xFFFFu >> kFSpare; // the positions of a section's third word

// line and offset of position i of a fused node's sequence: exact for every 64-bit i (wt_pc.hpp:371-399 is size_type throughout).
// 16-ary: kFusedPos is a multiple of 8, so below 2^35 the quotient is a 32-bit division by a constant after a shift — the form every
// walk takes on every text this part has met; positions from 2^35 on (a root node of 32 Gi symbols and more) take the 64-bit
// division.  kFusedLine32Bits ties the fast path's range to the shift: tests/cpp/fused_addr_check.cpp walks both sides of it.
constexpr unsigned kFusedLine32Bits = 35; // (i >> 3) fits 32 bits below this
static_assert(kFK == 3 || kFusedPos % 8 == 0, "the 32-bit form of fused_line divides (i >> 3) by kFusedPos / 8");
__device__ __host__ __forceinline__ uint64_t fused_line(uint64_t i)
{
    if constexpr (kFK == 3)
        return i >> 8;
    else
    {
        if (__builtin_expect((i >> kFusedLine32Bits) == 0, 1))
            return (uint64_t)((uint32_t)(i >> 3) / (kFusedPos >> 3));
        return i / kFusedPos;
    }
}
__device__ __host__ __forceinline__ unsigned fused_off(uint64_t i, uint64_t line)
{
    if constexpr (kFK == 3)
        return (unsigned)i & 255u;
    else
        return (unsigned)i - (unsigned)line * kFusedPos;
}
__device__ __host__ __forceinline__ uint64_t fused_lines_for(uint64_t size) // lines of a node of `size` positions (position `size` is addressable)
{
    return size / kFusedPos + 1;
}

struct FSec
{
    uint64_t h, p0, p1, p2; // header word; 8-ary: planes 0..2 of 64 positions; 16-ary: three words of 16 positions x 4 planes
};

template <bool NT>
__device__ __forceinline__ FSec load_fsec(const uint64_t * fl, uint64_t L, int s)
{
    typedef unsigned long long v2u64 __attribute__((ext_vector_type(2)));
    const v2u64 * ptr = reinterpret_cast<const v2u64 *>(fl + L * kFusedWords) + 2 * s;
    v2u64 a, b;
    if (NT) // (tried for every walk: non-temporal line loads cost a fifth of the rate, 16-ary or 8-ary)
    {
        a = __builtin_nontemporal_load(ptr);
        b = __builtin_nontemporal_load(ptr + 1);
    }
    else
    {
        a = ptr[0];
        b = ptr[1];
    }
    FSec x;
    x.h = a.x;
    x.p0 = a.y;
    x.p1 = b.x;
    x.p2 = b.y;
    return x;
}

// 16-ary: the 16 positions of word w that hold slot t, as a 16-bit mask.  xm = fsec16_xor(t): every plane's field inverted where the
// slot's bit is 0, so that a position matches where all four fields have a 1.
__device__ __host__ __forceinline__ uint64_t fsec16_xor(unsigned t)
{
    return ((t & 1) ? 0 : UINT64_C(0xFFFF)) | ((t & 2) ? 0 : UINT64_C(0xFFFF0000)) | ((t & 4) ? 0 : UINT64_C(0xFFFF00000000)) |
           ((t & 8) ? 0 : UINT64_C(0xFFFF000000000000));
}
__device__ __host__ __forceinline__ uint32_t fsec16_word_match(uint64_t w, uint64_t xm)
{
    const uint64_t f = w ^ xm;
    const uint32_t g = (uint32_t)f & (uint32_t)(f >> 32); // planes 0 & 2 | planes 1 & 3 << 16
    return g & (g >> 16) & 0xFFFFu;
}
// the words of a 16-ary section from its four 48-bit planes (the builder's ballots)
__device__ __host__ __forceinline__ uint64_t fsec16_pack(const uint64_t planes[4], unsigned j)
{
    return ((planes[0] >> (16 * j)) & 0xFFFF) | (((planes[1] >> (16 * j)) & 0xFFFF) << 16) | (((planes[2] >> (16 * j)) & 0xFFFF) << 32) |
           (((planes[3] >> (16 * j)) & 0xFFFF) << 48);
}

// the positions of the section that hold slot t (bit i = position i of the section's kFLane)
__device__ __host__ __forceinline__ uint64_t fsec_match_words(uint64_t p0, uint64_t p1, uint64_t p2, unsigned t)
{
    if constexpr (kFK == 3)
        return ((t & 1) ? p0 : ~p0) & ((t & 2) ? p1 : ~p1) & ((t & 4) ? p2 : ~p2);
    else
    {
        const uint64_t xm = fsec16_xor(t);
        const uint32_t lo = fsec16_word_match(p0, xm) | (fsec16_word_match(p1, xm) << 16);
        return (uint64_t)lo | ((uint64_t)(fsec16_word_match(p2, xm) & kFWord2Mask) << 32);
    }
}
__device__ __forceinline__ uint64_t fsec_match(const FSec & x, unsigned t)
{
    return fsec_match_words(x.p0, x.p1, x.p2, t);
}

// 16-ary: relative count k (0..3) of a section: 16 bits in the header word, kFSpare more in the top bits of field k of the third word
__device__ __host__ __forceinline__ unsigned fsec16_count_field(uint64_t h, uint64_t w2, unsigned k)
{
    unsigned c = (unsigned)(h >> (16 * k)) & 0xFFFFu;
    if constexpr (kFSpare != 0)
        c |= ((unsigned)(w2 >> (16 * k + 16 - kFSpare)) & ((1u << kFSpare) - 1u)) << 16;
    return c;
}

// this lane's part of the line's header count for slot t (8-ary: the low 32 bits of the absolute count; 16-ary: relative to the superblock)
__device__ __forceinline__ unsigned fsec_header(const FSec & x, int s, unsigned t)
{
    if constexpr (kFK == 3)
        return (s == (int)(t >> 1)) ? (unsigned)(x.h >> (32 * (t & 1))) : 0u;
    else
        return (s == (int)(t >> 2)) ? fsec16_count_field(x.h, x.p2, t & 3) : 0u;
}

// this lane's share of "slot t among the first `off` positions of the line, plus the line's count for t"
__device__ __forceinline__ unsigned fsec_count(const FSec & x, int s, unsigned off, unsigned t)
{
    const uint64_t m = fsec_match(x, t);
    const int tt = (int)off - (int)kFLane * s;
    const unsigned cnt = tt <= 0 ? 0u : (tt >= (int)kFLane ? popc64(m) : popc64(m << (64 - tt)));
    return cnt + fsec_header(x, s, t);
}

// 16-ary: the superblock's count for slot t (what the header of line `abs_line` is relative to); 0 on 8-ary lines.  Superblocks are
// counted in ABSOLUTE lines (superblock = abs_line >> kFSuperLog), nodes start anywhere: a line in the superblock its node starts
// in (`node_first` = the node's first line) counts from the node's start — base 0, nothing to read; every later superblock of the
// node has its record.  The address does not depend on the line's content: callers issue the read beside the line's.
__device__ __forceinline__ uint64_t fused_super(const uint32_t * f_super, bool wide, uint64_t node_first, uint64_t abs_line, unsigned t)
{
    if constexpr (kFK == 4)
    {
        const uint64_t sb = abs_line >> kFSuperLog;
        if (sb == (node_first >> kFSuperLog))
            return 0;
        const uint64_t idx = sb * kFSlots + t;
        if (wide) // kernel-uniform: 64-bit records
            return reinterpret_cast<const uint64_t *>(f_super)[idx];
        return f_super[idx];
    }
    else
        return 0;
}
__device__ __forceinline__ uint64_t fused_super(const WtView & wt, uint64_t node_first, uint64_t abs_line, unsigned t)
{
    return fused_super(wt.f_super, (wt.size >> 32) != 0, node_first, abs_line, t);
}

// the same summed over the quad — for every size of sequence: `u` the fused node, `abs_line` the line's index in the layout,
// `sup` = fused_super(wt, abs_line, t)
__device__ __forceinline__ uint64_t quad_fsec_count(const WtView & wt, const WtFusedTables * FT, const FSec & x, int s, unsigned off,
                                                    unsigned t, unsigned u, uint64_t abs_line, uint64_t sup)
{
    const unsigned lo = quad_sum(fsec_count(x, s, off, t)); // (modulo 2^32: header and in-line part may pass a multiple together)
    if constexpr (kFK == 4)
        return sup + lo;
    else
    {
        if (!(wt.size >> 32)) // kernel-uniform
            return lo;
        const uint64_t place = (abs_line << 8) + off;
        const unsigned key = (u << 3) | t, nc = FT->n_cross;
        unsigned hi = 0;
        for (unsigned e = 0; e < nc; ++e)
            hi += (FT->cross_key[e] == key && FT->cross_pos[e] <= place) ? 1u : 0u;
        return ((uint64_t)hi << 32) | lo;
    }
}

// the slot stored at position `off` of the line (all four lanes get it)
__device__ __forceinline__ unsigned quad_fsec_slot(const FSec & x, int s, unsigned off)
{
    unsigned t = 0;
    const int tt = (int)off - (int)kFLane * s;
    if (tt >= 0 && tt < (int)kFLane)
    {
        if constexpr (kFK == 3)
            t = (unsigned)((x.p0 >> tt) & 1) | ((unsigned)((x.p1 >> tt) & 1) << 1) | ((unsigned)((x.p2 >> tt) & 1) << 2);
        else
        {
            // (the word is picked by mask arithmetic: a select chain over the three words becomes an indexed access through scratch memory)
            const unsigned jw = (unsigned)tt >> 4;
            const uint64_t k1 = UINT64_C(0) - (uint64_t)(jw == 1), k2 = UINT64_C(0) - (uint64_t)(jw == 2);
            const uint64_t w = (x.p0 & ~(k1 | k2)) | (x.p1 & k1) | (x.p2 & k2);
            const uint64_t f = w >> (tt & 15);
            t = (unsigned)(f & 1) | ((unsigned)(f >> 15) & 2u) | ((unsigned)(f >> 30) & 4u) | ((unsigned)(f >> 45) & 8u);
        }
    }
    return quad_sum(t);
}

// follow slot t down the binary node table: kFK levels, or fewer when a leaf comes first (its slot pads with zeros).
// (A precomputed [node][slot] table would save two LDS reads per step but costs 8 KiB of LDS per workgroup, and the
// lost occupancy cost more than the reads: 28.9 -> 22.4 G wt.rank/s.)
__device__ __forceinline__ unsigned wt_descend(const WtTables * T, unsigned v, unsigned t)
{
#pragma unroll
    for (unsigned j = 0; j < kFK; ++j, t >>= 1)
    {
        const unsigned nv = T->child[v][t & 1];
        v = nv == kWtUndef ? v : nv;
    }
    return v;
}

// one fused step of two rank cascades for the same symbol: node v (depth 0, 3, ...), `left` path bits remaining in p
template <bool NT>
__device__ __forceinline__ void quad_wt8_rank2_step(const WtView & wt, const WtTables * T, const WtFusedTables * FT, int s,
                                                    unsigned & v, uint64_t & p, unsigned & left, uint64_t & a,
                                                    uint64_t & b)
{
    const unsigned k = left < kFK ? left : kFK;
    const unsigned t = (unsigned)p & ((1u << k) - 1u);
    const uint64_t base = FT->fline[v];
    const uint64_t la = fused_line(a), lb = fused_line(b);
    const uint64_t La = base + la, Lb = base + lb;
    FSec xb = load_fsec<NT>(wt.f_lines, Lb, s);
    FSec xa = xb;
    if (La != Lb) // quad-uniform
        xa = load_fsec<NT>(wt.f_lines, La, s);
    const uint64_t sb = fused_super(wt, base, Lb, t);
    uint64_t sa = sb;
    if constexpr (kFK == 4)
        if ((La >> kFSuperLog) != (Lb >> kFSuperLog))
            sa = fused_super(wt, base, La, t);
    const unsigned u = v;
    v = wt_descend(T, v, t); // LDS lookups overlap the line fetches
    a = quad_fsec_count(wt, FT, xa, s, fused_off(a, la), t, u, La, sa);
    b = quad_fsec_count(wt, FT, xb, s, fused_off(b, lb), t, u, Lb, sb);
    p >>= k;
    left -= k;
}

// wt_pc::rank(i, c) on the fused layout
template <bool NT>
__device__ __forceinline__ uint64_t quad_wt8_rank(const WtView & wt, const WtTables * T, const WtFusedTables * FT, int s,
                                                  uint64_t i, unsigned c)
{
    if (T->c_to_leaf[c] == kWtUndef)
        return 0;
    if (wt.sigma == 1)
        return i;
    uint64_t p = T->path[c];
    unsigned left = (unsigned)(p >> 56);
    uint64_t res = i;
    unsigned v = 0;
    while (left && res)
    {
        const unsigned k = left < kFK ? left : kFK;
        const unsigned t = (unsigned)p & ((1u << k) - 1u);
        const uint64_t nf = FT->fline[v], li = fused_line(res), L = nf + li;
        FSec x = load_fsec<NT>(wt.f_lines, L, s);
        const uint64_t sup = fused_super(wt, nf, L, t);
        const unsigned u = v;
        v = wt_descend(T, v, t); // LDS lookups overlap the line fetch
        res = quad_fsec_count(wt, FT, x, s, fused_off(res, li), t, u, L, sup);
        p >>= k;
        left -= k;
    }
    return res;
}

// ---- select on the fused layout ----------------------------------------------------------------
// position (inside node u's sequence) of the (k+1)-th occurrence of slot t — the inverse of the fused rank step, and
// three binary select levels in one.  The directory of (u, t) holds the POSITION of every 256th occurrence (and u's
// size as the last entry); the guess is interpolated between two (position, count) pairs, the probed line's header
// count for t and the popcount of its match mask say whether the occurrence is inside, and a miss replaces one end of
// the bracket by (line edge, exact count) — the scheme of §3.2 of DESIGN.md.
template <class P> // P: uint32_t, or uint64_t for sequences of 2^32 symbols and more (16-ary lines)
struct FselBracketT
{ // lo_cnt occurrences lie in front of position plo, hi_cnt in front of phi; lo_cnt <= k < hi_cnt
    P plo, phi, lo_cnt, hi_cnt;
};
typedef FselBracketT<uint32_t> FselBracket;

template <class P>
__device__ __forceinline__ FselBracketT<P> fsel_bracket_t(const P * dir, uint32_t off, P k, P total)
{
    const P j = k >> kFselLog;
    FselBracketT<P> b;
    b.plo = dir[off + j];
    b.phi = dir[off + j + 1];
    b.lo_cnt = j << kFselLog;
    b.hi_cnt = (j + 1) << kFselLog;
    if (b.hi_cnt > total)
        b.hi_cnt = total; // the last entry is the node's size: all `total` occurrences lie in front of it
    return b;
}
__device__ __forceinline__ FselBracket fsel_bracket(const uint32_t * dir, uint32_t off, uint32_t k, uint32_t total)
{
    return fsel_bracket_t<uint32_t>(dir, off, k, total);
}

// one probe; on a hit all four lanes get the position
template <bool NT, class P>
__device__ __forceinline__ bool quad_fsel_probe_t(const WtView & wt, uint64_t base_line, int s, unsigned t, P k, FselBracketT<P> & b,
                                                  int tries, uint64_t & pos_out)
{
    const P span = b.phi - b.plo; // > 0
    P pe;
    if (tries >= 3 && (tries & 1))
        pe = b.plo + (span >> 1);
    else
    {
        const float f = (float)(k - b.lo_cnt) * __builtin_amdgcn_rcpf((float)(b.hi_cnt - b.lo_cnt));
        const P o = (P)(f * (float)span);
        pe = b.plo + (o >= span ? span - 1 : o);
    }
    const uint32_t g = (uint32_t)fused_line(pe);
    const FSec x = load_fsec<NT>(wt.f_lines, base_line + g, s);
    const P sup = (P)fused_super(wt.f_super, sizeof(P) == 8, base_line, base_line + g, t);
    const uint64_t m = fsec_match(x, t);
    const unsigned c_lane = popc64(m);
    const unsigned hdr = fsec_header(x, s, t);
    const P c0 = sup + quad_sum(hdr);
    const uint32_t c_in = quad_sum(c_lane);
    if (k < c0)
    {
        b.phi = (P)g * kFusedPos;
        b.hi_cnt = c0;
        return false;
    }
    if (k >= c0 + c_in)
    {
        b.plo = (P)(g + 1) * kFusedPos;
        b.lo_cnt = c0 + c_in;
        return false;
    }
    const unsigned r = (unsigned)(k - c0), ex = quad_excl(c_lane, s);
    const bool mine = r >= ex && r < ex + c_lane;
    uint64_t pos = 0;
    if (mine)
        pos = (uint64_t)g * kFusedPos + kFLane * (unsigned)s + sel64(m, r - ex + 1);
    pos_out = quad_gather_u64(pos, mine);
    return true;
}
template <bool NT>
__device__ __forceinline__ bool quad_fsel_probe(const WtView & wt, uint64_t base_line, int s, unsigned t, uint32_t k, FselBracket & b,
                                                int tries, uint64_t & pos_out)
{
    return quad_fsel_probe_t<NT, uint32_t>(wt, base_line, s, t, k, b, tries, pos_out);
}

// both cascades of one LF step (backward_search)
template <bool NT>
__device__ __forceinline__ void quad_wt_rank2(const WtView & wt, const WtTables * T, const WtFusedTables * FT, int s,
                                              unsigned c, uint64_t & a, uint64_t & b)
{
    if (T->c_to_leaf[c] == kWtUndef)
    {
        a = b = 0;
        return;
    }
    if (wt.sigma == 1)
        return;
    uint64_t p = T->path[c];
    unsigned len = (unsigned)(p >> 56);
    unsigned v = 0;
    if (wt.f_lines)
        while (len && b)
            quad_wt8_rank2_step<NT>(wt, T, FT, s, v, p, len, a, b);
    else
        for (unsigned l = 0; l < len && b; ++l, p >>= 1) // a <= b always; b == 0 ends both chains
            quad_wt_rank2_level<NT>(wt, T, s, v, (unsigned)(p & 1), a, b);
    if (b == 0)
        a = 0;
}

// one fused step of wt_pc::inverse_select from node v (depth 0, 3, ...) at offset i: up to three levels
template <bool NT, class I>
__device__ __forceinline__ void quad_wt8_invsel_step(const WtView & wt, const WtTables * T, const WtFusedTables * FT,
                                                     int s, unsigned & v, I & i)
{
    const uint64_t nf = FT->fline[v], li = fused_line(i), L = nf + li;
    FSec x = load_fsec<NT>(wt.f_lines, L, s);
    const unsigned off = fused_off(i, li);
    if constexpr (kFK == 4)
    { // the slot is only known once the line has arrived: every lane fetches the superblock's counts of ITS four slots beside the
      // line (the quad reads the whole 64-byte record, one request to the cache) and the slot's owner adds its one
        typedef unsigned u32x4s __attribute__((ext_vector_type(4)));
        const bool wide = (wt.size >> 32) != 0; // kernel-uniform
        u32x4s sv = {0, 0, 0, 0};
        if (!wide) // (read whatever the line's superblock is — the record exists — and dropped below where the node starts in it: a
                   // load under a lane-dependent condition costs the walk its overlap with the line's)
            sv = *reinterpret_cast<const u32x4s *>(wt.f_super + (L >> kFSuperLog) * kFSlots + 4 * s);
        const unsigned t = quad_fsec_slot(x, s, off);
        if (!wide)
        {
            const unsigned k = t & 3u;
            const unsigned own = k == 0 ? sv.x : (k == 1 ? sv.y : (k == 2 ? sv.z : sv.w));
            const bool use = s == (int)(t >> 2) && (L >> kFSuperLog) != (nf >> kFSuperLog); // (the node's first superblock counts from 0)
            i = (I)quad_sum(fsec_count(x, s, off, t) + (use ? own : 0u));
        }
        else
            i = (I)quad_fsec_count(wt, FT, x, s, off, t, v, L, fused_super(wt, nf, L, t));
        v = wt_descend(T, v, t);
    }
    else
    {
        const unsigned t = quad_fsec_slot(x, s, off);
        i = (I)quad_fsec_count(wt, FT, x, s, off, t, v, L, 0);
        v = wt_descend(T, v, t);
    }
}

// the same from fused node r by the table of fused nodes (WtFusedWalk, in LDS): returns the table's entry for the slot found at i
template <bool NT, class I>
__device__ __forceinline__ unsigned quad_wtf_invsel_step(const WtView & wt, const WtFusedWalk * W, int s, unsigned r, I & i)
{
    unsigned v = 0; // (the node tables are not touched: 16-ary lines, or 8-ary lines of fewer than 2^32 symbols)
    const uint64_t nf = W->rline[r], li = fused_line(i), L = nf + li;
    FSec x = load_fsec<NT>(wt.f_lines, L, s);
    const unsigned off = fused_off(i, li);
    unsigned t;
    if constexpr (kFK == 4)
    {
        typedef unsigned u32x4s __attribute__((ext_vector_type(4)));
        const bool wide = (wt.size >> 32) != 0; // kernel-uniform
        u32x4s sv = {0, 0, 0, 0};
        if (!wide)
            sv = *reinterpret_cast<const u32x4s *>(wt.f_super + (L >> kFSuperLog) * kFSlots + 4 * s);
        t = quad_fsec_slot(x, s, off);
        if (!wide)
        {
            const unsigned k = t & 3u;
            const unsigned own = k == 0 ? sv.x : (k == 1 ? sv.y : (k == 2 ? sv.z : sv.w));
            const bool use = s == (int)(t >> 2) && (L >> kFSuperLog) != (nf >> kFSuperLog);
            i = (I)quad_sum(fsec_count(x, s, off, t) + (use ? own : 0u));
        }
        else
            i = (I)quad_fsec_count(wt, nullptr, x, s, off, t, v, L, fused_super(wt, nf, L, t));
    }
    else
    {
        t = quad_fsec_slot(x, s, off);
        i = (I)quad_sum(fsec_count(x, s, off, t));
    }
    return W->succ[r][t];
}

// one level of wt_pc::inverse_select from inner node v at offset i of its slice: the bit at the position and the
// rank up to it come from the same line
template <bool NT>
__device__ __forceinline__ void quad_wt_invsel_level(const WtView & wt, const WtTables * T, int s, unsigned & v,
                                                     uint64_t & i)
{
    uint64_t pos = T->bv_pos[v] + i;
    uint64_t L = pos / kDB;
    Pair w = load_pair<NT>(wt.bv.lines, L, s);
    unsigned off = (unsigned)(pos - L * kDB);
    // the lane owning data word (off>>6) extracts the bit, then it is summed over the quad
    int wi = (int)(off >> 6);
    unsigned mybit = 0;
    if (wi == 2 * s)
        mybit = (unsigned)((w.b >> (off & 63)) & 1);
    else if (s > 0 && wi == 2 * s - 1)
        mybit = (unsigned)((w.a >> (off & 63)) & 1);
    unsigned bit = quad_sum(mybit);
    uint64_t r = quad_rank1(w, s, pos, L) - T->bv_pos_rank[v];
    i = bit ? r : i - r;
    v = T->child[v][bit];
}

// wt_pc::inverse_select(i) (wt_pc.hpp:411-430): returns (rank of wt[i] in [0,i), wt[i]); i < size.
template <bool NT>
__device__ __forceinline__ uint64_t quad_wt_inverse_select(const WtView & wt, const WtTables * T,
                                                           const WtFusedTables * FT, int s, uint64_t i, unsigned & c_out)
{
    unsigned v = 0;
    if (wt.f_lines)
        while (T->child[v][0] != kWtUndef)
            quad_wt8_invsel_step<NT>(wt, T, FT, s, v, i);
    else
        while (T->child[v][0] != kWtUndef)
            quad_wt_invsel_level<NT>(wt, T, s, v, i);
    c_out = (unsigned)T->bv_pos_rank[v];
    return i;
}

} // namespace sdslhip
