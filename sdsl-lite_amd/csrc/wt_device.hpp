// wt_device.hpp — device view of a byte wavelet tree (wt_huff<bit_vector, rank_support_v5<>>)
// and the quad-cooperative traversals shared by wt.hip (rank/access/inverse_select/select) and
// fm.hip (backward search / count).
//
// The node table of SDSL's _byte_tree (wt_helper.hpp:200-327: BFS node array with bv_pos,
// bv_pos_rank, parent, child[2]; m_c_to_leaf[256]; m_path[256]) is tiny (<= 511 nodes), so each
// workgroup stages it in LDS once; the concatenated bit vector (wt_pc.hpp:90) lives in HBM as
// rank lines (bv_device.hpp).  One rank cascade level = ONE 64-byte line fetch.
#pragma once
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

struct WtView
{
    BvView bv;   // backend 0: the bit vector as rank lines
    RrrView rrr; // backend 1: the bit vector as an rrr_vector<63> (wt_huff<rrr_vector<63>>)
    uint32_t backend;
    const WtTables * tables;
    uint64_t size;  // number of symbols
    uint64_t sigma; // effective alphabet size
    uint32_t n_nodes;
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

template <bool NT>
__device__ __forceinline__ void quad_wt_rank2(const WtView & wt, const WtTables * T, int s, unsigned c, uint64_t & a,
                                              uint64_t & b)
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
    for (unsigned l = 0; l < len && b; ++l, p >>= 1) // a <= b always; b == 0 ends both chains
        quad_wt_rank2_level<NT>(wt, T, s, v, (unsigned)(p & 1), a, b);
    if (b == 0)
        a = 0;
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
__device__ __forceinline__ uint64_t quad_wt_inverse_select(const WtView & wt, const WtTables * T, int s, uint64_t i,
                                                           unsigned & c_out)
{
    unsigned v = 0;
    while (T->child[v][0] != kWtUndef)
        quad_wt_invsel_level<NT>(wt, T, s, v, i);
    c_out = (unsigned)T->bv_pos_rank[v];
    return i;
}

} // namespace sdslhip
