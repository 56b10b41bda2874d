// bv_serialize.cpp — writers for the serialised form of SDSL's supports of a plain bit vector, so that structures built
// on the device can be handed to unmodified SDSL code as its DEFAULT types (wt_huff<>, csa_wt<>, sd_vector<>):
//   rank_support_v5<b>::serialize      rank_support_v5.hpp:68-124,160-167
//   rank_support_v<b>::serialize       rank_support_v.hpp:71-122,156-163
//   select_support_mcl<b>::serialize   select_support_mcl.hpp:207-381 (what init_slow / init_fast leave behind), :474-518
// These are host-side passes over the exported words: serialisation is I/O, not a query path.
#include "bv_serialize.hpp"

#include <algorithm>

namespace sdslhip {

static inline unsigned hi_bit(uint64_t x) // bits::hi, hi(0) = 0
{
    return x ? 63u - (unsigned)__builtin_clzll(x) : 0u;
}

// word w of the vector as the support sees it: for b = 0 the complement (all 64 bits: the directories of the rank
// supports count the padding of the last word too, rank_support.hpp:111-134)
static inline uint64_t arg_word(const uint64_t * words, uint64_t w, int bit)
{
    return bit ? words[w] : ~words[w];
}

void rank_v5_serialize_host(const uint64_t * words, uint64_t n_bits, int bit, StreamWriter & out)
{
    std::vector<uint64_t> dir;
    const uint64_t W = (n_bits + 63) >> 6;
    if (n_bits == 0)
        dir.assign(2, 0); // rank_support_v5.hpp:76-80
    else
    {
        const uint64_t nsb = ((n_bits + 63) >> 11) + 1;
        dir.assign(2 * nsb, 0);
        uint64_t abs = 0;
        for (uint64_t s = 0; s < nsb; ++s)
        { // per 2048-bit superblock: the absolute count, then the counts before blocks 1..5 (384 bits each) at shifts
          // 48, 36, 24, 12, 0
            dir[2 * s] = abs;
            uint64_t rel = 0, packed = 0;
            for (unsigned j = 0; j < 32 && 32 * s + j < W; ++j)
            {
                rel += (uint64_t)__builtin_popcountll(arg_word(words, 32 * s + j, bit));
                if ((j + 1) % 6 == 0 && j + 1 < 32)
                    packed |= rel << (60 - 12 * ((j + 1) / 6));
            }
            dir[2 * s + 1] = packed;
            abs += rel;
        }
    }
    out.int_vector(dir.data(), dir.size() * 64, 64);
}

void rank_v_serialize_host(const uint64_t * words, uint64_t n_bits, int bit, StreamWriter & out)
{
    std::vector<uint64_t> dir;
    const uint64_t W = (n_bits + 63) >> 6;
    if (n_bits == 0)
        dir.assign(2, 0); // rank_support_v.hpp:78-82
    else
    { // per 512-bit superblock: the absolute count, then the counts before words 1..7 at shifts 54, 45, ..., 0
        const uint64_t nsb = ((n_bits + 63) >> 9) + 1;
        dir.assign(2 * nsb, 0);
        uint64_t abs = 0;
        for (uint64_t s = 0; s < nsb; ++s)
        {
            dir[2 * s] = abs;
            uint64_t rel = 0, packed = 0;
            unsigned j = 0;
            for (; j < 8 && 8 * s + j < W; ++j)
            {
                if (j)
                    packed |= rel << (63 - 9 * j);
                rel += (uint64_t)__builtin_popcountll(arg_word(words, 8 * s + j, bit));
            }
            if (j > 0 && j < 8) // the unfinished last superblock also records the count behind its last word (:107-111)
                packed |= rel << (63 - 9 * j);
            dir[2 * s + 1] = packed;
            abs += rel;
        }
    }
    out.int_vector(dir.data(), dir.size() * 64, 64);
}

namespace {

// the arguments (positions of the b-valued bits below n_bits) in increasing order, one at a time
struct ArgIter
{
    const uint64_t * words;
    uint64_t n_bits, W, w = 0, cur = 0;
    int bit;
    ArgIter(const uint64_t * wd, uint64_t n, int b) : words(wd), n_bits(n), W((n + 63) >> 6), bit(b)
    {
        load();
    }
    void load()
    {
        cur = 0;
        while (w < W)
        {
            uint64_t x = bit ? words[w] : ~words[w];
            if (w == W - 1 && (n_bits & 63))
                x &= (UINT64_C(1) << (n_bits & 63)) - 1;
            if (x)
            {
                cur = x;
                return;
            }
            ++w;
        }
    }
    bool next(uint64_t & pos)
    {
        if (w >= W)
            return false;
        pos = 64 * w + (uint64_t)__builtin_ctzll(cur);
        cur &= cur - 1;
        if (!cur)
        {
            ++w;
            load();
        }
        return true;
    }
};

struct Packed
{
    std::vector<uint64_t> words;
    uint64_t n = 0;
    uint8_t width = 64;
    void init(uint64_t count, uint8_t w)
    {
        n = count;
        width = w;
        words.assign(((count * w + 63) >> 6) + 1, 0);
    }
    void set(uint64_t i, uint64_t v)
    {
        const uint64_t pos = i * width;
        const unsigned off = (unsigned)(pos & 63);
        if (width < 64)
            v &= (UINT64_C(1) << width) - 1;
        words[pos >> 6] |= v << off;
        if (off + width > 64)
            words[(pos >> 6) + 1] |= v >> (64 - off);
    }
    void write(StreamWriter & out) const
    {
        out.int_vector(words.data(), n * width, width);
    }
};

} // namespace

void select_mcl_serialize_host(const uint64_t * words, uint64_t n_bits, int bit, StreamWriter & out)
{
    constexpr uint64_t SB = 4096;
    const uint64_t W = (n_bits + 63) >> 6;
    uint64_t ones = 0;
    for (uint64_t w = 0; w < W; ++w)
    {
        uint64_t x = words[w];
        if (w == W - 1 && (n_bits & 63))
            x &= (UINT64_C(1) << (n_bits & 63)) - 1;
        ones += (uint64_t)__builtin_popcountll(x);
    }
    const uint64_t A = bit ? ones : n_bits - ones; // select_support.hpp:132-135,171-174
    out.u64(A);
    if (A == 0)
        return;
    const unsigned logn = hi_bit(((n_bits + 63) >> 6) << 6) + 1; // initData, select_support_mcl.hpp:448-465
    const uint64_t logn4 = (uint64_t)logn * logn * logn * logn;
    const bool slow = n_bits < 100000; // dispatch :121-128
    const uint64_t sb = (A + SB - 1) / SB;

    Packed superblock;
    superblock.init(sb, (uint8_t)logn);
    std::vector<uint8_t> is_mini(sb, 1);
    std::vector<Packed> blocks(sb);
    bool any_long = false;

    ArgIter it(words, n_bits, bit);
    std::vector<uint64_t> P(SB);
    uint64_t pending = 0;
    bool have_pending = it.next(pending); // first argument of the block to come
    for (uint64_t k = 0; k < sb; ++k)
    {
        uint64_t c = 0;
        while (c < SB && have_pending)
        {
            P[c++] = pending;
            have_pending = it.next(pending);
        }
        const uint64_t first = P[0];
        bool is_long;
        uint64_t long_width, pos_diff;
        if (slow)
        { // init_slow :207-266: decided on the block's own last argument
            const uint64_t last = P[c - 1];
            pos_diff = last - first;
            is_long = pos_diff > logn4;
            long_width = hi_bit(last) + 1;
        }
        else if (c > SB - 64)
        { // init_fast :269-352: the block "completes" with its 4033rd argument; from there the scan runs 64 arguments
          // further, i.e. onto the first argument of the NEXT block if there is one
            const uint64_t pos_of_last = (c == SB && have_pending) ? pending : P[c - 1];
            pos_diff = pos_of_last - first;
            is_long = pos_diff > logn4;
            long_width = hi_bit(pos_of_last) + 1;
        }
        else
        { // init_fast :354-366: an unfinished last block is always long, as wide as the vector's last position — and
          // its m_superblock entry is never written (stays 0)
            pos_diff = 0;
            is_long = true;
            long_width = hi_bit(n_bits - 1) + 1;
        }
        if (slow || c > SB - 64)
            superblock.set(k, first);
        if (is_long)
        {
            any_long = true;
            is_mini[k] = 0;
            blocks[k].init(SB, (uint8_t)long_width);
            for (uint64_t j = 0; j < c; ++j)
                blocks[k].set(j, P[j]);
        }
        else
        {
            blocks[k].init(64, (uint8_t)(hi_bit(pos_diff) + 1));
            for (uint64_t j = 0; j < c; j += 64)
                blocks[k].set(j / 64, P[j] - first);
        }
    }
    superblock.write(out);
    Packed mol; // mini_or_long: empty unless some block is long (:487-494)
    mol.init(any_long ? sb : 0, 1);
    if (any_long)
        for (uint64_t k = 0; k < sb; ++k)
            if (is_mini[k])
                mol.set(k, 1);
    mol.write(out);
    for (uint64_t k = 0; k < sb; ++k)
        blocks[k].write(out);
}

} // namespace sdslhip
