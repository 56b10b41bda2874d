// bv_serialize.cpp — writers for the serialised form of SDSL's supports of a plain bit vector, so that structures built
// on the device can be handed to unmodified SDSL code as its DEFAULT types (wt_huff<>, csa_wt<>, sd_vector<>):
//   rank_support_v5<b>::serialize      rank_support_v5.hpp:68-124,160-167
//   rank_support_v<b>::serialize       rank_support_v.hpp:71-122,156-163
//   select_support_mcl<b>::serialize   select_support_mcl.hpp:207-381 (what init_slow / init_fast leave behind), :474-518
// These are host-side passes over the exported words: serialisation is I/O, not a query path.
#include "bv_serialize.hpp"

#include <algorithm>
#include <thread>

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
    // arguments (bits of value `bit` below n_bits) per chunk of 64 words, as a prefix sum: lets a worker start in the
    // middle of the vector
    constexpr uint64_t CH = 64;
    const uint64_t n_chunks = (W + CH - 1) / CH;
    std::vector<uint64_t> before(n_chunks + 1, 0);
    auto arg_word = [&](uint64_t w) -> uint64_t {
        uint64_t x = bit ? words[w] : ~words[w];
        if (w == W - 1 && (n_bits & 63))
            x &= (UINT64_C(1) << (n_bits & 63)) - 1;
        return x;
    };
    for (uint64_t c = 0; c < n_chunks; ++c)
    {
        uint64_t cnt = 0;
        const uint64_t e = std::min(W, (c + 1) * CH);
        for (uint64_t w = c * CH; w < e; ++w)
            cnt += (uint64_t)__builtin_popcountll(arg_word(w));
        before[c + 1] = before[c] + cnt;
    }
    const uint64_t A = before[n_chunks]; // select_support.hpp:132-135,171-174
    out.u64(A);
    if (A == 0)
        return;
    const unsigned logn = hi_bit(((n_bits + 63) >> 6) << 6) + 1; // initData, select_support_mcl.hpp:448-465
    const uint64_t logn4 = (uint64_t)logn * logn * logn * logn;
    const bool slow = n_bits < 100000; // dispatch :121-128
    const uint64_t sb = (A + SB - 1) / SB;

    std::vector<uint8_t> is_mini(sb, 1), has_entry(sb, 0);
    std::vector<uint64_t> entry(sb, 0);
    std::vector<Packed> blocks(sb);

    // superblocks [k0, k1): every worker walks its own stretch of the argument sequence
    auto work = [&](uint64_t k0, uint64_t k1) {
        // position the iterator on argument number SB * k0
        const uint64_t target = SB * k0;
        uint64_t c = (uint64_t)(std::upper_bound(before.begin(), before.end(), target) - before.begin()) - 1;
        uint64_t skip = target - before[c], w = c * CH, cur = 0;
        auto load = [&]() {
            cur = 0;
            while (w < W && !(cur = arg_word(w)))
                ++w;
        };
        load();
        auto next = [&](uint64_t & pos) -> bool {
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
        };
        uint64_t pending = 0;
        bool have_pending = true;
        for (uint64_t i = 0; i <= skip && have_pending; ++i)
            have_pending = next(pending); // `pending` = first argument of block k0
        std::vector<uint64_t> P(SB);
        for (uint64_t k = k0; k < k1; ++k)
        {
            uint64_t cn = 0;
            while (cn < SB && have_pending)
            {
                P[cn++] = pending;
                have_pending = next(pending);
            }
            const uint64_t first = P[0];
            bool is_long;
            uint64_t long_width, pos_diff;
            if (slow)
            { // init_slow :207-266: decided on the block's own last argument
                const uint64_t last = P[cn - 1];
                pos_diff = last - first;
                is_long = pos_diff > logn4;
                long_width = hi_bit(last) + 1;
            }
            else if (cn > SB - 64)
            { // init_fast :269-352: the block "completes" with its 4033rd argument; from there the scan runs 64 arguments
              // further, i.e. onto the first argument of the NEXT block if there is one
                const uint64_t pos_of_last = (cn == SB && have_pending) ? pending : P[cn - 1];
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
            if (slow || cn > SB - 64)
            {
                has_entry[k] = 1;
                entry[k] = first;
            }
            if (is_long)
            {
                is_mini[k] = 0;
                blocks[k].init(SB, (uint8_t)long_width);
                for (uint64_t j = 0; j < cn; ++j)
                    blocks[k].set(j, P[j]);
            }
            else
            {
                blocks[k].init(64, (uint8_t)(hi_bit(pos_diff) + 1));
                for (uint64_t j = 0; j < cn; j += 64)
                    blocks[k].set(j / 64, P[j] - first);
            }
        }
    };
    unsigned n_thr = std::thread::hardware_concurrency();
    n_thr = n_thr < 1 ? 1 : (n_thr > 32 ? 32 : n_thr);
    if (sb < 64)
        n_thr = 1;
    if (n_thr == 1)
        work(0, sb);
    else
    {
        std::vector<std::thread> th;
        for (unsigned t = 0; t < n_thr; ++t)
        {
            const uint64_t k0 = sb * t / n_thr, k1 = sb * (t + 1) / n_thr;
            if (k0 < k1)
                th.emplace_back(work, k0, k1);
        }
        for (auto & x : th)
            x.join();
    }
    Packed superblock;
    superblock.init(sb, (uint8_t)logn);
    bool any_long = false;
    for (uint64_t k = 0; k < sb; ++k)
    {
        if (has_entry[k])
            superblock.set(k, entry[k]);
        any_long = any_long || !is_mini[k];
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
