// workload.cpp — the benchmark inputs of SURVEY.md §8(d), produced on the host by integer-only, seeded generators so
// that the build container (where the real sdsl-lite answers them once: tests/golden/make_golden_large.py) and the GPU
// box (bench.py, tests -m gpu) hold byte-identical inputs:
//   * query streams: successive std::mt19937_64(seed) outputs, optionally reduced modulo a range (util.hpp:438-448 draws
//     its benchmark positions the same way);
//   * the configs[2] vector: bit i set iff the i-th output of mt19937_64(seed) % 100 < percent, drawn sequentially; a
//     table of generator states at regular draw counts ("checkpoints", tests/golden/mt9_checkpoints.bin) lets every host
//     thread produce its own stretch of the very same sequence;
//   * the configs[3]/[4] text: an English-class stand-in for Pizza&Chili english.1GB, which is not available offline
//     (benchmark/indexing_count/test_case.config:6 only names its download URL): Zipf-distributed words over a
//     generated vocabulary, mixed case, digits, punctuation and a tail of rare Latin-1 / control bytes — sigma > 200,
//     H0 ≈ 4.6 bits, every 64 KiB block seeded on its own so the text is produced in parallel.
// No device code; nothing here is on the query path.
#include <atomic>
#include <thread>

#include "common.hpp"

namespace sdslhip {

namespace {

// MT19937-64 (Matsumoto & Nishimura), the generator behind std::mt19937_64; own restatement because the checkpoints need
// the raw state.  tests/test_workload.py checks it against std::mt19937_64 through sdsl_hip_util_set_random_bits.
struct Mt64
{
    static constexpr int NN = 312, MM = 156;
    uint64_t mt[NN];
    uint64_t mti;
    explicit Mt64(uint64_t seed)
    {
        mt[0] = seed;
        for (int i = 1; i < NN; ++i)
            mt[i] = UINT64_C(6364136223846793005) * (mt[i - 1] ^ (mt[i - 1] >> 62)) + (uint64_t)i;
        mti = NN;
    }
    Mt64(const uint64_t * state313)
    {
        memcpy(mt, state313, sizeof(mt));
        mti = state313[NN];
    }
    void save(uint64_t * state313) const
    {
        memcpy(state313, mt, sizeof(mt));
        state313[NN] = mti;
    }
    void twist()
    {
        constexpr uint64_t A = UINT64_C(0xB5026F5AA96619E9), UM = UINT64_C(0xFFFFFFFF80000000), LM = UINT64_C(0x7FFFFFFF);
        int i = 0;
        for (; i < NN - MM; ++i)
        {
            const uint64_t x = (mt[i] & UM) | (mt[i + 1] & LM);
            mt[i] = mt[i + MM] ^ (x >> 1) ^ ((x & 1) ? A : 0);
        }
        for (; i < NN - 1; ++i)
        {
            const uint64_t x = (mt[i] & UM) | (mt[i + 1] & LM);
            mt[i] = mt[i + (MM - NN)] ^ (x >> 1) ^ ((x & 1) ? A : 0);
        }
        const uint64_t x = (mt[NN - 1] & UM) | (mt[0] & LM);
        mt[NN - 1] = mt[MM - 1] ^ (x >> 1) ^ ((x & 1) ? A : 0);
        mti = 0;
    }
    inline uint64_t operator()()
    {
        if (mti >= NN)
            twist();
        uint64_t x = mt[mti++];
        x ^= (x >> 29) & UINT64_C(0x5555555555555555);
        x ^= (x << 17) & UINT64_C(0x71D67FFFEDA60000);
        x ^= (x << 37) & UINT64_C(0xFFF7EEE000000000);
        x ^= (x >> 43);
        return x;
    }
};

unsigned host_threads()
{
    unsigned t = std::thread::hardware_concurrency();
    if (const char * e = getenv("SDSL_HIP_HOST_THREADS"))
        t = (unsigned)atoi(e);
    return t < 1 ? 1 : (t > 256 ? 256 : t);
}

template <class F>
void parallel_blocks(uint64_t n_blocks, F f)
{
    const unsigned T = (unsigned)std::min<uint64_t>(host_threads(), n_blocks ? n_blocks : 1);
    std::atomic<uint64_t> next{0};
    std::vector<std::thread> th;
    for (unsigned t = 0; t < T; ++t)
        th.emplace_back(
            [&]
            {
                for (;;)
                {
                    const uint64_t b = next.fetch_add(1);
                    if (b >= n_blocks)
                        break;
                    f(b);
                }
            });
    for (auto & x : th)
        x.join();
}

// ---- the text model ------------------------------------------------------------------------------------------------
struct TextModel
{
    static constexpr uint32_t V = 1u << 16;
    std::vector<uint8_t> pool;     // the words' letters
    std::vector<uint32_t> off;     // V + 1
    std::vector<uint8_t> proper;   // always capitalised
    std::vector<uint64_t> cum;     // cumulative Zipf-Mandelbrot weights
    uint8_t rare[160];             // bytes outside the printable set, most frequent first
    uint64_t rare_cum[160];
    explicit TextModel(uint64_t seed)
    {
        Mt64 rng(seed ^ UINT64_C(0x656E676C697368)); // "english"
        static const char letters[] = "etaoinshrdlcumwfgypbvkjxqz";
        static const uint16_t lw[26] = {127, 91, 82, 75, 70, 67, 63, 61, 60, 43, 40, 28, 28, 24, 24, 22, 20, 20, 19, 15, 10, 8, 2, 2, 1, 1};
        uint32_t lcum[26], ltot = 0;
        for (int i = 0; i < 26; ++i)
            lcum[i] = (ltot += lw[i]);
        static const uint8_t len_pct[12] = {3, 14, 20, 17, 13, 10, 8, 6, 4, 3, 1, 1}; // lengths 1..12
        off.resize(V + 1);
        proper.resize(V);
        cum.resize(V);
        uint64_t tot = 0;
        for (uint32_t r = 0; r < V; ++r)
        {
            off[r] = (uint32_t)pool.size();
            uint32_t u = (uint32_t)(rng() % 100), len = 1, acc = 0;
            for (int k = 0; k < 12; ++k)
                if (u < (acc += len_pct[k]))
                {
                    len = (uint32_t)k + 1;
                    break;
                }
            if (r < 64 && len > 4)
                len = 2 + (uint32_t)(rng() % 3); // the most frequent words are short
            if (r >= 4096 && len < 4)
                len += 3; // the tail is long
            for (uint32_t k = 0; k < len; ++k)
            {
                const uint32_t x = (uint32_t)(rng() % ltot);
                int c = 0;
                while (lcum[c] <= x)
                    ++c;
                pool.push_back((uint8_t)letters[c]);
            }
            proper[r] = r >= 200 && rng() % 12 == 0;
            tot += (UINT64_C(1) << 40) / (r + 3);
            cum[r] = tot;
        }
        off[V] = (uint32_t)pool.size();
        // rare symbols: 0x80..0xFF, then control bytes (never 0: SDSL's byte indexes reserve it, construct.hpp:83-86)
        int k = 0;
        for (int c = 0xE0; c <= 0xFF; ++c) rare[k++] = (uint8_t)c;
        for (int c = 0xC0; c <= 0xDF; ++c) rare[k++] = (uint8_t)c;
        for (int c = 0xA0; c <= 0xBF; ++c) rare[k++] = (uint8_t)c;
        for (int c = 0x80; c <= 0x9F; ++c) rare[k++] = (uint8_t)c;
        for (int c = 1; c <= 31 && k < 160; ++c)
            if (c != '\n' && c != '\t' && c != '\r')
                rare[k++] = (uint8_t)c;
        while (k < 160) rare[k++] = 0x7F;
        uint64_t rt = 0;
        for (int i = 0; i < 160; ++i)
            rare_cum[i] = (rt += (UINT64_C(1) << 32) / (uint64_t)(i + 2));
    }
    uint32_t draw_word(Mt64 & rng) const
    {
        const uint64_t u = rng() % cum[V - 1];
        uint32_t lo = 0, hi = V - 1; // first r with cum[r] > u
        while (lo < hi)
        {
            const uint32_t m = (lo + hi) >> 1;
            if (cum[m] > u)
                hi = m;
            else
                lo = m + 1;
        }
        return lo;
    }
    uint8_t draw_rare(Mt64 & rng) const
    {
        const uint64_t u = rng() % rare_cum[159];
        int i = 0;
        while (rare_cum[i] <= u)
            ++i;
        return rare[i];
    }
    // one block of the text, generated from its own stream
    void fill(uint8_t * out, uint64_t len, uint64_t seed, uint64_t block) const
    {
        Mt64 rng(seed * UINT64_C(0x9E3779B97F4A7C15) + block + 1);
        uint64_t p = 0;
        bool sentence_start = true, open_paren = false, open_quote = false;
        auto put = [&](uint8_t c)
        {
            if (p < len)
                out[p++] = c;
        };
        while (p < len)
        {
            const uint32_t kind = (uint32_t)(rng() % 1000);
            if (kind < 22)
            { // a number: 1-4 digits, sometimes a year or a decimal
                const uint32_t nd = 1 + (uint32_t)(rng() % 4);
                uint64_t v = rng();
                for (uint32_t i = 0; i < nd; ++i, v /= 10)
                    put((uint8_t)('0' + (i == 0 && nd > 1 ? 1 + v % 9 : v % 10)));
                if (rng() % 8 == 0)
                {
                    put((uint8_t)(rng() % 2 ? '.' : ','));
                    put((uint8_t)('0' + rng() % 10));
                    put((uint8_t)('0' + rng() % 10));
                }
            }
            else
            {
                const uint32_t w = draw_word(rng);
                const uint32_t b = off[w], e = off[w + 1];
                const bool all_caps = kind < 27;
                const bool cap = sentence_start || proper[w] || kind < 60;
                for (uint32_t i = b; i < e; ++i)
                {
                    uint8_t c = pool[i];
                    if (all_caps || (cap && i == b))
                        c = (uint8_t)(c - 32);
                    put(c);
                }
            }
            sentence_start = false;
            const uint32_t sep = (uint32_t)(rng() % 1000);
            if (sep < 740)
                put(' ');
            else if (sep < 810)
            {
                put(',');
                put(' ');
            }
            else if (sep < 868)
            {
                put('.');
                if (rng() % 5 == 0)
                {
                    put('\n');
                    if (rng() % 3 == 0)
                        put('\n');
                }
                else
                    put(' ');
                sentence_start = true;
            }
            else if (sep < 876)
            {
                put(';');
                put(' ');
            }
            else if (sep < 884)
            {
                put(':');
                put(' ');
            }
            else if (sep < 891)
            {
                put('?');
                put(' ');
                sentence_start = true;
            }
            else if (sep < 896)
            {
                put('!');
                put(' ');
                sentence_start = true;
            }
            else if (sep < 906)
            {
                put(' ');
                put('-');
                put(' ');
            }
            else if (sep < 920)
            {
                put('\'');
                put((uint8_t)(rng() % 4 ? 's' : 't'));
                put(' ');
            }
            else if (sep < 930)
                put('-');
            else if (sep < 944)
            {
                if (open_quote)
                {
                    put('"');
                    put(' ');
                }
                else
                {
                    put(' ');
                    put('"');
                }
                open_quote = !open_quote;
            }
            else if (sep < 954)
            {
                if (open_paren)
                {
                    put(')');
                    put(' ');
                }
                else
                {
                    put(' ');
                    put('(');
                }
                open_paren = !open_paren;
            }
            else if (sep < 962)
            { // markup-ish and other printable leftovers
                static const char misc[] = "/&*[]#%$@+=_<>|~^`{}\\\t";
                put(' ');
                put((uint8_t)misc[rng() % (sizeof(misc) - 1)]);
                put(' ');
            }
            else if (sep < 990)
                put(' ');
            else
            { // a rare byte, sometimes a run of them (non-English passages)
                put(' ');
                const uint32_t run = 1 + (uint32_t)(rng() % 3);
                for (uint32_t i = 0; i < run; ++i)
                    put(draw_rare(rng));
                put(' ');
            }
        }
    }
};

} // namespace

} // namespace sdslhip

using namespace sdslhip;

extern "C" {

sdsl_hip_status sdsl_hip_util_rnd_positions(uint64_t seed, uint64_t count, uint64_t mod, uint64_t add, uint64_t * out)
{
    if (!out && count)
    {
        set_error("rnd_positions: null output");
        return SDSL_HIP_ERR_INVALID;
    }
    // the generator is sequential; the 64-bit modulo (≈ 25 cycles) is what costs, so it runs on the other threads,
    // a chunk behind the generator
    Mt64 rng(seed);
    constexpr uint64_t kChunk = UINT64_C(1) << 22;
    const uint64_t n_chunks = (count + kChunk - 1) / kChunk;
    // first touch of a fresh array is the expensive part on a virtualised host (≈ 20 µs per page fault measured):
    // every thread faults in its share of the pages before the sequential generator walks over them
    parallel_blocks((count * 8 + (1u << 21) - 1) >> 21,
                    [&](uint64_t b)
                    {
                        volatile uint8_t * p = (volatile uint8_t *)out;
                        const uint64_t end = std::min<uint64_t>(count * 8, (b + 1) << 21);
                        for (uint64_t o = b << 21; o < end; o += 4096)
                            p[o] = 0;
                    });
    std::atomic<uint64_t> produced{0}, next{0};
    const unsigned T = mod ? std::min<unsigned>(host_threads(), 8) : 0;
    std::vector<std::thread> th;
    for (unsigned t = 0; t < T; ++t)
        th.emplace_back(
            [&]
            {
                for (;;)
                {
                    const uint64_t c = next.fetch_add(1);
                    if (c >= n_chunks)
                        break;
                    while (produced.load(std::memory_order_acquire) <= c)
                        std::this_thread::yield();
                    const uint64_t lo = c * kChunk, hi = std::min(count, lo + kChunk);
                    for (uint64_t i = lo; i < hi; ++i)
                        out[i] = add + out[i] % mod;
                }
            });
    for (uint64_t c = 0; c < n_chunks; ++c)
    {
        const uint64_t lo = c * kChunk, hi = std::min(count, lo + kChunk);
        for (uint64_t i = lo; i < hi; ++i)
            out[i] = rng();
        produced.store(c + 1, std::memory_order_release);
    }
    for (auto & x : th)
        x.join();
    if (!mod && add)
        for (uint64_t i = 0; i < count; ++i)
            out[i] += add;
    return SDSL_HIP_OK;
}

sdsl_hip_status sdsl_hip_util_set_random_bits(uint64_t * words, uint64_t n_bits, uint64_t seed)
{
    if (!words && n_bits)
    {
        set_error("set_random_bits: null words");
        return SDSL_HIP_ERR_INVALID;
    }
    // util.hpp:467-485 — one mt19937_64 output per 64-bit word; like SDSL the last word is NOT masked (stray bits above
    // n_bits are legal in an int_vector and every consumer ignores them)
    return sdsl_hip_util_rnd_positions(seed, (n_bits + 63) >> 6, 0, 0, words);
}

sdsl_hip_status sdsl_hip_util_mt_checkpoints(uint64_t seed, uint64_t stride, uint64_t n, uint64_t * out)
{
    if (!out || stride == 0)
    {
        set_error("mt_checkpoints: invalid argument");
        return SDSL_HIP_ERR_INVALID;
    }
    Mt64 rng(seed);
    for (uint64_t j = 0; j < n; ++j)
    {
        rng.save(out + j * 313);
        if (j + 1 < n)
            for (uint64_t i = 0; i < stride; ++i)
                (void)rng();
    }
    return SDSL_HIP_OK;
}

sdsl_hip_status sdsl_hip_util_density_bits(uint64_t * words, uint64_t n_bits, uint64_t seed, uint32_t percent,
                                           const uint64_t * checkpoints, uint64_t n_checkpoints, uint64_t stride)
{
    if ((!words && n_bits) || percent > 100 || (n_checkpoints && (!checkpoints || stride == 0 || stride % 64)))
    {
        set_error("density_bits: invalid argument");
        return SDSL_HIP_ERR_INVALID;
    }
    const uint64_t nw = (n_bits + 63) >> 6;
    auto run = [&](Mt64 rng, uint64_t w_lo, uint64_t w_hi)
    {
        for (uint64_t w = w_lo; w < w_hi; ++w)
        {
            const unsigned nb = w + 1 < nw || (n_bits & 63) == 0 ? 64u : (unsigned)(n_bits & 63);
            uint64_t x = 0;
            for (unsigned b = 0; b < nb; ++b)
                x |= (uint64_t)(rng() % 100 < percent) << b;
            words[w] = x;
        }
    };
    if (n_checkpoints == 0)
    {
        run(Mt64(seed), 0, nw);
        return SDSL_HIP_OK;
    }
    // checkpoint j = state before draw j * stride = before word j * stride / 64; stretches past the last checkpoint
    // continue from it
    const uint64_t wps = stride / 64;
    const uint64_t n_seg = std::min<uint64_t>(n_checkpoints, (nw + wps - 1) / wps);
    parallel_blocks(n_seg ? n_seg : 1,
                    [&](uint64_t j)
                    {
                        const uint64_t lo = j * wps, hi = j + 1 == n_seg ? nw : std::min(nw, lo + wps);
                        if (lo < hi)
                            run(Mt64(checkpoints + j * 313), lo, hi);
                    });
    return SDSL_HIP_OK;
}

sdsl_hip_status sdsl_hip_util_english_text(uint8_t * out, uint64_t n_bytes, uint64_t seed)
{
    if (!out && n_bytes)
    {
        set_error("english_text: null output");
        return SDSL_HIP_ERR_INVALID;
    }
    const TextModel model(seed);
    constexpr uint64_t kBlock = UINT64_C(1) << 16;
    parallel_blocks((n_bytes + kBlock - 1) / kBlock,
                    [&](uint64_t b)
                    {
                        const uint64_t lo = b * kBlock;
                        model.fill(out + lo, std::min(kBlock, n_bytes - lo), seed, b);
                    });
    return SDSL_HIP_OK;
}


// The repetitive stand-in (round 6).  The real english.1GB is a concatenation of Gutenberg books: licence boilerplate, headers and
// re-editions repeat whole passages, so a 20-byte pattern drawn from the text keeps a WIDE suffix-array interval for many characters
// (genpatterns.c:183-203 draws patterns from the text itself) — while in the independent blocks above it occurs 1.017 times on average
// and the search is down to one suffix after a dozen characters.  Here `percent` of the 64 KiB blocks are COPIES of an earlier block,
// rotated by a block-specific shift (copies are not aligned in real collections either): block b is a copy iff mix(seed, b) % 100 <
// percent; its source is drawn among ALL earlier blocks, and a source that is itself a copy hands on to its own source, so popular
// passages accumulate copies.  Every block is still produced from (seed, block index) alone — in parallel, prefix-stable.
static inline uint64_t mix64(uint64_t x)
{
    x += UINT64_C(0x9E3779B97F4A7C15);
    x = (x ^ (x >> 30)) * UINT64_C(0xBF58476D1CE4E5B9);
    x = (x ^ (x >> 27)) * UINT64_C(0x94D049BB133111EB);
    return x ^ (x >> 31);
}
sdsl_hip_status sdsl_hip_util_english_text_repetitive(uint8_t * out, uint64_t n_bytes, uint64_t seed, uint32_t percent)
{
    if (!out && n_bytes)
    {
        set_error("english_text_repetitive: null output");
        return SDSL_HIP_ERR_INVALID;
    }
    if (percent > 95)
    {
        set_error("english_text_repetitive: at most 95 %% of the blocks can be copies (got %u)", percent);
        return SDSL_HIP_ERR_INVALID;
    }
    const TextModel model(seed);
    constexpr uint64_t kBlock = UINT64_C(1) << 16;
    parallel_blocks((n_bytes + kBlock - 1) / kBlock,
                    [&](uint64_t b)
                    {
                        const uint64_t lo = b * kBlock, len = std::min(kBlock, n_bytes - lo);
                        uint64_t src = b, shift = 0;
                        while (src > 0 && mix64(seed * 31 + src) % 100 < percent)
                        { // a copy: of a block in front of it (the shifts of a chain add up)
                            shift += mix64(seed * 131 + src) % kBlock;
                            src = mix64(seed * 17 + src) % src;
                        }
                        if (src == b)
                        {
                            model.fill(out + lo, len, seed, b);
                            return;
                        }
                        std::vector<uint8_t> tmp(kBlock);
                        model.fill(tmp.data(), kBlock, seed, src);
                        shift %= kBlock;
                        for (uint64_t j = 0; j < len; ++j)
                            out[lo + j] = tmp[(j + shift) & (kBlock - 1)];
                    });
    return SDSL_HIP_OK;
}

} // extern "C"
