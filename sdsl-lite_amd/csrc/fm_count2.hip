// fm_count2.hip — count() of a large batch of fixed-length patterns on the fused wavelet tree, in three kernels:
//
//   k_fm_start   one pattern per quad, in the caller's order (patterns stream in, nothing is sorted): the SA interval of the
//                pattern's LAST k bytes comes from ONE 128-byte bucket of the k-mer hash table (FmDeep, below) instead of k LF
//                steps.  A pattern whose k-mer is absent cannot occur (count 0); one whose interval is a single suffix goes
//                straight to the text comparison (k_fm_verify2); one with no characters left is done; the rest are appended to
//                a work list as 32-byte records [q, l, e, characters left, the next 16 pattern bytes].
//   k_fm_count_flat  persistent quads over the work list.  The state of a search lives in registers; one loop iteration is ONE
//                fused tree step (three binary levels, one or two line fetches) of whatever pattern the quad holds, and a quad
//                that finishes takes the next record at once — its record was requested an iteration earlier, so the hand-over
//                costs no memory latency.  (k_fm_count, fm.hip, keeps its 16 quads in lock step per pattern: a wave lasts as
//                long as its longest search.)  Everything is 32-bit: the fused layout exists for fewer than 2^32 symbols.
//   k_fm_verify2 the searches that stopped at ONE suffix with characters left: SA[l] says where that suffix starts, the remaining
//                characters are compared with the text in front of it, eight bytes at a time.
//
// Reference semantics (unchanged, every answer is the number the reference computes):
//   backward_search(csa,l,r,begin,end,..)  suffix_array_algorithm.hpp:228-248  (characters from the pattern's end)
//   backward_search(csa,l,r,c,..)          suffix_array_algorithm.hpp:167-201  (l = C[c] + rank(l,c), r = C[c] + rank(r+1,c) - 1)
//   count(csa,begin,end)                   suffix_array_algorithm.hpp:464-471  (r + 1 - l, 0 for an empty interval)
//   csa_wt::rank_bwt -> wt_pc::rank        csa_wt.hpp:286-289, wt_pc.hpp:371-399
// The k-mer table holds, for every k-mer that occurs in the text, exactly the [l, r] the k LF steps produce (the SA range of
// the suffixes that start with it); a k-mer that does not occur has an empty range, i.e. count 0.
#include <chrono>
#include <type_traits>

#include "bv_host.hpp"
#include "fm_host.hpp"

namespace sdslhip {

// ---- tables of the flat count kernel ----------------------------------------------------------------------------------
__device__ __forceinline__ void fm_stage_ctab(FmCountTab * lds, const FmCountTab * g)
{
    static_assert(sizeof(FmCountTab) % 8 == 0, "FmCountTab is copied in 8-byte words");
    const uint64_t * src = reinterpret_cast<const uint64_t *>(g);
    uint64_t * dst = reinterpret_cast<uint64_t *>(lds);
    for (unsigned i = threadIdx.x; i < sizeof(FmCountTab) / 8; i += blockDim.x)
        dst[i] = src[i];
    __syncthreads();
}

// (u32x4, load_tail16, fm_tail_has_zero: fm_device.hpp)

struct FmRec
{
    uint32_t q, l, e, rem; // pattern number inside the slab, interval [l, e), characters still to process
    u32x4 w;               // pattern bytes [rem - 16, rem) of the pattern
};
static_assert(sizeof(FmRec) == 32, "FmRec is two 16-byte loads");
// WIDE (2^32 .. 2^39 symbols): l and e are 40-bit numbers, their top bytes ride in the record's fourth word:
// [rem : 16 | l >> 32 : 8 | e >> 32 : 8]; patterns are shorter than 65535 bytes there
template <bool WIDE>
struct FmPos
{
    typedef uint32_t type;
};
template <>
struct FmPos<true>
{
    typedef uint64_t type;
};

constexpr uint64_t kFmPending = UINT64_C(1) << 63;
constexpr uint32_t kFmDead = 0xFFFFFFFFu; // FmRec::rem of a pattern that k_fm_start has already answered

template <bool WIDE>
__device__ __forceinline__ u32x4 rec_head(uint32_t q, typename FmPos<WIDE>::type l, typename FmPos<WIDE>::type e, uint32_t rem, bool dead)
{
    u32x4 h;
    h.x = q;
    h.y = (uint32_t)l;
    h.z = (uint32_t)e;
    if (dead)
        h.w = kFmDead;
    else if (WIDE)
        h.w = (rem & 0xFFFFu) | ((uint32_t)((uint64_t)l >> 32) & 0xFFu) << 16 | ((uint32_t)((uint64_t)e >> 32) & 0xFFu) << 24;
    else
        h.w = rem;
    return h;
}
// What a search that has stopped at a FEW suffixes leaves for k_fm_verify2: [1 | suffixes - 1 : 3 | characters left | the first suffix's SA
// index].  Round 3 stopped at ONE suffix (a random 20-byte pattern of the first stand-in text occurs 1.017 times); in a collection with
// duplicated passages (tests: english_text_repetitive, 2.4 occurrences on average) half of the patterns never get down to one suffix and
// walked every remaining character.  The remaining characters stand in front of each of the interval's suffixes or not: comparing them at
// s <= kFmVerifyMax suffixes costs 1 + s fetches (the SA entries share a line) against 1.3 fused lines per remaining character.
// (kFmVerifyMax: fm_device.hpp)
template <bool WIDE>
__device__ __forceinline__ uint64_t pending_word(uint32_t rem, typename FmPos<WIDE>::type l, uint32_t s = 1)
{
    return WIDE ? kFmPending | ((uint64_t)(s - 1) << 60) | ((uint64_t)(rem & 0xFFFFFu) << 40) | (uint64_t)l
                : kFmPending | ((uint64_t)(s - 1) << 60) | ((uint64_t)(rem & 0x0FFFFFFFu) << 32) | (uint64_t)l;
}
// is the text comparison the cheaper end of a search with `rem` characters to go on the interval [l, e)?
// (`m` = the pattern's length: at least ONE character must have gone through the index — a pattern may END with the sentinel byte 0, which
// the index holds and the text buffer does not; a 0 anywhere else never matches, in the index or in the text)
// (`stable` = the interval has just survived a character without shrinking.  An interval of s > 1 suffixes that is still shrinking — a
// pattern on its way to its one occurrence — is a character away from s = 1, and 1.3 + 2 fetches beat 1 + s; one that holds its size is
// a passage the text repeats s times and will hold it to the end.  Measured: comparing at every s <= 8 at once cost the text without
// duplicates 3 % more fetches per pattern, 12.83 against 12.45.)
template <bool WIDE>
__device__ __forceinline__ bool verify_pays(uint64_t l, uint64_t e, uint32_t rem, uint32_t m, bool stable)
{
    const uint64_t s = e - l; // (one character left: its LF step is cheaper than SA[l] + the text)
    return s >= 1 && s <= kFmVerifyMax && (s == 1 || stable) && rem >= 2 && rem >= s && rem < m && rem < (WIDE ? (1u << 20) : (1u << 28));
}

// quad lookup of `key` in the k-mer table: true (and [l, e)) if present
template <bool WIDE>
__device__ __forceinline__ bool quad_deep_find(const FmDeep & D, int s, uint64_t key, typename FmPos<WIDE>::type & l,
                                               typename FmPos<WIDE>::type & e)
{
    constexpr uint64_t kmask = WIDE ? (UINT64_C(1) << 48) - 1 : ~UINT64_C(0);
    uint32_t b = deep_bucket(key, D.n_buckets);
    for (uint32_t tries = 0; tries < D.n_buckets; ++tries)
    {
        const ulonglong2 * bp = D.tab + (uint64_t)b * 8 + 2 * s;
        const ulonglong2 e0 = bp[0], e1 = bp[1];
        const bool h0 = (e0.x & kmask) == key, h1 = (e1.x & kmask) == key;
        const unsigned lh = h0 ? (unsigned)e0.y : (h1 ? (unsigned)e1.y : 0u);
        const unsigned eh = h0 ? (unsigned)(e0.y >> 32) : (h1 ? (unsigned)(e1.y >> 32) : 0u);
        const unsigned top = WIDE ? (h0 ? (unsigned)(e0.x >> 48) : (h1 ? (unsigned)(e1.x >> 48) : 0u)) : 0u;
        const unsigned hit = quad_sum((h0 || h1) ? 1u : 0u);
        if (hit)
        { // keys are unique: exactly one lane holds it
            l = quad_sum(lh);
            e = quad_sum(eh);
            if (WIDE)
            {
                const uint64_t t = quad_sum(top);
                l = (typename FmPos<WIDE>::type)((uint64_t)l | (t & 0xFFu) << 32);
                e = (typename FmPos<WIDE>::type)((uint64_t)e | (t >> 8) << 32);
            }
            return true;
        }
        const unsigned full = quad_sum((e0.x != 0 ? 1u : 0u) + (e1.x != 0 ? 1u : 0u));
        if (full < 8)
            return false; // an insertion would have used the free slot
        b = b + 1 == D.n_buckets ? 0 : b + 1;
    }
    return false;
}

template <bool VERIFY, bool WIDE>
__global__ __launch_bounds__(256) void k_fm_start(FmDeep D, uint64_t csa_size, const uint8_t * __restrict__ pats, uint32_t m,
                                                  uint32_t n_pat, uint64_t * __restrict__ out, FmRec * __restrict__ recs)
{
    typedef typename FmPos<WIDE>::type pos_t;
    const int s = threadIdx.x & 3;
    const uint32_t quads = (gridDim.x * blockDim.x) >> 2;
    for (uint32_t q = (blockIdx.x * blockDim.x + threadIdx.x) >> 2; q < n_pat; q += quads)
    {
        pos_t l = 0, e = (pos_t)csa_size;
        uint32_t rem = m;
        const uint64_t end = (uint64_t)(q + 1) * m;
        const uint64_t key = load_tail8(pats, end) >> (8 * (8 - D.k));
        uint64_t res = 0;
        bool done = false;
        if (has_zero_byte(key, D.k))
            ; // a 0 byte is the sentinel's character: left to the search, from the whole interval
        else if (!quad_deep_find<WIDE>(D, s, key, l, e))
            done = true; // the pattern's last k bytes do not occur in the text: count 0
        else
        {
            rem = m - D.k;
            if (rem == 0)
            {
                done = true;
                res = e - l;
            }
            else if (VERIFY && verify_pays<WIDE>(l, e, rem, m, false) && rem <= 16 && !fm_tail_has_zero(load_tail16(pats, (uint64_t)q * m + rem), rem))
            {
                done = true;
                res = pending_word<WIDE>(rem, l, (uint32_t)(e - l));
            }
        }
        const u32x4 h = rec_head<WIDE>(q, l, e, rem, done);
        u32x4 * dst = reinterpret_cast<u32x4 *>(recs + q);
        if (done)
        {
            if (s == 0)
            {
                out[q] = res;
                dst[0] = h;
            }
        }
        else
        {
            const u32x4 w = load_tail16(pats, (uint64_t)q * m + rem);
            if (s == 0)
            {
                dst[0] = h;
                dst[1] = w;
            }
        }
    }
}

// without a k-mer hash table: the dense table of fm.hip (FmJump: every k-mer over the compact alphabet, sigma^k entries) takes
// the pattern's last J.k characters — unsorted, but only the entries of k-mers that occur are ever touched (a few MiB that
// stay in the L2) — or the search starts from the whole interval.  One pattern per lane.
template <bool VERIFY, bool WIDE>
__global__ __launch_bounds__(256) void k_fm_start_dense(FmJump J, const FmTables * __restrict__ ftab, uint64_t csa_size,
                                                        const uint8_t * __restrict__ pats, uint32_t m, uint32_t n_pat,
                                                        uint64_t * __restrict__ out, FmRec * __restrict__ recs)
{
    typedef typename FmPos<WIDE>::type pos_t;
    __shared__ uint8_t c2c[256];
    for (unsigned c = threadIdx.x; c < 256; c += blockDim.x)
        c2c[c] = ftab->char2comp[c];
    __syncthreads();
    for (uint32_t q = blockIdx.x * blockDim.x + threadIdx.x; q < n_pat; q += gridDim.x * blockDim.x)
    {
        pos_t l = 0, e = (pos_t)csa_size;
        uint32_t rem = m;
        uint64_t res = 0;
        bool done = false;
        const uint64_t end = (uint64_t)(q + 1) * m;
        if (J.tab && m >= J.k)
        {
            const uint64_t tail = load_tail8(pats, end); // (J.k <= 8)
            uint64_t key = 0;
            bool ok = true;
            for (uint32_t t = 0; t < J.k; ++t)
            {
                const unsigned c = (unsigned)(tail >> (56 - 8 * t)) & 0xFFu;
                const unsigned cc = c2c[c];
                ok = ok && !(cc == 0 && c > 0); // a character that does not occur: left to the search (it ends there)
                key = key * J.sigma + cc;
            }
            if (ok)
            {
                const ulonglong2 en = *reinterpret_cast<const ulonglong2 *>(J.tab + 2 * key);
                rem = m - J.k;
                if (en.x > en.y)
                    done = true; // empty interval: count 0
                else
                {
                    l = (pos_t)en.x;
                    e = (pos_t)(en.y + 1);
                    if (rem == 0)
                    {
                        done = true;
                        res = e - l;
                    }
                    else if (VERIFY && verify_pays<WIDE>(l, e, rem, m, false) && rem <= 16 && !fm_tail_has_zero(load_tail16(pats, (uint64_t)q * m + rem), rem))
                    {
                        done = true;
                        res = pending_word<WIDE>(rem, l, (uint32_t)(e - l));
                    }
                }
            }
        }
        u32x4 * dst = reinterpret_cast<u32x4 *>(recs + q);
        dst[0] = rec_head<WIDE>(q, l, e, rem, done);
        if (done)
            out[q] = res;
        else
            dst[1] = load_tail16(pats, (uint64_t)q * m + rem);
    }
}

constexpr uint32_t kFlatChunk = 256; // records a wave claims at a time

template <bool WIDE>
struct FmCtabOf
{
    typedef FmCountTab type;
};
template <>
struct FmCtabOf<true>
{
    typedef FmCountTabW type;
};

template <bool VERIFY, bool WIDE>
__global__ __launch_bounds__(256) void k_fm_count_flat(const uint64_t * __restrict__ f_lines, const uint32_t * __restrict__ f_super,
                                                       const typename FmCtabOf<WIDE>::type * __restrict__ tab_g,
                                                       const FmRec * __restrict__ recs, uint32_t n_rec,
                                                       uint32_t * __restrict__ ticket, const uint8_t * __restrict__ pats,
                                                       uint32_t m, uint64_t * __restrict__ out)
{
    typedef typename FmPos<WIDE>::type pos_t;
    __shared__ typename FmCtabOf<WIDE>::type T;
    {
        static_assert(sizeof(T) % 8 == 0, "the tables are copied in 8-byte words");
        const uint64_t * src = reinterpret_cast<const uint64_t *>(tab_g);
        uint64_t * dst = reinterpret_cast<uint64_t *>(&T);
        for (unsigned i = threadIdx.x; i < sizeof(T) / 8; i += blockDim.x)
            dst[i] = src[i];
        __syncthreads();
    }
    const int lane = threadIdx.x & 63, s = lane & 3;
    const uint64_t quads_below = (UINT64_C(0x1111111111111111) & ((UINT64_C(1) << (lane & ~3)) - 1)); // leaders of the quads in front
    // the wave's current chunk of the work list, and the ticket of the next one (requested a chunk ahead)
    uint32_t cur = 0, cur_end = 0;
    bool drained = false; // no chunk left for this wave
    uint32_t next_ticket = 0;
    if (lane == 0)
        next_ticket = atomicAdd(ticket, 1u);
    // the search this quad holds ...
    bool act = false;
    uint32_t q = 0, rem = 0, wcnt = 0;
    pos_t l = 0, e = 0, a = 0, b = 0, cb = 0;
    u32x4 w = {0, 0, 0, 0};
    uint32_t left = 0, si = 0;
    uint64_t prev_s = 0; // the interval's size one character ago (0: a record fresh from the start kernel)
    // ... and the record it takes next
    bool nx = false;
    u32x4 nh = {0, 0, 0, 0}, nw = {0, 0, 0, 0};
    // WIDE: a count inside a node = the header's low 32 bits + the matches in the line (modulo 2^32), its high part = how many of the
    // (node, slot)'s listed places lie at or in front of the position (wt_device.hpp: quad_fsec_count)
    auto count_at = [&](const FSec & x, uint32_t off, uint32_t line, uint32_t t, uint32_t step, uint64_t sup) -> pos_t {
        const uint32_t lo = quad_sum(fsec_count(x, s, off, t));
        if constexpr (kFK == 4)
            return (pos_t)(sup + lo); // 16-ary lines: the superblock's count + the line's relative one
        else if constexpr (WIDE)
        {
            const uint64_t place = ((uint64_t)line << 8) + off;
            const unsigned key = ((unsigned)T.snode[step] << 3) | t, nc = T.n_cross;
            unsigned hi = 0;
            for (unsigned c = 0; c < nc; ++c)
                hi += (T.cross_key[c] == key && T.cross_pos[c] <= place) ? 1u : 0u;
            return (pos_t)(((uint64_t)hi << 32) | lo);
        }
        else
            return (pos_t)lo;
    };
    for (;;)
    {
        // 1. quads without a next record claim one (the wave's quads in order: the records of a wave stay neighbours)
        const uint64_t want = __ballot(!nx && s == 0);
        if (want && !drained)
        {
            if (cur == cur_end)
            {
                const uint32_t t = (uint32_t)__builtin_amdgcn_readfirstlane((int)next_ticket);
                const uint64_t lo = (uint64_t)t * kFlatChunk;
                if (lo >= n_rec)
                    drained = true;
                else
                {
                    cur = (uint32_t)lo;
                    cur_end = (uint32_t)(lo + kFlatChunk < n_rec ? lo + kFlatChunk : n_rec);
                    if (lane == 0)
                        next_ticket = atomicAdd(ticket, 1u);
                }
            }
            if (!drained)
            {
                const uint32_t avail = cur_end - cur, rank = (uint32_t)__popcll(want & quads_below);
                if (!nx && rank < avail)
                {
                    const u32x4 * src = reinterpret_cast<const u32x4 *>(recs + cur + rank);
                    nh = src[0];
                    nw = src[1];
                    nx = true;
                }
                const uint32_t wanted = (uint32_t)__popcll(want);
                cur += wanted < avail ? wanted : avail;
            }
        }
        // 2. one fused step (three tree levels) of both ends of the interval
        if (act && left)
        {
            const uint32_t step = si;
            const uint32_t st = T.steps[si];
            ++si;
            --left;
            const uint32_t base = st & 0x0FFFFFFFu, t = st >> 28;
            const uint32_t la = (uint32_t)fused_line(a), lb = (uint32_t)fused_line(b);
            const uint32_t La = base + la, Lb = base + lb;
            const FSec xb = load_fsec<false>(f_lines, Lb, s);
            FSec xa = xb;
            if (La != Lb) // quad-uniform
                xa = load_fsec<false>(f_lines, La, s);
            const uint64_t sb = fused_super(f_super, WIDE, base, Lb, t); // (16-ary lines; a cache-resident word beside the line)
            uint64_t sa = sb;
            if constexpr (kFK == 4)
                if ((La >> kFSuperLog) != (Lb >> kFSuperLog))
                    sa = fused_super(f_super, WIDE, base, La, t);
            a = count_at(xa, fused_off(a, la), La, t, step, sa);
            b = count_at(xb, fused_off(b, lb), Lb, t, step, sb);
            if (b == 0)
            { // a <= b: both chains stay 0 (wt_pc.hpp:386)
                a = 0;
                left = 0;
            }
            if (left == 0)
            {
                l = cb + a;
                e = cb + b;
            }
        }
        // 3. between two characters: is the search over?  then hand over to the next record; set up the next character
        if (act && left == 0)
        {
            uint64_t res = 0;
            bool fin = true;
            if (l >= e)
                res = 0;
            else if (rem == 0)
                res = e - l;
            else if (VERIFY && verify_pays<WIDE>(l, e, rem, m, (uint64_t)(e - l) == prev_s) && rem == wcnt && !fm_tail_has_zero(w, wcnt)) // (w: the next wcnt characters)
                res = pending_word<WIDE>(rem, l, (uint32_t)(e - l));
            else
            {
                fin = false;
                prev_s = (uint64_t)(e - l);
            }
            if (fin)
            {
                if (s == 0)
                    out[q] = res;
                act = false;
            }
        }
        if (!act && nx && nh.w == kFmDead)
            nx = false; // answered by k_fm_start already
        if (!act && nx)
        {
            q = nh.x;
            l = nh.y;
            e = nh.z;
            rem = nh.w;
            if (WIDE)
            {
                l = (pos_t)((uint64_t)l | (uint64_t)((nh.w >> 16) & 0xFFu) << 32);
                e = (pos_t)((uint64_t)e | (uint64_t)(nh.w >> 24) << 32);
                rem = nh.w & 0xFFFFu;
            }
            w = nw;
            wcnt = rem < 16 ? rem : 16;
            nx = false;
            act = true;
            left = 0;
            prev_s = 0;
        }
        if (act && left == 0)
        { // (records on the list have l < e and rem >= 1)
            if (wcnt == 0)
            { // a pattern with more than 16 characters to go: the next 16
                w = load_tail16(pats, (uint64_t)q * m + rem);
                wcnt = rem < 16 ? rem : 16;
            }
            const uint32_t c = w.w >> 24;
            w.w = __builtin_amdgcn_alignbit(w.w, w.z, 24);
            w.z = __builtin_amdgcn_alignbit(w.z, w.y, 24);
            w.y = __builtin_amdgcn_alignbit(w.y, w.x, 24);
            w.x <<= 8;
            --wcnt;
            --rem;
            const uint32_t meta = T.meta[c];
            if (meta == 0)
            { // the character does not occur (suffix_array_algorithm.hpp:180-184): ends at the top of the next iteration
                l = 1;
                e = 1;
            }
            else
            {
                cb = T.cb[c];
                if constexpr (WIDE)
                    cb = (pos_t)((uint64_t)cb | (uint64_t)T.cbh[c] << 32);
                a = l;
                b = e;
                si = meta & 0xFFFFu;
                left = meta >> 16;
            }
        }
        if (drained && !__any(act || nx))
            break; // (not drained and nobody busy: the claim at the top of the next iteration fetches a chunk or finds the end)
    }
}

// count() of the patterns whose search stopped at a few suffixes: suffix i of the interval stands at SA[i] in the text, so the pattern's
// remaining characters pats[begin .. begin + rem) occur right in front of it or not — the count is the number of suffixes where they
// do.  One lane per pattern (the SA entries of an interval are neighbours: one line).
template <class SA>
__global__ __launch_bounds__(256) void k_fm_verify2(const SA * __restrict__ sa, const uint8_t * __restrict__ text,
                                                    const uint8_t * __restrict__ pats, uint32_t m, uint32_t n_pat,
                                                    uint64_t * __restrict__ out)
{
    constexpr bool WIDE = sizeof(SA) == 8; // the pending word of an index of 2^32 suffixes and more: [1 | s - 1 : 3 | length : 20 | suffix : 40]
    for (uint32_t q = blockIdx.x * blockDim.x + threadIdx.x; q < n_pat; q += gridDim.x * blockDim.x)
    {
        const uint64_t v = out[q];
        if (!(v >> 63))
            continue;
        const uint64_t l = WIDE ? v & ((UINT64_C(1) << 40) - 1) : (uint64_t)(uint32_t)v;
        const uint32_t rem = WIDE ? (uint32_t)(v >> 40) & 0xFFFFFu : (uint32_t)(v >> 32) & 0x0FFFFFFFu;
        const uint32_t ns = (uint32_t)(v >> 60) & 7u; // suffixes - 1
        const uint8_t * p = pats + (uint64_t)q * m;
        uint64_t at[kFmVerifyMax];
#pragma unroll
        for (uint32_t j = 0; j < kFmVerifyMax; ++j)
            at[j] = j <= ns ? (uint64_t)sa[l + j] : 0; // (all requested before the first text byte is)
        uint64_t pw[2] = {0, 0}; // the pattern's first 16 of the remaining bytes (rem <= 16: every 20-byte pattern behind a k-mer table)
        if (rem >= 8)
        {
            __builtin_memcpy(&pw[0], p, 8);
            __builtin_memcpy(&pw[1], p + (rem >= 16 ? 8 : rem - 8), 8);
        }
        uint32_t cnt = 0;
#pragma unroll
        for (uint32_t j = 0; j < kFmVerifyMax; ++j)
        {
            if (j > ns)
                break;
            bool ok = at[j] >= rem;
            if (ok)
            {
                const uint8_t * t = text + (at[j] - rem);
                if (rem >= 8)
                {
                    uint64_t x, y;
                    __builtin_memcpy(&x, t, 8);
                    __builtin_memcpy(&y, t + (rem >= 16 ? 8 : rem - 8), 8);
                    uint64_t diff = (x ^ pw[0]) | (y ^ pw[1]);
                    for (uint32_t k = 16; k + 8 <= rem; k += 8)
                    {
                        uint64_t a, b;
                        __builtin_memcpy(&a, t + k, 8);
                        __builtin_memcpy(&b, p + k, 8);
                        diff |= a ^ b;
                    }
                    if (rem > 16)
                    { // the last eight bytes (they may overlap the chunk before)
                        uint64_t a, b;
                        __builtin_memcpy(&a, t + rem - 8, 8);
                        __builtin_memcpy(&b, p + rem - 8, 8);
                        diff |= a ^ b;
                    }
                    ok = diff == 0;
                }
                else
                    for (uint32_t k = 0; k < rem; ++k)
                        ok = ok && t[k] == p[k];
            }
            cnt += ok ? 1u : 0u;
        }
        out[q] = cnt;
    }
}

// ---- the k-mer table: built from the suffix array and the text --------------------------------------------------------
// the (up to) eight text bytes at p as a little-endian number, 0 beyond the text's end
__device__ __forceinline__ uint64_t text_key8(const uint8_t * __restrict__ text, uint64_t n_text, uint64_t p)
{
    uint64_t v = 0;
    if (p + 8 <= n_text)
        __builtin_memcpy(&v, text + p, 8);
    else
        for (uint64_t j = 0; p + j < n_text; ++j)
            v |= (uint64_t)text[p + j] << (8 * j);
    return v;
}
__device__ __forceinline__ uint64_t low_bytes(uint64_t x, uint32_t k)
{
    return k >= 8 ? x : x & ((UINT64_C(1) << (8 * k)) - 1);
}

// dk[k] = number of distinct k-mers of the text, k = 1..8: suffix i starts a new run of k-mers when its first k bytes differ
// from its predecessor's in suffix order (a suffix shorter than k has none; the text holds no 0 byte, so its padded key has one)
template <class SA>
__global__ __launch_bounds__(256) void k_deep_census(const SA * __restrict__ sa, uint64_t n, const uint8_t * __restrict__ text,
                                                     uint64_t n_text, unsigned long long * __restrict__ dk)
{
    __shared__ unsigned cnt[9];
    if (threadIdx.x < 9)
        cnt[threadIdx.x] = 0;
    __syncthreads();
    unsigned mine[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0};
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (uint64_t)gridDim.x * blockDim.x)
    {
        const uint64_t p = sa[i];
        const uint64_t x = text_key8(text, n_text, p), y = i ? text_key8(text, n_text, sa[i - 1]) : 0;
        const uint64_t d = x ^ y;
        const unsigned fd = d ? (unsigned)(__builtin_ctzll(d) >> 3) : 8u; // first differing byte
#pragma unroll
        for (unsigned k = 1; k <= 8; ++k)
            if (p + k <= n_text && (i == 0 || fd < k))
                ++mine[k];
    }
    for (unsigned k = 1; k <= 8; ++k)
        if (mine[k])
            atomicAdd(&cnt[k], mine[k]);
    __syncthreads();
    if (threadIdx.x >= 1 && threadIdx.x < 9 && cnt[threadIdx.x])
        atomicAdd(&dk[threadIdx.x], (unsigned long long)cnt[threadIdx.x]);
}

// PASS 0: the first suffix of every k-mer's run claims a slot and stores l; PASS 1: the last one finds the slot and stores e
// (SA = uint64_t: an index of 2^32 suffixes and more, k <= 6; bits 32..39 of l and e go into the two top bytes of the key word)
template <int PASS, class SA>
__global__ __launch_bounds__(256) void k_deep_fill(const SA * __restrict__ sa, uint64_t n, const uint8_t * __restrict__ text,
                                                   uint64_t n_text, uint32_t k, unsigned long long * __restrict__ tab,
                                                   uint32_t n_buckets, unsigned * __restrict__ failed)
{
    constexpr bool WIDE = sizeof(SA) == 8;
    constexpr unsigned long long kmask = WIDE ? (1ull << 48) - 1 : ~0ull;
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (uint64_t)gridDim.x * blockDim.x)
    {
        const uint64_t p = sa[i];
        if (p + k > n_text)
            continue;
        const uint64_t key = low_bytes(text_key8(text, n_text, p), k);
        const uint64_t j = PASS == 0 ? i - 1 : i + 1;
        if (j < n)
        { // (i - 1 wraps for i == 0)
            const uint64_t pj = sa[j];
            if (pj + k <= n_text && low_bytes(text_key8(text, n_text, pj), k) == key)
                continue; // not the end of the run this pass looks for
        }
        uint32_t b = deep_bucket(key, n_buckets);
        bool done = false;
        for (uint32_t tries = 0; tries < n_buckets && !done; ++tries)
        {
            unsigned long long * slot = tab + (uint64_t)b * 16;
            for (int t = 0; t < 8 && !done; ++t)
            {
                if (PASS == 0)
                {
                    const unsigned long long kw = (unsigned long long)key | (WIDE ? ((unsigned long long)(i >> 32) & 0xFFull) << 48 : 0ull);
                    if (atomicCAS(slot + 2 * t, 0ull, kw) == 0ull)
                    {
                        reinterpret_cast<uint32_t *>(slot + 2 * t + 1)[0] = (uint32_t)i;
                        done = true;
                    }
                }
                else if ((slot[2 * t] & kmask) == key)
                {
                    reinterpret_cast<uint32_t *>(slot + 2 * t + 1)[1] = (uint32_t)(i + 1);
                    if (WIDE)
                        atomicOr(slot + 2 * t, (((unsigned long long)(i + 1) >> 32) & 0xFFull) << 56);
                    done = true;
                }
            }
            b = b + 1 == n_buckets ? 0 : b + 1;
        }
        if (!done)
            atomicAdd(failed, 1u);
    }
}

} // namespace sdslhip

using namespace sdslhip;

namespace sdslhip {

// the per-character tables of k_fm_count_flat, from the node table the fused layout was built on
sdsl_hip_status fm_build_count_tab(sdsl_hip_fm_s * f)
{
    f->ctab_ok = false;
    f->d_ctab.release();
    const WtHost & w = sdsl_hip_wt_host(f->wt);
    if (w.backend != 0 || !w.d_fused.p || !w.d_ftables.p || f->size >= kLimFmFastSymbols || f->sigma < 2)
        return SDSL_HIP_OK;
    const bool wide = f->size >= (UINT64_C(1) << 32);
    std::vector<WtFusedTables> ft(1);
    SH_HIP(hipMemcpy(&ft[0], w.d_ftables.p, sizeof(WtFusedTables), hipMemcpyDeviceToHost));
    const WtTables & T = w.d_tables_f.p ? w.tables_f : w.tables;
    std::vector<FmCountTabW> store(1);
    FmCountTabW & C = store[0];
    memset(&C, 0, sizeof C);
    uint32_t used = 0;
    for (unsigned c = 0; c < 256; ++c)
    {
        const unsigned cc = f->tab.char2comp[c];
        if (cc == 0 && c > 0)
            continue;
        if (T.c_to_leaf[c] == kWtUndef)
            return SDSL_HIP_OK; // (an alphabet symbol without a leaf: not a tree this path understands)
        uint64_t p = T.path[c];
        unsigned left = (unsigned)(p >> 56), v = 0, steps = 0;
        const uint32_t first = used;
        while (left)
        {
            const unsigned k = left < kFK ? left : kFK, t = (unsigned)p & ((1u << k) - 1u);
            if (used >= kFmMaxSteps || ft[0].fline[v] >= (1u << kLimStepTableLineBits) || v >= w.n_nodes)
                return SDSL_HIP_OK;
            C.snode[used] = (uint16_t)v;
            C.steps[used++] = ft[0].fline[v] | (t << 28);
            for (unsigned j = 0, tt = t; j < kFK; ++j, tt >>= 1)
            { // wt_descend
                const unsigned nv = T.child[v][tt & 1];
                v = nv == kWtUndef ? v : nv;
            }
            p >>= k;
            left -= k;
            ++steps;
        }
        if (steps == 0 || steps > 255)
            return SDSL_HIP_OK;
        C.cb[c] = (uint32_t)f->tab.C[cc];
        C.cbh[c] = (uint32_t)(f->tab.C[cc] >> 32);
        C.meta[c] = first | (steps << 16);
    }
    C.n_cross = ft[0].n_cross;
    memcpy(C.cross_key, ft[0].cross_key, sizeof C.cross_key);
    memcpy(C.cross_pos, ft[0].cross_pos, sizeof C.cross_pos);
    // (the narrow kernel reads the first sizeof(FmCountTab) bytes, the wide one — 2^32 symbols and more — all of it)
    const size_t bytes = wide ? sizeof(FmCountTabW) : sizeof(FmCountTab);
    SH_TRY(f->d_ctab.alloc(bytes));
    SH_HIP(hipMemcpy(f->d_ctab.p, &C, bytes, hipMemcpyHostToDevice));
    f->ctab_ok = true;
    return SDSL_HIP_OK;
}

// Builds the k-mer table from the resident suffix array and text.  k = the deepest depth (<= k_max <= 8) whose table — 16 bytes
// per k-mer at a load of one half, up to 0.7 where the budget is that tight — stays within `budget_bytes`; k_max == 0 releases the table.
sdsl_hip_status fm_build_deep(sdsl_hip_fm_s * f, uint32_t k_max, uint64_t budget_bytes)
{
    auto release = [&]() {
        f->d_deep.release();
        f->deep_k = 0;
        f->deep_buckets = 0;
        f->deep_kmers = 0;
    };
    if (k_max == 0)
    {
        release();
        return SDSL_HIP_OK;
    }
    // preconditions first: a call that cannot build leaves the table the index has (it may be one that can no longer be rebuilt)
    const bool wide = f->size >= (UINT64_C(1) << 32); // 64-bit suffix array, 40-bit intervals, k <= 6
    if (!(wide ? f->d_sa64.p : f->d_sa.p) || !f->d_text.p || f->size < 2 || f->size >= kLimFmFastSymbols)
    {
        set_error("the k-mer table is built from the whole suffix array and the text: create the index from text (and before "
                  "sdsl_hip_fm_drop_sa), or sdsl_hip_fm_restore_suffix_array first");
        return SDSL_HIP_ERR_UNSUPPORTED;
    }
    SH_HIP(hipSetDevice(f->device));
    if (k_max > (wide ? 6u : 8u))
        k_max = wide ? 6u : 8u;
    const uint64_t n = f->size, n_text = f->size - 1;
    DevBuf d_dk;
    SH_TRY(d_dk.alloc(9 * 8 + 8, true));
    const unsigned grid = grid_for(n, 256, 256u * 16u);
    if (wide)
        hipLaunchKernelGGL(k_deep_census<uint64_t>, dim3(grid), dim3(256), 0, 0, f->d_sa64.as<uint64_t>(), n, f->d_text.as<uint8_t>(), n_text,
                           d_dk.as<unsigned long long>());
    else
        hipLaunchKernelGGL(k_deep_census<uint32_t>, dim3(grid), dim3(256), 0, 0, f->d_sa.as<uint32_t>(), n, f->d_text.as<uint8_t>(), n_text,
                           d_dk.as<unsigned long long>());
    SH_HIP(hipGetLastError());
    uint64_t dk[10];
    SH_HIP(hipMemcpy(dk, d_dk.p, 10 * 8, hipMemcpyDeviceToHost));
    uint32_t k = 0;
    // a depth fits when its k-mers fill the budget's buckets to 0.7 at most (eight slots per bucket: one bucket in twenty then hands a
    // k-mer on to its neighbour); with room to spare the table is laid out at a load of one half
    for (uint32_t t = 1; t <= k_max; ++t)
        if (dk[t] && dk[t] * 23 <= budget_bytes)
            k = t;
    if (getenv("SDSL_HIP_TRACE_BUILD"))
        fprintf(stderr, "[sdsl_hip] k-mer census: D1..D8 = %llu %llu %llu %llu %llu %llu %llu %llu, budget %llu MiB -> k = %u\n",
                (unsigned long long)dk[1], (unsigned long long)dk[2], (unsigned long long)dk[3], (unsigned long long)dk[4],
                (unsigned long long)dk[5], (unsigned long long)dk[6], (unsigned long long)dk[7], (unsigned long long)dk[8],
                (unsigned long long)(budget_bytes >> 20), k);
    const uint64_t nb = k ? std::max<uint64_t>(1, std::min<uint64_t>((dk[k] + 3) / 4, budget_bytes / 128)) : 0; // eight slots per bucket, four (at most 5.6) taken on average
    if (k == 0 || nb >= (UINT64_C(1) << 32))
    { // no depth fits the budget: the caller asked for a table smaller than the smallest
        release();
        return SDSL_HIP_OK;
    }
    if (k == f->deep_k && nb == f->deep_buckets)
        return SDSL_HIP_OK; // the table the index has is the one asked for
    DevBuf d_new; // built beside the old table, swapped in when it is complete
    SH_TRY(d_new.alloc(nb * 128, true));
    unsigned * failed = reinterpret_cast<unsigned *>(d_dk.as<unsigned long long>() + 9);
    if (wide)
    {
        hipLaunchKernelGGL((k_deep_fill<0, uint64_t>), dim3(grid), dim3(256), 0, 0, f->d_sa64.as<uint64_t>(), n, f->d_text.as<uint8_t>(),
                           n_text, k, d_new.as<unsigned long long>(), (uint32_t)nb, failed);
        hipLaunchKernelGGL((k_deep_fill<1, uint64_t>), dim3(grid), dim3(256), 0, 0, f->d_sa64.as<uint64_t>(), n, f->d_text.as<uint8_t>(),
                           n_text, k, d_new.as<unsigned long long>(), (uint32_t)nb, failed);
    }
    else
    {
        hipLaunchKernelGGL((k_deep_fill<0, uint32_t>), dim3(grid), dim3(256), 0, 0, f->d_sa.as<uint32_t>(), n, f->d_text.as<uint8_t>(), n_text,
                           k, d_new.as<unsigned long long>(), (uint32_t)nb, failed);
        hipLaunchKernelGGL((k_deep_fill<1, uint32_t>), dim3(grid), dim3(256), 0, 0, f->d_sa.as<uint32_t>(), n, f->d_text.as<uint8_t>(), n_text,
                           k, d_new.as<unsigned long long>(), (uint32_t)nb, failed);
    }
    SH_HIP(hipGetLastError());
    unsigned bad = 0;
    SH_HIP(hipMemcpy(&bad, failed, 4, hipMemcpyDeviceToHost));
    if (bad)
        return SDSL_HIP_OK; // cannot happen at this load; the index keeps what it had and is complete without any table
    f->d_deep = std::move(d_new);
    f->deep_k = k;
    f->deep_buckets = (uint32_t)nb;
    f->deep_kmers = dk[k];
    return SDSL_HIP_OK;
}

// the default table of an index created from text: as deep as fits into the wavelet tree's own size (SDSL_HIP_FM_DEEP=<k_max>,
// 0 = none; SDSL_HIP_FM_DEEP_MB=<budget>)
sdsl_hip_status fm_build_deep_default(sdsl_hip_fm_s * f)
{
    const char * ek = getenv("SDSL_HIP_FM_DEEP");
    const char * eb = getenv("SDSL_HIP_FM_DEEP_MB");
    const uint32_t k_max = ek ? (uint32_t)std::max(0, atoi(ek)) : 8u;
    const uint64_t budget = eb ? (uint64_t)atoll(eb) << 20 : std::max<uint64_t>(UINT64_C(1) << 20, sdsl_hip_wt_device_bytes(f->wt));
    // (the suffix array of the index's width: a small text sent through the 64-bit sorter by SDSL_HIP_SA64 has none of 32 bits)
    // who reads it: the flat kernels of the plain index (ctab_ok), and the lane kernel of the rrr-compressed one below 2^32 symbols
    const bool rrr_user = sdsl_hip_wt_host(f->wt).backend == 1 && f->size < (UINT64_C(1) << 32);
    if (!(f->ctab_ok || rrr_user) || !(f->size >= (UINT64_C(1) << 32) ? f->d_sa64.p : f->d_sa.p) || !f->d_text.p || f->size < 2)
        return SDSL_HIP_OK;
    return fm_build_deep(f, k_max, budget);
}

static bool fm_fast_enabled()
{ // SDSL_HIP_FM_FAST=0: count() keeps to k_fm_count (fm.hip), for A/B
    static const bool on = !(getenv("SDSL_HIP_FM_FAST") && atoi(getenv("SDSL_HIP_FM_FAST")) == 0);
    return on;
}

bool fm_fast_applies(const sdsl_hip_fm_s * f, uint32_t m, uint64_t n_pat)
{
    static const uint64_t min_pat = getenv("SDSL_HIP_FM_FAST_MIN") ? (uint64_t)atoll(getenv("SDSL_HIP_FM_FAST_MIN")) : 4096;
    const bool wide = f->size >= (UINT64_C(1) << 32); // (a wide record keeps the characters left in 16 bits)
    // (a pattern longer than the index counts 0 whatever it holds — suffix_array_algorithm.hpp:466-467; the index is cyclic through its
    // sentinel, so a search would let "a\\0a" occur in the text "a" — the lock-step kernel knows the rule, such batches go there)
    return fm_fast_enabled() && f->ctab_ok && m >= 1 && m <= f->size && m < (wide ? 65535u : (1u << 30)) && n_pat >= min_pat && (!f->deep_k || m >= f->deep_k);
}

// count() of n_pat patterns of m bytes each, all in device memory; slabs of at most 2^25 patterns share one scratch area
sdsl_hip_status fm_count_fast(sdsl_hip_fm_s * f, const uint8_t * d_pats, uint32_t m, uint64_t n_pat, uint64_t * d_out, bool verify,
                              hipStream_t s)
{
    static const uint64_t slab_log = getenv("SDSL_HIP_FM_SLAB_LOG2") ? (uint64_t)atoi(getenv("SDSL_HIP_FM_SLAB_LOG2")) : 25;
    const uint64_t slab = UINT64_C(1) << (slab_log >= 10 && slab_log <= 31 ? slab_log : 25);
    const uint64_t per = std::min(slab, n_pat);
    const WtHost & w = sdsl_hip_wt_host(f->wt);
    const uint64_t n_slabs = (n_pat + per - 1) / per;
    // the slab's records + one chunk counter per slab: from the device's scratch pool (bv_host.hpp: ScratchLease — one user at a time,
    // ordered across streams by the pool's event).  NOT the stream-ordered allocator: two handles answering on two streams of one
    // device were handed overlapping blocks by hipMallocAsync now and then (the first record of a slab came out as the other
    // stream's: tools/cpp/group_stress.cpp, 4 rounds in 1500).
    static DevBuf no_capture_scratch;
    ScratchLease L;
    SH_TRY(L.acquire(f->device, no_capture_scratch, per * sizeof(FmRec) + n_slabs * 4 + 64, s));
    if (!L.p)
        return SDSL_HIP_ERR_NOMEM; // (the caller falls back to the lock-step kernel)
    void * scratch = L.p;
    FmRec * recs = reinterpret_cast<FmRec *>(scratch);
    uint32_t * ctr = reinterpret_cast<uint32_t *>(reinterpret_cast<uint8_t *>(scratch) + per * sizeof(FmRec));
    SH_TRY(fill_u32_async(ctr, 0u, n_slabs * 4, s));
    hipError_t e = hipSuccess;
    const FmDeep D = f->deep();
    FmJump J = f->jump();
    if (J.k > 8)
        J.tab = nullptr;
    const bool wide = f->size >= (UINT64_C(1) << 32);
    verify = verify && (wide ? f->d_sa64.p : f->d_sa.p) && f->d_text.p;
    const uint64_t csa_size = f->size;
    // one slab: start kernel (k-mer table, or the dense table / the whole interval), flat search, text comparison
    auto run_slab = [&](auto verify_c, auto wide_c, const uint8_t * pp, uint32_t cnt, uint64_t * oo, uint32_t * ticket) {
        constexpr bool V = decltype(verify_c)::value, W = decltype(wide_c)::value;
        if (D.tab)
            hipLaunchKernelGGL((k_fm_start<V, W>), dim3(grid_for(cnt, 64, 256u * 8u)), dim3(256), 0, s, D, csa_size, pp, m, cnt, oo, recs);
        else
            hipLaunchKernelGGL((k_fm_start_dense<V, W>), dim3(grid_for(cnt, 256, 256u * 8u)), dim3(256), 0, s, J, f->d_tab.as<FmTables>(),
                               csa_size, pp, m, cnt, oo, recs);
        hipLaunchKernelGGL((k_fm_count_flat<V, W>), dim3(grid_for(cnt, 256, 256u * 8u)), dim3(256), 0, s, w.d_fused.as<uint64_t>(),
                           w.d_fsuper.as<uint32_t>(), f->d_ctab.as<typename FmCtabOf<W>::type>(), recs, cnt, ticket, pp,
                           m, oo);
        if (V)
        {
            if (W)
                hipLaunchKernelGGL(k_fm_verify2<uint64_t>, dim3(grid_for(cnt, 256, 256u * 16u)), dim3(256), 0, s, f->d_sa64.as<uint64_t>(),
                                   f->d_text.as<uint8_t>(), pp, m, cnt, oo);
            else
                hipLaunchKernelGGL(k_fm_verify2<uint32_t>, dim3(grid_for(cnt, 256, 256u * 16u)), dim3(256), 0, s, f->d_sa.as<uint32_t>(),
                                   f->d_text.as<uint8_t>(), pp, m, cnt, oo);
        }
    };
    for (uint64_t i = 0; i < n_slabs && e == hipSuccess; ++i)
    {
        const uint64_t lo = i * per;
        const uint32_t cnt = (uint32_t)std::min(per, n_pat - lo);
        const uint8_t * pp = d_pats + lo * m;
        uint64_t * oo = d_out + lo;
        uint32_t * ticket = ctr + i;
        if (wide)
        {
            if (verify)
                run_slab(std::true_type(), std::true_type(), pp, cnt, oo, ticket);
            else
                run_slab(std::false_type(), std::true_type(), pp, cnt, oo, ticket);
        }
        else if (verify)
            run_slab(std::true_type(), std::false_type(), pp, cnt, oo, ticket);
        else
            run_slab(std::false_type(), std::false_type(), pp, cnt, oo, ticket);
        e = hipGetLastError();
    }
    SH_HIP(e);
    return SDSL_HIP_OK;
}

} // namespace sdslhip
