// bv_swc.hip — the bucketed batch path with STATIC STREAMS and WRITE COMBINING (round 3).
//
// Same idea as bv_sorted.hip — partition the batch by the 64 KiB slice of the index each position addresses, answer every
// slice out of LDS, send the answers back through the partition — and the same key layout, slots and per-tile histograms.
// What changes is HOW a partition pass writes.  profiles/stream_probe_r03.txt: a pass that appends runs of ~32 keys to
// 196 608 streams wherever the previous run stopped moves 10 B/key in 3.0 ms per 10^9 keys with nothing else in the way;
// when every store covers whole 128-byte lines it takes 1.9 ms.  Whole lines need a stream whose consecutive runs come from
// ONE block (the block keeps the tail that does not fill a line in LDS and writes it with the next run — software write
// combining), and that needs every stream's start before the pass runs:
//
//   k_sw_hist        one pass over the positions (the only one besides the partition itself) counts, per input range,
//                    BOTH the keys per pass-1 bin (as before) and the keys per SLICE — a 2^16-counter histogram held in
//                    LDS as 16-bit fields (a field that reaches 2^15 is moved to the global row by the one thread that saw
//                    it cross: exact for any distribution)
//   tables           pass-1 stream starts; slice starts; per (segment, slice) starts for pass 2
//   k_sw_partition<1> unit = a contiguous range of the batch; streams (unit, bin), write-combined
//   k_sw_partition<2> unit = (pass-1 bin, segment of the batch); streams (unit, low digit) start at the slice's start +
//                    the slice's keys in earlier segments: no look-back, no second histogram pass, write-combined
//   answers          k_sr_rank_lds / k_sr_select_lds of bv_sorted.hip on tables in slice order
//   k_sw_unpermute<2>, <1>  the same units backwards (runs gathered into LDS, picked by slot)
//
// Units are handed to persistent blocks by a ticket counter; they are independent, so any order is correct.
// The answers are the reference's (rank_support_v5.hpp:131-149, select_support_mcl.hpp:384-439); batching is this
// library's addition.
#include <algorithm>
#include <type_traits>
#include <vector>

#include "bv_sorted_dev.hpp"

namespace sdslhip {

namespace {

constexpr unsigned kHB = 256;  // histogram blocks = rows of fine_h (one block per CU: the slice histogram fills its LDS)
constexpr unsigned kHT = 1024; // threads of a histogram block
constexpr unsigned kCA = 32;   // keys per write-combining chunk (128 B)
constexpr unsigned kMaxK = 64; // segments per pass-1 bin
constexpr unsigned kMoveAt = 0x1000; // every kMoveAt-th increment of a 16-bit slice counter moves that much to the global row
constexpr unsigned kCkS = 8;   // the partition notes every stream's position at every kCkS-th tile

struct SwGeom
{
    uint32_t U1;         // pass-1 units = streams per pass-1 bin (a multiple of kHB)
    uint32_t tpu;        // tiles per pass-1 unit
    uint32_t K;          // segments of the batch = pass-2 units per pass-1 bin (divides kHB)
    uint32_t nf;         // slices: 1 << (d1 + d2)
    uint32_t nfw;        // row stride of fine_h / segsum
    uint32_t fine_words; // LDS words of the slice histogram
};

struct SwBuf
{
    uint32_t *keys1, *keys2;
    uint16_t *slots1, *slots2;
    uint16_t *thist1, *thist2; // [tile][bin]
    uint32_t *counts1, *offs1, *bstart1, *btot;
    uint32_t *fine_h;          // [kHB][nfw]: keys per slice of each histogram block's input range
    uint32_t *segsum;          // [K][nfw]: a slice's keys in earlier segments
    uint32_t *tot, *fstart, *ioff;
    uint64_t * hf;
    uint32_t *in_lo, *tp2;     // pass-2 units: first key, first tile
    uint32_t *ck1, *ck2;       // [tile / kCkS][bin]: stream positions noted by the partition passes
    uint64_t *tdesc2;          // per tile of pass 2: first key | keys << 32 | unit << 46 | first tile of its unit << 63
    uint32_t *tickets;         // 4 counters
    uint32_t *marked;
};

size_t sw_carve(SwBuf & b, void * scratch, uint64_t n, unsigned tile, const SwGeom & w, unsigned bins1)
{
    uint8_t * p = (uint8_t *)scratch;
    auto take = [&](size_t bytes) -> void *
    {
        void * r = p;
        p += (bytes + 255) & ~(size_t)255;
        return r;
    };
    const uint64_t tiles1 = (n + tile - 1) / tile + w.U1, tiles2 = (n + tile - 1) / tile + 2 * (uint64_t)bins1 * w.K;
    b.keys1 = (uint32_t *)take(n * 4 + 256);
    b.keys2 = (uint32_t *)take(n * 4 + 256);
    b.slots1 = (uint16_t *)take(n * 2);
    b.slots2 = (uint16_t *)take(n * 2);
    b.thist1 = (uint16_t *)take(tiles1 * kBins * 2);
    b.thist2 = (uint16_t *)take(tiles2 * kBins * 2);
    b.counts1 = (uint32_t *)take((size_t)kBins * w.U1 * 4);
    b.offs1 = (uint32_t *)take(((size_t)kBins * w.U1 + 1) * 4);
    b.bstart1 = (uint32_t *)take((kBins + 1) * 4);
    b.btot = (uint32_t *)take((kBins + 1) * 4);
    b.fine_h = (uint32_t *)take((size_t)kHB * w.nfw * 4);
    b.segsum = (uint32_t *)take((size_t)w.K * w.nfw * 4);
    b.tot = (uint32_t *)take((size_t)w.nfw * 4);
    b.fstart = (uint32_t *)take(((size_t)w.nf + 1) * 4);
    b.ioff = (uint32_t *)take(((size_t)w.nf + 1) * 4);
    b.hf = (uint64_t *)take((size_t)w.nf * 8);
    b.in_lo = (uint32_t *)take(((size_t)kBins * kMaxK + 1) * 4);
    b.tp2 = (uint32_t *)take(((size_t)kBins * kMaxK + 1) * 4);
    b.ck1 = (uint32_t *)take((tiles1 / kCkS + 2) * kBins * 4);
    b.ck2 = (uint32_t *)take((tiles2 / kCkS + 2) * kBins * 4);
    b.tdesc2 = (uint64_t *)take((tiles2 + 64) * 8);
    b.tickets = (uint32_t *)take(256);
    b.marked = (uint32_t *)take(256);
    return (size_t)(p - (uint8_t *)scratch);
}

// ---- the one counting pass ------------------------------------------------------------------------------------------
// Block h counts the keys of pass-1 units [h * U1 / kHB, ...): per unit the keys per pass-1 bin (counts1[bin][unit]), over
// the whole range the keys per slice (row h of fine_h, zeroed by the caller).  The slice histogram lives in LDS as two
// 16-bit fields per word.  A field never overflows: the thread whose increment is the field's kMoveAt-th, 2 kMoveAt-th, ...
// (the returning atomic tells: old value = kMoveAt - 1 modulo kMoveAt — moving kMoveAt out does not change that residue,
// so exactly one thread per kMoveAt increments qualifies whenever the moves land) subtracts kMoveAt and adds it to the
// global row.  A move lands at most one LDS queue (a few 10^4 increments) behind its trigger, so the field stays below
// kMoveAt + that, and it is at least kMoveAt when the move lands: no overflow, no borrow, for any distribution.
// (The first form triggered on the value 2^15 - 1 itself: a move that landed more than 2^15 increments late left the field
// above the trigger for good, and a batch of 10^8 positions inside three slices lost 65536 keys of one — found by the
// reference-digest test on the windowed batch.)
template <unsigned PER, int OP>
__global__ __launch_bounds__(kHT) void k_sw_hist(SrGeom g, SwGeom w, const uint64_t * __restrict__ idx, uint32_t * __restrict__ counts1,
                                                 uint32_t * __restrict__ fine_h)
{
    if (g.go && !*g.go)
        return;
    extern __shared__ uint32_t sw_lds[];
    uint32_t * fine = sw_lds;
    uint32_t * uhist = sw_lds + w.fine_words + 1; // kBins + 1 (the last one takes what lies beyond the range)
    const unsigned t = threadIdx.x, h = blockIdx.x;
    for (unsigned i = t; i <= w.fine_words; i += kHT)
        fine[i] = 0;
    const unsigned upb = w.U1 / kHB;
    uint32_t * row = fine_h + (size_t)h * w.nfw;
    const uint64_t ulen = (uint64_t)w.tpu * g.tile;
    for (unsigned uu = 0; uu < upb; ++uu)
    {
        const unsigned unit = h * upb + uu;
        for (unsigned i = t; i <= kBins; i += kHT)
            uhist[i] = 0;
        __syncthreads();
        const uint64_t klo = (uint64_t)unit * ulen < g.n ? (uint64_t)unit * ulen : g.n;
        const uint64_t khi = klo + ulen < g.n ? klo + ulen : g.n;
        // chunks of kHT * PER positions; the next chunk is requested before this one is counted
        auto fetch = [&](uint64_t c, uint64_t (&p)[PER])
        {
#pragma unroll
            for (unsigned u = 0; u < PER; ++u)
            {
                const uint64_t q = c + u * kHT + t;
                p[u] = __builtin_nontemporal_load(idx + (q < khi ? q : khi - 1));
            }
        };
        uint64_t p[PER];
        if (klo < khi)
            fetch(klo, p);
        for (uint64_t c = klo; c < khi; c += (uint64_t)kHT * PER)
        {
            uint64_t pn[PER];
            const uint64_t cn = c + (uint64_t)kHT * PER;
            if (cn < khi)
                fetch(cn, pn);
            unsigned fidv[PER]; // slice of the key | what its counter held << 16
            // (a chunk that lies inside the unit — all but the last one — needs no per-key range checks: 5 of 44 instructions)
            auto count = [&](auto full)
            {
                constexpr bool FULL = decltype(full)::value;
#pragma unroll
                for (unsigned u = 0; u < PER; ++u)
                {
                    unsigned dig;
                    uint32_t key;
                    sr_key1_t<OP>(p[u], g, dig, key);
                    const bool on = FULL || c + u * kHT + t < khi;
                    atomicAdd(&uhist[on ? dig : kBins], 1u);
                    const unsigned fid = key >= kMark ? 0u : (dig << g.d2) | (key >> g.kb);
                    const unsigned sh = (fid & 1u) << 4;
                    const uint32_t old = atomicAdd(&fine[on ? fid >> 1 : w.fine_words], on ? 1u << sh : 0u);
                    fidv[u] = on ? fid | (((old >> sh) & 0xFFFFu) << 16) : 0u;
                }
            };
            if (cn <= khi)
                count(std::true_type{});
            else
                count(std::false_type{});
            // an increment that was its field's 4096th, 8192nd, ...: move that much to the global row.  Rare (one key in 4096 of
            // a slice's): one test per chunk instead of one branch per key
            bool any = false;
#pragma unroll
            for (unsigned u = 0; u < PER; ++u)
                any |= ((fidv[u] >> 16) & (kMoveAt - 1)) == kMoveAt - 1;
            if (any)
            {
#pragma unroll
                for (unsigned u = 0; u < PER; ++u)
                    if (((fidv[u] >> 16) & (kMoveAt - 1)) == kMoveAt - 1)
                    {
                        const unsigned fid = fidv[u] & 0xFFFFu;
                        atomicSub(&fine[fid >> 1], kMoveAt << ((fid & 1u) << 4));
                        atomicAdd(row + fid, kMoveAt);
                    }
            }
            if (cn < khi)
            {
#pragma unroll
                for (unsigned u = 0; u < PER; ++u)
                    p[u] = pn[u];
            }
        }
        __syncthreads();
        for (unsigned i = t; i < (1u << g.d1); i += kHT)
            counts1[(size_t)i * w.U1 + unit] = uhist[i];
        __syncthreads();
    }
    for (unsigned f = t; f < w.nf; f += kHT)
    {
        const uint32_t v = (fine[f >> 1] >> ((f & 1u) << 4)) & 0xFFFFu;
        if (v)
            atomicAdd(row + f, v);
    }
}

// per slice: its keys in earlier segments (segsum[k][f]) and in the whole batch (tot[f])
__global__ __launch_bounds__(256) void k_sw_seg_reduce(SwGeom w, const uint32_t * __restrict__ fine_h, uint32_t * __restrict__ segsum,
                                                       uint32_t * __restrict__ tot)
{
    const unsigned f = blockIdx.x * 256 + threadIdx.x;
    if (f >= w.nf)
        return;
    const unsigned hps = kHB / w.K;
    unsigned run = 0;
    for (unsigned k = 0; k < w.K; ++k)
    {
        segsum[(size_t)k * w.nfw + f] = run;
        for (unsigned j = 0; j < hps; ++j)
            run += fine_h[(size_t)(k * hps + j) * w.nfw + f];
    }
    tot[f] = run;
}

// pass-2 units x = (pass-1 bin, segment): first key in_lo[x] (the bin-major array of pass 1 is contiguous, so a unit ends
// where the next one starts) and first tile tp2[x]
__global__ __launch_bounds__(1024) void k_sw_units2(SrGeom g, SwGeom w, const uint32_t * __restrict__ offs1, uint32_t * __restrict__ in_lo,
                                                    uint32_t * __restrict__ tp2)
{
    if (g.go && !*g.go)
        return;
    __shared__ unsigned wred[16];
    const unsigned t = threadIdx.x;
    const unsigned n_units = (1u << g.d1) * w.K, upk = w.U1 / w.K;
    const unsigned per = (n_units + 1023) / 1024;
    auto lo_of = [&](unsigned x) -> uint32_t { return x < n_units ? offs1[(size_t)(x / w.K) * w.U1 + (x % w.K) * upk] : (uint32_t)g.n; };
    // (tiles end on 128-byte lines of the array: k_sw_partition<2>)
    auto tiles_of = [&](unsigned x) -> unsigned
    {
        const uint32_t a = lo_of(x), z = lo_of(x + 1);
        return z > a ? (z - (a & ~31u) + g.tile - 1) / g.tile : 0u;
    };
    const unsigned x0 = t * per, x1 = x0 + per < n_units ? x0 + per : n_units;
    unsigned s = 0;
    for (unsigned x = x0; x < x1; ++x)
        s += tiles_of(x);
    const unsigned inc = wave_incl_scan(s);
    if ((t & 63) == 63)
        wred[t >> 6] = inc;
    __syncthreads();
    unsigned base = inc - s;
    for (unsigned wv = 0; wv < (t >> 6); ++wv)
        base += wred[wv];
    for (unsigned x = x0; x < x1; ++x)
    {
        in_lo[x] = lo_of(x);
        tp2[x] = base;
        base += tiles_of(x);
    }
    if (t == 1023)
    { // (threads past the end have empty ranges, so the last thread holds the total)
        in_lo[n_units] = (uint32_t)g.n;
        tp2[n_units] = base;
    }
}

// what a unit of pass P covers and where its streams start
template <int P>
struct SwUnit
{
    uint64_t klo, khi;
    unsigned tbase, grp;
};
template <int P, unsigned TT>
__device__ __forceinline__ SwUnit<P> sw_unit_setup(const SrGeom & g, const SwGeom & w, unsigned unit, const uint32_t * __restrict__ offs1,
                                                   const uint32_t * __restrict__ in_lo, const uint32_t * __restrict__ tp2,
                                                   const uint32_t * __restrict__ fstart, const uint32_t * __restrict__ segsum,
                                                   unsigned * cursor)
{
    SwUnit<P> u;
    const unsigned t = threadIdx.x;
    if (P == 1)
    {
        const uint64_t ulen = (uint64_t)w.tpu * g.tile;
        u.klo = (uint64_t)unit * ulen < g.n ? (uint64_t)unit * ulen : g.n;
        u.khi = u.klo + ulen < g.n ? u.klo + ulen : g.n;
        u.tbase = unit * w.tpu;
        u.grp = 0;
        for (unsigned i = t; i < kBins; i += TT)
            cursor[i] = i < (1u << g.d1) ? offs1[(size_t)i * w.U1 + unit] : 0u;
    }
    else
    {
        u.grp = unit / w.K;
        const unsigned k = unit % w.K;
        u.klo = in_lo[unit];
        u.khi = in_lo[unit + 1];
        u.tbase = tp2[unit];
        for (unsigned i = t; i < kBins; i += TT)
        {
            const unsigned f = (u.grp << g.d2) | i;
            cursor[i] = i < (1u << g.d2) ? fstart[f] + segsum[(size_t)k * w.nfw + f] : 0u;
        }
    }
    return u;
}

// the carry row of bin b: position i of the row (bins of different parity use different bank halves: two bins share a half-wave)
__device__ __forceinline__ unsigned carry_at(unsigned b, unsigned i)
{
    return b * kCA + (i ^ ((b & 1u) << 4));
}


// ---- a partition pass with write combining ------------------------------------------------------------------------
// Per tile: counting sort in LDS as in bv_sorted.hip (returning atomic = place inside the tile's share of the bin, slot =
// place in the sorted tile).  Writing out: a stream's keys not yet written sit in its carry row (fewer than kCA); a tile's run
// is appended to them, every whole kCA-aligned chunk goes out (carry first), the tail becomes the new carry.  The global
// layout is exactly that of the unbuffered pass — only when and in which pieces a key is written changes — so slots and
// tile histograms mean what they always meant.  The next tile's keys are requested before this tile is written out.
template <int P, unsigned TT, unsigned PER, int OP>
__global__ __launch_bounds__(TT, 4) void k_sw_partition(SrGeom g, SwGeom w, const uint64_t * __restrict__ idx, const uint32_t * __restrict__ keys_in,
                                                       const uint32_t * __restrict__ offs1, const uint32_t * __restrict__ in_lo,
                                                       const uint32_t * __restrict__ tp2, const uint32_t * __restrict__ fstart,
                                                       const uint32_t * __restrict__ segsum, uint32_t * __restrict__ ticket,
                                                       uint32_t * __restrict__ keys_out, uint16_t * __restrict__ slots,
                                                       uint16_t * __restrict__ tile_hist, uint32_t * __restrict__ ckpt,
                                                       uint64_t * __restrict__ tdesc)
{
    if (g.go && !*g.go)
        return;
    constexpr unsigned kTile = TT * PER;
    typedef typename std::conditional<P == 1, uint64_t, uint32_t>::type raw_t;
    __shared__ uint32_t sorted[kTile + 1]; // (+ a place for what lies beyond a tile's end: stores need no branch)
    __shared__ uint32_t carry[kBins * kCA];
    __shared__ unsigned hist2[2][kBins + 1], start[kBins], cursor[kBins], ccnt[kBins], meta[kBins]; // hist2[.][kBins]: what lies beyond a tile's end
    __shared__ unsigned wsum[kBins / 64];
    __shared__ unsigned big[kTile / (kBigRun + 1) + 1], n_big, sh_unit;
    const unsigned t = threadIdx.x, l = t & 15;
    const unsigned bins = 1u << (P == 1 ? g.d1 : g.d2);
    const unsigned n_units = P == 1 ? w.U1 : (1u << g.d1) * w.K;
    const raw_t * __restrict__ in = P == 1 ? (const raw_t *)idx : (const raw_t *)keys_in;
    for (;;)
    {
        if (t == 0)
            sh_unit = atomicAdd(ticket, 1u);
        __syncthreads();
        const unsigned unit = __builtin_amdgcn_readfirstlane(sh_unit);
        if (unit >= n_units)
            break;
        SwUnit<P> un = sw_unit_setup<P, TT>(g, w, unit, offs1, in_lo, tp2, fstart, segsum, cursor);
        un.klo = uniform64(un.klo);
        un.khi = uniform64(un.khi);
        un.tbase = __builtin_amdgcn_readfirstlane(un.tbase);
        for (unsigned i = t; i < kBins; i += TT)
            ccnt[i] = 0;
        if (t == 0)
            n_big = 0;
        __syncthreads();
        if (un.klo >= un.khi)
            continue;
        // A tile's keys are requested one tile ahead, and they are WAITED FOR before the tile in front of them is written out:
        // vector loads and stores share one counter per wave, so a wait for loads that were issued after stores also waits for
        // the stores' acknowledgements — placed behind the write-out, every tile would drain its own stores before the next
        // one could even be counted (3.1 instead of 2.x ms per pass)
        // Pass 2: a unit starts wherever its stream of pass 1 starts, but its tiles END on 128-byte lines of the array (the first one
        // is short): the slots written here and, on the way back, the answers written by k_sw_unpermute_dma<2> then cover whole
        // lines — non-temporal stores that start mid-sector went out as partial writes (4.33 GB for 4.0 GB of answers).
        const uint64_t kal = P == 2 ? (un.klo & ~UINT64_C(31)) : un.klo;
        auto tile_end = [&](uint64_t lo) -> uint64_t
        {
            const uint64_t e = kal + ((lo - kal) / kTile + 1) * kTile;
            return e < un.khi ? e : un.khi;
        };
        auto fetch = [&](uint64_t lo, raw_t (&r)[PER])
        {
            const unsigned c = (unsigned)(tile_end(lo) - lo);
            const rsrc_t rs = make_rsrc(in + lo, c * (unsigned)sizeof(raw_t));
#pragma unroll
            for (unsigned u = 0; u < PER; ++u)
                buf_load(rs, t * (unsigned)sizeof(raw_t), u * TT * (unsigned)sizeof(raw_t), r[u]);
        };
        const rsrc_t rs_out = make_rsrc(keys_out, (uint32_t)g.n * 4u);
        raw_t raw[PER];
        fetch(un.klo, raw);
        unsigned ti = un.tbase;
        unsigned hb = 0; // which of the two histograms this tile counts into (the other one is cleared while this tile is written out)
        for (unsigned i = t; i <= kBins; i += TT)
            hist2[0][i] = 0;
        __syncthreads();
        for (uint64_t lo = un.klo, nlo; lo < un.khi; lo = nlo, ++ti, hb ^= 1u)
        {
            unsigned * hist = hist2[hb];
            nlo = tile_end(lo);
            const unsigned cnt_t = (unsigned)(nlo - lo);
            const bool has_next = nlo < un.khi;
            // every kCkS-th tile: where its runs start in the streams (what lets the way back begin there)
            const bool note = (ti % kCkS) == 0;
            unsigned ck = 0;
            if (note && t < bins)
                ck = cursor[t] + ccnt[t];
            uint32_t key[PER];
            unsigned br[PER]; // bin << 16 | rank inside the tile's share of the bin; later: the key's slot
            // (uniform; pass 1, all but a unit's last tile: no per-key range checks.  Pass 2 keeps them: the second copy of the loops
            // cost it more in registers than the checks cost in instructions — 2.63 -> 2.83 ms)
            const bool full_tile = P == 1 && cnt_t == kTile;
            auto count_keys = [&](auto full)
            {
                constexpr bool FULL = decltype(full)::value;
#pragma unroll
                for (unsigned u = 0; u < PER; ++u)
                {
                    const unsigned q = u * TT + t;
                    unsigned d;
                    if (P == 1)
                        sr_key1_t<OP>((uint64_t)raw[u], g, d, key[u]);
                    else
                        sr_key2((uint32_t)raw[u], g, d, key[u]);
                    // (what lies beyond the tile's end counts into a bin of its own: all the atomics are issued back to back)
                    br[u] = (d << 16) | atomicAdd(&hist[FULL || q < cnt_t ? d : kBins], 1u); // < 2^14
                }
            };
            if (full_tile)
                count_keys(std::true_type{});
            else
                count_keys(std::false_type{});
            raw_t nxt[PER]; // (requested once this tile's own keys are out of the way: the registers are the same)
            if (has_next)
                fetch(nlo, nxt);
            __syncthreads();
            { // exclusive scan of the counts -> start; the bins' write-out parameters packed into one word each
                unsigned v = 0, inc = 0;
                if (t < kBins)
                {
                    v = hist[t];
                    inc = wave_incl_scan(v);
                    if ((t & 63) == 63)
                        wsum[t >> 6] = inc;
                }
                __syncthreads();
                if (t < kBins)
                {
                    unsigned base = inc - v;
                    for (unsigned wv = 0; wv < (t >> 6); ++wv)
                        base += wsum[wv];
                    start[t] = base;
                    // first place in the sorted tile : 13 | keys : 14 | carried keys : 5 (a bin without keys may 'start' at kTile: 14 bits)
                    meta[t] = (v ? base : 0u) | (v << 13) | (ccnt[t] << 27);
                }
                __syncthreads();
            }
            if (full_tile)
            {
#pragma unroll
                for (unsigned u = 0; u < PER; ++u)
                {
                    const unsigned pos = start[br[u] >> 16] + (br[u] & 0xFFFFu);
                    br[u] = pos;
                    sorted[pos] = key[u];
                }
            }
            else
            {
#pragma unroll
                for (unsigned u = 0; u < PER; ++u)
                {
                    const unsigned q = u * TT + t;
                    const unsigned pos = start[br[u] >> 16] + (br[u] & 0xFFFFu);
                    br[u] = pos;
                    sorted[q < cnt_t ? pos : kTile] = key[u];
                }
            }
            __syncthreads();
            if (has_next)
            {
#pragma unroll
                for (unsigned u = 0; u < PER; ++u)
                    asm volatile("" : "+v"(nxt[u])); // the next tile's keys have landed; from here on only stores are in flight
            }
            for (unsigned i = t; i < bins; i += TT)
                tile_hist[(uint64_t)ti * bins + i] = (uint16_t)hist[i];
            if (note && t < bins)
                ckpt[(uint64_t)(ti / kCkS) * kBins + t] = ck;
            if (P == 2 && t == 0) // what the way back needs to know about this tile: first key | keys | unit | first tile of its unit
                tdesc[ti] = (uint64_t)lo | ((uint64_t)cnt_t << 32) | ((uint64_t)unit << 46) | ((uint64_t)(ti == un.tbase) << 63);
            {
                const rsrc_t rs = make_rsrc(slots + lo, cnt_t * 2u);
#pragma unroll
                for (unsigned u = 0; u < PER; ++u)
                    __builtin_amdgcn_raw_buffer_store_b16((unsigned short)br[u], rs, (int)(t * 2u), (int)(u * TT * 2u), kAuxNT);
            }
            for (unsigned i = t; i <= kBins; i += TT)
                hist2[hb ^ 1u][i] = 0;
            // chunks out: 16 lanes per bin; a bin's parameters are read while the bin in front of it is worked on
            constexpr unsigned kRounds = kBins / (TT / 16);
            unsigned m_nx = meta[t >> 4], c_nx = cursor[t >> 4];
#pragma nounroll
            for (unsigned k = 0; k < kRounds; ++k)
            {
                const unsigned b = (t >> 4) + k * (TT / 16);
                const unsigned m = m_nx, cur = c_nx;
                if (k + 1 < kRounds)
                {
                    m_nx = meta[b + TT / 16];
                    c_nx = cursor[b + TT / 16];
                }
                const unsigned st = m & 0x1FFFu, cnt = (m >> 13) & 0x3FFFu, cc = m >> 27;
                if (cnt == 0 || b >= bins)
                    continue;
                if (cnt > kBigRun)
                {
                    if (l == 0)
                        big[atomicAdd(&n_big, 1u)] = b;
                    continue;
                }
                const unsigned end = cur + cc + cnt, aend = end & ~(kCA - 1);
                if (aend > cur)
                {
                    const unsigned nw = aend - cur, rem = end - aend;
                    for (unsigned i = l; i < nw; i += 16)
                        __builtin_amdgcn_raw_buffer_store_b32(i < cc ? carry[carry_at(b, i)] : sorted[st + i - cc], rs_out, (int)((cur + i) * 4u), 0, 0);
                    const unsigned from = st + cnt - rem; // (the new carry always comes out of this tile's run)
                    if (l < rem)
                        carry[carry_at(b, l)] = sorted[from + l];
                    if (l + 16 < rem)
                        carry[carry_at(b, l + 16)] = sorted[from + l + 16];
                    if (l == 0)
                    {
                        ccnt[b] = rem;
                        cursor[b] = aend;
                    }
                }
                else
                { // not a whole chunk yet: the run joins the carry
                    for (unsigned j = l; j < cnt; j += 16)
                        carry[carry_at(b, cc + j)] = sorted[st + j];
                    if (l == 0)
                        ccnt[b] = cc + cnt;
                }
            }
            __syncthreads();
            const unsigned nb = n_big;
            if (nb)
            {
                __syncthreads();
                if (t == 0)
                    n_big = 0;
            }
            for (unsigned k = 0; k < nb; ++k)
            { // a long run (a skewed tile): the whole block writes it
                const unsigned b = big[k], cnt = hist[b], cc = ccnt[b], st = start[b], cur = cursor[b];
                const unsigned end = cur + cc + cnt, aend = end & ~(kCA - 1), nw = aend - cur, rem = end - aend;
                for (unsigned i = t; i < nw; i += TT)
                    keys_out[(uint64_t)cur + i] = i < cc ? carry[carry_at(b, i)] : sorted[st + i - cc];
                __syncthreads();
                if (t < rem)
                    carry[carry_at(b, t)] = sorted[st + cnt - rem + t];
                if (t == 0)
                {
                    ccnt[b] = rem;
                    cursor[b] = aend;
                }
                __syncthreads();
            }
            if (has_next)
            {
#pragma unroll
                for (unsigned u = 0; u < PER; ++u)
                    raw[u] = nxt[u];
            }
        }
        // the unit's last chunks (partial)
        for (unsigned b = t >> 4; b < bins; b += TT / 16)
        {
            const unsigned cc = ccnt[b], cur = cursor[b];
            if (l < cc)
                keys_out[cur + l] = carry[carry_at(b, l)];
            if (l + 16 < cc)
                keys_out[cur + l + 16] = carry[carry_at(b, l + 16)];
        }
        __syncthreads();
    }
}

// ---- the way back ------------------------------------------------------------------------------------------------------
template <int P>
struct SwTile
{
    uint64_t lo;
    unsigned cnt, unit, tb; // keys; unit; the unit's first tile
};
template <int P>
__device__ __forceinline__ SwTile<P> sw_tile(const SrGeom & g, const SwGeom & w, unsigned ti, unsigned unit_hint, const uint32_t * __restrict__ in_lo,
                                             const uint32_t * __restrict__ tp2)
{
    SwTile<P> d;
    if (P == 1)
    {
        const uint64_t ulen = (uint64_t)w.tpu * g.tile;
        d.unit = ti / w.tpu;
        d.tb = d.unit * w.tpu;
        d.lo = (uint64_t)d.unit * ulen + (uint64_t)(ti - d.tb) * g.tile;
        const uint64_t end = (uint64_t)(d.unit + 1) * ulen < g.n ? (uint64_t)(d.unit + 1) * ulen : g.n;
        d.cnt = d.lo < end ? (unsigned)(end - d.lo < g.tile ? end - d.lo : g.tile) : 0u;
    }
    else
    { // the unit of tile ti: the hint or one of the next (units without keys have no tiles)
        unsigned u = unit_hint;
        while (__builtin_amdgcn_readfirstlane(tp2[u + 1]) <= ti)
            ++u;
        d.unit = u;
        d.tb = __builtin_amdgcn_readfirstlane(tp2[u]);
        const uint32_t a = __builtin_amdgcn_readfirstlane(in_lo[u]), z = __builtin_amdgcn_readfirstlane(in_lo[u + 1]);
        const uint64_t al = a & ~31u, end = al + (uint64_t)(ti - d.tb + 1) * g.tile; // (tiles end on 128-byte lines: k_sw_partition<2>)
        d.lo = ti == d.tb ? (uint64_t)a : al + (uint64_t)(ti - d.tb) * g.tile;
        d.cnt = (unsigned)((end < z ? end : z) - d.lo);
    }
    return d;
}

// P == 2: slice-relative answers (order of partition 2) -> answers relative to their pass-1 bin, in the order of partition 1
// P == 1: those -> absolute answers in the caller's order
// The way back has no carry to keep, so it need not follow the units: the partition left every stream's position at every
// kCkS-th tile (ckpt) and, for pass 2, a descriptor of every tile (tdesc); the work is shared out in items of kCkS tiles.
// Per tile the runs are gathered into LDS (the bin-major image of the tile), then every position picks its answer by slot.
// The first form of the gather held the run elements in registers between request and use (24-32 VGPRs per lane, two rounds
// per tile at 512 threads: 2.7 ms per pass).  Here the fetch goes global -> LDS directly (buffer_load ... lds): position p of the bin-major tile image
// is fetched by lane p of its wave from  delta[bin(p)] + p  (delta = where the bin's run starts in the stream - where it starts
// in the image), sixteen requests per lane in flight and no register waiting for any of them; bin(p) comes from a byte map the
// bins' lane groups fill after the scan.  What makes an answer absolute is added when it is picked: one more table read per
// key instead of a pass over the image.  47 KB of LDS and < 80 VGPRs: three blocks per CU.
template <int P, unsigned TT, unsigned PER>
__global__ __launch_bounds__(TT, 6) void k_sw_unpermute_dma(const uint64_t * __restrict__ hf, int bit, SrGeom g, SwGeom w,
                                                           const uint32_t * __restrict__ offs1, const uint32_t * __restrict__ in_lo,
                                                           const uint32_t * __restrict__ tp2, const uint32_t * __restrict__ fstart,
                                                           const uint32_t * __restrict__ segsum, const uint32_t * __restrict__ ckpt,
                                                           uint32_t * __restrict__ ticket, const uint64_t * __restrict__ tdesc,
                                                           const uint32_t * __restrict__ res_lo,
                                                           uint32_t * __restrict__ any_marked, const uint16_t * __restrict__ slots,
                                                           const uint16_t * __restrict__ tile_hist, uint32_t * __restrict__ out_lo,
                                                           uint64_t * __restrict__ out)
{
    if (g.go && !*g.go)
        return;
    constexpr unsigned kTile = TT * PER;
    __shared__ uint32_t lo32[kTile];
    __shared__ uint8_t binof[kTile];
    __shared__ unsigned hist[kBins], start[kBins], cursor[kBins], delta[kBins];
    __shared__ uint64_t sbase[kBins]; // what makes the answers of bin b absolute (P == 1) / relative to the pass-1 bin (P == 2)
    __shared__ unsigned wsum[kBins / 64];
    __shared__ unsigned sh_item;
    const unsigned t = threadIdx.x, l = t & 15;
    const unsigned wbase = __builtin_amdgcn_readfirstlane(t & ~63u);
    const unsigned bins = 1u << (P == 1 ? g.d1 : g.d2);
    const unsigned n_units = P == 1 ? w.U1 : (1u << g.d1) * w.K;
    const unsigned T = P == 1 ? w.U1 * w.tpu : __builtin_amdgcn_readfirstlane(tp2[n_units]);
    const rsrc_t rs_res = make_rsrc(res_lo, (uint32_t)g.n * 4u);
    auto abs_base = [&](unsigned b1, unsigned b2) -> uint64_t
    {
        const unsigned f = (b1 << g.d2) | b2;
        const uint64_t h = hf[f];
        if (g.op == 1)
            return h;
        return bit ? h : (uint64_t)f * g.slice_bits - h;
    };
    if (P == 1)
        for (unsigned i = t; i < kBins; i += TT)
            sbase[i] = i < bins ? abs_base(i, 0) : 0;
    unsigned next_item = 0; // thread 0: the ticket after this one, taken while this item is worked on
    if (t == 0)
        next_item = atomicAdd(ticket, 1u);
    for (;;)
    {
        if (t == 0)
            sh_item = next_item;
        __syncthreads();
        const unsigned item = __builtin_amdgcn_readfirstlane(sh_item);
        const unsigned t_lo = item * kCkS;
        if (t_lo >= T)
            break;
        if (t == 0)
            next_item = atomicAdd(ticket, 1u);
        const unsigned t_hi = t_lo + kCkS < T ? t_lo + kCkS : T;
        // pass 2: what the partition noted about the item's tiles (one load; lane j holds tile t_lo + j)
        uint64_t dsc = 0;
        if (P == 2 && (t & 63) < kCkS && t_lo + (t & 63) < t_hi)
            dsc = tdesc[t_lo + (t & 63)];
        auto tile_of = [&](unsigned ti) -> SwTile<P>
        {
            if (P == 1)
                return sw_tile<P>(g, w, ti, 0u, in_lo, tp2);
            const unsigned j = ti - t_lo;
            const uint32_t lo = __builtin_amdgcn_readlane((unsigned)dsc, j), hi = __builtin_amdgcn_readlane((unsigned)(dsc >> 32), j);
            SwTile<P> r;
            r.lo = lo;
            r.cnt = hi & 0x3FFFu;
            r.unit = (hi >> 14) & 0x1FFFFu;
            r.tb = (hi >> 31) ? ti : 0xFFFFFFFFu;
            return r;
        };
        SwTile<P> d = tile_of(t_lo);
        unsigned have_unit = 0xFFFFFFFFu; // the unit sbase was loaded for
        unsigned nh = t < bins ? tile_hist[(uint64_t)t_lo * bins + t] : 0u;
        for (unsigned ti = t_lo; ti < t_hi; ++ti)
        {
            const bool more = ti + 1 < t_hi;
            SwTile<P> dn = d;
            if (more)
                dn = tile_of(ti + 1);
            if (d.cnt == 0)
            { // (pass 1: tiles past the batch's end)
                d = dn;
                continue;
            }
            const unsigned grp = P == 2 ? d.unit / w.K : 0u;
            if (ti == d.tb)
            {
                (void)sw_unit_setup<P, TT>(g, w, d.unit, offs1, in_lo, tp2, fstart, segsum, cursor);
            }
            else if (ti == t_lo)
            {
                for (unsigned i = t; i < kBins; i += TT)
                    cursor[i] = i < bins ? ckpt[(uint64_t)(ti / kCkS) * kBins + i] : 0u;
            }
            if (P == 2 && have_unit != d.unit)
            {
                const uint64_t gbase = abs_base(grp, 0);
                for (unsigned i = t; i < kBins; i += TT)
                    sbase[i] = i < bins ? abs_base(grp, i) - gbase : 0;
                have_unit = d.unit;
            }
            for (unsigned i = t; i < kBins; i += TT)
            {
                hist[i] = nh;
                start[i] = nh;
            }
            __syncthreads();
            if (more)
                nh = t < bins ? tile_hist[(uint64_t)(ti + 1) * bins + t] : 0u;
            uint16_t sl[PER];
            {
                const rsrc_t rs = make_rsrc(slots + d.lo, d.cnt * 2u);
#pragma unroll
                for (unsigned u = 0; u < PER; ++u)
                    sl[u] = __builtin_amdgcn_raw_buffer_load_b16(rs, (int)(t * 2u), (int)(u * TT * 2u), kAuxNT);
            }
            block_excl_scan_bins(start, wsum);
            for (unsigned i = t; i < kBins; i += TT)
                delta[i] = cursor[i] - start[i];
            // the byte map: 16 lanes per bin
            for (unsigned b = t >> 4; b < bins; b += TT / 16)
            {
                const unsigned cnt = hist[b], st = start[b];
                for (unsigned i = l; i < cnt; i += 16)
                    binof[st + i] = (uint8_t)b;
            }
            __syncthreads();
#pragma unroll
            for (unsigned u = 0; u < PER; ++u)
            {
                const unsigned p = u * TT + t;
                const unsigned src = delta[binof[p < d.cnt ? p : 0u]] + p;
                // (beyond the tile's end: an offset past the buffer's end, nothing is fetched)
                __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_res, &lo32[u * TT + wbase], 4, (int)(p < d.cnt ? src * 4u : 0xFFFFFFFCu), 0, 0, 0);
            }
            __builtin_amdgcn_s_waitcnt(0x0F70); // vmcnt(0): this wave's fetches have landed
            __syncthreads();
            bool mk = false;
            {
                typedef unsigned v2u32 __attribute__((ext_vector_type(2)));
                const rsrc_t rs = P == 2 ? make_rsrc(out_lo + d.lo, d.cnt * 4u) : make_rsrc(out + d.lo, d.cnt * 8u);
#pragma unroll
                for (unsigned u = 0; u < PER; ++u)
                {
                    const unsigned at = sl[u] & (kTile - 1); // (what lies beyond the tile's end is not written: any slot will do)
                    const uint32_t v = lo32[at];
                    const uint64_t full = sbase[binof[at]] + v;
                    if (P == 2)
                    { // an answer that does not fit 32 bits relative to its pass-1 bin is left to the fix-up pass
                        uint32_t r = (uint32_t)full;
                        if (v >= kMark)
                            r = v;
                        else if (full >= kMark)
                        {
                            r = kMark;
                            mk = true;
                        }
                        __builtin_amdgcn_raw_buffer_store_b32(r, rs, (int)(t * 4u), (int)(u * TT * 4u), kAuxNT);
                    }
                    else
                    {
                        const uint64_t a = v == kBad ? SDSL_HIP_NPOS : (v == kMark ? kMark64 : full);
                        v2u32 pr;
                        pr.x = (unsigned)a;
                        pr.y = (unsigned)(a >> 32);
                        __builtin_amdgcn_raw_buffer_store_b64(pr, rs, (int)(t * 8u), (int)(u * TT * 8u), kAuxNT);
                    }
                }
            }
            if (P == 2 && mk)
                *any_marked = 1;
            for (unsigned i = t; i < kBins; i += TT)
                cursor[i] += hist[i];
            __syncthreads();
            d = dn;
        }
    }
}

constexpr uint64_t kSwMaxPass = (UINT64_C(1) << 30) - (UINT64_C(1) << 20); // positions per pass over the batch (32-bit cursors)
constexpr unsigned kSwT = 512, kSwPer = 16, kSwTile = kSwT * kSwPer;

void sw_fill(SwGeom & w, const SrGeom & g, uint64_t cnt)
{
    const uint64_t tiles = (cnt + kSwTile - 1) / kSwTile;
    static const int u_env = getenv("SDSL_HIP_SWC_UPB") ? atoi(getenv("SDSL_HIP_SWC_UPB")) : 0;
    static const int k_env = getenv("SDSL_HIP_SWC_K") ? atoi(getenv("SDSL_HIP_SWC_K")) : 0;
    // streams should be long (a stream pays two partial chunks, at its ends) and units many (they are what the blocks share out)
    unsigned upb = (unsigned)(tiles / ((uint64_t)kHB * 64));
    upb = upb < 1 ? 1 : (upb > 8 ? 8 : upb);
    if (u_env >= 1 && u_env <= 8)
        upb = (unsigned)u_env;
    w.U1 = kHB * upb;
    w.tpu = (uint32_t)((tiles + w.U1 - 1) / w.U1);
    if (w.tpu == 0)
        w.tpu = 1;
    const uint64_t per_group = tiles >> g.d1; // (at least: the last bin of pass 1 is usually partly used)
    unsigned K = 1;
    while (K < 32 && per_group / (2 * K) >= 24)
        K *= 2;
    if (k_env >= 1 && k_env <= (int)kMaxK && (kHB % (unsigned)k_env) == 0)
        K = (unsigned)k_env;
    w.K = K;
    w.nf = 1u << (g.d1 + g.d2);
    w.nfw = (w.nf + 63) & ~63u;
    w.fine_words = (w.nf + 1) / 2;
}

} // namespace

size_t bv_swc_scratch_bytes(const BvView & v, uint64_t n)
{
    // (sized for the widest geometry a vector of this size can get: 2^16 slices, four units per histogram block, 16 segments)
    (void)v;
    SwGeom w{};
    w.U1 = kHB * 8;
    w.K = 32;
    w.nf = 1u << 16;
    w.nfw = w.nf;
    SwBuf b;
    return sw_carve(b, nullptr, n < kSwMaxPass ? n : kSwMaxPass, kSwTile, w, kBins) + 4096;
}

sdsl_hip_status sw_run(const BvView & v, int op, int bit, const SelectPlan & sp, const uint64_t * d_idx, uint64_t n, uint64_t * d_out,
                       hipStream_t s, void * scratch, size_t scratch_bytes, const uint32_t * go)
{
    SwCallbacks cb;
    cb.fill = [&](SrGeom & g, uint64_t cnt) { sr_fill_geom(g, v, op, sp, cnt); };
    cb.answers = [&](const SrGeom & g, unsigned nf, const uint32_t * fstart, const uint32_t * ioff, uint32_t * keys2, uint64_t * hf,
                     uint32_t * marked, hipStream_t st) { return sr_launch_answers(v, op, bit, sp, nf, g.d2, g.slog, fstart, ioff, keys2, hf, marked, g.go, st); };
    if (op == 1)
        cb.fixup = [&](const uint32_t * marked, const uint64_t * idx, uint64_t * out, uint64_t cnt, hipStream_t st)
        { sr_launch_select_fixup(v, bit, marked, idx, out, cnt, go, st); };
    return sw_run_with(cb, bit, d_idx, n, d_out, s, scratch, scratch_bytes, go);
}

sdsl_hip_status sw_run_with(const SwCallbacks & cb, int bit, const uint64_t * d_idx, uint64_t n, uint64_t * d_out, hipStream_t s,
                            void * scratch, size_t scratch_bytes, const uint32_t * go)
{
    static const bool trace_env = getenv("SDSL_HIP_TRACE_SORTED") != nullptr;
    const bool trace_opt = g_trace_phases.load() != 0;
    const bool trace = trace_env || trace_opt;
    static const int pb_env = getenv("SDSL_HIP_SWC_PART_BLOCKS") ? atoi(getenv("SDSL_HIP_SWC_PART_BLOCKS")) : 0;
    static const int ub_env = getenv("SDSL_HIP_SWC_UNP_BLOCKS") ? atoi(getenv("SDSL_HIP_SWC_UNP_BLOCKS")) : 0;
    // the slice histogram may fill the CU's LDS (a per-device attribute of the kernel; setting it is a host-side table write)
    {
        hipError_t e = hipSuccess;
        auto set = [&](const void * f) { const hipError_t r = hipFuncSetAttribute(f, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024); e = e == hipSuccess ? r : e; };
        set((const void *)k_sw_hist<8, 0>);
        set((const void *)k_sw_hist<8, 1>);
        set((const void *)k_sw_hist<8, 2>);
        set((const void *)k_sw_hist<8, 3>);
        set((const void *)k_sw_hist<8, 4>);
        SH_HIP(e);
    }
    for (uint64_t done = 0; done < n;)
    {
        const uint64_t cnt = n - done < kSwMaxPass ? n - done : kSwMaxPass;
        SrGeom g;
        cb.fill(g, cnt);
        const int op = (int)g.op;
        g.tile = kSwTile;
        g.tiles1 = (uint32_t)((cnt + g.tile - 1) / g.tile);
        g.G = 0;
        g.go = go;
        SwGeom w;
        sw_fill(w, g, cnt);
        const unsigned bins1 = 1u << g.d1;
        SwBuf b;
        if (sw_carve(b, scratch, cnt, g.tile, w, bins1) > scratch_bytes)
        {
            set_error("bucketed batch: scratch too small");
            return SDSL_HIP_ERR_INVALID;
        }
        const uint64_t * idx = d_idx + done;
        PhaseTimer pt(trace, s);
        pt.mark();
        SH_TRY(fill_u32_async(b.fine_h, 0u, (size_t)kHB * w.nfw * 4, s));
        SH_TRY(fill_u32_async(b.tickets, 0u, 16, s));
        // (eight positions per thread and chunk: with sixteen, the positions, the prefetched ones and their keys did not fit 128
        // VGPRs — 6 spilled, 1.52 against 1.45 ms)
        sr_dispatch_op(g, [&](auto op)
                       { hipLaunchKernelGGL((k_sw_hist<8, decltype(op)::value>), dim3(kHB), dim3(kHT), (w.fine_words + kBins + 2) * 4, s, g, w, idx, b.counts1, b.fine_h); });
        pt.mark("hist");
        sr_launch_bin_offsets(bins1, w.U1, b.counts1, b.btot, b.bstart1, b.offs1, s);
        hipLaunchKernelGGL(k_sw_seg_reduce, dim3((w.nf + 255) / 256), dim3(256), 0, s, w, b.fine_h, b.segsum, b.tot);
        sr_launch_fine_scan(w.nf, b.tot, b.fstart, b.ioff, s);
        pt.mark("tables");
        static const bool dbg = getenv("SDSL_HIP_SWC_DEBUG") != nullptr;
        if (dbg)
        { // the slice counts must add up to the batch
            std::vector<uint32_t> ht(w.nf), hrow((size_t)kHB * w.nfw);
            SH_HIP(hipStreamSynchronize(s));
            SH_HIP(hipMemcpy(ht.data(), b.tot, (size_t)w.nf * 4, hipMemcpyDeviceToHost));
            SH_HIP(hipMemcpy(hrow.data(), b.fine_h, hrow.size() * 4, hipMemcpyDeviceToHost));
            uint64_t sum = 0;
            for (uint32_t x : ht)
                sum += x;
            fprintf(stderr, "[swc debug] n=%llu sum(tot)=%llu diff=%lld U1=%u tpu=%u K=%u nf=%u\n", (unsigned long long)cnt, (unsigned long long)sum,
                    (long long)cnt - (long long)sum, w.U1, w.tpu, w.K, w.nf);
            for (unsigned h = 0; h < kHB; ++h)
            {
                uint64_t rs = 0;
                for (unsigned f = 0; f < w.nf; ++f)
                    rs += hrow[(size_t)h * w.nfw + f];
                const uint64_t lo = std::min<uint64_t>((uint64_t)h * (w.U1 / kHB) * w.tpu * g.tile, cnt), hi = std::min<uint64_t>(lo + (uint64_t)(w.U1 / kHB) * w.tpu * g.tile, cnt);
                if (rs != hi - lo)
                    fprintf(stderr, "[swc debug] row %u: %llu counted, %llu keys\n", h, (unsigned long long)rs, (unsigned long long)(hi - lo));
            }
        }
        const unsigned pblocks = pb_env >= 1 ? (unsigned)pb_env : 512u;
        // pass 1 holds 64-bit positions: 1024 threads x 8 keep a tile's keys and the next tile's within the register file (512 x 16
        // spilled 33 VGPRs: 4.2 against 3.1 ms); one block per CU
        sr_dispatch_op(g, [&](auto op)
                       {
                           hipLaunchKernelGGL((k_sw_partition<1, 1024, 8, decltype(op)::value>), dim3(pb_env >= 1 ? (unsigned)pb_env : 256u), dim3(1024), 0, s,
                                              g, w, idx, (const uint32_t *)nullptr, b.offs1, (const uint32_t *)nullptr, (const uint32_t *)nullptr,
                                              (const uint32_t *)nullptr, (const uint32_t *)nullptr, b.tickets + 0, b.keys1, b.slots1, b.thist1, b.ck1,
                                              (uint64_t *)nullptr);
                       });
        pt.mark("part1");
        hipLaunchKernelGGL(k_sw_units2, dim3(1), dim3(1024), 0, s, g, w, b.offs1, b.in_lo, b.tp2);
        hipLaunchKernelGGL((k_sw_partition<2, kSwT, kSwPer, 0>), dim3(pblocks), dim3(kSwT), 0, s, g, w, (const uint64_t *)nullptr, b.keys1, b.offs1,
                           b.in_lo, b.tp2, b.fstart, b.segsum, b.tickets + 1, b.keys2, b.slots2, b.thist2, b.ck2, b.tdesc2);
        pt.mark("part2");
        SH_TRY(cb.answers(g, w.nf, b.fstart, b.ioff, b.keys2, b.hf, b.marked, s));
        pt.mark("answer");
        // two blocks per CU on the way back (three fit): what a block gathers per tile then survives in its XCD's L2 until the next
        // tile continues the same streams — 7.0 + 8.5 GB read instead of 7.8 + 9.9 at the same 4.7 ms (profiles/swc_traffic_r03.txt)
        const unsigned ublocks = ub_env >= 1 ? (unsigned)ub_env : 512u;
        hipLaunchKernelGGL((k_sw_unpermute_dma<2, 512, 16>), dim3(ublocks), dim3(512), 0, s, b.hf, bit, g, w, b.offs1, b.in_lo, b.tp2, b.fstart,
                           b.segsum, b.ck2, b.tickets + 2, b.tdesc2, b.keys2, b.marked, b.slots2, b.thist2, b.keys1, (uint64_t *)nullptr);
        pt.mark("unperm2");
        hipLaunchKernelGGL((k_sw_unpermute_dma<1, 512, 16>), dim3(ublocks), dim3(512), 0, s, b.hf, bit, g, w, b.offs1, b.in_lo, b.tp2, b.fstart,
                           b.segsum, b.ck1, b.tickets + 3, (const uint64_t *)nullptr, b.keys1, b.marked, b.slots1, b.thist1, (uint32_t *)nullptr,
                           d_out + done);
        pt.mark("unperm1");
        if (dbg && cb.fixup)
        { // how much is left to the fix-up pass
            uint32_t mk = 0;
            SH_HIP(hipStreamSynchronize(s));
            SH_HIP(hipMemcpy(&mk, b.marked, 4, hipMemcpyDeviceToHost));
            std::vector<uint64_t> ho(std::min<uint64_t>(cnt, 1 << 24));
            SH_HIP(hipMemcpy(ho.data(), d_out + done, ho.size() * 8, hipMemcpyDeviceToHost));
            uint64_t m = 0;
            for (uint64_t x : ho)
                m += x == kMark64;
            fprintf(stderr, "[swc debug] any_marked=%u, %llu of the first %zu answers are left to the fix-up\n", mk, (unsigned long long)m, ho.size());
        }
        if (cb.fixup)
            cb.fixup(b.marked, idx, d_out + done, cnt, s);
        SH_HIP(hipGetLastError());
        if (trace_env)
            pt.report(g, cb.what);
        if (trace_opt)
        {
            pt.keep(op);
            if (go)
            { // (tracing only: the passes were skipped when the sample's verdict sent the batch to the direct kernel)
                uint32_t verdict = 1;
                SH_HIP(hipMemcpyAsync(&verdict, go, 4, hipMemcpyDeviceToHost, s));
                SH_HIP(hipStreamSynchronize(s));
                if (!verdict)
                    bv_sorted_set_phases("");
            }
        }
        done += cnt;
    }
    return SDSL_HIP_OK;
}

} // namespace sdslhip
