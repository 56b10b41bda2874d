// bv_sorted.hip — batched rank on a plain bit vector for LARGE batches: the batch is radix-partitioned by the
// 64 KiB slice of the index each position addresses, every slice is staged ONCE into LDS and answers all its
// positions from there, and the answers travel back through the partition in reverse.
//
// Why: a random rank costs one L2 miss = one 128-byte fabric read, and the part serves ≈ 40 G of
// those per second whatever the kernel does (profiles/gather_probe_r01.txt).  A batch of 10^9 positions over 2^34 bits
// addresses every 64-byte rank line ≈ 26 times.  The first cut of this file (profiles/sorted_rank_v1_single_level_r02.txt)
// partitioned once into 1 MiB buckets and gathered from the XCD's L2: the gather kernel then ran at 82 G/s (139 G/s
// with everything cache-resident) — a quad fetching a line costs a 128-byte L2->L1 transfer wherever the line lives.
// Only LDS serves random 64-byte reads an order of magnitude faster, so the buckets must be LDS-sized: two partition
// passes of <= 8 bits each (least significant digit first), both with runs of 32 keys per (tile, bin).
// The reference answers its queries one at a time (rank_support_v5.hpp:131-149); batching is this library's addition
// and the answers are the same numbers.
//
// line index of a position = [ digit 2 | digit 1 | 10 bits inside the slice ]      (slice = 2^10 lines = 64 KiB)
//
//   hist 1 / partition 1   by digit 1: 64-bit positions -> 32-bit keys (digit 2, line in slice, bit in line), u16 slots
//   hist 2 / partition 2   by digit 2, tiles never straddle a digit-1 group, so the result is ordered by
//                          (digit 2, digit 1) = by slice; keys shrink to (line in slice, bit in line)
//   k_sr_rank_lds          per slice: 64 KiB -> LDS, one LANE per position (8 words from LDS, masked popcounts),
//                          32-bit answers relative to the slice, in place
//   un-permute 2, 1        per tile the runs are gathered back into LDS, made absolute, and every position picks its
//                          answer by slot; all global accesses are coalesced or runs
// A partition pass: per tile of 8192 keys a counting sort in LDS (unstable inside a (tile, bin) pair — the slot array
// makes the way back exact), the sorted tile is written out bin by bin (16 lanes per bin).  No pass scatters single
// words: the un-permute that sank the idea in round 1 (random 8-byte scatter, 22.9 G/s,
// profiles/scatter_probe_r01.txt) is a gather of runs staged through LDS.
#include <mutex>
#include <string>
#include <type_traits>

#include "bv_sorted_dev.hpp"

namespace sdslhip {

namespace {


// Tiles of a pass.  Pass 1: tile i = keys [i * tile, ...).  Pass 2: the keys are grouped by digit 1 (group starts
// gs[], tp[] = exclusive prefix of ceil(size / tile)); a tile lies inside one group.
struct TileMap
{
    unsigned tp[kBins + 1], gs[kBins + 1];
};
template <int P, unsigned TT>
__device__ __forceinline__ void sr_load_map(const SrGeom & g, TileMap & m, const uint32_t * tprefix, const uint32_t * gstart)
{
    if (P == 2)
        for (unsigned i = threadIdx.x; i <= (1u << g.d1); i += TT)
        {
            m.tp[i] = tprefix[i];
            m.gs[i] = gstart[i];
        }
}
template <int P>
__device__ __forceinline__ unsigned sr_tiles(const SrGeom & g, const TileMap & m)
{
    return P == 1 ? g.tiles1 : m.tp[1u << g.d1];
}
template <int P>
__device__ __forceinline__ void sr_tile_range(const SrGeom & g, const TileMap & m, unsigned ti, uint64_t & lo, uint64_t & hi,
                                              unsigned & grp)
{
    const unsigned kTile = g.tile;
    if (P == 1)
    {
        grp = 0;
        lo = (uint64_t)ti * kTile;
        hi = lo + kTile < g.n ? lo + kTile : g.n;
        return;
    }
    unsigned a = 0, z = 1u << g.d1; // last group a with tp[a] <= ti (groups without tiles are skipped over)
    while (a + 1 < z)
    {
        const unsigned mid = (a + z) >> 1;
        if (m.tp[mid] <= ti)
            a = mid;
        else
            z = mid;
    }
    grp = a;
    lo = (uint64_t)m.gs[a] + (uint64_t)(ti - m.tp[a]) * kTile;
    hi = lo + kTile < m.gs[a + 1] ? lo + kTile : m.gs[a + 1];
}

// digits of the PER keys of this thread in tile [lo, hi); all loads are issued before the first is used
template <int P, unsigned TT, unsigned PER, bool FULL>
__device__ __forceinline__ void sr_load_keys(const SrGeom & g, const uint64_t * __restrict__ idx,
                                             const uint32_t * __restrict__ keys_in, uint64_t lo, uint64_t hi, unsigned (&dig)[PER],
                                             uint32_t (&key)[PER])
{
    const unsigned t = threadIdx.x;
    const unsigned cnt = (unsigned)(hi - lo); // offsets inside a tile are 32-bit (uniform base + lane offset addressing)
    idx += lo;
    keys_in += lo;
    if (P == 1)
    {
        uint64_t p[PER];
#pragma unroll
        for (unsigned u = 0; u < PER; ++u)
        {
            const unsigned q = u * TT + t;
            p[u] = FULL || q < cnt ? __builtin_nontemporal_load(idx + q) : 0;
        }
#pragma unroll
        for (unsigned u = 0; u < PER; ++u)
            sr_key1(p[u], g, dig[u], key[u]);
    }
    else
    {
        uint32_t k1[PER];
#pragma unroll
        for (unsigned u = 0; u < PER; ++u)
        {
            const unsigned q = u * TT + t;
            k1[u] = FULL || q < cnt ? __builtin_nontemporal_load(keys_in + q) : 0;
        }
#pragma unroll
        for (unsigned u = 0; u < PER; ++u)
            sr_key2(k1[u], g, dig[u], key[u]);
    }
}

// ---- histogram of a pass: counts[bin][block]; pass 2 also counts per slice ------------------------------------------
template <int P, unsigned TT, unsigned PER>
__global__ __launch_bounds__(TT) void k_sr_hist(SrGeom g, const uint64_t * __restrict__ idx, const uint32_t * __restrict__ keys1,
                                                const uint32_t * __restrict__ tprefix, const uint32_t * __restrict__ gstart,
                                                uint32_t * __restrict__ counts, uint32_t * __restrict__ fine_count)
{
    __shared__ unsigned hist[kBins], ghist[TT / 64][kBins]; // ghist: one per wave (fewer collisions)
    __shared__ unsigned ghist2[P == 1 ? TT / 64 : 1][kBins]; // pass 1 only: the digit-2 histogram of the whole batch (fine_count = its totals)
    __shared__ TileMap map;
    const unsigned t = threadIdx.x, wv = t >> 6;
    const bool both = P == 1 && fine_count != nullptr;
    sr_load_map<P, TT>(g, map, tprefix, gstart);
    for (unsigned i = t; i < kBins; i += TT)
    {
        hist[i] = 0;
        for (unsigned w = 0; w < TT / 64; ++w)
            ghist[w][i] = 0;
        if (P == 1)
            for (unsigned w = 0; w < TT / 64; ++w)
                ghist2[w][i] = 0;
    }
    __syncthreads();
    const unsigned nt = sr_tiles<P>(g, map);
    const unsigned tlo = (unsigned)((uint64_t)nt * blockIdx.x / g.G), thi = (unsigned)((uint64_t)nt * (blockIdx.x + 1) / g.G);
    unsigned cur = 0;
    auto flush = [&](unsigned grp)
    { // all threads; ghist -> hist (+ the per-slice counts of pass 2)
        __syncthreads();
        for (unsigned i = t; i < kBins; i += TT)
        {
            unsigned c = 0;
            for (unsigned w = 0; w < TT / 64; ++w)
            {
                c += ghist[w][i];
                ghist[w][i] = 0;
            }
            if (c)
            {
                hist[i] += c;
                if (P == 2)
                    atomicAdd(&fine_count[((uint64_t)i << g.d1) | grp], c);
            }
        }
        __syncthreads();
    };
    for (unsigned ti = tlo; ti < thi; ++ti)
    {
        uint64_t lo, hi;
        unsigned grp;
        sr_tile_range<P>(g, map, ti, lo, hi, grp);
        if (P == 2 && grp != cur && ti != tlo)
            flush(cur);
        cur = grp;
        // (a full tile — all but the last of a group — takes the branch-free body)
        auto count_tile = [&](auto full_c)
        {
            constexpr bool FULL = decltype(full_c)::value;
            unsigned dig[PER];
            uint32_t key[PER];
            sr_load_keys<P, TT, PER, FULL>(g, idx, keys1, lo, hi, dig, key);
#pragma unroll
            for (unsigned u = 0; u < PER; ++u)
                if (FULL || u * TT + t < (unsigned)(hi - lo))
                    atomicAdd(&ghist[wv][dig[u]], 1u);
            if (P == 1 && both)
            {
#pragma unroll
                for (unsigned u = 0; u < PER; ++u)
                    if (FULL || u * TT + t < (unsigned)(hi - lo))
                    {
                        unsigned d2;
                        uint32_t k2;
                        sr_key2(key[u], g, d2, k2);
                        atomicAdd(&ghist2[wv][d2], 1u);
                    }
            }
        };
        if (hi - lo == TT * PER)
            count_tile(std::true_type{});
        else
            count_tile(std::false_type{});
    }
    flush(cur);
    const unsigned bins = 1u << (P == 1 ? g.d1 : g.d2);
    for (unsigned i = t; i < bins; i += TT)
        counts[(uint64_t)i * g.G + blockIdx.x] = hist[i];
    if (P == 1 && both)
        for (unsigned i = t; i < (1u << g.d2); i += TT)
        {
            unsigned c = 0;
            for (unsigned w = 0; w < TT / 64; ++w)
                c += ghist2[w][i];
            if (c)
                atomicAdd(&fine_count[i], c);
        }
}

// ---- offsets: offs[b][g] = keys of bins < b + keys of bin b in blocks < g -------------------------------------------
__global__ __launch_bounds__(256) void k_sr_bucket_totals(unsigned G, const uint32_t * __restrict__ counts,
                                                          uint32_t * __restrict__ btot)
{
    __shared__ unsigned red[4];
    const unsigned b = blockIdx.x;
    unsigned s = 0;
    for (unsigned i = threadIdx.x; i < G; i += 256)
        s += counts[(uint64_t)b * G + i];
    for (int d = 32; d; d >>= 1)
        s += __shfl_xor(s, d, 64);
    if ((threadIdx.x & 63) == 0)
        red[threadIdx.x >> 6] = s;
    __syncthreads();
    if (threadIdx.x == 0)
        btot[b] = red[0] + red[1] + red[2] + red[3];
}

// bstart = exclusive scan of btot (bins + 1 entries); with tprefix: also the tile prefix of the groups
__global__ __launch_bounds__(kBins) void k_sr_bucket_scan(unsigned bins, unsigned tile, const uint32_t * __restrict__ btot,
                                                          uint32_t * __restrict__ bstart, uint32_t * __restrict__ tprefix)
{
    __shared__ unsigned wsum[kBins / 64];
    const unsigned t = threadIdx.x;
    auto scan = [&](unsigned v) -> unsigned
    { // exclusive
        const unsigned inc = wave_incl_scan(v);
        __syncthreads();
        if ((t & 63) == 63)
            wsum[t >> 6] = inc;
        __syncthreads();
        unsigned base = 0;
        for (unsigned w = 0; w < (t >> 6); ++w)
            base += wsum[w];
        return base + inc - v;
    };
    const unsigned v = t < bins ? btot[t] : 0;
    const unsigned e = scan(v);
    if (t < bins)
        bstart[t] = e;
    if (t == bins - 1)
        bstart[bins] = e + v;
    if (tprefix)
    {
        const unsigned nt = (v + tile - 1) / tile;
        const unsigned te = scan(nt);
        if (t < bins)
            tprefix[t] = te;
        if (t == bins - 1)
            tprefix[bins] = te + nt;
    }
}

__global__ __launch_bounds__(256) void k_sr_bucket_offsets(unsigned G, const uint32_t * __restrict__ counts,
                                                           const uint32_t * __restrict__ bstart,
                                                           uint32_t * __restrict__ offs)
{ // G <= 256 * 8: a thread owns 8 consecutive blocks
    __shared__ unsigned wtot[4];
    const unsigned b = blockIdx.x, t = threadIdx.x;
    unsigned c[8], s = 0;
#pragma unroll
    for (int u = 0; u < 8; ++u)
    {
        const unsigned gi = 8 * t + u;
        c[u] = gi < G ? counts[(uint64_t)b * G + gi] : 0;
        s += c[u];
    }
    const unsigned inc = wave_incl_scan(s);
    if ((t & 63) == 63)
        wtot[t >> 6] = inc;
    __syncthreads();
    unsigned base = bstart[b] + inc - s;
    for (unsigned w = 0; w < (t >> 6); ++w)
        base += wtot[w];
#pragma unroll
    for (int u = 0; u < 8; ++u)
    {
        const unsigned gi = 8 * t + u;
        if (gi < G)
            offs[(uint64_t)b * G + gi] = base;
        base += c[u];
    }
}

// per slice: first key (fstart) and first work item (ioff; an item = at most kItemKeys keys of one slice)
__global__ __launch_bounds__(1024) void k_sr_fine_scan(unsigned nf, const uint32_t * __restrict__ fine_count,
                                                       uint32_t * __restrict__ fstart, uint32_t * __restrict__ ioff)
{
    __shared__ unsigned wk[17], wi[17];
    const unsigned t = threadIdx.x;
    const unsigned per = (nf + 1023) / 1024;
    const unsigned lo = t * per, hi = lo + per < nf ? lo + per : nf;
    unsigned sk = 0, si = 0;
    for (unsigned f = lo; f < hi; ++f)
    {
        const unsigned c = fine_count[f];
        sk += c;
        si += (c + kItemKeys - 1) / kItemKeys;
    }
    const unsigned ik = wave_incl_scan(sk), ii = wave_incl_scan(si);
    if ((t & 63) == 63)
    {
        wk[t >> 6] = ik;
        wi[t >> 6] = ii;
    }
    __syncthreads();
    unsigned bk = ik - sk, bi = ii - si;
    for (unsigned w = 0; w < (t >> 6); ++w)
    {
        bk += wk[w];
        bi += wi[w];
    }
    for (unsigned f = lo; f < hi; ++f)
    {
        const unsigned c = fine_count[f];
        fstart[f] = bk;
        ioff[f] = bi;
        bk += c;
        bi += (c + kItemKeys - 1) / kItemKeys;
    }
    if (t == 1023)
    { // threads past the end have empty ranges, so the last thread holds the totals
        fstart[nf] = bk;
        ioff[nf] = bi;
    }
}

// ones in front of every slice (what makes a slice-relative answer absolute)
__global__ __launch_bounds__(256) void k_sr_slice_bases(BvView bv, unsigned nf, unsigned d1, unsigned d2, uint64_t * __restrict__ hf,
                                                        unsigned slog = kSliceLog)
{
    const unsigned f = blockIdx.x * 256 + threadIdx.x;
    if (f < nf)
    {
        const uint64_t L0 = (uint64_t)sr_slice_of(f, d1, d2) << slog;
        hf[f] = L0 < bv.n_lines ? bv.lines[L0 * kLW] : 0;
    }
}

// Runs of a sorted tile <-> the bin-major array: kLanes lanes per bin, the first element of every run of kUn bins is
// fetched before any is consumed (the loop-carried LDS traffic otherwise serialises the global accesses).
template <unsigned TT>
struct RunShape
{
    static constexpr unsigned kLanes = 16;
    static constexpr unsigned kBinsPerIter = TT / kLanes;
};

// ---- a partition pass -------------------------------------------------------------------------------------------
template <int P, unsigned TT, unsigned PER>
__global__ __launch_bounds__(TT, PER >= 16 ? 4 : 8) void k_sr_partition(SrGeom g, const uint64_t * __restrict__ idx,
                                                     const uint32_t * __restrict__ keys_in,
                                                     const uint32_t * __restrict__ tprefix, const uint32_t * __restrict__ gstart,
                                                     const uint32_t * __restrict__ offs, uint32_t * __restrict__ keys_out,
                                                     uint16_t * __restrict__ slots, uint16_t * __restrict__ tile_hist)
{
    constexpr unsigned kTile = TT * PER;
    __shared__ uint32_t sorted[kTile];
    __shared__ unsigned hist[kBins], start[kBins], cursor[kBins];
    __shared__ unsigned wsum[kBins / 64];
    __shared__ unsigned big[kTile / (kBigRun + 1) + 1], n_big;
    __shared__ TileMap map;
    const unsigned t = threadIdx.x;
    const unsigned bins = 1u << (P == 1 ? g.d1 : g.d2);
    sr_load_map<P, TT>(g, map, tprefix, gstart);
    for (unsigned i = t; i < kBins; i += TT)
        cursor[i] = i < bins ? offs[(uint64_t)i * g.G + blockIdx.x] : 0;
    __syncthreads();
    const unsigned nt = sr_tiles<P>(g, map);
    const unsigned tlo = (unsigned)((uint64_t)nt * blockIdx.x / g.G), thi = (unsigned)((uint64_t)nt * (blockIdx.x + 1) / g.G);
    for (unsigned ti = tlo; ti < thi; ++ti)
    {
        uint64_t lo, hi;
        unsigned grp;
        sr_tile_range<P>(g, map, ti, lo, hi, grp);
        for (unsigned i = t; i < kBins; i += TT)
            hist[i] = 0;
        if (t == 0)
            n_big = 0;
        __syncthreads();
        // load, count, place: a full tile — all but the last of a group — takes the body without per-key bounds checks
        auto sort_tile = [&](auto full_c)
        {
            constexpr bool FULL = decltype(full_c)::value;
            uint32_t key[PER];
            unsigned br[PER]; // bin << 16 | rank inside the tile's share of the bin
            sr_load_keys<P, TT, PER, FULL>(g, idx, keys_in, lo, hi, br, key);
            const unsigned cnt_t = (unsigned)(hi - lo);
            uint16_t * slots_t = slots + lo;
#pragma unroll
            for (unsigned u = 0; u < PER; ++u)
            {
                const unsigned q = u * TT + t;
                const unsigned d = br[u];
                br[u] = d << 16;
                if (FULL || q < cnt_t)
                    br[u] |= atomicAdd(&hist[d], 1u); // < 2^14
            }
            __syncthreads();
            for (unsigned i = t; i < kBins; i += TT)
                start[i] = hist[i];
            __syncthreads();
            block_excl_scan_bins(start, wsum);
            for (unsigned i = t; i < bins; i += TT)
                tile_hist[(uint64_t)ti * bins + i] = (uint16_t)hist[i];
#pragma unroll
            for (unsigned u = 0; u < PER; ++u)
            {
                const unsigned q = u * TT + t;
                if (FULL || q < cnt_t)
                {
                    const unsigned pos = start[br[u] >> 16] + (br[u] & 0xFFFFu);
                    sorted[pos] = key[u];
                    __builtin_nontemporal_store((uint16_t)pos, slots_t + q);
                }
            }
        };
        if (hi - lo == kTile)
            sort_tile(std::true_type{});
        else
            sort_tile(std::false_type{});
        __syncthreads();
        { // runs out: 16 lanes per bin
            const unsigned l = t & 15;
            for (unsigned b = t >> 4; b < bins; b += TT / 16)
            {
                const unsigned cnt = hist[b];
                if (cnt == 0)
                    continue;
                const unsigned st = start[b], cur = cursor[b];
                if (cnt > kBigRun)
                {
                    if (l == 0)
                        big[atomicAdd(&n_big, 1u)] = b;
                    continue;
                }
                uint32_t * dst = keys_out + cur;
                for (unsigned i = l; i < cnt; i += 16)
                    dst[i] = sorted[st + i];
            }
        }
        __syncthreads();
        const unsigned nb = n_big;
        for (unsigned k = 0; k < nb; ++k)
        {
            const unsigned b = big[k], cnt = hist[b], st = start[b], cur = cursor[b];
            for (unsigned i = t; i < cnt; i += TT)
                keys_out[(uint64_t)cur + i] = sorted[st + i];
        }
        __syncthreads();
        for (unsigned i = t; i < kBins; i += TT)
            cursor[i] += hist[i];
        __syncthreads();
    }
}

// ---- pass 2 in ONE sweep ------------------------------------------------------------------------------------------------
// The second partition without a histogram pass of its own (1.06 of the step's 18 ms): the digit-2 totals come out of the
// first histogram pass (k_sr_hist<1>), and where a tile's run of bin b starts is the bin's base + the keys of bin b in all
// tiles in front of it — a decoupled look-back over per-(tile, bin) status words [flag:2 | count:30]: a tile publishes its own
// count (AGGREGATE) as soon as it has it, walks back over its predecessors until it meets an INCLUSIVE prefix, and publishes its
// own inclusive prefix.  Tiles are handed out by a ticket counter, so every tile a block waits for belongs to a block that
// is already running: no deadlock whatever the residency.  Flag and count travel in one 32-bit word (relaxed device-scope
// atomics suffice).  The slice starts fall out of the same prefix (a slice = bin x digit-1 group: the bin's position at the
// group's first tile), and so do the per-(bin, block) offsets the way back (k_sr_unpermute<2>, static tile ranges) starts from.
constexpr uint32_t kStAgg = 1u << 30, kStIncl = 2u << 30, kStMask = (1u << 30) - 1;

template <unsigned TT, unsigned PER, unsigned WPE>
__global__ __launch_bounds__(TT, WPE) void k_sr_partition2_sweep(SrGeom g, const uint32_t * __restrict__ keys_in,
                                                     const uint32_t * __restrict__ tprefix, const uint32_t * __restrict__ gstart,
                                                     const uint32_t * __restrict__ bstart2, uint32_t * __restrict__ status,
                                                     uint32_t * __restrict__ ticket, uint32_t * __restrict__ keys_out,
                                                     uint16_t * __restrict__ slots, uint16_t * __restrict__ tile_hist,
                                                     uint32_t * __restrict__ offs_out, uint32_t * __restrict__ fstart)
{
    constexpr unsigned kTile = TT * PER;
    __shared__ uint32_t sorted[kTile];
    __shared__ unsigned hist[kBins], start[kBins], cursor[kBins];
    __shared__ unsigned wsum[kBins / 64];
    __shared__ unsigned big[kTile / (kBigRun + 1) + 1], n_big, sh_ti;
    __shared__ TileMap map;
    const unsigned t = threadIdx.x;
    const unsigned bins = 1u << g.d2;
    sr_load_map<2, TT>(g, map, tprefix, gstart);
    __syncthreads();
    const unsigned nt = sr_tiles<2>(g, map);
    unsigned next_ti = 0; // thread 0: the ticket of the tile after this one, taken while this one is written out
    if (t == 0)
        next_ti = atomicAdd(ticket, 1u);
    for (;;)
    {
        if (t == 0)
        {
            sh_ti = next_ti;
            n_big = 0;
        }
        for (unsigned i = t; i < kBins; i += TT)
            hist[i] = 0;
        __syncthreads();
        const unsigned ti = sh_ti;
        if (ti >= nt)
            break;
        uint64_t lo, hi;
        unsigned grp;
        sr_tile_range<2>(g, map, ti, lo, hi, grp);
        uint32_t * st_mine = status + (uint64_t)ti * kBins;
        // the words of the kLook tiles in front of this one are fetched together, right after this tile's own counts went out.
        // Two is the measured optimum: windows of 4 / 8 / 16 cost 3.67 / 3.90 / 4.31 ms against 3.48 (the polling traffic and
        // the registers they hold outweigh the round trips they save: the inclusive prefix is rarely more than two tiles away)
        constexpr unsigned kLook = 2;
        uint32_t w_prev[kLook] = {};
        auto sort_tile = [&](auto full_c)
        {
            constexpr bool FULL = decltype(full_c)::value;
            uint32_t key[PER];
            unsigned br[PER]; // bin << 16 | rank inside the tile's share of the bin
            sr_load_keys<2, TT, PER, FULL>(g, nullptr, keys_in, lo, hi, br, key);
            const unsigned cnt_t = (unsigned)(hi - lo);
            uint16_t * slots_t = slots + lo;
#pragma unroll
            for (unsigned u = 0; u < PER; ++u)
            {
                const unsigned q = u * TT + t;
                const unsigned d = br[u];
                br[u] = d << 16;
                if (FULL || q < cnt_t)
                    br[u] |= atomicAdd(&hist[d], 1u); // < 2^14
            }
            __syncthreads();
            if (t < bins)
            { // the earlier the successors see this tile's counts, the shorter their wait; and the predecessor's word is asked
              // for now, to be looked at after the keys have been placed
                __hip_atomic_store(st_mine + t, (ti == 0 ? kStIncl : kStAgg) | hist[t], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
#pragma unroll
                for (unsigned k = 0; k < kLook; ++k)
                    w_prev[k] = k < ti ? __hip_atomic_load(st_mine - (uint64_t)(k + 1) * kBins + t, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : 0u;
            }
            for (unsigned i = t; i < kBins; i += TT)
                start[i] = hist[i];
            __syncthreads();
            block_excl_scan_bins(start, wsum);
            for (unsigned i = t; i < bins; i += TT)
                tile_hist[(uint64_t)ti * bins + i] = (uint16_t)hist[i];
#pragma unroll
            for (unsigned u = 0; u < PER; ++u)
            {
                const unsigned q = u * TT + t;
                if (FULL || q < cnt_t)
                {
                    const unsigned pos = start[br[u] >> 16] + (br[u] & 0xFFFFu);
                    sorted[pos] = key[u];
                    __builtin_nontemporal_store((uint16_t)pos, slots_t + q);
                }
            }
        };
        if (hi - lo == kTile)
            sort_tile(std::true_type{});
        else
            sort_tile(std::false_type{});
        if (t < bins)
        { // look back: keys of bin t in the tiles in front of this one
            unsigned sum = 0;
            if (ti != 0)
            {
                bool done = false;
                for (unsigned base = ti; !done && base > 0;)
                { // a window of up to kLook predecessors: base-1 .. base-kLook
                    const unsigned wn = base < kLook ? base : kLook;
                    if (base != ti)
                    {
#pragma unroll
                        for (unsigned k = 0; k < kLook; ++k)
                            w_prev[k] = k < wn ? __hip_atomic_load(status + (uint64_t)(base - 1 - k) * kBins + t, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : 0u;
                    }
#pragma unroll
                    for (unsigned k = 0; k < kLook; ++k)
                    {
                        if (done || k >= wn)
                            continue;
                        uint32_t w = w_prev[k];
                        const uint32_t * sp = status + (uint64_t)(base - 1 - k) * kBins + t;
                        while ((w >> 30) == 0)
                        {
                            __builtin_amdgcn_s_sleep(1);
                            w = __hip_atomic_load(sp, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                        }
                        sum += w & kStMask;
                        done = (w & kStIncl) != 0;
                    }
                    base -= wn;
                }
                __hip_atomic_store(st_mine + t, kStIncl | (sum + hist[t]), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
            const unsigned at = bstart2[t] + sum;
            cursor[t] = at;
            if (ti == map.tp[grp]) // the group's first tile: here the slice (bin t, group) starts
                fstart[((unsigned)t << g.d1) | grp] = at;
            // blocks of the static tiling (the way back) whose tile range starts with this tile
            for (unsigned b = (unsigned)(((uint64_t)ti * g.G + nt - 1) / nt); b < g.G && (unsigned)((uint64_t)nt * b / g.G) == ti; ++b)
                offs_out[(uint64_t)t * g.G + b] = at;
        }
        if (t == 0) // (the answer arrives while the runs are written out)
            next_ti = atomicAdd(ticket, 1u);
        __syncthreads();
        { // runs out: 16 lanes per bin
            const unsigned l = t & 15;
            for (unsigned b = t >> 4; b < bins; b += TT / 16)
            {
                const unsigned cnt = hist[b];
                if (cnt == 0)
                    continue;
                const unsigned st = start[b], cur = cursor[b];
                if (cnt > kBigRun)
                {
                    if (l == 0)
                        big[atomicAdd(&n_big, 1u)] = b;
                    continue;
                }
                uint32_t * dst = keys_out + cur;
                for (unsigned i = l; i < cnt; i += 16)
                    dst[i] = sorted[st + i];
            }
        }
        __syncthreads();
        const unsigned nb = n_big;
        for (unsigned k = 0; k < nb; ++k)
        {
            const unsigned b = big[k], cnt = hist[b], st = start[b], cur = cursor[b];
            for (unsigned i = t; i < cnt; i += TT)
                keys_out[(uint64_t)cur + i] = sorted[st + i];
        }
        __syncthreads();
    }
}

// slice starts with holes (0xFFFFFFFF: a slice of a digit-1 group that has no tile, i.e. no keys) -> starts and first work
// items.  An empty slice starts where the next one does.  One block; the arrays are walked in chunks of 8192 entries staged
// through LDS so that global accesses stay coalesced (thread t owns entries 8t .. 8t+7 of a chunk).
__global__ __launch_bounds__(1024) void k_sr_fine_scan_starts(unsigned nf, unsigned total, uint32_t * __restrict__ fstart,
                                                              uint32_t * __restrict__ ioff)
{
    constexpr unsigned kChunk = 8192, kOwn = kChunk / 1024;
    __shared__ unsigned buf[kChunk + 1], first[1024], wred[16];
    const unsigned t = threadIdx.x;
    // backward: fill the holes
    unsigned carry = total; // the first set value behind the chunk
    for (unsigned chi = nf; chi > 0;)
    {
        const unsigned clo = chi > kChunk ? chi - kChunk : 0, cn = chi - clo;
        for (unsigned i = t; i < cn; i += 1024)
            buf[i] = fstart[clo + i];
        __syncthreads();
        const unsigned o = t * kOwn;
        unsigned fv = 0xFFFFFFFFu;
        for (unsigned j = 0; j < kOwn && o + j < cn && fv == 0xFFFFFFFFu; ++j)
            fv = buf[o + j];
        first[t] = fv;
        __syncthreads();
        unsigned nxt = carry;
        for (unsigned u = t + 1; u < 1024; ++u)
            if (first[u] != 0xFFFFFFFFu)
            {
                nxt = first[u];
                break;
            }
        for (unsigned j = kOwn; j-- > 0;)
            if (o + j < cn)
            {
                unsigned v = buf[o + j];
                if (v == 0xFFFFFFFFu)
                    v = nxt;
                buf[o + j] = v;
                nxt = v;
            }
        __syncthreads();
        for (unsigned i = t; i < cn; i += 1024)
            fstart[clo + i] = buf[i];
        carry = buf[0];
        __syncthreads();
        chi = clo;
    }
    if (t == 0)
        fstart[nf] = total;
    __syncthreads(); // (one block: its own global writes are visible to it after the barrier)
    // forward: first work item of every slice
    unsigned run = 0;
    for (unsigned clo = 0; clo < nf; clo += kChunk)
    {
        const unsigned cn = nf - clo < kChunk ? nf - clo : kChunk;
        for (unsigned i = t; i <= cn; i += 1024)
            buf[i] = fstart[clo + i]; // (entry cn: the start behind the chunk; fstart has nf + 1 entries)
        __syncthreads();
        const unsigned o = t * kOwn;
        unsigned si = 0;
        for (unsigned j = 0; j < kOwn && o + j < cn; ++j)
            si += (buf[o + j + 1] - buf[o + j] + kItemKeys - 1) / kItemKeys;
        const unsigned ii = wave_incl_scan(si);
        if ((t & 63) == 63)
            wred[t >> 6] = ii;
        __syncthreads();
        unsigned bi = run + ii - si, all = 0;
        for (unsigned w = 0; w < 16; ++w)
        {
            if (w < (t >> 6))
                bi += wred[w];
            all += wred[w];
        }
        unsigned items[kOwn];
        for (unsigned j = 0; j < kOwn && o + j < cn; ++j)
        {
            items[j] = bi;
            bi += (buf[o + j + 1] - buf[o + j] + kItemKeys - 1) / kItemKeys;
        }
        __syncthreads();
        for (unsigned j = 0; j < kOwn && o + j < cn; ++j)
            buf[o + j] = items[j];
        __syncthreads();
        for (unsigned i = t; i < cn; i += 1024)
            ioff[clo + i] = buf[i];
        run += all;
        __syncthreads();
    }
    if (t == 0)
        ioff[nf] = run;
}

// ---- rank out of LDS, in place over the final keys -----------------------------------------------------------------
// Once a slice is in LDS every line header is rewritten for the queries to come: [ones before the line, relative to
// the slice: 20 bits | ones in word 0: 9 bits | in words 0..2: 9 bits | in words 0..4: 9 bits].  A query then reads
// the header's 16 bytes and the one 16-byte pair that holds its word: two LDS reads and two popcounts instead of the
// whole line (the first form of this kernel spent 1.4 wave instructions per key, 65 % of its time in the VALU).
// MULTI (vectors of more than 2^26 lines): a slice covers 2^slog > 2^kSliceLog lines and is staged in rounds of 2^kSliceLog; every
// round walks all keys of the item and answers — and stores — those whose line lies in the staged part (the others' stores go
// beyond the buffer's end, i.e. nowhere).  The keys are read once per round (they sit in the L2 by then) and an answer is the
// ones in front of the position relative to the WHOLE slice's first line, as the way back expects.
// (MULTI, round 6: the keys are read once per round and a 128-byte line of answers is completed by up to eight rounds — through the
// caches (AUX 0) the re-reads hit the L2 and the partial stores of a line merge there before they leave for HBM; streamed
// (non-temporal) they went out as partial sectors.  SDSL_HIP_WIDE_NT=1 at compile time keeps the streamed form for A/B.)
#ifdef SDSL_HIP_WIDE_NT
constexpr bool kWideCached = false;
#else
constexpr bool kWideCached = true;
#endif
// SUBLOG / THREADS (round 6): lines staged per round and threads of a block.  The single-round kernel is <false, kSliceLog, kRT>: 64 KiB of
// lines, two blocks of 512 threads per CU.  Wide slices take <true, kSliceLog + 1, 2 * kRT>: 128 KiB per round and ONE block of 1024 threads
// per CU — the same waves per CU, half the rounds, i.e. half the walks over an item's keys (2^36 bits: two rounds instead of four).
template <bool MULTI, unsigned SUBLOG = kSliceLog, unsigned THREADS = kRT>
__global__ __launch_bounds__(THREADS) void k_sr_rank_lds(BvView bv, int bit, unsigned nf, unsigned d1, unsigned d2, const uint32_t * __restrict__ fstart,
                                                     const uint32_t * __restrict__ ioff, uint32_t * __restrict__ keys, const uint32_t * __restrict__ go,
                                                     unsigned slog)
{
    if (go && !*go)
        return;
    typedef unsigned long long v2u64 __attribute__((ext_vector_type(2)));
    __shared__ v2u64 slice[(kLW << SUBLOG) / 2]; // 64 KiB (SUBLOG 10) or 128 KiB (11)
    // the rewritten headers live in an array of their own: inside the lines they all sit at multiples of 64 bytes, i.e. in
    // 4 of the 64 LDS banks, and a wave's 64 random header reads serialise 16-fold (SQ_LDS_BANK_CONFLICT was 75 % of the
    // LDS cycles); packed, consecutive headers are 8 bytes apart and random reads spread over all banks
    __shared__ uint64_t hdr[1u << SUBLOG];
    __shared__ unsigned sh_f;
    constexpr int U = 8;
    const unsigned t = threadIdx.x;
    const unsigned n_items = ioff[nf];
    for (unsigned item = blockIdx.x; item < n_items; item += gridDim.x)
    {
        if (t == 0)
        { // the slice of this item: last f with ioff[f] <= item (empty slices have no items and are skipped over)
            unsigned a = 0, z = nf;
            while (a + 1 < z)
            {
                const unsigned m = (a + z) >> 1;
                if (ioff[m] <= item)
                    a = m;
                else
                    z = m;
            }
            sh_f = a;
        }
        __syncthreads(); // also: everybody is done with the previous slice
        const unsigned f = sh_f;
        const uint64_t Ls = (uint64_t)sr_slice_of(f, d1, d2) << (MULTI ? slog : SUBLOG); // the slice's first line
        const unsigned n_sub = MULTI ? 1u << (slog - SUBLOG) : 1u;
        const uint64_t Hs = MULTI ? bv.lines[Ls * kLW] : 0; // ones in front of the slice
      for (unsigned sub = 0; sub < n_sub; ++sub)
      {
        const uint64_t L0 = Ls + ((uint64_t)sub << SUBLOG);
        if (MULTI && L0 >= bv.n_lines)
            break;
        if (MULTI && sub)
            __syncthreads(); // everybody is done with the part staged before
        const unsigned nl = (unsigned)(bv.n_lines - L0 < (UINT64_C(1) << SUBLOG) ? bv.n_lines - L0 : (UINT64_C(1) << SUBLOG));
        const v2u64 * src = reinterpret_cast<const v2u64 *>(bv.lines + L0 * kLW);
        for (unsigned i = t; i < nl * (kLW / 2); i += THREADS)
            slice[i] = __builtin_nontemporal_load(src + i);
        const uint64_t lo = (uint64_t)fstart[f] + (uint64_t)(item - ioff[f]) * kItemKeys;
        const uint64_t fend = fstart[f + 1];
        const unsigned cnt = (unsigned)(lo + kItemKeys < fend ? kItemKeys : fend - lo);
        // the keys through a buffer of exactly this item's extent: what lies beyond reads as 0 and is not written back
        // (the item is worked on in whole 128-byte lines of the key array: the `head` keys of the line in front of its first one are
        // skipped by their lanes, so that every wave's loads and stores cover whole lines — unaligned non-temporal stores went out as
        // partial sectors: 4.5 GB written for 4.0 GB of answers)
        const unsigned head = (unsigned)lo & 31u, cnth = cnt + head;
        const unsigned vo0 = t < head ? 0xFFFFFFFCu : t * 4u; // first round, first key: beyond the buffer = not stored
        const rsrc_t rs_k = make_rsrc(keys + uniform64(lo - head), __builtin_amdgcn_readfirstlane(cnth) * 4u);
        // the first keys are asked for while the headers are being rewritten
        uint32_t key[U];
#pragma unroll
        for (int u = 0; u < U; ++u)
            {
                if constexpr (MULTI && kWideCached)
                    buf_load_cached(rs_k, t * 4u, (unsigned)u * THREADS * 4u, key[u]);
                else
                    buf_load(rs_k, t * 4u, (unsigned)u * THREADS * 4u, key[u]);
            }
        __syncthreads();
        const uint64_t H = slice[0].x;
        __syncthreads();
        for (unsigned ln = t; ln < nl; ln += THREADS)
        {
            v2u64 * w = slice + ln * (kLW / 2);
            const v2u64 a = w[0], b = w[1], c = w[2];
            const unsigned ca = popc64(a.y), cb = ca + popc64(b.x) + popc64(b.y), cc = cb + popc64(c.x) + popc64(c.y);
            hdr[ln] = (a.x - H) | ((uint64_t)ca << 20) | ((uint64_t)cb << 29) | ((uint64_t)cc << 38);
        }
        __syncthreads();
        for (unsigned i0 = 0; i0 < cnth; i0 += THREADS * U)
        { // the next round's keys are requested before this round's answers are stored (loads and stores share a counter)
            uint32_t nk[U];
            const unsigned n0 = (i0 + THREADS * U) * 4u;
            if (i0 + THREADS * U < cnth)
            {
#pragma unroll
                for (int u = 0; u < U; ++u)
                    {
                    if constexpr (MULTI && kWideCached)
                        buf_load_cached(rs_k, t * 4u, n0 + (unsigned)u * THREADS * 4u, nk[u]);
                    else
                        buf_load(rs_k, t * 4u, n0 + (unsigned)u * THREADS * 4u, nk[u]);
                }
            }
#pragma unroll
            for (int u = 0; u < U; ++u)
            {
                if (u == 0 && i0 == 0 && t < head)
                    key[u] = kBad; // (a key of the line's head: the item in front of this one answers it)
                const unsigned lnf = key[u] == kBad ? 0 : key[u] >> kOffBits; // line inside the slice
                const bool here = !MULTI || key[u] == kBad || (lnf >> SUBLOG) == sub; // staged in this round (kBad: answered in round 0)
                const unsigned ln = MULTI ? (here ? lnf & ((1u << SUBLOG) - 1) : 0u) : lnf;
                const unsigned off = key[u] & ((1u << kOffBits) - 1);
                const v2u64 * w = slice + ln * (kLW / 2);
                const unsigned wi = off >> 6, k = (wi + 1) >> 1; // data word of the position, 16-byte quarter holding it
                const uint64_t hx = hdr[ln];
                const v2u64 p = w[k]; // k == 0: (original header, word 0)
                const uint64_t x = k ? p.x : 0, y = p.y;
                const uint64_t m = lo_set(off & 63);
                const unsigned base = k ? (unsigned)(hx >> (11 + 9 * k)) & 0x1FFu : 0u;
                const unsigned part = (wi & 1) ? popc64(x & m) : popc64(x) + popc64(y & m);
                const uint32_t r1 = (MULTI ? (uint32_t)(H - Hs) : 0u) + ((uint32_t)hx & 0xFFFFFu) + base + part;
                uint32_t r = bit ? r1 : lnf * (uint32_t)kDB + off - r1;
                if (key[u] == kBad)
                    r = kBad;
                unsigned vo = u == 0 && i0 == 0 ? vo0 : t * 4u;
                if (MULTI && (!here || (key[u] == kBad && sub != 0)))
                    vo = 0xFFFFFFFCu; // another round's key: not stored
                __builtin_amdgcn_raw_buffer_store_b32(r, rs_k, (int)vo, (int)(i0 * 4u + (unsigned)u * THREADS * 4u), MULTI && kWideCached ? 0 : kAuxNT);
            }
            if (i0 + THREADS * U < cnth)
            {
#pragma unroll
                for (int u = 0; u < U; ++u)
                    key[u] = nk[u];
            }
        }
      } // rounds of a wide slice
    }
}

// ---- select out of LDS, in place over the final keys -----------------------------------------------------------------
// Bucket f holds the arguments of rank [f * 2^r, (f + 1) * 2^r); select is monotone, so they live in the lines
// [bnd[f], bnd[f + 1]] — usually fewer than 1024.  The lines are staged in LDS with their headers rewritten as in
// k_sr_rank_lds (arguments before the line relative to the slice | arguments in word 0 | in words 0..2 | in words 0..4) and,
// for BIT == 0, the words complemented under the validity mask, so that both bit values search the same way: an
// interpolated guess at the line, a short bisection over the headers in LDS, then the 16-byte pair that holds the word and
// sel64 inside it.  A bucket that spans more than an LDS slice (a sparse stretch) is left to the fix-up pass.
template <int BIT>
__global__ __launch_bounds__(kST) void k_sr_select_lds(BvView bv, unsigned nf, unsigned d1, unsigned d2, unsigned B, const uint32_t * __restrict__ bnd,
                                                       const uint32_t * __restrict__ fstart, const uint32_t * __restrict__ ioff,
                                                       uint32_t * __restrict__ keys, uint32_t * __restrict__ any_marked, const uint32_t * __restrict__ go)
{
    if (go && !*go)
        return;
    typedef unsigned long long v2u64 __attribute__((ext_vector_type(2)));
    __shared__ v2u64 slice[(kLW << kSliceLog) / 2]; // 64 KiB
    __shared__ uint64_t hdr[1u << kSliceLog];       // the rewritten headers, packed (see k_sr_rank_lds)
    constexpr unsigned kInv = 2048;
    __shared__ uint16_t inv[kInv + 1];              // inv[e]: the line that holds the slice's argument number e << s
    __shared__ unsigned sh_f;
    __shared__ unsigned sh_tot;
    constexpr int U = 4;
    const unsigned t = threadIdx.x;
    const unsigned n_items = ioff[nf];
    for (unsigned item = blockIdx.x; item < n_items; item += gridDim.x)
    {
        if (t == 0)
        {
            unsigned a = 0, z = nf;
            while (a + 1 < z)
            {
                const unsigned m = (a + z) >> 1;
                if (ioff[m] <= item)
                    a = m;
                else
                    z = m;
            }
            sh_f = a;
        }
        __syncthreads();
        const unsigned f = sh_f;
        const uint64_t lo = (uint64_t)fstart[f] + (uint64_t)(item - ioff[f]) * kItemKeys;
        const uint64_t fend = fstart[f + 1];
        const unsigned cnt = (unsigned)(lo + kItemKeys < fend ? kItemKeys : fend - lo);
        uint32_t * kp = keys + lo;
        const unsigned bk = sr_slice_of(f, d1, d2); // the bucket (f: its place in the tables)
        const uint64_t L0 = bnd[bk];
        const uint64_t L1 = (uint64_t)bnd[bk + 1] + 1 < bv.n_lines ? (uint64_t)bnd[bk + 1] + 1 : bv.n_lines;
        if (L1 - L0 > (UINT64_C(1) << kSliceLog))
        { // wider than a slice: the fix-up pass answers these
            for (unsigned i = t; i < cnt; i += kST)
                if (kp[i] != kBad)
                    kp[i] = kMark;
            if (t == 0)
                *any_marked = 1;
            __syncthreads(); // (thread 0 rewrites sh_f at the top of the next item: nobody may still be reading it)
            continue;
        }
        const unsigned nl = (unsigned)(L1 - L0);
        const v2u64 * src = reinterpret_cast<const v2u64 *>(bv.lines + L0 * kLW);
        for (unsigned i = t; i < nl * (kLW / 2); i += kST)
            slice[i] = __builtin_nontemporal_load(src + i);
        // the keys through a buffer of exactly this item's extent: what lies beyond reads as 0 and is not written back
        // (the item is worked on in whole 128-byte lines of the key array: the `head` keys of the line in front of its first one are
        // skipped by their lanes, so that every wave's loads and stores cover whole lines — unaligned non-temporal stores went out as
        // partial sectors: 4.5 GB written for 4.0 GB of answers)
        const unsigned head = (unsigned)lo & 31u, cnth = cnt + head;
        const unsigned vo0 = t < head ? 0xFFFFFFFCu : t * 4u; // first round, first key: beyond the buffer = not stored
        const rsrc_t rs_k = make_rsrc(keys + uniform64(lo - head), __builtin_amdgcn_readfirstlane(cnth) * 4u);
        uint32_t key[U];
#pragma unroll
        for (int u = 0; u < U; ++u)
            buf_load(rs_k, t * 4u, (unsigned)u * kST * 4u, key[u]);
        __syncthreads();
        const uint64_t h0 = slice[0].x;
        const uint64_t A0 = BIT ? h0 : L0 * kDB - h0; // arguments in front of the slice
        __syncthreads();
        for (unsigned ln = t; ln < nl; ln += kST)
        {
            v2u64 * w = slice + ln * (kLW / 2);
            v2u64 a = w[0], b = w[1], c = w[2], d = w[3];
            const uint64_t L = L0 + ln;
            uint64_t before = a.x;
            if (!BIT)
            {
                before = L * kDB - a.x;
                if ((L + 1) * kDB <= bv.n_bits)
                {
                    a.y = ~a.y, b.x = ~b.x, b.y = ~b.y, c.x = ~c.x, c.y = ~c.y, d.x = ~d.x, d.y = ~d.y;
                }
                else
                {
                    a.y = ~a.y & valid_mask(bv.n_bits, L, 0);
                    b.x = ~b.x & valid_mask(bv.n_bits, L, 1);
                    b.y = ~b.y & valid_mask(bv.n_bits, L, 2);
                    c.x = ~c.x & valid_mask(bv.n_bits, L, 3);
                    c.y = ~c.y & valid_mask(bv.n_bits, L, 4);
                    d.x = ~d.x & valid_mask(bv.n_bits, L, 5);
                    d.y = ~d.y & valid_mask(bv.n_bits, L, 6);
                }
                w[0] = a;
                w[1] = b;
                w[2] = c;
                w[3] = d;
            }
            const unsigned ca = popc64(a.y), cb = ca + popc64(b.x) + popc64(b.y), cc = cb + popc64(c.x) + popc64(c.y);
            hdr[ln] = (before - A0) | ((uint64_t)ca << 20) | ((uint64_t)cb << 29) | ((uint64_t)cc << 38);
            if (ln == nl - 1)
                sh_tot = (unsigned)(before - A0) + cc + popc64(d.x) + popc64(d.y); // arguments inside the slice
        }
        __syncthreads();
        const uint64_t t0 = (uint64_t)bk * B - A0; // rank of the bucket's first argument, relative to the slice
        // the inverse of the headers, sampled: argument number e << s of the slice lies in line inv[e] (every line enters the
        // samples that fall into it; lines without arguments enter none).  A query then starts at the sample below its
        // argument and walks forward — two or three dependent LDS reads instead of the six to eight of an interpolated search
        const unsigned tot = sh_tot;
        unsigned sh = 0;
        while ((tot >> sh) >= kInv)
            ++sh;
        const unsigned n_inv = ((tot + (1u << sh) - 1) >> sh); // samples 0 .. n_inv - 1 (arguments 0, 2^s, ...)
        for (unsigned ln = t; ln < nl; ln += kST)
        {
            const unsigned a0 = (unsigned)hdr[ln] & 0xFFFFFu, a1 = ln + 1 < nl ? (unsigned)hdr[ln + 1] & 0xFFFFFu : tot;
            for (unsigned e = (a0 + (1u << sh) - 1) >> sh; (e << sh) < a1; ++e)
                inv[e] = (uint16_t)ln;
        }
        if (t == 0)
            inv[n_inv] = (uint16_t)(nl - 1);
        __syncthreads();
        for (unsigned i0 = 0; i0 < cnth; i0 += kST * U)
        { // the next round's keys are requested before this round's answers are stored (loads and stores share a counter)
            uint32_t nk[U];
            const unsigned n0 = (i0 + kST * U) * 4u;
            if (i0 + kST * U < cnth)
            {
#pragma unroll
                for (int u = 0; u < U; ++u)
                    buf_load(rs_k, t * 4u, n0 + (unsigned)u * kST * 4u, nk[u]);
            }
#pragma unroll
            for (int u = 0; u < U; ++u)
            {
                if (u == 0 && i0 == 0 && t < head)
                    key[u] = kBad; // (a key of the line's head: the item in front of this one answers it)
                uint32_t res = kBad;
                if (key[u] != kBad)
                {
                    const unsigned tg = (unsigned)t0 + key[u]; // arguments of the slice in front of the wanted one
                    // line a with rel(a) <= tg < rel(a + 1): from the sample below tg, a short walk; a bisection up to the next
                    // sample's line if the walk does not get there (empty lines in between)
                    auto rel = [&](unsigned j) -> unsigned { return (unsigned)hdr[j] & 0xFFFFFu; };
                    const unsigned e = tg >> sh;
                    unsigned a = inv[e], z = (unsigned)inv[e + 1] + 1; // the line lies in [a, z)
                    z = z > nl ? nl : z;
#pragma unroll
                    for (int it = 0; it < 3; ++it)
                        if (a + 1 < z && rel(a + 1) <= tg)
                            ++a;
                    if (a + 1 < z && rel(a + 1) <= tg)
                    {
                        while (z - a > 1)
                        {
                            const unsigned m = (a + z) >> 1;
                            if (rel(m) <= tg)
                                a = m;
                            else
                                z = m;
                        }
                    }
                    const v2u64 * w = slice + a * (kLW / 2);
                    const uint64_t hx = hdr[a];
                    unsigned tl = tg - ((unsigned)hx & 0xFFFFFu); // index inside the line
                    const unsigned ca = (unsigned)(hx >> 20) & 0x1FFu, cb = (unsigned)(hx >> 29) & 0x1FFu,
                                   cc = (unsigned)(hx >> 38) & 0x1FFu;
                    // the 16-byte quarter that holds the argument: k = 0 -> word 0 (its .y), k >= 1 -> words 2k - 1, 2k
                    const unsigned k = (tl >= ca ? 1u : 0u) + (tl >= cb ? 1u : 0u) + (tl >= cc ? 1u : 0u);
                    tl -= k == 0 ? 0u : (k == 1 ? ca : (k == 2 ? cb : cc));
                    const v2u64 pr = w[k];
                    const unsigned px = k ? popc64(pr.x) : 0u;
                    const bool second = tl >= px;
                    const unsigned word = k ? 2 * k - 1 + (second ? 1u : 0u) : 0u;
                    const unsigned bitpos = sel64(second ? pr.y : pr.x, tl - (second ? px : 0u) + 1);
                    res = a * (uint32_t)kDB + 64u * word + bitpos; // relative to the slice's first bit
                }
                __builtin_amdgcn_raw_buffer_store_b32(res, rs_k, (int)(u == 0 && i0 == 0 ? vo0 : t * 4u), (int)(i0 * 4u + (unsigned)u * kST * 4u), kAuxNT);
            }
            if (i0 + kST * U < cnth)
            {
#pragma unroll
                for (int u = 0; u < U; ++u)
                    key[u] = nk[u];
            }
        }
    }
}

// first bit of every select bucket's slice (what makes a slice-relative position absolute)
__global__ __launch_bounds__(256) void k_sr_select_bases(unsigned nf, unsigned n_buckets, unsigned d1, unsigned d2,
                                                         const uint32_t * __restrict__ bnd, uint64_t * __restrict__ hf)
{
    const unsigned f = blockIdx.x * 256 + threadIdx.x;
    if (f < nf)
    {
        const unsigned bk = sr_slice_of(f, d1, d2);
        hf[f] = bk < n_buckets ? (uint64_t)bnd[bk] * kDB : 0; // first bit of the bucket's first line
    }
}

// The arguments that were left over (kMark64 in the output): one lane per query, bracket from the directory, bisection over
// the line headers in global memory, then the line itself.  Slow and rare: stretches where 2^r arguments span more than
// 1024 lines.
template <int BIT>
__global__ __launch_bounds__(256) void k_sr_select_fixup(BvView bv, const uint32_t * __restrict__ any_marked,
                                                         const uint64_t * __restrict__ iq, uint64_t * __restrict__ out, uint64_t n,
                                                         const uint32_t * __restrict__ go)
{
    if ((go && !*go) || !*any_marked)
        return; // the usual case: every bucket fitted a slice
    for (uint64_t q = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; q < n; q += (uint64_t)gridDim.x * blockDim.x)
    {
        if (out[q] != kMark64)
            continue;
        const uint64_t k = iq[q] - 1;
        const SelSamples sm = sel_samples<BIT>(bv, k);
        const SelBracket br = sel_bracket<BIT>(bv, k, sm);
        uint64_t a = br.lo_pos / kDB, z = (br.hi_pos + kDB - 1) / kDB; // line a has <= k arguments in front, line z more
        if (z > bv.n_lines)
            z = bv.n_lines;
        auto before = [&](uint64_t L) -> uint64_t
        {
            const uint64_t h = bv.lines[L * kLW];
            return BIT ? h : L * kDB - h;
        };
        while (z - a > 1)
        {
            const uint64_t m = (a + z) >> 1;
            if (before(m) <= k)
                a = m;
            else
                z = m;
        }
        unsigned tl = (unsigned)(k - before(a));
        uint64_t pos = SDSL_HIP_NPOS;
        for (int d = 0; d < kDW; ++d)
        {
            uint64_t x = bv.lines[a * kLW + 1 + d];
            if (!BIT)
                x = ~x & valid_mask(bv.n_bits, a, d);
            const unsigned pc = popc64(x);
            if (tl < pc)
            {
                pos = a * kDB + 64ull * d + sel64(x, tl + 1);
                break;
            }
            tl -= pc;
        }
        out[q] = pos;
    }
}

// ---- the way back ---------------------------------------------------------------------------------------------------
// P == 2: slice-relative answers (order of partition 2) -> answers relative to their pass-1 bin, in the order of partition 1
// P == 1: absolute answers in the order of partition 1 -> the caller's array
// V & 1: the runs are fetched four bins at a time (else bin after bin); V & 2: the slots are asked for before the gather
template <int P, unsigned TT, unsigned PER, int V>
__global__ __launch_bounds__(TT, PER >= 16 ? 6 : 8) void k_sr_unpermute(const uint64_t * __restrict__ hf, int bit, SrGeom g, const uint32_t * __restrict__ tprefix,
                                                     const uint32_t * __restrict__ gstart, const uint32_t * __restrict__ offs,
                                                     const uint32_t * __restrict__ res_lo, uint32_t * __restrict__ any_marked,
                                                     const uint16_t * __restrict__ slots,
                                                     const uint16_t * __restrict__ tile_hist, uint32_t * __restrict__ out_lo,
                                                     uint64_t * __restrict__ out)
{
    constexpr unsigned kTile = TT * PER;
    __shared__ uint32_t lo32[kTile];
    __shared__ uint8_t hi8[P == 1 ? kTile : 1]; // pass 1: bits 32.. of the absolute answers (0xFF: NPOS, 0xFE: left to the fix-up)
    __shared__ unsigned hist[kBins], start[kBins], cursor[kBins];
    __shared__ unsigned wsum[kBins / 64];
    __shared__ unsigned big[kTile / (kBigRun + 1) + 1], n_big;
    __shared__ TileMap map;
    const unsigned t = threadIdx.x;
    const unsigned bins = 1u << (P == 1 ? g.d1 : g.d2);
    sr_load_map<P, TT>(g, map, tprefix, gstart);
    for (unsigned i = t; i < kBins; i += TT)
        cursor[i] = i < bins ? offs[(uint64_t)i * g.G + blockIdx.x] : 0;
    __syncthreads();
    const unsigned nt = sr_tiles<P>(g, map);
    const unsigned tlo = (unsigned)((uint64_t)nt * blockIdx.x / g.G), thi = (unsigned)((uint64_t)nt * (blockIdx.x + 1) / g.G);
    for (unsigned ti = tlo; ti < thi; ++ti)
    {
        uint64_t lo, hi;
        unsigned grp;
        sr_tile_range<P>(g, map, ti, lo, hi, grp);
        uint16_t sl[PER];
        const unsigned cnt_t = (unsigned)(hi - lo);
        const uint16_t * slots_t = slots + lo;
        const bool full_tile = cnt_t == kTile; // all but the last tile of a group: no per-key bounds checks
        auto load_slots = [&](auto full_c)
        {
            constexpr bool FULL = decltype(full_c)::value;
#pragma unroll
            for (unsigned u = 0; u < PER; ++u)
            {
                const unsigned q = u * TT + t;
                sl[u] = FULL || q < cnt_t ? __builtin_nontemporal_load(slots_t + q) : (uint16_t)0;
            }
        };
        if (V & 2)
        { // the slots of this tile: asked for now, used after the gather
            if (full_tile)
                load_slots(std::true_type{});
            else
                load_slots(std::false_type{});
        }
        for (unsigned i = t; i < kBins; i += TT)
        {
            const unsigned c = i < bins ? tile_hist[(uint64_t)ti * bins + i] : 0;
            hist[i] = c;
            start[i] = c;
        }
        if (t == 0)
            n_big = 0;
        __syncthreads();
        block_excl_scan_bins(start, wsum);
        // what turns the answers of bin b of this tile into absolute ones (pass 2: the slice's first header)
        // what turns a slice-relative answer into an absolute one: rank: ones (zeros) in front of the slice; select: the first
        // bit of the bucket's first line.  f = place of the slice in the tables, (b2, b1) = its pass-2 / pass-1 digits
        auto abs_base = [&](unsigned b2, unsigned b1) -> uint64_t
        {
            const uint64_t h = hf[(b2 << g.d1) | b1];
            if (g.op == 1)
                return h;
            return bit ? h : ((uint64_t)((b1 << g.d2) | b2) << kSliceLog) * kDB - h;
        };
        // Between the two passes an answer travels as 32 bits RELATIVE TO ITS PASS-1 BIN: such a bin is a contiguous stretch of
        // the vector (2^d2 slices), so all its answers lie within 2^32 of its first slice's base (rank: always; select: unless
        // the stretch is mostly empty space — then the answer is left to the fix-up pass).  0xFFFFFFFF / ..FE stay NPOS / mark.
        const uint64_t gbase = P == 2 ? abs_base(0, grp) : 0;
        auto base_of = [&](unsigned b) -> uint64_t { return P == 2 ? abs_base(b, grp) - gbase : abs_base(0, b); };
        auto keep = [&](uint64_t base, unsigned at, uint32_t v)
        {
            if (P == 2)
            {
                const uint64_t rel = base + v;
                uint32_t r = (uint32_t)rel;
                if (v >= kMark)
                    r = v;
                else if (rel >= kMark)
                {
                    r = kMark;
                    *any_marked = 1;
                }
                lo32[at] = r;
            }
            else
            {
                const uint64_t full = base + v;
                lo32[at] = (uint32_t)full;
                hi8[at] = v == kBad ? (uint8_t)0xFF : (v == kMark ? (uint8_t)0xFE : (uint8_t)(full >> 32));
            }
        };
        if (!(V & 1))
        {
            const unsigned l = t & 15;
            for (unsigned b = t >> 4; b < bins; b += TT / 16)
            {
                const unsigned cnt = hist[b];
                if (cnt == 0)
                    continue;
                const unsigned st = start[b], cur = cursor[b];
                if (cnt > kBigRun)
                {
                    if (l == 0)
                        big[atomicAdd(&n_big, 1u)] = b;
                    continue;
                }
                const uint64_t base = base_of(b);
                for (unsigned i = l; i < cnt; i += 16)
                    keep(base, st + i, res_lo[(uint64_t)cur + i]);
            }
        }
        else
        { // two bins per step, up to four elements of each run per lane, all fetched before any is consumed
            constexpr unsigned kE = 4, kStep = TT / 16;
            const unsigned l = t & 15;
            for (unsigned b0 = t >> 4; b0 < bins; b0 += 2 * kStep)
            {
                unsigned cnt[2], st[2], cur[2];
                uint64_t base[2];
                uint32_t v[2][kE];
#pragma unroll
                for (unsigned k = 0; k < 2; ++k)
                {
                    const unsigned b = b0 + k * kStep;
                    cnt[k] = b < bins ? hist[b & (kBins - 1)] : 0;
                    st[k] = start[b & (kBins - 1)];
                    cur[k] = cursor[b & (kBins - 1)];
                    if (cnt[k] > kBigRun)
                    {
                        if (l == 0)
                            big[atomicAdd(&n_big, 1u)] = b;
                        cnt[k] = 0;
                    }
                }
#pragma unroll
                for (unsigned k = 0; k < 2; ++k)
                {
                    base[k] = cnt[k] ? base_of(b0 + k * kStep) : 0;
#pragma unroll
                    for (unsigned e = 0; e < kE; ++e)
                    {
                        const unsigned i = l + 16 * e;
                        const bool on = i < cnt[k];
                        v[k][e] = on ? res_lo[(uint64_t)cur[k] + i] : 0;
                    }
                }
#pragma unroll
                for (unsigned k = 0; k < 2; ++k)
                {
#pragma unroll
                    for (unsigned e = 0; e < kE; ++e)
                        if (l + 16 * e < cnt[k])
                            keep(base[k], st[k] + l + 16 * e, v[k][e]);
                    for (unsigned i = l + 16 * kE; i < cnt[k]; i += 16)
                        keep(base[k], st[k] + i, res_lo[(uint64_t)cur[k] + i]);
                }
            }
        }
        __syncthreads();
        const unsigned nb = n_big;
        for (unsigned k = 0; k < nb; ++k)
        {
            const unsigned b = big[k], c = hist[b], s0 = start[b], cu = cursor[b];
            const uint64_t base = base_of(b);
            for (unsigned i = t; i < c; i += TT)
                keep(base, s0 + i, res_lo[(uint64_t)cu + i]);
        }
        __syncthreads();
        if (!(V & 2))
        {
            if (full_tile)
                load_slots(std::true_type{});
            else
                load_slots(std::false_type{});
        }
        uint32_t * out_lo_t = P == 2 ? out_lo + lo : nullptr;
        uint64_t * out_t = P == 1 ? out + lo : nullptr;
        auto store_out = [&](auto full_c)
        {
            constexpr bool FULL = decltype(full_c)::value;
#pragma unroll
            for (unsigned u = 0; u < PER; ++u)
            {
                const unsigned q = u * TT + t;
                if (FULL || q < cnt_t)
                {
                    const unsigned h = P == 1 ? hi8[sl[u]] : 0u;
                    const uint32_t l32 = lo32[sl[u]];
                    if (P == 2)
                        __builtin_nontemporal_store(l32, out_lo_t + q);
                    else
                        __builtin_nontemporal_store(h == 0xFFu ? SDSL_HIP_NPOS : (h == 0xFEu ? kMark64 : ((uint64_t)h << 32) | l32),
                                                    out_t + q);
                }
            }
        };
        if (full_tile)
            store_out(std::true_type{});
        else
            store_out(std::false_type{});
        for (unsigned i = t; i < kBins; i += TT)
            cursor[i] += hist[i];
        __syncthreads();
    }
}

std::mutex g_phase_mutex;
std::string g_last_phases; // "name=ms;..." of the most recent traced call (option "trace_phases")

constexpr uint64_t kMaxPass = (UINT64_C(1) << 30) - (UINT64_C(1) << 20); // positions per pass over the batch (32-bit cursors; the
                                                                        // look-back words of the one-sweep pass hold 30-bit counts)
constexpr unsigned kMaxG = 2048;                 // partition blocks (k_sr_bucket_offsets: 8 per thread)

size_t carve(SrBuf & b, void * scratch, uint64_t n, unsigned tile)
{
    uint8_t * p = (uint8_t *)scratch;
    auto take = [&](size_t bytes) -> void *
    {
        void * r = p;
        p += (bytes + 255) & ~(size_t)255;
        return r;
    };
    const uint64_t tiles1 = (n + tile - 1) / tile, tiles2 = tiles1 + kBins;
    b.keys1 = (uint32_t *)take(n * 4);
    b.keys2 = (uint32_t *)take(n * 4);
    b.slots1 = (uint16_t *)take(n * 2);
    b.slots2 = (uint16_t *)take(n * 2);
    b.thist1 = (uint16_t *)take(tiles1 * kBins * 2);
    b.thist2 = (uint16_t *)take(tiles2 * kBins * 2);
    b.counts1 = (uint32_t *)take((size_t)kBins * kMaxG * 4);
    b.offs1 = (uint32_t *)take((size_t)kBins * kMaxG * 4);
    b.counts2 = (uint32_t *)take((size_t)kBins * kMaxG * 4);
    b.offs2 = (uint32_t *)take((size_t)kBins * kMaxG * 4);
    b.bstart1 = (uint32_t *)take((kBins + 1) * 4);
    b.bstart2 = (uint32_t *)take((kBins + 1) * 4);
    b.btot = (uint32_t *)take((kBins + 1) * 4);
    b.tprefix2 = (uint32_t *)take((kBins + 1) * 4);
    b.fine_count = (uint32_t *)take((size_t)kBins * kBins * 4);
    b.fstart = (uint32_t *)take(((size_t)kBins * kBins + 1) * 4);
    b.ioff = (uint32_t *)take(((size_t)kBins * kBins + 1) * 4);
    b.hf = (uint64_t *)take((size_t)kBins * kBins * 8);
    b.tot2 = (uint32_t *)take((kBins + 1) * 4);
    b.ticket = (uint32_t *)take(256);
    b.status = (uint32_t *)take(tiles2 * kBins * 4);
    return (size_t)(p - (uint8_t *)scratch);
}

// the partition / un-permute kernels of one block size
struct SrKernels
{
    void (*hist1)(SrGeom, const uint64_t *, const uint32_t *, const uint32_t *, const uint32_t *, uint32_t *, uint32_t *);
    void (*hist2)(SrGeom, const uint64_t *, const uint32_t *, const uint32_t *, const uint32_t *, uint32_t *, uint32_t *);
    void (*part1)(SrGeom, const uint64_t *, const uint32_t *, const uint32_t *, const uint32_t *, const uint32_t *, uint32_t *,
                  uint16_t *, uint16_t *);
    void (*part2)(SrGeom, const uint64_t *, const uint32_t *, const uint32_t *, const uint32_t *, const uint32_t *, uint32_t *,
                  uint16_t *, uint16_t *);
    void (*unp2)(const uint64_t *, int, SrGeom, const uint32_t *, const uint32_t *, const uint32_t *, const uint32_t *, uint32_t *,
                 const uint16_t *, const uint16_t *, uint32_t *, uint64_t *);
    void (*unp1)(const uint64_t *, int, SrGeom, const uint32_t *, const uint32_t *, const uint32_t *, const uint32_t *, uint32_t *,
                 const uint16_t *, const uint16_t *, uint32_t *, uint64_t *);
    void (*part2s)(SrGeom, const uint32_t *, const uint32_t *, const uint32_t *, const uint32_t *, uint32_t *, uint32_t *, uint32_t *,
                   uint16_t *, uint16_t *, uint32_t *, uint32_t *);
    void (*part2s3)(SrGeom, const uint32_t *, const uint32_t *, const uint32_t *, const uint32_t *, uint32_t *, uint32_t *, uint32_t *,
                    uint16_t *, uint16_t *, uint32_t *, uint32_t *); // the same at three blocks per CU (80 VGPRs)
    unsigned threads, blocks_per_cu, per;
};
template <unsigned TT, unsigned PER>
SrKernels sr_kernels(unsigned per_cu)
{
    return SrKernels{k_sr_hist<1, TT, PER>,      k_sr_hist<2, TT, PER>,         k_sr_partition<1, TT, PER>,
                     k_sr_partition<2, TT, PER>, k_sr_unpermute<2, TT, PER, 3>, k_sr_unpermute<1, TT, PER, 3>,
                     k_sr_partition2_sweep<TT, PER, (PER >= 16 ? 4 : 8)>, k_sr_partition2_sweep<TT, PER, (PER >= 16 ? 6 : 8)>,
                     TT,                         per_cu,                        PER};
}

} // namespace

// the slices' answers in place over the final keys (tables in SLICE order: d1 = 0 makes sr_slice_of the identity) + the
// per-slice bases the way back adds (bv_swc.hip)
sdsl_hip_status sr_launch_answers(const BvView & v, int op, int bit, const SelectPlan & sp, unsigned nf, unsigned d2, unsigned slog, const uint32_t * fstart,
                                  const uint32_t * ioff, uint32_t * keys2, uint64_t * hf, uint32_t * marked, const uint32_t * go, hipStream_t s)
{
    static const int rb_env = getenv("SDSL_HIP_SORTED_RANK_BLOCKS") ? atoi(getenv("SDSL_HIP_SORTED_RANK_BLOCKS")) : 0;
    const unsigned slice_blocks = rb_env >= 1 ? (unsigned)rb_env : 1024u;
    if (op == 0)
    {
        hipLaunchKernelGGL(k_sr_slice_bases, dim3((nf + 255) / 256), dim3(256), 0, s, v, nf, 0u, d2, hf, slog);
        if (slog > kSliceLog)
        {
#ifdef SDSL_HIP_WIDE_ROUNDS_1K
            hipLaunchKernelGGL((k_sr_rank_lds<true>), dim3(slice_blocks), dim3(kRT), 0, s, v, bit, nf, 0u, d2, fstart, ioff, keys2, go, slog);
#else
            hipLaunchKernelGGL((k_sr_rank_lds<true, kSliceLog + 1, 2 * kRT>), dim3(slice_blocks), dim3(2 * kRT), 0, s, v, bit, nf, 0u, d2, fstart, ioff, keys2, go, slog);
#endif
        }
        else
            hipLaunchKernelGGL(k_sr_rank_lds<false>, dim3(slice_blocks), dim3(kRT), 0, s, v, bit, nf, 0u, d2, fstart, ioff, keys2, go, slog);
    }
    else
    {
        SH_TRY(fill_u32_async(marked, 0u, 4, s));
        hipLaunchKernelGGL(k_sr_select_bases, dim3((nf + 255) / 256), dim3(256), 0, s, nf, sp.nf, 0u, d2, sp.bnd, hf);
        if (bit)
            hipLaunchKernelGGL(k_sr_select_lds<1>, dim3(slice_blocks), dim3(kST), 0, s, v, nf, 0u, d2, sp.bm << sp.bs, sp.bnd, fstart, ioff,
                               keys2, marked, go);
        else
            hipLaunchKernelGGL(k_sr_select_lds<0>, dim3(slice_blocks), dim3(kST), 0, s, v, nf, 0u, d2, sp.bm << sp.bs, sp.bnd, fstart, ioff,
                               keys2, marked, go);
    }
    SH_HIP(hipGetLastError());
    return SDSL_HIP_OK;
}

void sr_launch_select_fixup(const BvView & v, int bit, const uint32_t * marked, const uint64_t * idx, uint64_t * out, uint64_t cnt,
                            const uint32_t * go, hipStream_t s)
{
    if (bit)
        hipLaunchKernelGGL(k_sr_select_fixup<1>, dim3(256 * 8), dim3(256), 0, s, v, marked, idx, out, cnt, go);
    else
        hipLaunchKernelGGL(k_sr_select_fixup<0>, dim3(256 * 8), dim3(256), 0, s, v, marked, idx, out, cnt, go);
}

void sr_launch_bin_offsets(unsigned bins, unsigned G, const uint32_t * counts, uint32_t * btot, uint32_t * bstart, uint32_t * offs,
                           hipStream_t s)
{
    hipLaunchKernelGGL(k_sr_bucket_totals, dim3(bins), dim3(256), 0, s, G, counts, btot);
    hipLaunchKernelGGL(k_sr_bucket_scan, dim3(1), dim3(kBins), 0, s, bins, 1u, btot, bstart, (uint32_t *)nullptr);
    hipLaunchKernelGGL(k_sr_bucket_offsets, dim3(bins), dim3(256), 0, s, G, counts, bstart, offs);
}

void sr_launch_fine_scan(unsigned nf, const uint32_t * fine_count, uint32_t * fstart, uint32_t * ioff, hipStream_t s)
{
    hipLaunchKernelGGL(k_sr_fine_scan, dim3(1), dim3(1024), 0, s, nf, fine_count, fstart, ioff);
}

void bv_sorted_set_phases(const std::string & line)
{
    std::lock_guard<std::mutex> lock(g_phase_mutex);
    g_last_phases = line;
}

void bv_sorted_clear_phases()
{
    std::lock_guard<std::mutex> lock(g_phase_mutex);
    g_last_phases.clear();
}

std::string bv_sorted_last_phases()
{
    std::lock_guard<std::mutex> lock(g_phase_mutex);
    return g_last_phases;
}

// scratch: 12 bytes per position (two key arrays, two slot arrays) + the tables (incl. 1 KiB of look-back words per tile)
size_t bv_sorted_rank_scratch_bytes(const BvView & v, uint64_t n)
{
    SrBuf b;
    const size_t a = carve(b, nullptr, n < kMaxPass ? n : kMaxPass, 256 * 8) + 4096, c = bv_swc_scratch_bytes(v, n);
    return a > c ? a : c;
}

static bool sr_use_swc();
static_assert(kLimBvBucketedLinesLog == kSliceLog + 16 + kSliceExtraMax, "limits.hpp names what the slices allow");
bool bv_sorted_rank_possible(const BvView & v)
{
    return v.n_lines >= 2 && v.n_lines <= (UINT64_C(1) << (sr_use_swc() ? kLimBvBucketedLinesLog : kSliceLog + 16));
}

bool bv_sorted_rank_applicable(const BvView & v, uint64_t n)
{
    // worth it when the index is larger than the Infinity Cache (256 MiB = 2^22 lines; below that the direct kernel's
    // gathers are served on-die at 52-58 G/s, what this path reaches) and the batch addresses every line at least FOUR times:
    // the slices are streamed once per batch whatever its size and the table kernels cost 0.2 ms, so on 2^34 bits the passes take
    // 1.55 ms + 11.1 ns per 10^3 queries against the direct kernel's 23.1 — they cross at 1.3 x 10^8 queries = 3.4 per line
    // (bench.py extras.batch_sweep: 10^8 queries 37.6 against 43.2 G/s, 10^9: 73.9 against 43.2; round 3 switched at two per line)
    // A vector of 2^20..2^22 lines lives in the Infinity Cache and the direct kernel runs at 49 G/s whatever the batch; the passes
    // reach that at 10^8 queries and 61 G/s at 10^9 (extras.batch_sweep, 2^30 bits): from 2^29 queries on they are taken there too.
    if (bv_sorted_rank_possible(v) && v.n_lines >= (UINT64_C(1) << 20) && v.n_lines < (UINT64_C(1) << 22))
        return n >= (UINT64_C(1) << 29);
    return bv_sorted_rank_possible(v) && v.n_lines >= (UINT64_C(1) << 22) && n >= 4 * v.n_lines;
}

namespace {

// How spread out is a batch?  256 runs of 16 consecutive queries, evenly spaced over the batch: the number of DISTINCT slices
// (rank) / buckets (select) they touch.  Uniformly random positions on a large vector touch about as many slices as there
// are samples; a batch confined to a window, or a sorted one (every run inside one slice), touches few — and then the
// direct kernel, whose fetches hit in L2 / Infinity Cache, beats seven streaming passes (2^20-bit window on 2^34 bits:
// 9.7 ms against 21.7 ms per 10^9 queries).
constexpr unsigned kSampleRuns = 256, kSampleRun = 16;
__global__ __launch_bounds__(1024) void k_sr_sample_spread(SrGeom g, const uint64_t * __restrict__ idx, uint32_t * __restrict__ out)
{
    __shared__ uint32_t bm[2048]; // one bit per slice / bucket (at most 2^16)
    __shared__ unsigned n_valid, n_distinct;
    const unsigned t = threadIdx.x;
    for (unsigned i = t; i < 2048; i += 1024)
        bm[i] = 0;
    if (t == 0)
        n_valid = n_distinct = 0;
    __syncthreads();
    const uint64_t step = g.n / kSampleRuns;
    for (unsigned j = t; j < kSampleRuns * kSampleRun; j += 1024)
    {
        const uint64_t q = step * (j / kSampleRun) + (j % kSampleRun);
        if (q >= g.n)
            continue;
        unsigned dig;
        uint32_t key;
        sr_key1(idx[q], g, dig, key);
        if (key == kBad)
            continue;
        const unsigned id = (dig | ((key >> g.kb) << g.d1)) & 0xFFFFu;
        atomicOr(&bm[id >> 5], 1u << (id & 31));
        atomicAdd(&n_valid, 1u);
    }
    __syncthreads();
    unsigned c = 0;
    for (unsigned i = t; i < 2048; i += 1024)
        c += (unsigned)__builtin_popcount(bm[i]);
    if (c)
        atomicAdd(&n_distinct, c);
    __syncthreads();
    if (t == 0)
    {
        out[0] = n_distinct;
        out[1] = n_valid;
        // uniformly random samples over S slices touch S * (1 - exp(-m / S)) of them; half of the samples' own number is far below
        // that for every vector of >= 2^13 slices and far above what a window or a sorted batch gives.  A smaller vector has
        // between half and all of its 2^(d1 + d2) nominal slices: a quarter of the nominal number then (2^30 bits = 2341 slices of
        // 4096: uniform samples touch 1930 — the former limit of 2048 kept such a vector off the passes for good)
        const unsigned slices = 1u << (g.d1 + g.d2);
        const unsigned limit = n_valid / 2 < slices / 4 ? n_valid / 2 : slices / 4;
        out[2] = n_valid >= 1024 && n_distinct >= limit ? 1u : 0u; // the verdict: the batch is spread over the vector
    }
}

sdsl_hip_status sr_run(const BvView & v, int op, int bit, const SelectPlan & sp, const uint64_t * d_idx, uint64_t n, uint64_t * d_out,
                       hipStream_t s, void * scratch, size_t scratch_bytes)
{
    static const bool trace_env = getenv("SDSL_HIP_TRACE_SORTED") != nullptr;
    const bool trace_opt = g_trace_phases.load() != 0;
    const bool trace = trace_env || trace_opt;
    static const int t_env = getenv("SDSL_HIP_SORTED_THREADS") ? atoi(getenv("SDSL_HIP_SORTED_THREADS")) : 0;
    static const int g_env = getenv("SDSL_HIP_SORTED_G") ? atoi(getenv("SDSL_HIP_SORTED_G")) : 0;
    static const int rb_env = getenv("SDSL_HIP_SORTED_RANK_BLOCKS") ? atoi(getenv("SDSL_HIP_SORTED_RANK_BLOCKS")) : 0;
    static const bool sweep = !(getenv("SDSL_HIP_SORTED_SWEEP") && atoi(getenv("SDSL_HIP_SORTED_SWEEP")) == 0); // 0: histogram pass 2
    // tiles of 8192 keys (512 threads x 16) are the measured optimum on 2^34 bits (profiles/sorted_rank_v6_r02.txt:
    // 17.9 ms against 18.9 ms for 16384-key tiles and 21.4 ms for 4096-key tiles); the others stay selectable for profiling
    // three blocks per CU (768): the un-permute kernels are built for 80 VGPRs (six loop invariants in scratch) so that three fit —
    // 3.96 + 3.24 -> 3.22 + 2.79 ms on one box (SDSL_HIP_SORTED_G=512 for two per CU); the partition of pass 1 stays at 128 VGPRs
    const SrKernels K = t_env == 1024 ? sr_kernels<1024, 16>(1) : (t_env == 256 ? sr_kernels<256, 16>(4) : sr_kernels<512, 16>(3));
    for (uint64_t done = 0; done < n;)
    {
        const uint64_t cnt = n - done < kMaxPass ? n - done : kMaxPass;
        SrGeom g;
        sr_fill_geom(g, v, op, sp, cnt);
        g.tile = K.threads * K.per;
        g.tiles1 = (uint32_t)((cnt + g.tile - 1) / g.tile);
        g.G = g_env >= 1 && g_env <= (int)kMaxG ? (uint32_t)g_env : 256u * K.blocks_per_cu;
        const unsigned bins1 = 1u << g.d1, bins2 = 1u << g.d2, nf = bins1 * bins2;
        SrBuf b;
        if (carve(b, scratch, cnt, g.tile) > scratch_bytes)
        {
            set_error("bucketed batch: scratch too small");
            return SDSL_HIP_ERR_INVALID;
        }
        const uint64_t * idx = d_idx + done;
        const dim3 G(g.G), T(K.threads);
        PhaseTimer pt(trace, s);
        pt.mark();
        if (sweep)
            SH_TRY(fill_u32_async(b.tot2, 0u, (kBins + 1) * 4, s));
        hipLaunchKernelGGL(K.hist1, G, T, 0, s, g, idx, nullptr, nullptr, nullptr, b.counts1, sweep ? b.tot2 : nullptr);
        pt.mark();
        hipLaunchKernelGGL(k_sr_bucket_totals, dim3(bins1), dim3(256), 0, s, g.G, b.counts1, b.btot);
        hipLaunchKernelGGL(k_sr_bucket_scan, dim3(1), dim3(kBins), 0, s, bins1, g.tile, b.btot, b.bstart1, b.tprefix2);
        hipLaunchKernelGGL(k_sr_bucket_offsets, dim3(bins1), dim3(256), 0, s, g.G, b.counts1, b.bstart1, b.offs1);
        pt.mark();
        hipLaunchKernelGGL(K.part1, G, T, 0, s, g, idx, nullptr, nullptr, nullptr, b.offs1, b.keys1, b.slots1, b.thist1);
        pt.mark();
        if (sweep)
        { // pass 2 in one sweep: no histogram pass, offsets by look-back (k_sr_partition2_sweep)
            pt.mark();
            hipLaunchKernelGGL(k_sr_bucket_scan, dim3(1), dim3(kBins), 0, s, bins2, g.tile, b.tot2, b.bstart2, nullptr);
            SH_TRY(fill_u32_async(b.status, 0u, ((size_t)g.tiles1 + kBins) * kBins * 4, s));
            SH_TRY(fill_u32_async(b.ticket, 0u, 4, s));
            SH_TRY(fill_u32_async(b.fstart, 0xFFFFFFFFu, ((size_t)nf + 1) * 4, s));
            pt.mark();
            // three blocks per CU (the 80-VGPR build: a dozen loop invariants live in scratch) cover the look-back's round trips
            // better than two: 4.06 -> 3.92 ms, same box; SDSL_HIP_SWEEP_BLOCKS=0 runs the 128-VGPR build at two per CU
            static const int s3 = getenv("SDSL_HIP_SWEEP_BLOCKS") ? atoi(getenv("SDSL_HIP_SWEEP_BLOCKS")) : 768;
            hipLaunchKernelGGL(s3 > 0 ? K.part2s3 : K.part2s, s3 > 0 ? dim3((unsigned)s3) : G, T, 0, s, g, b.keys1, b.tprefix2, b.bstart1, b.bstart2, b.status, b.ticket, b.keys2, b.slots2,
                               b.thist2, b.offs2, b.fstart);
            pt.mark();
            hipLaunchKernelGGL(k_sr_fine_scan_starts, dim3(1), dim3(1024), 0, s, nf, (unsigned)cnt, b.fstart, b.ioff);
        }
        else
        {
            SH_TRY(fill_u32_async(b.fine_count, 0u, (size_t)nf * 4, s));
            hipLaunchKernelGGL(K.hist2, G, T, 0, s, g, nullptr, b.keys1, b.tprefix2, b.bstart1, b.counts2, b.fine_count);
            pt.mark();
            hipLaunchKernelGGL(k_sr_bucket_totals, dim3(bins2), dim3(256), 0, s, g.G, b.counts2, b.btot);
            hipLaunchKernelGGL(k_sr_bucket_scan, dim3(1), dim3(kBins), 0, s, bins2, g.tile, b.btot, b.bstart2, nullptr);
            hipLaunchKernelGGL(k_sr_bucket_offsets, dim3(bins2), dim3(256), 0, s, g.G, b.counts2, b.bstart2, b.offs2);
            pt.mark();
            hipLaunchKernelGGL(K.part2, G, T, 0, s, g, nullptr, b.keys1, b.tprefix2, b.bstart1, b.offs2, b.keys2, b.slots2,
                               b.thist2);
            pt.mark();
            hipLaunchKernelGGL(k_sr_fine_scan, dim3(1), dim3(1024), 0, s, nf, b.fine_count, b.fstart, b.ioff);
        }
        const unsigned slice_blocks = rb_env >= 1 ? (unsigned)rb_env : 1024u;
        if (op == 0)
        {
            hipLaunchKernelGGL(k_sr_slice_bases, dim3((nf + 255) / 256), dim3(256), 0, s, v, nf, g.d1, g.d2, b.hf);
            pt.mark();
            hipLaunchKernelGGL(k_sr_rank_lds<false>, dim3(slice_blocks), dim3(kRT), 0, s, v, bit, nf, g.d1, g.d2, b.fstart, b.ioff, b.keys2, (const uint32_t *)nullptr, kSliceLog);
        }
        else
        {
            SH_TRY(fill_u32_async(b.btot, 0u, 4, s)); // doubles as the "some answers are left to the fix-up pass" flag
            hipLaunchKernelGGL(k_sr_select_bases, dim3((nf + 255) / 256), dim3(256), 0, s, nf, sp.nf, g.d1, g.d2, sp.bnd, b.hf);
            pt.mark();
            // slices beyond sp.nf are empty (no items), so the kernel never reads bnd past sp.nf
            if (bit)
                hipLaunchKernelGGL(k_sr_select_lds<1>, dim3(slice_blocks), dim3(kST), 0, s, v, nf, g.d1, g.d2, sp.bm << sp.bs, sp.bnd, b.fstart, b.ioff,
                                   b.keys2, b.btot, (const uint32_t *)nullptr);
            else
                hipLaunchKernelGGL(k_sr_select_lds<0>, dim3(slice_blocks), dim3(kST), 0, s, v, nf, g.d1, g.d2, sp.bm << sp.bs, sp.bnd, b.fstart, b.ioff,
                                   b.keys2, b.btot, (const uint32_t *)nullptr);
        }
        pt.mark();
        hipLaunchKernelGGL(K.unp2, G, T, 0, s, b.hf, bit, g, b.tprefix2, b.bstart1, b.offs2, b.keys2, b.btot, b.slots2, b.thist2,
                           b.keys1, nullptr);
        pt.mark();
        hipLaunchKernelGGL(K.unp1, G, T, 0, s, b.hf, bit, g, nullptr, nullptr, b.offs1, b.keys1, nullptr, b.slots1, b.thist1,
                           nullptr, d_out + done);
        pt.mark();
        if (op == 1)
        { // whatever the slices left over (buckets wider than a slice): one lane per marked answer
            if (bit)
                hipLaunchKernelGGL(k_sr_select_fixup<1>, dim3(256 * 8), dim3(256), 0, s, v, b.btot, idx, d_out + done, cnt, (const uint32_t *)nullptr);
            else
                hipLaunchKernelGGL(k_sr_select_fixup<0>, dim3(256 * 8), dim3(256), 0, s, v, b.btot, idx, d_out + done, cnt, (const uint32_t *)nullptr);
        }
        SH_HIP(hipGetLastError());
        if (trace_env)
            pt.report(g, "bucketed (one-sweep)");
        if (trace_opt)
            pt.keep(op);
        done += cnt;
    }
    return SDSL_HIP_OK;
}

__global__ __launch_bounds__(256) void k_sr_bnd_args(unsigned nf, unsigned B, uint64_t * __restrict__ out)
{
    const unsigned f = blockIdx.x * 256 + threadIdx.x;
    if (f < nf)
        out[f] = (uint64_t)f * B + 1; // 1-based rank of the bucket's first argument
}
__global__ __launch_bounds__(256) void k_sr_bnd_lines(unsigned nf, const uint64_t * __restrict__ pos, uint64_t n_lines,
                                                      uint32_t * __restrict__ bnd)
{
    const unsigned f = blockIdx.x * 256 + threadIdx.x;
    if (f < nf)
        bnd[f] = (uint32_t)(pos[f] / kDB);
    if (f == nf)
        bnd[nf] = (uint32_t)(n_lines - 1);
}

} // namespace

// which pipeline: the write-combined one (bv_swc.hip) unless SDSL_HIP_SORTED_SWC=0 asks for the one-sweep passes of this file
static bool sr_use_swc()
{
    static const bool on = !(getenv("SDSL_HIP_SORTED_SWC") && atoi(getenv("SDSL_HIP_SORTED_SWC")) == 0);
    return on;
}

// true when the batch is spread over the vector (see k_sr_sample_spread); synchronises with stream s (one 8-byte read-back).
// `out2`: two device words of scratch.
static sdsl_hip_status sr_batch_is_spread(const BvView & v, int op, const SelectPlan & sp, const uint64_t * d_idx, uint64_t n,
                                          hipStream_t s, uint32_t * out2, bool & spread)
{
    SrGeom g;
    sr_fill_geom(g, v, op, sp, n);
    hipLaunchKernelGGL(k_sr_sample_spread, dim3(1), dim3(1024), 0, s, g, d_idx, out2);
    SH_HIP(hipGetLastError());
    uint32_t hres[2] = {0, 0};
    SH_HIP(hipMemcpyAsync(hres, out2, 8, hipMemcpyDeviceToHost, s));
    SH_HIP(hipStreamSynchronize(s));
    // uniformly random samples over S slices touch S * (1 - exp(-m / S)) of them; half of the samples' own number is far below
    // that for every vector the bucketed path applies to (>= 2^12 slices) and far above what a window or a sorted batch gives
    const unsigned slices = 1u << (g.d1 + g.d2);
    const unsigned limit = hres[1] / 2 < slices / 2 ? hres[1] / 2 : slices / 2;
    spread = hres[1] >= 1024 && hres[0] >= limit;
    return SDSL_HIP_OK;
}

sdsl_hip_status bv_sorted_rank_is_spread(const BvView & v, const uint64_t * d_idx, uint64_t n, hipStream_t s, void * scratch,
                                         bool & spread)
{
    return sr_batch_is_spread(v, 0, SelectPlan(), d_idx, n, s, (uint32_t *)scratch, spread);
}

sdsl_hip_status bv_sorted_select_is_spread(const BvHost & h, int bit, const uint64_t * d_idx, uint64_t n, hipStream_t s, void * scratch,
                                           bool & spread)
{
    const BvHost::SelPlan & P = h.sel_plan[bit];
    SelectPlan sp;
    sp.bnd = P.bnd.as<uint32_t>();
    sp.bm = P.bm;
    sp.bs = P.bs;
    sp.nf = P.nf;
    sp.total = bit ? h.view.ones : h.view.n_bits - h.view.ones;
    return sr_batch_is_spread(h.view, 1, sp, d_idx, n, s, (uint32_t *)scratch, spread);
}

void sr_launch_sample(const SrGeom & g, const uint64_t * d_idx, uint32_t * out3, hipStream_t s)
{
    hipLaunchKernelGGL(k_sr_sample_spread, dim3(1), dim3(1024), 0, s, g, d_idx, out3);
}

// The same sample without the read-back: the verdict (1: spread) lands in out3[2] and the routes enqueued behind it look at it
// themselves (the passes of bv_swc.hip through SrGeom::go, the direct kernels through BvView::skip_if) — nothing synchronises.
bool bv_sorted_device_verdict()
{
    return sr_use_swc();
}

sdsl_hip_status bv_sorted_rank_sample(const BvView & v, const uint64_t * d_idx, uint64_t n, hipStream_t s, uint32_t * out3)
{
    SrGeom g;
    sr_fill_geom(g, v, 0, SelectPlan(), n);
    hipLaunchKernelGGL(k_sr_sample_spread, dim3(1), dim3(1024), 0, s, g, d_idx, out3);
    SH_HIP(hipGetLastError());
    return SDSL_HIP_OK;
}

sdsl_hip_status bv_sorted_select_sample(const BvHost & h, int bit, const uint64_t * d_idx, uint64_t n, hipStream_t s, uint32_t * out3)
{
    const BvHost::SelPlan & P = h.sel_plan[bit];
    SelectPlan sp;
    sp.bnd = P.bnd.as<uint32_t>();
    sp.bm = P.bm;
    sp.bs = P.bs;
    sp.nf = P.nf;
    sp.total = bit ? h.view.ones : h.view.n_bits - h.view.ones;
    SrGeom g;
    sr_fill_geom(g, h.view, 1, sp, n);
    hipLaunchKernelGGL(k_sr_sample_spread, dim3(1), dim3(1024), 0, s, g, d_idx, out3);
    SH_HIP(hipGetLastError());
    return SDSL_HIP_OK;
}

sdsl_hip_status bv_launch_rank_sorted(const BvView & v, int bit, const uint64_t * d_idx, uint64_t n, uint64_t * d_out,
                                      hipStream_t s, void * scratch, size_t scratch_bytes, const uint32_t * go)
{
    if (!bv_sorted_rank_possible(v))
    {
        set_error("rank_sorted: vector too large for the bucketed path");
        return SDSL_HIP_ERR_INVALID;
    }
    if (sr_use_swc())
        return sw_run(v, 0, bit, SelectPlan{}, d_idx, n, d_out, s, scratch, scratch_bytes, go);
    return sr_run(v, 0, bit, SelectPlan{}, d_idx, n, d_out, s, scratch, scratch_bytes);
}

// Buckets of the bucketed select: B consecutive argument ranks each, at most 2^16 of them, sized so that a bucket of a
// uniformly dense vector spans about 768 lines; bnd[f] = line of the bucket's first argument.  Built once per handle and bit
// value (one small batch through the direct kernel).  wide_frac = share of the arguments that live in buckets spanning more
// than an LDS slice — those are answered by the fix-up pass, so the path only pays when the share is small.
sdsl_hip_status bv_select_sorted_prepare(BvHost & h, int bit)
{
    if (h.sel_plan[bit].ready)
        return SDSL_HIP_OK;
    BvHost::SelPlan & P = h.sel_plan[bit];
    const BvView & v = h.view;
    const uint64_t total = bit ? v.ones : v.n_bits - v.ones;
    // (ready only once the outcome is definitive: a failed allocation below leaves the plan to be tried again, and the caller
    // on the direct kernel meanwhile)
    P.ok = false;
    if (!v.sel[bit] || total < 2 || v.n_lines > UINT64_C(0xFFFFFFFF))
    {
        P.ready = true;
        return SDSL_HIP_OK;
    }
    // B = m << sh, m in 8..15: the largest such value whose bucket spans about 3/4 of a slice at the vector's mean density,
    // but at least total / 2^16 (two 8-bit digits address the buckets)
    auto round_down = [](uint64_t x, unsigned & m, unsigned & sh) { // largest m << sh <= x (x >= 64)
        sh = 0;
        while ((x >> sh) > 15)
            ++sh;
        m = (unsigned)(x >> sh);
    };
    const double per = 768.0 * (double)kDB * ((double)total / (double)v.n_bits); // arguments in 768 lines
    const uint64_t b_fit = per < 64.0 ? 64 : (per > 15.0 * 1048576.0 ? (uint64_t)15 << 20 : (uint64_t)per);
    const uint64_t b_min = (total + 65535) >> 16;
    unsigned bm, bs;
    round_down(b_fit, bm, bs);
    if (((uint64_t)bm << bs) < b_min)
    { // smallest m << sh >= b_min
        round_down(b_min, bm, bs);
        if (((uint64_t)bm << bs) < b_min && ++bm == 16)
            bm = 8, ++bs;
    }
    if (bs > 20)
    {
        P.ready = true;
        return SDSL_HIP_OK;
    }
    const uint64_t B = (uint64_t)bm << bs;
    const unsigned nf = (unsigned)((total + B - 1) / B);
    DevBuf args, pos;
    if (args.alloc((size_t)nf * 8) != SDSL_HIP_OK || pos.alloc((size_t)nf * 8) != SDSL_HIP_OK || P.bnd.alloc(((size_t)nf + 1) * 4) != SDSL_HIP_OK)
    { // no room right now: not an error of the query — it takes the direct kernel, and the plan is tried again next time
        P.bnd.release();
        return SDSL_HIP_OK;
    }
    struct BndGuard // a failure below must not leave a half-made plan's memory behind
    {
        DevBuf & b;
        bool keep = false;
        ~BndGuard()
        {
            if (!keep)
                b.release();
        }
    } bnd_guard{P.bnd};
    hipLaunchKernelGGL(k_sr_bnd_args, dim3((nf + 255) / 256), dim3(256), 0, 0, nf, (unsigned)B, args.as<uint64_t>());
    SH_HIP(hipGetLastError());
    {
        TimingPause pause;
        SH_TRY(bv_launch_select(v, bit, args.as<uint64_t>(), nf, pos.as<uint64_t>(), nullptr));
    }
    hipLaunchKernelGGL(k_sr_bnd_lines, dim3((nf + 256) / 256), dim3(256), 0, 0, nf, pos.as<uint64_t>(), v.n_lines,
                       P.bnd.as<uint32_t>());
    SH_HIP(hipGetLastError());
    std::vector<uint32_t> hb((size_t)nf + 1);
    SH_HIP(hipMemcpy(hb.data(), P.bnd.p, hb.size() * 4, hipMemcpyDeviceToHost));
    uint64_t wide = 0;
    for (unsigned f = 0; f < nf; ++f)
        if ((uint64_t)hb[f + 1] + 1 - hb[f] > (UINT64_C(1) << kSliceLog))
            wide += std::min<uint64_t>(B, total - (uint64_t)f * B);
    P.bm = bm;
    P.bs = bs;
    P.nf = nf;
    P.wide_frac = (double)wide / (double)total;
    P.ok = true;
    P.ready = true;
    bnd_guard.keep = true;
    return SDSL_HIP_OK;
}

bool bv_sorted_select_applicable(const BvHost & h, int bit, uint64_t n)
{
    const BvHost::SelPlan & P = h.sel_plan[bit];
    return P.ready && P.ok && P.wide_frac <= 0.01 && h.view.n_lines >= (UINT64_C(1) << 22) && n >= 2 * h.view.n_lines;
}

sdsl_hip_status bv_launch_select_sorted(BvHost & h, int bit, const uint64_t * d_i, uint64_t n, uint64_t * d_out, hipStream_t s,
                                        void * scratch, size_t scratch_bytes, const uint32_t * go)
{
    const BvHost::SelPlan & P = h.sel_plan[bit];
    if (!P.ready || !P.ok)
    {
        set_error("select_sorted: no bucket plan for this vector");
        return SDSL_HIP_ERR_INVALID;
    }
    SelectPlan sp;
    sp.bnd = P.bnd.as<uint32_t>();
    sp.bm = P.bm;
    sp.bs = P.bs;
    sp.nf = P.nf;
    sp.total = bit ? h.view.ones : h.view.n_bits - h.view.ones;
    if (sr_use_swc())
        return sw_run(h.view, 1, bit, sp, d_i, n, d_out, s, scratch, scratch_bytes, go);
    return sr_run(h.view, 1, bit, sp, d_i, n, d_out, s, scratch, scratch_bytes);
}

} // namespace sdslhip
