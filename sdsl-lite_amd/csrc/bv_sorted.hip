// bv_sorted.hip — batched rank on a plain bit vector for LARGE batches: the batch is partitioned by the region of
// the index it addresses, answered region by region out of the XCD's L2, and put back into the caller's order.
//
// Why: a random rank costs one L2 miss = one 128-byte fabric read, and the part serves ≈ 40 G of
// those per second whatever the kernel does (profiles/gather_probe_r01.txt).  A batch of 10^9 positions over 2^34 bits
// addresses every 64-byte rank line ≈ 26 times; fetched in the caller's order that is 26 misses, fetched bucket by
// bucket it is one miss and 25 L2 hits.  The reference answers its queries one at a time
// (rank_support_v5.hpp:131-149); batching is this library's addition and the answers are the same numbers.
//
// Pipeline (all passes stream; every global access is coalesced or a run):
//   1. k_sr_hist      block g counts, per bucket, the positions of its contiguous share of the batch
//   2. k_sr_bucket_*  exclusive scan over (bucket, block) -> where block g's entries of bucket b start
//   3. k_sr_partition per tile of 16384 positions: counting sort by bucket in LDS, the sorted tile is written out
//                     bucket run by bucket run as 32-bit keys (line inside the bucket, bit inside the line); every
//                     position remembers its 16-bit slot in the sorted tile; the tile's histogram is kept (u16)
//   4. k_sr_rank      sorted keys -> 32-bit answers relative to the bucket's first line, in place; XCD x walks the
//                     x-th eighth of the sorted array, so a bucket's slice (2^14 lines = 1 MiB) lives in ONE L2
//   5. k_sr_unpermute per tile: the runs are gathered back into LDS (coalesced), made absolute, and every position
//                     picks its answer by slot; the result array is written in the caller's order, coalesced
// No pass scatters single words: the un-permute that sank the idea in round 1 (random 8-byte scatter, 22.9 G/s,
// profiles/scatter_probe_r01.txt) is a gather of runs staged through LDS.
#include "bv_host.hpp"

namespace sdslhip {

namespace {

constexpr unsigned kSrThreads = 1024;            // partition / un-permute block
constexpr unsigned kSrPer = 16;                  // positions per thread per tile
constexpr unsigned kSrTile = kSrThreads * kSrPer; // 16384 (slots fit 16 bits)
constexpr unsigned kSrBMax = 3 * kSrThreads;     // buckets a block can scan (3 per thread)
constexpr uint32_t kSrBad = 0xFFFFFFFFu;         // key / answer of a position beyond the vector (answer NPOS)
constexpr unsigned kSrOffBits = 9;               // 448 < 2^9 in-line offsets

struct SrGeom
{
    uint32_t k;        // log2(lines per bucket)
    uint32_t B;        // buckets
    uint32_t G;        // partition blocks
    uint64_t tiles;    // tiles in the batch
    uint64_t n;        // positions in the batch (< 2^31)
    bool small;        // 32-bit division path of line_of
};

__device__ __forceinline__ unsigned wave_incl_scan(unsigned v)
{
#pragma unroll
    for (int d = 1; d < 64; d <<= 1)
    {
        unsigned u = __shfl_up(v, d, 64);
        if ((int)(threadIdx.x & 63) >= d)
            v += u;
    }
    return v;
}

// In-place exclusive scan of a[0 .. 3*kSrThreads) in LDS by a block of kSrThreads; returns the total.  `wsum` is
// scratch for 16 wave totals + 1.  Ends with a barrier.
__device__ __forceinline__ unsigned block_excl_scan3(unsigned * a, unsigned * wsum)
{
    const unsigned t = threadIdx.x;
    const unsigned a0 = a[3 * t], a1 = a[3 * t + 1], a2 = a[3 * t + 2];
    const unsigned s = a0 + a1 + a2;
    const unsigned inc = wave_incl_scan(s);
    if ((t & 63) == 63)
        wsum[t >> 6] = inc;
    __syncthreads();
    if (t < 64)
    {
        unsigned w = t < kSrThreads / 64 ? wsum[t] : 0;
        unsigned wi = wave_incl_scan(w);
        if (t < kSrThreads / 64)
            wsum[t] = wi - w;
        if (t == kSrThreads / 64 - 1)
            wsum[kSrThreads / 64] = wi;
    }
    __syncthreads();
    const unsigned base = wsum[t >> 6] + inc - s;
    a[3 * t] = base;
    a[3 * t + 1] = base + a0;
    a[3 * t + 2] = base + a0 + a1;
    const unsigned total = wsum[kSrThreads / 64];
    __syncthreads();
    return total;
}

// bucket and key of a position
__device__ __forceinline__ void sr_key(uint64_t pos, uint64_t n_bits, const SrGeom & g, unsigned & b, uint32_t & key)
{
    if (pos > n_bits)
    {
        b = 0;
        key = kSrBad;
        return;
    }
    uint64_t L;
    unsigned off;
    line_of(pos, g.small, L, off);
    b = (unsigned)(L >> g.k);
    key = ((uint32_t)(L & ((UINT64_C(1) << g.k) - 1)) << kSrOffBits) | off;
}

// tiles [lo, hi) of block gi
__device__ __forceinline__ void sr_share(const SrGeom & g, unsigned gi, uint64_t & lo, uint64_t & hi)
{
    lo = g.tiles * gi / g.G;
    hi = g.tiles * (gi + 1) / g.G;
}

// ---- 1. histogram -----------------------------------------------------------------------------------------------
__global__ __launch_bounds__(kSrThreads) void k_sr_hist(uint64_t n_bits, SrGeom g, const uint64_t * __restrict__ idx,
                                                        uint32_t * __restrict__ counts /* [B][G] */)
{
    __shared__ unsigned hist[kSrBMax];
    for (unsigned i = threadIdx.x; i < kSrBMax; i += kSrThreads)
        hist[i] = 0;
    __syncthreads();
    uint64_t tlo, thi;
    sr_share(g, blockIdx.x, tlo, thi);
    const uint64_t qlo = tlo * kSrTile, qhi = thi * kSrTile < g.n ? thi * kSrTile : g.n;
    for (uint64_t q0 = qlo + threadIdx.x; q0 < qhi; q0 += (uint64_t)kSrThreads * 4)
    {
        uint64_t p[4];
#pragma unroll
        for (int u = 0; u < 4; ++u)
        {
            const uint64_t q = q0 + (uint64_t)u * kSrThreads;
            p[u] = q < qhi ? __builtin_nontemporal_load(idx + q) : ~UINT64_C(0);
        }
#pragma unroll
        for (int u = 0; u < 4; ++u)
        {
            const uint64_t q = q0 + (uint64_t)u * kSrThreads;
            if (q < qhi)
            {
                unsigned b;
                uint32_t key;
                sr_key(p[u], n_bits, g, b, key);
                atomicAdd(&hist[b], 1u);
            }
        }
    }
    __syncthreads();
    for (unsigned b = threadIdx.x; b < g.B; b += kSrThreads)
        counts[(uint64_t)b * g.G + blockIdx.x] = hist[b];
}

// ---- 2. offsets: offs[b][g] = entries of buckets < b + entries of bucket b in blocks < g ---------------------------
__global__ __launch_bounds__(256) void k_sr_bucket_totals(SrGeom g, const uint32_t * __restrict__ counts,
                                                          uint32_t * __restrict__ btot)
{
    __shared__ unsigned red[4];
    const unsigned b = blockIdx.x;
    unsigned s = 0;
    for (unsigned i = threadIdx.x; i < g.G; i += 256)
        s += counts[(uint64_t)b * g.G + i];
    for (int d = 32; d; d >>= 1)
        s += __shfl_xor(s, d, 64);
    if ((threadIdx.x & 63) == 0)
        red[threadIdx.x >> 6] = s;
    __syncthreads();
    if (threadIdx.x == 0)
        btot[b] = red[0] + red[1] + red[2] + red[3];
}

__global__ __launch_bounds__(kSrThreads) void k_sr_bucket_scan(SrGeom g, const uint32_t * __restrict__ btot,
                                                               uint32_t * __restrict__ bstart /* B + 1 */)
{
    __shared__ unsigned a[kSrBMax];
    __shared__ unsigned wsum[kSrThreads / 64 + 1];
    for (unsigned i = threadIdx.x; i < kSrBMax; i += kSrThreads)
        a[i] = i < g.B ? btot[i] : 0;
    __syncthreads();
    const unsigned total = block_excl_scan3(a, wsum);
    for (unsigned i = threadIdx.x; i < g.B; i += kSrThreads)
        bstart[i] = a[i];
    if (threadIdx.x == 0)
        bstart[g.B] = total;
}

__global__ __launch_bounds__(256) void k_sr_bucket_offsets(SrGeom g, const uint32_t * __restrict__ counts,
                                                           const uint32_t * __restrict__ bstart,
                                                           uint32_t * __restrict__ offs)
{ // G <= 256 * 4: a thread owns 4 consecutive blocks
    __shared__ unsigned wtot[5];
    const unsigned b = blockIdx.x, t = threadIdx.x;
    unsigned c[4], s = 0;
#pragma unroll
    for (int u = 0; u < 4; ++u)
    {
        const unsigned gi = 4 * t + u;
        c[u] = gi < g.G ? counts[(uint64_t)b * g.G + gi] : 0;
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
    for (int u = 0; u < 4; ++u)
    {
        const unsigned gi = 4 * t + u;
        if (gi < g.G)
            offs[(uint64_t)b * g.G + gi] = base;
        base += c[u];
    }
}

// runs of a sorted tile <-> the bucket-major array.  8 lanes per bucket; a bucket with more than kSrBigRun entries in
// this tile (skewed batches) is left to the whole block afterwards.  COPY(b, src_in_tile, dst_global, count, lane, step)
constexpr unsigned kSrBigRun = 64;

// ---- 3. partition -----------------------------------------------------------------------------------------------
__global__ __launch_bounds__(kSrThreads) void k_sr_partition(uint64_t n_bits, SrGeom g, const uint64_t * __restrict__ idx,
                                                             const uint32_t * __restrict__ offs,
                                                             uint32_t * __restrict__ keys, uint16_t * __restrict__ slots,
                                                             uint16_t * __restrict__ tile_hist /* [tiles][B] */)
{
    __shared__ uint32_t sorted[kSrTile];
    __shared__ unsigned hist[kSrBMax];  // counts of the tile, kept beside their scan
    __shared__ unsigned start[kSrBMax]; // exclusive scan
    __shared__ unsigned cursor[kSrBMax];
    __shared__ unsigned wsum[kSrThreads / 64 + 1];
    __shared__ unsigned big[256], n_big;
    const unsigned t = threadIdx.x;
    for (unsigned b = t; b < kSrBMax; b += kSrThreads)
        cursor[b] = b < g.B ? offs[(uint64_t)b * g.G + blockIdx.x] : 0;
    uint64_t tlo, thi;
    sr_share(g, blockIdx.x, tlo, thi);
    for (uint64_t tile = tlo; tile < thi; ++tile)
    {
        for (unsigned b = t; b < kSrBMax; b += kSrThreads)
            hist[b] = 0;
        if (t == 0)
            n_big = 0;
        __syncthreads();
        const uint64_t q0 = tile * kSrTile + t;
        uint64_t p[kSrPer];
#pragma unroll
        for (unsigned u = 0; u < kSrPer; ++u)
        {
            const uint64_t q = q0 + (uint64_t)u * kSrThreads;
            p[u] = q < g.n ? __builtin_nontemporal_load(idx + q) : ~UINT64_C(0);
        }
        uint32_t key[kSrPer], br[kSrPer]; // br = bucket << 16 | rank inside the tile's share of the bucket
#pragma unroll
        for (unsigned u = 0; u < kSrPer; ++u)
        {
            const uint64_t q = q0 + (uint64_t)u * kSrThreads;
            unsigned b = 0;
            key[u] = kSrBad;
            br[u] = 0;
            if (q < g.n)
            {
                sr_key(p[u], n_bits, g, b, key[u]);
                const unsigned r = atomicAdd(&hist[b], 1u);
                br[u] = (b << 16) | r; // r < 16384, b < 3072
            }
        }
        __syncthreads();
        for (unsigned b = t; b < kSrBMax; b += kSrThreads)
            start[b] = hist[b];
        __syncthreads();
        block_excl_scan3(start, wsum);
        for (unsigned b = t; b < g.B; b += kSrThreads)
            tile_hist[tile * g.B + b] = (uint16_t)hist[b];
#pragma unroll
        for (unsigned u = 0; u < kSrPer; ++u)
        {
            const uint64_t q = q0 + (uint64_t)u * kSrThreads;
            if (q < g.n)
            {
                const unsigned pos = start[br[u] >> 16] + (br[u] & 0xFFFFu);
                sorted[pos] = key[u];
                __builtin_nontemporal_store((uint16_t)pos, slots + q);
            }
        }
        __syncthreads();
        { // runs out: 8 lanes per bucket
            const unsigned l = t & 7;
            for (unsigned b = t >> 3; b < g.B; b += kSrThreads / 8)
            {
                const unsigned cnt = hist[b];
                if (cnt == 0)
                    continue;
                const unsigned st = start[b], cur = cursor[b];
                if (cnt > kSrBigRun)
                {
                    if (l == 0)
                        big[atomicAdd(&n_big, 1u)] = b; // at most kSrTile / (kSrBigRun + 1) = 252 of them
                    continue;
                }
                for (unsigned i = l; i < cnt; i += 8)
                    keys[(uint64_t)cur + i] = sorted[st + i];
            }
        }
        __syncthreads();
        const unsigned nb = n_big;
        for (unsigned k = 0; k < nb; ++k)
        {
            const unsigned b = big[k], cnt = hist[b], st = start[b], cur = cursor[b];
            for (unsigned i = t; i < cnt; i += kSrThreads)
                keys[(uint64_t)cur + i] = sorted[st + i];
        }
        __syncthreads();
        for (unsigned b = t; b < g.B; b += kSrThreads)
            cursor[b] += hist[b];
        __syncthreads();
    }
}

// ---- 4. rank over the sorted keys, in place ------------------------------------------------------------------------
// 256 threads = 64 quads.  XCD x (= blockIdx % 8, the observed placement; any placement is correct) takes the x-th
// eighth of the sorted array in chunks of kSrChunk, its blocks interleaved, so the blocks of an XCD stay within a
// window of a few hundred thousand keys = one or two buckets = 1-2 MiB of index in that XCD's 4 MiB L2.
constexpr unsigned kSrChunk = 2048;

template <int U>
__global__ __launch_bounds__(256) void k_sr_rank(BvView bv, int bit, SrGeom g, const uint32_t * __restrict__ bstart,
                                                 uint32_t * __restrict__ keys)
{
    __shared__ unsigned sh_b;
    const int s = threadIdx.x & (kG - 1);
    const unsigned gq = threadIdx.x / kG;
    const unsigned xcd = blockIdx.x & 7, j = blockIdx.x >> 3, nbx = gridDim.x >> 3;
    const uint64_t n_chunks = (g.n + kSrChunk - 1) / kSrChunk;
    const uint64_t c_lo = n_chunks * xcd / 8, c_hi = n_chunks * (xcd + 1) / 8;
    for (uint64_t c = c_lo + j; c < c_hi; c += nbx)
    {
        uint64_t lo = c * kSrChunk;
        const uint64_t chi = lo + kSrChunk < g.n ? lo + kSrChunk : g.n;
        if (threadIdx.x == 0)
        { // last bucket whose start is <= lo
            unsigned a = 0, z = g.B; // invariant: bstart[a] <= lo, bstart[z] > lo or z == B
            while (z - a > 1)
            {
                const unsigned m = (a + z) >> 1;
                if (bstart[m] <= lo)
                    a = m;
                else
                    z = m;
            }
            sh_b = a;
        }
        __syncthreads();
        unsigned b = sh_b;
        __syncthreads();
        while (lo < chi)
        {
            const uint64_t bend = bstart[b + 1];
            const uint64_t hi = bend < chi ? bend : chi;
            if (hi > lo)
            {
                const uint64_t L0 = (uint64_t)b << g.k;
                const uint64_t H = bv.lines[L0 * kLW];
                for (uint64_t base = lo; base < hi; base += 64 * U)
                {
                    uint32_t key[U];
                    Pair w[U];
#pragma unroll
                    for (int u = 0; u < U; ++u)
                    {
                        const uint64_t q = base + (uint64_t)u * 64 + gq;
                        key[u] = q < hi ? __builtin_nontemporal_load(keys + q) : kSrBad;
                    }
#pragma unroll
                    for (int u = 0; u < U; ++u)
                        w[u] = load_pair<false>(bv.lines, key[u] == kSrBad ? L0 : L0 + (key[u] >> kSrOffBits), s);
#pragma unroll
                    for (int u = 0; u < U; ++u)
                    {
                        const uint64_t q = base + (uint64_t)u * 64 + gq;
                        const unsigned off = key[u] & ((1u << kSrOffBits) - 1);
                        const uint64_t r1 = quad_rank1_at(w[u], s, off) - H;
                        uint32_t r = (uint32_t)r1;
                        if (!bit)
                            r = (uint32_t)((uint64_t)(key[u] >> kSrOffBits) * kDB + off - r1);
                        if (key[u] == kSrBad)
                            r = kSrBad;
                        if (s == 0 && q < hi)
                            __builtin_nontemporal_store(r, keys + q);
                    }
                }
            }
            lo = hi;
            if (lo < chi)
                ++b;
        }
    }
}

// ---- 5. back into the caller's order --------------------------------------------------------------------------------
__global__ __launch_bounds__(kSrThreads) void k_sr_unpermute(BvView bv, int bit, SrGeom g,
                                                             const uint32_t * __restrict__ offs,
                                                             const uint32_t * __restrict__ res,
                                                             const uint16_t * __restrict__ slots,
                                                             const uint16_t * __restrict__ tile_hist,
                                                             uint64_t * __restrict__ out)
{
    __shared__ uint32_t lo32[kSrTile];
    __shared__ uint8_t hi8[kSrTile];
    __shared__ unsigned hist[kSrBMax];
    __shared__ unsigned start[kSrBMax];
    __shared__ unsigned cursor[kSrBMax];
    __shared__ uint64_t hb[kSrBMax]; // what turns a bucket-relative answer into the absolute one
    __shared__ unsigned wsum[kSrThreads / 64 + 1];
    __shared__ unsigned big[256], n_big;
    const unsigned t = threadIdx.x;
    for (unsigned b = t; b < kSrBMax; b += kSrThreads)
    {
        cursor[b] = b < g.B ? offs[(uint64_t)b * g.G + blockIdx.x] : 0;
        uint64_t h = 0;
        if (b < g.B)
        {
            const uint64_t L0 = (uint64_t)b << g.k;
            h = bv.lines[L0 * kLW];
            if (!bit)
                h = L0 * kDB - h;
        }
        hb[b] = h;
    }
    uint64_t tlo, thi;
    sr_share(g, blockIdx.x, tlo, thi);
    for (uint64_t tile = tlo; tile < thi; ++tile)
    {
        for (unsigned b = t; b < kSrBMax; b += kSrThreads)
        {
            const unsigned c = b < g.B ? tile_hist[tile * g.B + b] : 0;
            hist[b] = c;
            start[b] = c;
        }
        if (t == 0)
            n_big = 0;
        __syncthreads();
        block_excl_scan3(start, wsum);
        auto put = [&](unsigned b, unsigned st, unsigned cur, unsigned i)
        {
            const uint32_t v = res[(uint64_t)cur + i];
            const uint64_t full = hb[b] + v;
            lo32[st + i] = (uint32_t)full;
            hi8[st + i] = v == kSrBad ? (uint8_t)0xFF : (uint8_t)(full >> 32);
        };
        {
            const unsigned l = t & 7;
            for (unsigned b = t >> 3; b < g.B; b += kSrThreads / 8)
            {
                const unsigned cnt = hist[b];
                if (cnt == 0)
                    continue;
                const unsigned st = start[b], cur = cursor[b];
                if (cnt > kSrBigRun)
                {
                    if (l == 0)
                        big[atomicAdd(&n_big, 1u)] = b;
                    continue;
                }
                for (unsigned i = l; i < cnt; i += 8)
                    put(b, st, cur, i);
            }
        }
        __syncthreads();
        const unsigned nb = n_big;
        for (unsigned k = 0; k < nb; ++k)
        {
            const unsigned b = big[k], cnt = hist[b], st = start[b], cur = cursor[b];
            for (unsigned i = t; i < cnt; i += kSrThreads)
                put(b, st, cur, i);
        }
        __syncthreads();
        const uint64_t q0 = tile * kSrTile + t;
        uint16_t sl[kSrPer];
#pragma unroll
        for (unsigned u = 0; u < kSrPer; ++u)
        {
            const uint64_t q = q0 + (uint64_t)u * kSrThreads;
            sl[u] = q < g.n ? __builtin_nontemporal_load(slots + q) : (uint16_t)0;
        }
#pragma unroll
        for (unsigned u = 0; u < kSrPer; ++u)
        {
            const uint64_t q = q0 + (uint64_t)u * kSrThreads;
            if (q < g.n)
            {
                const unsigned h = hi8[sl[u]];
                const uint64_t r = h == 0xFFu ? SDSL_HIP_NPOS : ((uint64_t)h << 32) | lo32[sl[u]];
                __builtin_nontemporal_store(r, out + q);
            }
        }
        for (unsigned b = t; b < g.B; b += kSrThreads)
            cursor[b] += hist[b];
        __syncthreads();
    }
}

struct PhaseTimer
{
    bool on;
    hipStream_t s;
    hipEvent_t ev[8];
    int n = 0;
    PhaseTimer(bool on_, hipStream_t s_) : on(on_), s(s_)
    {
        if (on)
            for (auto & e : ev)
                (void)hipEventCreate(&e);
    }
    void mark()
    {
        if (on && n < 8)
            (void)hipEventRecord(ev[n++], s);
    }
    void report(const SrGeom & g)
    {
        if (!on)
            return;
        (void)hipEventSynchronize(ev[n - 1]);
        static const char * name[] = {"hist", "offsets", "partition", "rank", "unpermute"};
        float total = 0;
        fprintf(stderr, "[sdsl_hip] sorted rank: n=%llu B=%u k=%u G=%u |", (unsigned long long)g.n, g.B, g.k, g.G);
        for (int i = 0; i + 1 < n; ++i)
        {
            float ms = 0;
            (void)hipEventElapsedTime(&ms, ev[i], ev[i + 1]);
            total += ms;
            fprintf(stderr, " %s %.3f ms", name[i], ms);
        }
        fprintf(stderr, " | total %.3f ms = %.2f G/s\n", total, g.n / total / 1e6);
    }
    ~PhaseTimer()
    {
        if (on)
            for (auto & e : ev)
                (void)hipEventDestroy(e);
    }
};

} // namespace

// scratch per position: 4 (key / answer) + 2 (slot) bytes, + 2 B per (tile, bucket) + the offset tables
size_t bv_sorted_rank_scratch_bytes(const BvView & v, uint64_t n)
{
    (void)v;
    const uint64_t tiles = (n + kSrTile - 1) / kSrTile;
    return (size_t)(n * 6 + tiles * kSrBMax * 2 + (size_t)kSrBMax * 1024 * 8 + (1u << 16));
}

bool bv_sorted_rank_applicable(const BvView & v, uint64_t n)
{
    // worth it when the index is much larger than the L2s (else the direct kernel already hits) and the batch
    // addresses every line several times
    return v.n_lines >= (UINT64_C(1) << 19) && n >= (UINT64_C(1) << 24) && n >= 4 * v.n_lines;
}

sdsl_hip_status bv_launch_rank_sorted(const BvView & v, int bit, const uint64_t * d_idx, uint64_t n, uint64_t * d_out,
                                      hipStream_t s, void * scratch, size_t scratch_bytes)
{
    static const bool trace = getenv("SDSL_HIP_TRACE_SORTED") != nullptr;
    static const int k_env = getenv("SDSL_HIP_SORTED_K") ? atoi(getenv("SDSL_HIP_SORTED_K")) : 0;
    static const int g_env = getenv("SDSL_HIP_SORTED_G") ? atoi(getenv("SDSL_HIP_SORTED_G")) : 0;
    static const int rb_env = getenv("SDSL_HIP_SORTED_RANK_BLOCKS") ? atoi(getenv("SDSL_HIP_SORTED_RANK_BLOCKS")) : 0;
    const uint64_t kMaxPass = UINT64_C(1) << 30; // positions per pass (32-bit cursors)
    for (uint64_t done = 0; done < n;)
    {
        const uint64_t cnt = n - done < kMaxPass ? n - done : kMaxPass;
        SrGeom g;
        g.k = k_env >= 8 && k_env <= 22 ? (uint32_t)k_env : 14;
        while (((v.n_lines + (UINT64_C(1) << g.k) - 1) >> g.k) > kSrBMax)
            ++g.k;
        if (g.k > 22)
        {
            set_error("rank_sorted: vector too large for the bucketed path");
            return SDSL_HIP_ERR_INVALID;
        }
        g.B = (uint32_t)((v.n_lines + (UINT64_C(1) << g.k) - 1) >> g.k);
        g.n = cnt;
        g.tiles = (cnt + kSrTile - 1) / kSrTile;
        g.G = g_env >= 1 && g_env <= 1024 ? (uint32_t)g_env : 256;
        if (g.G > g.tiles)
            g.G = (uint32_t)g.tiles;
        g.small = v.n_bits < (UINT64_C(1) << 38);
        // carve the scratch
        uint8_t * p = (uint8_t *)scratch;
        auto take = [&](size_t bytes) -> void *
        {
            void * r = p;
            p += (bytes + 255) & ~(size_t)255;
            return r;
        };
        uint32_t * keys = (uint32_t *)take(cnt * 4);
        uint16_t * slots = (uint16_t *)take(cnt * 2);
        uint16_t * tile_hist = (uint16_t *)take(g.tiles * g.B * 2);
        uint32_t * counts = (uint32_t *)take((size_t)g.B * g.G * 4);
        uint32_t * offs = (uint32_t *)take((size_t)g.B * g.G * 4);
        uint32_t * btot = (uint32_t *)take((size_t)(g.B + 1) * 4);
        uint32_t * bstart = (uint32_t *)take((size_t)(g.B + 1) * 4);
        if ((size_t)(p - (uint8_t *)scratch) > scratch_bytes)
        {
            set_error("rank_sorted: scratch too small");
            return SDSL_HIP_ERR_INVALID;
        }
        PhaseTimer pt(trace, s);
        pt.mark();
        hipLaunchKernelGGL(k_sr_hist, dim3(g.G), dim3(kSrThreads), 0, s, v.n_bits, g, d_idx + done, counts);
        pt.mark();
        hipLaunchKernelGGL(k_sr_bucket_totals, dim3(g.B), dim3(256), 0, s, g, counts, btot);
        hipLaunchKernelGGL(k_sr_bucket_scan, dim3(1), dim3(kSrThreads), 0, s, g, btot, bstart);
        hipLaunchKernelGGL(k_sr_bucket_offsets, dim3(g.B), dim3(256), 0, s, g, counts, bstart, offs);
        pt.mark();
        hipLaunchKernelGGL(k_sr_partition, dim3(g.G), dim3(kSrThreads), 0, s, v.n_bits, g, d_idx + done, offs, keys, slots,
                           tile_hist);
        pt.mark();
        const unsigned rank_blocks = rb_env >= 8 ? (unsigned)rb_env & ~7u : 256u * 8u;
        hipLaunchKernelGGL((k_sr_rank<4>), dim3(rank_blocks), dim3(256), 0, s, v, bit, g, bstart, keys);
        pt.mark();
        hipLaunchKernelGGL(k_sr_unpermute, dim3(g.G), dim3(kSrThreads), 0, s, v, bit, g, offs, keys, slots, tile_hist,
                           d_out + done);
        pt.mark();
        SH_HIP(hipGetLastError());
        pt.report(g);
        done += cnt;
    }
    return SDSL_HIP_OK;
}

} // namespace sdslhip
