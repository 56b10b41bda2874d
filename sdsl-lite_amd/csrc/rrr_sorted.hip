// rrr_sorted.hip — large batches of rank on rrr_vector<63>: the passes of bv_swc.hip around an answering kernel of its own.
//
// The direct kernel (rrr.hip, k_rrr_rank) fetches one 128-byte record per query and runs the block decoder once per query: on
// 2^34 bits at 5 % density it sits at the part's random-fetch ceiling (33 G/s, 1.2 fabric requests per query,
// profiles/bench_r03_pmc.md).  A batch that is partitioned by slice of the record array reads every record once — and, what
// decides it here, DECODES EVERY BLOCK ONCE: a slice of 2^7 records (16 KiB, 4352 blocks) is turned into its plain 63-bit
// blocks in LDS, and every key of the slice is then a table read and a popcount.  Round 2 tried the partition with the
// decoder left per key and lost (decoder-bound: a wave pays for its lane with the most set bits, ~800 wave instructions per 64
// keys, profiles/rrr_bucketed_rank_r02.txt).  Per slice the blocks are first sorted by class (a counting sort in LDS), so the
// lanes of a wave decode blocks of ONE class: the decoder's cost is the mean class, not the maximum over 64 lanes.
// Answers are rank_support_rrr<1,63>::rank's (rrr_vector.hpp:503-544); batching is this library's addition.
#include <mutex>

#include "bv_sorted_dev.hpp"
#include "rrr_device.hpp"
#include "rrr_host.hpp"

namespace sdslhip {

namespace {

static_assert(kSrRecBits == RrrFmtW::SB && kSrRecBitsSlim == RrrFmtS::SB, "bv_sorted_dev.hpp: kSrRecBits* must be the record lengths of rrr_device.hpp");
constexpr unsigned kRsT = 1024;      // threads of an answering block
constexpr unsigned kRsCol0 = 3;      // first binomial column the sparse decoder reads (the last two set bits come in closed form)
// ... and how many: classes 3..10 after the complement for a vector whose enumerative classes end at 10 (every vector until round 5), 3..20
// for one that keeps the classes up to 20 enumerative (option "rrr_sparse_limit", the default of stand-alone vectors since round 6:
// 10-30 % density at SDSL's space).  5 KiB more of LDS; a wide-record slice then takes 76 KiB and two answering blocks still fit a CU.
__host__ __device__ constexpr unsigned rs_cols(unsigned sparse_max)
{
    return sparse_max > 10 ? 18u : 8u;
}
// C(63, k) is needed for the complemented classes only: k >= 63 - sparse_max
__host__ __device__ constexpr unsigned rs_top0(unsigned sparse_max)
{
    return sparse_max > 10 ? 43u : 53u;
}
constexpr unsigned kRsBins = 16;     // decode-cost classes of the per-slice counting sort

// Records may be decoded in CHUNKS of a slice (what only the decoder needs is then sized for a chunk of NC blocks, what the keys
// read for the slice).  Slim records (42 blocks: 5376 per slice) took two chunks to leave room for two answering blocks per CU,
// at the price of a second round of barriers and of exposed fetches (5.35 ms against the wide format's 4.65); with the class fields
// in the spare bits of the offset positions (slim classes are 4-bit fields, positions below 2^12) and only the tables the
// decoder still reads, a whole slice fits in 79.3 KiB and is decoded in one.
template <class F>
constexpr unsigned rs_chunks()
{
    return 1u;
}
struct RsLds
{ // carved out of dynamic LDS; NB = blocks of a slice, NC = blocks of a chunk
    uint64_t * raw;      // [NB] the slice's blocks, plain
    uint64_t * cbin;     // [64][cols]: C(m, kRsCol0 + j)
    uint64_t * top;      // [64 - top0]: C(63, top0 + j)
    unsigned cols, top0; // rs_cols / rs_top0 of the vector
    uint64_t * rptr;     // [S]: first word of the record's stretch of the overflow stream
    uint32_t * rones;    // [S]: ones in front of the record, relative to the slice
    uint16_t * pre;      // [NB] ones of the record in front of the block
    uint16_t * obit;     // [NC] where the block's field starts in the record's offset bits (slim: | class field << 12)
    uint16_t * ord;      // [NC] blocks in order of decode cost
    uint8_t * cls;       // [NC] class fields (wide records only)
    uint8_t * space;     // [64]
    unsigned * cnt;      // [kRsBins + 1]
};

__device__ __forceinline__ RsLds rs_carve(unsigned char * base, unsigned S, unsigned K, unsigned chunks, bool packed_cls, unsigned sparse_max)
{
    const unsigned NB = S * K, NC = NB / chunks;
    RsLds L;
    L.cols = rs_cols(sparse_max);
    L.top0 = rs_top0(sparse_max);
    unsigned char * p = base;
    auto take = [&](size_t bytes) -> unsigned char *
    {
        unsigned char * r = p;
        p += (bytes + 15) & ~(size_t)15;
        return r;
    };
    L.raw = (uint64_t *)take((size_t)NB * 8);
    L.cbin = (uint64_t *)take(64 * L.cols * 8);
    L.top = (uint64_t *)take((64 - L.top0) * 8);
    L.rptr = (uint64_t *)take((size_t)S * 8);
    L.rones = (uint32_t *)take((size_t)S * 4);
    L.pre = (uint16_t *)take((size_t)NB * 2);
    L.obit = (uint16_t *)take((size_t)NC * 2);
    L.ord = (uint16_t *)take((size_t)NC * 2);
    L.cls = (uint8_t *)take(packed_cls ? 0 : NC);
    L.space = (uint8_t *)take(64);
    L.cnt = (unsigned *)take((kRsBins + 1) * 4);
    return L;
}

size_t rs_lds_bytes(unsigned S, unsigned fmt, unsigned sparse_max)
{
    const size_t NB = (size_t)S * (fmt ? RrrFmtS::K : RrrFmtW::K), NC = NB / (fmt ? rs_chunks<RrrFmtS>() : rs_chunks<RrrFmtW>());
    auto up = [](size_t b) { return (b + 15) & ~(size_t)15; };
    return up(NB * 8) + up(64 * rs_cols(sparse_max) * 8) + up((64 - rs_top0(sparse_max)) * 8) + up((size_t)S * 8) + up((size_t)S * 4) + up(NB * 2) + 2 * up(NC * 2)
           + up(fmt ? 0 : NC) + up(64) + up((kRsBins + 1) * 4);
}

// the block of class k whose field holds f, with the compact tables (rrr_decode_block, rrr_device.hpp)
__device__ __forceinline__ uint64_t rs_decode(const RsLds & L, unsigned k, uint64_t f)
{
    if (rrr_raw_width(L.space[k]))
        return f;
    const bool flip = k > 31;
    unsigned kk = flip ? kRrrBS - k : k;
    uint64_t nr = flip ? L.top[k - L.top0] - 1 - f : f;
    uint64_t bits = 0;
    int hi = 62;
    while (kk > 2)
    { // largest m in [kk - 1, hi] with C(m, kk) <= nr
        int lo = (int)kk - 1, h = hi;
        while (lo < h)
        {
            const int mid = (lo + h + 1) >> 1;
            if (L.cbin[mid * L.cols + kk - kRsCol0] <= nr)
                lo = mid;
            else
                h = mid - 1;
        }
        bits |= UINT64_C(1) << (62 - lo);
        nr -= L.cbin[lo * L.cols + kk - kRsCol0];
        --kk;
        hi = lo - 1;
    }
    // the last two set bits in closed form (no table reads): nr = C(m2, 2) + m3 with m2 > m3 — m2 from a square root, checked
    // in integers; a block of class k costs max(0, k - 2) bisections instead of k (at 5 % density: 1.3 instead of 3.2 on average)
    if (kk == 2)
    {
        const unsigned x = (unsigned)nr; // < C(63, 2)
        unsigned m = (unsigned)((1.0f + __builtin_sqrtf(1.0f + 8.0f * (float)x)) * 0.5f);
        if (m * (m - 1) / 2 > x)
            --m;
        else if ((m + 1) * m / 2 <= x)
            ++m;
        bits |= UINT64_C(1) << (62 - m);
        nr = x - m * (m - 1) / 2;
        kk = 1;
    }
    if (kk == 1)
        bits |= UINT64_C(1) << (62 - (unsigned)nr);
    if (flip)
        bits = ~bits & lo_set(kRrrBS);
    return bits;
}

// The records [recs, recs + nrec) as plain blocks in LDS: classes / ones / offset bits in front of every block, the blocks
// in order of decode cost, every block decoded once.  All threads of the block; L.cnt must be zero on entry; ends with a barrier.
// S: records of a slice (a chunk holds S / rs_chunks of them).
template <class F>
__device__ __forceinline__ void rs_decode_records(const RsLds & L, const RrrView & v, const uint64_t * recs, unsigned nrec, unsigned S)
{
    const unsigned t = threadIdx.x, CH = S / rs_chunks<F>();
    const uint64_t ones0 = F::ones_before(recs[0]);
    // step 1 gives a (record, group of blocks) to one lane: its header and class words are requested a chunk ahead (the fetch of
    // a record that nobody has touched yet was fully exposed: a fifth of the kernel's time)
    static_assert(128u * 4u <= kRsT, "one step-1 item per thread");
    struct Hdr
    {
        uint64_t r0, r1, p, cw;
    };
    auto fetch_hdr = [&](unsigned c0) -> Hdr
    {
        Hdr h{0, 0, 0, 0};
        const unsigned nr = nrec - c0 < CH ? nrec - c0 : CH;
        if (c0 < nrec && t < nr * F::NCW)
        {
            const unsigned lr = t / F::NCW, gi = t - lr * F::NCW;
            const uint64_t * rec = recs + (uint64_t)(c0 + lr) * kRecWords;
            h.r0 = rec[0];
            h.r1 = rec[1];
            h.p = F::id == 0 ? rec[2] : 0;
            h.cw = rec[F::CLS0 + gi];
        }
        return h;
    };
    Hdr hd = fetch_hdr(0);
    for (unsigned c0 = 0; c0 < nrec; c0 += CH)
    {
        const unsigned nr = nrec - c0 < CH ? nrec - c0 : CH, nb = nr * F::K;
        if (c0)
        {
            for (unsigned i = t; i <= kRsBins; i += kRsT)
                L.cnt[i] = 0;
            __syncthreads();
        }
        // 1. per (record, group of blocks): classes, ones and offset bits in front of every block
        if (t < nr * F::NCW)
        {
            const unsigned lr = t / F::NCW, gi = t - lr * F::NCW, r = c0 + lr;
            const uint64_t * rec = recs + (uint64_t)r * kRecWords;
            const uint64_t r0 = hd.r0, r1 = hd.r1, cw = hd.cw;
            unsigned ones, bits;
            if constexpr (F::id == 0)
                rrr_prefix(hd.p, gi, ones, bits);
            else
                rrs_prefix(r0, r1, gi, ones, bits);
            if (gi == 0)
            {
                L.rones[r] = (uint32_t)(F::ones_before(r0) - ones0);
                L.rptr[r] = F::ptr(r1);
            }
            const unsigned nblk = gi + 1 < F::NCW ? F::GRP : F::K - (F::NCW - 1) * F::GRP;
            for (unsigned u = 0; u < nblk; ++u)
            {
                unsigned k;
                if constexpr (F::id == 0)
                    k = rrr_cls(cw, u);
                else
                    k = rrs_cls(cw, u);
                const unsigned lb = lr * F::K + gi * F::GRP + u, len = L.space[k];
                if constexpr (F::id == 0)
                {
                    L.cls[lb] = (uint8_t)k;
                    L.obit[lb] = (uint16_t)bits;
                }
                else
                    L.obit[lb] = (uint16_t)(bits | (k << 12));
                L.pre[c0 * F::K + lb] = (uint16_t)ones;
                unsigned real = k;
                if (F::id && k == kEsc) // an escaped block says how many ones it has
                    real = popc64(rrr_field_t<F::INL0, F::INLW>(v, rec, F::ptr(r1), bits, kRrrBS));
                ones += real;
                bits += len;
                // decode cost: nothing for the raw classes and for k = 0 / 63, else the set bits the decoder walks
                const unsigned c = rrr_raw_width(len) ? 0u : (k > 31 ? kRrrBS - k : k);
                atomicAdd(&L.cnt[c < kRsBins ? c : kRsBins - 1], 1u);
            }
        }
        __syncthreads();
        // 2. exclusive scan of the cost classes (sixteen values: one thread), then the order
        if (t == 0)
        {
            unsigned run = 0;
            for (unsigned c = 0; c < kRsBins; ++c)
            {
                const unsigned x = L.cnt[c];
                L.cnt[c] = run;
                run += x;
            }
        }
        __syncthreads();
        for (unsigned b = t; b < nb; b += kRsT)
        {
            const unsigned k = F::id ? L.obit[b] >> 12 : L.cls[b], len = L.space[k];
            const unsigned c = rrr_raw_width(len) ? 0u : (k > 31 ? kRrrBS - k : k);
            L.ord[atomicAdd(&L.cnt[c < kRsBins ? c : kRsBins - 1], 1u)] = (uint16_t)b;
        }
        __syncthreads();
        const Hdr hn = fetch_hdr(c0 + CH); // (lands while this chunk is decoded)
        // 3. every block decoded once; the lanes of a wave hold blocks of one cost class.  Two blocks per lane and round: both
        // fields are requested before the first one is decoded
        for (unsigned i = t; i < nb; i += 2 * kRsT)
        {
            const unsigned i2 = i + kRsT;
            const bool two = i2 < nb;
            const unsigned b1 = L.ord[i], b2 = two ? L.ord[i2] : b1;
            const unsigned ra = c0 + b1 / F::K, rb = c0 + b2 / F::K, o1 = L.obit[b1], o2 = L.obit[b2];
            const unsigned k1 = F::id ? o1 >> 12 : L.cls[b1], k2 = F::id ? o2 >> 12 : L.cls[b2];
            const uint64_t f1 = rrr_field_t<F::INL0, F::INLW>(v, recs + (uint64_t)ra * kRecWords, L.rptr[ra], o1 & 0xFFFu, L.space[k1]);
            const uint64_t f2 = rrr_field_t<F::INL0, F::INLW>(v, recs + (uint64_t)rb * kRecWords, L.rptr[rb], o2 & 0xFFFu, L.space[k2]);
            L.raw[c0 * F::K + b1] = rs_decode(L, k1, f1);
            if (two)
                L.raw[c0 * F::K + b2] = rs_decode(L, k2, f2);
        }
        __syncthreads();
        hd = hn;
    }
}

// ones in front of every slice of records
__global__ __launch_bounds__(256) void k_rs_slice_bases(RrrView v, unsigned nf, unsigned rlog, uint64_t * __restrict__ hf)
{
    const unsigned f = blockIdx.x * 256 + threadIdx.x;
    if (f < nf)
    {
        const uint64_t r0 = (uint64_t)f << rlog;
        hf[f] = r0 < v.n_sb ? (v.fmt ? RrrFmtS::ones_before(v.rec[r0 * kRecWords]) : v.rec[r0 * kRecWords]) : 0;
    }
}

// ---- rank out of LDS, in place over the final keys -------------------------------------------------------------------
// key = [record in the slice : 8 | block in the record : 6 | bits of the block in front of the position : 6 (0..63)]
template <class F>
__global__ __launch_bounds__(kRsT) void k_rs_rank_lds(RrrView v, int bit, unsigned nf, unsigned rlog, const uint32_t * __restrict__ fstart,
                                                      const uint32_t * __restrict__ ioff, uint32_t * __restrict__ keys,
                                                      const uint32_t * __restrict__ go)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char rs_lds[];
    __shared__ unsigned sh_f;
    if (go && !*go)
        return;
    const unsigned S = 1u << rlog, t = threadIdx.x;
    const RsLds L = rs_carve(rs_lds, S, F::K, rs_chunks<F>(), F::id != 0, v.sparse_max);
    // the tables the decoder reads, once per block of threads
    for (unsigned i = t; i < 64 * L.cols; i += kRsT)
        L.cbin[i] = v.tables->binom[i / L.cols][kRsCol0 + i % L.cols];
    for (unsigned i = t; i < 64; i += kRsT)
    {
        if (i >= L.top0)
            L.top[i - L.top0] = v.tables->binom[63][i];
        L.space[i] = v.tables->space[i];
    }
    const unsigned n_items = ioff[nf];
    for (unsigned item = blockIdx.x; item < n_items; item += gridDim.x)
    {
        if (t == 0)
        { // the slice of this item: last f with ioff[f] <= item
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
        for (unsigned i = t; i <= kRsBins; i += kRsT)
            L.cnt[i] = 0;
        __syncthreads(); // also: everybody is done with the previous slice
        const unsigned f = __builtin_amdgcn_readfirstlane(sh_f);
        const uint64_t R0 = (uint64_t)f << rlog;
        const unsigned nrec = (unsigned)(v.n_sb - R0 < S ? v.n_sb - R0 : S);
        const uint64_t * recs = v.rec + R0 * kRecWords;
        const uint64_t lo = (uint64_t)fstart[f] + (uint64_t)(item - ioff[f]) * kItemKeys;
        const uint64_t fend = fstart[f + 1];
        const unsigned cnt = (unsigned)(lo + kItemKeys < fend ? kItemKeys : fend - lo);
        // (whole 128-byte lines of the key array, as in k_sr_rank_lds: the lanes of the `head` keys in front of the item's first one skip)
        const unsigned head = (unsigned)lo & 31u, cnth = cnt + head;
        const unsigned vo0 = t < head ? 0xFFFFFFFCu : t * 4u;
        const rsrc_t rs_k = make_rsrc(keys + uniform64(lo - head), __builtin_amdgcn_readfirstlane(cnth) * 4u);
        constexpr int U = 4;
        uint32_t key[U];
#pragma unroll
        for (int u = 0; u < U; ++u)
            buf_load(rs_k, t * 4u, (unsigned)u * kRsT * 4u, key[u]);
        rs_decode_records<F>(L, v, recs, nrec, S);
        // 4. the keys
        for (unsigned i0 = 0; i0 < cnth; i0 += kRsT * U)
        {
            uint32_t nk[U];
            const unsigned n0 = (i0 + kRsT * U) * 4u;
            if (i0 + kRsT * U < cnth)
            {
#pragma unroll
                for (int u = 0; u < U; ++u)
                    buf_load(rs_k, t * 4u, n0 + (unsigned)u * kRsT * 4u, nk[u]);
            }
#pragma unroll
            for (int u = 0; u < U; ++u)
            {
                if (u == 0 && i0 == 0 && t < head)
                    key[u] = kBad; // (a key of the line's head: the item in front of this one answers it)
                const uint32_t kq = key[u];
                uint32_t res = kBad;
                if (kq != kBad)
                {
                    const unsigned r = (kq >> 12) & 255u, j = (kq >> 6) & 63u, o = kq & 63u, b = r * F::K + j;
                    const uint32_t r1 = L.rones[r] + L.pre[b] + popc64(L.raw[b] & lo_set(o));
                    res = bit ? r1 : r * (uint32_t)F::SB + j * kRrrBS + o - r1;
                }
                __builtin_amdgcn_raw_buffer_store_b32(res, rs_k, (int)(u == 0 && i0 == 0 ? vo0 : t * 4u), (int)(i0 * 4u + (unsigned)u * kRsT * 4u), kAuxNT);
            }
            if (i0 + kRsT * U < cnth)
            {
#pragma unroll
                for (int u = 0; u < U; ++u)
                    key[u] = nk[u];
            }
        }
    }
}

// ---- select out of LDS ----------------------------------------------------------------------------------------------------
// Bucket f holds the arguments of rank [f * B, (f + 1) * B); select is monotone, so they live in the records bnd[f] .. bnd[f + 1]
// — usually fewer than a slice's worth.  The records are decoded as for rank; a key is then two bisections over counts that
// are already in LDS (records of the slice, blocks of the record) and sel64 inside the plain block.  A bucket that spans more
// records than a slice holds (a sparse stretch) is left to the fix-up pass.  Answers: select_support_rrr<BIT,63>::select
// (rrr_vector.hpp:639-726).
template <int BIT, class F>
__global__ __launch_bounds__(kRsT) void k_rs_select_lds(RrrView v, unsigned nf, unsigned rlog, unsigned B, const uint32_t * __restrict__ bnd,
                                                        const uint32_t * __restrict__ fstart, const uint32_t * __restrict__ ioff,
                                                        uint32_t * __restrict__ keys, uint32_t * __restrict__ any_marked,
                                                        const uint32_t * __restrict__ go)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char rs_lds[];
    __shared__ unsigned sh_f;
    if (go && !*go)
        return;
    const unsigned S = 1u << rlog, t = threadIdx.x;
    const RsLds L = rs_carve(rs_lds, S, F::K, rs_chunks<F>(), F::id != 0, v.sparse_max);
    for (unsigned i = t; i < 64 * L.cols; i += kRsT)
        L.cbin[i] = v.tables->binom[i / L.cols][kRsCol0 + i % L.cols];
    for (unsigned i = t; i < 64; i += kRsT)
    {
        if (i >= L.top0)
            L.top[i - L.top0] = v.tables->binom[63][i];
        L.space[i] = v.tables->space[i];
    }
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
        for (unsigned i = t; i <= kRsBins; i += kRsT)
            L.cnt[i] = 0;
        __syncthreads();
        const unsigned f = __builtin_amdgcn_readfirstlane(sh_f); // (tables in slice order: f is the bucket)
        const uint64_t lo = (uint64_t)fstart[f] + (uint64_t)(item - ioff[f]) * kItemKeys;
        const uint64_t fend = fstart[f + 1];
        const unsigned cnt = (unsigned)(lo + kItemKeys < fend ? kItemKeys : fend - lo);
        const uint64_t R0 = bnd[f];
        const uint64_t R1 = (uint64_t)bnd[f + 1] + 1 < v.n_sb ? (uint64_t)bnd[f + 1] + 1 : v.n_sb;
        if (R1 - R0 > S)
        { // wider than a slice: the fix-up pass answers these
            uint32_t * kp = keys + lo;
            for (unsigned i = t; i < cnt; i += kRsT)
                if (kp[i] != kBad)
                    kp[i] = kMark;
            if (t == 0)
                *any_marked = 1;
            __syncthreads();
            continue;
        }
        const unsigned nrec = (unsigned)(R1 - R0);
        const uint64_t * recs = v.rec + R0 * kRecWords;
        // (whole 128-byte lines of the key array, as in k_sr_rank_lds: the lanes of the `head` keys in front of the item's first one skip)
        const unsigned head = (unsigned)lo & 31u, cnth = cnt + head;
        const unsigned vo0 = t < head ? 0xFFFFFFFCu : t * 4u;
        const rsrc_t rs_k = make_rsrc(keys + uniform64(lo - head), __builtin_amdgcn_readfirstlane(cnth) * 4u);
        constexpr int U = 4;
        uint32_t key[U];
#pragma unroll
        for (int u = 0; u < U; ++u)
            buf_load(rs_k, t * 4u, (unsigned)u * kRsT * 4u, key[u]);
        rs_decode_records<F>(L, v, recs, nrec, S);
        const uint64_t ones0 = F::ones_before(recs[0]);
        const uint64_t A0 = BIT ? ones0 : R0 * F::SB - ones0; // arguments in front of the slice
        const unsigned t0 = (unsigned)((uint64_t)f * B - A0);   // rank of the bucket's first argument, relative to the slice
        auto rargs = [&](unsigned r) -> unsigned { return BIT ? L.rones[r] : r * (unsigned)F::SB - L.rones[r]; };
        bool mk = false;
        for (unsigned i0 = 0; i0 < cnth; i0 += kRsT * U)
        {
            uint32_t nk[U];
            const unsigned n0 = (i0 + kRsT * U) * 4u;
            if (i0 + kRsT * U < cnth)
            {
#pragma unroll
                for (int u = 0; u < U; ++u)
                    buf_load(rs_k, t * 4u, n0 + (unsigned)u * kRsT * 4u, nk[u]);
            }
#pragma unroll
            for (int u = 0; u < U; ++u)
            {
                if (u == 0 && i0 == 0 && t < head)
                    key[u] = kBad; // (a key of the line's head: the item in front of this one answers it)
                const uint32_t kq = key[u];
                uint32_t res = kq; // (NPOS and "size()" keys travel on as they are)
                mk |= kq == kMark;
                if (kq < kMark)
                {
                    const unsigned tg = t0 + kq; // arguments of the slice in front of the wanted one
                    unsigned a = 0, z = nrec;    // last record with rargs <= tg
                    while (z - a > 1)
                    {
                        const unsigned m = (a + z) >> 1;
                        if (rargs(m) <= tg)
                            a = m;
                        else
                            z = m;
                    }
                    unsigned want = tg - rargs(a);
                    const unsigned b0 = a * F::K;
                    unsigned ja = 0, jz = F::K; // last block of the record with (arguments of the record in front of it) <= want
                    while (jz - ja > 1)
                    {
                        const unsigned m = (ja + jz) >> 1;
                        const unsigned in_front = BIT ? L.pre[b0 + m] : m * kRrrBS - L.pre[b0 + m];
                        if (in_front <= want)
                            ja = m;
                        else
                            jz = m;
                    }
                    want -= BIT ? L.pre[b0 + ja] : ja * kRrrBS - L.pre[b0 + ja];
                    const uint64_t bits = BIT ? L.raw[b0 + ja] : ~L.raw[b0 + ja] & lo_set(kRrrBS);
                    res = a * (uint32_t)F::SB + ja * kRrrBS + sel64(bits, want + 1); // relative to the slice's first bit
                }
                __builtin_amdgcn_raw_buffer_store_b32(res, rs_k, (int)(u == 0 && i0 == 0 ? vo0 : t * 4u), (int)(i0 * 4u + (unsigned)u * kRsT * 4u), kAuxNT);
            }
            if (i0 + kRsT * U < cnth)
            {
#pragma unroll
                for (int u = 0; u < U; ++u)
                    key[u] = nk[u];
            }
        }
        if (mk)
            *any_marked = 1;
    }
}

// first bit of every bucket's first record (what makes a slice-relative position absolute)
__global__ __launch_bounds__(256) void k_rs_select_bases(unsigned nf, unsigned n_buckets, uint64_t rec_bits, const uint32_t * __restrict__ bnd,
                                                         uint64_t * __restrict__ hf)
{
    const unsigned f = blockIdx.x * 256 + threadIdx.x;
    if (f < nf)
        hf[f] = f < n_buckets ? (uint64_t)bnd[f] * rec_bits : 0;
}

// the arguments that were left over (kMark64 in the output): the direct search, one lane per marked answer
template <int BIT, class F>
__global__ __launch_bounds__(kRrrBlock) void k_rs_select_fixup(RrrView v, const uint32_t * __restrict__ any_marked, const uint64_t * __restrict__ iq,
                                                               uint64_t * __restrict__ out, uint64_t n, const uint32_t * __restrict__ go)
{
    __shared__ RrrTables T;
    if ((go && !*go) || !*any_marked)
        return;
    rrr_stage_tables(&T, v.tables);
    for (uint64_t q = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; q < n; q += (uint64_t)gridDim.x * blockDim.x)
        if (out[q] == kMark64)
        {
            const uint64_t i = iq[q], total = BIT ? v.ones : v.n_bits - v.ones;
            out[q] = i > total ? v.n_bits : rrr_select<BIT, F>(v, &T, i - 1); // (beyond the last argument: size(), rrr_vector.hpp:641-642)
        }
}

__global__ __launch_bounds__(256) void k_rs_bnd_args(unsigned nf, unsigned B, uint64_t * __restrict__ out)
{
    const unsigned f = blockIdx.x * 256 + threadIdx.x;
    if (f < nf)
        out[f] = (uint64_t)f * B + 1; // 1-based rank of the bucket's first argument
}
__global__ __launch_bounds__(256) void k_rs_bnd_records(unsigned nf, const uint64_t * __restrict__ pos, uint64_t n_sb, uint64_t rec_bits,
                                                        uint32_t * __restrict__ bnd)
{
    const unsigned f = blockIdx.x * 256 + threadIdx.x;
    if (f < nf)
        bnd[f] = (uint32_t)(pos[f] / rec_bits);
    if (f == nf)
        bnd[nf] = (uint32_t)(n_sb - 1);
}

} // namespace

// records per slice: 2^7 (two answering blocks per CU) while that keeps the slices within two 8-bit digits, else 2^8
static unsigned rs_rlog(const RrrView & v)
{
    return ((v.n_sb + 127) >> 7) <= 65536 ? 7u : 8u;
}

bool rrr_sorted_rank_possible(const RrrView & v)
{
    return v.n_sb >= 2 && v.n_sb <= kLimRrrBucketedRecords;
}

bool rrr_sorted_rank_applicable(const RrrView & v, uint64_t n)
{
    // worth it when the records exceed the Infinity Cache (2^21 records = 256 MiB) and the batch addresses every record a few
    // times over (the passes cost about what twelve bytes of streaming per key cost, whatever the vector)
    return rrr_sorted_rank_possible(v) && v.n_sb >= (UINT64_C(1) << 21) && n >= 8 * v.n_sb;
}

static void rs_fill(SrGeom & g, const RrrView & v, uint64_t cnt)
{
    g = SrGeom{};
    g.n = cnt;
    g.n_bits = v.n_bits;
    g.n_lines = 0;
    const uint64_t rec_bits = v.fmt ? RrrFmtS::SB : RrrFmtW::SB;
    g.op = v.fmt ? 3 : 2;
    g.rbits = (uint32_t)rec_bits;
    g.rlog = rs_rlog(v);
    const uint64_t slices = (v.n_sb + (UINT64_C(1) << g.rlog) - 1) >> g.rlog;
    unsigned f = 0;
    while (((slices - 1) >> f) != 0)
        ++f;
    g.d2 = f < 8 ? f : 8;
    g.d1 = f - g.d2;
    g.kb = 20;
    g.slice_bits = rec_bits << g.rlog;
    g.small = false;
    g.go = nullptr;
}

void rrr_sorted_rank_sample(const RrrView & v, const uint64_t * d_idx, uint64_t n, hipStream_t s, uint32_t * out3)
{
    SrGeom g;
    rs_fill(g, v, n);
    sr_launch_sample(g, d_idx, out3, s);
}

sdsl_hip_status rrr_launch_rank_sorted(const RrrView & v, int bit, const uint64_t * d_idx, uint64_t n, uint64_t * d_out, hipStream_t s,
                                       void * scratch, size_t scratch_bytes, const uint32_t * go)
{
    if (!rrr_sorted_rank_possible(v))
    {
        set_error("rrr rank_sorted: vector too large for the bucketed path");
        return SDSL_HIP_ERR_INVALID;
    }
    const unsigned rlog = rs_rlog(v);
    const size_t lds = rs_lds_bytes(1u << rlog, v.fmt, v.sparse_max);
    SH_HIP(hipFuncSetAttribute((const void *)k_rs_rank_lds<RrrFmtW>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    SH_HIP(hipFuncSetAttribute((const void *)k_rs_rank_lds<RrrFmtS>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    SwCallbacks cb;
    cb.what = "bucketed rrr rank";
    cb.fill = [&](SrGeom & g, uint64_t cnt) { rs_fill(g, v, cnt); };
    cb.answers = [&](const SrGeom & g, unsigned nf, const uint32_t * fstart, const uint32_t * ioff, uint32_t * keys2, uint64_t * hf, uint32_t *,
                     hipStream_t st) -> sdsl_hip_status
    {
        hipLaunchKernelGGL(k_rs_slice_bases, dim3((nf + 255) / 256), dim3(256), 0, st, v, nf, rlog, hf);
        if (v.fmt)
            hipLaunchKernelGGL(k_rs_rank_lds<RrrFmtS>, dim3(rlog == 7 ? 512u : 256u), dim3(kRsT), lds, st, v, bit, nf, rlog, fstart, ioff, keys2, g.go);
        else
            hipLaunchKernelGGL(k_rs_rank_lds<RrrFmtW>, dim3(rlog == 7 ? 512u : 256u), dim3(kRsT), lds, st, v, bit, nf, rlog, fstart, ioff, keys2, g.go);
        SH_HIP(hipGetLastError());
        return SDSL_HIP_OK;
    };
    return sw_run_with(cb, bit, d_idx, n, d_out, s, scratch, scratch_bytes, go);
}

// Buckets of the bucketed select: B consecutive argument ranks each, at most 2^16 of them, sized so that a bucket of a uniformly
// dense vector spans about 70 % of a slice's records; bnd[f] = record of the bucket's first argument (one small batch through the
// direct kernel, once per handle and bit value).
sdsl_hip_status rrr_select_sorted_prepare(RrrHost & h, int bit)
{
    RrrHost::SelPlan & P = h.sel_plan[bit];
    if (P.ready)
        return SDSL_HIP_OK;
    const RrrView & v = h.view;
    const uint64_t total = bit ? v.ones : v.n_bits - v.ones;
    P.ok = false;
    if (!v.sel[bit] || total < 2 || !rrr_sorted_rank_possible(v))
    {
        P.ready = true;
        return SDSL_HIP_OK;
    }
    auto round_down = [](uint64_t x, unsigned & m, unsigned & sh) { // largest m << sh <= x (x >= 64), m in 8..15
        sh = 0;
        while ((x >> sh) > 15)
            ++sh;
        m = (unsigned)(x >> sh);
    };
    // records per slice: 2^7 (two answering blocks per CU) unless 2^16 buckets of that capacity cannot hold the arguments
    const uint64_t b_min = (total + 65535) >> 16;
    const uint64_t rec_bits = v.fmt ? RrrFmtS::SB : RrrFmtW::SB;
    unsigned rlog = 7;
    if (0.7 * 128 * (double)rec_bits * ((double)total / (double)v.n_bits) < 1.15 * (double)b_min)
        rlog = 8;
    P.rlog = rlog;
    const unsigned S = 1u << rlog;
    const double per = 0.7 * S * (double)rec_bits * ((double)total / (double)v.n_bits);
    const uint64_t b_fit = per < 64.0 ? 64 : (per > 15.0 * 1048576.0 ? (uint64_t)15 << 20 : (uint64_t)per);
    unsigned bm, bs;
    round_down(b_fit, bm, bs);
    if (((uint64_t)bm << bs) < b_min)
    {
        round_down(b_min < 64 ? 64 : b_min, bm, bs);
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
    {
        P.bnd.release();
        return SDSL_HIP_OK; // (no room right now: the query takes the direct kernel, the plan is tried again next time)
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
    hipLaunchKernelGGL(k_rs_bnd_args, dim3((nf + 255) / 256), dim3(256), 0, 0, nf, (unsigned)B, args.as<uint64_t>());
    SH_HIP(hipGetLastError());
    {
        TimingPause pause;
        SH_TRY(rrr_launch_select(v, bit, args.as<uint64_t>(), nf, pos.as<uint64_t>(), nullptr));
    }
    hipLaunchKernelGGL(k_rs_bnd_records, dim3((nf + 256) / 256), dim3(256), 0, 0, nf, pos.as<uint64_t>(), v.n_sb, rec_bits, P.bnd.as<uint32_t>());
    SH_HIP(hipGetLastError());
    std::vector<uint32_t> hb((size_t)nf + 1);
    SH_HIP(hipMemcpy(hb.data(), P.bnd.p, hb.size() * 4, hipMemcpyDeviceToHost));
    uint64_t wide = 0;
    for (unsigned f = 0; f < nf; ++f)
        if ((uint64_t)hb[f + 1] + 1 - hb[f] > S)
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

bool rrr_sorted_select_applicable(const RrrHost & h, int bit, uint64_t n)
{
    const RrrHost::SelPlan & P = h.sel_plan[bit];
    return P.ready && P.ok && P.wide_frac <= 0.01 && h.view.n_sb >= (UINT64_C(1) << 21) && n >= 8 * h.view.n_sb;
}

static void rs_fill_select(SrGeom & g, const RrrHost & h, int bit, uint64_t cnt)
{
    const RrrHost::SelPlan & P = h.sel_plan[bit];
    SelectPlan sp;
    sp.bnd = P.bnd.as<uint32_t>();
    sp.bm = P.bm;
    sp.bs = P.bs;
    sp.nf = P.nf;
    sp.total = bit ? h.view.ones : h.view.n_bits - h.view.ones;
    BvView none{};
    none.n_bits = h.view.n_bits;
    none.n_lines = 2;
    sr_fill_geom(g, none, 1, sp, cnt); // (select keys depend on the bucket plan only)
    g.over_is_size = 1;
}

void rrr_sorted_select_sample(const RrrHost & h, int bit, const uint64_t * d_idx, uint64_t n, hipStream_t s, uint32_t * out3)
{
    SrGeom g;
    rs_fill_select(g, h, bit, n);
    sr_launch_sample(g, d_idx, out3, s);
}

sdsl_hip_status rrr_launch_select_sorted(RrrHost & h, int bit, const uint64_t * d_i, uint64_t n, uint64_t * d_out, hipStream_t s, void * scratch,
                                         size_t scratch_bytes, const uint32_t * go)
{
    const RrrHost::SelPlan & P = h.sel_plan[bit];
    if (!P.ready || !P.ok)
    {
        set_error("rrr select_sorted: no bucket plan for this vector");
        return SDSL_HIP_ERR_INVALID;
    }
    const RrrView & v = h.view;
    const unsigned rlog = P.rlog;
    const size_t lds = rs_lds_bytes(1u << rlog, v.fmt, v.sparse_max);
    const uint64_t rec_bits = v.fmt ? RrrFmtS::SB : RrrFmtW::SB;
    SH_HIP(hipFuncSetAttribute((const void *)k_rs_select_lds<1, RrrFmtW>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    SH_HIP(hipFuncSetAttribute((const void *)k_rs_select_lds<0, RrrFmtW>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    SH_HIP(hipFuncSetAttribute((const void *)k_rs_select_lds<1, RrrFmtS>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    SH_HIP(hipFuncSetAttribute((const void *)k_rs_select_lds<0, RrrFmtS>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    const uint32_t * bnd = P.bnd.as<uint32_t>();
    const unsigned B = P.bm << P.bs, n_buckets = P.nf;
    SwCallbacks cb;
    cb.what = "bucketed rrr select";
    cb.fill = [&](SrGeom & g, uint64_t cnt) { rs_fill_select(g, h, bit, cnt); };
    cb.answers = [&](const SrGeom & g, unsigned nf, const uint32_t * fstart, const uint32_t * ioff, uint32_t * keys2, uint64_t * hf, uint32_t * marked,
                     hipStream_t st) -> sdsl_hip_status
    {
        SH_TRY(fill_u32_async(marked, 0u, 4, st));
        hipLaunchKernelGGL(k_rs_select_bases, dim3((nf + 255) / 256), dim3(256), 0, st, nf, n_buckets, rec_bits, bnd, hf);
        // (slices beyond the plan's buckets have no keys, so the kernel never reads bnd past n_buckets)
        const dim3 grid(rlog == 7 ? 512u : 256u);
        if (v.fmt)
        {
            if (bit)
                hipLaunchKernelGGL((k_rs_select_lds<1, RrrFmtS>), grid, dim3(kRsT), lds, st, v, nf, rlog, B, bnd, fstart, ioff, keys2, marked, g.go);
            else
                hipLaunchKernelGGL((k_rs_select_lds<0, RrrFmtS>), grid, dim3(kRsT), lds, st, v, nf, rlog, B, bnd, fstart, ioff, keys2, marked, g.go);
        }
        else
        {
            if (bit)
                hipLaunchKernelGGL((k_rs_select_lds<1, RrrFmtW>), grid, dim3(kRsT), lds, st, v, nf, rlog, B, bnd, fstart, ioff, keys2, marked, g.go);
            else
                hipLaunchKernelGGL((k_rs_select_lds<0, RrrFmtW>), grid, dim3(kRsT), lds, st, v, nf, rlog, B, bnd, fstart, ioff, keys2, marked, g.go);
        }
        SH_HIP(hipGetLastError());
        return SDSL_HIP_OK;
    };
    cb.fixup = [&](const uint32_t * marked, const uint64_t * idx, uint64_t * out, uint64_t cnt, hipStream_t st)
    {
        if (v.fmt)
        {
            if (bit)
                hipLaunchKernelGGL((k_rs_select_fixup<1, RrrFmtS>), dim3(256 * 4), dim3(kRrrBlock), 0, st, v, marked, idx, out, cnt, go);
            else
                hipLaunchKernelGGL((k_rs_select_fixup<0, RrrFmtS>), dim3(256 * 4), dim3(kRrrBlock), 0, st, v, marked, idx, out, cnt, go);
        }
        else
        {
            if (bit)
                hipLaunchKernelGGL((k_rs_select_fixup<1, RrrFmtW>), dim3(256 * 4), dim3(kRrrBlock), 0, st, v, marked, idx, out, cnt, go);
            else
                hipLaunchKernelGGL((k_rs_select_fixup<0, RrrFmtW>), dim3(256 * 4), dim3(kRrrBlock), 0, st, v, marked, idx, out, cnt, go);
        }
    };
    return sw_run_with(cb, bit, d_i, n, d_out, s, scratch, scratch_bytes, go);
}

} // namespace sdslhip
