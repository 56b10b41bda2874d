// sd.hip — sd_vector<> (Elias-Fano coded sparse bit vector) on the device: rank / select / access.
//
// Reference semantics reproduced (bit-identical answers; SDSL's serialised arrays are accepted as input):
//   sd_vector(bit_vector const&) / (begin, end)   sd_vector.hpp:217-305   wl = logn - logm, low = wl low bits of
//                                                 every position, high = unary-coded high parts (m ones, 2^logm zeros)
//   sd_vector::operator[]                         sd_vector.hpp:328-349
//   rank_support_sd<b>::rank                      sd_vector.hpp:553-575
//   select_support_sd<1>::select                  sd_vector.hpp:621-631   low[i-1] + ((high_1_select(i) + 1 - i) << wl)
//   select_support_sd<0>::select                  sd_vector.hpp:633-664   (reference: binary search over select_1;
//                                                 here: interpolated search over buckets, same positions)
//
// Device layout: `high` as rank lines with both select directories (bv_device.hpp), `low` as SDSL's packed words.
// One query per quad.  rank(x): the bucket of x's high part is delimited by two select_0 on `high` (they land in the
// same or neighbouring windows), then a binary search over the bucket's low parts — the reference scans the bucket
// linearly from its end, which is the same count.  select_1: one select_1 on `high` and one read of `low`.
#include <algorithm>

#include "bv_host.hpp"
#include "bv_serialize.hpp"
#include "sdsl_stream.hpp"

namespace sdslhip {

struct SdView
{
    BvView high;
    const uint64_t * low; // packed, wl bits per entry, padded by one word
    uint64_t n, m;        // size of the bit vector, number of ones
    uint32_t wl;
    // select_0 directory (device only, quad_sd_select0): entry q = [bucket c : 32 | entries in front of it : 32] for the largest
    // bucket c whose first position has at most q << sel0_shift zeros in front of it; nullptr when the vector has none
    const uint64_t * sel0_dir;
    uint32_t sel0_shift;
    uint32_t per_bucket_fx; // entries per bucket of a uniformly filled vector, in 1/65536 (the estimate of quad_sd_select0)
    // rank directory (device only, lane kernels): entry j = entries with a high part below j << rank_shift; nullptr = none
    const uint64_t * rank_dir;
    uint32_t rank_shift;
};
constexpr uint64_t kSdRedo = ~UINT64_C(0) - 1; // what a lane kernel writes where its short road did not lead (k_sd_redo answers those)
constexpr uint8_t kSdRedo8 = 0xFE;

struct SdHost
{
    int device = 0;
    BvHost high;
    DevBuf low, sel0_dir, rank_dir;
    bool lane_rank_ok = false, lane_sel0_ok = false; // the lane kernels answer this vector's queries on their short road (sd_probe_lanes)
    SdView view{};
    uint32_t low_width_when_empty = 64; // width SDSL's `low` reports for m == 0 (wl for built vectors)
    size_t device_bytes() const
    {
        return high.device_bytes() + low.bytes + sel0_dir.bytes + rank_dir.bytes;
    }
};

__device__ __forceinline__ uint64_t sd_low(const SdView & v, uint64_t i)
{
    return read_bits(v.low, i * v.wl, v.wl);
}

// position of the first zero at or after p inside the rank line of p (all four lanes); NPOS if that line has none
template <bool NT>
__device__ __forceinline__ uint64_t quad_next_zero_in_line(const BvView & bv, int s, uint64_t p)
{
    uint64_t L;
    unsigned off;
    line_of(p, bv.n_bits < (UINT64_C(1) << 38), L, off);
    const Pair w = load_pair<NT>(bv.lines, L, s);
    unsigned best = 0xFFFFFFFFu;
    // lane s holds data words 2s-1 (.a; lane 0 holds the header there) and 2s (.b); data word d covers [64d, 64d+64)
    if (s > 0)
    {
        const unsigned base = 64u * (unsigned)(2 * s - 1);
        uint64_t z = ~w.a;
        if (base + 64 <= off)
            z = 0;
        else if (base < off)
            z &= ~lo_set(off - base);
        if (z)
            best = base + (unsigned)__builtin_ctzll(z);
    }
    if (best == 0xFFFFFFFFu)
    {
        const unsigned base = 64u * (unsigned)(2 * s);
        uint64_t z = ~w.b;
        if (base + 64 <= off)
            z = 0;
        else if (base < off)
            z &= ~lo_set(off - base);
        if (z)
            best = base + (unsigned)__builtin_ctzll(z);
    }
    unsigned o = quad_xor1(best);
    best = best < o ? best : o;
    o = quad_xor2(best);
    best = best < o ? best : o;
    return best == 0xFFFFFFFFu ? SDSL_HIP_NPOS : L * kDB + best;
}

// number of ones in [0, x), x in [0, n]; all four lanes return it.  *hit (optional) = "bit x is set" (x < n)
template <bool NT>
__device__ __forceinline__ uint64_t quad_sd_rank1(const SdView & v, int s, uint64_t x, bool * hit)
{
    const uint64_t h = x >> v.wl, val_low = x & lo_set(v.wl);
    // The bucket of high part h is the run of ones between the h-th and the (h+1)-th zero of `high`.  One select_0
    // finds the h-th zero; the next zero almost always lies in the same 448-bit line (a bucket holds ~1 entry on
    // average), so it is read off that line instead of paying a second select.
    bool mine;
    uint64_t start = 0; // first position of the bucket's run in `high`
    if (h > 0)
    {
        const uint64_t p = quad_select<0, NT>(v.high, s, h - 1, mine);
        start = quad_gather_u64(p, mine) + 1;
    }
    const uint64_t begin = start - h; // entries with high part < h
    uint64_t nz = quad_next_zero_in_line<NT>(v.high, s, start);
    if (nz == SDSL_HIP_NPOS)
    { // the run leaves the line: a clustered bucket
        const uint64_t p = quad_select<0, NT>(v.high, s, h, mine);
        nz = quad_gather_u64(p, mine);
    }
    const uint64_t end = nz - h; // entries with high part <= h
    uint64_t lo = begin, hi = end; // first entry of the bucket with low part >= val_low
    while (lo < hi)
    {
        const uint64_t mid = lo + ((hi - lo) >> 1);
        if (sd_low(v, mid) < val_low)
            lo = mid + 1;
        else
            hi = mid;
    }
    if (hit)
        *hit = lo < end && sd_low(v, lo) == val_low;
    return lo;
}

// position of the i-th one, i in [1, m]
template <bool NT>
__device__ __forceinline__ uint64_t quad_sd_select1(const SdView & v, int s, uint64_t i)
{
    bool mine;
    const uint64_t p = quad_select<1, NT>(v.high, s, i - 1, mine);
    const uint64_t hp = quad_gather_u64(p, mine);
    return sd_low(v, i - 1) + ((hp + 1 - i) << v.wl);
}

// position of the i-th zero, i in [1, n - m].  The reference bisects over the ones with one select_1 per step
// (sd_vector.hpp:633-664: ~log2(m) dependent selects).  Here the search runs over BUCKETS (high parts): f(b) = zeros in
// front of position b << wl = (b << wl) - (entries with high part < b) is monotone and — the zeros being what a sparse
// vector mostly consists of — nearly linear in b.
//  * With the directory (sel0_dir): entry k >> shift names a bucket c at or in front of the wanted one and the entries in front
//    of c.  Buckets are 2^wl positions wide and an entry takes one zero away, so b = (k + entries in front of c) >> wl is the
//    wanted bucket unless more than a bucket's worth of entries lies between c and it: ONE select_0 on `high` gives the
//    entries in front of b, the bucket's end is read off the same line, and f(b + 1) confirms it.  If not, the general
//    search goes on from b + 1 with exact counts.
//  * Without it, or from there: an interpolated guess lands within a few buckets and the bracket (exact f at both ends) closes in
//    two or three probes, one select_0 on `high` each; every second late probe bisects.
// Inside the bucket the t-th one (0-based) sits in front of the wanted zero iff low_t - t <= r, r the zero's rank
// inside the bucket: monotone in t, found by bisection over the bucket's few entries.  The answer is the same position.
// DIR_ONLY (building the directory): returns [bucket : 32 | entries in front of it : 32] instead of the position.
template <bool NT, bool DIR_ONLY = false>
__device__ __forceinline__ uint64_t quad_sd_select0(const SdView & v, int s, uint64_t i)
{
    const uint64_t k = i - 1;
    uint64_t blo = 0, flo = 0, bhi = (v.n >> v.wl) + 1, fhi = v.n - v.m; // f(blo) <= k < f(bhi)
    bool mine;
    uint64_t nz = SDSL_HIP_NPOS; // the zero that closes bucket blo, once known
    bool found = false;
    if (!DIR_ONLY && v.sel0_dir)
    {
        const uint64_t e = v.sel0_dir[k >> v.sel0_shift];
        const uint64_t c = e >> 32, bef = e & 0xFFFFFFFFu;
        const uint64_t b = (k + bef) >> v.wl; // >= c, and f(b) <= k whatever lies between c and b
        uint64_t before = bef;
        if (b > c)
        { // the zero that closes bucket b - 1: b - c zeros and about (b - c) x (entries per bucket) ones behind the start of
          // bucket c — the window there holds it unless the entries in between are far from average; no directory lookup
            const uint64_t x_est = bef + c + (b - c - 1) + (((b - c) * (uint64_t)v.per_bucket_fx) >> 16);
            uint64_t p;
            {
                uint64_t W;
                unsigned off_unused;
                line_of(x_est, v.high.n_bits < (UINT64_C(1) << 38), W, off_unused);
                W >>= 1;
                const uint64_t last_win = (v.high.n_lines >> 1) - 1;
                W = W > last_win ? last_win : W;
                const Pair wa = load_pair<NT>(v.high.lines, 2 * W, s), wb = load_pair<NT>(v.high.lines, 2 * W + 1, s);
                SelBracket br{0, 0, v.high.n_bits, v.high.n_bits - v.high.ones};
                if (!sel_eval<0>(v.high, s, b - 1, W, wa, wb, br, mine, p))
                    p = quad_select<0, NT>(v.high, s, b - 1, mine);
            }
            before = quad_gather_u64(p, mine) + 1 - b;
        }
        const uint64_t fb = (b << v.wl) - before, start = before + b;
        uint64_t z = quad_next_zero_in_line<NT>(v.high, s, start);
        if (z == SDSL_HIP_NPOS)
        {
            const uint64_t p = quad_select<0, NT>(v.high, s, b, mine);
            z = quad_gather_u64(p, mine);
        }
        uint64_t bb = b, fbb = fb, f_next = fb + (UINT64_C(1) << v.wl) - (z - start); // f(b + 1)
        // the entries between c and b may push the answer into one of the next buckets: their ends are read off the same line
        for (int step = 0; step < 4 && k >= f_next; ++step)
        {
            const uint64_t st2 = z + 1;
            uint64_t z2 = quad_next_zero_in_line<NT>(v.high, s, st2);
            if (z2 == SDSL_HIP_NPOS)
                break;
            ++bb;
            fbb = f_next;
            f_next = fbb + (UINT64_C(1) << v.wl) - (z2 - st2);
            z = z2;
        }
        if (k < f_next)
        {
            blo = bb;
            flo = fbb;
            nz = z;
            found = true;
        }
        else
        { // (far more entries between c and the answer than a bucket is wide: clustered data; the general search goes on from
          // exact counts)
            blo = bb + 1;
            flo = f_next;
        }
    }
    if (!found)
    {
        for (int tries = 0; bhi - blo > 1; ++tries)
        {
            const uint64_t span = bhi - blo;
            uint64_t b;
            if (tries >= 3 && (tries & 1))
                b = blo + (span >> 1);
            else
            {
                const double f = (double)(k - flo) / (double)(fhi - flo);
                b = blo + (uint64_t)(f * (double)span);
            }
            b = b <= blo ? blo + 1 : (b >= bhi ? bhi - 1 : b);
            const uint64_t p = quad_select<0, NT>(v.high, s, b - 1, mine); // the zero that closes bucket b - 1
            const uint64_t before = quad_gather_u64(p, mine) + 1 - b;     // entries with high part < b
            const uint64_t fb = (b << v.wl) - before;
            if (fb <= k)
            {
                blo = b;
                flo = fb;
            }
            else
            {
                bhi = b;
                fhi = fb;
            }
        }
    }
    // bucket blo: its entries are [begin, end) of `low`
    const uint64_t begin = (blo << v.wl) - flo, start = begin + blo; // start: the bucket's run in `high`
    if (DIR_ONLY)
        return (blo << 32) | begin;
    if (nz == SDSL_HIP_NPOS)
    {
        nz = quad_next_zero_in_line<NT>(v.high, s, start);
        if (nz == SDSL_HIP_NPOS)
        {
            const uint64_t p = quad_select<0, NT>(v.high, s, blo, mine);
            nz = quad_gather_u64(p, mine);
        }
    }
    const uint64_t cnt = nz - blo - begin, r = k - flo;
    uint64_t lo = 0, hi = cnt; // first t with low_t - t > r
    while (lo < hi)
    {
        const uint64_t mid = lo + ((hi - lo) >> 1);
        if (sd_low(v, begin + mid) <= r + mid)
            lo = mid + 1;
        else
            hi = mid;
    }
    return (blo << v.wl) + r + lo;
}

// directory entry q: the bucket of zero number q << shift
__global__ __launch_bounds__(kBlock) void k_sd_sel0_dir(SdView v, uint32_t shift, uint64_t n_entries, uint64_t * __restrict__ dir)
{
    const int s = threadIdx.x & (kG - 1);
    const unsigned gq = threadIdx.x / kG;
    for (uint64_t base = (uint64_t)blockIdx.x * kQPB; base < n_entries; base += (uint64_t)gridDim.x * kQPB)
    {
        const uint64_t q = base + gq;
        if (q >= n_entries)
            continue;
        const uint64_t e = quad_sd_select0<false, true>(v, s, (q << shift) + 1);
        if (s == 0)
            dir[q] = e;
    }
}

template <int MODE> // 0: rank (bit b), 1: access
__global__ __launch_bounds__(kBlock) void k_sd_rank(SdView v, int bit, const uint64_t * __restrict__ xq,
                                                    uint64_t * __restrict__ out, uint8_t * __restrict__ out8, uint64_t n)
{
    const int s = threadIdx.x & (kG - 1);
    const unsigned gq = threadIdx.x / kG;
    for (uint64_t base = (uint64_t)blockIdx.x * kQPB; base < n; base += (uint64_t)gridDim.x * kQPB)
    {
        const uint64_t q = base + gq;
        if (q >= n)
            continue;
        const uint64_t x = __builtin_nontemporal_load(xq + q);
        if (MODE == 1)
        {
            bool hit = false;
            if (x < v.n)
                quad_sd_rank1<false>(v, s, x, &hit);
            if (s == 0)
                out8[q] = x < v.n ? (hit ? 1 : 0) : 0xFF;
        }
        else
        {
            uint64_t r = SDSL_HIP_NPOS;
            if (x <= v.n)
            {
                const uint64_t r1 = quad_sd_rank1<false>(v, s, x, nullptr);
                r = bit ? r1 : x - r1;
            }
            if (s == 0)
                __builtin_nontemporal_store(r, out + q);
        }
    }
}

template <int BIT>
__global__ __launch_bounds__(kBlock) void k_sd_select(SdView v, const uint64_t * __restrict__ iq,
                                                      uint64_t * __restrict__ out, uint64_t n)
{
    const int s = threadIdx.x & (kG - 1);
    const unsigned gq = threadIdx.x / kG;
    const uint64_t total = BIT ? v.m : v.n - v.m;
    for (uint64_t base = (uint64_t)blockIdx.x * kQPB; base < n; base += (uint64_t)gridDim.x * kQPB)
    {
        const uint64_t q = base + gq;
        if (q >= n)
            continue;
        const uint64_t i = __builtin_nontemporal_load(iq + q);
        uint64_t r = SDSL_HIP_NPOS; // outside SDSL's precondition (sd_vector.hpp:639)
        if (i >= 1 && i <= total)
        {
            if (BIT)
                r = quad_sd_select1<false>(v, s, i);
            else
                r = quad_sd_select0<false>(v, s, i);
        }
        if (s == 0)
            __builtin_nontemporal_store(r, out + q);
    }
}

// ---- one LANE per query (round 4) ------------------------------------------------------------------------------
// The quad kernels above spend four lanes on a chain of small dependent steps (580 instructions per lane and select_0, three
// fabric requests per rank) and are bound by exactly that.  Here a lane walks alone and takes only the SHORT road: a directory
// entry names a bucket at or in front of the wanted one and the entries in front of it, the zero that closes the bucket in front
// of the wanted one is looked for in the line where a uniformly filled vector has it (or its neighbour), the bucket's end is read
// off the same line.  Wherever that road does not lead — clustered data, runs that leave their line — the lane writes kSdRedo and
// k_sd_redo answers the query with the quad code; a vector on which that happens often keeps the quad kernels (sd_probe_lanes).
struct LaneLine
{
    uint64_t L;
    uint64_t w[8]; // w[0] = ones in front of the line
};
__device__ __forceinline__ void lane_load_line(const BvView & hv, uint64_t L, LaneLine & x)
{
    typedef unsigned long long v2u64 __attribute__((ext_vector_type(2)));
    const v2u64 * ln = reinterpret_cast<const v2u64 *>(hv.lines + L * kLW);
    x.L = L;
#pragma unroll
    for (int i = 0; i < 4; ++i)
    {
        const v2u64 t = ln[i];
        x.w[2 * i] = t.x;
        x.w[2 * i + 1] = t.y;
    }
}
// position of zero number z (0-based) if it lies in the line
__device__ __forceinline__ bool lane_zero_in_line(const LaneLine & x, uint64_t z, uint64_t & pos)
{
    const uint64_t zb = x.L * kDB - x.w[0];
    if (z < zb || z - zb >= kDB)
        return false;
    unsigned r = (unsigned)(z - zb);
#pragma unroll
    for (int i = 1; i < 8; ++i)
    {
        const unsigned c = 64u - popc64(x.w[i]);
        if (r < c)
        {
            pos = x.L * kDB + 64u * (unsigned)(i - 1) + sel64(~x.w[i], r + 1);
            return true;
        }
        r -= c;
    }
    return false;
}
// the first zero at or behind position p of the line (p inside it), NPOS if the line has none there
__device__ __forceinline__ uint64_t lane_next_zero(const LaneLine & x, uint64_t p)
{
    const unsigned off = (unsigned)(p - x.L * kDB);
#pragma unroll
    for (int i = 1; i < 8; ++i)
    {
        const unsigned base = 64u * (unsigned)(i - 1);
        if (base + 64u <= off)
            continue;
        uint64_t z = ~x.w[i];
        if (base < off)
            z &= ~lo_set(off - base);
        if (z)
            return x.L * kDB + base + (unsigned)__builtin_ctzll(z);
    }
    return SDSL_HIP_NPOS;
}
// zero number z of `high`, looked for in the line of x_est and the neighbour the header points to; the line it was found in stays in x
__device__ __forceinline__ bool lane_find_zero(const BvView & hv, uint64_t z, uint64_t x_est, LaneLine & x, uint64_t & pos)
{
    uint64_t L = x_est / kDB;
    L = L < hv.n_lines ? L : hv.n_lines - 1;
    lane_load_line(hv, L, x);
    if (lane_zero_in_line(x, z, pos))
        return true;
    const uint64_t zb = L * kDB - x.w[0];
    if (z < zb ? L == 0 : L + 1 >= hv.n_lines)
        return false;
    lane_load_line(hv, z < zb ? L - 1 : L + 1, x);
    return lane_zero_in_line(x, z, pos);
}
// the first zero at or behind `start` (x holds some line; reloaded when start lies elsewhere); looks one line further if needed
__device__ __forceinline__ uint64_t lane_zero_from(const BvView & hv, uint64_t start, LaneLine & x)
{
    const uint64_t L = start / kDB;
    if (L != x.L)
        lane_load_line(hv, L, x);
    uint64_t z = lane_next_zero(x, start);
    if (z == SDSL_HIP_NPOS && L + 1 < hv.n_lines)
    {
        lane_load_line(hv, L + 1, x);
        z = lane_next_zero(x, (L + 1) * kDB);
    }
    return z;
}

// rank_1(x) for x <= n (MODE 0) / "is bit x set" for x < n (MODE 1); false: not on the short road
template <int MODE>
__device__ __forceinline__ bool lane_sd_rank1(const SdView & v, uint64_t x, uint64_t & rank, bool & hit)
{
    const uint64_t h = x >> v.wl, val_low = x & lo_set(v.wl);
    LaneLine ln;
    uint64_t start = 0;
    if (h > 0)
    {
        const uint64_t j = h >> v.rank_shift, c = j << v.rank_shift, bef = v.rank_dir[j];
        // zero number h - 1 stands behind h - 1 zeros and the entries of the buckets below h: bef + those of the buckets c .. h - 1
        const uint64_t x_est = bef + (h - 1) + (((h - c) * (uint64_t)v.per_bucket_fx) >> 16);
        uint64_t p;
        if (!lane_find_zero(v.high, h - 1, x_est, ln, p))
            return false;
        start = p + 1;
    }
    else
        lane_load_line(v.high, 0, ln);
    const uint64_t begin = start - h;
    const uint64_t nz = lane_zero_from(v.high, start, ln);
    if (nz == SDSL_HIP_NPOS)
        return false; // (a run of entries longer than a line: a clustered bucket)
    const uint64_t end = nz - h;
    uint64_t lo = begin, hi = end;
    while (lo < hi)
    {
        const uint64_t mid = lo + ((hi - lo) >> 1);
        if (sd_low(v, mid) < val_low)
            lo = mid + 1;
        else
            hi = mid;
    }
    if (MODE == 1)
        hit = lo < end && sd_low(v, lo) == val_low;
    rank = lo;
    return true;
}

// select_0(i), i in [1, n - m]; false: not on the short road
__device__ __forceinline__ bool lane_sd_select0(const SdView & v, uint64_t i, uint64_t & pos_out)
{
    const uint64_t k = i - 1;
    const uint64_t e = v.sel0_dir[k >> v.sel0_shift];
    const uint64_t c = e >> 32, bef = e & 0xFFFFFFFFu;
    const uint64_t b = (k + bef) >> v.wl; // >= c, and f(b) <= k whatever lies between c and b
    uint64_t before = bef;
    LaneLine ln;
    ln.L = ~UINT64_C(0);
    if (b > c)
    {
        const uint64_t x_est = bef + c + (b - c - 1) + (((b - c) * (uint64_t)v.per_bucket_fx) >> 16);
        uint64_t p;
        if (!lane_find_zero(v.high, b - 1, x_est, ln, p))
            return false;
        before = p + 1 - b;
    }
    uint64_t fb = (b << v.wl) - before, start = before + b;
    uint64_t z = lane_zero_from(v.high, start, ln);
    if (z == SDSL_HIP_NPOS)
        return false;
    uint64_t bb = b, f_next = fb + (UINT64_C(1) << v.wl) - (z - start); // f(b + 1)
    for (int step = 0; step < 4 && k >= f_next; ++step)
    { // the entries between c and b push the answer into one of the next buckets
        const uint64_t st2 = z + 1;
        const uint64_t z2 = lane_zero_from(v.high, st2, ln);
        if (z2 == SDSL_HIP_NPOS)
            return false;
        ++bb;
        fb = f_next;
        f_next = fb + (UINT64_C(1) << v.wl) - (z2 - st2);
        start = st2;
        z = z2;
    }
    if (k >= f_next)
        return false; // (clustered data: the bracket search of the quad kernel)
    const uint64_t begin = (bb << v.wl) - fb, cnt = z - bb - begin, r = k - fb;
    uint64_t lo = 0, hi = cnt; // first t with low_t - t > r
    while (lo < hi)
    {
        const uint64_t mid = lo + ((hi - lo) >> 1);
        if (sd_low(v, begin + mid) <= r + mid)
            lo = mid + 1;
        else
            hi = mid;
    }
    pos_out = (bb << v.wl) + r + lo;
    return true;
}

template <int MODE> // 0: rank (bit b), 1: access
__global__ __launch_bounds__(kBlock) void k_sd_rank_lane(SdView v, int bit, const uint64_t * __restrict__ xq, uint64_t * __restrict__ out,
                                                         uint8_t * __restrict__ out8, uint64_t n, uint32_t * __restrict__ redo_count)
{
    for (uint64_t q = (uint64_t)blockIdx.x * kBlock + threadIdx.x; q < n; q += (uint64_t)gridDim.x * kBlock)
    {
        const uint64_t x = __builtin_nontemporal_load(xq + q);
        uint64_t r1 = 0;
        bool hit = false, ok = true;
        const bool inside = MODE == 1 ? x < v.n : x <= v.n;
        if (inside)
            ok = lane_sd_rank1<MODE>(v, x, r1, hit);
        if (!ok && redo_count)
            atomicAdd(redo_count, 1u);
        if (MODE == 1)
            out8[q] = !ok ? kSdRedo8 : (inside ? (hit ? 1 : 0) : 0xFF);
        else
            __builtin_nontemporal_store(!ok ? kSdRedo : (inside ? (bit ? r1 : x - r1) : SDSL_HIP_NPOS), out + q);
    }
}

__global__ __launch_bounds__(kBlock) void k_sd_select0_lane(SdView v, const uint64_t * __restrict__ iq, uint64_t * __restrict__ out, uint64_t n,
                                                            uint32_t * __restrict__ redo_count)
{
    const uint64_t total = v.n - v.m;
    for (uint64_t q = (uint64_t)blockIdx.x * kBlock + threadIdx.x; q < n; q += (uint64_t)gridDim.x * kBlock)
    {
        const uint64_t i = __builtin_nontemporal_load(iq + q);
        uint64_t r = SDSL_HIP_NPOS;
        if (i >= 1 && i <= total && !lane_sd_select0(v, i, r))
        {
            r = kSdRedo;
            if (redo_count)
                atomicAdd(redo_count, 1u);
        }
        __builtin_nontemporal_store(r, out + q);
    }
}

// the queries a lane kernel left: a wave looks at 64 answers, its quads take the marked ones sixteen at a time
template <int KIND> // 0: rank, 1: access, 2: select_0
__global__ __launch_bounds__(kBlock) void k_sd_redo(SdView v, int bit, const uint64_t * __restrict__ in, uint64_t * __restrict__ out,
                                                    uint8_t * __restrict__ out8, uint64_t n)
{
    const unsigned lane = threadIdx.x & 63u;
    const int s = (int)(lane & 3u);
    const unsigned gq = lane >> 2;
    for (uint64_t base = (uint64_t)blockIdx.x * kBlock + (threadIdx.x - lane); base < n; base += (uint64_t)gridDim.x * kBlock)
    {
        const uint64_t q = base + lane;
        const bool flag = q < n && (KIND == 1 ? out8[q] == kSdRedo8 : out[q] == kSdRedo);
        uint64_t mask = __ballot(flag);
        while (mask)
        { // (wave-uniform)
            int mybit = -1;
            for (unsigned j = 0; j < 16 && mask; ++j)
            {
                const int bpos = __builtin_ctzll(mask);
                mask &= mask - 1;
                if (j == gq)
                    mybit = bpos;
            }
            if (mybit >= 0)
            {
                const uint64_t qq = base + (unsigned)mybit;
                const uint64_t x = in[qq];
                if (KIND == 2)
                {
                    const uint64_t r = quad_sd_select0<false>(v, s, x);
                    if (s == 0)
                        out[qq] = r;
                }
                else
                {
                    bool hit = false;
                    const uint64_t r1 = quad_sd_rank1<false>(v, s, x, &hit);
                    if (s == 0)
                    {
                        if (KIND == 1)
                            out8[qq] = hit ? 1 : 0;
                        else
                            out[qq] = bit ? r1 : x - r1;
                    }
                }
            }
        }
    }
}

// rank directory: entries with a high part below j << shift = (position of zero number (j << shift) - 1) + 1 - (j << shift)
__global__ __launch_bounds__(kBlock) void k_sd_rank_dir(SdView v, uint32_t shift, uint64_t n_entries, uint64_t * __restrict__ dir)
{
    const int s = threadIdx.x & (kG - 1);
    const unsigned gq = threadIdx.x / kG;
    for (uint64_t base = (uint64_t)blockIdx.x * kQPB; base < n_entries; base += (uint64_t)gridDim.x * kQPB)
    {
        const uint64_t j = base + gq;
        if (j >= n_entries)
            continue;
        const uint64_t c = j << shift;
        uint64_t val = 0;
        if (c > 0)
        {
            bool mine;
            const uint64_t p = quad_select<0, false>(v.high, s, c - 1, mine);
            val = quad_gather_u64(p, mine) + 1 - c;
        }
        if (s == 0)
            dir[j] = val;
    }
}
__global__ __launch_bounds__(256) void k_sd_probe_args(uint64_t n_args, uint64_t mod, uint64_t add, uint64_t * __restrict__ a)
{
    const uint64_t i = (uint64_t)blockIdx.x * 256 + threadIdx.x;
    if (i < n_args)
    {
        uint64_t x = (i + 1) * UINT64_C(0x9E3779B97F4A7C15);
        x ^= x >> 29;
        x *= UINT64_C(0xBF58476D1CE4E5B9);
        x ^= x >> 32;
        a[i] = add + x % mod;
    }
}

// ---- construction --------------------------------------------------------------------------------------------
// entry i of the sorted position list: wl low bits into `low`, a one at (pos >> wl) + i into `high`
__global__ __launch_bounds__(256) void k_sd_fill(const uint64_t * __restrict__ pos, uint64_t m, uint32_t wl,
                                                 unsigned long long * __restrict__ low,
                                                 unsigned long long * __restrict__ high)
{
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < m; i += (uint64_t)gridDim.x * blockDim.x)
    {
        const uint64_t p = pos[i];
        const uint64_t lv = p & lo_set(wl), at = i * wl;
        const unsigned off = (unsigned)(at & 63);
        atomicOr(&low[at >> 6], (unsigned long long)(lv << off));
        if (off + wl > 64)
            atomicOr(&low[(at >> 6) + 1], (unsigned long long)(lv >> (64 - off)));
        const uint64_t hp = (p >> wl) + i;
        atomicOr(&high[hp >> 6], 1ull << (hp & 63));
    }
}

// 0 if pos[0..m) is strictly increasing and below n
__global__ __launch_bounds__(256) void k_sd_check_sorted(const uint64_t * __restrict__ pos, uint64_t m, uint64_t n,
                                                         unsigned * __restrict__ bad)
{
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < m; i += (uint64_t)gridDim.x * blockDim.x)
        if (pos[i] >= n || (i && pos[i - 1] >= pos[i]))
            atomicOr(bad, 1u);
}

__global__ __launch_bounds__(256) void k_sd_iota1(uint64_t * __restrict__ a, uint64_t m)
{
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < m; i += (uint64_t)gridDim.x * blockDim.x)
        a[i] = i + 1;
}

static unsigned hi64_host(uint64_t x)
{ // bits::hi: position of the most significant set bit, 0 for x == 0
    unsigned r = 0;
    while (x >>= 1)
        ++r;
    return r;
}

static void sd_finish_view(SdHost & h, uint64_t n, uint64_t m, uint32_t wl)
{
    h.view.high = h.high.view;
    h.view.low = h.low.as<uint64_t>();
    h.view.n = n;
    h.view.m = m;
    h.view.wl = wl;
    h.view.sel0_dir = nullptr;
    h.view.sel0_shift = 0;
    h.view.rank_dir = nullptr;
    h.view.rank_shift = 0;
    h.lane_rank_ok = h.lane_sel0_ok = false;
    {
        const double per = (double)m / (double)((n >> wl) + 1) * 65536.0;
        h.view.per_bucket_fx = per > 4e9 ? 0xFFFFFFFFu : (uint32_t)per;
    }
    h.low_width_when_empty = wl;
}

// The select_0 directory: one entry per 2^shift zeros, shift >= wl + 2 and large enough to keep the directory within 2^19
// entries (4 MiB: mostly resident in the XCDs' L2) and within 1/16 of the structure.  Measured on 2^28 ones in a universe of
// 2^40 (7.8 G/s without it): 2^15 entries 10.8, 2^17 10.9, 2^19 11.9, 2^22 12.4 G/s at +0.1 / +0.2 / +0.8 / +3.2 % of space — the
// kernel is bound by its instructions (about 580 per lane and query, 2.2 fabric requests), not by the lookups, and a coarser
// directory only makes the window guess of quad_sd_select0 miss more often.  Not built when buckets or entries do not fit 32
// bits, or when there are few zeros.  Errors leave the vector without one.
static void sd_build_sel0_dir(SdHost & h)
{
    const SdView & v = h.view;
    const uint64_t zeros = v.n - v.m;
    if (getenv("SDSL_HIP_SD_NO_SEL0_DIR") || zeros < (UINT64_C(1) << 20) || v.m >= (UINT64_C(1) << 32) || (v.n >> v.wl) + 2 >= (UINT64_C(1) << 32))
        return;
    uint32_t shift = v.wl + 2;
    const char * cap_env = getenv("SDSL_HIP_SD_SEL0_LOG2"); // (experiment knob: log2 of the directory's size in entries)
    const unsigned cap_log2 = cap_env && atoi(cap_env) >= 4 && atoi(cap_env) <= 26 ? (unsigned)atoi(cap_env) : 19u;
    const size_t budget = (h.high.device_bytes() + h.low.bytes) / 16;
    while ((((zeros - 1) >> shift) + 1) * 8 > budget || (((zeros - 1) >> shift) + 1) > (UINT64_C(1) << cap_log2))
        ++shift;
    const uint64_t n_entries = ((zeros - 1) >> shift) + 1;
    if (h.sel0_dir.alloc(n_entries * 8) != SDSL_HIP_OK)
        return;
    hipLaunchKernelGGL(k_sd_sel0_dir, dim3(grid_for(n_entries, kQPB, 256u * 8u)), dim3(kBlock), 0, 0, v, shift, n_entries, h.sel0_dir.as<uint64_t>());
    if (hipGetLastError() != hipSuccess || hipDeviceSynchronize() != hipSuccess)
    {
        h.sel0_dir.release();
        return;
    }
    h.view.sel0_dir = h.sel0_dir.as<uint64_t>();
    h.view.sel0_shift = shift;
}

// The rank directory of the lane kernels (at most 2^19 entries, like the select_0 directory) and the verdict whether this vector's
// queries stay on the lane kernels' short road: 2^14 pseudo-random arguments through either kernel, at most 3 % marked for the
// quad code.  (Clustered data: the window guess misses, runs leave their line — there the quad kernels, with their bracket search
// and select directories, are the faster road.)  Errors leave the vector on the quad kernels.
static void sd_build_lane_tables(SdHost & h)
{
    const SdView & v = h.view;
    if (getenv("SDSL_HIP_SD_NO_LANES") || v.m == 0 || v.n < (UINT64_C(1) << 16))
        return;
    const uint64_t n_buckets = (v.n >> v.wl) + 1;
    uint32_t shift = 0;
    while ((n_buckets >> shift) + 1 > (UINT64_C(1) << 19))
        ++shift;
    const uint64_t n_entries = (n_buckets >> shift) + 1;
    if (h.rank_dir.alloc(n_entries * 8) != SDSL_HIP_OK)
        return;
    hipLaunchKernelGGL(k_sd_rank_dir, dim3(grid_for(n_entries, kQPB, 256u * 8u)), dim3(kBlock), 0, 0, v, shift, n_entries, h.rank_dir.as<uint64_t>());
    if (hipGetLastError() != hipSuccess || hipDeviceSynchronize() != hipSuccess)
    {
        h.rank_dir.release();
        return;
    }
    h.view.rank_dir = h.rank_dir.as<uint64_t>();
    h.view.rank_shift = shift;
    constexpr uint64_t kProbe = 1u << 14;
    DevBuf args, outs, cnt;
    if (args.alloc(kProbe * 8) != SDSL_HIP_OK || outs.alloc(kProbe * 8) != SDSL_HIP_OK || cnt.alloc(8, true) != SDSL_HIP_OK)
        return;
    uint32_t redo[2] = {~0u, ~0u};
    hipLaunchKernelGGL(k_sd_probe_args, dim3(kProbe / 256), dim3(256), 0, 0, kProbe, v.n + 1, (uint64_t)0, args.as<uint64_t>());
    hipLaunchKernelGGL((k_sd_rank_lane<0>), dim3(kProbe / kBlock), dim3(kBlock), 0, 0, h.view, 1, args.as<uint64_t>(), outs.as<uint64_t>(),
                       (uint8_t *)nullptr, kProbe, cnt.as<uint32_t>());
    const uint64_t zeros = v.n - v.m;
    if (h.view.sel0_dir && zeros)
    {
        hipLaunchKernelGGL(k_sd_probe_args, dim3(kProbe / 256), dim3(256), 0, 0, kProbe, zeros, (uint64_t)1, args.as<uint64_t>());
        hipLaunchKernelGGL(k_sd_select0_lane, dim3(kProbe / kBlock), dim3(kBlock), 0, 0, h.view, args.as<uint64_t>(), outs.as<uint64_t>(), kProbe,
                           cnt.as<uint32_t>() + 1);
    }
    if (hipGetLastError() != hipSuccess || hipMemcpy(redo, cnt.p, 8, hipMemcpyDeviceToHost) != hipSuccess)
        return;
    const uint32_t limit = (uint32_t)(kProbe * 3 / 100);
    h.lane_rank_ok = redo[0] <= limit;
    h.lane_sel0_ok = h.view.sel0_dir && zeros && redo[1] <= limit;
    if (getenv("SDSL_HIP_TRACE_BUILD"))
        fprintf(stderr, "[sdsl_hip] sd_vector: lane kernels: %u / %u of %llu probe queries left their short road (rank / select_0): %s / %s\n", redo[0],
                redo[1], (unsigned long long)kProbe, h.lane_rank_ok ? "lanes" : "quads", h.lane_sel0_ok ? "lanes" : "quads");
}

// the lane kernels mark what they leave to k_sd_redo in `out` and k_sd_redo reads the marked queries' ARGUMENTS again: argument and
// answer arrays must not share a byte (any overlap, not only in == out; the quad kernels read an argument before they write its answer
// and take either)
static bool sd_ranges_disjoint(const void * in, const void * out, uint64_t n, size_t out_elem)
{
    const uintptr_t a = (uintptr_t)in, b = (uintptr_t)out;
    return a + n * 8 <= b || b + n * out_elem <= a;
}

// enqueue rank / access / select_0 on whichever kernels the vector takes
static void sd_launch_rank(const SdHost & h, int mode, int bit, const uint64_t * d_in, uint64_t * d_out, uint8_t * d_out8, uint64_t n, hipStream_t st)
{
    if (h.lane_rank_ok && sd_ranges_disjoint(d_in, mode == 1 ? (const void *)d_out8 : (const void *)d_out, n, mode == 1 ? 1 : 8))
    {
        const dim3 g(grid_for(n, kBlock, 256u * 8u)), gr(grid_for(n, kBlock, 256u * 4u));
        if (mode == 1)
        {
            hipLaunchKernelGGL((k_sd_rank_lane<1>), g, dim3(kBlock), 0, st, h.view, bit, d_in, d_out, d_out8, n, (uint32_t *)nullptr);
            hipLaunchKernelGGL((k_sd_redo<1>), gr, dim3(kBlock), 0, st, h.view, bit, d_in, d_out, d_out8, n);
        }
        else
        {
            hipLaunchKernelGGL((k_sd_rank_lane<0>), g, dim3(kBlock), 0, st, h.view, bit, d_in, d_out, d_out8, n, (uint32_t *)nullptr);
            hipLaunchKernelGGL((k_sd_redo<0>), gr, dim3(kBlock), 0, st, h.view, bit, d_in, d_out, d_out8, n);
        }
        return;
    }
    if (mode == 1)
        hipLaunchKernelGGL((k_sd_rank<1>), dim3(grid_for(n, kQPB, 256u * 8u)), dim3(kBlock), 0, st, h.view, bit, d_in, d_out, d_out8, n);
    else
        hipLaunchKernelGGL((k_sd_rank<0>), dim3(grid_for(n, kQPB, 256u * 8u)), dim3(kBlock), 0, st, h.view, bit, d_in, d_out, d_out8, n);
}
static void sd_launch_select(const SdHost & h, int bit, const uint64_t * d_in, uint64_t * d_out, uint64_t n, hipStream_t st)
{
    if (bit)
        hipLaunchKernelGGL((k_sd_select<1>), dim3(grid_for(n, kQPB, 256u * 8u)), dim3(kBlock), 0, st, h.view, d_in, d_out, n);
    else if (h.lane_sel0_ok && sd_ranges_disjoint(d_in, d_out, n, 8))
    {
        hipLaunchKernelGGL(k_sd_select0_lane, dim3(grid_for(n, kBlock, 256u * 8u)), dim3(kBlock), 0, st, h.view, d_in, d_out, n, (uint32_t *)nullptr);
        hipLaunchKernelGGL((k_sd_redo<2>), dim3(grid_for(n, kBlock, 256u * 4u)), dim3(kBlock), 0, st, h.view, 0, d_in, d_out, (uint8_t *)nullptr, n);
    }
    else
        hipLaunchKernelGGL((k_sd_select<0>), dim3(grid_for(n, kQPB, 256u * 8u)), dim3(kBlock), 0, st, h.view, d_in, d_out, n);
}

// sd_vector(begin, end) with an explicit size (sd_vector.hpp:217-305; the iterator constructor takes size = last + 1)
static sdsl_hip_status sd_build_from_device_positions(SdHost & h, const uint64_t * d_pos, uint64_t m, uint64_t n, int device)
{
    h.device = device;
    // validate before anything is sized from (m, n): m strictly increasing positions below n need m <= n
    if (m > n)
    {
        set_error("sd_vector: %llu positions cannot be strictly increasing below the size %llu (sd_vector.hpp:266-269)",
                  (unsigned long long)m, (unsigned long long)n);
        return SDSL_HIP_ERR_INVALID;
    }
    if (m)
    {
        DevBuf bad;
        SH_TRY(bad.alloc(4, true));
        hipLaunchKernelGGL(k_sd_check_sorted, dim3(grid_for(m, 256, 65536)), dim3(256), 0, 0, d_pos, m, n, bad.as<unsigned>());
        SH_HIP(hipGetLastError());
        unsigned hb = 0;
        SH_HIP(hipMemcpy(&hb, bad.p, 4, hipMemcpyDeviceToHost));
        if (hb)
        {
            set_error("sd_vector: the positions must be strictly increasing and smaller than the size (sd_vector.hpp:266-269)");
            return SDSL_HIP_ERR_INVALID;
        }
    }
    unsigned logm = hi64_host(m) + 1, logn = hi64_host(n) + 1;
    if (logm == logn)
        --logm; // to ensure logn - logm > 0 (:226-229)
    const uint32_t wl = logn - logm;
    const uint64_t high_bits = m + (UINT64_C(1) << logm);
    DevBuf d_high;
    SH_TRY(d_high.alloc((((high_bits + 63) >> 6) + 1) * 8, true));
    SH_TRY(h.low.alloc((((m * wl + 63) >> 6) + 2) * 8, true));
    if (m)
    {
        hipLaunchKernelGGL(k_sd_fill, dim3(grid_for(m, 256, 65536)), dim3(256), 0, 0, d_pos, m, wl,
                           h.low.as<unsigned long long>(), d_high.as<unsigned long long>());
        SH_HIP(hipGetLastError());
    }
    h.high.device = device;
    SH_TRY(bv_build_from_device_words(h.high, d_high.as<uint64_t>(), high_bits, SDSL_HIP_BV_SELECT1 | SDSL_HIP_BV_SELECT0,
                                      default_sel_shift()));
    SH_HIP(hipDeviceSynchronize());
    sd_finish_view(h, n, m, wl);
    sd_build_sel0_dir(h);
    sd_build_lane_tables(h);
    return SDSL_HIP_OK;
}

// sd_vector::load (sd_vector.hpp:447-456): size, wl, low, high, the two select supports of high (skipped)
static sdsl_hip_status sd_build_from_stream(SdHost & h, StreamReader & rd, int device)
{
    h.device = device;
    uint64_t n = 0;
    uint8_t wl = 0;
    HostIntVec low, high;
    if (!rd.u64(n) || !rd.raw(&wl, 1) || !rd.int_vector(low) || !rd.int_vector(high, 1) || !rd.skip_select_mcl()
        || !rd.skip_select_mcl())
    {
        set_error("malformed sd_vector stream (offset %zu of %zu)", rd.pos, rd.len);
        return SDSL_HIP_ERR_FORMAT;
    }
    const uint64_t m = low.size();
    uint64_t ones = 0;
    for (uint64_t w = 0; w < (high.bit_size + 63) >> 6; ++w)
        ones += popc64(w == (high.bit_size >> 6) && (high.bit_size & 63) ? high.words[w] & lo_set((unsigned)(high.bit_size & 63))
                                                                        : high.words[w]);
    // the kernels index with these: every rank needs the (x >> wl) + 1-th zero of high, every select the i-th one
    const bool sane = wl >= 1 && wl <= 63 && (m == 0 || low.width == wl) && ones == m && high.bit_size >= m
                      && high.bit_size - m > (n >> wl);
    if (!sane)
    {
        set_error("sd_vector stream: size, wl, low and high do not describe one vector");
        return SDSL_HIP_ERR_FORMAT;
    }
    DevBuf d_high;
    const uint64_t hw = (high.bit_size + 63) >> 6;
    SH_TRY(d_high.alloc((hw + 1) * 8, true));
    if (hw)
        SH_HIP(hipMemcpy(d_high.p, high.words.data(), hw * 8, hipMemcpyHostToDevice));
    const uint64_t lw = (m * wl + 63) >> 6;
    SH_TRY(h.low.alloc((lw + 2) * 8, true));
    if (lw)
        SH_HIP(hipMemcpy(h.low.p, low.words.data(), lw * 8, hipMemcpyHostToDevice));
    h.high.device = device;
    SH_TRY(bv_build_from_device_words(h.high, d_high.as<uint64_t>(), high.bit_size, SDSL_HIP_BV_SELECT1 | SDSL_HIP_BV_SELECT0,
                                      default_sel_shift()));
    SH_HIP(hipDeviceSynchronize());
    sd_finish_view(h, n, m, wl);
    sd_build_sel0_dir(h);
    sd_build_lane_tables(h);
    return SDSL_HIP_OK;
}

} // namespace sdslhip

using namespace sdslhip;

struct sdsl_hip_sd_s
{
    SdHost h;
    uint64_t uid = next_handle_uid(); // key of the serialiser's size-query cache
};

extern "C" {

static sdsl_hip_status sdsl_hip_sd_create_from_positions_impl(const uint64_t * positions, uint64_t m, uint64_t n_bits, int32_t device,
                                                  sdsl_hip_sd_t * out)
{
    if (!out || (!positions && m))
    {
        set_error("sd_create_from_positions: null argument");
        return SDSL_HIP_ERR_INVALID;
    }
    *out = nullptr;
    SH_TRY(check_device(device));
    sdsl_hip_sd_s * r = new (std::nothrow) sdsl_hip_sd_s();
    if (!r)
        return SDSL_HIP_ERR_NOMEM;
    Staged p;
    sdsl_hip_status st = p.in(positions, m * 8, nullptr);
    if (st == SDSL_HIP_OK)
        st = sd_build_from_device_positions(r->h, (const uint64_t *)p.dev, m, n_bits, device);
    if (st != SDSL_HIP_OK)
    {
        delete r;
        return st;
    }
    *out = r;
    return SDSL_HIP_OK;
}
// no exception crosses the C ABI: a malformed stream or an exhausted host becomes a status code
sdsl_hip_status sdsl_hip_sd_create_from_positions(const uint64_t * positions, uint64_t m, uint64_t n_bits, int32_t device,
                                                  sdsl_hip_sd_t * out)
{
    return guarded("sd_create_from_positions", [&] { return sdsl_hip_sd_create_from_positions_impl(positions, m, n_bits, device, out); });
}

static sdsl_hip_status sdsl_hip_sd_create_impl(const uint64_t * words, uint64_t n_bits, int32_t device, sdsl_hip_sd_t * out)
{
    if (!out || (!words && n_bits))
    {
        set_error("sd_create: null argument");
        return SDSL_HIP_ERR_INVALID;
    }
    *out = nullptr;
    SH_TRY(check_device(device));
    // positions of the ones through a temporary plain vector: select_1(1..m)
    BvHost tmp;
    tmp.device = device;
    Staged w;
    SH_TRY(w.in(words, ((n_bits + 63) >> 6) * 8, nullptr));
    SH_TRY(bv_build_from_device_words(tmp, (const uint64_t *)w.dev, n_bits, SDSL_HIP_BV_SELECT1, default_sel_shift()));
    const uint64_t m = tmp.view.ones;
    DevBuf d_pos;
    SH_TRY(d_pos.alloc((m + 1) * 8));
    if (m)
    {
        hipLaunchKernelGGL(k_sd_iota1, dim3(grid_for(m, 256, 65536)), dim3(256), 0, 0, d_pos.as<uint64_t>(), m);
        SH_HIP(hipGetLastError());
        SH_TRY(bv_launch_select(tmp.view, 1, d_pos.as<uint64_t>(), m, d_pos.as<uint64_t>(), nullptr));
    }
    sdsl_hip_sd_s * r = new (std::nothrow) sdsl_hip_sd_s();
    if (!r)
        return SDSL_HIP_ERR_NOMEM;
    sdsl_hip_status st = sd_build_from_device_positions(r->h, d_pos.as<uint64_t>(), m, n_bits, device);
    if (st != SDSL_HIP_OK)
    {
        delete r;
        return st;
    }
    *out = r;
    return SDSL_HIP_OK;
}
// no exception crosses the C ABI: a malformed stream or an exhausted host becomes a status code
sdsl_hip_status sdsl_hip_sd_create(const uint64_t * words, uint64_t n_bits, int32_t device, sdsl_hip_sd_t * out)
{
    return guarded("sd_create", [&] { return sdsl_hip_sd_create_impl(words, n_bits, device, out); });
}

static sdsl_hip_status sdsl_hip_sd_create_from_sdsl_impl(const void * bytes, size_t len, int32_t device, sdsl_hip_sd_t * out,
                                             size_t * consumed)
{
    if (!out || !bytes)
    {
        set_error("sd_create_from_sdsl: null argument");
        return SDSL_HIP_ERR_INVALID;
    }
    *out = nullptr;
    SH_TRY(check_device(device));
    sdsl_hip_sd_s * r = new (std::nothrow) sdsl_hip_sd_s();
    if (!r)
        return SDSL_HIP_ERR_NOMEM;
    StreamReader rd(bytes, len);
    sdsl_hip_status st = sd_build_from_stream(r->h, rd, device);
    if (st != SDSL_HIP_OK)
    {
        delete r;
        return st;
    }
    if (consumed)
        *consumed = rd.pos;
    *out = r;
    return SDSL_HIP_OK;
}
// no exception crosses the C ABI: a malformed stream or an exhausted host becomes a status code
sdsl_hip_status sdsl_hip_sd_create_from_sdsl(const void * bytes, size_t len, int32_t device, sdsl_hip_sd_t * out,
                                             size_t * consumed)
{
    return guarded("sd_create_from_sdsl", [&] { return sdsl_hip_sd_create_from_sdsl_impl(bytes, len, device, out, consumed); });
}

// sd_vector<>::serialize (sd_vector.hpp:435-445): size, wl, low, high, select_support_mcl<1> and <0> of high
static sdsl_hip_status sdsl_hip_sd_serialize_impl(sdsl_hip_sd_t v, void * buf, size_t cap, size_t * written)
{
    if (!v)
    {
        set_error("sd_serialize: null handle");
        return SDSL_HIP_ERR_INVALID;
    }
    sdsl_hip_status cached;
    if (deliver_cached(v->uid, 0, buf, cap, written, cached))
        return cached;
    SH_HIP(hipSetDevice(v->h.device));
    const SdView & sv = v->h.view;
    const uint64_t hb = sv.high.n_bits, HW = (hb + 63) >> 6, LW = (sv.m * sv.wl + 63) >> 6;
    std::vector<uint64_t> high(HW + 1, 0), low(LW + 1, 0);
    if (HW)
    {
        DevBuf d;
        SH_TRY(d.alloc(HW * 8));
        SH_TRY(bv_export_words_device(sv.high, d.as<uint64_t>(), HW, nullptr));
        SH_HIP(hipMemcpy(high.data(), d.p, HW * 8, hipMemcpyDeviceToHost));
    }
    if (LW)
        SH_HIP(hipMemcpy(low.data(), sv.low, LW * 8, hipMemcpyDeviceToHost));
    StreamWriter w;
    w.u64(sv.n);
    const uint8_t wl = (uint8_t)sv.wl;
    w.raw(&wl, 1);
    w.int_vector(low.data(), sv.m * sv.wl, sv.m ? wl : (uint8_t)v->h.low_width_when_empty);
    w.int_vector(high.data(), hb, 1);
    select_mcl_serialize_host(high.data(), hb, 1, w);
    select_mcl_serialize_host(high.data(), hb, 0, w);
    return deliver_and_cache(v->uid, 0, w, buf, cap, written);
}
// no exception crosses the C ABI: a malformed stream or an exhausted host becomes a status code
sdsl_hip_status sdsl_hip_sd_serialize(sdsl_hip_sd_t v, void * buf, size_t cap, size_t * written)
{
    return guarded("sd_serialize", [&] { return sdsl_hip_sd_serialize_impl(v, buf, cap, written); });
}

sdsl_hip_status sdsl_hip_sd_destroy(sdsl_hip_sd_t v)
{
    if (!v)
        return SDSL_HIP_OK;
    (void)hipSetDevice(v->h.device);
    delete v;
    return SDSL_HIP_OK;
}
uint64_t sdsl_hip_sd_size(sdsl_hip_sd_t v)
{
    return v ? v->h.view.n : 0;
}
uint64_t sdsl_hip_sd_ones(sdsl_hip_sd_t v)
{
    return v ? v->h.view.m : 0;
}
uint32_t sdsl_hip_sd_lane_kernels(sdsl_hip_sd_t v)
{
    return v ? (v->h.lane_rank_ok ? 1u : 0u) | (v->h.lane_sel0_ok ? 2u : 0u) : 0u;
}
uint32_t sdsl_hip_sd_low_width(sdsl_hip_sd_t v)
{
    return v ? v->h.view.wl : 0;
}
uint64_t sdsl_hip_sd_device_bytes(sdsl_hip_sd_t v)
{
    return v ? v->h.device_bytes() : 0;
}

sdsl_hip_status sdsl_hip_sd_rank_batch(sdsl_hip_sd_t v, int32_t bit, const uint64_t * idx, uint64_t n, uint64_t * out,
                                       void * stream)
{
    if (!v || (bit != 0 && bit != 1) || (n && (!idx || !out)))
    {
        set_error("sd_rank_batch: invalid argument");
        return SDSL_HIP_ERR_INVALID;
    }
    hipStream_t s = (hipStream_t)stream;
    SH_HIP(hipSetDevice(v->h.device));
    if (n == 0)
        return SDSL_HIP_OK;
    if (n >= kPipelineMinQueries && !is_device_ptr(idx) && !is_device_ptr(out))
    { // host arrays on both sides: chunked over two streams (common.hpp host_pipeline_u64)
        const SdHost * hp = &v->h;
        return host_pipeline_u64(v->h.device, idx, out, n,
                                 [hp, bit](const uint64_t * d_in, uint64_t * d_out, uint64_t cnt, hipStream_t st) -> sdsl_hip_status
                                 {
                                     sd_launch_rank(*hp, 0, bit, d_in, d_out, (uint8_t *)nullptr, cnt, st);
                                     SH_HIP(hipGetLastError());
                                     return SDSL_HIP_OK;
                                 });
    }
    Staged in, o;
    SH_TRY(in.in(idx, n * 8, s));
    SH_TRY(o.out(out, n * 8));
    {
        KernelTimer t(s);
        sd_launch_rank(v->h, 0, bit, (const uint64_t *)in.dev, (uint64_t *)o.dev, (uint8_t *)nullptr, n, s);
    }
    SH_HIP(hipGetLastError());
    SH_TRY(o.finish(s));
    if (in.host && !o.host)
        SH_HIP(hipStreamSynchronize(s));
    return SDSL_HIP_OK;
}

sdsl_hip_status sdsl_hip_sd_access_batch(sdsl_hip_sd_t v, const uint64_t * idx, uint64_t n, uint8_t * out, void * stream)
{
    if (!v || (n && (!idx || !out)))
    {
        set_error("sd_access_batch: invalid argument");
        return SDSL_HIP_ERR_INVALID;
    }
    hipStream_t s = (hipStream_t)stream;
    SH_HIP(hipSetDevice(v->h.device));
    if (n == 0)
        return SDSL_HIP_OK;
    Staged in, o;
    SH_TRY(in.in(idx, n * 8, s));
    SH_TRY(o.out(out, n));
    {
        KernelTimer t(s);
        sd_launch_rank(v->h, 1, 1, (const uint64_t *)in.dev, (uint64_t *)nullptr, (uint8_t *)o.dev, n, s);
    }
    SH_HIP(hipGetLastError());
    SH_TRY(o.finish(s));
    if (in.host && !o.host)
        SH_HIP(hipStreamSynchronize(s));
    return SDSL_HIP_OK;
}

sdsl_hip_status sdsl_hip_sd_select_batch(sdsl_hip_sd_t v, int32_t bit, const uint64_t * i, uint64_t n, uint64_t * out,
                                         void * stream)
{
    if (!v || (bit != 0 && bit != 1) || (n && (!i || !out)))
    {
        set_error("sd_select_batch: invalid argument");
        return SDSL_HIP_ERR_INVALID;
    }
    hipStream_t s = (hipStream_t)stream;
    SH_HIP(hipSetDevice(v->h.device));
    if (n == 0)
        return SDSL_HIP_OK;
    if (n >= kPipelineMinQueries && !is_device_ptr(i) && !is_device_ptr(out))
    {
        const SdHost * hp = &v->h;
        return host_pipeline_u64(v->h.device, i, out, n,
                                 [hp, bit](const uint64_t * d_in, uint64_t * d_out, uint64_t cnt, hipStream_t st) -> sdsl_hip_status
                                 {
                                     sd_launch_select(*hp, bit, d_in, d_out, cnt, st);
                                     SH_HIP(hipGetLastError());
                                     return SDSL_HIP_OK;
                                 });
    }
    Staged in, o;
    SH_TRY(in.in(i, n * 8, s));
    SH_TRY(o.out(out, n * 8));
    {
        KernelTimer t(s);
        sd_launch_select(v->h, bit, (const uint64_t *)in.dev, (uint64_t *)o.dev, n, s);
    }
    SH_HIP(hipGetLastError());
    SH_TRY(o.finish(s));
    if (in.host && !o.host)
        SH_HIP(hipStreamSynchronize(s));
    return SDSL_HIP_OK;
}
}
