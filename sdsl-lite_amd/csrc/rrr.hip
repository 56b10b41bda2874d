// rrr.hip — rrr_vector<63, int_vector<>, 32> on the device: H0-compressed bit vector with
// class/offset coding (block size 63, one sample per 32 blocks), its rank / select / access.
//
// Reference semantics reproduced (bit-identical answers; SDSL's serialised arrays are accepted as
// input, the device layout differs):
//   rrr_vector(bit_vector const&)        rrr_vector.hpp:158-270   (+ bin_to_nr rrr_helper.hpp:346-366)
//   rank_support_rrr<b,63>::rank         rrr_vector.hpp:503-544
//   select_support_rrr<b,63>::select     rrr_vector.hpp:639-726   (overflow -> size())
//   rrr_vector::operator[]               rrr_vector.hpp:276-298
//   binomial table / space[]             rrr_helper.hpp:194-237, 262-295
//
// Device layout: one 128-byte RECORD per 34 blocks (2142 bits; SDSL samples every 32 blocks — the parser
// and the serialiser regroup)
//   word 0      ones before the record's first block              (what SDSL's m_rank holds every 32 blocks)
//   word 1      bits 0..47  WORD pointer into the overflow stream (the role of SDSL's m_btnrp)
//               bits 48..59 ones inside the record
//   word 2      prefix word: ones and offset bits in blocks [0,9), [0,18), [0,27) (rrr_device.hpp rrr_pack_prefix)
//   words 3..6  the 36 block classes, nine 7-bit fields per word, inversion already undone (SDSL m_bt + m_invert)
//   words 7..15 the first 576 bits of the record's offsets (SDSL m_btnr)
// plus a stream that holds, per record, only the offset bits beyond those 576 (in whole words).  An L2 miss costs the same for 64 and 128 bytes on MI355X (the
// random-request rate is the bound, profiles/gather_probe_r01.txt), so packing header, classes and
// the usually-sufficient head of the offsets into ONE line turns SDSL's five scattered arrays into
// one fetch for most queries; only long offset runs touch the stream (second fetch).  The prefix word lets ONE
// lane locate a block with byte arithmetic on a single class word, so queries run one per lane.
#include <algorithm>
#include <atomic>
#include <mutex>
#include <thread>

#include "rrr_host.hpp"

namespace sdslhip {

// ---- host: tables + encoder --------------------------------------------------------------------
static RrrTables g_host_tables;
static std::atomic<bool> g_tables_ready{false};

static const RrrTables & host_tables()
{
    if (!g_tables_ready.load())
    {
        static std::mutex m;
        std::lock_guard<std::mutex> lk(m);
        if (!g_tables_ready.load())
        {
            RrrTables & T = g_host_tables;
            memset(&T, 0, sizeof T);
            for (int m2 = 0; m2 < 64; ++m2)
            {
                T.binom[m2][0] = 1;
                for (int k = 1; k <= m2; ++k)
                    T.binom[m2][k] = T.binom[m2 - 1][k - 1] + (k <= m2 - 1 ? T.binom[m2 - 1][k] : 0);
            }
            for (int k = 0; k < 64; ++k)
            {
                uint64_t c = T.binom[63][k];
                T.sdsl_space[k] = c == 1 ? 0 : (uint8_t)(hi64(c) + 1);
                T.space[k] = T.sdsl_space[k]; // (per vector: tables_for)
            }
            g_tables_ready.store(true);
        }
    }
    return g_host_tables;
}

// the tables of a vector whose classes t+1 .. 62-t are stored raw (rrr_device.hpp)
static RrrTables tables_for(unsigned sparse_max)
{
    RrrTables T = host_tables();
    for (unsigned k = sparse_max + 1; k + sparse_max < kRrrBS; ++k)
        T.space[k] = (uint8_t)kRrrBS;
    for (unsigned b = 0; b < 256; ++b) // (field 15 of a slim record is the escape: 63 bits, which is what space[15] says anyway)
        T.space2[b] = (uint8_t)(T.space[b & 15] + T.space[b >> 4]);
    return T;
}

// geometry of the two record formats for the host-side code (rrr_device.hpp: RrrFmtW, RrrFmtS)
struct RrrGeo
{
    unsigned fmt, K, GRP, CLSW, NCW, CLS0, INL0, INLW, INLB;
    uint64_t SB;
    unsigned field_value(unsigned k) const { return fmt && k >= kEsc ? kEsc : k; }             // what the class word holds
    unsigned dev_len(const RrrTables & T, unsigned k) const { return fmt && k >= kEsc ? kRrrBS : T.space[k]; } // bits of its field
};
static RrrGeo geo_of(unsigned fmt)
{
    if (fmt)
        return RrrGeo{1, RrrFmtS::K, RrrFmtS::GRP, 4, RrrFmtS::NCW, RrrFmtS::CLS0, RrrFmtS::INL0, RrrFmtS::INLW, RrrFmtS::INLB, RrrFmtS::SB};
    return RrrGeo{0, kRecK, kGrp, kClsW, 4, kRecClasses, kRecInline, kInlineWords, kInlineBits, kRecSB};
}

// Which record format a stand-alone vector gets (option "rrr_format" / SDSL_HIP_RRR_FORMAT overrides): slim when escapes are rare
// (at most one block in 512 has 15 or more ones) and the estimate of its size — records plus the offset bits beyond a record's
// inline area — is at least 5 % below the wide format's.
static unsigned choose_format(const uint64_t hist[64], uint64_t n_blocks, const RrrTables & T, bool allow_slim)
{
    if (!allow_slim || T.space[kEsc] != kRrrBS)
        return 0; // (the slim format's escape is "class 15, raw": a vector that keeps class 15 enumerative stays wide)
    const int forced = g_rrr_format.load();
    if (forced == 0 || forced == 1)
        return (unsigned)forced;
    if (n_blocks == 0)
        return 0;
    uint64_t esc = 0;
    double bits_w = 0, bits_s = 0; // offset bits of the whole vector in either format
    for (unsigned k = 0; k < 64; ++k)
    {
        if (k >= kEsc)
            esc += hist[k];
        bits_w += (double)hist[k] * T.space[k];
        bits_s += (double)hist[k] * (k >= kEsc ? kRrrBS : T.space[k]);
    }
    if (esc * 512 > n_blocks)
        return 0;
    auto size = [&](const RrrGeo & G, double bits)
    {
        const double nrec = (double)n_blocks / G.K, per = bits / nrec;
        return nrec * (1024.0 + (per > G.INLB ? per - G.INLB + 32.0 : 16.0));
    };
    return size(geo_of(1), bits_s) < 0.95 * size(geo_of(0), bits_w) ? 1u : 0u;
}

// smallest t whose raw classes cost at most 2 % (option "rrr_raw_budget", in permille) of the vector's compressed size on top of t = 10 (the classes 11..52 are
// always raw: the sparse decoder does not take them); hist[k] = blocks of class k.  Measured on the compressed FM-index of
// the 1 GiB text (profiles/rrr_raw_classes_r02.txt): t = 10 -> 369, 9 -> 384, 6 -> 437, 3 -> 469, 0 -> 460 Mcount/s at
// 5.10 .. 5.16 GB; a 5 %-dense vector gets t = 7 (+0.6 % of its length) and keeps its speed (it is bound by the fetches).
static unsigned choose_sparse_max(const uint64_t hist[64], uint64_t n_bits, bool standalone)
{
    if (const char * e = getenv("SDSL_HIP_RRR_SPARSE_MAX"))
    { // experiment knob
        const int v = atoi(e);
        if (v >= 0 && v <= 20)
            return (unsigned)(standalone ? v : std::min(v, 10)); // (the kernels over a wavelet tree's vectors stage the columns 0..10 only)
    }
    // Where the search starts (option "rrr_sparse_limit"; default 20 since round 6, 10 before).  A vector of 10-30 % density consists of
    // classes 6..25, and with every class above 10 raw it takes 1.3-1.4 times SDSL's space; with the limit at 20 the classes up to 20
    // stay enumerative (1.12-1.16 times SDSL's: what is left is the 128-byte record) and a block costs up to eighteen bisections
    // instead of eight.  The bucketed route decodes every block of a slice ONCE whatever its class (rrr_sorted.hip: its LDS image
    // holds the binomial columns 3..20 for such a vector), so large batches keep their speed.  The vectors INSIDE a wavelet tree stay
    // at 10 at most: count() on csa_wt<wt_huff<rrr_vector<63>>> decodes a block per tree level and lane, and that index is sized by
    // its raw middle classes anyway.
    const int opt = g_rrr_sparse_limit.load();
    const unsigned limit = (unsigned)std::min(standalone ? 20 : 10, std::max(0, opt));
    const RrrTables & T = host_tables();
    uint64_t size10 = 0; // about what the vector takes at t = limit: 13 bits of record per block + its field
    for (unsigned k = 0; k < 64; ++k)
        size10 += hist[k] * (13 + (k > limit && k < kRrrBS - limit ? kRrrBS : (unsigned)T.sdsl_space[k]));
    const int budget = g_rrr_raw_budget.load(); // permille of size10 (option "rrr_raw_budget", default 20)
    unsigned t = limit;
    uint64_t extra = 0;
    while (t > 0)
    { // making classes t and 63 - t raw as well
        const uint64_t more = hist[t] * (kRrrBS - T.sdsl_space[t]) + hist[kRrrBS - t] * (kRrrBS - T.sdsl_space[kRrrBS - t]);
        if ((extra + more) * 1000 > size10 * (uint64_t)budget)
            break;
        extra += more;
        --t;
    }
    if (getenv("SDSL_HIP_TRACE_BUILD"))
        fprintf(stderr, "[sdsl_hip] rrr_vector<63> of %llu bits: classes %u..%u raw, +%.2f %% of the length\n", (unsigned long long)n_bits,
                t + 1, kRrrBS - t - 1, 100.0 * (double)extra / (double)(n_bits ? n_bits : 1));
    return t;
}

struct RrrArrays // host image of a parsed SDSL stream
{
    uint64_t n_bits = 0, n_blocks = 0, n_sb = 0, ones = 0; // n_sb: SDSL's superblocks (32 blocks)
    std::vector<uint8_t> cls;      // actual classes per block (padded with zeros to whole SDSL superblocks and records)
    std::vector<uint64_t> stream;  // offset bits
    uint64_t stream_bits = 0;
    std::vector<uint64_t> sb_rank; // ones before each SDSL superblock
    std::vector<uint64_t> sb_ptr;  // stream position of each SDSL superblock
};

// rrr_vector<63>::load layout (rrr_vector.hpp:366-378,381-392)
static sdsl_hip_status rrr_parse_sdsl(StreamReader & rd, RrrArrays & A)
{
    HostIntVec bt, btnr, btnrp, rank, inv;
    uint64_t n = 0;
    if (!rd.u64(n) || !rd.int_vector(bt) || !rd.int_vector(btnr, 1) || !rd.int_vector(btnrp) || !rd.int_vector(rank)
        || !rd.int_vector(inv, 1))
        goto bad;
    if (n >= kLimRrrBits)
    { // the limit the plain bit vector enforces too; also keeps n + 63 and everything derived from it from wrapping
        set_error("rrr_vector<63> stream declares %llu bits: beyond the supported 2^40", (unsigned long long)n);
        return SDSL_HIP_ERR_FORMAT;
    }
    {
        A.n_bits = n;
        A.n_blocks = (n + kRrrBS) / kRrrBS;
        A.n_sb = (A.n_blocks + kRrrK - 1) / kRrrK;
        if (bt.width != 6 || bt.size() != A.n_blocks || btnrp.size() != A.n_sb || inv.size() != A.n_sb
            || rank.size() != A.n_sb + ((n % kRrrSB) > 0))
            goto bad;
        const RrrTables & T = host_tables();
        A.cls.assign(A.n_blocks + 64, 0); // (padded: a device record may reach up to 41 blocks beyond the last one)
        A.sb_rank.assign(A.n_sb + 1, 0);
        A.sb_ptr.assign(A.n_sb + 1, 0);
        uint64_t run = 0, ptr = 0;
        for (uint64_t s = 0; s < A.n_sb; ++s)
        {
            bool iv = inv.get(s) != 0;
            if (rank.get(s) != run || btnrp.get(s) != ptr)
            { // the dummy-block superblock stores pointer 0 (never dereferenced): tolerate exactly that
                if (!(rank.get(s) == run && s * kRrrK + 1 == A.n_blocks && n % kRrrBS == 0 && btnrp.get(s) == 0))
                    goto bad;
            }
            A.sb_rank[s] = run;
            A.sb_ptr[s] = ptr;
            for (unsigned j = 0; j < kRrrK; ++j)
            {
                uint64_t b = s * kRrrK + j;
                if (b >= A.n_blocks)
                    break;
                unsigned k = (unsigned)bt.get(b);
                if (iv)
                    k = kRrrBS - k;
                A.cls[b] = (uint8_t)k;
                run += k;
                ptr += T.sdsl_space[k];
            }
        }
        A.sb_rank[A.n_sb] = run;
        A.sb_ptr[A.n_sb] = ptr;
        A.ones = run;
        A.stream_bits = ptr;
        if (ptr > btnr.bit_size || (rank.size() > A.n_sb && rank.get(A.n_sb) != run) || run > n)
            goto bad;
        A.stream = btnr.words;
        A.stream.resize(((std::max<uint64_t>(btnr.bit_size, 64) + 63) >> 6) + 2, 0);
        return SDSL_HIP_OK;
    }
bad:
    set_error("malformed rrr_vector<63> stream (offset %zu of %zu)", rd.pos, rd.len);
    return SDSL_HIP_ERR_FORMAT;
}

// rank / access, one query per lane
template <int MODE, class F> // MODE 0: rank, 1: access; F: record format
__global__ __launch_bounds__(kRrrBlock) void k_rrr_rank(RrrView v, int bit, const uint64_t * __restrict__ iq,
                                                        uint64_t * __restrict__ out, uint8_t * __restrict__ out8,
                                                        uint64_t n)
{
    __shared__ RrrTables T;
    if (v.skip_if && *v.skip_if)
        return;
    rrr_stage_tables(&T, v.tables);
    for (uint64_t q = (uint64_t)blockIdx.x * kRrrBlock + threadIdx.x; q < n; q += (uint64_t)gridDim.x * kRrrBlock)
    {
        const uint64_t i = __builtin_nontemporal_load(iq + q);
        const bool ok = MODE == 0 ? i <= v.n_bits : i < v.n_bits;
        if (MODE == 1)
        {
            unsigned b = 0xFF;
            if (ok)
                rrr_rank1<F>(v, &T, i, &b);
            out8[q] = (uint8_t)b;
        }
        else
        {
            uint64_t r = SDSL_HIP_NPOS;
            if (ok)
            {
                const uint64_t r1 = rrr_rank1<F>(v, &T, i);
                r = bit ? r1 : i - r1;
            }
            out[q] = r;
        }
    }
}

// rrr_vector::get_int(idx, len) (rrr_vector.hpp:308-356): the len <= 64 bits starting at idx, bit idx in the lowest
// position; a window touches at most two 63-bit blocks.  One query per lane.
template <class F>
__global__ __launch_bounds__(kRrrBlock) void k_rrr_get_int(RrrView v, unsigned len, const uint64_t * __restrict__ iq,
                                                           uint64_t * __restrict__ out, uint64_t n)
{
    __shared__ RrrTables T;
    rrr_stage_tables(&T, v.tables);
    for (uint64_t q = (uint64_t)blockIdx.x * kRrrBlock + threadIdx.x; q < n; q += (uint64_t)gridDim.x * kRrrBlock)
    {
        const uint64_t i = __builtin_nontemporal_load(iq + q);
        uint64_t r = SDSL_HIP_NPOS;
        if (i <= v.n_bits && len <= v.n_bits - i)
        {
            r = 0;
            if (len)
            {
                const RankTail t = rrr_rank_head_f<F>(v, &T, i);
                r = rrr_decode_block(&T, t.k, t.nr) >> t.off;
                if (t.off + len > kRrrBS)
                {
                    const RankTail u = rrr_rank_head_f<F>(v, &T, i - t.off + kRrrBS);
                    r |= rrr_decode_block(&T, u.k, u.nr) << (kRrrBS - t.off);
                }
                r &= lo_set(len);
            }
        }
        out[q] = r;
    }
}

// select, one query per lane
// Flat variant: every wave owns a contiguous range of the batch; a lane that needs a query takes the next unassigned
// index of the range (rank among the needing lanes, from a ballot), so the queries in flight in a wave stay within a
// short window of the arrays (argument loads coalesce, result stores merge in L2).  One iteration = one probe.
template <int BIT, unsigned DECODE_AT, class F>
__global__ __launch_bounds__(kRrrBlock) void k_rrr_select_flat(RrrView v, const uint64_t * __restrict__ iq,
                                                               uint64_t * __restrict__ out, uint64_t n)
{
    __shared__ RrrTables T;
    if (v.skip_if && *v.skip_if)
        return;
    rrr_stage_tables(&T, v.tables);
    const uint64_t total = BIT ? v.ones : v.n_bits - v.ones;
    const uint32_t * __restrict__ smp = v.sel[BIT];
    const uint64_t n_waves = (uint64_t)gridDim.x * (kRrrBlock / 64);
    const uint64_t wave = (uint64_t)blockIdx.x * (kRrrBlock / 64) + threadIdx.x / 64;
    const uint64_t per = ((n + n_waves - 1) / n_waves + 63) / 64 * 64;
    uint64_t base = wave * per; // wave-uniform: next unassigned query
    const uint64_t end = base + per < n ? base + per : n;
    const uint64_t below = (UINT64_C(1) << (threadIdx.x & 63)) - 1;
    bool have = false, ready = false, fetched = false;
    uint64_t q = 0, nr_far = 0;
    RrrSelState st{};
    RrrSelHit h{};
    RrrSelLoc loc{};
    for (;;)
    {
        const uint64_t m_need = __ballot(!have);
        if (!have)
        {
            const uint64_t my = base + (uint64_t)__popcll(m_need & below);
            if (my < end)
            {
                q = my;
                const uint64_t i = iq[q];
                if (i >= 1 && i <= total)
                {
                    const uint64_t j = (i - 1) >> v.sel_shift[BIT];
                    rrr_sel_init<BIT>(v, st, i - 1, smp[j], smp[j + 1]);
                    have = true;
                }
                else // i > #args: SDSL returns size() (rrr_vector.hpp:641-642, 686-689); i == 0 is outside its domain
                    out[q] = i == 0 ? SDSL_HIP_NPOS : v.n_bits;
            }
        }
        base += (uint64_t)__popcll(m_need);
        if (base > end)
            base = end;
        if (__ballot(have) == 0 && base >= end)
            break;
        if (have && !ready)
            ready = rrr_sel_probe<BIT, F>(v, st, h);
        // the decode is the expensive half in instructions: run it when most of the wave can take part, or when
        // nobody is left probing
        const unsigned n_ready = (unsigned)__popcll(__ballot(ready));
        const bool probing = __ballot(have && !ready) != 0;
        if (n_ready >= DECODE_AT || !probing)
        {
            if (ready)
            {
                // a block whose offset field reaches into the overflow stream needs a second, random fetch: such a lane
                // only issues it in this round and decodes in the next one, so the rest of the wave does not wait for it
                bool now = true;
                uint64_t nr = nr_far;
                if (!fetched)
                {
                    loc = rrr_sel_locate<BIT, F>(v, &T, st.k0, h);
                    const unsigned len = T.space[loc.k];
                    if (rrr_sel_in_stream<F>(&T, loc))
                    {
                        nr_far = rrr_field_t<F::INL0, F::INLW>(v, h.r, loc.ptr, loc.rel, len);
                        fetched = true;
                        now = false;
                    }
                    else
                        nr = rrr_field_inline<F>(h.r, loc.rel, len);
                }
                if (now)
                {
                    out[q] = rrr_sel_decode<BIT>(v, &T, loc, nr);
                    have = ready = fetched = false;
                }
            }
        }
    }
}

template <int BIT, class F>
__global__ __launch_bounds__(kRrrBlock) void k_rrr_select(RrrView v, const uint64_t * __restrict__ iq,
                                                          uint64_t * __restrict__ out, uint64_t n)
{
    __shared__ RrrTables T;
    if (v.skip_if && *v.skip_if)
        return;
    rrr_stage_tables(&T, v.tables);
    const uint64_t total = BIT ? v.ones : v.n_bits - v.ones;
    // Software pipeline over the lane's queries: a select is a chain of dependent accesses (argument -> directory
    // samples -> record probe(s) -> offset field).  The argument is loaded two queries ahead and the samples one query
    // ahead, which takes two memory latencies off the chain.
    const uint64_t stride = (uint64_t)gridDim.x * kRrrBlock;
    const uint32_t * __restrict__ smp = v.sel[BIT];
    auto load_arg = [&](uint64_t q) -> uint64_t { return q < n ? __builtin_nontemporal_load(iq + q) : 0; };
    auto arg_ok = [&](uint64_t i) -> bool { return i >= 1 && i <= total; };
    uint64_t q = (uint64_t)blockIdx.x * kRrrBlock + threadIdx.x;
    uint64_t i_cur = load_arg(q), i_nxt = load_arg(q + stride);
    uint64_t j_cur = arg_ok(i_cur) ? (i_cur - 1) >> v.sel_shift[BIT] : 0;
    uint32_t s0_cur = smp[j_cur], s1_cur = smp[j_cur + 1];
    for (; q < n; q += stride)
    {
        const uint64_t i_nn = load_arg(q + 2 * stride);
        const uint64_t j_nxt = arg_ok(i_nxt) ? (i_nxt - 1) >> v.sel_shift[BIT] : 0;
        const uint32_t s0_nxt = smp[j_nxt], s1_nxt = smp[j_nxt + 1];
        const uint64_t i = i_cur;
        uint64_t r;
        if (arg_ok(i))
            r = rrr_select<BIT, F>(v, &T, i - 1, s0_cur, s1_cur);
        else // i > #args: SDSL returns size() (rrr_vector.hpp:641-642, 686-689); i == 0 is outside its domain
            r = i == 0 ? SDSL_HIP_NPOS : v.n_bits;
        __builtin_nontemporal_store(r, out + q);
        i_cur = i_nxt;
        i_nxt = i_nn;
        s0_cur = s0_nxt;
        s1_cur = s1_nxt;
    }
}

template <int BIT, class F>
static auto rrr_select_kernel() -> void (*)(RrrView, const uint64_t *, uint64_t *, uint64_t)
{
    const char * e = getenv("SDSL_HIP_RRR_SEL_FLAT"); // experiment knob: 0 = the nested loop (profiles/rrr_select_flat_r01.txt)
    if (e && atoi(e) == 0)
        return k_rrr_select<BIT, F>;
    return k_rrr_select_flat<BIT, 40, F>; // the threshold hardly matters: 24 .. 56 measured within 2 %
}

// ---- device-side encoder (rrr_vector(bit_vector const&), rrr_vector.hpp:158-270) -------------------------------
// the 63 bits of block b (masked to n_bits)
__device__ __forceinline__ uint64_t rrr_block_bits(const uint64_t * __restrict__ words, uint64_t n_bits, uint64_t b)
{
    uint64_t pos = b * kRrrBS;
    if (pos >= n_bits)
        return 0;
    unsigned len = (unsigned)(n_bits - pos < kRrrBS ? n_bits - pos : kRrrBS);
    uint64_t wi = pos >> 6;
    unsigned off = (unsigned)(pos & 63);
    uint64_t v = words[wi] >> off;
    if (off + len > 64)
        v |= words[wi + 1] << (64 - off); // the caller pads the word array by one
    return v & lo_set(len);
}

// pass 0: blocks per class
__global__ __launch_bounds__(256) void k_rrr_class_hist(const uint64_t * __restrict__ words, uint64_t n_bits, uint64_t n_blocks,
                                                        unsigned long long * __restrict__ hist)
{
    __shared__ unsigned h[64];
    if (threadIdx.x < 64)
        h[threadIdx.x] = 0;
    __syncthreads();
    for (uint64_t b = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; b < n_blocks; b += (uint64_t)gridDim.x * blockDim.x)
        atomicAdd(&h[popc64(rrr_block_bits(words, n_bits, b))], 1u);
    __syncthreads();
    if (threadIdx.x < 64 && h[threadIdx.x])
        atomicAdd(&hist[threadIdx.x], (unsigned long long)h[threadIdx.x]);
}

// pass 1, one thread per record: classes into the record, ones and offset bits of the record
// (slim format: the prefix counts wait in word 5 — the first inline word, written last by pass 2 — until the header words exist)
template <class F>
__global__ __launch_bounds__(256) void k_rrr_enc_classes(const uint64_t * __restrict__ words, uint64_t n_bits,
                                                         uint64_t n_blocks, uint64_t n_sb,
                                                         const RrrTables * __restrict__ tables, uint64_t * __restrict__ rec,
                                                         uint32_t * __restrict__ sb_ones, uint32_t * __restrict__ sb_len)
{
    __shared__ uint8_t space[64];
    if (threadIdx.x < 64)
        space[threadIdx.x] = tables->space[threadIdx.x];
    __syncthreads();
    constexpr unsigned CLSW = F::id ? 4u : kClsW;
    for (uint64_t sb = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; sb < n_sb; sb += (uint64_t)gridDim.x * blockDim.x)
    {
        unsigned ones = 0, len = 0;
        uint64_t cw[4] = {0, 0, 0, 0};
        unsigned po[3] = {0, 0, 0}, pb[3] = {0, 0, 0};
        for (unsigned j = 0; j < F::K; ++j)
        {
            const unsigned g = j / F::GRP, u = j - g * F::GRP;
            if (j && u == 0)
            {
                po[g - 1] = ones;
                pb[g - 1] = len;
            }
            uint64_t b = sb * F::K + j;
            if (b >= n_blocks)
                continue; // blocks behind the end count as class 0
            unsigned k = popc64(rrr_block_bits(words, n_bits, b));
            const bool esc = F::id && k >= kEsc;
            cw[g] |= (uint64_t)(esc ? kEsc : k) << (CLSW * u);
            ones += k;
            len += esc ? kRrrBS : space[k];
        }
        uint64_t * r = rec + sb * kRecWords;
        if (F::id == 0)
        {
            r[2] = rrr_pack_prefix(po, pb);
            r[kRecClasses + 3] = cw[3];
        }
        else
            r[F::INL0] = (uint64_t)po[0] | ((uint64_t)po[1] << 10) | ((uint64_t)pb[0] << 21) | ((uint64_t)pb[1] << 31) | ((uint64_t)ones << 42);
        r[F::CLS0 + 0] = cw[0];
        r[F::CLS0 + 1] = cw[1];
        r[F::CLS0 + 2] = cw[2];
        sb_ones[sb] = ones;
        sb_len[sb] = len > F::INLB ? (len - F::INLB + 63) >> 6 : 0; // WORDS of this record in the overflow stream (rrr_device.hpp)
    }
}

// pass 2, one thread per superblock (headers r[0] = ones before, r[1] = stream pointer are in place): offsets into
// the stream and the record's inline area, select samples for both bit values
template <class F>
__global__ __launch_bounds__(256) void k_rrr_enc_offsets(const uint64_t * __restrict__ words, uint64_t n_bits,
                                                         uint64_t n_blocks, uint64_t n_sb,
                                                         const RrrTables * __restrict__ tables, uint64_t * __restrict__ rec,
                                                         const uint32_t * __restrict__ sb_ones,
                                                         unsigned long long * __restrict__ stream, uint32_t sh1, uint32_t sh0, uint32_t ps,
                                                         uint32_t * __restrict__ sel1, uint32_t * __restrict__ sel0)
{
    __shared__ RrrTables T;
    rrr_stage_tables(&T, tables);
    const uint64_t S1 = UINT64_C(1) << sh1, S0 = UINT64_C(1) << sh0;
    for (uint64_t sb = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; sb < n_sb; sb += (uint64_t)gridDim.x * blockDim.x)
    {
        uint64_t * r = rec + sb * kRecWords;
        const uint64_t ones_before = r[0], ptr = r[1];
        const uint64_t stash = F::id ? r[F::INL0] : 0;
        const uint64_t start = sb * F::SB;
        uint64_t acc1 = ones_before, acc0 = start - ones_before; // arguments before the current block
        uint64_t j1 = (acc1 + S1 - 1) >> sh1, j0 = (acc0 + S0 - 1) >> sh0;
        uint64_t inl[F::INLW] = {};
        unsigned rel = 0;
        for (unsigned j = 0; j < F::K; ++j)
        {
            const uint64_t b = sb * F::K + j;
            if (b >= n_blocks)
                break;
            const uint64_t bstart = b * kRrrBS;
            const unsigned blen = bstart >= n_bits ? 0u : (unsigned)(n_bits - bstart < kRrrBS ? n_bits - bstart : kRrrBS);
            const uint64_t bits = rrr_block_bits(words, n_bits, b);
            const unsigned k = popc64(bits), len = F::id && k >= kEsc ? kRrrBS : T.space[k];
            if (len)
            {
                uint64_t nr = bits, x = rrr_raw_width(len) ? 0 : bits; // a raw class stores the block itself
                unsigned kk = k;
                if (x)
                    nr = 0;
                while (x)
                { // combinatorial number system, positions from the least significant bit
                    unsigned p = (unsigned)__ffsll((long long)x) - 1;
                    nr += T.binom[62 - p][kk];
                    --kk;
                    x &= x - 1;
                }
                // the record's offsets = its inline area followed by its stretch of the stream; a field may straddle the seam
                if (rel < F::INLB)
                {
                    const unsigned w = rel >> 6, o = rel & 63;
                    inl[w] |= nr << o;
                    if (o + len > 64 && w + 1 < F::INLW)
                        inl[w + 1] |= nr >> (64 - o);
                }
                if (rel + len > F::INLB)
                {
                    const unsigned cut = rel < F::INLB ? F::INLB - rel : 0u; // bits of the field that went inline
                    const uint64_t pos = ptr * 64 + (rel + cut - F::INLB), val = nr >> cut;
                    const unsigned off = (unsigned)(pos & 63);
                    atomicOr(&stream[pos >> 6], (unsigned long long)(val << off));
                    if (off + (len - cut) > 64)
                        atomicOr(&stream[(pos >> 6) + 1], (unsigned long long)(val >> (64 - off)));
                }
            }
            rel += len;
            // select samples that fall into this block
            while ((j1 << sh1) < acc1 + k)
            {
                sel1[j1] = (uint32_t)((bstart + sel64(bits, (unsigned)((j1 << sh1) - acc1) + 1)) >> ps);
                ++j1;
            }
            const uint64_t zb = ~bits & lo_set(blen);
            while ((j0 << sh0) < acc0 + (blen - k))
            {
                sel0[j0] = (uint32_t)((bstart + sel64(zb, (unsigned)((j0 << sh0) - acc0) + 1)) >> ps);
                ++j0;
            }
            acc1 += k;
            acc0 += blen - k;
        }
        if (rel < F::INLB && (rel & 63))
            inl[rel >> 6] &= lo_set(rel & 63);
        for (unsigned w = 0; w < F::INLW; ++w)
            r[F::INL0 + w] = w * 64 < rel ? inl[w] : 0;
        if (F::id == 0)
            r[1] = ptr | ((uint64_t)sb_ones[sb] << 48);
        else
        {
            const unsigned o16 = (unsigned)stash & 0x3FFu, o32 = (unsigned)(stash >> 10) & 0x7FFu, b16 = (unsigned)(stash >> 21) & 0x3FFu,
                           b32 = (unsigned)(stash >> 31) & 0x7FFu, ones_in = (unsigned)(stash >> 42) & 0xFFFu;
            r[0] = rrs_pack0(ones_before, o16, o32);
            r[1] = rrs_pack1(ptr, b16, b32, ones_in - o32);
        }
    }
}

// host mirror of rrr_decode_block (used only to place the select samples)
static uint64_t decode_block_host(const RrrTables & T, unsigned k, uint64_t nr)
{
    if (k == 0)
        return 0;
    if (k == kRrrBS)
        return lo_set(kRrrBS);
    uint64_t bits = 0;
    for (int m = 62; m >= 0 && k > 0; --m)
    {
        uint64_t c = T.binom[m][k];
        if (nr >= c)
        {
            nr -= c;
            --k;
            bits |= UINT64_C(1) << (62 - m);
        }
    }
    return bits;
}

// Sampling rates of the two select directories, per bit value: the smallest power of two >= 16 that keeps the
// directory within 2^16 samples (256 KiB: resident in every L2 next to the streaming traffic).  A superblock spans
// 2016 bits, so even a coarse directory interpolates to the right record or its neighbour; what a denser directory
// saves in probes it loses in fabric requests for the samples (2^34 bits at 5 % density: select_0 16.0 -> 20.9 Gq/s
// going from 2^21 to 2^15 samples, select_1 best at 2^16; profiles/rrr_select_sweep_r01.txt).
// SDSL_HIP_RRR_SEL_LOG2 overrides the rate of both (profiling).
static void rrr_sel_shifts(uint64_t ones, uint64_t zeros, uint64_t rec_bits, uint32_t shb[2])
{
    const uint64_t cnt[2] = {zeros, ones};
    for (int b = 0; b < 2; ++b)
    {
        uint32_t sh = 4; // few arguments: sample densely, the directory stays far below 2^16 entries
        while (sh < 24 && (cnt[b] >> sh) > (UINT64_C(1) << 16))
            ++sh;
        // a RARE bit value (mean gap of more than a record): interpolation inside a sample interval of many records misses,
        // and every miss is a random fetch (isolated ones every 2^16 bits: 4.5 G/s).  Sample so that an interval spans about
        // two records, as long as that directory stays small (2^18 samples = 1 MiB)
        if (cnt[b] && (ones + zeros) / cnt[b] > rec_bits / 4)
        {
            const uint64_t gap = (ones + zeros) / cnt[b];
            uint32_t ss = 0;
            while (ss < sh && (gap << (ss + 1)) <= 2 * rec_bits)
                ++ss;
            if ((cnt[b] >> ss) <= (UINT64_C(1) << 18))
                sh = ss;
        }
        shb[b] = sh;
    }
    if (const char * e = getenv("SDSL_HIP_RRR_SEL_LOG2"))
    {
        int v = atoi(e);
        if (v >= 6 && v <= 24)
            shb[0] = shb[1] = (uint32_t)v;
    }
}

static sdsl_hip_status rrr_upload(RrrHost & h, const RrrArrays & A, int device)
{
    h.device = device;
    {
        uint64_t hist[64] = {};
        for (uint64_t b = 0; b < A.n_blocks; ++b)
            ++hist[A.cls[b]];
        h.sparse_max = choose_sparse_max(hist, A.n_bits, h.allow_slim);
        h.fmt = choose_format(hist, A.n_blocks, tables_for(h.sparse_max), h.allow_slim);
    }
    const RrrTables T = tables_for(h.sparse_max);
    RrrGeo G = geo_of(h.fmt);
    uint64_t n_rec = (A.n_blocks + G.K - 1) / G.K;
    if (A.stream_bits >= (UINT64_C(1) << 48) || n_rec > UINT64_C(0xFFFFFFFF))
    {
        set_error("rrr_vector too large for the device record format");
        return SDSL_HIP_ERR_UNSUPPORTED;
    }
    // per device record: ones in front of it, position of its first field in SDSL's stream, words of overflow stream in front of it
    // (the device keeps stream storage only for the offset bits beyond a record's inline area)
    std::vector<uint64_t> rec_rank, rec_ptr, cptr;
    for (;;)
    {
        rec_rank.assign((size_t)n_rec + 1, 0);
        rec_ptr.assign((size_t)n_rec + 1, 0);
        cptr.assign((size_t)n_rec + 1, 0);
        uint64_t run = 0, ptr = 0;
        for (uint64_t s = 0; s < n_rec; ++s)
        {
            rec_rank[s] = run;
            rec_ptr[s] = ptr;
            uint64_t len = 0; // on the device: raw classes take 63 bits
            for (unsigned j = 0; j < G.K; ++j)
            {
                const unsigned k = A.cls[(size_t)s * G.K + j];
                run += k;
                ptr += T.sdsl_space[k];
                len += G.dev_len(T, k);
            }
            cptr[s + 1] = cptr[s] + (len > G.INLB ? (len - G.INLB + 63) >> 6 : 0); // words
        }
        rec_rank[n_rec] = run;
        rec_ptr[n_rec] = ptr;
        if (G.fmt == 0 || cptr[n_rec] + 3 < kSlimMaxStream)
            break;
        h.fmt = 0; // (a slim vector with more overflow than its 33-bit pointers address: only a forced format gets here)
        G = geo_of(0);
        n_rec = (A.n_blocks + G.K - 1) / G.K;
    }
    std::vector<uint64_t> rec((size_t)n_rec * kRecWords, 0);
    std::vector<uint64_t> cstream(cptr[n_rec] + 3, 0);
    const uint64_t zeros = A.n_bits - A.ones;
    // sampling rate: smallest power of two >= 256 that keeps a directory within 2^21 samples
    uint32_t shb[2];
    rrr_sel_shifts(A.ones, zeros, G.SB, shb);
    uint32_t ps = 0;
    while ((A.n_bits >> ps) >= UINT64_C(0xFFFFFFFF))
        ++ps;
    const uint64_t ns1 = (A.ones + (UINT64_C(1) << shb[1]) - 1) >> shb[1], ns0 = (zeros + (UINT64_C(1) << shb[0]) - 1) >> shb[0];
    std::vector<uint32_t> sel1(ns1 + 2, 0), sel0(ns0 + 2, 0);
    auto fill = [&](uint64_t s0, uint64_t s1)
    {
        for (uint64_t s = s0; s < s1; ++s)
        {
            uint64_t * r = &rec[(size_t)s * kRecWords];
            uint64_t ones_in = rec_rank[s + 1] - rec_rank[s];
            {
                unsigned po[3] = {0, 0, 0}, pb[3] = {0, 0, 0}, ones = 0, bits = 0;
                for (unsigned j = 0; j < G.K; ++j)
                {
                    const unsigned g = j / G.GRP, u = j - g * G.GRP;
                    if (j && u == 0)
                    {
                        po[g - 1] = ones;
                        pb[g - 1] = bits;
                    }
                    unsigned k = A.cls[(size_t)s * G.K + j];
                    r[G.CLS0 + g] |= (uint64_t)G.field_value(k) << (G.CLSW * u);
                    ones += k;
                    bits += G.dev_len(T, k);
                }
                if (G.fmt == 0)
                {
                    r[0] = rec_rank[s];
                    r[1] = cptr[s] | (ones_in << 48);
                    r[2] = rrr_pack_prefix(po, pb);
                }
                else
                {
                    r[0] = rrs_pack0(rec_rank[s], po[0], po[1]);
                    r[1] = rrs_pack1(cptr[s], pb[0], pb[1], (unsigned)ones_in - po[1]);
                }
            }
            { // the record's fields: SDSL's offsets, or the decoded block for a raw class; nine words inline, the rest in
              // the record's own words of the overflow stream (no other thread writes those)
                uint64_t sp = rec_ptr[s];
                unsigned rel = 0;
                for (unsigned j = 0; j < G.K; ++j)
                {
                    const uint64_t blk = s * G.K + j;
                    if (blk >= A.n_blocks)
                        break;
                    const unsigned k = A.cls[blk], sl = T.sdsl_space[k], dl = G.dev_len(T, k);
                    if (dl)
                    {
                        uint64_t f = read_bits(A.stream.data(), sp, sl);
                        if (rrr_raw_width(dl))
                            f = decode_block_host(T, k, f);
                        for (unsigned done = 0; done < dl;)
                        { // word by word of the record's offset string
                            const unsigned at = rel + done, w = at >> 6, o = at & 63, n = std::min(dl - done, 64 - o);
                            uint64_t * dst = w < G.INLW ? &r[G.INL0 + w] : &cstream[cptr[s] + (w - G.INLW)];
                            *dst |= ((f >> done) & lo_set(n)) << o;
                            done += n;
                        }
                    }
                    sp += sl;
                    rel += dl;
                }
            }
            // select samples falling into this superblock: walk its blocks, decode only where needed
            uint64_t start = s * G.SB;
            uint64_t len_in = A.n_bits - start < G.SB ? A.n_bits - start : G.SB;
            uint64_t h[2] = {start - rec_rank[s], rec_rank[s]};
            uint64_t c[2] = {len_in - ones_in, ones_in};
            for (int b = 0; b < 2; ++b)
            {
                const uint32_t sh = shb[b];
                const uint64_t S = UINT64_C(1) << sh;
                uint64_t jj = (h[b] + S - 1) >> sh;
                if (c[b] == 0 || (jj << sh) >= h[b] + c[b])
                    continue;
                uint64_t acc = h[b], ptr = rec_ptr[s];
                for (unsigned t = 0; t < G.K && (jj << sh) < h[b] + c[b]; ++t)
                {
                    uint64_t blk = s * G.K + t;
                    if (blk >= A.n_blocks)
                        break;
                    unsigned k = A.cls[blk], len = T.sdsl_space[k];
                    uint64_t bstart = blk * kRrrBS;
                    unsigned blen = bstart >= A.n_bits ? 0u : (unsigned)std::min<uint64_t>(kRrrBS, A.n_bits - bstart);
                    unsigned a = b ? k : blen - k;
                    if ((jj << sh) < acc + a)
                    {
                        uint64_t bits = decode_block_host(T, k, read_bits(A.stream.data(), ptr, len));
                        if (!b)
                            bits = ~bits & lo_set(blen);
                        while ((jj << sh) < acc + a)
                        {
                            uint64_t pos = bstart + sel64(bits, (unsigned)((jj << sh) - acc) + 1);
                            (b ? sel1 : sel0)[jj] = (uint32_t)(pos >> ps);
                            ++jj;
                        }
                    }
                    acc += a;
                    ptr += len;
                }
            }
        }
    };
    {
        unsigned nt = std::max(1u, std::min(std::thread::hardware_concurrency(), 64u));
        if (n_rec < 4096)
            nt = 1;
        std::vector<std::thread> th;
        for (unsigned t = 0; t < nt; ++t)
            th.emplace_back(fill, n_rec * t / nt, n_rec * (t + 1) / nt);
        for (auto & x : th)
            x.join();
    }
    sel1[ns1] = (uint32_t)(A.n_bits >> ps);
    sel0[ns0] = (uint32_t)(A.n_bits >> ps);
    SH_TRY(h.rec.alloc(rec.size() * 8));
    if (!rec.empty())
        SH_HIP(hipMemcpy(h.rec.p, rec.data(), rec.size() * 8, hipMemcpyHostToDevice));
    SH_TRY(h.stream.alloc(cstream.size() * 8));
    SH_HIP(hipMemcpy(h.stream.p, cstream.data(), cstream.size() * 8, hipMemcpyHostToDevice));
    SH_TRY(h.tables.alloc(sizeof(RrrTables)));
    SH_HIP(hipMemcpy(h.tables.p, &T, sizeof(RrrTables), hipMemcpyHostToDevice));
    SH_TRY(h.sel[1].alloc(sel1.size() * 4));
    SH_HIP(hipMemcpy(h.sel[1].p, sel1.data(), sel1.size() * 4, hipMemcpyHostToDevice));
    SH_TRY(h.sel[0].alloc(sel0.size() * 4));
    SH_HIP(hipMemcpy(h.sel[0].p, sel0.data(), sel0.size() * 4, hipMemcpyHostToDevice));
    h.view.rec = h.rec.as<uint64_t>();
    h.view.stream = h.stream.as<uint64_t>();
    h.view.tables = h.tables.as<RrrTables>();
    h.view.sel[0] = h.sel[0].as<uint32_t>();
    h.view.sel[1] = h.sel[1].as<uint32_t>();
    h.view.n_bits = A.n_bits;
    h.view.n_blocks = A.n_blocks;
    h.view.n_sb = n_rec;
    h.view.ones = A.ones;
    h.view.sel_shift[0] = shb[0];
    h.view.sel_shift[1] = shb[1];
    h.view.sel_pshift = ps;
    h.view.fmt = h.fmt;
    h.view.sparse_max = h.sparse_max;
    return SDSL_HIP_OK;
}

sdsl_hip_status rrr_build_from_stream(RrrHost & h, StreamReader & rd, int device)
{
    RrrArrays A;
    SH_TRY(rrr_parse_sdsl(rd, A));
    return rrr_upload(h, A, device);
}

__global__ void k_rrr_set_sentinels(uint32_t * a, uint64_t ia, uint32_t * b, uint64_t ib, uint32_t v)
{
    a[ia] = v;
    b[ib] = v;
}

// rrr_vector<63>(bit_vector const&) on the device: words (device memory) -> records, stream, directories
sdsl_hip_status rrr_build_device(RrrHost & h, const uint64_t * d_words, uint64_t n_bits, int device)
{
    h.device = device;
    const uint64_t n_blocks = (n_bits + kRrrBS) / kRrrBS; // one all-zero dummy block when 63 | n (rrr_vector.hpp:163)
    if ((n_blocks + kRecK - 1) / kRecK > UINT64_C(0xFFFFFFFF))
    {
        set_error("rrr_vector too large for the device record format");
        return SDSL_HIP_ERR_UNSUPPORTED;
    }
    { // pass 0: how many blocks of every class -> which classes are stored raw (rrr_device.hpp)
        DevBuf dh;
        SH_TRY(dh.alloc(64 * 8, true));
        hipLaunchKernelGGL(k_rrr_class_hist, dim3(grid_for(n_blocks, 256, 4096)), dim3(256), 0, 0, d_words, n_bits, n_blocks,
                           dh.as<unsigned long long>());
        SH_HIP(hipGetLastError());
        uint64_t hist[64];
        SH_HIP(hipMemcpy(hist, dh.p, sizeof hist, hipMemcpyDeviceToHost));
        h.sparse_max = choose_sparse_max(hist, n_bits, h.allow_slim);
        h.fmt = choose_format(hist, n_blocks, tables_for(h.sparse_max), h.allow_slim);
        if (h.fmt == 1 && n_bits >= (UINT64_C(1) << 40))
            h.fmt = 0; // (a slim record holds 40 bits of rank: also when the format was forced)
    }
    const RrrTables T = tables_for(h.sparse_max);
again:
    const RrrGeo G = geo_of(h.fmt);
    const uint64_t n_sb = (n_blocks + G.K - 1) / G.K; // records
    SH_TRY(h.tables.alloc(sizeof(RrrTables)));
    SH_HIP(hipMemcpy(h.tables.p, &T, sizeof(RrrTables), hipMemcpyHostToDevice));
    SH_TRY(h.rec.alloc(n_sb * kRecWords * 8, true));
    DevBuf sb_ones, sb_len;
    SH_TRY(sb_ones.alloc(n_sb * 4));
    SH_TRY(sb_len.alloc(n_sb * 4));
    const unsigned grid = grid_for(n_sb, 256, 65536);
    if (G.fmt)
        hipLaunchKernelGGL(k_rrr_enc_classes<RrrFmtS>, dim3(grid), dim3(256), 0, 0, d_words, n_bits, n_blocks, n_sb,
                           h.tables.as<RrrTables>(), h.rec.as<uint64_t>(), sb_ones.as<uint32_t>(), sb_len.as<uint32_t>());
    else
        hipLaunchKernelGGL(k_rrr_enc_classes<RrrFmtW>, dim3(grid), dim3(256), 0, 0, d_words, n_bits, n_blocks, n_sb,
                           h.tables.as<RrrTables>(), h.rec.as<uint64_t>(), sb_ones.as<uint32_t>(), sb_len.as<uint32_t>());
    SH_HIP(hipGetLastError());
    uint64_t ones = 0, stream_bits = 0; // stream_bits: total of sb_len
    SH_TRY(device_exclusive_scan_u32(sb_ones.as<uint32_t>(), n_sb, h.rec.as<uint64_t>(), kRecWords, &ones));
    SH_TRY(device_exclusive_scan_u32(sb_len.as<uint32_t>(), n_sb, h.rec.as<uint64_t>() + 1, kRecWords, &stream_bits));
    const uint64_t stream_words = stream_bits; // (the scan summed words)
    if (G.fmt && stream_words + 3 >= kSlimMaxStream)
    { // more overflow than a slim record's 33-bit pointer addresses (only a forced format gets here)
        h.fmt = 0;
        h.rec.release();
        h.tables.release();
        goto again;
    }
    const uint64_t zeros = n_bits - ones;
    uint32_t shb[2];
    rrr_sel_shifts(ones, zeros, G.SB, shb);
    uint32_t ps = 0;
    while ((n_bits >> ps) >= UINT64_C(0xFFFFFFFF))
        ++ps;
    const uint64_t ns1 = (ones + (UINT64_C(1) << shb[1]) - 1) >> shb[1], ns0 = (zeros + (UINT64_C(1) << shb[0]) - 1) >> shb[0];
    SH_TRY(h.stream.alloc((stream_words + 3) * 8, true));
    SH_TRY(h.sel[1].alloc((ns1 + 2) * 4, true));
    SH_TRY(h.sel[0].alloc((ns0 + 2) * 4, true));
    if (G.fmt)
        hipLaunchKernelGGL(k_rrr_enc_offsets<RrrFmtS>, dim3(grid), dim3(256), 0, 0, d_words, n_bits, n_blocks, n_sb,
                           h.tables.as<RrrTables>(), h.rec.as<uint64_t>(), sb_ones.as<uint32_t>(),
                           h.stream.as<unsigned long long>(), shb[1], shb[0], ps, h.sel[1].as<uint32_t>(), h.sel[0].as<uint32_t>());
    else
        hipLaunchKernelGGL(k_rrr_enc_offsets<RrrFmtW>, dim3(grid), dim3(256), 0, 0, d_words, n_bits, n_blocks, n_sb,
                           h.tables.as<RrrTables>(), h.rec.as<uint64_t>(), sb_ones.as<uint32_t>(),
                           h.stream.as<unsigned long long>(), shb[1], shb[0], ps, h.sel[1].as<uint32_t>(), h.sel[0].as<uint32_t>());
    SH_HIP(hipGetLastError());
    hipLaunchKernelGGL(k_rrr_set_sentinels, dim3(1), dim3(1), 0, 0, h.sel[1].as<uint32_t>(), ns1, h.sel[0].as<uint32_t>(),
                       ns0, (uint32_t)(n_bits >> ps));
    SH_HIP(hipGetLastError());
    SH_HIP(hipDeviceSynchronize());
    h.view.rec = h.rec.as<uint64_t>();
    h.view.stream = h.stream.as<uint64_t>();
    h.view.tables = h.tables.as<RrrTables>();
    h.view.sel[0] = h.sel[0].as<uint32_t>();
    h.view.sel[1] = h.sel[1].as<uint32_t>();
    h.view.n_bits = n_bits;
    h.view.n_blocks = n_blocks;
    h.view.n_sb = n_sb;
    h.view.ones = ones;
    h.view.sel_shift[0] = shb[0];
    h.view.sel_shift[1] = shb[1];
    h.view.sel_pshift = ps;
    h.view.fmt = h.fmt;
    h.view.sparse_max = h.sparse_max;
    return SDSL_HIP_OK;
}

} // namespace sdslhip

namespace sdslhip {
sdsl_hip_status rrr_serialize_host(const RrrHost & h, StreamWriter & w)
{
    SH_HIP(hipSetDevice(h.device));
    const RrrView & rv = h.view;
    const RrrGeo G = geo_of(h.fmt);
    const uint64_t n = rv.n_bits, nb = rv.n_blocks, nrec = rv.n_sb;
    std::vector<uint64_t> rec((size_t)nrec * kRecWords);
    if (nrec)
        SH_HIP(hipMemcpy(rec.data(), rv.rec, rec.size() * 8, hipMemcpyDeviceToHost));
    std::vector<uint64_t> cs(h.stream.bytes / 8 + 2, 0); // the overflow stream
    if (h.stream.bytes)
        SH_HIP(hipMemcpy(cs.data(), rv.stream, h.stream.bytes, hipMemcpyDeviceToHost));
    const RrrTables T = tables_for(h.sparse_max);
    auto rec_ptr = [&](const uint64_t * rp) -> uint64_t { return G.fmt ? RrrFmtS::ptr(rp[1]) : RrrFmtW::ptr(rp[1]); };
    // the device fields of a record, in order: f(j, class, field bits, field).  An escaped block of a slim record says its class itself
    auto walk = [&](uint64_t r, auto && f)
    {
        const uint64_t * rp = &rec[r * kRecWords];
        const uint64_t * far = cs.data() + rec_ptr(rp);
        unsigned rel = 0;
        for (unsigned j = 0; j < G.K; ++j)
        {
            if (r * G.K + j >= nb)
                break;
            const unsigned fv = (unsigned)(rp[G.CLS0 + j / G.GRP] >> (G.CLSW * (j % G.GRP))) & ((1u << G.CLSW) - 1);
            const unsigned dl = T.space[fv]; // (space[15] == 63: the escape reads as the raw class it is)
            uint64_t fld = 0;
            for (unsigned done = 0; done < dl;)
            {
                const unsigned p = rel + done, wd = p >> 6, o = p & 63, cnt = std::min(dl - done, 64 - o);
                const uint64_t word = wd < G.INLW ? rp[G.INL0 + wd] : far[wd - G.INLW];
                fld |= ((word >> o) & lo_set(cnt)) << done;
                done += cnt;
            }
            const unsigned k = G.fmt && fv == kEsc ? (unsigned)__builtin_popcountll(fld) : fv;
            f(j, k, dl, fld);
            rel += dl;
        }
    };
    unsigned nt = std::max(1u, std::min(std::thread::hardware_concurrency(), 64u));
    if (nrec < 4096)
        nt = 1;
    auto spread = [&](auto && fn)
    {
        std::vector<std::thread> th;
        for (unsigned t = 0; t < nt; ++t)
            th.emplace_back(fn, nrec * t / nt, nrec * (t + 1) / nt);
        for (auto & x : th)
            x.join();
    };
    std::vector<uint8_t> clsv((size_t)nb + 64, 0); // classes per block
    spread([&](uint64_t r0, uint64_t r1)
           {
               for (uint64_t r = r0; r < r1; ++r)
                   walk(r, [&](unsigned j, unsigned k, unsigned, uint64_t) { clsv[r * G.K + j] = (uint8_t)k; });
           });
    auto cls = [&](uint64_t b) -> unsigned { return clsv[b]; };
    // SDSL's samples every 32 blocks (m_rank, m_btnrp) and its offset stream (m_btnr) from the classes, the inline areas and
    // the overflow stream of the device records
    const uint64_t nsb = (nb + kRrrK - 1) / kRrrK;
    std::vector<uint64_t> sptr((size_t)nsb + 1, 0), srank((size_t)nsb + 1, 0);
    std::vector<uint64_t> rat((size_t)nrec + 1, 0);  // SDSL stream position of every record's first field
    {
        uint64_t len = 0, ones = 0;
        for (uint64_t b = 0; b < nb; ++b)
        {
            if (b % kRrrK == 0)
            {
                sptr[b / kRrrK] = len;
                srank[b / kRrrK] = ones;
            }
            if (b % G.K == 0)
                rat[b / G.K] = len;
            const unsigned k = cls(b);
            len += T.sdsl_space[k];
            ones += k;
        }
        sptr[nsb] = len;
        srank[nsb] = ones;
        rat[nrec] = len;
    }
    const uint64_t stream_bits = sptr[nsb];
    const uint64_t btnr_bits = std::max<uint64_t>(stream_bits, 64); // rrr_vector.hpp:183
    std::vector<uint64_t> btnr(((btnr_bits + 63) >> 6) + 1, 0);
    // field by field: a raw block is turned back into SDSL's offset (bin_to_nr, rrr_helper.hpp:346-366).  Records are
    // spread over threads; neighbouring records share words of m_btnr, hence the atomic OR
    spread([&](uint64_t r0, uint64_t r1)
           {
               for (uint64_t r = r0; r < r1; ++r)
               {
                   uint64_t at = rat[r];
                   walk(r, [&](unsigned, unsigned k, unsigned dl, uint64_t f)
                        {
                            const unsigned sl = T.sdsl_space[k];
                            if (sl)
                            {
                                if (rrr_raw_width(dl))
                                {
                                    uint64_t nr = 0, x = f;
                                    unsigned kk = k;
                                    while (x)
                                    { // combinatorial number system, positions from the least significant bit
                                        const unsigned p = (unsigned)__builtin_ctzll(x);
                                        nr += T.binom[62 - p][kk];
                                        --kk;
                                        x &= x - 1;
                                    }
                                    f = nr;
                                }
                                const unsigned o = (unsigned)(at & 63);
                                __atomic_fetch_or(&btnr[at >> 6], f << o, __ATOMIC_RELAXED);
                                if (o + sl > 64)
                                    __atomic_fetch_or(&btnr[(at >> 6) + 1], f >> (64 - o), __ATOMIC_RELAXED);
                            }
                            at += sl;
                        });
               }
           });
    PackedBuilder bt(nb, 6), btnrp(nsb, (uint8_t)(hi64(stream_bits) + 1)), invert(nsb, 1);
    const uint64_t n_rank = nsb + ((n % kRrrSB) > 0); // rrr_vector.hpp:185-186
    PackedBuilder rank(n_rank, (uint8_t)(hi64(rv.ones) + 1));
    for (uint64_t s = 0; s < nsb; ++s)
    {
        const uint64_t i = s * kRrrK;
        // superblock inversion (rrr_vector.hpp:203-228): only decided inside the full-block loop, i.e. when block i
        // is a complete 63-bit block, and only for superblocks with all 32 blocks present
        bool inv = false;
        if ((i + 1) * kRrrBS <= n && i + kRrrK <= nb)
        {
            unsigned gt = 0;
            for (unsigned j = 0; j < kRrrK; ++j)
                gt += cls(i + j) > kRrrBS / 2;
            inv = gt > kRrrK / 2;
        }
        invert.set(s, inv);
        for (uint64_t b = i; b < std::min(nb, i + kRrrK); ++b)
            bt.set(b, inv ? kRrrBS - cls(b) : cls(b));
        const bool dummy_only = i * kRrrBS >= n; // superblock that starts with the dummy block: never initialised by SDSL
        btnrp.set(s, dummy_only ? 0 : sptr[s]);
        if (s + 1 < n_rank)
            rank.set(s, srank[s]);
    }
    if (n_rank)
        rank.set(n_rank - 1, rv.ones); // the last entry always holds the total (:268)
    w.u64(n);
    bt.write(w);
    w.int_vector(btnr.data(), btnr_bits, 1);
    btnrp.write(w);
    rank.write(w);
    invert.write(w);
    return SDSL_HIP_OK;
}

// plain words of the bit vector described by parsed SDSL arrays (host-side decode; used to validate wavelet trees
// that arrive with an rrr-compressed bit vector)
void rrr_arrays_to_words(const RrrArrays & A, std::vector<uint64_t> & words)
{
    const RrrTables & T = host_tables();
    words.assign(((A.n_bits + 63) >> 6) + 2, 0);
    {
        uint64_t ptr = 0;
        for (uint64_t b = 0; b < A.n_blocks; ++b)
        {
            unsigned k = A.cls[b], len = T.sdsl_space[k];
            uint64_t bits = decode_block_host(T, k, read_bits(A.stream.data(), ptr, len));
            ptr += len;
            uint64_t pos = b * kRrrBS;
            if (pos >= A.n_bits || !bits)
                continue;
            unsigned off = (unsigned)(pos & 63);
            words[pos >> 6] |= bits << off;
            if (off + kRrrBS > 64)
                words[(pos >> 6) + 1] |= bits >> (64 - off);
        }
    }
}

sdsl_hip_status rrr_parse_and_upload(RrrHost & h, StreamReader & rd, int device, std::vector<uint64_t> * words_out)
{
    RrrArrays A;
    SH_TRY(rrr_parse_sdsl(rd, A));
    if (words_out)
        rrr_arrays_to_words(A, *words_out);
    return rrr_upload(h, A, device);
}
} // namespace sdslhip

using namespace sdslhip;

struct sdsl_hip_rrr_s
{
    RrrHost h;
    uint64_t uid = next_handle_uid(); // key of the serialiser's size-query cache
};

extern "C" {

static sdsl_hip_status sdsl_hip_rrr_create_impl(const uint64_t * words, uint64_t n_bits, int32_t device, sdsl_hip_rrr_t * out)
{
    if (!out || (!words && n_bits))
    {
        set_error("rrr_create: null argument");
        return SDSL_HIP_ERR_INVALID;
    }
    *out = nullptr;
    SH_TRY(check_device(device));
    sdsl_hip_rrr_s * r = new (std::nothrow) sdsl_hip_rrr_s();
    if (!r)
        return SDSL_HIP_ERR_NOMEM;
    Staged w; // host words are uploaded, device words are encoded where they are
    sdsl_hip_status st = w.in(words, ((n_bits + 63) >> 6) * 8, nullptr);
    r->h.allow_slim = true; // (a stand-alone vector: the record format follows its density)
    if (st == SDSL_HIP_OK)
        st = rrr_build_device(r->h, (const uint64_t *)w.dev, n_bits, device);
    if (st != SDSL_HIP_OK)
    {
        delete r;
        return st;
    }
    *out = r;
    return SDSL_HIP_OK;
}
// no exception crosses the C ABI: a malformed stream or an exhausted host becomes a status code
sdsl_hip_status sdsl_hip_rrr_create(const uint64_t * words, uint64_t n_bits, int32_t device, sdsl_hip_rrr_t * out)
{
    return guarded("rrr_create", [&] { return sdsl_hip_rrr_create_impl(words, n_bits, device, out); });
}

static sdsl_hip_status sdsl_hip_rrr_create_from_sdsl_impl(const void * bytes, size_t len, int32_t device, sdsl_hip_rrr_t * out)
{
    if (!out || !bytes)
    {
        set_error("rrr_create_from_sdsl: null argument");
        return SDSL_HIP_ERR_INVALID;
    }
    *out = nullptr;
    SH_TRY(check_device(device));
    StreamReader rd(bytes, len);
    RrrArrays A;
    SH_TRY(rrr_parse_sdsl(rd, A));
    sdsl_hip_rrr_s * r = new (std::nothrow) sdsl_hip_rrr_s();
    if (!r)
        return SDSL_HIP_ERR_NOMEM;
    r->h.allow_slim = true;
    sdsl_hip_status st = rrr_upload(r->h, A, device);
    if (st != SDSL_HIP_OK)
    {
        delete r;
        return st;
    }
    *out = r;
    return SDSL_HIP_OK;
}
// no exception crosses the C ABI: a malformed stream or an exhausted host becomes a status code
sdsl_hip_status sdsl_hip_rrr_create_from_sdsl(const void * bytes, size_t len, int32_t device, sdsl_hip_rrr_t * out)
{
    return guarded("rrr_create_from_sdsl", [&] { return sdsl_hip_rrr_create_from_sdsl_impl(bytes, len, device, out); });
}

// A sibling representation (bit_vector_il<>, the rrr_vector<15> specialisation, any rrr_vector<t_bs, int_vector<>, t_k>) kept COMPRESSED in
// HBM: its stream is decoded on the device (bv_compressed.hip) and the bits are re-encoded at once as this library's rrr records — the
// plain words exist for the time of the call only.  The resident form is the device's rrr_vector<63> layout (a 5 %-dense rrr_vector<15>
// of 0.58 bits/bit in SDSL is 0.40 here); rank / select / access do not depend on the representation.
static sdsl_hip_status sdsl_hip_rrr_create_from_sibling_impl(const void * bytes, size_t len, int32_t kind, int32_t device, sdsl_hip_rrr_t * out)
{
    if (!out || !bytes)
    {
        set_error("rrr_create_from_sibling: null argument");
        return SDSL_HIP_ERR_INVALID;
    }
    *out = nullptr;
    SH_TRY(check_device(device));
    DevBuf d_words;
    uint64_t n_bits = 0;
    SH_TRY(compressed_stream_to_device_words(bytes, len, kind, d_words, n_bits));
    sdsl_hip_rrr_s * r = new (std::nothrow) sdsl_hip_rrr_s();
    if (!r)
        return SDSL_HIP_ERR_NOMEM;
    r->h.allow_slim = true;
    const sdsl_hip_status st = rrr_build_device(r->h, d_words.as<uint64_t>(), n_bits, device);
    if (st != SDSL_HIP_OK)
    {
        delete r;
        return st;
    }
    *out = r;
    return SDSL_HIP_OK;
}
sdsl_hip_status sdsl_hip_rrr_create_from_sibling(const void * bytes, size_t len, int32_t kind, int32_t device, sdsl_hip_rrr_t * out)
{
    return guarded("rrr_create_from_sibling", [&] { return sdsl_hip_rrr_create_from_sibling_impl(bytes, len, kind, device, out); });
}

static sdsl_hip_status sdsl_hip_rrr_serialize_impl(sdsl_hip_rrr_t v, void * buf, size_t cap, size_t * written)
{
    if (!v)
    {
        set_error("rrr_serialize: null handle");
        return SDSL_HIP_ERR_INVALID;
    }
    sdsl_hip_status cached;
    if (deliver_cached(v->uid, 0, buf, cap, written, cached))
        return cached;
    StreamWriter w;
    SH_TRY(rrr_serialize_host(v->h, w));
    return deliver_and_cache(v->uid, 0, w, buf, cap, written);
}
// no exception crosses the C ABI: a malformed stream or an exhausted host becomes a status code
sdsl_hip_status sdsl_hip_rrr_serialize(sdsl_hip_rrr_t v, void * buf, size_t cap, size_t * written)
{
    return guarded("rrr_serialize", [&] { return sdsl_hip_rrr_serialize_impl(v, buf, cap, written); });
}

sdsl_hip_status sdsl_hip_rrr_destroy(sdsl_hip_rrr_t v)
{
    if (!v)
        return SDSL_HIP_OK;
    (void)hipSetDevice(v->h.device);
    device_scratch_quiesce(v->h.device); // (a bucketed batch over this vector may still be running on some stream)
    delete v;
    return SDSL_HIP_OK;
}
sdsl_hip_status sdsl_hip_rrr_reserve_capture_scratch(sdsl_hip_rrr_t v, uint64_t max_queries)
{
    if (!v)
    {
        set_error("rrr_reserve_capture_scratch: null handle");
        return SDSL_HIP_ERR_INVALID;
    }
    RrrHost & h = v->h;
    std::lock_guard<std::mutex> lock(h.scratch_mutex);
    SH_HIP(hipSetDevice(h.device));
    if (max_queries == 0)
    {
        h.capture_scratch.release();
        return SDSL_HIP_OK;
    }
    const uint64_t pass = max_queries < (UINT64_C(1) << 30) ? max_queries : (UINT64_C(1) << 30);
    const size_t need = bv_swc_scratch_bytes(BvView{}, pass);
    if (h.capture_scratch.bytes < need)
        SH_TRY(h.capture_scratch.alloc(need));
    if (!h.spread_probe.p)
        SH_TRY(h.spread_probe.alloc(64));
    for (int bit = 0; bit < 2; ++bit)
        if (h.view.sel[bit])
            SH_TRY(rrr_select_sorted_prepare(h, bit));
    return SDSL_HIP_OK;
}
uint64_t sdsl_hip_rrr_size(sdsl_hip_rrr_t v)
{
    return v ? v->h.view.n_bits : 0;
}
uint64_t sdsl_hip_rrr_ones(sdsl_hip_rrr_t v)
{
    return v ? v->h.view.ones : 0;
}
uint64_t sdsl_hip_rrr_device_bytes(sdsl_hip_rrr_t v)
{
    return v ? v->h.device_bytes() : 0;
}

static unsigned rrr_grid(uint64_t n)
{
    return grid_for(n, kRrrBlock, 256u * 4u);
}

// the direct kernels for the vector's record format
static void rrr_launch_rank_direct(const RrrView & v, int mode, int bit, const uint64_t * d_in, uint64_t * d_out, uint8_t * d_out8, uint64_t n,
                                   hipStream_t s)
{
    const dim3 grid(rrr_grid(n)), block(kRrrBlock);
    if (mode == 0)
    {
        if (v.fmt)
            hipLaunchKernelGGL((k_rrr_rank<0, RrrFmtS>), grid, block, 0, s, v, bit, d_in, d_out, d_out8, n);
        else
            hipLaunchKernelGGL((k_rrr_rank<0, RrrFmtW>), grid, block, 0, s, v, bit, d_in, d_out, d_out8, n);
    }
    else
    {
        if (v.fmt)
            hipLaunchKernelGGL((k_rrr_rank<1, RrrFmtS>), grid, block, 0, s, v, bit, d_in, d_out, d_out8, n);
        else
            hipLaunchKernelGGL((k_rrr_rank<1, RrrFmtW>), grid, block, 0, s, v, bit, d_in, d_out, d_out8, n);
    }
}
static void rrr_launch_select_direct(const RrrView & v, int bit, const uint64_t * d_in, uint64_t * d_out, uint64_t n, hipStream_t s)
{
    const dim3 grid(rrr_grid(n)), block(kRrrBlock);
    if (v.fmt)
    {
        if (bit)
            hipLaunchKernelGGL((rrr_select_kernel<1, RrrFmtS>()), grid, block, 0, s, v, d_in, d_out, n);
        else
            hipLaunchKernelGGL((rrr_select_kernel<0, RrrFmtS>()), grid, block, 0, s, v, d_in, d_out, n);
    }
    else
    {
        if (bit)
            hipLaunchKernelGGL((rrr_select_kernel<1, RrrFmtW>()), grid, block, 0, s, v, d_in, d_out, n);
        else
            hipLaunchKernelGGL((rrr_select_kernel<0, RrrFmtW>()), grid, block, 0, s, v, d_in, d_out, n);
    }
}

sdsl_hip_status sdsl_hip_rrr_rank_batch(sdsl_hip_rrr_t v, int32_t bit, const uint64_t * idx, uint64_t n,
                                        uint64_t * out, void * stream)
{
    if (!v || (bit != 0 && bit != 1) || (n && (!idx || !out)))
    {
        set_error("rrr_rank_batch: invalid argument");
        return SDSL_HIP_ERR_INVALID;
    }
    hipStream_t s = (hipStream_t)stream;
    SH_HIP(hipSetDevice(v->h.device));
    if (n == 0)
        return SDSL_HIP_OK;
    if (n >= kPipelineMinQueries && !is_device_ptr(idx) && !is_device_ptr(out))
    { // host arrays on both sides: chunked over two streams (common.hpp host_pipeline_u64)
        const RrrView rv = v->h.view;
        return host_pipeline_u64(v->h.device, idx, out, n,
                                 [rv, bit](const uint64_t * d_in, uint64_t * d_out, uint64_t cnt, hipStream_t st) -> sdsl_hip_status
                                 {
                                     rrr_launch_rank_direct(rv, 0, bit, d_in, d_out, nullptr, cnt, st);
                                     SH_HIP(hipGetLastError());
                                     return SDSL_HIP_OK;
                                 });
    }
    Staged in, o;
    SH_TRY(in.in(idx, n * 8, s));
    SH_TRY(o.out(out, n * 8));
    // A large batch over a large vector that is spread over it goes through the passes of bv_swc.hip around the slice-wise
    // decoder of rrr_sorted.hip (option "rrr_sorted": 0 never, 1 whenever possible, -1 automatic: the spread sample's verdict
    // stays on the device, both routes are enqueued, the one whose turn it is not returns at once — as for the plain vector)
    RrrHost & h = v->h;
    const int mode = g_rrr_sorted_mode.load();
    // (a vector that keeps classes above 10 enumerative — option "rrr_sparse_limit" — takes the same route since round 6: the slice
    // decoder of rrr_sorted.hip stages the binomial columns 3..20 for it)
    const bool want = mode == 0 ? false : (mode > 0 ? rrr_sorted_rank_possible(h.view) : rrr_sorted_rank_applicable(h.view, n));
    if (want)
    {
        std::lock_guard<std::mutex> lock(h.scratch_mutex);
        const uint64_t pass = n < (UINT64_C(1) << 30) ? n : (UINT64_C(1) << 30);
        ScratchLease L; // (ends — and records the pool's event — when this block is left, on every path)
        SH_TRY(L.acquire(h.device, h.capture_scratch, bv_swc_scratch_bytes(BvView{}, pass), s));
        bool have = L.p != nullptr;
        if (have && !h.spread_probe.p)
            have = !L.capturing && h.spread_probe.alloc(64) == SDSL_HIP_OK;
        if (have)
        {
            sdsl_hip_status st;
            {
                KernelTimer t(s);
                const uint32_t * go = nullptr;
                if (mode < 0)
                {
                    rrr_sorted_rank_sample(h.view, (const uint64_t *)in.dev, n, s, h.spread_probe.as<uint32_t>());
                    go = h.spread_probe.as<uint32_t>() + 2;
                }
                st = rrr_launch_rank_sorted(h.view, bit, (const uint64_t *)in.dev, n, (uint64_t *)o.dev, s, L.p, L.bytes, go);
                if (st == SDSL_HIP_OK && go)
                {
                    RrrView dv = h.view;
                    dv.skip_if = go;
                    rrr_launch_rank_direct(dv, 0, bit, (const uint64_t *)in.dev, (uint64_t *)o.dev, nullptr, n, s);
                }
            }
            SH_TRY(st);
            SH_HIP(hipGetLastError());
            SH_TRY(o.finish(s));
            if (in.host && !o.host)
                SH_HIP(hipStreamSynchronize(s));
            return SDSL_HIP_OK;
        }
    }
    {
        KernelTimer t(s);
        rrr_launch_rank_direct(v->h.view, 0, bit, (const uint64_t *)in.dev, (uint64_t *)o.dev, nullptr, n, s);
    }
    SH_HIP(hipGetLastError());
    SH_TRY(o.finish(s));
    if (in.host && !o.host)
        SH_HIP(hipStreamSynchronize(s));
    return SDSL_HIP_OK;
}

sdsl_hip_status sdsl_hip_rrr_access_batch(sdsl_hip_rrr_t v, const uint64_t * idx, uint64_t n, uint8_t * out,
                                          void * stream)
{
    if (!v || (n && (!idx || !out)))
    {
        set_error("rrr_access_batch: invalid argument");
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
        rrr_launch_rank_direct(v->h.view, 1, 1, (const uint64_t *)in.dev, nullptr, (uint8_t *)o.dev, n, s);
    }
    SH_HIP(hipGetLastError());
    SH_TRY(o.finish(s));
    if (in.host && !o.host)
        SH_HIP(hipStreamSynchronize(s));
    return SDSL_HIP_OK;
}

sdsl_hip_status sdsl_hip_rrr_get_int_batch(sdsl_hip_rrr_t v, const uint64_t * idx, uint32_t len, uint64_t n, uint64_t * out,
                                           void * stream)
{
    if (!v || len > 64 || (n && (!idx || !out)))
    {
        set_error("rrr_get_int_batch: invalid argument (len must be <= 64)");
        return SDSL_HIP_ERR_INVALID;
    }
    hipStream_t s = (hipStream_t)stream;
    SH_HIP(hipSetDevice(v->h.device));
    if (n == 0)
        return SDSL_HIP_OK;
    Staged in, o;
    SH_TRY(in.in(idx, n * 8, s));
    SH_TRY(o.out(out, n * 8));
    {
        KernelTimer t(s);
        if (v->h.view.fmt)
            hipLaunchKernelGGL(k_rrr_get_int<RrrFmtS>, dim3(rrr_grid(n)), dim3(kRrrBlock), 0, s, v->h.view, len, (const uint64_t *)in.dev,
                               (uint64_t *)o.dev, n);
        else
            hipLaunchKernelGGL(k_rrr_get_int<RrrFmtW>, dim3(rrr_grid(n)), dim3(kRrrBlock), 0, s, v->h.view, len, (const uint64_t *)in.dev,
                               (uint64_t *)o.dev, n);
    }
    SH_HIP(hipGetLastError());
    SH_TRY(o.finish(s));
    if (in.host && !o.host)
        SH_HIP(hipStreamSynchronize(s));
    return SDSL_HIP_OK;
}

} // extern "C"
namespace sdslhip {
sdsl_hip_status rrr_launch_select(const RrrView & v, int bit, const uint64_t * d_i, uint64_t n, uint64_t * d_out, hipStream_t s)
{
    if (n == 0)
        return SDSL_HIP_OK;
    KernelTimer t(s);
    rrr_launch_select_direct(v, bit, d_i, d_out, n, s);
    SH_HIP(hipGetLastError());
    return SDSL_HIP_OK;
}
} // namespace sdslhip
extern "C" {

sdsl_hip_status sdsl_hip_rrr_select_batch(sdsl_hip_rrr_t v, int32_t bit, const uint64_t * i, uint64_t n,
                                          uint64_t * out, void * stream)
{
    if (!v || (bit != 0 && bit != 1) || (n && (!i || !out)))
    {
        set_error("rrr_select_batch: invalid argument");
        return SDSL_HIP_ERR_INVALID;
    }
    hipStream_t s = (hipStream_t)stream;
    SH_HIP(hipSetDevice(v->h.device));
    if (n == 0)
        return SDSL_HIP_OK;
    if (n >= kPipelineMinQueries && !is_device_ptr(i) && !is_device_ptr(out))
    {
        const RrrView rv = v->h.view;
        return host_pipeline_u64(v->h.device, i, out, n,
                                 [rv, bit](const uint64_t * d_in, uint64_t * d_out, uint64_t cnt, hipStream_t st) -> sdsl_hip_status
                                 {
                                     rrr_launch_select_direct(rv, bit, d_in, d_out, cnt, st);
                                     SH_HIP(hipGetLastError());
                                     return SDSL_HIP_OK;
                                 });
    }
    Staged in, o;
    SH_TRY(in.in(i, n * 8, s));
    SH_TRY(o.out(out, n * 8));
    // large spread batches: the bucketed path (rrr_sorted.hip), chosen as for rank (option "rrr_sorted")
    RrrHost & h = v->h;
    const int mode = g_rrr_sorted_mode.load();
    if (mode != 0 && h.view.sel[bit] && (mode > 0 || (h.view.n_sb >= (UINT64_C(1) << 21) && n >= 8 * h.view.n_sb)))
    {
        std::lock_guard<std::mutex> lock(h.scratch_mutex);
        const bool cap = stream_is_capturing(s); // (nothing may be built or allocated then)
        bool have = !cap || (h.sel_plan[bit].ready && h.spread_probe.p && h.capture_scratch.p);
        if (have)
            SH_TRY(rrr_select_sorted_prepare(h, bit)); // first use: the bucket boundaries (one small batch, synchronous)
        have = have && (mode > 0 ? h.sel_plan[bit].ok : rrr_sorted_select_applicable(h, bit, n));
        const uint64_t pass = n < (UINT64_C(1) << 30) ? n : (UINT64_C(1) << 30);
        ScratchLease L;
        if (have)
        {
            SH_TRY(L.acquire(h.device, h.capture_scratch, bv_swc_scratch_bytes(BvView{}, pass), s));
            have = L.p != nullptr;
            if (have && !h.spread_probe.p)
                have = !L.capturing && h.spread_probe.alloc(64) == SDSL_HIP_OK;
        }
        if (have)
        {
            sdsl_hip_status st;
            {
                KernelTimer t(s);
                const uint32_t * go = nullptr;
                if (mode < 0)
                {
                    rrr_sorted_select_sample(h, bit, (const uint64_t *)in.dev, n, s, h.spread_probe.as<uint32_t>());
                    go = h.spread_probe.as<uint32_t>() + 2;
                }
                st = rrr_launch_select_sorted(h, bit, (const uint64_t *)in.dev, n, (uint64_t *)o.dev, s, L.p, L.bytes, go);
                if (st == SDSL_HIP_OK && go)
                {
                    TimingPause pause;
                    RrrView dv = h.view;
                    dv.skip_if = go;
                    st = rrr_launch_select(dv, bit, (const uint64_t *)in.dev, n, (uint64_t *)o.dev, s);
                }
            }
            SH_TRY(st);
            SH_HIP(hipGetLastError());
            SH_TRY(o.finish(s));
            if (in.host && !o.host)
                SH_HIP(hipStreamSynchronize(s));
            return SDSL_HIP_OK;
        }
    }
    SH_TRY(rrr_launch_select(v->h.view, bit, (const uint64_t *)in.dev, n, (uint64_t *)o.dev, s));
    SH_HIP(hipGetLastError());
    SH_TRY(o.finish(s));
    if (in.host && !o.host)
        SH_HIP(hipStreamSynchronize(s));
    return SDSL_HIP_OK;
}
}
