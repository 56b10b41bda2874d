// bv_sorted_dev.hpp — what the two bucketed pipelines (bv_sorted.hip: one-sweep look-back; bv_swc.hip: static streams with
// write combining) share: key layout, digits, block scans.  DESIGN.md §3.5.
#pragma once
#include <type_traits>
#include <functional>
#include <string>

#include "bv_host.hpp"

namespace sdslhip {

// geometry of one pass over a batch (a plain struct with external linkage: it crosses translation units)
struct SrGeom
{
    uint64_t n;      // positions in this pass over the batch (< 2^31)
    uint64_t n_bits;
    uint64_t n_lines;
    uint32_t d1, d2; // digit widths
    uint32_t G;      // blocks of the partition kernels
    uint32_t tiles1; // tiles of pass 1
    uint32_t tile;   // keys per tile
    uint32_t op;     // 0: rank (keys = positions), 1: select (keys = argument ranks)
    uint32_t kb;     // bits of the final key: line in slice + bit in line (rank), rank inside the bucket (select)
    uint32_t B;      // select: ranks per bucket = m << bs with m in 8..15 (a power of two would waste up to half an LDS slice)
    uint32_t bs;     // select: the shift of B
    uint32_t binv;   // select: ceil(2^32 / m); floor(x / m) = (x * binv) >> 32 for x < 2^28
    uint64_t total;  // select: arguments of the vector (ones or zeros)
    bool small;      // 32-bit division path of line_of
    uint64_t slice_bits; // bits of the vector a slice covers (rank_0: zeros in front of a slice = its first bit - ones in front)
    uint32_t rbits;  // op 2 (rank on rrr records): bits per record; a key is [record in the slice : 8 | block : 6 | bit : 6]
    uint32_t rlog;   // op 2: log2 of the records per slice (7 or 8)
    uint32_t slog;   // op 0 / 4: log2 of the lines a slice covers: kSliceLog, or up to kSliceLog + kSliceExtraMax on vectors of more than
                     // 2^26 lines — the answering kernel then stages a slice in rounds of 2^kSliceLog lines (bv_sorted.hip: k_sr_rank_lds)
    uint32_t over_is_size; // op 1: an argument beyond the last one is not NPOS but size() (select_support_rrr, rrr_vector.hpp:641-642):
                           // its key is kMark and the fix-up pass writes the answer
    const uint32_t * go; // automatic dispatch: the passes return at once when this word is zero (nullptr: always run)
};

namespace {

constexpr uint32_t kSrRecBits = 63 * 34;   // bits per record of the rrr_vector<63> device layout (rrr_device.hpp: kRecSB)
constexpr uint32_t kSrRecBitsSlim = 63 * 42; // ... of its slim format (RrrFmtS::SB)
constexpr unsigned kRT = 512;            // threads of a rank block (two per CU)
constexpr unsigned kST = 1024;           // threads of a select block: its search is a chain of dependent LDS reads per key, and with 8
                                         // waves per SIMD instead of 4 the VALU stays busy (k_sr_select_lds: 4.42 -> 3.54 ms); rank loses (1.90 -> 1.98)
constexpr unsigned kBins = 256;          // bins per pass (8-bit digits)
constexpr unsigned kSliceLog = 10;       // lines per slice (64 KiB)
constexpr unsigned kOffBits = 9;         // 448 < 2^9 in-line offsets
constexpr unsigned kKey2Bits = kSliceLog + kOffBits;
constexpr unsigned kSliceExtraMax = 3;   // a slice of up to 2^13 lines: vectors of up to 2^29 lines (2^37.8 bits)
constexpr uint32_t kBad = 0xFFFFFFFFu;   // key / answer of a position beyond the vector (answer NPOS)
constexpr uint32_t kMark = 0xFFFFFFFEu;  // select: answer left to the fix-up pass (bucket wider than an LDS slice)
constexpr uint64_t kMark64 = SDSL_HIP_NPOS - 1;
constexpr unsigned kBigRun = 512;        // a (tile, bin) run longer than this is copied by the whole block
constexpr unsigned kItemKeys = 32768;    // keys of one slice handled by one block before the slice is reloaded


struct SrBuf
{ // carved out of the scratch allocation
    uint32_t *keys1, *keys2;   // keys1 doubles as the low 32 bits of the absolute answers on the way back
    uint16_t *slots1, *slots2;
    uint16_t *thist1, *thist2; // [tile][bin]
    uint32_t *counts1, *offs1, *bstart1;
    uint32_t *counts2, *offs2, *bstart2;
    uint32_t *btot, *tprefix2;
    uint32_t *tot2, *status, *ticket; // one-sweep pass 2: digit-2 totals, per (tile, bin) look-back words, tile tickets
    uint32_t *fine_count, *fstart, *ioff; // per slice: keys, first key, first work item
    uint64_t * hf;                        // per slice: ones in front of it
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

// In-place exclusive scan of a[0 .. kBins) in LDS by the first kBins threads of the block (all threads must call);
// returns the total.  Ends with a barrier.
__device__ __forceinline__ unsigned block_excl_scan_bins(unsigned * a, unsigned * wsum)
{
    const unsigned t = threadIdx.x;
    unsigned v = 0, inc = 0;
    if (t < kBins)
    {
        v = a[t];
        inc = wave_incl_scan(v);
        if ((t & 63) == 63)
            wsum[t >> 6] = inc;
    }
    __syncthreads();
    if (t < kBins)
    {
        unsigned base = 0;
        for (unsigned w = 0; w < (t >> 6); ++w)
            base += wsum[w];
        a[t] = base + inc - v;
    }
    unsigned total = 0;
    for (unsigned w = 0; w < kBins / 64; ++w)
        total += wsum[w];
    __syncthreads();
    return total;
}

template <uint32_t REC_BITS>
__device__ __forceinline__ void sr_key1_rrr(uint64_t pos, const SrGeom & g, unsigned & dig, uint32_t & key)
{
    // pos / REC_BITS in 32-bit arithmetic (positions are below 2^40): pos = hi 2^20 + lo, 2^20 = q0 REC + r0, so
    // pos / REC = hi q0 + (hi r0 + lo) / REC with a 32-bit numerator — half the instructions of the 64-bit division by a constant
    constexpr uint32_t q0 = (1u << 20) / REC_BITS, r0 = (1u << 20) % REC_BITS;
    static_assert((uint64_t)(r0 + 1) << 20 < (UINT64_C(1) << 32), "numerator of the second term must fit 32 bits");
    const uint32_t hi = (uint32_t)(pos >> 20), lo = (uint32_t)pos & 0xFFFFFu;
    const uint32_t num = hi * r0 + lo, q1 = num / REC_BITS;
    uint32_t rec = hi * q0 + q1;
    unsigned in_rec = num - q1 * REC_BITS;
    if (in_rec == 0 && pos == g.n_bits && pos != 0)
    { // rank(size()) when the vector ends with a record: "all 63 bits of its last block"
        --rec;
        in_rec = REC_BITS;
    }
    const unsigned blk = in_rec == REC_BITS ? REC_BITS / 63 - 1 : in_rec / 63, off = in_rec - blk * 63; // off <= 63
    const uint32_t sl = rec >> g.rlog;
    dig = sl >> g.d2;
    key = ((sl & ((1u << g.d2) - 1)) << g.kb) | ((rec & ((1u << g.rlog) - 1)) << 12) | (blk << 6) | off;
}

// pass 1: digit and 32-bit key of a position
__device__ __forceinline__ void sr_key1(uint64_t pos, const SrGeom & g, unsigned & dig, uint32_t & key)
{
    if (g.op == 1)
    { // select: the argument is the 1-based rank i of the wanted bit; buckets of 2^r consecutive ranks
        if (pos == 0 || pos > g.total)
        {
            dig = 0;
            key = pos != 0 && g.over_is_size ? kMark : kBad;
            return;
        }
        const uint64_t k = pos - 1;
        const uint32_t f = (uint32_t)(((k >> g.bs) * g.binv) >> 32); // k / B
        dig = f >> g.d2;
        key = ((f & ((1u << g.d2) - 1)) << g.kb) | (uint32_t)(k - (uint64_t)f * g.B);
        return;
    }
    if (pos > g.n_bits)
    {
        dig = 0;
        key = kBad;
        return;
    }
    if (g.op >= 2)
    { // rank on an rrr_vector<63>'s records (rrr_sorted.hip): record, 63-bit block inside it, bit inside the block
        // (the record length is a compile-time constant here: a 64-bit division by a run-time value costs more than the rest
        // of the pass — 4.4 instead of 1.6 ms per 10^9 keys in the counting pass; op 2: wide records, op 3: slim ones)
        if (g.op == 2)
            sr_key1_rrr<kSrRecBits>(pos, g, dig, key);
        else
            sr_key1_rrr<kSrRecBitsSlim>(pos, g, dig, key);
        return;
    }
    uint64_t L;
    unsigned off;
    line_of(pos, g.small, L, off);
    const uint32_t l = (uint32_t)L; // < 2^29
    dig = l >> (g.slog + g.d2);
    key = (((l >> g.slog) & ((1u << g.d2) - 1)) << g.kb) | ((l & ((1u << g.slog) - 1)) << kOffBits) | off;
}
// The same with the kind of query fixed at compile time (OP = SrGeom::op; plain rank: 0 for vectors below 2^38 bits, 4 above) and
// without per-key branches: the counting pass and pass 1 are bound by the instructions they issue per key (47 and 54 of them —
// 1.2 and 1.4 ms of VALU issue per 10^9 keys), and with the run-time dispatch every one of the sixteen unrolled keys carried its
// own ladder of scalar branches (the counting kernel had grown to 4800 instructions).
template <int OP>
__device__ __forceinline__ void sr_key1_t(uint64_t pos, const SrGeom & g, unsigned & dig, uint32_t & key)
{
    if constexpr (OP == 1)
    { // (with its early exit: computed for all keys side by side, the 64-bit products of sixteen unrolled keys spilled 44 VGPRs
      // in the counting kernel — 1.5 -> 3.0 ms)
        if (pos == 0 || pos > g.total)
        {
            dig = 0;
            key = pos != 0 && g.over_is_size ? kMark : kBad;
            return;
        }
        const uint64_t k = pos - 1;
        const uint32_t f = (uint32_t)(((k >> g.bs) * g.binv) >> 32); // k / B
        dig = f >> g.d2;
        key = ((f & ((1u << g.d2) - 1)) << g.kb) | (uint32_t)(k - (uint64_t)f * g.B);
    }
    else if constexpr (OP == 2 || OP == 3)
    {
        const bool bad = pos > g.n_bits;
        unsigned d;
        uint32_t ky;
        sr_key1_rrr<OP == 2 ? kSrRecBits : kSrRecBitsSlim>(pos, g, d, ky);
        dig = bad ? 0u : d;
        key = bad ? kBad : ky;
    }
    else
    {
        const bool bad = pos > g.n_bits;
        uint64_t L;
        unsigned off;
        line_of(pos, OP == 0, L, off);
        const uint32_t l = (uint32_t)L; // < 2^29
        dig = bad ? 0u : l >> (g.slog + g.d2);
        key = bad ? kBad : (((l >> g.slog) & ((1u << g.d2) - 1)) << g.kb) | ((l & ((1u << g.slog) - 1)) << kOffBits) | off;
    }
}
// calls f with the compile-time kind of a geometry's queries
template <class Fn>
inline void sr_dispatch_op(const SrGeom & g, Fn && f)
{
    switch (g.op)
    {
        case 1: f(std::integral_constant<int, 1>{}); break;
        case 2: f(std::integral_constant<int, 2>{}); break;
        case 3: f(std::integral_constant<int, 3>{}); break;
        default:
            if (g.small)
                f(std::integral_constant<int, 0>{});
            else
                f(std::integral_constant<int, 4>{});
    }
}
// pass 2: digit and final key of a pass-1 key
__device__ __forceinline__ void sr_key2(uint32_t k1, const SrGeom & g, unsigned & dig, uint32_t & key)
{
    if (k1 >= kMark)
    { // NPOS / left to the fix-up pass: travels with slice 0
        dig = 0;
        key = k1;
        return;
    }
    dig = k1 >> g.kb;
    key = k1 & ((1u << g.kb) - 1);
}

// The tables per slice (fstart, ioff, hf) are indexed in the order of the final array: f = (pass-2 digit << d1) | pass-1 digit.
// The slice itself (rank: 2^10 lines; select: a bucket of argument ranks) is (pass-1 digit << d2) | pass-2 digit.
__device__ __forceinline__ unsigned sr_slice_of(unsigned f, unsigned d1, unsigned d2)
{
    return ((f & ((1u << d1) - 1)) << d2) | (f >> d1);
}

// ---- buffer addressing: a uniform base in SGPRs + a 32-bit lane offset; whatever lies beyond num_records reads as 0 and is not
// written — tiles need no per-key bounds checks and no 64-bit address arithmetic in VGPRs ----
typedef __amdgpu_buffer_rsrc_t rsrc_t;
constexpr int kAuxNT = 2; // streamed once: non-temporal
__device__ __forceinline__ rsrc_t make_rsrc(const void * p, uint32_t bytes)
{
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<void *>(p), (short)0, (int)bytes, 0x00020000);
}
__device__ __forceinline__ uint64_t uniform64(uint64_t v)
{
    return ((uint64_t)__builtin_amdgcn_readfirstlane((unsigned)(v >> 32)) << 32) | __builtin_amdgcn_readfirstlane((unsigned)v);
}
__device__ __forceinline__ void buf_load(rsrc_t r, unsigned voff, unsigned soff, uint32_t & out)
{
    out = __builtin_amdgcn_raw_buffer_load_b32(r, (int)voff, (int)soff, kAuxNT);
}
// the same through the caches (AUX 0): for data that is read again soon
__device__ __forceinline__ void buf_load_cached(rsrc_t r, unsigned voff, unsigned soff, uint32_t & out)
{
    out = __builtin_amdgcn_raw_buffer_load_b32(r, (int)voff, (int)soff, 0);
}
__device__ __forceinline__ void buf_load(rsrc_t r, unsigned voff, unsigned soff, uint64_t & out)
{
    typedef unsigned v2u32 __attribute__((ext_vector_type(2)));
    const v2u32 v = __builtin_amdgcn_raw_buffer_load_b64(r, (int)voff, (int)soff, kAuxNT);
    out = ((uint64_t)v.y << 32) | v.x;
}


} // namespace

void bv_sorted_set_phases(const std::string & line); // what sdsl_hip_last_phases reports (bv_sorted.hip)

// HIP events around the passes of one bucketed call (options "trace_phases" / SDSL_HIP_TRACE_SORTED)
struct PhaseTimer
{
    static constexpr int kMax = 16;
    bool on;
    hipStream_t s;
    hipEvent_t ev[kMax];
    const char * nm[kMax];
    int n = 0;
    PhaseTimer(bool on_, hipStream_t s_) : on(on_), s(s_)
    {
        if (on)
            for (auto & e : ev)
                (void)hipEventCreate(&e);
    }
    // the first call opens the first phase; every later one closes the phase `name` (legacy table when null)
    void mark(const char * name = nullptr)
    {
        static const char * legacy[] = {"", "hist1", "offs1", "part1", "hist2", "offs2", "part2", "slices", "rank", "unperm2", "unperm1",
                                        "p11",  "p12",   "p13",   "p14",   "p15"};
        if (on && n < kMax)
        {
            nm[n] = name ? name : legacy[n];
            (void)hipEventRecord(ev[n++], s);
        }
    }
    std::string line(int op, float * total = nullptr)
    {
        std::string out = op ? "select=1" : "select=0"; // which query the passes below belong to
        if (!on || n == 0)
            return out;
        (void)hipEventSynchronize(ev[n - 1]);
        float sum = 0;
        for (int i = 0; i + 1 < n; ++i)
        {
            float ms = 0;
            (void)hipEventElapsedTime(&ms, ev[i], ev[i + 1]);
            sum += ms;
            char buf[64];
            snprintf(buf, sizeof(buf), ";%s=%.4f", nm[i + 1], ms);
            out += buf;
        }
        if (total)
            *total = sum;
        return out;
    }
    void report(const SrGeom & g, const char * what)
    {
        if (!on)
            return;
        float total = 0;
        const std::string l = line((int)g.op, &total);
        fprintf(stderr, "[sdsl_hip] %s: n=%llu d1=%u d2=%u | %s | total %.3f ms = %.2f G/s\n", what, (unsigned long long)g.n, g.d1, g.d2,
                l.c_str(), total, g.n / total / 1e6);
    }
    void keep(int op)
    {
        if (on)
            bv_sorted_set_phases(line(op));
    }
    ~PhaseTimer()
    {
        if (on)
            for (auto & e : ev)
                (void)hipEventDestroy(e);
    }
};

struct SelectPlan
{ // select only
    const uint32_t * bnd = nullptr; // nf + 1 line indices
    unsigned bm = 8, bs = 3, nf = 0; // buckets of bm << bs ranks
    uint64_t total = 0;
};

// digits and key widths of a pass over `cnt` positions (everything but the tiling)
inline void sr_fill_geom(SrGeom & g, const BvView & v, int op, const SelectPlan & sp, uint64_t cnt)
{
    g.n = cnt;
    g.n_bits = v.n_bits;
    g.n_lines = v.n_lines;
    g.op = (uint32_t)op;
    unsigned f = 0; // bits of a slice / bucket index
    g.slog = kSliceLog;
    if (op == 0)
    {
        unsigned lb = 0; // bits of a line index
        while ((v.n_lines - 1) >> lb)
            ++lb;
        if (lb > kSliceLog + 16) // more than 2^16 slices of 2^kSliceLog lines: wider slices, answered in rounds
            g.slog = lb - 16;
        f = lb > g.slog ? lb - g.slog : 0;
        g.kb = g.slog + kOffBits;
        g.B = 0;
        g.bs = 0;
        g.binv = 0;
        g.total = 0;
        g.slice_bits = (UINT64_C(1) << g.slog) * kDB;
    }
    else
    {
        while (sp.nf > (1u << f))
            ++f;
        g.kb = sp.bs + 4; // B <= 15 << bs
        g.B = sp.bm << sp.bs;
        g.bs = sp.bs;
        g.binv = (uint32_t)(((UINT64_C(1) << 32) + sp.bm - 1) / sp.bm);
        g.total = sp.total;
    }
    g.d2 = f < 8 ? f : 8; // pass 2: the LOW bits of the slice index; pass 1: the high ones, so that a pass-1 bin is a contiguous
    g.d1 = f - g.d2;      // stretch of the vector and the answers of its keys fit 32 bits relative to the stretch's first one
    g.small = v.n_bits < (UINT64_C(1) << 38);
    g.go = nullptr;
    g.rbits = 0;
    g.rlog = 0;
    g.over_is_size = 0;
    if (op != 0)
        g.slice_bits = 0;
}

sdsl_hip_status sr_launch_answers(const BvView & v, int op, int bit, const SelectPlan & sp, unsigned nf, unsigned d2, unsigned slog, const uint32_t * fstart,
                                  const uint32_t * ioff, uint32_t * keys2, uint64_t * hf, uint32_t * marked, const uint32_t * go, hipStream_t s);
void sr_launch_select_fixup(const BvView & v, int bit, const uint32_t * marked, const uint64_t * idx, uint64_t * out, uint64_t cnt,
                            const uint32_t * go, hipStream_t s);

// the write-combined pipeline (bv_swc.hip): same contract as the passes of bv_sorted.hip
size_t bv_swc_scratch_bytes(const BvView & v, uint64_t n);
sdsl_hip_status sw_run(const BvView & v, int op, int bit, const SelectPlan & sp, const uint64_t * d_idx, uint64_t n, uint64_t * d_out,
                       hipStream_t s, void * scratch, size_t scratch_bytes, const uint32_t * go);
// the passes around any answering kernel: `fill` sets digits / key width / op of a pass over cnt positions (everything but the
// tiling), `answers` enqueues the kernel(s) that answer the slices in place over the final keys and fill the per-slice bases,
// `fixup` (may be empty) what runs on the caller's arrays afterwards
struct SwCallbacks
{
    std::function<void(SrGeom &, uint64_t)> fill;
    std::function<sdsl_hip_status(const SrGeom &, unsigned, const uint32_t *, const uint32_t *, uint32_t *, uint64_t *, uint32_t *, hipStream_t)> answers;
    std::function<void(const uint32_t *, const uint64_t *, uint64_t *, uint64_t, hipStream_t)> fixup;
    const char * what = "bucketed (write-combined)";
};
sdsl_hip_status sw_run_with(const SwCallbacks & cb, int bit, const uint64_t * d_idx, uint64_t n, uint64_t * d_out, hipStream_t s,
                            void * scratch, size_t scratch_bytes, const uint32_t * go);
// the spread sample on any geometry (out3[2] = verdict)
void sr_launch_sample(const SrGeom & g, const uint64_t * d_idx, uint32_t * out3, hipStream_t s);

// table kernels of bv_sorted.hip, launched on behalf of bv_swc.hip
// offs[b][g] = keys of bins < b + keys of bin b in streams < g, from counts[b][g]; bstart = bin starts (bins + 1 entries)
void sr_launch_bin_offsets(unsigned bins, unsigned G, const uint32_t * counts, uint32_t * btot, uint32_t * bstart, uint32_t * offs,
                           hipStream_t s);
// per slice: first key (fstart) and first work item (ioff) from the keys per slice
void sr_launch_fine_scan(unsigned nf, const uint32_t * fine_count, uint32_t * fstart, uint32_t * ioff, hipStream_t s);

} // namespace sdslhip
