// locate.hip — the rest of the csa_wt query API on the device: SA and ISA access, LF and psi, extract, locate.
//
// Reference semantics reproduced:
//   csa_wt::operator[] (SA[i])             csa_wt.hpp:363-381          walk LF until a sampled SA index
//   sa_order_sa_sampling                   csa_sampling_strategy.hpp:72-135   SA[k * dens]
//   csa.isa[i]                             suffix_array_helper.hpp:519-537    nearest ISA sample to the right + LF
//   isa_sampling::sample_qeq               csa_sampling_strategy.hpp:795-799
//   csa.lf[i]                              suffix_array_helper.hpp:346-360
//   csa.psi[i]                             suffix_array_helper.hpp:330-342
//   extract(csa, begin, end, text) lf_tag  suffix_array_algorithm.hpp:578-600
//   locate(csa, begin, end)                suffix_array_algorithm.hpp:505-523 (occurrences in SA order)
//
// All of them are LF walks: one LF step is one wt.inverse_select (wt_pc.hpp:411-430) plus a C[] lookup, i.e. one rank
// line per tree level.  The walks have very different lengths (SA[i] stops at the first sampled index: geometric with
// mean `dens`), so the kernel is a FLAT loop — one iteration is one tree level of whatever walk the quad (plain
// bit vector) or lane (rrr) is on — and a finished walker takes its next query at once instead of waiting for the
// slowest walker of its wave.
//
// An index created from text keeps its whole suffix array on the device (4 bytes per symbol — cheap against 288 GB),
// so SA[i] and locate are plain gathers there; sdsl_hip_fm_drop_sa() reduces it to SDSL's default samples.
#include "fm_host.hpp"

namespace sdslhip {

// ---- execution policies: who walks, and how one tree level is taken --------------------------------------------
struct LocPlain
{
    static constexpr unsigned kLanes = kG, kThreads = kBlock;
    struct Shared
    {
        WtTables T;
        WtFusedTables FT;
        FmTables F;
    };
    static __device__ __forceinline__ void stage(Shared * S, const WtView & wt, const FmTables * ftab)
    {
        fm_stage_tables(&S->F, ftab);
        wt_stage_tables(&S->T, wt.tables);
        wt_stage_fused(&S->FT, wt);
    }
    // one step from node v at offset i; true: a leaf is reached, c = its symbol
    template <class I>
    static __device__ __forceinline__ bool step(const WtView & wt, const Shared * S, int s, unsigned & v, I & i, unsigned & c)
    {
        if (S->T.child[v][0] != kWtUndef) // (a one-symbol tree is a single leaf)
        {
            if (wt.f_lines) // several levels per step
                quad_wt8_invsel_step<false>(wt, &S->T, &S->FT, s, v, i);
            else
            {
                uint64_t i64 = i;
                quad_wt_invsel_level<false>(wt, &S->T, s, v, i64);
                i = (I)i64;
            }
        }
        c = (unsigned)S->T.bv_pos_rank[v];
        return S->T.child[v][0] == kWtUndef;
    }
};

// the fused lines walked by fused node (wt_device.hpp: WtFusedWalk): v is the fused node's index, no binary node table in LDS
struct LocFused
{
    static constexpr unsigned kLanes = kG, kThreads = kBlock;
    struct Shared
    {
        WtFusedWalk W;
        FmTables F;
    };
    static __device__ __forceinline__ void stage(Shared * S, const WtView & wt, const FmTables * ftab)
    {
        fm_stage_tables(&S->F, ftab);
        const uint64_t * src = reinterpret_cast<const uint64_t *>(wt.f_walk);
        uint64_t * dst = reinterpret_cast<uint64_t *>(&S->W);
        static_assert(sizeof(WtFusedWalk) % 8 == 0, "copied in 8-byte words");
        for (unsigned k = threadIdx.x; k < sizeof(WtFusedWalk) / 8; k += blockDim.x)
            dst[k] = src[k];
        __syncthreads();
    }
    template <class I>
    static __device__ __forceinline__ bool step(const WtView & wt, const Shared * S, int s, unsigned & v, I & i, unsigned & c)
    {
        const unsigned e = quad_wtf_invsel_step<false>(wt, &S->W, s, v, i);
        c = e & 0xFFu;
        v = e;
        return (e & kFWalkLeaf) != 0;
    }
};

struct LocRrr
{
    static constexpr unsigned kLanes = 1, kThreads = 512;
    struct Shared
    {
        WtTables T;
        FmTables F;
        RrrTables RT;
    };
    static __device__ __forceinline__ void stage(Shared * S, const WtView & wt, const FmTables * ftab)
    {
        fm_stage_tables(&S->F, ftab);
        rrr_stage_tables(&S->RT, wt.rrr.tables);
        wt_stage_tables(&S->T, wt.tables);
    }
    template <class I>
    static __device__ __forceinline__ bool step(const WtView & wt, const Shared * S, int, unsigned & v, I & i, unsigned & c)
    {
        if (S->T.child[v][0] != kWtUndef)
        {
            unsigned bit = 0;
            const uint64_t r = rrr_rank1(wt.rrr, &S->RT, S->T.bv_pos[v] + i, &bit) - S->T.bv_pos_rank[v];
            i = (I)(bit ? r : i - r);
            v = S->T.child[v][bit];
        }
        c = (unsigned)S->T.bv_pos_rank[v];
        return S->T.child[v][0] == kWtUndef;
    }
};

enum
{
    kWalkSa = 0,      // in: SA index; walk to the first sampled index; out: SA value
    kWalkIsa = 1,     // in: text position; walk from the ISA sample to its right; out: ISA value
    kWalkLf = 2,      // in: SA index; one step; out: LF value
    kWalkExtract = 3, // in: (begin, end); walk from the ISA sample right of end; out: the bytes of text[begin..end]
};

// ISA sample to the right of text position i and its position (csa_sampling_strategy.hpp:795-799); past the last
// sample this is the sample of position 0, which stands for position n of the cyclic text
__device__ __forceinline__ void isa_sample_right(const FmLocView & L, uint64_t i, uint64_t & order, uint64_t & pos)
{
    const uint64_t ci = (i / L.isa_dens + 1) % L.n_isa_s;
    pos = ci * L.isa_dens;
    order = loc_sample(L.isa_s, L.s32, ci);
}

// WIDE = false: an index of fewer than 2^32 symbols — positions, step counts and SA values are 32-bit in the loop (registers and
// instructions: the walks are bound by both)
template <class P, int MODE, bool WIDE>
__global__ __launch_bounds__(P::kThreads) void k_fm_walk(WtView wt, const FmTables * __restrict__ ftab, FmLocView L,
                                                         const uint64_t * __restrict__ in0,
                                                         const uint64_t * __restrict__ in1,
                                                         const uint64_t * __restrict__ out_off, uint64_t n,
                                                         uint64_t * __restrict__ out, uint8_t * __restrict__ out_text)
{
    __shared__ typename P::Shared S;
    P::stage(&S, wt, ftab);
    const int s = threadIdx.x % P::kLanes;
    constexpr unsigned kWalkers = P::kThreads / P::kLanes;
    const uint64_t stride = (uint64_t)gridDim.x * kWalkers;
    uint64_t q = (uint64_t)blockIdx.x * kWalkers + threadIdx.x / P::kLanes;
    uint64_t a_next = q < n ? in0[q] : 0; // the walker's next argument, loaded one query ahead
    bool busy = false;
    typedef typename std::conditional<WIDE, uint64_t, uint32_t>::type pos_t;
    const pos_t size = (pos_t)L.size; // (WIDE = false: size < 2^32)
    pos_t j = 0, i = 0, steps = 0, taken = 0, emit_from = 0;
    uint64_t base = 0;
    unsigned v = 0;
    // extract: the walk yields the text from its end, one byte per LF step at descending addresses.  Byte stores reached the fabric as
    // 52 write requests per 64-byte snippet (profiles/walk_extract_r05_pmc.md: 27 x the bytes written); the bytes are collected in a
    // word — the byte for the lowest address arrives last, so shifting left as they come builds the little-endian word — and an
    // aligned group of eight leaves as ONE store.
    uint64_t acc = 0;
    unsigned acc_n = 0;
    auto emit = [&](unsigned c, bool last) {
        acc = (acc << 8) | c;
        ++acc_n;
        uint8_t * at = out_text + (base - taken);
        if (acc_n == 8 && (reinterpret_cast<uintptr_t>(at) & 7) == 0)
        {
            if (s == 0)
                *reinterpret_cast<uint64_t *>(at) = acc;
            acc_n = 0;
        }
        else if (last || (reinterpret_cast<uintptr_t>(at) & 7) == 0)
        { // the ragged ends of a snippet: fewer than eight bytes above an aligned address, or the snippet's first bytes
            if (s == 0)
                for (unsigned k = 0; k < acc_n; ++k)
                    at[k] = (uint8_t)(acc >> (8 * k));
            acc_n = 0;
        }
    };
    // "is SA index j sampled": a 64-bit division by a run-time value is some hundred instructions, paid after EVERY LF step of a csa[i]
    // walk — more than the step's own counting; the densities of real csa_wt types are powers of two (kernel-uniform test): a mask
    const bool sa_pow2 = (L.sa_dens & (L.sa_dens - 1)) == 0;
    const unsigned sa_shift = (unsigned)__builtin_ctzll(L.sa_dens | (UINT64_C(1) << 63));
    auto sa_sampled = [&](pos_t x) { return sa_pow2 ? (x & (pos_t)(L.sa_dens - 1)) == 0 : x % (pos_t)L.sa_dens == 0; };
    // between two LF steps: is the walk finished?  Then the answer is written and the walker is free again.
    auto settle = [&]() {
        bool done;
        const bool hit = MODE == kWalkSa && sa_sampled(j);
        if (MODE == kWalkSa) // LF is one cycle of length n in a consistent index: a longer walk means a broken one
            done = hit || taken > size;
        else
            done = taken == steps;
        if (!done)
        { // next LF step starts at the root
            v = 0;
            i = j;
            return;
        }
        if (MODE == kWalkSa)
        {
            uint64_t r = SDSL_HIP_NPOS;
            if (hit)
            {
                r = loc_sample(L.sa_s, L.s32, sa_pow2 ? j >> sa_shift : j / (pos_t)L.sa_dens) + taken; // (csa_wt.hpp:373-380)
                r = r < L.size ? r : r - L.size;
            }
            if (s == 0)
                out[q] = r;
        }
        else if (MODE != kWalkExtract)
        {
            if (s == 0)
                out[q] = j;
        }
        q += stride;
        busy = false;
    };
    // Every iteration of the outer loop is ONE tree level (one memory access) for every walker that has work: taking
    // the next query, finishing an LF step at a leaf and testing for the end of the walk all happen around it.
    for (;;)
    {
        while (!busy && q < n)
        { // next query of this walker
            const uint64_t a = a_next;
            if (q + stride < n)
                a_next = in0[q + stride];
            bool ok = a < L.size;
            taken = 0;
            if (MODE == kWalkSa || MODE == kWalkLf)
            {
                j = (pos_t)a; // (a < size where it is used)
                steps = 1;
            }
            else if (MODE == kWalkIsa)
            {
                if (ok)
                { // (suffix_array_helper.hpp:522-531)
                    uint64_t pos, j64;
                    isa_sample_right(L, a, j64, pos);
                    j = (pos_t)j64;
                    steps = (pos_t)(pos < a ? pos + L.size - a : pos - a);
                }
            }
            else
            { // extract text[a..e]: the walk from the sample at P > e yields text[P-1], text[P-2], ...
                const uint64_t e = in1[q];
                ok = ok && a <= e && e < L.size;
                if (ok)
                {
                    uint64_t pos, j64;
                    isa_sample_right(L, e, j64, pos);
                    j = (pos_t)j64;
                    steps = (pos_t)(pos <= e ? pos + L.size - e : pos - e); // P - e with P > e (position 0 taken as n)
                    emit_from = steps;                             // the steps-th character of the walk is text[e]
                    steps += (pos_t)(e - a);   // ... and text[a] is the last one
                    base = out_off[q] + (e - a) + emit_from; // text[P-k] goes to out_off[q] + (P-k-a) = base - k
                }
            }
            if (!ok)
            {
                if (MODE != kWalkExtract && s == 0)
                    out[q] = SDSL_HIP_NPOS;
                q += stride;
                continue;
            }
            busy = true;
            settle(); // a walk of zero steps ends here
        }
        if (!busy)
            break; // out of queries
        unsigned c;
        if (P::step(wt, &S, s, v, i, c))
        { // leaf: the LF step is complete (suffix_array_helper.hpp:352-358)
            j = (pos_t)S.F.C[S.F.char2comp[c]] + i;
            ++taken;
            if (MODE == kWalkExtract && taken >= emit_from)
                emit(c, taken == steps);
            settle();
        }
    }
}

// SA[i] from the whole suffix array
template <class SA>
__global__ __launch_bounds__(256) void k_fm_sa_full(const SA * __restrict__ sa, uint64_t size,
                                                    const uint64_t * __restrict__ idx, uint64_t n,
                                                    uint64_t * __restrict__ out)
{
    for (uint64_t q = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; q < n; q += (uint64_t)gridDim.x * blockDim.x)
    {
        const uint64_t i = idx[q];
        out[q] = i < size ? (uint64_t)sa[i] : SDSL_HIP_NPOS;
    }
}

// psi[i] = wt.select(i - C[cc] + 1, comp2char[cc]) with cc the symbol whose F-range holds i
// (suffix_array_helper.hpp:330-342, first_row_symbol :28-48); absent/invalid i select the 1st occurrence of an absent
// symbol, patched to NPOS afterwards
__global__ __launch_bounds__(256) void k_fm_psi_args(const FmTables * __restrict__ ftab, uint32_t sigma, uint64_t size,
                                                     const uint64_t * __restrict__ idx, uint64_t n,
                                                     uint64_t * __restrict__ sel_i, uint8_t * __restrict__ sel_c)
{
    __shared__ FmTables F;
    __shared__ uint8_t comp2char[256];
    fm_stage_tables(&F, ftab);
    __syncthreads();
    for (unsigned c = threadIdx.x; c < 256; c += blockDim.x)
        if (F.char2comp[c] || c == 0)
            comp2char[F.char2comp[c]] = (uint8_t)c;
    __syncthreads();
    for (uint64_t q = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; q < n; q += (uint64_t)gridDim.x * blockDim.x)
    {
        const uint64_t i = idx[q];
        if (i >= size)
        {
            sel_i[q] = 0; // wt.select answers NPOS for i == 0
            sel_c[q] = comp2char[0];
            continue;
        }
        unsigned lo = 0, hi = sigma; // largest cc with C[cc] <= i
        while (hi - lo > 1)
        {
            unsigned mid = (lo + hi) >> 1;
            if (F.C[mid] <= i)
                lo = mid;
            else
                hi = mid;
        }
        sel_i[q] = i - F.C[lo] + 1;
        sel_c[q] = comp2char[lo];
    }
}

// lengths of the answers of a ragged batch: extract -> end-begin+1, SA ranges -> r+1-l (0 for empty or invalid ones)
template <bool EXTRACT>
__global__ __launch_bounds__(256) void k_fm_lengths(const uint64_t * __restrict__ a, const uint64_t * __restrict__ b,
                                                    uint64_t size, uint64_t n, uint64_t * __restrict__ len)
{
    for (uint64_t q = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; q <= n; q += (uint64_t)gridDim.x * blockDim.x)
    {
        uint64_t v = 0;
        if (q < n)
        {
            const uint64_t x = a[q], y = b[q];
            if (x <= y && y < size)
                v = y - x + 1;
            (void)EXTRACT;
        }
        len[q] = v; // entry n is 0: the scan turns it into the total
    }
}

// extract in pieces: a range is cut at the ISA sample positions (multiples of d), so that every piece starts its walk
// at the sample right behind it and the pieces of one long range are walked in parallel.
// pieces of query q: floor(e/d) - floor(b/d) + 1 (0 for an invalid range); entry n is 0 (the scan turns it into the total)
__global__ __launch_bounds__(256) void k_fm_piece_count(const uint64_t * __restrict__ b, const uint64_t * __restrict__ e,
                                                        uint64_t size, uint64_t d, uint64_t n, uint64_t * __restrict__ cnt)
{
    for (uint64_t q = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; q <= n; q += (uint64_t)gridDim.x * blockDim.x)
    {
        uint64_t v = 0;
        if (q < n && b[q] <= e[q] && e[q] < size)
            v = e[q] / d - b[q] / d + 1;
        cnt[q] = v;
    }
}

// piece z belongs to the query q with poff[q] <= z < poff[q+1]; it covers [max(b, k*d), min(e, (k+1)*d - 1)] with
// k = floor(b/d) + (z - poff[q]) and writes at toff[q] + (its begin - b)
__global__ __launch_bounds__(256) void k_fm_piece_fill(const uint64_t * __restrict__ b, const uint64_t * __restrict__ e,
                                                       uint64_t d, const uint64_t * __restrict__ toff,
                                                       const uint64_t * __restrict__ poff, uint64_t n, uint64_t pieces,
                                                       uint64_t * __restrict__ pb, uint64_t * __restrict__ pe,
                                                       uint64_t * __restrict__ po)
{
    for (uint64_t z = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; z < pieces; z += (uint64_t)gridDim.x * blockDim.x)
    {
        uint64_t lo = 0, hi = n; // largest q with poff[q] <= z
        while (hi - lo > 1)
        {
            uint64_t mid = (lo + hi) >> 1;
            if (poff[mid] <= z)
                lo = mid;
            else
                hi = mid;
        }
        const uint64_t bq = b[lo], eq = e[lo];
        const uint64_t k = bq / d + (z - poff[lo]);
        const uint64_t first = k * d > bq ? k * d : bq;
        const uint64_t last = (k + 1) * d - 1 < eq ? (k + 1) * d - 1 : eq;
        pb[z] = first;
        pe[z] = last;
        po[z] = toff[lo] + (first - bq);
    }
}

// out[z] = l[p] + (z - off[p]) for the p with off[p] <= z < off[p+1]: the SA indices of all occurrences
__global__ __launch_bounds__(256) void k_fm_expand(const uint64_t * __restrict__ l, const uint64_t * __restrict__ off,
                                                   uint64_t n, uint64_t total, uint64_t * __restrict__ out)
{
    for (uint64_t z = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; z < total; z += (uint64_t)gridDim.x * blockDim.x)
    {
        uint64_t lo = 0, hi = n; // largest p with off[p] <= z
        while (hi - lo > 1)
        {
            uint64_t mid = (lo + hi) >> 1;
            if (off[mid] <= z)
                lo = mid;
            else
                hi = mid;
        }
        out[z] = l[lo] + (z - off[lo]);
    }
}

// extract on an index that still holds its text (created from text, before drop_sa / set_footprint): the answer is a copy.  A thread
// fills 16 consecutive output bytes; off[] says which ranges they belong to (a range's bytes are text[b .. e], position size - 1 being
// the sentinel's 0).  Invalid ranges have length 0 (k_fm_lengths) and are stepped over.
__global__ __launch_bounds__(256) void k_fm_extract_copy(const uint8_t * __restrict__ text, uint64_t n_text, const uint64_t * __restrict__ b,
                                                         const uint64_t * __restrict__ off, uint64_t n, uint64_t total,
                                                         uint8_t * __restrict__ out)
{
    for (uint64_t z0 = ((uint64_t)blockIdx.x * blockDim.x + threadIdx.x) * 16; z0 < total; z0 += (uint64_t)gridDim.x * blockDim.x * 16)
    {
        uint64_t lo = 0, hi = n; // largest p with off[p] <= z0
        while (hi - lo > 1)
        {
            const uint64_t mid = (lo + hi) >> 1;
            if (off[mid] <= z0)
                lo = mid;
            else
                hi = mid;
        }
        uint64_t p = lo, o_lo = off[p], o_hi = off[p + 1], src = b[p];
        const uint64_t z1 = z0 + 16 < total ? z0 + 16 : total;
        for (uint64_t z = z0; z < z1; ++z)
        {
            while (z >= o_hi)
            { // next range (empty ones have o_lo == o_hi)
                ++p;
                o_lo = o_hi;
                o_hi = off[p + 1];
                src = b[p];
            }
            const uint64_t at = src + (z - o_lo);
            out[z] = at < n_text ? text[at] : (uint8_t)0;
        }
    }
}

} // namespace sdslhip

using namespace sdslhip;

static FmLocView loc_view(const sdsl_hip_fm_s * f)
{
    FmLocView L;
    L.sa_full = f->d_sa.as<uint32_t>();
    L.sa_s = f->d_sa_s.as<uint64_t>();
    L.isa_s = f->d_isa_s.as<uint64_t>();
    L.sa_dens = f->sa_dens;
    L.isa_dens = f->isa_dens;
    L.n_isa_s = f->n_isa_s;
    L.size = f->size;
    L.s32 = f->samples32 ? 1u : 0u;
    return L;
}

// ISA samples for an index that still has its whole suffix array but no samples yet (SDSL's default density)
static sdsl_hip_status ensure_isa_samples(sdsl_hip_fm_s * f)
{
    if (f->isa_dens)
        return SDSL_HIP_OK;
    if (!f->d_sa.p)
    {
        set_error("this FM-index has no ISA samples (loaded without densities, or created from a BWT)");
        return SDSL_HIP_ERR_UNSUPPORTED;
    }
    SH_TRY(sa_samples_device(f->d_sa.as<uint32_t>(), f->size, 32, 64, nullptr, &f->d_isa_s));
    f->isa_dens = 64;
    f->n_isa_s = (f->size + 63) / 64;
    return SDSL_HIP_OK;
}

template <int MODE>
static sdsl_hip_status launch_walk(sdsl_hip_fm_s * f, const uint64_t * d_in0, const uint64_t * d_in1,
                                   const uint64_t * d_off, uint64_t n, uint64_t * d_out, uint8_t * d_text, hipStream_t s)
{
    if (n == 0)
        return SDSL_HIP_OK;
    const WtHost & w = sdsl_hip_wt_host(f->wt);
    const FmLocView L = loc_view(f);
    KernelTimer t(s);
    if (w.backend == 1)
        hipLaunchKernelGGL((k_fm_walk<LocRrr, MODE, true>), dim3(grid_for(n, LocRrr::kThreads, 256u * 3u)),
                           dim3(LocRrr::kThreads), 0, s, w.view(), f->d_tab.as<FmTables>(), L, d_in0, d_in1, d_off, n,
                           d_out, d_text);
    else if (w.d_fwalk.p && w.d_fused.p && (f->size >> 32))
        hipLaunchKernelGGL((k_fm_walk<LocFused, MODE, true>), dim3(grid_for(n, kQPB, 256u * 8u)), dim3(LocFused::kThreads), 0, s,
                           w.view(), f->d_tab.as<FmTables>(), L, d_in0, d_in1, d_off, n, d_out, d_text);
    else if (w.d_fwalk.p && w.d_fused.p)
        hipLaunchKernelGGL((k_fm_walk<LocFused, MODE, false>), dim3(grid_for(n, kQPB, 256u * 8u)), dim3(LocFused::kThreads), 0, s,
                           w.view(), f->d_tab.as<FmTables>(), L, d_in0, d_in1, d_off, n, d_out, d_text);
    else if (f->size >> 32)
        hipLaunchKernelGGL((k_fm_walk<LocPlain, MODE, true>), dim3(grid_for(n, kQPB, 256u * 8u)), dim3(LocPlain::kThreads), 0, s,
                           w.view(), f->d_tab.as<FmTables>(), L, d_in0, d_in1, d_off, n, d_out, d_text);
    else
        hipLaunchKernelGGL((k_fm_walk<LocPlain, MODE, false>), dim3(grid_for(n, kQPB, 256u * 8u)), dim3(LocPlain::kThreads), 0, s,
                           w.view(), f->d_tab.as<FmTables>(), L, d_in0, d_in1, d_off, n, d_out, d_text);
    SH_HIP(hipGetLastError());
    return SDSL_HIP_OK;
}

// SA values for device-resident indices (in place is fine: every walker reads its index before it writes)
static sdsl_hip_status sa_lookup(sdsl_hip_fm_s * f, const uint64_t * d_idx, uint64_t n, uint64_t * d_out, hipStream_t s)
{
    if (n == 0)
        return SDSL_HIP_OK;
    if (f->d_sa.p)
    {
        KernelTimer t(s);
        hipLaunchKernelGGL(k_fm_sa_full<uint32_t>, dim3(grid_for(n, 256, 256u * 16u)), dim3(256), 0, s, f->d_sa.as<uint32_t>(),
                           f->size, d_idx, n, d_out);
        SH_HIP(hipGetLastError());
        return SDSL_HIP_OK;
    }
    if (f->d_sa64.p)
    {
        KernelTimer t(s);
        hipLaunchKernelGGL(k_fm_sa_full<uint64_t>, dim3(grid_for(n, 256, 256u * 16u)), dim3(256), 0, s, f->d_sa64.as<uint64_t>(),
                           f->size, d_idx, n, d_out);
        SH_HIP(hipGetLastError());
        return SDSL_HIP_OK;
    }
    if (!f->sa_dens)
    {
        set_error("this FM-index has no SA samples (loaded without densities, or created from a BWT)");
        return SDSL_HIP_ERR_UNSUPPORTED;
    }
    return launch_walk<kWalkSa>(f, d_idx, nullptr, nullptr, n, d_out, nullptr, s);
}

extern "C" {

sdsl_hip_status sdsl_hip_fm_sampling(sdsl_hip_fm_t fm, uint32_t * sa_dens, uint32_t * isa_dens, int32_t * has_full_sa)
{
    if (!fm)
    {
        set_error("fm_sampling: null handle");
        return SDSL_HIP_ERR_INVALID;
    }
    if (sa_dens)
        *sa_dens = fm->sa_dens;
    if (isa_dens)
        *isa_dens = fm->isa_dens;
    if (has_full_sa)
        *has_full_sa = fm->d_sa.p || fm->d_sa64.p ? 1 : 0;
    return SDSL_HIP_OK;
}

static sdsl_hip_status simple_batch(sdsl_hip_fm_t fm, const uint64_t * idx, uint64_t n, uint64_t * out, void * stream,
                                    int what, const char * name)
{
    if (!fm || (n && (!idx || !out)))
    {
        set_error("%s: invalid argument", name);
        return SDSL_HIP_ERR_INVALID;
    }
    hipStream_t s = (hipStream_t)stream;
    SH_HIP(hipSetDevice(fm->device));
    if (n == 0)
        return SDSL_HIP_OK;
    if (what == kWalkIsa)
        SH_TRY(ensure_isa_samples(fm));
    Staged si, so;
    SH_TRY(si.in(idx, n * 8, s));
    SH_TRY(so.out(out, n * 8));
    if (what == kWalkSa)
        SH_TRY(sa_lookup(fm, (const uint64_t *)si.dev, n, (uint64_t *)so.dev, s));
    else if (what == kWalkIsa)
        SH_TRY(launch_walk<kWalkIsa>(fm, (const uint64_t *)si.dev, nullptr, nullptr, n, (uint64_t *)so.dev, nullptr, s));
    else
        SH_TRY(launch_walk<kWalkLf>(fm, (const uint64_t *)si.dev, nullptr, nullptr, n, (uint64_t *)so.dev, nullptr, s));
    SH_TRY(so.finish(s));
    if (si.host && !so.host)
        SH_HIP(hipStreamSynchronize(s));
    return SDSL_HIP_OK;
}

sdsl_hip_status sdsl_hip_fm_sa_batch(sdsl_hip_fm_t fm, const uint64_t * idx, uint64_t n, uint64_t * out, void * stream)
{
    return simple_batch(fm, idx, n, out, stream, kWalkSa, "fm_sa_batch");
}

sdsl_hip_status sdsl_hip_fm_isa_batch(sdsl_hip_fm_t fm, const uint64_t * idx, uint64_t n, uint64_t * out, void * stream)
{
    return simple_batch(fm, idx, n, out, stream, kWalkIsa, "fm_isa_batch");
}

sdsl_hip_status sdsl_hip_fm_lf_batch(sdsl_hip_fm_t fm, const uint64_t * idx, uint64_t n, uint64_t * out, void * stream)
{
    return simple_batch(fm, idx, n, out, stream, kWalkLf, "fm_lf_batch");
}

sdsl_hip_status sdsl_hip_fm_psi_batch(sdsl_hip_fm_t fm, const uint64_t * idx, uint64_t n, uint64_t * out, void * stream)
{
    if (!fm || (n && (!idx || !out)))
    {
        set_error("fm_psi_batch: invalid argument");
        return SDSL_HIP_ERR_INVALID;
    }
    hipStream_t s = (hipStream_t)stream;
    SH_HIP(hipSetDevice(fm->device));
    if (n == 0)
        return SDSL_HIP_OK;
    Staged si, so;
    SH_TRY(si.in(idx, n * 8, s));
    SH_TRY(so.out(out, n * 8));
    DevBuf d_i, d_c;
    SH_TRY(d_i.alloc(n * 8));
    SH_TRY(d_c.alloc(n));
    hipLaunchKernelGGL(k_fm_psi_args, dim3(grid_for(n, 256, 256u * 8u)), dim3(256), 0, s, fm->d_tab.as<FmTables>(),
                       fm->sigma, fm->size, (const uint64_t *)si.dev, n, d_i.as<uint64_t>(), d_c.as<uint8_t>());
    SH_HIP(hipGetLastError());
    SH_TRY(sdsl_hip_wt_select_batch(fm->wt, d_i.as<uint64_t>(), d_c.as<uint8_t>(), n, (uint64_t *)so.dev, s));
    SH_TRY(so.finish(s));
    SH_HIP(hipStreamSynchronize(s)); // the scratch buffers die with this frame
    return SDSL_HIP_OK;
}

sdsl_hip_status sdsl_hip_fm_extract_batch(sdsl_hip_fm_t fm, const uint64_t * begin, const uint64_t * end, uint64_t n,
                                          uint64_t * out_offsets, uint8_t * out_text, uint64_t cap, uint64_t * total,
                                          void * stream)
{
    if (!fm || !total || (n && (!begin || !end)))
    {
        set_error("fm_extract_batch: invalid argument");
        return SDSL_HIP_ERR_INVALID;
    }
    hipStream_t s = (hipStream_t)stream;
    SH_HIP(hipSetDevice(fm->device));
    *total = 0;
    if (!fm->d_text.p)
        SH_TRY(ensure_isa_samples(fm)); // (with the text resident the answer is a copy: no walk, no samples)
    Staged sb, se;
    SH_TRY(sb.in(begin, n * 8, s));
    SH_TRY(se.in(end, n * 8, s));
    // ONE allocation for the call's four (n + 1)-entry arrays and the scans' working memory (seven hipMalloc / hipFree pairs and two
    // synchronising scans used to cost 6 ms of a 39 ms call)
    const size_t col = ((size_t)(n + 1) * 8 + 255) & ~(size_t)255, scan_tmp = (exclusive_scan_u64_temp_bytes(n + 1) + 255) & ~(size_t)255;
    DevBuf work;
    SH_TRY(work.alloc(4 * col + scan_tmp));
    uint64_t * d_len = work.as<uint64_t>();
    uint64_t * d_off = reinterpret_cast<uint64_t *>(work.as<uint8_t>() + col);
    uint64_t * d_cnt = reinterpret_cast<uint64_t *>(work.as<uint8_t>() + 2 * col);
    uint64_t * d_poff = reinterpret_cast<uint64_t *>(work.as<uint8_t>() + 3 * col);
    void * d_scan = work.as<uint8_t>() + 4 * col;
    hipLaunchKernelGGL((k_fm_lengths<true>), dim3(grid_for(n + 1, 256, 256u * 8u)), dim3(256), 0, s,
                       (const uint64_t *)sb.dev, (const uint64_t *)se.dev, fm->size, n, d_len);
    SH_HIP(hipGetLastError());
    SH_TRY(exclusive_scan_u64(d_len, d_off, n + 1, s, d_scan, scan_tmp));
    SH_HIP(hipMemcpyAsync(total, d_off + n, 8, hipMemcpyDeviceToHost, s));
    SH_HIP(hipStreamSynchronize(s));
    if (out_offsets)
    {
        Staged so;
        SH_TRY(so.out(out_offsets, (n + 1) * 8));
        SH_HIP(hipMemcpyAsync(so.dev, d_off, (n + 1) * 8, hipMemcpyDeviceToDevice, s));
        SH_TRY(so.finish(s));
        SH_HIP(hipStreamSynchronize(s));
    }
    if (!out_text)
        return SDSL_HIP_OK; // size query
    if (cap < *total)
    {
        set_error("fm_extract_batch: the output needs %llu bytes, %llu given", (unsigned long long)*total,
                  (unsigned long long)cap);
        return SDSL_HIP_ERR_INVALID;
    }
    if (*total == 0)
        return SDSL_HIP_OK;
    Staged st;
    SH_TRY(st.out(out_text, *total));
    if (fm->d_text.p)
    { // the text is resident (an index created from text that has not given it back): a copy instead of LF walks
        {
            KernelTimer t(s);
            hipLaunchKernelGGL(k_fm_extract_copy, dim3(grid_for((*total + 15) / 16, 256, 256u * 16u)), dim3(256), 0, s, fm->d_text.as<uint8_t>(),
                               fm->size - 1, (const uint64_t *)sb.dev, d_off, n, *total, (uint8_t *)st.dev);
        }
        SH_HIP(hipGetLastError());
        SH_TRY(st.finish(s));
        SH_HIP(hipStreamSynchronize(s));
        return SDSL_HIP_OK;
    }
    // cut the ranges at the ISA sample positions: one walk per piece
    const uint64_t d = fm->isa_dens;
    hipLaunchKernelGGL(k_fm_piece_count, dim3(grid_for(n + 1, 256, 256u * 8u)), dim3(256), 0, s, (const uint64_t *)sb.dev,
                       (const uint64_t *)se.dev, fm->size, d, n, d_cnt);
    SH_HIP(hipGetLastError());
    SH_TRY(exclusive_scan_u64(d_cnt, d_poff, n + 1, s, d_scan, scan_tmp));
    uint64_t pieces = 0;
    SH_HIP(hipMemcpyAsync(&pieces, d_poff + n, 8, hipMemcpyDeviceToHost, s));
    SH_HIP(hipStreamSynchronize(s));
    DevBuf d_pieces; // begin, end and output offset of every piece: one allocation
    SH_TRY(d_pieces.alloc(std::max<uint64_t>(pieces, 1) * 24));
    uint64_t *d_pb = d_pieces.as<uint64_t>(), *d_pe = d_pb + pieces, *d_po = d_pe + pieces;
    hipLaunchKernelGGL(k_fm_piece_fill, dim3(grid_for(pieces, 256, 256u * 16u)), dim3(256), 0, s, (const uint64_t *)sb.dev,
                       (const uint64_t *)se.dev, d, d_off, d_poff, n, pieces, d_pb, d_pe, d_po);
    SH_HIP(hipGetLastError());
    SH_TRY(launch_walk<kWalkExtract>(fm, d_pb, d_pe, d_po, pieces, nullptr, (uint8_t *)st.dev, s));
    SH_TRY(st.finish(s));
    SH_HIP(hipStreamSynchronize(s));
    return SDSL_HIP_OK;
}

sdsl_hip_status sdsl_hip_fm_sa_range_batch(sdsl_hip_fm_t fm, const uint64_t * l, const uint64_t * r, uint64_t n,
                                           uint64_t * out_offsets, uint64_t * out_pos, uint64_t cap, uint64_t * total,
                                           void * stream)
{
    if (!fm || !total || (n && (!l || !r)))
    {
        set_error("fm_sa_range_batch: invalid argument");
        return SDSL_HIP_ERR_INVALID;
    }
    hipStream_t s = (hipStream_t)stream;
    SH_HIP(hipSetDevice(fm->device));
    *total = 0;
    Staged sl, sr;
    SH_TRY(sl.in(l, n * 8, s));
    SH_TRY(sr.in(r, n * 8, s));
    const size_t col = ((size_t)(n + 1) * 8 + 255) & ~(size_t)255, scan_tmp = (exclusive_scan_u64_temp_bytes(n + 1) + 255) & ~(size_t)255;
    DevBuf work; // lengths, offsets and the scan's working memory: one allocation, no synchronising scan
    SH_TRY(work.alloc(2 * col + scan_tmp));
    uint64_t * d_len = work.as<uint64_t>();
    uint64_t * d_off = reinterpret_cast<uint64_t *>(work.as<uint8_t>() + col);
    hipLaunchKernelGGL((k_fm_lengths<false>), dim3(grid_for(n + 1, 256, 256u * 8u)), dim3(256), 0, s,
                       (const uint64_t *)sl.dev, (const uint64_t *)sr.dev, fm->size, n, d_len);
    SH_HIP(hipGetLastError());
    SH_TRY(exclusive_scan_u64(d_len, d_off, n + 1, s, work.as<uint8_t>() + 2 * col, scan_tmp));
    SH_HIP(hipMemcpyAsync(total, d_off + n, 8, hipMemcpyDeviceToHost, s));
    SH_HIP(hipStreamSynchronize(s));
    if (out_offsets)
    {
        Staged so;
        SH_TRY(so.out(out_offsets, (n + 1) * 8));
        SH_HIP(hipMemcpyAsync(so.dev, d_off, (n + 1) * 8, hipMemcpyDeviceToDevice, s));
        SH_TRY(so.finish(s));
        SH_HIP(hipStreamSynchronize(s));
    }
    if (!out_pos)
        return SDSL_HIP_OK; // size query
    if (cap < *total)
    {
        set_error("fm_sa_range_batch: the output needs %llu entries, %llu given", (unsigned long long)*total,
                  (unsigned long long)cap);
        return SDSL_HIP_ERR_INVALID;
    }
    if (*total == 0)
        return SDSL_HIP_OK;
    Staged sp;
    SH_TRY(sp.out(out_pos, *total * 8));
    hipLaunchKernelGGL(k_fm_expand, dim3(grid_for(*total, 256, 256u * 16u)), dim3(256), 0, s, (const uint64_t *)sl.dev,
                       d_off, n, *total, (uint64_t *)sp.dev);
    SH_HIP(hipGetLastError());
    SH_TRY(sa_lookup(fm, (const uint64_t *)sp.dev, *total, (uint64_t *)sp.dev, s));
    SH_TRY(sp.finish(s));
    SH_HIP(hipStreamSynchronize(s));
    return SDSL_HIP_OK;
}

sdsl_hip_status sdsl_hip_fm_locate_batch(sdsl_hip_fm_t fm, const uint8_t * patterns, uint32_t m, uint64_t n_pat,
                                         uint64_t * out_offsets, uint64_t * out_pos, uint64_t cap, uint64_t * total,
                                         void * stream)
{
    if (!fm || !total || (n_pat && !patterns))
    {
        set_error("fm_locate_batch: invalid argument");
        return SDSL_HIP_ERR_INVALID;
    }
    hipStream_t s = (hipStream_t)stream;
    SH_HIP(hipSetDevice(fm->device));
    DevBuf d_l, d_r;
    SH_TRY(d_l.alloc((n_pat + 1) * 8));
    SH_TRY(d_r.alloc((n_pat + 1) * 8));
    Staged sp;
    SH_TRY(sp.in(patterns, n_pat * (uint64_t)m, s));
    SH_TRY(sdsl_hip_fm_interval_batch(fm, (const uint8_t *)sp.dev, m, n_pat, d_l.as<uint64_t>(), d_r.as<uint64_t>(), s));
    return sdsl_hip_fm_sa_range_batch(fm, d_l.as<uint64_t>(), d_r.as<uint64_t>(), n_pat, out_offsets, out_pos, cap, total,
                                      s);
}
}
