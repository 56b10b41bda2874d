// fm.hip — csa_wt<wt_huff<bit_vector, rank_support_v5<>>> restricted to the count() path:
// byte alphabet (C, char2comp), batched backward_search steps and whole-pattern count.
//
// Reference semantics reproduced:
//   backward_search(csa,l,r,c,..)          suffix_array_algorithm.hpp:167-201
//   backward_search(csa,l,r,begin,end,..)  suffix_array_algorithm.hpp:228-248
//   count(csa,begin,end)                   suffix_array_algorithm.hpp:464-471
//   byte_alphabet                          csa_alphabet_strategy.hpp:175-212
//   csa_wt::rank_bwt                       csa_wt.hpp:286-289
#include "fm_host.hpp"

namespace sdslhip {

// Sort key of a pattern: its last eight bytes, the LAST byte most significant — backward search consumes a pattern
// from its end, so patterns that are neighbours in this order walk the same SA intervals (the same rank lines) for
// their first eight LF steps.
__global__ __launch_bounds__(256) void k_fm_keys(const uint8_t * __restrict__ pats, uint32_t m,
                                                 const uint64_t * __restrict__ offsets, uint64_t n_pat,
                                                 uint64_t * __restrict__ keys, uint32_t * __restrict__ idx)
{
    for (uint64_t p = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; p < n_pat; p += (uint64_t)gridDim.x * blockDim.x)
    {
        uint64_t begin = offsets ? offsets[p] : p * (uint64_t)m;
        uint64_t end = offsets ? offsets[p + 1] : begin + m;
        uint64_t key = 0;
#pragma unroll
        for (int t = 0; t < 8; ++t)
            key = (key << 8) | (end > begin + (uint64_t)t ? (uint64_t)pats[end - 1 - t] : 0);
        keys[p] = key;
        idx[p] = (uint32_t)p;
    }
}

// One pattern per quad.  The two rank cascades of every LF step run level by level with both line fetches in
// flight (wt_device.hpp: quad_wt_rank2_level).  The loop is FLAT — one iteration is one tree level of whatever
// character the quad is at — so the 16 quads of a wave do not wait for each other at character boundaries (Huffman
// paths differ in length; a nested loop would cost max(len) instead of len per character).  The next pattern byte is
// fetched one character ahead.
static bool fm_verify_enabled()
{ // SDSL_HIP_FM_VERIFY=0: count() walks every character of every pattern (the round-2 behaviour, for A/B)
    static const bool on = !(getenv("SDSL_HIP_FM_VERIFY") && atoi(getenv("SDSL_HIP_FM_VERIFY")) == 0);
    return on;
}

// VERIFY (count only, index created from a text that is still resident with its whole suffix array): once the interval is ONE
// suffix the rest of the pattern can only occur right in front of that suffix in the text — SA[l] says where, and the remaining
// characters are compared with the text there: two fetches instead of one or two per remaining character (the late steps are
// the random ones: 25.6 fabric requests per 20-byte pattern without it).  The answer is the same number: 1 if the characters
// match, else 0 — what the remaining LF steps would have found.
template <bool NT, bool WANT_IVAL, bool FUSED, bool VERIFY = false>
__global__ __launch_bounds__(kBlock) void k_fm_count(WtView wt, const FmTables * __restrict__ ftab, FmJump J,
                                                     uint64_t csa_size, const uint8_t * __restrict__ pats,
                                                     uint32_t m, const uint64_t * __restrict__ offsets,
                                                     const uint32_t * __restrict__ order, uint64_t n_pat,
                                                     uint64_t * __restrict__ out_cnt, uint64_t * __restrict__ out_l,
                                                     uint64_t * __restrict__ out_r, const uint32_t * __restrict__ sa = nullptr,
                                                     const uint8_t * __restrict__ text = nullptr)
{
    __shared__ WtTables T;
    __shared__ FmTables F;
    fm_stage_tables(&F, ftab);
    wt_stage_tables(&T, wt.tables); // ends with __syncthreads()
    __shared__ WtFusedTables FT;
    if (FUSED)
        wt_stage_fused(&FT, wt);
    const int s = threadIdx.x & (kG - 1);
    const unsigned gq = threadIdx.x / kG;
    for (uint64_t base = (uint64_t)blockIdx.x * kQPB; base < n_pat; base += (uint64_t)gridDim.x * kQPB)
    {
        uint64_t q = base + gq;
        if (q >= n_pat)
            continue;
        if (order)
            q = order[q]; // batch ordered by pattern suffix: neighbouring quads share lines; results go back in place
        uint64_t begin = offsets ? offsets[q] : q * (uint64_t)m;
        uint64_t end = offsets ? offsets[q + 1] : begin + m;
        uint64_t l = 0, r = csa_size - 1;
        if (!WANT_IVAL && end - begin > csa_size)
        { // count(): a pattern longer than the text cannot occur (:466-467)
            l = 1;
            r = 0;
            end = begin;
        }
        uint64_t it = fm_jump_start(J, F, pats, begin, end, l, r);
        unsigned c_next = it > begin ? pats[it - 1] : 0;
        // state of the character being processed
        unsigned left = 0, v = 0;
        uint64_t a = 0, b = 0, p = 0, cb = 0;
        bool pending = false; // VERIFY: the search stopped at a few suffixes with characters left (see k_fm_verify)
        uint64_t prev_size = 0;
        for (;;)
        {
            if (left == 0)
            { // next character (suffix_array_algorithm.hpp:176-200)
                if (!(it > begin && r + 1 - l > 0))
                    break;
                const bool stable = r + 1 - l == prev_size; // (fm_count2.hip, verify_pays: s > 1 only for an interval that has stopped shrinking)
                prev_size = r + 1 - l;
                if (VERIFY && !WANT_IVAL && r - l < kFmVerifyMax && (l == r || stable) && it > begin + 1 && it < end && it - begin > r - l &&
                    it - begin < (csa_size >> 32 ? (UINT64_C(1) << 20) : (UINT64_C(1) << 28)) && it - begin <= 16 &&
                    !fm_tail_has_zero(load_tail16(pats, it), (uint32_t)(it - begin)))
                { // a few suffixes left and at least that many (two) characters to go: k_fm_verify compares them with the text at each
                    pending = true;
                    break;
                }
                --it;
                const unsigned c = c_next;
                if (it > begin)
                    c_next = pats[it - 1];
                const unsigned cc = F.char2comp[c];
                if (cc == 0 && c > 0)
                { // character does not occur (:180-184)
                    l = 1;
                    r = 0;
                    continue;
                }
                cb = F.C[cc];
                if (l == 0 && r + 1 == csa_size)
                { // whole interval: no rank needed (:188-192)
                    l = cb;
                    r = F.C[cc + 1] - 1;
                    continue;
                }
                a = l;
                b = r + 1;
                if (wt.sigma == 1)
                { // one symbol: rank(i, c) == i
                    l = cb + a;
                    r = cb + b - 1;
                    continue;
                }
                p = T.path[c]; // the symbol occurs (char2comp said so), so it has a leaf and a path
                left = (unsigned)(p >> 56);
                v = 0;
            }
            if (FUSED) // one iteration = up to three levels (wt_device.hpp: fused layout)
                quad_wt8_rank2_step<NT>(wt, &T, &FT, s, v, p, left, a, b);
            else
            {
                quad_wt_rank2_level<NT>(wt, &T, s, v, (unsigned)(p & 1), a, b);
                p >>= 1;
                --left;
            }
            if (b == 0)
            { // a <= b: both chains are 0 from here on (wt_pc.hpp:386)
                a = 0;
                left = 0;
            }
            if (left == 0)
            {
                l = cb + a;
                r = cb + b - 1;
            }
        }
        if (s == 0)
        {
            if (WANT_IVAL)
            {
                out_l[q] = l;
                out_r[q] = r;
            }
            else // (pending: fm_device.hpp, fm_pending_word)
                out_cnt[q] = VERIFY && pending ? fm_pending_word(csa_size >> 32 ? 40 : 32, l, r + 1 - l, it - begin) : r + 1 - l;
        }
    }
}

// count() of the patterns whose search stopped at a few suffixes: suffix i of the interval stands at SA[i] in the text, so the pattern's
// remaining characters pats[begin .. begin + rem) stand right in front of it or not; the count is the number of suffixes where they do.
// One lane per pattern.
template <class SA>
__global__ __launch_bounds__(256) void k_fm_verify(const SA * __restrict__ sa, const uint8_t * __restrict__ text,
                                                   const uint8_t * __restrict__ pats, uint32_t m, const uint64_t * __restrict__ offsets,
                                                   uint64_t n_pat, uint64_t * __restrict__ out_cnt, uint64_t csa_size)
{
    const unsigned LB = csa_size >> 32 ? 40 : 32; // (the packing of the pending word: fm_device.hpp)
    for (uint64_t q = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; q < n_pat; q += (uint64_t)gridDim.x * blockDim.x)
    {
        const uint64_t v = out_cnt[q];
        if (!(v >> 63))
            continue;
        const uint64_t l = v & ((UINT64_C(1) << LB) - 1);
        const uint32_t rem = (uint32_t)((v >> LB) & ((UINT64_C(1) << (60 - LB)) - 1));
        const uint32_t ns = (uint32_t)(v >> 60) & 7u; // suffixes - 1
        const uint64_t begin = offsets ? offsets[q] : q * (uint64_t)m;
        const uint8_t * p = pats + begin;
        uint64_t at[kFmVerifyMax];
#pragma unroll
        for (uint32_t j = 0; j < kFmVerifyMax; ++j)
            at[j] = j <= ns ? (uint64_t)sa[l + j] : 0;
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
                for (uint32_t k = 0; k < rem && ok; ++k)
                    ok = t[k] == p[k];
            }
            cnt += ok ? 1u : 0u;
        }
        out_cnt[q] = cnt;
    }
}

// all k-mers over the compact alphabet as patterns: pattern `idx` spells the key digits of fm_jump_start (first digit =
// the LAST character)
__global__ __launch_bounds__(256) void k_fm_kmers(const FmTables * __restrict__ ftab, uint32_t sigma, uint32_t k,
                                                  uint64_t count, uint8_t * __restrict__ pats)
{
    __shared__ uint8_t comp2char[256];
    for (unsigned c = threadIdx.x; c < 256; c += blockDim.x)
        if (ftab->char2comp[c] || c == 0)
            comp2char[ftab->char2comp[c]] = (uint8_t)c;
    __syncthreads();
    for (uint64_t idx = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; idx < count; idx += (uint64_t)gridDim.x * blockDim.x)
    {
        uint64_t key = idx;
        for (uint32_t t = 0; t < k; ++t)
        { // least significant digit = the character processed last = the FIRST byte of the k-mer
            pats[idx * k + t] = comp2char[key % sigma];
            key /= sigma;
        }
    }
}

__global__ __launch_bounds__(256) void k_fm_pack_jump(const uint64_t * __restrict__ l, const uint64_t * __restrict__ r,
                                                      uint64_t count, uint64_t * __restrict__ tab)
{
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < count; i += (uint64_t)gridDim.x * blockDim.x)
    {
        tab[2 * i] = l[i];
        tab[2 * i + 1] = r[i];
    }
}

// one LF step per element
template <bool NT>
__global__ __launch_bounds__(kBlock) void k_fm_backward_step(WtView wt, const FmTables * __restrict__ ftab,
                                                             uint64_t csa_size, const uint64_t * __restrict__ lq,
                                                             const uint64_t * __restrict__ rq,
                                                             const uint8_t * __restrict__ cq, uint64_t n,
                                                             uint64_t * __restrict__ out_l,
                                                             uint64_t * __restrict__ out_r)
{
    __shared__ WtTables T;
    __shared__ WtFusedTables FT;
    __shared__ FmTables F;
    fm_stage_tables(&F, ftab);
    wt_stage_tables(&T, wt.tables);
    wt_stage_fused(&FT, wt);
    const int s = threadIdx.x & (kG - 1);
    const unsigned gq = threadIdx.x / kG;
    for (uint64_t base = (uint64_t)blockIdx.x * kQPB; base < n; base += (uint64_t)gridDim.x * kQPB)
    {
        uint64_t q = base + gq;
        if (q >= n)
            continue;
        uint64_t l = lq[q], r = rq[q];
        unsigned c = cq[q];
        uint64_t lo, ro;
        if (!(l <= r && r < csa_size))
        { // outside SDSL's precondition (asserts :177-178)
            lo = ro = SDSL_HIP_NPOS;
        }
        else
        {
            unsigned cc = F.char2comp[c];
            if (cc == 0 && c > 0)
            {
                lo = 1;
                ro = 0;
            }
            else
            {
                uint64_t cb = F.C[cc];
                if (l == 0 && r + 1 == csa_size)
                {
                    lo = cb;
                    ro = F.C[cc + 1] - 1;
                }
                else
                {
                    uint64_t a = l, b = r + 1;
                    quad_wt_rank2<NT>(wt, &T, &FT, s, c, a, b);
                    lo = cb + a;
                    ro = cb + b - 1;
                }
            }
        }
        if (s == 0)
        {
            out_l[q] = lo;
            out_r[q] = ro;
        }
    }
}

} // namespace sdslhip

using namespace sdslhip;

static void fm_free(sdsl_hip_fm_s * f)
{
    if (!f)
        return;
    if (f->wt)
        sdsl_hip_wt_destroy(f->wt);
    delete f;
}

// byte_alphabet from the symbol histogram (csa_alphabet_strategy.hpp:175-212)
static void alphabet_from_counts(sdsl_hip_fm_s * f, const uint64_t occ[256])
{
    memset(&f->tab, 0, sizeof f->tab);
    uint32_t sigma = 0;
    uint64_t run = 0;
    for (int c = 0; c < 256; ++c)
        if (occ[c])
        {
            f->tab.char2comp[c] = (uint8_t)sigma;
            f->tab.C[sigma] = run;
            run += occ[c];
            ++sigma;
        }
    f->tab.C[sigma] = run;
    f->sigma = sigma;
}

static sdsl_hip_status fm_upload_tables(sdsl_hip_fm_s * f)
{
    SH_TRY(f->d_tab.alloc(sizeof(FmTables)));
    SH_HIP(hipMemcpy(f->d_tab.p, &f->tab, sizeof(FmTables), hipMemcpyHostToDevice));
    return SDSL_HIP_OK;
}

// position of the first 0 byte of a device-resident text (the sentinel must be unique, construct.hpp:41)
__global__ __launch_bounds__(256) void k_first_zero_byte(const uint8_t * __restrict__ text, uint64_t n,
                                                         unsigned long long * __restrict__ first)
{
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (uint64_t)gridDim.x * blockDim.x)
        if (text[i] == 0)
            atomicMin(first, (unsigned long long)i);
}

static sdsl_hip_status fm_build_jump(sdsl_hip_fm_s * f);
static sdsl_hip_status fm_build_jump_k(sdsl_hip_fm_s * f, uint32_t k);

static sdsl_hip_status fm_from_device_bwt(sdsl_hip_fm_s * f, const uint8_t * d_bwt, uint64_t n, int device,
                                          uint32_t flags)
{
    f->device = device;
    f->size = n;
    f->wt = sdsl_hip_wt_alloc();
    if (!f->wt)
        return SDSL_HIP_ERR_NOMEM;
    WtHost & w = sdsl_hip_wt_host(f->wt);
    const bool trace = getenv("SDSL_HIP_TRACE_BUILD") != nullptr;
    auto stamp = [&](const char * what, std::chrono::steady_clock::time_point & t) {
        if (!trace)
            return;
        (void)hipDeviceSynchronize();
        const auto now = std::chrono::steady_clock::now();
        fprintf(stderr, "[sdsl_hip] fm build: %s %.1f ms\n", what, std::chrono::duration<double, std::milli>(now - t).count());
        t = now;
    };
    auto t = std::chrono::steady_clock::now();
    SH_TRY(wt_build_from_device_text(w, d_bwt, n, device, flags)); // csa_wt.hpp:337-343: the WT is built over the BWT
    stamp("wavelet tree levels", t);
    SH_TRY(sdsl_hip_wt_finish(f->wt));
    stamp("fused layout + occ", t);
    if (n == 0 || w.occ[0] != 1)
    {
        set_error("the BWT must contain exactly one 0 byte (the sentinel appended by construct.hpp:100-108); found %llu",
                  (unsigned long long)(n ? w.occ[0] : 0));
        return SDSL_HIP_ERR_INVALID;
    }
    alphabet_from_counts(f, w.occ);
    SH_TRY(fm_upload_tables(f));
    sdsl_hip_status st = fm_build_jump(f);
    stamp("k-mer interval table", t);
    if (st == SDSL_HIP_OK)
        st = fm_build_count_tab(f);
    return st;
}

extern "C" {

static sdsl_hip_status sdsl_hip_fm_create_from_bwt_ex_impl(const uint8_t * bwt, uint64_t n, int32_t device, uint32_t flags,
                                               sdsl_hip_fm_t * out)
{
    if (!out || !bwt || n == 0)
    {
        set_error("fm_create_from_bwt: null/empty argument");
        return SDSL_HIP_ERR_INVALID;
    }
    *out = nullptr;
    SH_TRY(check_device(device));
    sdsl_hip_fm_s * f = new (std::nothrow) sdsl_hip_fm_s();
    if (!f)
        return SDSL_HIP_ERR_NOMEM;
    Staged b;
    sdsl_hip_status st = b.in(bwt, n, nullptr);
    if (st == SDSL_HIP_OK)
        st = fm_from_device_bwt(f, (const uint8_t *)b.dev, n, device, flags);
    if (st != SDSL_HIP_OK)
    {
        fm_free(f);
        return st;
    }
    *out = f;
    return SDSL_HIP_OK;
}
// no exception crosses the C ABI: a malformed stream or an exhausted host becomes a status code
sdsl_hip_status sdsl_hip_fm_create_from_bwt_ex(const uint8_t * bwt, uint64_t n, int32_t device, uint32_t flags,
                                               sdsl_hip_fm_t * out)
{
    return guarded("fm_create_from_bwt_ex", [&] { return sdsl_hip_fm_create_from_bwt_ex_impl(bwt, n, device, flags, out); });
}

static sdsl_hip_status sdsl_hip_fm_create_from_text_ex_impl(const uint8_t * text, uint64_t n_text, int32_t device, uint32_t flags,
                                                sdsl_hip_fm_t * out)
{
    if (!out || (!text && n_text))
    {
        set_error("fm_create_from_text: null argument");
        return SDSL_HIP_ERR_INVALID;
    }
    *out = nullptr;
    SH_TRY(check_device(device));
    uint64_t zero_at = ~UINT64_C(0);
    if (n_text && is_device_ptr(text))
    { // a text that already lives in HBM is checked and sorted where it is
        SH_HIP(hipSetDevice(device));
        DevBuf d_first;
        SH_TRY(d_first.alloc(8));
        SH_HIP(hipMemset(d_first.p, 0xFF, 8));
        hipLaunchKernelGGL(k_first_zero_byte, dim3(grid_for(n_text, 256, 256u * 16u)), dim3(256), 0, 0, text, n_text,
                           d_first.as<unsigned long long>());
        SH_HIP(hipGetLastError());
        SH_HIP(hipMemcpy(&zero_at, d_first.p, 8, hipMemcpyDeviceToHost));
    }
    else if (const void * z = n_text ? memchr(text, 0, n_text) : nullptr)
        zero_at = (uint64_t)((const uint8_t *)z - text);
    if (zero_at != ~UINT64_C(0))
    {
        set_error("text contains a 0 byte at %llu (SDSL's construct refuses this too, construct.hpp:41)",
                  (unsigned long long)zero_at);
        return SDSL_HIP_ERR_INVALID;
    }
    DevBuf d_bwt, d_sa;
    const char * force64 = getenv("SDSL_HIP_SA64"); // (test knob: the 64-bit sorter on a text of any size)
    if (n_text + 1 >= kLimSorter32Symbols || (force64 && atoi(force64) != 0))
    { // 2^32 symbols and more: 64-bit suffixes (sa.hip).  The index keeps SA / ISA samples at SDSL's default densities
      // (csa_wt<..., 32, 64>: csa_wt.hpp:56) AND, room permitting, the whole 64-bit suffix array, the text and a k-mer table with
      // 40-bit intervals (k <= 6): count() of large batches takes the wide variants of the flat kernels (fm_count2.hip).
        SH_TRY(sa_build_bwt_device64(text, n_text, device, d_bwt, d_sa));
        sdsl_hip_fm_s * f = new (std::nothrow) sdsl_hip_fm_s();
        if (!f)
            return SDSL_HIP_ERR_NOMEM;
        const uint64_t n = n_text + 1;
        sdsl_hip_status st = sa_samples_device64(d_sa.as<uint64_t>(), n, 32, 64, &f->d_sa_s, &f->d_isa_s);
        // the whole suffix array (8 bytes per suffix) and the text stay beside the samples, as for a small index: count() finishes
        // single suffixes against the text, csa[i] and locate are gathers; sdsl_hip_fm_drop_sa releases both (SDSL_HIP_FM_KEEP_SA64=0:
        // never kept).  No room for them: the index works from its samples.
        const char * keep_env = getenv("SDSL_HIP_FM_KEEP_SA64");
        if (keep_env && atoi(keep_env) == 0)
            d_sa.release();
        else
        {
            f->d_sa64 = std::move(d_sa);
            if (fm_verify_enabled() && f->d_text.alloc(n_text) == SDSL_HIP_OK)
                if (hipMemcpy(f->d_text.p, text, n_text, hipMemcpyDefault) != hipSuccess)
                    f->d_text.release();
        }
        if (st == SDSL_HIP_OK)
        {
            f->sa_dens = 32;
            f->isa_dens = 64;
            f->n_sa_s = (n + 31) / 32;
            f->n_isa_s = (n + 63) / 64;
            st = fm_from_device_bwt(f, d_bwt.as<uint8_t>(), n, device, flags);
        }
        if (st == SDSL_HIP_OK)
            st = fm_build_deep_default(f);
        if (st != SDSL_HIP_OK)
        {
            fm_free(f);
            return st;
        }
        *out = f;
        return SDSL_HIP_OK;
    }
    SH_TRY(sa_build_bwt_device(text, n_text, device, d_bwt, d_sa)); // suffix array and BWT never leave the device
    sdsl_hip_fm_s * f = new (std::nothrow) sdsl_hip_fm_s();
    if (!f)
        return SDSL_HIP_ERR_NOMEM;
    f->d_sa = std::move(d_sa);
    if (n_text && fm_verify_enabled() && f->d_text.alloc(n_text) == SDSL_HIP_OK) // (no room: count() simply walks every character)
        if (hipMemcpy(f->d_text.p, text, n_text, hipMemcpyDefault) != hipSuccess)
            f->d_text.release();
    sdsl_hip_status st = fm_from_device_bwt(f, d_bwt.as<uint8_t>(), n_text + 1, device, flags);
    if (st == SDSL_HIP_OK)
        st = fm_build_deep_default(f); // the k-mer table of count() (fm_count2.hip), from the suffix array and the text
    if (st != SDSL_HIP_OK)
    {
        fm_free(f);
        return st;
    }
    *out = f;
    return SDSL_HIP_OK;
}
// no exception crosses the C ABI: a malformed stream or an exhausted host becomes a status code
sdsl_hip_status sdsl_hip_fm_create_from_text_ex(const uint8_t * text, uint64_t n_text, int32_t device, uint32_t flags,
                                                sdsl_hip_fm_t * out)
{
    return guarded("fm_create_from_text_ex", [&] { return sdsl_hip_fm_create_from_text_ex_impl(text, n_text, device, flags, out); });
}

static sdsl_hip_status sdsl_hip_fm_create_from_bwt_impl(const uint8_t * bwt, uint64_t n, int32_t device, sdsl_hip_fm_t * out)
{
    return sdsl_hip_fm_create_from_bwt_ex(bwt, n, device, 0, out);
}
// no exception crosses the C ABI: a malformed stream or an exhausted host becomes a status code
sdsl_hip_status sdsl_hip_fm_create_from_bwt(const uint8_t * bwt, uint64_t n, int32_t device, sdsl_hip_fm_t * out)
{
    return guarded("fm_create_from_bwt", [&] { return sdsl_hip_fm_create_from_bwt_impl(bwt, n, device, out); });
}

static sdsl_hip_status sdsl_hip_fm_create_from_text_impl(const uint8_t * text, uint64_t n_text, int32_t device, sdsl_hip_fm_t * out)
{
    return sdsl_hip_fm_create_from_text_ex(text, n_text, device, 0, out);
}
// no exception crosses the C ABI: a malformed stream or an exhausted host becomes a status code
sdsl_hip_status sdsl_hip_fm_create_from_text(const uint8_t * text, uint64_t n_text, int32_t device, sdsl_hip_fm_t * out)
{
    return guarded("fm_create_from_text", [&] { return sdsl_hip_fm_create_from_text_impl(text, n_text, device, out); });
}

static sdsl_hip_status sdsl_hip_fm_create_from_sdsl_impl(const void * bytes, size_t len, int32_t layout, int32_t device,
                                             sdsl_hip_fm_t * out)
{
    return sdsl_hip_fm_create_from_sdsl_ex(bytes, len, layout, 0, 0, device, out);
}
// no exception crosses the C ABI: a malformed stream or an exhausted host becomes a status code
sdsl_hip_status sdsl_hip_fm_create_from_sdsl(const void * bytes, size_t len, int32_t layout, int32_t device,
                                             sdsl_hip_fm_t * out)
{
    return guarded("fm_create_from_sdsl", [&] { return sdsl_hip_fm_create_from_sdsl_impl(bytes, len, layout, device, out); });
}

// uploads an SDSL sample vector (int_vector<0>) as u64; every value must be < n
static sdsl_hip_status fm_upload_samples(const HostIntVec & v, uint64_t expect, uint64_t n, const char * what, DevBuf & d)
{
    if (v.size() != expect)
    {
        set_error("csa_wt stream: %s holds %llu entries, %llu expected for this density", what,
                  (unsigned long long)v.size(), (unsigned long long)expect);
        return SDSL_HIP_ERR_FORMAT;
    }
    std::vector<uint64_t> h(expect);
    for (uint64_t i = 0; i < expect; ++i)
    {
        h[i] = v.get(i);
        if (h[i] >= n)
        {
            set_error("csa_wt stream: %s[%llu] is out of range", what, (unsigned long long)i);
            return SDSL_HIP_ERR_FORMAT;
        }
    }
    SH_TRY(d.alloc(expect * 8));
    if (expect)
        SH_HIP(hipMemcpy(d.p, h.data(), expect * 8, hipMemcpyHostToDevice));
    return SDSL_HIP_OK;
}

static sdsl_hip_status sdsl_hip_fm_create_from_sdsl_ex_impl(const void * bytes, size_t len, int32_t layout, uint32_t sa_dens,
                                                uint32_t isa_dens, int32_t device, sdsl_hip_fm_t * out)
{
    if (!out || !bytes)
    {
        set_error("fm_create_from_sdsl: null argument");
        return SDSL_HIP_ERR_INVALID;
    }
    *out = nullptr;
    SH_TRY(check_device(device));
    sdsl_hip_fm_s * f = new (std::nothrow) sdsl_hip_fm_s();
    if (!f)
        return SDSL_HIP_ERR_NOMEM;
    f->device = device;
    f->wt = sdsl_hip_wt_alloc();
    StreamReader rd(bytes, len);
    sdsl_hip_status st = f->wt ? SDSL_HIP_OK : SDSL_HIP_ERR_NOMEM;
    if (st == SDSL_HIP_OK)
        st = wt_build_from_stream(sdsl_hip_wt_host(f->wt), rd, layout, device);
    if (st == SDSL_HIP_OK)
        st = sdsl_hip_wt_finish(f->wt);
    if (st == SDSL_HIP_OK)
    {
        // csa_wt::serialize (csa_wt.hpp:389-402): wt, sa_sample, isa_sample, alphabet.  The default
        // sampling policies are plain int_vector<0> (csa_sampling_strategy.hpp:72,735) — skipped.
        // With the type's densities (template arguments, not part of the stream) the samples are kept for SA / ISA
        // access, locate and extract; with 0 they are skipped and the handle answers count-type queries only.
        HostIntVec c2c, comp2char, Cv, sa_s, isa_s;
        uint16_t sigma = 0;
        const bool keep = sa_dens != 0 && isa_dens != 0;
        if (!(keep ? rd.int_vector(sa_s) : rd.skip_int_vector()) || !(keep ? rd.int_vector(isa_s) : rd.skip_int_vector())
            || !rd.int_vector(c2c, 8) || !rd.int_vector(comp2char, 8)
            || !rd.int_vector(Cv, 64) || !rd.u16(sigma) || c2c.size() != 256 || sigma == 0 || sigma > 256
            || Cv.size() != (uint64_t)sigma + 1)
        {
            set_error("malformed csa_wt stream (offset %zu of %zu)", rd.pos, rd.len);
            st = SDSL_HIP_ERR_FORMAT;
        }
        else
        {
            memset(&f->tab, 0, sizeof f->tab);
            for (int c = 0; c < 256; ++c)
                f->tab.char2comp[c] = (uint8_t)c2c.get((uint64_t)c);
            for (uint32_t i = 0; i <= sigma; ++i)
                f->tab.C[i] = Cv.get(i);
            f->sigma = sigma;
            f->size = sdsl_hip_wt_host(f->wt).size;
            // the alphabet must describe the wavelet tree's symbol counts exactly (the kernels index with it)
            const WtHost & w = sdsl_hip_wt_host(f->wt);
            bool good = f->tab.C[sigma] == f->size && f->tab.C[0] == 0 && f->size >= 1;
            uint32_t next_cc = 0;
            for (int c = 0; c < 256 && good; ++c)
            {
                uint32_t cc = f->tab.char2comp[c];
                if (w.occ[c] == 0)
                    good = cc == 0;
                else
                    good = cc == next_cc && cc < sigma && f->tab.C[cc + 1] - f->tab.C[cc] == w.occ[c] && ++next_cc;
            }
            if (!good || next_cc != sigma)
            {
                set_error("csa_wt stream: alphabet (C, char2comp, sigma) is inconsistent with the wavelet tree");
                st = SDSL_HIP_ERR_FORMAT;
            }
            else
                st = fm_upload_tables(f);
            if (st == SDSL_HIP_OK)
                st = fm_build_jump(f);
            if (st == SDSL_HIP_OK)
                st = fm_build_count_tab(f);
            if (st == SDSL_HIP_OK && keep)
            {
                const uint64_t n = f->size;
                st = fm_upload_samples(sa_s, (n + sa_dens - 1) / sa_dens, n, "sa_sample", f->d_sa_s);
                if (st == SDSL_HIP_OK)
                    st = fm_upload_samples(isa_s, (n + isa_dens - 1) / isa_dens, n, "isa_sample", f->d_isa_s);
                if (st == SDSL_HIP_OK)
                {
                    f->sa_dens = sa_dens;
                    f->isa_dens = isa_dens;
                    f->n_sa_s = (n + sa_dens - 1) / sa_dens;
                    f->n_isa_s = (n + isa_dens - 1) / isa_dens;
                }
            }
        }
    }
    if (st != SDSL_HIP_OK)
    {
        fm_free(f);
        return st;
    }
    *out = f;
    return SDSL_HIP_OK;
}
// no exception crosses the C ABI: a malformed stream or an exhausted host becomes a status code
sdsl_hip_status sdsl_hip_fm_create_from_sdsl_ex(const void * bytes, size_t len, int32_t layout, uint32_t sa_dens,
                                                uint32_t isa_dens, int32_t device, sdsl_hip_fm_t * out)
{
    return guarded("fm_create_from_sdsl_ex", [&] { return sdsl_hip_fm_create_from_sdsl_ex_impl(bytes, len, layout, sa_dens, isa_dens, device, out); });
}

sdsl_hip_status sdsl_hip_fm_set_jump_depth(sdsl_hip_fm_t fm, uint32_t k)
{
    if (!fm)
    {
        set_error("fm_set_jump_depth: null handle");
        return SDSL_HIP_ERR_INVALID;
    }
    SH_HIP(hipSetDevice(fm->device));
    return fm_build_jump_k(fm, k);
}
uint32_t sdsl_hip_fm_jump_depth(sdsl_hip_fm_t fm)
{
    return fm ? fm->jump_k : 0;
}

sdsl_hip_status sdsl_hip_fm_set_kmer_table(sdsl_hip_fm_t fm, uint32_t k_max, uint64_t budget_bytes)
{
    if (!fm)
    {
        set_error("fm_set_kmer_table: null handle");
        return SDSL_HIP_ERR_INVALID;
    }
    SH_HIP(hipSetDevice(fm->device));
    SH_HIP(hipDeviceSynchronize()); // batches in flight on any stream may still read the table that is about to be replaced
    return fm_build_deep(fm, k_max, budget_bytes);
}
uint32_t sdsl_hip_fm_kmer_table_depth(sdsl_hip_fm_t fm)
{
    return fm ? fm->deep_k : 0;
}
uint64_t sdsl_hip_fm_kmer_table_bytes(sdsl_hip_fm_t fm)
{
    return fm ? fm->d_deep.bytes : 0;
}

// The opposite of drop_sa, for an index that came from an SDSL stream: the text is read back through the ISA samples
// (sdsl_hip_fm_extract_batch: thousands of independent walks), suffix-sorted on the device like a text handed to
// sdsl_hip_fm_create_from_text, and the result is checked against the stream's own SA samples before it is kept.
} // extern "C" (a kernel template)
template <class SA>
__global__ __launch_bounds__(256) void k_fm_check_samples(const SA * __restrict__ sa, const uint64_t * __restrict__ samples,
                                                          uint32_t s32, uint64_t n_samples, uint64_t dens, unsigned * __restrict__ bad)
{
    for (uint64_t k = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; k < n_samples; k += (uint64_t)gridDim.x * blockDim.x)
        if ((uint64_t)sa[k * dens] != loc_sample(samples, s32, k))
            atomicAdd(bad, 1u);
}

extern "C" {
__global__ __launch_bounds__(256) void k_fm_narrow_samples(const uint64_t * __restrict__ in, uint32_t * __restrict__ out, uint64_t n)
{
    for (uint64_t k = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; k < n; k += (uint64_t)gridDim.x * blockDim.x)
        out[k] = (uint32_t)in[k];
}

// the whole suffix array and the text back in HBM (no tables rebuilt)
static sdsl_hip_status fm_ensure_sa_text(sdsl_hip_fm_t fm)
{
    SH_HIP(hipSetDevice(fm->device));
    const bool wide = fm->d_sa64.p || (!fm->d_sa.p && fm->size >= kLimSorter32Symbols); // the 64-bit sorter and suffix array (sa.hip), as at creation
    if (!((wide ? fm->d_sa64.p : fm->d_sa.p) && fm->d_text.p))
    {
        if (fm->size < 2 || fm->size >= kLimFmFastSymbols || !fm->isa_dens || !fm->sa_dens)
        {
            set_error("fm_restore_suffix_array: needs the index's SA and ISA samples (load the stream with its densities: "
                      "sdsl_hip_fm_create_from_sdsl_ex) and fewer than 2^39 symbols");
            return SDSL_HIP_ERR_UNSUPPORTED;
        }
        const uint64_t n_text = fm->size - 1;
        DevBuf d_text, d_bwt, d_sa, d_bad;
        SH_TRY(d_text.alloc(n_text));
        const uint64_t b = 0, e = n_text - 1;
        uint64_t total = 0;
        SH_TRY(sdsl_hip_fm_extract_batch(fm, &b, &e, 1, nullptr, d_text.as<uint8_t>(), n_text, &total, nullptr));
        if (total != n_text)
        {
            set_error("fm_restore_suffix_array: the index gave back %llu of %llu symbols", (unsigned long long)total, (unsigned long long)n_text);
            return SDSL_HIP_ERR_FORMAT;
        }
        SH_TRY(d_bad.alloc(4, true));
        if (wide)
        {
            SH_TRY(sa_build_bwt_device64(d_text.as<uint8_t>(), n_text, fm->device, d_bwt, d_sa));
            hipLaunchKernelGGL(k_fm_check_samples<uint64_t>, dim3(grid_for(fm->n_sa_s, 256, 256u * 8u)), dim3(256), 0, 0, d_sa.as<uint64_t>(),
                               fm->d_sa_s.as<uint64_t>(), fm->samples32 ? 1u : 0u, fm->n_sa_s, (uint64_t)fm->sa_dens, d_bad.as<unsigned>());
        }
        else
        {
            SH_TRY(sa_build_bwt_device(d_text.as<uint8_t>(), n_text, fm->device, d_bwt, d_sa));
            hipLaunchKernelGGL(k_fm_check_samples<uint32_t>, dim3(grid_for(fm->n_sa_s, 256, 256u * 8u)), dim3(256), 0, 0, d_sa.as<uint32_t>(),
                               fm->d_sa_s.as<uint64_t>(), fm->samples32 ? 1u : 0u, fm->n_sa_s, (uint64_t)fm->sa_dens, d_bad.as<unsigned>());
        }
        SH_HIP(hipGetLastError());
        unsigned bad = 0;
        SH_HIP(hipMemcpy(&bad, d_bad.p, 4, hipMemcpyDeviceToHost));
        if (bad)
        { // (a stream whose samples and wavelet tree do not describe the same text)
            set_error("fm_restore_suffix_array: %u of the stream's SA samples disagree with the suffix array of the text the index spells", bad);
            return SDSL_HIP_ERR_FORMAT;
        }
        if (wide)
            fm->d_sa64 = std::move(d_sa);
        else
            fm->d_sa = std::move(d_sa);
        fm->d_text = std::move(d_text);
    }
    return SDSL_HIP_OK;
}

sdsl_hip_status sdsl_hip_fm_restore_suffix_array(sdsl_hip_fm_t fm)
{
    if (!fm)
    {
        set_error("fm_restore_suffix_array: null handle");
        return SDSL_HIP_ERR_INVALID;
    }
    SH_TRY(fm_ensure_sa_text(fm));
    if (!fm->ctab_ok)
        SH_TRY(fm_build_count_tab(fm));
    if (!fm->deep_k)
        SH_TRY(fm_build_deep_default(fm));
    return SDSL_HIP_OK;
}

// drop_sa with the sampling densities of the caller's csa_wt type (t_dens / t_inv_dens, csa_wt.hpp:51-57; 0 = keep what the index
// has, or SDSL's defaults 32 / 64 if it has none).  Samples of other densities than the index holds can only be taken while the whole
// suffix array is still there.
sdsl_hip_status sdsl_hip_fm_drop_sa_ex(sdsl_hip_fm_t fm, uint32_t sa_dens, uint32_t isa_dens)
{
    if (!fm)
        return SDSL_HIP_ERR_INVALID;
    SH_HIP(hipSetDevice(fm->device));
    const uint32_t want_sa = sa_dens ? sa_dens : (fm->sa_dens ? fm->sa_dens : 32u), want_isa = isa_dens ? isa_dens : (fm->isa_dens ? fm->isa_dens : 64u);
    if (want_sa != fm->sa_dens || want_isa != fm->isa_dens)
    {
        if (!fm->d_sa.p && !fm->d_sa64.p)
        {
            set_error("fm_drop_sa: samples of densities %u / %u can only be taken from the whole suffix array, which this index no longer "
                      "has (it holds %u / %u)", want_sa, want_isa, fm->sa_dens, fm->isa_dens);
            return SDSL_HIP_ERR_UNSUPPORTED;
        }
        const uint64_t n = fm->size;
        if (fm->d_sa.p)
            SH_TRY(sa_samples_device(fm->d_sa.as<uint32_t>(), n, want_sa, want_isa, &fm->d_sa_s, &fm->d_isa_s));
        else
            SH_TRY(sa_samples_device64(fm->d_sa64.as<uint64_t>(), n, want_sa, want_isa, &fm->d_sa_s, &fm->d_isa_s));
        fm->sa_dens = want_sa;
        fm->isa_dens = want_isa;
        fm->n_sa_s = (n + want_sa - 1) / want_sa;
        fm->n_isa_s = (n + want_isa - 1) / want_isa;
        fm->samples32 = false;
    }
    fm->d_sa.release();
    fm->d_sa64.release();
    fm->d_text.release(); // (only useful beside the whole suffix array)
    return SDSL_HIP_OK;
}

sdsl_hip_status sdsl_hip_fm_drop_sa(sdsl_hip_fm_t fm)
{
    return sdsl_hip_fm_drop_sa_ex(fm, 0, 0);
}

// ---- the index at a chosen footprint ------------------------------------------------------------------------------------
// What the reference's csa_wt holds (csa_wt.hpp:389-402: wavelet tree, SA samples, ISA samples, alphabet) is 0.93 bytes per symbol
// for csa_wt<wt_huff<>, 32, 64> on English text; an index created from text here holds 8.3 (whole suffix array, text, both tree
// layouts, k-mer table), because HBM is there to be used.  sdsl_hip_fm_set_footprint gives memory back in the order that costs
// count() least per byte:
//   1. SDSL's binary tree levels with their select directories (serialisation and select on trees without the fused directory
//      rebuild them from the fused lines when asked: wt_restore_binary) — count() / rank / access / LF walks lose nothing;
//   2. the whole suffix array and the text (the single-suffix shortcut of count(), csa[i] as a gather) -> SDSL's default samples
//      (32 / 64), packed to 32 bits per sample; the dense jump table shrinks to <= 4 MiB; the k-mer table is rebuilt first, as deep
//      as the budget still allows (it is built FROM suffix array and text; an index that has dropped them gets them back for the
//      time of the call: fm_ensure_sa_text).
// What remains at the floor: the fused lines (4 bits per symbol and fused level), the samples, the alphabet tables.
void sdsl_hip_fm_footprint_parts(sdsl_hip_fm_t fm, uint64_t parts[8])
{
    if (!parts)
        return;
    for (int i = 0; i < 8; ++i)
        parts[i] = 0;
    if (!fm)
        return;
    const WtHost & w = sdsl_hip_wt_host(fm->wt);
    parts[0] = w.bv.device_bytes() + w.rrr.device_bytes();
    parts[1] = w.d_fused.bytes + w.d_ftables.bytes + w.d_fsuper.bytes + w.d_fwalk.bytes + w.d_fsteps.bytes + w.d_fsel.bytes + w.d_fsel_tables.bytes + w.d_tables_f.bytes;
    parts[2] = fm->d_sa.bytes + fm->d_sa64.bytes;
    parts[3] = fm->d_text.bytes;
    parts[4] = fm->d_sa_s.bytes + fm->d_isa_s.bytes;
    parts[5] = fm->d_deep.bytes;
    parts[6] = fm->d_jump.bytes;
    parts[7] = fm->d_tab.bytes + fm->d_ctab.bytes + w.d_tables.bytes;
}

// 32-bit copies of `n_sa` SA samples and `n_isa` ISA samples (u64 each) — into buffers of the caller's
static sdsl_hip_status fm_narrow_samples(const DevBuf & sa_s, uint64_t n_sa, const DevBuf & isa_s, uint64_t n_isa, DevBuf & a, DevBuf & b)
{
    SH_TRY(a.alloc(std::max<uint64_t>(n_sa, 1) * 4));
    SH_TRY(b.alloc(std::max<uint64_t>(n_isa, 1) * 4));
    hipLaunchKernelGGL(k_fm_narrow_samples, dim3(grid_for(n_sa, 256, 256u * 8u)), dim3(256), 0, 0, sa_s.as<uint64_t>(), a.as<uint32_t>(), n_sa);
    hipLaunchKernelGGL(k_fm_narrow_samples, dim3(grid_for(n_isa, 256, 256u * 8u)), dim3(256), 0, 0, isa_s.as<uint64_t>(), b.as<uint32_t>(), n_isa);
    SH_HIP(hipGetLastError());
    SH_HIP(hipDeviceSynchronize());
    return SDSL_HIP_OK;
}

// TRANSACTIONAL (round 6; VERDICT r05 / ADVICE r05: a failure half-way used to leave a half-converted index).  Everything that can
// fail — new samples, their 32-bit copies, the k-mer table, the smaller jump table — is built into temporaries while the index still
// holds all it had; the commit at the end consists of moves and releases only.  Two things may already have changed when an error is
// returned, and both leave a complete index with the same answers: the whole suffix array and the text may be BACK (an index that had
// dropped them gets them back to build the k-mer table from), and the k-mer table may be the new one.
//
// backend 1, csa_wt<wt_huff<rrr_vector<63>>> (round 6): the tree is the rrr structure itself (there are no binary levels to drop and no
// fused lines); suffix array and text -> SDSL's samples packed to 32 bits, the k-mer table as deep as the budget allows (the lane kernel
// of wt_rrr.hip starts from it), the jump table <= 4 MiB.  The floor is rrr tree + samples + alphabet.
static sdsl_hip_status sdsl_hip_fm_set_footprint_impl(sdsl_hip_fm_t fm, uint64_t max_bytes)
{
    if (!fm)
    {
        set_error("fm_set_footprint: null handle");
        return SDSL_HIP_ERR_INVALID;
    }
    SH_HIP(hipSetDevice(fm->device));
    SH_HIP(hipDeviceSynchronize()); // nothing in flight may still read what is released below
    if (sdsl_hip_fm_device_bytes(fm) <= max_bytes)
        return SDSL_HIP_OK;
    WtHost & w = sdsl_hip_wt_host(fm->wt);
    const bool rrr = w.backend == 1;
    if (!rrr)
    {
        if (w.backend == 0 && w.d_fused.p && !fm->ctab_ok)
            SH_TRY(fm_build_count_tab(fm)); // (an index loaded from a stream gets the flat kernel's tables on first need)
        if (w.backend != 0 || !w.d_fused.p || !w.d_ftables.p || !fm->ctab_ok)
        {
            set_error("fm_set_footprint: the compact forms exist for an index on the plain wavelet tree with its fused layout (fewer than "
                      "2^36 symbols) and for csa_wt<wt_huff<rrr_vector<63>>>; this one holds %llu bytes",
                      (unsigned long long)sdsl_hip_fm_device_bytes(fm));
            return SDSL_HIP_ERR_UNSUPPORTED;
        }
    }
    uint64_t parts[8];
    sdsl_hip_fm_footprint_parts(fm, parts);
    // 1. the binary levels alone (plain tree)
    if (!rrr && sdsl_hip_fm_device_bytes(fm) - parts[0] <= max_bytes)
        return wt_drop_binary(w);
    // 2. samples instead of suffix array and text: what is the floor, what is left for the k-mer table?
    const uint64_t n = fm->size;
    const bool wide = n >= (UINT64_C(1) << 32); // 40-bit intervals; the samples stay 64 bits wide
    const bool have_sa = fm->d_sa.p || fm->d_sa64.p;
    const bool make_samples = !fm->sa_dens || !fm->isa_dens;
    if (make_samples && !have_sa)
    {
        set_error("fm_set_footprint: this index has neither its suffix array nor SA / ISA samples (created from a BWT, or loaded "
                  "without densities): there is nothing smaller to fall back to");
        return SDSL_HIP_ERR_UNSUPPORTED;
    }
    const uint64_t n_sa_s = make_samples ? (n + 31) / 32 : fm->n_sa_s, n_isa_s = make_samples ? (n + 63) / 64 : fm->n_isa_s;
    // the dense jump table (small batches, intervals, ragged patterns) keeps a depth of at most 4 MiB and 1/16 byte per symbol
    const uint64_t jump_cap = std::min<uint64_t>(UINT64_C(4) << 20, n / 16);
    uint32_t jk = 0;
    uint64_t jump_bytes = 16;
    while (jk < fm->jump_k && jump_bytes * fm->sigma <= jump_cap)
    {
        jump_bytes *= fm->sigma;
        ++jk;
    }
    const uint64_t tree_bytes = rrr ? parts[0] : parts[1];
    const uint64_t floor_bytes = tree_bytes + parts[7] + (n_sa_s + n_isa_s) * (wide ? 8 : 4) + (jk ? jump_bytes : 0);
    if (floor_bytes > max_bytes)
    {
        set_error("fm_set_footprint: the smallest form of this index (%s, SA / ISA samples at %u / %u, alphabet) is %llu "
                  "bytes; %llu were asked for.%s", rrr ? "the rrr-compressed tree" : "fused tree lines",
                  make_samples ? 32u : fm->sa_dens, make_samples ? 64u : fm->isa_dens, (unsigned long long)floor_bytes,
                  (unsigned long long)max_bytes, rrr ? "" : "  csa_wt<wt_huff<rrr_vector<63>>> (SDSL_HIP_WT_RRR63) is the smaller structure");
        return SDSL_HIP_ERR_INVALID;
    }
    // ---- everything that can fail: built beside what the index holds ----
    DevBuf new_sa_s, new_isa_s; // u64 samples at 32 / 64, when the index has none
    if (make_samples)
    {
        if (fm->d_sa64.p)
            SH_TRY(sa_samples_device64(fm->d_sa64.as<uint64_t>(), n, 32, 64, &new_sa_s, &new_isa_s));
        else
            SH_TRY(sa_samples_device(fm->d_sa.as<uint32_t>(), n, 32, 64, &new_sa_s, &new_isa_s));
    }
    DevBuf sa_32, isa_32; // their 32-bit copies (fewer than 2^32 symbols)
    const bool pack = !wide && !(fm->samples32 && !make_samples);
    if (pack)
        SH_TRY(fm_narrow_samples(make_samples ? new_sa_s : fm->d_sa_s, n_sa_s, make_samples ? new_isa_s : fm->d_isa_s, n_isa_s, sa_32, isa_32));
    // the k-mer table, as deep as the rest of the budget allows.  Who reads it: the flat kernels of the plain index, the lane kernel of the
    // rrr one (below 2^32 symbols).  Built from suffix array and text: an index that has dropped them gets them back for the time of the call.
    const uint64_t table_budget = max_bytes - floor_bytes;
    const bool table_user = !rrr || n < (UINT64_C(1) << 32);
    bool drop_table = false;
    sdsl_hip_status deep_st = SDSL_HIP_OK;
    if (!table_user || table_budget < 128)
        drop_table = fm->deep_k != 0;
    else if (fm->d_deep.bytes > table_budget || (!fm->deep_k && table_budget >= (UINT64_C(64) << 10)))
    {
        if (make_samples)
        { // (fm_ensure_sa_text reads the text back through samples: an index that still has to make them has its suffix array)
            if (!fm->d_text.p)
                drop_table = fm->deep_k != 0; // created without a resident text: no table can be built
            else
                deep_st = fm_build_deep(fm, 8, table_budget);
        }
        else
        {
            SH_TRY(fm_ensure_sa_text(fm));
            deep_st = fm_build_deep(fm, 8, table_budget);
        }
        // (an index whose suffix array is not of the width its intervals call for — 2^32 - 2 or 2^32 - 1 symbols, or a small
        // text sent through the 64-bit sorter by SDSL_HIP_SA64 — cannot build the table: it goes without one, the call does not fail)
        if (deep_st != SDSL_HIP_OK && deep_st != SDSL_HIP_ERR_UNSUPPORTED)
            return deep_st;
        if (fm->d_deep.bytes > table_budget)
            drop_table = true; // (no depth fits: fm_build_deep kept what there was)
    }
    if (jk != fm->jump_k)
        SH_TRY(fm_build_jump_k(fm, jk)); // (keeps the old table if the new one cannot be built)
    // ---- commit: moves and releases, nothing that can fail ----
    if (make_samples)
    {
        fm->d_sa_s = std::move(new_sa_s);
        fm->d_isa_s = std::move(new_isa_s);
        fm->sa_dens = 32;
        fm->isa_dens = 64;
        fm->n_sa_s = n_sa_s;
        fm->n_isa_s = n_isa_s;
        fm->samples32 = false;
    }
    if (pack)
    {
        fm->d_sa_s = std::move(sa_32);
        fm->d_isa_s = std::move(isa_32);
        fm->samples32 = true;
    }
    if (drop_table)
        (void)fm_build_deep(fm, 0, 0);
    fm->d_sa.release();
    fm->d_sa64.release();
    fm->d_text.release();
    if (!rrr)
        (void)wt_drop_binary(w); // (its only failure is the precondition checked at the top)
    if (sdsl_hip_fm_device_bytes(fm) > max_bytes)
    {
        set_error("internal: fm_set_footprint left %llu bytes, %llu were asked for", (unsigned long long)sdsl_hip_fm_device_bytes(fm),
                  (unsigned long long)max_bytes);
        return SDSL_HIP_ERR_HIP;
    }
    return SDSL_HIP_OK;
}
sdsl_hip_status sdsl_hip_fm_set_footprint(sdsl_hip_fm_t fm, uint64_t max_bytes)
{
    return guarded("fm_set_footprint", [&] { return sdsl_hip_fm_set_footprint_impl(fm, max_bytes); });
}

static sdsl_hip_status sdsl_hip_fm_serialize_impl(sdsl_hip_fm_t fm, uint32_t sa_dens, uint32_t isa_dens, void * buf, size_t cap,
                                      size_t * written)
{
    return sdsl_hip_fm_serialize_ex(fm, SDSL_HIP_LAYOUT_BV_SCAN, sa_dens, isa_dens, buf, cap, written);
}
// no exception crosses the C ABI: a malformed stream or an exhausted host becomes a status code
sdsl_hip_status sdsl_hip_fm_serialize(sdsl_hip_fm_t fm, uint32_t sa_dens, uint32_t isa_dens, void * buf, size_t cap,
                                      size_t * written)
{
    return guarded("fm_serialize", [&] { return sdsl_hip_fm_serialize_impl(fm, sa_dens, isa_dens, buf, cap, written); });
}

static sdsl_hip_status sdsl_hip_fm_serialize_ex_impl(sdsl_hip_fm_t fm, int32_t layout, uint32_t sa_dens, uint32_t isa_dens, void * buf,
                                         size_t cap, size_t * written)
{
    if (!fm || sa_dens == 0 || isa_dens == 0)
    {
        set_error("fm_serialize: invalid argument");
        return SDSL_HIP_ERR_INVALID;
    }
    // samples come from the whole suffix array or, without it, from the samples the index holds — at their densities only (an
    // index of 2^32 symbols and more built from text keeps csa_wt's default 32 / 64; so does one loaded from a stream)
    const bool from_samples = !fm->d_sa.p && fm->d_sa_s.p && fm->d_isa_s.p && fm->sa_dens == sa_dens && fm->isa_dens == isa_dens;
    if (!fm->d_sa.p && !from_samples)
    {
        set_error("fm_serialize: the suffix array is not available (index not created from text, or dropped) and the index holds no "
                  "samples of the densities %u / %u", sa_dens, isa_dens);
        return SDSL_HIP_ERR_UNSUPPORTED;
    }
    SH_HIP(hipSetDevice(fm->device));
    const uint64_t key = ((uint64_t)(uint32_t)layout << 58) ^ ((uint64_t)sa_dens << 29) ^ (uint64_t)isa_dens;
    sdsl_hip_status cached;
    if (deliver_cached(fm->uid, key, buf, cap, written, cached))
        return cached;
    // 1. the wavelet tree in the requested flavour
    StreamWriter w;
    SH_TRY(sdsl_hip_wt_serialize_into(fm->wt, layout, w));
    // 2. SA and ISA samples as int_vector<0> of width hi(n)+1
    const uint64_t n = fm->size;
    std::vector<uint64_t> sa_s, isa_s;
    if (from_samples)
    {
        sa_s.resize(fm->n_sa_s);
        isa_s.resize(fm->n_isa_s);
        if (fm->samples32)
        { // packed by sdsl_hip_fm_set_footprint: u32 per sample
            std::vector<uint32_t> t(std::max(fm->n_sa_s, fm->n_isa_s));
            SH_HIP(hipMemcpy(t.data(), fm->d_sa_s.p, fm->n_sa_s * 4, hipMemcpyDeviceToHost));
            std::copy(t.begin(), t.begin() + fm->n_sa_s, sa_s.begin());
            SH_HIP(hipMemcpy(t.data(), fm->d_isa_s.p, fm->n_isa_s * 4, hipMemcpyDeviceToHost));
            std::copy(t.begin(), t.begin() + fm->n_isa_s, isa_s.begin());
        }
        else
        {
            SH_HIP(hipMemcpy(sa_s.data(), fm->d_sa_s.p, fm->n_sa_s * 8, hipMemcpyDeviceToHost));
            SH_HIP(hipMemcpy(isa_s.data(), fm->d_isa_s.p, fm->n_isa_s * 8, hipMemcpyDeviceToHost));
        }
    }
    else
        SH_TRY(sa_samples_to_host(fm->d_sa.as<uint32_t>(), n, sa_dens, isa_dens, sa_s, isa_s));
    const uint8_t width = (uint8_t)(hi64(n) + 1);
    PackedBuilder ps(sa_s.size(), width), pi(isa_s.size(), width);
    for (uint64_t i = 0; i < sa_s.size(); ++i)
        ps.set(i, sa_s[i]);
    for (uint64_t i = 0; i < isa_s.size(); ++i)
        pi.set(i, isa_s[i]);
    ps.write(w);
    pi.write(w);
    // 3. byte_alphabet::serialize (csa_alphabet_strategy.hpp:258-268)
    uint64_t c2c[32], comp2char[32];
    memset(c2c, 0, sizeof c2c);
    memset(comp2char, 0, sizeof comp2char);
    memcpy(c2c, fm->tab.char2comp, 256);
    const WtHost & wh = sdsl_hip_wt_host(fm->wt);
    for (int c = 0; c < 256; ++c)
        if (wh.occ[c])
            ((uint8_t *)comp2char)[fm->tab.char2comp[c]] = (uint8_t)c;
    w.int_vector(c2c, 256 * 8, 8);
    w.int_vector(comp2char, (uint64_t)fm->sigma * 8, 8);
    w.int_vector(fm->tab.C, ((uint64_t)fm->sigma + 1) * 64, 64);
    w.u16((uint16_t)fm->sigma);
    return deliver_and_cache(fm->uid, key, w, buf, cap, written);
}
// no exception crosses the C ABI: a malformed stream or an exhausted host becomes a status code
sdsl_hip_status sdsl_hip_fm_serialize_ex(sdsl_hip_fm_t fm, int32_t layout, uint32_t sa_dens, uint32_t isa_dens, void * buf,
                                         size_t cap, size_t * written)
{
    return guarded("fm_serialize_ex", [&] { return sdsl_hip_fm_serialize_ex_impl(fm, layout, sa_dens, isa_dens, buf, cap, written); });
}

sdsl_hip_status sdsl_hip_fm_destroy(sdsl_hip_fm_t fm)
{
    if (fm)
        (void)hipSetDevice(fm->device);
    fm_free(fm);
    return SDSL_HIP_OK;
}

uint64_t sdsl_hip_fm_size(sdsl_hip_fm_t fm)
{
    return fm ? fm->size : 0;
}
uint64_t sdsl_hip_fm_sigma(sdsl_hip_fm_t fm)
{
    return fm ? fm->sigma : 0;
}
uint64_t sdsl_hip_fm_device_bytes(sdsl_hip_fm_t fm)
{
    return fm ? sdsl_hip_wt_device_bytes(fm->wt) + fm->d_tab.bytes + fm->d_sa_s.bytes + fm->d_isa_s.bytes + fm->d_jump.bytes + fm->d_sa.bytes + fm->d_sa64.bytes + fm->d_text.bytes + fm->d_ctab.bytes + fm->d_deep.bytes : 0;
}
sdsl_hip_wt_t sdsl_hip_fm_wavelet_tree(sdsl_hip_fm_t fm)
{
    return fm ? fm->wt : nullptr;
}

sdsl_hip_status sdsl_hip_fm_alphabet(sdsl_hip_fm_t fm, uint8_t char2comp_out[256], uint64_t C_out[257])
{
    if (!fm || !char2comp_out || !C_out)
        return SDSL_HIP_ERR_INVALID;
    memcpy(char2comp_out, fm->tab.char2comp, 256);
    memcpy(C_out, fm->tab.C, sizeof fm->tab.C);
    return SDSL_HIP_OK;
}

static sdsl_hip_status fm_run(sdsl_hip_fm_t fm, const uint8_t * pats, uint32_t m, const uint64_t * offsets,
                              uint64_t total_bytes, uint64_t n_pat, uint64_t * out_cnt, uint64_t * out_l,
                              uint64_t * out_r, hipStream_t s, bool use_jump = true)
{
    const FmJump jump = use_jump ? fm->jump() : FmJump{nullptr, 0, 0};
    SH_HIP(hipSetDevice(fm->device));
    if (n_pat == 0)
        return SDSL_HIP_OK;
    Staged sp, so, sc, sl, sr;
    SH_TRY(sp.in(pats, total_bytes, s));
    if (offsets)
        SH_TRY(so.in(offsets, (n_pat + 1) * 8, s));
    const bool ival = out_l != nullptr;
    if (ival)
    {
        SH_TRY(sl.out(out_l, n_pat * 8));
        SH_TRY(sr.out(out_r, n_pat * 8));
    }
    else
        SH_TRY(sc.out(out_cnt, n_pat * 8));
    const WtHost & w = sdsl_hip_wt_host(fm->wt);
    if (!ival && !offsets && use_jump && w.backend == 0 && fm_fast_applies(fm, m, n_pat))
    { // large batches of fixed-length patterns on the fused layout: k-mer table + flat kernel + text comparison (fm_count2.hip)
        sdsl_hip_status st;
        {
            KernelTimer t(s);
            st = fm_count_fast(fm, (const uint8_t *)sp.dev, m, n_pat, (uint64_t *)sc.dev, fm_verify_enabled(), s);
        }
        if (st == SDSL_HIP_OK)
        {
            SH_TRY(sc.finish(s));
            if (sp.host)
                SH_HIP(hipStreamSynchronize(s));
            return SDSL_HIP_OK;
        }
        if (st != SDSL_HIP_ERR_NOMEM)
            return st; // (no working memory — also: the stream is being captured — leaves the batch to the lock-step kernel below)
    }
    // everything that uses the device's scratch pool is enqueued inside this scope: the lease (the pool's mutex) ends with the last
    // kernel that reads the scratch, BEFORE the results are copied out and the stream is waited for — other handles' large batches on
    // this device are not held up for the length of this one
    auto enqueue = [&]() -> sdsl_hip_status {
        unsigned grid = grid_for(n_pat, kQPB, 256u * 8u);
        // Large batches are answered in suffix order (see k_fm_keys): one key kernel + one radix sort, no synchronisation
        static const int sort_knob = getenv("SDSL_HIP_FM_SORT") ? atoi(getenv("SDSL_HIP_FM_SORT")) : -1;
        const bool ordered = sort_knob < 0 ? (n_pat >= (UINT64_C(1) << 16) && n_pat < UINT64_C(0xFFFFFFFF)) : sort_knob != 0;
        uint32_t * d_order = nullptr;
        // (keys, order and the sort's working memory: from the device's scratch pool, held until the search is enqueued — bv_host.hpp:
        // ScratchLease; not the stream-ordered allocator, sa.hip: sort_pairs_u64_u32)
        static DevBuf no_capture_scratch;
        ScratchLease lease;
        KernelTimer t(s); // covers key generation + sort + search: the whole cost of the batch
        if (ordered)
        {
            const size_t kb = (size_t)n_pat * 8, ib = (size_t)n_pat * 4, tb = (sort_pairs_u64_u32_temp_bytes(n_pat, 64u) + 255) & ~(size_t)255;
            SH_TRY(lease.acquire(fm->device, no_capture_scratch, 2 * kb + 2 * ib + tb + 256, s));
        }
        if (ordered && lease.p)
        {
            const size_t kb = (size_t)n_pat * 8, ib = (size_t)n_pat * 4;
            uint64_t * k0 = (uint64_t *)lease.p;
            uint64_t * k1 = k0 + n_pat;
            uint32_t * i0 = (uint32_t *)(k1 + n_pat);
            uint32_t * i1 = i0 + n_pat;
            uint8_t * tmp = reinterpret_cast<uint8_t *>(lease.p) + ((2 * kb + 2 * ib + 255) & ~(size_t)255);
            hipLaunchKernelGGL(k_fm_keys, dim3(grid_for(n_pat, 256, 256u * 8u)), dim3(256), 0, s, (const uint8_t *)sp.dev, m,
                               offsets ? (const uint64_t *)so.dev : nullptr, n_pat, k0, i0);
            SH_TRY(sort_pairs_u64_u32(k0, k1, i0, i1, n_pat, 64u, s, tmp, lease.bytes - (size_t)(tmp - reinterpret_cast<uint8_t *>(lease.p))));
            d_order = i1;
        }
        if (w.backend == 1)
        {
            const bool verify = !ival && fm->d_sa.p && fm->d_text.p && fm_verify_enabled() && fm->size < (UINT64_C(1) << 32);
            SH_TRY(fm_rrr_launch_count(w, fm->d_tab.as<FmTables>(), jump, use_jump ? fm->deep() : FmDeep{nullptr, 0, 0}, fm->size, (const uint8_t *)sp.dev, m,
                                       offsets ? (const uint64_t *)so.dev : nullptr, d_order, n_pat,
                                       ival ? nullptr : (uint64_t *)sc.dev, ival ? (uint64_t *)sl.dev : nullptr,
                                       ival ? (uint64_t *)sr.dev : nullptr, s, verify));
            if (verify)
                hipLaunchKernelGGL(k_fm_verify<uint32_t>, dim3(grid_for(n_pat, 256, 256u * 16u)), dim3(256), 0, s, fm->d_sa.as<uint32_t>(),
                                   fm->d_text.as<uint8_t>(), (const uint8_t *)sp.dev, m, offsets ? (const uint64_t *)so.dev : nullptr, n_pat,
                                   (uint64_t *)sc.dev, fm->size);
        }
        else
        {
            const FmTables * tab = fm->d_tab.as<FmTables>();
            const uint8_t * pp = (const uint8_t *)sp.dev;
            const uint64_t * oo = offsets ? (const uint64_t *)so.dev : nullptr;
            uint64_t *oc = ival ? nullptr : (uint64_t *)sc.dev, *ol = ival ? (uint64_t *)sl.dev : nullptr,
                     *orr = ival ? (uint64_t *)sr.dev : nullptr;
            const WtView v = w.view();
            if (ival && v.f_lines)
                hipLaunchKernelGGL((k_fm_count<false, true, true>), dim3(grid), dim3(kBlock), 0, s, v, tab, jump, fm->size, pp, m,
                                   oo, d_order, n_pat, oc, ol, orr, (const uint32_t *)nullptr, (const uint8_t *)nullptr);
            else if (ival)
                hipLaunchKernelGGL((k_fm_count<false, true, false>), dim3(grid), dim3(kBlock), 0, s, v, tab, jump, fm->size, pp,
                                   m, oo, d_order, n_pat, oc, ol, orr, (const uint32_t *)nullptr, (const uint8_t *)nullptr);
            else if (v.f_lines && fm->d_sa.p && fm->d_text.p && fm_verify_enabled())
            {
                hipLaunchKernelGGL((k_fm_count<false, false, true, true>), dim3(grid), dim3(kBlock), 0, s, v, tab, jump, fm->size, pp,
                                   m, oo, d_order, n_pat, oc, ol, orr, fm->d_sa.as<uint32_t>(), fm->d_text.as<uint8_t>());
                hipLaunchKernelGGL(k_fm_verify<uint32_t>, dim3(grid_for(n_pat, 256, 256u * 16u)), dim3(256), 0, s, fm->d_sa.as<uint32_t>(),
                                   fm->d_text.as<uint8_t>(), pp, m, oo, n_pat, oc, fm->size);
            }
            else if (v.f_lines && fm->d_sa64.p && fm->d_text.p && fm_verify_enabled() && fm->size < (UINT64_C(1) << 40))
            { // 2^32 suffixes and more: the pending word holds 40 bits of suffix and 23 of length (the kernel walks longer remainders)
                hipLaunchKernelGGL((k_fm_count<false, false, true, true>), dim3(grid), dim3(kBlock), 0, s, v, tab, jump, fm->size, pp,
                                   m, oo, d_order, n_pat, oc, ol, orr, (const uint32_t *)nullptr, fm->d_text.as<uint8_t>());
                hipLaunchKernelGGL(k_fm_verify<uint64_t>, dim3(grid_for(n_pat, 256, 256u * 16u)), dim3(256), 0, s, fm->d_sa64.as<uint64_t>(),
                                   fm->d_text.as<uint8_t>(), pp, m, oo, n_pat, oc, fm->size);
            }
            else if (v.f_lines)
                hipLaunchKernelGGL((k_fm_count<false, false, true>), dim3(grid), dim3(kBlock), 0, s, v, tab, jump, fm->size, pp,
                                   m, oo, d_order, n_pat, oc, ol, orr, (const uint32_t *)nullptr, (const uint8_t *)nullptr);
            else
                hipLaunchKernelGGL((k_fm_count<false, false, false>), dim3(grid), dim3(kBlock), 0, s, v, tab, jump, fm->size, pp,
                                   m, oo, d_order, n_pat, oc, ol, orr, (const uint32_t *)nullptr, (const uint8_t *)nullptr);
        }
        return SDSL_HIP_OK;
    };
    SH_TRY(enqueue());
    SH_HIP(hipGetLastError());
    if (ival)
    {
        SH_TRY(sl.finish(s));
        SH_TRY(sr.finish(s));
    }
    else
        SH_TRY(sc.finish(s));
    if (sp.host || so.host)
        SH_HIP(hipStreamSynchronize(s));
    return SDSL_HIP_OK;
}

// The jump-start table (fm_device.hpp FmJump), filled by the search kernel itself on all k-mers.  Default depth: the
// largest k whose table (16 bytes per k-mer) stays within half the size of the wavelet tree (at least 64 MiB) and
// within 4 * size entries; sdsl_hip_fm_set_jump_depth() chooses another one, SDSL_HIP_FM_JUMP=<k> overrides the default
// (0 = none; profiling).
static sdsl_hip_status fm_build_jump_k(sdsl_hip_fm_s * f, uint32_t k);

static sdsl_hip_status fm_build_jump(sdsl_hip_fm_s * f)
{
    const uint64_t sigma = f->sigma;
    if (sigma < 2)
        return SDSL_HIP_OK;
    if (const char * e = getenv("SDSL_HIP_FM_JUMP"))
        return fm_build_jump_k(f, (uint32_t)std::max(0, atoi(e)));
    const uint64_t budget = std::max<uint64_t>(UINT64_C(64) << 20, sdsl_hip_wt_device_bytes(f->wt) / 2) / 16;
    const uint64_t cap = std::min<uint64_t>(budget, 4 * f->size);
    uint32_t k = 0;
    uint64_t count = 1;
    while (count * sigma <= cap && k < 8)
    {
        count *= sigma;
        ++k;
    }
    return fm_build_jump_k(f, k);
}

static sdsl_hip_status fm_build_jump_k_fresh(sdsl_hip_fm_s * f, uint32_t k);
// the table of depth k in place of the one the index has; a failure keeps the old one
static sdsl_hip_status fm_build_jump_k(sdsl_hip_fm_s * f, uint32_t k)
{
    DevBuf old = std::move(f->d_jump);
    const uint32_t old_k = f->jump_k;
    f->jump_k = 0; // (the searches that fill the new table start without one)
    const sdsl_hip_status st = fm_build_jump_k_fresh(f, k);
    if (st != SDSL_HIP_OK)
    {
        f->d_jump = std::move(old);
        f->jump_k = old_k;
    }
    return st;
}
static sdsl_hip_status fm_build_jump_k_fresh(sdsl_hip_fm_s * f, uint32_t k)
{
    f->jump_k = 0;
    f->d_jump.release();
    const uint64_t sigma = f->sigma;
    if (k == 0 || sigma < 2)
        return SDSL_HIP_OK;
    uint64_t count = 1;
    for (uint32_t t = 0; t < k; ++t)
    {
        if (count > (UINT64_C(1) << 34) / sigma)
        {
            set_error("fm jump table: %u characters over an alphabet of %llu symbols is too deep", k,
                      (unsigned long long)sigma);
            return SDSL_HIP_ERR_INVALID;
        }
        count *= sigma;
    }
    DevBuf d_p, d_l, d_r;
    SH_TRY(d_p.alloc(count * k));
    SH_TRY(d_l.alloc(count * 8));
    SH_TRY(d_r.alloc(count * 8));
    SH_TRY(f->d_jump.alloc(count * 16));
    hipLaunchKernelGGL(k_fm_kmers, dim3(grid_for(count, 256, 65536)), dim3(256), 0, 0, f->d_tab.as<FmTables>(), (uint32_t)sigma,
                       k, count, d_p.as<uint8_t>());
    SH_HIP(hipGetLastError());
    SH_TRY(fm_run(f, d_p.as<uint8_t>(), k, nullptr, count * k, count, nullptr, d_l.as<uint64_t>(), d_r.as<uint64_t>(), nullptr,
                  false));
    hipLaunchKernelGGL(k_fm_pack_jump, dim3(grid_for(count, 256, 65536)), dim3(256), 0, 0, d_l.as<uint64_t>(),
                       d_r.as<uint64_t>(), count, f->d_jump.as<uint64_t>());
    SH_HIP(hipGetLastError());
    SH_HIP(hipDeviceSynchronize());
    f->jump_k = k;
    return SDSL_HIP_OK;
}

sdsl_hip_status sdsl_hip_fm_count_batch(sdsl_hip_fm_t fm, const uint8_t * patterns, uint32_t m,
                                        uint64_t n_patterns, uint64_t * out, void * stream)
{
    if (!fm || (n_patterns && (!out || (!patterns && m))))
    {
        set_error("fm_count_batch: invalid argument");
        return SDSL_HIP_ERR_INVALID;
    }
    if (m && n_patterns >= (UINT64_C(1) << 22) && !is_device_ptr(patterns) && !is_device_ptr(out))
    { // host arrays on both sides: 2^20 patterns at a time on several streams, so that uploads, searches and downloads
      // overlap (common.hpp host_pipeline_bytes); every piece is answered in suffix order on its own
        return host_pipeline_bytes(fm->device, patterns, m, (uint8_t *)out, 8, n_patterns, UINT64_C(1) << 20,
                                   [fm, m](const void * d_in, void * d_out, uint64_t cnt, hipStream_t st) -> sdsl_hip_status
                                   {
                                       return fm_run(fm, (const uint8_t *)d_in, m, nullptr, (uint64_t)m * cnt, cnt,
                                                     (uint64_t *)d_out, nullptr, nullptr, st);
                                   });
    }
    return fm_run(fm, patterns, m, nullptr, (uint64_t)m * n_patterns, n_patterns, out, nullptr, nullptr,
                  (hipStream_t)stream);
}

sdsl_hip_status sdsl_hip_fm_interval_batch(sdsl_hip_fm_t fm, const uint8_t * patterns, uint32_t m,
                                           uint64_t n_patterns, uint64_t * l_out, uint64_t * r_out, void * stream)
{
    if (!fm || (n_patterns && (!l_out || !r_out || (!patterns && m))))
    {
        set_error("fm_interval_batch: invalid argument");
        return SDSL_HIP_ERR_INVALID;
    }
    return fm_run(fm, patterns, m, nullptr, (uint64_t)m * n_patterns, n_patterns, nullptr, l_out, r_out,
                  (hipStream_t)stream);
}

sdsl_hip_status sdsl_hip_fm_count_ragged(sdsl_hip_fm_t fm, const uint8_t * bytes, const uint64_t * offsets,
                                         uint64_t n_patterns, uint64_t * out, void * stream)
{
    if (!fm || (n_patterns && (!out || !offsets)))
    {
        set_error("fm_count_ragged: invalid argument");
        return SDSL_HIP_ERR_INVALID;
    }
    if (n_patterns == 0)
        return SDSL_HIP_OK;
    // total byte count = offsets[n]; fetch it wherever the offsets live
    uint64_t total = 0;
    if (is_device_ptr(offsets))
        SH_HIP(hipMemcpy(&total, offsets + n_patterns, 8, hipMemcpyDeviceToHost));
    else
        total = offsets[n_patterns];
    return fm_run(fm, bytes, 0, offsets, total, n_patterns, out, nullptr, nullptr, (hipStream_t)stream);
}

sdsl_hip_status sdsl_hip_fm_backward_search_batch(sdsl_hip_fm_t fm, const uint64_t * l, const uint64_t * r,
                                                  const uint8_t * c, uint64_t n, uint64_t * l_out,
                                                  uint64_t * r_out, void * stream)
{
    if (!fm || (n && (!l || !r || !c || !l_out || !r_out)))
    {
        set_error("fm_backward_search_batch: invalid argument");
        return SDSL_HIP_ERR_INVALID;
    }
    hipStream_t s = (hipStream_t)stream;
    SH_HIP(hipSetDevice(fm->device));
    if (n == 0)
        return SDSL_HIP_OK;
    Staged sl, sr, sc, ol, orr;
    SH_TRY(sl.in(l, n * 8, s));
    SH_TRY(sr.in(r, n * 8, s));
    SH_TRY(sc.in(c, n, s));
    SH_TRY(ol.out(l_out, n * 8));
    SH_TRY(orr.out(r_out, n * 8));
    const WtHost & w = sdsl_hip_wt_host(fm->wt);
    {
        KernelTimer t(s);
        if (w.backend == 1)
            SH_TRY(fm_rrr_launch_backward_step(w, fm->d_tab.as<FmTables>(), fm->size, (const uint64_t *)sl.dev,
                                               (const uint64_t *)sr.dev, (const uint8_t *)sc.dev, n, (uint64_t *)ol.dev,
                                               (uint64_t *)orr.dev, s));
        else
            hipLaunchKernelGGL((k_fm_backward_step<false>), dim3(grid_for(n, kQPB, 256u * 8u)), dim3(kBlock), 0, s,
                               w.view(), fm->d_tab.as<FmTables>(), fm->size, (const uint64_t *)sl.dev,
                               (const uint64_t *)sr.dev, (const uint8_t *)sc.dev, n, (uint64_t *)ol.dev,
                               (uint64_t *)orr.dev);
    }
    SH_HIP(hipGetLastError());
    SH_TRY(ol.finish(s));
    SH_TRY(orr.finish(s));
    if (sl.host || sr.host || sc.host)
        SH_HIP(hipStreamSynchronize(s));
    return SDSL_HIP_OK;
}
}
