// sdsl_ref.cpp — extern "C" wrappers around the REAL sdsl-lite headers (compiled from
// /root/reference/include where they lie; nothing of the reference is copied into this repo).
// Built by oracle/Makefile into oracle/_ref/libsdsl_ref.so.  TEST INFRASTRUCTURE ONLY: it pins the
// C restatement (oracle.c), generates the golden fixtures (tests/golden/make_golden.py) and serves
// as bench.py's cpu_baseline of kind "reference".  It is never loaded by the product library.
#include <atomic>
#include <sched.h>
#include <pthread.h>
#include <thread>
#include <random>
#include <sdsl/bit_vectors.hpp>
#include <sdsl/suffix_arrays.hpp>
#include <sdsl/wavelet_trees.hpp>

#include <chrono>
#include <cstdint>
#include <cstdlib>
#include <cstring>
#include <istream>
#include <sstream>
#include <streambuf>
#include <string>
#include <vector>

using namespace sdsl;

namespace {

template <class T>
void to_bytes(T const & x, uint8_t ** out, uint64_t * len)
{
    std::ostringstream os;
    x.serialize(os);
    std::string s = os.str();
    *len = s.size();
    *out = (uint8_t *)malloc(s.size() ? s.size() : 1);
    memcpy(*out, s.data(), s.size());
}

struct RefBv
{
    bit_vector bv;
    rank_support_v5<1> r1;
    rank_support_v5<0> r0;
    rank_support_v<1> rv1;
    rank_support_v<0> rv0;
    select_support_mcl<1> s1;
    select_support_mcl<0> s0;
};

typedef rrr_vector<63> rrr_t;
struct RefRrr
{
    rrr_t v;
    rrr_t::rank_1_type r1;
    rrr_t::rank_0_type r0;
    rrr_t::select_1_type s1;
    rrr_t::select_0_type s0;
};

typedef wt_huff<bit_vector, rank_support_v5<>> wt_t; // selects default to select_support_mcl<1>/<0>
typedef wt_huff<bit_vector, rank_support_v5<>, select_support_scan<>, select_support_scan<0>> wt_scan_t;
typedef wt_huff<rrr_vector<63>> wt_rrr_t;                    // rank/select_support_rrr by default
typedef csa_wt<wt_rrr_t, 32, 64> csa_rrr_t;
typedef csa_wt<wt_t> csa_t;                                   // t_dens 32, t_inv_dens 64
typedef csa_wt<wt_scan_t, 1 << 20, 1 << 20> csa_fmhuff_t;   // benchmark/indexing_count/index.config:8

struct RefWt
{
    wt_t wt;
    wt_scan_t wts;
};
struct RefCsa
{
    csa_t csa;
    csa_fmhuff_t csa2;
    bool have2 = false;
};

} // namespace

extern "C" {

void ref_free(void * p)
{
    free(p);
}

// ---------------- plain bit vector -------------------------------------------------------
void * ref_bv_create(const uint64_t * words, uint64_t n_bits)
{
    RefBv * h = new RefBv();
    h->bv = bit_vector(n_bits, 0);
    if (n_bits)
        memcpy(h->bv.data(), words, ((n_bits + 63) >> 6) * 8); // keeps stray bits of the last word
    h->r1 = rank_support_v5<1>(&h->bv);
    h->r0 = rank_support_v5<0>(&h->bv);
    h->rv1 = rank_support_v<1>(&h->bv);
    h->rv0 = rank_support_v<0>(&h->bv);
    h->s1 = select_support_mcl<1>(&h->bv);
    h->s0 = select_support_mcl<0>(&h->bv);
    return h;
}
// rank supports only (v5, both bit values): for vectors whose select supports would take minutes to build
void * ref_bv_create_rank(const uint64_t * words, uint64_t n_bits)
{
    RefBv * h = new RefBv();
    h->bv = bit_vector(n_bits, 0);
    if (n_bits)
        memcpy(h->bv.data(), words, ((n_bits + 63) >> 6) * 8);
    h->r1 = rank_support_v5<1>(&h->bv);
    h->r0 = rank_support_v5<0>(&h->bv);
    return h;
}
void ref_bv_destroy(void * p)
{
    delete (RefBv *)p;
}
void ref_bv_rank(void * p, int bit, const uint64_t * idx, uint64_t n, uint64_t * out)
{
    RefBv * h = (RefBv *)p;
    if (bit)
        for (uint64_t q = 0; q < n; ++q)
            out[q] = h->r1(idx[q]);
    else
        for (uint64_t q = 0; q < n; ++q)
            out[q] = h->r0(idx[q]);
}
// the same scalar loop on `threads` host threads, contiguous slices (queries are const-safe: SURVEY.md §8(b))
void ref_bv_rank_mt(void * p, int bit, const uint64_t * idx, uint64_t n, uint64_t * out, int threads)
{
    if (threads < 1)
        threads = 1;
    std::vector<std::thread> th;
    for (int t = 0; t < threads; ++t)
    {
        uint64_t lo = n * (uint64_t)t / threads, hi = n * (uint64_t)(t + 1) / threads;
        th.emplace_back([=] { ref_bv_rank(p, bit, idx + lo, hi - lo, out + lo); });
    }
    for (auto & x : th)
        x.join();
}
// All host cores, measured the way a throughput number should be: every thread pinned to one CPU of the process's
// affinity mask, the output pages touched beforehand, the threads released together and the clock stopped when the last
// one is done (thread creation and first-touch page faults are outside), `reps` passes over the slice so that a thread
// runs long enough.  Returns the seconds of the slowest thread.
double ref_bv_rank_mt_timed(void * p, int bit, const uint64_t * idx, uint64_t n, uint64_t * out, int threads, int reps)
{
    if (threads < 1)
        threads = 1;
    if (reps < 1)
        reps = 1;
    cpu_set_t allowed;
    CPU_ZERO(&allowed);
    sched_getaffinity(0, sizeof(allowed), &allowed);
    std::vector<int> cpus;
    for (int c = 0; c < CPU_SETSIZE; ++c)
        if (CPU_ISSET(c, &allowed))
            cpus.push_back(c);
    std::atomic<int> ready{0};
    std::atomic<bool> go{false};
    std::vector<double> secs(threads, 0.0);
    std::vector<std::thread> th;
    for (int t = 0; t < threads; ++t)
        th.emplace_back(
            [&, t]
            {
                if (!cpus.empty())
                {
                    cpu_set_t one;
                    CPU_ZERO(&one);
                    CPU_SET(cpus[t % cpus.size()], &one);
                    pthread_setaffinity_np(pthread_self(), sizeof(one), &one);
                }
                const uint64_t lo = n * (uint64_t)t / threads, hi = n * (uint64_t)(t + 1) / threads;
                for (uint64_t q = lo; q < hi; q += 512)
                    out[q] = 0; // first touch
                ready.fetch_add(1);
                while (!go.load(std::memory_order_acquire))
                    ;
                auto t0 = std::chrono::steady_clock::now();
                for (int r = 0; r < reps; ++r)
                    ref_bv_rank(p, bit, idx + lo, hi - lo, out + lo);
                secs[t] = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
            });
    while (ready.load() < threads)
        std::this_thread::yield();
    go.store(true, std::memory_order_release);
    double worst = 0;
    for (int t = 0; t < threads; ++t)
    {
        th[t].join();
        worst = std::max(worst, secs[t]);
    }
    return worst;
}
void ref_bv_rank_v(void * p, const uint64_t * idx, uint64_t n, uint64_t * out)
{
    RefBv * h = (RefBv *)p;
    for (uint64_t q = 0; q < n; ++q)
        out[q] = h->rv1(idx[q]);
}
void ref_bv_select(void * p, int bit, const uint64_t * i, uint64_t n, uint64_t * out)
{
    RefBv * h = (RefBv *)p;
    if (bit)
        for (uint64_t q = 0; q < n; ++q)
            out[q] = h->s1(i[q]);
    else
        for (uint64_t q = 0; q < n; ++q)
            out[q] = h->s0(i[q]);
}
// which: 0 bit_vector, 1 rank_support_v5<1>, 2 rank_support_v5<0>, 3 select_support_mcl<1>, 4 select_support_mcl<0>,
// 5 rank_support_v<1>, 6 rank_support_v<0>
void ref_bv_serialize(void * p, int which, uint8_t ** out, uint64_t * len)
{
    RefBv * h = (RefBv *)p;
    switch (which)
    {
    case 0: to_bytes(h->bv, out, len); break;
    case 1: to_bytes(h->r1, out, len); break;
    case 2: to_bytes(h->r0, out, len); break;
    case 3: to_bytes(h->s1, out, len); break;
    case 4: to_bytes(h->s0, out, len); break;
    case 5: to_bytes(h->rv1, out, len); break;
    default: to_bytes(h->rv0, out, len); break;
    }
}

// ---------------- rrr_vector<63> -----------------------------------------------------------
void * ref_rrr_create(const uint64_t * words, uint64_t n_bits)
{
    bit_vector bv(n_bits, 0);
    if (n_bits)
        memcpy(bv.data(), words, ((n_bits + 63) >> 6) * 8);
    if (n_bits & 63) // rrr_vector reads whole blocks through get_int: keep the padding clean like SDSL users do
        bv.data()[(n_bits - 1) >> 6] &= bits::lo_set[n_bits & 63];
    RefRrr * h = new RefRrr();
    h->v = rrr_t(bv);
    h->r1 = rrr_t::rank_1_type(&h->v);
    h->r0 = rrr_t::rank_0_type(&h->v);
    h->s1 = rrr_t::select_1_type(&h->v);
    h->s0 = rrr_t::select_0_type(&h->v);
    return h;
}
void ref_rrr_destroy(void * p)
{
    delete (RefRrr *)p;
}
void ref_rrr_rank(void * p, int bit, const uint64_t * i, uint64_t n, uint64_t * out)
{
    RefRrr * h = (RefRrr *)p;
    if (bit)
        for (uint64_t q = 0; q < n; ++q)
            out[q] = h->r1(i[q]);
    else
        for (uint64_t q = 0; q < n; ++q)
            out[q] = h->r0(i[q]);
}
void ref_rrr_select(void * p, int bit, const uint64_t * i, uint64_t n, uint64_t * out)
{
    RefRrr * h = (RefRrr *)p;
    if (bit)
        for (uint64_t q = 0; q < n; ++q)
            out[q] = h->s1(i[q]);
    else
        for (uint64_t q = 0; q < n; ++q)
            out[q] = h->s0(i[q]);
}
void ref_rrr_access(void * p, const uint64_t * i, uint64_t n, uint8_t * out)
{
    RefRrr * h = (RefRrr *)p;
    for (uint64_t q = 0; q < n; ++q)
        out[q] = h->v[i[q]];
}
void ref_rrr_get_int(void * p, const uint64_t * i, uint32_t len, uint64_t n, uint64_t * out)
{
    RefRrr * h = (RefRrr *)p;
    for (uint64_t q = 0; q < n; ++q)
        out[q] = h->v.get_int(i[q], (uint8_t)len);
}
void ref_rrr_serialize(void * p, uint8_t ** out, uint64_t * len)
{
    to_bytes(((RefRrr *)p)->v, out, len);
}

// ---------------- wt_huff<bit_vector, rank_support_v5<>> -----------------------------------
void * ref_wt_create(const uint8_t * text, uint64_t n)
{
    RefWt * h = new RefWt();
    h->wt = wt_t(text, text + n);       // unsigned bytes (wt_helper.hpp:50-57 pitfall in SURVEY §3.4)
    h->wts = wt_scan_t(text, text + n);
    return h;
}
void ref_wt_destroy(void * p)
{
    delete (RefWt *)p;
}
uint64_t ref_wt_size(void * p)
{
    return ((RefWt *)p)->wt.size();
}
uint64_t ref_wt_sigma(void * p)
{
    return ((RefWt *)p)->wt.sigma;
}
uint64_t ref_wt_bv_size(void * p)
{
    return ((RefWt *)p)->wt.bv.size();
}
void ref_wt_rank(void * p, const uint64_t * i, const uint8_t * c, uint64_t n, uint64_t * out)
{
    RefWt * h = (RefWt *)p;
    for (uint64_t q = 0; q < n; ++q)
        out[q] = h->wt.rank(i[q], c[q]);
}
void ref_wt_access(void * p, const uint64_t * i, uint64_t n, uint8_t * out)
{
    RefWt * h = (RefWt *)p;
    for (uint64_t q = 0; q < n; ++q)
        out[q] = h->wt[i[q]];
}
void ref_wt_inverse_select(void * p, const uint64_t * i, uint64_t n, uint64_t * out_rank, uint8_t * out_c)
{
    RefWt * h = (RefWt *)p;
    for (uint64_t q = 0; q < n; ++q)
    {
        auto r = h->wt.inverse_select(i[q]);
        out_rank[q] = r.first;
        out_c[q] = r.second;
    }
}
void ref_wt_select(void * p, const uint64_t * i, const uint8_t * c, uint64_t n, uint64_t * out)
{
    RefWt * h = (RefWt *)p;
    for (uint64_t q = 0; q < n; ++q)
        out[q] = h->wt.select(i[q], c[q]);
}
void ref_wt_serialize(void * p, int select_is_mcl, uint8_t ** out, uint64_t * len)
{
    RefWt * h = (RefWt *)p;
    if (select_is_mcl)
        to_bytes(h->wt, out, len);
    else
        to_bytes(h->wts, out, len);
}

// ---------------- csa_wt<wt_huff<...>> -----------------------------------------------------
void * ref_csa_create(const uint8_t * text, uint64_t n, int also_fm_huff)
{
    RefCsa * h = new RefCsa();
    std::string s((const char *)text, n);
    try
    {
        construct_im(h->csa, s, 1);
        if (also_fm_huff)
        {
            construct_im(h->csa2, s, 1);
            h->have2 = true;
        }
    }
    catch (std::exception const &)
    { // e.g. construct.hpp:41 "contains zero symbol"
        delete h;
        return nullptr;
    }
    return h;
}
// csa_wt<wt_huff<bit_vector, rank_support_v5<>>> (32 / 64) from its serialised bytes — SDSL's own, or the ones the GPU
// engine writes for an index it built: the unmodified library then answers on it (bench.py times it as the CPU baseline)
void * ref_csa_load(const uint8_t * bytes, uint64_t len)
{
    struct membuf : std::streambuf
    {
        membuf(char * b, char * e)
        {
            setg(b, b, e);
        }
    };
    RefCsa * h = new RefCsa();
    try
    {
        membuf mb((char *)bytes, (char *)bytes + len);
        std::istream is(&mb);
        h->csa.load(is);
        if (!is)
        {
            delete h;
            return nullptr;
        }
    }
    catch (std::exception const &)
    {
        delete h;
        return nullptr;
    }
    return h;
}
void ref_csa_destroy(void * p)
{
    delete (RefCsa *)p;
}
uint64_t ref_csa_size(void * p)
{
    return ((RefCsa *)p)->csa.size();
}
uint64_t ref_csa_sigma(void * p)
{
    return ((RefCsa *)p)->csa.sigma;
}
void ref_csa_bwt(void * p, uint8_t * out)
{
    RefCsa * h = (RefCsa *)p;
    for (uint64_t i = 0; i < h->csa.size(); ++i)
        out[i] = h->csa.bwt[i];
}
void ref_csa_alphabet(void * p, uint8_t * char2comp, uint64_t * C)
{
    RefCsa * h = (RefCsa *)p;
    for (int c = 0; c < 256; ++c)
        char2comp[c] = h->csa.char2comp[c];
    for (uint64_t i = 0; i <= h->csa.sigma; ++i)
        C[i] = h->csa.C[i];
}
void ref_csa_count(void * p, const uint8_t * pats, uint32_t m, uint64_t n_pat, uint64_t * out)
{
    RefCsa * h = (RefCsa *)p;
    for (uint64_t q = 0; q < n_pat; ++q)
    {
        const uint8_t * b = pats + q * (uint64_t)m;
        out[q] = count(h->csa, b, b + m);
    }
}
void ref_csa_count_ragged(void * p, const uint8_t * bytes, const uint64_t * offs, uint64_t n_pat, uint64_t * out)
{
    RefCsa * h = (RefCsa *)p;
    for (uint64_t q = 0; q < n_pat; ++q)
        out[q] = count(h->csa, bytes + offs[q], bytes + offs[q + 1]);
}
void ref_csa_interval(void * p, const uint8_t * pats, uint32_t m, uint64_t n_pat, uint64_t * l_out, uint64_t * r_out)
{
    RefCsa * h = (RefCsa *)p;
    for (uint64_t q = 0; q < n_pat; ++q)
    {
        const uint8_t * b = pats + q * (uint64_t)m;
        uint64_t l, r;
        backward_search(h->csa, 0, h->csa.size() - 1, b, b + m, l, r);
        l_out[q] = l;
        r_out[q] = r;
    }
}
void ref_csa_backward_search(void * p, const uint64_t * l, const uint64_t * r, const uint8_t * c, uint64_t n,
                             uint64_t * l_out, uint64_t * r_out)
{
    RefCsa * h = (RefCsa *)p;
    for (uint64_t q = 0; q < n; ++q)
    {
        uint64_t a, b;
        backward_search(h->csa, l[q], r[q], (csa_t::char_type)c[q], a, b);
        l_out[q] = a;
        r_out[q] = b;
    }
}
// the rest of the csa_wt API on the default-density index csa_wt<wt_huff<bit_vector, rank_support_v5<>>, 32, 64>
// what: 0 = csa[i], 1 = csa.isa[i], 2 = csa.lf[i], 3 = csa.psi[i]
void ref_csa_access(void * p, int what, const uint64_t * idx, uint64_t n, uint64_t * out)
{
    RefCsa * h = (RefCsa *)p;
    for (uint64_t q = 0; q < n; ++q)
        out[q] = what == 0 ? h->csa[idx[q]]
                           : (what == 1 ? h->csa.isa[idx[q]] : (what == 2 ? h->csa.lf[idx[q]] : h->csa.psi[idx[q]]));
}
uint64_t ref_csa_extract(void * p, uint64_t begin, uint64_t end, uint8_t * out)
{
    RefCsa * h = (RefCsa *)p;
    return extract(h->csa, begin, end, out);
}
uint64_t ref_csa_locate(void * p, const uint8_t * pat, uint64_t m, uint64_t * out, uint64_t cap)
{
    RefCsa * h = (RefCsa *)p;
    auto occ = locate(h->csa, pat, pat + m);
    for (uint64_t i = 0; i < occ.size() && i < cap; ++i)
        out[i] = occ[i];
    return occ.size();
}
// which: 0 = csa_wt<wt_huff<bit_vector,rank_support_v5<>>>, 1 = the FM_HUFF type of the count benchmark
void ref_csa_serialize(void * p, int which, uint8_t ** out, uint64_t * len)
{
    RefCsa * h = (RefCsa *)p;
    if (which == 0)
        to_bytes(h->csa, out, len);
    else
        to_bytes(h->csa2, out, len);
}
void ref_csa_wt_rank(void * p, const uint64_t * i, const uint8_t * c, uint64_t n, uint64_t * out)
{
    RefCsa * h = (RefCsa *)p;
    for (uint64_t q = 0; q < n; ++q)
        out[q] = h->csa.wavelet_tree.rank(i[q], c[q]);
}

// ---------------- wt_huff<rrr_vector<63>> and csa_wt over it: serialised bytes only (answers equal the plain tree's)
void ref_wt_rrr_serialize(const uint8_t * text, uint64_t n, uint8_t ** out, uint64_t * len)
{
    wt_rrr_t wt(text, text + n);
    to_bytes(wt, out, len);
}
int ref_csa_rrr_serialize(const uint8_t * text, uint64_t n, uint8_t ** out, uint64_t * len)
{
    csa_rrr_t csa;
    std::string s((const char *)text, n);
    try
    {
        construct_im(csa, s, 1);
    }
    catch (std::exception const &)
    {
        return 1;
    }
    to_bytes(csa, out, len);
    return 0;
}

// two-bit pattern supports on a plain bit vector: pat 0 = <10,2>, 1 = <01,2>, 2 = <00,2>, 3 = <11,2>;
// which 0 = rank_support_v5, 1 = rank_support_v, 2 = select_support_mcl (arguments 1-based).  Returns the number of
// arguments the select support counts (0 for rank).
} // extern "C"
template <uint8_t t_b>
static uint64_t pattern_run(bit_vector const & bv, int which, const uint64_t * q, uint64_t n, uint64_t * out)
{
    if (which == 0)
    {
        rank_support_v5<t_b, 2> r(&bv);
        for (uint64_t k = 0; k < n; ++k)
            out[k] = r(q[k]);
    }
    else if (which == 1)
    {
        rank_support_v<t_b, 2> r(&bv);
        for (uint64_t k = 0; k < n; ++k)
            out[k] = r(q[k]);
    }
    else
    {
        select_support_mcl<t_b, 2> sl(&bv);
        for (uint64_t k = 0; k < n; ++k)
            out[k] = sl(q[k]);
    }
    return 0;
}
extern "C" {
void ref_bv_pattern(const uint64_t * words, uint64_t n_bits, int pat, int which, const uint64_t * q, uint64_t n,
                    uint64_t * out)
{
    bit_vector bv(n_bits, 0);
    memcpy(bv.data(), words, ((n_bits + 63) >> 6) * 8);
    switch (pat)
    {
    case 0: pattern_run<10>(bv, which, q, n, out); break;
    case 1: pattern_run<01>(bv, which, q, n, out); break;
    case 2: pattern_run<00>(bv, which, q, n, out); break;
    default: pattern_run<11>(bv, which, q, n, out); break;
    }
}

// ---------------- sd_vector<> ---------------------------------------------------------------
struct RefSd
{
    sd_vector<> v;
    sd_vector<>::rank_1_type r1;
    sd_vector<>::rank_0_type r0;
    sd_vector<>::select_1_type s1;
    sd_vector<>::select_0_type s0;
    void init()
    {
        r1.set_vector(&v);
        r0.set_vector(&v);
        s1.set_vector(&v);
        s0.set_vector(&v);
    }
};
void * ref_sd_create(const uint64_t * words, uint64_t n_bits)
{
    bit_vector bv(n_bits, 0);
    memcpy(bv.data(), words, ((n_bits + 63) >> 6) * 8);
    RefSd * h = new RefSd();
    h->v = sd_vector<>(bv);
    h->init();
    return h;
}
void * ref_sd_create_from_positions(const uint64_t * pos, uint64_t m)
{
    RefSd * h = new RefSd();
    std::vector<uint64_t> pv(pos, pos + m); // (the constructor finds is_sorted through ADL on the iterator type)
    h->v = sd_vector<>(pv.begin(), pv.end());
    h->init();
    return h;
}
void ref_sd_destroy(void * p)
{
    delete (RefSd *)p;
}
uint64_t ref_sd_size(void * p)
{
    return ((RefSd *)p)->v.size();
}
// what: 0 = rank_0, 1 = rank_1, 2 = select_0, 3 = select_1, 4 = operator[]
void ref_sd_query(void * p, int what, const uint64_t * q, uint64_t n, uint64_t * out)
{
    RefSd * h = (RefSd *)p;
    for (uint64_t k = 0; k < n; ++k)
        out[k] = what == 0 ? h->r0(q[k])
                           : (what == 1 ? h->r1(q[k]) : (what == 2 ? h->s0(q[k]) : (what == 3 ? h->s1(q[k]) : (uint64_t)h->v[q[k]])));
}
void ref_sd_serialize(void * p, uint8_t ** out, uint64_t * len)
{
    to_bytes(((RefSd *)p)->v, out, len);
}

// SDSL's default types: wt_huff<> (rank_support_v, select_support_mcl) and csa_wt<> over it (32 / 64)
void ref_wt_default_serialize(const uint8_t * text, uint64_t n, uint8_t ** out, uint64_t * len)
{
    to_bytes(wt_huff<>(text, text + n), out, len);
}
int ref_csa_default_serialize(const uint8_t * text, uint64_t n, uint8_t ** out, uint64_t * len)
{
    csa_wt<> csa;
    std::string s((const char *)text, n);
    try
    {
        construct_im(csa, s, 1);
    }
    catch (std::exception const &)
    {
        return 1;
    }
    to_bytes(csa, out, len);
    return 0;
}

// other wt_pc shapes over bytes: shape 1 = wt_blcd (balanced), 2 = wt_hutu (Hu-Tucker); flavour 0 = the type with
// its default template arguments (rank_support_v, select_support_mcl), 1 = <bit_vector, rank_support_v5<>,
// select_support_scan<>, select_support_scan<0>> (what sdsl_hip_wt_serialize writes)
void ref_wt_shape_serialize(const uint8_t * text, uint64_t n, int shape, int flavour, uint8_t ** out, uint64_t * len)
{
    if (shape == 1 && flavour == 0)
        to_bytes(wt_blcd<>(text, text + n), out, len);
    else if (shape == 1)
        to_bytes(wt_blcd<bit_vector, rank_support_v5<>, select_support_scan<>, select_support_scan<0>>(text, text + n), out,
                 len);
    else if (shape == 2 && flavour == 0)
        to_bytes(wt_hutu<>(text, text + n), out, len);
    else
        to_bytes(wt_hutu<bit_vector, rank_support_v5<>, select_support_scan<>, select_support_scan<0>>(text, text + n), out,
                 len);
}
// csa_wt<wt_blcd<bit_vector, rank_support_v5<>, scan, scan>, 32, 64>: the balanced-tree FM-index as the device writes it
int ref_csa_blcd_serialize(const uint8_t * text, uint64_t n, uint8_t ** out, uint64_t * len)
{
    csa_wt<wt_blcd<bit_vector, rank_support_v5<>, select_support_scan<>, select_support_scan<0>>, 32, 64> csa;
    std::string s((const char *)text, n);
    try
    {
        construct_im(csa, s, 1);
    }
    catch (std::exception const &)
    {
        return 1;
    }
    to_bytes(csa, out, len);
    return 0;
}

void ref_set_random_bits(uint64_t * words, uint64_t n_bits, int seed)
{
    bit_vector bv(n_bits, 0);
    util::set_random_bits(bv, seed);
    memcpy(words, bv.data(), ((n_bits + 63) >> 6) * 8);
}

// SURVEY.md 8(d) configs[2] vector with the STANDARD LIBRARY's generator (the product restates MT19937-64 on its own,
// csrc/workload.cpp): bit i = (i-th output of std::mt19937_64(seed) % 100 < percent), drawn sequentially
void ref_density_bits(uint64_t * words, uint64_t n_bits, uint64_t seed, uint32_t percent)
{
    std::mt19937_64 rng(seed);
    for (uint64_t w = 0; w < (n_bits + 63) / 64; ++w)
    {
        uint64_t x = 0;
        const unsigned nb = (w + 1) * 64 <= n_bits ? 64u : (unsigned)(n_bits & 63);
        for (unsigned b = 0; b < nb; ++b)
            x |= (uint64_t)(rng() % 100 < percent) << b;
        words[w] = x;
    }
}

// the serialised bytes of the sibling representations of bit_vectors.hpp over the same bits (in THIS translation unit
// rrr_vector<15> is the generic template; the specialisation lives in sdsl_ref_r15.cpp)
void ref_sibling_serialize(const uint64_t * words, uint64_t n_bits, int kind, uint8_t ** out, uint64_t * len)
{
    bit_vector bv(n_bits, 0);
    if (n_bits)
        memcpy(bv.data(), words, ((n_bits + 63) >> 6) * 8);
    if (n_bits & 63)
        bv.data()[n_bits >> 6] &= bits::lo_set[n_bits & 63]; // these constructors read whole words
    if (kind == 0)
        to_bytes(bit_vector_il<512>(bv), out, len);
    else if (kind == 1)
        to_bytes(rrr_vector<15>(bv), out, len);
    else if (kind == 2)
        to_bytes(bit_vector_il<64>(bv), out, len);
    else if (kind == 3)
        to_bytes(rrr_vector<15, int_vector<>, 8>(bv), out, len);
    else if (kind == 4)
        to_bytes(rrr_vector<31>(bv), out, len);
    else
        to_bytes(rrr_vector<62, int_vector<>, 16>(bv), out, len);
}

uint32_t ref_bits_sel(uint64_t x, uint32_t i)
{
    return bits::sel(x, i);
}
uint32_t ref_bits_hi(uint64_t x)
{
    return bits::hi(x);
}
}
