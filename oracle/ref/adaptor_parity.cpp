// adaptor_parity.cpp — compiles the header-only adaptors (include/sdsl_hip/adaptors.hpp) against the REAL
// sdsl-lite headers and checks, in one process, every batched answer of the HIP engine against the
// scalar answer of the unmodified reference object it was built from.  TEST INFRASTRUCTURE: built by
// oracle/Makefile into oracle/_ref/adaptor_parity (needs /root/reference at build time), run on the GPU
// box by tests/test_gpu_adaptors.py.  Exit code 0 = all comparisons equal.
#include <chrono>
#include <sdsl_hip/adaptors.hpp>

#include <cstdio>
#include <fstream>
#include <random>

using namespace sdsl;

static int g_fail = 0;
#define CHECK(cond, what)                                                                                          \
    do {                                                                                                           \
        if (getenv("PARITY_TRACE"))                                                                                \
            fprintf(stderr, "[parity] %s (line %d)\n", what, __LINE__);                                            \
        if (!(cond))                                                                                               \
        {                                                                                                          \
            ++g_fail;                                                                                              \
            printf("FAIL %s (%s:%d)\n", what, __FILE__, __LINE__);                                                 \
        }                                                                                                          \
    } while (0)

template <class t_rs, class t_hip>
static void check_rank(bit_vector const & bv, std::mt19937_64 & rng, char const * what)
{
    t_rs rs(&bv);
    t_hip hs(&bv);
    size_t n = 20000;
    std::vector<uint64_t> idx(n), out(n);
    for (auto & x : idx)
        x = rng() % (bv.size() + 1);
    idx[0] = 0;
    idx[1] = bv.size();
    hs.rank_batch(idx.data(), n, out.data());
    bool ok = true;
    for (size_t q = 0; q < n; ++q)
        ok &= out[q] == rs(idx[q]);
    CHECK(ok, what);
    CHECK(hs(idx[7]) == rs(idx[7]) and hs.size() == rs.size(), "scalar operator()");
    // serialised bytes are SDSL's own
    std::ostringstream a, b;
    rs.serialize(a);
    hs.serialize(b);
    CHECK(a.str() == b.str(), "rank serialize bytes");
    std::istringstream in(a.str());
    t_hip loaded;
    loaded.load(in, &bv);
    CHECK(loaded(idx[5]) == rs(idx[5]), "rank load");
}

template <class t_ss, class t_hip>
static void check_select(bit_vector const & bv, uint64_t args, std::mt19937_64 & rng, char const * what)
{
    if (!args)
        return;
    t_ss ss(&bv);
    t_hip hs(&bv);
    size_t n = 20000;
    std::vector<uint64_t> i(n), out(n);
    for (auto & x : i)
        x = 1 + rng() % args;
    i[0] = 1;
    i[1] = args;
    hs.select_batch(i.data(), n, out.data());
    bool ok = true;
    for (size_t q = 0; q < n; ++q)
        ok &= out[q] == ss(i[q]);
    CHECK(ok, what);
    std::ostringstream a, b;
    ss.serialize(a);
    hs.serialize(b);
    CHECK(a.str() == b.str(), "select serialize bytes");
}

int main(int argc, char ** argv)
{
    setvbuf(stderr, nullptr, _IONBF, 0);
    setvbuf(stdout, nullptr, _IOLBF, 0); // (piped into a test: a hang must show how far the run got)
    std::mt19937_64 rng(4242);
    for (uint64_t n : {1000ull, 100000ull, 1000003ull})
        for (int dens : {50, 3, 97})
        {
            bit_vector bv(n, 0);
            for (uint64_t i = 0; i < n; ++i)
                bv[i] = (rng() % 100) < (uint64_t)dens;
            uint64_t ones = util::cnt_one_bits(bv);
            check_rank<rank_support_v5<1>, rank_support_v5_hip<1>>(bv, rng, "rank_support_v5<1>");
            check_rank<rank_support_v5<0>, rank_support_v5_hip<0>>(bv, rng, "rank_support_v5<0>");
            check_rank<rank_support_v<1>, rank_support_v_hip<1>>(bv, rng, "rank_support_v<1>");
            check_rank<rank_support_v<0>, rank_support_v_hip<0>>(bv, rng, "rank_support_v<0>");
            check_select<select_support_mcl<1>, select_support_mcl_hip<1>>(bv, ones, rng, "select_support_mcl<1>");
            check_select<select_support_mcl<0>, select_support_mcl_hip<0>>(bv, n - ones, rng, "select_support_mcl<0>");
            // two-bit patterns (rank_support.hpp:160-284, select_support.hpp:206-409): occurrences end at position i
            {
                rank_support_v5<10, 2> c10(&bv);
                rank_support_v5<01, 2> c01(&bv);
                rank_support_v5<00, 2> c00(&bv);
                rank_support_v5<11, 2> c11(&bv);
                check_rank<rank_support_v5<10, 2>, rank_support_v5_hip<10, 2>>(bv, rng, "rank_support_v5<10,2>");
                check_rank<rank_support_v5<01, 2>, rank_support_v5_hip<01, 2>>(bv, rng, "rank_support_v5<01,2>");
                check_rank<rank_support_v<00, 2>, rank_support_v_hip<00, 2>>(bv, rng, "rank_support_v<00,2>");
                check_rank<rank_support_v<11, 2>, rank_support_v_hip<11, 2>>(bv, rng, "rank_support_v<11,2>");
                check_select<select_support_mcl<10, 2>, select_support_mcl_hip<10, 2>>(bv, c10(n), rng, "select_support_mcl<10,2>");
                check_select<select_support_mcl<01, 2>, select_support_mcl_hip<01, 2>>(bv, c01(n), rng, "select_support_mcl<01,2>");
                check_select<select_support_mcl<00, 2>, select_support_mcl_hip<00, 2>>(bv, c00(n), rng, "select_support_mcl<00,2>");
                check_select<select_support_mcl<11, 2>, select_support_mcl_hip<11, 2>>(bv, c11(n), rng, "select_support_mcl<11,2>");
            }
            // any other SDSL bit-vector type through its plain bits: bit_vector_il<512>, rrr_vector<15>
            if (n <= 100000)
            {
                bit_vector_il<512> il(bv);
                rrr_vector<15> r15(bv);
                bit_vector_il<512>::rank_1_type ilr(&il);
                rrr_vector<15>::select_1_type r15s(&r15);
                bit_vector from_il = to_bit_vector(il), from_r15 = to_bit_vector(r15);
                CHECK(from_il == bv and from_r15 == bv, "to_bit_vector(bit_vector_il / rrr_vector<15>)");
                rank_support_v5_hip<1> hr(&from_il);
                rrr_vector_hip dr(from_r15);
                select_support_rrr_hip<1> hsel(&dr);
                bool ok = true;
                for (int t = 0; t < 300; ++t)
                {
                    uint64_t x = rng() % (n + 1);
                    ok &= hr(x) == ilr(x);
                    if (ones)
                    {
                        uint64_t k = 1 + rng() % ones;
                        ok &= hsel(k) == r15s(k) and (t >= 20 or hsel.select_on_device(k) == r15s(k));
                    }
                }
                CHECK(ok, "rank on bit_vector_il<512>, select on rrr_vector<15> through their bits");
                // the named adaptors for them
                rank_support_il_hip<1, 512> ilh(&il);
                select_support_il_hip<0, 512> ils(&il);
                bit_vector_il<512>::select_0_type ils_ref(&il);
                rrr_vector_hip d15(r15);
                rank_support_rrr_hip<0> r15h(&d15);
                rrr_vector<15>::rank_0_type r15r(&r15);
                bool ok2 = true;
                for (int t = 0; t < 200; ++t)
                {
                    uint64_t x = rng() % (n + 1);
                    ok2 &= ilh(x) == ilr(x) and r15h(x) == r15r(x);
                    if (n - ones)
                    {
                        uint64_t k = 1 + rng() % (n - ones);
                        ok2 &= ils(k) == ils_ref(k);
                    }
                }
                CHECK(ok2, "rank_support_il_hip / select_support_il_hip / rrr_vector_hip(rrr_vector<15>)");
                // rrr_vector<15> decoded on the device (no host conversion), batch members included
                rank_support_rrr_bits_hip<1, rrr_vector<15>> r15dev(&r15);
                select_support_rrr_bits_hip<1, rrr_vector<15>> s15dev(&r15);
                rrr_vector<15>::rank_1_type r15r1(&r15);
                std::vector<uint64_t> qq(500), oo(500);
                for (auto & x : qq)
                    x = rng() % (n + 1);
                r15dev.rank_batch(qq.data(), qq.size(), oo.data());
                bool ok3 = r15dev.size() == n;
                for (size_t t = 0; t < qq.size(); ++t)
                    ok3 &= oo[t] == r15r1(qq[t]);
                if (ones)
                    for (int t = 0; t < 100; ++t)
                    {
                        uint64_t k = 1 + rng() % ones;
                        ok3 &= s15dev(k) == r15s(k);
                    }
                CHECK(ok3, "rank_support_rrr_bits_hip / select_support_rrr_bits_hip (device decode of rrr_vector<15>)");
            }
            // sd_vector<>
            {
                sd_vector<> sv(bv);
                sd_vector<>::rank_1_type sr1(&sv);
                sd_vector<>::rank_0_type sr0(&sv);
                sd_vector<>::select_1_type ss1(&sv);
                sd_vector<>::select_0_type ss0(&sv);
                sd_vector_hip dsv(sv), dsv2(bv);
                rank_support_sd_hip<1> hr1(&dsv);
                rank_support_sd_hip<0> hr0(&dsv2);
                select_support_sd_hip<1> hs1(&dsv2);
                select_support_sd_hip<0> hs0(&dsv);
                size_t q = 5000;
                std::vector<uint64_t> a(q), o(q);
                for (auto & x : a)
                    x = rng() % (n + 1);
                bool ok = true;
                hr1.rank_batch(a.data(), q, o.data());
                for (size_t k = 0; k < q; ++k)
                    ok &= o[k] == sr1(a[k]);
                hr0.rank_batch(a.data(), q, o.data());
                for (size_t k = 0; k < q; ++k)
                    ok &= o[k] == sr0(a[k]);
                CHECK(ok, "rank_support_sd<1> / <0>");
                if (ones)
                {
                    for (auto & x : a)
                        x = 1 + rng() % ones;
                    hs1.select_batch(a.data(), q, o.data());
                    for (size_t k = 0; k < q; ++k)
                        ok &= o[k] == ss1(a[k]);
                }
                if (n - ones)
                {
                    for (auto & x : a)
                        x = 1 + rng() % (n - ones);
                    hs0.select_batch(a.data(), 500, o.data());
                    for (size_t k = 0; k < 500; ++k)
                        ok &= o[k] == ss0(a[k]);
                }
                CHECK(ok, "select_support_sd<1> / <0>");
                CHECK(dsv[n / 2] == sv[n / 2] and dsv.access_on_device(n / 2) == sv[n / 2] and dsv.size() == sv.size(),
                      "sd_vector::operator[] (host) / access_on_device / size");
                // scalar members: the host object (dsv: the caller's sd_vector; dsv2, made from plain bits: an sd_vector<> loaded from
                // the device image's own bytes) — and the same queries through the device
                bool oksc = true;
                for (int t = 0; t < 40; ++t)
                {
                    uint64_t x = rng() % (n + 1);
                    oksc &= hr1(x) == sr1(x) and hr0(x) == sr0(x) and hr1.rank_on_device(x) == sr1(x) and hr0.rank_on_device(x) == sr0(x);
                    if (ones)
                    {
                        uint64_t k = 1 + rng() % ones;
                        oksc &= hs1(k) == ss1(k) and hs1.select_on_device(k) == ss1(k);
                    }
                    if (n - ones)
                    {
                        uint64_t k = 1 + rng() % (n - ones);
                        oksc &= hs0(k) == ss0(k) and hs0.select_on_device(k) == ss0(k);
                    }
                }
                CHECK(oksc, "sd supports: scalar calls (host) and *_on_device agree with sd_vector<>'s supports");
            }
            // rrr_vector<63>
            rrr_vector<63> rv(bv);
            rrr_vector<63>::rank_1_type r1(&rv);
            rrr_vector<63>::select_1_type s1(&rv);
            rrr_vector<63>::select_0_type s0(&rv);
            rrr_vector_hip dv(rv), dv2(bv);
            rank_support_rrr_hip<1> hr1(&dv);
            select_support_rrr_hip<1> hs1(&dv2);
            select_support_rrr_hip<0> hs0(&dv);
            size_t q = 5000;
            std::vector<uint64_t> a(q), o(q);
            for (auto & x : a)
                x = rng() % (n + 1);
            hr1.rank_batch(a.data(), q, o.data());
            bool ok = true;
            for (size_t k = 0; k < q; ++k)
                ok &= o[k] == r1(a[k]);
            CHECK(ok, "rank_support_rrr<1,63>");
            for (auto & x : a)
                x = 1 + rng() % (ones + 1); // includes the overflow value ones+1 -> size()
            hs1.select_batch(a.data(), q, o.data());
            ok = true;
            for (size_t k = 0; k < q; ++k)
                ok &= o[k] == s1(a[k]);
            CHECK(ok, "select_support_rrr<1,63>");
            for (auto & x : a)
                x = 1 + rng() % (n - ones + 1);
            hs0.select_batch(a.data(), q, o.data());
            ok = true;
            for (size_t k = 0; k < q; ++k)
                ok &= o[k] == s0(a[k]);
            CHECK(ok, "select_support_rrr<0,63>");
            CHECK(dv[n / 2] == rv[n / 2] and dv.access_on_device(n / 2) == rv[n / 2] and dv2[n / 2] == rv[n / 2], "rrr operator[] (host) / access_on_device");
            {
                bool oksc = true;
                for (int t = 0; t < 40; ++t)
                {
                    uint64_t x = rng() % (n + 1);
                    oksc &= hr1(x) == r1(x) and hr1.rank_on_device(x) == r1(x);
                    if (ones)
                    {
                        uint64_t k = 1 + rng() % ones;
                        oksc &= hs1(k) == s1(k) and hs1.select_on_device(k) == s1(k);
                    }
                    if (n - ones)
                    {
                        uint64_t k = 1 + rng() % (n - ones);
                        oksc &= hs0(k) == s0(k) and hs0.select_on_device(k) == s0(k);
                    }
                    if (n >= 64)
                    {
                        uint64_t y = rng() % (n - 63);
                        oksc &= dv.get_int(y, 64) == rv.get_int(y, 64) and dv.get_int_on_device(y, 37) == rv.get_int(y, 37) and dv2.get_int(y, 64) == rv.get_int(y, 64);
                    }
                }
                CHECK(oksc, "rrr supports: scalar calls (host: the caller's vector, or one loaded from the device image) and *_on_device");
            }
            // encoded on the GPU, loaded into an unmodified rrr_vector<63>
            {
                sdsl_hip_rrr_t h = nullptr;
                CHECK(sdsl_hip_rrr_create(bv.data(), bv.bit_size(), 0, &h) == SDSL_HIP_OK, "rrr_create");
                size_t len = 0;
                sdsl_hip_rrr_serialize(h, nullptr, 0, &len);
                std::string bytes(len, 0);
                CHECK(sdsl_hip_rrr_serialize(h, &bytes[0], len, &len) == SDSL_HIP_OK, "rrr_serialize");
                std::istringstream iss(bytes);
                std::istream & in = iss; // (an istringstream lvalue would select the cereal overload of load)
                rrr_vector<63> loaded;
                loaded.load(in);
                CHECK(loaded == rv, "GPU-encoded rrr_vector loads into SDSL and equals the CPU-built one");
                sdsl_hip_rrr_destroy(h);
            }
        }
    // wavelet tree + FM-index over a text file (argv[1]) or a built-in sample
    std::string text;
    if (argc > 1)
    {
        std::ifstream f(argv[1], std::ios::binary);
        text.assign(std::istreambuf_iterator<char>(f), std::istreambuf_iterator<char>());
    }
    if (text.empty())
        for (int i = 0; i < 3000; ++i)
            text += "she sells sea shells by the sea shore; ";
    typedef wt_huff<bit_vector, rank_support_v5<>> wt_t;
    typedef csa_wt<wt_t> csa_t;
    typedef csa_wt<wt_huff<bit_vector, rank_support_v5<>, select_support_scan<>, select_support_scan<0>>, 1 << 20, 1 << 20>
        fm_huff_t; // benchmark/indexing_count/index.config:8
    {
        std::vector<uint8_t> bytes(text.begin(), text.end());
        wt_t wt(bytes.begin(), bytes.end());
        wt_huff_hip dw(wt, true);
        size_t q = 20000;
        std::vector<uint64_t> i(q), o(q);
        std::vector<uint8_t> c(q);
        for (size_t k = 0; k < q; ++k)
        {
            i[k] = rng() % (bytes.size() + 1);
            c[k] = (k & 3) ? bytes[rng() % bytes.size()] : (uint8_t)(rng() & 0xFF);
        }
        dw.rank_batch(i.data(), c.data(), q, o.data());
        bool ok = true;
        for (size_t k = 0; k < q; ++k)
            ok &= o[k] == wt.rank(i[k], c[k]);
        CHECK(ok, "wt_huff::rank");
        uint64_t p = bytes.size() / 3;
        CHECK(dw[p] == wt[p] and dw.inverse_select(p) == wt.inverse_select(p), "wt_huff::operator[] / inverse_select (host)");
        CHECK(dw.select(1, bytes[p]) == wt.select(1, bytes[p]) and dw.rank(p, bytes[p]) == wt.rank(p, bytes[p]), "wt_huff::select / rank (host)");
        CHECK(dw.access_on_device(p) == wt[p] and dw.inverse_select_on_device(p) == wt.inverse_select(p) and
                  dw.select_on_device(1, bytes[p]) == wt.select(1, bytes[p]) and dw.rank_on_device(p, bytes[p]) == wt.rank(p, bytes[p]),
              "wt_huff: the same scalar queries through the device");
        // the compressed flavour through the same adaptor
        wt_huff<rrr_vector<63>> wr(bytes.begin(), bytes.end());
        wt_huff_hip dr(wr, SDSL_HIP_LAYOUT_RRR63);
        dr.rank_batch(i.data(), c.data(), q, o.data());
        bool okr = true;
        for (size_t k = 0; k < q; ++k)
            okr &= o[k] == wr.rank(i[k], c[k]);
        CHECK(okr, "wt_huff<rrr_vector<63>>::rank");
        CHECK(dr[p] == wr[p] and dr.inverse_select(p) == wr.inverse_select(p) and dr.access_on_device(p) == wr[p] and
                  dr.inverse_select_on_device(p) == wr.inverse_select(p),
              "wt_huff<rrr>::operator[] / inverse_select (host and device)");
        std::vector<uint64_t> si(q);
        std::vector<uint8_t> sc(q);
        for (size_t k = 0; k < q; ++k)
        {
            sc[k] = bytes[rng() % bytes.size()];
            si[k] = 1 + rng() % wr.rank(bytes.size(), sc[k]);
        }
        dr.select_batch(si.data(), sc.data(), q, o.data());
        for (size_t k = 0; k < q; ++k)
            okr &= o[k] == wr.select(si[k], sc[k]);
        CHECK(okr, "wt_huff<rrr_vector<63>>::select");
        // other byte shapes through the same adaptor: wt_blcd<> and wt_hutu<> (rank_support_v, mcl selects)
        wt_blcd<> wb(bytes.begin(), bytes.end());
        wt_hutu<> wh(bytes.begin(), bytes.end());
        wt_blcd_hip db(wb, SDSL_HIP_LAYOUT_BV_MCL);
        wt_hutu_hip dh(wh, SDSL_HIP_LAYOUT_BV_MCL);
        std::vector<uint64_t> o2(q);
        db.rank_batch(i.data(), c.data(), q, o.data());
        dh.rank_batch(i.data(), c.data(), q, o2.data());
        bool oks = true;
        for (size_t k = 0; k < q; ++k)
            oks &= o[k] == wb.rank(i[k], c[k]) and o2[k] == wh.rank(i[k], c[k]);
        CHECK(oks, "wt_blcd::rank / wt_hutu::rank");
        CHECK(db[p] == wb[p] and dh.inverse_select(p) == wh.inverse_select(p) and db.select(2, bytes[p]) == wb.select(2, bytes[p]),
              "wt_blcd / wt_hutu access, inverse_select, select (host)");
        CHECK(db.access_on_device(p) == wb[p] and dh.inverse_select_on_device(p) == wh.inverse_select(p) and
                  db.select_on_device(2, bytes[p]) == wb.select(2, bytes[p]),
              "wt_blcd / wt_hutu access, inverse_select, select (device)");
        // what a scalar call costs through the adaptors of the compressed / tree types: 10^6 wt.rank and 10^6
        // rank on an rrr_vector<63>, one at a time, against the unmodified objects — the adaptor forwards to them
        {
            const size_t nq = 1000000;
            std::vector<uint64_t> qi(nq);
            std::vector<uint8_t> qc(nq);
            for (size_t k = 0; k < nq; ++k)
            {
                qi[k] = rng() % (bytes.size() + 1);
                qc[k] = bytes[rng() % bytes.size()];
            }
            bit_vector big(1 << 26, 0);
            for (uint64_t x = 0; x < big.size(); x += 1 + rng() % 40)
                big[x] = 1;
            rrr_vector<63> brv(big);
            rrr_vector<63>::rank_1_type br1(&brv);
            rrr_vector_hip bdv(brv);
            rank_support_rrr_hip<1> bhr(&bdv);
            std::vector<uint64_t> qr(nq);
            for (auto & x : qr)
                x = rng() % (big.size() + 1);
            (void)dw.rank(qi[0], qc[0]);
            (void)bhr(qr[0]);
            double ns_wt = 1e30, ns_wt_ref = 1e30, ns_rrr = 1e30, ns_rrr_ref = 1e30;
            uint64_t s_a = 0, s_b = 0, s_c = 0, s_d = 0;
            for (int rep = 0; rep < 5; ++rep)
            { // (best of five passes each, interleaved: the box's CPUs are shared)
                auto t0 = std::chrono::steady_clock::now();
                s_a = 0;
                for (size_t k = 0; k < nq; ++k)
                    s_a += dw.rank(qi[k], qc[k]);
                ns_wt = std::min(ns_wt, std::chrono::duration<double, std::nano>(std::chrono::steady_clock::now() - t0).count() / nq);
                t0 = std::chrono::steady_clock::now();
                s_b = 0;
                for (size_t k = 0; k < nq; ++k)
                    s_b += wt.rank(qi[k], qc[k]);
                ns_wt_ref = std::min(ns_wt_ref, std::chrono::duration<double, std::nano>(std::chrono::steady_clock::now() - t0).count() / nq);
                t0 = std::chrono::steady_clock::now();
                s_c = 0;
                for (size_t k = 0; k < nq; ++k)
                    s_c += bhr(qr[k]);
                ns_rrr = std::min(ns_rrr, std::chrono::duration<double, std::nano>(std::chrono::steady_clock::now() - t0).count() / nq);
                t0 = std::chrono::steady_clock::now();
                s_d = 0;
                for (size_t k = 0; k < nq; ++k)
                    s_d += br1(qr[k]);
                ns_rrr_ref = std::min(ns_rrr_ref, std::chrono::duration<double, std::nano>(std::chrono::steady_clock::now() - t0).count() / nq);
            }
            auto t0 = std::chrono::steady_clock::now();
            uint64_t s_dev = 0;
            for (size_t k = 0; k < 2000; ++k)
                s_dev += dw.rank_on_device(qi[k], qc[k]);
            const double us_dev = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count() / 2000;
            CHECK(s_a == s_b and s_c == s_d, "scalar loops through wt_huff_hip / rank_support_rrr_hip sum like the reference objects");
            printf("scalar wt.rank(i, c), 10^6 calls: %.1f ns per call through wt_huff_hip (wt_huff<> itself: %.1f ns; through the device: %.2f us); "
                   "scalar rank on rrr_vector<63> (2^26 bits), 10^6 calls: %.1f ns through rank_support_rrr_hip (rank_support_rrr<1, 63> itself: %.1f ns)\n",
                   ns_wt, ns_wt_ref, us_dev, ns_rrr, ns_rrr_ref);
            CHECK(ns_wt <= 1.3 * ns_wt_ref + 5.0, "a scalar wt.rank loop through the adaptor runs within 1.3x of wt_huff<>");
            CHECK(ns_rrr <= 1.3 * ns_rrr_ref + 5.0, "a scalar rank loop through rank_support_rrr_hip runs within 1.3x of rank_support_rrr<1, 63>");
        }
    }
    {
        csa_t csa;
        construct_im(csa, text, 1);
        fm_huff_t fm;
        construct_im(fm, text, 1);
        csa_wt_hip d1(csa, true), d2(fm, false);
        size_t q = 5000;
        uint32_t m = 20;
        std::vector<uint8_t> pats(q * m);
        for (size_t k = 0; k < q; ++k)
        {
            size_t st = rng() % (text.size() - m);
            for (uint32_t j = 0; j < m; ++j)
                pats[k * m + j] = (uint8_t)text[st + j];
            if (k % 7 == 0)
                pats[k * m + (rng() % m)] ^= 0x55; // some misses
        }
        std::vector<uint64_t> o1(q), o2(q);
        count_batch(d1, pats.data(), m, q, o1.data());
        count_batch(d2, pats.data(), m, q, o2.data());
        bool ok = true;
        for (size_t k = 0; k < q; ++k)
        {
            uint64_t ref = count(csa, pats.begin() + k * m, pats.begin() + (k + 1) * m);
            ok &= o1[k] == ref and o2[k] == ref;
        }
        CHECK(ok, "count(csa_wt)");
        std::string und = "sea";
        CHECK(count(d1, und.begin(), und.end()) == count(csa, und.begin(), und.end()), "count single (host)");
        CHECK(count_on_device(d1, und.begin(), und.end()) == count(csa, und.begin(), und.end()), "count single (device)");
        CHECK(d1.size() == csa.size(), "csa size");
        // the rest of the API on the default-density index (samples travel with the adaptor)
        {
            size_t qq = 3000;
            std::vector<uint64_t> idx(qq), o(qq);
            for (auto & x : idx)
                x = rng() % csa.size();
            bool oks = true;
            d1.sa_batch(idx.data(), qq, o.data());
            for (size_t k = 0; k < qq; ++k)
                oks &= o[k] == csa[idx[k]];
            CHECK(oks, "csa_wt::operator[]");
            d1.isa_batch(idx.data(), qq, o.data());
            for (size_t k = 0; k < qq; ++k)
                oks &= o[k] == csa.isa[idx[k]];
            CHECK(oks, "csa.isa[]");
            d1.lf_batch(idx.data(), qq, o.data());
            for (size_t k = 0; k < qq; ++k)
                oks &= o[k] == csa.lf[idx[k]];
            d1.psi_batch(idx.data(), qq, o.data());
            for (size_t k = 0; k < qq; ++k)
                oks &= o[k] == csa.psi[idx[k]];
            CHECK(oks, "csa.lf[] / csa.psi[]");
            CHECK(d1[7] == csa[7] and d1.sa_on_device(7) == csa[7], "csa_wt_hip::operator[] (host) / sa_on_device");
            std::vector<uint64_t> off, pos;
            locate_batch(d1, pats.data(), m, 200, off, pos);
            bool okl = off.size() == 201;
            for (size_t k = 0; k < 200 && okl; ++k)
            {
                auto ref = locate(csa, pats.begin() + k * m, pats.begin() + (k + 1) * m);
                okl &= ref.size() == off[k + 1] - off[k];
                for (size_t t = 0; t < ref.size() && okl; ++t)
                    okl &= ref[t] == pos[off[k] + t];
            }
            CHECK(okl, "locate(csa_wt)");
            auto one = locate_on_device(d1, und.begin(), und.end());
            auto one_host = locate(d1, und.begin(), und.end());
            auto one_ref = locate(csa, und.begin(), und.end());
            CHECK(one.size() == one_ref.size() and std::equal(one.begin(), one.end(), one_ref.begin()), "locate single (device)");
            CHECK(one_host.size() == one_ref.size() and std::equal(one_host.begin(), one_host.end(), one_ref.begin()), "locate single (host)");
            CHECK(extract_on_device(d1, 10, 60) == extract(csa, 10, 60) and extract(d1, 10, 60) == extract(csa, 10, 60), "extract(csa_wt)");
            CHECK(extract_on_device(d1, csa.size() - 3, csa.size() - 1) == extract(csa, csa.size() - 3, csa.size() - 1),
                  "extract at the end (sentinel included)");
        }
        // the same device image with everything HBM offers (suffix array, text, k-mer table) and back at the host type's footprint
        {
            const uint64_t stream_bytes = size_in_bytes(csa), as_loaded = d1.device_bytes();
            d1.restore_suffix_array();
            CHECK(d1.device_bytes() > as_loaded + 4 * csa.size(), "restore_suffix_array keeps suffix array and text");
            std::vector<uint64_t> o4(q);
            count_batch(d1, pats.data(), m, q, o4.data());
            CHECK(o4 == o1, "count after restore_suffix_array");
            CHECK(extract_on_device(d1, 10, 60) == extract(csa, 10, 60), "extract from the resident text");
            // (what the image took as loaded from the stream is always reachable; on a text of a GiB the floor is 1.1 x stream_bytes,
            // on this small one the fixed tables weigh more)
            d1.set_footprint(as_loaded);
            CHECK(d1.device_bytes() <= as_loaded, "set_footprint");
            (void)stream_bytes;
            count_batch(d1, pats.data(), m, q, o4.data());
            CHECK(o4 == o1, "count at the reduced footprint");
            CHECK(d1.sa_on_device(7) == csa[7] and extract_on_device(d1, 10, 60) == extract(csa, 10, 60), "csa[i] / extract at the reduced footprint");
        }
        // SDSL's README index family: csa_wt<wt_huff<rrr_vector<63>>>
        {
            csa_wt<wt_huff<rrr_vector<63>>, 32, 64> crrr;
            construct_im(crrr, text, 1);
            csa_wt_hip d3(crrr, SDSL_HIP_LAYOUT_RRR63);
            std::vector<uint64_t> o3(q);
            count_batch(d3, pats.data(), m, q, o3.data());
            bool ok3 = true;
            for (size_t k = 0; k < q; ++k)
                ok3 &= o3[k] == count(crrr, pats.begin() + k * m, pats.begin() + (k + 1) * m);
            CHECK(ok3, "count(csa_wt<wt_huff<rrr_vector<63>>>)");
        }
        // FM-index built from the raw text on the GPU, loaded into the unmodified SDSL type: locate / extract work
        sdsl_hip_fm_t h = nullptr;
        CHECK(sdsl_hip_fm_create_from_text((const uint8_t *)text.data(), text.size(), 0, &h) == SDSL_HIP_OK, "fm_create_from_text");
        size_t len = 0;
        sdsl_hip_fm_serialize(h, 1 << 20, 1 << 20, nullptr, 0, &len);
        std::string bytes(len, 0);
        CHECK(sdsl_hip_fm_serialize(h, 1 << 20, 1 << 20, &bytes[0], len, &len) == SDSL_HIP_OK, "fm_serialize");
        std::istringstream iss(bytes);
        std::istream & in = iss;
        fm_huff_t loaded;
        loaded.load(in);
        CHECK(loaded == fm, "GPU-built csa_wt loads into SDSL and equals construct()'s result");
        auto occ1 = locate(loaded, und.begin(), und.end());
        auto occ2 = locate(fm, und.begin(), und.end());
        std::sort(occ1.begin(), occ1.end());
        std::sort(occ2.begin(), occ2.end());
        CHECK(occ1.size() == occ2.size() and std::equal(occ1.begin(), occ1.end(), occ2.begin()), "locate on the loaded index");
        CHECK(extract(loaded, 10, 40) == extract(fm, 10, 40), "extract on the loaded index");
        // ... and as SDSL's DEFAULT index type csa_wt<> (wt_huff<> with rank_support_v and select_support_mcl)
        {
            size_t len2 = 0;
            sdsl_hip_fm_serialize_ex(h, SDSL_HIP_LAYOUT_BV_DEFAULT, 32, 64, nullptr, 0, &len2);
            std::string b2(len2, 0);
            CHECK(sdsl_hip_fm_serialize_ex(h, SDSL_HIP_LAYOUT_BV_DEFAULT, 32, 64, &b2[0], len2, &len2) == SDSL_HIP_OK,
                  "fm_serialize_ex(default)");
            std::istringstream iss2(b2);
            std::istream & in2 = iss2;
            csa_wt<> def_loaded, def_ref;
            def_loaded.load(in2);
            construct_im(def_ref, text, 1);
            CHECK(def_loaded == def_ref, "GPU-built index loads as csa_wt<> and equals construct()'s result");
            CHECK(def_loaded[17] == def_ref[17] and def_loaded.wavelet_tree.select(3, 'e') == def_ref.wavelet_tree.select(3, 'e'),
                  "SA access and wt.select on the loaded default index");
        }
        sdsl_hip_fm_destroy(h);
        // an sd_vector built on the device loads into sd_vector<>
        {
            bit_vector sparse(500000, 0);
            for (int t = 0; t < 3000; ++t)
                sparse[rng() % sparse.size()] = 1;
            sdsl_hip_sd_t sh = nullptr;
            CHECK(sdsl_hip_sd_create(sparse.data(), sparse.bit_size(), 0, &sh) == SDSL_HIP_OK, "sd_create");
            size_t l3 = 0;
            sdsl_hip_sd_serialize(sh, nullptr, 0, &l3);
            std::string b3(l3, 0);
            CHECK(sdsl_hip_sd_serialize(sh, &b3[0], l3, &l3) == SDSL_HIP_OK, "sd_serialize");
            std::istringstream iss3(b3);
            std::istream & in3 = iss3;
            sd_vector<> sdl;
            sdl.load(in3);
            sd_vector<> sdr(sparse);
            CHECK(sdl == sdr, "GPU-built sd_vector loads into sd_vector<> and equals the host-built one");
            sdsl_hip_sd_destroy(sh);
        }
    }
    // what a scalar call costs: rs(i) one at a time — answered by the caller's own rank_support_v5, built on
    // first use — against the unmodified reference object and against one batch call
    {
        bit_vector bv(1 << 28, 0);
        for (uint64_t i = 0; i < bv.size(); i += 3)
            bv[i] = 1;
        rank_support_v5_hip<1> hr(&bv);
        rank_support_v5<1> r1(&bv);
        const size_t nq = 1000000;
        std::vector<uint64_t> q(nq), out(nq);
        for (auto & x : q)
            x = rng() % (bv.size() + 1);
        (void)hr(q[0]); // (builds the host support)
        // (best of five passes each, interleaved: the box's CPUs are shared and a single pass can be preempted)
        auto t0 = std::chrono::steady_clock::now();
        uint64_t sum = 0, want = 0;
        double ns = 1e30, ns_ref = 1e30;
        for (int rep = 0; rep < 5; ++rep)
        {
            t0 = std::chrono::steady_clock::now();
            sum = 0;
            for (size_t i = 0; i < nq; ++i)
                sum += hr(q[i]);
            ns = std::min(ns, std::chrono::duration<double, std::nano>(std::chrono::steady_clock::now() - t0).count() / nq);
            t0 = std::chrono::steady_clock::now();
            want = 0;
            for (size_t i = 0; i < nq; ++i)
                want += r1(q[i]);
            ns_ref = std::min(ns_ref, std::chrono::duration<double, std::nano>(std::chrono::steady_clock::now() - t0).count() / nq);
        }
        t0 = std::chrono::steady_clock::now();
        hr.rank_batch(q.data(), nq, out.data());
        double ns_b = std::chrono::duration<double, std::nano>(std::chrono::steady_clock::now() - t0).count() / nq;
        uint64_t got_b = 0;
        for (size_t i = 0; i < nq; ++i)
            got_b += out[i];
        CHECK(sum == want && got_b == want, "scalar operator() loop and batch agree with rank_support_v5");
        CHECK(ns < 2.0 * ns_ref + 20.0, "a scalar loop over the adaptor runs within 2x of the reference object");
        uint64_t dsum = 0;
        t0 = std::chrono::steady_clock::now();
        for (size_t i = 0; i < 2000; ++i)
            dsum += hr.rank_on_device(q[i]);
        double us_dev = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count() / 2000;
        uint64_t dwant = 0;
        for (size_t i = 0; i < 2000; ++i)
            dwant += r1(q[i]);
        CHECK(dsum == dwant, "rank_on_device");
        {
            double best = 1e30;
            for (int rep = 0; rep < 3; ++rep)
            {
                auto tf = std::chrono::steady_clock::now();
                volatile uint64_t sink = hip_detail::fingerprint(&bv).a;
                (void)sink;
                best = std::min(best, std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - tf).count());
            }
            printf("content fingerprint of the 2^28-bit vector (32 MiB, taken once per support construction): %.2f ms = %.1f GB/s\n", best,
                   (double)(bv.size() / 8) / best / 1e6);
        }
        printf("scalar rs(i), 10^6 calls on 2^28 bits: %.1f ns per call (rank_support_v5<1> itself: %.1f ns; through the device: %.2f us); "
               "the same queries as one rank_batch from host arrays: %.2f ns per query\n", ns, ns_ref, us_dev, ns_b);
        // supports of one vector share one device replica; a select support adds its directory to it
        select_support_mcl_hip<1> hs1(&bv);
        select_support_mcl_hip<0> hs0(&bv);
        rank_support_v5_hip<0> hr0(&bv);
        CHECK(hs1.device_handle() == hr.device_handle() && hs0.device_handle() == hr.device_handle() &&
                  hr0.device_handle() == hr.device_handle(),
              "rank and select supports of one bit_vector share one device replica");
        select_support_mcl<1> s1(&bv);
        select_support_mcl<0> s0(&bv);
        std::vector<uint64_t> a(5000), o1(5000), o0(5000);
        const uint64_t ones = r1(bv.size());
        for (auto & x : a)
            x = 1 + rng() % ones;
        hs1.select_batch(a.data(), a.size(), o1.data());
        bool ok = true;
        for (size_t i = 0; i < a.size(); ++i)
            ok &= o1[i] == s1(a[i]) && hs1(a[i]) == s1(a[i]);
        for (auto & x : a)
            x = 1 + rng() % (bv.size() - ones);
        hs0.select_batch(a.data(), a.size(), o0.data());
        for (size_t i = 0; i < a.size(); ++i)
            ok &= o0[i] == s0(a[i]);
        CHECK(ok, "select on the shared replica (directories added on demand)");
        // a support made after the vector changed must not see the old replica
        bv[1] = !bv[1];
        rank_support_v5_hip<1> hr2(&bv);
        rank_support_v5<1> r2(&bv);
        CHECK(hr2.device_handle() != hr.device_handle(), "a modified vector gets a fresh replica");
        uint64_t two[2] = {2, bv.size()}, got2[2];
        hr2.rank_batch(two, 2, got2);
        CHECK(got2[0] == r2(2) && got2[1] == r2(bv.size()) && hr2(2) == r2(2), "fresh replica answers for the modified vector");
    }
    // an UNMODIFIED SDSL container over the adaptors: wt_pc's level walk calls m_bv_rank(pos) once per level (wt_pc.hpp:371-399)
    {
        std::string text;
        for (int i = 0; i < 400000; ++i)
            text += (char)('a' + (rng() % 100 < 60 ? rng() % 4 : rng() % 26));
        wt_huff<bit_vector, rank_support_v5_hip<1>, select_support_mcl_hip<1>, select_support_mcl_hip<0>> wth;
        wt_huff<> wtr;
        construct_im(wth, text, 1);
        construct_im(wtr, text, 1);
        bool ok = wth.size() == wtr.size();
        auto t0 = std::chrono::steady_clock::now();
        uint64_t acc = 0;
        for (int i = 0; i < 200000; ++i)
        {
            uint64_t pos = rng() % (wtr.size() + 1);
            unsigned char c = (unsigned char)text[rng() % text.size()];
            acc += wth.rank(pos, c);
            ok &= wth.rank(pos, c) == wtr.rank(pos, c);
        }
        double ns = std::chrono::duration<double, std::nano>(std::chrono::steady_clock::now() - t0).count() / 400000;
        for (int i = 0; i < 2000; ++i)
        {
            unsigned char c = (unsigned char)text[rng() % text.size()];
            uint64_t occ = wtr.rank(wtr.size(), c);
            uint64_t k = 1 + rng() % occ;
            ok &= wth.select(k, c) == wtr.select(k, c) && wth[k % wtr.size()] == wtr[k % wtr.size()];
        }
        CHECK(ok, "wt_huff<bit_vector, rank_support_v5_hip, select_support_mcl_hip...> answers like wt_huff<>");
        printf("unmodified wt_huff over the hip supports: %.0f ns per rank(i, c) (scalar path = the caller's SDSL)\n", ns);
        (void)acc;
    }
    // several GPUs of one node (all the box has; a group of one still goes through the group driver and RCCL's setup), then the same
    // through a group of two members on device 0 over the copy transport: shards and pipelined pieces for G > 1 on a one-GPU box
    for (int pass = 0; pass < 2; ++pass)
    {
        int n_dev = sdsl_hip_device_count();
        std::vector<int32_t> devs;
        for (int d = 0; d < n_dev; ++d)
            devs.push_back(d);
        if (pass == 1)
        {
            devs.assign(2, 0);
            setenv("SDSL_HIP_GROUP_TRANSPORT", "copy", 1);
        }
        device_group grp(devs);
        unsetenv("SDSL_HIP_GROUP_TRANSPORT");
        bit_vector bv(3000017, 0);
        for (uint64_t i = 0; i < bv.size(); ++i)
            bv[i] = (rng() % 100) < 37;
        rank_support_v5<1> r1(&bv);
        rank_support_v5<0> r0(&bv);
        select_support_mcl<1> s1(&bv);
        bit_vector_multi_hip mv(bv, grp);
        const size_t nq = 700001;
        std::vector<uint64_t> q(nq), out(nq);
        for (auto & x : q)
            x = rng() % (bv.size() + 1);
        mv.rank_batch<1>(q.data(), nq, out.data(), 3);
        bool ok = true;
        for (size_t i = 0; i < nq; ++i)
            ok &= out[i] == r1(q[i]);
        CHECK(ok, "bit_vector_multi_hip rank_batch<1>");
        mv.rank_batch<0>(q.data(), nq, out.data(), 1);
        ok = true;
        for (size_t i = 0; i < nq; ++i)
            ok &= out[i] == r0(q[i]);
        CHECK(ok, "bit_vector_multi_hip rank_batch<0>");
        const uint64_t ones = r1(bv.size());
        for (auto & x : q)
            x = 1 + rng() % ones;
        mv.select_batch<1>(q.data(), nq, out.data(), 4);
        ok = true;
        for (size_t i = 0; i < nq; ++i)
            ok &= out[i] == s1(q[i]);
        CHECK(ok, "bit_vector_multi_hip select_batch<1>");
        // count over a group against sdsl::count on the host index
        std::string text;
        for (int i = 0; i < 300000; ++i)
            text.push_back((char)('a' + rng() % 7));
        csa_wt<wt_huff<bit_vector, rank_support_v5<>>> csa;
        construct_im(csa, text, 1);
        csa_wt_multi_hip mcsa((uint8_t const *)text.data(), text.size(), grp);
        const uint32_t m = 6;
        const size_t np = 50000;
        std::vector<uint8_t> pats(np * m);
        for (size_t i = 0; i < np; ++i)
        {
            size_t at = rng() % (text.size() - m);
            for (uint32_t k = 0; k < m; ++k)
                pats[i * m + k] = (uint8_t)text[at + k];
        }
        std::vector<uint64_t> cnt(np);
        mcsa.count_batch(pats.data(), m, np, cnt.data(), 2);
        ok = mcsa.size() == csa.size();
        size_t shown = 0;
        for (size_t i = 0; i < np; ++i)
        {
            const uint64_t want = count(csa, pats.begin() + i * m, pats.begin() + (i + 1) * m);
            if (cnt[i] != want && shown++ < 8) // (where and how a mismatch looks decides what to look for)
                fprintf(stderr, "count_batch over the group, pass %d (%zu members): pattern %zu got %llu want %llu\n", pass, devs.size(), i,
                        (unsigned long long)cnt[i], (unsigned long long)want);
            ok &= cnt[i] == want;
        }
        if (shown)
            fprintf(stderr, "count_batch over the group, pass %d: %zu of %zu patterns differ (sizes %llu / %llu)\n", pass, shown, np,
                    (unsigned long long)mcsa.size(), (unsigned long long)csa.size());
        CHECK(ok, "csa_wt_multi_hip count_batch");
    }
    printf(g_fail ? "adaptor parity: %d FAILED\n" : "adaptor parity: all equal\n", g_fail);
    return g_fail ? 1 : 0;
}
