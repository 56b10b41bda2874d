// group_stress.cpp — hunts the intermittent mismatch of count() over a device group seen in oracle/ref/adaptor_parity.cpp
// (csa_wt_multi_hip count_batch), with nothing but the C ABI and the SYSTEM HIP runtime (the Python tests run on the runtime
// torch ships).  Phases as tools/group_stress.py:  A fixed replicas / B replicas rebuilt every round / C single handle rebuilt.
// build: g++ -O2 -std=c++17 tools/cpp/group_stress.cpp -Iinclude -Lsdsl-lite_amd/lib -lsdsl_hip -Wl,-rpath,$PWD/sdsl-lite_amd/lib -o sdsl-lite_amd/lib/group_stress
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <random>
#include <string>
#include <vector>

#include "sdsl_hip.h"

#define OK(x)                                                                                                        \
    do {                                                                                                             \
        if ((x) != SDSL_HIP_OK)                                                                                      \
        {                                                                                                            \
            fprintf(stderr, "%s failed: %s\n", #x, sdsl_hip_last_error());                                           \
            return 2;                                                                                                \
        }                                                                                                            \
    } while (0)

int main(int argc, char ** argv)
{
    const int rounds = argc > 1 ? atoi(argv[1]) : 150;
    const std::string transport = argc > 2 ? argv[2] : "copy2";
    const std::string phases = argc > 3 ? argv[3] : "ABC";
    std::mt19937_64 rng(3);
    std::vector<uint8_t> text(300000);
    for (auto & c : text)
        c = (uint8_t)('a' + rng() % 7);
    const uint32_t m = 6;
    const size_t np = 50000;
    std::vector<uint8_t> pats(np * m);
    for (size_t i = 0; i < np; ++i)
    {
        size_t at = rng() % (text.size() - m);
        memcpy(&pats[i * m], &text[at], m);
    }
    sdsl_hip_fm_t ref = nullptr;
    OK(sdsl_hip_fm_create_from_text(text.data(), text.size(), 0, &ref));
    std::vector<uint64_t> want(np), got(np);
    OK(sdsl_hip_fm_count_batch(ref, pats.data(), m, np, want.data(), nullptr));
    for (size_t i = 0; i < np; ++i)
        if (want[i] < 1)
        {
            fprintf(stderr, "reference count of pattern %zu is %llu\n", i, (unsigned long long)want[i]);
            return 2;
        }
    std::vector<int32_t> devs;
    if (transport == "copy2")
    {
        devs.assign(2, 0);
        setenv("SDSL_HIP_GROUP_TRANSPORT", "copy", 1);
    }
    else
        devs.assign(1, 0);
    sdsl_hip_group_t grp = nullptr;
    fprintf(stderr, "reference built\n");
    if (transport != "none")
        OK(sdsl_hip_group_create(devs.data(), (int32_t)devs.size(), &grp));
    unsetenv("SDSL_HIP_GROUP_TRANSPORT");
    fprintf(stderr, "group created\n");
    auto report = [&](const char * tag, int it) -> int
    {
        size_t bad = 0, first = 0;
        for (size_t i = 0; i < np; ++i)
            if (got[i] != want[i] && bad++ == 0)
                first = i;
        if (bad)
            printf("%s round %d: %zu of %zu differ; first %zu got %llu want %llu\n", tag, it, bad, np, first, (unsigned long long)got[first],
                   (unsigned long long)want[first]);
        return bad ? 1 : 0;
    };
    std::vector<sdsl_hip_fm_t> reps(devs.size(), nullptr);
    if (grp)
        OK(sdsl_hip_group_fm_create_from_text(grp, text.data(), text.size(), 0, reps.data()));
    fprintf(stderr, "replicas built\n");
    int bad = 0;
    if (phases.find('A') != std::string::npos)
    {
        for (int it = 0; it < rounds; ++it)
        {
            std::fill(got.begin(), got.end(), 0xDEADBEEFull);
            OK(sdsl_hip_group_fm_count_batch(grp, reps.data(), pats.data(), m, np, got.data(), 2));
            if (it < 3 || it % 50 == 0)
                fprintf(stderr, "A round %d done\n", it);
            bad += report("A", it);
        }
        printf("A (fixed replicas, %s): %d of %d rounds with mismatches\n", transport.c_str(), bad, rounds);
    }
    if (phases.find('B') != std::string::npos)
    {
        bad = 0;
        for (int it = 0; it < rounds; ++it)
        {
            const bool say = getenv("STRESS_VERBOSE") != nullptr;
            for (auto & r : reps)
                sdsl_hip_fm_destroy(r);
            if (say)
                fprintf(stderr, "B %d: replicas destroyed\n", it);
            OK(sdsl_hip_group_fm_create_from_text(grp, text.data(), text.size(), 0, reps.data()));
            if (say)
                fprintf(stderr, "B %d: replicas created\n", it);
            std::fill(got.begin(), got.end(), 0xDEADBEEFull);
            OK(sdsl_hip_group_fm_count_batch(grp, reps.data(), pats.data(), m, np, got.data(), 2));
            if (say)
                fprintf(stderr, "B %d: counted\n", it);
            bad += report("B", it);
        }
        printf("B (replicas rebuilt by the group's builder threads): %d of %d rounds with mismatches\n", bad, rounds);
    }
    if (phases.find('C') != std::string::npos)
    {
        bad = 0;
        for (int it = 0; it < rounds; ++it)
        {
            sdsl_hip_fm_t one = nullptr;
            OK(sdsl_hip_fm_create_from_text(text.data(), text.size(), 0, &one));
            std::fill(got.begin(), got.end(), 0xDEADBEEFull);
            OK(sdsl_hip_fm_count_batch(one, pats.data(), m, np, got.data(), nullptr));
            bad += report("C", it);
            sdsl_hip_fm_destroy(one);
        }
        printf("C (single handle rebuilt on the main thread): %d of %d rounds with mismatches\n", bad, rounds);
    }
    if (phases.find('D') != std::string::npos)
    { // which part of a rebuilt index differs: its serialised bytes (BWT / wavelet tree / samples) or only the answers
        std::vector<uint8_t> ref_bytes, bytes;
        size_t len = 0;
        OK(sdsl_hip_fm_serialize(ref, 32, 64, nullptr, 0, &len));
        ref_bytes.resize(len);
        OK(sdsl_hip_fm_serialize(ref, 32, 64, ref_bytes.data(), len, &len));
        int bad_bytes = 0, bad_cnt = 0;
        for (int it = 0; it < rounds; ++it)
        {
            const bool say = getenv("STRESS_VERBOSE") != nullptr;
            sdsl_hip_fm_t one = nullptr;
            if (say)
                fprintf(stderr, "D %d: create ...", it);
            OK(sdsl_hip_fm_create_from_text(text.data(), text.size(), 0, &one));
            size_t l2 = 0;
            if (say)
                fprintf(stderr, " serialize ...");
            OK(sdsl_hip_fm_serialize(one, 32, 64, nullptr, 0, &l2));
            bytes.assign(l2, 0);
            OK(sdsl_hip_fm_serialize(one, 32, 64, bytes.data(), l2, &l2));
            if (say)
                fprintf(stderr, " count ...");
            size_t diff = 0, first = 0;
            if (l2 != len)
                diff = 1;
            else
                for (size_t i = 0; i < len; ++i)
                    if (bytes[i] != ref_bytes[i] && diff++ == 0)
                        first = i;
            if (diff)
            {
                ++bad_bytes;
                printf("D round %d: serialised index differs in %zu bytes of %zu (first at %zu)\n", it, diff, len, first);
            }
            std::fill(got.begin(), got.end(), 0xDEADBEEFull);
            OK(sdsl_hip_fm_count_batch(one, pats.data(), m, np, got.data(), nullptr));
            const int b = report("D", it);
            bad_cnt += b;
            if (b && !diff)
            { // the same handle once more: a wrong table or a wrong batch?
                OK(sdsl_hip_fm_count_batch(one, pats.data(), m, np, got.data(), nullptr));
                printf("D round %d: index bytes equal; second batch on the same handle: %s\n", it, report("D again", it) ? "wrong too" : "right");
            }
            if (say)
                fprintf(stderr, " destroy ...");
            sdsl_hip_fm_destroy(one);
            if (say)
                fprintf(stderr, " done\n");
        }
        printf("D: %d rounds with different index bytes, %d with wrong counts, of %d\n", bad_bytes, bad_cnt, rounds);
    }
    fflush(stdout);
    for (auto & r : reps)
        sdsl_hip_fm_destroy(r);
    if (grp)
        sdsl_hip_group_destroy(grp);
    sdsl_hip_fm_destroy(ref);
    return 0;
}
