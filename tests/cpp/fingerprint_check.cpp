// CPU check of include/sdsl_hip/adaptors.hpp: the content fingerprint that decides whether a device replica of a bit_vector may be
// reused.  Built against the real SDSL headers by tests/test_adaptor_fingerprint.py (build container only).
#include <cstdio>
#include <sdsl_hip/adaptors.hpp>

int main()
{
    using namespace sdsl;
    int bad = 0;
    for (uint64_t n : {0ull, 1ull, 63ull, 64ull, 65ull, 100003ull, (1ull << 28) + 77, (1ull << 29) + (1ull << 22) * 64 + 5})
    { // (the larger sizes span several 32 MiB stretches: hashed by several threads, folded in order)
        bit_vector v(n, 0);
        util::set_random_bits(v, 815);
        const auto f0 = hip_detail::fingerprint(&v), f1 = hip_detail::fingerprint(&v);
        if (!(f0 == f1))
            ++bad, printf("n %llu: not deterministic\n", (unsigned long long)n);
        if (n)
        {
            for (uint64_t pos : {uint64_t(0), n / 2, n - 1})
            {
                v[pos] = !v[pos];
                if (hip_detail::fingerprint(&v) == f0)
                    ++bad, printf("n %llu: a flipped bit at %llu went unnoticed\n", (unsigned long long)n, (unsigned long long)pos);
                v[pos] = !v[pos];
            }
            if (!(hip_detail::fingerprint(&v) == f0))
                ++bad, printf("n %llu: restored content, other fingerprint\n", (unsigned long long)n);
            bit_vector w(n + 1, 0); // same words, other length
            std::copy(v.data(), v.data() + ((n + 63) >> 6), w.data());
            if (((n + 1 + 63) >> 6) == ((n + 63) >> 6) && hip_detail::fingerprint(&w) == f0)
                ++bad, printf("n %llu: the length is not part of the fingerprint\n", (unsigned long long)n);
        }
    }
    printf("fingerprint check: %d failed\n", bad);
    return bad != 0;
}
