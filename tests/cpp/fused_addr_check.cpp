// Host-side check of the fused wavelet-tree lines' address arithmetic (sdsl-lite_amd/csrc/wt_device.hpp: fused_line, fused_off,
// fused_lines_for) against plain 64-bit arithmetic, for every position a sequence the builder admits can have (wt.hip:
// wt_build_fused takes sequences below 2^36 symbols) and well beyond.  The reference is size_type (64-bit) throughout:
// wt_pc.hpp:371-399.  Compiled by tests/test_fused_addressing.py with hipcc (the functions are __host__ __device__); nothing is
// launched, so it runs in the CPU-only container.
#include <cinttypes>
#include <cstdint>
#include <cstdio>
#include <random>

#include "wt_device.hpp"

using namespace sdslhip;

static int failures = 0;

static void check(uint64_t i)
{
    const uint64_t want_line = i / kFusedPos, want_off = i % kFusedPos;
    const uint64_t line = fused_line(i);
    const unsigned off = fused_off(i, line);
    if (line != want_line || off != want_off)
    {
        if (failures < 10)
            fprintf(stderr, "position %" PRIu64 ": line %" PRIu64 " off %u, want %" PRIu64 " / %" PRIu64 "\n", i, line, off, want_line, want_off);
        ++failures;
    }
}

int main()
{
    printf("kFK %u kFusedPos %u kFSuperLog %u\n", kFK, kFusedPos, kFSuperLog);
    // both sides of every power of two up to 2^40 (2^32: the 32-bit word; 2^35: where (i >> 3) leaves 32 bits; 2^36: the builder's gate)
    for (unsigned b = 0; b <= 40; ++b)
        for (int64_t d = -3 * (int64_t)kFusedPos; d <= 3 * (int64_t)kFusedPos; ++d)
        {
            const int64_t i = (int64_t)(UINT64_C(1) << b) + d;
            if (i >= 0)
                check((uint64_t)i);
        }
    // line boundaries at random lines
    std::mt19937_64 rng(12345);
    for (int r = 0; r < 2000000; ++r)
    {
        const uint64_t line = rng() % ((UINT64_C(1) << 40) / kFusedPos);
        check(line * kFusedPos);
        check(line * kFusedPos + kFusedPos - 1);
    }
    // random positions: below 2^35, 2^35 .. 2^36, up to 2^40, and the full 64-bit range
    for (int r = 0; r < 4000000; ++r)
    {
        check(rng() & ((UINT64_C(1) << 35) - 1));
        check((UINT64_C(1) << 35) + (rng() & ((UINT64_C(1) << 35) - 1)));
        check(rng() & ((UINT64_C(1) << 40) - 1));
        check(rng());
    }
    // the number of lines of a node: position `size` is addressable, and no line beyond it is counted
    for (unsigned b = 0; b <= 40; ++b)
        for (int64_t d = -2 * (int64_t)kFusedPos; d <= 2 * (int64_t)kFusedPos; ++d)
        {
            const int64_t sz = (int64_t)(UINT64_C(1) << b) + d;
            if (sz < 0)
                continue;
            const uint64_t n = fused_lines_for((uint64_t)sz);
            if (fused_line((uint64_t)sz) != n - 1)
            {
                if (failures < 10)
                    fprintf(stderr, "size %" PRId64 ": %" PRIu64 " lines, position size on line %" PRIu64 "\n", sz, n, fused_line((uint64_t)sz));
                ++failures;
            }
        }
    if (failures)
    {
        fprintf(stderr, "%d mismatches\n", failures);
        return 1;
    }
    printf("ok\n");
    return 0;
}
