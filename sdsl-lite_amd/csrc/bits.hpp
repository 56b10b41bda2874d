// bits.hpp — word-level primitives of the device kernels (the gfx950 counterpart of SDSL's
// bits::cnt / bits::sel / bits::hi / lo_set, bits.hpp:486-502,586-612,653-684,194).
// gfx950 has no PDEP, so in-word select is a 6-step popcount bisection instead of SDSL's
// pdep+tzcnt (bits.hpp:588-591).  Everything here is __host__ __device__ so the host-side
// builders of this library share the exact same arithmetic.
#pragma once
#include <hip/hip_runtime.h>

#include <cstdint>

#define SH_HD __host__ __device__ __forceinline__

namespace sdslhip {

SH_HD unsigned popc64(uint64_t x)
{
#if defined(__HIP_DEVICE_COMPILE__)
    return (unsigned)__popcll(x);
#else
    return (unsigned)__builtin_popcountll(x);
#endif
}

SH_HD unsigned popc32(uint32_t x)
{
#if defined(__HIP_DEVICE_COMPILE__)
    return (unsigned)__popc(x);
#else
    return (unsigned)__builtin_popcount(x);
#endif
}

// k low bits set, k in [0,64]   (SDSL: bits::lo_set[k])
SH_HD uint64_t lo_set(unsigned k)
{
    return k >= 64 ? ~UINT64_C(0) : ((UINT64_C(1) << k) - 1);
}

// index of the most significant set bit, hi(0) = 0   (SDSL: bits::hi)
SH_HD unsigned hi64(uint64_t x)
{
#if defined(__HIP_DEVICE_COMPILE__)
    return x ? 63u - (unsigned)__clzll((long long)x) : 0u;
#else
    return x ? 63u - (unsigned)__builtin_clzll(x) : 0u;
#endif
}

// position (0-based) of the r-th set bit of w, r in [1, popc64(w)]   (SDSL: bits::sel(w, r))
SH_HD unsigned sel64(uint64_t w, unsigned r)
{
    unsigned pos = 0;
    uint32_t x = (uint32_t)w;
    unsigned c = popc32(x);
    if (r > c)
    {
        r -= c;
        x = (uint32_t)(w >> 32);
        pos = 32;
    }
    c = popc32(x & 0xFFFFu);
    if (r > c)
    {
        r -= c;
        x >>= 16;
        pos += 16;
    }
    c = popc32(x & 0xFFu);
    if (r > c)
    {
        r -= c;
        x >>= 8;
        pos += 8;
    }
    c = popc32(x & 0xFu);
    if (r > c)
    {
        r -= c;
        x >>= 4;
        pos += 4;
    }
    c = popc32(x & 0x3u);
    if (r > c)
    {
        r -= c;
        x >>= 2;
        pos += 2;
    }
    if (r > (x & 1u))
        pos += 1;
    return pos;
}

// Read `len` (0..64) bits starting at bit position `pos` from a packed little-endian u64 array
// (SDSL: bits::read_int, bits.hpp:777-790).  Touches word pos>>6 and, if the field straddles,
// the next one.
SH_HD uint64_t read_bits(const uint64_t * data, uint64_t pos, unsigned len)
{
    if (len == 0)
        return 0;
    const uint64_t * p = data + (pos >> 6);
    unsigned off = (unsigned)(pos & 63);
    uint64_t v = p[0] >> off;
    if (off + len > 64)
        v |= p[1] << (64 - off);
    return v & lo_set(len);
}

// OR `len` (0..64) bits of v into a zero-initialised packed array at bit position `pos` (host-side builders)
inline void write_bits(uint64_t * data, uint64_t pos, unsigned len, uint64_t v)
{
    if (len == 0)
        return;
    v &= lo_set(len);
    uint64_t * p = data + (pos >> 6);
    const unsigned off = (unsigned)(pos & 63);
    p[0] |= v << off;
    if (off + len > 64)
        p[1] |= v >> (64 - off);
}

} // namespace sdslhip
