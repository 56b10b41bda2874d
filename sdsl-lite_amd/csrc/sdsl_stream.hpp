// sdsl_stream.hpp — reader for SDSL's serialised byte format (the second drop-in boundary,
// SURVEY.md §8(b)).  Little-endian, no magic, no versioning:
//   int_vector<w>      : u64 (width<<56 | bit_size), then ceil(bit_size/64) u64 words
//                        (int_vector.hpp:884-916, 1978-2004)
//   write_member(x)    : raw bytes of x (io.hpp:93-101)
// The reader copies words into aligned host vectors (streams may place them at odd offsets,
// e.g. after the 22-byte wavelet-tree nodes or the u16 sigma).
#pragma once
#include <cstdint>
#include <cstring>
#include <vector>

#include "bits.hpp"
#include "common.hpp"

namespace sdslhip {

struct HostIntVec
{
    std::vector<uint64_t> words; // always one extra zero word so read_bits may touch p[1]
    uint64_t bit_size = 0;
    uint8_t width = 64;
    uint64_t size() const
    {
        return width ? bit_size / width : 0;
    }
    uint64_t get(uint64_t i) const
    {
        return read_bits(words.data(), i * width, width);
    }
    bool empty() const
    {
        return bit_size == 0;
    }
};

struct StreamReader
{
    const uint8_t * p;
    size_t len;
    size_t pos = 0;
    bool ok = true;
    StreamReader(const void * bytes, size_t n) : p((const uint8_t *)bytes), len(n)
    {}
    bool raw(void * dst, size_t n)
    {
        if (!ok || n > len - pos)
        {
            ok = false;
            return false;
        }
        memcpy(dst, p + pos, n);
        pos += n;
        return true;
    }
    bool skip(size_t n)
    {
        if (!ok || n > len - pos)
        {
            ok = false;
            return false;
        }
        pos += n;
        return true;
    }
    bool u64(uint64_t & v)
    {
        return raw(&v, 8);
    }
    bool u16(uint16_t & v)
    {
        return raw(&v, 2);
    }
    // int_vector<w>::load; expect_width = 0 accepts any width (int_vector<0>)
    bool int_vector(HostIntVec & out, uint8_t expect_width = 0)
    {
        uint64_t hdr;
        if (!u64(hdr))
            return false;
        out.bit_size = hdr & ((UINT64_C(1) << 56) - 1);
        out.width = (uint8_t)(hdr >> 56);
        if (out.width == 0 || out.width > 64 || (expect_width && out.width != expect_width))
        {
            ok = false;
            return false;
        }
        uint64_t nw = (out.bit_size + 63) >> 6;
        if (nw > (len - pos) / 8)
        {
            ok = false;
            return false;
        }
        out.words.assign(nw + 1, 0);
        return raw(out.words.data(), nw * 8);
    }
    bool skip_int_vector()
    {
        uint64_t hdr;
        if (!u64(hdr))
            return false;
        uint64_t bits = hdr & ((UINT64_C(1) << 56) - 1);
        uint64_t nw = (bits + 63) >> 6;
        if (nw > (len - pos) / 8)
        {
            ok = false;
            return false;
        }
        return skip(nw * 8);
    }
    // select_support_mcl<b>::load layout (select_support_mcl.hpp:521-554): only skipped here —
    // the device select directory is rebuilt from the bits.
    bool skip_select_mcl()
    {
        uint64_t arg_cnt;
        if (!u64(arg_cnt))
            return false;
        if (arg_cnt == 0)
            return true;
        uint64_t sb = (arg_cnt + 4095) >> 12;
        if (!skip_int_vector()) // m_superblock
            return false;
        HostIntVec mol;
        if (!int_vector(mol, 1)) // mini_or_long
            return false;
        for (uint64_t i = 0; i < sb; ++i)
            if (!skip_int_vector()) // either the long or the mini vector of superblock i
                return false;
        return ok;
    }
};

// Writer for the same format (used by the *_serialize entry points: structures built on the GPU are handed
// back to unmodified SDSL code as the bytes its own serialize() would have produced).
struct StreamWriter
{
    std::vector<uint8_t> bytes;
    void raw(const void * p, size_t n)
    {
        const uint8_t * b = (const uint8_t *)p;
        bytes.insert(bytes.end(), b, b + n);
    }
    void u64(uint64_t v)
    {
        raw(&v, 8);
    }
    void u16(uint16_t v)
    {
        raw(&v, 2);
    }
    // int_vector<w>::serialize: header + ceil(bit_size/64) words (int_vector.hpp:904-916, 1978-2004)
    void int_vector(const uint64_t * words, uint64_t bit_size, uint8_t width)
    {
        u64(((uint64_t)width << 56) | bit_size);
        raw(words, (size_t)((bit_size + 63) >> 6) * 8);
    }
};

// packed vector of fixed-width integers under construction (host)
struct PackedBuilder
{
    std::vector<uint64_t> words;
    uint64_t n;
    uint8_t width;
    PackedBuilder(uint64_t n_, uint8_t w) : words(((n_ * w + 63) >> 6) + 1, 0), n(n_), width(w)
    {}
    void set(uint64_t i, uint64_t v)
    {
        uint64_t pos = i * width;
        unsigned off = (unsigned)(pos & 63);
        v &= lo_set(width);
        words[pos >> 6] |= v << off;
        if (off + width > 64)
            words[(pos >> 6) + 1] |= v >> (64 - off);
    }
    void write(StreamWriter & w) const
    {
        w.int_vector(words.data(), n * width, width);
    }
};

inline sdsl_hip_status deliver(const StreamWriter & w, void * buf, size_t cap, size_t * written)
{
    if (written)
        *written = w.bytes.size();
    if (!buf)
        return SDSL_HIP_OK; // size query
    if (cap < w.bytes.size())
    {
        set_error("serialize: buffer of %zu bytes is too small for %zu", cap, w.bytes.size());
        return SDSL_HIP_ERR_INVALID;
    }
    memcpy(buf, w.bytes.data(), w.bytes.size());
    return SDSL_HIP_OK;
}

// The size-query / fill protocol calls a serialiser twice.  Building a large stream twice is wasted work (seconds for a
// 1 GiB index), so the size query keeps what it built, per thread, and the fill call that follows with the same handle
// and arguments takes it.  `uid` is unique per handle for the life of the process (no address reuse).
struct SerCache
{
    uint64_t uid = 0, key = 0;
    std::vector<uint8_t> bytes;
};
inline SerCache & ser_cache()
{
    static thread_local SerCache c;
    return c;
}
// true if the call was answered from the cache (status in st)
inline bool deliver_cached(uint64_t uid, uint64_t key, void * buf, size_t cap, size_t * written, sdsl_hip_status & st)
{
    SerCache & c = ser_cache();
    if (!buf || c.uid != uid || c.key != key || c.bytes.empty())
        return false;
    if (written)
        *written = c.bytes.size();
    if (cap < c.bytes.size())
    {
        set_error("serialize: buffer of %zu bytes is too small for %zu", cap, c.bytes.size());
        st = SDSL_HIP_ERR_INVALID;
        return true;
    }
    memcpy(buf, c.bytes.data(), c.bytes.size());
    c = SerCache();
    st = SDSL_HIP_OK;
    return true;
}
inline sdsl_hip_status deliver_and_cache(uint64_t uid, uint64_t key, StreamWriter & w, void * buf, size_t cap, size_t * written)
{
    if (buf)
        return deliver(w, buf, cap, written);
    if (written)
        *written = w.bytes.size();
    SerCache & c = ser_cache();
    c.uid = uid;
    c.key = key;
    c.bytes = std::move(w.bytes);
    return SDSL_HIP_OK;
}
uint64_t next_handle_uid(); // common.cpp

} // namespace sdslhip
