// sdsl_hip/adaptors.hpp — header-only C++ adaptors that put the batched MI355X engine (sdsl_hip.h)
// back behind SDSL's own concepts, for header-only callers of xxsds/sdsl-lite.
//
//   #include <sdsl/bit_vectors.hpp>          // the caller's SDSL, unchanged
//   #include <sdsl_hip/adaptors.hpp>         // this file; link with -lsdsl_hip
//
//   sdsl::bit_vector bv = ...;
//   sdsl::rank_support_v5_hip<1> rs(&bv);            // same constructor shape as rank_support_v5<1>
//   rs(i);                                           // SDSL's scalar operator(): answered by the CALLER'S OWN rank_support_v5 over
//                                                    //   the same bit_vector (built on first use) — a one-element launch is never
//                                                    //   the right thing; ~40 ns as before the switch
//   rs.rank_batch(idx, n, out);                      // the reason to switch: one launch for n queries
//
// Every class mirrors the reference interface it replaces (constructor from `bit_vector const*`,
// rank/select/operator(), size(), set_vector, serialize/load writing and reading SDSL's OWN byte
// format, ==/!=, nested typedefs) so that it satisfies the t_rank / t_select concepts
// (rank_support_v5.hpp:44-200, select_support_mcl.hpp:64-117, rrr_vector.hpp:455-600).
// The batch members have no CPU path.  SCALAR members never reach the device, for any type (a one-element launch
// is never the right thing): they forward to the caller's own SDSL object (the header includes SDSL anyway: it is the caller's
// library, not this one's).  The supports of a plain bit_vector build SDSL's support over the caller's vector on the first scalar
// call; rrr_vector_hip / sd_vector_hip / wt_huff_hip / csa_wt_hip keep a NON-OWNING pointer to the host object they were
// constructed from — SDSL's own convention for everything that supports another object (rank_support.hpp:33) — so that object must
// outlive the adaptor's scalar calls (batch calls only need the device image).  An adaptor that was constructed from something
// else (plain bits, a position list) loads SDSL's type from the device image's own serialised bytes on the first scalar call
// (std::call_once).  The old road — a one-element batch through a pinned mailbox, about 14 microseconds — stays as *_on_device
// members for checking the device image.  Supports of one
// bit_vector share ONE device replica per device (ref-counted; select directories are added when a select support asks).
// Errors surface as
// std::runtime_error carrying sdsl_hip_last_error() (SDSL itself throws std::logic_error /
// std::bad_alloc on its own failures, memory_management.hpp:907-910).
#pragma once
#include <cstdint>
#include <functional>
#include <map>
#include <memory>
#include <mutex>
#include <iterator>
#include <sstream>
#include <stdexcept>
#include <string>
#include <tuple>
#include <vector>

#include <sdsl/bit_vectors.hpp>
#include <type_traits>

#include <thread>

#include <sdsl/suffix_arrays.hpp>
#include <sdsl/wavelet_trees.hpp>

#include "../sdsl_hip.h"

namespace sdsl
{
namespace hip_detail
{
inline void check(sdsl_hip_status st, char const * what)
{
    if (st != SDSL_HIP_OK)
        throw std::runtime_error(std::string(what) + ": " + sdsl_hip_last_error());
}
template <class T>
inline std::string to_stream(T const & x)
{
    std::ostringstream os;
    x.serialize(os);
    return os.str();
}
struct bv_deleter
{
    void operator()(sdsl_hip_bv_s * p) const
    {
        sdsl_hip_bv_destroy(p);
    }
};
typedef std::shared_ptr<sdsl_hip_bv_s> bv_ptr;
//! 128-bit fingerprint of a bit_vector's content: four independent multiply-fold lanes per stretch (runs at memory bandwidth), the
//! stretches of a large vector hashed by several threads (11.7 GB/s per thread on the GPU box's host: a 2^34-bit vector, 2 GiB, is 0.18 s
//! on one thread — a wavelet tree's four supports each ask once) and folded in order, so the value does not depend on how many threads ran.
struct fingerprint_t
{
    uint64_t a = 0, b = 0;
    bool operator==(fingerprint_t const & o) const
    {
        return a == o.a and b == o.b;
    }
};
inline void fingerprint_stretch(uint64_t const * w, uint64_t n, uint64_t h[4])
{
    h[0] = 0x9E3779B97F4A7C15ull, h[1] = 0xC2B2AE3D27D4EB4Full, h[2] = 0x165667B19E3779F9ull, h[3] = 0x27D4EB2F165667C5ull;
    uint64_t i = 0;
    for (; i + 4 <= n; i += 4)
        for (int k = 0; k < 4; ++k)
        {
            h[k] = (h[k] ^ w[i + k]) * 0xFF51AFD7ED558CCDull;
            h[k] ^= h[k] >> 29;
        }
    for (; i < n; ++i)
        h[0] = ((h[0] ^ w[i]) * 0xFF51AFD7ED558CCDull) ^ (h[0] >> 31);
}
inline fingerprint_t fingerprint(bit_vector const * v)
{
    uint64_t const * w = v->data();
    const uint64_t n = (v->bit_size() + 63) >> 6;
    constexpr uint64_t kStretch = uint64_t(1) << 22; // words (32 MiB): the unit of work AND of the fold, whatever the thread count
    const uint64_t n_st = (n + kStretch - 1) / kStretch;
    std::vector<uint64_t> part(4 * std::max<uint64_t>(n_st, 1), 0);
    auto work = [&](uint64_t s0, uint64_t step) {
        for (uint64_t s = s0; s < n_st; s += step)
            fingerprint_stretch(w + s * kStretch, std::min(kStretch, n - s * kStretch), &part[4 * s]);
    };
    unsigned nt = (unsigned)std::min<uint64_t>(n_st, std::min(8u, std::max(1u, std::thread::hardware_concurrency())));
    if (nt <= 1)
        work(0, 1);
    else
    {
        std::vector<std::thread> th;
        for (unsigned t = 1; t < nt; ++t)
            th.emplace_back(work, (uint64_t)t, (uint64_t)nt);
        work(0, nt);
        for (auto & t : th)
            t.join();
    }
    fingerprint_t f;
    f.a = v->bit_size();
    f.b = 0x2545F4914F6CDD1Dull;
    for (uint64_t s = 0; s < n_st; ++s)
    {
        f.a = (f.a ^ part[4 * s] ^ (part[4 * s + 1] * 3)) * 0xD6E8FEB86659FD93ull;
        f.a ^= f.a >> 32;
        f.b = (f.b ^ part[4 * s + 2] ^ (part[4 * s + 3] * 5)) * 0xCA5A826395121157ull;
        f.b ^= f.b >> 31;
    }
    return f;
}
//! One device replica per (bit_vector object, device, pattern): SDSL's supports hold a non-owning pointer to the vector they
//! support (rank_support.hpp:33, select_support.hpp:35) and three supports of one vector cost no copy of it; here they share
//! one replica.  A replica is reused only while the vector's size AND content fingerprint are what they were when it was made
//! (a support constructed after the vector was modified gets a fresh replica, as a fresh SDSL support would read the new bits).
//! The fingerprint is taken OUTSIDE the registry's lock; the lock covers a map lookup, and the sweep of expired entries runs every
//! 32nd call only — constructing supports from many threads does not serialise on it.
struct replica_registry
{
    struct entry
    {
        std::weak_ptr<sdsl_hip_bv_s> dev;
        uint64_t bits = 0;
        fingerprint_t print;
    };
    std::mutex m;
    std::map<std::tuple<void const *, int, unsigned>, entry> map;
    unsigned calls = 0;
};
inline replica_registry & replicas()
{
    static replica_registry r; // (inline function: one registry per program)
    return r;
}
//! (t_b, t_pat_len) as SDSL's template arguments; two-bit patterns get their occurrence vector on the device
inline bv_ptr make_device_bv(bit_vector const * v, int device, uint32_t flags, uint32_t t_b = 1, uint32_t t_pat_len = 1)
{
    const unsigned pat = t_pat_len == 2 ? 100u + t_b : 0u; // plain supports of both bit values share the replica
    const fingerprint_t print = fingerprint(v); // (no lock held)
    replica_registry & R = replicas();
    auto key = std::make_tuple((void const *)v, device, pat);
    bv_ptr have;
    {
        std::lock_guard<std::mutex> lock(R.m);
        if ((++R.calls & 31u) == 0)
            for (auto it = R.map.begin(); it != R.map.end();) // (replicas nobody holds any more leave the table)
                it = it->second.dev.expired() ? R.map.erase(it) : std::next(it);
        auto hit = R.map.find(key);
        if (hit != R.map.end() and hit->second.bits == v->bit_size() and hit->second.print == print)
            have = hit->second.dev.lock();
    }
    if (have)
    { // (the directory is added outside the registry's lock: it is a device build of its own, serialised by the handle)
        if (flags)
            check(sdsl_hip_bv_add_select(have.get(), flags), "sdsl_hip_bv_add_select");
        return have;
    }
    sdsl_hip_bv_t h = nullptr; // the upload and the device build run without the lock, too
    check(sdsl_hip_bv_create_pattern(v->data(), v->bit_size(), device, t_b, t_pat_len, flags, &h),
          "sdsl_hip_bv_create_pattern");
    bv_ptr made(h, bv_deleter());
    replica_registry::entry e;
    e.dev = made;
    e.bits = v->bit_size();
    e.print = print;
    bv_ptr other;
    {
        std::lock_guard<std::mutex> lock(R.m);
        auto hit = R.map.find(key);
        if (hit != R.map.end() and hit->second.bits == e.bits and hit->second.print == print)
            other = hit->second.dev.lock(); // another thread made the same replica meanwhile: keep one
        if (!other)
            R.map[key] = e;
    }
    if (!other)
        return made;
    if (flags) // (a device build of its own: outside the lock, like the one above)
        check(sdsl_hip_bv_add_select(other.get(), flags), "sdsl_hip_bv_add_select");
    return other;
}
//! The caller's own SDSL support over the same vector, built when the first scalar query arrives (thread-safe; shared by
//! copies of the adaptor, which support the same vector)
template <class t_support>
struct lazy_host_support
{
    std::once_flag once;
    std::unique_ptr<t_support> s;
    t_support const & get(bit_vector const * v)
    {
        std::call_once(once, [&] { s.reset(new t_support(v)); });
        return *s;
    }
};
//! ---- the host side of the compressed / tree adaptors --------------------------------------------------------------------
//! What a scalar member needs from the caller's SDSL object, behind one virtual call (a nanosecond beside SDSL's 100+ ns for
//! a rank on rrr_vector<63> or 600 ns for wt.rank on a GiB of text), so that one adaptor class serves every host type it
//! can be constructed from.
struct host_bits // rrr_vector<...>, sd_vector<...>: the vector and its four supports
{
    virtual ~host_bits() = default;
    virtual uint64_t rank(uint64_t i, bool b) const = 0;
    virtual uint64_t select(uint64_t i, bool b) const = 0;
    virtual bool access(uint64_t i) const = 0;
    virtual uint64_t get_int(uint64_t i, uint8_t len) const = 0;
};
template <class t_vec>
struct host_bits_of final : host_bits
{
    std::unique_ptr<t_vec> owned; // (only when the adaptor was not constructed from a t_vec: loaded from the device image)
    t_vec const * v;
    typename t_vec::rank_1_type r1;
    typename t_vec::rank_0_type r0;
    typename t_vec::select_1_type s1;
    typename t_vec::select_0_type s0;
    explicit host_bits_of(t_vec const * p) : v(p), r1(p), r0(p), s1(p), s0(p)
    {}
    explicit host_bits_of(std::unique_ptr<t_vec> o) : owned(std::move(o)), v(owned.get()), r1(v), r0(v), s1(v), s0(v)
    {}
    uint64_t rank(uint64_t i, bool b) const override
    {
        return b ? r1(i) : r0(i);
    }
    uint64_t select(uint64_t i, bool b) const override
    {
        return b ? s1(i) : s0(i);
    }
    bool access(uint64_t i) const override
    {
        return (*v)[i];
    }
    uint64_t get_int(uint64_t i, uint8_t len) const override
    {
        return v->get_int(i, len);
    }
};
struct host_wt // wt_huff / wt_blcd / wt_hutu over any bit vector
{
    virtual ~host_wt() = default;
    virtual uint64_t rank(uint64_t i, uint8_t c) const = 0;
    virtual uint64_t select(uint64_t i, uint8_t c) const = 0;
    virtual uint8_t access(uint64_t i) const = 0;
    virtual std::pair<uint64_t, uint8_t> inverse_select(uint64_t i) const = 0;
};
template <class t_wt>
struct host_wt_of final : host_wt
{
    t_wt const * w;
    explicit host_wt_of(t_wt const * p) : w(p)
    {}
    uint64_t rank(uint64_t i, uint8_t c) const override
    {
        return w->rank(i, c);
    }
    uint64_t select(uint64_t i, uint8_t c) const override
    {
        return w->select(i, c);
    }
    uint8_t access(uint64_t i) const override
    {
        return (uint8_t)(*w)[i];
    }
    std::pair<uint64_t, uint8_t> inverse_select(uint64_t i) const override
    {
        auto r = w->inverse_select(i);
        return std::make_pair((uint64_t)r.first, (uint8_t)r.second);
    }
};
struct host_csa // csa_wt<...>: csa[i] and the one-pattern forms of count / locate / extract
{
    virtual ~host_csa() = default;
    virtual uint64_t sa(uint64_t i) const = 0;
    virtual uint64_t count(uint8_t const * p, size_t m) const = 0;
    virtual std::vector<uint64_t> locate(uint8_t const * p, size_t m) const = 0;
    virtual std::string extract(uint64_t begin, uint64_t end) const = 0;
};
template <class t_csa>
struct host_csa_of final : host_csa
{
    t_csa const * c;
    explicit host_csa_of(t_csa const * p) : c(p)
    {}
    uint64_t sa(uint64_t i) const override
    {
        return (*c)[i];
    }
    uint64_t count(uint8_t const * p, size_t m) const override
    {
        return sdsl::count(*c, p, p + m);
    }
    std::vector<uint64_t> locate(uint8_t const * p, size_t m) const override
    {
        auto occ = sdsl::locate(*c, p, p + m);
        return std::vector<uint64_t>(occ.begin(), occ.end());
    }
    std::string extract(uint64_t begin, uint64_t end) const override
    {
        auto t = sdsl::extract(*c, begin, end);
        return std::string(t.begin(), t.end());
    }
};
//! made when the first scalar call arrives (thread-safe; copies of an adaptor share it)
template <class t_iface>
struct lazy_host
{
    std::once_flag once;
    std::function<std::unique_ptr<t_iface>()> make;
    std::unique_ptr<t_iface> h;
    explicit lazy_host(std::function<std::unique_ptr<t_iface>()> f) : make(std::move(f))
    {}
    t_iface const & get()
    {
        std::call_once(once, [&] {
            h = make();
            make = nullptr;
        });
        return *h;
    }
};
template <class t_iface>
inline t_iface const & host_of(std::shared_ptr<lazy_host<t_iface>> const & h, char const * who)
{
    if (!h)
        throw std::runtime_error(std::string(who) + ": scalar call on an adaptor without a host object (default-constructed)");
    return h->get();
}
//! SDSL's type loaded from the bytes the device image serialises to (they are SDSL's own format)
template <class t_vec, class t_serialize>
inline std::unique_ptr<t_vec> load_from_device(t_serialize && serialize, char const * what)
{
    size_t len = 0;
    (void)serialize((void *)nullptr, (size_t)0, &len); // size query
    std::string bytes(len, 0);
    check(serialize((void *)&bytes[0], len, &len), what);
    std::istringstream iss(bytes);
    std::istream & in = iss; // (an istringstream lvalue would select the cereal overload of load)
    std::unique_ptr<t_vec> v(new t_vec());
    v->load(in);
    return v;
}
constexpr bool pattern_ok(unsigned t_b, unsigned t_pat_len)
{
    return (t_pat_len == 1 and t_b <= 1) or (t_pat_len == 2 and (t_b == 10 or t_b == 01 or t_b == 00 or t_b == 11));
}
} // namespace hip_detail

//! The plain bits of ANY SDSL bit-vector type that offers size() and get_int(idx, len) — bit_vector_il<>, rrr_vector<15>,
//! hyb_vector<>, sd_vector<>, ... — as a bit_vector.  rank / select answers do not depend on the representation, so
//! such a vector is served by handing its bits to rank_support_v5_hip / select_support_mcl_hip (plain rank lines),
//! rrr_vector_hip (H0-compressed on the device) or sd_vector_hip (sparse).
template <class t_bv>
inline bit_vector to_bit_vector(t_bv const & v)
{
    bit_vector out(v.size(), 0);
    uint64_t * w = out.data();
    const uint64_t n = v.size();
    for (uint64_t i = 0; i + 64 <= n; i += 64)
        w[i >> 6] = v.get_int(i, 64);
    if (n & 63)
        w[n >> 6] = v.get_int(n & ~UINT64_C(63), (uint8_t)(n & 63));
    return out;
}

//! Drop-in for rank_support_v5<t_b, 1> (rank_support_v5.hpp:44) with a batched member.
template <uint8_t t_b = 1, uint8_t t_pat_len = 1>
class rank_support_v5_hip
{
    static_assert(hip_detail::pattern_ok(t_b, t_pat_len), "rank_support_v5_hip: patterns 0, 1, 10, 01, 00, 11");
    // a two-bit pattern is answered as rank_1 on its occurrence vector
    static constexpr int dev_bit = t_pat_len == 2 ? 1 : t_b;

public:
    typedef bit_vector bit_vector_type;
    typedef bit_vector::size_type size_type;
    enum
    {
        bit_pat = t_b
    };
    enum
    {
        bit_pat_len = t_pat_len
    };

private:
    typedef hip_detail::lazy_host_support<rank_support_v5<t_b, t_pat_len>> host_type;
    bit_vector const * m_v = nullptr;
    hip_detail::bv_ptr m_dev;
    std::shared_ptr<host_type> m_host;
    int m_device = 0;

public:
    explicit rank_support_v5_hip(bit_vector const * v = nullptr, int device = 0) : m_device(device)
    {
        set_vector(v);
    }
    //! Number of t_b bits in [0, idx), idx in [0, size()]   (rank_support_v5.hpp:131-149)
    //! A scalar query is answered by the caller's own rank_support_v5 over the same vector, built when the first one arrives
    //! (unmodified SDSL loops such as wt_pc's level walk keep their speed); batches go to the GPU.
    size_type rank(size_type idx) const
    {
        if (!m_v)
            throw std::runtime_error("rank_support_v5_hip: no vector set");
        return m_host->get(m_v).rank(idx);
    }
    //! the same query through the device (one launch + one synchronisation, about 14 microseconds): for checking the replica
    size_type rank_on_device(size_type idx) const
    {
        if (!m_dev)
            throw std::runtime_error("rank_support_v5_hip: no vector set");
        uint64_t r = 0;
        hip_detail::check(sdsl_hip_bv_query_one(m_dev.get(), 0, dev_bit, idx, &r), "sdsl_hip_bv_query_one");
        return r;
    }
    size_type operator()(size_type idx) const
    {
        return rank(idx);
    }
    //! out[q] = rank(idx[q]); host or device pointers; asynchronous on `stream` for device pointers
    void rank_batch(size_type const * idx, size_t n, size_type * out, void * stream = nullptr) const
    {
        if (!m_dev)
            throw std::runtime_error("rank_support_v5_hip: no vector set");
        hip_detail::check(sdsl_hip_bv_rank_batch(m_dev.get(), dev_bit, idx, n, out, stream), "sdsl_hip_bv_rank_batch");
    }
    //! working memory for batches of up to n queries enqueued while `stream` is being captured into a HIP graph (sdsl_hip.h:
    //! stream capture); shared by the supports of this vector on this device; 0 releases it
    void reserve_capture_scratch(size_t n) const
    {
        if (m_dev)
            hip_detail::check(sdsl_hip_bv_reserve_capture_scratch(m_dev.get(), n), "sdsl_hip_bv_reserve_capture_scratch");
    }
    size_type size() const
    {
        return m_v->size();
    }
    //! Writes exactly the bytes rank_support_v5<t_b>::serialize writes (rank_support_v5.hpp:160-167)
    size_type serialize(std::ostream & out, structure_tree_node * v = nullptr, std::string name = "") const
    {
        rank_support_v5<t_b, t_pat_len> host(m_v);
        return host.serialize(out, v, name);
    }
    //! Reads (and skips) SDSL's directory, then re-lays the supported vector on the device (:169-173)
    void load(std::istream & in, bit_vector const * v = nullptr)
    {
        int_vector<64> skipped;
        skipped.load(in);
        set_vector(v);
    }
    void set_vector(bit_vector const * v = nullptr)
    {
        m_v = v;
        m_dev = v ? hip_detail::make_device_bv(v, m_device, 0, t_b, t_pat_len) : hip_detail::bv_ptr();
        m_host = std::make_shared<host_type>();
    }
    //! the device replica (shared with the other supports of the same vector on this device)
    sdsl_hip_bv_t device_handle() const
    {
        return m_dev.get();
    }
    bool operator==(rank_support_v5_hip const & o) const noexcept
    {
        return m_v == o.m_v or (m_v and o.m_v and *m_v == *o.m_v);
    }
    bool operator!=(rank_support_v5_hip const & o) const noexcept
    {
        return !(*this == o);
    }
};

//! Drop-in for rank_support_v<t_b, 1> (rank_support_v.hpp:40), the default bit_vector::rank_1_type: its answers are
//! those of rank_support_v5 (only the directory differs), so it shares the device structure; serialize/load speak
//! rank_support_v's own byte format.
template <uint8_t t_b = 1, uint8_t t_pat_len = 1>
class rank_support_v_hip : public rank_support_v5_hip<t_b, t_pat_len>
{
    typedef rank_support_v5_hip<t_b, t_pat_len> base;
    bit_vector const * m_vv = nullptr;

public:
    typedef typename base::size_type size_type;
    explicit rank_support_v_hip(bit_vector const * v = nullptr, int device = 0) : base(v, device), m_vv(v)
    {}
    size_type serialize(std::ostream & out, structure_tree_node * v = nullptr, std::string name = "") const
    {
        rank_support_v<t_b, t_pat_len> host(m_vv);
        return host.serialize(out, v, name);
    }
    void load(std::istream & in, bit_vector const * v = nullptr)
    {
        rank_support_v<t_b, t_pat_len> skipped;
        skipped.load(in, v);
        set_vector(v);
    }
    void set_vector(bit_vector const * v = nullptr)
    {
        m_vv = v;
        base::set_vector(v);
    }
};

//! Drop-in for select_support_mcl<t_b, 1> (select_support_mcl.hpp:64) with a batched member.
template <uint8_t t_b = 1, uint8_t t_pat_len = 1>
class select_support_mcl_hip
{
    static_assert(hip_detail::pattern_ok(t_b, t_pat_len), "select_support_mcl_hip: patterns 0, 1, 10, 01, 00, 11");
    static constexpr int dev_bit = t_pat_len == 2 ? 1 : t_b;

public:
    typedef bit_vector bit_vector_type;
    typedef bit_vector::size_type size_type;
    enum
    {
        bit_pat = t_b
    };
    enum
    {
        bit_pat_len = t_pat_len
    };

private:
    typedef hip_detail::lazy_host_support<select_support_mcl<t_b, t_pat_len>> host_type;
    bit_vector const * m_v = nullptr;
    hip_detail::bv_ptr m_dev;
    std::shared_ptr<host_type> m_host;
    int m_device = 0;

public:
    explicit select_support_mcl_hip(bit_vector const * v = nullptr, int device = 0) : m_device(device)
    {
        set_vector(v);
    }
    //! Position of the i-th t_b bit, i in [1, #t_b bits]   (select_support_mcl.hpp:384-439)
    //! Scalar: the caller's own select_support_mcl over the same vector, built when the first scalar query arrives.
    size_type select(size_type i) const
    {
        if (!m_v)
            throw std::runtime_error("select_support_mcl_hip: no vector set");
        return m_host->get(m_v).select(i);
    }
    //! the same query through the device (one launch + one synchronisation)
    size_type select_on_device(size_type i) const
    {
        if (!m_dev)
            throw std::runtime_error("select_support_mcl_hip: no vector set");
        uint64_t r = 0;
        hip_detail::check(sdsl_hip_bv_query_one(m_dev.get(), 1, dev_bit, i, &r), "sdsl_hip_bv_query_one");
        return r;
    }
    size_type operator()(size_type i) const
    {
        return select(i);
    }
    void select_batch(size_type const * i, size_t n, size_type * out, void * stream = nullptr) const
    {
        if (!m_dev)
            throw std::runtime_error("select_support_mcl_hip: no vector set");
        hip_detail::check(sdsl_hip_bv_select_batch(m_dev.get(), dev_bit, i, n, out, stream), "sdsl_hip_bv_select_batch");
    }
    //! working memory for batches of up to n queries enqueued while `stream` is being captured into a HIP graph (sdsl_hip.h:
    //! stream capture); shared by the supports of this vector on this device; 0 releases it
    void reserve_capture_scratch(size_t n) const
    {
        if (m_dev)
            hip_detail::check(sdsl_hip_bv_reserve_capture_scratch(m_dev.get(), n), "sdsl_hip_bv_reserve_capture_scratch");
    }
    size_type size() const
    {
        return m_v->size();
    }
    //! Writes exactly the bytes select_support_mcl<t_b>::serialize writes (select_support_mcl.hpp:474-518)
    size_type serialize(std::ostream & out, structure_tree_node * v = nullptr, std::string name = "") const
    {
        select_support_mcl<t_b, t_pat_len> host(m_v);
        return host.serialize(out, v, name);
    }
    void load(std::istream & in, bit_vector const * v = nullptr)
    {
        select_support_mcl<t_b, t_pat_len> skipped;
        skipped.load(in, v);
        set_vector(v);
    }
    void set_vector(bit_vector const * v = nullptr)
    {
        m_v = v;
        m_dev = v ? hip_detail::make_device_bv(v, m_device, dev_bit ? SDSL_HIP_BV_SELECT1 : SDSL_HIP_BV_SELECT0, t_b, t_pat_len)
                  : hip_detail::bv_ptr();
        m_host = std::make_shared<host_type>();
    }
    sdsl_hip_bv_t device_handle() const
    {
        return m_dev.get();
    }
    bool operator==(select_support_mcl_hip const & o) const noexcept
    {
        return m_v == o.m_v or (m_v and o.m_v and *m_v == *o.m_v);
    }
    bool operator!=(select_support_mcl_hip const & o) const noexcept
    {
        return !(*this == o);
    }
};

namespace hip_detail
{
//! the plain bits of a bit_vector_il<> / rrr_vector<15> on the device: the object's own serialised bytes are decoded there
template <class t_bv>
inline bv_ptr make_device_bv_from_sibling(t_bv const & v, int32_t kind, int device, uint32_t flags)
{
    std::string s = to_stream(v);
    sdsl_hip_bv_t h = nullptr;
    check(sdsl_hip_bv_create_from_sdsl(s.data(), s.size(), kind, device, flags, &h), "sdsl_hip_bv_create_from_sdsl");
    return bv_ptr(h, bv_deleter());
}
//! rank over device bits that came from some other representation
template <uint8_t t_b>
class device_bits_rank
{
protected:
    bv_ptr m_dev;
    uint64_t m_size = 0;
    int m_device = 0;

public:
    typedef bit_vector::size_type size_type;
    size_type rank(size_type i) const
    {
        uint64_t r = 0;
        check(sdsl_hip_bv_query_one(m_dev.get(), 0, t_b, i, &r), "sdsl_hip_bv_query_one");
        return r;
    }
    size_type operator()(size_type i) const
    {
        return rank(i);
    }
    void rank_batch(size_type const * i, size_t n, size_type * out, void * stream = nullptr) const
    {
        check(sdsl_hip_bv_rank_batch(m_dev.get(), t_b, i, n, out, stream), "sdsl_hip_bv_rank_batch");
    }
    size_type size() const
    {
        return m_size;
    }
};
template <uint8_t t_b>
class device_bits_select
{
protected:
    bv_ptr m_dev;
    uint64_t m_size = 0;
    int m_device = 0;

public:
    typedef bit_vector::size_type size_type;
    size_type select(size_type i) const
    {
        uint64_t r = 0;
        check(sdsl_hip_bv_query_one(m_dev.get(), 1, t_b, i, &r), "sdsl_hip_bv_query_one");
        return r;
    }
    size_type operator()(size_type i) const
    {
        return select(i);
    }
    void select_batch(size_type const * i, size_t n, size_type * out, void * stream = nullptr) const
    {
        check(sdsl_hip_bv_select_batch(m_dev.get(), t_b, i, n, out, stream), "sdsl_hip_bv_select_batch");
    }
    size_type size() const
    {
        return m_size;
    }
};
} // namespace hip_detail

//! rank_support_il<t_b, t_bs> / select_support_il<t_b, t_bs> look-alikes (bit_vector_il.hpp:303-317,  :398-460) over a
//! bit_vector_il<t_bs>: the interleaving is a host cache layout (the device's rank lines are its counterpart); the
//! vector's serialised bytes are de-interleaved on the device and the supports answer from rank lines.
template <uint8_t t_b = 1, uint32_t t_bs = 512>
class rank_support_il_hip : public hip_detail::device_bits_rank<t_b>
{
public:
    typedef bit_vector_il<t_bs> bit_vector_type;
    explicit rank_support_il_hip(bit_vector_type const * v = nullptr, int device = 0)
    {
        this->m_device = device;
        set_vector(v);
    }
    void set_vector(bit_vector_type const * v = nullptr)
    {
        this->m_size = v ? v->size() : 0;
        this->m_dev = v ? hip_detail::make_device_bv_from_sibling(*v, SDSL_HIP_SIBLING_IL, this->m_device, 0) : hip_detail::bv_ptr();
    }
};

template <uint8_t t_b = 1, uint32_t t_bs = 512>
class select_support_il_hip : public hip_detail::device_bits_select<t_b>
{
public:
    typedef bit_vector_il<t_bs> bit_vector_type;
    explicit select_support_il_hip(bit_vector_type const * v = nullptr, int device = 0)
    {
        this->m_device = device;
        set_vector(v);
    }
    void set_vector(bit_vector_type const * v = nullptr)
    {
        this->m_size = v ? v->size() : 0;
        this->m_dev = v ? hip_detail::make_device_bv_from_sibling(*v, SDSL_HIP_SIBLING_IL, this->m_device,
                                                                  t_b ? SDSL_HIP_BV_SELECT1 : SDSL_HIP_BV_SELECT0)
                        : hip_detail::bv_ptr();
    }
};

namespace hip_detail
{
//! which stream format an rrr_vector<t_bs, t_rac, t_k> type writes: the generic template's (rrr_vector.hpp:366-378) or,
//! when the translation unit includes <sdsl/rrr_vector_15.hpp>, the rrr_vector<15> specialisation's (no invert vector)
template <class T, class = void>
struct is_rrr15_spec : std::false_type
{};
template <class T>
struct is_rrr15_spec<T, std::void_t<typename T::bi_type>> : std::true_type
{};
template <class T>
struct rrr_params;
template <uint16_t t_bs, class t_rac, uint16_t t_k>
struct rrr_params<rrr_vector<t_bs, t_rac, t_k>>
{
    static constexpr int32_t kind()
    {
        return is_rrr15_spec<rrr_vector<t_bs, t_rac, t_k>>::value ? SDSL_HIP_SIBLING_RRR15 : SDSL_HIP_SIBLING_RRR(t_bs, t_k);
    }
    static_assert(t_bs >= 2 and t_bs <= 63, "rrr_vector blocks of 2..63 bits are decoded on the device");
};
} // namespace hip_detail

//! rank_support_rrr<t_b, t_bs, ...> / select_support_rrr<t_b, t_bs, ...> look-alikes over an rrr_vector<t_bs, t_rac, t_k>
//! with 2 <= t_bs <= 63 (rrr_vector<15>, <31>, ...; for <63> prefer rrr_vector_hip, which keeps the vector compressed): its
//! classes and offsets are decoded to plain bits on the device and the supports answer from rank lines.
template <uint8_t t_b, class t_rrr>
class rank_support_rrr_bits_hip : public hip_detail::device_bits_rank<t_b>
{
public:
    typedef t_rrr bit_vector_type;
    explicit rank_support_rrr_bits_hip(bit_vector_type const * v = nullptr, int device = 0)
    {
        this->m_device = device;
        set_vector(v);
    }
    void set_vector(bit_vector_type const * v = nullptr)
    {
        this->m_size = v ? v->size() : 0;
        this->m_dev = v ? hip_detail::make_device_bv_from_sibling(*v, hip_detail::rrr_params<t_rrr>::kind(), this->m_device, 0)
                        : hip_detail::bv_ptr();
    }
};

template <uint8_t t_b, class t_rrr>
class select_support_rrr_bits_hip : public hip_detail::device_bits_select<t_b>
{
public:
    typedef t_rrr bit_vector_type;
    explicit select_support_rrr_bits_hip(bit_vector_type const * v = nullptr, int device = 0)
    {
        this->m_device = device;
        set_vector(v);
    }
    void set_vector(bit_vector_type const * v = nullptr)
    {
        this->m_size = v ? v->size() : 0;
        this->m_dev = v ? hip_detail::make_device_bv_from_sibling(*v, hip_detail::rrr_params<t_rrr>::kind(), this->m_device,
                                                                  t_b ? SDSL_HIP_BV_SELECT1 : SDSL_HIP_BV_SELECT0)
                        : hip_detail::bv_ptr();
    }
};

//! Device image of an rrr_vector<63> (rrr_vector.hpp:68) built from the host object's own serialised
//! arrays; exposes the rank/select/access members of rank_support_rrr / select_support_rrr in batch form.
//! Scalar members (operator[], get_int, and rank / select of the supports below) are answered by the HOST object the adaptor was
//! constructed from (non-owning pointer: it must outlive those calls); constructed from plain bits, the adaptor loads an
//! rrr_vector<63> from the device image's serialised bytes when the first scalar call arrives.
class rrr_vector_hip
{
public:
    typedef rrr_vector<63> host_type;
    typedef host_type::size_type size_type;

private:
    struct deleter
    {
        void operator()(sdsl_hip_rrr_s * p) const
        {
            sdsl_hip_rrr_destroy(p);
        }
    };
    typedef hip_detail::lazy_host<hip_detail::host_bits> lazy_type;
    std::shared_ptr<sdsl_hip_rrr_s> m_dev;
    std::shared_ptr<lazy_type> m_host;

    template <class t_vec>
    void host_is(t_vec const * v)
    {
        m_host = std::make_shared<lazy_type>(
            [v] { return std::unique_ptr<hip_detail::host_bits>(new hip_detail::host_bits_of<t_vec>(v)); });
    }

public:
    rrr_vector_hip() = default;
    explicit rrr_vector_hip(host_type const & v, int device = 0)
    {
        std::string s = hip_detail::to_stream(v);
        sdsl_hip_rrr_t h = nullptr;
        hip_detail::check(sdsl_hip_rrr_create_from_sdsl(s.data(), s.size(), device, &h), "sdsl_hip_rrr_create_from_sdsl");
        m_dev.reset(h, deleter());
        host_is(&v);
    }
    //! from plain bits (no pointer to `bv` is kept: the host side, if a scalar call ever asks for it, is the rrr_vector<63> the
    //! device image serialises to)
    explicit rrr_vector_hip(bit_vector const & bv, int device = 0)
    {
        sdsl_hip_rrr_t h = nullptr;
        hip_detail::check(sdsl_hip_rrr_create(bv.data(), bv.bit_size(), device, &h), "sdsl_hip_rrr_create");
        m_dev.reset(h, deleter());
        std::shared_ptr<sdsl_hip_rrr_s> dev = m_dev;
        m_host = std::make_shared<lazy_type>([dev] {
            auto ser = [&](void * buf, size_t cap, size_t * len) { return sdsl_hip_rrr_serialize(dev.get(), buf, cap, len); };
            return std::unique_ptr<hip_detail::host_bits>(
                new hip_detail::host_bits_of<host_type>(hip_detail::load_from_device<host_type>(ser, "sdsl_hip_rrr_serialize")));
        });
    }
    //! any other rrr_vector<t_bs, t_rac, t_k> (block sizes 15, 31, 127, ..., other sample rates): the device keeps its
    //! own layout (block size 63), so the vector is handed over through its plain bits — the answers are the same
    template <uint16_t t_bs, class t_rac, uint16_t t_k>
    explicit rrr_vector_hip(rrr_vector<t_bs, t_rac, t_k> const & v, int device = 0) : rrr_vector_hip(to_bit_vector(v), device)
    {
        host_is(&v); // scalar calls: the caller's own vector
    }
    size_type size() const
    {
        return sdsl_hip_rrr_size(m_dev.get());
    }
    template <uint8_t t_b>
    void rank_batch(size_type const * i, size_t n, size_type * out, void * stream = nullptr) const
    {
        hip_detail::check(sdsl_hip_rrr_rank_batch(m_dev.get(), t_b, i, n, out, stream), "sdsl_hip_rrr_rank_batch");
    }
    template <uint8_t t_b>
    void select_batch(size_type const * i, size_t n, size_type * out, void * stream = nullptr) const
    {
        hip_detail::check(sdsl_hip_rrr_select_batch(m_dev.get(), t_b, i, n, out, stream), "sdsl_hip_rrr_select_batch");
    }
    void access_batch(size_type const * i, size_t n, uint8_t * out, void * stream = nullptr) const
    {
        hip_detail::check(sdsl_hip_rrr_access_batch(m_dev.get(), i, n, out, stream), "sdsl_hip_rrr_access_batch");
    }
    //! as rank_support_v5_hip::reserve_capture_scratch
    void reserve_capture_scratch(size_t n) const
    {
        hip_detail::check(sdsl_hip_rrr_reserve_capture_scratch(m_dev.get(), n), "sdsl_hip_rrr_reserve_capture_scratch");
    }
    //! out[q] = get_int(idx[q], len)   (rrr_vector.hpp:308-356)
    void get_int_batch(size_type const * idx, uint8_t len, size_t n, uint64_t * out, void * stream = nullptr) const
    {
        hip_detail::check(sdsl_hip_rrr_get_int_batch(m_dev.get(), idx, len, n, out, stream), "sdsl_hip_rrr_get_int_batch");
    }
    //! the host side of the scalar members
    hip_detail::host_bits const & host() const
    {
        return hip_detail::host_of(m_host, "rrr_vector_hip");
    }
    uint64_t get_int(size_type idx, uint8_t len = 64) const
    {
        return host().get_int(idx, len);
    }
    bool operator[](size_type i) const
    {
        return host().access(i);
    }
    //! the same through the device (one launch + one synchronisation each): for checking the device image
    uint64_t get_int_on_device(size_type idx, uint8_t len = 64) const
    {
        uint64_t r = 0;
        get_int_batch(&idx, len, 1, &r);
        return r;
    }
    bool access_on_device(size_type i) const
    {
        uint8_t b = 0;
        access_batch(&i, 1, &b);
        return b != 0;
    }
};

//! rank_support_rrr<t_b, 63> look-alike (rrr_vector.hpp:455) over an rrr_vector_hip
template <uint8_t t_b = 1>
class rank_support_rrr_hip
{
    rrr_vector_hip const * m_v;

public:
    typedef rrr_vector_hip bit_vector_type;
    typedef rrr_vector_hip::size_type size_type;
    explicit rank_support_rrr_hip(rrr_vector_hip const * v = nullptr) : m_v(v)
    {}
    //! answered by the host vector's own rank_support_rrr (rrr_vector.hpp:503-544)
    size_type rank(size_type i) const
    {
        return m_v->host().rank(i, t_b != 0);
    }
    size_type rank_on_device(size_type i) const
    {
        size_type r = 0;
        m_v->rank_batch<t_b>(&i, 1, &r);
        return r;
    }
    size_type operator()(size_type i) const
    {
        return rank(i);
    }
    void rank_batch(size_type const * i, size_t n, size_type * out, void * stream = nullptr) const
    {
        m_v->rank_batch<t_b>(i, n, out, stream);
    }
    size_type size() const
    {
        return m_v->size();
    }
    void set_vector(rrr_vector_hip const * v = nullptr)
    {
        m_v = v;
    }
};

//! select_support_rrr<t_b, 63> look-alike (rrr_vector.hpp:602)
template <uint8_t t_b = 1>
class select_support_rrr_hip
{
    rrr_vector_hip const * m_v;

public:
    typedef rrr_vector_hip bit_vector_type;
    typedef rrr_vector_hip::size_type size_type;
    explicit select_support_rrr_hip(rrr_vector_hip const * v = nullptr) : m_v(v)
    {}
    //! answered by the host vector's own select_support_rrr (rrr_vector.hpp:602)
    size_type select(size_type i) const
    {
        return m_v->host().select(i, t_b != 0);
    }
    size_type select_on_device(size_type i) const
    {
        size_type r = 0;
        m_v->select_batch<t_b>(&i, 1, &r);
        return r;
    }
    size_type operator()(size_type i) const
    {
        return select(i);
    }
    void select_batch(size_type const * i, size_t n, size_type * out, void * stream = nullptr) const
    {
        m_v->select_batch<t_b>(i, n, out, stream);
    }
    size_type size() const
    {
        return m_v->size();
    }
    void set_vector(rrr_vector_hip const * v = nullptr)
    {
        m_v = v;
    }
};

//! Device image of an sd_vector<> (sd_vector.hpp:134), built from the host object's own serialised arrays, from a
//! plain bit_vector or from a sorted position list; rank_support_sd / select_support_sd members in batch form.
//! Scalar members: the host sd_vector<> the adaptor was constructed from (non-owning pointer), else an sd_vector<> loaded from
//! the device image's serialised bytes on the first scalar call.
class sd_vector_hip
{
public:
    typedef uint64_t size_type;
    typedef sd_vector<> host_type;

private:
    struct deleter
    {
        void operator()(sdsl_hip_sd_s * p) const
        {
            sdsl_hip_sd_destroy(p);
        }
    };
    typedef hip_detail::lazy_host<hip_detail::host_bits> lazy_type;
    std::shared_ptr<sdsl_hip_sd_s> m_dev;
    std::shared_ptr<lazy_type> m_host;

    void host_from_device()
    {
        std::shared_ptr<sdsl_hip_sd_s> dev = m_dev;
        m_host = std::make_shared<lazy_type>([dev] {
            auto ser = [&](void * buf, size_t cap, size_t * len) { return sdsl_hip_sd_serialize(dev.get(), buf, cap, len); };
            return std::unique_ptr<hip_detail::host_bits>(
                new hip_detail::host_bits_of<host_type>(hip_detail::load_from_device<host_type>(ser, "sdsl_hip_sd_serialize")));
        });
    }

public:
    sd_vector_hip() = default;
    explicit sd_vector_hip(host_type const & v, int device = 0)
    {
        std::string s = hip_detail::to_stream(v);
        sdsl_hip_sd_t h = nullptr;
        hip_detail::check(sdsl_hip_sd_create_from_sdsl(s.data(), s.size(), device, &h, nullptr), "sdsl_hip_sd_create_from_sdsl");
        m_dev.reset(h, deleter());
        host_type const * p = &v;
        m_host = std::make_shared<lazy_type>(
            [p] { return std::unique_ptr<hip_detail::host_bits>(new hip_detail::host_bits_of<host_type>(p)); });
    }
    explicit sd_vector_hip(bit_vector const & bv, int device = 0)
    {
        sdsl_hip_sd_t h = nullptr;
        hip_detail::check(sdsl_hip_sd_create(bv.data(), bv.bit_size(), device, &h), "sdsl_hip_sd_create");
        m_dev.reset(h, deleter());
        host_from_device();
    }
    //! strictly increasing positions of the ones and the size of the vector (sd_vector_builder's arguments)
    sd_vector_hip(uint64_t const * positions, size_t m, size_type n, int device = 0)
    {
        sdsl_hip_sd_t h = nullptr;
        hip_detail::check(sdsl_hip_sd_create_from_positions(positions, m, n, device, &h), "sdsl_hip_sd_create_from_positions");
        m_dev.reset(h, deleter());
        host_from_device();
    }
    size_type size() const
    {
        return sdsl_hip_sd_size(m_dev.get());
    }
    template <uint8_t t_b>
    void rank_batch(size_type const * i, size_t n, size_type * out, void * stream = nullptr) const
    {
        hip_detail::check(sdsl_hip_sd_rank_batch(m_dev.get(), t_b, i, n, out, stream), "sdsl_hip_sd_rank_batch");
    }
    template <uint8_t t_b>
    void select_batch(size_type const * i, size_t n, size_type * out, void * stream = nullptr) const
    {
        hip_detail::check(sdsl_hip_sd_select_batch(m_dev.get(), t_b, i, n, out, stream), "sdsl_hip_sd_select_batch");
    }
    void access_batch(size_type const * i, size_t n, uint8_t * out, void * stream = nullptr) const
    {
        hip_detail::check(sdsl_hip_sd_access_batch(m_dev.get(), i, n, out, stream), "sdsl_hip_sd_access_batch");
    }
    hip_detail::host_bits const & host() const
    {
        return hip_detail::host_of(m_host, "sd_vector_hip");
    }
    bool operator[](size_type i) const
    {
        return host().access(i);
    }
    bool access_on_device(size_type i) const
    {
        uint8_t b = 0;
        access_batch(&i, 1, &b);
        return b != 0;
    }
};

//! rank_support_sd<t_b> look-alike (sd_vector.hpp:527) over an sd_vector_hip
template <uint8_t t_b = 1>
class rank_support_sd_hip
{
    sd_vector_hip const * m_v;

public:
    typedef sd_vector_hip bit_vector_type;
    typedef sd_vector_hip::size_type size_type;
    explicit rank_support_sd_hip(sd_vector_hip const * v = nullptr) : m_v(v)
    {}
    //! answered by the host vector's own rank_support_sd (sd_vector.hpp:527)
    size_type rank(size_type i) const
    {
        return m_v->host().rank(i, t_b != 0);
    }
    size_type rank_on_device(size_type i) const
    {
        size_type r = 0;
        m_v->rank_batch<t_b>(&i, 1, &r);
        return r;
    }
    size_type operator()(size_type i) const
    {
        return rank(i);
    }
    void rank_batch(size_type const * i, size_t n, size_type * out, void * stream = nullptr) const
    {
        m_v->rank_batch<t_b>(i, n, out, stream);
    }
    size_type size() const
    {
        return m_v->size();
    }
    void set_vector(sd_vector_hip const * v = nullptr)
    {
        m_v = v;
    }
};

//! select_support_sd<t_b> look-alike (sd_vector.hpp:676)
template <uint8_t t_b = 1>
class select_support_sd_hip
{
    sd_vector_hip const * m_v;

public:
    typedef sd_vector_hip bit_vector_type;
    typedef sd_vector_hip::size_type size_type;
    explicit select_support_sd_hip(sd_vector_hip const * v = nullptr) : m_v(v)
    {}
    //! answered by the host vector's own select_support_sd (sd_vector.hpp:676)
    size_type select(size_type i) const
    {
        return m_v->host().select(i, t_b != 0);
    }
    size_type select_on_device(size_type i) const
    {
        size_type r = 0;
        m_v->select_batch<t_b>(&i, 1, &r);
        return r;
    }
    size_type operator()(size_type i) const
    {
        return select(i);
    }
    void select_batch(size_type const * i, size_t n, size_type * out, void * stream = nullptr) const
    {
        m_v->select_batch<t_b>(i, n, out, stream);
    }
    size_type size() const
    {
        return m_v->size();
    }
    void set_vector(sd_vector_hip const * v = nullptr)
    {
        m_v = v;
    }
};

//! Device image of a byte wavelet tree of the wt_pc family (wt_huff / wt_blcd / wt_hutu share wt_pc's layout,
//! wt_pc.hpp:53-59), built from the host object's serialised form.  `layout` names the host type's bit vector and
//! select supports: SDSL_HIP_LAYOUT_BV_SCAN (0: bit_vector + rank_support_v5 + select_support_scan, zero bytes),
//! SDSL_HIP_LAYOUT_BV_MCL (1: ... + select_support_mcl, the wt_huff<bit_vector, rank_support_v5<>> default) or
//! SDSL_HIP_LAYOUT_RRR63 (2: wt_huff<rrr_vector<63>>).  `false` / `true` still mean 0 / 1.
class wt_huff_hip
{
public:
    typedef uint64_t size_type;
    typedef uint8_t value_type;

private:
    struct deleter
    {
        void operator()(sdsl_hip_wt_s * p) const
        {
            sdsl_hip_wt_destroy(p);
        }
    };
    typedef hip_detail::lazy_host<hip_detail::host_wt> lazy_type;
    std::shared_ptr<sdsl_hip_wt_s> m_dev;
    std::shared_ptr<lazy_type> m_host;

public:
    wt_huff_hip() = default;
    //! `wt` is the caller's tree: the device image is made from its serialised bytes, the scalar members below forward to it
    //! through a non-owning pointer (it must outlive them, as a bit_vector outlives its supports: rank_support.hpp:33)
    template <class t_wt>
    explicit wt_huff_hip(t_wt const & wt, int layout, int device = 0)
    {
        std::string s = hip_detail::to_stream(wt);
        sdsl_hip_wt_t h = nullptr;
        size_t used = 0;
        hip_detail::check(sdsl_hip_wt_create_from_sdsl(s.data(), s.size(), layout, device, &h, &used),
                          "sdsl_hip_wt_create_from_sdsl");
        m_dev.reset(h, deleter());
        t_wt const * p = &wt;
        m_host = std::make_shared<lazy_type>(
            [p] { return std::unique_ptr<hip_detail::host_wt>(new hip_detail::host_wt_of<t_wt>(p)); });
    }
    size_type size() const
    {
        return sdsl_hip_wt_size(m_dev.get());
    }
    hip_detail::host_wt const & host() const
    {
        return hip_detail::host_of(m_host, "wt_huff_hip");
    }
    //! wt.rank(i, c) (wt_pc.hpp:371-399), one query: the host tree's own walk
    size_type rank(size_type i, value_type c) const
    {
        return host().rank(i, c);
    }
    //! out[q] = wt.rank(i[q], c[q])   (wt_pc.hpp:371-399)
    void rank_batch(size_type const * i, value_type const * c, size_t n, size_type * out, void * stream = nullptr) const
    {
        hip_detail::check(sdsl_hip_wt_rank_batch(m_dev.get(), i, c, n, out, stream), "sdsl_hip_wt_rank_batch");
    }
    void access_batch(size_type const * i, size_t n, value_type * out, void * stream = nullptr) const
    {
        hip_detail::check(sdsl_hip_wt_access_batch(m_dev.get(), i, n, out, stream), "sdsl_hip_wt_access_batch");
    }
    value_type operator[](size_type i) const
    {
        return host().access(i);
    }
    void select_batch(size_type const * i, value_type const * c, size_t n, size_type * out, void * stream = nullptr) const
    {
        hip_detail::check(sdsl_hip_wt_select_batch(m_dev.get(), i, c, n, out, stream), "sdsl_hip_wt_select_batch");
    }
    size_type select(size_type i, value_type c) const
    {
        return host().select(i, c);
    }
    std::pair<size_type, value_type> inverse_select(size_type i) const
    {
        return host().inverse_select(i);
    }
    //! the same queries through the device (one launch + one synchronisation each): for checking the device image
    size_type rank_on_device(size_type i, value_type c) const
    {
        size_type r = 0;
        rank_batch(&i, &c, 1, &r);
        return r;
    }
    value_type access_on_device(size_type i) const
    {
        value_type c = 0;
        access_batch(&i, 1, &c);
        return c;
    }
    size_type select_on_device(size_type i, value_type c) const
    {
        size_type r = 0;
        select_batch(&i, &c, 1, &r);
        return r;
    }
    std::pair<size_type, value_type> inverse_select_on_device(size_type i) const
    {
        size_type r = 0;
        value_type c = 0;
        hip_detail::check(sdsl_hip_wt_inverse_select_batch(m_dev.get(), &i, 1, &r, &c, nullptr),
                          "sdsl_hip_wt_inverse_select_batch");
        return std::make_pair(r, c);
    }
};

//! Device image of a csa_wt over such a wavelet tree.  Scalar forms — csa[i], count / locate / extract of ONE pattern or range —
//! are answered by the caller's csa_wt (non-owning pointer, as above); the batch forms by the device.
class csa_wt_hip
{
public:
    typedef uint64_t size_type;

private:
    struct deleter
    {
        void operator()(sdsl_hip_fm_s * p) const
        {
            sdsl_hip_fm_destroy(p);
        }
    };
    typedef hip_detail::lazy_host<hip_detail::host_csa> lazy_type;
    std::shared_ptr<sdsl_hip_fm_s> m_dev;
    std::shared_ptr<lazy_type> m_host;

public:
    csa_wt_hip() = default;
    //! layout: as for wt_huff_hip (0 scan selects, 1 mcl selects, 2 = csa_wt<wt_huff<rrr_vector<63>>>)
    template <class t_csa>
    explicit csa_wt_hip(t_csa const & csa, int layout, int device = 0)
    {
        std::string s = hip_detail::to_stream(csa);
        sdsl_hip_fm_t h = nullptr;
        // the densities are template arguments of the host type, not part of its stream: hand them over so that the
        // SA / ISA samples are kept and sa_batch, isa_batch, locate_batch, extract_batch work on the device
        hip_detail::check(sdsl_hip_fm_create_from_sdsl_ex(s.data(), s.size(), layout, t_csa::sa_sample_dens,
                                                          t_csa::isa_sample_dens, device, &h),
                          "sdsl_hip_fm_create_from_sdsl_ex");
        m_dev.reset(h, deleter());
        t_csa const * p = &csa;
        m_host = std::make_shared<lazy_type>(
            [p] { return std::unique_ptr<hip_detail::host_csa>(new hip_detail::host_csa_of<t_csa>(p)); });
    }
    size_type size() const
    {
        return sdsl_hip_fm_size(m_dev.get());
    }
    hip_detail::host_csa const & host() const
    {
        return hip_detail::host_of(m_host, "csa_wt_hip");
    }
    //! csa[i] (csa_wt.hpp:363-381): the host index's own walk — use sa_batch for many
    size_type operator[](size_type i) const
    {
        return host().sa(i);
    }
    size_type sa_on_device(size_type i) const
    {
        size_type r = 0;
        sa_batch(&i, 1, &r);
        return r;
    }
    void sa_batch(size_type const * idx, size_t n, size_type * out, void * stream = nullptr) const
    {
        hip_detail::check(sdsl_hip_fm_sa_batch(m_dev.get(), idx, n, out, stream), "sdsl_hip_fm_sa_batch");
    }
    //! csa.isa[i], csa.lf[i], csa.psi[i] (suffix_array_helper.hpp:519-537, 346-360, 330-342), batched
    void isa_batch(size_type const * idx, size_t n, size_type * out, void * stream = nullptr) const
    {
        hip_detail::check(sdsl_hip_fm_isa_batch(m_dev.get(), idx, n, out, stream), "sdsl_hip_fm_isa_batch");
    }
    void lf_batch(size_type const * idx, size_t n, size_type * out, void * stream = nullptr) const
    {
        hip_detail::check(sdsl_hip_fm_lf_batch(m_dev.get(), idx, n, out, stream), "sdsl_hip_fm_lf_batch");
    }
    void psi_batch(size_type const * idx, size_t n, size_type * out, void * stream = nullptr) const
    {
        hip_detail::check(sdsl_hip_fm_psi_batch(m_dev.get(), idx, n, out, stream), "sdsl_hip_fm_psi_batch");
    }
    sdsl_hip_fm_t handle() const
    {
        return m_dev.get();
    }
    //! Resident bytes of the device image, and the two knobs that trade them against the speed of count():
    //! restore_suffix_array() reads the text back through the samples, sorts its suffixes on the device and keeps the whole suffix array,
    //! the text and a k-mer table beside the tree (8 bytes per symbol more: count() of large batches gets its fastest road, csa[i] and
    //! extract become gathers); set_footprint(max_bytes) gives memory back down to the host type's own footprint (csa_wt.hpp:389-402:
    //! tree + samples + alphabet), keeping the k-mer table the budget still holds.  Answers are the same in every state.
    size_type device_bytes() const
    {
        return sdsl_hip_fm_device_bytes(m_dev.get());
    }
    void restore_suffix_array()
    {
        hip_detail::check(sdsl_hip_fm_restore_suffix_array(m_dev.get()), "sdsl_hip_fm_restore_suffix_array");
    }
    void set_footprint(size_type max_bytes)
    {
        hip_detail::check(sdsl_hip_fm_set_footprint(m_dev.get(), max_bytes), "sdsl_hip_fm_set_footprint");
    }
};

//! The serialised form of wt_pc carries its node table, so the same adaptor serves every byte-alphabet shape:
//! wt_blcd<...> (wt_blcd.hpp:50) and wt_hutu<...> (wt_hutu.hpp) objects are handed over exactly like wt_huff<...>.
typedef wt_huff_hip wt_blcd_hip;
typedef wt_huff_hip wt_hutu_hip;

//! sdsl::count (suffix_array_algorithm.hpp:464-471) for n fixed-length patterns
inline void count_batch(csa_wt_hip const & csa, uint8_t const * patterns, uint32_t m, size_t n, uint64_t * out,
                        void * stream = nullptr)
{
    hip_detail::check(sdsl_hip_fm_count_batch(csa.handle(), patterns, m, n, out, stream), "sdsl_hip_fm_count_batch");
}
//! sdsl::count for one pattern, the reference's own call shape: answered by the caller's own csa_wt (one pattern is no batch)
template <class t_pat_iter>
inline uint64_t count(csa_wt_hip const & csa, t_pat_iter begin, t_pat_iter end)
{
    std::vector<uint8_t> p(begin, end);
    uint8_t dummy = 0;
    return csa.host().count(p.empty() ? &dummy : p.data(), p.size());
}
//! the same through the device (a ragged batch of one): for checking the device image
template <class t_pat_iter>
inline uint64_t count_on_device(csa_wt_hip const & csa, t_pat_iter begin, t_pat_iter end)
{
    std::vector<uint8_t> p(begin, end);
    uint64_t offs[2] = {0, p.size()}, r = 0;
    uint8_t dummy = 0;
    hip_detail::check(sdsl_hip_fm_count_ragged(csa.handle(), p.empty() ? &dummy : p.data(), offs, 1, &r, nullptr),
                      "sdsl_hip_fm_count_ragged");
    return r;
}
//! sdsl::locate (suffix_array_algorithm.hpp:505-523) for n fixed-length patterns: the occurrences of pattern p are
//! positions[offsets[p] .. offsets[p+1]), in SA order like the reference's
inline void locate_batch(csa_wt_hip const & csa, uint8_t const * patterns, uint32_t m, size_t n,
                         std::vector<uint64_t> & offsets, std::vector<uint64_t> & positions)
{
    std::vector<uint64_t> l(n), r(n);
    hip_detail::check(sdsl_hip_fm_interval_batch(csa.handle(), patterns, m, n, l.data(), r.data(), nullptr),
                      "sdsl_hip_fm_interval_batch");
    uint64_t total = 0;
    offsets.assign(n + 1, 0);
    hip_detail::check(sdsl_hip_fm_sa_range_batch(csa.handle(), l.data(), r.data(), n, offsets.data(), nullptr, 0, &total,
                                                 nullptr),
                      "sdsl_hip_fm_sa_range_batch");
    positions.assign(total, 0);
    if (total)
        hip_detail::check(sdsl_hip_fm_sa_range_batch(csa.handle(), l.data(), r.data(), n, nullptr, positions.data(), total,
                                                     &total, nullptr),
                          "sdsl_hip_fm_sa_range_batch");
}
//! sdsl::locate for one pattern, the reference's own call shape: the caller's own csa_wt answers
template <class t_pat_iter>
inline std::vector<uint64_t> locate(csa_wt_hip const & csa, t_pat_iter begin, t_pat_iter end)
{
    std::vector<uint8_t> p(begin, end);
    uint8_t dummy = 0;
    return csa.host().locate(p.empty() ? &dummy : p.data(), p.size());
}
//! the same through the device
template <class t_pat_iter>
inline std::vector<uint64_t> locate_on_device(csa_wt_hip const & csa, t_pat_iter begin, t_pat_iter end)
{
    std::vector<uint8_t> p(begin, end);
    std::vector<uint64_t> off, pos;
    if (p.empty())
    { // the empty pattern matches every suffix: SA[0..size)
        uint64_t l = 0, r = csa.size() - 1, total = 0;
        hip_detail::check(sdsl_hip_fm_sa_range_batch(csa.handle(), &l, &r, 1, nullptr, nullptr, 0, &total, nullptr),
                          "sdsl_hip_fm_sa_range_batch");
        pos.assign(total, 0);
        hip_detail::check(sdsl_hip_fm_sa_range_batch(csa.handle(), &l, &r, 1, nullptr, pos.data(), total, &total, nullptr),
                          "sdsl_hip_fm_sa_range_batch");
        return pos;
    }
    locate_batch(csa, p.data(), (uint32_t)p.size(), 1, off, pos);
    return pos;
}
//! sdsl::extract (suffix_array_algorithm.hpp:578-600) for n ranges [begin[q], end[q]] (inclusive): the text of range q
//! is text[offsets[q] .. offsets[q+1])
inline void extract_batch(csa_wt_hip const & csa, uint64_t const * begin, uint64_t const * end, size_t n,
                          std::vector<uint64_t> & offsets, std::vector<uint8_t> & text)
{
    uint64_t total = 0;
    offsets.assign(n + 1, 0);
    hip_detail::check(sdsl_hip_fm_extract_batch(csa.handle(), begin, end, n, offsets.data(), nullptr, 0, &total, nullptr),
                      "sdsl_hip_fm_extract_batch");
    text.assign(total, 0);
    if (total)
        hip_detail::check(sdsl_hip_fm_extract_batch(csa.handle(), begin, end, n, nullptr, text.data(), total, &total,
                                                    nullptr),
                          "sdsl_hip_fm_extract_batch");
}
//! sdsl::extract for one range, the reference's own call shape (returns the string): the caller's own csa_wt answers
inline std::string extract(csa_wt_hip const & csa, uint64_t begin, uint64_t end)
{
    return csa.host().extract(begin, end);
}
//! the same through the device
inline std::string extract_on_device(csa_wt_hip const & csa, uint64_t begin, uint64_t end)
{
    std::vector<uint64_t> off;
    std::vector<uint8_t> t;
    extract_batch(csa, &begin, &end, 1, off, t);
    return std::string(t.begin(), t.end());
}
//! backward_search(csa, l, r, c, l_res, r_res) (suffix_array_algorithm.hpp:167-201), batched
inline void backward_search_batch(csa_wt_hip const & csa, uint64_t const * l, uint64_t const * r, uint8_t const * c,
                                  size_t n, uint64_t * l_res, uint64_t * r_res, void * stream = nullptr)
{
    hip_detail::check(sdsl_hip_fm_backward_search_batch(csa.handle(), l, r, c, n, l_res, r_res, stream),
                      "sdsl_hip_fm_backward_search_batch");
}

// ---- several GPUs of one node ------------------------------------------------------------------
//! The devices a batch is sharded over.  Index structures built on a group are replicated (one RCCL broadcast at load
//! time); a batch call scatters the arguments from the root device, runs the single-GPU kernels everywhere and gathers
//! the answers (sdsl_hip.h, "several GPUs of one node").
class device_group
{
    struct deleter
    {
        void operator()(sdsl_hip_group_s * p) const
        {
            sdsl_hip_group_destroy(p);
        }
    };
    std::shared_ptr<sdsl_hip_group_s> m_g;

public:
    explicit device_group(std::vector<int32_t> const & devices)
    {
        sdsl_hip_group_t g = nullptr;
        hip_detail::check(sdsl_hip_group_create(devices.data(), (int32_t)devices.size(), &g), "sdsl_hip_group_create");
        m_g.reset(g, deleter());
    }
    int size() const
    {
        return sdsl_hip_group_size(m_g.get());
    }
    int device(int r) const
    {
        return sdsl_hip_group_device(m_g.get(), r);
    }
    sdsl_hip_group_t handle() const
    {
        return m_g.get();
    }
};

//! rank_support_v5<0/1> and select_support_mcl<0/1> of one bit_vector on every device of a group.
class bit_vector_multi_hip
{
public:
    typedef bit_vector::size_type size_type;

private:
    device_group m_g;
    std::shared_ptr<std::vector<sdsl_hip_bv_t>> m_rep;

public:
    bit_vector_multi_hip(bit_vector const & v, device_group const & g, uint32_t flags = SDSL_HIP_BV_SELECT1 | SDSL_HIP_BV_SELECT0)
        : m_g(g)
    {
        sdsl_hip_bv_t root = nullptr;
        hip_detail::check(sdsl_hip_bv_create(v.data(), v.bit_size(), g.device(0), flags, &root), "sdsl_hip_bv_create");
        m_rep.reset(new std::vector<sdsl_hip_bv_t>((size_t)g.size(), nullptr),
                    [](std::vector<sdsl_hip_bv_t> * r)
                    {
                        for (sdsl_hip_bv_t h : *r)
                            sdsl_hip_bv_destroy(h);
                        delete r;
                    });
        (*m_rep)[0] = root;
        hip_detail::check(sdsl_hip_group_bv_replicate(g.handle(), root, m_rep->data()), "sdsl_hip_group_bv_replicate");
    }
    //! out[q] = rank_support_v5<t_b>(&v)(idx[q]); arrays in host memory or in the root device's memory
    template <uint8_t t_b = 1>
    void rank_batch(size_type const * idx, size_t n, size_type * out, int chunks = 4) const
    {
        hip_detail::check(sdsl_hip_group_bv_rank_batch(m_g.handle(), m_rep->data(), t_b, idx, n, out, chunks),
                          "sdsl_hip_group_bv_rank_batch");
    }
    //! out[q] = select_support_mcl<t_b>(&v)(i[q])
    template <uint8_t t_b = 1>
    void select_batch(size_type const * i, size_t n, size_type * out, int chunks = 4) const
    {
        hip_detail::check(sdsl_hip_group_bv_select_batch(m_g.handle(), m_rep->data(), t_b, i, n, out, chunks),
                          "sdsl_hip_group_bv_select_batch");
    }
};

//! csa_wt<wt_huff<>> of one text on every device of a group (the text is broadcast once, every device lays out its own
//! index); count_batch shards the patterns.
class csa_wt_multi_hip
{
    device_group m_g;
    std::shared_ptr<std::vector<sdsl_hip_fm_t>> m_rep;

public:
    csa_wt_multi_hip(uint8_t const * text, size_t n, device_group const & g, uint32_t flags = 0) : m_g(g)
    {
        m_rep.reset(new std::vector<sdsl_hip_fm_t>((size_t)g.size(), nullptr),
                    [](std::vector<sdsl_hip_fm_t> * r)
                    {
                        for (sdsl_hip_fm_t h : *r)
                            sdsl_hip_fm_destroy(h);
                        delete r;
                    });
        hip_detail::check(sdsl_hip_group_fm_create_from_text(g.handle(), text, n, flags, m_rep->data()),
                          "sdsl_hip_group_fm_create_from_text");
    }
    uint64_t size() const
    {
        return sdsl_hip_fm_size((*m_rep)[0]);
    }
    //! sdsl::count (suffix_array_algorithm.hpp:464-471) for n patterns of m bytes each
    void count_batch(uint8_t const * patterns, uint32_t m, size_t n, uint64_t * out, int chunks = 4) const
    {
        hip_detail::check(sdsl_hip_group_fm_count_batch(m_g.handle(), m_rep->data(), patterns, m, n, out, chunks),
                          "sdsl_hip_group_fm_count_batch");
    }
};

} // namespace sdsl
