// sa.hip — suffix array / BWT construction of text+'\0' on the device by prefix doubling
// (Manber-Myers with radix sorts), feeding sdsl_hip_fm_create_from_text.
//
// This is SURVEY.md §8(f) row n4: the reference builds SA with divsufsort on one CPU core
// (construct_sa.hpp:120-153) and derives the BWT from it (construct_bwt.hpp:38-80); any correct
// suffix sorter yields the same SA, hence the same BWT, hence the same index.  The device version
// keeps everything in HBM: per round one 64-bit-key radix sort of n (key, suffix) pairs
// (rocPRIM — a plain library sort is exactly what this step is), one adjacent-difference + scan to
// re-rank, one gather to form the next keys.  k starts at 8 (first 8 bytes packed big-endian).
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>

#include "common.hpp"

#include <rocprim/device/device_radix_sort.hpp>
#include <rocprim/device/device_scan.hpp>

namespace sdslhip {

__global__ void k_sa_init_keys(const uint8_t * __restrict__ s, uint64_t n, uint64_t * __restrict__ keys,
                               uint32_t * __restrict__ idx)
{
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (uint64_t)gridDim.x * blockDim.x)
    {
        uint64_t k = 0;
#pragma unroll
        for (int j = 0; j < 8; ++j)
            k = (k << 8) | (i + j < n ? (uint64_t)s[i + j] : 0);
        keys[i] = k;
        idx[i] = (uint32_t)i;
    }
}

// flags[i] = 1 if sorted key i differs from sorted key i-1 (flags[0] = 0)
__global__ void k_sa_flags(const uint64_t * __restrict__ keys, uint64_t n, uint32_t * __restrict__ flags)
{
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (uint64_t)gridDim.x * blockDim.x)
        flags[i] = (i > 0 && keys[i] != keys[i - 1]) ? 1u : 0u;
}

// SA samples in SDSL's two layouts (csa_sampling_strategy.hpp:97-114, 755-777)
template <class SA>
__global__ void k_sa_sample(const SA * __restrict__ sa, uint64_t n, uint64_t dens, uint64_t * __restrict__ out)
{
    const uint64_t m = (n + dens - 1) / dens;
    for (uint64_t j = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; j < m; j += (uint64_t)gridDim.x * blockDim.x)
        out[j] = sa[j * dens];
}
template <class SA>
__global__ void k_isa_sample(const SA * __restrict__ sa, uint64_t n, uint64_t dens, uint64_t * __restrict__ out)
{
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (uint64_t)gridDim.x * blockDim.x)
    {
        uint64_t v = sa[i];
        if (v % dens == 0)
            out[v / dens] = i;
    }
}

sdsl_hip_status sa_samples_to_host(const uint32_t * d_sa, uint64_t n, uint64_t sa_dens, uint64_t isa_dens,
                                   std::vector<uint64_t> & sa_s, std::vector<uint64_t> & isa_s)
{
    const uint64_t ms = (n + sa_dens - 1) / sa_dens, mi = n ? (n - 1) / isa_dens + 1 : 0;
    DevBuf a, b;
    SH_TRY(a.alloc(ms * 8, true));
    SH_TRY(b.alloc(mi * 8, true));
    hipLaunchKernelGGL(k_sa_sample<uint32_t>, dim3(grid_for(ms, 256, 65536)), dim3(256), 0, 0, d_sa, n, sa_dens, a.as<uint64_t>());
    SH_HIP(hipGetLastError());
    hipLaunchKernelGGL(k_isa_sample<uint32_t>, dim3(grid_for(n, 256, 65536)), dim3(256), 0, 0, d_sa, n, isa_dens, b.as<uint64_t>());
    SH_HIP(hipGetLastError());
    sa_s.resize(ms);
    isa_s.resize(mi);
    if (ms)
        SH_HIP(hipMemcpy(sa_s.data(), a.p, ms * 8, hipMemcpyDeviceToHost));
    if (mi)
        SH_HIP(hipMemcpy(isa_s.data(), b.p, mi * 8, hipMemcpyDeviceToHost));
    return SDSL_HIP_OK;
}

template <class SA>
static sdsl_hip_status sa_samples_device_t(const SA * d_sa, uint64_t n, uint64_t sa_dens, uint64_t isa_dens, DevBuf * sa_s,
                                           DevBuf * isa_s)
{
    if (sa_s)
    {
        const uint64_t ms = (n + sa_dens - 1) / sa_dens;
        SH_TRY(sa_s->alloc(ms * 8, true));
        hipLaunchKernelGGL(k_sa_sample<SA>, dim3(grid_for(ms, 256, 65536)), dim3(256), 0, 0, d_sa, n, sa_dens,
                           sa_s->as<uint64_t>());
        SH_HIP(hipGetLastError());
    }
    if (isa_s)
    {
        const uint64_t mi = (n + isa_dens - 1) / isa_dens;
        SH_TRY(isa_s->alloc(mi * 8, true));
        hipLaunchKernelGGL(k_isa_sample<SA>, dim3(grid_for(n, 256, 65536)), dim3(256), 0, 0, d_sa, n, isa_dens,
                           isa_s->as<uint64_t>());
        SH_HIP(hipGetLastError());
    }
    SH_HIP(hipDeviceSynchronize());
    return SDSL_HIP_OK;
}

sdsl_hip_status sa_samples_device(const uint32_t * d_sa, uint64_t n, uint64_t sa_dens, uint64_t isa_dens, DevBuf * sa_s,
                                  DevBuf * isa_s)
{
    return sa_samples_device_t(d_sa, n, sa_dens, isa_dens, sa_s, isa_s);
}
sdsl_hip_status sa_samples_device64(const uint64_t * d_sa, uint64_t n, uint64_t sa_dens, uint64_t isa_dens, DevBuf * sa_s,
                                    DevBuf * isa_s)
{
    return sa_samples_device_t(d_sa, n, sa_dens, isa_dens, sa_s, isa_s);
}

size_t exclusive_scan_u64_temp_bytes(uint64_t n)
{
    size_t bytes = 0;
    if (n == 0 || rocprim::exclusive_scan(nullptr, bytes, (const uint64_t *)nullptr, (uint64_t *)nullptr, (uint64_t)0, (size_t)n,
                                          rocprim::plus<uint64_t>(), nullptr) != hipSuccess)
        return 16;
    return bytes ? bytes : 16;
}

// with the caller's working memory (>= exclusive_scan_u64_temp_bytes(n)): stream-ordered, no allocation, no synchronisation; without:
// a plain hipMalloc / hipFree around the scan and a stream synchronise before the memory goes back (see sort_pairs_u64_u32)
sdsl_hip_status exclusive_scan_u64(const uint64_t * in, uint64_t * out, uint64_t n, hipStream_t s, void * tmp_p, size_t tmp_bytes)
{
    if (n == 0)
        return SDSL_HIP_OK;
    size_t bytes = 0;
    SH_HIP(rocprim::exclusive_scan(nullptr, bytes, in, out, (uint64_t)0, (size_t)n, rocprim::plus<uint64_t>(), s));
    if (tmp_p && tmp_bytes >= bytes)
    {
        SH_HIP(rocprim::exclusive_scan(tmp_p, bytes, in, out, (uint64_t)0, (size_t)n, rocprim::plus<uint64_t>(), s));
        return SDSL_HIP_OK;
    }
    DevBuf tmp;
    SH_TRY(tmp.alloc(bytes ? bytes : 16));
    SH_HIP(rocprim::exclusive_scan(tmp.p, bytes, in, out, (uint64_t)0, (size_t)n, rocprim::plus<uint64_t>(), s));
    SH_HIP(hipStreamSynchronize(s)); // (tmp is freed on return)
    return SDSL_HIP_OK;
}

// rank[sa[i]] = scanned[i]
__global__ void k_sa_scatter_rank(const uint32_t * __restrict__ sa, const uint32_t * __restrict__ scanned, uint64_t n,
                                  uint32_t * __restrict__ rank)
{
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (uint64_t)gridDim.x * blockDim.x)
        rank[sa[i]] = scanned[i];
}

// keys for the next round, in suffix order i = 0..n-1: (rank[i], rank[i+k]+1 or 0 past the end)
__global__ void k_sa_next_keys(const uint32_t * __restrict__ rank, uint64_t n, uint64_t k, unsigned shift,
                               uint64_t * __restrict__ keys, uint32_t * __restrict__ idx)
{
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (uint64_t)gridDim.x * blockDim.x)
    {
        uint64_t second = i + k < n ? (uint64_t)rank[i + k] + 1 : 0;
        keys[i] = ((uint64_t)rank[i] << shift) | second;
        idx[i] = (uint32_t)i;
    }
}

__global__ void k_sa_bwt(const uint8_t * __restrict__ s, const uint32_t * __restrict__ sa, uint64_t n,
                         uint8_t * __restrict__ bwt)
{
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (uint64_t)gridDim.x * blockDim.x)
    {
        uint64_t p = sa[i];
        bwt[i] = s[p == 0 ? n - 1 : p - 1];
    }
}

// Sort of (u64 key, u32 value) pairs, used by fm.hip to order a pattern batch.  The working memory is the caller's (`tmp`, at
// least sort_pairs_u64_u32_temp_bytes(n, end_bit) bytes: the query path takes it from the device's scratch pool) or, with
// tmp == nullptr, a plain allocation that is freed — and the stream synchronised — before the call returns.
// NOT hipMallocAsync / hipFreeAsync: on the ROCm 7.2 runtime a loop of index builds through these helpers hung inside the level
// sort once in a few thousand builds, died with a GPU memory fault now and then and left wrong tables behind
// (tools/cpp/group_stress.cpp, phase D) — none of it since the stream-ordered allocator is out of the library.
size_t sort_pairs_u64_u32_temp_bytes(uint64_t n, unsigned end_bit)
{
    size_t bytes = 0;
    if (rocprim::radix_sort_pairs(nullptr, bytes, (uint64_t *)nullptr, (uint64_t *)nullptr, (uint32_t *)nullptr, (uint32_t *)nullptr, (size_t)n, 0u,
                                  end_bit) != hipSuccess)
        return 0;
    return bytes ? bytes : 16;
}
sdsl_hip_status sort_pairs_u64_u32(uint64_t * keys_in, uint64_t * keys_out, uint32_t * vals_in, uint32_t * vals_out,
                                   uint64_t n, unsigned end_bit, hipStream_t s, void * tmp, size_t tmp_bytes)
{
    size_t bytes = 0;
    SH_HIP(rocprim::radix_sort_pairs(nullptr, bytes, keys_in, keys_out, vals_in, vals_out, (size_t)n, 0u, end_bit, s));
    DevBuf own;
    if (!tmp)
    {
        SH_TRY(own.alloc(bytes ? bytes : 16));
        tmp = own.p;
        tmp_bytes = own.bytes;
    }
    if (tmp_bytes < bytes)
    {
        set_error("sort_pairs: working memory too small");
        return SDSL_HIP_ERR_INVALID;
    }
    SH_HIP(rocprim::radix_sort_pairs(tmp, bytes, keys_in, keys_out, vals_in, vals_out, (size_t)n, 0u, end_bit, s));
    if (own.p)
        SH_HIP(hipStreamSynchronize(s));
    return SDSL_HIP_OK;
}

// Stable sort of u16 keys by the bit range [begin_bit, end_bit) only (wt.hip: one wavelet-tree level).  The two key buffers are the
// sort's ping and pong (rocPRIM's double_buffer form: the working memory is histograms only — the plain form keeps a third buffer of n
// keys, 69 GB for a sequence of 2^35 symbols); *sorted_out = whichever of the two holds the result.
size_t sort_keys_u16_temp_bytes(uint64_t n, unsigned begin_bit, unsigned end_bit)
{
    size_t bytes = 0;
    rocprim::double_buffer<uint16_t> db((uint16_t *)nullptr, (uint16_t *)nullptr);
    if (rocprim::radix_sort_keys(nullptr, bytes, db, (size_t)n, begin_bit, end_bit) != hipSuccess)
        return 0;
    return bytes ? bytes : 16;
}
sdsl_hip_status sort_keys_u16(uint16_t * keys, uint16_t * other, uint64_t n, unsigned begin_bit, unsigned end_bit, hipStream_t s, void * tmp,
                              size_t tmp_bytes, uint16_t ** sorted_out)
{
    size_t bytes = 0;
    rocprim::double_buffer<uint16_t> db(keys, other);
    SH_HIP(rocprim::radix_sort_keys(nullptr, bytes, db, (size_t)n, begin_bit, end_bit, s));
    DevBuf own;
    if (!tmp)
    {
        SH_TRY(own.alloc(bytes ? bytes : 16));
        tmp = own.p;
        tmp_bytes = own.bytes;
    }
    if (tmp_bytes < bytes)
    {
        set_error("sort_keys: working memory too small");
        return SDSL_HIP_ERR_INVALID;
    }
    SH_HIP(rocprim::radix_sort_keys(tmp, bytes, db, (size_t)n, begin_bit, end_bit, s));
    *sorted_out = db.current();
    if (own.p)
        SH_HIP(hipStreamSynchronize(s));
    return SDSL_HIP_OK;
}

// BWT of text+'\\0' left in device memory (d_bwt, n_text+1 bytes); the suffix array (u32 per suffix) is handed over too
sdsl_hip_status sa_build_bwt_device(const uint8_t * text, uint64_t n_text, int device, DevBuf & d_bwt, DevBuf & d_sa)
{
    const uint64_t n = n_text + 1;
    if (n >= kLimSorter32Symbols)
    {
        set_error("device suffix sorter handles texts below 2^32-2 bytes (got %llu)", (unsigned long long)n_text);
        return SDSL_HIP_ERR_UNSUPPORTED;
    }
    SH_HIP(hipSetDevice(device));
    const bool trace = getenv("SDSL_HIP_TRACE_BUILD") != nullptr;
    auto t_prev = std::chrono::steady_clock::now();
    auto stamp = [&](const char * what) {
        if (!trace)
            return;
        (void)hipDeviceSynchronize();
        const auto now = std::chrono::steady_clock::now();
        fprintf(stderr, "[sdsl_hip] suffix sort: %s %.1f ms\n", what, std::chrono::duration<double, std::milli>(now - t_prev).count());
        t_prev = now;
    };
    DevBuf d_s, d_k0, d_k1, d_i0, d_i1, d_rank, d_flags, d_tmp;
    SH_TRY(d_s.alloc(n));
    SH_HIP(hipMemcpy(d_s.p, text, n_text, hipMemcpyDefault)); // host or device text
    SH_HIP(hipMemset((uint8_t *)d_s.p + n_text, 0, 1));
    SH_TRY(d_k0.alloc(n * 8));
    SH_TRY(d_k1.alloc(n * 8));
    SH_TRY(d_i0.alloc(n * 4));
    SH_TRY(d_i1.alloc(n * 4));
    SH_TRY(d_rank.alloc(n * 4));
    SH_TRY(d_flags.alloc(n * 4));
    uint64_t *k0 = d_k0.as<uint64_t>(), *k1 = d_k1.as<uint64_t>();
    uint32_t *i0 = d_i0.as<uint32_t>(), *i1 = d_i1.as<uint32_t>();
    uint32_t *rank = d_rank.as<uint32_t>(), *flags = d_flags.as<uint32_t>();
    const unsigned grid = grid_for(n, 256, 256u * 16u);

    unsigned rank_bits = 1;
    while ((UINT64_C(1) << rank_bits) < n + 1)
        ++rank_bits;

    size_t tmp_sort = 0, tmp_scan = 0;
    SH_HIP(rocprim::radix_sort_pairs(nullptr, tmp_sort, k0, k1, i0, i1, (size_t)n, 0u, 64u));
    SH_HIP(rocprim::inclusive_scan(nullptr, tmp_scan, flags, flags, (size_t)n, rocprim::plus<uint32_t>()));
    SH_TRY(d_tmp.alloc(std::max(tmp_sort, tmp_scan)));

    stamp("allocations + text copy");
    hipLaunchKernelGGL(k_sa_init_keys, dim3(grid), dim3(256), 0, 0, d_s.as<uint8_t>(), n, k0, i0);
    SH_HIP(hipGetLastError());
    uint64_t k = 8;
    unsigned end_bit = 64;
    for (int round = 0; round < 40; ++round)
    {
        size_t ts = d_tmp.bytes;
        SH_HIP(rocprim::radix_sort_pairs(d_tmp.p, ts, k0, k1, i0, i1, (size_t)n, 0u, end_bit));
        hipLaunchKernelGGL(k_sa_flags, dim3(grid), dim3(256), 0, 0, k1, n, flags);
        SH_HIP(hipGetLastError());
        ts = d_tmp.bytes;
        SH_HIP(rocprim::inclusive_scan(d_tmp.p, ts, flags, flags, (size_t)n, rocprim::plus<uint32_t>()));
        uint32_t max_rank = 0;
        SH_HIP(hipMemcpy(&max_rank, flags + (n - 1), 4, hipMemcpyDeviceToHost));
        if ((uint64_t)max_rank == n - 1)
            break; // all suffixes distinct: i1 is the suffix array
        if (k >= n)
        {
            set_error("suffix sorter did not converge (duplicate suffixes?)");
            return SDSL_HIP_ERR_HIP;
        }
        hipLaunchKernelGGL(k_sa_scatter_rank, dim3(grid), dim3(256), 0, 0, i1, flags, n, rank);
        SH_HIP(hipGetLastError());
        hipLaunchKernelGGL(k_sa_next_keys, dim3(grid), dim3(256), 0, 0, rank, n, k, rank_bits, k0, i0);
        SH_HIP(hipGetLastError());
        end_bit = 2 * rank_bits;
        k <<= 1;
    }
    stamp("doubling rounds");
    SH_TRY(d_bwt.alloc(n));
    hipLaunchKernelGGL(k_sa_bwt, dim3(grid), dim3(256), 0, 0, d_s.as<uint8_t>(), i1, n, d_bwt.as<uint8_t>());
    SH_HIP(hipGetLastError());
    SH_HIP(hipDeviceSynchronize());
    d_sa = std::move(d_i1);
    stamp("bwt");
    return SDSL_HIP_OK;
}

// ---- texts of 2^32 - 2 bytes and more: 64-bit suffixes ------------------------------------------------------------------
// Two ranks of 33 bits and more do not share one 64-bit key, so a doubling round sorts twice (LSD: by the rank k symbols
// ahead, then — the sort is stable — by the suffix's own rank); keys and suffixes live in rocPRIM double buffers, so the sorts
// need no pair-sized temporary: 5 arrays of 8 bytes per symbol (two key buffers, two suffix buffers, the ranks), the flags of
// a round and their scan sitting in the key buffer that is free at that moment.  The reference sorts with 64-bit divsufsort above
// 2^31 symbols (construct_sa.hpp:120-153) — any correct sorter gives the same array.
__global__ void k_sa64_init_keys(const uint8_t * __restrict__ s, uint64_t n, uint64_t * __restrict__ keys, uint64_t * __restrict__ idx)
{
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (uint64_t)gridDim.x * blockDim.x)
    {
        uint64_t k = 0;
#pragma unroll
        for (int j = 0; j < 8; ++j)
            k = (k << 8) | (i + j < n ? (uint64_t)s[i + j] : 0);
        keys[i] = k;
        idx[i] = i;
    }
}
// flags[j] = 1 if the j-th suffix in sorted order differs from its predecessor in (own key, key `ahead`)
__global__ void k_sa64_flags(const uint64_t * __restrict__ key, const uint64_t * __restrict__ sa, const uint64_t * __restrict__ rank,
                             uint64_t n, uint64_t ahead, uint64_t * __restrict__ flags)
{
    for (uint64_t j = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; j < n; j += (uint64_t)gridDim.x * blockDim.x)
    {
        bool diff = false;
        if (j > 0)
        {
            diff = key[j] != key[j - 1];
            if (!diff && ahead)
            { // (ahead == 0: the first round, whose key is the whole comparison)
                const uint64_t a = sa[j] + ahead, b = sa[j - 1] + ahead;
                diff = (a < n ? rank[a] + 1 : 0) != (b < n ? rank[b] + 1 : 0);
            }
        }
        flags[j] = diff ? 1u : 0u;
    }
}
__global__ void k_sa64_scatter_rank(const uint64_t * __restrict__ sa, const uint64_t * __restrict__ scanned, uint64_t n,
                                    uint64_t * __restrict__ rank)
{
    for (uint64_t j = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; j < n; j += (uint64_t)gridDim.x * blockDim.x)
        rank[sa[j]] = scanned[j];
}
__global__ void k_sa64_keys_ahead(const uint64_t * __restrict__ rank, uint64_t n, uint64_t k, uint64_t * __restrict__ keys,
                                  uint64_t * __restrict__ idx)
{
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (uint64_t)gridDim.x * blockDim.x)
    {
        keys[i] = i + k < n ? rank[i + k] + 1 : 0;
        idx[i] = i;
    }
}
__global__ void k_sa64_keys_own(const uint64_t * __restrict__ rank, const uint64_t * __restrict__ sa, uint64_t n, uint64_t * __restrict__ keys)
{
    for (uint64_t j = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; j < n; j += (uint64_t)gridDim.x * blockDim.x)
        keys[j] = rank[sa[j]];
}
__global__ void k_sa64_bwt(const uint8_t * __restrict__ s, const uint64_t * __restrict__ sa, uint64_t n, uint8_t * __restrict__ bwt)
{
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (uint64_t)gridDim.x * blockDim.x)
    {
        const uint64_t p = sa[i];
        bwt[i] = s[p == 0 ? n - 1 : p - 1];
    }
}

sdsl_hip_status sa_build_bwt_device64(const uint8_t * text, uint64_t n_text, int device, DevBuf & d_bwt, DevBuf & d_sa)
{
    const uint64_t n = n_text + 1;
    if (n >= kLimSorter64Symbols)
    {
        set_error("device suffix sorter handles texts below 2^40 bytes (got %llu)", (unsigned long long)n_text);
        return SDSL_HIP_ERR_UNSUPPORTED;
    }
    SH_HIP(hipSetDevice(device));
    const bool trace = getenv("SDSL_HIP_TRACE_BUILD") != nullptr;
    auto t_prev = std::chrono::steady_clock::now();
    auto stamp = [&](const char * what) {
        if (!trace)
            return;
        (void)hipDeviceSynchronize();
        const auto now = std::chrono::steady_clock::now();
        fprintf(stderr, "[sdsl_hip] suffix sort (64-bit): %s %.1f ms\n", what, std::chrono::duration<double, std::milli>(now - t_prev).count());
        t_prev = now;
    };
    DevBuf d_s, d_k0, d_k1, d_i0, d_i1, d_rank, d_tmp;
    SH_TRY(d_s.alloc(n));
    SH_HIP(hipMemcpy(d_s.p, text, n_text, hipMemcpyDefault)); // host or device text
    SH_HIP(hipMemset((uint8_t *)d_s.p + n_text, 0, 1));
    SH_TRY(d_k0.alloc(n * 8));
    SH_TRY(d_k1.alloc(n * 8));
    SH_TRY(d_i0.alloc(n * 8));
    SH_TRY(d_i1.alloc(n * 8));
    SH_TRY(d_rank.alloc(n * 8));
    rocprim::double_buffer<uint64_t> keys(d_k0.as<uint64_t>(), d_k1.as<uint64_t>()), sufs(d_i0.as<uint64_t>(), d_i1.as<uint64_t>());
    uint64_t * rank = d_rank.as<uint64_t>();
    const unsigned grid = grid_for(n, 256, 256u * 16u);
    unsigned rank_bits = 1;
    while ((UINT64_C(1) << rank_bits) < n + 1)
        ++rank_bits;
    size_t tmp_sort = 0, tmp_scan = 0;
    SH_HIP(rocprim::radix_sort_pairs(nullptr, tmp_sort, keys, sufs, (size_t)n, 0u, 64u));
    SH_HIP(rocprim::inclusive_scan(nullptr, tmp_scan, keys.current(), keys.alternate(), (size_t)n, rocprim::plus<uint64_t>()));
    SH_TRY(d_tmp.alloc(std::max<size_t>(std::max(tmp_sort, tmp_scan), 16)));
    stamp("allocations + text copy");
    hipLaunchKernelGGL(k_sa64_init_keys, dim3(grid), dim3(256), 0, 0, d_s.as<uint8_t>(), n, keys.current(), sufs.current());
    SH_HIP(hipGetLastError());
    size_t ts = d_tmp.bytes;
    SH_HIP(rocprim::radix_sort_pairs(d_tmp.p, ts, keys, sufs, (size_t)n, 0u, 64u));
    uint64_t k = 8, ahead = 0; // the sorted order reflects the first k symbols; `ahead`: what the last sort's second key looked at
    for (int round = 0; round < 48; ++round)
    {
        // new ranks: flags and their scan in the key buffer that is free now
        uint64_t * flags = keys.alternate();
        hipLaunchKernelGGL(k_sa64_flags, dim3(grid), dim3(256), 0, 0, keys.current(), sufs.current(), rank, n, ahead, flags);
        SH_HIP(hipGetLastError());
        ts = d_tmp.bytes;
        SH_HIP(rocprim::inclusive_scan(d_tmp.p, ts, flags, flags, (size_t)n, rocprim::plus<uint64_t>()));
        uint64_t max_rank = 0;
        SH_HIP(hipMemcpy(&max_rank, flags + (n - 1), 8, hipMemcpyDeviceToHost));
        if (trace)
            fprintf(stderr, "[sdsl_hip] suffix sort (64-bit): %llu symbols compared, %llu of %llu ranks\n", (unsigned long long)k,
                    (unsigned long long)max_rank + 1, (unsigned long long)n);
        if (max_rank == n - 1)
            break; // all suffixes distinct: sufs.current() is the suffix array
        if (k >= n)
        {
            set_error("suffix sorter did not converge (duplicate suffixes?)");
            return SDSL_HIP_ERR_HIP;
        }
        hipLaunchKernelGGL(k_sa64_scatter_rank, dim3(grid), dim3(256), 0, 0, sufs.current(), flags, n, rank);
        SH_HIP(hipGetLastError());
        // sort by (rank[i], rank[i + k]): least significant key first
        hipLaunchKernelGGL(k_sa64_keys_ahead, dim3(grid), dim3(256), 0, 0, rank, n, k, keys.current(), sufs.current());
        SH_HIP(hipGetLastError());
        ts = d_tmp.bytes;
        SH_HIP(rocprim::radix_sort_pairs(d_tmp.p, ts, keys, sufs, (size_t)n, 0u, rank_bits));
        hipLaunchKernelGGL(k_sa64_keys_own, dim3(grid), dim3(256), 0, 0, rank, sufs.current(), n, keys.current());
        SH_HIP(hipGetLastError());
        ts = d_tmp.bytes;
        SH_HIP(rocprim::radix_sort_pairs(d_tmp.p, ts, keys, sufs, (size_t)n, 0u, rank_bits));
        ahead = k;
        k <<= 1;
    }
    stamp("doubling rounds");
    d_k0.release(); // (the BWT and whatever the caller builds next need the room)
    d_k1.release();
    d_rank.release();
    const bool in_first = sufs.current() == d_i0.as<uint64_t>();
    (in_first ? d_i1 : d_i0).release();
    SH_TRY(d_bwt.alloc(n));
    hipLaunchKernelGGL(k_sa64_bwt, dim3(grid), dim3(256), 0, 0, d_s.as<uint8_t>(), sufs.current(), n, d_bwt.as<uint8_t>());
    SH_HIP(hipGetLastError());
    SH_HIP(hipDeviceSynchronize());
    d_sa = std::move(in_first ? d_i0 : d_i1);
    stamp("bwt");
    return SDSL_HIP_OK;
}


} // namespace sdslhip
