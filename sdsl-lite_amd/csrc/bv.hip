// bv.hip — plain bit vector on the device: construction of the rank-line layout and of the
// select sample directories, and the batched rank / select / access kernels.
//
// Reference semantics reproduced (results are bit-identical, the layout is not):
//   rank_support_v5<b>::rank            rank_support_v5.hpp:131-149
//   select_support_mcl<b>::select       select_support_mcl.hpp:384-439
//   bit_vector::operator[]              int_vector.hpp:1900-1904
#include <algorithm>

#include "bv_host.hpp"
#include "bv_serialize.hpp"

namespace sdslhip {

__global__ __launch_bounds__(256) void k_fill_u32(uint32_t * __restrict__ p, uint32_t word, uint64_t n)
{
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (uint64_t)gridDim.x * blockDim.x)
        p[i] = word;
}
sdsl_hip_status fill_u32_async(void * p, uint32_t word, size_t bytes, hipStream_t s)
{
    if (bytes == 0)
        return SDSL_HIP_OK;
    const uint64_t n = bytes / 4;
    hipLaunchKernelGGL(k_fill_u32, dim3(grid_for(n, 256, 256u * 8u)), dim3(256), 0, s, (uint32_t *)p, word, n);
    SH_HIP(hipGetLastError());
    return SDSL_HIP_OK;
}


// =========================================================================================
// construction
// =========================================================================================

// one thread per line: copy the 7 data words (masked to n_bits) and record the line popcount
__global__ __launch_bounds__(256) void k_build_lines(const uint64_t * __restrict__ words, uint64_t n_bits,
                                                     uint64_t n_lines, uint64_t * __restrict__ lines,
                                                     uint32_t * __restrict__ cnts)
{
    const uint64_t n_words = (n_bits + 63) >> 6;
    for (uint64_t L = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; L < n_lines;
         L += (uint64_t)gridDim.x * blockDim.x)
    {
        uint32_t c = 0;
#pragma unroll
        for (int d = 0; d < kDW; ++d)
        {
            uint64_t wi = L * kDW + d;
            uint64_t w = wi < n_words ? words[wi] : 0;
            if (wi + 1 == n_words && (n_bits & 63))
                w &= lo_set((unsigned)(n_bits & 63));
            lines[L * kLW + 1 + d] = w;
            c += popc64(w);
        }
        cnts[L] = c;
    }
}

// ones per line, from the lines themselves (a select directory added after the build: sdsl_hip_bv_add_select)
__global__ __launch_bounds__(256) void k_line_counts(const uint64_t * __restrict__ lines, uint64_t n_lines, uint32_t * __restrict__ cnts)
{
    for (uint64_t L = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; L < n_lines; L += (uint64_t)gridDim.x * blockDim.x)
    {
        uint32_t c = 0;
#pragma unroll
        for (int d = 0; d < kDW; ++d)
            c += popc64(lines[L * kLW + 1 + d]);
        cnts[L] = c;
    }
}

constexpr int kScanPerThread = 8;
constexpr int kScanPerBlock = 256 * kScanPerThread;

__device__ __forceinline__ uint64_t block_excl_scan_256(uint64_t v, uint64_t * sh, uint64_t & total)
{
    // simple LDS Hillis-Steele over 256 thread sums
    const int t = threadIdx.x;
    sh[t] = v;
    __syncthreads();
    for (int d = 1; d < 256; d <<= 1)
    {
        uint64_t x = t >= d ? sh[t - d] : 0;
        __syncthreads();
        sh[t] += x;
        __syncthreads();
    }
    total = sh[255];
    uint64_t incl = sh[t];
    __syncthreads();
    return incl - v;
}

__global__ __launch_bounds__(256) void k_scan_reduce(const uint32_t * __restrict__ in, uint64_t n,
                                                     uint64_t * __restrict__ bsum)
{
    __shared__ uint64_t sh[256];
    uint64_t base = (uint64_t)blockIdx.x * kScanPerBlock + (uint64_t)threadIdx.x * kScanPerThread;
    uint64_t s = 0;
#pragma unroll
    for (int i = 0; i < kScanPerThread; ++i)
        if (base + i < n)
            s += in[base + i];
    uint64_t tot;
    (void)block_excl_scan_256(s, sh, tot);
    if (threadIdx.x == 0)
        bsum[blockIdx.x] = tot;
}

// single block: exclusive scan of bsum[0..nb) in place, total written to bsum[nb]
__global__ __launch_bounds__(256) void k_scan_top(uint64_t * bsum, uint64_t nb)
{
    __shared__ uint64_t sh[256];
    uint64_t carry = 0;
    for (uint64_t base = 0; base < nb; base += 256)
    {
        uint64_t i = base + threadIdx.x;
        uint64_t v = i < nb ? bsum[i] : 0;
        uint64_t tot;
        uint64_t ex = block_excl_scan_256(v, sh, tot);
        if (i < nb)
            bsum[i] = carry + ex;
        carry += tot;
    }
    if (threadIdx.x == 0)
        bsum[nb] = carry;
}

// out[i*stride] = exclusive prefix of in[0..i)
__global__ __launch_bounds__(256) void k_scan_apply(const uint32_t * __restrict__ in, uint64_t n,
                                                    const uint64_t * __restrict__ bsum, uint64_t * __restrict__ out,
                                                    uint64_t stride)
{
    __shared__ uint64_t sh[256];
    uint64_t base = (uint64_t)blockIdx.x * kScanPerBlock + (uint64_t)threadIdx.x * kScanPerThread;
    uint32_t v[kScanPerThread];
    uint64_t s = 0;
#pragma unroll
    for (int i = 0; i < kScanPerThread; ++i)
    {
        v[i] = base + i < n ? in[base + i] : 0;
        s += v[i];
    }
    uint64_t tot;
    uint64_t ex = block_excl_scan_256(s, sh, tot) + bsum[blockIdx.x];
#pragma unroll
    for (int i = 0; i < kScanPerThread; ++i)
    {
        if (base + i < n)
            out[(base + i) * stride] = ex;
        ex += v[i];
    }
}

// out[i * stride] = sum of in[0..i); *total (host) = sum of all.  Asynchronous on the null stream except for the total.
sdsl_hip_status device_exclusive_scan_u32(const uint32_t * d_in, uint64_t n, uint64_t * d_out, uint64_t stride,
                                          uint64_t * total)
{
    const uint64_t nb = (n + kScanPerBlock - 1) / kScanPerBlock;
    DevBuf bsum;
    SH_TRY(bsum.alloc((nb + 1) * sizeof(uint64_t), true));
    if (n)
    {
        hipLaunchKernelGGL(k_scan_reduce, dim3((unsigned)nb), dim3(256), 0, 0, d_in, n, bsum.as<uint64_t>());
        SH_HIP(hipGetLastError());
        hipLaunchKernelGGL(k_scan_top, dim3(1), dim3(256), 0, 0, bsum.as<uint64_t>(), nb);
        SH_HIP(hipGetLastError());
        hipLaunchKernelGGL(k_scan_apply, dim3((unsigned)nb), dim3(256), 0, 0, d_in, n, bsum.as<uint64_t>(), d_out, stride);
        SH_HIP(hipGetLastError());
    }
    if (total)
        SH_HIP(hipMemcpy(total, bsum.as<uint64_t>() + nb, sizeof(uint64_t), hipMemcpyDeviceToHost));
    else
        SH_HIP(hipDeviceSynchronize()); // bsum is released on return
    return SDSL_HIP_OK;
}

// select samples: sample[j] = (position of the BIT-argument of 0-based rank j << shift) >> pshift
template <int BIT>
__global__ __launch_bounds__(256) void k_build_sel(const uint64_t * __restrict__ lines,
                                                   const uint32_t * __restrict__ cnts, uint64_t n_bits,
                                                   uint64_t n_lines, uint32_t shift, uint32_t pshift,
                                                   uint32_t * __restrict__ sample)
{
    for (uint64_t L = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; L < n_lines;
         L += (uint64_t)gridDim.x * blockDim.x)
    {
        const uint64_t start = L * kDB;
        if (start >= n_bits)
            continue;
        const uint64_t valid = n_bits - start < kDB ? n_bits - start : kDB;
        const uint64_t h1 = lines[L * kLW], c1 = cnts[L];
        const uint64_t h = BIT ? h1 : start - h1;
        const uint64_t c = BIT ? c1 : valid - c1;
        const uint64_t S = UINT64_C(1) << shift;
        uint64_t j = (h + S - 1) >> shift;
        if (c == 0 || (j << shift) >= h + c)
            continue;
        uint64_t w[kDW];
#pragma unroll
        for (int d = 0; d < kDW; ++d)
        {
            uint64_t x = lines[L * kLW + 1 + d];
            if (!BIT)
            {
                uint64_t ws = 64ull * d;
                uint64_t m = ws >= valid ? 0 : (valid - ws >= 64 ? ~UINT64_C(0) : lo_set((unsigned)(valid - ws)));
                x = ~x & m;
            }
            w[d] = x;
        }
        for (; (j << shift) < h + c; ++j)
        {
            unsigned r = (unsigned)((j << shift) - h); // 0-based rank inside the line
            uint64_t pos = 0;
#pragma unroll
            for (int d = 0; d < kDW; ++d)
            {
                unsigned pc = popc64(w[d]);
                if (r < pc)
                {
                    pos = start + 64ull * d + sel64(w[d], r + 1);
                    r = 0xFFFFFFFFu; // found
                }
                else if (r != 0xFFFFFFFFu)
                    r -= pc;
            }
            sample[j] = (uint32_t)(pos >> pshift);
        }
    }
}

__global__ void k_set_u32(uint32_t * p, uint32_t v)
{
    *p = v;
}

__global__ __launch_bounds__(256) void k_fill_u32(uint32_t * p, uint64_t n, uint32_t v)
{
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (uint64_t)gridDim.x * blockDim.x)
        p[i] = v;
}

// positions of ALL arguments of the long sample intervals (one thread per line; a line contributes to such an interval
// only with a few arguments), plus, as entry S of every long interval, the first position of the next interval
template <int BIT>
__global__ __launch_bounds__(256) void k_build_sel_long(const uint64_t * __restrict__ lines, const uint32_t * __restrict__ cnts,
                                                        uint64_t n_bits, uint64_t n_lines, uint32_t shift, uint32_t pshift,
                                                        const uint32_t * __restrict__ sample, uint64_t ns,
                                                        const uint32_t * __restrict__ lmask, const uint32_t * __restrict__ lidx,
                                                        uint32_t * __restrict__ lpos)
{
    const uint64_t S = UINT64_C(1) << shift;
    for (uint64_t j = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; j < ns; j += (uint64_t)gridDim.x * blockDim.x)
        if ((lmask[j >> 5] >> (j & 31)) & 1u)
            lpos[(uint64_t)lidx[j] * (S + 1) + S] = sample[j + 1];
    for (uint64_t L = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; L < n_lines; L += (uint64_t)gridDim.x * blockDim.x)
    {
        const uint64_t start = L * kDB;
        if (start >= n_bits)
            continue;
        const uint64_t valid = n_bits - start < kDB ? n_bits - start : kDB;
        const uint64_t h1 = lines[L * kLW], c1 = cnts[L];
        const uint64_t h = BIT ? h1 : start - h1;
        const uint64_t c = BIT ? c1 : valid - c1;
        if (c == 0)
            continue;
        // is any interval this line's arguments fall into a long one?
        const uint64_t j0 = h >> shift, j1 = (h + c - 1) >> shift;
        bool any = false;
        for (uint64_t j = j0; j <= j1; ++j)
            any |= ((lmask[j >> 5] >> (j & 31)) & 1u) != 0;
        if (!any)
            continue;
        uint64_t k = h;
        for (int d = 0; d < kDW; ++d)
        {
            uint64_t x = lines[L * kLW + 1 + d];
            if (!BIT)
            {
                const uint64_t ws = 64ull * d;
                const uint64_t m = ws >= valid ? 0 : (valid - ws >= 64 ? ~UINT64_C(0) : lo_set((unsigned)(valid - ws)));
                x = ~x & m;
            }
            while (x)
            {
                const unsigned b = (unsigned)__ffsll((long long)x) - 1;
                const uint64_t j = k >> shift;
                if ((lmask[j >> 5] >> (j & 31)) & 1u)
                    lpos[(uint64_t)lidx[j] * (S + 1) + (k & (S - 1))] = (uint32_t)((start + 64ull * d + b) >> pshift);
                ++k;
                x &= x - 1;
            }
        }
    }
}

sdsl_hip_status build_select_dir(BvHost & bv, int bit)
{
    uint64_t total = bit ? bv.view.ones : bv.view.n_bits - bv.view.ones;
    uint32_t sh = bv.view.sel_shift;
    uint64_t ns = (total + (UINT64_C(1) << sh) - 1) >> sh;
    SH_TRY(bv.sel[bit].alloc((ns + 2) * sizeof(uint32_t), true));
    uint32_t * smp = bv.sel[bit].as<uint32_t>();
    unsigned grid = grid_for(bv.view.n_lines, 256, 65536);
    if (bit)
        hipLaunchKernelGGL(k_build_sel<1>, dim3(grid), dim3(256), 0, 0, bv.view.lines, bv.cnts.as<uint32_t>(),
                           bv.view.n_bits, bv.view.n_lines, sh, bv.view.sel_pshift, smp);
    else
        hipLaunchKernelGGL(k_build_sel<0>, dim3(grid), dim3(256), 0, 0, bv.view.lines, bv.cnts.as<uint32_t>(),
                           bv.view.n_bits, bv.view.n_lines, sh, bv.view.sel_pshift, smp);
    SH_HIP(hipGetLastError());
    hipLaunchKernelGGL(k_set_u32, dim3(1), dim3(1), 0, 0, smp + ns, (uint32_t)(bv.view.n_bits >> bv.view.sel_pshift));
    SH_HIP(hipGetLastError());
    bv.view.sel[bit] = smp;
    // sparse stretches: intervals whose arguments lie more than kSelLongGap bits apart on average keep every position
    bv.view.lmask[bit] = bv.view.lidx[bit] = bv.view.lpos[bit] = nullptr;
    if (ns == 0 || getenv("SDSL_HIP_SELECT_NO_LONG"))
        return SDSL_HIP_OK;
    std::vector<uint32_t> h_smp(ns + 1);
    SH_HIP(hipMemcpy(h_smp.data(), smp, (ns + 1) * 4, hipMemcpyDeviceToHost));
    std::vector<uint32_t> h_mask((ns + 31) / 32 + 1, 0), h_idx(ns, 0);
    const uint64_t S = UINT64_C(1) << sh;
    uint64_t n_long = 0;
    for (uint64_t j = 0; j < ns; ++j)
    {
        const uint64_t cnt = std::min<uint64_t>(S, total - j * S);
        const uint64_t span = ((uint64_t)h_smp[j + 1] - h_smp[j]) << bv.view.sel_pshift;
        if (cnt > 1 && span > kSelLongGap * cnt)
        {
            h_mask[j >> 5] |= 1u << (j & 31);
            h_idx[j] = (uint32_t)n_long++;
        }
    }
    if (n_long == 0)
        return SDSL_HIP_OK;
    SH_TRY(bv.lmask[bit].alloc(h_mask.size() * 4));
    SH_TRY(bv.lidx[bit].alloc(h_idx.size() * 4));
    SH_TRY(bv.lpos[bit].alloc(n_long * (S + 1) * 4));
    SH_HIP(hipMemcpy(bv.lmask[bit].p, h_mask.data(), h_mask.size() * 4, hipMemcpyHostToDevice));
    SH_HIP(hipMemcpy(bv.lidx[bit].p, h_idx.data(), h_idx.size() * 4, hipMemcpyHostToDevice));
    const uint32_t sentinel = (uint32_t)(bv.view.n_bits >> bv.view.sel_pshift);
    hipLaunchKernelGGL(k_fill_u32, dim3(grid_for(n_long * (S + 1), 256, 65536)), dim3(256), 0, 0, bv.lpos[bit].as<uint32_t>(),
                       n_long * (S + 1), sentinel);
    SH_HIP(hipGetLastError());
    if (bit)
        hipLaunchKernelGGL(k_build_sel_long<1>, dim3(grid), dim3(256), 0, 0, bv.view.lines, bv.cnts.as<uint32_t>(), bv.view.n_bits,
                           bv.view.n_lines, sh, bv.view.sel_pshift, smp, ns, bv.lmask[bit].as<uint32_t>(), bv.lidx[bit].as<uint32_t>(),
                           bv.lpos[bit].as<uint32_t>());
    else
        hipLaunchKernelGGL(k_build_sel_long<0>, dim3(grid), dim3(256), 0, 0, bv.view.lines, bv.cnts.as<uint32_t>(), bv.view.n_bits,
                           bv.view.n_lines, sh, bv.view.sel_pshift, smp, ns, bv.lmask[bit].as<uint32_t>(), bv.lidx[bit].as<uint32_t>(),
                           bv.lpos[bit].as<uint32_t>());
    SH_HIP(hipGetLastError());
    bv.view.lmask[bit] = bv.lmask[bit].as<uint32_t>();
    bv.view.lidx[bit] = bv.lidx[bit].as<uint32_t>();
    bv.view.lpos[bit] = bv.lpos[bit].as<uint32_t>();
    return SDSL_HIP_OK;
}

// The occurrences of a two-bit pattern as a bit vector: bit i is set iff (x[i-1], x[i]) is the pattern, with SDSL's
// convention for the bit in front of position 0 (rank_support.hpp:160-284: init_carry 0 for 10 and 11, 1 for 01 and
// 00; bits.hpp:558-583 map10 / map01).  rank / select on pattern supports are rank_1 / select_1 on this vector.
// pat: 0 = "10", 1 = "01", 2 = "00", 3 = "11"
__global__ __launch_bounds__(256) void k_pattern_words(const uint64_t * __restrict__ x, uint64_t n_bits, uint64_t nw, int pat,
                                                       uint64_t * __restrict__ d)
{
    for (uint64_t w = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; w < nw; w += (uint64_t)gridDim.x * blockDim.x)
    {
        const uint64_t cur = x[w];
        const uint64_t carry = w ? x[w - 1] >> 63 : ((pat == 1 || pat == 2) ? 1u : 0u);
        const uint64_t prev = (cur << 1) | carry;
        uint64_t v;
        if (pat == 0)
            v = prev & ~cur;
        else if (pat == 1)
            v = (cur ^ prev) & cur;
        else if (pat == 2)
            v = ~(cur | prev);
        else
            v = cur & prev;
        if (w == nw - 1 && (n_bits & 63))
            v &= lo_set((unsigned)(n_bits & 63)); // the padding of the last word holds no positions
        d[w] = v;
    }
}

// Build the device layout from SDSL words that already live on the device.
sdsl_hip_status bv_build_from_device_words(BvHost & bv, const uint64_t * d_words, uint64_t n_bits, uint32_t flags,
                                           uint32_t sel_shift)
{
    bv.view = BvView{};
    bv.view.n_bits = n_bits;
    bv.view.n_lines = n_bits / kDB + 1;
    bv.view.n_lines += bv.view.n_lines & 1; // even: select probes aligned pairs of lines
    if (n_bits >= kLimBvBits)
    {
        set_error("bit vector of %llu bits exceeds the 2^40-bit limit of the select directory",
                  (unsigned long long)n_bits);
        return SDSL_HIP_ERR_UNSUPPORTED;
    }
    bv.view.sel_pshift = 0;
    while ((n_bits >> bv.view.sel_pshift) >= UINT64_C(0xFFFFFFFF))
        ++bv.view.sel_pshift;
    bv.view.sel_shift = sel_shift; // 0 = choose per directory below
    const uint64_t nl = bv.view.n_lines;
    SH_TRY(bv.lines.alloc(nl * kLW * sizeof(uint64_t)));
    SH_TRY(bv.cnts.alloc(nl * sizeof(uint32_t)));
    uint64_t * lines = bv.lines.as<uint64_t>();
    uint32_t * cnts = bv.cnts.as<uint32_t>();
    bv.view.lines = lines;

    hipLaunchKernelGGL(k_build_lines, dim3(grid_for(nl, 256, 65536)), dim3(256), 0, 0, d_words, n_bits, nl, lines,
                       cnts);
    SH_HIP(hipGetLastError());

    SH_TRY(device_exclusive_scan_u32(cnts, nl, lines, (uint64_t)kLW, &bv.view.ones)); // headers = word 0 of every line

    if (bv.view.sel_shift == 0)
    { // one rate for both directories: at most 2^19 four-byte samples (2 MiB) per directory, so that the
      // directory is mostly L2/Infinity-Cache resident.  With the retry-queue kernel the rate is flat between
      // 2^12 and 2^16 on 2^34 bits (profiles/select_sweep_r01.txt); 2^14 is the measured optimum.
        uint64_t most = std::max(bv.view.ones, n_bits - bv.view.ones);
        uint32_t sh = 9;
        while (sh < 20 && (most >> sh) > (UINT64_C(1) << 19))
            ++sh;
        bv.view.sel_shift = sh;
    }
    if (flags & SDSL_HIP_BV_SELECT1)
        SH_TRY(build_select_dir(bv, 1));
    if (flags & SDSL_HIP_BV_SELECT0)
        SH_TRY(build_select_dir(bv, 0));
    SH_HIP(hipDeviceSynchronize());
    bv.cnts.release(); // only needed while building
    return SDSL_HIP_OK;
}

// =========================================================================================
// query kernels
// =========================================================================================

// U queries per quad per round: all U line loads are issued before the first is consumed.
template <int U, bool NT, bool IO_NT = false>
__global__ __launch_bounds__(kBlock) void k_rank(BvView bv, int bit, const uint64_t * __restrict__ idx,
                                                 uint64_t * __restrict__ out, uint64_t n)
{
    if (bv.skip_if && *bv.skip_if)
        return;
    const int s = threadIdx.x & (kG - 1);
    const unsigned gq = threadIdx.x / kG;
    const uint64_t stride = (uint64_t)gridDim.x * kQPB * U;
    for (uint64_t base = (uint64_t)blockIdx.x * kQPB * U; base < n; base += stride)
    {
        uint64_t id[U], L[U];
        Pair w[U];
        bool ok[U];
#pragma unroll
        for (int u = 0; u < U; ++u)
        {
            uint64_t q = base + (uint64_t)u * kQPB + gq;
            id[u] = q < n ? (IO_NT ? __builtin_nontemporal_load(idx + q) : idx[q]) : 0;
            ok[u] = id[u] <= bv.n_bits;
            L[u] = ok[u] ? id[u] / kDB : 0;
        }
#pragma unroll
        for (int u = 0; u < U; ++u)
            w[u] = load_pair<NT>(bv.lines, L[u], s);
#pragma unroll
        for (int u = 0; u < U; ++u)
        {
            uint64_t q = base + (uint64_t)u * kQPB + gq;
            uint64_t r = quad_rank1(w[u], s, ok[u] ? id[u] : 0, L[u]);
            if (!bit)
                r = id[u] - r;
            if (s == 0 && q < n)
            {
                if (IO_NT)
                    __builtin_nontemporal_store(ok[u] ? r : SDSL_HIP_NPOS, out + q);
                else
                    out[q] = ok[u] ? r : SDSL_HIP_NPOS;
            }
        }
    }
}

// Select with a per-block RETRY QUEUE in LDS.  Every round each quad starts one new query (argument and
// sample loads stay coalesced because the block advances in lock step); a quad whose interpolated window missed
// parks (query, bracket) in the queue instead of re-probing while the other 15 quads of its wave wait.  As soon as
// a full round's worth of entries has piled up the whole block spends one round on them.  Misses thus cost one
// probe of one quad, which is what makes a coarse, L2-resident sample directory pay off (profiles/select_sweep_r01.txt).
struct RetryEntry
{
    uint64_t q, k;
    SelBracket br;
    uint32_t tries, pad;
};

// The access skeleton of k_rank with the arithmetic taken out: streamed position -> the quad fetches the 64-byte rank
// line of that position -> one streamed word back.  What this kernel reaches on a given table is what the memory system
// allows a batched rank to reach (sdsl_hip_bv_gather_probe; bench.py times it next to the real kernel).
template <int U>
__global__ __launch_bounds__(kBlock) void k_rank_access_skeleton(BvView bv, const uint64_t * __restrict__ idx,
                                                                 uint64_t * __restrict__ out, uint64_t n)
{
    const int s = threadIdx.x & (kG - 1);
    const unsigned gq = threadIdx.x / kG;
    const uint64_t stride = (uint64_t)gridDim.x * kQPB * U;
    for (uint64_t base = (uint64_t)blockIdx.x * kQPB * U; base < n; base += stride)
    {
        uint64_t L[U];
        Pair w[U];
#pragma unroll
        for (int u = 0; u < U; ++u)
        {
            const uint64_t q = base + (uint64_t)u * kQPB + gq;
            const uint64_t id = q < n ? __builtin_nontemporal_load(idx + q) : 0;
            L[u] = id <= bv.n_bits ? id / kDB : 0;
        }
#pragma unroll
        for (int u = 0; u < U; ++u)
            w[u] = load_pair<false>(bv.lines, L[u], s);
#pragma unroll
        for (int u = 0; u < U; ++u)
        {
            const uint64_t q = base + (uint64_t)u * kQPB + gq;
            const unsigned x = quad_sum((unsigned)(w[u].a ^ w[u].b)); // every lane's 16 bytes are consumed
            if (s == 0 && q < n)
                __builtin_nontemporal_store((uint64_t)x, out + q);
        }
    }
}

template <int BIT, bool NT, bool IO_NT = false>
__global__ __launch_bounds__(kBlock) void k_select_rq(BvView bv, const uint64_t * __restrict__ iq,
                                                      uint64_t * __restrict__ out, uint64_t n)
{
    __shared__ RetryEntry rq[2 * kQPB];
    __shared__ unsigned rq_n;
    if (bv.skip_if && *bv.skip_if)
        return;
    const int s = threadIdx.x & (kG - 1);
    const unsigned gq = threadIdx.x / kG;
    const uint64_t total = BIT ? bv.ones : bv.n_bits - bv.ones;
    if (threadIdx.x == 0)
        rq_n = 0;
    __syncthreads();

    // one probe of query (q, k); on a miss the tightened bracket goes to the queue.  Quad-uniform.
    auto probe = [&](uint64_t q, uint64_t k, SelBracket br, uint32_t tries)
    {
        const uint64_t W = sel_guess(bv, br, k, (int)tries);
        Pair wa = load_pair<NT>(bv.lines, 2 * W, s);
        Pair wb = load_pair<NT>(bv.lines, 2 * W + 1, s);
        bool mine = false;
        uint64_t pos = 0;
        if (sel_eval<BIT>(bv, s, k, W, wa, wb, br, mine, pos))
        {
            if (mine)
            {
                if (IO_NT)
                    __builtin_nontemporal_store(pos, out + q);
                else
                    out[q] = pos;
            }
        }
        else if (s == 0)
        {
            unsigned slot = atomicAdd(&rq_n, 1u);
            RetryEntry e;
            e.q = q;
            e.k = k;
            e.br = br;
            e.tries = tries + 1;
            e.pad = 0;
            rq[slot] = e;
        }
    };

    // Software pipeline over the rounds: a query costs three dependent memory accesses (argument -> directory samples
    // -> window).  The argument is loaded two rounds ahead and the samples one round ahead, so that only the window
    // fetch is on the critical path of a round.
    const uint64_t stride = (uint64_t)gridDim.x * kQPB;
    auto load_arg = [&](uint64_t q) -> uint64_t { return q < n ? (IO_NT ? __builtin_nontemporal_load(iq + q) : iq[q]) : 0; };
    auto arg_ok = [&](uint64_t i) -> bool { return i >= 1 && i <= total; }; // outside: SDSL's precondition (select_support_mcl.hpp:386)
    const uint64_t q_first = (uint64_t)blockIdx.x * kQPB + gq;
    uint64_t i_cur = load_arg(q_first), i_nxt = load_arg(q_first + stride);
    SelSamples sm_cur = sel_samples<BIT>(bv, arg_ok(i_cur) ? i_cur - 1 : 0);
    for (uint64_t base = (uint64_t)blockIdx.x * kQPB; base < n; base += stride) // block-uniform trip count
    {
        const uint64_t q = base + gq;
        const uint64_t i_nn = load_arg(q + 2 * stride);                                 // argument of round r+2
        const SelSamples sm_nxt = sel_samples<BIT>(bv, arg_ok(i_nxt) ? i_nxt - 1 : 0); // samples of round r+1
        const uint64_t i = i_cur;
        const bool ok = arg_ok(i);
        if (q < n && !ok && s == 0)
            out[q] = SDSL_HIP_NPOS;
        if (ok)
        {
            const uint64_t k = i - 1;
            probe(q, k, sel_bracket<BIT>(bv, k, sm_cur), 0u);
        }
        i_cur = i_nxt;
        i_nxt = i_nn;
        sm_cur = sm_nxt;
        __syncthreads();
        unsigned cnt = rq_n;
        __syncthreads(); // everybody has seen the same count before anyone pushes again
        while (cnt >= kQPB)
        {
            RetryEntry e = rq[cnt - kQPB + gq];
            __syncthreads();
            if (threadIdx.x == 0)
                rq_n = cnt - kQPB;
            __syncthreads();
            probe(e.q, e.k, e.br, e.tries);
            __syncthreads();
            cnt = rq_n;
            __syncthreads();
        }
    }
    // drain what is left
    __syncthreads();
    unsigned cnt = rq_n;
    __syncthreads();
    while (cnt > 0)
    {
        const unsigned take = cnt < kQPB ? cnt : kQPB;
        const bool act = gq < take;
        RetryEntry e{};
        if (act)
            e = rq[cnt - take + gq];
        __syncthreads();
        if (threadIdx.x == 0)
            rq_n = cnt - take;
        __syncthreads();
        if (act)
            probe(e.q, e.k, e.br, e.tries);
        __syncthreads();
        cnt = rq_n;
        __syncthreads();
    }
}

// The same scheme with one retry queue PER WAVE and no block-level barrier: the 16 quads of a wave are in lock step
// anyway, so pushing, counting and popping need nothing but the in-order LDS pipeline of that wave.  A round of one
// wave no longer waits for the slowest window fetch of the other three waves of its block.
constexpr unsigned kQPW = 64 / kG; // quads (queries in flight) per wave

__device__ __forceinline__ void wave_lds_sync()
{
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    __builtin_amdgcn_wave_barrier();
}

template <int BIT, bool NT>
__global__ __launch_bounds__(kBlock) void k_select_wq(BvView bv, const uint64_t * __restrict__ iq,
                                                      uint64_t * __restrict__ out, uint64_t n)
{
    constexpr unsigned kWaves = kBlock / 64;
    __shared__ RetryEntry rq_all[kWaves][2 * kQPW];
    __shared__ unsigned rq_cnt[kWaves];
    if (bv.skip_if && *bv.skip_if)
        return;
    const int s = threadIdx.x & (kG - 1);
    const unsigned wave = threadIdx.x / 64, wq = (threadIdx.x & 63) / kG; // quad index inside the wave
    const unsigned gq = threadIdx.x / kG;
    RetryEntry * rq = rq_all[wave];
    unsigned * rq_n = &rq_cnt[wave];
    const uint64_t total = BIT ? bv.ones : bv.n_bits - bv.ones;
    if ((threadIdx.x & 63) == 0)
        *rq_n = 0;
    wave_lds_sync();

    auto probe = [&](uint64_t q, uint64_t k, SelBracket br, uint32_t tries)
    {
        const uint64_t W = sel_guess(bv, br, k, (int)tries);
        Pair wa = load_pair<NT>(bv.lines, 2 * W, s);
        Pair wb = load_pair<NT>(bv.lines, 2 * W + 1, s);
        bool mine = false;
        uint64_t pos = 0;
        if (sel_eval<BIT>(bv, s, k, W, wa, wb, br, mine, pos))
        {
            if (mine)
                __builtin_nontemporal_store(pos, out + q);
        }
        else if (s == 0)
        {
            unsigned slot = atomicAdd(rq_n, 1u);
            RetryEntry e;
            e.q = q;
            e.k = k;
            e.br = br;
            e.tries = tries + 1;
            e.pad = 0;
            rq[slot] = e;
        }
    };

    const uint64_t stride = (uint64_t)gridDim.x * kQPB;
    auto load_arg = [&](uint64_t q) -> uint64_t { return q < n ? __builtin_nontemporal_load(iq + q) : 0; };
    auto arg_ok = [&](uint64_t i) -> bool { return i >= 1 && i <= total; }; // SDSL's precondition (select_support_mcl.hpp:386)
    const uint64_t q_first = (uint64_t)blockIdx.x * kQPB + gq;
    uint64_t i_cur = load_arg(q_first), i_nxt = load_arg(q_first + stride);
    SelSamples sm_cur = sel_samples<BIT>(bv, arg_ok(i_cur) ? i_cur - 1 : 0);
    for (uint64_t base = (uint64_t)blockIdx.x * kQPB; base < n; base += stride) // wave-uniform trip count
    {
        const uint64_t q = base + gq;
        const uint64_t i_nn = load_arg(q + 2 * stride);              // argument of round r+2
        const SelSamples sm_nxt = sel_samples<BIT>(bv, arg_ok(i_nxt) ? i_nxt - 1 : 0); // samples of round r+1
        const uint64_t i = i_cur;
        const bool ok = arg_ok(i);
        if (q < n && !ok && s == 0)
            out[q] = SDSL_HIP_NPOS;
        if (ok)
        {
            const uint64_t k = i - 1;
            probe(q, k, sel_bracket<BIT>(bv, k, sm_cur), 0u);
        }
        i_cur = i_nxt;
        i_nxt = i_nn;
        sm_cur = sm_nxt;
        wave_lds_sync();
        unsigned cnt = *rq_n;
        while (cnt >= kQPW)
        { // a full wave's worth of parked queries: spend one round on them
            RetryEntry e = rq[cnt - kQPW + wq];
            wave_lds_sync();
            if ((threadIdx.x & 63) == 0)
                *rq_n = cnt - kQPW;
            wave_lds_sync();
            probe(e.q, e.k, e.br, e.tries);
            wave_lds_sync();
            cnt = *rq_n;
        }
    }
    // drain what is left
    wave_lds_sync();
    unsigned cnt = *rq_n;
    while (cnt > 0)
    {
        const unsigned take = cnt < kQPW ? cnt : kQPW;
        const bool act = wq < take;
        RetryEntry e{};
        if (act)
            e = rq[cnt - take + wq];
        wave_lds_sync();
        if ((threadIdx.x & 63) == 0)
            *rq_n = cnt - take;
        wave_lds_sync();
        if (act)
            probe(e.q, e.k, e.br, e.tries);
        wave_lds_sync();
        cnt = *rq_n;
    }
}

__global__ __launch_bounds__(256) void k_access(BvView bv, const uint64_t * __restrict__ idx,
                                                uint8_t * __restrict__ out, uint64_t n)
{
    for (uint64_t q = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; q < n; q += (uint64_t)gridDim.x * blockDim.x)
    {
        uint64_t i = idx[q];
        if (i >= bv.n_bits)
        {
            out[q] = 0xFF;
            continue;
        }
        uint64_t L = i / kDB;
        unsigned off = (unsigned)(i - L * kDB);
        uint64_t w = bv.lines[L * kLW + 1 + (off >> 6)];
        out[q] = (uint8_t)((w >> (off & 63)) & 1);
    }
}

__global__ __launch_bounds__(256) void k_export_words(BvView bv, uint64_t * __restrict__ words, uint64_t n_words)
{
    for (uint64_t wi = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; wi < n_words;
         wi += (uint64_t)gridDim.x * blockDim.x)
    {
        uint64_t L = wi / kDW;
        unsigned d = (unsigned)(wi - L * kDW);
        words[wi] = bv.lines[L * kLW + 1 + d];
    }
}

sdsl_hip_status bv_export_words_device(const BvView & v, uint64_t * d_words, uint64_t n_words, hipStream_t s)
{
    if (n_words)
        hipLaunchKernelGGL(k_export_words, dim3(grid_for(n_words, 256, 65536)), dim3(256), 0, s, v, d_words, n_words);
    SH_HIP(hipGetLastError());
    return SDSL_HIP_OK;
}

static unsigned query_grid(uint64_t n, unsigned q_per_block)
{
    // memory-latency bound gathers: fill every CU with 8 blocks of 256 threads, grid-stride the rest
    return grid_for(n, q_per_block, 256u * 8u);
}

sdsl_hip_status bv_launch_rank(const BvView & v, int bit, const uint64_t * d_idx, uint64_t n, uint64_t * d_out,
                               hipStream_t s)
{
    if (n == 0)
        return SDSL_HIP_OK;
    // 4 queries per quad per round, plain (cached) loads: the best of the variants measured in round 1
    // (U = 1/2/4/8, nontemporal or not: 36.2-38.8 G/s, profiles/gather_probe_r01.txt)
    constexpr int U = 4;
    KernelTimer t(s);
    const char * io_env = getenv("SDSL_HIP_RANK_IO_NT"); // experiment knob (profiles/rank_io_nt_r01.txt)
    const int io_nt = io_env ? atoi(io_env) : 1; // default on: +2.2 % rank, +0.5 % select (same allocation A/B)
    if (io_nt)
        hipLaunchKernelGGL((k_rank<U, false, true>), dim3(query_grid(n, kQPB * U)), dim3(kBlock), 0, s, v, bit, d_idx, d_out,
                           n);
    else
        hipLaunchKernelGGL((k_rank<U, false>), dim3(query_grid(n, kQPB * U)), dim3(kBlock), 0, s, v, bit, d_idx, d_out, n);
    SH_HIP(hipGetLastError());
    return SDSL_HIP_OK;
}

sdsl_hip_status bv_launch_select(const BvView & v, int bit, const uint64_t * d_i, uint64_t n, uint64_t * d_out,
                                 hipStream_t s)
{
    if (n == 0)
        return SDSL_HIP_OK;
    if (!v.sel[bit])
    {
        set_error("select_%d directory was not built (pass SDSL_HIP_BV_SELECT%d to sdsl_hip_bv_create)", bit, bit);
        return SDSL_HIP_ERR_INVALID;
    }
    KernelTimer t(s);
    const char * io_env = getenv("SDSL_HIP_SELECT_IO_NT");
    const int io_nt = io_env ? atoi(io_env) : 1; // default on: +2.2 % rank, +0.5 % select (same allocation A/B)
    const char * var_env = getenv("SDSL_HIP_SELECT_VARIANT"); // "wq" (default): per-wave retry queues; "rq": per-block
    const bool wave_queues = !(var_env && var_env[0] == 'r');
    if (wave_queues && bit)
        hipLaunchKernelGGL((k_select_wq<1, false>), dim3(query_grid(n, kQPB)), dim3(kBlock), 0, s, v, d_i, d_out, n);
    else if (wave_queues)
        hipLaunchKernelGGL((k_select_wq<0, false>), dim3(query_grid(n, kQPB)), dim3(kBlock), 0, s, v, d_i, d_out, n);
    else if (bit && io_nt)
        hipLaunchKernelGGL((k_select_rq<1, false, true>), dim3(query_grid(n, kQPB)), dim3(kBlock), 0, s, v, d_i, d_out, n);
    else if (bit)
        hipLaunchKernelGGL((k_select_rq<1, false>), dim3(query_grid(n, kQPB)), dim3(kBlock), 0, s, v, d_i, d_out, n);
    else if (io_nt)
        hipLaunchKernelGGL((k_select_rq<0, false, true>), dim3(query_grid(n, kQPB)), dim3(kBlock), 0, s, v, d_i, d_out, n);
    else
        hipLaunchKernelGGL((k_select_rq<0, false>), dim3(query_grid(n, kQPB)), dim3(kBlock), 0, s, v, d_i, d_out, n);
    SH_HIP(hipGetLastError());
    return SDSL_HIP_OK;
}

// direct kernel or, for a large batch over a large vector, the bucketed path (bv_sorted.hip / bv_swc.hip).  Option "rank_sorted"
// (sdsl_hip_set_option; initial value from SDSL_HIP_RANK_SORTED): 0 = never, 1 = whenever possible, -1 = automatic.
//
// Automatic: a batch that is SPREAD over the vector goes through the passes, one confined to a window (or sorted) is better
// served by the direct kernel out of L2 / Infinity Cache.  Which it is, a sample of the batch says (k_sr_sample_spread) — on
// the device: the verdict is a word in device memory, BOTH routes are enqueued behind the sample, and the one whose turn it is
// not returns at once (SrGeom::go / BvView::skip_if).  Nothing is read back, nothing synchronises: the call stays asynchronous
// on the caller's stream and can be captured into a graph (once the handle's scratch has its size: the first call allocates).

static size_t bv_pass_scratch_bytes(const BvHost & h, uint64_t n)
{
    const uint64_t pass = n < (UINT64_C(1) << 30) ? n : (UINT64_C(1) << 30);
    return bv_sorted_rank_scratch_bytes(h.view, pass);
}

sdsl_hip_status bv_rank_dispatch(BvHost & h, int bit, const uint64_t * d_idx, uint64_t n, uint64_t * d_out, hipStream_t s)
{
    const int mode = g_rank_sorted_mode.load();
    const bool want = mode == 0 ? false : (mode > 0 ? bv_sorted_rank_possible(h.view) : bv_sorted_rank_applicable(h.view, n));
    if (want && n > 0)
    {
        // the scratch belongs to the handle; queries stay safe to issue from several threads / on several streams: the
        // host side is serialised here, the device side by an event the next user of the scratch waits for
        std::lock_guard<std::mutex> lock(h.scratch_mutex);
        const bool on_device = mode < 0 && bv_sorted_device_verdict();
        if (mode < 0 && !on_device)
        { // (the one-sweep passes, SDSL_HIP_SORTED_SWC=0: the verdict is read back, which synchronises s)
            bool spread = true;
            if (!h.spread_probe.p)
                SH_TRY(h.spread_probe.alloc(64));
            SH_TRY(bv_sorted_rank_is_spread(h.view, d_idx, n, s, h.spread_probe.p, spread));
            if (!spread)
                return bv_launch_rank(h.view, bit, d_idx, n, d_out, s);
        }
        sdsl_hip_status st;
        ScratchLease L; // (ends — and records the pool's event — when this block is left, on every path)
        SH_TRY(L.acquire(h.device, h.capture_scratch, bv_pass_scratch_bytes(h, n), s));
        if (!L.p || (L.capturing && !h.spread_probe.p))
            return bv_launch_rank(h.view, bit, d_idx, n, d_out, s); // no room for the scratch (or nothing reserved for a capture): direct kernel
        const uint32_t * go = nullptr;
        if (on_device && !h.spread_probe.p)
            SH_TRY(h.spread_probe.alloc(64));
        {
            KernelTimer t(s);
            if (on_device)
            {
                SH_TRY(bv_sorted_rank_sample(h.view, d_idx, n, s, h.spread_probe.as<uint32_t>()));
                go = h.spread_probe.as<uint32_t>() + 2;
            }
            st = bv_launch_rank_sorted(h.view, bit, d_idx, n, d_out, s, L.p, L.bytes, go);
            if (st == SDSL_HIP_OK && go)
            {
                TimingPause pause; // (one timer around both routes)
                BvView dv = h.view;
                dv.skip_if = go;
                st = bv_launch_rank(dv, bit, d_idx, n, d_out, s);
            }
        }
        return st;
    }
    return bv_launch_rank(h.view, bit, d_idx, n, d_out, s);
}

// select: the same choice (option "select_sorted", initial value SDSL_HIP_SELECT_SORTED)
sdsl_hip_status bv_select_dispatch(BvHost & h, int bit, const uint64_t * d_i, uint64_t n, uint64_t * d_out, hipStream_t s)
{
    const int mode = g_select_sorted_mode.load();
    if (mode != 0 && n > 0 && h.view.sel[bit] && (mode > 0 || (h.view.n_lines >= (UINT64_C(1) << 22) && n >= 2 * h.view.n_lines)))
    {
        std::lock_guard<std::mutex> lock(h.scratch_mutex);
        if (stream_is_capturing(s) && (!h.sel_plan[bit].ready || !h.spread_probe.p || !h.capture_scratch.p))
            return bv_launch_select(h.view, bit, d_i, n, d_out, s); // nothing may be built or allocated during a capture
        SH_TRY(bv_select_sorted_prepare(h, bit)); // first use: the bucket boundaries (one small batch, synchronous)
        bool want = mode > 0 ? h.sel_plan[bit].ok : bv_sorted_select_applicable(h, bit, n);
        const bool on_device = mode < 0 && bv_sorted_device_verdict();
        if (want && mode < 0 && !on_device)
        { // (the one-sweep passes: read-back, synchronises s)
            if (!h.spread_probe.p)
                SH_TRY(h.spread_probe.alloc(64));
            SH_TRY(bv_sorted_select_is_spread(h, bit, d_i, n, s, h.spread_probe.p, want));
        }
        ScratchLease L;
        if (want)
            SH_TRY(L.acquire(h.device, h.capture_scratch, bv_pass_scratch_bytes(h, n), s));
        if (L.p)
        {
            sdsl_hip_status st = SDSL_HIP_OK;
            const uint32_t * go = nullptr;
            if (on_device && !h.spread_probe.p)
                SH_TRY(h.spread_probe.alloc(64));
            {
                KernelTimer t(s);
                if (on_device)
                {
                    SH_TRY(bv_sorted_select_sample(h, bit, d_i, n, s, h.spread_probe.as<uint32_t>()));
                    go = h.spread_probe.as<uint32_t>() + 2;
                }
                st = bv_launch_select_sorted(h, bit, d_i, n, d_out, s, L.p, L.bytes, go);
                if (st == SDSL_HIP_OK && go)
                {
                    TimingPause pause;
                    BvView dv = h.view;
                    dv.skip_if = go;
                    st = bv_launch_select(dv, bit, d_i, n, d_out, s);
                }
            }
            return st;
        }
    }
    return bv_launch_select(h.view, bit, d_i, n, d_out, s);
}

uint32_t default_sel_shift()
{
    uint32_t sh = 0; // 0 = automatic: smallest rate >= 512 that keeps a directory within 2^21 samples
    if (const char * e = getenv("SDSL_HIP_SELECT_SAMPLE_LOG2"))
    {
        int v = atoi(e);
        if (v >= 6 && v <= 20)
            sh = (uint32_t)v;
    }
    return sh;
}

} // namespace sdslhip

using namespace sdslhip;

struct sdsl_hip_bv_s
{
    BvHost h;
    uint64_t uid = next_handle_uid(); // key of the serialiser's size-query cache
};

namespace sdslhip {

BvHost & bv_host_of(sdsl_hip_bv_t bv)
{
    return bv->h;
}

sdsl_hip_status bv_new_replica(const BvHost & src, int device, sdsl_hip_bv_t * out)
{
    *out = nullptr;
    SH_TRY(check_device(device));
    sdsl_hip_bv_s * bv = new (std::nothrow) sdsl_hip_bv_s();
    if (!bv)
        return SDSL_HIP_ERR_NOMEM;
    BvHost & d = bv->h;
    d.device = device;
    d.view = src.view;
    sdsl_hip_status st = d.lines.alloc(src.lines.bytes);
    for (int b = 0; b < 2 && st == SDSL_HIP_OK; ++b)
    {
        if (src.sel[b].p)
            st = d.sel[b].alloc(src.sel[b].bytes);
        if (st == SDSL_HIP_OK && src.lmask[b].p)
            st = d.lmask[b].alloc(src.lmask[b].bytes);
        if (st == SDSL_HIP_OK && src.lidx[b].p)
            st = d.lidx[b].alloc(src.lidx[b].bytes);
        if (st == SDSL_HIP_OK && src.lpos[b].p)
            st = d.lpos[b].alloc(src.lpos[b].bytes);
    }
    if (st != SDSL_HIP_OK)
    {
        delete bv;
        return st;
    }
    d.view.lines = d.lines.as<uint64_t>();
    for (int b = 0; b < 2; ++b)
    {
        d.view.sel[b] = src.sel[b].p ? d.sel[b].as<uint32_t>() : nullptr;
        d.view.lmask[b] = src.lmask[b].p ? d.lmask[b].as<uint32_t>() : nullptr;
        d.view.lidx[b] = src.lidx[b].p ? d.lidx[b].as<uint32_t>() : nullptr;
        d.view.lpos[b] = src.lpos[b].p ? d.lpos[b].as<uint32_t>() : nullptr;
    }
    *out = bv;
    return SDSL_HIP_OK;
}

} // namespace sdslhip

extern "C" {

static sdsl_hip_status sdsl_hip_bv_create_impl(const uint64_t * words, uint64_t n_bits, int32_t device, uint32_t flags,
                                   sdsl_hip_bv_t * out)
{
    if (!out || (!words && n_bits))
    {
        set_error("bv_create: null argument");
        return SDSL_HIP_ERR_INVALID;
    }
    *out = nullptr;
    SH_TRY(check_device(device));
    sdsl_hip_bv_s * bv = new (std::nothrow) sdsl_hip_bv_s();
    if (!bv)
        return SDSL_HIP_ERR_NOMEM;
    bv->h.device = device;
    Staged w;
    sdsl_hip_status st = w.in(words, ((n_bits + 63) >> 6) * sizeof(uint64_t), nullptr);
    if (st == SDSL_HIP_OK)
        st = bv_build_from_device_words(bv->h, (const uint64_t *)w.dev, n_bits, flags, default_sel_shift());
    if (st != SDSL_HIP_OK)
    {
        delete bv;
        return st;
    }
    *out = bv;
    return SDSL_HIP_OK;
}
// no exception crosses the C ABI: a malformed stream or an exhausted host becomes a status code
sdsl_hip_status sdsl_hip_bv_create(const uint64_t * words, uint64_t n_bits, int32_t device, uint32_t flags,
                                   sdsl_hip_bv_t * out)
{
    return guarded("bv_create", [&] { return sdsl_hip_bv_create_impl(words, n_bits, device, flags, out); });
}

static sdsl_hip_status sdsl_hip_bv_create_from_sdsl_impl(const void * bytes, size_t len, int32_t kind, int32_t device, uint32_t flags,
                                             sdsl_hip_bv_t * out)
{
    if (!out || !bytes)
    {
        set_error("bv_create_from_sdsl: null argument");
        return SDSL_HIP_ERR_INVALID;
    }
    *out = nullptr;
    SH_TRY(check_device(device));
    DevBuf d_words;
    uint64_t n_bits = 0;
    SH_TRY(compressed_stream_to_device_words(bytes, len, kind, d_words, n_bits));
    sdsl_hip_bv_s * bv = new (std::nothrow) sdsl_hip_bv_s();
    if (!bv)
        return SDSL_HIP_ERR_NOMEM;
    bv->h.device = device;
    sdsl_hip_status st = bv_build_from_device_words(bv->h, d_words.as<uint64_t>(), n_bits, flags, default_sel_shift());
    if (st != SDSL_HIP_OK)
    {
        delete bv;
        return st;
    }
    *out = bv;
    return SDSL_HIP_OK;
}
// no exception crosses the C ABI: a malformed stream or an exhausted host becomes a status code
sdsl_hip_status sdsl_hip_bv_create_from_sdsl(const void * bytes, size_t len, int32_t kind, int32_t device, uint32_t flags,
                                             sdsl_hip_bv_t * out)
{
    return guarded("bv_create_from_sdsl", [&] { return sdsl_hip_bv_create_from_sdsl_impl(bytes, len, kind, device, flags, out); });
}

static sdsl_hip_status sdsl_hip_bv_create_pattern_impl(const uint64_t * words, uint64_t n_bits, int32_t device, uint32_t t_b,
                                           uint32_t t_pat_len, uint32_t flags, sdsl_hip_bv_t * out)
{
    if (t_pat_len == 1 && t_b <= 1)
        return sdsl_hip_bv_create(words, n_bits, device, flags, out);
    int pat = -1; // (previous bit, this bit)
    if (t_pat_len == 2)
        pat = t_b == 10 ? 0 : (t_b == 1 ? 1 : (t_b == 0 ? 2 : (t_b == 11 ? 3 : -1)));
    if (!out || (!words && n_bits) || pat < 0)
    {
        set_error("bv_create_pattern: the pattern must be one of <0,1> <1,1> <10,2> <01,2> <00,2> <11,2>");
        return SDSL_HIP_ERR_INVALID;
    }
    *out = nullptr;
    SH_TRY(check_device(device));
    sdsl_hip_bv_s * bv = new (std::nothrow) sdsl_hip_bv_s();
    if (!bv)
        return SDSL_HIP_ERR_NOMEM;
    bv->h.device = device;
    const uint64_t nw = (n_bits + 63) >> 6;
    Staged w;
    DevBuf d;
    sdsl_hip_status st = w.in(words, nw * sizeof(uint64_t), nullptr);
    if (st == SDSL_HIP_OK)
        st = d.alloc((nw + 1) * 8, true);
    if (st == SDSL_HIP_OK && nw)
    {
        hipLaunchKernelGGL(k_pattern_words, dim3(grid_for(nw, 256, 65536)), dim3(256), 0, 0, (const uint64_t *)w.dev, n_bits,
                           nw, pat, d.as<uint64_t>());
        if (hipGetLastError() != hipSuccess)
        {
            set_error("bv_create_pattern: kernel launch failed");
            st = SDSL_HIP_ERR_HIP;
        }
    }
    if (st == SDSL_HIP_OK)
        st = bv_build_from_device_words(bv->h, d.as<uint64_t>(), n_bits, flags, default_sel_shift());
    if (st != SDSL_HIP_OK)
    {
        delete bv;
        return st;
    }
    *out = bv;
    return SDSL_HIP_OK;
}
// no exception crosses the C ABI: a malformed stream or an exhausted host becomes a status code
sdsl_hip_status sdsl_hip_bv_create_pattern(const uint64_t * words, uint64_t n_bits, int32_t device, uint32_t t_b,
                                           uint32_t t_pat_len, uint32_t flags, sdsl_hip_bv_t * out)
{
    return guarded("bv_create_pattern", [&] { return sdsl_hip_bv_create_pattern_impl(words, n_bits, device, t_b, t_pat_len, flags, out); });
}

static sdsl_hip_status sdsl_hip_bv_add_select_impl(sdsl_hip_bv_t bv, uint32_t flags)
{
    if (!bv)
    {
        set_error("bv_add_select: null handle");
        return SDSL_HIP_ERR_INVALID;
    }
    BvHost & h = bv->h;
    std::lock_guard<std::mutex> lock(h.scratch_mutex);
    const bool need1 = (flags & SDSL_HIP_BV_SELECT1) && !h.view.sel[1], need0 = (flags & SDSL_HIP_BV_SELECT0) && !h.view.sel[0];
    if (!need1 && !need0)
        return SDSL_HIP_OK;
    SH_HIP(hipSetDevice(h.device));
    SH_TRY(h.cnts.alloc(h.view.n_lines * sizeof(uint32_t)));
    hipLaunchKernelGGL(k_line_counts, dim3(grid_for(h.view.n_lines, 256, 65536)), dim3(256), 0, 0, h.view.lines, h.view.n_lines,
                       h.cnts.as<uint32_t>());
    sdsl_hip_status st = hipGetLastError() == hipSuccess ? SDSL_HIP_OK : SDSL_HIP_ERR_HIP;
    if (st == SDSL_HIP_OK && need1)
        st = build_select_dir(h, 1);
    if (st == SDSL_HIP_OK && need0)
        st = build_select_dir(h, 0);
    if (hipDeviceSynchronize() != hipSuccess && st == SDSL_HIP_OK)
        st = SDSL_HIP_ERR_HIP;
    h.cnts.release();
    return st;
}
// a select directory for a handle that was created without it (several supports of one vector share one device replica:
// include/sdsl_hip/adaptors.hpp); queries already in flight are not disturbed — they cannot be select queries of that bit value
sdsl_hip_status sdsl_hip_bv_add_select(sdsl_hip_bv_t bv, uint32_t flags)
{
    return guarded("bv_add_select", [&] { return sdsl_hip_bv_add_select_impl(bv, flags); });
}

static sdsl_hip_status sdsl_hip_bv_serialize_impl(sdsl_hip_bv_t bv, int32_t what, void * buf, size_t cap, size_t * written)
{
    if (!bv || what < 0 || what > SDSL_HIP_SER_RANK_V_0)
    {
        set_error("bv_serialize: invalid argument");
        return SDSL_HIP_ERR_INVALID;
    }
    sdsl_hip_status cached;
    if (deliver_cached(bv->uid, (uint64_t)what, buf, cap, written, cached))
        return cached;
    SH_HIP(hipSetDevice(bv->h.device));
    const uint64_t n = bv->h.view.n_bits, W = (n + 63) >> 6;
    std::vector<uint64_t> words(W + 1, 0);
    if (W)
    {
        DevBuf d;
        SH_TRY(d.alloc(W * 8));
        SH_TRY(bv_export_words_device(bv->h.view, d.as<uint64_t>(), W, nullptr));
        SH_HIP(hipMemcpy(words.data(), d.p, W * 8, hipMemcpyDeviceToHost));
    }
    StreamWriter w;
    switch (what)
    {
    case SDSL_HIP_SER_BIT_VECTOR: w.int_vector(words.data(), n, 1); break;
    case SDSL_HIP_SER_RANK_V5_1: rank_v5_serialize_host(words.data(), n, 1, w); break;
    case SDSL_HIP_SER_RANK_V5_0: rank_v5_serialize_host(words.data(), n, 0, w); break;
    case SDSL_HIP_SER_SELECT_MCL_1: select_mcl_serialize_host(words.data(), n, 1, w); break;
    case SDSL_HIP_SER_SELECT_MCL_0: select_mcl_serialize_host(words.data(), n, 0, w); break;
    case SDSL_HIP_SER_RANK_V_1: rank_v_serialize_host(words.data(), n, 1, w); break;
    default: rank_v_serialize_host(words.data(), n, 0, w); break;
    }
    return deliver_and_cache(bv->uid, (uint64_t)what, w, buf, cap, written);
}
// no exception crosses the C ABI: a malformed stream or an exhausted host becomes a status code
sdsl_hip_status sdsl_hip_bv_serialize(sdsl_hip_bv_t bv, int32_t what, void * buf, size_t cap, size_t * written)
{
    return guarded("bv_serialize", [&] { return sdsl_hip_bv_serialize_impl(bv, what, buf, cap, written); });
}

sdsl_hip_status sdsl_hip_bv_destroy(sdsl_hip_bv_t bv)
{
    if (!bv)
        return SDSL_HIP_OK;
    (void)hipSetDevice(bv->h.device);
    device_scratch_quiesce(bv->h.device); // (a bucketed batch over this vector may still be running on some stream)
    delete bv;
    return SDSL_HIP_OK;
}

sdsl_hip_status sdsl_hip_last_phases(char * buf, size_t cap)
{
    if (!buf || cap == 0)
        return SDSL_HIP_ERR_INVALID;
    const std::string p = bv_sorted_last_phases();
    snprintf(buf, cap, "%s", p.c_str());
    return SDSL_HIP_OK;
}

sdsl_hip_status sdsl_hip_bv_layout_info(sdsl_hip_bv_t bv, uint64_t out[4])
{
    if (!bv || !out)
        return SDSL_HIP_ERR_INVALID;
    out[0] = (uint64_t)(uintptr_t)bv->h.lines.p;
    out[1] = bv->h.lines.bytes;
    out[2] = (uint64_t)(uintptr_t)bv->h.sel[1].p;
    out[3] = (uint64_t)(uintptr_t)device_scratch(bv->h.device).buf.p;
    return SDSL_HIP_OK;
}

sdsl_hip_status sdsl_hip_bv_release_scratch(sdsl_hip_bv_t bv)
{
    if (!bv)
        return SDSL_HIP_OK;
    std::lock_guard<std::mutex> lock(bv->h.scratch_mutex);
    SH_HIP(hipSetDevice(bv->h.device));
    bv->h.spread_probe.release();
    return device_scratch_release(bv->h.device);
}

sdsl_hip_status sdsl_hip_bv_reserve_capture_scratch(sdsl_hip_bv_t bv, uint64_t max_queries)
{
    if (!bv)
    {
        set_error("bv_reserve_capture_scratch: null handle");
        return SDSL_HIP_ERR_INVALID;
    }
    BvHost & h = bv->h;
    std::lock_guard<std::mutex> lock(h.scratch_mutex);
    SH_HIP(hipSetDevice(h.device));
    if (max_queries == 0)
    {
        h.capture_scratch.release();
        return SDSL_HIP_OK;
    }
    const size_t need = bv_pass_scratch_bytes(h, max_queries);
    if (h.capture_scratch.bytes < need)
        SH_TRY(h.capture_scratch.alloc(need));
    if (!h.spread_probe.p)
        SH_TRY(h.spread_probe.alloc(64));
    for (int bit = 0; bit < 2; ++bit) // the bucket plans of the select directories the handle has (synchronous, once)
        if (h.view.sel[bit])
            SH_TRY(bv_select_sorted_prepare(h, bit));
    return SDSL_HIP_OK;
}

uint64_t sdsl_hip_device_scratch_bytes(int32_t device)
{
    if (device < 0 || device >= 64)
        return 0;
    DeviceScratch & P = device_scratch(device);
    std::lock_guard<std::mutex> lock(P.m);
    return P.buf.bytes;
}

uint64_t sdsl_hip_bv_size(sdsl_hip_bv_t bv)
{
    return bv ? bv->h.view.n_bits : 0;
}
uint64_t sdsl_hip_bv_ones(sdsl_hip_bv_t bv)
{
    return bv ? bv->h.view.ones : 0;
}
uint64_t sdsl_hip_bv_device_bytes(sdsl_hip_bv_t bv)
{
    return bv ? bv->h.device_bytes() : 0;
}

sdsl_hip_status sdsl_hip_bv_rank_batch(sdsl_hip_bv_t bv, int32_t bit, const uint64_t * idx, uint64_t n,
                                       uint64_t * out, void * stream)
{
    if (!bv || (bit != 0 && bit != 1) || (n && (!idx || !out)))
    {
        set_error("bv_rank_batch: invalid argument");
        return SDSL_HIP_ERR_INVALID;
    }
    hipStream_t s = (hipStream_t)stream;
    SH_HIP(hipSetDevice(bv->h.device));
    if (n >= kPipelineMinQueries && !is_device_ptr(idx) && !is_device_ptr(out))
    { // host arrays on both sides: chunked, uploads / kernels / downloads overlapped on several streams
        const BvView v = bv->h.view;
        return host_pipeline_u64(bv->h.device, idx, out, n,
                                 [v, bit](const uint64_t * d_in, uint64_t * d_out, uint64_t cnt, hipStream_t st)
                                 { return bv_launch_rank(v, bit, d_in, cnt, d_out, st); });
    }
    Staged in, o;
    SH_TRY(in.in(idx, n * 8, s));
    SH_TRY(o.out(out, n * 8));
    SH_TRY(bv_rank_dispatch(bv->h, bit, (const uint64_t *)in.dev, n, (uint64_t *)o.dev, s));
    SH_TRY(o.finish(s));
    if (in.host && !o.host)
        SH_HIP(hipStreamSynchronize(s)); // staging buffer must outlive the kernel
    return SDSL_HIP_OK;
}

// one query, host value in, host value out (the adaptors' scalar operator())
sdsl_hip_status sdsl_hip_bv_query_one(sdsl_hip_bv_t bv, int32_t what, int32_t bit, uint64_t arg, uint64_t * out)
{
    if (!bv || !out || (bit != 0 && bit != 1) || what < 0 || what > 1)
    {
        set_error("bv_query_one: invalid argument");
        return SDSL_HIP_ERR_INVALID;
    }
    Mailbox * mb = nullptr;
    SH_TRY(mailbox_for(bv->h.device, &mb));
    std::lock_guard<std::mutex> lock(mb->m);
    SH_HIP(hipSetDevice(bv->h.device));
    mb->host[0] = arg;
    TimingPause pause;
    if (what == 0)
        SH_TRY(bv_launch_rank(bv->h.view, bit, mb->dev, 1, mb->dev + 8, mb->stream));
    else
        SH_TRY(bv_launch_select(bv->h.view, bit, mb->dev, 1, mb->dev + 8, mb->stream));
    SH_HIP(hipStreamSynchronize(mb->stream));
    *out = mb->host[8];
    return SDSL_HIP_OK;
}

sdsl_hip_status sdsl_hip_bv_gather_probe(sdsl_hip_bv_t bv, const uint64_t * d_idx, uint64_t n, uint64_t * d_out,
                                         void * stream)
{
    if (!bv || (n && (!d_idx || !d_out)) || (n && (!is_device_ptr(d_idx) || !is_device_ptr(d_out))))
    {
        set_error("bv_gather_probe: needs a handle and device arrays");
        return SDSL_HIP_ERR_INVALID;
    }
    hipStream_t s = (hipStream_t)stream;
    SH_HIP(hipSetDevice(bv->h.device));
    if (n == 0)
        return SDSL_HIP_OK;
    KernelTimer t(s);
    hipLaunchKernelGGL((k_rank_access_skeleton<4>), dim3(query_grid(n, kQPB * 4)), dim3(kBlock), 0, s, bv->h.view, d_idx,
                       d_out, n);
    SH_HIP(hipGetLastError());
    return SDSL_HIP_OK;
}

sdsl_hip_status sdsl_hip_bv_select_batch(sdsl_hip_bv_t bv, int32_t bit, const uint64_t * i, uint64_t n,
                                         uint64_t * out, void * stream)
{
    if (!bv || (bit != 0 && bit != 1) || (n && (!i || !out)))
    {
        set_error("bv_select_batch: invalid argument");
        return SDSL_HIP_ERR_INVALID;
    }
    hipStream_t s = (hipStream_t)stream;
    SH_HIP(hipSetDevice(bv->h.device));
    if (n >= kPipelineMinQueries && !is_device_ptr(i) && !is_device_ptr(out))
    {
        const BvView v = bv->h.view;
        return host_pipeline_u64(bv->h.device, i, out, n,
                                 [v, bit](const uint64_t * d_in, uint64_t * d_out, uint64_t cnt, hipStream_t st)
                                 { return bv_launch_select(v, bit, d_in, cnt, d_out, st); });
    }
    Staged in, o;
    SH_TRY(in.in(i, n * 8, s));
    SH_TRY(o.out(out, n * 8));
    SH_TRY(bv_select_dispatch(bv->h, bit, (const uint64_t *)in.dev, n, (uint64_t *)o.dev, s));
    SH_TRY(o.finish(s));
    if (in.host && !o.host)
        SH_HIP(hipStreamSynchronize(s));
    return SDSL_HIP_OK;
}

sdsl_hip_status sdsl_hip_bv_access_batch(sdsl_hip_bv_t bv, const uint64_t * idx, uint64_t n, uint8_t * out,
                                         void * stream)
{
    if (!bv || (n && (!idx || !out)))
    {
        set_error("bv_access_batch: invalid argument");
        return SDSL_HIP_ERR_INVALID;
    }
    hipStream_t s = (hipStream_t)stream;
    SH_HIP(hipSetDevice(bv->h.device));
    if (n == 0)
        return SDSL_HIP_OK;
    Staged in, o;
    SH_TRY(in.in(idx, n * 8, s));
    SH_TRY(o.out(out, n));
    {
        KernelTimer t(s);
        hipLaunchKernelGGL(k_access, dim3(grid_for(n, 256, 256u * 8u)), dim3(256), 0, s, bv->h.view,
                           (const uint64_t *)in.dev, (uint8_t *)o.dev, n);
    }
    SH_HIP(hipGetLastError());
    SH_TRY(o.finish(s));
    if (in.host && !o.host)
        SH_HIP(hipStreamSynchronize(s));
    return SDSL_HIP_OK;
}

sdsl_hip_status sdsl_hip_bv_export_words(sdsl_hip_bv_t bv, uint64_t * words_out, void * stream)
{
    if (!bv || (!words_out && bv->h.view.n_bits))
    {
        set_error("bv_export_words: invalid argument");
        return SDSL_HIP_ERR_INVALID;
    }
    hipStream_t s = (hipStream_t)stream;
    SH_HIP(hipSetDevice(bv->h.device));
    uint64_t nw = (bv->h.view.n_bits + 63) >> 6;
    if (nw == 0)
        return SDSL_HIP_OK;
    Staged o;
    SH_TRY(o.out(words_out, nw * 8));
    hipLaunchKernelGGL(k_export_words, dim3(grid_for(nw, 256, 65536)), dim3(256), 0, s, bv->h.view,
                       (uint64_t *)o.dev, nw);
    SH_HIP(hipGetLastError());
    SH_TRY(o.finish(s));
    return SDSL_HIP_OK;
}
}
