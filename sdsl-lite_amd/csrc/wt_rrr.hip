// wt_rrr.hip — the wavelet-tree / FM-index kernels over an rrr-compressed bit vector:
// wt_huff<rrr_vector<63>> and csa_wt<wt_huff<rrr_vector<63>>> (the reference's FULL test
// list, test/wt_byte_test.cpp:42-58; SDSL's README FM-index is this family).  Answers are those of the plain
// wavelet tree — the bit vector is the same, only its representation differs — so the parity tests reuse the same
// oracle and golden vectors.
//
// Execution model: one QUERY PER LANE, no cooperation (rrr_device.hpp: a lane needs 40 bytes of a superblock record
// plus the offset field for one rank).  These kernels are VALU-bound by the block decoder, not by memory, so the
// loops are flat — one iteration is one tree level of whatever the lane is working on — and lanes never wait for
// each other at query or character boundaries.  LDS holds the node table (13.5 KiB) and the binomial columns 0..10 (5.8 KiB).
#include "fm_device.hpp"
#include "wt_host.hpp"

namespace sdslhip {

// 256 threads and the compact binomial table (rrr_device.hpp: RrrTablesWt): 21 KiB of LDS per block, so the kernels' 70 VGPRs set the
// occupancy — 7 waves per SIMD; with the full table (33 KiB) in blocks of 512 it was 6, set by LDS (SDSL_HIP_WT_RRR_BLOCK=512 at compile
// time: that block size, for A/B)
#ifndef SDSL_HIP_WT_RRR_BLOCK
#define SDSL_HIP_WT_RRR_BLOCK 256
#endif
constexpr unsigned kWtRrrBlock = SDSL_HIP_WT_RRR_BLOCK;
constexpr unsigned kWtRrrWaves = kWtRrrBlock == 256 ? 7 : 6; // waves per SIMD the register allocation aims at (second argument of __launch_bounds__)

// wt_pc::rank (wt_pc.hpp:371-399)
__global__ __launch_bounds__(kWtRrrBlock, kWtRrrWaves) void k_wt_rank_rrr(WtView wt, const uint64_t * __restrict__ iq,
                                                             const uint8_t * __restrict__ cq, uint64_t * __restrict__ out,
                                                             uint64_t n)
{
    __shared__ WtTables T;
    __shared__ RrrTablesWt RT;
    rrr_stage_tables(&RT, wt.rrr.tables);
    wt_stage_tables(&T, wt.tables);
    for (uint64_t q = (uint64_t)blockIdx.x * kWtRrrBlock + threadIdx.x; q < n; q += (uint64_t)gridDim.x * kWtRrrBlock)
    {
        const uint64_t i = iq[q];
        const unsigned c = cq[q];
        uint64_t result;
        if (i > wt.size)
            result = SDSL_HIP_NPOS;
        else if (T.c_to_leaf[c] == kWtUndef)
            result = 0; // c does not occur (:374-377)
        else if (wt.sigma == 1)
            result = i; // (:378-381)
        else
        {
            uint64_t p = T.path[c];
            unsigned left = (unsigned)(p >> 56);
            unsigned v = 0;
            result = i;
            while (left && result)
            { // early exit on 0 like the reference (:386)
                const uint64_t r = rrr_rank1(wt.rrr, &RT, T.bv_pos[v] + result) - T.bv_pos_rank[v];
                const unsigned bit = (unsigned)(p & 1);
                result = bit ? r : result - r;
                v = T.child[v][bit];
                p >>= 1;
                --left;
            }
        }
        out[q] = result;
    }
}

// wt_pc::inverse_select / operator[] (wt_pc.hpp:411-430, 336-357)
template <bool WITH_RANK>
__global__ __launch_bounds__(kWtRrrBlock, kWtRrrWaves) void k_wt_invsel_rrr(WtView wt, const uint64_t * __restrict__ iq,
                                                               uint64_t * __restrict__ out_rank,
                                                               uint8_t * __restrict__ out_c, uint64_t n)
{
    __shared__ WtTables T;
    __shared__ RrrTablesWt RT;
    rrr_stage_tables(&RT, wt.rrr.tables);
    wt_stage_tables(&T, wt.tables);
    for (uint64_t q = (uint64_t)blockIdx.x * kWtRrrBlock + threadIdx.x; q < n; q += (uint64_t)gridDim.x * kWtRrrBlock)
    {
        uint64_t i = iq[q];
        const bool valid = i < wt.size;
        unsigned v = 0;
        while (valid && T.child[v][0] != kWtUndef)
        {
            unsigned bit = 0;
            const uint64_t r = rrr_rank1(wt.rrr, &RT, T.bv_pos[v] + i, &bit) - T.bv_pos_rank[v];
            i = bit ? r : i - r;
            v = T.child[v][bit];
        }
        out_c[q] = valid ? (uint8_t)T.bv_pos_rank[v] : 0xFF;
        if (WITH_RANK)
            out_rank[q] = valid ? i : SDSL_HIP_NPOS;
    }
}

// wt_pc::select (wt_pc.hpp:441-480): bottom-up, one rrr select per level whose bit value is the path bit
__global__ __launch_bounds__(kWtRrrBlock, kWtRrrWaves) void k_wt_select_rrr(WtView wt, const uint64_t * __restrict__ occ,
                                                               const uint64_t * __restrict__ iq,
                                                               const uint8_t * __restrict__ cq,
                                                               uint64_t * __restrict__ out, uint64_t n)
{
    __shared__ WtTables T;
    __shared__ RrrTablesWt RT;
    rrr_stage_tables(&RT, wt.rrr.tables);
    wt_stage_tables(&T, wt.tables);
    for (uint64_t q = (uint64_t)blockIdx.x * kWtRrrBlock + threadIdx.x; q < n; q += (uint64_t)gridDim.x * kWtRrrBlock)
    {
        const uint64_t i = iq[q];
        const unsigned c = cq[q];
        unsigned v = T.c_to_leaf[c];
        uint64_t res;
        if (v == kWtUndef)
            res = wt.size; // c not in the text (wt_pc.hpp:447-450)
        else if (i == 0 || i > occ[c])
            res = SDSL_HIP_NPOS; // outside SDSL's precondition
        else if (wt.sigma == 1)
            res = i - 1 < wt.size ? i - 1 : wt.size;
        else
        {
            res = i - 1;
            uint64_t p = T.path[c];
            unsigned left = (unsigned)(p >> 56);
            p <<= (64 - left);
            for (; left; --left, p <<= 1)
            {
                const unsigned par = T.parent[v];
                uint64_t pos;
                if ((p >> 63) == 0) // v is a left child: zeros of the parent's slice
                    pos = rrr_select<0>(wt.rrr, &RT, T.bv_pos[par] - T.bv_pos_rank[par] + res);
                else
                    pos = rrr_select<1>(wt.rrr, &RT, T.bv_pos_rank[par] + res);
                res = pos - T.bv_pos[par];
                v = par;
            }
        }
        out[q] = res;
    }
}

// count / interval (suffix_array_algorithm.hpp:228-248, 464-471), one pattern per lane.  Both cascades of an LF
// step (rank at l and at r+1, :195-196) advance level by level (rrr_rank2).
// VERIFY (count only; the whole suffix array and the text are resident): a search that is down to a FEW suffixes (<= kFmVerifyMax; round 3:
// one) with two or more characters to go stops and leaves the pending word of fm_device.hpp for k_fm_verify (fm.hip), which compares
// them with the text in front of each suffix — on this index every LF step it saves is a cascade of block decodes (8 fabric requests per
// character against 1 + s for the comparison), so the text takes over as soon as the interval is that narrow.
template <bool WANT_IVAL, bool VERIFY = false>
__global__ __launch_bounds__(kWtRrrBlock, kWtRrrWaves) void k_fm_count_rrr(WtView wt, const FmTables * __restrict__ ftab, FmJump J, FmDeep D,
                                                              uint64_t csa_size, const uint8_t * __restrict__ pats,
                                                              uint32_t m, const uint64_t * __restrict__ offsets,
                                                              const uint32_t * __restrict__ order, uint64_t n_pat,
                                                              uint64_t * __restrict__ out_cnt,
                                                              uint64_t * __restrict__ out_l, uint64_t * __restrict__ out_r)
{
    __shared__ WtTables T;
    __shared__ RrrTablesWt RT;
    __shared__ FmTables F;
    fm_stage_tables(&F, ftab);
    rrr_stage_tables(&RT, wt.rrr.tables);
    wt_stage_tables(&T, wt.tables);
    for (uint64_t slot = (uint64_t)blockIdx.x * kWtRrrBlock + threadIdx.x; slot < n_pat;
         slot += (uint64_t)gridDim.x * kWtRrrBlock)
    {
        const uint64_t q = order ? order[slot] : slot;
        uint64_t begin = offsets ? offsets[q] : q * (uint64_t)m;
        uint64_t end = offsets ? offsets[q + 1] : begin + m;
        uint64_t l = 0, r = csa_size - 1;
        if (!WANT_IVAL && end - begin > csa_size)
        { // count(): a pattern longer than the text cannot occur (:466-467)
            l = 1;
            r = 0;
            end = begin;
        }
        // the pattern's last D.k bytes in ONE bucket of the k-mer hash table (fm_count2.hip: the SA interval of every k-mer that occurs
        // in the text; a k-mer that is not in it does not occur) — every LF step it saves is a cascade of block decodes here; else the
        // dense table over the compact alphabet
        uint64_t it = end;
        bool started = false;
        if (D.tab && end - begin >= D.k && end - begin <= csa_size)
        {
            const uint64_t key = load_tail8(pats, end) >> (8 * (8 - D.k));
            if (!has_zero_byte(key, D.k))
            {
                uint64_t lo = 0, hi = 0;
                if (lane_deep_find(D, key, lo, hi))
                {
                    l = lo;
                    r = hi - 1;
                    it = end - D.k;
                }
                else
                { // count 0; the interval of an absent pattern is (l, r) with r + 1 == l as the reference leaves it after its last step
                    l = 1;
                    r = 0;
                    it = begin;
                }
                started = true;
            }
        }
        if (!started)
            it = fm_jump_start(J, F, pats, begin, end, l, r);
        unsigned c_next = it > begin ? pats[it - 1] : 0;
        unsigned left = 0, v = 0;
        uint64_t a = 0, b = 0, p = 0, cb = 0;
        bool pending = false;
        for (;;)
        {
            if (left == 0)
            { // next character (suffix_array_algorithm.hpp:176-200)
                if (!(it > begin && r + 1 - l > 0))
                    break;
                if (VERIFY && !WANT_IVAL && r - l < kFmVerifyMax && it > begin + 1 && it < end && it - begin < (1u << 28) && it - begin <= 16 &&
                    !fm_tail_has_zero(load_tail16(pats, it), (uint32_t)(it - begin)))
                {
                    pending = true;
                    break;
                }
                --it;
                const unsigned c = c_next;
                if (it > begin)
                    c_next = pats[it - 1];
                const unsigned cc = F.char2comp[c];
                if (cc == 0 && c > 0)
                { // character does not occur (:180-184)
                    l = 1;
                    r = 0;
                    continue;
                }
                cb = F.C[cc];
                if (l == 0 && r + 1 == csa_size)
                { // whole interval: no rank needed (:188-192)
                    l = cb;
                    r = F.C[cc + 1] - 1;
                    continue;
                }
                a = l;
                b = r + 1;
                if (wt.sigma == 1)
                {
                    l = cb + a;
                    r = cb + b - 1;
                    continue;
                }
                p = T.path[c]; // the symbol occurs (char2comp said so): walk its path
                left = (unsigned)(p >> 56);
                v = 0;
            }
            const uint64_t bp = T.bv_pos[v], br = T.bv_pos_rank[v];
            uint64_t ra, rb;
            rrr_rank2(wt.rrr, &RT, bp + a, bp + b, ra, rb);
            ra -= br;
            rb -= br;
            const unsigned bit = (unsigned)(p & 1);
            a = bit ? ra : a - ra;
            b = bit ? rb : b - rb;
            v = T.child[v][bit];
            p >>= 1;
            --left;
            if (b == 0)
            { // a <= b: both chains are 0 from here on (wt_pc.hpp:386)
                a = 0;
                left = 0;
            }
            if (left == 0)
            {
                l = cb + a;
                r = cb + b - 1;
            }
        }
        if (WANT_IVAL)
        {
            out_l[q] = l;
            out_r[q] = r;
        }
        else
            out_cnt[q] = VERIFY && pending ? fm_pending_word(32, l, r + 1 - l, it - begin) : r + 1 - l;
    }
}

// one LF step per element (suffix_array_algorithm.hpp:167-201)
__global__ __launch_bounds__(kWtRrrBlock, kWtRrrWaves) void k_fm_backward_step_rrr(WtView wt, const FmTables * __restrict__ ftab,
                                                                      uint64_t csa_size, const uint64_t * __restrict__ lq,
                                                                      const uint64_t * __restrict__ rq,
                                                                      const uint8_t * __restrict__ cq, uint64_t n,
                                                                      uint64_t * __restrict__ out_l,
                                                                      uint64_t * __restrict__ out_r)
{
    __shared__ WtTables T;
    __shared__ RrrTablesWt RT;
    __shared__ FmTables F;
    fm_stage_tables(&F, ftab);
    rrr_stage_tables(&RT, wt.rrr.tables);
    wt_stage_tables(&T, wt.tables);
    for (uint64_t q = (uint64_t)blockIdx.x * kWtRrrBlock + threadIdx.x; q < n; q += (uint64_t)gridDim.x * kWtRrrBlock)
    {
        const uint64_t l = lq[q], r = rq[q];
        const unsigned c = cq[q];
        uint64_t lo = SDSL_HIP_NPOS, ro = SDSL_HIP_NPOS; // outside SDSL's precondition (asserts :177-178)
        if (l <= r && r < csa_size)
        {
            const unsigned cc = F.char2comp[c];
            if (cc == 0 && c > 0)
            {
                lo = 1;
                ro = 0;
            }
            else
            {
                const uint64_t cb = F.C[cc];
                if (l == 0 && r + 1 == csa_size)
                {
                    lo = cb;
                    ro = F.C[cc + 1] - 1;
                }
                else
                {
                    uint64_t a = l, b = r + 1;
                    if (wt.sigma != 1)
                    {
                        uint64_t p = T.path[c];
                        unsigned left = (unsigned)(p >> 56);
                        unsigned v = 0;
                        for (; left && b; --left, p >>= 1)
                        {
                            const uint64_t bp = T.bv_pos[v], br = T.bv_pos_rank[v];
                            uint64_t ra, rb;
                            rrr_rank2(wt.rrr, &RT, bp + a, bp + b, ra, rb);
                            ra -= br;
                            rb -= br;
                            const unsigned bit = (unsigned)(p & 1);
                            a = bit ? ra : a - ra;
                            b = bit ? rb : b - rb;
                            v = T.child[v][bit];
                        }
                        if (b == 0)
                            a = 0;
                    }
                    lo = cb + a;
                    ro = cb + b - 1;
                }
            }
        }
        out_l[q] = lo;
        out_r[q] = ro;
    }
}

static unsigned wt_rrr_grid(uint64_t n)
{
    return grid_for(n, kWtRrrBlock, 256u * (kWtRrrBlock == 256 ? 7u : 3u));
}

sdsl_hip_status wt_rrr_launch_rank(const WtHost & wt, const uint64_t * d_i, const uint8_t * d_c, uint64_t n,
                                   uint64_t * d_out, hipStream_t s)
{
    if (n == 0)
        return SDSL_HIP_OK;
    KernelTimer t(s);
    hipLaunchKernelGGL(k_wt_rank_rrr, dim3(wt_rrr_grid(n)), dim3(kWtRrrBlock), 0, s, wt.view(), d_i, d_c, d_out, n);
    SH_HIP(hipGetLastError());
    return SDSL_HIP_OK;
}

sdsl_hip_status wt_rrr_launch_select(const WtHost & wt, const uint64_t * d_occ, const uint64_t * d_i, const uint8_t * d_c,
                                     uint64_t n, uint64_t * d_out, hipStream_t s)
{
    if (n == 0)
        return SDSL_HIP_OK;
    KernelTimer t(s);
    hipLaunchKernelGGL(k_wt_select_rrr, dim3(wt_rrr_grid(n)), dim3(kWtRrrBlock), 0, s, wt.view(), d_occ, d_i, d_c, d_out,
                       n);
    SH_HIP(hipGetLastError());
    return SDSL_HIP_OK;
}

sdsl_hip_status wt_rrr_launch_inverse_select(const WtHost & wt, const uint64_t * d_i, uint64_t n, uint64_t * d_rank,
                                             uint8_t * d_c, hipStream_t s)
{
    if (n == 0)
        return SDSL_HIP_OK;
    KernelTimer t(s);
    if (d_rank)
        hipLaunchKernelGGL((k_wt_invsel_rrr<true>), dim3(wt_rrr_grid(n)), dim3(kWtRrrBlock), 0, s, wt.view(), d_i, d_rank,
                           d_c, n);
    else
        hipLaunchKernelGGL((k_wt_invsel_rrr<false>), dim3(wt_rrr_grid(n)), dim3(kWtRrrBlock), 0, s, wt.view(), d_i,
                           (uint64_t *)nullptr, d_c, n);
    SH_HIP(hipGetLastError());
    return SDSL_HIP_OK;
}

sdsl_hip_status fm_rrr_launch_count(const WtHost & wt, const FmTables * d_tab, FmJump jump, FmDeep deep, uint64_t csa_size,
                                    const uint8_t * d_pats, uint32_t m, const uint64_t * d_offsets, const uint32_t * d_order,
                                    uint64_t n_pat, uint64_t * d_cnt, uint64_t * d_l, uint64_t * d_r, hipStream_t s, bool verify)
{
    if (n_pat == 0)
        return SDSL_HIP_OK;
    if (csa_size >= (UINT64_C(1) << 32) || d_l)
        deep.tab = nullptr; // (the lane lookup reads narrow entries; intervals keep the reference's exact (l, r) of an absent pattern)
    if (d_l)
        hipLaunchKernelGGL((k_fm_count_rrr<true>), dim3(wt_rrr_grid(n_pat)), dim3(kWtRrrBlock), 0, s, wt.view(), d_tab,
                           jump, deep, csa_size, d_pats, m, d_offsets, d_order, n_pat, (uint64_t *)nullptr, d_l, d_r);
    else if (verify && csa_size < (UINT64_C(1) << 32))
        hipLaunchKernelGGL((k_fm_count_rrr<false, true>), dim3(wt_rrr_grid(n_pat)), dim3(kWtRrrBlock), 0, s, wt.view(), d_tab,
                           jump, deep, csa_size, d_pats, m, d_offsets, d_order, n_pat, d_cnt, (uint64_t *)nullptr, (uint64_t *)nullptr);
    else
        hipLaunchKernelGGL((k_fm_count_rrr<false>), dim3(wt_rrr_grid(n_pat)), dim3(kWtRrrBlock), 0, s, wt.view(), d_tab,
                           jump, deep, csa_size, d_pats, m, d_offsets, d_order, n_pat, d_cnt, (uint64_t *)nullptr, (uint64_t *)nullptr);
    SH_HIP(hipGetLastError());
    return SDSL_HIP_OK;
}

sdsl_hip_status fm_rrr_launch_backward_step(const WtHost & wt, const FmTables * d_tab, uint64_t csa_size,
                                            const uint64_t * d_l, const uint64_t * d_r, const uint8_t * d_c, uint64_t n,
                                            uint64_t * d_lo, uint64_t * d_ro, hipStream_t s)
{
    if (n == 0)
        return SDSL_HIP_OK;
    hipLaunchKernelGGL(k_fm_backward_step_rrr, dim3(wt_rrr_grid(n)), dim3(kWtRrrBlock), 0, s, wt.view(), d_tab, csa_size,
                       d_l, d_r, d_c, n, d_lo, d_ro);
    SH_HIP(hipGetLastError());
    return SDSL_HIP_OK;
}

} // namespace sdslhip
