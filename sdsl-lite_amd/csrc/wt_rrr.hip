// wt_rrr.hip — the wavelet-tree / FM-index kernels over an rrr-compressed bit vector:
// wt_huff<rrr_vector<63>> and csa_wt<wt_huff<rrr_vector<63>>> (the reference's FULL test
// list, test/wt_byte_test.cpp:42-58; SDSL's README FM-index is this family).  Answers are those of the plain
// wavelet tree — the bit vector is the same, only its representation differs — so the parity tests reuse the same
// oracle and golden vectors.
//
// Execution model: one QUERY PER LANE, four lanes cooperating.  All four queries of a quad advance one tree level
// per iteration; per level the quad runs the cooperative half of an rrr rank (record fetch, class-byte prefix, offset
// field; rrr_device.hpp rrr_rank_head) once for each of its queries, then every lane decodes the 63-bit block of its
// own query (wt_device.hpp quad4_rrr_rank1).  LDS holds the node table (13.5 KiB) and the binomial table (32 KiB).
#include "fm_device.hpp"
#include "wt_host.hpp"

namespace sdslhip {

constexpr unsigned kWtRrrBlock = 512;

// wt_pc::rank (wt_pc.hpp:371-399), one query per lane
__global__ __launch_bounds__(kWtRrrBlock) void k_wt_rank_rrr(WtView wt, const uint64_t * __restrict__ iq,
                                                             const uint8_t * __restrict__ cq, uint64_t * __restrict__ out,
                                                             uint64_t n)
{
    __shared__ WtTables T;
    __shared__ RrrTables RT;
    rrr_stage_tables(&RT, wt.rrr.tables);
    wt_stage_tables(&T, wt.tables);
    const int s = threadIdx.x & 3;
    for (uint64_t base = (uint64_t)blockIdx.x * kWtRrrBlock; base < n; base += (uint64_t)gridDim.x * kWtRrrBlock)
    {
        const uint64_t q = base + threadIdx.x;
        const uint64_t i = q < n ? iq[q] : 0;
        const unsigned c = q < n ? cq[q] : 0;
        const bool valid = q < n && i <= wt.size;
        const bool leaf = T.c_to_leaf[c] != kWtUndef;
        uint64_t p = T.path[c];
        const unsigned len = (unsigned)(p >> 56);
        uint64_t result = i;
        unsigned v = 0;
        const bool run = valid && leaf && wt.sigma != 1;
        for (unsigned l = 0;; ++l)
        {
            const bool act = run && l < len && result != 0; // early exit on 0 like the reference (:386)
            if (!quad_any(act))
                break;
            uint64_t r = quad4_rrr_rank1(wt.rrr, &RT, s, T.bv_pos[v] + result, act) - T.bv_pos_rank[v];
            if (act)
            {
                unsigned bit = (unsigned)(p & 1);
                result = bit ? r : result - r;
                v = T.child[v][bit];
                p >>= 1;
            }
        }
        if (q < n)
            out[q] = !valid ? SDSL_HIP_NPOS : (!leaf ? 0 : result);
    }
}

// wt_pc::inverse_select / operator[] (wt_pc.hpp:411-430, 336-357), one query per lane
template <bool WITH_RANK>
__global__ __launch_bounds__(kWtRrrBlock) void k_wt_invsel_rrr(WtView wt, const uint64_t * __restrict__ iq,
                                                               uint64_t * __restrict__ out_rank,
                                                               uint8_t * __restrict__ out_c, uint64_t n)
{
    __shared__ WtTables T;
    __shared__ RrrTables RT;
    rrr_stage_tables(&RT, wt.rrr.tables);
    wt_stage_tables(&T, wt.tables);
    const int s = threadIdx.x & 3;
    for (uint64_t base = (uint64_t)blockIdx.x * kWtRrrBlock; base < n; base += (uint64_t)gridDim.x * kWtRrrBlock)
    {
        const uint64_t q = base + threadIdx.x;
        uint64_t i = q < n ? iq[q] : 0;
        const bool valid = q < n && i < wt.size;
        unsigned v = 0;
        for (;;)
        {
            const bool act = valid && T.child[v][0] != kWtUndef;
            if (!quad_any(act))
                break;
            unsigned bit = 0;
            uint64_t r = quad4_rrr_rank1(wt.rrr, &RT, s, T.bv_pos[v] + i, act, &bit) - T.bv_pos_rank[v];
            if (act)
            {
                i = bit ? r : i - r;
                v = T.child[v][bit];
            }
        }
        if (q < n)
        {
            out_c[q] = valid ? (uint8_t)T.bv_pos_rank[v] : 0xFF;
            if (WITH_RANK)
                out_rank[q] = valid ? i : SDSL_HIP_NPOS;
        }
    }
}

// count / interval (suffix_array_algorithm.hpp:228-248, 464-471), one pattern per lane.  Both cascades of an LF
// step (rank at l and at r+1, :195-196) advance level by level.
template <bool WANT_IVAL>
__global__ __launch_bounds__(kWtRrrBlock) void k_fm_count_rrr(WtView wt, const FmTables * __restrict__ ftab,
                                                              uint64_t csa_size, const uint8_t * __restrict__ pats,
                                                              uint32_t m, const uint64_t * __restrict__ offsets,
                                                              const uint32_t * __restrict__ order, uint64_t n_pat,
                                                              uint64_t * __restrict__ out_cnt,
                                                              uint64_t * __restrict__ out_l, uint64_t * __restrict__ out_r)
{
    __shared__ WtTables T;
    __shared__ RrrTables RT;
    __shared__ FmTables F;
    fm_stage_tables(&F, ftab);
    rrr_stage_tables(&RT, wt.rrr.tables);
    wt_stage_tables(&T, wt.tables);
    const int s = threadIdx.x & 3;
    for (uint64_t base = (uint64_t)blockIdx.x * kWtRrrBlock; base < n_pat; base += (uint64_t)gridDim.x * kWtRrrBlock)
    {
        const uint64_t slot = base + threadIdx.x;
        const bool valid = slot < n_pat;
        const uint64_t q = valid ? (order ? order[slot] : slot) : 0;
        uint64_t begin = 0, end = 0;
        if (valid)
        {
            begin = offsets ? offsets[q] : q * (uint64_t)m;
            end = offsets ? offsets[q + 1] : begin + m;
        }
        uint64_t l = 0, r = csa_size - 1;
        if (!WANT_IVAL && end - begin > csa_size)
        { // count(): a pattern longer than the text cannot occur (:466-467)
            l = 1;
            r = 0;
            end = begin;
        }
        uint64_t it = end;
        unsigned c_next = valid && it > begin ? pats[it - 1] : 0;
        // Flat loop: one iteration is one tree level of whatever character the lane is at, so lanes do not wait for
        // each other at character boundaries (Huffman paths differ in length).
        bool alive = valid;
        unsigned left = 0, v = 0;
        uint64_t a = 0, b = 0, p = 0, cb = 0;
        for (;;)
        {
            while (alive && left == 0)
            { // next character (suffix_array_algorithm.hpp:176-200); lane-divergent and short
                if (!(it > begin && r + 1 - l > 0))
                {
                    alive = false;
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
            if (!quad_any(alive))
                break;
            const uint64_t bp = T.bv_pos[v], br = T.bv_pos_rank[v];
            uint64_t ra, rb;
            quad4_rrr_rank2(wt.rrr, &RT, s, bp + a, bp + b, alive, ra, rb);
            if (alive)
            {
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
                    v = 0;
                }
            }
        }
        if (valid)
        {
            if (WANT_IVAL)
            {
                out_l[q] = l;
                out_r[q] = r;
            }
            else
                out_cnt[q] = r + 1 - l;
        }
    }
}

// one LF step per element (suffix_array_algorithm.hpp:167-201)
__global__ __launch_bounds__(kWtRrrBlock) void k_fm_backward_step_rrr(WtView wt, const FmTables * __restrict__ ftab,
                                                                      uint64_t csa_size, const uint64_t * __restrict__ lq,
                                                                      const uint64_t * __restrict__ rq,
                                                                      const uint8_t * __restrict__ cq, uint64_t n,
                                                                      uint64_t * __restrict__ out_l,
                                                                      uint64_t * __restrict__ out_r)
{
    __shared__ WtTables T;
    __shared__ RrrTables RT;
    __shared__ FmTables F;
    fm_stage_tables(&F, ftab);
    rrr_stage_tables(&RT, wt.rrr.tables);
    wt_stage_tables(&T, wt.tables);
    const int s = threadIdx.x & 3;
    for (uint64_t base = (uint64_t)blockIdx.x * kWtRrrBlock; base < n; base += (uint64_t)gridDim.x * kWtRrrBlock)
    {
        const uint64_t q = base + threadIdx.x;
        const bool valid = q < n;
        const uint64_t l = valid ? lq[q] : 0, r = valid ? rq[q] : 0;
        const unsigned c = valid ? cq[q] : 0;
        uint64_t lo = SDSL_HIP_NPOS, ro = SDSL_HIP_NPOS, cb = 0, a = 0, b = 0, p = 0;
        unsigned len = 0;
        bool need = false;
        if (valid && l <= r && r < csa_size)
        {
            unsigned cc = F.char2comp[c];
            if (cc == 0 && c > 0)
            {
                lo = 1;
                ro = 0;
            }
            else
            {
                cb = F.C[cc];
                if (l == 0 && r + 1 == csa_size)
                {
                    lo = cb;
                    ro = F.C[cc + 1] - 1;
                }
                else
                {
                    a = l;
                    b = r + 1;
                    if (wt.sigma != 1)
                    {
                        p = T.path[c];
                        len = (unsigned)(p >> 56);
                        need = true;
                    }
                    else
                    {
                        lo = cb + a;
                        ro = cb + b - 1;
                    }
                }
            }
        }
        unsigned v = 0;
        for (unsigned lev = 0;; ++lev)
        {
            const bool act = need && lev < len && b != 0;
            if (!quad_any(act))
                break;
            const uint64_t bp = T.bv_pos[v], br = T.bv_pos_rank[v];
            uint64_t ra, rb;
            quad4_rrr_rank2(wt.rrr, &RT, s, bp + a, bp + b, act, ra, rb);
            ra -= br;
            rb -= br;
            if (act)
            {
                unsigned bit = (unsigned)(p & 1);
                a = bit ? ra : a - ra;
                b = bit ? rb : b - rb;
                v = T.child[v][bit];
                p >>= 1;
            }
        }
        if (need)
        {
            if (b == 0)
                a = 0;
            lo = cb + a;
            ro = cb + b - 1;
        }
        if (valid)
        {
            out_l[q] = lo;
            out_r[q] = ro;
        }
    }
}

// select on the rrr vector for this lane's own argument: position of the (k0+1)-th BIT-valued bit, where the bit
// value differs per lane.  The cooperative head (rrr_select_head) runs once per active query of the quad; queries
// are rotated through quad lane 0 so the loop body exists once per bit value.
__device__ __forceinline__ uint64_t quad4_rrr_select(const RrrView & v, const RrrTables * RT, int s, uint64_t k0,
                                                     unsigned bit, bool act)
{
    SelTail mine;
    mine.r = v.rec;
    mine.bstart = 0;
    mine.k = mine.blen = mine.rel = mine.want = 0;
    uint64_t rk = act ? k0 : 0;
    unsigned rflags = (act ? 1u : 0u) | (bit << 1);
#pragma unroll 1
    for (int u = 0; u < 4; ++u)
    {
        const unsigned f = quad_bcast0(rflags);
        const uint64_t k = quad_bcast0_u64(rk);
        if (f & 1u)
        { // quad-uniform
            SelTail t = (f & 2u) ? rrr_select_head<1>(v, RT, s, k) : rrr_select_head<0>(v, RT, s, k);
            if (s == u)
                mine = t;
        }
        // rotate the queries by one lane: quad_perm:[1,2,3,0]
        rflags = (unsigned)__builtin_amdgcn_update_dpp(0, (int)rflags, 0x39, 0xF, 0xF, true);
        unsigned lo = (unsigned)__builtin_amdgcn_update_dpp(0, (int)(unsigned)rk, 0x39, 0xF, 0xF, true);
        unsigned hi = (unsigned)__builtin_amdgcn_update_dpp(0, (int)(unsigned)(rk >> 32), 0x39, 0xF, 0xF, true);
        rk = ((uint64_t)hi << 32) | lo;
    }
    if (!act)
        return 0;
    const uint64_t ptr = mine.r[1] & ((UINT64_C(1) << 48) - 1);
    const uint64_t nr = rrr_field(v, mine.r, ptr, mine.rel, RT->space[mine.k]);
    uint64_t bits = rrr_decode_block(RT, mine.k, nr);
    if (!bit)
        bits = ~bits & lo_set(mine.blen);
    return mine.bstart + sel64(bits, mine.want + 1);
}

// wt_pc::select (wt_pc.hpp:441-480), one query per lane: bottom-up, one rrr select per level whose bit value is
// the path bit of that level.
__global__ __launch_bounds__(kWtRrrBlock) void k_wt_select_rrr(WtView wt, const uint64_t * __restrict__ occ,
                                                               const uint64_t * __restrict__ iq,
                                                               const uint8_t * __restrict__ cq,
                                                               uint64_t * __restrict__ out, uint64_t n)
{
    __shared__ WtTables T;
    __shared__ RrrTables RT;
    rrr_stage_tables(&RT, wt.rrr.tables);
    wt_stage_tables(&T, wt.tables);
    const int s = threadIdx.x & 3;
    for (uint64_t base = (uint64_t)blockIdx.x * kWtRrrBlock; base < n; base += (uint64_t)gridDim.x * kWtRrrBlock)
    {
        const uint64_t q = base + threadIdx.x;
        const uint64_t i = q < n ? iq[q] : 0;
        const unsigned c = q < n ? cq[q] : 0;
        unsigned v = T.c_to_leaf[c];
        const bool present = v != kWtUndef;
        const bool in_dom = present && i >= 1 && i <= occ[c];
        const bool run = q < n && in_dom && wt.sigma != 1;
        uint64_t res = i - 1;
        uint64_t p = T.path[c];
        const unsigned len = (unsigned)(p >> 56);
        p = len ? p << (64 - len) : 0;
        for (unsigned l = 0;; ++l)
        {
            const bool act = run && l < len;
            if (!quad_any(act))
                break;
            const unsigned par = act ? T.parent[v] : 0;
            const unsigned bit = (unsigned)(p >> 63);
            const uint64_t k0 = bit ? T.bv_pos_rank[par] + res : T.bv_pos[par] - T.bv_pos_rank[par] + res;
            const uint64_t pos = quad4_rrr_select(wt.rrr, &RT, s, k0, bit, act);
            if (act)
            {
                res = pos - T.bv_pos[par];
                v = par;
                p <<= 1;
            }
        }
        if (q < n)
        {
            uint64_t r;
            if (!present)
                r = wt.size; // c not in the text (wt_pc.hpp:447-450)
            else if (!in_dom)
                r = SDSL_HIP_NPOS; // outside SDSL's precondition
            else if (wt.sigma == 1)
                r = i - 1 < wt.size ? i - 1 : wt.size;
            else
                r = res;
            out[q] = r;
        }
    }
}

static unsigned wt_rrr_grid(uint64_t n)
{
    return grid_for(n, kWtRrrBlock, 256u * 3u);
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

sdsl_hip_status fm_rrr_launch_count(const WtHost & wt, const FmTables * d_tab, uint64_t csa_size, const uint8_t * d_pats,
                                    uint32_t m, const uint64_t * d_offsets, const uint32_t * d_order, uint64_t n_pat,
                                    uint64_t * d_cnt, uint64_t * d_l, uint64_t * d_r, hipStream_t s)
{
    if (n_pat == 0)
        return SDSL_HIP_OK;
    if (d_l)
        hipLaunchKernelGGL((k_fm_count_rrr<true>), dim3(wt_rrr_grid(n_pat)), dim3(kWtRrrBlock), 0, s, wt.view(), d_tab,
                           csa_size, d_pats, m, d_offsets, d_order, n_pat, (uint64_t *)nullptr, d_l, d_r);
    else
        hipLaunchKernelGGL((k_fm_count_rrr<false>), dim3(wt_rrr_grid(n_pat)), dim3(kWtRrrBlock), 0, s, wt.view(), d_tab,
                           csa_size, d_pats, m, d_offsets, d_order, n_pat, d_cnt, (uint64_t *)nullptr, (uint64_t *)nullptr);
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
