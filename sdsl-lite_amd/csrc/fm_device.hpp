// fm_device.hpp — byte alphabet tables of the FM-index (csa_alphabet_strategy.hpp:136-308: C, char2comp) as staged in
// LDS by the count kernels, shared by fm.hip (plain wavelet tree) and wt_rrr.hip (rrr-compressed wavelet tree).
#pragma once
#include "common.hpp"
#include "wt_device.hpp"

namespace sdslhip {

struct FmTables
{
    uint64_t C[257];        // C[cc] = number of symbols smaller than comp2char[cc]; C[sigma] = size
    uint8_t char2comp[256]; // 0 for absent bytes (and for the sentinel itself)
};

__device__ __forceinline__ void fm_stage_tables(FmTables * lds, const FmTables * g)
{
    const uint64_t * src = reinterpret_cast<const uint64_t *>(g);
    uint64_t * dst = reinterpret_cast<uint64_t *>(lds);
    for (unsigned i = threadIdx.x; i < sizeof(FmTables) / 8; i += blockDim.x)
        dst[i] = src[i];
    // callers follow with wt_stage_tables(), which ends in __syncthreads()
}

struct WtHost;
sdsl_hip_status fm_rrr_launch_count(const WtHost & wt, const FmTables * d_tab, uint64_t csa_size, const uint8_t * d_pats,
                                    uint32_t m, const uint64_t * d_offsets, const uint32_t * d_order, uint64_t n_pat,
                                    uint64_t * d_cnt, uint64_t * d_l, uint64_t * d_r, hipStream_t s);
sdsl_hip_status fm_rrr_launch_backward_step(const WtHost & wt, const FmTables * d_tab, uint64_t csa_size,
                                            const uint64_t * d_l, const uint64_t * d_r, const uint8_t * d_c, uint64_t n,
                                            uint64_t * d_lo, uint64_t * d_ro, hipStream_t s);

} // namespace sdslhip
