// workload_dev.hip — the benchmark's seeded input streams (workload.cpp: std::mt19937_64 outputs) produced ON
// the device from generator checkpoints, so that a rank of the multi-GPU bench holds no host copy of its 8 GB of positions
// (eight ranks on one node would hold 80 GB before the first kernel).  Not on the query path.
//
// MT19937-64 is sequential, but its state can be saved: sdsl_hip_util_mt_checkpoints walks the generator once on the host and
// keeps the 312-word state every `stride` draws (a few MB).  One block per checkpoint then regenerates its stretch: the twist
// of the state is two phases of 156 independent updates (the second reads the first's results), tempering and the modulo are
// per element.
#include "common.hpp"

namespace sdslhip {

constexpr int kMtN = 312, kMtM = 156;

__global__ __launch_bounds__(320) void k_mt_draw(const uint64_t * __restrict__ ckpt, uint64_t stride, uint64_t count, uint64_t mod,
                                                 uint64_t add, uint64_t * __restrict__ out)
{
    constexpr uint64_t A = UINT64_C(0xB5026F5AA96619E9), UM = UINT64_C(0xFFFFFFFF80000000), LM = UINT64_C(0x7FFFFFFF);
    __shared__ uint64_t mt[kMtN];
    const unsigned t = threadIdx.x;
    const uint64_t seg = blockIdx.x;
    uint64_t pos = seg * stride;
    const uint64_t end = count < pos + stride ? count : pos + stride;
    if (t < kMtN)
        mt[t] = ckpt[seg * 313 + t];
    uint64_t mti = ckpt[seg * 313 + kMtN];
    __syncthreads();
    while (pos < end)
    {
        if (mti >= kMtN)
        { // twist: phase 1 = elements [0, 156), phase 2 = [156, 312) (they read phase 1's results)
            uint64_t v = 0;
            if (t < kMtN - kMtM)
            {
                const uint64_t x = (mt[t] & UM) | (mt[t + 1] & LM);
                v = mt[t + kMtM] ^ (x >> 1) ^ ((x & 1) ? A : 0);
            }
            uint64_t old_t = 0, old_t1 = 0;
            if (t >= kMtN - kMtM && t < kMtN)
            {
                old_t = mt[t];
                old_t1 = t + 1 < kMtN ? mt[t + 1] : 0;
            }
            __syncthreads();
            if (t < kMtN - kMtM)
                mt[t] = v;
            __syncthreads();
            if (t >= kMtN - kMtM && t < kMtN)
            {
                const uint64_t nxt = t + 1 < kMtN ? old_t1 : mt[0]; // (the last element pairs with the NEW first one)
                const uint64_t x = (old_t & UM) | (nxt & LM);
                v = mt[t - (kMtN - kMtM)] ^ (x >> 1) ^ ((x & 1) ? A : 0);
            }
            __syncthreads();
            if (t >= kMtN - kMtM && t < kMtN)
                mt[t] = v;
            __syncthreads();
            mti = 0;
        }
        const uint64_t avail = kMtN - mti, left = end - pos;
        const uint64_t n = avail < left ? avail : left;
        if (t < n)
        {
            uint64_t x = mt[mti + t];
            x ^= (x >> 29) & UINT64_C(0x5555555555555555);
            x ^= (x << 17) & UINT64_C(0x71D67FFFEDA60000);
            x ^= (x << 37) & UINT64_C(0xFFF7EEE000000000);
            x ^= (x >> 43);
            out[pos + t] = mod ? add + x % mod : x + add;
        }
        pos += n;
        mti += n;
        __syncthreads();
    }
}

} // namespace sdslhip

using namespace sdslhip;

extern "C" sdsl_hip_status sdsl_hip_util_rnd_positions_device(const uint64_t * checkpoints, uint64_t n_checkpoints, uint64_t stride,
                                                              uint64_t count, uint64_t mod, uint64_t add, uint64_t * d_out,
                                                              int32_t device, void * stream)
{
    if (count == 0)
        return SDSL_HIP_OK;
    if (!checkpoints || !d_out || stride == 0 || n_checkpoints < (count + stride - 1) / stride || !is_device_ptr(d_out))
    {
        set_error("rnd_positions_device: needs a device array and a checkpoint for every `stride` draws");
        return SDSL_HIP_ERR_INVALID;
    }
    SH_TRY(check_device(device));
    const uint64_t segs = (count + stride - 1) / stride;
    if (segs >= (UINT64_C(1) << 31))
    {
        set_error("rnd_positions_device: stride too small for %llu draws", (unsigned long long)count);
        return SDSL_HIP_ERR_INVALID;
    }
    DevBuf d_ck;
    SH_TRY(d_ck.alloc(segs * 313 * 8));
    hipStream_t s = (hipStream_t)stream;
    SH_HIP(hipMemcpyAsync(d_ck.p, checkpoints, segs * 313 * 8, hipMemcpyHostToDevice, s));
    hipLaunchKernelGGL(k_mt_draw, dim3((unsigned)segs), dim3(320), 0, s, d_ck.as<uint64_t>(), stride, count, mod, add, d_out);
    SH_HIP(hipGetLastError());
    SH_HIP(hipStreamSynchronize(s)); // (the checkpoint buffer is freed on return)
    return SDSL_HIP_OK;
}
