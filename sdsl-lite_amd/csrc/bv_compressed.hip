// bv_compressed.hip — sibling bit-vector representations handed over as their own serialised bytes and turned into plain
// bits ON THE DEVICE: bit_vector_il<t_bs> (bit_vector_il.hpp:120-163: data words interleaved with
// one cumulative count per block of t_bs bits) and rrr_vector<15> (the specialisation of rrr_vector_15.hpp and the generic
// rrr_vector<t_bs> for t_bs <= 63: classes + offsets in the combinatorial number system).  rank / select answers do not depend on the representation, so
// both are then served from rank lines (bv.hip).  Round 1 converted them on the host, one get_int per word.
#include "bv_host.hpp"
#include "sdsl_stream.hpp"

namespace sdslhip {

namespace {

// word i of the plain vector sits at data[i + i / wpb + 1]: every block of wpb words is preceded by its count
__global__ __launch_bounds__(256) void k_il_to_plain(const uint64_t * __restrict__ data, uint64_t n_words, uint32_t wpb_shift,
                                                     uint64_t * __restrict__ out)
{
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n_words; i += (uint64_t)gridDim.x * blockDim.x)
        out[i] = data[i + (i >> wpb_shift) + 1];
}

// rrr_vector<t_bs, int_vector<>, t_k> for t_bs <= 63 (rrr_vector.hpp:366-378: size, bt, btnr, btnrp, rank, invert) and the
// rrr_vector<15> specialisation of rrr_vector_15.hpp (:409-420: the same without invert).  Blocks of a class are numbered
// in lexicographic order of (bit 0, bit 1, ...), 0 < 1 (rrr_helper.hpp:346-366; rrr_vector_15.hpp:60-78); a flagged
// superblock stores t_bs - class (rrr_vector.hpp:203-228).
struct RrrGenTables
{
    uint64_t C[64][64]; // binomials
    uint8_t space[64];  // bits of an offset field: hi(C(t_bs, k)) + 1, 0 when the class has one member
};

struct RrrGenParams
{
    uint32_t bs, k, w; // block size, blocks per superblock, bits of a class
    bool has_invert;
    uint64_t n_blocks;
};

__device__ __forceinline__ unsigned rg_class(const RrrGenParams & P, const uint64_t * __restrict__ bt,
                                             const uint64_t * __restrict__ inv, uint64_t b)
{
    unsigned c = (unsigned)read_bits(bt, b * P.w, P.w);
    if (P.has_invert)
    {
        const uint64_t sb = b / P.k;
        if ((inv[sb >> 6] >> (sb & 63)) & 1)
            c = P.bs - c;
    }
    return c;
}

// A GROUP is 64 consecutive blocks (= bs words of plain bits, starting on a word boundary).  Both kernels give a group to a WAVE,
// one block per lane: the classes are read side by side, the offset widths are summed / prefix-summed across the wave, every
// lane decodes its own block, the bs words of the group are assembled in LDS and written out side by side.  (The first form
// walked the 64 blocks of a group in ONE thread, OR-ing into global memory: 64 dependent decodes per thread.)
__device__ __forceinline__ unsigned rg_wave_incl_scan(unsigned v)
{
    const unsigned lane = threadIdx.x & 63;
#pragma unroll
    for (unsigned d = 1; d < 64; d <<= 1)
    {
        const unsigned o = (unsigned)__shfl_up((int)v, d, 64);
        if (lane >= d)
            v += o;
    }
    return v;
}

// offset bits of every group
__global__ __launch_bounds__(256) void k_rg_group_len(RrrGenParams P, const uint64_t * __restrict__ bt,
                                                      const uint64_t * __restrict__ inv, uint64_t n_groups,
                                                      const RrrGenTables * __restrict__ T, uint32_t * __restrict__ glen)
{
    const unsigned lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    for (uint64_t g = (uint64_t)blockIdx.x * 4 + wv; g < n_groups; g += (uint64_t)gridDim.x * 4)
    {
        const uint64_t b = g * 64 + lane;
        unsigned len = 0;
        if (b < P.n_blocks)
        {
            const unsigned c = rg_class(P, bt, inv, b);
            len = c <= P.bs ? T->space[c] : 0;
        }
        const unsigned total = rg_wave_incl_scan(len);
        if (lane == 63)
            glen[g] = total;
    }
}

__global__ __launch_bounds__(256) void k_rg_decode(RrrGenParams P, const uint64_t * __restrict__ bt, const uint64_t * __restrict__ inv,
                                                   const uint64_t * __restrict__ btnr, uint64_t btnr_bits,
                                                   const uint64_t * __restrict__ gptr, uint64_t n_groups, uint64_t n_words,
                                                   const RrrGenTables * __restrict__ T, unsigned long long * __restrict__ out)
{
    __shared__ unsigned long long wbuf[4][64];
    const unsigned lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    unsigned long long * wb = wbuf[wv]; // (a wave's own words: LDS operations of one wave complete in order, no block barrier)
    for (uint64_t g = (uint64_t)blockIdx.x * 4 + wv; g < n_groups; g += (uint64_t)gridDim.x * 4)
    {
        wb[lane] = 0;
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        const uint64_t b = g * 64 + lane;
        unsigned k = 0, len = 0;
        if (b < P.n_blocks)
        {
            k = rg_class(P, bt, inv, b);
            if (k > P.bs)
                k = 0; // a malformed class: the block decodes to zeros
            len = T->space[k];
        }
        const uint64_t ptr = gptr[g] + (rg_wave_incl_scan(len) - len);
        uint64_t nr = len && ptr + len <= btnr_bits ? read_bits(btnr, ptr, len) : 0;
        uint64_t bits = 0;
        for (unsigned p = 0; p < P.bs && k; ++p)
        {
            const uint64_t c = T->C[P.bs - 1 - p][k]; // members that have a 0 at position p
            if (nr >= c)
            {
                nr -= c;
                --k;
                bits |= UINT64_C(1) << p;
            }
        }
        if (bits)
        {
            const unsigned at = lane * P.bs, wi = at >> 6, o = at & 63; // (lane 63 with bs = 63 ends exactly at word 62's last bit)
            atomicOr(&wb[wi], (unsigned long long)(bits << o));
            if (o + P.bs > 64)
                atomicOr(&wb[wi + 1], (unsigned long long)(bits >> (64 - o)));
        }
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        __builtin_amdgcn_wave_barrier();
        const uint64_t wi = g * P.bs + lane;
        if (lane < P.bs && wi < n_words)
            out[wi] = wb[lane];
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        __builtin_amdgcn_wave_barrier();
    }
}

sdsl_hip_status fmt(const char * what, const StreamReader & rd)
{
    set_error("malformed %s stream (offset %zu of %zu)", what, rd.pos, rd.len);
    return SDSL_HIP_ERR_FORMAT;
}

} // namespace

// plain words (device) of a serialised bit_vector_il / rrr_vector<15>
sdsl_hip_status compressed_stream_to_device_words(const void * bytes, size_t len, int kind, DevBuf & d_words, uint64_t & n_bits)
{
    StreamReader rd(bytes, len);
    if (kind == SDSL_HIP_SIBLING_IL)
    {
        uint64_t size = 0, block_num = 0, superblocks = 0, block_shift = 0;
        HostIntVec data;
        if (!rd.u64(size) || !rd.u64(block_num) || !rd.u64(superblocks) || !rd.u64(block_shift) || !rd.int_vector(data, 64))
            return fmt("bit_vector_il", rd);
        if (size >= (UINT64_C(1) << 40) || block_shift < 6 || block_shift > 30)
            return fmt("bit_vector_il", rd);
        const uint64_t nw = (size + 63) >> 6, wpb_shift = block_shift - 6;
        if (nw && data.size() < nw + ((nw - 1) >> wpb_shift) + 2)
            return fmt("bit_vector_il", rd);
        n_bits = size;
        DevBuf d_data;
        SH_TRY(d_data.alloc((data.size() + 1) * 8));
        SH_HIP(hipMemcpy(d_data.p, data.words.data(), data.size() * 8, hipMemcpyHostToDevice));
        SH_TRY(d_words.alloc((nw + 1) * 8, true));
        if (nw)
            hipLaunchKernelGGL(k_il_to_plain, dim3(grid_for(nw, 256, 65536)), dim3(256), 0, 0, d_data.as<uint64_t>(), nw,
                               (uint32_t)wpb_shift, d_words.as<uint64_t>());
        SH_HIP(hipGetLastError());
        SH_HIP(hipDeviceSynchronize());
        return SDSL_HIP_OK;
    }
    if (kind == SDSL_HIP_SIBLING_RRR15 || (kind & 0xFF) == 2)
    {
        RrrGenParams P;
        P.has_invert = kind != SDSL_HIP_SIBLING_RRR15;
        P.bs = P.has_invert ? (uint32_t)(kind >> 8) & 0xFF : 15u;
        P.k = P.has_invert ? (uint32_t)(kind >> 16) & 0xFFFF : 32u;
        const char * what = P.has_invert ? "rrr_vector<t_bs>" : "rrr_vector<15> (rrr_vector_15.hpp)";
        if (P.bs < 2 || P.bs > 63 || P.k == 0)
        {
            set_error("bv_create_from_sdsl: rrr block sizes 2..63 are decoded on the device (got %u, k = %u)", P.bs, P.k);
            return SDSL_HIP_ERR_UNSUPPORTED;
        }
        P.w = hi64(P.bs) + 1;
        uint64_t size = 0;
        HostIntVec bt, btnr, btnrp, rank, inv;
        if (!rd.u64(size) || !rd.int_vector(bt) || !rd.int_vector(btnr, 1) || !rd.int_vector(btnrp) || !rd.int_vector(rank)
            || (P.has_invert && !rd.int_vector(inv, 1)))
            return fmt(what, rd);
        if (size >= (UINT64_C(1) << 40) || bt.width != P.w || bt.size() != (size + P.bs) / P.bs
            || (P.has_invert && inv.size() != (bt.size() + P.k - 1) / P.k))
            return fmt(what, rd);
        P.n_blocks = (size + P.bs - 1) / P.bs; // the trailing (dummy) class of a vector whose size is a multiple of bs is unused
        std::vector<RrrGenTables> Th(1);
        RrrGenTables & T = Th[0];
        memset(&T, 0, sizeof(T));
        for (int m = 0; m < 64; ++m)
            for (int k = 0; k <= m; ++k)
                T.C[m][k] = k == 0 || k == m ? 1 : T.C[m - 1][k - 1] + T.C[m - 1][k];
        for (unsigned k = 0; k <= P.bs; ++k)
        {
            const uint64_t c = T.C[P.bs][k];
            T.space[k] = c == 1 ? 0 : (uint8_t)(hi64(c) + 1);
        }
        n_bits = size;
        const uint64_t nw = (size + 63) >> 6, n_groups = (P.n_blocks + 63) / 64;
        DevBuf d_bt, d_btnr, d_inv, d_T, d_glen, d_gptr;
        SH_TRY(d_bt.alloc((bt.words.size() + 1) * 8, true));
        SH_HIP(hipMemcpy(d_bt.p, bt.words.data(), bt.words.size() * 8, hipMemcpyHostToDevice));
        SH_TRY(d_btnr.alloc((btnr.words.size() + 2) * 8, true));
        SH_HIP(hipMemcpy(d_btnr.p, btnr.words.data(), btnr.words.size() * 8, hipMemcpyHostToDevice));
        SH_TRY(d_inv.alloc((inv.words.size() + 2) * 8, true));
        if (!inv.words.empty())
            SH_HIP(hipMemcpy(d_inv.p, inv.words.data(), inv.words.size() * 8, hipMemcpyHostToDevice));
        SH_TRY(d_T.alloc(sizeof(RrrGenTables)));
        SH_HIP(hipMemcpy(d_T.p, &T, sizeof(RrrGenTables), hipMemcpyHostToDevice));
        SH_TRY(d_words.alloc((nw + 1) * 8, true));
        if (n_groups)
        {
            SH_TRY(d_glen.alloc(n_groups * 4));
            SH_TRY(d_gptr.alloc(n_groups * 8));
            hipLaunchKernelGGL(k_rg_group_len, dim3(grid_for(n_groups * 64, 256, 65536)), dim3(256), 0, 0, P, d_bt.as<uint64_t>(),
                               d_inv.as<uint64_t>(), n_groups, d_T.as<RrrGenTables>(), d_glen.as<uint32_t>());
            SH_HIP(hipGetLastError());
            uint64_t total = 0;
            SH_TRY(device_exclusive_scan_u32(d_glen.as<uint32_t>(), n_groups, d_gptr.as<uint64_t>(), 1, &total));
            if (total > btnr.bit_size)
                return fmt(what, rd); // the classes ask for more offset bits than the stream holds
            hipLaunchKernelGGL(k_rg_decode, dim3(grid_for(n_groups * 64, 256, 65536)), dim3(256), 0, 0, P, d_bt.as<uint64_t>(),
                               d_inv.as<uint64_t>(), d_btnr.as<uint64_t>(), btnr.bit_size, d_gptr.as<uint64_t>(), n_groups, nw,
                               d_T.as<RrrGenTables>(), d_words.as<unsigned long long>());
            SH_HIP(hipGetLastError());
        }
        SH_HIP(hipDeviceSynchronize());
        return SDSL_HIP_OK;
    }
    set_error("bv_create_from_sdsl: unknown kind %d", kind);
    return SDSL_HIP_ERR_INVALID;
}

} // namespace sdslhip
