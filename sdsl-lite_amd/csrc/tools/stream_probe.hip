// stream_probe.hip — calibration micro-benchmark behind DESIGN.md §3.5 (round 3): what does gfx950 do with the
// memory pattern of a radix-partition pass?  G blocks each own 256 output streams (one per bin) in a bin-major
// array; per tile step a block appends a run of ~32 keys (128 B) to every stream.  The probe issues exactly that
// pattern with nothing else in the way (no LDS sort, values from registers) in three flavours:
//   unaligned   runs start wherever the previous run of the stream stopped (what k_sr_partition did in round 2)
//   aligned A   the block holds back the tail that does not fill an A-key-aligned chunk and writes it with the next
//               run (software write combining): every store instruction covers whole aligned chunks of A keys
// and the same for the way back (reads of the runs).  Optionally a sequential 4-byte-per-key read and a 2-byte-per-key
// sequential write travel with every step, as in the real passes.  Not part of the library; run by hand via gpurun.
#include <hip/hip_runtime.h>

#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>

#define CK(x)                                                                                                      \
    do {                                                                                                           \
        hipError_t e = (x);                                                                                        \
        if (e != hipSuccess)                                                                                       \
        {                                                                                                          \
            fprintf(stderr, "HIP error %s at %s:%d\n", hipGetErrorString(e), __FILE__, __LINE__);                 \
            exit(1);                                                                                               \
        }                                                                                                          \
    } while (0)

__device__ __forceinline__ uint32_t mix32(uint32_t x)
{
    x ^= x >> 16;
    x *= 0x7feb352du;
    x ^= x >> 15;
    x *= 0x846ca68bu;
    x ^= x >> 16;
    return x;
}

constexpr unsigned kT = 512, kBins = 256, kPer = 16, kTile = kT * kPer;

// ALIGN: 0 = unaligned runs, else keys per aligned chunk (16 = 64 B, 32 = 128 B).  READ: the runs are read instead of written.
template <unsigned ALIGN, bool READ>
__global__ __launch_bounds__(kT) void k_streams(uint32_t * __restrict__ arr, uint64_t cap, unsigned steps, unsigned G,
                                                const uint32_t * __restrict__ seq_in, uint16_t * __restrict__ seq_out,
                                                uint64_t * __restrict__ sink, unsigned jitter)
{
    __shared__ unsigned cursor_lo[kBins], carry[kBins];
    const unsigned t = threadIdx.x, g = blockIdx.x, l = t & 15;
    for (unsigned b = t; b < kBins; b += kT)
    {
        cursor_lo[b] = mix32(b * 7919u + g) & 31u; // arbitrary start alignment inside the stream's region
        carry[b] = 0;
    }
    __syncthreads();
    uint32_t acc = 0;
    for (unsigned step = 0; step < steps; ++step)
    {
        uint32_t v = step;
        if (seq_in)
        {
            const uint32_t * p = seq_in + ((uint64_t)g * steps + step) * kTile;
            uint32_t k[kPer];
#pragma unroll
            for (unsigned u = 0; u < kPer; ++u)
                k[u] = __builtin_nontemporal_load(p + u * kT + t);
#pragma unroll
            for (unsigned u = 0; u < kPer; ++u)
                v ^= k[u];
        }
        if (seq_out)
        {
            uint16_t * p = seq_out + ((uint64_t)g * steps + step) * kTile;
#pragma unroll
            for (unsigned u = 0; u < kPer; ++u)
                __builtin_nontemporal_store((uint16_t)(v + u), p + u * kT + t);
        }
        // a 16-lane group owns bins grp, grp + 32, ...: the group's eight runs of a step are all requested before any is consumed
        constexpr unsigned kB = kBins / (kT / 16), kE = 3;
        unsigned cur[kB], n[kB];
#pragma unroll
        for (unsigned k = 0; k < kB; ++k)
        {
            const unsigned b = (t >> 4) + k * (kT / 16);
            // run lengths of a step sum to the tile: 32 +- jitter, deterministic
            const unsigned h = mix32((g * 1315423911u) ^ (step * 2654435761u) ^ b);
            const unsigned cnt = 32 - jitter + (jitter ? h % (2 * jitter + 1) : 0);
            cur[k] = cursor_lo[b];
            n[k] = cnt;
            if (ALIGN)
            {
                const unsigned avail = carry[b] + cnt, end = cur[k] + avail, aend = end & ~(ALIGN - 1);
                if (aend > cur[k])
                {
                    n[k] = aend - cur[k];
                    if (l == 0)
                    {
                        carry[b] = end - aend;
                        cursor_lo[b] = aend;
                    }
                }
                else
                {
                    n[k] = 0;
                    if (l == 0)
                        carry[b] = avail;
                }
            }
            else if (l == 0)
                cursor_lo[b] = cur[k] + cnt;
        }
        uint32_t r[kB][kE];
#pragma unroll
        for (unsigned k = 0; k < kB; ++k)
        {
            const unsigned b = (t >> 4) + k * (kT / 16);
            uint32_t * base = arr + ((uint64_t)b * G + g) * cap + cur[k];
#pragma unroll
            for (unsigned e = 0; e < kE; ++e)
            {
                const unsigned i = l + 16 * e;
                if (READ)
                    r[k][e] = i < n[k] ? base[i] : 0;
                else if (i < n[k])
                    base[i] = v + i;
            }
            for (unsigned i = l + 16 * kE; i < n[k]; i += 16)
            {
                if (READ)
                    acc += base[i];
                else
                    base[i] = v + i;
            }
        }
        if (READ)
        {
#pragma unroll
            for (unsigned k = 0; k < kB; ++k)
#pragma unroll
                for (unsigned e = 0; e < kE; ++e)
                    acc += r[k][e];
        }
        __syncthreads();
    }
    if (READ || seq_in)
        if (acc == 0x12345678u)
            sink[0] = acc;
}

template <unsigned ALIGN, bool READ>
static float run(uint32_t * arr, uint64_t cap, unsigned steps, unsigned G, const uint32_t * seq_in, uint16_t * seq_out,
                 uint64_t * sink, unsigned jitter)
{
    hipEvent_t a, b;
    CK(hipEventCreate(&a));
    CK(hipEventCreate(&b));
    float best = 1e30f;
    for (int rep = 0; rep < 3; ++rep)
    {
        CK(hipEventRecord(a));
        hipLaunchKernelGGL((k_streams<ALIGN, READ>), dim3(G), dim3(kT), 0, 0, arr, cap, steps, G, seq_in, seq_out, sink, jitter);
        CK(hipEventRecord(b));
        CK(hipEventSynchronize(b));
        float ms;
        CK(hipEventElapsedTime(&ms, a, b));
        if (ms < best)
            best = ms;
    }
    return best;
}

int main(int argc, char ** argv)
{
    const unsigned G = argc > 1 ? (unsigned)atoi(argv[1]) : 768;
    const unsigned steps = argc > 2 ? (unsigned)atoi(argv[2]) : 159;
    const unsigned jitter = argc > 3 ? (unsigned)atoi(argv[3]) : 12;
    const uint64_t cap = ((uint64_t)steps * (32 + jitter) + 64 + 31) & ~UINT64_C(31);
    const uint64_t n_arr = (uint64_t)kBins * G * cap, n_seq = (uint64_t)G * steps * kTile;
    uint32_t *arr, *seq_in;
    uint16_t * seq_out;
    uint64_t * sink;
    CK(hipMalloc(&arr, n_arr * 4 + 4096));
    CK(hipMalloc(&seq_in, n_seq * 4));
    CK(hipMalloc(&seq_out, n_seq * 2));
    CK(hipMalloc(&sink, 64));
    CK(hipMemset(arr, 1, n_arr * 4));
    CK(hipMemset(seq_in, 2, n_seq * 4));
    const double keys = (double)n_seq;
    printf("streams: G=%u blocks x 256 bins, %u steps, runs of 32 +- %u keys; %.2f G keys, array %.2f GiB\n", G, steps, jitter, keys / 1e9,
           n_arr * 4.0 / (1 << 30));
    auto line = [&](const char * name, float ms, double bytes_per_key)
    { printf("%-58s: %8.3f ms  %6.2f Gkeys/s  %6.2f TB/s algorithmic\n", name, ms, keys / ms / 1e6, keys * bytes_per_key / ms / 1e9); };
    for (int with_seq = 0; with_seq < 2; ++with_seq)
    {
        const uint32_t * si = with_seq ? seq_in : nullptr;
        uint16_t * so = with_seq ? seq_out : nullptr;
        const char * tag = with_seq ? " + 4 B seq read + 2 B seq write" : "";
        char nm[128];
        const double wb = 4 + (with_seq ? 6 : 0);
        snprintf(nm, sizeof nm, "write runs, unaligned%s", tag);
        line(nm, run<0, false>(arr, cap, steps, G, si, so, sink, jitter), wb);
        snprintf(nm, sizeof nm, "write runs, 64-B chunks (carry)%s", tag);
        line(nm, run<16, false>(arr, cap, steps, G, si, so, sink, jitter), wb);
        snprintf(nm, sizeof nm, "write runs, 128-B chunks (carry)%s", tag);
        line(nm, run<32, false>(arr, cap, steps, G, si, so, sink, jitter), wb);
        snprintf(nm, sizeof nm, "read runs, unaligned%s", tag);
        line(nm, run<0, true>(arr, cap, steps, G, si, so, sink, jitter), wb);
        snprintf(nm, sizeof nm, "read runs, 64-B chunks (carry)%s", tag);
        line(nm, run<16, true>(arr, cap, steps, G, si, so, sink, jitter), wb);
        snprintf(nm, sizeof nm, "read runs, 128-B chunks (carry)%s", tag);
        line(nm, run<32, true>(arr, cap, steps, G, si, so, sink, jitter), wb);
    }
    return 0;
}
