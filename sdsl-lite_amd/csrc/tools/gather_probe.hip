// gather_probe.hip — calibration micro-benchmark behind DESIGN.md §3: how fast can gfx950 fetch
// RANDOM aligned chunks of 32 / 64 / 128 bytes from a multi-GiB table, as a function of how
// many lanes cooperate on one chunk and of the cache policy of the load?  Not part of the
// library; built by build.py into sdsl-lite_amd/lib/gather_probe and run by hand via gpurun.
#include <hip/hip_runtime.h>

#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

#define CK(x)                                                                                                      \
    do {                                                                                                           \
        hipError_t e = (x);                                                                                        \
        if (e != hipSuccess)                                                                                       \
        {                                                                                                          \
            fprintf(stderr, "HIP error %s at %s:%d\n", hipGetErrorString(e), __FILE__, __LINE__);                 \
            exit(1);                                                                                               \
        }                                                                                                          \
    } while (0)

typedef unsigned long long v2u64 __attribute__((ext_vector_type(2)));

__device__ __forceinline__ uint64_t mix(uint64_t x)
{ // splitmix64 finaliser: query positions are generated in-kernel so only the table is read
    x += 0x9E3779B97F4A7C15ull;
    x = (x ^ (x >> 30)) * 0xBF58476D1CE4E5B9ull;
    x = (x ^ (x >> 27)) * 0x94D049BB133111EBull;
    return x ^ (x >> 31);
}

// CHUNK bytes per query, LPC lanes per chunk (each lane loads CHUNK/LPC bytes as 16-B vectors),
// U queries in flight per lane group.
template <int CHUNK, int LPC, int U, bool NT, bool IDX_FROM_MEM>
__global__ __launch_bounds__(256) void k_gather(const v2u64 * __restrict__ table, uint64_t n_chunks,
                                                const uint64_t * __restrict__ idx, uint64_t * __restrict__ out,
                                                uint64_t n_q)
{
    constexpr int V = CHUNK / 16 / LPC; // 16-B vectors per lane
    const int s = threadIdx.x % LPC;
    const unsigned gq = threadIdx.x / LPC;
    constexpr unsigned QPB = 256 / LPC;
    for (uint64_t base = (uint64_t)blockIdx.x * QPB * U; base < n_q; base += (uint64_t)gridDim.x * QPB * U)
    {
        v2u64 v[U][V];
#pragma unroll
        for (int u = 0; u < U; ++u)
        {
            uint64_t q = base + (uint64_t)u * QPB + gq;
            uint64_t c;
            if (IDX_FROM_MEM)
                c = q < n_q ? idx[q] : 0;
            else
                c = mix(q) % n_chunks;
            const v2u64 * p = table + c * (CHUNK / 16) + s * V;
#pragma unroll
            for (int k = 0; k < V; ++k)
                v[u][k] = NT ? __builtin_nontemporal_load(p + k) : p[k];
        }
#pragma unroll
        for (int u = 0; u < U; ++u)
        {
            uint64_t q = base + (uint64_t)u * QPB + gq;
            uint64_t acc = 0;
#pragma unroll
            for (int k = 0; k < V; ++k)
                acc += __popcll(v[u][k].x) + __popcll(v[u][k].y);
            // reduce over the LPC lanes with shuffles (cost is irrelevant here)
            for (int m = 1; m < LPC; m <<= 1)
                acc += __shfl_xor(acc, m, 64);
            if (s == 0 && q < n_q)
                out[q] = acc;
        }
    }
}

__global__ void k_fill(uint64_t * p, uint64_t n)
{
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (uint64_t)gridDim.x * blockDim.x)
        p[i] = mix(i);
}
__global__ void k_fill_idx(uint64_t * p, uint64_t n, uint64_t mod)
{
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (uint64_t)gridDim.x * blockDim.x)
        p[i] = mix(i * 7 + 1) % mod;
}
__global__ void k_stream(const v2u64 * __restrict__ t, uint64_t n16, uint64_t * out)
{
    uint64_t acc = 0;
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n16; i += (uint64_t)gridDim.x * blockDim.x)
    {
        v2u64 v = t[i];
        acc += v.x ^ v.y;
    }
    if (acc == 0x1234567)
        out[0] = acc;
}

template <class F>
static float time_ms(F f, int reps)
{
    hipEvent_t a, b;
    CK(hipEventCreate(&a));
    CK(hipEventCreate(&b));
    f();
    CK(hipDeviceSynchronize());
    CK(hipEventRecord(a));
    for (int i = 0; i < reps; ++i)
        f();
    CK(hipEventRecord(b));
    CK(hipEventSynchronize(b));
    float ms;
    CK(hipEventElapsedTime(&ms, a, b));
    return ms / reps;
}

template <int CHUNK, int LPC, int U, bool NT, bool IDXM>
static void run(const char * name, const v2u64 * table, uint64_t table_bytes, const uint64_t * idx, uint64_t * out,
                uint64_t nq, unsigned blocks_per_cu)
{
    uint64_t n_chunks = table_bytes / CHUNK;
    unsigned grid = 256u * blocks_per_cu;
    float ms = time_ms(
        [&] {
            hipLaunchKernelGGL((k_gather<CHUNK, LPC, U, NT, IDXM>), dim3(grid), dim3(256), 0, 0, table, n_chunks, idx,
                               out, nq);
        },
        3);
    double gq = nq / (ms * 1e-3) / 1e9;
    printf("%-34s chunk=%3d lpc=%d U=%d nt=%d idxmem=%d bpc=%u : %8.3f ms  %7.2f Gq/s  chunk-GB/s=%8.1f\n", name, CHUNK,
           LPC, U, (int)NT, (int)IDXM, blocks_per_cu, ms, gq, gq * CHUNK);
    fflush(stdout);
}

// random 8-byte scatter: out[(mix(q) % n_slots)] = q   (the un-permute step of a bucketed batch)
template <bool NT>
__global__ __launch_bounds__(256) void k_scatter8(uint64_t * __restrict__ out, uint64_t n_slots, uint64_t n_q)
{
    for (uint64_t q = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; q < n_q; q += (uint64_t)gridDim.x * blockDim.x)
    {
        uint64_t p = mix(q) % n_slots;
        if (NT)
            __builtin_nontemporal_store(q, out + p);
        else
            out[p] = q;
    }
}
// scatter confined to a window of `win` slots that moves with q (results of one bucket land in one L2-sized tile)
__global__ __launch_bounds__(256) void k_scatter8_windowed(uint64_t * __restrict__ out, uint64_t n_slots, uint64_t win,
                                                          uint64_t n_q)
{
    for (uint64_t q = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; q < n_q; q += (uint64_t)gridDim.x * blockDim.x)
    {
        uint64_t base = (q / win) * win;
        uint64_t p = base + mix(q) % win;
        if (p < n_slots)
            out[p] = q;
    }
}

static int scatter(int argc, char ** argv)
{
    uint64_t nq = strtoull(argv[2], 0, 10);
    uint64_t * out;
    CK(hipMalloc(&out, nq * 8));
    CK(hipMemset(out, 0, nq * 8));
    float ms = time_ms([&] { hipLaunchKernelGGL(k_scatter8<false>, dim3(2048), dim3(256), 0, 0, out, nq, nq); }, 3);
    printf("random 8-B scatter over %.1f GiB            : %8.3f ms  %7.2f Gw/s\n", nq * 8 / 1073741824.0, ms, nq / ms / 1e6);
    ms = time_ms([&] { hipLaunchKernelGGL(k_scatter8<true>, dim3(2048), dim3(256), 0, 0, out, nq, nq); }, 3);
    printf("random 8-B scatter (nontemporal)           : %8.3f ms  %7.2f Gw/s\n", ms, nq / ms / 1e6);
    for (uint64_t win : {1ull << 17, 1ull << 19, 1ull << 21, 1ull << 23})
    {
        ms = time_ms([&] { hipLaunchKernelGGL(k_scatter8_windowed, dim3(2048), dim3(256), 0, 0, out, nq, win, nq); }, 3);
        printf("8-B scatter inside moving %6.1f MiB windows : %8.3f ms  %7.2f Gw/s\n", win * 8 / 1048576.0, ms, nq / ms / 1e6);
    }
    return 0;
}

static int sweep(int argc, char ** argv)
{ // gather_probe sweep <nq> <MiB> <MiB> ... : random-gather rate as a function of table size
    uint64_t nq = strtoull(argv[2], 0, 10);
    uint64_t max_bytes = 0;
    for (int i = 3; i < argc; ++i)
        if ((strtoull(argv[i], 0, 10) << 20) > max_bytes)
            max_bytes = strtoull(argv[i], 0, 10) << 20;
    v2u64 * table;
    uint64_t *idx, *out;
    const char * mode = getenv("PROBE_ALLOC");
    if (mode && !strcmp(mode, "uncached"))
        CK(hipExtMallocWithFlags((void **)&table, max_bytes, hipDeviceMallocUncached));
    else if (mode && !strcmp(mode, "finegrained"))
        CK(hipExtMallocWithFlags((void **)&table, max_bytes, hipDeviceMallocFinegrained));
    else
        CK(hipMalloc(&table, max_bytes));
    printf("alloc mode: %s\n", mode ? mode : "default");
    CK(hipMalloc(&idx, nq * 8));
    CK(hipMalloc(&out, nq * 8));
    hipLaunchKernelGGL(k_fill, dim3(4096), dim3(256), 0, 0, (uint64_t *)table, max_bytes / 8);
    CK(hipDeviceSynchronize());
    for (int i = 3; i < argc; ++i)
    {
        uint64_t tb = strtoull(argv[i], 0, 10) << 20;
        printf("--- table %llu MiB\n", (unsigned long long)(tb >> 20));
        run<64, 4, 4, false, false>("sweep", table, tb, idx, out, nq, 8);
        run<64, 4, 4, true, false>("sweep", table, tb, idx, out, nq, 8);
        run<128, 8, 4, false, false>("sweep", table, tb, idx, out, nq, 8);
        run<32, 2, 4, false, false>("sweep", table, tb, idx, out, nq, 8);
    }
    return 0;
}

static int pairs(int argc, char ** argv)
{ // gather_probe pairs <nq> <MiB>: is a second, ADJACENT line cheaper than a second random line?  And what does a
  // lane-individual 16-byte read of a random line cost (one query per lane layouts)?
    uint64_t nq = strtoull(argv[2], 0, 10);
    uint64_t tb = strtoull(argv[3], 0, 10) << 20;
    v2u64 * table;
    uint64_t *idx, *out;
    CK(hipMalloc(&table, tb));
    CK(hipMalloc(&idx, 8));
    CK(hipMalloc(&out, nq * 8));
    hipLaunchKernelGGL(k_fill, dim3(4096), dim3(256), 0, 0, (uint64_t *)table, tb / 8);
    CK(hipDeviceSynchronize());
    run<128, 8, 4, false, false>("one line, 8 lanes", table, tb, idx, out, nq, 8);
    run<256, 16, 4, false, false>("two adjacent lines, 16 lanes", table, tb, idx, out, nq, 8);
    run<256, 8, 4, false, false>("two adjacent lines, 8 lanes", table, tb, idx, out, nq, 8);
    run<512, 16, 2, false, false>("four adjacent lines, 16 lanes", table, tb, idx, out, nq, 8);
    run<16, 1, 4, false, false>("16 B of a random line per lane", table, tb, idx, out, nq, 8);
    run<32, 1, 4, false, false>("32 B of a random line per lane", table, tb, idx, out, nq, 8);
    run<64, 1, 4, false, false>("64 B of a random line per lane", table, tb, idx, out, nq, 8);
    return 0;
}

// The access skeleton of select with (almost) no arithmetic: argument (streamed) -> two adjacent 4-byte samples of a
// small directory (cache resident) -> ONE dependent, random, aligned 128-byte window of the big table fetched by a quad
// (two 64-byte halves) -> streamed result.  Its rate is the ceiling of that access mix.
template <int U, int SAMPLES>
__global__ __launch_bounds__(256) void k_chain(const v2u64 * __restrict__ table, uint64_t n_windows,
                                               const uint32_t * __restrict__ dir, uint64_t dir_n,
                                               const uint64_t * __restrict__ idx, uint64_t * __restrict__ out, uint64_t n_q)
{
    const int s = threadIdx.x & 3;
    const unsigned gq = threadIdx.x >> 2;
    constexpr unsigned QPB = 64;
    for (uint64_t base = (uint64_t)blockIdx.x * QPB * U; base < n_q; base += (uint64_t)gridDim.x * QPB * U)
    {
        v2u64 va[U], vb[U];
#pragma unroll
        for (int u = 0; u < U; ++u)
        {
            uint64_t q = base + (uint64_t)u * QPB + gq;
            uint64_t a = q < n_q ? __builtin_nontemporal_load(idx + q) : 0;
            uint64_t c = a;
            if (SAMPLES == 1)
            {
                uint64_t j = a % dir_n;
                c += dir[j] + dir[j + 1]; // zeros: the window index DEPENDS on the samples without changing
            }
            else if (SAMPLES == 2)
            { // both samples with ONE 8-byte load (4-byte aligned)
                uint64_t j = a % dir_n, two;
                __builtin_memcpy(&two, dir + j, 8);
                c += two;
            }
            c %= n_windows;
            const v2u64 * p = table + c * 8 + s;
            va[u] = p[0];
            vb[u] = p[4];
        }
#pragma unroll
        for (int u = 0; u < U; ++u)
        {
            uint64_t q = base + (uint64_t)u * QPB + gq;
            uint64_t acc = __popcll(va[u].x) + __popcll(va[u].y) + __popcll(vb[u].x) + __popcll(vb[u].y);
            acc += __shfl_xor(acc, 1, 64);
            acc += __shfl_xor(acc, 2, 64);
            if (s == 0 && q < n_q)
                __builtin_nontemporal_store(acc, out + q);
        }
    }
}

template <int U, int SAMPLES>
static void run_chain(const v2u64 * table, uint64_t tb, const uint32_t * dir, uint64_t dir_n, const uint64_t * idx,
                      uint64_t * out, uint64_t nq)
{
    float ms = time_ms(
        [&] {
            hipLaunchKernelGGL((k_chain<U, SAMPLES>), dim3(256 * 8), dim3(256), 0, 0, table, tb / 128, dir, dir_n, idx, out,
                               nq);
        },
        3);
    printf("chain U=%d samples=%d dir=%6.2f MiB : %8.3f ms  %7.2f Gq/s\n", U, (int)SAMPLES, dir_n * 4 / 1048576.0, ms,
           nq / ms / 1e6);
    fflush(stdout);
}

static int chain(int argc, char ** argv)
{ // gather_probe chain <nq> <table MiB>
    uint64_t nq = strtoull(argv[2], 0, 10);
    uint64_t tb = strtoull(argv[3], 0, 10) << 20;
    v2u64 * table;
    uint64_t *idx, *out;
    uint32_t * dir;
    const uint64_t dir_max = 1ull << 23;
    CK(hipMalloc(&table, tb));
    CK(hipMalloc(&idx, nq * 8));
    CK(hipMalloc(&out, nq * 8));
    CK(hipMalloc(&dir, (dir_max + 1) * 4));
    CK(hipMemset(dir, 0, (dir_max + 1) * 4));
    hipLaunchKernelGGL(k_fill, dim3(4096), dim3(256), 0, 0, (uint64_t *)table, tb / 8);
    hipLaunchKernelGGL(k_fill_idx, dim3(4096), dim3(256), 0, 0, idx, nq, ~0ull);
    CK(hipDeviceSynchronize());
    run_chain<1, 0>(table, tb, dir, 1, idx, out, nq);
    run_chain<2, 0>(table, tb, dir, 1, idx, out, nq);
    run_chain<4, 0>(table, tb, dir, 1, idx, out, nq);
    for (uint64_t dn : {1ull << 17, 1ull << 19, 1ull << 21, 1ull << 23})
    {
        run_chain<1, 1>(table, tb, dir, dn, idx, out, nq);
        run_chain<2, 1>(table, tb, dir, dn, idx, out, nq);
        run_chain<4, 1>(table, tb, dir, dn, idx, out, nq);
        run_chain<1, 2>(table, tb, dir, dn, idx, out, nq);
        run_chain<4, 2>(table, tb, dir, dn, idx, out, nq);
    }
    return 0;
}

int main(int argc, char ** argv)
{
    if (argc > 3 && !strcmp(argv[1], "pairs"))
        return pairs(argc, argv);
    if (argc > 3 && !strcmp(argv[1], "chain"))
        return chain(argc, argv);
    if (argc > 3 && !strcmp(argv[1], "sweep"))
        return sweep(argc, argv);
    if (argc > 2 && !strcmp(argv[1], "scatter"))
        return scatter(argc, argv);
    uint64_t table_bytes = (argc > 1 ? strtoull(argv[1], 0, 10) : 2304ull) << 20; // MiB
    uint64_t nq = argc > 2 ? strtoull(argv[2], 0, 10) : (1ull << 28);
    hipDeviceProp_t prop;
    CK(hipGetDeviceProperties(&prop, 0));
    printf("device: %s  CUs=%d  clock=%d MHz  L2=%d KiB\n", prop.gcnArchName, prop.multiProcessorCount,
           prop.clockRate / 1000, prop.l2CacheSize / 1024);
    v2u64 * table;
    uint64_t *idx, *out;
    CK(hipMalloc(&table, table_bytes));
    CK(hipMalloc(&idx, nq * 8));
    CK(hipMalloc(&out, nq * 8));
    hipLaunchKernelGGL(k_fill, dim3(4096), dim3(256), 0, 0, (uint64_t *)table, table_bytes / 8);
    CK(hipDeviceSynchronize());
    printf("table = %.2f GiB, queries = %llu\n", table_bytes / 1073741824.0, (unsigned long long)nq);

    float ms = time_ms([&] { hipLaunchKernelGGL(k_stream, dim3(2048), dim3(256), 0, 0, table, table_bytes / 16, out); }, 3);
    printf("streaming read of the table: %.3f ms = %.1f GB/s\n", ms, table_bytes / (ms * 1e-3) / 1e9);

    // ---- in-kernel generated positions: isolates the gather itself -------------------------
    run<32, 2, 4, false, false>("gen idx", table, table_bytes, idx, out, nq, 8);
    run<32, 2, 4, true, false>("gen idx", table, table_bytes, idx, out, nq, 8);
    run<32, 1, 4, true, false>("gen idx", table, table_bytes, idx, out, nq, 8);
    run<64, 4, 4, false, false>("gen idx", table, table_bytes, idx, out, nq, 8);
    run<64, 4, 4, true, false>("gen idx", table, table_bytes, idx, out, nq, 8);
    run<64, 4, 2, true, false>("gen idx", table, table_bytes, idx, out, nq, 8);
    run<64, 4, 8, true, false>("gen idx", table, table_bytes, idx, out, nq, 8);
    run<64, 2, 4, true, false>("gen idx", table, table_bytes, idx, out, nq, 8);
    run<64, 1, 4, true, false>("gen idx", table, table_bytes, idx, out, nq, 8);
    run<64, 1, 2, true, false>("gen idx", table, table_bytes, idx, out, nq, 8);
    run<128, 8, 4, false, false>("gen idx", table, table_bytes, idx, out, nq, 8);
    run<128, 8, 4, true, false>("gen idx", table, table_bytes, idx, out, nq, 8);
    run<128, 4, 4, true, false>("gen idx", table, table_bytes, idx, out, nq, 8);
    run<64, 4, 4, true, false>("gen idx bpc4", table, table_bytes, idx, out, nq, 4);
    run<64, 4, 4, true, false>("gen idx bpc16", table, table_bytes, idx, out, nq, 16);

    // ---- positions read from memory + results written: the real query I/O ------------------
    for (int chunk : {32, 64, 128})
    {
        hipLaunchKernelGGL(k_fill_idx, dim3(4096), dim3(256), 0, 0, idx, nq, table_bytes / chunk);
        CK(hipDeviceSynchronize());
        if (chunk == 32)
            run<32, 2, 4, true, true>("mem idx", table, table_bytes, idx, out, nq, 8);
        if (chunk == 64)
        {
            run<64, 4, 4, true, true>("mem idx", table, table_bytes, idx, out, nq, 8);
            run<64, 4, 4, false, true>("mem idx", table, table_bytes, idx, out, nq, 8);
            run<64, 1, 4, true, true>("mem idx", table, table_bytes, idx, out, nq, 8);
        }
        if (chunk == 128)
            run<128, 8, 4, true, true>("mem idx", table, table_bytes, idx, out, nq, 8);
    }
    return 0;
}
