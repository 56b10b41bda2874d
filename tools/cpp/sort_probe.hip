// (hipcc --offload-arch=gfx950 -O2 -std=c++17 tools/cpp/sort_probe.hip -o sdsl-lite_amd/lib/sort_probe)
// Does rocPRIM's radix_sort_pairs / inclusive_scan handle MORE than 2^32 items?  (sa.hip's 64-bit suffix sorter depends on it.)
// keys: a hash of the index truncated to `bits` bits; values: the index.  Checks: keys non-decreasing, equal keys keep their
// values in increasing order (stability), every value's key is its hash, scan total.
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <hip/hip_runtime.h>
#include <rocprim/device/device_radix_sort.hpp>
#include <rocprim/device/device_scan.hpp>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e), __FILE__, __LINE__); return 2; } } while (0)
__device__ __host__ inline uint64_t mix(uint64_t x) { x ^= x >> 33; x *= 0xff51afd7ed558ccdULL; x ^= x >> 33; x *= 0xc4ceb9fe1a85ec53ULL; x ^= x >> 33; return x; }
__global__ void fill(uint64_t * k, uint64_t * v, uint64_t n, uint64_t mask)
{
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (uint64_t)gridDim.x * blockDim.x) { k[i] = mix(i) & mask; v[i] = i; }
}
__global__ void check(const uint64_t * k, const uint64_t * v, uint64_t n, uint64_t mask, unsigned long long * bad, uint64_t * ones)
{
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (uint64_t)gridDim.x * blockDim.x)
    {
        bool b = (mix(v[i]) & mask) != k[i];
        if (i) b = b || k[i - 1] > k[i] || (k[i - 1] == k[i] && v[i - 1] >= v[i]);
        if (b) atomicAdd(bad, 1ull);
        ones[i] = 1;
    }
}
int main(int argc, char ** argv)
{
    const uint64_t n = argc > 1 ? strtoull(argv[1], nullptr, 0) : (1ull << 32) + 12345;
    const unsigned bits = argc > 2 ? atoi(argv[2]) : 34;
    uint64_t *k0, *k1, *v0, *v1; unsigned long long * bad; void * tmp;
    CK(hipMalloc(&k0, n * 8)); CK(hipMalloc(&k1, n * 8)); CK(hipMalloc(&v0, n * 8)); CK(hipMalloc(&v1, n * 8)); CK(hipMalloc(&bad, 8)); CK(hipMemset(bad, 0, 8));
    const uint64_t mask = bits >= 64 ? ~0ull : (1ull << bits) - 1;
    fill<<<4096, 256>>>(k0, v0, n, mask);
    size_t bytes = 0;
    CK(rocprim::radix_sort_pairs(nullptr, bytes, k0, k1, v0, v1, (size_t)n, 0u, bits));
    printf("n = %llu, %u key bits, temporary storage %.1f MB\n", (unsigned long long)n, bits, bytes / 1e6);
    CK(hipMalloc(&tmp, bytes ? bytes : 16));
    hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    CK(hipEventRecord(a));
    CK(rocprim::radix_sort_pairs(tmp, bytes, k0, k1, v0, v1, (size_t)n, 0u, bits));
    CK(hipEventRecord(b)); CK(hipDeviceSynchronize());
    float ms = 0; CK(hipEventElapsedTime(&ms, a, b));
    check<<<4096, 256>>>(k1, v1, n, mask, bad, k0);
    unsigned long long h = 0; CK(hipMemcpy(&h, bad, 8, hipMemcpyDeviceToHost));
    printf("sort: %.1f ms, violations: %llu\n", ms, h);
    size_t sb = 0;
    CK(rocprim::inclusive_scan(nullptr, sb, k0, v0, (size_t)n, rocprim::plus<uint64_t>()));
    void * t2; CK(hipMalloc(&t2, sb ? sb : 16));
    CK(hipEventRecord(a));
    CK(rocprim::inclusive_scan(t2, sb, k0, v0, (size_t)n, rocprim::plus<uint64_t>()));
    CK(hipEventRecord(b)); CK(hipDeviceSynchronize()); CK(hipEventElapsedTime(&ms, a, b));
    uint64_t last = 0, mid = 0; CK(hipMemcpy(&last, v0 + (n - 1), 8, hipMemcpyDeviceToHost)); CK(hipMemcpy(&mid, v0 + (1ull << 32), 8, hipMemcpyDeviceToHost));
    printf("scan: %.1f ms, last %llu (want %llu), at 2^32: %llu (want %llu)\n", ms, (unsigned long long)last, (unsigned long long)n, (unsigned long long)mid, (1ull << 32) + 1);
    return h == 0 && last == n && mid == (1ull << 32) + 1 ? 0 : 1;
}
