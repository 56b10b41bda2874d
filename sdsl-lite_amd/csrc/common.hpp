// common.hpp — shared host-side plumbing of libsdsl_hip (error state, pointer classification,
// device buffers, staging of host-resident batches, kernel timing).  gfx950 only.
#pragma once
#include <hip/hip_runtime.h>

#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <new>
#include <string>
#include <utility>
#include <vector>

#include "../../include/sdsl_hip.h"

namespace sdslhip {

void set_error(const char * fmt, ...);
sdsl_hip_status hip_fail(hipError_t e, const char * what, const char * file, int line);

#define SH_HIP(expr)                                                                                               \
    do {                                                                                                           \
        hipError_t _e = (expr);                                                                                    \
        if (_e != hipSuccess)                                                                                      \
            return ::sdslhip::hip_fail(_e, #expr, __FILE__, __LINE__);                                             \
    } while (0)

#define SH_TRY(expr)                                                                                               \
    do {                                                                                                           \
        sdsl_hip_status _s = (expr);                                                                               \
        if (_s != SDSL_HIP_OK)                                                                                     \
            return _s;                                                                                             \
    } while (0)

// true if p points into device (or managed) memory
bool is_device_ptr(const void * p);
sdsl_hip_status check_device(int32_t device); // validates index + gfx950, sets the device current

// RAII device allocation (hipMalloc/hipFree on a fixed device)
struct DevBuf
{
    void * p = nullptr;
    size_t bytes = 0;
    DevBuf() = default;
    DevBuf(const DevBuf &) = delete;
    DevBuf & operator=(const DevBuf &) = delete;
    DevBuf(DevBuf && o) noexcept : p(o.p), bytes(o.bytes)
    {
        o.p = nullptr;
        o.bytes = 0;
    }
    DevBuf & operator=(DevBuf && o) noexcept
    {
        if (this != &o)
        {
            release();
            p = o.p;
            bytes = o.bytes;
            o.p = nullptr;
            o.bytes = 0;
        }
        return *this;
    }
    ~DevBuf()
    {
        release();
    }
    sdsl_hip_status alloc(size_t n, bool zero = false);
    void release();
    template <class T>
    T * as() const
    {
        return reinterpret_cast<T *>(p);
    }
};

// A batch argument that may live on the host: gives a device view, uploads on construction
// (inputs) and downloads on finish() (outputs).
struct Staged
{
    void * dev = nullptr;   // device view used by the kernel
    void * host = nullptr;  // original host pointer (nullptr if the caller passed device memory)
    size_t bytes = 0;
    DevBuf tmp;
    sdsl_hip_status in(const void * p, size_t nbytes, hipStream_t s);   // read-only argument
    sdsl_hip_status out(void * p, size_t nbytes);                       // write-only argument
    sdsl_hip_status finish(hipStream_t s);                              // D2H for host outputs
};

// kernel timing hook (sdsl_hip_set_timing / sdsl_hip_last_kernel_ms)
struct KernelTimer
{
    hipStream_t s;
    bool on;
    explicit KernelTimer(hipStream_t stream);
    ~KernelTimer();
};

inline unsigned grid_for(uint64_t work_items, unsigned per_block, unsigned max_blocks = 1u << 30)
{
    uint64_t b = (work_items + per_block - 1) / per_block;
    if (b == 0)
        b = 1;
    if (b > max_blocks)
        b = max_blocks;
    return (unsigned)b;
}

} // namespace sdslhip
