// common.hpp — shared host-side plumbing of libsdsl_hip (error state, pointer classification,
// device buffers, staging of host-resident batches, kernel timing).  gfx950 only.
#pragma once
#include "limits.hpp"
#include <algorithm>
#include <atomic>
#include <mutex>
#include <string>
#include <thread>
#include <vector>
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

extern std::atomic<int> g_rank_sorted_mode; // sdsl_hip_set_option("rank_sorted", ...)
extern std::atomic<int> g_rrr_sorted_mode;  // sdsl_hip_set_option("rrr_sorted", ...)
extern std::atomic<int> g_select_sorted_mode; // sdsl_hip_set_option("select_sorted", ...)
extern std::atomic<int> g_wt_select_sorted_mode; // sdsl_hip_set_option("wt_select_sorted", ...)
extern std::atomic<int64_t> g_group_timeout_ms; // sdsl_hip_set_option("group_timeout_ms", ...), SDSL_HIP_GROUP_TIMEOUT_MS
extern std::atomic<int> g_group_test_stall;
void group_test_stall_set(int member); // group.cpp
extern std::atomic<int> g_trace_phases;     // sdsl_hip_set_option("trace_phases", ...)
void bv_sorted_clear_phases();              // bv_sorted.hip
extern std::atomic<int> g_rrr_format;       // sdsl_hip_set_option("rrr_format", -1 | 0 | 1)
extern std::atomic<int> g_rrr_sparse_limit; // sdsl_hip_set_option("rrr_sparse_limit", 0..20)
extern std::atomic<int> g_rrr_raw_budget;   // sdsl_hip_set_option("rrr_raw_budget", permille)
const char * last_error_message();
void suppress_timing_in_this_thread(); // for good (pipeline worker threads)
bool set_timing_suppressed(bool v);     // returns the previous state
// helper launches inside a call must not overwrite the caller's sdsl_hip_last_kernel_ms
struct TimingPause
{
    bool prev;
    TimingPause() : prev(set_timing_suppressed(true))
    {}
    ~TimingPause()
    {
        set_timing_suppressed(prev);
    }
};
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
// `bytes` (a multiple of 4) at p = the 32-bit pattern `word`, as a KERNEL on `s`: the large-batch paths clear their counters this
// way and not with hipMemsetAsync, whose node in a captured graph did not stay ordered with the kernels around it (a replayed
// bucketed batch read counters that were being cleared and walked off its tables: tools/capture_probe.py tds)
sdsl_hip_status fill_u32_async(void * p, uint32_t word, size_t bytes, hipStream_t s);
// true while `s` is being captured into a graph (then nothing may be allocated, built or synchronised)
bool stream_is_capturing(hipStream_t s);
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
    // gives the memory up WITHOUT freeing it: hipFree waits for the device, and a device with work that will never finish (a device
    // group past its deadline, group.cpp) must not take the caller's thread with it
    void leak()
    {
        p = nullptr;
        bytes = 0;
    }
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

// One query at a time (the scalar operator() of the adaptors): a mapped, pinned mailbox per device — the kernel reads
// the argument from host memory and writes the answer back into it, so a call costs one launch and one stream
// synchronisation: no pointer classification, no staging allocation, no copies.  lock() the mailbox for the call.
struct Mailbox
{
    std::mutex m;
    uint64_t * host = nullptr; // 16 words: [0..8) arguments, [8..16) answers
    uint64_t * dev = nullptr;  // the same memory as the device sees it
    hipStream_t stream = nullptr;
};
sdsl_hip_status mailbox_for(int device, Mailbox ** out);

// kernel timing hook (sdsl_hip_set_timing / sdsl_hip_last_kernel_ms)
struct KernelTimer
{
    hipStream_t s;
    bool on;
    explicit KernelTimer(hipStream_t stream);
    ~KernelTimer();
};

// A batch whose argument and result arrays BOTH live in host memory (the shape an unmodified SDSL caller has:
// std::vector in, std::vector out) is cut into chunks that travel on several HIP streams, one host thread per stream:
// while one chunk is being uploaded, another one is in its kernel and a third one is on its way back, so the two
// directions of the PCIe link and the kernels overlap.  `launch(d_in, d_out, count, stream)` enqueues the kernel(s).
// Returns false if the batch is too small to be worth it (the caller then takes the single-shot path).
constexpr uint64_t kPipelineMinQueries = UINT64_C(1) << 23;
constexpr int kPipelineMaxStreams = 8;
// queries per chunk (default 2^22: 32 MiB up, 32 MiB down) and streams (default 2: more of them contend in the
// runtime's pageable staging — 2: 4.7, 3: 4.0, 4: 3.8, 8: 3.0 Gq/s for 10^8 ranks); the environment overrides are
// for profiling (SDSL_HIP_PIPE_CHUNK_LOG2, SDSL_HIP_PIPE_STREAMS)
inline uint64_t pipeline_chunk()
{
    const char * e = getenv("SDSL_HIP_PIPE_CHUNK_LOG2");
    int v = e ? atoi(e) : 22;
    return UINT64_C(1) << (v >= 16 && v <= 28 ? v : 22);
}
inline int pipeline_streams()
{
    const char * e = getenv("SDSL_HIP_PIPE_STREAMS");
    int v = e ? atoi(e) : 2;
    return v >= 1 && v <= kPipelineMaxStreams ? v : 2;
}

// n items of in_bytes (and, optionally, a second column of in2_bytes) each in host memory -> n items of out_bytes each
// in host memory, `chunk` items at a time
template <class Launch>
sdsl_hip_status host_pipeline_bytes2(int device, const uint8_t * h_in, size_t in_bytes, const uint8_t * h_in2, size_t in2_bytes,
                                     uint8_t * h_out, size_t out_bytes, uint64_t n, uint64_t chunk, Launch launch)
{
    std::vector<std::thread> workers;
    sdsl_hip_status status[kPipelineMaxStreams];
    std::string msg[kPipelineMaxStreams];
    const uint64_t kPipelineChunk = chunk;
    const int kPipelineStreams = pipeline_streams();
    const uint64_t n_chunks = (n + kPipelineChunk - 1) / kPipelineChunk;
    for (int t = 0; t < kPipelineStreams; ++t)
    {
        status[t] = SDSL_HIP_OK;
        workers.emplace_back(
            [&, t]
            {
                auto run = [&]() -> sdsl_hip_status
                {
                    suppress_timing_in_this_thread();
                    SH_HIP(hipSetDevice(device));
                    hipStream_t st = nullptr;
                    SH_HIP(hipStreamCreateWithFlags(&st, hipStreamNonBlocking));
                    DevBuf d_in, d_in2, d_out;
                    sdsl_hip_status r = d_in.alloc(kPipelineChunk * in_bytes);
                    if (r == SDSL_HIP_OK && h_in2)
                        r = d_in2.alloc(kPipelineChunk * in2_bytes);
                    if (r == SDSL_HIP_OK)
                        r = d_out.alloc(kPipelineChunk * out_bytes);
                    for (uint64_t c = (uint64_t)t; r == SDSL_HIP_OK && c < n_chunks; c += kPipelineStreams)
                    {
                        const uint64_t lo = c * kPipelineChunk, cnt = std::min(kPipelineChunk, n - lo);
                        hipError_t e = hipMemcpyAsync(d_in.p, h_in + lo * in_bytes, cnt * in_bytes, hipMemcpyHostToDevice, st);
                        if (e == hipSuccess && h_in2)
                            e = hipMemcpyAsync(d_in2.p, h_in2 + lo * in2_bytes, cnt * in2_bytes, hipMemcpyHostToDevice, st);
                        if (e == hipSuccess)
                        {
                            r = launch((const void *)d_in.p, (const void *)d_in2.p, (void *)d_out.p, cnt, st);
                            if (r != SDSL_HIP_OK)
                                break;
                            e = hipMemcpyAsync(h_out + lo * out_bytes, d_out.p, cnt * out_bytes, hipMemcpyDeviceToHost, st);
                        }
                        if (e == hipSuccess)
                            e = hipStreamSynchronize(st);
                        if (e != hipSuccess)
                            r = hip_fail(e, "host pipeline", __FILE__, __LINE__);
                    }
                    (void)hipStreamSynchronize(st);
                    (void)hipStreamDestroy(st);
                    return r;
                };
                status[t] = run();
                if (status[t] != SDSL_HIP_OK)
                    msg[t] = last_error_message(); // the error text is thread-local: carry it to the caller's thread
            });
    }
    for (auto & w : workers)
        w.join();
    for (int t = 0; t < kPipelineStreams; ++t)
        if (status[t] != SDSL_HIP_OK)
        {
            set_error("%s", msg[t].c_str());
            return status[t];
        }
    return SDSL_HIP_OK;
}

template <class Launch>
sdsl_hip_status host_pipeline_bytes(int device, const uint8_t * h_in, size_t in_bytes, uint8_t * h_out, size_t out_bytes,
                                    uint64_t n, uint64_t chunk, Launch launch)
{
    return host_pipeline_bytes2(device, h_in, in_bytes, nullptr, 0, h_out, out_bytes, n, chunk,
                                [&](const void * d_in, const void *, void * d_out, uint64_t cnt, hipStream_t st) -> sdsl_hip_status
                                { return launch(d_in, d_out, cnt, st); });
}

template <class Launch>
sdsl_hip_status host_pipeline_u64(int device, const uint64_t * h_in, uint64_t * h_out, uint64_t n, Launch launch)
{
    return host_pipeline_bytes(device, (const uint8_t *)h_in, 8, (uint8_t *)h_out, 8, n, pipeline_chunk(),
                               [&](const void * d_in, void * d_out, uint64_t cnt, hipStream_t st) -> sdsl_hip_status
                               { return launch((const uint64_t *)d_in, (uint64_t *)d_out, cnt, st); });
}

// runs the body of an extern "C" entry point that builds host-side containers from caller-supplied sizes
template <class F>
sdsl_hip_status guarded(const char * what, F body) noexcept
{
    try
    {
        return body();
    }
    catch (const std::bad_alloc &)
    {
        set_error("%s: out of host memory", what);
        return SDSL_HIP_ERR_NOMEM;
    }
    catch (const std::exception & e)
    {
        set_error("%s: %s", what, e.what());
        return SDSL_HIP_ERR_INVALID;
    }
    catch (...)
    {
        set_error("%s: unknown failure", what);
        return SDSL_HIP_ERR_INVALID;
    }
}

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
