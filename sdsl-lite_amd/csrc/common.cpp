// common.cpp — error state, device checks, staging buffers, timing hooks, library-level C ABI.
#include "common.hpp"
#include "bv_host.hpp"

#include <atomic>
#include <cstdarg>

namespace sdslhip {

static thread_local std::string g_err;
static bool g_timing = false;
// batched rank on a plain vector: -1 automatic, 0 always the direct kernel, 1 the bucketed path whenever it applies
std::atomic<int> g_trace_phases{0};
std::atomic<int64_t> g_group_timeout_ms{getenv("SDSL_HIP_GROUP_TIMEOUT_MS") ? atoll(getenv("SDSL_HIP_GROUP_TIMEOUT_MS")) : 120000}; // group.cpp: deadline of a batch
std::atomic<int> g_group_test_stall{-1}; // group.cpp: test hook (a member whose scatter stream never gets going)
std::atomic<int> g_rrr_sparse_limit{20}; // largest class rrr_vector<63> may keep enumerative (rrr.hip: choose_sparse_max); up to 20
std::atomic<int> g_rrr_raw_budget{20}; // permille of the compressed size rrr_vector<63> may spend on raw classes (rrr.hip)
std::atomic<int> g_select_sorted_mode{getenv("SDSL_HIP_SELECT_SORTED") ? atoi(getenv("SDSL_HIP_SELECT_SORTED")) : -1};
std::atomic<int> g_wt_select_sorted_mode{getenv("SDSL_HIP_WT_SELECT_SORTED") ? atoi(getenv("SDSL_HIP_WT_SELECT_SORTED")) : -1};
std::atomic<int> g_rank_sorted_mode{getenv("SDSL_HIP_RANK_SORTED") ? atoi(getenv("SDSL_HIP_RANK_SORTED")) : -1};
std::atomic<int> g_rrr_sorted_mode{getenv("SDSL_HIP_RRR_SORTED") ? atoi(getenv("SDSL_HIP_RRR_SORTED")) : -1};
std::atomic<int> g_rrr_format{getenv("SDSL_HIP_RRR_FORMAT") ? atoi(getenv("SDSL_HIP_RRR_FORMAT")) : -1}; // record format of new rrr vectors
static thread_local bool g_timing_suppressed = false; // pipeline workers: the event pair is global
static hipEvent_t g_ev_start = nullptr, g_ev_stop = nullptr;
static bool g_ev_valid = false;

const char * last_error_message()
{
    return g_err.c_str();
}

void set_error(const char * fmt, ...)
{
    char buf[1024];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof buf, fmt, ap);
    va_end(ap);
    g_err = buf;
}

sdsl_hip_status hip_fail(hipError_t e, const char * what, const char * file, int line)
{
    set_error("HIP error %d (%s) in `%s` at %s:%d", (int)e, hipGetErrorString(e), what, file, line);
    return e == hipErrorOutOfMemory ? SDSL_HIP_ERR_NOMEM : SDSL_HIP_ERR_HIP;
}

bool is_device_ptr(const void * p)
{
    if (p == nullptr)
        return false;
    hipPointerAttribute_t a;
    hipError_t e = hipPointerGetAttributes(&a, p);
    if (e != hipSuccess)
    {
        (void)hipGetLastError(); // plain malloc'ed memory: not an error for us
        return false;
    }
    return a.type == hipMemoryTypeDevice || a.type == hipMemoryTypeManaged;
}

sdsl_hip_status mailbox_for(int device, Mailbox ** out)
{
    // one mailbox per device, made on first use.  Everything happens under the mutex (a scalar query costs microseconds of
    // launch and synchronisation anyway): no unsynchronised read of a half-published entry, and a failure half way frees what
    // it had pinned instead of leaking it on every retry
    static Mailbox boxes[64];
    static bool ready[64];
    static std::mutex init;
    if (device < 0 || device >= 64)
        return SDSL_HIP_ERR_INVALID;
    std::lock_guard<std::mutex> lock(init);
    Mailbox & b = boxes[device];
    if (!ready[device])
    {
        SH_HIP(hipSetDevice(device));
        void * h = nullptr;
        SH_HIP(hipHostMalloc(&h, 16 * sizeof(uint64_t), hipHostMallocMapped));
        void * d = nullptr;
        hipStream_t st = nullptr;
        if (hipHostGetDevicePointer(&d, h, 0) != hipSuccess || hipStreamCreateWithFlags(&st, hipStreamNonBlocking) != hipSuccess)
        {
            (void)hipHostFree(h);
            set_error("scalar query mailbox: %s", hipGetErrorString(hipGetLastError()));
            return SDSL_HIP_ERR_HIP;
        }
        b.stream = st;
        b.dev = (uint64_t *)d;
        b.host = (uint64_t *)h;
        ready[device] = true;
    }
    *out = &b;
    return SDSL_HIP_OK;
}

sdsl_hip_status check_device(int32_t device)
{
    int n = 0;
    hipError_t e = hipGetDeviceCount(&n);
    if (e != hipSuccess || n <= 0)
    {
        (void)hipGetLastError();
        set_error("no HIP device visible (hipGetDeviceCount: %s); this engine has no CPU path",
                  e == hipSuccess ? "0 devices" : hipGetErrorString(e));
        return SDSL_HIP_ERR_NO_DEVICE;
    }
    if (device < 0 || device >= n)
    {
        set_error("device index %d out of range [0,%d)", device, n);
        return SDSL_HIP_ERR_INVALID;
    }
    hipDeviceProp_t prop;
    SH_HIP(hipGetDeviceProperties(&prop, device));
    if (strncmp(prop.gcnArchName, "gfx950", 6) != 0)
    {
        set_error("device %d is %s; the kernels in this library are built for gfx950 only", device,
                  prop.gcnArchName);
        return SDSL_HIP_ERR_NO_DEVICE;
    }
    SH_HIP(hipSetDevice(device));
    return SDSL_HIP_OK;
}

static std::atomic<uint64_t> g_dev_bytes{0}; // device memory currently held through DevBuf (handles + calls in flight)

sdsl_hip_status DevBuf::alloc(size_t n, bool zero)
{
    release();
    if (n == 0)
        n = 16; // keep a valid pointer for empty structures
    void * q = nullptr;
    SH_HIP(hipMalloc(&q, n));
    p = q;
    bytes = n;
    g_dev_bytes += n;
    static const bool trace_alloc = getenv("SDSL_HIP_TRACE_ALLOC") != nullptr;
    if (trace_alloc && n >= (1u << 20))
        fprintf(stderr, "[sdsl_hip] alloc %p .. %p (%zu MiB)\n", q, (void *)((char *)q + n), n >> 20);
    // SDSL_HIP_POISON=<byte>: every allocation that is not asked to be zero starts filled with that byte — fresh device memory is
    // usually zero, which hides reads of memory nobody has written (tests/test_gpu_poison.py runs the large-batch paths this way)
    static const int poison = getenv("SDSL_HIP_POISON") ? atoi(getenv("SDSL_HIP_POISON")) & 0xFF : -1;
    if (zero)
        SH_HIP(hipMemset(p, 0, n));
    else if (poison >= 0)
        SH_HIP(hipMemset(p, poison, n));
    return SDSL_HIP_OK;
}

void DevBuf::release()
{
    if (p)
    {
        static const bool trace_alloc = getenv("SDSL_HIP_TRACE_ALLOC") != nullptr;
        if (trace_alloc && bytes >= (1u << 20))
            fprintf(stderr, "[sdsl_hip] free  %p (%zu MiB)\n", p, bytes >> 20);
        (void)hipFree(p);
        g_dev_bytes -= bytes;
    }
    p = nullptr;
    bytes = 0;
}

sdsl_hip_status Staged::in(const void * ptr, size_t nbytes, hipStream_t s)
{
    bytes = nbytes;
    if (nbytes == 0 || is_device_ptr(ptr))
    {
        dev = const_cast<void *>(ptr);
        host = nullptr;
        return SDSL_HIP_OK;
    }
    SH_TRY(tmp.alloc(nbytes));
    SH_HIP(hipMemcpyAsync(tmp.p, ptr, nbytes, hipMemcpyHostToDevice, s));
    dev = tmp.p;
    host = const_cast<void *>(ptr);
    return SDSL_HIP_OK;
}

sdsl_hip_status Staged::out(void * ptr, size_t nbytes)
{
    bytes = nbytes;
    if (nbytes == 0 || is_device_ptr(ptr))
    {
        dev = ptr;
        host = nullptr;
        return SDSL_HIP_OK;
    }
    SH_TRY(tmp.alloc(nbytes));
    dev = tmp.p;
    host = ptr;
    return SDSL_HIP_OK;
}

sdsl_hip_status Staged::finish(hipStream_t s)
{
    if (host && bytes)
    {
        SH_HIP(hipMemcpyAsync(host, dev, bytes, hipMemcpyDeviceToHost, s));
        SH_HIP(hipStreamSynchronize(s));
    }
    return SDSL_HIP_OK;
}

void suppress_timing_in_this_thread()
{
    g_timing_suppressed = true;
}

bool set_timing_suppressed(bool v)
{
    const bool prev = g_timing_suppressed;
    g_timing_suppressed = v;
    return prev;
}

KernelTimer::KernelTimer(hipStream_t stream) : s(stream), on(g_timing && !g_timing_suppressed)
{
    if (!on)
        return;
    if (!g_ev_start)
    {
        if (hipEventCreate(&g_ev_start) != hipSuccess || hipEventCreate(&g_ev_stop) != hipSuccess)
        {
            on = false;
            return;
        }
    }
    (void)hipEventRecord(g_ev_start, s);
}

KernelTimer::~KernelTimer()
{
    if (!on)
        return;
    (void)hipEventRecord(g_ev_stop, s);
    g_ev_valid = true;
}

uint64_t next_handle_uid()
{
    static std::atomic<uint64_t> n{0};
    return ++n;
}

} // namespace sdslhip

using namespace sdslhip;

namespace sdslhip {
DeviceScratch & device_scratch(int device)
{
    static std::mutex mk;
    static DeviceScratch * pools[64] = {};
    const int d = device < 0 ? 0 : (device > 63 ? 63 : device);
    std::lock_guard<std::mutex> lock(mk);
    if (!pools[d])
        pools[d] = new DeviceScratch();
    return *pools[d];
}
bool stream_is_capturing(hipStream_t s)
{
    hipStreamCaptureStatus cs = hipStreamCaptureStatusNone;
    if (hipStreamIsCapturing(s, &cs) != hipSuccess)
    {
        (void)hipGetLastError();
        return false;
    }
    return cs != hipStreamCaptureStatusNone;
}

sdsl_hip_status ScratchLease::acquire(int device, DevBuf & capture_buf, size_t need, hipStream_t s)
{
    hipStreamCaptureStatus cs = hipStreamCaptureStatusNone;
    if (hipStreamIsCapturing(s, &cs) != hipSuccess)
    {
        (void)hipGetLastError(); // (the legacy stream of a thread whose other streams are capturing: treated as not capturing)
        cs = hipStreamCaptureStatusNone;
    }
    if (cs != hipStreamCaptureStatusNone)
    {
        capturing = true;
        if (cs == hipStreamCaptureStatusActive && capture_buf.p && capture_buf.bytes >= need)
        {
            p = capture_buf.p;
            bytes = capture_buf.bytes;
        }
        return SDSL_HIP_OK;
    }
    DeviceScratch & P = device_scratch(device);
    std::unique_lock<std::mutex> l(P.m);
    if (P.ev)
        SH_HIP(hipStreamWaitEvent(s, P.ev, 0));
    if (P.buf.bytes < need)
    {
        if (P.ev)
            SH_HIP(hipEventSynchronize(P.ev)); // the old buffer may still be in use
        P.buf.release();
        if (P.buf.alloc(need) != SDSL_HIP_OK)
            return SDSL_HIP_OK; // no room: direct kernel
    }
    if (!P.ev)
        SH_HIP(hipEventCreateWithFlags(&P.ev, hipEventDisableTiming));
    p = P.buf.p;
    bytes = P.buf.bytes;
    pool = &P;
    stream = s;
    lock = std::move(l);
    return SDSL_HIP_OK;
}

ScratchLease::~ScratchLease()
{
    if (pool && pool->ev)
        (void)hipEventRecord(pool->ev, stream); // whatever was enqueued against the pool lies in front of this
}

void device_scratch_quiesce(int device)
{
    DeviceScratch & P = device_scratch(device);
    std::lock_guard<std::mutex> lock(P.m);
    if (P.ev)
        (void)hipEventSynchronize(P.ev);
}
sdsl_hip_status device_scratch_release(int device)
{
    DeviceScratch & P = device_scratch(device);
    std::lock_guard<std::mutex> lock(P.m);
    SH_HIP(hipSetDevice(device));
    if (P.ev)
    {
        SH_HIP(hipEventSynchronize(P.ev));
        SH_HIP(hipEventDestroy(P.ev));
        P.ev = nullptr;
    }
    P.buf.release();
    return SDSL_HIP_OK;
}
} // namespace sdslhip

extern "C" {

const char * sdsl_hip_last_error(void)
{
    return g_err.c_str();
}

const char * sdsl_hip_version(void)
{
    return "sdsl_hip 0.1 (gfx950)";
}

int32_t sdsl_hip_device_count(void)
{
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess)
    {
        (void)hipGetLastError();
        return 0;
    }
    int ok = 0;
    for (int d = 0; d < n; ++d)
    {
        hipDeviceProp_t prop;
        if (hipGetDeviceProperties(&prop, d) == hipSuccess && strncmp(prop.gcnArchName, "gfx950", 6) == 0)
            ++ok;
    }
    return ok;
}

// The size gates (limits.hpp), by name: what INTEGRATION.md 5 tabulates and tests/test_size_limits.py compares the table with.
// 0 = no such name.
uint64_t sdsl_hip_limit(const char * what)
{
    using namespace sdslhip;
    if (!what)
        return 0;
    static const struct
    {
        const char * name;
        uint64_t value;
    } table[] = {
        {"bv_bits", kLimBvBits},
        {"bv_bucketed_bits", (UINT64_C(1) << kLimBvBucketedLinesLog) * 448},
        {"rrr_bits", kLimRrrBits},
        {"rrr_bucketed_bits", kLimRrrBucketedRecords * 34 * 63},
        {"wt_fused_symbols", kLimWtFusedSymbols},
        {"wt_select_bucketed_symbols", kLimWtSelectBucketedSymbols},
        {"fm_fast_symbols", kLimFmFastSymbols},
        {"sorter32_symbols", kLimSorter32Symbols},
        {"sorter64_symbols", kLimSorter64Symbols},
        {"step_table_lines", UINT64_C(1) << kLimStepTableLineBits},
    };
    for (const auto & e : table)
        if (!strcmp(what, e.name))
            return e.value;
    return 0;
}

sdsl_hip_status sdsl_hip_set_option(const char * name, int64_t value)
{
    if (name && !strcmp(name, "rank_sorted"))
    {
        sdslhip::g_rank_sorted_mode.store((int)value);
        return SDSL_HIP_OK;
    }
    if (name && !strcmp(name, "select_sorted"))
    {
        sdslhip::g_select_sorted_mode.store((int)value);
        return SDSL_HIP_OK;
    }
    if (name && !strcmp(name, "wt_select_sorted"))
    { // wavelet-tree select of large batches through the bucketed passes (wt_sorted.hip): 0 never, 1 from 4096 queries on, -1 automatic
        sdslhip::g_wt_select_sorted_mode.store((int)value);
        return SDSL_HIP_OK;
    }
    if (name && !strcmp(name, "rrr_sorted"))
    {
        sdslhip::g_rrr_sorted_mode.store((int)value);
        return SDSL_HIP_OK;
    }
    if (name && !strcmp(name, "rrr_format"))
    { // record format of the stand-alone rrr_vector<63> handles created from now on: 0 wide, 1 slim, -1 chosen per vector
        if (value < -1 || value > 1)
        {
            set_error("set_option: rrr_format is -1 (automatic), 0 (wide records) or 1 (slim records)");
            return SDSL_HIP_ERR_INVALID;
        }
        sdslhip::g_rrr_format.store((int)value);
        return SDSL_HIP_OK;
    }
    if (name && !strcmp(name, "rrr_raw_budget"))
    {
        if (value < 0 || value > 1000)
        {
            set_error("set_option: rrr_raw_budget is in permille of the compressed size (0..1000)");
            return SDSL_HIP_ERR_INVALID;
        }
        sdslhip::g_rrr_raw_budget.store((int)value);
        return SDSL_HIP_OK;
    }
    if (name && !strcmp(name, "rrr_sparse_limit"))
    { // the space / speed trade of rrr_vector<63> handles created from now on (rrr.hip: choose_sparse_max)
        if (value < 0 || value > 20)
        {
            set_error("set_option: rrr_sparse_limit is the largest class kept enumerative (0..20, default 20)");
            return SDSL_HIP_ERR_INVALID;
        }
        sdslhip::g_rrr_sparse_limit.store((int)value);
        return SDSL_HIP_OK;
    }
    if (name && !strcmp(name, "group_timeout_ms"))
    { // deadline of one device-group batch (group.cpp); 0 = wait for ever
        if (value < 0)
        {
            set_error("set_option: group_timeout_ms is a number of milliseconds (0 = no deadline)");
            return SDSL_HIP_ERR_INVALID;
        }
        sdslhip::g_group_timeout_ms.store(value);
        return SDSL_HIP_OK;
    }
    if (name && !strcmp(name, "group_test_stall"))
    { // TEST HOOK: member `value` of the next group batch gets a kernel on its scatter stream that spins until the option is set to -1
        sdslhip::group_test_stall_set((int)value);
        return SDSL_HIP_OK;
    }
    if (name && !strcmp(name, "trace_phases"))
    {
        sdslhip::g_trace_phases.store((int)value);
        sdslhip::bv_sorted_clear_phases(); // what sdsl_hip_last_phases reports from now on belongs to calls made after this one
        return SDSL_HIP_OK;
    }
    set_error("set_option: unknown option '%s'", name ? name : "(null)");
    return SDSL_HIP_ERR_INVALID;
}

sdsl_hip_status sdsl_hip_set_timing(int32_t enabled)
{
    g_timing = enabled != 0;
    g_ev_valid = false;
    return SDSL_HIP_OK;
}

uint64_t sdsl_hip_allocated_bytes(void)
{
    return g_dev_bytes.load();
}

sdsl_hip_status sdsl_hip_last_kernel_ms(float * ms_out)
{
    if (!ms_out)
        return SDSL_HIP_ERR_INVALID;
    if (!g_ev_valid)
    {
        set_error("no timed kernel launch recorded (call sdsl_hip_set_timing(1) first)");
        return SDSL_HIP_ERR_INVALID;
    }
    SH_HIP(hipEventSynchronize(g_ev_stop));
    SH_HIP(hipEventElapsedTime(ms_out, g_ev_start, g_ev_stop));
    return SDSL_HIP_OK;
}
}
