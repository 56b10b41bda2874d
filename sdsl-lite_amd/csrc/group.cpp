// group.cpp — one process, several MI355X: the multi-GPU split of SURVEY.md §8(e) behind the C ABI.
//
// The path shards by QUERY: the index is read-only and small next to 288 GB of HBM, so every device holds a replica
// (one ncclBroadcast per device buffer at load time) and a batch owned by the root device is cut into one contiguous
// shard per device.  Per batch the root SCATTERS the shards (ncclGroupStart + ncclSend x (G-1) on the root, ncclRecv on
// each peer: every peer on its own xGMI link, no ring), every device answers its shard with the single-GPU kernels, and
// the answers are GATHERED the same way.  The batch travels in `chunks` pieces over three streams per device (scatter,
// kernels, gather; two communicators so that the scatter of piece c+1 and the gather of piece c-1 overlap the kernels
// of piece c).  No collective exchanges index data during queries.
// The reference has no multi-device layer; the caller-visible contract is the batch call of the single-GPU ABI
// (sdsl_hip_bv_rank_batch etc.), same arguments, same answers.
// RCCL is loaded with dlopen at group creation, so a process that never creates a group does not pay for it.
#include <chrono>
#include <dlfcn.h>
#include <memory>
#include <string>
#include <thread>
#include <rccl/rccl.h>

#include "bv_host.hpp"

namespace sdslhip {

namespace {

struct Rccl
{
    void * so = nullptr;
    ncclResult_t (*CommInitAll)(ncclComm_t *, int, const int *) = nullptr;
    ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
    ncclResult_t (*GroupStart)() = nullptr;
    ncclResult_t (*GroupEnd)() = nullptr;
    ncclResult_t (*Send)(const void *, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*Recv)(void *, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*Broadcast)(const void *, void *, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
    const char * (*GetErrorString)(ncclResult_t) = nullptr;
};

const Rccl * rccl()
{
    static Rccl R;
    static bool tried = false;
    if (!tried)
    {
        tried = true;
        for (const char * name : {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"})
            if ((R.so = dlopen(name, RTLD_NOW | RTLD_GLOBAL)))
                break;
        if (R.so)
        {
            auto sym = [&](const char * s) { return dlsym(R.so, s); };
            R.CommInitAll = (decltype(R.CommInitAll))sym("ncclCommInitAll");
            R.CommDestroy = (decltype(R.CommDestroy))sym("ncclCommDestroy");
            R.GroupStart = (decltype(R.GroupStart))sym("ncclGroupStart");
            R.GroupEnd = (decltype(R.GroupEnd))sym("ncclGroupEnd");
            R.Send = (decltype(R.Send))sym("ncclSend");
            R.Recv = (decltype(R.Recv))sym("ncclRecv");
            R.Broadcast = (decltype(R.Broadcast))sym("ncclBroadcast");
            R.GetErrorString = (decltype(R.GetErrorString))sym("ncclGetErrorString");
            if (!R.CommInitAll || !R.CommDestroy || !R.GroupStart || !R.GroupEnd || !R.Send || !R.Recv || !R.Broadcast)
            {
                dlclose(R.so);
                R.so = nullptr;
            }
        }
    }
    return R.so ? &R : nullptr;
}

sdsl_hip_status rccl_fail(ncclResult_t r, const char * what)
{
    const Rccl * R = rccl();
    set_error("RCCL: %s failed: %s", what, R && R->GetErrorString ? R->GetErrorString(r) : "?");
    return SDSL_HIP_ERR_HIP;
}

#define SH_NCCL(expr)                                                                                              \
    do {                                                                                                           \
        ncclResult_t _r = (expr);                                                                                  \
        if (_r != ncclSuccess)                                                                                     \
            return rccl_fail(_r, #expr);                                                                           \
    } while (0)
// between ncclGroupStart and ncclGroupEnd: a failing call must not leave the group open on this thread (every later RCCL
// call would silently queue into it) — close it, then report the FIRST error
#define SH_NCCL_G(expr)                                                                                            \
    do {                                                                                                           \
        ncclResult_t _r = (expr);                                                                                  \
        if (_r != ncclSuccess)                                                                                     \
        {                                                                                                          \
            const sdsl_hip_status _st = rccl_fail(_r, #expr);                                                      \
            (void)R->GroupEnd();                                                                                   \
            return _st;                                                                                            \
        }                                                                                                          \
    } while (0)

} // namespace

} // namespace sdslhip

using namespace sdslhip;

struct sdsl_hip_group_s
{
    int n = 0;
    std::vector<int> dev;
    std::vector<ncclComm_t> comm_a, comm_b; // scatter / gather
    std::vector<hipStream_t> s_in, s_k, s_out;
    std::vector<DevBuf> in, out; // per device: its shard of the batch in flight (peers only)
    // Transport.  Default: RCCL (two communicators).  copy == true (SDSL_HIP_GROUP_TRANSPORT=copy at creation): every transfer is a
    // device-to-device hipMemcpyAsync on the RECEIVING member's stream, ordered behind the sender's stream by an event — for
    // processes that cannot load librccl, and the only transport that accepts the same device twice (a group of two on a
    // one-GPU box: the sharding, the chunk pipeline and the event chains of G > 1 then run where only one GPU is at hand).
    bool copy = false;
    bool poisoned = false; // a batch ran into its deadline: work that cannot be cancelled may still sit in the streams
    ~sdsl_hip_group_s()
    {
        if (poisoned)
        { // stuck work may still read the buffers, and CommDestroy / hipFree would wait for it: the group's device resources are leaked
            (void)new std::vector<DevBuf>(std::move(in));
            (void)new std::vector<DevBuf>(std::move(out));
            return;
        }
        const Rccl * R = rccl();
        for (int r = 0; r < n; ++r)
        {
            (void)hipSetDevice(dev[r]);
            if (R && r < (int)comm_a.size() && comm_a[r])
                (void)R->CommDestroy(comm_a[r]);
            if (R && r < (int)comm_b.size() && comm_b[r])
                (void)R->CommDestroy(comm_b[r]);
            for (auto * v : {&s_in, &s_k, &s_out})
                if (r < (int)v->size() && (*v)[r])
                    (void)hipStreamDestroy((*v)[r]);
            if (r < (int)in.size())
                in[r].release();
            if (r < (int)out.size())
                out[r].release();
        }
    }
};

namespace sdslhip {
// TEST HOOK (sdsl_hip_set_option("group_test_stall", member)): a peer that never gets to its part of the batch — in the copy
// transport "a receive that is never posted" is a scatter stream that never reaches its copy.  The kernel spins on a word of pinned
// host memory; setting the option to -1 releases it.
static uint32_t * g_stall_flag = nullptr; // pinned, mapped
__global__ void k_group_stall(volatile uint32_t * flag)
{
    while (*flag == 0)
        __builtin_amdgcn_s_sleep(127);
}
void group_test_stall_set(int member)
{
    if (member < 0)
    {
        if (g_stall_flag)
            __atomic_store_n(g_stall_flag, 1u, __ATOMIC_SEQ_CST);
        g_group_test_stall.store(-1);
        return;
    }
    if (!g_stall_flag && hipHostMalloc((void **)&g_stall_flag, 64, hipHostMallocMapped) != hipSuccess)
        return;
    __atomic_store_n(g_stall_flag, 0u, __ATOMIC_SEQ_CST);
    g_group_test_stall.store(member);
}
} // namespace sdslhip

namespace {

void shard(uint64_t n, int G, int r, uint64_t & lo, uint64_t & hi)
{
    lo = n * (uint64_t)r / (uint64_t)G;
    hi = n * (uint64_t)(r + 1) / (uint64_t)G;
}

// copy transport: `bytes` from member ra's memory to member rb's, behind everything queued on sa, carried out on sb
sdsl_hip_status copy_xfer(sdsl_hip_group_s * g, const void * src, int ra, hipStream_t sa, void * dst, int rb, hipStream_t sb, size_t bytes)
{
    if (!bytes)
        return SDSL_HIP_OK;
    hipEvent_t e;
    SH_HIP(hipSetDevice(g->dev[ra]));
    SH_HIP(hipEventCreateWithFlags(&e, hipEventDisableTiming));
    SH_HIP(hipEventRecord(e, sa));
    SH_HIP(hipSetDevice(g->dev[rb]));
    SH_HIP(hipStreamWaitEvent(sb, e, 0));
    SH_HIP(hipEventDestroy(e)); // (released once the wait has been satisfied)
    SH_HIP(hipMemcpyPeerAsync(dst, g->dev[rb], src, g->dev[ra], bytes, sb));
    return SDSL_HIP_OK;
}

// The root-owned batch: n items of in_bytes each (device memory of devices[0]) -> n items of out_bytes each.
// launch(r, d_in, d_out, count, stream) enqueues device r's kernels on `stream`.
template <class Launch>
sdsl_hip_status group_run(sdsl_hip_group_s * g, const uint8_t * d_in, size_t in_bytes, uint8_t * d_out, size_t out_bytes, uint64_t n,
                          int chunks, Launch launch)
{
    const Rccl * R = rccl();
    const int G = g->n;
    if (chunks < 1)
        chunks = 1;
    if (chunks > 64)
        chunks = 64;
    // staging on the peers: the whole shard (pieces land at their offsets)
    for (int r = 1; r < G; ++r)
    {
        uint64_t lo, hi;
        shard(n, G, r, lo, hi);
        SH_HIP(hipSetDevice(g->dev[r]));
        if (g->in[r].bytes < (hi - lo) * in_bytes)
            SH_TRY(g->in[r].alloc((hi - lo) * in_bytes));
        if (g->out[r].bytes < (hi - lo) * out_bytes)
            SH_TRY(g->out[r].alloc((hi - lo) * out_bytes));
    }
    std::vector<hipEvent_t> ev;
    auto new_event = [&](int device, hipEvent_t & e) -> sdsl_hip_status
    {
        SH_HIP(hipSetDevice(device));
        SH_HIP(hipEventCreateWithFlags(&e, hipEventDisableTiming));
        ev.push_back(e);
        return SDSL_HIP_OK;
    };
    auto body = [&]() -> sdsl_hip_status
    {
        for (int c = 0; c < chunks; ++c)
        {
            // piece c of every shard
            std::vector<uint64_t> plo(G), pcnt(G), slo(G);
            for (int r = 0; r < G; ++r)
            {
                uint64_t lo, hi, a, b;
                shard(n, G, r, lo, hi);
                shard(hi - lo, chunks, c, a, b);
                slo[r] = lo;
                plo[r] = a;
                pcnt[r] = b - a;
            }
            // scatter
            if (G > 1 && g->copy)
            {
                for (int r = 1; r < G; ++r)
                    SH_TRY(copy_xfer(g, d_in + (slo[r] + plo[r]) * in_bytes, 0, g->s_in[0], g->in[r].as<uint8_t>() + plo[r] * in_bytes, r, g->s_in[r],
                                     pcnt[r] * in_bytes));
            }
            else if (G > 1)
            {
                SH_NCCL(R->GroupStart());
                for (int r = 1; r < G; ++r)
                    if (pcnt[r])
                    {
                        SH_NCCL_G(R->Send(d_in + (slo[r] + plo[r]) * in_bytes, pcnt[r] * in_bytes, ncclUint8, r, g->comm_a[0], g->s_in[0]));
                        SH_NCCL_G(R->Recv(g->in[r].as<uint8_t>() + plo[r] * in_bytes, pcnt[r] * in_bytes, ncclUint8, 0, g->comm_a[r],
                                        g->s_in[r]));
                    }
                SH_NCCL(R->GroupEnd());
            }
            // kernels
            for (int r = 0; r < G; ++r)
            {
                if (!pcnt[r])
                    continue;
                SH_HIP(hipSetDevice(g->dev[r]));
                if (r > 0)
                {
                    hipEvent_t e;
                    SH_TRY(new_event(g->dev[r], e));
                    SH_HIP(hipEventRecord(e, g->s_in[r]));
                    SH_HIP(hipStreamWaitEvent(g->s_k[r], e, 0));
                }
                const uint8_t * pin = r == 0 ? d_in + (slo[0] + plo[0]) * in_bytes : g->in[r].as<uint8_t>() + plo[r] * in_bytes;
                uint8_t * pout = r == 0 ? d_out + (slo[0] + plo[0]) * out_bytes : g->out[r].as<uint8_t>() + plo[r] * out_bytes;
                SH_TRY(launch(r, pin, pout, pcnt[r], g->s_k[r]));
                if (r > 0)
                {
                    hipEvent_t e;
                    SH_TRY(new_event(g->dev[r], e));
                    SH_HIP(hipEventRecord(e, g->s_k[r]));
                    SH_HIP(hipStreamWaitEvent(g->s_out[r], e, 0));
                }
            }
            // gather
            if (G > 1 && g->copy)
            {
                for (int r = 1; r < G; ++r)
                    SH_TRY(copy_xfer(g, g->out[r].as<uint8_t>() + plo[r] * out_bytes, r, g->s_out[r], d_out + (slo[r] + plo[r]) * out_bytes, 0,
                                     g->s_out[r], pcnt[r] * out_bytes)); // (on the peer's stream: the call ends with every stream synchronised)
            }
            else if (G > 1)
            {
                SH_NCCL(R->GroupStart());
                for (int r = 1; r < G; ++r)
                    if (pcnt[r])
                    {
                        SH_NCCL_G(R->Send(g->out[r].as<uint8_t>() + plo[r] * out_bytes, pcnt[r] * out_bytes, ncclUint8, 0, g->comm_b[r],
                                        g->s_out[r]));
                        SH_NCCL_G(R->Recv(d_out + (slo[r] + plo[r]) * out_bytes, pcnt[r] * out_bytes, ncclUint8, r, g->comm_b[0],
                                        g->s_out[0]));
                    }
                SH_NCCL(R->GroupEnd());
            }
        }
        return SDSL_HIP_OK;
    };
    if (g->poisoned)
    {
        set_error("device group: an earlier batch ran into its deadline and may still hold the group's streams; destroy the group");
        return SDSL_HIP_ERR_HIP;
    }
    const int stall = g_group_test_stall.load();
    if (stall >= 0 && stall < G && g_stall_flag)
    { // (test hook: this member's scatter stream is held)
        (void)hipSetDevice(g->dev[stall]);
        void * dflag = nullptr;
        if (hipHostGetDevicePointer(&dflag, g_stall_flag, 0) == hipSuccess)
            hipLaunchKernelGGL(k_group_stall, dim3(1), dim3(1), 0, g->s_in[stall], (volatile uint32_t *)dflag);
    }
    sdsl_hip_status st = body();
    // The call is synchronous for the caller: everything has landed in d_out on return — or the DEADLINE has passed (option
    // "group_timeout_ms", SDSL_HIP_GROUP_TIMEOUT_MS; default 120 s, 0 = none).  A collective whose peer never turns up does not
    // raise an error anywhere: its stream simply never drains, and hipStreamSynchronize would wait with it for ever (a first 8-GPU run
    // is the first time RCCL point-to-point between distinct devices executes at all).  So the wait is a poll of one event per
    // stream against the clock; on expiry the call names the member, its device and the stage that did not finish, returns
    // SDSL_HIP_ERR_HIP and marks the group unusable (the work that is stuck cannot be cancelled: destroy the group).
    const int64_t limit_ms = g_group_timeout_ms.load();
    const auto t_start = std::chrono::steady_clock::now();
    static const char * const stage[3] = {"scatter", "kernels", "gather"};
    std::vector<hipEvent_t> done((size_t)G * 3, nullptr);
    for (int r = 0; r < G && st == SDSL_HIP_OK; ++r)
    {
        (void)hipSetDevice(g->dev[r]);
        int k = 0;
        for (hipStream_t s : {g->s_in[r], g->s_k[r], g->s_out[r]})
        {
            hipError_t e = hipEventCreateWithFlags(&done[(size_t)r * 3 + k], hipEventDisableTiming);
            if (e == hipSuccess)
                e = hipEventRecord(done[(size_t)r * 3 + k], s);
            if (e != hipSuccess && st == SDSL_HIP_OK)
                st = hip_fail(e, "group stream wait", __FILE__, __LINE__);
            ++k;
        }
    }
    for (size_t i = 0; i < done.size() && st == SDSL_HIP_OK;)
    { // (in order: when the last one is ready, all are)
        const hipError_t e = done[i] ? hipEventQuery(done[i]) : hipSuccess;
        if (e == hipSuccess)
        {
            ++i;
            continue;
        }
        if (e != hipErrorNotReady)
        {
            st = hip_fail(e, "group stream wait", __FILE__, __LINE__);
            break;
        }
        const int64_t waited = std::chrono::duration_cast<std::chrono::milliseconds>(std::chrono::steady_clock::now() - t_start).count();
        if (limit_ms > 0 && waited > limit_ms)
        { // every stage that has not finished, by member (stages behind a stuck one wait for it; so may another member's stream that
          // shares its hardware queue — the first member listed with ALL its later stages is where to look)
            std::string who;
            for (size_t j = 0; j < done.size(); ++j)
                if (done[j] && hipEventQuery(done[j]) == hipErrorNotReady)
                {
                    char buf[96];
                    snprintf(buf, sizeof buf, "%smember %d (device %d) %s", who.empty() ? "" : ", ", (int)(j / 3), g->dev[j / 3], stage[j % 3]);
                    who += buf;
                }
            (void)hipGetLastError();
            set_error("device group: not finished within %lld ms (option group_timeout_ms): %s — a peer that never posted its side of a "
                      "transfer, or a kernel that does not end.  The group is unusable now; destroy it", (long long)limit_ms, who.c_str());
            st = SDSL_HIP_ERR_HIP;
            g->poisoned = true;
            break;
        }
        std::this_thread::sleep_for(std::chrono::microseconds(waited < 5 ? 20 : 200));
    }
    (void)hipGetLastError(); // (hipErrorNotReady of the polls is not an error of the call)
    for (hipEvent_t e : done)
        if (e)
            ev.push_back(e);
    if (!g->poisoned)
        for (hipEvent_t e : ev)
            (void)hipEventDestroy(e);
    (void)hipSetDevice(g->dev[0]);
    return st;
}

// arguments of a group batch that may live on the host: staged on the root device
struct RootStaged
{
    Staged in, out;
};

} // namespace

extern "C" {

sdsl_hip_status sdsl_hip_group_create(const int32_t * devices, int32_t n, sdsl_hip_group_t * out)
{
    if (!out || !devices || n < 1 || n > 64)
    {
        set_error("group_create: invalid argument");
        return SDSL_HIP_ERR_INVALID;
    }
    *out = nullptr;
    const char * tr = getenv("SDSL_HIP_GROUP_TRANSPORT");
    const bool copy_transport = tr && !strcmp(tr, "copy");
    for (int i = 0; i < n; ++i)
    {
        SH_TRY(check_device(devices[i]));
        for (int j = 0; j < i; ++j)
            if (devices[j] == devices[i] && !copy_transport)
            {
                set_error("group_create: device %d listed twice (only the copy transport accepts that)", devices[i]);
                return SDSL_HIP_ERR_INVALID;
            }
    }
    const Rccl * R = copy_transport ? nullptr : rccl();
    if (!R && !copy_transport)
    {
        set_error("group_create: librccl.so not found (%s)", dlerror() ? dlerror() : "dlopen failed");
        return SDSL_HIP_ERR_NO_DEVICE;
    }
    return guarded("group_create",
                   [&]() -> sdsl_hip_status
                   {
                       std::unique_ptr<sdsl_hip_group_s> g(new sdsl_hip_group_s());
                       g->n = n;
                       g->dev.assign(devices, devices + n);
                       g->comm_a.assign(n, nullptr);
                       g->comm_b.assign(n, nullptr);
                       g->s_in.assign(n, nullptr);
                       g->s_k.assign(n, nullptr);
                       g->s_out.assign(n, nullptr);
                       g->in.resize(n);
                       g->out.resize(n);
                       g->copy = copy_transport;
                       if (!copy_transport)
                       {
                           SH_NCCL(R->CommInitAll(g->comm_a.data(), n, g->dev.data()));
                           SH_NCCL(R->CommInitAll(g->comm_b.data(), n, g->dev.data()));
                       }
                       for (int r = 0; r < n; ++r)
                       {
                           SH_HIP(hipSetDevice(g->dev[r]));
                           SH_HIP(hipStreamCreateWithFlags(&g->s_in[r], hipStreamNonBlocking));
                           SH_HIP(hipStreamCreateWithFlags(&g->s_k[r], hipStreamNonBlocking));
                           SH_HIP(hipStreamCreateWithFlags(&g->s_out[r], hipStreamNonBlocking));
                           for (int q = 0; q < n; ++q)
                               if (q != r)
                               {
                                   int can = 0;
                                   (void)hipDeviceCanAccessPeer(&can, g->dev[r], g->dev[q]);
                                   if (can)
                                       (void)hipDeviceEnablePeerAccess(g->dev[q], 0); // already enabled is fine
                               }
                       }
                       (void)hipGetLastError();
                       SH_HIP(hipSetDevice(g->dev[0]));
                       *out = g.release();
                       return SDSL_HIP_OK;
                   });
}

sdsl_hip_status sdsl_hip_group_destroy(sdsl_hip_group_t g)
{
    delete g;
    return SDSL_HIP_OK;
}

int32_t sdsl_hip_group_size(sdsl_hip_group_t g)
{
    return g ? g->n : 0;
}

int32_t sdsl_hip_group_device(sdsl_hip_group_t g, int32_t r)
{
    return g && r >= 0 && r < g->n ? g->dev[r] : -1;
}

// every device sends `bytes` to the next one of the group (to itself in a group of one) through both communicators and
// the received pattern is verified: a link check that also exercises the send/recv path on a single-GPU box
sdsl_hip_status sdsl_hip_group_loopback(sdsl_hip_group_t g, uint64_t bytes, float * ms_out)
{
    if (!g || bytes == 0)
    {
        set_error("group_loopback: invalid argument");
        return SDSL_HIP_ERR_INVALID;
    }
    const Rccl * R = rccl();
    const int G = g->n;
    std::vector<DevBuf> src(G), dst(G);
    std::vector<uint8_t> h(bytes);
    for (int r = 0; r < G; ++r)
    {
        SH_HIP(hipSetDevice(g->dev[r]));
        SH_TRY(src[r].alloc(bytes));
        SH_TRY(dst[r].alloc(bytes, true));
        for (uint64_t i = 0; i < bytes; ++i)
            h[i] = (uint8_t)(i * 131 + r * 17 + 1);
        SH_HIP(hipMemcpy(src[r].p, h.data(), bytes, hipMemcpyHostToDevice));
    }
    hipEvent_t e0, e1;
    SH_HIP(hipSetDevice(g->dev[0]));
    SH_HIP(hipEventCreate(&e0));
    SH_HIP(hipEventCreate(&e1));
    sdsl_hip_status st = SDSL_HIP_OK;
    for (int pass = 0; pass < 2 && st == SDSL_HIP_OK; ++pass)
    {
        auto & comm = pass ? g->comm_b : g->comm_a;
        auto & str = pass ? g->s_out : g->s_in;
        if (pass == 0)
            SH_HIP(hipEventRecord(e0, str[0]));
        auto go = [&]() -> sdsl_hip_status
        {
            if (g->copy)
            {
                for (int r = 0; r < G; ++r)
                    SH_TRY(copy_xfer(g, src[r].p, r, str[r], dst[(r + 1) % G].p, (r + 1) % G, str[(r + 1) % G], bytes));
                return SDSL_HIP_OK;
            }
            SH_NCCL(R->GroupStart());
            for (int r = 0; r < G; ++r)
            {
                SH_NCCL_G(R->Send(src[r].p, bytes, ncclUint8, (r + 1) % G, comm[r], str[r]));
                SH_NCCL_G(R->Recv(dst[r].p, bytes, ncclUint8, (r + G - 1) % G, comm[r], str[r]));
            }
            SH_NCCL(R->GroupEnd());
            return SDSL_HIP_OK;
        };
        st = go();
        if (pass == 0 && st == SDSL_HIP_OK)
            SH_HIP(hipEventRecord(e1, str[0]));
        for (int r = 0; r < G && st == SDSL_HIP_OK; ++r)
        {
            SH_HIP(hipSetDevice(g->dev[r]));
            SH_HIP(hipStreamSynchronize(str[r]));
            std::vector<uint8_t> got(bytes);
            SH_HIP(hipMemcpy(got.data(), dst[r].p, bytes, hipMemcpyDeviceToHost));
            const int from = (r + G - 1) % G;
            for (uint64_t i = 0; i < bytes; ++i)
                if (got[i] != (uint8_t)(i * 131 + from * 17 + 1))
                {
                    set_error("group_loopback: device %d received wrong data at byte %llu", g->dev[r], (unsigned long long)i);
                    st = SDSL_HIP_ERR_HIP;
                    break;
                }
            SH_HIP(hipMemset(dst[r].p, 0, bytes));
        }
    }
    SH_HIP(hipSetDevice(g->dev[0]));
    if (ms_out && st == SDSL_HIP_OK)
    {
        SH_HIP(hipEventSynchronize(e1));
        SH_HIP(hipEventElapsedTime(ms_out, e0, e1));
    }
    (void)hipEventDestroy(e0);
    (void)hipEventDestroy(e1);
    return st;
}

sdsl_hip_status sdsl_hip_group_bv_replicate(sdsl_hip_group_t g, sdsl_hip_bv_t root, sdsl_hip_bv_t * replicas)
{
    if (!g || !root || !replicas)
    {
        set_error("group_bv_replicate: null argument");
        return SDSL_HIP_ERR_INVALID;
    }
    const Rccl * R = rccl();
    BvHost & src = bv_host_of(root);
    if (src.device != g->dev[0])
    {
        set_error("group_bv_replicate: the root handle lives on device %d, the group's root is device %d", src.device, g->dev[0]);
        return SDSL_HIP_ERR_INVALID;
    }
    const int G = g->n;
    replicas[0] = root;
    for (int r = 1; r < G; ++r)
        replicas[r] = nullptr;
    auto cleanup = [&]()
    {
        for (int r = 1; r < G; ++r)
            if (replicas[r])
            {
                (void)sdsl_hip_bv_destroy(replicas[r]);
                replicas[r] = nullptr;
            }
    };
    for (int r = 1; r < G; ++r)
    {
        sdsl_hip_status st = bv_new_replica(src, g->dev[r], &replicas[r]);
        if (st != SDSL_HIP_OK)
        {
            cleanup();
            return st;
        }
    }
    // one broadcast per device buffer of the layout (rank lines, the select directories and their sparse-stretch tables)
    auto buf_of = [](BvHost & h, int which) -> DevBuf &
    {
        switch (which)
        {
        case 0: return h.lines;
        case 1: return h.sel[0];
        case 2: return h.sel[1];
        case 3: return h.lmask[0];
        case 4: return h.lmask[1];
        case 5: return h.lidx[0];
        case 6: return h.lidx[1];
        case 7: return h.lpos[0];
        default: return h.lpos[1];
        }
    };
    auto bcast = [&](int which) -> sdsl_hip_status
    {
        const DevBuf & b0 = buf_of(src, which);
        if (!b0.p || !b0.bytes)
            return SDSL_HIP_OK;
        if (g->copy)
        {
            for (int r = 1; r < G; ++r)
                SH_TRY(copy_xfer(g, b0.p, 0, g->s_in[0], buf_of(bv_host_of(replicas[r]), which).p, r, g->s_in[r], b0.bytes));
            return SDSL_HIP_OK;
        }
        SH_NCCL(R->GroupStart());
        for (int r = 0; r < G; ++r)
        {
            BvHost & d = bv_host_of(replicas[r]);
            DevBuf & br = buf_of(d, which);
            SH_NCCL_G(R->Broadcast(b0.p, br.p, b0.bytes, ncclUint8, 0, g->comm_a[r], g->s_in[r]));
        }
        SH_NCCL(R->GroupEnd());
        return SDSL_HIP_OK;
    };
    sdsl_hip_status st = SDSL_HIP_OK;
    for (int which = 0; which < 9 && st == SDSL_HIP_OK; ++which)
        st = bcast(which);
    for (int r = 0; r < G; ++r)
    {
        (void)hipSetDevice(g->dev[r]);
        hipError_t e = hipStreamSynchronize(g->s_in[r]);
        if (e != hipSuccess && st == SDSL_HIP_OK)
            st = hip_fail(e, "group_bv_replicate", __FILE__, __LINE__);
    }
    (void)hipSetDevice(g->dev[0]);
    if (st != SDSL_HIP_OK)
        cleanup();
    return st;
}

static sdsl_hip_status group_bv_query(sdsl_hip_group_t g, const sdsl_hip_bv_t * replicas, int32_t bit, const uint64_t * arg, uint64_t n,
                                      uint64_t * out, int32_t chunks, bool select)
{
    if (!g || !replicas || (bit != 0 && bit != 1) || (n && (!arg || !out)))
    {
        set_error("group_bv_%s_batch: invalid argument", select ? "select" : "rank");
        return SDSL_HIP_ERR_INVALID;
    }
    for (int r = 0; r < g->n; ++r)
        if (!replicas[r] || bv_host_of(replicas[r]).device != g->dev[r])
        {
            set_error("group_bv_%s_batch: replicas[%d] is not a handle on device %d", select ? "select" : "rank", r, g->dev[r]);
            return SDSL_HIP_ERR_INVALID;
        }
    if (n == 0)
        return SDSL_HIP_OK;
    SH_HIP(hipSetDevice(g->dev[0]));
    Staged in, o;
    SH_TRY(in.in(arg, n * 8, nullptr));
    SH_TRY(o.out(out, n * 8));
    SH_HIP(hipDeviceSynchronize()); // staging copies ran on the null stream; the group's streams are non-blocking
    const sdsl_hip_status st =
        group_run(g, (const uint8_t *)in.dev, 8, (uint8_t *)o.dev, 8, n, chunks,
                  [&](int r, const uint8_t * pin, uint8_t * pout, uint64_t cnt, hipStream_t s) -> sdsl_hip_status
                  {
                      return select ? sdsl_hip_bv_select_batch(replicas[r], bit, (const uint64_t *)pin, cnt, (uint64_t *)pout, s)
                                    : sdsl_hip_bv_rank_batch(replicas[r], bit, (const uint64_t *)pin, cnt, (uint64_t *)pout, s);
                  });
    if (st != SDSL_HIP_OK)
    {
        if (g->poisoned)
        { // stuck work may still read the staging copies, and freeing them would wait for it
            in.tmp.leak();
            o.tmp.leak();
        }
        return st;
    }
    return o.finish(nullptr);
}

sdsl_hip_status sdsl_hip_group_bv_rank_batch(sdsl_hip_group_t g, const sdsl_hip_bv_t * replicas, int32_t bit, const uint64_t * idx,
                                             uint64_t n, uint64_t * out, int32_t chunks)
{
    return group_bv_query(g, replicas, bit, idx, n, out, chunks, false);
}

sdsl_hip_status sdsl_hip_group_bv_select_batch(sdsl_hip_group_t g, const sdsl_hip_bv_t * replicas, int32_t bit, const uint64_t * i,
                                               uint64_t n, uint64_t * out, int32_t chunks)
{
    return group_bv_query(g, replicas, bit, i, n, out, chunks, true);
}

// load time of configs[4]: the text goes to every device with one broadcast and every device lays out its own FM-index
sdsl_hip_status sdsl_hip_group_fm_create_from_text(sdsl_hip_group_t g, const uint8_t * text, uint64_t n_text, uint32_t flags,
                                                   sdsl_hip_fm_t * replicas)
{
    if (!g || !replicas || (!text && n_text))
    {
        set_error("group_fm_create_from_text: null argument");
        return SDSL_HIP_ERR_INVALID;
    }
    const Rccl * R = rccl();
    const int G = g->n;
    for (int r = 0; r < G; ++r)
        replicas[r] = nullptr;
    SH_HIP(hipSetDevice(g->dev[0]));
    Staged t0;
    SH_TRY(t0.in(text, n_text, nullptr));
    SH_HIP(hipDeviceSynchronize());
    std::vector<DevBuf> copy(G);
    std::vector<const uint8_t *> tp(G);
    tp[0] = (const uint8_t *)t0.dev;
    for (int r = 1; r < G; ++r)
    {
        SH_HIP(hipSetDevice(g->dev[r]));
        SH_TRY(copy[r].alloc(n_text));
        tp[r] = copy[r].as<uint8_t>();
    }
    if (G > 1 && n_text && g->copy)
    {
        for (int r = 1; r < G; ++r)
            SH_TRY(copy_xfer(g, tp[0], 0, g->s_in[0], (void *)tp[r], r, g->s_in[r], n_text));
        for (int r = 0; r < G; ++r)
        {
            SH_HIP(hipSetDevice(g->dev[r]));
            SH_HIP(hipStreamSynchronize(g->s_in[r]));
        }
    }
    else if (G > 1 && n_text)
    {
        SH_NCCL(R->GroupStart());
        for (int r = 0; r < G; ++r)
            SH_NCCL_G(R->Broadcast(tp[0], (void *)tp[r], n_text, ncclUint8, 0, g->comm_a[r], g->s_in[r]));
        SH_NCCL(R->GroupEnd());
        for (int r = 0; r < G; ++r)
        {
            SH_HIP(hipSetDevice(g->dev[r]));
            SH_HIP(hipStreamSynchronize(g->s_in[r]));
        }
    }
    sdsl_hip_status st = SDSL_HIP_OK;
    std::vector<std::thread> th;
    std::vector<sdsl_hip_status> sts(G, SDSL_HIP_OK);
    std::vector<std::string> msg(G);
    for (int r = 0; r < G; ++r) // the builds are independent: one host thread per device
        th.emplace_back(
            [&, r]
            {
                sts[r] = sdsl_hip_fm_create_from_text_ex(tp[r], n_text, g->dev[r], flags, &replicas[r]);
                if (sts[r] != SDSL_HIP_OK)
                    msg[r] = last_error_message();
            });
    for (auto & x : th)
        x.join();
    for (int r = 0; r < G; ++r)
        if (sts[r] != SDSL_HIP_OK && st == SDSL_HIP_OK)
        {
            st = sts[r];
            set_error("%s", msg[r].c_str());
        }
    if (st != SDSL_HIP_OK)
        for (int r = 0; r < G; ++r)
            if (replicas[r])
            {
                (void)sdsl_hip_fm_destroy(replicas[r]);
                replicas[r] = nullptr;
            }
    (void)hipSetDevice(g->dev[0]);
    return st;
}

sdsl_hip_status sdsl_hip_group_fm_count_batch(sdsl_hip_group_t g, const sdsl_hip_fm_t * replicas, const uint8_t * patterns, uint32_t m,
                                              uint64_t n_patterns, uint64_t * out, int32_t chunks)
{
    if (!g || !replicas || (n_patterns && (!out || (!patterns && m))))
    {
        set_error("group_fm_count_batch: invalid argument");
        return SDSL_HIP_ERR_INVALID;
    }
    for (int r = 0; r < g->n; ++r)
        if (!replicas[r])
        {
            set_error("group_fm_count_batch: replicas[%d] is null", r);
            return SDSL_HIP_ERR_INVALID;
        }
    if (n_patterns == 0)
        return SDSL_HIP_OK;
    if (m == 0)
        return sdsl_hip_fm_count_batch(replicas[0], patterns, 0, n_patterns, out, nullptr); // count("") = size(): no data to move
    SH_HIP(hipSetDevice(g->dev[0]));
    Staged in, o;
    SH_TRY(in.in(patterns, n_patterns * m, nullptr));
    SH_TRY(o.out(out, n_patterns * 8));
    SH_HIP(hipDeviceSynchronize());
    const sdsl_hip_status st = group_run(g, (const uint8_t *)in.dev, m, (uint8_t *)o.dev, 8, n_patterns, chunks,
                                         [&](int r, const uint8_t * pin, uint8_t * pout, uint64_t cnt, hipStream_t s) -> sdsl_hip_status
                                         { return sdsl_hip_fm_count_batch(replicas[r], pin, m, cnt, (uint64_t *)pout, s); });
    if (st != SDSL_HIP_OK)
    {
        if (g->poisoned)
        {
            in.tmp.leak();
            o.tmp.leak();
        }
        return st;
    }
    return o.finish(nullptr);
}

} // extern "C"
