// In-library neighbour transport over RCCL (xGMI): the wxa_comm callbacks of the host layer implemented with
// ncclSend / ncclRecv groups enqueued on the caller's HIP stream.
//
// Replaces, for GPU runs, the Python callbacks of warpx_amd/distributed.py (torch.distributed P2P): no ctypes hop and
// no host-blocking wait per exchange, no reliance on message tags (RCCL has none: the messages of one peer pair are
// matched in posting order, and BrickComm posts both sides in the same canonical order -- "to minus, to plus" on the
// sender is "from plus, from minus" on the receiver, i.e. the same two messages in the same order when both faces
// of a direction lead to the same peer, as on a 2 x 2 x 2 layout).  The reference moves the same data with MPI
// inside amrex FillBoundary / SumBoundary / Redistribute (Source/ablastr/utils/Communication.cpp:71-175,
// Source/Parallelization/WarpXComm.cpp:699-827,1386-1424).
//
// RCCL is loaded at run time (dlopen), so the library itself has no link dependency on it: single-GPU runs and the
// CPU test builds never touch it, and wxa_rccl_comm_create fails loudly where it is missing.
#include "common.hpp"

#include <dlfcn.h>
#include <stdlib.h>
#include <string.h>

#include <mutex>
#include <vector>

namespace wxa {

// the slice of the RCCL API used here (rccl.h of ROCm 7.x; opaque handles and plain enums)
typedef struct ncclComm* ncclComm_t;
typedef struct { char internal[128]; } ncclUniqueId;
typedef int ncclResult_t;       // 0 = ncclSuccess
constexpr int kNcclChar = 0;    // ncclInt8 / ncclChar

struct RcclApi {
    void* dl = nullptr;
    ncclResult_t (*GetUniqueId)(ncclUniqueId*) = nullptr;
    ncclResult_t (*CommInitRank)(ncclComm_t*, int, ncclUniqueId, int) = nullptr;
    ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
    ncclResult_t (*CommSplit)(ncclComm_t, int, int, ncclComm_t*, void*) = nullptr;   // optional (NCCL >= 2.18 API)
    ncclResult_t (*GroupStart)() = nullptr;
    ncclResult_t (*GroupEnd)() = nullptr;
    ncclResult_t (*Send)(const void*, size_t, int, int, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*Recv)(void*, size_t, int, int, ncclComm_t, hipStream_t) = nullptr;
    const char* (*GetErrorString)(ncclResult_t) = nullptr;
};

static RcclApi* rccl_api() {
    static RcclApi api;
    static std::once_flag once;
    std::call_once(once, [] {
        // a copy that the process already holds (PyTorch ships its own librccl.so) is reused by the loader
        const char* names[] = {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"};
        for (const char* n : names)
            if ((api.dl = dlopen(n, RTLD_NOW | RTLD_GLOBAL)) != nullptr) break;
        if (api.dl) {
            *(void**)&api.GetUniqueId = dlsym(api.dl, "ncclGetUniqueId");
            *(void**)&api.CommInitRank = dlsym(api.dl, "ncclCommInitRank");
            *(void**)&api.CommDestroy = dlsym(api.dl, "ncclCommDestroy");
            *(void**)&api.CommSplit = dlsym(api.dl, "ncclCommSplit");
            *(void**)&api.GroupStart = dlsym(api.dl, "ncclGroupStart");
            *(void**)&api.GroupEnd = dlsym(api.dl, "ncclGroupEnd");
            *(void**)&api.Send = dlsym(api.dl, "ncclSend");
            *(void**)&api.Recv = dlsym(api.dl, "ncclRecv");
            *(void**)&api.GetErrorString = dlsym(api.dl, "ncclGetErrorString");
            if (!api.GetUniqueId || !api.CommInitRank || !api.CommDestroy || !api.GroupStart || !api.GroupEnd ||
                !api.Send || !api.Recv) {
                dlclose(api.dl);
                api.dl = nullptr;
            }
        }
    });
    return api.dl ? &api : nullptr;
}

struct RcclCtx {
    RcclApi* api = nullptr;
    ncclComm_t comm = nullptr;
    // The count round has a communicator of its own (ncclCommSplit of `comm`, all ranks, same order): RCCL runs the
    // operations of ONE communicator in posting order whatever their streams, so on the shared communicator the 8-byte
    // count messages of step n + 1 queued behind the data exchanges of step n that the compute stream had not reached
    // yet -- the host, which blocks on the counts, then waited for the whole step.  Where ncclCommSplit is missing, or
    // with WXA_RCCL_SHARED_COMM=1, ccomm == comm as before.
    ncclComm_t ccomm = nullptr;
    int rank = 0, nranks = 1;
    bool loopback = false;        // test mode: messages to this rank go through ncclSend / ncclRecv too
    hipStream_t cstream = nullptr;   // count exchanges: their own stream -- the host waits for these 8-byte messages only,
                                     // not for the kernels of the main stream
    int64_t* dcounts = nullptr;      // device staging, 2 x kMaxMsg
    int64_t* hcounts = nullptr;      // pinned host staging, 2 x kMaxMsg
    // statistics (wxa_rccl_comm_stats)
    int64_t n_exchanges = 0, n_messages = 0, bytes_sent = 0, n_count_exchanges = 0;
    bool timing = false;       // wxa_rccl_comm_set_timing: a pair of events around every exchange (diagnostic runs only)
    std::vector<std::pair<hipEvent_t, hipEvent_t>> events;   // recorded pairs, not yet read
    std::vector<std::pair<hipEvent_t, hipEvent_t>> pool;     // read pairs, ready for reuse (no event creation per exchange)
    double timed_ms = 0.0;
    int64_t timed_exchanges = 0;
};
constexpr int kMaxMsg = 32;

#define WXA_NCCL(ctx, expr)                                                                        \
    do {                                                                                           \
        const ncclResult_t _r = (expr);                                                            \
        if (_r != 0) {                                                                             \
            set_last_error("%s failed: %s", #expr,                                                 \
                           (ctx)->api->GetErrorString ? (ctx)->api->GetErrorString(_r) : "RCCL error"); \
            return -1;                                                                             \
        }                                                                                          \
    } while (0)
// inside ncclGroupStart .. ncclGroupEnd: leave the group before returning, or every later call of this thread would
// be queued into it
#define WXA_NCCL_IN_GROUP(ctx, expr)                                                               \
    do {                                                                                           \
        const ncclResult_t _r = (expr);                                                            \
        if (_r != 0) {                                                                             \
            set_last_error("%s failed: %s", #expr,                                                 \
                           (ctx)->api->GetErrorString ? (ctx)->api->GetErrorString(_r) : "RCCL error"); \
            (void)(ctx)->api->GroupEnd();                                                          \
            return -1;                                                                             \
        }                                                                                          \
    } while (0)
#define WXA_HIP_RC(expr)                                                                           \
    do {                                                                                           \
        const hipError_t _e = (expr);                                                              \
        if (_e != hipSuccess) {                                                                    \
            set_last_error("%s failed: %s", #expr, hipGetErrorString(_e));                         \
            return -1;                                                                             \
        }                                                                                          \
    } while (0)

static void drain_events(RcclCtx* c) {
    for (auto& ev : c->events) {
        float ms = 0.f;
        if (hipEventSynchronize(ev.second) == hipSuccess && hipEventElapsedTime(&ms, ev.first, ev.second) == hipSuccess) {
            c->timed_ms += ms;
            c->timed_exchanges += 1;
        }
        c->pool.push_back(ev);
    }
    c->events.clear();
}
static void destroy_events(RcclCtx* c) {
    drain_events(c);
    for (auto& ev : c->pool) { (void)hipEventDestroy(ev.first); (void)hipEventDestroy(ev.second); }
    c->pool.clear();
}

// nmsg sends + nmsg receives of device buffers, enqueued on `stream`; returns as soon as they are enqueued
static int rccl_exchange(void* vctx, int nmsg, const int32_t* send_peer, void* const* send_buf, const int64_t* send_bytes,
                         const int32_t* recv_peer, void* const* recv_buf, const int64_t* recv_bytes, void* vstream) {
    RcclCtx* c = static_cast<RcclCtx*>(vctx);
    hipStream_t st = (hipStream_t)vstream;
    hipEvent_t e0 = nullptr, e1 = nullptr;
    if (c->timing) {
        if (c->pool.empty() && c->events.size() >= 4096) drain_events(c);   // (a host wait: timing runs only)
        if (!c->pool.empty()) {
            e0 = c->pool.back().first; e1 = c->pool.back().second;
            c->pool.pop_back();
        } else {
            WXA_HIP_RC(hipEventCreate(&e0));
            WXA_HIP_RC(hipEventCreate(&e1));
        }
        WXA_HIP_RC(hipEventRecord(e0, st));
    }
    // messages to this rank itself (a direction of one brick handled by the caller never gets here; this is the
    // 1 x 1 x N style corner where both neighbours are this rank): message i of the send list pairs with receive i
    if (!c->loopback) {
        for (int i = 0; i < nmsg; ++i) {
            if (send_peer[i] != c->rank) continue;
            if (recv_peer[i] != c->rank || recv_bytes[i] != send_bytes[i]) {
                set_last_error("rccl_exchange: self message %d without its matching receive", i);
                return -1;
            }
            if (send_bytes[i] > 0)
                WXA_HIP_RC(hipMemcpyAsync(recv_buf[i], send_buf[i], (size_t)send_bytes[i], hipMemcpyDeviceToDevice, st));
        }
    }
    bool any = false;
    for (int i = 0; i < nmsg; ++i) {
        const bool s = (c->loopback || send_peer[i] != c->rank) && send_bytes[i] > 0;
        const bool r = (c->loopback || recv_peer[i] != c->rank) && recv_bytes[i] > 0;
        any = any || s || r;
    }
    if (any) {
        WXA_NCCL(c, c->api->GroupStart());
        for (int i = 0; i < nmsg; ++i) {
            if ((c->loopback || send_peer[i] != c->rank) && send_bytes[i] > 0) {
                WXA_NCCL_IN_GROUP(c, c->api->Send(send_buf[i], (size_t)send_bytes[i], kNcclChar, send_peer[i], c->comm, st));
                c->bytes_sent += send_bytes[i];
                c->n_messages += 1;
            }
        }
        for (int i = 0; i < nmsg; ++i) {
            if ((c->loopback || recv_peer[i] != c->rank) && recv_bytes[i] > 0)
                WXA_NCCL_IN_GROUP(c, c->api->Recv(recv_buf[i], (size_t)recv_bytes[i], kNcclChar, recv_peer[i], c->comm, st));
        }
        WXA_NCCL(c, c->api->GroupEnd());
    }
    if (c->timing) {
        WXA_HIP_RC(hipEventRecord(e1, st));
        c->events.emplace_back(e0, e1);
    }
    c->n_exchanges += 1;
    return 0;
}

// one int64 per message each way: staged through device memory (RCCL moves device buffers), on the transport's own
// stream, so the host waits for these 8-byte messages only -- not for whatever the compute streams still hold
static int rccl_exchange_counts(void* vctx, int nmsg, const int32_t* send_peer, const int64_t* send_val,
                                const int32_t* recv_peer, int64_t* recv_val) {
    RcclCtx* c = static_cast<RcclCtx*>(vctx);
    if (nmsg > kMaxMsg) {
        set_last_error("rccl_exchange_counts: more than %d messages", kMaxMsg);
        return -1;
    }
    bool any = false;
    for (int i = 0; i < nmsg; ++i) {
        c->hcounts[i] = send_val[i];
        if (!c->loopback && send_peer[i] == c->rank) recv_val[i] = send_val[i];   // pairs with receive i, as above
        any = any || c->loopback || send_peer[i] != c->rank || recv_peer[i] != c->rank;
    }
    c->n_count_exchanges += 1;
    if (!any) return 0;
    WXA_HIP_RC(hipMemcpyAsync(c->dcounts, c->hcounts, sizeof(int64_t) * nmsg, hipMemcpyHostToDevice, c->cstream));
    WXA_NCCL(c, c->api->GroupStart());
    for (int i = 0; i < nmsg; ++i)
        if (c->loopback || send_peer[i] != c->rank)
            WXA_NCCL_IN_GROUP(c, c->api->Send(c->dcounts + i, sizeof(int64_t), kNcclChar, send_peer[i], c->ccomm, c->cstream));
    for (int i = 0; i < nmsg; ++i)
        if (c->loopback || recv_peer[i] != c->rank)
            WXA_NCCL_IN_GROUP(c, c->api->Recv(c->dcounts + kMaxMsg + i, sizeof(int64_t), kNcclChar, recv_peer[i], c->ccomm, c->cstream));
    WXA_NCCL(c, c->api->GroupEnd());
    WXA_HIP_RC(hipMemcpyAsync(c->hcounts + kMaxMsg, c->dcounts + kMaxMsg, sizeof(int64_t) * nmsg, hipMemcpyDeviceToHost,
                              c->cstream));
    WXA_HIP_RC(hipStreamSynchronize(c->cstream));
    for (int i = 0; i < nmsg; ++i)
        if (c->loopback || recv_peer[i] != c->rank) recv_val[i] = c->hcounts[kMaxMsg + i];
    return 0;
}

}  // namespace wxa

using namespace wxa;

extern "C" {

wxa_status wxa_rccl_unique_id(char id[WXA_RCCL_ID_BYTES]) {
    WXA_REQUIRE(id, "null argument");
    RcclApi* api = rccl_api();
    if (!api) {
        set_last_error("wxa_rccl_unique_id: librccl.so could not be loaded");
        return WXA_ERR_UNSUPPORTED;
    }
    ncclUniqueId u;
    const ncclResult_t r = api->GetUniqueId(&u);
    if (r != 0) {
        set_last_error("ncclGetUniqueId failed: %s", api->GetErrorString ? api->GetErrorString(r) : "RCCL error");
        return WXA_ERR_HIP;
    }
    static_assert(sizeof(u.internal) == WXA_RCCL_ID_BYTES, "unique id size");
    memcpy(id, u.internal, WXA_RCCL_ID_BYTES);
    return WXA_OK;
}

wxa_status wxa_rccl_comm_create(const char id[WXA_RCCL_ID_BYTES], int32_t rank, int32_t nranks, int32_t flags,
                                wxa_comm* out) {
    WXA_REQUIRE(id && out && nranks >= 1 && rank >= 0 && rank < nranks, "bad arguments");
    RcclApi* api = rccl_api();
    if (!api) {
        set_last_error("wxa_rccl_comm_create: librccl.so could not be loaded");
        return WXA_ERR_UNSUPPORTED;
    }
    RcclCtx* c = new RcclCtx();
    c->api = api;
    c->rank = rank;
    c->nranks = nranks;
    c->loopback = (flags & WXA_RCCL_LOOPBACK) != 0;
    c->timing = (flags & WXA_RCCL_TIMING) != 0;
    ncclUniqueId u;
    memcpy(u.internal, id, WXA_RCCL_ID_BYTES);
    const ncclResult_t r = api->CommInitRank(&c->comm, nranks, u, rank);
    if (r != 0) {
        set_last_error("ncclCommInitRank failed: %s", api->GetErrorString ? api->GetErrorString(r) : "RCCL error");
        delete c;
        return WXA_ERR_HIP;
    }
    c->ccomm = c->comm;
    {
        const char* shared = getenv("WXA_RCCL_SHARED_COMM");
        if (api->CommSplit && !(shared && shared[0] == '1')) {
            ncclComm_t cc = nullptr;
            // collective over the parent communicator: every rank creates its transport at the same point
            if (api->CommSplit(c->comm, /*color=*/0, /*key=*/rank, &cc, nullptr) == 0 && cc) c->ccomm = cc;
        }
    }
    if (hipStreamCreateWithFlags(&c->cstream, hipStreamNonBlocking) != hipSuccess ||
        hipMalloc((void**)&c->dcounts, sizeof(int64_t) * 2 * kMaxMsg) != hipSuccess ||
        hipHostMalloc((void**)&c->hcounts, sizeof(int64_t) * 2 * kMaxMsg) != hipSuccess) {
        set_last_error("wxa_rccl_comm_create: staging allocation failed");
        if (c->cstream) (void)hipStreamDestroy(c->cstream);
        if (c->dcounts) (void)hipFree(c->dcounts);
        if (c->hcounts) (void)hipHostFree(c->hcounts);
        if (c->ccomm && c->ccomm != c->comm) (void)api->CommDestroy(c->ccomm);
        (void)api->CommDestroy(c->comm);
        delete c;
        return WXA_ERR_NOMEM;
    }
    out->ctx = c;
    out->rank = rank;
    out->nranks = nranks;
    out->exchange = rccl_exchange;
    out->exchange_counts = rccl_exchange_counts;
    return WXA_OK;
}

void wxa_rccl_comm_destroy(wxa_comm* comm) {
    if (!comm || !comm->ctx || comm->exchange != rccl_exchange) return;
    RcclCtx* c = static_cast<RcclCtx*>(comm->ctx);
    destroy_events(c);
    if (c->cstream) { (void)hipStreamSynchronize(c->cstream); (void)hipStreamDestroy(c->cstream); }
    if (c->dcounts) (void)hipFree(c->dcounts);
    if (c->hcounts) (void)hipHostFree(c->hcounts);
    if (c->ccomm && c->ccomm != c->comm) c->api->CommDestroy(c->ccomm);
    if (c->comm) c->api->CommDestroy(c->comm);
    delete c;
    comm->ctx = nullptr;
}

wxa_status wxa_rccl_comm_set_timing(wxa_comm* comm, int32_t on) {
    WXA_REQUIRE(comm && comm->ctx && comm->exchange == rccl_exchange, "not an RCCL transport");
    RcclCtx* c = static_cast<RcclCtx*>(comm->ctx);
    if (!on) drain_events(c);
    c->timing = on != 0;
    return WXA_OK;
}

wxa_status wxa_rccl_comm_stats(wxa_comm* comm, wxa_rccl_stats* out, int32_t reset) {
    WXA_REQUIRE(comm && comm->ctx && out && comm->exchange == rccl_exchange, "not an RCCL transport");
    RcclCtx* c = static_cast<RcclCtx*>(comm->ctx);
    drain_events(c);
    out->n_exchanges = c->n_exchanges;
    out->n_messages = c->n_messages;
    out->bytes_sent = c->bytes_sent;
    out->n_count_exchanges = c->n_count_exchanges;
    out->timed_exchanges = c->timed_exchanges;
    out->timed_ms = c->timed_ms;
    if (reset) {
        c->n_exchanges = c->n_messages = c->bytes_sent = c->n_count_exchanges = c->timed_exchanges = 0;
        c->timed_ms = 0.0;
    }
    return WXA_OK;
}

}  // extern "C"
