// The one exchange step of sharded meta-training through the C ABI: all-reduce(sum) of the flat meta-gradient over the ranks'
// GPUs with RCCL (xGMI), issued on a SIDE stream and event-ordered against the stream that produces the buffer, so that the
// next task's inner forward runs beside it (reference site being replaced: the `_updates[n] += p.grad` /
// `_updates[n] /= counter` of a single process, src/fo_meta_interface.py:190-196,200-202 -- the reference has no collective).
//
//   masr_allreduce_unique_id  rank 0 draws the RCCL id; the caller hands its 128 bytes to the other ranks (any side channel)
//   masr_allreduce_init       ncclCommInitRank on the CURRENT device; owns a side stream + events
//   masr_allreduce            chunk c: [clip-scale chunk c on the producer stream] -> event -> side stream: ncclAllReduce(chunk c)
//                             (the clip coefficient of clip_grad_norm_ is known only after the whole backward, :148-149, so the
//                             "scale by coef" pass is pipelined chunk by chunk with the collective instead of preceding it)
//   masr_allreduce_wait       makes a stream wait for everything issued so far
//   masr_allreduce_check      host side: RCCL's asynchronous error state + (optionally) a bounded wait for the last exchange -- a stuck
//                             collective ends the job with an error instead of hanging it; a failed call aborts the communicator
//
// librccl is bound with dlopen at init (no link-time dependency: the library loads on hosts without RCCL, and the single-GPU
// path never touches it).  xGMI is point-to-point: a ring all-reduce is bound by one link (~153 GB/s), so the 99.5 MB payload
// goes out as a FEW large chunks (default 4 x 25 MB), not per-tensor buckets.
#include <dlfcn.h>
#include <rccl/rccl.h>

#include <cstdio>
#include <ctime>
#include <cstring>

#include "../../include/masr.h"
#include "common.h"
#include "kernels.h"

namespace {
struct Rccl {
    void* so = nullptr;
    ncclResult_t (*GetUniqueId)(ncclUniqueId*) = nullptr;
    ncclResult_t (*CommInitRank)(ncclComm_t*, int, ncclUniqueId, int) = nullptr;
    ncclResult_t (*AllReduce)(const void*, void*, size_t, ncclDataType_t, ncclRedOp_t, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
    ncclResult_t (*CommAbort)(ncclComm_t) = nullptr;                              // (optional symbols: older libraries lack them)
    ncclResult_t (*CommGetAsyncError)(ncclComm_t, ncclResult_t*) = nullptr;
    const char* (*GetErrorString)(ncclResult_t) = nullptr;
};
Rccl* rccl() {
    static Rccl r;
    static bool tried = false;
    if (!tried) {
        tried = true;
        for (const char* name : {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"}) {
            r.so = dlopen(name, RTLD_NOW | RTLD_GLOBAL);
            if (r.so) break;
        }
        if (r.so) {
            r.GetUniqueId = (decltype(r.GetUniqueId))dlsym(r.so, "ncclGetUniqueId");
            r.CommInitRank = (decltype(r.CommInitRank))dlsym(r.so, "ncclCommInitRank");
            r.AllReduce = (decltype(r.AllReduce))dlsym(r.so, "ncclAllReduce");
            r.CommDestroy = (decltype(r.CommDestroy))dlsym(r.so, "ncclCommDestroy");
            r.GetErrorString = (decltype(r.GetErrorString))dlsym(r.so, "ncclGetErrorString");
            r.CommAbort = (decltype(r.CommAbort))dlsym(r.so, "ncclCommAbort");
            r.CommGetAsyncError = (decltype(r.CommGetAsyncError))dlsym(r.so, "ncclCommGetAsyncError");
            if (!r.GetUniqueId || !r.CommInitRank || !r.AllReduce || !r.CommDestroy) { dlclose(r.so); r.so = nullptr; }
        }
    }
    if (!r.so) { mk_set_error("masr_allreduce", "librccl.so could not be loaded (dlopen)"); return nullptr; }
    return &r;
}
int nccl_fail(const char* what, ncclResult_t rc) {
    Rccl* r = rccl();
    mk_set_error(what, r && r->GetErrorString ? r->GetErrorString(rc) : "RCCL error");
    return -1;
}
static_assert(sizeof(ncclUniqueId) == MASR_UNIQUE_ID_BYTES, "RCCL unique id size");
}  // namespace

constexpr int MAX_CHUNKS = 16;
struct masr_comm {
    ncclComm_t comm = nullptr;
    int rank = 0, world = 1;
    hipStream_t side = nullptr;
    hipEvent_t ready[MAX_CHUNKS] = {}, done = nullptr;
    bool issued = false, broken = false;
};
// a call failed between chunks: some chunks of the exchange are on the wire and some are not, the ranks no longer agree on the buffer's
// content -- the communicator is aborted (pending work is cancelled, peers see an error) and every later call on it fails
static int comm_break(masr_comm* c) {
    Rccl* r = rccl();
    c->broken = true;
    if (c->comm && r && r->CommAbort) { r->CommAbort(c->comm); c->comm = nullptr; }
    return -1;
}

int masr_allreduce_unique_id(char* id) {
    Rccl* r = rccl();
    if (!r) return -1;
    ncclUniqueId u;
    const ncclResult_t rc = r->GetUniqueId(&u);
    if (rc != ncclSuccess) return nccl_fail("ncclGetUniqueId", rc);
    memcpy(id, &u, sizeof u);
    return 0;
}

masr_comm* masr_allreduce_init(int rank, int world, const char* id) {
    Rccl* r = rccl();
    if (!r) return nullptr;
    if (world < 1 || rank < 0 || rank >= world || !id) { mk_set_error("masr_allreduce_init", "need 0 <= rank < world and a unique id"); return nullptr; }
    masr_comm* c = new masr_comm;
    c->rank = rank; c->world = world;
    ncclUniqueId u;
    memcpy(&u, id, sizeof u);
    const ncclResult_t rc = r->CommInitRank(&c->comm, world, u, rank);
    if (rc != ncclSuccess) { nccl_fail("ncclCommInitRank", rc); delete c; return nullptr; }
    bool ok = hipStreamCreateWithFlags(&c->side, hipStreamNonBlocking) == hipSuccess;
    for (auto& e : c->ready) ok = ok && hipEventCreateWithFlags(&e, hipEventDisableTiming) == hipSuccess;
    ok = ok && hipEventCreateWithFlags(&c->done, hipEventDisableTiming) == hipSuccess;
    if (!ok) { mk_set_error("masr_allreduce_init", "stream / event creation failed"); masr_allreduce_destroy(c); return nullptr; }
    return c;
}

void masr_allreduce_destroy(masr_comm* c) {
    if (!c) return;
    if (c->side && !c->broken) hipStreamSynchronize(c->side);
    Rccl* r = rccl();
    if (c->comm && r) r->CommDestroy(c->comm);
    for (auto& e : c->ready) if (e) hipEventDestroy(e);
    if (c->done) hipEventDestroy(c->done);
    if (c->side) hipStreamDestroy(c->side);
    delete c;
}

int masr_allreduce(masr_comm* c, float* buf, int64_t n, const float* norm, float max_norm, int nchunks, void* producer_stream) {
    Rccl* r = rccl();
    if (!r) return -1;
    if (!c || !buf || n <= 0) { mk_set_error("masr_allreduce", "null communicator / buffer or n <= 0"); return -1; }
    if (c->broken || !c->comm) { mk_set_error("masr_allreduce", "the communicator was aborted after an earlier failure"); return -1; }
    hipStream_t prod = (hipStream_t)producer_stream;
    nchunks = nchunks < 1 ? 1 : (nchunks > MAX_CHUNKS ? MAX_CHUNKS : nchunks);
    // chunk boundaries on 1024-float multiples (16-byte lanes of the scale pass, whole 4 KiB pages for the transport)
    const int64_t per = ((n + nchunks - 1) / nchunks + 1023) / 1024 * 1024;
    // (a second exchange on this communicator queues behind the first on the side stream; re-recording a chunk event is safe: a
    // stream wait binds to the record that was current when the wait was enqueued)
    for (int k = 0; k < nchunks; ++k) {
        const int64_t off = (int64_t)k * per;
        if (off >= n) break;
        const int64_t len = n - off < per ? n - off : per;
        if (norm) { if (mk_clip_scale(buf + off, len, norm, max_norm, prod) != 0) return comm_break(c); }
        if (hipEventRecord(c->ready[k], prod) != hipSuccess || hipStreamWaitEvent(c->side, c->ready[k], 0) != hipSuccess) {
            mk_set_error("masr_allreduce", "event ordering between the producer and the side stream failed"); return comm_break(c);
        }
        const ncclResult_t rc = r->AllReduce(buf + off, buf + off, (size_t)len, ncclFloat32, ncclSum, c->comm, c->side);
        if (rc != ncclSuccess) { nccl_fail("ncclAllReduce", rc); return comm_break(c); }
    }
    if (hipEventRecord(c->done, c->side) != hipSuccess) { mk_set_error("masr_allreduce", "hipEventRecord(done) failed"); return comm_break(c); }
    c->issued = true;
    return 0;
}

int masr_allreduce_wait(masr_comm* c, void* stream) {
    if (!c) { mk_set_error("masr_allreduce_wait", "null communicator"); return -1; }
    if (c->issued) HIP_CHECK_RET(hipStreamWaitEvent((hipStream_t)stream, c->done, 0));
    return 0;
}

// Host-side health check of the exchange (the waits above are device-side: nothing else would ever notice a collective that never
// completes).  Returns 0 = the last exchange has completed (or none was issued), 1 = still running (timeout_ms == 0: one poll), -1 = RCCL
// reports an asynchronous error or the wait ran past timeout_ms (the communicator is aborted: the caller must end the job).
int masr_allreduce_check(masr_comm* c, int timeout_ms) {
    if (!c) { mk_set_error("masr_allreduce_check", "null communicator"); return -1; }
    if (c->broken) { mk_set_error("masr_allreduce_check", "the communicator was aborted after an earlier failure"); return -1; }
    Rccl* r = rccl();
    for (int waited = 0;; ++waited) {
        if (r && r->CommGetAsyncError && c->comm) {
            ncclResult_t st = ncclSuccess;
            const ncclResult_t rc = r->CommGetAsyncError(c->comm, &st);
            if (rc != ncclSuccess || (st != ncclSuccess && st != ncclInProgress)) { nccl_fail("ncclCommGetAsyncError", rc != ncclSuccess ? rc : st); return comm_break(c); }
        }
        if (!c->issued) return 0;
        const hipError_t q = hipEventQuery(c->done);
        if (q == hipSuccess) return 0;
        if (q != hipErrorNotReady) { mk_set_error("masr_allreduce_check", hipGetErrorString(q)); return comm_break(c); }
        if (timeout_ms <= 0) return 1;
        if (waited >= timeout_ms) { mk_set_error("masr_allreduce_check", "the meta-gradient all-reduce did not complete within the time limit"); return comm_break(c); }
        struct timespec ts = {0, 1000000};
        nanosleep(&ts, nullptr);
    }
}
