// comm.cpp -- the one collective of the multi-GPU frame: the all-gather of the composed RGBA8 row strips,
// on RCCL over xGMI, behind the C ABI (include/crychic_hip.h "multi-GPU exchange").
//
// The reference is a single-GPU program (NodeMask 0, CRYCHIC.cpp:96,105; one device, Common/d3dApp.cpp:429-432): this
// file has no counterpart there.  What it completes is the back buffer CRYCHIC::Draw hands to Present
// (CRYCHIC.cpp:282-297) when the frame is rendered by several GPUs.
//
// RCCL is bound at run time (dlopen "librccl.so.1"): a process that already carries an RCCL (PyTorch ships one under the
// same soname) shares that instance instead of loading a second one; a plain C++ host gets /opt/rocm's.  Nothing here
// synchronises with the host: every exchange is enqueued on the caller's stream behind the strip's lighting pass.
#include <hip/hip_runtime.h>
#include <rccl/rccl.h>
#include <dlfcn.h>
#include <cstdio>
#include <cstring>
#include <mutex>
#include <new>
#include <vector>
#include "crychic_hip.h"
#include "internal.hpp"

static_assert(sizeof(ncclUniqueId) == CRYCHIC_COMM_ID_BYTES, "crychic_hip.h promises a 128-byte rendezvous id");

namespace {
using cry::fail;

struct Rccl {
    void* handle = nullptr;
    ncclResult_t (*GetUniqueId)(ncclUniqueId*) = nullptr;
    ncclResult_t (*CommInitRank)(ncclComm_t*, int, ncclUniqueId, int) = nullptr;
    ncclResult_t (*CommInitAll)(ncclComm_t*, int, const int*) = nullptr;
    ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
    ncclResult_t (*CommAbort)(ncclComm_t) = nullptr;
    ncclResult_t (*AllGather)(const void*, void*, size_t, ncclDataType_t, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*Broadcast)(const void*, void*, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*AllReduce)(const void*, void*, size_t, ncclDataType_t, ncclRedOp_t, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*GroupStart)() = nullptr;
    ncclResult_t (*GroupEnd)() = nullptr;
    ncclResult_t (*CommGetAsyncError)(ncclComm_t, ncclResult_t*) = nullptr;
    const char* (*GetErrorString)(ncclResult_t) = nullptr;
    const char* (*GetLastError)(ncclComm_t) = nullptr;
    char why[256] = "";
};

Rccl g_rccl;
std::once_flag g_rccl_once;

template <class F>
bool sym(void* h, const char* name, F& out)
{
    out = reinterpret_cast<F>(dlsym(h, name));
    return out != nullptr;
}

void load_rccl()
{
    Rccl& r = g_rccl;
    const char* names[] = { "librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1" };
    for (const char* n : names) {                  // an instance the process already carries wins (same soname)
        r.handle = dlopen(n, RTLD_NOW | RTLD_LOCAL | RTLD_NOLOAD);
        if (r.handle) break;
    }
    for (size_t i = 0; !r.handle && i < sizeof names / sizeof names[0]; ++i) r.handle = dlopen(names[i], RTLD_NOW | RTLD_LOCAL);
    if (!r.handle) {
        snprintf(r.why, sizeof r.why, "librccl.so.1 not found (%s)", dlerror());
        return;
    }
    const bool ok = sym(r.handle, "ncclGetUniqueId", r.GetUniqueId) && sym(r.handle, "ncclCommInitRank", r.CommInitRank) &&
                    sym(r.handle, "ncclCommInitAll", r.CommInitAll) && sym(r.handle, "ncclCommDestroy", r.CommDestroy) &&
                    sym(r.handle, "ncclCommAbort", r.CommAbort) && sym(r.handle, "ncclAllGather", r.AllGather) &&
                    sym(r.handle, "ncclBroadcast", r.Broadcast) && sym(r.handle, "ncclAllReduce", r.AllReduce) &&
                    sym(r.handle, "ncclGroupStart", r.GroupStart) && sym(r.handle, "ncclGroupEnd", r.GroupEnd) &&
                    sym(r.handle, "ncclCommGetAsyncError", r.CommGetAsyncError) && sym(r.handle, "ncclGetErrorString", r.GetErrorString);
    sym(r.handle, "ncclGetLastError", r.GetLastError);      // optional (diagnostics only)
    if (!ok) {
        snprintf(r.why, sizeof r.why, "the loaded RCCL lacks a required entry point");
        r.handle = nullptr;
    }
}

const Rccl* rccl()
{
    std::call_once(g_rccl_once, load_rccl);
    return g_rccl.handle ? &g_rccl : nullptr;
}

}  // namespace

struct crychic_comm {
    crychic_ctx* ctx;
    ncclComm_t nccl;
    int nranks, rank;
    uint32_t* barrier_word;     // device scratch of crychic_comm_barrier
    // crychic_draw_hot_path_shared with nparts > 1: the exchange runs on a stream of its own, ordered by events (created on first use)
    hipStream_t side;
    hipEvent_t partDone[CRYCHIC_MAX_EXCHANGE_PARTS];    // recorded on the caller's stream behind each lighting part
    hipEvent_t sideDone;                                // recorded on `side` behind the last part's exchange
    hipEvent_t sideFree;                                // recorded on the caller's stream at entry: `side` starts no earlier
};

namespace {

int comm_fail(const Rccl* r, const crychic_comm* c, const char* what, ncclResult_t e)
{
    const char* detail = (r->GetLastError && c) ? r->GetLastError(c->nccl) : "";
    return fail(CRYCHIC_E_COMM, "%s failed: %s%s%s", what, r->GetErrorString(e), detail && *detail ? " -- " : "", detail ? detail : "");
}

#define CRY_NCCL(r, c, expr)                                            \
    do {                                                                \
        ncclResult_t e_ = (expr);                                       \
        if (e_ != ncclSuccess) return comm_fail(r, c, #expr, e_);       \
    } while (0)

// Byte range [offset, offset + bytes) of every rank's strip inside a W x H RGBA8 frame; bounds == NULL -> crychic_strip_rows.
int strip_table(const crychic_comm* c, uint32_t W, uint32_t H, const uint32_t* bounds, std::vector<size_t>& off, std::vector<size_t>& len)
{
    if (W == 0 || H == 0 || (H & 1u)) return fail(CRYCHIC_E_INVALID_ARG, "frame %ux%u: H must be even and both non-zero", W, H);
    const size_t pitch = (size_t)W * 4u;
    off.resize((size_t)c->nranks);
    len.resize((size_t)c->nranks);
    uint64_t next = 0;
    for (int r = 0; r < c->nranks; ++r) {
        uint32_t row0, rows;
        if (bounds) { row0 = bounds[2 * r]; rows = bounds[2 * r + 1]; }
        else if (int rc = crychic_strip_rows(H, c->nranks, r, &row0, &rows)) return rc;
        if (row0 != next) return fail(CRYCHIC_E_INVALID_ARG, "strip %d starts at row %u, expected %llu (strips must tile the frame in rank order)", r, row0, (unsigned long long)next);
        next += rows;
        off[(size_t)r] = (size_t)row0 * pitch;
        len[(size_t)r] = (size_t)rows * pitch;
    }
    if (next != H) return fail(CRYCHIC_E_INVALID_ARG, "strips cover %llu rows of %u", (unsigned long long)next, H);
    return 0;
}

// The exchange proper, inside or outside an enclosing ncclGroup.  Equal strips: one in-place ncclAllGather (rank r's send
// buffer is its own slot of the receive buffer).  Ragged strips: one ncclBroadcast per strip, rooted at its owner, in
// place, issued as one group -- on the fully connected xGMI mesh every link then carries each strip exactly once.
int enqueue_gather(const Rccl* r, crychic_comm* c, uint8_t* frame, const std::vector<size_t>& off, const std::vector<size_t>& len, hipStream_t stream,
                   bool grouped)
{
    bool equal = true;
    for (int k = 0; k < c->nranks; ++k) equal = equal && len[(size_t)k] == len[0] && off[(size_t)k] == (size_t)k * len[0];
    if (equal) {
        if (len[0] == 0) return 0;
        CRY_NCCL(r, c, r->AllGather(frame + off[(size_t)c->rank], frame, len[0], ncclUint8, c->nccl, stream));
        return 0;
    }
    if (!grouped) CRY_NCCL(r, c, r->GroupStart());
    ncclResult_t first = ncclSuccess;
    for (int k = 0; k < c->nranks; ++k) {
        if (len[(size_t)k] == 0) continue;
        const ncclResult_t e = r->Broadcast(frame + off[(size_t)k], frame + off[(size_t)k], len[(size_t)k], ncclUint8, k, c->nccl, stream);
        if (e != ncclSuccess && first == ncclSuccess) first = e;
    }
    if (!grouped) {
        const ncclResult_t e = r->GroupEnd();          // always close the group, even after a failed enqueue
        if (first == ncclSuccess) first = e;
    }
    if (first != ncclSuccess) return comm_fail(r, c, "ncclBroadcast group", first);
    return 0;
}

void free_comm(crychic_comm* c)        // the HIP objects only; the RCCL communicator is the caller's business
{
    if (c->barrier_word) (void)hipFree(c->barrier_word);
    for (hipEvent_t e : c->partDone) if (e) (void)hipEventDestroy(e);
    if (c->sideDone) (void)hipEventDestroy(c->sideDone);
    if (c->sideFree) (void)hipEventDestroy(c->sideFree);
    if (c->side) (void)hipStreamDestroy(c->side);
    delete c;
}

int side_stream(crychic_comm* c)
{
    if (c->side) return 0;
    hipError_t e = hipStreamCreateWithFlags(&c->side, hipStreamNonBlocking);
    for (hipEvent_t& ev : c->partDone) if (e == hipSuccess) e = hipEventCreateWithFlags(&ev, hipEventDisableTiming);
    if (e == hipSuccess) e = hipEventCreateWithFlags(&c->sideDone, hipEventDisableTiming);
    if (e == hipSuccess) e = hipEventCreateWithFlags(&c->sideFree, hipEventDisableTiming);
    if (e != hipSuccess) return fail(CRYCHIC_E_HIP, "creating the exchange stream failed: %s", hipGetErrorString(e));
    return 0;
}

crychic_comm* new_comm(crychic_ctx* ctx, ncclComm_t nccl, int nranks, int rank)
{
    crychic_comm* c = new (std::nothrow) crychic_comm();
    if (!c) return nullptr;
    c->ctx = ctx;
    c->nccl = nccl;
    c->nranks = nranks;
    c->rank = rank;
    c->barrier_word = nullptr;
    c->side = nullptr;
    c->sideDone = c->sideFree = nullptr;
    for (hipEvent_t& e : c->partDone) e = nullptr;
    if (hipMalloc((void**)&c->barrier_word, sizeof(uint32_t)) != hipSuccess || hipMemset(c->barrier_word, 0, sizeof(uint32_t)) != hipSuccess) {
        if (c->barrier_word) (void)hipFree(c->barrier_word);
        delete c;
        return nullptr;
    }
    return c;
}

}  // namespace

extern "C" {

int crychic_comm_unique_id(uint8_t id[CRYCHIC_COMM_ID_BYTES])
{
    if (!id) return fail(CRYCHIC_E_INVALID_ARG, "id is null");
    const Rccl* r = rccl();
    if (!r) return fail(CRYCHIC_E_COMM, "RCCL unavailable: %s", g_rccl.why);
    ncclUniqueId u;
    CRY_NCCL(r, (crychic_comm*)nullptr, r->GetUniqueId(&u));
    memcpy(id, &u, sizeof u);
    return 0;
}

int crychic_comm_create(crychic_ctx* ctx, int nranks, int rank, const uint8_t id[CRYCHIC_COMM_ID_BYTES], crychic_comm** out)
{
    if (!out) return fail(CRYCHIC_E_INVALID_ARG, "out is null");
    *out = nullptr;
    if (!ctx || !id || nranks < 1 || rank < 0 || rank >= nranks) return fail(CRYCHIC_E_INVALID_ARG, "bad communicator request (nranks=%d rank=%d)", nranks, rank);
    const Rccl* r = rccl();
    if (!r) return fail(CRYCHIC_E_COMM, "RCCL unavailable: %s", g_rccl.why);
    hipError_t he = hipSetDevice(ctx->device);
    if (he != hipSuccess) return fail(CRYCHIC_E_HIP, "hipSetDevice(%d) failed: %s", ctx->device, hipGetErrorString(he));
    ncclUniqueId u;
    memcpy(&u, id, sizeof u);
    ncclComm_t nccl = nullptr;
    CRY_NCCL(r, (crychic_comm*)nullptr, r->CommInitRank(&nccl, nranks, u, rank));
    crychic_comm* c = new_comm(ctx, nccl, nranks, rank);
    if (!c) { r->CommAbort(nccl); return fail(CRYCHIC_E_HIP, "out of memory creating the communicator"); }
    *out = c;
    return 0;
}

int crychic_comm_create_all(crychic_ctx* const* ctxs, int nranks, crychic_comm** out)
{
    if (!ctxs || !out || nranks < 1 || nranks > 64) return fail(CRYCHIC_E_INVALID_ARG, "bad communicator request (nranks=%d)", nranks);
    for (int k = 0; k < nranks; ++k) { out[k] = nullptr; if (!ctxs[k]) return fail(CRYCHIC_E_INVALID_ARG, "context %d is null", k); }
    for (int a = 0; a < nranks; ++a)
        for (int b = a + 1; b < nranks; ++b)
            if (ctxs[a]->device == ctxs[b]->device) return fail(CRYCHIC_E_INVALID_ARG, "contexts %d and %d share device %d (one rank per GPU)", a, b, ctxs[a]->device);
    const Rccl* r = rccl();
    if (!r) return fail(CRYCHIC_E_COMM, "RCCL unavailable: %s", g_rccl.why);
    std::vector<int> devs((size_t)nranks);
    std::vector<ncclComm_t> comms((size_t)nranks, nullptr);
    for (int k = 0; k < nranks; ++k) devs[(size_t)k] = ctxs[k]->device;
    CRY_NCCL(r, (crychic_comm*)nullptr, r->CommInitAll(comms.data(), nranks, devs.data()));
    for (int k = 0; k < nranks; ++k) {
        (void)hipSetDevice(ctxs[k]->device);
        out[k] = new_comm(ctxs[k], comms[(size_t)k], nranks, k);
        if (!out[k]) {
            for (int j = 0; j < nranks; ++j) {
                if (out[j]) { free_comm(out[j]); out[j] = nullptr; }
                r->CommAbort(comms[(size_t)j]);
            }
            return fail(CRYCHIC_E_HIP, "out of memory creating the communicators");
        }
    }
    return 0;
}

void crychic_comm_destroy(crychic_comm* c)
{
    if (!c) return;
    const Rccl* r = rccl();
    (void)hipSetDevice(c->ctx->device);
    if (c->side) (void)hipStreamSynchronize(c->side);
    if (r && c->nccl) r->CommDestroy(c->nccl);
    free_comm(c);
}

int crychic_comm_abort(crychic_comm* c)
{
    if (!c) return fail(CRYCHIC_E_INVALID_ARG, "null communicator");
    const Rccl* r = rccl();
    if (!r) return fail(CRYCHIC_E_COMM, "RCCL unavailable: %s", g_rccl.why);
    if (c->nccl) { CRY_NCCL(r, c, r->CommAbort(c->nccl)); c->nccl = nullptr; }
    return 0;
}

int crychic_comm_rank(const crychic_comm* c) { return c ? c->rank : -1; }
int crychic_comm_size(const crychic_comm* c) { return c ? c->nranks : -1; }

int crychic_comm_async_error(crychic_comm* c)
{
    if (!c || !c->nccl) return fail(CRYCHIC_E_INVALID_ARG, "null or aborted communicator");
    const Rccl* r = rccl();
    if (!r) return fail(CRYCHIC_E_COMM, "RCCL unavailable: %s", g_rccl.why);
    ncclResult_t async = ncclSuccess;
    CRY_NCCL(r, c, r->CommGetAsyncError(c->nccl, &async));
    if (async != ncclSuccess && async != ncclInProgress) return comm_fail(r, c, "asynchronous RCCL operation", async);
    return 0;
}

int crychic_allgather_frame(crychic_comm* c, uint8_t* frame_rgba8_dev, uint32_t W, uint32_t H, const uint32_t* bounds, void* stream)
{
    if (!c || !c->nccl || !frame_rgba8_dev) return fail(CRYCHIC_E_INVALID_ARG, "null communicator / frame");
    const Rccl* r = rccl();
    if (!r) return fail(CRYCHIC_E_COMM, "RCCL unavailable: %s", g_rccl.why);
    hipError_t he = hipSetDevice(c->ctx->device);
    if (he != hipSuccess) return fail(CRYCHIC_E_HIP, "hipSetDevice(%d) failed: %s", c->ctx->device, hipGetErrorString(he));
    std::vector<size_t> off, len;
    if (int rc = strip_table(c, W, H, bounds, off, len)) return rc;
    return enqueue_gather(r, c, frame_rgba8_dev, off, len, (hipStream_t)stream, false);
}

int crychic_allgather_frame_all(crychic_comm* const* comms, int nranks, uint8_t* const* frames_rgba8_dev, uint32_t W, uint32_t H,
                                const uint32_t* bounds, void* const* streams)
{
    if (!comms || !frames_rgba8_dev || nranks < 1) return fail(CRYCHIC_E_INVALID_ARG, "null argument");
    const Rccl* r = rccl();
    if (!r) return fail(CRYCHIC_E_COMM, "RCCL unavailable: %s", g_rccl.why);
    for (int k = 0; k < nranks; ++k)
        if (!comms[k] || !comms[k]->nccl || !frames_rgba8_dev[k] || comms[k]->nranks != nranks || comms[k]->rank != k)
            return fail(CRYCHIC_E_INVALID_ARG, "communicator %d is null, aborted or not rank %d of %d", k, k, nranks);
    std::vector<size_t> off, len;
    if (int rc = strip_table(comms[0], W, H, bounds, off, len)) return rc;
    // one thread drives every rank: all enqueues sit inside one group, so no rank's call waits for a peer's
    CRY_NCCL(r, comms[0], r->GroupStart());
    int rc = 0;
    for (int k = 0; k < nranks && rc == 0; ++k) {
        (void)hipSetDevice(comms[k]->ctx->device);
        rc = enqueue_gather(r, comms[k], frames_rgba8_dev[k], off, len, streams ? (hipStream_t)streams[k] : nullptr, true);
    }
    const ncclResult_t e = r->GroupEnd();
    if (rc) return rc;
    if (e != ncclSuccess) return comm_fail(r, comms[0], "ncclGroupEnd", e);
    return 0;
}

namespace {
struct SharedDraw {
    const Rccl* r;
    crychic_comm* c;
    uint8_t* frame;
    size_t pitch;
    uint32_t nparts;
    const uint32_t* row0;       // per rank
    const uint32_t* rows;
    hipStream_t main;
};

// Behind lighting part p of this rank's strip: part p of EVERY rank's strip travels (one group of in-place ncclBroadcasts, each
// rooted at the strip's owner), on the side stream, while the caller's stream goes on to light part p + 1.
int exchange_part(void* user, uint32_t p, uint32_t, uint32_t)
{
    SharedDraw& d = *static_cast<SharedDraw*>(user);
    crychic_comm* c = d.c;
    hipError_t he = hipEventRecord(c->partDone[p], d.main);
    if (he == hipSuccess) he = hipStreamWaitEvent(c->side, c->partDone[p], 0);
    if (he != hipSuccess) return fail(CRYCHIC_E_HIP, "ordering the exchange behind lighting part %u failed: %s", p, hipGetErrorString(he));
    CRY_NCCL(d.r, c, d.r->GroupStart());
    ncclResult_t first = ncclSuccess;
    for (int k = 0; k < c->nranks; ++k) {
        const uint32_t pairs = d.rows[k] / 2u, per = pairs / d.nparts;          // the same cut api.cpp's hot_path_parts makes
        const uint32_t r0 = d.row0[k] + 2u * per * p;
        const uint32_t r1 = (p + 1u == d.nparts) ? d.row0[k] + d.rows[k] : r0 + 2u * per;
        if (r1 == r0) continue;
        uint8_t* at = d.frame + (size_t)r0 * d.pitch;
        const ncclResult_t e = d.r->Broadcast(at, at, (size_t)(r1 - r0) * d.pitch, ncclUint8, k, c->nccl, c->side);
        if (e != ncclSuccess && first == ncclSuccess) first = e;
    }
    const ncclResult_t e = d.r->GroupEnd();
    if (first == ncclSuccess) first = e;
    if (first != ncclSuccess) return comm_fail(d.r, c, "ncclBroadcast group (exchange part)", first);
    return 0;
}
}  // namespace

int crychic_draw_hot_path_shared(crychic_comm* c, const crychic_ssao_constants* ssaoCB, const crychic_pass_constants* passCB,
                                 const crychic_frame_desc* f, const uint32_t* bounds, uint32_t nparts, void* stream_)
{
    if (!c || !c->nccl || !f) return fail(CRYCHIC_E_INVALID_ARG, "null communicator / frame descriptor");
    const Rccl* r = rccl();
    if (!r) return fail(CRYCHIC_E_COMM, "RCCL unavailable: %s", g_rccl.why);
    if (nparts == 0 || nparts > CRYCHIC_MAX_EXCHANGE_PARTS) return fail(CRYCHIC_E_INVALID_ARG, "nparts %u (1 .. %d)", nparts, CRYCHIC_MAX_EXCHANGE_PARTS);
    hipStream_t stream = (hipStream_t)stream_;
    std::vector<size_t> off, len;
    if (int rc = strip_table(c, f->W, f->H, bounds, off, len)) return rc;
    const size_t pitch = (size_t)f->W * 4u;
    if ((size_t)f->row0 * pitch != off[(size_t)c->rank] || (size_t)f->rows * pitch != len[(size_t)c->rank])
        return fail(CRYCHIC_E_INVALID_ARG, "the frame descriptor's strip [%u,+%u) is not rank %d's strip of the plan", f->row0, f->rows, c->rank);
    if (nparts == 1u) {                         // nothing to overlap: the strip, then the one-call exchange, on the caller's stream
        if (int rc = cry::hot_path_parts(c->ctx, ssaoCB, passCB, f, stream, 1u, nullptr, nullptr)) return rc;
        return enqueue_gather(r, c, f->out_rgba8_dev, off, len, stream, false);
    }
    hipError_t he = hipSetDevice(c->ctx->device);
    if (he != hipSuccess) return fail(CRYCHIC_E_HIP, "hipSetDevice(%d) failed: %s", c->ctx->device, hipGetErrorString(he));
    if (int rc = side_stream(c)) return rc;
    std::vector<uint32_t> row0((size_t)c->nranks), rows((size_t)c->nranks);
    for (int k = 0; k < c->nranks; ++k) { row0[(size_t)k] = (uint32_t)(off[(size_t)k] / pitch); rows[(size_t)k] = (uint32_t)(len[(size_t)k] / pitch); }
    SharedDraw d{ r, c, f->out_rgba8_dev, pitch, nparts, row0.data(), rows.data(), stream };
    // the peers' rows of this buffer may still be read by work enqueued earlier on the caller's stream (a copy-out of the previous
    // frame): the side stream writes them no earlier than this call's position in that stream
    he = hipEventRecord(c->sideFree, stream);
    if (he == hipSuccess) he = hipStreamWaitEvent(c->side, c->sideFree, 0);
    if (he != hipSuccess) return fail(CRYCHIC_E_HIP, "ordering the exchange stream failed: %s", hipGetErrorString(he));
    const int rc = cry::hot_path_parts(c->ctx, ssaoCB, passCB, f, stream, nparts, exchange_part, &d);
    // whatever happened above, the caller's stream joins the side stream again: the frame is complete behind this call
    he = hipEventRecord(c->sideDone, c->side);
    if (he == hipSuccess) he = hipStreamWaitEvent(stream, c->sideDone, 0);
    if (rc) return rc;
    if (he != hipSuccess) return fail(CRYCHIC_E_HIP, "joining the exchange stream failed: %s", hipGetErrorString(he));
    return 0;
}

int crychic_comm_barrier(crychic_comm* c, void* stream)
{
    if (!c || !c->nccl) return fail(CRYCHIC_E_INVALID_ARG, "null or aborted communicator");
    const Rccl* r = rccl();
    if (!r) return fail(CRYCHIC_E_COMM, "RCCL unavailable: %s", g_rccl.why);
    hipError_t he = hipSetDevice(c->ctx->device);
    if (he != hipSuccess) return fail(CRYCHIC_E_HIP, "hipSetDevice(%d) failed: %s", c->ctx->device, hipGetErrorString(he));
    CRY_NCCL(r, c, r->AllReduce(c->barrier_word, c->barrier_word, 1, ncclUint32, ncclSum, c->nccl, (hipStream_t)stream));
    return 0;
}

}  // extern "C"
