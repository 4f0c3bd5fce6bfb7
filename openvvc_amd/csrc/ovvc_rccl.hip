// ovvc_rccl.hip -- the multi-process exchange of the stream driver over RCCL (include/ovvc_hip.h, "RCCL transport").
//
// One process per GPU (north_star: "reference pictures pushed over xGMI via RCCL point-to-point only when a frame's refs live on
// another GPU"): a finished reference picture goes to exactly the ranks whose pictures list it, as ncclSend / ncclRecv of its three
// planes inside ONE ncclGroup per picture, on a stream of the transport's own, from the stream driver's communication thread
// (ovvc_stream.c: comm_thread walks the stream in decoding order on every rank, so each pair of ranks issues its sends and receives
// in the same order; openvvc_amd/gop.py check_programs proves the schedules deadlock-free under rendezvous semantics).  No
// collective.  The reference's analogue is the frame thread's wait on a picture decoded by another thread
// (ovdpb_synchro_ref_decoded_ctus, dpb.c:1242-1270; ovdec.c:188-248 deals the pictures to the sub-decoders).
//
// librccl.so is opened at run time (dlopen): a one-GPU decoder does not load it, and the library has no link-time dependency on it.
#include <dlfcn.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <hip/hip_runtime.h>
#include <pthread.h>
#include "ovvc_hip.h"

// The few types and constants of rccl.h this file needs, declared here: the library must build on a ROCm install without the RCCL
// development headers -- the functions themselves come from dlsym.  (ncclUniqueId is 128 opaque bytes, ncclComm_t an
// opaque pointer, the enumerators below are fixed by NCCL's ABI: nccl.h "ncclResult_t", "ncclDataType_t".)
typedef struct ncclComm *ncclComm_t;
typedef struct { char internal[128]; } ncclUniqueId;
typedef int ncclResult_t;
typedef int ncclDataType_t;
enum { ncclSuccess = 0 };
enum { ncclUint8 = 1 };

struct ovhip_rccl {
    void *lib;
    ncclComm_t comm;
    hipStream_t stream;
    hipEvent_t ev_done;            // behind a picture's group: the comm thread waits for THIS, not for the stream (ovvc_common.hip.h: hipStreamSynchronize
                                   // from several threads at once is what the runtime does badly; an event wait costs nothing)
    int device, rank, world;
    ncclResult_t (*CommInitRank)(ncclComm_t *, int, ncclUniqueId, int);
    ncclResult_t (*CommDestroy)(ncclComm_t);
    ncclResult_t (*Send)(const void *, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t);
    ncclResult_t (*Recv)(void *, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t);
    ncclResult_t (*GroupStart)(void);
    ncclResult_t (*GroupEnd)(void);
    const char *(*GetErrorString)(ncclResult_t);
    ovhip_stream_xfer xfer;
    uint64_t bytes_sent, bytes_received, n_sent, n_received;
    char err[192];
};

// one handle per process, opened on first use and kept (RCCL keeps threads and device state of its own: it is never unloaded)
static void *open_rccl(char *err, size_t cap)
{
    static void *g_handle;
    static pthread_mutex_t g_mtx = PTHREAD_MUTEX_INITIALIZER;
    static const char *names[] = { "librccl.so", "librccl.so.1", "/opt/rocm/lib/librccl.so" };
    pthread_mutex_lock(&g_mtx);
    for (unsigned i = 0; !g_handle && i < sizeof(names) / sizeof(names[0]); ++i) g_handle = dlopen(names[i], RTLD_NOW | RTLD_LOCAL);
    if (!g_handle) snprintf(err, cap, "librccl.so: %s", dlerror());
    void *h = g_handle;
    pthread_mutex_unlock(&g_mtx);
    return h;
}

extern "C" int ovhip_rccl_unique_id(uint8_t out[128])
{
    if (!out) return OVHIP_EINVAL;
    char err[192];
    void *h = open_rccl(err, sizeof(err));
    if (!h) return OVHIP_ENODEV;
    auto get = (ncclResult_t (*)(ncclUniqueId *))dlsym(h, "ncclGetUniqueId");
    ncclUniqueId id;
    static_assert(sizeof(id) == 128, "ncclUniqueId");
    if (!get || get(&id) != ncclSuccess) return OVHIP_ELAUNCH;
    memcpy(out, &id, 128);
    return OVHIP_OK;                      // (the handle stays open: the library is in use from here on)
}

static int rfail(ovhip_rccl *r, const char *what, ncclResult_t e, hipError_t h)
{
    snprintf(r->err, sizeof(r->err), "%s: %s", what, e != ncclSuccess ? (r->GetErrorString ? r->GetErrorString(e) : "rccl error") : hipGetErrorString(h));
    return OVHIP_ELAUNCH;
}

static size_t plane_bytes(const ovhip_pic *p, int k) { return k ? (size_t)p->stride_c * (size_t)(p->h / 2) * 2 : (size_t)p->stride_y * (size_t)p->h * 2; }

// the three planes of one picture in one group; dir 0: send to peer, 1: receive from peer, 2 (self test): send src / receive dst
static int exchange(ovhip_rccl *r, const ovhip_pic *src, const ovhip_pic *dst, int peer, int dir)
{
    if (hipSetDevice(r->device) != hipSuccess) return rfail(r, "hipSetDevice", ncclSuccess, hipErrorInvalidDevice);
    ncclResult_t e = r->GroupStart();
    if (e != ncclSuccess) return rfail(r, "ncclGroupStart", e, hipSuccess);
    for (int k = 0; k < 3 && e == ncclSuccess; ++k) {
        if (dir != 1) { const void *p = k == 0 ? (const void *)src->y : k == 1 ? (const void *)src->cb : (const void *)src->cr; e = r->Send(p, plane_bytes(src, k), ncclUint8, peer, r->comm, r->stream); }
        if (dir != 0 && e == ncclSuccess) { void *p = k == 0 ? (void *)dst->y : k == 1 ? (void *)dst->cb : (void *)dst->cr; e = r->Recv(p, plane_bytes(dst, k), ncclUint8, peer, r->comm, r->stream); }
    }
    const ncclResult_t e2 = r->GroupEnd();
    if (e != ncclSuccess || e2 != ncclSuccess) return rfail(r, "ncclSend / ncclRecv", e != ncclSuccess ? e : e2, hipSuccess);
    // the callbacks of ovhip_stream_xfer are synchronous by contract (the driver's comm thread publishes the picture to the device DPB when
    // the call returns): an event behind the group, waited for on the host
    hipError_t h = hipSuccess;
    if (!r->ev_done) h = hipEventCreateWithFlags(&r->ev_done, hipEventDisableTiming);
    if (h == hipSuccess) h = hipEventRecord(r->ev_done, r->stream);
    if (h == hipSuccess) h = hipEventSynchronize(r->ev_done);
    if (h != hipSuccess) return rfail(r, "event behind the picture's group (transport)", ncclSuccess, h);
    return OVHIP_OK;
}

static int cb_send(void *user, uint32_t idx, const ovhip_pic *pic, int dst_rank)
{
    (void)idx;
    ovhip_rccl *r = (ovhip_rccl *)user;
    const int q = exchange(r, pic, nullptr, dst_rank, 0);
    if (q == OVHIP_OK) { r->n_sent++; r->bytes_sent += plane_bytes(pic, 0) + 2 * plane_bytes(pic, 1); }
    return q;
}

static int cb_recv(void *user, uint32_t idx, const ovhip_pic *pic, int src_rank)
{
    (void)idx;
    ovhip_rccl *r = (ovhip_rccl *)user;
    const int q = exchange(r, nullptr, pic, src_rank, 1);
    if (q == OVHIP_OK) { r->n_received++; r->bytes_received += plane_bytes(pic, 0) + 2 * plane_bytes(pic, 1); }
    return q;
}

extern "C" int ovhip_rccl_create(ovhip_rccl **out, const uint8_t unique_id[128], int rank, int world, int hip_device)
{
    if (!out || !unique_id || rank < 0 || world < 1 || rank >= world) return OVHIP_EINVAL;
    *out = nullptr;
    ovhip_rccl *r = (ovhip_rccl *)calloc(1, sizeof(*r));
    if (!r) return OVHIP_ENOMEM;
    r->device = hip_device; r->rank = rank; r->world = world;
    r->lib = open_rccl(r->err, sizeof(r->err));
    if (!r->lib) { fprintf(stderr, "ovhip_rccl_create: %s\n", r->err); free(r); return OVHIP_ENODEV; }
#define SYM(field, name) do { *(void **)&r->field = dlsym(r->lib, name); if (!r->field) { fprintf(stderr, "ovhip_rccl_create: %s missing in librccl\n", name); free(r); return OVHIP_ENODEV; } } while (0)
    SYM(CommInitRank, "ncclCommInitRank"); SYM(CommDestroy, "ncclCommDestroy"); SYM(Send, "ncclSend"); SYM(Recv, "ncclRecv");
    SYM(GroupStart, "ncclGroupStart"); SYM(GroupEnd, "ncclGroupEnd"); SYM(GetErrorString, "ncclGetErrorString");
#undef SYM
    if (hipSetDevice(hip_device) != hipSuccess || hipStreamCreateWithFlags(&r->stream, hipStreamNonBlocking) != hipSuccess) { free(r); return OVHIP_ENODEV; }
    ncclUniqueId id;
    memcpy(&id, unique_id, 128);
    const ncclResult_t e = r->CommInitRank(&r->comm, world, id, rank);
    if (e != ncclSuccess) {
        fprintf(stderr, "ovhip_rccl_create: ncclCommInitRank(rank %d of %d, device %d): %s\n", rank, world, hip_device, r->GetErrorString(e));
        (void)hipStreamDestroy(r->stream); free(r);
        return OVHIP_ELAUNCH;
    }
    r->xfer.user = r; r->xfer.send = cb_send; r->xfer.recv = cb_recv;
    *out = r;
    return OVHIP_OK;
}

extern "C" void ovhip_rccl_destroy(ovhip_rccl *r)
{
    if (!r) return;
    (void)hipSetDevice(r->device);
    if (r->comm) (void)r->CommDestroy(r->comm);
    if (r->ev_done) (void)hipEventDestroy(r->ev_done);
    (void)hipStreamDestroy(r->stream);
    free(r);
}

extern "C" const ovhip_stream_xfer *ovhip_rccl_xfer(ovhip_rccl *r) { return r ? &r->xfer : nullptr; }
extern "C" const char *ovhip_rccl_last_error(const ovhip_rccl *r) { return r ? r->err : "no transport"; }

extern "C" int ovhip_rccl_stats(const ovhip_rccl *r, uint64_t out[4])
{
    if (!r || !out) return OVHIP_EINVAL;
    out[0] = r->n_sent; out[1] = r->bytes_sent; out[2] = r->n_received; out[3] = r->bytes_received;
    return OVHIP_OK;
}

// one rank sends a picture to ITSELF and receives it into another (send and receive in the same group): the whole machinery --
// dlopen, communicator, grouped point-to-point of the three planes, stream -- on a box with one GPU
extern "C" int ovhip_rccl_self_exchange(ovhip_rccl *r, const ovhip_pic *src, const ovhip_pic *dst)
{
    if (!r || !src || !dst || src->w != dst->w || src->h != dst->h) return OVHIP_EINVAL;
    return exchange(r, src, dst, r->rank, 2);
}
