// ovvc_dpb_hip.hip -- HIP back-end of the device DPB (ovvc_dpb.c): device pictures and the transfer of a finished reference
// picture to another device of the same process.
//
// Transfer = hipMemcpyPeerAsync on a copy stream of the DESTINATION device (peer access enabled once per ordered pair: over xGMI
// on an MI355X node the copy engine then moves the picture link to link, 24.9 MB at 4K ~= 0.16 ms per hop), one event per copy;
// the reader waits for the event on the host after its own uploads are under way (ovhip_frame_submit), the same way it waits for
// a reference picture decoded on its own device.  The source is complete when the copy is issued (a picture is published only
// after ovhip_job_wait), so no cross-device stream dependency is needed.  No collective: the path has no exchange step other than
// "this picture, to the devices whose queued pictures list it" (SURVEY 8e).
#include "ovvc_common.hip.h"
#include "ovvc_dpb_priv.h"
#include <stdlib.h>

namespace {

struct HipDpb {
    int n;
    int dev[OVHIP_MAX_DEVICES];
    hipStream_t copy_stream[OVHIP_MAX_DEVICES];
    unsigned char peer[OVHIP_MAX_DEVICES][OVHIP_MAX_DEVICES];      // peer access dst <- src enabled (by logical device)
};

// The DPB's memory calls run on whatever thread released / began a picture -- often a frame thread bound to ANOTHER device: they
// leave the caller's current device as they found it.
struct DeviceScope {
    int prev = -1;
    DeviceScope() { if (hipGetDevice(&prev) != hipSuccess) prev = -1; }
    ~DeviceScope() { if (prev >= 0) (void)hipSetDevice(prev); }
};

size_t plane_bytes(int32_t w, int32_t h, size_t *ysz, size_t *csz)
{
    *ysz = ((size_t)w * h * 2 + 255) & ~(size_t)255;
    *csz = ((size_t)(w / 2) * (h / 2) * 2 + 255) & ~(size_t)255;
    return *ysz + 2 * *csz;
}

int hd_stream(HipDpb *u, int k, hipStream_t *s)
{
    if (hipSetDevice(u->dev[k]) != hipSuccess) return OVHIP_ENODEV;
    if (!u->copy_stream[k] && hipStreamCreateWithFlags(&u->copy_stream[k], hipStreamNonBlocking) != hipSuccess) return OVHIP_ENODEV;
    *s = u->copy_stream[k];
    return OVHIP_OK;
}

int hd_pic_alloc(void *user, int k, int32_t w, int32_t h, ovhip_pic *pic)
{
    DeviceScope keep;
    HipDpb *u = (HipDpb *)user;
    if ((w & 1) || (h & 1)) return OVHIP_EINVAL;
    hipStream_t s;
    int r = hd_stream(u, k, &s);
    if (r != OVHIP_OK) return r;
    size_t ysz, csz;
    const size_t total = plane_bytes(w, h, &ysz, &csz);
    void *base = nullptr;
    if (hipMalloc(&base, total) != hipSuccess) return OVHIP_ENOMEM;
    // zero-filled and complete on return, like ovhip_pic_alloc: no sample may carry the ordered pass's hand-over bit in
    if (hipMemsetAsync(base, 0, total, s) != hipSuccess || hipStreamSynchronize(s) != hipSuccess) { (void)hipFree(base); return OVHIP_ELAUNCH; }
    memset(pic, 0, sizeof(*pic));
    pic->y = (uint16_t *)base;
    pic->cb = (uint16_t *)((char *)base + ysz);
    pic->cr = (uint16_t *)((char *)base + ysz + csz);
    pic->w = w; pic->h = h; pic->stride_y = w; pic->stride_c = w / 2;
    return OVHIP_OK;
}

void hd_pic_free(void *user, int k, ovhip_pic *pic)
{
    DeviceScope keep;
    HipDpb *u = (HipDpb *)user;
    if (hipSetDevice(u->dev[k]) == hipSuccess && pic->y) (void)hipFree(pic->y);
    memset(pic, 0, sizeof(*pic));
}

int hd_pic_clear(void *user, int k, const ovhip_pic *pic)
{
    DeviceScope keep;
    HipDpb *u = (HipDpb *)user;
    hipStream_t s;
    int r = hd_stream(u, k, &s);
    if (r != OVHIP_OK) return r;
    size_t ysz, csz;
    const size_t total = plane_bytes(pic->w, pic->h, &ysz, &csz);
    if (hipMemsetAsync(pic->y, 0, total, s) != hipSuccess || hipStreamSynchronize(s) != hipSuccess) return OVHIP_ELAUNCH;
    return OVHIP_OK;
}

int hd_copy_start(void *user, int kd, const ovhip_pic *dst, int ks, const ovhip_pic *src, void **event)
{
    DeviceScope keep;
    HipDpb *u = (HipDpb *)user;
    hipStream_t s;
    int r = hd_stream(u, kd, &s);
    if (r != OVHIP_OK) return r;
    if (u->dev[kd] != u->dev[ks] && !u->peer[kd][ks]) {
        // direct access dst <- src over xGMI; without it the runtime stages the copy through host memory
        int can = 0;
        if (hipDeviceCanAccessPeer(&can, u->dev[kd], u->dev[ks]) == hipSuccess && can) {
            hipError_t e = hipDeviceEnablePeerAccess(u->dev[ks], 0);
            if (e != hipSuccess && e != hipErrorPeerAccessAlreadyEnabled) return OVHIP_ENODEV;
            (void)hipGetLastError();
        }
        u->peer[kd][ks] = 1;
    }
    size_t ysz, csz;
    const size_t total = plane_bytes(src->w, src->h, &ysz, &csz);            // both pictures are ONE allocation of this layout
    hipEvent_t ev;
    if (hipEventCreateWithFlags(&ev, hipEventDisableTiming) != hipSuccess) return OVHIP_ENODEV;
    hipError_t e = u->dev[kd] == u->dev[ks] ? hipMemcpyAsync(dst->y, src->y, total, hipMemcpyDeviceToDevice, s)
                                           : hipMemcpyPeerAsync(dst->y, u->dev[kd], src->y, u->dev[ks], total, s);
    if (e == hipSuccess) e = hipEventRecord(ev, s);
    if (e != hipSuccess) { (void)hipEventDestroy(ev); return OVHIP_ELAUNCH; }
    *event = (void *)ev;
    return OVHIP_OK;
}

int hd_copy_wait(void *user, int kd, void *event)
{
    DeviceScope keep;
    HipDpb *u = (HipDpb *)user;
    if (hipSetDevice(u->dev[kd]) != hipSuccess) return OVHIP_ENODEV;
    return hipEventSynchronize((hipEvent_t)event) == hipSuccess ? OVHIP_OK : OVHIP_ELAUNCH;
}

void hd_copy_done(void *user, int kd, void *event)
{
    DeviceScope keep;
    HipDpb *u = (HipDpb *)user;
    if (hipSetDevice(u->dev[kd]) == hipSuccess) (void)hipEventDestroy((hipEvent_t)event);
}

// row progress: the producer's events (recorded on its job's stream; any thread may ask)
int hd_event_query(void *user, int k, void *event)
{
    (void)user; (void)k;
    const hipError_t e = hipEventQuery((hipEvent_t)event);
    if (e == hipSuccess) return 1;
    (void)hipGetLastError();
    return e == hipErrorNotReady ? 0 : OVHIP_ELAUNCH;
}

int hd_event_wait(void *user, int k, void *event)
{
    (void)user; (void)k;
    return hipEventSynchronize((hipEvent_t)event) == hipSuccess ? OVHIP_OK : OVHIP_ELAUNCH;
}

} // namespace

extern "C" int ovhip_dpb_hip_ops_(const int *devices, int n_devices, ovhip_dpb_ops *ops, void **user)
{
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess || n <= 0) return OVHIP_ENODEV;        // no CPU fallback
    for (int i = 0; i < n_devices; ++i) if (devices[i] < 0 || devices[i] >= n) return OVHIP_ENODEV;
    HipDpb *u = (HipDpb *)calloc(1, sizeof(*u));
    if (!u) return OVHIP_ENOMEM;
    u->n = n_devices;
    for (int i = 0; i < n_devices; ++i) u->dev[i] = devices[i];
    ops->user = u;
    ops->pic_alloc = hd_pic_alloc; ops->pic_free = hd_pic_free; ops->pic_clear = hd_pic_clear;
    ops->copy_start = hd_copy_start; ops->copy_wait = hd_copy_wait; ops->copy_done = hd_copy_done;
    ops->event_query = hd_event_query; ops->event_wait = hd_event_wait;
    *user = u;
    return OVHIP_OK;
}

extern "C" void ovhip_dpb_hip_ops_free_(void *user)
{
    HipDpb *u = (HipDpb *)user;
    if (!u) return;
    for (int k = 0; k < u->n; ++k)
        if (u->copy_stream[k] && hipSetDevice(u->dev[k]) == hipSuccess) (void)hipStreamDestroy(u->copy_stream[k]);
    free(u);
}
