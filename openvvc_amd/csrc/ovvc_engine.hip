// ovvc_engine.hip -- context, device-memory and picture plumbing of the C ABI (include/ovvc_hip.h).
// gfx950 only; every entry point fails loudly (OVHIP_ENODEV + ovhip_last_error) when the HIP
// runtime or device is missing -- there is no CPU fallback in the product library.
#include "ovvc_common.hip.h"
#include <stdlib.h>

#include <pthread.h>

// Streams are pooled per device, never destroyed while the process lives: an event keeps a reference to the stream it was last
// recorded on, and on this runtime hipEventSynchronize on an event whose stream has been destroyed fails ("operation not permitted
// when stream is capturing", tools/micro/event_after_stream_destroy.hip) -- a job's completion event must survive the frame
// thread (context) that last flushed it.  Creating a stream also costs milliseconds; a frame thread pool that comes and goes
// (one per sequence) gets them back at once.
namespace {
enum { POOL_DEVS = 64, POOL_CAP = 512 };
pthread_mutex_t g_pool_mtx = PTHREAD_MUTEX_INITIALIZER;
hipStream_t g_pool[POOL_DEVS][POOL_CAP];
int g_pool_n[POOL_DEVS];

hipError_t stream_get(int device, hipStream_t *out)
{
    pthread_mutex_lock(&g_pool_mtx);
    if (device < POOL_DEVS && g_pool_n[device] > 0) { *out = g_pool[device][--g_pool_n[device]]; pthread_mutex_unlock(&g_pool_mtx); return hipSuccess; }
    pthread_mutex_unlock(&g_pool_mtx);
    return hipStreamCreateWithFlags(out, hipStreamNonBlocking);
}

void stream_put(int device, hipStream_t s)
{
    pthread_mutex_lock(&g_pool_mtx);
    const bool kept = device < POOL_DEVS && g_pool_n[device] < POOL_CAP;
    if (kept) g_pool[device][g_pool_n[device]++] = s;
    pthread_mutex_unlock(&g_pool_mtx);
    if (!kept) (void)hipStreamDestroy(s);
}
} // namespace

__global__ void k_spin(unsigned long long ticks, unsigned *sink)
{
    const unsigned long long t0 = wall_clock64();
    while (wall_clock64() - t0 < ticks) __builtin_amdgcn_s_sleep(32);
    if (sink && ticks == ~0ull) *sink = 1;
}
__global__ void k_nop(unsigned *sink) { if (sink && threadIdx.x == 12345) *sink = 1; }

extern "C" {

int ovhip_abi_version(void) { return OVHIP_ABI_VERSION; }

/* Do the streams of two contexts share a hardware queue?  The runtime spreads its streams over GPU_MAX_HW_QUEUES (4) hardware
 * queues, and a hardware queue runs the packets of its streams in order: a kernel of several milliseconds on one stream (an I
 * picture's ordered pass) holds up every stream that shares its queue.  Nothing in the API says which stream got which queue, so it
 * is measured: a 1.5 ms spin kernel on a, an empty kernel on b right behind it -- if b's kernel only completes when a's has, they
 * share.  1: share, 0: do not, < 0: error.  Both contexts idle, same device. */
int ovhip_ctx_shares_queue(ovhip_ctx *a, ovhip_ctx *b)
{
    if (!a || !b || a->device != b->device) return OVHIP_EINVAL;
    if (a->stream == b->stream) return 1;
    OV_DEVICE(a);
    OV_HIP(a, hipStreamSynchronize(a->stream));
    OV_HIP(a, hipStreamSynchronize(b->stream));
    hipEvent_t ea, eb;
    OV_HIP(a, hipEventCreateWithFlags(&ea, hipEventDisableTiming));
    OV_HIP(a, hipEventCreateWithFlags(&eb, hipEventDisableTiming));
    hipLaunchKernelGGL(k_nop, dim3(1), dim3(64), 0, a->stream, (unsigned *)nullptr);       // (first launches of a process also load the code object)
    hipLaunchKernelGGL(k_nop, dim3(1), dim3(64), 0, b->stream, (unsigned *)nullptr);
    (void)hipStreamSynchronize(a->stream); (void)hipStreamSynchronize(b->stream);
    hipLaunchKernelGGL(k_spin, dim3(1), dim3(64), 0, a->stream, 150000ull, (unsigned *)nullptr);     // wall_clock64: 100 MHz -> 1.5 ms
    (void)hipEventRecord(ea, a->stream);
    hipLaunchKernelGGL(k_nop, dim3(1), dim3(64), 0, b->stream, (unsigned *)nullptr);
    (void)hipEventRecord(eb, b->stream);
    hipError_t e = hipEventSynchronize(eb);
    const int shared = e == hipSuccess && hipEventQuery(ea) == hipSuccess;       // a's spin was over when b's empty kernel completed
    (void)hipEventSynchronize(ea);
    (void)hipGetLastError();
    (void)hipEventDestroy(ea); (void)hipEventDestroy(eb);
    return e == hipSuccess ? shared : ov_fail(a, OVHIP_ELAUNCH, "ovhip_ctx_shares_queue", e);
}

/* Gives the context another stream (its own is parked in the pool): what a caller does with a context whose stream shares a
 * hardware queue with one it must not be held up by. */
int ovhip_ctx_new_stream(ovhip_ctx *ctx)
{
    if (!ctx || !ctx->owns_stream) return OVHIP_EINVAL;
    OV_DEVICE(ctx);
    OV_HIP(ctx, hipStreamSynchronize(ctx->main_stream));
    hipStream_t s = nullptr;
    OV_HIP(ctx, hipStreamCreateWithFlags(&s, hipStreamNonBlocking));          // a NEW one: the pool would hand the old one back
    stream_put(ctx->device, ctx->main_stream);
    if (ctx->stream == ctx->main_stream) ctx->stream = s;
    ctx->main_stream = s;
    return OVHIP_OK;
}

int ovhip_ctx_create(ovhip_ctx **out, int device, void *stream)
{
    if (!out) return OVHIP_EINVAL;
    *out = nullptr;
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess || n <= 0 || device < 0 || device >= n) return OVHIP_ENODEV;
    if (hipSetDevice(device) != hipSuccess) return OVHIP_ENODEV;
    ovhip_ctx *ctx = (ovhip_ctx *)calloc(1, sizeof(*ctx));
    if (!ctx) return OVHIP_ENOMEM;
    ctx->device = device;
    if (hipDeviceGetAttribute(&ctx->num_cus, hipDeviceAttributeMultiprocessorCount, device) != hipSuccess || ctx->num_cus <= 0)
        ctx->num_cus = 256;
    if (stream) {
        ctx->stream = (hipStream_t)stream;
    } else {
        if (stream_get(device, &ctx->stream) != hipSuccess) { free(ctx); return OVHIP_ENODEV; }
        ctx->owns_stream = 1;
    }
    ctx->main_stream = ctx->stream;
    *out = ctx;
    return OVHIP_OK;
}

void ovhip_ctx_destroy(ovhip_ctx *ctx)
{
    if (!ctx) return;
    (void)hipSetDevice(ctx->device);
    // (whatever is still queued on the streams completes before they are handed to the next context)
    for (int i = 0; i < OV_MAX_LANES; ++i) {
        if (ctx->lane[i]) { (void)hipStreamSynchronize(ctx->lane[i]); stream_put(ctx->device, ctx->lane[i]); }
        if (ctx->have_events) (void)hipEventDestroy(ctx->ev_lane[i]);
    }
    if (ctx->have_events) (void)hipEventDestroy(ctx->ev_fork);
    if (ctx->scratch_d || ctx->scratch_h) (void)hipSetDevice(ctx->device);
    if (ctx->scratch_d) (void)hipFree(ctx->scratch_d);
    if (ctx->scratch_h) (void)hipHostFree(ctx->scratch_h);
    if (ctx->ev_sync) (void)hipEventDestroy(ctx->ev_sync);
    if (ctx->owns_stream) { (void)hipStreamSynchronize(ctx->main_stream); stream_put(ctx->device, ctx->main_stream); }
    free(ctx);
}

/* Independent launches of one stage (plain / refined / affine prediction units write disjoint samples; the two
 * size classes of transform blocks likewise) may overlap: ovhip_ctx_fork(ctx, k) routes the following launches to
 * side stream k (k = 1 .. 3; k = 0 = back to the main stream) after everything enqueued on the main stream so far;
 * ovhip_ctx_join(ctx) makes the main stream wait for every side stream used since the last join.  Side streams
 * only fill the ramp-up / tail gaps of the main kernel -- the data dependencies stay those of the stage order. */
int ovhip_ctx_fork(ovhip_ctx *ctx, int k)
{
    if (!ctx || k < 0 || k >= OV_MAX_LANES) return OVHIP_EINVAL;
    OV_DEVICE(ctx);
    if (k == 0) { ctx->stream = ctx->main_stream; return OVHIP_OK; }
    if (!ctx->have_events) {
        OV_HIP(ctx, hipEventCreateWithFlags(&ctx->ev_fork, hipEventDisableTiming));
        for (int i = 0; i < OV_MAX_LANES; ++i) OV_HIP(ctx, hipEventCreateWithFlags(&ctx->ev_lane[i], hipEventDisableTiming));
        ctx->have_events = 1;
    }
    if (!ctx->lane[k]) OV_HIP(ctx, stream_get(ctx->device, &ctx->lane[k]));
    // every fork orders the lane behind everything enqueued on the main stream so far (header contract), also a
    // second fork(k) before the join
    OV_HIP(ctx, hipEventRecord(ctx->ev_fork, ctx->main_stream));
    OV_HIP(ctx, hipStreamWaitEvent(ctx->lane[k], ctx->ev_fork, 0));
    ctx->lane_used[k] = 1;
    ctx->stream = ctx->lane[k];
    return OVHIP_OK;
}

int ovhip_ctx_join(ovhip_ctx *ctx)
{
    if (!ctx) return OVHIP_EINVAL;
    OV_DEVICE(ctx);
    ctx->stream = ctx->main_stream;
    for (int k = 1; k < OV_MAX_LANES; ++k) {
        if (!ctx->lane_used[k]) continue;
        OV_HIP(ctx, hipEventRecord(ctx->ev_lane[k], ctx->lane[k]));
        OV_HIP(ctx, hipStreamWaitEvent(ctx->main_stream, ctx->ev_lane[k], 0));
        ctx->lane_used[k] = 0;
    }
    return OVHIP_OK;
}

int ovhip_ctx_sync(ovhip_ctx *ctx)
{
    if (!ctx) return OVHIP_EINVAL;
    OV_DEVICE(ctx);
    int r = ovhip_ctx_join(ctx);
    if (r != OVHIP_OK) return r;
    hipError_t e = hipStreamSynchronize(ctx->main_stream);
    if (e != hipSuccess) return ov_fail(ctx, OVHIP_ELAUNCH, "hipStreamSynchronize", e);
    return OVHIP_OK;
}

const char *ovhip_last_error(const ovhip_ctx *ctx) { return ctx ? ctx->err : "no context"; }
void *ovhip_ctx_stream(ovhip_ctx *ctx) { return ctx ? (void *)ctx->main_stream : nullptr; }

int ovhip_malloc(ovhip_ctx *ctx, size_t bytes, void **dptr)
{
    if (!ctx || !dptr) return OVHIP_EINVAL;
    OV_DEVICE(ctx);
    *dptr = nullptr;
    hipError_t e = hipMalloc(dptr, bytes ? bytes : 16);
    if (e != hipSuccess) return ov_fail(ctx, OVHIP_ENOMEM, "hipMalloc", e);
    return OVHIP_OK;
}

int ovhip_free(ovhip_ctx *ctx, void *dptr)
{
    if (!ctx) return OVHIP_EINVAL;
    OV_DEVICE(ctx);
    if (dptr) OV_HIP(ctx, hipFree(dptr));
    return OVHIP_OK;
}

int ovhip_h2d(ovhip_ctx *ctx, void *dptr, const void *host, size_t bytes)
{
    if (!ctx || (bytes && (!dptr || !host))) return OVHIP_EINVAL;
    OV_DEVICE(ctx);
    if (bytes) OV_HIP(ctx, hipMemcpyAsync(dptr, host, bytes, hipMemcpyHostToDevice, ctx->stream));
    return OVHIP_OK;
}

int ovhip_d2h(ovhip_ctx *ctx, void *host, const void *dptr, size_t bytes)
{
    if (!ctx || (bytes && (!dptr || !host))) return OVHIP_EINVAL;
    OV_DEVICE(ctx);
    if (bytes) {
        OV_HIP(ctx, hipMemcpyAsync(host, dptr, bytes, hipMemcpyDeviceToHost, ctx->stream));
        OV_HIP(ctx, hipStreamSynchronize(ctx->stream));
    }
    return OVHIP_OK;
}

/* page-locked host memory for buffers the device reads or writes by DMA (output frames, call logs) */
void *ovhip_host_alloc(size_t bytes)
{
    void *p = nullptr;
    return hipHostMalloc(&p, bytes ? bytes : 16, hipHostMallocDefault) == hipSuccess ? p : nullptr;
}

void ovhip_host_free(void *p) { if (p) (void)hipHostFree(p); }

/* device-to-device copy on the context's stream, complete on return (a picture to / from a staging buffer of the host harness) */
int ovhip_d2d(ovhip_ctx *ctx, void *dst, const void *src, size_t bytes)
{
    if (!ctx || (bytes && (!dst || !src))) return OVHIP_EINVAL;
    OV_DEVICE(ctx);
    if (bytes) {
        OV_HIP(ctx, hipMemcpyAsync(dst, src, bytes, hipMemcpyDeviceToDevice, ctx->stream));
        OV_HIP(ctx, hipStreamSynchronize(ctx->stream));
    }
    return OVHIP_OK;
}

int ovhip_pic_alloc(ovhip_ctx *ctx, int32_t w, int32_t h, ovhip_pic *pic)
{
    if (!ctx || !pic || w <= 0 || h <= 0 || (w & 1) || (h & 1)) return OVHIP_EINVAL;
    OV_DEVICE(ctx);
    memset(pic, 0, sizeof(*pic));
    // one allocation, three tight planes (what the reference's frame pool hands out), each 256-B aligned
    const size_t ysz = ((size_t)w * h * 2 + 255) & ~(size_t)255;
    const size_t csz = ((size_t)(w / 2) * (h / 2) * 2 + 255) & ~(size_t)255;
    void *base = nullptr;
    hipError_t e = hipMalloc(&base, ysz + 2 * csz);
    if (e != hipSuccess) return ov_fail(ctx, OVHIP_ENOMEM, "hipMalloc(picture)", e);
    // zero-filled, and complete before the call returns (any context's stream may decode into it next): the ordered intra pass
    // hands samples over by their bit 15, so a destination picture must not carry that bit in on entry -- recycled device memory can
    e = hipMemsetAsync(base, 0, ysz + 2 * csz, ctx->stream);
    if (e == hipSuccess) e = hipStreamSynchronize(ctx->stream);
    if (e != hipSuccess) { (void)hipFree(base); return ov_fail(ctx, OVHIP_ELAUNCH, "hipMemset(picture)", e); }
    pic->y = (uint16_t *)base;
    pic->cb = (uint16_t *)((char *)base + ysz);
    pic->cr = (uint16_t *)((char *)base + ysz + csz);
    pic->w = w; pic->h = h; pic->stride_y = w; pic->stride_c = w / 2;
    return OVHIP_OK;
}

int ovhip_pic_free(ovhip_ctx *ctx, ovhip_pic *pic)
{
    if (!ctx || !pic) return OVHIP_EINVAL;
    OV_DEVICE(ctx);
    if (pic->y) OV_HIP(ctx, hipFree(pic->y));
    memset(pic, 0, sizeof(*pic));
    return OVHIP_OK;
}

static int copy_planes(ovhip_ctx *ctx, const ovhip_pic *pic, uint16_t *y, uint16_t *cb, uint16_t *cr,
                       int32_t hs_y, int32_t hs_c, int to_device)
{
    if (!ctx || !pic || !pic->y) return OVHIP_EINVAL;
    OV_DEVICE(ctx);
    uint16_t *host[3] = { y, cb, cr };
    uint16_t *dev[3] = { pic->y, pic->cb, pic->cr };
    for (int p = 0; p < 3; ++p) {
        if (!host[p]) continue;
        const int w = p ? pic->w / 2 : pic->w, h = p ? pic->h / 2 : pic->h;
        const size_t dpitch = (size_t)(p ? pic->stride_c : pic->stride_y) * 2, hpitch = (size_t)(p ? hs_c : hs_y) * 2;
        if (to_device) OV_HIP(ctx, hipMemcpy2DAsync(dev[p], dpitch, host[p], hpitch, (size_t)w * 2, h, hipMemcpyHostToDevice, ctx->stream));
        else           OV_HIP(ctx, hipMemcpy2DAsync(host[p], hpitch, dev[p], dpitch, (size_t)w * 2, h, hipMemcpyDeviceToHost, ctx->stream));
    }
    OV_HIP(ctx, ov_sync_stream(ctx));
    return OVHIP_OK;
}

int ovhip_pic_upload(ovhip_ctx *ctx, const ovhip_pic *pic, const uint16_t *y, const uint16_t *cb,
                     const uint16_t *cr, int32_t hs_y, int32_t hs_c)
{
    return copy_planes(ctx, pic, (uint16_t *)y, (uint16_t *)cb, (uint16_t *)cr, hs_y, hs_c, 1);
}

int ovhip_pic_download(ovhip_ctx *ctx, const ovhip_pic *pic, uint16_t *y, uint16_t *cb, uint16_t *cr,
                       int32_t hs_y, int32_t hs_c)
{
    return copy_planes(ctx, pic, y, cb, cr, hs_y, hs_c, 0);
}

} // extern "C"
