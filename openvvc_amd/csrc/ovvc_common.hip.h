// ovvc_common.hip.h -- shared device helpers and the engine context (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <string.h>
#include "ovvc_hip.h"

// Measured on MI355X: an empty one-wave workgroup per item costs ~0.4 ns/item (48.9k workgroups in ~20 us incl.
// the command fetch); capping the grid to a resident set and striding was SLOWER for every latency-bound kernel
// here (fewer loads in flight), so the kernels keep their grid-stride loops but are launched with one workgroup
// per item.

#define OV_BD 10
#define OV_PIX_MAX ((1 << OV_BD) - 1)

#define OV_MAX_LANES 4
struct ovhip_ctx {
    int device;
    int num_cus;                   // compute units of the device (sizes the resident grids of the pipelined kernels)
    hipStream_t stream;            // the stream launches go to: the main stream, or a side lane after ovhip_ctx_fork
    int owns_stream;
    hipStream_t main_stream;       // what the caller passed / what ctx_create made
    hipStream_t lane[OV_MAX_LANES];// side streams for independent launches of one stage (created on first use)
    hipEvent_t ev_fork, ev_lane[OV_MAX_LANES];
    int lane_used[OV_MAX_LANES];
    int have_events;
    // grow-only scratch of the synchronous conveniences (ovhip_pic_output / _digest): a hipMalloc + hipFree pair per call would
    // synchronise the whole device every time a picture is output
    void *scratch_d; size_t scratch_d_cap;
    void *scratch_h; size_t scratch_h_cap;       // page-locked
    hipEvent_t ev_sync;            // the synchronous conveniences wait for an event behind their last command (measured in round 4: hipStreamSynchronize from
                                   // 16 frame threads cost the stream a fifth of its rate, an event wait nothing)
    char err[256];
};

static inline int ov_fail(ovhip_ctx *ctx, int code, const char *what, hipError_t e)
{
    if (ctx) snprintf(ctx->err, sizeof(ctx->err), "%s: %s", what, e == hipSuccess ? "invalid argument" : hipGetErrorString(e));
    return code;
}

#define OV_HIP(ctx, call)                                                     \
    do {                                                                      \
        hipError_t e__ = (call);                                              \
        if (e__ != hipSuccess) return ov_fail((ctx), OVHIP_ENODEV, #call, e__); \
    } while (0)

// The HIP current device is per THREAD: a context created on one decoder thread and driven from another (one ctx per
// frame thread) must re-select its device in every entry point, or allocations and launches land on
// device 0.  hipSetDevice on the already-current device is a thread-local compare.
#define OV_DEVICE(ctx)                                                        \
    do {                                                                      \
        hipError_t e__ = hipSetDevice((ctx)->device);                         \
        if (e__ != hipSuccess) return ov_fail((ctx), OVHIP_ENODEV, "hipSetDevice", e__); \
        (void)hipGetLastError();   /* the last-error slot is per thread and sticky: what OV_LAUNCH_CHECK reads must be this entry point's */ \
    } while (0)

/* wait for everything enqueued on the context's current stream: an event behind the last command, waited for on the host (see
 * ovhip_ctx.ev_sync: hipStreamSynchronize from many threads at once is what the runtime does badly) */
static inline hipError_t ov_sync_stream(ovhip_ctx *ctx)
{
    hipError_t e = hipSuccess;
    if (!ctx->ev_sync) e = hipEventCreateWithFlags(&ctx->ev_sync, hipEventDisableTiming);
    if (e == hipSuccess) e = hipEventRecord(ctx->ev_sync, ctx->stream);
    if (e == hipSuccess) e = hipEventSynchronize(ctx->ev_sync);
    return e;
}

static inline int ov_scratch(ovhip_ctx *ctx, size_t dev_bytes, size_t host_bytes)
{
    if (dev_bytes > ctx->scratch_d_cap) {
        if (ctx->scratch_d) (void)hipFree(ctx->scratch_d);
        ctx->scratch_d = nullptr; ctx->scratch_d_cap = 0;
        hipError_t e = hipMalloc(&ctx->scratch_d, dev_bytes);
        if (e != hipSuccess) return ov_fail(ctx, OVHIP_ENOMEM, "hipMalloc(scratch)", e);
        ctx->scratch_d_cap = dev_bytes;
    }
    if (host_bytes > ctx->scratch_h_cap) {
        if (ctx->scratch_h) (void)hipHostFree(ctx->scratch_h);
        ctx->scratch_h = nullptr; ctx->scratch_h_cap = 0;
        hipError_t e = hipHostMalloc(&ctx->scratch_h, host_bytes, hipHostMallocDefault);
        if (e != hipSuccess) return ov_fail(ctx, OVHIP_ENOMEM, "hipHostMalloc(scratch)", e);
        ctx->scratch_h_cap = host_bytes;
    }
    return OVHIP_OK;
}

#define OV_LAUNCH_CHECK(ctx, name)                                            \
    do {                                                                      \
        hipError_t e__ = hipGetLastError();                                   \
        if (e__ != hipSuccess) return ov_fail((ctx), OVHIP_ELAUNCH, name, e__); \
    } while (0)

// Workgroups are dealt to the 8 XCDs round-robin (workgroup i -> XCD i % 8).  Map workgroup i of a grid of
// n to item slot: XCD k = i % 8 takes the k-th contiguous chunk of the item list, in order.
#define OV_NUM_XCD 8
__device__ __forceinline__ uint32_t ov_xcd_slot(uint32_t i, uint32_t n)
{
    const uint32_t k = i % OV_NUM_XCD, j = i / OV_NUM_XCD;
    const uint32_t base = n / OV_NUM_XCD, rem = n % OV_NUM_XCD;          // chunks 0..rem-1 hold base + 1 items
    return k * base + (k < rem ? k : rem) + j;
}

// The same for the workgroups [first, first + n) of a larger grid (several planes' tiles in one launch): workgroup b runs on XCD
// b % 8 whatever `first` is, and XCD k takes the k-th contiguous chunk of THIS range's items -- the k-th band of every plane, so the
// planes' bands an XCD works on lie on top of each other (a chroma tile's CC-ALF taps read the luma band its own L2 holds).
__device__ __forceinline__ uint32_t ov_xcd_slot_at(uint32_t b, uint32_t first, uint32_t n)
{
    const uint32_t i = b - first, k = b % OV_NUM_XCD, f = first % OV_NUM_XCD;
    uint32_t start = 0;
#pragma unroll
    for (uint32_t q = 0; q < OV_NUM_XCD; ++q) {           // workgroups of the range on the XCDs before k
        const uint32_t r = (q + OV_NUM_XCD - f) % OV_NUM_XCD;                   // XCD q's first workgroup of the range is first + r
        if (q < k) start += (n + OV_NUM_XCD - 1 - r) / OV_NUM_XCD;
    }
    return start + i / OV_NUM_XCD;
}

// row * stride for sample addressing: both fit 24 bits, and v_mul_i32_i24 / v_mad_i32_i24 are full rate where the 32-bit
// v_mul_lo_u32 takes four issue slots
__device__ __forceinline__ int ov_rowoff(int row, int stride) { return __mul24(row, stride); }

__device__ __forceinline__ int ov_clip3(int v, int lo, int hi) { return min(max(v, lo), hi); }
__device__ __forceinline__ int ov_clip16(int v) { return ov_clip3(v, -32768, 32767); }
__device__ __forceinline__ int ov_clip_bd(int v) { return ov_clip3(v, 0, OV_PIX_MAX); }

__device__ __forceinline__ uint16_t *ov_plane(const ovhip_pic &p, int plane, int &stride)
{
    stride = plane ? p.stride_c : p.stride_y;
    return plane == 0 ? p.y : (plane == 1 ? p.cb : p.cr);
}
