// ovvc_picture.hip -- the per-picture flush of the rcn back-end, in C (include/ovvc_hip.h "Picture job").
//
// The reference reconstructs block by block while it parses (coding_unit() -> rcn slots, vcl_coding_unit.c:711-840);
// the device path defers: the slots record, and the LAST alf.rcn_alf_filter_line of a picture (slicedec.c:940-955) hands
// the recorded picture over.  This file is that hand-over: asynchronous H2D of the recorder's page-locked arrays, the
// launch chain prediction -> residual -> (ordered intra pass) -> inverse luma mapping -> deblocking -> SAO -> ALF, and
// the D2H of the motion vectors DMVR refined (the `OVMV *mv0, *mv1` in/out contract of rcn_dmvr_mv_refine,
// rcn_structures.h:628-632).  Host code above the launches is plain C; nothing here falls back to a CPU path.
#include "ovvc_common.hip.h"
#include <stdlib.h>

extern "C" int ovhip_dmvr_search_launch(ovhip_ctx *ctx, const ovhip_pic *geom, const ovhip_pic *refs, uint32_t n_refs,
                                        const ovhip_mc_unit *d_units, uint32_t n_units, int32_t *d_mv_out);

struct BandState;
namespace {

struct DevBuf { void *p; size_t cap; };

enum { B_TB, B_COEF, B_MC, B_MCX, B_MV, B_AFF, B_SIDE, B_REG, B_SCALE, B_EV, B_EH, B_PARAM, B_CLASS, B_CIIP, B_ITASK, B_ICTU, B_IITEM, B_TMVP, B_COUNT };

// layout of the parameter block (one pinned staging copy, one H2D)
struct ParamLayout { size_t sao, alf_ctus, lcoef, lclip, ccoef, cclip, cc, fwd, bwd, total; };

} // namespace

struct ovhip_job {
    ovhip_ctx *ctx;
    ovhip_ctx *home;                     // the context the job was created on (ovhip_job_bind(job, NULL) returns to it)
    int32_t w, h;
    ovhip_recorder *rec;
    DevBuf dev[B_COUNT];
    ovhip_pic tmp;                       // SAO destination / ALF source
    ovhip_pic res;                       // residuals of the ordered tasks (allocated with the first picture that has any)
    uint32_t *d_sync; uint32_t epoch;    // CTU flags of the one-launch ordered pass (zeroed once; a new epoch per picture)
    uint32_t *d_flow;                    // unit state words of the flow launch (zeroed once)
    uint32_t *items_host; size_t items_cap;   // pinned: items of the flow launch
    uint32_t *abort_host;                // pinned word the ordered pass writes when a bounded wait expired
    char *param_host; size_t param_cap;  // pinned staging of the picture-level tables
    int32_t *mv_host; size_t mv_cap;     // pinned: refined vectors, 4 int32 per refined unit
    struct { int valid, has_intra; ovhip_pic dst, refs[16], intra; uint32_t n_refs; ovhip_job_params pr; } again;   // the last flush's arguments
    uint32_t n_retries;                  // second passes of the last picture (ovhip_job_wait)
    int flow_launched;                   // the last flush had a flow launch (ovhip_job_wait: a clean one counts towards the decay of g_flow_shift)
    int test_abort, test_abort_seen;     // ovhip_job_test_abort_next_flow (a forced abort does not count as evidence of starvation)
    ovhip_tmvp_cell *tmvp_host; size_t tmvp_cap, n_tmvp;   // pinned: TMVP plane cells of the refined units (ovhip_job_params.tmvp_cells)
    size_t n_mv;                         // units covered by the last flush / eager pass
    size_t dmvr_first;                   // refined units [0, dmvr_first) already went through the eager search
    size_t rows_end; int rows_pending;   // an eager pass is in flight: it covers [.., rows_end), ev_rows follows its copies
    hipEvent_t ev_rows;
    hipEvent_t ev_h2d, ev_done;
    hipStream_t up;                      // uploads on one of the device's shared upload streams (upload_stream_for): up != NULL during such a flush
    int flow_on_device;                  // the last full flush uploaded the flow launch's item list (a resident replay may use it)
    int flushed;                         // ev_* recorded at least once
    const void *packed_prev[24];         // where the last full flush placed the arrays that rode in the parameter block
    int resident;                        // this flush reuses the device copies of the previous one (OVHIP_STAGE_RESIDENT)
    ovhip_job_stats st;
    struct BandState *bs;                // band-wise submission (ovhip_job_band): allocated with the first band of the job's life
    // optional: HIP-event bracket around ONE launch group of the flush (ovhip_job_time_stage)
    int t_stage;                         // OVHIP_TIME_* or -1
    hipEvent_t t_ev[32][2]; uint8_t t_pending[32]; int t_next;
    double t_sum_ms; uint64_t t_count;
};

static void band_free(ovhip_job *j);
static int  band_reset(ovhip_job *j);
static int  band_active(const ovhip_job *j);
static int  band_wait_done(ovhip_job *j);

// workers of a flow launch = (6 x CUs) >> g_flow_shift[device]: grows with every launch of that device that was abandoned (co-resident
// flow launches starving each other: another GPU_MAX_HW_QUEUES, another process on the GPU) and decays again -- one step per
// FLOW_DECAY pictures whose flow launch went through -- so that one transient event does not halve the device's pictures for the life of
// the process; ovhip_job_stats.flow_shift reports it
enum { FLOW_DEVS = 64, FLOW_DECAY = 512 };
static int g_flow_shift[FLOW_DEVS], g_flow_clean[FLOW_DEVS];
static inline int flow_shift_of(int device)          // clamped where it is read: a shift count is never negative, never above 4
{
    const int s = __atomic_load_n(&g_flow_shift[device & (FLOW_DEVS - 1)], __ATOMIC_RELAXED);
    return s < 0 ? 0 : s > 4 ? 4 : s;
}

namespace {

// Small results that the host reads after the picture is complete (refined vectors, TMVP plane entries) leave the device from a
// kernel that stores into page-locked host memory: a hipMemcpyAsync between two kernels of a picture's chain is a round trip
// compute queue -> DMA engine -> compute queue with the hardware queue (shared by four pictures' streams) held at the barrier.
__global__ __launch_bounds__(256) void k_store_host(uint4 *__restrict__ dst, const uint4 *__restrict__ src, size_t n)
{
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i < n) dst[i] = src[i];
}

int store_host(ovhip_ctx *ctx, void *dst_host, const void *src_dev, size_t bytes)
{
    const size_t n = (bytes + 15) / 16;                 // (both buffers are allocated in multiples of 16 bytes)
    if (!n) return OVHIP_OK;
    hipLaunchKernelGGL(k_store_host, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, ctx->stream, (uint4 *)dst_host, (const uint4 *)src_dev, n);
    OV_LAUNCH_CHECK(ctx, "k_store_host");
    return OVHIP_OK;
}

double host_now_us();
int grow_trace() { static int on = -1; if (on < 0) on = getenv("OVVC_HIP_GROW_TRACE") != nullptr; return on; }

void *pinned_alloc(void *user, size_t bytes)
{
    (void)user;
    void *p = nullptr;
    const double t0 = grow_trace() ? host_now_us() : 0.0;
    const bool ok = hipHostMalloc(&p, bytes ? bytes : 16, hipHostMallocDefault) == hipSuccess;
    if (grow_trace()) fprintf(stderr, "grow: pinned %zu bytes at %.3f ms, took %.3f ms\n", bytes, 1e-3 * t0, 1e-3 * (host_now_us() - t0));
    return ok ? p : nullptr;
}
void pinned_free(void *user, void *p) { (void)user; if (p) (void)hipHostFree(p); }

int dev_reserve(ovhip_job *j, int k, size_t bytes)
{
    DevBuf &b = j->dev[k];
    if (bytes <= b.cap) return OVHIP_OK;
    size_t nc = b.cap ? b.cap : (size_t)1 << 16;
    while (nc < bytes) nc *= 2;
    // growth is rare (first pictures of a sequence); hipFree synchronises the device, which is what makes it safe here
    const double t0 = grow_trace() ? host_now_us() : 0.0;
    if (b.p) OV_HIP(j->ctx, hipFree(b.p));
    b.p = nullptr; b.cap = 0;
    hipError_t e = hipMalloc(&b.p, nc);
    if (grow_trace()) fprintf(stderr, "grow: device buffer %d to %zu bytes at %.3f ms, took %.3f ms\n", k, nc, 1e-3 * t0, 1e-3 * (host_now_us() - t0));
    if (e != hipSuccess) return ov_fail(j->ctx, OVHIP_ENOMEM, "hipMalloc(job buffer)", e);
    b.cap = nc;
    return OVHIP_OK;
}

// The uploads of all pictures of a device on a FEW shared streams instead of every picture's own (tools/micro/h2d_concurrent.py, round 4:
// 16 streams copying at once reach 31-38 GB/s, 1-4 streams 52-56): a picture's copies go to one of the device's upload streams, dealt
// round-robin per flush, and the picture's own stream waits for the event behind them.  Measured on the headline stream, interleaved
// runs on one box (tools/debug/ab_upload_streams.sh; LABBOOK 12): own stream 3309 +- 73 pictures/s, two upload streams 3506 +- 25; on
// a second box 3336 / 3419 (three: 3427, four: 3247, eight: 3366).  OVVC_HIP_UPLOAD_STREAMS=0: every picture uploads on its own stream.
enum { UP_DEVS = 64, UP_MAX = 8 };
static pthread_mutex_t g_up_mtx = PTHREAD_MUTEX_INITIALIZER;
static hipStream_t g_up[UP_DEVS][UP_MAX];
static unsigned g_up_next[UP_DEVS];
static int g_up_n = -1;
static int upload_stream_count();
static hipStream_t upload_stream_for(ovhip_job *j)
{
    const int n = upload_stream_count(), dev = j->ctx->device;
    if (n <= 0 || dev < 0 || dev >= UP_DEVS || j->t_stage >= 0) return nullptr;
    pthread_mutex_lock(&g_up_mtx);
    const unsigned k = g_up_next[dev]++ % (unsigned)n;
    if (!g_up[dev][k] && hipStreamCreateWithFlags(&g_up[dev][k], hipStreamNonBlocking) != hipSuccess) g_up[dev][k] = nullptr;
    hipStream_t st = g_up[dev][k];
    pthread_mutex_unlock(&g_up_mtx);
    return st;
}

static int upload_stream_count()
{
    if (__atomic_load_n(&g_up_n, __ATOMIC_RELAXED) < 0) {
        const char *e = getenv("OVVC_HIP_UPLOAD_STREAMS"); const int n = e ? atoi(e) : 2;
        __atomic_store_n(&g_up_n, n < 0 ? 0 : n > UP_MAX ? UP_MAX : n, __ATOMIC_RELAXED);
    }
    return __atomic_load_n(&g_up_n, __ATOMIC_RELAXED);
}

int h2d(ovhip_job *j, int k, const void *host, size_t bytes)
{
    if (!bytes) return OVHIP_OK;
    if (j->resident) return j->dev[k].cap >= bytes ? OVHIP_OK : ov_fail(j->ctx, OVHIP_EINVAL, "resident flush before a full one", hipSuccess);
    int r = dev_reserve(j, k, bytes);
    if (r) return r;
    OV_HIP(j->ctx, hipMemcpyAsync(j->dev[k].p, host, bytes, hipMemcpyHostToDevice, j->up ? j->up : j->ctx->stream));
    j->st.h2d_bytes += bytes; j->st.n_h2d++;
    return OVHIP_OK;
}

int pinned_reserve(ovhip_job *j, void **p, size_t *cap, size_t bytes)
{
    if (bytes <= *cap) return OVHIP_OK;
    size_t nc = *cap ? *cap : 4096;
    while (nc < bytes) nc *= 2;
    void *q = pinned_alloc(nullptr, nc);
    if (!q) return ov_fail(j->ctx, OVHIP_ENOMEM, "hipHostMalloc", hipSuccess);
    if (*p) { memcpy(q, *p, *cap); pinned_free(nullptr, *p); }
    *p = q; *cap = nc;
    return OVHIP_OK;
}

extern "C" const ovhip_tb_cmd *ovhip_rec_tb_cmds_split_tiny_(ovhip_recorder *r, size_t counts[4], size_t tiny[4][4], size_t *n);
extern "C" int ovhip_itx_launch_ex_(ovhip_ctx *ctx, const ovhip_pic *dst, const ovhip_pic *res, const ovhip_tb_cmd *d_cmds, uint32_t n_large,
                                    uint32_t n_small, const uint32_t tiny3[4], const int16_t *d_coefs, const int16_t *d_lmcs_scales, const uint16_t *d_bwd_lut);

#define CHK(x) do { int r__ = (x); if (r__ != OVHIP_OK) return r__; } while (0)

#ifndef OVHIP_PACK_LIMIT
#define OVHIP_PACK_LIMIT (1 << 20)       /* bytes: arrays up to this size ride in the staging block (round 6, uploads on shared streams:
                                          * 128 KB 3414 +- 147 / 3491 +- 79, 1 MB 3492 +- 69 / 3566 +- 49, 2 MB 3456, everything 3440 pictures/s;
                                          * tools/debug/ab_pack_limit.sh) */
#endif

double host_now_us()
{
    timespec t;
    clock_gettime(CLOCK_MONOTONIC, &t);
    return 1e6 * (double)t.tv_sec + 1e-3 * (double)t.tv_nsec;
}

int t_collect(ovhip_job *j, int k)
{
    if (!j->t_pending[k]) return OVHIP_OK;
    float ms = 0.f;
    OV_HIP(j->ctx, hipEventSynchronize(j->t_ev[k][1]));
    OV_HIP(j->ctx, hipEventElapsedTime(&ms, j->t_ev[k][0], j->t_ev[k][1]));
    j->t_sum_ms += ms; j->t_count++;
    j->t_pending[k] = 0;
    return OVHIP_OK;
}

// bracket the launches of `stage` with an event pair on the launch stream (a pair costs a few us of stream time, which
// is why only ONE stage is bracketed at a time)
struct StageTimer {
    ovhip_job *j; int k;
    StageTimer(ovhip_job *job, int stage) : j(job), k(-1)
    {
        if (j->t_stage != stage) return;
        k = j->t_next; j->t_next = (j->t_next + 1) % 32;
        if (t_collect(j, k) != OVHIP_OK || hipEventRecord(j->t_ev[k][0], j->ctx->stream) != hipSuccess) k = -1;
    }
    ~StageTimer()
    {
        if (k >= 0 && hipEventRecord(j->t_ev[k][1], j->ctx->stream) == hipSuccess) j->t_pending[k] = 1;
    }
};

// plane-wise device copy (the two pictures may come from different allocators)
int copy_pic(ovhip_ctx *ctx, const ovhip_pic *d, const ovhip_pic *s)
{
    uint16_t *dp[3] = { d->y, d->cb, d->cr };
    uint16_t *sp[3] = { s->y, s->cb, s->cr };
    for (int p = 0; p < 3; ++p) {
        const int w = p ? d->w / 2 : d->w, h = p ? d->h / 2 : d->h;
        OV_HIP(ctx, hipMemcpy2DAsync(dp[p], (size_t)(p ? d->stride_c : d->stride_y) * 2, sp[p], (size_t)(p ? s->stride_c : s->stride_y) * 2,
                                     (size_t)w * 2, h, hipMemcpyDeviceToDevice, ctx->stream));
    }
    return OVHIP_OK;
}

} // namespace

extern "C" {

int ovhip_job_create(ovhip_ctx *ctx, int32_t w, int32_t h, ovhip_job **out)
{
    if (!ctx || !out || w <= 0 || h <= 0 || (w & 1) || (h & 1)) return OVHIP_EINVAL;
    OV_DEVICE(ctx);
    *out = nullptr;
    ovhip_job *j = (ovhip_job *)calloc(1, sizeof(*j));
    if (!j) return OVHIP_ENOMEM;
    j->ctx = ctx; j->home = ctx; j->w = w; j->h = h; j->t_stage = -1;
    const ovhip_allocator al = { pinned_alloc, pinned_free, nullptr };
    j->rec = ovhip_rec_create_ex(w, h, &al);
    if (!j->rec) { free(j); return OVHIP_ENOMEM; }
    ovhip_rec_set_dense_dbf_planes(j->rec, 0);
    int r = ovhip_pic_alloc(ctx, w, h, &j->tmp);
    if (r == OVHIP_OK && hipEventCreateWithFlags(&j->ev_h2d, hipEventDisableTiming) != hipSuccess) r = OVHIP_ENODEV;
    if (r == OVHIP_OK && hipEventCreateWithFlags(&j->ev_done, hipEventDisableTiming) != hipSuccess) r = OVHIP_ENODEV;
    if (r == OVHIP_OK && hipEventCreateWithFlags(&j->ev_rows, hipEventDisableTiming) != hipSuccess) r = OVHIP_ENODEV;
    if (r != OVHIP_OK) { ovhip_job_destroy(j); return r; }
    *out = j;
    return OVHIP_OK;
}

void ovhip_job_destroy(ovhip_job *j)
{
    if (!j) return;
    (void)hipSetDevice(j->ctx->device);
    if (j->flushed) (void)hipEventSynchronize(j->ev_done);
    for (int k = 0; k < B_COUNT; ++k) if (j->dev[k].p) (void)hipFree(j->dev[k].p);
    if (j->tmp.y) (void)ovhip_pic_free(j->ctx, &j->tmp);
    if (j->res.y) (void)ovhip_pic_free(j->ctx, &j->res);
    if (j->d_sync) (void)hipFree(j->d_sync);
    if (j->d_flow) (void)hipFree(j->d_flow);
    pinned_free(nullptr, j->items_host);
    pinned_free(nullptr, j->abort_host);
    pinned_free(nullptr, j->param_host); pinned_free(nullptr, j->mv_host); pinned_free(nullptr, j->tmvp_host);
    for (int k = 0; k < 32; ++k) for (int q = 0; q < 2; ++q) if (j->t_ev[k][q]) (void)hipEventDestroy(j->t_ev[k][q]);
    if (j->ev_h2d) (void)hipEventDestroy(j->ev_h2d);
    if (j->ev_done) (void)hipEventDestroy(j->ev_done);
    if (j->ev_rows) { if (j->rows_pending) (void)hipEventSynchronize(j->ev_rows); (void)hipEventDestroy(j->ev_rows); }
    band_free(j);
    ovhip_rec_destroy(j->rec);
    free(j);
}

ovhip_recorder *ovhip_job_recorder(ovhip_job *j) { return j ? j->rec : nullptr; }

/* Every buffer of the job sized for a picture of its size NOW (the recorder's page-locked arrays, ovhip_rec_reserve_for_picture; the
 * device copies; the staging blocks), so that no picture of a running decoder meets a growth: a device buffer that grows is hipFree
 * (a device-wide synchronisation) + hipMalloc, a page-locked one hipHostMalloc + copy + hipHostFree.  What a frame thread's job does
 * when it is created (ovhip_frame_job); a harness that keeps a hundred pre-recorded jobs does not call it.  ~30 MB of page-locked
 * and ~30 MB of device memory at 4K. */
int ovhip_job_reserve_for_picture(ovhip_job *j)
{
    if (!j) return OVHIP_EINVAL;
    OV_DEVICE(j->ctx);
    if (ovhip_rec_reserve_for_picture(j->rec) != OVHIP_OK) return ov_fail(j->ctx, OVHIP_ENOMEM, "ovhip_rec_reserve_for_picture", hipSuccess);
    const size_t P = (size_t)j->w * j->h;
    const struct { int k; size_t bytes; } want[] = {
        { B_TB, (P / 64 + 1024) * sizeof(ovhip_tb_cmd) }, { B_COEF, (P / 2 + 1024) * 2 }, { B_MC, (P / 128 + 1024) * sizeof(ovhip_mc_unit) },
        { B_MCX, (P / 256 + 1024) * sizeof(ovhip_mc_unit) }, { B_MV, (P / 256 + 1024) * 16 }, { B_AFF, (P / 1024 + 256) * sizeof(ovhip_aff_unit) },
        { B_SIDE, (P / 6 + 1024) * 4 }, { B_REG, (P / 4096 + 64) * sizeof(ovhip_lmcs_region) }, { B_SCALE, 65536 },
        { B_EV, (P / 32 + 2048) * sizeof(ovhip_dbf_edge) }, { B_EH, (P / 32 + 2048) * sizeof(ovhip_dbf_edge) }, { B_PARAM, P / 2 + ((size_t)2 << 20) },
        { B_CLASS, (size_t)((j->w + 3) / 4) * ((j->h + 3) / 4) }, { B_ITASK, (P / 64 + 1024) * sizeof(ovhip_itask) }, { B_IITEM, (P / 32 + 1024) * 4 },
        { B_TMVP, 4 * (P / 256 + 1024) * sizeof(ovhip_tmvp_cell) },
    };
    for (const auto &q : want) CHK(dev_reserve(j, q.k, q.bytes));
    CHK(pinned_reserve(j, (void **)&j->param_host, &j->param_cap, P / 2 + ((size_t)2 << 20)));
    CHK(pinned_reserve(j, (void **)&j->mv_host, &j->mv_cap, (P / 256 + 1024) * 16));
    CHK(pinned_reserve(j, (void **)&j->tmvp_host, &j->tmvp_cap, 4 * (P / 256 + 1024) * sizeof(ovhip_tmvp_cell)));
    if (j->items_cap < P / 32 + 1024) {
        pinned_free(nullptr, j->items_host);
        j->items_cap = P / 32 + 1024;
        j->items_host = (uint32_t *)pinned_alloc(nullptr, j->items_cap * sizeof(uint32_t));
        if (!j->items_host) { j->items_cap = 0; return ov_fail(j->ctx, OVHIP_ENOMEM, "ovhip_job_reserve_for_picture: item list", hipSuccess); }
    }
    if (!j->res.y) CHK(ovhip_pic_alloc(j->ctx, j->w, j->h, &j->res));
    return OVHIP_OK;
}

int ovhip_job_begin(ovhip_job *j)
{
    if (!j) return OVHIP_EINVAL;
    OV_DEVICE(j->ctx);
    // the DMA engines may still be reading the recorder's arrays and the parameter staging block
    if (j->flushed) OV_HIP(j->ctx, hipEventSynchronize(j->ev_h2d));
    if (j->rows_pending) { OV_HIP(j->ctx, hipEventSynchronize(j->ev_rows)); j->rows_pending = 0; }
    { const int rb = band_reset(j); if (rb != OVHIP_OK) return rb; }
    ovhip_rec_reset(j->rec);
    j->dmvr_first = 0; j->n_mv = 0; j->n_tmvp = 0; j->rows_end = 0;
    return OVHIP_OK;
}

int ovhip_job_bind(ovhip_job *j, ovhip_ctx *ctx)
{
    if (!j) return OVHIP_EINVAL;
    if (!ctx) ctx = j->home;             // back to the context it was created on (which outlives it by contract)
    // (errors go to the context the caller holds)
    if (ctx->device != j->ctx->device) return ov_fail(ctx, OVHIP_EINVAL, "ovhip_job_bind: context of another device", hipSuccess);
    OV_DEVICE(ctx);
    if (j->flushed) {
        hipError_t e = hipEventSynchronize(j->ev_done);
        if (e != hipSuccess) return ov_fail(ctx, OVHIP_ELAUNCH, "ovhip_job_bind: hipEventSynchronize(previous flush)", e);
    }
    j->ctx = ctx;
    return OVHIP_OK;
}

static int job_flush_impl(ovhip_job *j, const ovhip_pic *dst, const ovhip_pic *refs, uint32_t n_refs, const ovhip_pic *intra,
                          const ovhip_job_params *pr);

int ovhip_job_wait(ovhip_job *j)
{
    if (!j) return OVHIP_EINVAL;
    OV_DEVICE(j->ctx);
    if (!j->flushed) return OVHIP_OK;
    hipError_t e = hipEventSynchronize(j->ev_done);
    if (e != hipSuccess) return ov_fail(j->ctx, OVHIP_ELAUNCH, "hipEventSynchronize(job)", e);
    if (band_active(j)) return band_wait_done(j);
    if (j->abort_host && *(volatile uint32_t *)j->abort_host) {
        // a workgroup of the ordered pass gave up waiting for its inputs (workgroups of several pictures' flow launches can
        // fill the compute units with pollers whose producers then find no slot): the picture is incomplete.  Re-arm, and decode
        // the picture again with one launch per level -- no workgroup of such a launch waits for another -- from the recorder's
        // arrays, which are untouched until the next ovhip_job_begin.  Every sample of dst is rewritten by a flush.
        *(volatile uint32_t *)j->abort_host = 0;
        const int fd = j->ctx->device & (FLOW_DEVS - 1);
        if (!j->test_abort_seen) {
            if (__atomic_load_n(&g_flow_shift[fd], __ATOMIC_RELAXED) < 4) __atomic_fetch_add(&g_flow_shift[fd], 1, __ATOMIC_RELAXED);
            __atomic_store_n(&g_flow_clean[fd], 0, __ATOMIC_RELAXED);
        }
        j->test_abort_seen = 0;
        if (j->d_sync) (void)hipMemset(j->d_sync, 0, sizeof(uint32_t));
        if (j->d_flow) (void)hipMemset(j->d_flow, 0, sizeof(uint32_t));
        j->flow_launched = 0;                //  an abandoned launch is not a clean one for the decay below
        if (j->again.valid && j->n_retries == 0) {
            j->n_retries = 1;
            ovhip_job_params pr = j->again.pr;
            pr.stages = ((pr.stages ? pr.stages : (uint32_t)(OVHIP_STAGE_MC | OVHIP_STAGE_ITX | OVHIP_STAGE_DBF | OVHIP_STAGE_SAO | OVHIP_STAGE_ALF | OVHIP_STAGE_INTRA)) | OVHIP_STAGE_INTRA_LEVELS) & ~(uint32_t)OVHIP_STAGE_INTRA_CTU;
            pr.wait_events = nullptr; pr.n_wait_events = 0; pr.before_launch = nullptr;      // the references were done the first time
            int r = job_flush_impl(j, &j->again.dst, j->again.refs, j->again.n_refs, j->again.has_intra ? &j->again.intra : nullptr, &pr);
            if (r) return r;
            e = hipEventSynchronize(j->ev_done);
            if (e != hipSuccess) return ov_fail(j->ctx, OVHIP_ELAUNCH, "hipEventSynchronize(job, second pass)", e);
            return OVHIP_OK;
        }
        return ov_fail(j->ctx, OVHIP_ELAUNCH, "ordered pass: a CTU's bounded wait for its neighbours expired (picture incomplete)", hipSuccess);
    }
    if (j->flow_launched) {
        const int fd = j->ctx->device & (FLOW_DEVS - 1);
        j->flow_launched = 0;
        //  exactly ONE thread takes the step -- the one whose increment lands on FLOW_DECAY -- and the shift never goes
        // below zero (two threads that both saw ">= FLOW_DECAY" used to subtract twice: 1 -> -1, a negative shift count)
        if (__atomic_load_n(&g_flow_shift[fd], __ATOMIC_RELAXED) > 0 && __atomic_add_fetch(&g_flow_clean[fd], 1, __ATOMIC_RELAXED) == FLOW_DECAY) {
            __atomic_store_n(&g_flow_clean[fd], 0, __ATOMIC_RELAXED);
            int cur = __atomic_load_n(&g_flow_shift[fd], __ATOMIC_RELAXED);
            while (cur > 0 && !__atomic_compare_exchange_n(&g_flow_shift[fd], &cur, cur - 1, true, __ATOMIC_RELAXED, __ATOMIC_RELAXED)) {}
        }
    }
    return OVHIP_OK;
}

const int32_t *ovhip_job_refined_mvs(ovhip_job *j, size_t *n_units)
{
    if (!j || !n_units) return nullptr;
    *n_units = j->n_mv;
    return j->mv_host;
}

const ovhip_tmvp_cell *ovhip_job_tmvp_cells(ovhip_job *j, size_t *n_entries)
{
    if (!j || !n_entries) return nullptr;
    *n_entries = j->n_tmvp;
    return j->tmvp_host;
}

int ovhip_job_last_stats(const ovhip_job *j, ovhip_job_stats *out)
{
    if (!j || !out) return OVHIP_EINVAL;
    *out = j->st;
    return OVHIP_OK;
}

int ovhip_job_time_stage(ovhip_job *j, int stage)
{
    if (!j || stage < -1 || stage >= OVHIP_TIME_COUNT) return OVHIP_EINVAL;
    OV_DEVICE(j->ctx);
    for (int k = 0; k < 32; ++k) {
        CHK(t_collect(j, k));
        for (int q = 0; q < 2 && stage >= 0; ++q)
            if (!j->t_ev[k][q]) OV_HIP(j->ctx, hipEventCreate(&j->t_ev[k][q]));
    }
    j->t_stage = stage; j->t_sum_ms = 0.0; j->t_count = 0;
    return OVHIP_OK;
}

int ovhip_job_stage_time(ovhip_job *j, double *sum_ms, uint64_t *count)
{
    if (!j || !sum_ms || !count) return OVHIP_EINVAL;
    OV_DEVICE(j->ctx);
    for (int k = 0; k < 32; ++k) CHK(t_collect(j, k));
    *sum_ms = j->t_sum_ms; *count = j->t_count;
    return OVHIP_OK;
}

int64_t ovhip_job_dmvr_rows_collect(ovhip_job *j)
{
    if (!j) return OVHIP_EINVAL;
    if (j->rows_pending) {
        ovhip_ctx *ctx = j->ctx;
        OV_DEVICE(ctx);
        hipError_t e = hipEventSynchronize(j->ev_rows);
        j->rows_pending = 0;
        if (e != hipSuccess) return ov_fail(ctx, OVHIP_ELAUNCH, "hipEventSynchronize(dmvr rows)", e);
        if (j->rows_end > j->n_mv) j->n_mv = j->rows_end;
    }
    return (int64_t)j->rows_end;
}

int64_t ovhip_job_dmvr_rows_begin_upto(ovhip_job *j, const ovhip_pic *refs, uint32_t n_refs, int32_t log2_ctu_s, size_t upto);
int64_t ovhip_job_dmvr_rows_begin(ovhip_job *j, const ovhip_pic *refs, uint32_t n_refs, int32_t log2_ctu_s)
{
    return ovhip_job_dmvr_rows_begin_upto(j, refs, n_refs, log2_ctu_s, (size_t)-1);
}

/* the same over the refined units [.., upto) only (a caller that works through a parsed picture row by row as its reference
 * pictures' rows arrive: shim/rcn_hip.c final_progressive) */
int64_t ovhip_job_dmvr_rows_begin_upto(ovhip_job *j, const ovhip_pic *refs, uint32_t n_refs, int32_t log2_ctu_s, size_t upto)
{
    if (!j) return OVHIP_EINVAL;
    // one pass in flight: the pinned result arrays may have to grow, and they keep the recorder's indexing
    const int64_t c = ovhip_job_dmvr_rows_collect(j);
    if (c < 0) return c;
    ovhip_ctx *ctx = j->ctx;
    OV_DEVICE(ctx);
    size_t n = 0;
    const ovhip_mc_unit *u = ovhip_rec_mcx_units(j->rec, &n);
    if (n > upto) n = upto;
    const size_t first = j->dmvr_first;
    if (n <= first) return (int64_t)first;
    int any = 0;
    for (size_t i = first; i < n && !any; ++i) any = (u[i].flags & OVHIP_MC_DMVR) != 0;
    if (any) {
        // the device copy of the unit list and the vector buffer keep the recorder's indexing: [first, n) lands at first
        CHK(pinned_reserve(j, (void **)&j->mv_host, &j->mv_cap, n * 16));
        if (log2_ctu_s) CHK(pinned_reserve(j, (void **)&j->tmvp_host, &j->tmvp_cap, 4 * n * sizeof(ovhip_tmvp_cell)));
        if (j->dev[B_MCX].cap < n * sizeof(ovhip_mc_unit) || j->dev[B_MV].cap < n * 16 ||
            (log2_ctu_s && j->dev[B_TMVP].cap < 4 * n * sizeof(ovhip_tmvp_cell))) {
            // grow to the picture's upper bound at once so that earlier rows' results are never moved: one refined unit
            // covers at least 8x8 luma samples
            const size_t ub = (size_t)((j->w + 7) / 8) * ((j->h + 7) / 8);
            CHK(dev_reserve(j, B_MCX, (ub > n ? ub : n) * sizeof(ovhip_mc_unit)));
            CHK(dev_reserve(j, B_MV, (ub > n ? ub : n) * 16));
            if (log2_ctu_s) CHK(dev_reserve(j, B_TMVP, 4 * (ub > n ? ub : n) * sizeof(ovhip_tmvp_cell)));
        }
        char *d_units = (char *)j->dev[B_MCX].p + first * sizeof(ovhip_mc_unit);
        int32_t *d_mv = (int32_t *)j->dev[B_MV].p + 4 * first;
        OV_HIP(ctx, hipMemcpyAsync(d_units, u + first, (n - first) * sizeof(ovhip_mc_unit), hipMemcpyHostToDevice, ctx->stream));
        ovhip_pic geom = j->tmp;
        CHK(ovhip_dmvr_search_launch(ctx, &geom, refs, n_refs, (const ovhip_mc_unit *)d_units, (uint32_t)(n - first), d_mv));
        CHK(store_host(ctx, j->mv_host + 4 * first, d_mv, (n - first) * 16));
        if (log2_ctu_s) {
            // the same vectors as entries of the picture's collocated motion plane: what the caller patches before it publishes
            // the row (4 entries per unit, recorder order)
            ovhip_tmvp_cell *d_cells = (ovhip_tmvp_cell *)j->dev[B_TMVP].p + 4 * first;
            CHK(ovhip_tmvp_cells_launch(ctx, (const ovhip_mc_unit *)d_units, (uint32_t)(n - first), d_mv, log2_ctu_s,
                                        (j->w + (1 << log2_ctu_s) - 1) >> log2_ctu_s, d_cells));
            CHK(store_host(ctx, j->tmvp_host + 4 * first, d_cells, 4 * (n - first) * sizeof(ovhip_tmvp_cell)));
            j->n_tmvp = 4 * n;
        }
        OV_HIP(ctx, hipEventRecord(j->ev_rows, ctx->stream));
        j->rows_pending = 1;
    }
    else if (log2_ctu_s) {
        // no DMVR unit among them: their plane entries are all "none" (the caller indexes the entries by unit)
        CHK(pinned_reserve(j, (void **)&j->tmvp_host, &j->tmvp_cap, 4 * n * sizeof(ovhip_tmvp_cell)));
        for (size_t i = 4 * first; i < 4 * n; ++i) { j->tmvp_host[i] = ovhip_tmvp_cell{}; j->tmvp_host[i].cell = OVHIP_TMVP_NONE; }
        j->n_tmvp = 4 * n;
    }
    j->dmvr_first = n;
    j->rows_end = n;
    return (int64_t)n;
}

int64_t ovhip_job_dmvr_rows(ovhip_job *j, const ovhip_pic *refs, uint32_t n_refs)
{
    const int64_t n = ovhip_job_dmvr_rows_begin(j, refs, n_refs, 0);
    return n < 0 ? n : ovhip_job_dmvr_rows_collect(j);
}

static int job_flush_impl(ovhip_job *j, const ovhip_pic *dst, const ovhip_pic *refs, uint32_t n_refs, const ovhip_pic *intra,
                          const ovhip_job_params *pr);

int ovhip_job_flush(ovhip_job *j, const ovhip_pic *dst, const ovhip_pic *refs, uint32_t n_refs, const ovhip_pic *intra,
                    const ovhip_job_params *pr)
{
    if (!j || !dst || !pr) return OVHIP_EINVAL;
    // what a second flush of the same picture needs (ovhip_job_wait re-runs the picture with one launch per level when the
    // flow launch gave up): the pictures by value, the parameter block as the caller passed it (its tables stay the caller's
    // until ovhip_job_wait has returned)
    j->again.valid = 0;
    j->n_retries = 0;
    const int r = job_flush_impl(j, dst, refs, n_refs, intra, pr);
    // (after the flush: a before_launch callback may be what fills refs[] -- ovhip_frame_submit resolves the reference pictures
    // there, while the uploads already run)
    j->again.valid = r == OVHIP_OK && n_refs <= 16;
    j->again.dst = *dst;
    for (uint32_t i = 0; i < n_refs && i < 16; ++i) j->again.refs[i] = refs[i];
    j->again.n_refs = n_refs;
    j->again.has_intra = intra != nullptr;
    if (intra) j->again.intra = *intra;
    j->again.pr = *pr;
    return r;
}

int ovhip_job_test_abort_next_flow(ovhip_job *j)
{
    if (!j) return OVHIP_EINVAL;
    j->test_abort = 1;
    return OVHIP_OK;
}

static int job_flush_impl(ovhip_job *j, const ovhip_pic *dst, const ovhip_pic *refs, uint32_t n_refs, const ovhip_pic *intra,
                          const ovhip_job_params *pr)
{
    ovhip_ctx *ctx = j->ctx;
    OV_DEVICE(ctx);
    if ((dst->w != j->w || dst->h != j->h)) return ov_fail(ctx, OVHIP_EINVAL, "ovhip_job_flush: picture size differs from the job's", hipSuccess);
    // an eager pass nobody collected (ovhip_job_dmvr_rows_begin): its copies land in arrays this flush may re-allocate
    { const int64_t c_ = ovhip_job_dmvr_rows_collect(j); if (c_ < 0) return (int)c_; }
    const uint32_t stages = pr->stages ? pr->stages : 0xffffffffu;
    const int log2_ctu = pr->log2_ctu_s ? pr->log2_ctu_s : 7;
    if (log2_ctu < 5 || log2_ctu > 7) return ov_fail(ctx, OVHIP_EINVAL, "ovhip_job_flush: log2_ctu_s", hipSuccess);
    const size_t n_ctu = (size_t)((j->w + (1 << log2_ctu) - 1) >> log2_ctu) * ((j->h + (1 << log2_ctu) - 1) >> log2_ctu);
    memset(&j->st, 0, sizeof(j->st));
    j->st.n_ordered_retries = j->n_retries;
    const double t_flush0 = host_now_us();
    j->resident = (stages & OVHIP_STAGE_RESIDENT) && pr->stages;
    ovhip_recorder *rec = j->rec;
    const bool no_upload = j->resident;

    // ---- host: class split of the transform blocks (luma first; big / small), recorder arrays ----
    size_t cls[4] = { 0, 0, 0, 0 }, n_tb = 0, n_coef = 0, n_mc = 0, n_mcx = 0, n_aff = 0, n_side = 0, n_reg = 0, n_ev = 0, n_eh = 0;
    const ovhip_tb_cmd *tb;
    size_t tiny[4][4] = { { 0, 0, 0, 0 }, { 0, 0, 0, 0 }, { 0, 0, 0, 0 }, { 0, 0, 0, 0 } };      // the tiny blocks at the end of each class (a lane per sample)
    tb = ovhip_rec_tb_cmds_split_tiny_(rec, cls, tiny, &n_tb);
    if (!tb && n_tb) return ov_fail(ctx, OVHIP_ENOMEM, "ovhip_rec_tb_cmds_split", hipSuccess);
    const int16_t *coef = ovhip_rec_coefs(rec, &n_coef);
    const ovhip_mc_unit *mc = ovhip_rec_mc_units(rec, &n_mc);
    const ovhip_mc_unit *mcx = ovhip_rec_mcx_units(rec, &n_mcx);
    const ovhip_aff_unit *aff = ovhip_rec_aff_units(rec, &n_aff);
    const int32_t *side = ovhip_rec_aff_side(rec, &n_side);
    const ovhip_lmcs_region *reg = ovhip_rec_lmcs_regions(rec, &n_reg);
    size_t n_ciip = 0;
    const ovhip_ciip_unit *ciip = ovhip_rec_ciip_units(rec, &n_ciip);
    if (n_ciip && !intra)
        return ov_fail(ctx, OVHIP_EUNSUP, "ovhip_job_flush: CIIP blend units recorded but no picture with their intra prediction", hipSuccess);
    // ordered tasks: grouped by CTU for the one-launch pass, or sorted by level for one launch per level
    size_t n_it = 0, n_ictu = 0; uint32_t n_lv = 0; const uint32_t *lv_start = nullptr; const ovhip_ictu *ictu = nullptr;
    const int by_ctu = pr->stages && (stages & OVHIP_STAGE_INTRA_CTU);
    int by_flow = !by_ctu && !(pr->stages && (stages & OVHIP_STAGE_INTRA_LEVELS));
    // a resident replay runs from what the last FULL flush left on the device: after a second pass (ovhip_job_wait: per-level launches,
    // no item list in the parameter block) that is a picture without the flow launch's items
    if (j->resident && by_flow && !j->flow_on_device) by_flow = 0;
    const int by_level = !by_ctu;          // the flow launch also takes the level-sorted list
    const ovhip_itask *it;
    it = by_level ? ovhip_rec_itasks_sorted(rec, &n_it, &lv_start, &n_lv)
                       : ovhip_rec_itasks_by_ctu(rec, pr->log2_ctu_s ? pr->log2_ctu_s : 7, &n_it, &ictu, &n_ictu);
    size_t n_items = 0;
    if (by_flow && n_it) {
        if (j->items_cap < 4 * n_it + 16) {
            pinned_free(nullptr, j->items_host);
            j->items_cap = 8 * n_it + 1024;
            j->items_host = (uint32_t *)pinned_alloc(nullptr, j->items_cap * sizeof(uint32_t));
            if (!j->items_host) { j->items_cap = 0; return ov_fail(ctx, OVHIP_ENOMEM, "ovhip_job_flush: item list", hipSuccess); }
        }
        n_items = ovhip_intra_flow_items(it, n_it, j->items_host, j->items_cap);
        if (!n_items) by_flow = 0;           // more tasks / strips than an item word holds: per-level launches
    }
    if (!it && ovhip_rec_itask_levels(rec)) return ov_fail(ctx, OVHIP_ENOMEM, "ovhip_job_flush: sorting the ordered tasks", hipSuccess);
    if (!by_level) n_lv = ovhip_rec_itask_levels(rec);
    if (!(stages & OVHIP_STAGE_INTRA)) { n_it = 0; n_lv = 0; n_ictu = 0; }
    const int ordered = n_it != 0;       // a picture with an ordered pass keeps its luma in the mapped domain until the pass has run
    if (ordered && !j->res.y) CHK(ovhip_pic_alloc(ctx, j->w, j->h, &j->res));
    if (ordered && (dst->stride_y != j->res.stride_y || dst->stride_c != j->res.stride_c))
        return ov_fail(ctx, OVHIP_EUNSUP, "ovhip_job_flush: pictures with ordered tasks need tight planes (stride = width)", hipSuccess);
    j->st.n_itasks = (uint32_t)n_it; j->st.n_ilevels = n_lv;
    ovhip_dbf_offsets offs;
    const ovhip_dbf_edge *ev = ovhip_rec_dbf_edges(rec, 0, &n_ev, &offs);
    const ovhip_dbf_edge *eh = ovhip_rec_dbf_edges(rec, 1, &n_eh, nullptr);
    j->st.n_tb = (uint32_t)n_tb; j->st.n_mc = (uint32_t)n_mc; j->st.n_mcx = (uint32_t)n_mcx; j->st.n_aff = (uint32_t)n_aff;
    j->st.n_edges_v = (uint32_t)n_ev; j->st.n_edges_h = (uint32_t)n_eh; j->st.n_regions = (uint32_t)n_reg;

    // ---- host: picture-level tables into one staging block ----
    const int sao_on = pr->sao && (stages & OVHIP_STAGE_SAO), alf_on = pr->alf_ctus && (stages & OVHIP_STAGE_ALF);
    if (alf_on && (!pr->alf_luma_coeff || !pr->alf_luma_clip || !pr->alf_chroma_coeff || !pr->alf_chroma_clip || !pr->alf_cc_coeff))
        return ov_fail(ctx, OVHIP_EINVAL, "ovhip_job_flush: ALF tables missing", hipSuccess);
    ParamLayout L;
    size_t o = 0;
    auto put = [&o](size_t bytes) { size_t at = o; o += (bytes + 255) & ~(size_t)255; return at; };
    L.sao = put(sao_on ? n_ctu * sizeof(ovhip_sao_ctu) : 0);
    L.alf_ctus = put(alf_on ? n_ctu * sizeof(ovhip_alf_ctu) : 0);
    L.lcoef = put(alf_on ? 24 * OVHIP_ALF_LUMA_SET_SIZE * 2 : 0);
    L.lclip = put(alf_on ? 24 * OVHIP_ALF_LUMA_SET_SIZE * 2 : 0);
    L.ccoef = put(alf_on ? 8 * 7 * 2 : 0);
    L.cclip = put(alf_on ? 8 * 7 * 2 : 0);
    L.cc = put(alf_on ? 2 * 4 * 8 * 2 : 0);
    L.fwd = put(pr->lmcs ? 2048 : 0);
    L.bwd = put(pr->lmcs ? 2048 : 0);
    // Uploads are what the stream is bound by once the kernels are fast (round 4, tools/micro/h2d_concurrent.py on the GPU box: a 4K
    // picture's 8.7 MB in 10 copies on 16 streams at once = 3600 pictures/s of DMA and nothing else; one copy ~20 us of fixed cost;
    // 16 streams copying at once reach 31-38 GB/s where 1-4 reach 52-56).  So: arrays up to PACK_LIMIT ride in the staging block (one
    // host memcpy each, out of the recorder's page-locked arrays), the few larger ones get a copy of their own.
#ifdef OVHIP_TUNING
    static const size_t PACK_LIMIT = getenv("OVHIP_X_PACK_LIMIT") ? (size_t)atol(getenv("OVHIP_X_PACK_LIMIT")) : (size_t)OVHIP_PACK_LIMIT;
#else
    const size_t PACK_LIMIT = OVHIP_PACK_LIMIT;
#endif
    struct Small { const void *host; size_t bytes; int buf; size_t at; } small[] = {
        { mc, n_mc * sizeof(*mc), B_MC, 0 }, { tb, n_tb * sizeof(*tb), B_TB, 0 }, { coef, n_coef * sizeof(*coef), B_COEF, 0 },
        { it, n_it * sizeof(*it), B_ITASK, 0 }, { ictu, n_ictu * sizeof(*ictu), B_ICTU, 0 },
        { j->items_host, by_flow ? n_items * sizeof(uint32_t) : 0, B_IITEM, 0 },
        { mcx, n_mcx * sizeof(*mcx), B_MCX, 0 }, { ciip, n_ciip * sizeof(*ciip), B_CIIP, 0 }, { aff, n_aff * sizeof(*aff), B_AFF, 0 },
        { side, n_side * sizeof(*side), B_SIDE, 0 }, { reg, n_reg * sizeof(*reg), B_REG, 0 },
        { ev, (stages & OVHIP_STAGE_DBF) ? n_ev * sizeof(*ev) : 0, B_EV, 0 }, { eh, (stages & OVHIP_STAGE_DBF) ? n_eh * sizeof(*eh) : 0, B_EH, 0 },
    };
    static_assert(B_COUNT <= 24, "packed_prev");
    const void *packed[24] = { nullptr };           // device address of a packed array (inside the parameter block)
    for (auto &sm : small) if (sm.bytes && sm.bytes <= PACK_LIMIT && !j->resident) sm.at = put(sm.bytes) + 1;
    L.total = o;
    if (L.total) {
        CHK(pinned_reserve(j, (void **)&j->param_host, &j->param_cap, L.total));
        char *ph = j->param_host;
        for (auto &sm : small) if (sm.at) memcpy(ph + sm.at - 1, sm.host, sm.bytes);
        if (sao_on) memcpy(ph + L.sao, pr->sao, n_ctu * sizeof(ovhip_sao_ctu));
        if (alf_on) {
            memcpy(ph + L.alf_ctus, pr->alf_ctus, n_ctu * sizeof(ovhip_alf_ctu));
            memcpy(ph + L.lcoef, pr->alf_luma_coeff, 24 * OVHIP_ALF_LUMA_SET_SIZE * 2);
            memcpy(ph + L.lclip, pr->alf_luma_clip, 24 * OVHIP_ALF_LUMA_SET_SIZE * 2);
            memcpy(ph + L.ccoef, pr->alf_chroma_coeff, 8 * 7 * 2);
            memcpy(ph + L.cclip, pr->alf_chroma_clip, 8 * 7 * 2);
            memcpy(ph + L.cc, pr->alf_cc_coeff, 2 * 4 * 8 * 2);
        }
        if (pr->lmcs) { memcpy(ph + L.fwd, pr->lmcs->fwd_lut, 2048); memcpy(ph + L.bwd, pr->lmcs->bwd_lut, 2048); }
    }

    const double t_flush1 = host_now_us();
    // ---- H2D (asynchronous DMA out of page-locked memory, in stage order so that prediction can start early) ----
    j->up = (j->resident || no_upload) ? nullptr : upload_stream_for(j);
    if (j->up) {
        // The copies overwrite buffers the previous picture's launches and this picture's eager refinement rows read: the upload stream
        // waits for THEIR events (ev_done: the end of the last flush; ev_rows: the end of an eager pass not collected yet) -- with a frame
        // thread that waits for its picture before it takes the next, both long complete.  NOT for a fresh marker on the picture's own
        // stream: a marker queues behind whatever other streams put into the same hardware queue (17 streams share 4: an I picture's
        // 7 ms ordered pass, a 0.8 ms frame download), and the shared upload stream would hold up every picture behind it (measured:
        // the in-order variant 2309 -> 1476, frames leaving the device 1247 -> 1022 pictures/s with such a marker)
        hipError_t e0 = hipSuccess;
        if (j->flushed) e0 = hipStreamWaitEvent(j->up, j->ev_done, 0);
        if (e0 == hipSuccess && j->rows_pending) e0 = hipStreamWaitEvent(j->up, j->ev_rows, 0);
        if (e0 != hipSuccess) { j->up = nullptr; return ov_fail(ctx, OVHIP_ENODEV, "ovhip_job_flush: upload stream", e0); }
    }
    {
    StageTimer t_(j, OVHIP_TIME_H2D);
    CHK(h2d(j, B_PARAM, j->param_host, L.total));
    if (n_mcx) {
        CHK(dev_reserve(j, B_MV, n_mcx * 16));
        CHK(pinned_reserve(j, (void **)&j->mv_host, &j->mv_cap, n_mcx * 16));
    }
    if (n_reg) CHK(dev_reserve(j, B_SCALE, n_reg * 2));
    // (refined units that went through the eager per-row search are uploaded again with the rest: the list is small and
    // the full kernel repeats the search with the identical result)
    for (auto &sm : small) {
        if (j->resident) continue;
        if (sm.at) packed[sm.buf] = (const char *)j->dev[B_PARAM].p + sm.at - 1;
        else CHK(h2d(j, sm.buf, sm.host, sm.bytes));
    }
    }
    // a resident replay re-uses the placement of the flush before it
    if (no_upload) memcpy(packed, j->packed_prev, sizeof(packed)); else { memcpy(j->packed_prev, packed, sizeof(packed)); j->flow_on_device = by_flow && n_items; }
    auto DEV = [&](int k) -> const void * { return packed[k] ? packed[k] : j->dev[k].p; };
    if (j->up) {
        hipStream_t up = j->up; j->up = nullptr;
        OV_HIP(ctx, hipEventRecord(j->ev_h2d, up));
        OV_HIP(ctx, hipStreamWaitEvent(ctx->stream, j->ev_h2d, 0));
    } else OV_HIP(ctx, hipEventRecord(j->ev_h2d, ctx->stream));
    const double t_flush2 = host_now_us();
    // everything below reads or writes pictures: behind the pictures this one depends on
    for (uint32_t i = 0; i < pr->n_wait_events; ++i)
        if (pr->wait_events && pr->wait_events[i]) {
            if (pr->wait_on_host) OV_HIP(ctx, hipEventSynchronize((hipEvent_t)pr->wait_events[i]));
            else OV_HIP(ctx, hipStreamWaitEvent(ctx->stream, (hipEvent_t)pr->wait_events[i], 0));
        }
    if (pr->before_launch && pr->before_launch(pr->before_launch_user))
        return ov_fail(ctx, OVHIP_EINVAL, "ovhip_job_flush: before_launch callback failed", hipSuccess);
#ifdef OVHIP_TUNING
    static const int X_H2D_HOSTWAIT = getenv("OVHIP_X_H2D_HOSTWAIT") ? atoi(getenv("OVHIP_X_H2D_HOSTWAIT")) : 0;
    static const int X_MV_D2H = getenv("OVHIP_X_MV_D2H") ? atoi(getenv("OVHIP_X_MV_D2H")) : 3;       // 0: skipped, 1: DMA on the picture's stream, 3: stored by a kernel
    if (X_H2D_HOSTWAIT && !j->resident) OV_HIP(ctx, hipEventSynchronize(j->ev_h2d));
#else
    const int X_MV_D2H = 3;
#endif
    const double t_flush3 = host_now_us();
    j->st.host_us_prepare = (uint32_t)(t_flush1 - t_flush0); j->st.host_us_upload = (uint32_t)(t_flush2 - t_flush1);
    j->st.host_us_wait = (uint32_t)(t_flush3 - t_flush2);

    const char *dp = (const char *)j->dev[B_PARAM].p;
    const uint16_t *d_fwd = pr->lmcs ? (const uint16_t *)(dp + L.fwd) : nullptr;
    const uint16_t *d_bwd = pr->lmcs ? (const uint16_t *)(dp + L.bwd) : nullptr;
    const int16_t *d_scales = n_reg ? (const int16_t *)j->dev[B_SCALE].p : nullptr;

    // ---- prediction ----
    if (stages & OVHIP_STAGE_MC) {
        { StageTimer t_(j, OVHIP_TIME_MC);
        CHK(ovhip_mc_launch(ctx, dst, refs, n_refs, (const ovhip_mc_unit *)DEV(B_MC), (uint32_t)n_mc, d_fwd, intra)); }
        j->st.n_launches += n_mc != 0;
        // Nothing on the device reads the refined vectors unless the TMVP entries are asked for: k_mcxa then stores them (one 16-byte
        // store per unit) straight into the page-locked array the host reads -- not even k_store_host's launch is left
        const bool mv_direct = n_mcx && !pr->tmvp_cells && !j->resident && X_MV_D2H == 3;
        if (n_mcx || n_aff) {
            StageTimer t_(j, OVHIP_TIME_MCXA);
            CHK(ovhip_mcxa_launch(ctx, dst, refs, n_refs, (const ovhip_mc_unit *)DEV(B_MCX), (uint32_t)n_mcx,
                                  mv_direct ? j->mv_host : (int32_t *)j->dev[B_MV].p, (const ovhip_aff_unit *)DEV(B_AFF), (uint32_t)n_aff,
                                  (const int32_t *)DEV(B_SIDE), d_fwd));
            j->st.n_launches++;
        }
        if (mv_direct) j->st.d2h_bytes += n_mcx * 16;
        else if (n_mcx && !j->resident && X_MV_D2H) {
            // refined vectors back to the host as early as the stream allows (the decoder's TMVP field needs them)
            // (round 4: as a DMA between k_mcxa and the residual kernels it cost the stream 7 % -- tools/x_mvd2h.sh: 2963 -> 3177 pictures/s)
            if (X_MV_D2H == 3) CHK(store_host(ctx, j->mv_host, j->dev[B_MV].p, n_mcx * 16));
            else OV_HIP(ctx, hipMemcpyAsync(j->mv_host, j->dev[B_MV].p, n_mcx * 16, hipMemcpyDeviceToHost, ctx->stream));
            j->st.d2h_bytes += n_mcx * 16;
        }
        j->n_mv = n_mcx;
        j->n_tmvp = 0;
        if (n_mcx && pr->tmvp_cells) {
            // the same vectors as a delta of the picture's collocated motion plane
            const size_t bytes = 4 * n_mcx * sizeof(ovhip_tmvp_cell);
            CHK(dev_reserve(j, B_TMVP, bytes));
            CHK(pinned_reserve(j, (void **)&j->tmvp_host, &j->tmvp_cap, bytes));
            CHK(ovhip_tmvp_cells_launch(ctx, (const ovhip_mc_unit *)DEV(B_MCX), (uint32_t)n_mcx, (const int32_t *)j->dev[B_MV].p, log2_ctu,
                                        (j->w + (1 << log2_ctu) - 1) >> log2_ctu, (ovhip_tmvp_cell *)j->dev[B_TMVP].p));
            CHK(store_host(ctx, j->tmvp_host, j->dev[B_TMVP].p, bytes));
            j->st.n_launches++; j->st.d2h_bytes += bytes;
            j->n_tmvp = 4 * n_mcx;
        }
        if (n_ciip) { CHK(ovhip_ciip_launch(ctx, dst, intra, (const ovhip_ciip_unit *)DEV(B_CIIP), (uint32_t)n_ciip)); j->st.n_launches++; }
    }
    // ---- the flow launch of the ordered pass: state block, abort word, this picture's epoch (before the residual stage: the
    //      chroma-scale launch prepares the state words as a rider) ----
    int flow_prepared = 0;
    const ovhip_itask *d_it_early = (const ovhip_itask *)DEV(B_ITASK);
    if ((stages & OVHIP_STAGE_INTRA) && by_flow && n_items) {
        if (!j->d_flow) {
            const size_t words = ovhip_intra_flow_words(j->w, j->h);
            OV_HIP(ctx, hipMalloc((void **)&j->d_flow, words * sizeof(uint32_t)));
            OV_HIP(ctx, hipMemsetAsync(j->d_flow, 0, words * sizeof(uint32_t), ctx->stream));
        }
        if (!j->abort_host) {
            j->abort_host = (uint32_t *)pinned_alloc(nullptr, 64);
            if (!j->abort_host) return ov_fail(ctx, OVHIP_ENOMEM, "ovhip_job_flush: pinned abort word", hipSuccess);
            *j->abort_host = 0;
        }
        if (++j->epoch >= 0x7ffffff0u) j->epoch = 1;
        // test hook (ovhip_job_test_abort_next_flow): the abort word is set on the device BEFORE the launch, so every item that has to
        // wait for another gives up at once -- a real abandoned first pass, picture incomplete and partly tagged
        if (j->test_abort && j->n_retries == 0) {
            j->test_abort = 0; j->test_abort_seen = 1;
            OV_HIP(ctx, hipMemsetAsync(j->d_flow, 0xff, sizeof(uint32_t), ctx->stream));
            *(volatile uint32_t *)j->abort_host = 1;
        }
    }
    // ---- residual: luma blocks, chroma-scale derivation on the reconstructed luma, chroma blocks (+ inverse mapping) ----
    if (stages & OVHIP_STAGE_ITX) {
        const ovhip_tb_cmd *d_tb = (const ovhip_tb_cmd *)DEV(B_TB);
        const int16_t *d_coef = (const int16_t *)DEV(B_COEF);
        const uint32_t t_luma[4] = { (uint32_t)tiny[1][0], (uint32_t)tiny[1][1], (uint32_t)tiny[1][2], (uint32_t)tiny[1][3] };
        const uint32_t t_chroma[4] = { (uint32_t)tiny[3][0], (uint32_t)tiny[3][1], (uint32_t)tiny[3][2], (uint32_t)tiny[3][3] };
        if (cls[0] + cls[1]) {
            StageTimer t_(j, OVHIP_TIME_ITX_LUMA);
            CHK(ovhip_itx_launch_ex_(ctx, dst, ordered ? &j->res : nullptr, d_tb, (uint32_t)cls[0], (uint32_t)cls[1], t_luma, d_coef, nullptr, nullptr));
            j->st.n_launches++;
        }
        if (n_reg) {
            if (!pr->lmcs) return ov_fail(ctx, OVHIP_EINVAL, "ovhip_job_flush: chroma-scale regions recorded without LMCS tables", hipSuccess);
            StageTimer t_(j, OVHIP_TIME_LMCS_SCALE);
            if ((stages & OVHIP_STAGE_INTRA) && by_flow && n_items) {
                CHK(ovhip_lmcs_scale_prepare_launch(ctx, dst, (const ovhip_lmcs_region *)DEV(B_REG), (uint32_t)n_reg, pr->lmcs,
                                                    (int16_t *)j->dev[B_SCALE].p, d_it_early, (uint32_t)n_it, j->d_flow, j->epoch));
                flow_prepared = 1;
            } else
            CHK(ovhip_lmcs_scale_launch(ctx, dst, (const ovhip_lmcs_region *)DEV(B_REG), (uint32_t)n_reg, pr->lmcs,
                                        (int16_t *)j->dev[B_SCALE].p));
            j->st.n_launches++;
        }
        const ovhip_tb_cmd *d_tbc = d_tb + cls[0] + cls[1];
        StageTimer t_(j, OVHIP_TIME_ITX_CHROMA);
        if (pr->lmcs && cls[3] && !ordered) {
            CHK(ovhip_itx_launch_ex_(ctx, dst, nullptr, d_tbc, (uint32_t)cls[2], (uint32_t)cls[3], t_chroma, d_coef, d_scales, d_bwd));
            j->st.n_launches++;
        } else {
            if (cls[2] + cls[3]) {
                CHK(ovhip_itx_launch_ex_(ctx, dst, ordered ? &j->res : nullptr, d_tbc, (uint32_t)cls[2], (uint32_t)cls[3], t_chroma, d_coef, d_scales, nullptr));
                j->st.n_launches++;
            }
            if (pr->lmcs && !ordered) { CHK(ovhip_lmcs_inverse_launch(ctx, dst, d_bwd)); j->st.n_launches++; }
        }
    }
    // ---- ordered pass: one launch per level (the launch boundary is the inter-level synchronisation) or, on request, ONE launch
    // (a workgroup per CTU with tasks, CTU samples in LDS, neighbour CTUs chained by flags); then the inverse luma mapping ----
    if (ordered) {
        StageTimer t_(j, OVHIP_TIME_INTRA);
        const ovhip_itask *d_it = (const ovhip_itask *)DEV(B_ITASK);
        if (!by_level) {
            if (!j->d_sync) {
                const size_t words = ovhip_intra_sync_words(j->w, j->h, 5);          // the smallest CTU: enough for every size
                OV_HIP(ctx, hipMalloc((void **)&j->d_sync, words * sizeof(uint32_t)));
                OV_HIP(ctx, hipMemsetAsync(j->d_sync, 0, words * sizeof(uint32_t), ctx->stream));
            }
            if (!j->abort_host) {
                j->abort_host = (uint32_t *)pinned_alloc(nullptr, 64);
                if (!j->abort_host) return ov_fail(ctx, OVHIP_ENOMEM, "ovhip_job_flush: pinned abort word", hipSuccess);
                *j->abort_host = 0;
            }
            if (++j->epoch >= 0x7ffffff0u) j->epoch = 1;
            CHK(ovhip_intra_ctu_launch(ctx, dst, &j->res, d_it, (const ovhip_ictu *)DEV(B_ICTU), (uint32_t)n_ictu,
                                       (const ovhip_lmcs_region *)DEV(B_REG), pr->lmcs, (int16_t *)j->dev[B_SCALE].p, log2_ctu, j->d_sync, j->epoch,
                                       j->abort_host));
            j->st.n_launches++;
        }
        if (by_flow && n_items) {
            // ONE launch of W persistent workers (k_intra_flow): worker b takes the items b, b + W, ... in level order.  W bounds the
            // pollers of the launch, and the launches in flight together must fit the device for the forward-progress argument (an item
            // waits only for lower items; the lowest unfinished item's worker is resident or will be): the kernel holds 109 VGPRs = 16
            // waves per compute unit, HIP runs the process's streams on 4 hardware queues, so W = 16 CUs / 4 = 1024 on MI355X.
            // Measured on the stream of bench.py (pictures/s, second passes): one workgroup per item 2834-2890 / 0 (the device full of
            // the pollers of levels far ahead); W = 512 3280-3320 / 0; 1024 3353-3490 / 0; 2048 1561 / 9; 4096 35 / 1168 -- above
            // the bound the launches starve each other, exactly as the argument says.  Alone, a B picture's wide levels want more
            // workers (its pass: 104 us with a workgroup per item, 156 with 1024 workers, 229 with 512); beside other pictures that
            // does not show.  GPU_MAX_HW_QUEUES = 8 with W = 512: 3000-3060 / 0, no gain.  (OVHIP_FLOW_WORKERS: a tuning knob, read once)
#ifdef OVHIP_TUNING
            static const long WORKERS = getenv("OVHIP_FLOW_WORKERS") ? atol(getenv("OVHIP_FLOW_WORKERS")) : -1;
#else
            static const long WORKERS = -1;
#endif
            //  the 4 x CUs default assumes 4 hardware queues and 16 resident waves per CU; where that does not hold (another
            // GPU_MAX_HW_QUEUES, another process on the GPU) flow launches starve each other and pictures fall into second passes:
            // every abandoned launch halves the default for the launches that follow (g_flow_shift, down to CUs / 4)
            // Round 4: 6 x CUs (1536).  The argument above sized 4 x CUs for four launches of the full count side by side; since the
            // widest-level rule below an I picture's launch holds ~256 workers, and B launches of 1536 were never abandoned in 4600
            // pictures (tools/sweep_bpic_workers.sh, interleaved, six runs each: 1024 workers 3220 pictures/s, 1536 3308, 2048 2-10
            // second passes per run); an abandoned launch still halves the default (g_flow_shift).
            int n_workers = pr->flow_workers ? (int)pr->flow_workers : (WORKERS >= 0 ? (int)WORKERS : (6 * ctx->num_cus) >> flow_shift_of(ctx->device));
            if (!pr->flow_workers && WORKERS < 0) {
                // No more workers than the picture's widest level can use (round 4): a worker beyond that only ever holds an item that is
                // levels ahead of the front -- and its wave slot, registers and LDS are then missing to the kernels of the pictures beside
                // this one for as long as the pass runs.  An I picture has ~100 items per level and runs 5 ms: 4 x CUs = 1024 workers
                // gave 2830-2930 pictures/s on bench.py's stream, 128-512 gave 3000-3150 (tools/sweep_ipic_workers.sh); a B picture's
                // levels are 1000+ items wide and keep the full count.
                size_t widest = 0, run = 0;
                for (size_t q = 0; q < n_items; ++q) {
                    run = (q && it[j->items_host[q] & 0xffffff].level == it[j->items_host[q - 1] & 0xffffff].level) ? run + 1 : 1;
                    if (run > widest) widest = run;
                }
                const int want = (int)((2 * widest + 63) & ~(size_t)63);
                if (want < n_workers) n_workers = want < 64 ? 64 : want;
            }
            if (n_workers < 1) n_workers = 1;
            CHK(ovhip_intra_flow_launch(ctx, dst, &j->res, d_it, (uint32_t)n_it, (const uint32_t *)DEV(B_IITEM), (uint32_t)n_items,
                                        (const ovhip_lmcs_region *)DEV(B_REG), pr->lmcs, (int16_t *)j->dev[B_SCALE].p, log2_ctu, j->d_flow, j->epoch,
                                        j->abort_host, !flow_prepared, n_workers));
            j->st.n_launches += 1 + !flow_prepared;
            j->flow_launched = 1;
            j->st.flow_shift = (uint32_t)flow_shift_of(ctx->device);
        }
        for (uint32_t l = 0; by_level && !by_flow && l < n_lv; ++l) {
            const uint32_t a = lv_start[l], b = lv_start[l + 1];
            if (b > a) {
                CHK(ovhip_intra_level_launch(ctx, dst, &j->res, d_it + a, b - a, (const ovhip_lmcs_region *)DEV(B_REG), pr->lmcs,
                                             (int16_t *)j->dev[B_SCALE].p, log2_ctu, ovhip_intra_level_geom(it + a, b - a)));
                j->st.n_launches++;
            }
        }
        const bool inverse = pr->lmcs && (stages & OVHIP_STAGE_ITX);
        // the flow launches leave a hand-over bit in what they wrote: dropped by the inverse mapping's launch, or by one of its own
        if (inverse) { CHK(ovhip_lmcs_inverse_untag_launch(ctx, dst, d_bwd, d_it, by_flow && n_items ? (uint32_t)n_it : 0u)); j->st.n_launches++; }
        else if (by_flow && n_items) { CHK(ovhip_intra_flow_untag_launch(ctx, dst, d_it, (uint32_t)n_it, 1)); j->st.n_launches++; }
    }
    // ---- in-loop filters ----
    if (stages & OVHIP_STAGE_DBF) {
        StageTimer t_(j, OVHIP_TIME_DBF);
        CHK(ovhip_dbf_launch_edges_ex(ctx, dst, (const ovhip_dbf_edge *)DEV(B_EV), (uint32_t)n_ev,
                                      (const ovhip_dbf_edge *)DEV(B_EH), (uint32_t)n_eh, &offs));
        j->st.n_launches += (n_ev != 0) + (n_eh != 0);
    }
    // SAO writes tmp, ALF writes dst; with only one of the two the result is copied back so that dst always holds it
    if (sao_on) {
        StageTimer t_(j, OVHIP_TIME_SAO);
        CHK(ovhip_sao_launch(ctx, &j->tmp, dst, (const ovhip_sao_ctu *)(dp + L.sao), log2_ctu));
        j->st.n_launches++;
    }
    if (alf_on) {
        CHK(dev_reserve(j, B_CLASS, (size_t)((j->w + 3) / 4) * ((j->h + 3) / 4)));
        ovhip_alf_pic ap;
        ap.ctus = (const ovhip_alf_ctu *)(dp + L.alf_ctus);
        ap.luma_coeff = (const int16_t *)(dp + L.lcoef); ap.luma_clip = (const int16_t *)(dp + L.lclip);
        ap.chroma_coeff = (const int16_t *)(dp + L.ccoef); ap.chroma_clip = (const int16_t *)(dp + L.cclip);
        ap.cc_coeff = (const int16_t *)(dp + L.cc);
        ap.class_scratch = (uint8_t *)j->dev[B_CLASS].p;
        ap.log2_ctu_s = log2_ctu;
        if (!sao_on) CHK(copy_pic(ctx, &j->tmp, dst));
        StageTimer t_(j, OVHIP_TIME_ALF);
        CHK(ovhip_alf_launch(ctx, dst, &j->tmp, &ap));
        j->st.n_launches++;
    } else if (sao_on) {
        CHK(copy_pic(ctx, dst, &j->tmp));
    }
    OV_HIP(ctx, hipEventRecord(j->ev_done, ctx->stream));
    j->flushed = 1;
    j->st.host_us_launch = (uint32_t)(host_now_us() - t_flush3);
    return OVHIP_OK;
}

} // extern "C"

// =====================================================================================================================================
// Band-wise submission: a picture enters the device while it is still being parsed.
//
// The reference reconstructs a CTU row right after parsing it and reports it (slicedec.c:815-975, dpb.c:1309-1323); a picture that
// references it runs a few rows behind (rcn_inter.c:131-146).  ovhip_job_flush takes a picture when its parse has ENDED, so every level
// of a GOP's reference hierarchy cost a whole parse plus a whole launch chain.  Here the recorder's arrays are cut at CTU-row bands:
//
//     ovhip_job_band(k)   = upload of band k's slices (ONE copy out of a staging block) + recon(k) + tail(k)
//     recon(k)            = MC -> refined / affine MC -> luma residual -> chroma-scale regions -> chroma residual -> ordered pass,
//                           the picture-wide launches over the band's slice of every list
//     tail(k)             = inverse luma mapping of the band's rows (+ un-tag) -> deblocking of the band's edge lists (V then H)
//                           -> SAO rows [.., s1) -> ALF rows [.., a1)
//
// The horizontal edge on the boundary between bands k - 1 and k belongs to band k's lists and changes up to 7 rows above it, so after
// tail(k) the rows < end_k - 8 are final for the deblocking; SAO then runs up to the last multiple of 8 rows below that (its edge
// classes read one row further), ALF up to the last multiple of 8 that keeps its 3-row reach and the classification windows inside the
// SAO output: s1 = end_k - 16, a1 = end_k - 24 for CTU-row bands (the tiles a window cuts are staged whole, rows outside it are not
// stored).  Intra prediction of band k + 1 reads the UNFILTERED, still mapped bottom row of band k (the reference keeps saved lines for
// this, rcn_ctu.c:246-510): k_band_row below sets it aside before tail(k) and puts it back for recon(k + 1).  Every launch reads exactly
// the samples the picture-wide launch reads, so the fixtures' parity carries over (tests/test_gpu_bands.py: bands of one CTU row, of
// two, of three and one band = the whole picture give identical pictures).
//
// Indices inside the commands (coefficient / side-arena offsets, region numbers) stay what the recorder wrote: the band's slice is
// addressed through a pointer moved back by the slice's first index.  The flow launches of the bands never wait for an item of another
// launch (stream order), and the workers of all band launches in flight are accounted against the device's wave slots (flow budget
// below), so that every launch's workers can be resident: the bounded waits cannot expire by starvation.  If one does anyway the
// picture FAILS (the rows already published to readers cannot be taken back); ovhip_job_wait reports it.
// =====================================================================================================================================
extern "C" void ovhip_rec_tb_split_range_(const ovhip_recorder *r, size_t first, size_t n, ovhip_tb_cmd *out, size_t counts[4], size_t tiny[4][4]);
extern "C" int  ovhip_rec_itasks_sorted_range_(const ovhip_recorder *r, size_t first, size_t n, ovhip_itask *out, uint32_t *level_start, size_t cap, uint32_t *n_levels);
extern "C" int  ovhip_sao_launch_rows(ovhip_ctx *ctx, const ovhip_pic *dst, const ovhip_pic *src, const ovhip_sao_ctu *d_params, int32_t log2_ctu_s, int32_t row0, int32_t row1);
extern "C" int  ovhip_alf_launch_rows(ovhip_ctx *ctx, const ovhip_pic *dst, const ovhip_pic *src, const ovhip_alf_pic *alf, int32_t row0, int32_t row1);

enum { MAX_BANDS = 96, ARENA_CHUNKS = 16, BAND_LEVELS = 4096 };
struct ArenaChunk { char *host, *dev; size_t cap, used; };
struct BandRec {
    int32_t row0, row1;
    ovhip_band_counts c0, c1;
    const ovhip_itask *d_it; uint32_t n_it; int tagged;          // the band's ordered tasks on the device; tagged: a flow launch wrote them
    const ovhip_dbf_edge *d_ev, *d_eh; uint32_t n_ev, n_eh;
    int flow_charge;                                             // workers charged to the device's flow budget until ev_recon is seen
    int32_t rows_final;                                          // picture rows final once this band's tail has run
};
struct BandState {
    int active, n, tails, closed, failed;
    ovhip_pic dst;
    ovhip_band_counts cur;
    int32_t row_prev, dbf_rows, sao_rows, alf_rows, rows_final;
    int log2_ctu, sao_on, alf_on, filters_latched;
    uint32_t stages;
    const uint16_t *d_fwd, *d_bwd; int lmcs_up;                   // LMCS tables: in the first band's block
    ovhip_alf_pic d_alf; int alf_up;                              // ALF picture-level tables: in the block of the first call that has them
    ovhip_lmcs_luts luts; int have_luts;
    ovhip_dbf_offsets offs;
    ArenaChunk chunk[ARENA_CHUNKS]; int n_chunks, cur_chunk;
    BandRec band[MAX_BANDS];
    hipEvent_t ev_recon[MAX_BANDS], ev_tail[MAX_BANDS];
    uint32_t level_start[BAND_LEVELS + 2];
    void *last_event; int32_t last_rows;                          // what ovhip_job_band_progress hands out
    uint16_t *keep; int keep_valid;                               // device: the previous band's bottom row before / after its filters (4 w samples)
};

// The bottom row of a band as the band below must see it.  Intra prediction, the cross-component model and the chroma-scale
// derivation of band k + 1 read the row above it UNFILTERED and in the mapped domain (the reference keeps saved lines for this,
// rcn_ctu.c:246-510), but band k's filters run with band k, so that its rows are final one band earlier: the row (luma row end - 1,
// chroma rows end / 2 - 1) is set aside before the filters and put back for the time band k + 1 is reconstructed.  Nobody else reads
// it meanwhile: rows within 8 of a band's end are not final -- not posted to readers, not reached by SAO / ALF -- before the
// deblocking of the band below has run.  mode 0: keep_unf <- picture; 1: keep_fil <- picture, picture <- keep_unf (& mask);
// 2: picture <- keep_fil.  keep = [unf: Y w | Cb w/2 | Cr w/2][fil: the same].
__global__ __launch_bounds__(256) void k_band_row(ovhip_pic pic, uint16_t *keep, int row_y, int mode, unsigned mask)
{
    const int i = blockIdx.x * 256 + threadIdx.x, w = pic.w, wc = w >> 1;
    if (i >= 2 * w) return;
    uint16_t *p = i < w ? pic.y + (size_t)row_y * pic.stride_y + i
                        : (i < w + wc ? pic.cb + (size_t)(row_y >> 1) * pic.stride_c + (i - w) : pic.cr + (size_t)(row_y >> 1) * pic.stride_c + (i - w - wc));
    uint16_t *unf = keep + i, *fil = keep + 2 * w + i;
    if (mode == 0) *unf = *p;
    else if (mode == 1) { *fil = *p; *p = (uint16_t)(*unf & mask); }
    else *p = *fil;
}
static int band_row(ovhip_job *j, const ovhip_pic *pic, int row_y, int mode, unsigned mask)
{
    hipLaunchKernelGGL(k_band_row, dim3((2 * pic->w + 255) / 256), dim3(256), 0, j->ctx->stream, *pic, j->bs->keep, row_y, mode, mask);
    OV_LAUNCH_CHECK(j->ctx, "k_band_row");
    j->st.n_launches++;
    return OVHIP_OK;
}


// ---- flow budget: the workers of the band flow launches in flight on a device never exceed the wave slots k_intra_flow can hold
// there (16 one-wave workgroups per compute unit, three quarters of them: the whole-picture launches of I pictures run beside).  A
// launch is charged when it is enqueued and released when its job sees the event behind it -- so every charged launch's workers can
// be resident together and no worker waits for a workgroup that cannot start.
static int g_band_flow_inflight[FLOW_DEVS];
static int band_flow_take(ovhip_ctx *ctx, int want)
{
    const int fd = ctx->device & (FLOW_DEVS - 1), cap = 12 * ctx->num_cus;
    int cur = __atomic_load_n(&g_band_flow_inflight[fd], __ATOMIC_RELAXED);
    for (;;) {
        int give = cap - cur < want ? cap - cur : want;
        give &= ~63;
        if (give < 64) return 0;
        if (__atomic_compare_exchange_n(&g_band_flow_inflight[fd], &cur, cur + give, true, __ATOMIC_RELAXED, __ATOMIC_RELAXED)) return give;
    }
}
static void band_flow_give(ovhip_ctx *ctx, int n) { if (n) __atomic_fetch_sub(&g_band_flow_inflight[ctx->device & (FLOW_DEVS - 1)], n, __ATOMIC_RELAXED); }

// the charges of bands whose reconstruction has completed go back (in order: a later band's launches are behind the earlier ones')
static void band_flow_reclaim(ovhip_job *j, int wait)
{
    BandState *bs = j->bs;
    for (int b = 0; bs && b < bs->n; ++b) {
        BandRec &B = bs->band[b];
        if (!B.flow_charge) continue;
        if (wait) (void)hipEventSynchronize(bs->ev_recon[b]);
        else if (hipEventQuery(bs->ev_recon[b]) != hipSuccess) { (void)hipGetLastError(); break; }
        band_flow_give(j->ctx, B.flow_charge);
        B.flow_charge = 0;
    }
}

static int band_active(const ovhip_job *j) { return j->bs && j->bs->active; }

static void band_free(ovhip_job *j)
{
    BandState *bs = j->bs;
    if (!bs) return;
    band_flow_reclaim(j, 1);
    for (int i = 0; i < bs->n_chunks; ++i) { pinned_free(nullptr, bs->chunk[i].host); if (bs->chunk[i].dev) (void)hipFree(bs->chunk[i].dev); }
    for (int i = 0; i < MAX_BANDS; ++i) { if (bs->ev_recon[i]) (void)hipEventDestroy(bs->ev_recon[i]); if (bs->ev_tail[i]) (void)hipEventDestroy(bs->ev_tail[i]); }
    if (bs->keep) (void)hipFree(bs->keep);
    free(bs);
    j->bs = nullptr;
}

// ovhip_job_begin: the previous picture's uploads have ended (ev_h2d); the arena starts over -- as ONE chunk if the last picture needed several
static int band_reset(ovhip_job *j)
{
    BandState *bs = j->bs;
    if (!bs) return OVHIP_OK;
    band_flow_reclaim(j, 1);
    if (bs->n_chunks > 1) {
        size_t total = 0;
        if (j->flushed) OV_HIP(j->ctx, hipEventSynchronize(j->ev_done));        // the launches read the device halves
        for (int i = 0; i < bs->n_chunks; ++i) { total += bs->chunk[i].cap; pinned_free(nullptr, bs->chunk[i].host); if (bs->chunk[i].dev) (void)hipFree(bs->chunk[i].dev); }
        memset(bs->chunk, 0, sizeof(bs->chunk));
        bs->n_chunks = 0;
        total += total / 4;
        bs->chunk[0].host = (char *)pinned_alloc(nullptr, total);
        if (!bs->chunk[0].host) return ov_fail(j->ctx, OVHIP_ENOMEM, "band arena (pinned)", hipSuccess);
        if (hipMalloc((void **)&bs->chunk[0].dev, total) != hipSuccess) { pinned_free(nullptr, bs->chunk[0].host); bs->chunk[0].host = nullptr; return ov_fail(j->ctx, OVHIP_ENOMEM, "band arena (device)", hipSuccess); }
        bs->chunk[0].cap = total; bs->n_chunks = 1;
    }
    for (int i = 0; i < bs->n_chunks; ++i) bs->chunk[i].used = 0;
    bs->cur_chunk = 0;
    bs->active = 0; bs->n = 0; bs->tails = 0; bs->closed = 0; bs->failed = 0;
    return OVHIP_OK;
}

// a block of `bytes` in the arena: the same offset in a page-locked host chunk and in its device twin (one copy moves it)
static int arena_take(ovhip_job *j, size_t bytes, char **host, char **dev)
{
    BandState *bs = j->bs;
    bytes = (bytes + 255) & ~(size_t)255;
    for (;;) {
        if (bs->cur_chunk < bs->n_chunks) {
            ArenaChunk &c = bs->chunk[bs->cur_chunk];
            if (c.cap - c.used >= bytes) { *host = c.host + c.used; *dev = c.dev + c.used; c.used += bytes; return OVHIP_OK; }
            bs->cur_chunk++;
            continue;
        }
        if (bs->n_chunks == ARENA_CHUNKS) return ov_fail(j->ctx, OVHIP_ENOMEM, "band arena: too many chunks", hipSuccess);
        // first chunk: ~ a 4K B picture's arrays (they sum to 9 MB); later ones double
        size_t cap = bs->n_chunks ? 2 * bs->chunk[bs->n_chunks - 1].cap : ((size_t)j->w * j->h * 3 / 2 < ((size_t)4 << 20) ? (size_t)4 << 20 : (size_t)j->w * j->h * 3 / 2);
        while (cap < bytes) cap *= 2;
        ArenaChunk &c = bs->chunk[bs->n_chunks];
        c.host = (char *)pinned_alloc(nullptr, cap);
        if (!c.host) return ov_fail(j->ctx, OVHIP_ENOMEM, "band arena (pinned)", hipSuccess);
        if (hipMalloc((void **)&c.dev, cap) != hipSuccess) { pinned_free(nullptr, c.host); c.host = nullptr; return ov_fail(j->ctx, OVHIP_ENOMEM, "band arena (device)", hipSuccess); }
        c.cap = cap; c.used = 0;
        bs->n_chunks++;
    }
}

static int band_wait_done(ovhip_job *j)
{
    BandState *bs = j->bs;
    band_flow_reclaim(j, 1);
    if (j->abort_host && *(volatile uint32_t *)j->abort_host) {
        *(volatile uint32_t *)j->abort_host = 0;
        if (j->d_flow) (void)hipMemset(j->d_flow, 0, sizeof(uint32_t));
        bs->failed = 1;
        return ov_fail(j->ctx, OVHIP_ELAUNCH, "band-wise picture: a bounded wait of the ordered pass expired (picture incomplete; its bands may have been read)", hipSuccess);
    }
    return bs->failed ? ov_fail(j->ctx, OVHIP_ELAUNCH, "band-wise picture failed", hipSuccess) : OVHIP_OK;
}

static inline int32_t floor8(int32_t v) { return v <= 0 ? 0 : v & ~7; }

/* (internal, ovvc_stream.c) The device's upload streams out of the hardware queue of `keep_clear`'s stream -- the look-ahead thread's: the event
 * behind a picture's copies is a packet of the upload stream's hardware queue, and behind an I picture's ordered pass it would arrive
 * milliseconds late for every picture whose copies follow.  An upload stream that shares the queue is replaced (the old one stays
 * alive: events may refer to it).  Returns the number still sharing, or < 0. */
extern "C" int ovhip_upload_streams_clear_of_(ovhip_ctx *keep_clear)
{
    if (!keep_clear) return OVHIP_EINVAL;
    const int n = upload_stream_count(), dev = keep_clear->device;
    if (n <= 0 || dev < 0 || dev >= UP_DEVS) return 0;
    OV_DEVICE(keep_clear);
    int sharing = 0;
    for (int k = 0; k < n; ++k) {
        pthread_mutex_lock(&g_up_mtx);
        if (!g_up[dev][k] && hipStreamCreateWithFlags(&g_up[dev][k], hipStreamNonBlocking) != hipSuccess) g_up[dev][k] = nullptr;
        hipStream_t st = g_up[dev][k];
        pthread_mutex_unlock(&g_up_mtx);
        if (!st) continue;
        for (int tries = 0; tries < 12; ++tries) {
            ovhip_ctx tmp; memset(&tmp, 0, sizeof(tmp)); tmp.device = dev; tmp.stream = st;
            const int q = ovhip_ctx_shares_queue(keep_clear, &tmp);
            if (q != 1) { st = q == 0 ? st : nullptr; break; }
            hipStream_t fresh = nullptr;
            if (hipStreamCreateWithFlags(&fresh, hipStreamNonBlocking) != hipSuccess) { st = nullptr; break; }
            st = fresh;
            if (tries == 11) st = nullptr;
        }
        if (!st) { ++sharing; continue; }
        pthread_mutex_lock(&g_up_mtx);
        g_up[dev][k] = st;
        pthread_mutex_unlock(&g_up_mtx);
    }
    return sharing;
}

extern "C" int ovhip_job_band_active(const ovhip_job *j) { return j && band_active(j); }

// the band arena's first chunk and the bottom-row buffer now, not in the first band of the job's first band-wise picture (what
// ovhip_frame_set_band_mode(f, 1) asks for: a frame thread's job is sized when it is created, see ovhip_job_reserve_for_picture)
extern "C" int ovhip_job_band_reserve(ovhip_job *j)
{
    if (!j) return OVHIP_EINVAL;
    OV_DEVICE(j->ctx);
    if (!j->bs) {
        j->bs = (BandState *)calloc(1, sizeof(BandState));
        if (!j->bs) return OVHIP_ENOMEM;
    }
    BandState *bs = j->bs;
    if (!bs->n_chunks) {
        char *h = nullptr, *d = nullptr;
        CHK(arena_take(j, 1, &h, &d));
        bs->chunk[0].used = 0; bs->cur_chunk = 0;
    }
    if (!bs->keep) OV_HIP(j->ctx, hipMalloc((void **)&bs->keep, (size_t)4 * j->w * sizeof(uint16_t)));
    if (!j->res.y) CHK(ovhip_pic_alloc(j->ctx, j->w, j->h, &j->res));
    if (!j->d_flow) {
        const size_t words = ovhip_intra_flow_words(j->w, j->h);
        OV_HIP(j->ctx, hipMalloc((void **)&j->d_flow, words * sizeof(uint32_t)));
        OV_HIP(j->ctx, hipMemsetAsync(j->d_flow, 0, words * sizeof(uint32_t), j->ctx->stream));
    }
    return OVHIP_OK;
}

// 1: the reconstruction of the last band submitted is still running on the device.  A caller that is ahead of the device leaves its
// next band to a later hook (it then covers more CTU rows): the launches stay few and full when the device is the slower side -- and
// an I picture, whose ordered pass is one dependency chain per band, keeps its wavefront across as many rows as the parse has delivered
extern "C" int ovhip_job_band_busy(ovhip_job *j)
{
    if (!j || !band_active(j) || !j->bs->n) return 0;
    (void)hipSetDevice(j->ctx->device);
    if (hipEventQuery(j->bs->ev_recon[j->bs->n - 1]) == hipSuccess) return 0;
    (void)hipGetLastError();
    return 1;
}

extern "C" int ovhip_job_band_progress(ovhip_job *j, int32_t *rows_final, void **event, const volatile uint32_t **abort_word)
{
    if (!j || !rows_final) return OVHIP_EINVAL;
    *rows_final = 0;
    if (event) *event = nullptr;
    if (abort_word) *abort_word = j->abort_host;
    if (!band_active(j)) return OVHIP_OK;
    *rows_final = j->bs->last_rows;
    if (event) *event = j->bs->last_event;
    return OVHIP_OK;
}

extern "C" int ovhip_job_band(ovhip_job *j, const ovhip_pic *dst, const ovhip_pic *refs, uint32_t n_refs, const ovhip_job_params *pr,
                   const ovhip_band_counts *upto, int32_t row_end, int32_t last)
{
    if (!j || !dst || !pr) return OVHIP_EINVAL;
    ovhip_ctx *ctx = j->ctx;
    OV_DEVICE(ctx);
    if (dst->w != j->w || dst->h != j->h) return ov_fail(ctx, OVHIP_EINVAL, "ovhip_job_band: picture size differs from the job's", hipSuccess);
    if (dst->stride_y != j->tmp.stride_y || dst->stride_c != j->tmp.stride_c)
        return ov_fail(ctx, OVHIP_EUNSUP, "ovhip_job_band: tight planes only (stride = width)", hipSuccess);
    if (!j->bs) {
        j->bs = (BandState *)calloc(1, sizeof(BandState));
        if (!j->bs) return OVHIP_ENOMEM;
    }
    BandState *bs = j->bs;
    ovhip_recorder *rec = j->rec;
    const int log2_ctu = pr->log2_ctu_s ? pr->log2_ctu_s : 7;
    if (log2_ctu < 5 || log2_ctu > 7) return ov_fail(ctx, OVHIP_EINVAL, "ovhip_job_band: log2_ctu_s", hipSuccess);
    if (bs->closed) return ov_fail(ctx, OVHIP_EINVAL, "ovhip_job_band: the picture's last band was already submitted", hipSuccess);
    if (last) row_end = j->h;
    if (row_end > j->h) row_end = j->h;
    const int first_band = !bs->active;
    if (first_band) {
        memset(&j->st, 0, sizeof(j->st));
        bs->active = 1; bs->n = 0; bs->tails = 0; bs->closed = 0; bs->failed = 0;
        bs->dst = *dst; bs->row_prev = 0; bs->dbf_rows = bs->sao_rows = bs->alf_rows = bs->rows_final = 0;
        memset(&bs->cur, 0, sizeof(bs->cur));
        bs->log2_ctu = log2_ctu; bs->filters_latched = 0; bs->lmcs_up = 0; bs->alf_up = 0; bs->have_luts = 0;
        bs->stages = pr->stages ? pr->stages : 0xffffffffu;
        bs->last_event = nullptr; bs->last_rows = 0; bs->keep_valid = 0;
        if (!bs->keep) OV_HIP(ctx, hipMalloc((void **)&bs->keep, (size_t)4 * j->w * sizeof(uint16_t)));
        j->again.valid = 0; j->n_retries = 0;       // (n_mv / n_tmvp: the eager DMVR rows' -- a pass may have run before the first band)
        if (!j->res.y) CHK(ovhip_pic_alloc(ctx, j->w, j->h, &j->res));
        if (!j->d_flow) {
            const size_t words = ovhip_intra_flow_words(j->w, j->h);
            OV_HIP(ctx, hipMalloc((void **)&j->d_flow, words * sizeof(uint32_t)));
            OV_HIP(ctx, hipMemsetAsync(j->d_flow, 0, words * sizeof(uint32_t), ctx->stream));
        }
        if (!j->abort_host) {
            j->abort_host = (uint32_t *)pinned_alloc(nullptr, 64);
            if (!j->abort_host) return ov_fail(ctx, OVHIP_ENOMEM, "ovhip_job_band: pinned abort word", hipSuccess);
            *j->abort_host = 0;
        }
        if (++j->epoch >= 0x7ffffff0u) j->epoch = 1;
        CHK(dev_reserve(j, B_SCALE, 65536));                               // 32767 regions at most (ovhip_rec_lmcs_region)
        CHK(dev_reserve(j, B_CLASS, (size_t)((j->w + 3) / 4) * ((j->h + 3) / 4)));
    }
    if (!(dst->y == bs->dst.y) || log2_ctu != bs->log2_ctu) return ov_fail(ctx, OVHIP_EINVAL, "ovhip_job_band: another picture than the first band's", hipSuccess);
    if (row_end < bs->row_prev || (!last && (row_end & ((1 << log2_ctu) - 1)) && row_end != j->h))
        return ov_fail(ctx, OVHIP_EINVAL, "ovhip_job_band: row_end must be a CTU-row boundary not above the previous band's", hipSuccess);
    if (bs->n >= MAX_BANDS) return ov_fail(ctx, OVHIP_EUNSUP, "ovhip_job_band: too many bands", hipSuccess);
    const uint32_t stages = bs->stages;
    band_flow_reclaim(j, 0);

    // ---- the band's slices ----
    ovhip_band_counts c1;
    ovhip_rec_counts(rec, &c1);
    if (upto) {
        const uint32_t *u = &upto->n_tb, *m = &c1.n_tb, *lo = &bs->cur.n_tb;
        for (int i = 0; i < 10; ++i) if (u[i] > m[i] || u[i] < lo[i]) return ov_fail(ctx, OVHIP_EINVAL, "ovhip_job_band: counts outside the recorded arrays", hipSuccess);
        c1 = *upto;
    }
    const ovhip_band_counts c0 = bs->cur;
    const int b = bs->n;
    BandRec &B = bs->band[b];
    memset(&B, 0, sizeof(B));
    B.row0 = bs->row_prev; B.row1 = row_end; B.c0 = c0; B.c1 = c1;
    size_t dummy = 0;
    if (ovhip_rec_ciip_units(rec, &dummy) && dummy) return ov_fail(ctx, OVHIP_EUNSUP, "ovhip_job_band: stand-alone CIIP blend units (a second picture with the caller's intra prediction)", hipSuccess);
    const size_t n_tb = c1.n_tb - c0.n_tb, n_coef = c1.n_coef - c0.n_coef, n_mc = c1.n_mc - c0.n_mc, n_mcx = c1.n_mcx - c0.n_mcx,
                 n_aff = c1.n_aff - c0.n_aff, n_side = c1.n_side - c0.n_side, n_reg = c1.n_reg - c0.n_reg, n_it_all = c1.n_itask - c0.n_itask,
                 n_ev = (stages & OVHIP_STAGE_DBF) ? c1.n_edge_v - c0.n_edge_v : 0, n_eh = (stages & OVHIP_STAGE_DBF) ? c1.n_edge_h - c0.n_edge_h : 0;
    const size_t n_it = (stages & OVHIP_STAGE_INTRA) ? n_it_all : 0;
    const ovhip_itask *it_all = ovhip_rec_itasks(rec, &dummy);
    // flow items of the band: counted first (the count does not depend on the order), built after the sort
    size_t n_items = 0; int by_flow = n_it != 0 && !(pr->stages && (stages & OVHIP_STAGE_INTRA_LEVELS));
    for (size_t i = 0; i < n_it && by_flow; ++i) {
        const ovhip_itask &t = it_all[c0.n_itask + i];
        if (t.kind == OVHIP_IT_REGION) { ++n_items; continue; }
        const int npx = 1 << (t.log2_w + t.log2_h), strips = (npx + 255) / 256;          // (FSTRIP = 256, kernels_intra.hip)
        if (strips > 32) by_flow = 0;
        n_items += (size_t)strips * (t.kind == OVHIP_IT_LUMA ? 1 : 2);
    }
    if (!by_flow) n_items = 0;

    // ---- which tails this call runs, and the filter rows they make final ----
    const int t_first = bs->tails, t_end = b + 1;                     // tails [t_first, t_end): this band's (see "bottom row" below)
    if (t_end > t_first && !bs->filters_latched) {
        bs->sao_on = pr->sao && (stages & OVHIP_STAGE_SAO); bs->alf_on = pr->alf_ctus && (stages & OVHIP_STAGE_ALF);
        bs->filters_latched = 1;
    }
    if (bs->filters_latched && (bs->sao_on != (pr->sao && (stages & OVHIP_STAGE_SAO)) || bs->alf_on != (pr->alf_ctus && (stages & OVHIP_STAGE_ALF))))
        return ov_fail(ctx, OVHIP_EINVAL, "ovhip_job_band: SAO / ALF switched on or off inside a picture", hipSuccess);
    const int sao_on = bs->filters_latched && bs->sao_on, alf_on = bs->filters_latched && bs->alf_on;
    if (alf_on && (!pr->alf_luma_coeff || !pr->alf_luma_clip || !pr->alf_chroma_coeff || !pr->alf_chroma_clip || !pr->alf_cc_coeff))
        return ov_fail(ctx, OVHIP_EINVAL, "ovhip_job_band: ALF tables missing", hipSuccess);
    int32_t dbf_new = bs->dbf_rows, sao_new = bs->sao_rows, alf_new = bs->alf_rows;
    if (t_end > t_first) {
        const int32_t E = t_end - 1 == b ? row_end : bs->band[t_end - 1].row1;
        const bool fin = last != 0;
        dbf_new = fin ? j->h : ((stages & OVHIP_STAGE_DBF) ? (E - 8 > dbf_new ? E - 8 : dbf_new) : E);
        // the second stage (SAO, or the copy that stands in for it) reads one row below its window; the third (ALF, or the copy back)
        // three rows below its own -- and the second stage's next window re-reads the row above it, which the third must leave alone
        sao_new = fin ? j->h : (floor8(dbf_new - 1) > sao_new ? floor8(dbf_new - 1) : sao_new);
        alf_new = fin ? j->h : (floor8(sao_new - 3) > alf_new ? floor8(sao_new - 3) : alf_new);
    }
    const int nb_ctu_w = (j->w + (1 << log2_ctu) - 1) >> log2_ctu;
    const int sao_r0 = bs->sao_rows >> log2_ctu, sao_r1 = sao_new > bs->sao_rows ? ((sao_new - 1) >> log2_ctu) + 1 : sao_r0;
    const int alf_r0 = bs->alf_rows >> log2_ctu, alf_r1 = alf_new > bs->alf_rows ? ((alf_new - 1) >> log2_ctu) + 1 : alf_r0;
    const size_t n_sao = sao_on ? (size_t)(sao_r1 - sao_r0) * nb_ctu_w : 0, n_alf = alf_on ? (size_t)(alf_r1 - alf_r0) * nb_ctu_w : 0;

    // ---- one staging block: layout ----
    size_t o = 0;
    auto put = [&o](size_t bytes) { size_t at = o; o += (bytes + 255) & ~(size_t)255; return at; };
    const bool lmcs_now = pr->lmcs && !bs->lmcs_up, alf_now = alf_on && !bs->alf_up;
    const size_t o_fwd = put(lmcs_now ? 2048 : 0), o_bwd = put(lmcs_now ? 2048 : 0);
    const size_t o_lco = put(alf_now ? 24 * OVHIP_ALF_LUMA_SET_SIZE * 2 : 0), o_lcl = put(alf_now ? 24 * OVHIP_ALF_LUMA_SET_SIZE * 2 : 0),
                 o_cco = put(alf_now ? 8 * 7 * 2 : 0), o_ccl = put(alf_now ? 8 * 7 * 2 : 0), o_cc = put(alf_now ? 2 * 4 * 8 * 2 : 0);
    const size_t o_tb = put(n_tb * sizeof(ovhip_tb_cmd)), o_coef = put(n_coef * 2), o_mc = put(n_mc * sizeof(ovhip_mc_unit)),
                 o_mcx = put(n_mcx * sizeof(ovhip_mc_unit)), o_aff = put(n_aff * sizeof(ovhip_aff_unit)), o_side = put(n_side * 4),
                 o_reg = put(n_reg * sizeof(ovhip_lmcs_region)), o_it = put(n_it * sizeof(ovhip_itask)), o_items = put(n_items * 4),
                 o_ev = put(n_ev * sizeof(ovhip_dbf_edge)), o_eh = put(n_eh * sizeof(ovhip_dbf_edge)),
                 o_sao = put(n_sao * sizeof(ovhip_sao_ctu)), o_alf = put(n_alf * sizeof(ovhip_alf_ctu));
    const size_t upload_bytes = o;
    const size_t o_mv = put(n_mcx * 16);                                // device only: the refined vectors k_mcxa leaves (nobody reads them here)
    char *hb = nullptr, *db = nullptr;
    if (o) CHK(arena_take(j, o, &hb, &db));

    // ---- fill it ----
    size_t cls[4] = { 0, 0, 0, 0 }, tiny[4][4] = { { 0 } };
    if (n_tb) ovhip_rec_tb_split_range_(rec, c0.n_tb, n_tb, (ovhip_tb_cmd *)(hb + o_tb), cls, tiny);
    uint32_t n_lv = 0; int have_levels = 0;
    if (n_it) have_levels = ovhip_rec_itasks_sorted_range_(rec, c0.n_itask, n_it, (ovhip_itask *)(hb + o_it), bs->level_start, BAND_LEVELS + 2, &n_lv) == 0;
    if (n_it && !have_levels && !by_flow) return ov_fail(ctx, OVHIP_EUNSUP, "ovhip_job_band: a band with more levels than the table holds and blocks the flow launch cannot take", hipSuccess);
    if (n_items) {
        const size_t k = ovhip_intra_flow_items((const ovhip_itask *)(hb + o_it), n_it, (uint32_t *)(hb + o_items), n_items);
        if (k != n_items) return ov_fail(ctx, OVHIP_EINVAL, "ovhip_job_band: flow item count", hipSuccess);
    }
    {
        size_t n;
        if (n_coef) memcpy(hb + o_coef, ovhip_rec_coefs(rec, &n) + c0.n_coef, n_coef * 2);
        if (n_mc) memcpy(hb + o_mc, ovhip_rec_mc_units(rec, &n) + c0.n_mc, n_mc * sizeof(ovhip_mc_unit));
        if (n_mcx) memcpy(hb + o_mcx, ovhip_rec_mcx_units(rec, &n) + c0.n_mcx, n_mcx * sizeof(ovhip_mc_unit));
        if (n_aff) memcpy(hb + o_aff, ovhip_rec_aff_units(rec, &n) + c0.n_aff, n_aff * sizeof(ovhip_aff_unit));
        if (n_side) memcpy(hb + o_side, ovhip_rec_aff_side(rec, &n) + c0.n_side, n_side * 4);
        if (n_reg) memcpy(hb + o_reg, ovhip_rec_lmcs_regions(rec, &n) + c0.n_reg, n_reg * sizeof(ovhip_lmcs_region));
        ovhip_dbf_offsets offs;
        const ovhip_dbf_edge *ev = ovhip_rec_dbf_edges(rec, 0, &n, &offs), *eh = ovhip_rec_dbf_edges(rec, 1, &n, nullptr);
        bs->offs = offs;
        if (n_ev) memcpy(hb + o_ev, ev + c0.n_edge_v, n_ev * sizeof(ovhip_dbf_edge));
        if (n_eh) memcpy(hb + o_eh, eh + c0.n_edge_h, n_eh * sizeof(ovhip_dbf_edge));
        if (n_sao) memcpy(hb + o_sao, pr->sao + (size_t)sao_r0 * nb_ctu_w, n_sao * sizeof(ovhip_sao_ctu));
        if (n_alf) memcpy(hb + o_alf, pr->alf_ctus + (size_t)alf_r0 * nb_ctu_w, n_alf * sizeof(ovhip_alf_ctu));
        if (lmcs_now) { memcpy(hb + o_fwd, pr->lmcs->fwd_lut, 2048); memcpy(hb + o_bwd, pr->lmcs->bwd_lut, 2048); }
        if (alf_now) {
            memcpy(hb + o_lco, pr->alf_luma_coeff, 24 * OVHIP_ALF_LUMA_SET_SIZE * 2); memcpy(hb + o_lcl, pr->alf_luma_clip, 24 * OVHIP_ALF_LUMA_SET_SIZE * 2);
            memcpy(hb + o_cco, pr->alf_chroma_coeff, 8 * 7 * 2); memcpy(hb + o_ccl, pr->alf_chroma_clip, 8 * 7 * 2);
            memcpy(hb + o_cc, pr->alf_cc_coeff, 2 * 4 * 8 * 2);
        }
    }
    if (upload_bytes) {
        OV_HIP(ctx, hipMemcpyAsync(db, hb, upload_bytes, hipMemcpyHostToDevice, ctx->stream));
        j->st.h2d_bytes += upload_bytes; j->st.n_h2d++;
    }
    OV_HIP(ctx, hipEventRecord(j->ev_h2d, ctx->stream));
    if (lmcs_now) { bs->d_fwd = (const uint16_t *)(db + o_fwd); bs->d_bwd = (const uint16_t *)(db + o_bwd); bs->lmcs_up = 1; bs->luts = *pr->lmcs; bs->have_luts = 1; }
    if (alf_now) {
        bs->d_alf.luma_coeff = (const int16_t *)(db + o_lco); bs->d_alf.luma_clip = (const int16_t *)(db + o_lcl);
        bs->d_alf.chroma_coeff = (const int16_t *)(db + o_cco); bs->d_alf.chroma_clip = (const int16_t *)(db + o_ccl);
        bs->d_alf.cc_coeff = (const int16_t *)(db + o_cc);
        bs->d_alf.class_scratch = (uint8_t *)j->dev[B_CLASS].p; bs->d_alf.log2_ctu_s = log2_ctu;
        bs->alf_up = 1;
    }
    if (pr->lmcs && !bs->lmcs_up) return ov_fail(ctx, OVHIP_EINVAL, "ovhip_job_band: LMCS tables", hipSuccess);
    const ovhip_lmcs_luts *luts = pr->lmcs ? &bs->luts : nullptr;
    const uint16_t *d_fwd = pr->lmcs ? bs->d_fwd : nullptr, *d_bwd = pr->lmcs ? bs->d_bwd : nullptr;
    int16_t *d_scales = (int16_t *)j->dev[B_SCALE].p;
    // the band's slices through pointers moved back by the slice's first index: the commands' own indices stay what the recorder wrote
    const ovhip_lmcs_region *d_reg = (const ovhip_lmcs_region *)(db + o_reg);
    const ovhip_lmcs_region *d_reg_g = d_reg - c0.n_reg;
    const int16_t *d_coef_g = (const int16_t *)(db + o_coef) - c0.n_coef;
    const int32_t *d_side_g = (const int32_t *)(db + o_side) - c0.n_side;
    const ovhip_itask *d_it = (const ovhip_itask *)(db + o_it);
    B.d_it = d_it; B.n_it = (uint32_t)n_it;
    B.d_ev = (const ovhip_dbf_edge *)(db + o_ev); B.n_ev = (uint32_t)n_ev; B.d_eh = (const ovhip_dbf_edge *)(db + o_eh); B.n_eh = (uint32_t)n_eh;
    j->st.n_tb += (uint32_t)n_tb; j->st.n_mc += (uint32_t)n_mc; j->st.n_mcx += (uint32_t)n_mcx; j->st.n_aff += (uint32_t)n_aff;
    j->st.n_edges_v += (uint32_t)n_ev; j->st.n_edges_h += (uint32_t)n_eh; j->st.n_regions += (uint32_t)n_reg; j->st.n_itasks += (uint32_t)n_it; j->st.n_ilevels += n_lv;

    // ---- recon(b) ----
    if (stages & OVHIP_STAGE_MC) {
        if (n_mc) { CHK(ovhip_mc_launch(ctx, dst, refs, n_refs, (const ovhip_mc_unit *)(db + o_mc), (uint32_t)n_mc, d_fwd, nullptr)); j->st.n_launches++; }
        if (n_mcx || n_aff) {
            CHK(ovhip_mcxa_launch(ctx, dst, refs, n_refs, (const ovhip_mc_unit *)(db + o_mcx), (uint32_t)n_mcx, (int32_t *)(db + o_mv),
                                  (const ovhip_aff_unit *)(db + o_aff), (uint32_t)n_aff, d_side_g, d_fwd));
            j->st.n_launches++;
        }
    }
    int flow_workers = 0;
    if (by_flow && n_items) {
        // workers: no more than the band's widest level can use, no more than the device's budget has left (else: one launch per level)
        size_t widest = 0, run = 0;
        const ovhip_itask *hs = (const ovhip_itask *)(hb + o_it); const uint32_t *items = (const uint32_t *)(hb + o_items);
        for (size_t q = 0; q < n_items; ++q) {
            run = (q && hs[items[q] & 0xffffff].level == hs[items[q - 1] & 0xffffff].level) ? run + 1 : 1;
            if (run > widest) widest = run;
        }
        int want = pr->flow_workers ? (int)pr->flow_workers : (int)((2 * widest + 63) & ~(size_t)63);
        const int most = (6 * ctx->num_cus) >> flow_shift_of(ctx->device);
        if (want > most) want = most;
        if (want < 64) want = 64;
        flow_workers = band_flow_take(ctx, (want + 63) & ~63);
        if (!flow_workers) { by_flow = 0; if (!have_levels) return ov_fail(ctx, OVHIP_EUNSUP, "ovhip_job_band: flow budget exhausted and no level table", hipSuccess); }
        B.flow_charge = flow_workers;
    }
    // the row above the band as the band's reconstruction must see it (k_band_row): unfiltered, mapped -- and without the hand-over bit
    // when the readers are the per-level kernels, which take samples as they are
    const bool swap_row = b > 0 && bs->keep_valid && (n_it || n_reg) && B.row0 > 0;
    if (swap_row) CHK(band_row(j, dst, B.row0 - 1, 1, (by_flow && n_items) ? 0xffffu : 0x03ffu));
    int flow_prepared = 0;
    if (stages & OVHIP_STAGE_ITX) {
        const ovhip_tb_cmd *d_tb = (const ovhip_tb_cmd *)(db + o_tb);
        const uint32_t t_luma[4] = { (uint32_t)tiny[1][0], (uint32_t)tiny[1][1], (uint32_t)tiny[1][2], (uint32_t)tiny[1][3] };
        const uint32_t t_chroma[4] = { (uint32_t)tiny[3][0], (uint32_t)tiny[3][1], (uint32_t)tiny[3][2], (uint32_t)tiny[3][3] };
        if (cls[0] + cls[1]) { CHK(ovhip_itx_launch_ex_(ctx, dst, &j->res, d_tb, (uint32_t)cls[0], (uint32_t)cls[1], t_luma, d_coef_g, nullptr, nullptr)); j->st.n_launches++; }
        if (n_reg) {
            if (!luts) return ov_fail(ctx, OVHIP_EINVAL, "ovhip_job_band: chroma-scale regions recorded without LMCS tables", hipSuccess);
            if (by_flow && n_items) {
                CHK(ovhip_lmcs_scale_prepare_launch(ctx, dst, d_reg, (uint32_t)n_reg, luts, d_scales + c0.n_reg, d_it, (uint32_t)n_it, j->d_flow, j->epoch));
                flow_prepared = 1;
            } else CHK(ovhip_lmcs_scale_launch(ctx, dst, d_reg, (uint32_t)n_reg, luts, d_scales + c0.n_reg));
            j->st.n_launches++;
        }
        if (cls[2] + cls[3]) {
            CHK(ovhip_itx_launch_ex_(ctx, dst, &j->res, d_tb + cls[0] + cls[1], (uint32_t)cls[2], (uint32_t)cls[3], t_chroma, d_coef_g, d_scales, nullptr));
            j->st.n_launches++;
        }
    }
    if (n_it) {
        if (by_flow && n_items) {
            CHK(ovhip_intra_flow_launch(ctx, dst, &j->res, d_it, (uint32_t)n_it, (const uint32_t *)(db + o_items), (uint32_t)n_items, d_reg_g, luts, d_scales,
                                        log2_ctu, j->d_flow, j->epoch, j->abort_host, !flow_prepared, flow_workers));
            j->st.n_launches += 1 + !flow_prepared;
            B.tagged = 1;
        } else {
            // one launch per level: its kernels read plain samples -- the band above must not carry the flow launches' hand-over bit any more
            if (b > 0 && bs->band[b - 1].tagged && bs->tails < b) {
                CHK(ovhip_intra_flow_untag_launch(ctx, dst, bs->band[b - 1].d_it, bs->band[b - 1].n_it, 1));
                bs->band[b - 1].tagged = 2;                          // (un-tagged early; the tail's own un-tag is then idempotent)
                j->st.n_launches++;
            }
            const ovhip_itask *hs = (const ovhip_itask *)(hb + o_it);
            for (uint32_t l = 0; l < n_lv; ++l) {
                const uint32_t a = bs->level_start[l], e = bs->level_start[l + 1];
                if (e > a) {
                    CHK(ovhip_intra_level_launch(ctx, dst, &j->res, d_it + a, e - a, d_reg_g, luts, d_scales, log2_ctu, ovhip_intra_level_geom(hs + a, e - a)));
                    j->st.n_launches++;
                }
            }
        }
    }
    if (swap_row) CHK(band_row(j, dst, B.row0 - 1, 2, 0));
    if (!last && B.row1 > B.row0) { CHK(band_row(j, dst, B.row1 - 1, 0, 0)); bs->keep_valid = 1; }
    if (!bs->ev_recon[b]) OV_HIP(ctx, hipEventCreateWithFlags(&bs->ev_recon[b], hipEventDisableTiming));
    OV_HIP(ctx, hipEventRecord(bs->ev_recon[b], ctx->stream));
    bs->n = b + 1; bs->cur = c1; bs->row_prev = row_end;

    // ---- tails: inverse luma mapping + un-tag, deblocking, per band; then the SAO / ALF rows they made final, once ----
    for (int t = t_first; t < t_end; ++t) {
        BandRec &T = bs->band[t];
        if (T.row1 > T.row0) {
            const bool inverse = pr->lmcs && (stages & OVHIP_STAGE_ITX);
            if (inverse) {
                ovhip_pic view = *dst;                                // the band's luma rows; the chroma un-tag takes picture coordinates
                view.y = dst->y + (size_t)T.row0 * dst->stride_y; view.h = T.row1 - T.row0;
                CHK(ovhip_lmcs_inverse_untag_launch(ctx, &view, d_bwd, T.d_it, T.tagged ? T.n_it : 0u));
                j->st.n_launches++;
            } else if (T.tagged == 1 && T.n_it) { CHK(ovhip_intra_flow_untag_launch(ctx, dst, T.d_it, T.n_it, 1)); j->st.n_launches++; }
        }
        if ((stages & OVHIP_STAGE_DBF) && (T.n_ev || T.n_eh)) {
            CHK(ovhip_dbf_launch_edges_ex(ctx, dst, T.d_ev, T.n_ev, T.d_eh, T.n_eh, &bs->offs));
            j->st.n_launches += (T.n_ev != 0) + (T.n_eh != 0);
        }
    }
    if (t_end > t_first) {
        const ovhip_sao_ctu *d_sao = sao_on ? (const ovhip_sao_ctu *)(db + o_sao) - (size_t)sao_r0 * nb_ctu_w : nullptr;
        ovhip_alf_pic ap = bs->d_alf;
        ap.ctus = alf_on ? (const ovhip_alf_ctu *)(db + o_alf) - (size_t)alf_r0 * nb_ctu_w : nullptr;
        auto copy_rows = [&](const ovhip_pic *d, const ovhip_pic *s_, int32_t r0, int32_t r1) -> int {
            if (r1 <= r0) return OVHIP_OK;
            OV_HIP(ctx, hipMemcpyAsync(d->y + (size_t)r0 * d->stride_y, s_->y + (size_t)r0 * s_->stride_y, (size_t)(r1 - r0) * d->stride_y * 2, hipMemcpyDeviceToDevice, ctx->stream));
            const int32_t c0_ = r0 / 2, c1_ = r1 == j->h ? j->h / 2 : r1 / 2;
            OV_HIP(ctx, hipMemcpyAsync(d->cb + (size_t)c0_ * d->stride_c, s_->cb + (size_t)c0_ * s_->stride_c, (size_t)(c1_ - c0_) * d->stride_c * 2, hipMemcpyDeviceToDevice, ctx->stream));
            OV_HIP(ctx, hipMemcpyAsync(d->cr + (size_t)c0_ * d->stride_c, s_->cr + (size_t)c0_ * s_->stride_c, (size_t)(c1_ - c0_) * d->stride_c * 2, hipMemcpyDeviceToDevice, ctx->stream));
            return OVHIP_OK;
        };
        if (sao_new > bs->sao_rows) {
            if (sao_on) { CHK(ovhip_sao_launch_rows(ctx, &j->tmp, dst, d_sao, log2_ctu, bs->sao_rows, sao_new)); j->st.n_launches++; }
            else if (alf_on) CHK(copy_rows(&j->tmp, dst, bs->sao_rows, sao_new));
        }
        if (alf_new > bs->alf_rows) {
            if (alf_on) { CHK(ovhip_alf_launch_rows(ctx, dst, &j->tmp, &ap, bs->alf_rows, alf_new)); j->st.n_launches++; }
            else if (sao_on) CHK(copy_rows(dst, &j->tmp, bs->alf_rows, alf_new));
        }
        bs->dbf_rows = dbf_new; bs->sao_rows = sao_new; bs->alf_rows = alf_new;
        bs->rows_final = (sao_on || alf_on) ? alf_new : dbf_new;
        bs->tails = t_end;
        const int te = t_end - 1;
        bs->band[te].rows_final = bs->rows_final;
        if (!bs->ev_tail[te]) OV_HIP(ctx, hipEventCreateWithFlags(&bs->ev_tail[te], hipEventDisableTiming));
        OV_HIP(ctx, hipEventRecord(bs->ev_tail[te], ctx->stream));
        bs->last_event = (void *)bs->ev_tail[te]; bs->last_rows = bs->rows_final;
    }
    // (behind every band: ovhip_job_wait / _begin / _destroy wait for what has been enqueued, whether or not the picture was completed)
    OV_HIP(ctx, hipEventRecord(j->ev_done, ctx->stream));
    j->flushed = 1; j->flow_launched = 0;
    if (last) bs->closed = 1;
    return OVHIP_OK;
}

