/* ovvc_frame.c -- one decoder frame thread on the device path (include/ovvc_hip.h, "Frame thread").
 *
 * What shim/rcn_hip.c keeps per OVCTUDec, moved into the library so that it can be executed and tested without the reference's
 * headers: context + job + the picture's place in the device DPB + its reference table, and the ORDER of a picture's end:
 *
 *     uploads enqueued -> wait for the reference pictures (host) -> launches -> ovhip_job_wait (incl. its second pass)
 *     -> publish + unpin the references -> output (optional; after the wait, never before: r2 downloaded first)
 *
 * The reference's order for the same events: slicedec.c:934-956 (filters of the last rows, then ovdpb_report_decoded_ctu_line),
 * rcn_inter.c:131-146 (a reader waits for its reference's rows), dectest.c:372-409 (the application reads the frame after the
 * picture was output).  Plain C over the C ABI; no HIP calls of its own.
 */
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>
#include "ovvc_hip.h"
#include "ovvc_dpb_priv.h"

#define MAX_REFS 16

static double mono_s(void) { struct timespec t; clock_gettime(CLOCK_MONOTONIC, &t); return (double)t.tv_sec + 1e-9 * (double)t.tv_nsec; }

struct ovhip_frame {
    ovhip_dpb *dpb;
    int dev;
    int32_t w, h;
    int id;                               /* creation order: the frame's name in the event trace */
    ovhip_ctx *ctx;
    ovhip_job *job;                       /* own job, created on first use */
    /* dry frame (a DPB on a test back-end, no device): the same state machine and the same DPB calls, nothing launched.  Its own
     * plain recorder takes the picture's commands; the eager-DMVR counters move as the device's would. */
    double done_at, published_at;         /* CLOCK_MONOTONIC seconds: ovhip_job_wait returned / the picture was published (stall analysis) */
    int dry;
    ovhip_recorder *dry_rec;
    int64_t dry_pending, dry_done;
    const void *key; ovhip_pic dst; int live;
    const void *ref_key[MAX_REFS]; uint64_t ref_tag[MAX_REFS]; ovhip_pic ref_pic[MAX_REFS]; void *ref_ev[MAX_REFS]; unsigned char ref_pinned[MAX_REFS];
    int n_refs;
    int status;
    char err[192];
};

/* ---- event trace (include/ovvc_hip.h, ovhip_frame_set_trace): WHEN a caller (shim/rcn_hip.c) makes its frame-level calls ---- */
static void (*g_trace)(void *user, const ovhip_frame_event *ev);
static void *g_trace_user;
static int g_next_id;

void
ovhip_frame_set_trace(void (*sink)(void *user, const ovhip_frame_event *ev), void *user)
{
    g_trace_user = user;
    __atomic_store_n(&g_trace, sink, __ATOMIC_RELEASE);
}

static void
trace(const ovhip_frame *f, uint32_t op, const void *key, uint64_t tag, int64_t a, int64_t b, int64_t result)
{
    void (*sink)(void *, const ovhip_frame_event *) = __atomic_load_n(&g_trace, __ATOMIC_ACQUIRE);
    if (!sink) return;
    ovhip_frame_event ev;
    memset(&ev, 0, sizeof(ev));
    ev.op = op; ev.frame = f->id; ev.key = (uint64_t)(uintptr_t)key; ev.tag = tag; ev.a = a; ev.b = b; ev.result = result;
    sink(g_trace_user, &ev);
}

static size_t
n_refined_units(ovhip_frame *f)
{
    size_t n = 0;
    ovhip_recorder *r = ovhip_frame_recorder(f);
    if (r) (void)ovhip_rec_mcx_units(r, &n);
    return n;
}

static int
fail(ovhip_frame *f, int code, const char *what)
{
    if (code < 0 && !f->status) {
        f->status = code;
        snprintf(f->err, sizeof(f->err), "%s (%d)%s%s", what, code, f->ctx ? ": " : "", f->ctx ? ovhip_last_error(f->ctx) : "");
    }
    return code;
}

int
ovhip_frame_create(ovhip_dpb *dpb, int dev, int32_t w, int32_t h, ovhip_frame **out)
{
    if (!dpb || !out || dev < 0 || dev >= ovhip_dpb_n_devices(dpb) || w <= 0 || h <= 0) return OVHIP_EINVAL;
    *out = NULL;
    const int hipdev = ovhip_dpb_device(dpb, dev);
    ovhip_frame *f = (ovhip_frame *)calloc(1, sizeof(*f));
    if (!f) return OVHIP_ENOMEM;
    f->dpb = dpb; f->dev = dev; f->w = w; f->h = h;
    f->id = __atomic_fetch_add(&g_next_id, 1, __ATOMIC_RELAXED);
    if (hipdev < 0) {
        /* a DPB on a test back-end has no device to decode on: a dry frame (the caller's sequence of frame-level calls is the
         * subject, tests/test_shim_device_cpu.py); pictures "decode" to whatever the back-end's pic_alloc handed out */
        f->dry = 1;
        f->dry_rec = ovhip_rec_create(w, h);
        if (!f->dry_rec) { free(f); return OVHIP_ENOMEM; }
        *out = f;
        return OVHIP_OK;
    }
    int r = ovhip_ctx_create(&f->ctx, hipdev, NULL);
    if (r != OVHIP_OK) { free(f); return r; }
    *out = f;
    return OVHIP_OK;
}

static void
unpin_refs(ovhip_frame *f)
{
    for (int i = 0; i < f->n_refs; ++i)
        if (f->ref_pinned[i]) { (void)ovhip_dpb_unpin(f->dpb, f->ref_key[i]); f->ref_pinned[i] = 0; }
}

void
ovhip_frame_destroy(ovhip_frame *f)
{
    if (!f) return;
    if (f->live) (void)ovhip_frame_fail(f, OVHIP_EINVAL);
    if (f->job) ovhip_job_destroy(f->job);
    if (f->dry_rec) ovhip_rec_destroy(f->dry_rec);
    if (f->ctx) ovhip_ctx_destroy(f->ctx);
    free(f);
}

ovhip_ctx *ovhip_frame_ctx(ovhip_frame *f) { return f ? f->ctx : NULL; }
double ovhip_frame_published_at(const ovhip_frame *f) { return f ? f->published_at : 0.0; }
double ovhip_frame_done_at(const ovhip_frame *f) { return f ? f->done_at : 0.0; }

ovhip_job *
ovhip_frame_job(ovhip_frame *f)
{
    if (!f || f->dry) return NULL;
    if (!f->job && fail(f, ovhip_job_create(f->ctx, f->w, f->h, &f->job), "ovhip_job_create") != OVHIP_OK) f->job = NULL;
    return f->job;
}

ovhip_recorder *
ovhip_frame_recorder(ovhip_frame *f)
{
    if (f && f->dry) return f->dry_rec;
    ovhip_job *j = ovhip_frame_job(f);
    return j ? ovhip_job_recorder(j) : NULL;
}
const char *ovhip_frame_last_error(const ovhip_frame *f) { return f ? f->err : "no frame"; }

int ovhip_frame_begin(ovhip_frame *f, const void *key) { return ovhip_frame_begin_tag(f, key, 0); }

int
ovhip_frame_begin_tag(ovhip_frame *f, const void *key, uint64_t tag)
{
    if (!f || !key) return OVHIP_EINVAL;
    /* a picture that was begun and never submitted must not leave its readers waiting */
    if (f->live) (void)ovhip_frame_fail(f, OVHIP_EINVAL);
    f->status = 0; f->err[0] = 0; f->n_refs = 0;
    int r = ovhip_dpb_begin_tag(f->dpb, key, tag, f->dev, f->w, f->h, &f->dst);
    trace(f, OVHIP_FE_BEGIN, key, tag, f->dev, 0, r);
    if (r != OVHIP_OK) return fail(f, r, "ovhip_dpb_begin");
    f->key = key; f->live = 1;
    if (f->dry) { ovhip_rec_reset(f->dry_rec); f->dry_pending = f->dry_done = 0; }
    if (f->job) {
        r = ovhip_job_begin(f->job);
        if (r != OVHIP_OK) { fail(f, r, "ovhip_job_begin"); (void)ovhip_frame_fail(f, r); return r; }
    }
    return OVHIP_OK;
}

int ovhip_frame_ref(ovhip_frame *f, const void *ref_key) { return ovhip_frame_ref_tag(f, ref_key, 0); }

int
ovhip_frame_ref_tag(ovhip_frame *f, const void *ref_key, uint64_t tag)
{
    if (!f || !ref_key || !f->live) return OVHIP_EINVAL;
    for (int i = 0; i < f->n_refs; ++i) if (f->ref_key[i] == ref_key) return i;
    if (f->n_refs >= MAX_REFS) return fail(f, OVHIP_EUNSUP, "more than 16 distinct reference pictures");
    const int i = f->n_refs++;
    f->ref_key[i] = ref_key; f->ref_tag[i] = tag; f->ref_pinned[i] = 0; f->ref_ev[i] = NULL;
    memset(&f->ref_pic[i], 0, sizeof(f->ref_pic[i]));
    /* as soon as the reference lists are known: a picture decoded on another device is pushed here when it is done */
    (void)ovhip_dpb_want_tag(f->dpb, ref_key, tag, f->dev);
    trace(f, OVHIP_FE_REF, ref_key, tag, i, 0, i);
    return i;
}

int
ovhip_frame_ref_at(ovhip_frame *f, int slot, const void *ref_key)
{
    if (!f || !ref_key || !f->live || slot != f->n_refs) return OVHIP_EINVAL;
    if (f->n_refs >= MAX_REFS) return fail(f, OVHIP_EUNSUP, "more than 16 reference table entries");
    f->n_refs++;
    f->ref_key[slot] = ref_key; f->ref_tag[slot] = 0; f->ref_pinned[slot] = 0; f->ref_ev[slot] = NULL;
    memset(&f->ref_pic[slot], 0, sizeof(f->ref_pic[slot]));
    (void)ovhip_dpb_want(f->dpb, ref_key, f->dev);
    return slot;
}

/* the device analogue of ovdpb_frame_synchro: every reference complete and, if it was decoded elsewhere, here */
static int
acquire_refs(ovhip_frame *f)
{
    for (int i = 0; i < f->n_refs; ++i) {
        if (f->ref_pinned[i]) continue;
        int r = ovhip_dpb_acquire_tag(f->dpb, f->ref_key[i], f->ref_tag[i], f->dev, &f->ref_pic[i], &f->ref_ev[i]);
        if (r != OVHIP_OK) return fail(f, r, r == OVHIP_EREF ? "a reference picture failed to decode" : "reference picture unknown to the device DPB");
        f->ref_pinned[i] = 1;
    }
    for (int i = 0; i < f->n_refs; ++i) {
        if (!f->ref_ev[i]) continue;
        int r = ovhip_dpb_wait_copy(f->dpb, f->dev, f->ref_ev[i]);
        if (r != OVHIP_OK) return fail(f, r, "transfer of a reference picture");
        f->ref_ev[i] = NULL;
    }
    return OVHIP_OK;
}

/* 1: every reference picture named so far is complete (acquired and pinned: ovhip_frame_dmvr_rows_begin / ovhip_frame_submit will not
 * wait for a decode), 0: not yet, < 0: one of them failed.  Never waits for a decode -- what a caller that may go on parsing asks before
 * it starts the eager DMVR rows (shim/rcn_hip.c with the caller patch: the CTU rows are then reported later instead of the parse
 * stopping here). */
int
ovhip_frame_refs_ready(ovhip_frame *f)
{
    if (!f || !f->live) return OVHIP_EINVAL;
    for (int i = 0; i < f->n_refs; ++i) {
        if (f->ref_pinned[i]) continue;
        const int r = ovhip_dpb_poll_tag(f->dpb, f->ref_key[i], f->ref_tag[i]);
        if (r <= 0) return r < 0 ? fail(f, r, "a reference picture failed to decode") : 0;
    }
    const int r = acquire_refs(f);
    return r == OVHIP_OK ? 1 : r;
}

static int
before_launch_cb(void *user)
{
    ovhip_frame *f = (ovhip_frame *)user;
    return acquire_refs(f) != OVHIP_OK;
}

int64_t
ovhip_frame_dmvr_rows(ovhip_frame *f)
{
    if (!f || !f->live || (!f->job && !f->dry)) return OVHIP_EINVAL;
    int r = acquire_refs(f);
    if (r != OVHIP_OK) return r;
    int64_t n;
    if (f->dry) n = f->dry_done = f->dry_pending = (int64_t)n_refined_units(f);
    else n = ovhip_job_dmvr_rows(f->job, f->ref_pic, (uint32_t)f->n_refs);
    if (n < 0) fail(f, (int)n, "ovhip_job_dmvr_rows");
    trace(f, OVHIP_FE_DMVR_ROWS, f->key, 0, (int64_t)n_refined_units(f), 0, n);
    return n;
}

int64_t
ovhip_frame_dmvr_rows_begin(ovhip_frame *f, int32_t log2_ctu_s)
{
    if (!f || !f->live || (!f->job && !f->dry)) return OVHIP_EINVAL;
    /* the references are needed (and waited for) only if a unit the pass would cover is a DMVR unit: rows of BDOF-only units do not
     * stop the parse */
    int64_t c = f->dry ? (f->dry_done = f->dry_pending) : ovhip_job_dmvr_rows_collect(f->job);
    if (c < 0) { fail(f, (int)c, "ovhip_job_dmvr_rows_collect"); return c; }
    size_t nu = 0;
    const ovhip_mc_unit *u = ovhip_rec_mcx_units(ovhip_frame_recorder(f), &nu);
    int any = 0;
    for (size_t i = (size_t)c; i < nu && !any; ++i) any = (u[i].flags & OVHIP_MC_DMVR) != 0;
    if (any) {
        int r = acquire_refs(f);
        if (r != OVHIP_OK) return r;
    }
    int64_t n;
    if (f->dry) n = f->dry_pending = (int64_t)nu;
    else n = ovhip_job_dmvr_rows_begin(f->job, f->ref_pic, (uint32_t)f->n_refs, log2_ctu_s);
    if (n < 0) fail(f, (int)n, "ovhip_job_dmvr_rows_begin");
    trace(f, OVHIP_FE_DMVR_BEGIN, f->key, 0, (int64_t)nu, any, n);
    return n;
}

int64_t
ovhip_frame_dmvr_rows_collect(ovhip_frame *f)
{
    if (!f || !f->live || (!f->job && !f->dry)) return OVHIP_EINVAL;
    int64_t n = f->dry ? (f->dry_done = f->dry_pending) : ovhip_job_dmvr_rows_collect(f->job);
    if (n < 0) fail(f, (int)n, "ovhip_job_dmvr_rows_collect");
    trace(f, OVHIP_FE_DMVR_COLLECT, f->key, 0, (int64_t)n_refined_units(f), 0, n);
    return n;
}

static int
publish(ovhip_frame *f, int status)
{
    if (!f->live) return OVHIP_EINVAL;
    f->live = 0;
    int r = ovhip_dpb_publish(f->dpb, f->key, status);
    f->published_at = mono_s();
    unpin_refs(f);
    return r;
}

int
ovhip_frame_fail(ovhip_frame *f, int status)
{
    if (!f) return OVHIP_EINVAL;
    if (!f->status) f->status = status ? status : OVHIP_EINVAL;
    trace(f, OVHIP_FE_FAIL, f->key, 0, f->status, 0, 0);
    return publish(f, f->status);
}

int
ovhip_frame_submit(ovhip_frame *f, ovhip_job *job, const ovhip_pic *intra, const ovhip_job_params *params, ovhip_frame_output *out)
{
    if (!f || !params || !f->live) return OVHIP_EINVAL;
    int r = f->status;                    /* a latched recorder error: the picture is published as failed, never launched */
    ovhip_job *j = job ? job : f->job;
    if (f->dry) {
        /* dry: the same waits and the same publication, no launches, no output */
        if (r == OVHIP_OK) r = acquire_refs(f);
        trace(f, OVHIP_FE_SUBMIT, f->key, 0, (int64_t)n_refined_units(f), f->n_refs, r);
        (void)publish(f, r);
        return r;
    }
    trace(f, OVHIP_FE_SUBMIT, f->key, 0, (int64_t)n_refined_units(f), f->n_refs, r);
    if (r == OVHIP_OK && !j) r = fail(f, OVHIP_EINVAL, "ovhip_frame_submit: nothing was recorded");
    if (r == OVHIP_OK && job) r = fail(f, ovhip_job_bind(job, f->ctx), "ovhip_job_bind");
    if (r == OVHIP_OK) {
        ovhip_job_params pr = *params;
        /* the references are waited for on the host, by this thread, after the uploads have been enqueued (no barrier in the
         * stream: a blocked stream blocks the hardware queue it shares) */
        pr.before_launch = before_launch_cb; pr.before_launch_user = f;
        r = ovhip_job_flush(j, &f->dst, f->ref_pic, (uint32_t)f->n_refs, intra, &pr);
        if (r != OVHIP_OK) fail(f, r, "ovhip_job_flush");
        /* ONLY the wait marks the picture complete: it may run the ordered pass a second time */
        int q = ovhip_job_wait(j);
        f->done_at = mono_s();
        if (q != OVHIP_OK && r == OVHIP_OK) r = fail(f, q, "ovhip_job_wait");
    }
    /* a borrowed job goes back to its own context: this frame (and its context) may be destroyed before the job */
    if (job) { int q = ovhip_job_bind(job, NULL); if (q != OVHIP_OK && r == OVHIP_OK) r = fail(f, q, "ovhip_job_bind(home)"); }
    if (f->status) r = f->status;         /* the first error, e.g. a failed reference rather than "callback failed" */
    /* the picture is complete: its readers go on while this thread copies it out (the output only reads; whoever keeps the key
     * alive -- the decoder's DPB, the stream driver's hold -- does so until this call returns) */
    (void)publish(f, r);
    if (r == OVHIP_OK && out && out->mode != OVHIP_OUT_NONE) {
        switch (out->mode) {
        case OVHIP_OUT_DIGEST: r = ovhip_pic_digest(f->ctx, &f->dst, &out->window, out->digest); break;
        case OVHIP_OUT_PLANES: r = ovhip_pic_download(f->ctx, &f->dst, out->y, out->cb, out->cr, out->stride_y, out->stride_c); break;
        case OVHIP_OUT_PACKED: r = ovhip_pic_output(f->ctx, &f->dst, &out->window, out->packed); break;
        default: r = OVHIP_EINVAL;
        }
        if (r != OVHIP_OK) fail(f, r, "picture output");
    }
    return r;
}
