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
    /* band-wise submission (ovhip_frame_band): what has been handed to the job so far, how many rows were posted to the DPB */
    int band_mode;
    ovhip_band_counts band_prev;
    int32_t band_row_prev, band_rows_posted, dry_row_prev2;
    int n_bands, n_deferred;
    /* OVVC_HIP_FRAME_PROF=1: where the frame-level calls spend their wall time, printed when the frame is destroyed (seconds) */
    double pt_collect, pt_rows_begin, pt_band_refs, pt_band_job, pt_final_refs, pt_job_wait, pt_output, pt_submit; int pn_pics;
    double pt_fl_prepare, pt_fl_upload, pt_fl_refs, pt_fl_launch;      /* ovhip_job_flush by phase (ovhip_job_stats.host_us_*) */
};
static int g_frame_prof = -1;
#define PT_ON() (__atomic_load_n(&g_frame_prof, __ATOMIC_RELAXED) > 0)
#define PT0() const double pt0_ = PT_ON() ? mono_s() : 0.0
#define PT(acc) do { if (PT_ON()) (acc) += mono_s() - pt0_; } while (0)

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
    if (__atomic_load_n(&g_frame_prof, __ATOMIC_RELAXED) < 0) __atomic_store_n(&g_frame_prof, getenv("OVVC_HIP_FRAME_PROF") != NULL, __ATOMIC_RELAXED);
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
    if (PT_ON() && f->pn_pics)
        fprintf(stderr, "frame %d: %d pictures, ms per picture: dmvr collect %.3f, dmvr begin %.3f, band refs check %.3f, band enqueue %.3f, last band: refs wait %.3f, job wait %.3f, output %.3f; whole submit %.3f (flush: prepare %.3f, uploads %.3f, reference wait %.3f, launches %.3f)\n",
                f->id, f->pn_pics, 1e3 * f->pt_collect / f->pn_pics, 1e3 * f->pt_rows_begin / f->pn_pics, 1e3 * f->pt_band_refs / f->pn_pics, 1e3 * f->pt_band_job / f->pn_pics,
                1e3 * f->pt_final_refs / f->pn_pics, 1e3 * f->pt_job_wait / f->pn_pics, 1e3 * f->pt_output / f->pn_pics, 1e3 * f->pt_submit / f->pn_pics,
                1e3 * f->pt_fl_prepare / f->pn_pics, 1e3 * f->pt_fl_upload / f->pn_pics, 1e3 * f->pt_fl_refs / f->pn_pics, 1e3 * f->pt_fl_launch / f->pn_pics);
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
    if (!f->job) {
        if (fail(f, ovhip_job_create(f->ctx, f->w, f->h, &f->job), "ovhip_job_create") != OVHIP_OK) f->job = NULL;
        /* a frame thread's job: sized for its pictures now, not grown picture by picture (OVVC_HIP_NO_RESERVE=1: the A / B).  The
         * reservation is an optimisation: when it cannot be had (page-locked memory is short) the buffers grow with the pictures as they
         * did before, and a picture that then does not fit fails with the allocation's own error */
        else if (!getenv("OVVC_HIP_NO_RESERVE")) (void)ovhip_job_reserve_for_picture(f->job);
    }
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
    memset(&f->band_prev, 0, sizeof(f->band_prev));
    f->band_row_prev = f->band_rows_posted = f->dry_row_prev2 = 0; f->n_bands = f->n_deferred = 0;
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
static int refs_rows_ready(ovhip_frame *f, const int32_t *need, int block);
static void units_need_rows(ovhip_frame *f, const ovhip_band_counts *c0, const ovhip_band_counts *c1, int only_dmvr, int32_t *need);

int
ovhip_frame_refs_ready(ovhip_frame *f)
{
    if (!f || !f->live) return OVHIP_EINVAL;
    if (f->band_mode) {
        /* row-granular: the rows the DMVR units recorded since the last eager pass reach in each reference picture (rcn_inter.c:131-146
         * waits for exactly those, per block) */
        ovhip_band_counts c0, c1;
        int32_t need[MAX_REFS];
        memset(&c0, 0, sizeof(c0));
        ovhip_rec_counts(ovhip_frame_recorder(f), &c1);
        const int64_t done = f->dry ? f->dry_done : (f->job ? ovhip_job_dmvr_rows_collect(f->job) : 0);
        if (done < 0) return fail(f, (int)done, "ovhip_job_dmvr_rows_collect");
        c0.n_mcx = (uint32_t)done; c0.n_mc = c1.n_mc; c0.n_aff = c1.n_aff;
        units_need_rows(f, &c0, &c1, 1, need);
        return refs_rows_ready(f, need, 0);
    }
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

int64_t ovhip_frame_dmvr_rows_begin(ovhip_frame *f, int32_t log2_ctu_s) { return ovhip_frame_dmvr_rows_begin_upto(f, log2_ctu_s, (size_t)-1); }

int64_t
ovhip_frame_dmvr_rows_begin_upto(ovhip_frame *f, int32_t log2_ctu_s, size_t upto)
{
    if (!f || !f->live || (!f->job && !f->dry)) return OVHIP_EINVAL;
    /* the references are needed (and waited for) only if a unit the pass would cover is a DMVR unit: rows of BDOF-only units do not
     * stop the parse */
    PT0();
    int64_t c = f->dry ? (f->dry_done = f->dry_pending) : ovhip_job_dmvr_rows_collect(f->job);
    if (c < 0) { fail(f, (int)c, "ovhip_job_dmvr_rows_collect"); return c; }
    size_t nu = 0;
    const ovhip_mc_unit *u = ovhip_rec_mcx_units(ovhip_frame_recorder(f), &nu);
    if (nu > upto) nu = upto;
    int any = 0;
    for (size_t i = (size_t)c; i < nu && !any; ++i) any = (u[i].flags & OVHIP_MC_DMVR) != 0;
    if (any && f->band_mode) {
        ovhip_band_counts c0, c1;
        int32_t need[MAX_REFS];
        memset(&c0, 0, sizeof(c0));
        ovhip_rec_counts(ovhip_frame_recorder(f), &c1);
        c0.n_mcx = (uint32_t)c; c0.n_mc = c1.n_mc; c0.n_aff = c1.n_aff; c1.n_mcx = (uint32_t)nu;
        units_need_rows(f, &c0, &c1, 1, need);
        int r = refs_rows_ready(f, need, 1);
        if (r < 0) return r;
    } else if (any) {
        int r = acquire_refs(f);
        if (r != OVHIP_OK) return r;
    }
    int64_t n;
    if (f->dry) n = f->dry_pending = (int64_t)nu > f->dry_pending ? (int64_t)nu : f->dry_pending;
    else n = ovhip_job_dmvr_rows_begin_upto(f->job, f->ref_pic, (uint32_t)f->n_refs, log2_ctu_s, upto);
    PT(f->pt_rows_begin);
    if (n < 0) fail(f, (int)n, "ovhip_job_dmvr_rows_begin");
    trace(f, OVHIP_FE_DMVR_BEGIN, f->key, 0, (int64_t)nu, any, n);
    return n;
}

int64_t
ovhip_frame_dmvr_rows_collect(ovhip_frame *f)
{
    if (!f || !f->live || (!f->job && !f->dry)) return OVHIP_EINVAL;
    PT0();
    int64_t n = f->dry ? (f->dry_done = f->dry_pending) : ovhip_job_dmvr_rows_collect(f->job);
    PT(f->pt_collect);
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
    PT0();
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
        if (PT_ON()) {
            ovhip_job_stats st;
            if (ovhip_job_last_stats(j, &st) == OVHIP_OK) { f->pt_fl_prepare += 1e-6 * st.host_us_prepare; f->pt_fl_upload += 1e-6 * st.host_us_upload; f->pt_fl_refs += 1e-6 * st.host_us_wait; f->pt_fl_launch += 1e-6 * st.host_us_launch; }
        }
        /* ONLY the wait marks the picture complete: it may run the ordered pass a second time */
        const double tw0 = PT_ON() ? mono_s() : 0.0;
        int q = ovhip_job_wait(j);
        if (PT_ON()) f->pt_job_wait += mono_s() - tw0;
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
        const double to0 = PT_ON() ? mono_s() : 0.0;
        switch (out->mode) {
        case OVHIP_OUT_DIGEST: r = ovhip_pic_digest(f->ctx, &f->dst, &out->window, out->digest); break;
        case OVHIP_OUT_PLANES: r = ovhip_pic_download(f->ctx, &f->dst, out->y, out->cb, out->cr, out->stride_y, out->stride_c); break;
        case OVHIP_OUT_PACKED: r = ovhip_pic_output(f->ctx, &f->dst, &out->window, out->packed); break;
        default: r = OVHIP_EINVAL;
        }
        if (PT_ON()) f->pt_output += mono_s() - to0;
        if (r != OVHIP_OK) fail(f, r, "picture output");
    }
    PT(f->pt_submit); f->pn_pics += !f->band_mode;
    return r;
}

/* ------------------------------------------------------------------------------------------------ band-wise submission
 * ovhip_job_band under the device DPB's row progress: a band goes to the device when the rows its units read are final in every
 * reference picture (the producer's posted records, seen complete on the HOST: no barrier enters the stream); a band that is not
 * there yet is simply left to the next call, which takes everything recorded since the last band that went -- the parse never waits,
 * except in the picture's last call.  Mirrors rcn_inter_synchronization (rcn_inter.c:131-146) + ovdpb_report_decoded_ctu_line
 * (dpb.c:1309-1323) at band granularity. */
int
ovhip_frame_set_band_mode(ovhip_frame *f, int on)
{
    if (!f) return OVHIP_EINVAL;
    f->band_mode = on != 0;
    if (on && !f->dry && f->job && !getenv("OVVC_HIP_NO_RESERVE")) return fail(f, ovhip_job_band_reserve(f->job), "ovhip_job_band_reserve");
    return OVHIP_OK;
}

/* luma rows [0, need[i]) of reference i are read by the units [c0, c1) (0: not used): the unit's last row + the vertical vector +
 * 8 rows -- the 8-tap window's 4 rows below, DMVR's 2-sample search range, BDOF's extension, and the chroma windows (2 chroma rows
 * = 4 luma rows below the co-located row) all stay inside it */
static void
units_need_rows(ovhip_frame *f, const ovhip_band_counts *c0, const ovhip_band_counts *c1, int only_dmvr, int32_t *need)
{
    ovhip_recorder *rec = ovhip_frame_recorder(f);
    size_t n = 0;
    for (int i = 0; i < MAX_REFS; ++i) need[i] = 0;
    if (!rec) return;
#define NEED(ref, y_last, mvy) do { int32_t v_ = (int32_t)(y_last) + ((mvy) >> 4) + 1 + 8; if (v_ < 1) v_ = 1; if (v_ > f->h) v_ = f->h; \
                                    if ((ref) < MAX_REFS && need[ref] < v_) need[ref] = v_; } while (0)
    for (int pass = 0; pass < 2; ++pass) {
        const ovhip_mc_unit *u = pass ? ovhip_rec_mcx_units(rec, &n) : ovhip_rec_mc_units(rec, &n);
        const size_t a = pass ? c0->n_mcx : c0->n_mc, b = pass ? c1->n_mcx : c1->n_mc;
        for (size_t i = a; i < b && i < n; ++i) {
            if (only_dmvr && !(u[i].flags & OVHIP_MC_DMVR)) continue;
            const int y_last = u[i].y + u[i].h - 1;
            if (u[i].dir & 1) NEED(u[i].ref0, y_last, u[i].mv0y);
            if (u[i].dir & 2) NEED(u[i].ref1, y_last, u[i].mv1y);
        }
    }
    if (!only_dmvr) {
        const ovhip_aff_unit *au = ovhip_rec_aff_units(rec, &n);
        size_t ns = 0;
        const int32_t *side = ovhip_rec_aff_side(rec, &ns);
        for (size_t i = c0->n_aff; i < c1->n_aff && i < n; ++i) {
            const int n_sub = (au[i].w / 4) * (au[i].h / 4) + (au[i].w / 8) * (au[i].h / 8);
            const int y_last = au[i].y + au[i].h - 1;
            int32_t m0 = -(1 << 30), m1 = -(1 << 30);
            for (int k = 0; k < n_sub && (size_t)au[i].side_off + 4 * (size_t)k + 3 < ns; ++k) {
                const int32_t *q = side + au[i].side_off + 4 * k;
                if (q[1] > m0) m0 = q[1];
                if (q[3] > m1) m1 = q[3];
            }
            if (m0 == -(1 << 30)) m0 = m1 = f->h * 16;                 /* (no vectors found: the whole picture) */
            if (au[i].dir & 1) NEED(au[i].ref0, y_last, m0);
            if (au[i].dir & 2) NEED(au[i].ref1, y_last, m1);
        }
    }
#undef NEED
}

/* 1: every reference has the rows (pinned, f->ref_pic[] filled), 0: not yet (block == 0 only), < 0: error */
static int
refs_rows_ready(ovhip_frame *f, const int32_t *need, int block)
{
    for (int i = 0; i < f->n_refs; ++i) {
        /* a table entry none of these units reads stays a placeholder of the right geometry (the launchers check every entry of the
         * table; no unit of the launch indexes this one): the picture being decoded itself */
        if (!need[i]) { if (!f->ref_pinned[i]) f->ref_pic[i] = f->dst; continue; }
        void *ev = NULL;
        ovhip_pic pic;
        int r = ovhip_dpb_rows_tag(f->dpb, f->ref_key[i], f->ref_tag[i], f->dev, need[i], block, !f->ref_pinned[i], &pic, &ev);
        if (r < 0) return fail(f, r, r == OVHIP_EREF ? "a reference picture failed to decode" : "reference picture unknown to the device DPB");
        if (!r) return 0;
        f->ref_pic[i] = pic; f->ref_pinned[i] = 1;
        if (ev) {
            r = ovhip_dpb_wait_copy(f->dpb, f->dev, ev);
            if (r != OVHIP_OK) return fail(f, r, "transfer of a reference picture");
        }
    }
    return 1;
}

/* rows a band-wise job has made final once the band ending at `end` has had its filters (ovvc_picture.hip: deblocking leaves
 * the 8 rows above a band's end to the next band, SAO and ALF follow in steps of 8 rows) -- what a dry frame posts */
static int32_t
dry_rows_after(int32_t end)
{
    int32_t v = end - 8 - 1;
    v = v <= 0 ? 0 : v & ~7;
    v -= 3;
    return v <= 0 ? 0 : v & ~7;
}

static int frame_band(ovhip_frame *f, const ovhip_job_params *params, int32_t row_end, int32_t last, ovhip_frame_output *out, const ovhip_band_counts *upto, int block);

int ovhip_frame_band(ovhip_frame *f, const ovhip_job_params *params, int32_t row_end, int32_t last, ovhip_frame_output *out)
{
    return frame_band(f, params, row_end, last, out, NULL, last != 0);
}

int ovhip_frame_band_upto(ovhip_frame *f, const ovhip_job_params *params, int32_t row_end, const ovhip_band_counts *upto, int32_t block)
{
    return frame_band(f, params, row_end, 0, NULL, upto, block != 0);
}

static int
frame_band(ovhip_frame *f, const ovhip_job_params *params, int32_t row_end, int32_t last, ovhip_frame_output *out, const ovhip_band_counts *upto, int block)
{
    if (!f || !params || !f->live) return OVHIP_EINVAL;
    int r = f->status;
    if (r != OVHIP_OK) {
        /* a latched error: nothing more goes to the device; the last call publishes the picture as failed */
        trace(f, OVHIP_FE_BAND, f->key, 0, row_end, last, r);
        if (last) (void)publish(f, r);
        return r;
    }
    if (last) row_end = f->h;
    if (row_end > f->h) row_end = f->h;
    ovhip_job *j = f->dry ? NULL : ovhip_frame_job(f);
    if (!f->dry && !j) return fail(f, OVHIP_EINVAL, "ovhip_frame_band: no job");
    /* the device is behind (the band before this one is still being reconstructed): this band is left to the next call, which takes
     * its rows too -- fewer, fuller launches when the device is the slower side; an I picture's wavefront then spans the rows */
    /* (a picture of more CTU rows than the job keeps bands -- 8K with 32-sample CTUs -- goes on in fewer, larger bands: the rest is
     * left to the last call) */
    if (!last && f->n_bands >= 90) { f->n_deferred++; trace(f, OVHIP_FE_BAND, f->key, 0, row_end, last, 0); return 0; }
    if (!block && j && f->n_bands && ovhip_job_band_busy(j)) { f->n_deferred++; trace(f, OVHIP_FE_BAND, f->key, 0, row_end, last, 0); return 0; }
    ovhip_band_counts now;
    ovhip_rec_counts(ovhip_frame_recorder(f), &now);
    if (upto) {
        const uint32_t *u = &upto->n_tb, *m = &now.n_tb, *lo = &f->band_prev.n_tb;
        for (int i = 0; i < 10; ++i) if (u[i] > m[i] || u[i] < lo[i]) return fail(f, OVHIP_EINVAL, "ovhip_frame_band_upto: counts outside what has been recorded since the last band");
        now = *upto;
    }
    int32_t need[MAX_REFS];
    { PT0();
    units_need_rows(f, &f->band_prev, &now, 0, need);
    r = refs_rows_ready(f, need, block);
    if (last) PT(f->pt_final_refs); else PT(f->pt_band_refs); }
    if (r < 0) { trace(f, OVHIP_FE_BAND, f->key, 0, row_end, last, r); if (last) (void)publish(f, r); return r; }
    if (!r) { f->n_deferred++; trace(f, OVHIP_FE_BAND, f->key, 0, row_end, last, 0); return 0; }
    int32_t rows = 0; void *ev = NULL; const volatile uint32_t *abw = NULL;
    if (f->dry) {
        rows = last ? f->h : dry_rows_after(row_end);
        r = OVHIP_OK;
    } else {
        PT0();
        r = ovhip_job_band(j, &f->dst, f->ref_pic, (uint32_t)f->n_refs, params, upto, row_end, last);
        PT(f->pt_band_job);
        if (r != OVHIP_OK) fail(f, r, "ovhip_job_band");
        else (void)ovhip_job_band_progress(j, &rows, &ev, &abw);
    }
    f->band_prev = now; f->band_row_prev = row_end; f->n_bands++;
    trace(f, OVHIP_FE_BAND, f->key, 0, row_end, last, r == OVHIP_OK ? 1 : r);
    if (r == OVHIP_OK && !last && rows > f->band_rows_posted) {
        (void)ovhip_dpb_post_rows(f->dpb, f->key, rows, ev, abw);
        f->band_rows_posted = rows;
    }
    if (!last) return r == OVHIP_OK ? 1 : r;
    /* the picture's end: as ovhip_frame_submit -- only the wait marks it complete */
    if (r == OVHIP_OK && !f->dry) {
        PT0();
        int q = ovhip_job_wait(j);
        PT(f->pt_job_wait);
        f->done_at = mono_s();
        if (q != OVHIP_OK) r = fail(f, q, "ovhip_job_wait");
    }
    if (f->status) r = f->status;
    (void)publish(f, r);
    f->pn_pics++;
    if (r == OVHIP_OK && !f->dry && out && out->mode != OVHIP_OUT_NONE) {
        PT0();
        switch (out->mode) {
        case OVHIP_OUT_DIGEST: r = ovhip_pic_digest(f->ctx, &f->dst, &out->window, out->digest); break;
        case OVHIP_OUT_PLANES: r = ovhip_pic_download(f->ctx, &f->dst, out->y, out->cb, out->cr, out->stride_y, out->stride_c); break;
        case OVHIP_OUT_PACKED: r = ovhip_pic_output(f->ctx, &f->dst, &out->window, out->packed); break;
        default: r = OVHIP_EINVAL;
        }
        PT(f->pt_output);
        if (r != OVHIP_OK) fail(f, r, "picture output");
    }
    return r == OVHIP_OK ? 1 : r;
}

int
ovhip_frame_band_stats(const ovhip_frame *f, int32_t *n_bands, int32_t *n_deferred)
{
    if (!f) return OVHIP_EINVAL;
    if (n_bands) *n_bands = f->n_bands;
    if (n_deferred) *n_deferred = f->n_deferred;
    return OVHIP_OK;
}
