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
#include "ovvc_hip.h"

#define MAX_REFS 16

struct ovhip_frame {
    ovhip_dpb *dpb;
    int dev;
    int32_t w, h;
    ovhip_ctx *ctx;
    ovhip_job *job;                       /* own job, created on first use */
    const void *key; ovhip_pic dst; int live;
    const void *ref_key[MAX_REFS]; uint64_t ref_tag[MAX_REFS]; ovhip_pic ref_pic[MAX_REFS]; void *ref_ev[MAX_REFS]; unsigned char ref_pinned[MAX_REFS];
    int n_refs;
    int status;
    char err[192];
};

static int
fail(ovhip_frame *f, int code, const char *what)
{
    if (code < 0 && !f->status) {
        f->status = code;
        snprintf(f->err, sizeof(f->err), "%s (%d)%s%s", what, code, f->ctx ? ": " : "", f->ctx ? ovhip_last_error(f->ctx) : "");
    }
    return code;
}

int ovhip_frame_create(ovhip_dpb *dpb, int dev, int32_t w, int32_t h, ovhip_frame **out) { return ovhip_frame_create_ex(dpb, dev, w, h, 0, out); }

int
ovhip_frame_create_ex(ovhip_dpb *dpb, int dev, int32_t w, int32_t h, int stream_priority, ovhip_frame **out)
{
    if (!dpb || !out || dev < 0 || dev >= ovhip_dpb_n_devices(dpb) || w <= 0 || h <= 0) return OVHIP_EINVAL;
    *out = NULL;
    const int hipdev = ovhip_dpb_device(dpb, dev);
    if (hipdev < 0) return OVHIP_ENODEV;          /* a DPB on a test back-end has no device to decode on */
    ovhip_frame *f = (ovhip_frame *)calloc(1, sizeof(*f));
    if (!f) return OVHIP_ENOMEM;
    f->dpb = dpb; f->dev = dev; f->w = w; f->h = h;
    int r = stream_priority ? ovhip_ctx_create_prio(&f->ctx, hipdev, stream_priority) : ovhip_ctx_create(&f->ctx, hipdev, NULL);
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
    ovhip_ctx_destroy(f->ctx);
    free(f);
}

ovhip_ctx *ovhip_frame_ctx(ovhip_frame *f) { return f ? f->ctx : NULL; }

ovhip_job *
ovhip_frame_job(ovhip_frame *f)
{
    if (!f) return NULL;
    if (!f->job && fail(f, ovhip_job_create(f->ctx, f->w, f->h, &f->job), "ovhip_job_create") != OVHIP_OK) f->job = NULL;
    return f->job;
}

ovhip_recorder *ovhip_frame_recorder(ovhip_frame *f) { ovhip_job *j = ovhip_frame_job(f); return j ? ovhip_job_recorder(j) : NULL; }
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
    if (r != OVHIP_OK) return fail(f, r, "ovhip_dpb_begin");
    f->key = key; f->live = 1;
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

static int before_launch_cb(void *user) { return acquire_refs((ovhip_frame *)user) != OVHIP_OK; }

int64_t
ovhip_frame_dmvr_rows(ovhip_frame *f)
{
    if (!f || !f->live || !f->job) return OVHIP_EINVAL;
    int r = acquire_refs(f);
    if (r != OVHIP_OK) return r;
    int64_t n = ovhip_job_dmvr_rows(f->job, f->ref_pic, (uint32_t)f->n_refs);
    if (n < 0) fail(f, (int)n, "ovhip_job_dmvr_rows");
    return n;
}

int64_t
ovhip_frame_dmvr_rows_begin(ovhip_frame *f, int32_t log2_ctu_s)
{
    if (!f || !f->live || !f->job) return OVHIP_EINVAL;
    /* the references are needed (and waited for) only if a unit the pass would cover is a DMVR unit: rows of BDOF-only units do not
     * stop the parse */
    int64_t c = ovhip_job_dmvr_rows_collect(f->job);
    if (c < 0) { fail(f, (int)c, "ovhip_job_dmvr_rows_collect"); return c; }
    size_t nu = 0;
    const ovhip_mc_unit *u = ovhip_rec_mcx_units(ovhip_job_recorder(f->job), &nu);
    int any = 0;
    for (size_t i = (size_t)c; i < nu && !any; ++i) any = (u[i].flags & OVHIP_MC_DMVR) != 0;
    if (any) {
        int r = acquire_refs(f);
        if (r != OVHIP_OK) return r;
    }
    int64_t n = ovhip_job_dmvr_rows_begin(f->job, f->ref_pic, (uint32_t)f->n_refs, log2_ctu_s);
    if (n < 0) fail(f, (int)n, "ovhip_job_dmvr_rows_begin");
    return n;
}

int64_t
ovhip_frame_dmvr_rows_collect(ovhip_frame *f)
{
    if (!f || !f->live || !f->job) return OVHIP_EINVAL;
    int64_t n = ovhip_job_dmvr_rows_collect(f->job);
    if (n < 0) fail(f, (int)n, "ovhip_job_dmvr_rows_collect");
    return n;
}

static int
publish(ovhip_frame *f, int status)
{
    if (!f->live) return OVHIP_EINVAL;
    f->live = 0;
    int r = ovhip_dpb_publish(f->dpb, f->key, status);
    unpin_refs(f);
    return r;
}

int
ovhip_frame_fail(ovhip_frame *f, int status)
{
    if (!f) return OVHIP_EINVAL;
    if (!f->status) f->status = status ? status : OVHIP_EINVAL;
    return publish(f, f->status);
}

int
ovhip_frame_submit(ovhip_frame *f, ovhip_job *job, const ovhip_pic *intra, const ovhip_job_params *params, ovhip_frame_output *out)
{
    if (!f || !params || !f->live) return OVHIP_EINVAL;
    int r = f->status;                    /* a latched recorder error: the picture is published as failed, never launched */
    ovhip_job *j = job ? job : f->job;
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
