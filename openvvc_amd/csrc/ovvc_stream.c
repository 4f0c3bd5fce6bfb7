/* ovvc_stream.c -- the frame-thread pool of the device path, in C (include/ovvc_hip.h, "Stream driver").
 *
 * The reference deals pictures to sub-decoders in decoding order (ovdec_select_subdec, ovdec.c:188-248); a sub-decoder parses and
 * reconstructs its picture on its own thread, waiting for reference pictures as it meets them (rcn_inter.c:131-146), and the
 * application takes finished pictures in output order (ovdec_receive_picture / ovdpb_output_pic; dectest.c:304-368).  Here:
 *
 *   frame thread   takes the next picture of its device in decoding order, records it (call-log replay = the parse thread's
 *                  share of the device path) or takes its pre-recorded job, ovhip_frame_submit (uploads, wait for references,
 *                  launches, wait, publish), drops its holds on the reference pictures
 *   output thread  pictures in POC order: waits for each, crops / packs / copies it out (or fingerprints it), drops the hold
 *   comm thread    multi-process runs: hands pictures other ranks need to xfer.send and receives pictures decoded elsewhere
 *                  into the DPB, in the one global transfer order every rank derives (producer picture, then destination)
 *
 * A picture leaves the device DPB when its decode, its readers, the output and the sends have all dropped their hold
 * (dpb.c: ovdpb_unref_pic when neither a reference list nor the output process needs the frame).
 */
#define _GNU_SOURCE
#include <pthread.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>
#include "ovvc_hip.h"
#include "ovvc_dpb_priv.h"

#define STAGE_ALL (OVHIP_STAGE_MC | OVHIP_STAGE_ITX | OVHIP_STAGE_DBF | OVHIP_STAGE_SAO | OVHIP_STAGE_ALF | OVHIP_STAGE_INTRA)

struct run_state;

int ovhip_upload_streams_clear_of_(ovhip_ctx *keep_clear);     /* ovvc_picture.hip */

struct ovhip_stream {
    ovhip_dpb *dpb;
    ovhip_stream_cfg cfg;
    const ovhip_stream_content *contents; uint32_t n_contents;
    ovhip_job *const *jobs; uint32_t n_jobs;
    pthread_mutex_t *job_mtx;
    int n_dev, tpd, n_reg;                /* frames per device: n_reg in-order frame threads + (tpd - n_reg) look-ahead threads */
    ovhip_frame **frames;                 /* [n_dev * tpd] */
    ovhip_ctx **out_ctx;                  /* output thread: one context per device */
    void *out_host; size_t out_host_bytes;
    /* the stream being decoded (continued over several runs) */
    const ovhip_stream_pic *pics; uint32_t n_total;
    uintptr_t key_base;
    uint32_t *holds;                      /* per picture: decode / receive + local readers + output + sends still to come */
    unsigned char *alive;                 /* begun in the DPB and not released yet */
    unsigned char *begun;                 /* has entered the DPB at some point (the output / comm threads wait for that before they
                                           * ask the DPB for it: a key the DPB never saw is an error there, not a wait) */
    pthread_mutex_t begun_mtx; pthread_cond_t begun_cnd;
    int n_moved, n_sharing;               /* streams replaced to clear the look-ahead thread's hardware queue / still sharing it */
    uint8_t *dg;                          /* OVHIP_OUT_DIGEST: the pictures' digests, computed by their frame threads (begun[idx] == 2: there) */
};

struct dev_queue { uint32_t *order; unsigned char *taken; uint32_t n, next; pthread_mutex_t take; pthread_cond_t moved; };

struct run_state {
    ovhip_stream *s;
    const ovhip_stream_pic *pics;
    uint32_t first, n, flags;
    uint8_t *digests;
    struct dev_queue *q;
    pthread_mutex_t mtx;                  /* result fields */
    ovhip_stream_result *res;
    volatile int abort;
    uint32_t *out_order; uint32_t n_out;
    ovhip_md5_state md5;
    double t0; double *trace;
};

struct thread_arg { struct run_state *rs; int dev, t; };

static double now_s(void) { struct timespec t; clock_gettime(CLOCK_MONOTONIC, &t); return (double)t.tv_sec + 1e-9 * (double)t.tv_nsec; }
static const void *key_of(const ovhip_stream *s, uint32_t idx) { return (const void *)(s->key_base + idx + 1); }
static int is_local(const ovhip_stream *s, const ovhip_stream_pic *p) { return !s->cfg.xfer || p->owner == s->cfg.rank; }

static void
run_fail(struct run_state *rs, int code, const char *what, const char *detail)
{
    pthread_mutex_lock(&rs->mtx);
    if (!rs->res->status) {
        rs->res->status = code ? code : OVHIP_EINVAL;
        snprintf(rs->res->error, sizeof(rs->res->error), "%s%s%s", what, detail && detail[0] ? ": " : "", detail ? detail : "");
    }
    pthread_mutex_unlock(&rs->mtx);
    rs->abort = 1;
    ovhip_dpb_shutdown(rs->s->dpb);        /* nobody keeps waiting for a picture that will not come */
    for (int k = 0; rs->q && k < rs->s->n_dev; ++k) {
        pthread_mutex_lock(&rs->q[k].take); pthread_cond_broadcast(&rs->q[k].moved); pthread_mutex_unlock(&rs->q[k].take);
    }
    pthread_mutex_lock(&rs->s->begun_mtx);
    pthread_cond_broadcast(&rs->s->begun_cnd);
    pthread_mutex_unlock(&rs->s->begun_mtx);
}

static void
mark_begun(ovhip_stream *s, uint32_t idx)
{
    pthread_mutex_lock(&s->begun_mtx);
    s->alive[idx] = 1; s->begun[idx] = 1;
    pthread_cond_broadcast(&s->begun_cnd);
    pthread_mutex_unlock(&s->begun_mtx);
}

/* 0 once picture idx has entered the DPB, != 0 when the run was aborted first */
static int
wait_begun(struct run_state *rs, uint32_t idx)
{
    ovhip_stream *s = rs->s;
    pthread_mutex_lock(&s->begun_mtx);
    while (!s->begun[idx] && !rs->abort) pthread_cond_wait(&s->begun_cnd, &s->begun_mtx);
    const int ok = s->begun[idx];
    pthread_mutex_unlock(&s->begun_mtx);
    return !ok;
}

static void
drop_hold(ovhip_stream *s, uint32_t idx)
{
    if (__atomic_sub_fetch(&s->holds[idx], 1, __ATOMIC_ACQ_REL) == 0 && s->alive[idx]) {
        s->alive[idx] = 0;
        (void)ovhip_dpb_release(s->dpb, key_of(s, idx));
    }
}

/* distinct reference pictures of p (a picture may sit in several table entries) */
static int
distinct_refs(const ovhip_stream_pic *p, uint32_t out[OVHIP_STREAM_MAX_REFS])
{
    int n = 0;
    for (int k = 0; k < p->n_refs && k < OVHIP_STREAM_MAX_REFS; ++k) {
        int seen = 0;
        for (int j = 0; j < n; ++j) seen |= out[j] == p->refs[k];
        if (!seen) out[n++] = p->refs[k];
    }
    return n;
}

static void
release_stream(ovhip_stream *s)
{
    for (uint32_t i = 0; i < s->n_total; ++i)
        if (s->alive && s->alive[i]) { s->alive[i] = 0; (void)ovhip_dpb_release(s->dpb, key_of(s, i)); }
    free(s->holds); free(s->alive); free(s->begun); free(s->dg);
    s->holds = NULL; s->alive = NULL; s->begun = NULL; s->dg = NULL;
    s->key_base += (uintptr_t)s->n_total + 1;
    s->pics = NULL; s->n_total = 0;
}

static int
adopt_stream(ovhip_stream *s, const ovhip_stream_pic *pics, uint32_t n_total, uint32_t flags)
{
    release_stream(s);
    s->holds = (uint32_t *)calloc(n_total ? n_total : 1, sizeof(uint32_t));
    s->alive = (unsigned char *)calloc(n_total ? n_total : 1, 1);
    s->begun = (unsigned char *)calloc(n_total ? n_total : 1, 1);
    s->dg = (uint8_t *)calloc(n_total ? n_total : 1, 16);
    if (!s->holds || !s->alive || !s->begun || !s->dg) return OVHIP_ENOMEM;
    for (uint32_t i = 0; i < n_total; ++i) {
        const ovhip_stream_pic *p = &pics[i];
        if (p->content >= s->n_contents || p->device >= (uint32_t)s->n_dev || p->n_refs > OVHIP_STREAM_MAX_REFS) return OVHIP_EINVAL;
        const int local = is_local(s, p);
        uint32_t refs[OVHIP_STREAM_MAX_REFS];
        const int nr = distinct_refs(p, refs);
        for (int k = 0; k < nr; ++k) {
            if (refs[k] >= i) return OVHIP_EINVAL;                  /* decoding order: references come first */
            if (local) s->holds[refs[k]]++;
        }
        if (local) {
            s->holds[i] += 1 + (s->cfg.output != OVHIP_OUT_NONE) + ((flags & OVHIP_STREAM_HOLD_ALL) != 0);
            for (uint32_t m = s->cfg.xfer ? p->send_mask : 0; m; m &= m - 1) s->holds[i]++;
        }
    }
    /* a picture decoded elsewhere that a local picture lists: one hold for its reception */
    for (uint32_t i = 0; i < n_total; ++i) if (!is_local(s, &pics[i]) && s->holds[i]) s->holds[i]++;
    s->pics = pics; s->n_total = n_total;
    return OVHIP_OK;
}

/* ---------------------------------------------------------------- frame threads */
/* the parameter block of a picture's flush */
static ovhip_job_params
picture_params(const struct run_state *rs, const ovhip_stream_content *c)
{
    const ovhip_stream *s = rs->s;
    ovhip_job_params pr = c->params;
    pr.stages = (pr.stages ? pr.stages : STAGE_ALL) | s->cfg.extra_stages | ((rs->flags & OVHIP_STREAM_RESIDENT) ? OVHIP_STAGE_RESIDENT : 0);
    if (pr.stages == STAGE_ALL) pr.stages = 0;
    return pr;
}

static void
decode_picture(struct run_state *rs, ovhip_frame *f, uint32_t idx, int job_locked, int ahead, int thread)
{
    double *tr = rs->trace ? rs->trace + 8 * (size_t)(idx - rs->first) : NULL;
    if (tr) { tr[0] = now_s() - rs->t0; tr[3] = (double)thread; }
    ovhip_stream *s = rs->s;
    const ovhip_stream_pic *p = &rs->pics[idx];
    const ovhip_stream_content *c = &s->contents[p->content];
    const int record = (rs->flags & OVHIP_STREAM_RECORD) != 0;
    ovhip_job *job = record ? NULL : s->jobs[p->job];
    int r = ovhip_frame_begin(f, key_of(s, idx));
    if (r != OVHIP_OK) { run_fail(rs, r, "ovhip_frame_begin", ovhip_frame_last_error(f)); goto out; }
    mark_begun(s, idx);
    for (uint32_t k = 0; p->n_refs && k < c->n_ref_slots && r >= 0; ++k) r = ovhip_frame_ref_at(f, (int)k, key_of(s, p->refs[k % p->n_refs]));
    if (r < 0) { (void)ovhip_frame_fail(f, r); run_fail(rs, r, "ovhip_frame_ref_at", ovhip_frame_last_error(f)); goto out; }
    if (record) {
        /* what the parse thread does for the device path: every slot call of the picture, into this thread's recorder */
        const double t0 = now_s();
        ovhip_recorder *rec = ovhip_frame_recorder(f);
        int64_t nc = rec ? ovhip_calllog_replay(c->calllog, c->calllog_bytes, rec) : OVHIP_ENOMEM;
        const double dt = now_s() - t0;
        pthread_mutex_lock(&rs->mtx); rs->res->record_seconds += dt; pthread_mutex_unlock(&rs->mtx);
        if (nc < 0) { (void)ovhip_frame_fail(f, (int)nc); run_fail(rs, (int)nc, "ovhip_calllog_replay", ""); goto out; }
    }
    {
        ovhip_job_params pr = picture_params(rs, c);
        ovhip_frame_output out;
        memset(&out, 0, sizeof(out));
        /* the fingerprint is taken by the picture's own frame thread, after the picture was published (16 threads hash 16 pictures;
         * the output thread only puts them in output order) */
        out.mode = ((rs->flags & OVHIP_STREAM_DIGESTS) || s->cfg.output == OVHIP_OUT_DIGEST) ? OVHIP_OUT_DIGEST : OVHIP_OUT_NONE;
        out.window = s->cfg.window;
        if (tr) tr[1] = now_s() - rs->t0;
        const double t_sub = now_s();
        r = ovhip_frame_submit(f, job, NULL, &pr, &out);
        const double dt_sub = now_s() - t_sub;
        if (tr) { tr[6] = now_s() - rs->t0; tr[2] = tr[6]; }
        if (r != OVHIP_OK) { run_fail(rs, r, "ovhip_frame_submit", ovhip_frame_last_error(f)); goto out; }
        if (out.mode == OVHIP_OUT_DIGEST) {
            if (rs->digests) memcpy(rs->digests + 16 * (size_t)(idx - rs->first), out.digest, 16);
            pthread_mutex_lock(&s->begun_mtx);
            memcpy(s->dg + 16 * (size_t)idx, out.digest, 16);
            s->begun[idx] = 2;
            pthread_cond_broadcast(&s->begun_cnd);
            pthread_mutex_unlock(&s->begun_mtx);
            /* the output thread needs these 16 bytes, not the picture: the picture's hold for the output ends HERE, not when its turn
             * in output order comes (a key picture is decoded 31 pictures before it is output: held that long, the DPB ran with twice
             * the device pictures) */
            if (s->cfg.output == OVHIP_OUT_DIGEST) drop_hold(s, idx);
        }
        ovhip_job_stats st;
        if (ovhip_job_last_stats(job ? job : ovhip_frame_job(f), &st) == OVHIP_OK) {
            pthread_mutex_lock(&rs->mtx);
            rs->res->n_decoded++; rs->res->n_second_passes += st.n_ordered_retries;
            const double ph[4] = { 1e-6 * st.host_us_prepare, 1e-6 * st.host_us_upload, 1e-6 * st.host_us_wait, 1e-6 * st.host_us_launch };
            if (tr) {
                tr[4] = tr[1] + ph[0] + ph[1] + ph[2];           /* references in hand */
                tr[5] = tr[4] + ph[3];                            /* launches enqueued */
                tr[2] = ovhip_frame_published_at(f) - rs->t0;     /* published (before the output) */
                tr[7] = ovhip_frame_done_at(f) - rs->t0;          /* complete on the device (ovhip_job_wait returned) */
            }
            for (int k = 0; k < 4; ++k) rs->res->host_seconds[k] += ph[k];
            rs->res->host_seconds[4] += dt_sub - ph[0] - ph[1] - ph[2] - ph[3];      /* ovhip_job_wait, publish, output */
            pthread_mutex_unlock(&rs->mtx);
        }
    }
out:
    if (job_locked) pthread_mutex_unlock(&s->job_mtx[p->job]);
    {
        /* this picture's decode is over (or will never happen): its references lose a reader, it loses the hold of its own decode */
        uint32_t refs[OVHIP_STREAM_MAX_REFS];
        const int nr = distinct_refs(p, refs);
        for (int k = 0; k < nr; ++k) drop_hold(s, refs[k]);
        drop_hold(s, idx);
    }
}

static void *
frame_thread(void *argp)
{
    struct thread_arg *a = (struct thread_arg *)argp;
    struct run_state *rs = a->rs;
    ovhip_stream *s = rs->s;
    struct dev_queue *q = &rs->q[a->dev];
    ovhip_frame *f = s->frames[a->dev * s->tpd + a->t];
    const int ahead = a->t >= s->n_reg;            /* a look-ahead thread: pictures WITHOUT reference pictures, before their turn */
    for (;;) {
        pthread_mutex_lock(&q->take);
        uint32_t idx = 0;
        int have = 0;
        while (!have) {
            while (q->next < q->n && q->taken[q->next]) ++q->next;
            if (rs->abort || q->next >= q->n) break;
            if (!ahead) { q->taken[q->next] = 1; idx = q->order[q->next++]; have = 1; pthread_cond_broadcast(&q->moved); break; }
            /* An intra picture depends on nothing: started up to intra_lookahead pictures before its turn in decoding order it runs
             * beside the pictures that precede it, instead of in front of the pictures that wait for it (what a decoder with that
             * many frame threads would do by itself; here one thread and one picture buffer do it) */
            const uint32_t lim = q->next + (uint32_t)s->cfg.intra_lookahead < q->n ? q->next + (uint32_t)s->cfg.intra_lookahead : q->n;
            for (uint32_t k = q->next; k < lim && !have; ++k)
                if (!q->taken[k] && rs->pics[q->order[k]].n_refs == 0) { q->taken[k] = 1; idx = q->order[k]; have = 1; }
            if (!have) pthread_cond_wait(&q->moved, &q->take);          /* the window moves when the in-order threads take pictures */
        }
        if (!have) { pthread_cond_broadcast(&q->moved); pthread_mutex_unlock(&q->take); break; }
        int locked = 0;
        if (!(rs->flags & OVHIP_STREAM_RECORD)) {
            /* a pre-recorded job is in flight once at a time; taken in decoding order (still under the queue's lock), so a
             * later picture can never hold a job an earlier one -- which it may depend on -- is waiting for */
            pthread_mutex_lock(&s->job_mtx[rs->pics[idx].job]);
            locked = 1;
        }
        pthread_mutex_unlock(&q->take);
        decode_picture(rs, f, idx, locked, ahead, a->dev * s->tpd + a->t);
    }
    return NULL;
}

/* ---------------------------------------------------------------- output thread */
static void *
output_thread(void *argp)
{
    struct run_state *rs = (struct run_state *)argp;
    ovhip_stream *s = rs->s;
    for (uint32_t k = 0; k < rs->n_out && !rs->abort; ++k) {
        const uint32_t idx = rs->out_order[k];
        const ovhip_stream_pic *p = &rs->pics[idx];
        ovhip_pic pic;
        if (s->cfg.output == OVHIP_OUT_DIGEST) {
            pthread_mutex_lock(&s->begun_mtx);
            while (s->begun[idx] != 2 && !rs->abort) pthread_cond_wait(&s->begun_cnd, &s->begun_mtx);
            const int ok = s->begun[idx] == 2;
            pthread_mutex_unlock(&s->begun_mtx);
            if (!ok) break;
            ovhip_md5_update(&rs->md5, s->dg + 16 * (size_t)idx, 16);
            rs->res->out_bytes += 16; rs->res->out_frames++;
            continue;                                  /* (the picture's output hold was dropped by its frame thread) */
        }
        if (wait_begun(rs, idx)) break;
        int r = ovhip_dpb_acquire(s->dpb, key_of(s, idx), p->device, &pic, NULL);
        if (r != OVHIP_OK) { if (!rs->abort) run_fail(rs, r, "output: picture not available", ""); break; }
        ovhip_ctx *ctx = s->out_ctx[p->device];
        r = ovhip_pic_output(ctx, &pic, &s->cfg.window, s->out_host);
        if (r == OVHIP_OK) {
            if (rs->flags & OVHIP_STREAM_FILE_MD5) ovhip_md5_update(&rs->md5, s->out_host, s->out_host_bytes);
            rs->res->out_bytes += s->out_host_bytes;
        }
        (void)ovhip_dpb_unpin(s->dpb, key_of(s, idx));
        if (r != OVHIP_OK) { run_fail(rs, r, "output", ovhip_last_error(ctx)); break; }
        rs->res->out_frames++;
        drop_hold(s, idx);
    }
    return NULL;
}

/* ---------------------------------------------------------------- multi-process exchange */
static void *
comm_thread(void *argp)
{
    struct run_state *rs = (struct run_state *)argp;
    ovhip_stream *s = rs->s;
    const ovhip_stream_xfer *x = s->cfg.xfer;
    const int rank = s->cfg.rank;
    for (uint32_t idx = rs->first; idx < rs->first + rs->n && !rs->abort; ++idx) {
        const ovhip_stream_pic *p = &rs->pics[idx];
        if (p->owner == rank) {
            if (!p->send_mask) continue;
            ovhip_pic pic;
            if (wait_begun(rs, idx)) break;
            int r = ovhip_dpb_acquire(s->dpb, key_of(s, idx), p->device, &pic, NULL);
            if (r != OVHIP_OK) { if (!rs->abort) run_fail(rs, r, "send: picture not available", ""); break; }
            for (uint32_t m = p->send_mask; m && r == OVHIP_OK; m &= m - 1) {
                const int dst = __builtin_ctz(m);
                r = x->send(x->user, idx, &pic, dst);
                if (r == OVHIP_OK) { rs->res->n_sent++; drop_hold(s, idx); }
            }
            (void)ovhip_dpb_unpin(s->dpb, key_of(s, idx));
            if (r != OVHIP_OK) { run_fail(rs, r, "xfer.send", ""); break; }
        } else if (((p->send_mask >> rank) & 1) && s->holds[idx]) {
            ovhip_pic pic;
            const int dev = 0;           /* one process per GPU: its only device */
            int r = ovhip_dpb_begin(s->dpb, key_of(s, idx), dev, s->cfg.w, s->cfg.h, &pic);
            if (r != OVHIP_OK) { run_fail(rs, r, "recv: ovhip_dpb_begin", ""); break; }
            mark_begun(s, idx);
            r = x->recv(x->user, idx, &pic, p->owner);
            (void)ovhip_dpb_publish(s->dpb, key_of(s, idx), r);
            if (r != OVHIP_OK) { run_fail(rs, r, "xfer.recv", ""); break; }
            rs->res->n_received++;
            drop_hold(s, idx);
        }
    }
    return NULL;
}

/* ---------------------------------------------------------------- life cycle */
int
ovhip_stream_create(ovhip_stream **out, ovhip_dpb *dpb, const ovhip_stream_cfg *cfg, const ovhip_stream_content *contents,
                    uint32_t n_contents, ovhip_job *const *jobs, uint32_t n_jobs)
{
    if (!out || !dpb || !cfg || !contents || !n_contents || cfg->w <= 0 || cfg->h <= 0 || cfg->threads_per_device < 1 || (n_jobs && !jobs))
        return OVHIP_EINVAL;
    *out = NULL;
    ovhip_stream *s = (ovhip_stream *)calloc(1, sizeof(*s));
    if (!s) return OVHIP_ENOMEM;
    pthread_mutex_init(&s->begun_mtx, NULL); pthread_cond_init(&s->begun_cnd, NULL);
    s->dpb = dpb; s->cfg = *cfg; s->contents = contents; s->n_contents = n_contents; s->jobs = jobs; s->n_jobs = n_jobs;
    s->n_dev = ovhip_dpb_n_devices(dpb); s->n_reg = cfg->threads_per_device;
    s->tpd = s->n_reg + (cfg->intra_lookahead > 0 ? 1 : 0);
    /* several stream objects may share one DPB (bench.py: one per configuration): each gets a key space of its own */
    static uintptr_t next_space = 1;
    s->key_base = __atomic_fetch_add(&next_space, 1, __ATOMIC_RELAXED) << 40;
    int r = OVHIP_OK;
    s->frames = (ovhip_frame **)calloc((size_t)s->n_dev * s->tpd, sizeof(*s->frames));
    s->out_ctx = (ovhip_ctx **)calloc((size_t)s->n_dev, sizeof(*s->out_ctx));
    s->job_mtx = (pthread_mutex_t *)calloc(n_jobs ? n_jobs : 1, sizeof(*s->job_mtx));
    if (!s->frames || !s->out_ctx || !s->job_mtx) r = OVHIP_ENOMEM;
    for (uint32_t i = 0; i < n_jobs && r == OVHIP_OK; ++i) pthread_mutex_init(&s->job_mtx[i], NULL);
    for (int i = 0; i < s->n_dev * s->tpd && r == OVHIP_OK; ++i)
        r = ovhip_frame_create(dpb, i / s->tpd, cfg->w, cfg->h, &s->frames[i]);
    /* The look-ahead thread's stream gets a hardware queue to itself: an in-order thread whose stream shares it would sit behind an
     * I picture's ordered pass for milliseconds.  Which stream got which queue is measured (ovhip_ctx_shares_queue); a frame in the
     * wrong company takes new streams until it is out of it (the runtime deals its queues round robin: a few tries). */
    for (int k = 0; k < s->n_dev && r == OVHIP_OK && s->tpd > s->n_reg && cfg->ahead_own_queue; ++k) {
        ovhip_ctx *ahead = ovhip_frame_ctx(s->frames[k * s->tpd + s->n_reg]);
        for (int t = 0; t < s->n_reg; ++t) {
            ovhip_ctx *c = ovhip_frame_ctx(s->frames[k * s->tpd + t]);
            for (int tries = 0; tries < 12 && ovhip_ctx_shares_queue(ahead, c) == 1; ++tries) { if (ovhip_ctx_new_stream(c) != OVHIP_OK) break; s->n_moved++; }
            if (ovhip_ctx_shares_queue(ahead, c) == 1) s->n_sharing++;
        }
        /* the device's shared upload streams likewise: the event behind a picture's copies must not queue behind an I picture */
        { const int q = ovhip_upload_streams_clear_of_(ahead); if (q > 0) s->n_sharing += q; }
    }
    if (cfg->output == OVHIP_OUT_PACKED) {
        for (int k = 0; k < s->n_dev && r == OVHIP_OK; ++k) r = ovhip_ctx_create(&s->out_ctx[k], ovhip_dpb_device(dpb, k), NULL);
        if (r == OVHIP_OK) {
            s->out_host_bytes = ovhip_output_bytes(cfg->w, cfg->h, &cfg->window);
            s->out_host = s->out_host_bytes ? ovhip_host_alloc(s->out_host_bytes) : NULL;
            if (!s->out_host) r = s->out_host_bytes ? OVHIP_ENOMEM : OVHIP_EINVAL;
        }
    }
    if (r != OVHIP_OK) { ovhip_stream_destroy(s); return r; }
    *out = s;
    return OVHIP_OK;
}

void
ovhip_stream_destroy(ovhip_stream *s)
{
    if (!s) return;
    release_stream(s);
    for (int i = 0; s->frames && i < s->n_dev * s->tpd; ++i) ovhip_frame_destroy(s->frames[i]);
    for (int k = 0; s->out_ctx && k < s->n_dev; ++k) ovhip_ctx_destroy(s->out_ctx[k]);
    ovhip_host_free(s->out_host);
    for (uint32_t i = 0; s->job_mtx && i < s->n_jobs; ++i) pthread_mutex_destroy(&s->job_mtx[i]);
    free(s->frames); free(s->out_ctx); free(s->job_mtx);
    pthread_cond_destroy(&s->begun_cnd); pthread_mutex_destroy(&s->begun_mtx);
    free(s);
}

int ovhip_stream_queue_info(const ovhip_stream *s, int *moved, int *sharing)
{
    if (!s) return OVHIP_EINVAL;
    if (moved) *moved = s->n_moved;
    if (sharing) *sharing = s->n_sharing;
    return OVHIP_OK;
}

ovhip_frame *
ovhip_stream_frame(ovhip_stream *s, int dev, int thread)
{
    return s && dev >= 0 && dev < s->n_dev && thread >= 0 && thread < s->tpd ? s->frames[dev * s->tpd + thread] : NULL;
}

const void *ovhip_stream_key(const ovhip_stream *s, uint32_t idx) { return s && idx < s->n_total ? key_of(s, idx) : NULL; }

static int cmp_out(const void *a, const void *b, void *arg)
{
    const ovhip_stream_pic *pics = (const ovhip_stream_pic *)arg;
    const uint32_t i = *(const uint32_t *)a, j = *(const uint32_t *)b;
    if (pics[i].poc != pics[j].poc) return pics[i].poc < pics[j].poc ? -1 : 1;
    return i < j ? -1 : i > j;
}

int
ovhip_stream_run(ovhip_stream *s, const ovhip_stream_pic *pics, uint32_t n_total, uint32_t first, uint32_t n, uint32_t flags, uint8_t *digests,
                 ovhip_stream_result *res)
{
    if (!s || !pics || !res || first > n_total || n > n_total - first) return OVHIP_EINVAL;
    double *trace = res->trace;
    memset(res, 0, sizeof(*res));
    res->trace = trace;
    flags |= s->cfg.flags;
    if (!(flags & OVHIP_STREAM_RECORD)) {
        if (!s->n_jobs) return OVHIP_EINVAL;
        for (uint32_t i = first; i < first + n; ++i) if (pics[i].job >= s->n_jobs) return OVHIP_EINVAL;
    } else {
        for (uint32_t i = first; i < first + n; ++i)
            if (pics[i].content < s->n_contents && !s->contents[pics[i].content].calllog) return OVHIP_EINVAL;
    }
    int r;
    if (pics != s->pics || n_total != s->n_total || first == 0) {
        if ((r = adopt_stream(s, pics, n_total, flags)) != OVHIP_OK) { release_stream(s); return r; }
    }
    struct run_state rs;
    memset(&rs, 0, sizeof(rs));
    rs.s = s; rs.pics = pics; rs.first = first; rs.n = n; rs.flags = flags; rs.digests = digests; rs.res = res;
    pthread_mutex_init(&rs.mtx, NULL);
    ovhip_md5_init(&rs.md5);
    rs.q = (struct dev_queue *)calloc((size_t)s->n_dev, sizeof(*rs.q));
    rs.out_order = (uint32_t *)malloc((n ? n : 1) * sizeof(uint32_t));
    const int nthr = s->n_dev * s->tpd;
    pthread_t *th = (pthread_t *)calloc((size_t)nthr + 2, sizeof(*th));
    struct thread_arg *ta = (struct thread_arg *)calloc((size_t)nthr, sizeof(*ta));
    r = rs.q && rs.out_order && th && ta ? OVHIP_OK : OVHIP_ENOMEM;
    for (int k = 0; k < s->n_dev && r == OVHIP_OK; ++k) {
        pthread_mutex_init(&rs.q[k].take, NULL); pthread_cond_init(&rs.q[k].moved, NULL);
        rs.q[k].order = (uint32_t *)malloc((n ? n : 1) * sizeof(uint32_t));
        rs.q[k].taken = (unsigned char *)calloc(n ? n : 1, 1);
        if (!rs.q[k].order || !rs.q[k].taken) r = OVHIP_ENOMEM;
    }
    if (r == OVHIP_OK) {
        for (uint32_t i = first; i < first + n; ++i) {
            if (!is_local(s, &pics[i])) continue;
            struct dev_queue *q = &rs.q[pics[i].device];
            q->order[q->n++] = i;
            if (s->cfg.output != OVHIP_OUT_NONE) rs.out_order[rs.n_out++] = i;
        }
        qsort_r(rs.out_order, rs.n_out, sizeof(uint32_t), cmp_out, (void *)pics);
        int started = 0, aux = 0;
        const double t0 = now_s();
        rs.t0 = t0; rs.trace = trace;
        for (int i = 0; i < nthr; ++i) {
            ta[i].rs = &rs; ta[i].dev = i / s->tpd; ta[i].t = i % s->tpd;
            if (pthread_create(&th[i], NULL, frame_thread, &ta[i])) { run_fail(&rs, OVHIP_ENOMEM, "pthread_create", ""); break; }
            ++started;
        }
        if (rs.n_out && !rs.abort && !pthread_create(&th[nthr + aux], NULL, output_thread, &rs)) ++aux;
        if (s->cfg.xfer && !rs.abort && !pthread_create(&th[nthr + aux], NULL, comm_thread, &rs)) ++aux;
        for (int i = 0; i < started; ++i) pthread_join(th[i], NULL);
        for (int i = 0; i < aux; ++i) pthread_join(th[nthr + i], NULL);
        /* the devices are idle: every picture was waited for by its own thread, the output by the output thread */
        res->seconds = now_s() - t0;
        if (rs.n_out) ovhip_md5_final(&rs.md5, res->out_md5);
    } else {
        res->status = r;
    }
    if (rs.abort) ovhip_dpb_rearm_(s->dpb);
    for (int k = 0; rs.q && k < s->n_dev; ++k) { free(rs.q[k].order); free(rs.q[k].taken); pthread_cond_destroy(&rs.q[k].moved); pthread_mutex_destroy(&rs.q[k].take); }
    free(rs.q); free(rs.out_order); free(th); free(ta);
    pthread_mutex_destroy(&rs.mtx);
    if (res->status || !(flags & OVHIP_STREAM_KEEP)) release_stream(s);
    return res->status;
}
