/* ovvc_dpb.c -- device-side mirror of the decoded picture buffer (include/ovvc_hip.h, "Frame threads and the device DPB").
 *
 * Plain C + pthreads: the state machine only.  Memory and copies go through ovhip_dpb_ops (HIP back-end: ovvc_dpb_hip.hip), so
 * the waits, the release / re-use rules and the multi-device bookkeeping are exercised without a GPU (tests/test_dpb_cpu.py).
 *
 * What it mirrors in the reference (restated, nothing copied):
 *   ovdpb_init_picture / ovdpb_unref_pic            libovvc/dpb.c       picture enters / leaves the DPB
 *   ovdpb_report_decoded_ctu_line                   dpb.c:1309-1323     producer publishes progress (here: once, the whole picture)
 *   ovdpb_synchro_ref_decoded_ctus                  dpb.c:1242-1270     consumer waits for it (rcn_inter.c:131-146)
 *   frame pool re-use of an OVFrame                 ovframepool.c       a key may come back for a new picture once unreferenced
 */
#include <errno.h>
#include <pthread.h>
#include <stdlib.h>
#include <time.h>
#include <string.h>
#include "ovvc_hip.h"
#include "ovvc_dpb_priv.h"

enum { S_FREE = 0, S_DECODING, S_DONE, S_FAILED };
enum { C_NONE = 0, C_STARTED };
enum { PROG_MAX = 8 };

struct dpb_copy { ovhip_pic pic; int state; void *event; };

struct dpb_slot {
    const void *key;
    int state, status;
    int home;
    int32_t w, h;
    ovhip_pic pic;
    int pins, released;
    uint32_t want;
    uint64_t serial;
    uint64_t tag;                         /* caller's picture identity (0: none): a recycled key with another tag is another picture */
    struct dpb_copy copy[OVHIP_MAX_DEVICES];
    /* row progress of a picture that is decoded band by band (ovhip_dpb_post_rows): rows_final = what has been SEEN complete; prog[] =
     * the producer's records not seen complete yet, oldest first */
    int32_t rows_final;
    struct { int32_t rows; void *event; } prog[PROG_MAX]; int n_prog;
    const volatile uint32_t *abort_word;
};

struct pool_ent { ovhip_pic pic; int32_t w, h; };
/* a reclaimed picture that needs a blocking step before it may be handed out again: a copy into it may still run (event), or an
 * abandoned decode left hand-over bits in it (clear) */
struct limbo_ent { ovhip_pic pic; int dev; int32_t w, h; void *event; int clear; };

struct ovhip_dpb {
    pthread_mutex_t mtx;
    pthread_cond_t cnd;
    int n_dev;
    int hipdev[OVHIP_MAX_DEVICES];
    ovhip_dpb_ops ops;
    void *hip_user;                       /* owned HIP back-end state (ovhip_dpb_create), NULL with caller ops */
    struct dpb_slot *slots; size_t n_slots;
    struct pool_ent *pool[OVHIP_MAX_DEVICES]; size_t n_pool[OVHIP_MAX_DEVICES], cap_pool[OVHIP_MAX_DEVICES];
    int shutdown;
    int unknown_ms;                       /* how long ovhip_dpb_acquire waits for a key nobody has begun yet */
    uint64_t serial;
    struct limbo_ent *limbo; size_t n_limbo, cap_limbo;       /* finish_deferred() takes these to the pools OUTSIDE the mutex */
    struct { const void *key; uint64_t tag; uint32_t devs; } pend[64]; int n_pend;      /* wants for pictures nobody has begun yet */
    ovhip_dpb_stats st;
};

/* ---- free pools (per logical device; same-size pictures are re-used, others are freed when they get in the way) ---- */
static int
pool_push(ovhip_dpb *d, int dev, const ovhip_pic *pic, int32_t w, int32_t h)
{
    if (d->n_pool[dev] == d->cap_pool[dev]) {
        size_t nc = d->cap_pool[dev] ? 2 * d->cap_pool[dev] : 16;
        struct pool_ent *q = (struct pool_ent *)realloc(d->pool[dev], nc * sizeof(*q));
        if (!q) { ovhip_pic tmp = *pic; d->ops.pic_free(d->ops.user, dev, &tmp); return OVHIP_ENOMEM; }
        d->pool[dev] = q; d->cap_pool[dev] = nc;
    }
    struct pool_ent *e = &d->pool[dev][d->n_pool[dev]++];
    e->pic = *pic; e->w = w; e->h = h;
    d->st.n_pool++;
    return OVHIP_OK;
}

static int
pool_pop(ovhip_dpb *d, int dev, int32_t w, int32_t h, ovhip_pic *pic)
{
    for (size_t i = d->n_pool[dev]; i-- > 0;) {
        struct pool_ent *e = &d->pool[dev][i];
        if (e->w == w && e->h == h) {
            *pic = e->pic;
            *e = d->pool[dev][--d->n_pool[dev]];
            d->st.n_pool--; d->st.n_recycled++;
            return OVHIP_OK;
        }
    }
    /* a resolution change: pictures of the old size only hold memory */
    while (d->n_pool[dev]) {
        struct pool_ent *e = &d->pool[dev][--d->n_pool[dev]];
        d->ops.pic_free(d->ops.user, dev, &e->pic);
        d->st.n_pool--;
    }
    int r = d->ops.pic_alloc(d->ops.user, dev, w, h, pic);
    if (r == OVHIP_OK) d->st.n_alloc++;
    return r;
}

static struct dpb_slot *
find(ovhip_dpb *d, const void *key)
{
    for (size_t i = 0; i < d->n_slots; ++i)
        if (d->slots[i].state != S_FREE && d->slots[i].key == key) return &d->slots[i];
    return NULL;
}

/* The reader names the PICTURE it means (tag != 0): a slot that holds the key for another picture -- the frame pool handed the
 * OVFrame to a new picture whose frame thread has not begun it yet, and the slot still shows the previous, DONE picture -- is
 * "not there yet", exactly like an unknown key (predicting from the stale picture was silent). */
static struct dpb_slot *
find_tag(ovhip_dpb *d, const void *key, uint64_t tag)
{
    struct dpb_slot *s = find(d, key);
    return s && tag && s->tag && s->tag != tag ? NULL : s;
}

/* the blocking part of giving a picture back: wait for the copy into it / clear it (no mutex needed) */
static void
settle(ovhip_dpb *d, const struct limbo_ent *e)
{
    if (e->event) {
        (void)d->ops.copy_wait(d->ops.user, e->dev, e->event);          /* a copy nobody waited for may still be running */
        if (d->ops.copy_done) d->ops.copy_done(d->ops.user, e->dev, e->event);
    }
    if (e->clear && d->ops.pic_clear) (void)d->ops.pic_clear(d->ops.user, e->dev, &e->pic);
}

/* mutex held: the picture goes to its pool -- at once when nothing has to be waited for, else through the limbo list, which the
 * public call that got here empties after it has dropped the mutex (event waits, clears and the allocations behind
 * them ran under the DPB's one lock and stalled every begin / acquire / publish of every device) */
static void
give_back(ovhip_dpb *d, int dev, const ovhip_pic *pic, int32_t w, int32_t h, void *event, int clear)
{
    if (!event && !(clear && d->ops.pic_clear)) { (void)pool_push(d, dev, pic, w, h); return; }
    struct limbo_ent e = { *pic, dev, w, h, event, clear };
    if (d->n_limbo == d->cap_limbo) {
        const size_t nc = d->cap_limbo ? 2 * d->cap_limbo : 16;
        struct limbo_ent *q = (struct limbo_ent *)realloc(d->limbo, nc * sizeof(*q));
        if (!q) { settle(d, &e); (void)pool_push(d, dev, pic, w, h); return; }      /* no memory for the list: the old, blocking way */
        d->limbo = q; d->cap_limbo = nc;
    }
    d->limbo[d->n_limbo++] = e;
}

/* mutex NOT held */
static void
finish_deferred(ovhip_dpb *d)
{
    pthread_mutex_lock(&d->mtx);
    struct limbo_ent *l = d->limbo;
    const size_t n = d->n_limbo;
    d->limbo = NULL; d->n_limbo = d->cap_limbo = 0;
    pthread_mutex_unlock(&d->mtx);
    if (!n) { free(l); return; }
    for (size_t i = 0; i < n; ++i) settle(d, &l[i]);
    pthread_mutex_lock(&d->mtx);
    for (size_t i = 0; i < n; ++i) (void)pool_push(d, l[i].dev, &l[i].pic, l[i].w, l[i].h);
    pthread_mutex_unlock(&d->mtx);
    free(l);
}

/* leaves the mutex; settles what the call put into limbo */
static void
unlock_and_finish(ovhip_dpb *d)
{
    const int pending = d->n_limbo > 0;
    pthread_mutex_unlock(&d->mtx);
    if (pending) finish_deferred(d);
}

/* the slot's pictures go back to the pools (mutex held; nobody has it pinned) */
static void
reclaim(ovhip_dpb *d, struct dpb_slot *s)
{
    /* an incomplete decode may have left hand-over bits in the samples (ovhip_intra_flow_untag_launch) */
    give_back(d, s->home, &s->pic, s->w, s->h, NULL, s->state == S_FAILED || s->state == S_DECODING);
    d->st.n_live--;
    for (int k = 0; k < d->n_dev; ++k) {
        struct dpb_copy *c = &s->copy[k];
        if (c->state == C_NONE) continue;
        give_back(d, k, &c->pic, s->w, s->h, c->event, 0);
        d->st.n_live--;
        c->state = C_NONE; c->event = NULL;
    }
    s->state = S_FREE; s->key = NULL; s->pins = 0; s->released = 0; s->want = 0;
}

/* DONE picture -> device k (mutex held) */
static int
start_copy(ovhip_dpb *d, struct dpb_slot *s, int k)
{
    struct dpb_copy *c = &s->copy[k];
    if (k == s->home || c->state != C_NONE) return OVHIP_OK;
    int r = pool_pop(d, k, s->w, s->h, &c->pic);
    if (r != OVHIP_OK) return r;
    c->event = NULL;
    r = d->ops.copy_start(d->ops.user, k, &c->pic, s->home, &s->pic, &c->event);
    if (r != OVHIP_OK) { (void)pool_push(d, k, &c->pic, s->w, s->h); return r; }
    c->state = C_STARTED;
    d->st.n_live++; d->st.n_copies++;
    d->st.copy_bytes += (uint64_t)s->w * s->h * 3;
    return OVHIP_OK;
}

int
ovhip_dpb_create_ex(ovhip_dpb **out, int n_devices, const ovhip_dpb_ops *ops)
{
    if (!out || !ops || n_devices < 1 || n_devices > OVHIP_MAX_DEVICES || !ops->pic_alloc || !ops->pic_free || !ops->copy_start || !ops->copy_wait)
        return OVHIP_EINVAL;
    *out = NULL;
    ovhip_dpb *d = (ovhip_dpb *)calloc(1, sizeof(*d));
    if (!d) return OVHIP_ENOMEM;
    pthread_mutex_init(&d->mtx, NULL);
    pthread_cond_init(&d->cnd, NULL);
    d->n_dev = n_devices; d->ops = *ops;
    d->unknown_ms = 10000;
    for (int i = 0; i < OVHIP_MAX_DEVICES; ++i) d->hipdev[i] = -1;
    *out = d;
    return OVHIP_OK;
}

int
ovhip_dpb_create(ovhip_dpb **out, const int *devices, int n_devices)
{
    if (!out || !devices || n_devices < 1 || n_devices > OVHIP_MAX_DEVICES) return OVHIP_EINVAL;
    ovhip_dpb_ops ops;
    void *user = NULL;
    int r = ovhip_dpb_hip_ops_(devices, n_devices, &ops, &user);
    if (r != OVHIP_OK) return r;
    r = ovhip_dpb_create_ex(out, n_devices, &ops);
    if (r != OVHIP_OK) { ovhip_dpb_hip_ops_free_(user); return r; }
    (*out)->hip_user = user;
    for (int i = 0; i < n_devices; ++i) (*out)->hipdev[i] = devices[i];
    return OVHIP_OK;
}

void
ovhip_dpb_destroy(ovhip_dpb *d)
{
    if (!d) return;
    pthread_mutex_lock(&d->mtx);
    for (size_t i = 0; i < d->n_slots; ++i)
        if (d->slots[i].state != S_FREE) { d->slots[i].pins = 0; reclaim(d, &d->slots[i]); }
    unlock_and_finish(d);
    pthread_mutex_lock(&d->mtx);
    for (int k = 0; k < d->n_dev; ++k) {
        for (size_t i = 0; i < d->n_pool[k]; ++i) d->ops.pic_free(d->ops.user, k, &d->pool[k][i].pic);
        free(d->pool[k]);
    }
    pthread_mutex_unlock(&d->mtx);
    free(d->slots); free(d->limbo);
    if (d->hip_user) ovhip_dpb_hip_ops_free_(d->hip_user);
    pthread_cond_destroy(&d->cnd);
    pthread_mutex_destroy(&d->mtx);
    free(d);
}

int ovhip_dpb_n_devices(const ovhip_dpb *d) { return d ? d->n_dev : 0; }
int ovhip_dpb_device(const ovhip_dpb *d, int dev) { return d && dev >= 0 && dev < d->n_dev ? d->hipdev[dev] : -1; }

int ovhip_dpb_begin(ovhip_dpb *d, const void *key, int dev, int32_t w, int32_t h, ovhip_pic *pic) { return ovhip_dpb_begin_tag(d, key, 0, dev, w, h, pic); }

int
ovhip_dpb_begin_tag(ovhip_dpb *d, const void *key, uint64_t tag, int dev, int32_t w, int32_t h, ovhip_pic *pic)
{
    if (!d || !key || !pic || dev < 0 || dev >= d->n_dev || w <= 0 || h <= 0) return OVHIP_EINVAL;
    int r = OVHIP_OK;
    pthread_mutex_lock(&d->mtx);
    struct dpb_slot *s = find(d, key);
    if (s) {
        /* the key comes back for a new picture: the host DPB re-uses a frame only once it is unreferenced */
        if (s->pins) { pthread_mutex_unlock(&d->mtx); return OVHIP_EINVAL; }
        reclaim(d, s);
        if (d->n_limbo) {
            /* its buffers need a wait or a clear first: outside the mutex, then they are in the pools for the pop below */
            unlock_and_finish(d);
            pthread_mutex_lock(&d->mtx);
            s = find(d, key);                            /* (nobody else begins this key; a racing begin is the caller's error) */
            if (s) { pthread_mutex_unlock(&d->mtx); return OVHIP_EINVAL; }
        }
    }
    if (!s || s->state != S_FREE) {
        s = NULL;
        for (size_t i = 0; i < d->n_slots && !s; ++i) if (d->slots[i].state == S_FREE) s = &d->slots[i];
        if (!s) {
            const size_t nc = d->n_slots ? 2 * d->n_slots : 32;
            struct dpb_slot *q = (struct dpb_slot *)realloc(d->slots, nc * sizeof(*q));
            if (!q) { pthread_mutex_unlock(&d->mtx); return OVHIP_ENOMEM; }
            memset(q + d->n_slots, 0, (nc - d->n_slots) * sizeof(*q));
            d->slots = q; s = &q[d->n_slots]; d->n_slots = nc;
        }
    }
    memset(s, 0, sizeof(*s));
    r = pool_pop(d, dev, w, h, &s->pic);
    if (r == OVHIP_OK) {
        s->key = key; s->tag = tag; s->state = S_DECODING; s->home = dev; s->w = w; s->h = h; s->serial = ++d->serial;
        /* devices that asked for this picture before it existed */
        for (int i = 0; i < d->n_pend;) {
            if (d->pend[i].key == key && (!tag || !d->pend[i].tag || d->pend[i].tag == tag)) {
                s->want |= d->pend[i].devs & ~(1u << dev);
                d->pend[i] = d->pend[--d->n_pend];
            } else ++i;
        }
        *pic = s->pic;
        d->st.n_live++; d->st.n_begin++;
        pthread_cond_broadcast(&d->cnd);                 /* a reader may already be waiting for this key to appear */
    }
    pthread_mutex_unlock(&d->mtx);
    return r;
}

int ovhip_dpb_want(ovhip_dpb *d, const void *key, int dev) { return ovhip_dpb_want_tag(d, key, 0, dev); }

int
ovhip_dpb_want_tag(ovhip_dpb *d, const void *key, uint64_t tag, int dev)
{
    if (!d || !key || dev < 0 || dev >= d->n_dev) return OVHIP_EINVAL;
    int r = OVHIP_OK;
    pthread_mutex_lock(&d->mtx);
    struct dpb_slot *s = find_tag(d, key, tag);
    if (!s) {
        /* not begun yet (its frame thread is behind): remembered, ovhip_dpb_begin picks it up (the request was lost) */
        int i = 0;
        while (i < d->n_pend && !(d->pend[i].key == key && d->pend[i].tag == tag)) ++i;
        if (i == d->n_pend && d->n_pend == 64) {
            /* full of wants for pictures that were never begun (a decoder that skipped them): the oldest goes (the table
             * filled up for good and every later early want was lost) */
            memmove(&d->pend[0], &d->pend[1], 63 * sizeof(d->pend[0]));
            i = --d->n_pend;
        }
        if (i == d->n_pend) { d->pend[i].key = key; d->pend[i].tag = tag; d->pend[i].devs = 0; d->n_pend++; }
        d->pend[i].devs |= 1u << dev;
    }
    else if (dev != s->home) {
        s->want |= 1u << dev;
        if (s->state == S_DONE) r = start_copy(d, s, dev);
    }
    pthread_mutex_unlock(&d->mtx);
    return r;
}

int
ovhip_dpb_publish(ovhip_dpb *d, const void *key, int status)
{
    if (!d || !key) return OVHIP_EINVAL;
    int r = OVHIP_OK;
    pthread_mutex_lock(&d->mtx);
    struct dpb_slot *s = find(d, key);
    if (!s || s->state != S_DECODING) r = OVHIP_EINVAL;
    else {
        s->state = status ? S_FAILED : S_DONE;
        s->status = status;
        if (status) d->st.n_failed++;
        /* push: exactly the devices whose queued pictures list this one */
        for (int k = 0; k < d->n_dev && !status; ++k)
            if ((s->want >> k) & 1) { int q = start_copy(d, s, k); if (q != OVHIP_OK && r == OVHIP_OK) r = q; }
        if (s->released && !s->pins) reclaim(d, s);
    }
    pthread_cond_broadcast(&d->cnd);
    unlock_and_finish(d);
    return r;
}

int ovhip_dpb_acquire(ovhip_dpb *d, const void *key, int dev, ovhip_pic *pic, void **event) { return ovhip_dpb_acquire_tag(d, key, 0, dev, pic, event); }

int
ovhip_dpb_acquire_tag(ovhip_dpb *d, const void *key, uint64_t tag, int dev, ovhip_pic *pic, void **event)
{
    if (!d || !key || !pic || dev < 0 || dev >= d->n_dev) return OVHIP_EINVAL;
    if (event) *event = NULL;
    int r = OVHIP_OK;
    pthread_mutex_lock(&d->mtx);
    struct dpb_slot *s = find_tag(d, key, tag);
    int waited = 0;
    if (!s && d->unknown_ms > 0) {
        /* Frame threads start in decoding order but run on their own: a reader can get here before the thread that decodes its
         * reference picture has begun it.  Wait for the key to appear -- bounded: a key that never appears is an error */
        struct timespec until;
        clock_gettime(CLOCK_REALTIME, &until);
        until.tv_sec += d->unknown_ms / 1000; until.tv_nsec += (long)(d->unknown_ms % 1000) * 1000000L;
        if (until.tv_nsec >= 1000000000L) { until.tv_sec++; until.tv_nsec -= 1000000000L; }
        while (!s && !d->shutdown) {
            waited = 1;
            if (pthread_cond_timedwait(&d->cnd, &d->mtx, &until) == ETIMEDOUT) { s = find_tag(d, key, tag); break; }
            s = find_tag(d, key, tag);
        }
    }
    if (!s) { d->st.n_waits += waited; pthread_mutex_unlock(&d->mtx); return d->shutdown ? OVHIP_EREF : OVHIP_EINVAL; }
    const uint64_t serial = s->serial;
    while (s && s->serial == serial && s->state == S_DECODING && !d->shutdown) {
        waited = 1;
        pthread_cond_wait(&d->cnd, &d->mtx);
        s = find(d, key);                                   /* the slot array may have moved */
    }
    d->st.n_waits += waited;
    if (!s || s->serial != serial || d->shutdown || s->state != S_DONE) r = OVHIP_EREF;
    else if (dev == s->home) { *pic = s->pic; s->pins++; }
    else {
        r = start_copy(d, s, dev);
        if (r == OVHIP_OK) {
            if (!event && s->copy[dev].event) {
                /* a caller without a place for the handle waits right here */
                void *ev = s->copy[dev].event;
                s->pins++;
                pthread_mutex_unlock(&d->mtx);
                r = d->ops.copy_wait(d->ops.user, dev, ev);
                pthread_mutex_lock(&d->mtx);
                s = find(d, key);
                if (s && s->serial == serial) { if (r != OVHIP_OK) s->pins--; else *pic = s->copy[dev].pic; }
                pthread_mutex_unlock(&d->mtx);
                return r;
            }
            *pic = s->copy[dev].pic;
            if (event) *event = s->copy[dev].event;
            s->pins++;
        }
    }
    pthread_mutex_unlock(&d->mtx);
    return r;
}

/* ---- row progress ---- */
int
ovhip_dpb_post_rows(ovhip_dpb *d, const void *key, int32_t rows, void *event, const volatile uint32_t *abort_word)
{
    if (!d || !key || rows < 0) return OVHIP_EINVAL;
    int r = OVHIP_OK;
    pthread_mutex_lock(&d->mtx);
    struct dpb_slot *s = find(d, key);
    if (!s || s->state != S_DECODING) r = OVHIP_EINVAL;
    else {
        const int32_t last = s->n_prog ? s->prog[s->n_prog - 1].rows : s->rows_final;
        if (rows < last) r = OVHIP_EINVAL;
        else if (rows > last || (s->n_prog && event)) {
            s->abort_word = abort_word;
            /* the table is full (readers have not looked for a while), or the same rows behind a later event: the newest record takes the
             * place of the one before it -- an event recorded later on the producer's stream completes after the earlier one */
            if (s->n_prog == PROG_MAX || (s->n_prog && rows == last)) s->n_prog--;
            s->prog[s->n_prog].rows = rows; s->prog[s->n_prog].event = event;
            s->n_prog++;
            if (!d->ops.event_query && !event) { s->rows_final = rows; s->n_prog = 0; }
            pthread_cond_broadcast(&d->cnd);
        }
    }
    pthread_mutex_unlock(&d->mtx);
    return r;
}

/* mutex held: take the records whose events have completed (in order) */
static int
advance_rows(ovhip_dpb *d, struct dpb_slot *s)
{
    while (s->n_prog) {
        int done = 1;
        if (s->prog[0].event && d->ops.event_query) done = d->ops.event_query(d->ops.user, s->home, s->prog[0].event);
        if (done < 0) return done;
        if (!done) break;
        if (s->abort_word && *s->abort_word) break;          /* the producer's ordered pass gave up: it will fail the picture */
        s->rows_final = s->prog[0].rows;
        memmove(&s->prog[0], &s->prog[1], (size_t)(--s->n_prog) * sizeof(s->prog[0]));
    }
    return OVHIP_OK;
}

int
ovhip_dpb_rows_tag(ovhip_dpb *d, const void *key, uint64_t tag, int dev, int32_t need_rows, int block, int pin, ovhip_pic *pic, void **event)
{
    if (!d || !key || !pic || dev < 0 || dev >= d->n_dev) return OVHIP_EINVAL;
    if (event) *event = NULL;
    int r = 0, waited = 0;
    pthread_mutex_lock(&d->mtx);
    struct timespec until;
    int have_until = 0;
    for (;;) {
        if (d->shutdown) { r = OVHIP_EREF; break; }
        struct dpb_slot *s = find_tag(d, key, tag);
        if (!s) {
            /* not begun yet: its frame thread is behind (bounded, as in ovhip_dpb_acquire_tag) */
            if (!block) break;
            if (d->unknown_ms <= 0) { r = OVHIP_EINVAL; break; }
            if (!have_until) {
                clock_gettime(CLOCK_REALTIME, &until);
                until.tv_sec += d->unknown_ms / 1000; until.tv_nsec += (long)(d->unknown_ms % 1000) * 1000000L;
                if (until.tv_nsec >= 1000000000L) { until.tv_sec++; until.tv_nsec -= 1000000000L; }
                have_until = 1;
            }
            waited = 1;
            if (pthread_cond_timedwait(&d->cnd, &d->mtx, &until) == ETIMEDOUT && !find_tag(d, key, tag)) { r = OVHIP_EINVAL; break; }
            continue;
        }
        if (s->state == S_FAILED) { r = OVHIP_EREF; break; }
        if (s->state == S_DONE) {
            if (dev == s->home) { *pic = s->pic; if (pin) s->pins++; r = 1; break; }
            r = start_copy(d, s, dev);
            if (r == OVHIP_OK) { *pic = s->copy[dev].pic; if (event) *event = s->copy[dev].event; if (pin) s->pins++; r = 1; }
            break;
        }
        /* DECODING */
        if (dev == s->home && need_rows < s->h) {
            const int q = advance_rows(d, s);
            if (q < 0) { r = q; break; }
            if (s->rows_final >= need_rows) { *pic = s->pic; if (pin) s->pins++; r = 1; break; }
        }
        if (!block) break;
        waited = 1;
        /* a record that covers the rows: wait for its event outside the mutex; else for the producer's next record / its publication */
        void *ev = NULL;
        if (dev == s->home && need_rows < s->h && d->ops.event_wait && !(s->abort_word && *s->abort_word))
            for (int i = 0; i < s->n_prog && !ev; ++i) if (s->prog[i].rows >= need_rows) ev = s->prog[i].event;
        if (ev) {
            const int home = s->home;
            pthread_mutex_unlock(&d->mtx);
            const int q = d->ops.event_wait(d->ops.user, home, ev);
            pthread_mutex_lock(&d->mtx);
            if (q != OVHIP_OK) { r = q; break; }
        } else pthread_cond_wait(&d->cnd, &d->mtx);
    }
    d->st.n_waits += waited;
    pthread_mutex_unlock(&d->mtx);
    return r;
}

/* Is the picture there?  1: DONE (ovhip_dpb_acquire_tag would not block on the decode), 0: not yet -- unknown key, another picture under
 * the key, still DECODING -- OVHIP_EREF: it FAILED / the DPB was shut down.  Never blocks. */
int
ovhip_dpb_poll_tag(ovhip_dpb *d, const void *key, uint64_t tag)
{
    if (!d || !key) return OVHIP_EINVAL;
    pthread_mutex_lock(&d->mtx);
    const struct dpb_slot *s = find_tag(d, key, tag);
    const int r = d->shutdown ? OVHIP_EREF : !s ? 0 : s->state == S_DONE ? 1 : s->state == S_FAILED ? OVHIP_EREF : 0;
    pthread_mutex_unlock(&d->mtx);
    return r;
}

int
ovhip_dpb_wait_copy(ovhip_dpb *d, int dev, void *event)
{
    if (!d || dev < 0 || dev >= d->n_dev) return OVHIP_EINVAL;
    return event ? d->ops.copy_wait(d->ops.user, dev, event) : OVHIP_OK;
}

int
ovhip_dpb_unpin(ovhip_dpb *d, const void *key)
{
    if (!d || !key) return OVHIP_EINVAL;
    int r = OVHIP_OK;
    pthread_mutex_lock(&d->mtx);
    struct dpb_slot *s = find(d, key);
    if (!s || s->pins <= 0) r = OVHIP_EINVAL;
    else if (--s->pins == 0 && s->released && s->state != S_DECODING) reclaim(d, s);
    unlock_and_finish(d);
    return r;
}

int
ovhip_dpb_release(ovhip_dpb *d, const void *key)
{
    if (!d || !key) return OVHIP_EINVAL;
    int r = OVHIP_OK;
    pthread_mutex_lock(&d->mtx);
    struct dpb_slot *s = find(d, key);
    if (!s) r = OVHIP_EINVAL;
    else {
        s->released = 1;
        /* a picture still being decoded is reclaimed by its publish, a pinned one by its last unpin */
        if (!s->pins && s->state != S_DECODING) reclaim(d, s);
    }
    unlock_and_finish(d);
    return r;
}

int
ovhip_dpb_lookup(ovhip_dpb *d, const void *key, int *home_dev, ovhip_pic *pic)
{
    if (!d || !key || !pic) return OVHIP_EINVAL;
    int r = OVHIP_OK;
    pthread_mutex_lock(&d->mtx);
    struct dpb_slot *s = find(d, key);
    if (!s) r = OVHIP_EINVAL;
    else if (s->state != S_DONE) r = OVHIP_EREF;
    else { *pic = s->pic; if (home_dev) *home_dev = s->home; }
    pthread_mutex_unlock(&d->mtx);
    return r;
}

void
ovhip_dpb_set_unknown_key_timeout(ovhip_dpb *d, int ms)
{
    if (!d) return;
    pthread_mutex_lock(&d->mtx);
    d->unknown_ms = ms < 0 ? 0 : ms;
    pthread_mutex_unlock(&d->mtx);
}

void
ovhip_dpb_shutdown(ovhip_dpb *d)
{
    if (!d) return;
    pthread_mutex_lock(&d->mtx);
    d->shutdown = 1;
    pthread_cond_broadcast(&d->cnd);
    pthread_mutex_unlock(&d->mtx);
}

/* the stream driver re-arms a DPB it shut down after a failed run */
void
ovhip_dpb_rearm_(ovhip_dpb *d)
{
    pthread_mutex_lock(&d->mtx);
    d->shutdown = 0;
    pthread_mutex_unlock(&d->mtx);
}

int
ovhip_dpb_get_stats(ovhip_dpb *d, ovhip_dpb_stats *out)
{
    if (!d || !out) return OVHIP_EINVAL;
    pthread_mutex_lock(&d->mtx);
    *out = d->st;
    pthread_mutex_unlock(&d->mtx);
    return OVHIP_OK;
}
