/* ovvc_record_intra.c -- host recorder, ordered tasks (plain C, no GPU needed).
 *
 * The reference reconstructs in decoding order, so an intra block simply finds its neighbours reconstructed
 * (rcn_intra_tu -> intra_pred -> rcn_tu_st, rcn_transform_tree.c:1384-1451).  The device path runs everything that does
 * not read the current picture first, stage-parallel; what does -- intra prediction, CIIP's planar part, chroma-scale
 * regions next to such blocks and the chroma residuals scaled by them -- is recorded here as ovhip_itask with a LEVEL:
 * 1 + the highest level among the tasks that produce the samples it reads.  Level maps on the 4x4-luma grid (one for luma,
 * one for the chroma pair) hold the level of the task that last wrote each unit (0: written by the unordered launches).
 * Availability (which neighbours exist at all) is the caller's information (progress bit-fields, rcn_fill_ref.h:40-64);
 * only available neighbours create dependencies.
 */
#include <stdlib.h>
#include <string.h>
#include "ovvc_hip.h"
#include "ovvc_record_priv.h"

static int
maps_ready(ovhip_recorder *r)
{
    const int w4 = (r->pic_w + 3) >> 2, h4 = (r->pic_h + 3) >> 2;
    if (!r->lvl_y) {
        r->lvl_y = (uint16_t *)calloc((size_t)w4 * h4, 2);
        r->lvl_c = (uint16_t *)calloc((size_t)w4 * h4, 2);
        if (!r->lvl_y || !r->lvl_c) return -1;
        r->lvl_w4 = w4; r->lvl_h4 = h4; r->lvl_dirty = 0;
    }
    if (r->lvl_dirty) {
        memset(r->lvl_y, 0, (size_t)w4 * h4 * 2); memset(r->lvl_c, 0, (size_t)w4 * h4 * 2);
        r->lvl_dirty = 0;
    }
    return 0;
}

int
ovhip_rec_intra_reset_(ovhip_recorder *r)
{
    r->n_itask = 0; r->n_ilevels = 0; r->max_ilevel = 0;
    if (r->lvl_y) r->lvl_dirty = 1;          /* cleared lazily: pictures without ordered tasks never touch the maps */
    return 0;
}

void
ovhip_rec_intra_free_(ovhip_recorder *r)
{
    free(r->lvl_y); free(r->lvl_c); free(r->reg_level);
    r->lvl_y = r->lvl_c = NULL; r->reg_level = NULL;
}

/* highest level over the units [ux0, ux0 + nx) x [uy0, uy0 + ny) of a map, clipped to the picture.  Units written by an
 * ordered task (level > 0) that lie in ANOTHER CTU than the scanning task's add that CTU to r->scan_deps (bit 0 left,
 * 1 above-left, 2 above, 3 above-right): what the one-launch pass (k_intra_ctu) must wait for. */
static int
max_level(ovhip_recorder *r, const uint16_t *map, int ux0, int uy0, int nx, int ny)
{
    int m = 0;
    int x1 = ux0 + nx, y1 = uy0 + ny;
    const int sh = r->log2_ctu - 2;
    if (ux0 < 0) ux0 = 0;
    if (uy0 < 0) uy0 = 0;
    if (x1 > r->lvl_w4) x1 = r->lvl_w4;
    if (y1 > r->lvl_h4) y1 = r->lvl_h4;
    for (int y = uy0; y < y1; ++y)
        for (int x = ux0; x < x1; ++x) {
            const int l = map[y * r->lvl_w4 + x];
            if (!l) continue;
            if (l > m) m = l;
            const int dx = (x >> sh) - r->scan_cx, dy = (y >> sh) - r->scan_cy;
            if (dx | dy) {
                if (dy == 0 && dx == -1) r->scan_deps |= 1;
                else if (dy == -1 && dx == -1) r->scan_deps |= 2;
                else if (dy == -1 && dx == 0) r->scan_deps |= 4;
                else if (dy == -1 && dx == 1) r->scan_deps |= 8;
                else r->scan_deps |= 0x10;           /* not a wavefront neighbour: the CTU pass cannot serve this picture */
            }
        }
    return m;
}

static void
scan_begin(ovhip_recorder *r, int luma_x, int luma_y)
{
    r->scan_cx = luma_x >> r->log2_ctu; r->scan_cy = luma_y >> r->log2_ctu; r->scan_deps = 0;
}

static void
set_level(ovhip_recorder *r, uint16_t *map, int ux0, int uy0, int nx, int ny, int level)
{
    int x1 = ux0 + nx, y1 = uy0 + ny;
    if (x1 > r->lvl_w4) x1 = r->lvl_w4;
    if (y1 > r->lvl_h4) y1 = r->lvl_h4;
    for (int y = uy0; y < y1; ++y)
        for (int x = ux0; x < x1; ++x) map[y * r->lvl_w4 + x] = (uint16_t)level;
}

uint16_t
ovhip_rec_region_level_(ovhip_recorder *r, int32_t x0, int32_t y0, int n_abv, int n_lft)
{
    r->region_deps = 0;
    if (!r->n_itask || maps_ready(r)) return 0;
    const int ux = x0 >> 2, uy = y0 >> 2;
    int m = 0, k;
    scan_begin(r, x0, y0);
    if (n_abv && (k = max_level(r, r->lvl_y, ux, uy - 1, n_abv, 1)) > m) m = k;
    if (n_lft && (k = max_level(r, r->lvl_y, ux - 1, uy, 1, n_lft)) > m) m = k;
    r->region_deps = r->scan_deps;
    return (uint16_t)(m ? m + 1 : 0);
}

/* Appends a task; its level = 1 + max(levels of the units it reads, extra_level).  Returns the task index or <0. */
int
ovhip_rec_itask_add_(ovhip_recorder *r, const ovhip_itask *in, uint16_t extra_level)
{
    if (maps_ready(r)) return OVHIP_ENOMEM;
    if (r->n_itask + 1 > r->cap_itask && ovhip_rec_grow_(r, (void **)&r->itask, &r->cap_itask, r->n_itask + 1, sizeof(ovhip_itask))) return OVHIP_ENOMEM;
    ovhip_itask t = *in;
    const int w = 1 << t.log2_w, h = 1 << t.log2_h;
    int m = extra_level, k;
    const int cs = t.kind == OVHIP_IT_CHROMA || t.kind == OVHIP_IT_RES_C;
    scan_begin(r, t.x << cs, t.y << cs);
    if (t.kind == OVHIP_IT_REGION) r->scan_deps = r->region_deps;
    if (t.kind == OVHIP_IT_LUMA) {
        const int ux = t.x >> 2, uy = t.y >> 2, nx = (w + 3) >> 2, ny = (h + 3) >> 2;
        /* ISP: the arms are the coding unit's -- above from the CU's left edge, left from the CU's top edge (the partitions of
         * the CU itself are chained by ovhip_rec_isp_cu through extra_level) */
        const int isp = !!(t.flags & OVHIP_IF_ISP);
        const int uxa = isp ? (t.x - t.isp_off_x) >> 2 : ux, uyl = isp ? (t.y - t.isp_off_y) >> 2 : uy;
        if ((t.flags & OVHIP_IF_CORNER) && (k = max_level(r, r->lvl_y, uxa - 1, uy - 1, 1, 1)) > m) m = k;
        if (t.avl_abv && (k = max_level(r, r->lvl_y, uxa, uy - 1, t.avl_abv, 1)) > m) m = k;
        if (t.avl_lft && (k = max_level(r, r->lvl_y, ux - 1, uyl, 1, t.avl_lft)) > m) m = k;
        if (t.ciip_wt && (k = max_level(r, r->lvl_y, ux, uy, nx, ny)) > m) m = k;     /* blends into what is there */
        if (m >= 65534) return OVHIP_EUNSUP;
        t.level = (uint16_t)(m + 1);
        set_level(r, r->lvl_y, ux, uy, nx, ny, t.level);
    } else if (t.kind == OVHIP_IT_CHROMA || t.kind == OVHIP_IT_RES_C) {
        /* chroma units are 2 chroma samples = the same 4x4-luma grid */
        const int ux = t.x >> 1, uy = t.y >> 1, nx = (w + 1) >> 1, ny = (h + 1) >> 1;
        if (t.kind == OVHIP_IT_CHROMA) {
            if (t.mode >= 67) {
                /* CCLM / MDLM: the co-located reconstructed luma block and the luma + chroma lines around it the parameter
                 * derivation may read (up to w + min(w, h) above, h + min(w, h) left; rcn_intra_cclm.c:660-880) */
                const int ext = w < h ? w : h;
                const int na = (w + ext + 1) >> 1, nl = (h + ext + 1) >> 1;
                if ((k = max_level(r, r->lvl_y, ux, uy, nx, ny)) > m) m = k;
                if (t.avl_abv) { if ((k = max_level(r, r->lvl_y, ux - 1, uy - 1, na + 1, 1)) > m) m = k; if ((k = max_level(r, r->lvl_c, ux, uy - 1, na, 1)) > m) m = k; }
                if (t.avl_lft) { if ((k = max_level(r, r->lvl_y, ux - 1, uy - 1, 1, nl + 1)) > m) m = k; if ((k = max_level(r, r->lvl_c, ux - 1, uy, 1, nl)) > m) m = k; }
            } else {
                if ((t.flags & OVHIP_IF_CORNER) && (k = max_level(r, r->lvl_c, ux - 1, uy - 1, 1, 1)) > m) m = k;
                if (t.avl_abv && (k = max_level(r, r->lvl_c, ux, uy - 1, t.avl_abv, 1)) > m) m = k;
                if (t.avl_lft && (k = max_level(r, r->lvl_c, ux - 1, uy, 1, t.avl_lft)) > m) m = k;
            }
        }
        if ((t.kind == OVHIP_IT_RES_C || t.ciip_wt) && (k = max_level(r, r->lvl_c, ux, uy, nx, ny)) > m) m = k;
        if (m >= 65534) return OVHIP_EUNSUP;
        t.level = (uint16_t)(m + 1);
        set_level(r, r->lvl_c, ux, uy, nx, ny, t.level);
    } else if (t.kind == OVHIP_IT_REGION) {
        if (m >= 65534) return OVHIP_EUNSUP;
        t.level = (uint16_t)(m + 1);          /* extra_level = what ovhip_rec_region_level_ found, minus one */
    } else {
        return OVHIP_EINVAL;
    }
    t.ctu_deps = (uint16_t)(0x8000u | ((unsigned)r->log2_ctu << 8) | (r->scan_deps & 0x1f));
    r->itask[r->n_itask] = t;
    if (t.level > r->max_ilevel) r->max_ilevel = t.level;
    return (int)r->n_itask++;
}

int
ovhip_rec_set_ctu_size(ovhip_recorder *r, int32_t log2_ctu_s)
{
    if (!r || log2_ctu_s < 5 || log2_ctu_s > 7) return OVHIP_EINVAL;
    if (r->log) ovhip_calllog_ctu_size_(r->log, log2_ctu_s);
    r->log2_ctu = log2_ctu_s;
    return OVHIP_OK;
}

uint32_t
ovhip_rec_itask_levels(const ovhip_recorder *r)
{
    if (!r) return 0;
    uint32_t m = r->max_ilevel;
    if (!m) for (size_t i = 0; i < r->n_itask; ++i) if (r->itask[i].level > m) m = r->itask[i].level;    /* appended raw */
    return m;
}

const ovhip_itask *
ovhip_rec_itasks(const ovhip_recorder *r, size_t *n)
{
    if (!r || !n) return NULL;
    *n = r->n_itask;
    return r->itask;
}

/* counting sort by level (stable: decoding order inside a level) */
const ovhip_itask *
ovhip_rec_itasks_sorted(ovhip_recorder *r, size_t *n, const uint32_t **level_start, uint32_t *n_levels)
{
    if (!r || !n || !level_start || !n_levels) return NULL;
    *n = r->n_itask; *n_levels = 0; *level_start = NULL;
    if (!r->n_itask) return r->itask;
    uint32_t maxl = 0;
    for (size_t i = 0; i < r->n_itask; ++i) if (r->itask[i].level > maxl) maxl = r->itask[i].level;
    if (ovhip_rec_grow_(r, (void **)&r->itask_sorted, &r->cap_isorted, r->n_itask, sizeof(ovhip_itask))) { *n = 0; return NULL; }
    {   /* level table: plain host memory */
        if (r->cap_ilevel < (size_t)(maxl + 2) * 2) {
            uint32_t *q = (uint32_t *)realloc(r->ilevel_start, (size_t)(maxl + 2) * 2 * sizeof(uint32_t));
            if (!q) { *n = 0; return NULL; }
            r->ilevel_start = q; r->cap_ilevel = (size_t)(maxl + 2) * 2;
        }
    }
    uint32_t *start = r->ilevel_start;
    memset(start, 0, (maxl + 2) * sizeof(uint32_t));
    for (size_t i = 0; i < r->n_itask; ++i) start[r->itask[i].level]++;         /* counts at [level], level >= 1 */
    uint32_t acc = 0;
    for (uint32_t l = 1; l <= maxl; ++l) { const uint32_t c = start[l]; start[l] = acc; acc += c; }
    start[maxl + 1] = acc;
    uint32_t *fill = start + maxl + 2;                                          /* second half: running positions */
    memcpy(fill, start, (maxl + 2) * sizeof(uint32_t));
    for (size_t i = 0; i < r->n_itask; ++i) r->itask_sorted[fill[r->itask[i].level]++] = r->itask[i];
    r->n_ilevels = maxl;
    *n_levels = maxl;
    *level_start = start + 1;                                                   /* entry l = first task of level l + 1 */
    return r->itask_sorted;
}

/* The tasks [first, first + n) sorted by level into `out` (the picture job's staging block of a band of CTU rows).  level_start
 * (caller's array of cap entries): entry l = first task of the band's l-th level value in use ... entry *n_levels = n; levels the
 * band does not hold take no entry.  Returns 0, or -1 when cap is too small (the caller falls back to one flow launch, which needs
 * no table). */
int
ovhip_rec_itasks_sorted_range_(const ovhip_recorder *r, size_t first, size_t n, ovhip_itask *out, uint32_t *level_start, size_t cap, uint32_t *n_levels)
{
    *n_levels = 0;
    if (!n) return 0;
    const ovhip_itask *t = r->itask + first;
    uint32_t minl = 0xffffffffu, maxl = 0;
    for (size_t i = 0; i < n; ++i) { if (t[i].level < minl) minl = t[i].level; if (t[i].level > maxl) maxl = t[i].level; }
    const size_t span = (size_t)(maxl - minl) + 1;
    /* counting sort over the band's level range (levels are uint16: at most 64 K counters, on the heap of the caller's table when it
     * fits, else a scratch allocation) */
    uint32_t *cnt = (uint32_t *)calloc(span + 1, sizeof(uint32_t));
    if (!cnt) return -1;
    for (size_t i = 0; i < n; ++i) cnt[t[i].level - minl]++;
    uint32_t acc = 0, nl = 0;
    int fits = 1;
    for (size_t l = 0; l < span; ++l) {
        const uint32_t c = cnt[l];
        cnt[l] = acc;
        if (c) { if (nl < cap) level_start[nl] = acc; else fits = 0; ++nl; }
        acc += c;
    }
    if (nl < cap) level_start[nl] = acc; else fits = 0;
    for (size_t i = 0; i < n; ++i) out[cnt[t[i].level - minl]++] = t[i];
    free(cnt);
    *n_levels = nl;
    return fits ? 0 : -1;
}

/* Grouped by CTU for the one-launch ordered pass (k_intra_ctu): counting sort of the level-sorted list by CTU index, so a
 * CTU's tasks stay in level order, decoding order inside a level. */
const ovhip_itask *
ovhip_rec_itasks_by_ctu(ovhip_recorder *r, int32_t log2_ctu_s, size_t *n, const ovhip_ictu **ctus, size_t *n_ctus)
{
    if (!r || !n || !ctus || !n_ctus || log2_ctu_s < 5 || log2_ctu_s > 7) return NULL;
    *n = 0; *ctus = NULL; *n_ctus = 0;
    if (!r->n_itask) return r->itask;
    const uint32_t *lv; uint32_t nlv; size_t nt;
    const ovhip_itask *sorted = ovhip_rec_itasks_sorted(r, &nt, &lv, &nlv);
    if (!sorted) return NULL;
    const int ncx = (r->pic_w + (1 << log2_ctu_s) - 1) >> log2_ctu_s, ncy = (r->pic_h + (1 << log2_ctu_s) - 1) >> log2_ctu_s;
    const size_t nctu = (size_t)ncx * ncy;
    if (r->cap_ctu_count < 2 * nctu + 1) {
        uint32_t *q = (uint32_t *)realloc(r->ctu_count, (2 * nctu + 1) * sizeof(uint32_t));
        if (!q) return NULL;
        r->ctu_count = q; r->cap_ctu_count = 2 * nctu + 1;
    }
    if (ovhip_rec_grow_(r, (void **)&r->itask_ctu, &r->cap_itask_ctu, nt, sizeof(ovhip_itask))) return NULL;
    uint32_t *start = r->ctu_count, *fill = r->ctu_count + nctu + 1;
    memset(start, 0, (nctu + 1) * sizeof(uint32_t));
#define CTU_OF(t) ((size_t)(((t).kind == OVHIP_IT_CHROMA || (t).kind == OVHIP_IT_RES_C ? (t).y * 2 : (t).y) >> log2_ctu_s) * ncx \
                   + (((t).kind == OVHIP_IT_CHROMA || (t).kind == OVHIP_IT_RES_C ? (t).x * 2 : (t).x) >> log2_ctu_s))
    size_t used = 0;
    int precise = 1;                 /* every task carries the CTUs it reads ordered samples from, for this CTU size */
    for (size_t i = 0; i < nt; ++i) {
        const size_t c = CTU_OF(sorted[i]);
        if (c >= nctu) return NULL;
        if (!start[c]++) ++used;
        const unsigned d = sorted[i].ctu_deps;
        if (!(d & 0x8000u) || ((d >> 8) & 7) != (unsigned)log2_ctu_s || (d & 0x10)) precise = 0;
    }
    if (ovhip_rec_grow_(r, (void **)&r->ictu, &r->cap_ictu, used, sizeof(ovhip_ictu))) return NULL;
    uint32_t acc = 0; size_t k = 0;
    for (size_t c = 0; c < nctu; ++c) {
        const uint32_t cnt = start[c];
        fill[c] = acc;
        if (cnt) {
            const int cx = (int)(c % ncx), cy = (int)(c / ncx);
            ovhip_ictu d = { (uint16_t)cx, (uint16_t)cy, acc, cnt, 0 };
            r->ictu[k++] = d;
        }
        acc += cnt;
    }
    /* neighbour masks from the raw counts (start[] still holds them) */
    for (size_t i = 0; i < used; ++i) {
        ovhip_ictu *d = &r->ictu[i];
        const int cx = d->cx, cy = d->cy;
        uint32_t m = 0;
        if (cx > 0 && start[(size_t)cy * ncx + cx - 1]) m |= 1;
        if (cx > 0 && cy > 0 && start[(size_t)(cy - 1) * ncx + cx - 1]) m |= 2;
        if (cy > 0 && start[(size_t)(cy - 1) * ncx + cx]) m |= 4;
        if (cy > 0 && cx + 1 < ncx && start[(size_t)(cy - 1) * ncx + cx + 1]) m |= 8;
        d->deps = precise ? 0 : m;
    }
    for (size_t i = 0; i < nt; ++i) r->itask_ctu[fill[CTU_OF(sorted[i])]++] = sorted[i];
    if (precise)
        for (size_t i = 0; i < used; ++i) {
            ovhip_ictu *d = &r->ictu[i];
            for (uint32_t k = 0; k < d->n; ++k) d->deps |= r->itask_ctu[d->first + k].ctu_deps & 0xf;
        }
#undef CTU_OF
    *n = nt; *ctus = r->ictu; *n_ctus = used;
    return r->itask_ctu;
}
