/* ovvc_calllog.c -- serialised recorder calls of one picture (include/ovvc_hip.h, "Call log").
 *
 * The reference hands a CU to reconstruction the moment it is parsed (vcl_coding_unit.c:711-840, vcl_transform_unit.c:1819-1963);
 * a call log is those hand-overs of one picture written down: the descriptors the shim's hooks fill from the OVCTUDec plus the
 * coefficient blocks the descriptors point to (ctudec->residual_y / _cb / _cr, sub-block-major, rcn_dequant.c:160-236).  Replaying
 * it does what a parse thread does per picture on the device path, minus CABAC and minus the OVCTUDec snapshot: every
 * ovhip_rec_* call, in order, on a recorder.  Record: [u32 type][u32 payload bytes][payload, padded to 8].
 */
#include <stdlib.h>
#include <string.h>
#include "ovvc_hip.h"
#include "ovvc_record_priv.h"

enum { CL_CTU_SIZE = 1, CL_TU, CL_ISP, CL_PU, CL_AFF, CL_REGION, CL_DBF, CL_CIIP };

struct ovhip_calllog { unsigned char *data; size_t n, cap; int oom; };   /* oom: the log is unusable (allocation failed or a call could not be written down): _data returns NULL */

struct cl_tu {
    ovhip_tu_state st;
    ovhip_tu_desc tu;             /* coef[] = int16 offsets into the payload's coefficient area + 1 (0: NULL) */
    ovhip_itask tl, tc;
    uint32_t has_l, has_c;
};
struct cl_isp { ovhip_tu_state st; ovhip_isp_desc cu; };
struct cl_region { int32_t x0, y0; uint32_t abv, lft; };
struct cl_ciip { int32_t x0, y0, log2_w, log2_h, mode_abv, mode_lft; };

ovhip_calllog *ovhip_calllog_create(void) { return (ovhip_calllog *)calloc(1, sizeof(ovhip_calllog)); }
void ovhip_calllog_destroy(ovhip_calllog *l) { if (l) { free(l->data); free(l); } }
void ovhip_calllog_reset(ovhip_calllog *l) { if (l) { l->n = 0; l->oom = 0; } }

const void *
ovhip_calllog_data(const ovhip_calllog *l, size_t *bytes)
{
    if (!l || !bytes || l->oom) return NULL;
    *bytes = l->n;
    return l->data;
}

void ovhip_rec_set_calllog(ovhip_recorder *r, ovhip_calllog *l) { if (r) r->log = l; }

/* room for one record; returns the payload address */
static unsigned char *
cl_open(ovhip_calllog *l, uint32_t type, size_t payload)
{
    const size_t pad = (payload + 7) & ~(size_t)7, need = l->n + 8 + pad;
    if (need > l->cap) {
        size_t nc = l->cap ? l->cap : (size_t)1 << 20;
        while (nc < need) nc *= 2;
        unsigned char *q = (unsigned char *)realloc(l->data, nc);
        if (!q) { l->oom = 1; return NULL; }
        l->data = q; l->cap = nc;
    }
    uint32_t hdr[2] = { type, (uint32_t)pad };
    memcpy(l->data + l->n, hdr, 8);
    unsigned char *p = l->data + l->n + 8;
    if (pad != payload) memset(p + payload, 0, pad - payload);
    l->n = need;
    return p;
}

/* int16s of component comp's coefficient block a TU call may read (capture_sbs / capture_raster, ovvc_record.c) */
static size_t
tu_coef_extent(const ovhip_tu_desc *tu, int comp)
{
    int l2w = tu->log2_tb_w, l2h = tu->log2_tb_h;
    if (comp != 2 && tu->tree != 2) { --l2w; --l2h; }
    if (l2w < 0 || l2h < 0) return 0;
    const int w = 1 << l2w, h = 1 << l2h;
    const int stride = w > 32 ? 32 : w, rows = h > 32 ? 32 : h;       /* what the parser's buffer holds of the block at most */
    if (l2w < 2 || l2h < 2 || tu->tr_skip_mask || (tu->cu_flags & ((1u << 8) | (1u << 9)))) return (size_t)stride * rows;      /* raster forms */
    uint64_t m = tu->sig_sb_map[comp] | 1;
    int top = 0;
    for (int sy = 0; sy < 8; ++sy) if ((m >> (8 * sy)) & 0xff) top = sy;
    int need_rows = 4 * (top + 1);
    return (size_t)stride * (need_rows < rows ? need_rows : rows);
}

void
ovhip_calllog_tu_(ovhip_calllog *l, const ovhip_tu_state *st, const ovhip_tu_desc *tu, const ovhip_itask *il, const ovhip_itask *ic)
{
    size_t ext[3], tot = 0;
    for (int c = 0; c < 3; ++c) { ext[c] = tu->coef[c] ? tu_coef_extent(tu, c) : 0; tot += (ext[c] + 3) & ~(size_t)3; }
    unsigned char *p = cl_open(l, CL_TU, sizeof(struct cl_tu) + 2 * tot);
    if (!p) return;
    struct cl_tu *q = (struct cl_tu *)p;
    memset(q, 0, sizeof(*q));
    q->st = *st; q->tu = *tu;
    if (il) { q->tl = *il; q->has_l = 1; }
    if (ic) { q->tc = *ic; q->has_c = 1; }
    int16_t *dst = (int16_t *)(p + sizeof(*q));
    size_t off = 0;
    for (int c = 0; c < 3; ++c) {
        q->tu.coef[c] = NULL;
        if (!ext[c]) continue;
        memcpy(dst + off, tu->coef[c], 2 * ext[c]);
        q->tu.coef[c] = (const int16_t *)(uintptr_t)(off + 1);
        off += (ext[c] + 3) & ~(size_t)3;
    }
}

void
ovhip_calllog_isp_(ovhip_calllog *l, const ovhip_tu_state *st, const ovhip_isp_desc *cu)
{
    const size_t n = cu->coef ? (size_t)1 << (cu->log2_cb_w + cu->log2_cb_h) : 0;
    unsigned char *p = cl_open(l, CL_ISP, sizeof(struct cl_isp) + 2 * n);
    if (!p) return;
    struct cl_isp *q = (struct cl_isp *)p;
    q->st = *st; q->cu = *cu; q->cu.coef = NULL;
    if (n) memcpy(p + sizeof(*q), cu->coef, 2 * n);
}

void
ovhip_calllog_pu_(ovhip_calllog *l, const ovhip_pu_desc *pu)
{
    unsigned char *p = cl_open(l, CL_PU, sizeof(*pu));
    if (p) memcpy(p, pu, sizeof(*pu));
}

void
ovhip_calllog_affine_(ovhip_calllog *l, const ovhip_affine_desc *cu)
{
    const int nx = (1 << cu->log2_w) >> 2, ny = (1 << cu->log2_h) >> 2;
    if (!cu->mv0 || !cu->mv1 || cu->log2_w < 2 || cu->log2_h < 2 || cu->log2_w > 7 || cu->log2_h > 7) { l->oom = 1; return; }   /* (ovhip_rec_affine_cu refuses these) */
    unsigned char *p = cl_open(l, CL_AFF, sizeof(*cu) + 2 * (size_t)nx * ny * 8);
    if (!p) return;
    ovhip_affine_desc *q = (ovhip_affine_desc *)p;
    *q = *cu; q->mv0 = q->mv1 = NULL; q->mv_stride = nx;
    int32_t *m0 = (int32_t *)(p + sizeof(*cu)), *m1 = m0 + 2 * nx * ny;
    for (int y = 0; y < ny; ++y) {
        memcpy(m0 + 2 * y * nx, cu->mv0 + 2 * y * cu->mv_stride, 8 * (size_t)nx);
        memcpy(m1 + 2 * y * nx, cu->mv1 + 2 * y * cu->mv_stride, 8 * (size_t)nx);
    }
}

void
ovhip_calllog_region_(ovhip_calllog *l, int32_t x0, int32_t y0, uint32_t abv, uint32_t lft)
{
    struct cl_region g = { x0, y0, abv, lft };
    unsigned char *p = cl_open(l, CL_REGION, sizeof(g));
    if (p) memcpy(p, &g, sizeof(g));
}

void
ovhip_calllog_dbf_(ovhip_calllog *l, const ovhip_dbf_ctu *c)
{
    unsigned char *p = cl_open(l, CL_DBF, sizeof(*c));
    if (p) memcpy(p, c, sizeof(*c));
}

void
ovhip_calllog_ciip_(ovhip_calllog *l, int32_t x0, int32_t y0, int32_t log2_w, int32_t log2_h, int32_t mode_abv, int32_t mode_lft)
{
    struct cl_ciip g = { x0, y0, log2_w, log2_h, mode_abv, mode_lft };
    unsigned char *p = cl_open(l, CL_CIIP, sizeof(g));
    if (p) memcpy(p, &g, sizeof(g));
}

void
ovhip_calllog_ctu_size_(ovhip_calllog *l, int32_t log2_ctu_s)
{
    unsigned char *p = cl_open(l, CL_CTU_SIZE, sizeof(log2_ctu_s));
    if (p) memcpy(p, &log2_ctu_s, sizeof(log2_ctu_s));
}

int64_t
ovhip_calllog_replay(const void *data, size_t bytes, ovhip_recorder *rec)
{
    if (!rec || (bytes && !data) || (bytes & 7)) return OVHIP_EINVAL;
    const unsigned char *p = (const unsigned char *)data, *end = p + bytes;
    ovhip_calllog *attached = rec->log;
    rec->log = NULL;                       /* a replay is not recorded again */
    int64_t n = 0;
    int r = 0;
    while (p < end && r >= 0) {
        uint32_t hdr[2];
        if ((size_t)(end - p) < 8) { r = OVHIP_EINVAL; break; }
        memcpy(hdr, p, 8);
        const unsigned char *q = p + 8;
        if (hdr[1] > (size_t)(end - q) || (hdr[1] & 7)) { r = OVHIP_EINVAL; break; }
        /* A log is a byte buffer from a file or another process: every record's fixed part, and every extent derived from its
         * fields (coefficient blocks, sub-block vectors), must lie inside the payload before a recorder call reads through it. */
        const size_t len = hdr[1];
#define CL_NEED(bytes_) if (len < (size_t)(bytes_)) { r = OVHIP_EINVAL; break; }
        switch (hdr[0]) {
        case CL_CTU_SIZE: { int32_t v; CL_NEED(4) memcpy(&v, q, 4); r = ovhip_rec_set_ctu_size(rec, v); break; }
        case CL_TU: {
            CL_NEED(sizeof(struct cl_tu))
            const struct cl_tu *t = (const struct cl_tu *)q;
            ovhip_tu_desc d = t->tu;
            if (d.log2_tb_w > 7 || d.log2_tb_h > 7 || t->has_l > 1 || t->has_c > 1) { r = OVHIP_EINVAL; break; }
            const int16_t *coefs = (const int16_t *)(q + sizeof(*t));
            const size_t have = (len - sizeof(*t)) / 2;
            int ok = 1;
            for (int c = 0; c < 3; ++c) {
                const uintptr_t o = (uintptr_t)t->tu.coef[c];
                if (!o) { d.coef[c] = NULL; continue; }
                if (o - 1 > have || tu_coef_extent(&d, c) > have - (o - 1)) { ok = 0; break; }
                d.coef[c] = coefs + (o - 1);
            }
            if (!ok) { r = OVHIP_EINVAL; break; }
            r = ovhip_rec_tu_intra(rec, &t->st, &d, t->has_l ? &t->tl : NULL, t->has_c ? &t->tc : NULL);
            break;
        }
        case CL_ISP: {
            CL_NEED(sizeof(struct cl_isp))
            const struct cl_isp *t = (const struct cl_isp *)q;
            ovhip_isp_desc d = t->cu;
            if (d.log2_cb_w > 7 || d.log2_cb_h > 7) { r = OVHIP_EINVAL; break; }
            const size_t have = (len - sizeof(*t)) / 2, want = (size_t)1 << (d.log2_cb_w + d.log2_cb_h);
            if (have >= want) d.coef = (const int16_t *)(q + sizeof(*t));
            else if (have < 4) d.coef = NULL;                    /* written without coefficients (only the record's padding follows) */
            else { r = OVHIP_EINVAL; break; }
            r = ovhip_rec_isp_cu(rec, &t->st, &d);
            break;
        }
        case CL_PU: CL_NEED(sizeof(ovhip_pu_desc)) r = ovhip_rec_pu(rec, (const ovhip_pu_desc *)q); break;
        case CL_AFF: {
            CL_NEED(sizeof(ovhip_affine_desc))
            ovhip_affine_desc d = *(const ovhip_affine_desc *)q;
            if (d.log2_w < 2 || d.log2_h < 2 || d.log2_w > 7 || d.log2_h > 7) { r = OVHIP_EINVAL; break; }
            const int nx = (1 << d.log2_w) >> 2, ny = (1 << d.log2_h) >> 2;
            CL_NEED(sizeof(d) + 16 * (size_t)nx * ny)
            d.mv0 = (const int32_t *)(q + sizeof(d)); d.mv1 = d.mv0 + 2 * nx * ny; d.mv_stride = nx;
            r = ovhip_rec_affine_cu(rec, &d);
            break;
        }
        case CL_REGION: { CL_NEED(sizeof(struct cl_region)) const struct cl_region *g = (const struct cl_region *)q; r = ovhip_rec_lmcs_region(rec, g->x0, g->y0, g->abv, g->lft); break; }
        case CL_DBF: CL_NEED(sizeof(ovhip_dbf_ctu)) r = ovhip_rec_dbf_ctu(rec, (const ovhip_dbf_ctu *)q); break;
        case CL_CIIP: { CL_NEED(sizeof(struct cl_ciip)) const struct cl_ciip *g = (const struct cl_ciip *)q; r = ovhip_rec_ciip(rec, g->x0, g->y0, g->log2_w, g->log2_h, g->mode_abv, g->mode_lft); break; }
        default: r = OVHIP_EINVAL;
        }
#undef CL_NEED
        p = q + hdr[1];
        ++n;
    }
    rec->log = attached;
    return r < 0 ? r : n;
}
