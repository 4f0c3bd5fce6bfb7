/* ovvc_record_dbf.c -- host recorder, deblocking part (plain C, no GPU needed).
 *
 * Input: the CTU-local deblocking bit maps the reference hands to df.rcn_dbf_ctu()
 * (struct DBFInfo, libovvc/ctudec.h:130-170, after dbf_load_info()'s neighbour rotation,
 * drv_lines.c:618-761).  Output: one 16-bit parameter word per 4-sample edge segment in
 * picture-level planes (include/ovvc_hip.h).  This file restates ONLY the control flow of
 *   vvc_dbf_ctu_hor / vvc_dbf_ctu_ver            libovvc/rcn_df.c:1940-2167
 *   set_edge_context / derive_filter_length      libovvc/rcn_df.c:1890-1938
 *   vvc_dbf_chroma_hor / vvc_dbf_chroma_ver      libovvc/rcn_df.c:1151-1431
 *   derive_size_3_map / derive_large_map_from_ngh libovvc/rcn_df.c:190-207, :1087-1105
 * i.e. which segments are filtered, with which bS, average QP and maximum filter lengths.
 * The sample arithmetic (decisions + filters) runs on the device (kernels_dbf.hip).
 */
#include <stdlib.h>
#include <string.h>
#include "ovvc_hip.h"
#include "ovvc_record_priv.h"

static int
dbf_alloc(ovhip_recorder *r)
{
    const int w4 = (r->pic_w + 3) >> 2, h4 = (r->pic_h + 3) >> 2;
    const int w4c = (w4 + 1) >> 1, h4c = (h4 + 1) >> 1;
    r->dbf_w4 = w4; r->dbf_h4 = h4;
    if (r->dbf_luma_v || !r->dense_planes) return 0;
    r->dbf_luma_v = (uint16_t *)calloc((size_t)w4 * h4, 2);
    r->dbf_luma_h = (uint16_t *)calloc((size_t)w4 * h4, 2);
    r->dbf_cb_v = (uint16_t *)calloc((size_t)w4c * h4, 2);
    r->dbf_cr_v = (uint16_t *)calloc((size_t)w4c * h4, 2);
    r->dbf_cb_h = (uint16_t *)calloc((size_t)w4 * h4c, 2);
    r->dbf_cr_h = (uint16_t *)calloc((size_t)w4 * h4c, 2);
    if (!r->dbf_luma_v || !r->dbf_luma_h || !r->dbf_cb_v || !r->dbf_cr_v || !r->dbf_cb_h || !r->dbf_cr_h) return -1;
    return 0;
}

void
ovhip_rec_dbf_reset_(ovhip_recorder *r)
{
    if (!r->dbf_luma_v) return;
    const size_t w4 = r->dbf_w4, h4 = r->dbf_h4, w4c = (w4 + 1) >> 1, h4c = (h4 + 1) >> 1;
    memset(r->dbf_luma_v, 0, w4 * h4 * 2); memset(r->dbf_luma_h, 0, w4 * h4 * 2);
    memset(r->dbf_cb_v, 0, w4c * h4 * 2);  memset(r->dbf_cr_v, 0, w4c * h4 * 2);
    memset(r->dbf_cb_h, 0, w4 * h4c * 2);  memset(r->dbf_cr_h, 0, w4 * h4c * 2);
}

void
ovhip_rec_dbf_free_(ovhip_recorder *r)
{
    free(r->dbf_luma_v); free(r->dbf_luma_h); free(r->dbf_cb_v); free(r->dbf_cr_v); free(r->dbf_cb_h); free(r->dbf_cr_h);
    r->dbf_luma_v = r->dbf_luma_h = r->dbf_cb_v = r->dbf_cr_v = r->dbf_cb_h = r->dbf_cr_h = NULL;
}

int
ovhip_rec_dbf_planes(const ovhip_recorder *r, ovhip_dbf_planes *out)
{
    if (!r || !out || !r->dbf_luma_v) return OVHIP_EINVAL;
    if (r->n_dbf_off > 1) return OVHIP_EUNSUP;   /* the dense planes carry one (beta, tc) pair: use the edge lists */
    out->luma_v = r->dbf_luma_v; out->luma_h = r->dbf_luma_h;
    out->cb_v = r->dbf_cb_v; out->cr_v = r->dbf_cr_v; out->cb_h = r->dbf_cb_h; out->cr_h = r->dbf_cr_h;
    out->w4 = r->dbf_w4; out->h4 = r->dbf_h4;
    out->beta_offset = r->dbf_beta_offset; out->tc_offset = r->dbf_tc_offset;
    return OVHIP_OK;
}

/* ~(OR of 7 consecutive map words): no edge within the next 7 units <=> block >= 32 samples */
static uint64_t size_3_map(const uint64_t *m) { return ~(m[0] | m[1] | m[2] | m[3] | m[4] | m[5] | m[6]); }

struct edge_ctx { uint64_t large_p, large_q, small, aff1; };

static void
edge_context(struct edge_ctx *e, const uint64_t *edg, const uint64_t *sb, int i, int p_allowed)
{
    e->large_p = (i % 4 || !p_allowed) ? 0 : size_3_map(&edg[i - 7]);
    e->large_q = (i % 4) ? 0 : size_3_map(&edg[i + 1]);
    e->small = edg[i - 1] | edg[i + 1] | sb[i - 1] | sb[i + 1];
    e->aff1 = sb[i] & (edg[i - 2] | edg[i + 2]) & ~edg[i];
    e->large_p &= ~(sb[i] & ~edg[i]);
    e->large_q &= ~(sb[i] & ~edg[i]);
}

static void
filter_length(const struct edge_ctx *e, uint64_t aff_p, uint64_t aff_q, uint64_t pos, int *lp, int *lq)
{
    if (e->small & pos) { *lp = *lq = 1; return; }
    if (e->aff1 & pos)  { *lp = *lq = 2; return; }
    *lp = *lq = 3;
    if (e->large_p & pos) *lp = (aff_p & pos) ? 5 : 7;
    if (e->large_q & pos) *lq = (aff_q & pos) ? 5 : 7;
}

/* in-place transpose of a 32 x 32 bit matrix (row r = word r, column c = bit c): five rounds of masked block swaps */
static void
transpose32(uint32_t a[32])
{
    uint32_t m = 0x0000ffffu;
    for (int j = 16; j; j >>= 1, m ^= m << j)
        for (int k = 0; k < 32; k = (k + j + 1) & ~j) {
            const uint32_t t = ((a[k] >> j) ^ a[k + j]) & m;
            a[k] ^= t << j; a[k + j] ^= t;
        }
}

static uint64_t large_from_ngh(const uint64_t *m) { return ~(m[-1] | m[1] | m[-2] | m[2] | m[-3] | m[3]); }

/* One segment: into the compact list the device kernel consumes (never an edge ON the picture boundary, the rule
 * ovhip_dbf_compact applies too) and, when kept, into the dense plane.  The lists have room for a whole CTU's segments when
 * this runs (ovhip_rec_dbf_ctu reserves it once): no capacity test per segment. */
static inline void
emit_edge(ovhip_recorder *r, int dir, int comp, int ux, int uy, uint16_t word, int off_idx, uint16_t *plane_word)
{
    if (plane_word) *plane_word = word;
    if ((dir ? uy : ux) == 0) return;
    const ovhip_dbf_edge e = { (uint16_t)ux, (uint16_t)uy, word, (uint8_t)comp, (uint8_t)off_idx };
    if (dir) r->edge_h[r->n_edge_h++] = e; else r->edge_v[r->n_edge_v++] = e;
}

/* The (beta, tc) offsets are slice-level state (slicedec.c:1416-1417): every distinct pair of the picture gets an index
 * that travels in the edge record. */
static int
offset_index(ovhip_recorder *r, int beta, int tc)
{
    for (int i = 0; i < r->n_dbf_off; ++i)
        if (r->dbf_off.beta[i] == beta && r->dbf_off.tc[i] == tc) return i;
    if (r->n_dbf_off >= OVHIP_DBF_MAX_OFFSETS || beta < -128 || beta > 127 || tc < -128 || tc > 127) return -1;
    r->dbf_off.beta[r->n_dbf_off] = (int8_t)beta; r->dbf_off.tc[r->n_dbf_off] = (int8_t)tc;
    return r->n_dbf_off++;
}

/* the maps of one CTU by pointer: what ovhip_rec_dbf_row takes, and what ovhip_rec_dbf_ctu makes of its by-value descriptor */
static void
view_of(ovhip_dbf_view *v, const ovhip_dbf_ctu *c)
{
    v->ctb_bound_ver = c->ctb_bound_ver; v->ctb_bound_hor = c->ctb_bound_hor; v->ctb_bound_ver_c = c->ctb_bound_ver_c; v->ctb_bound_hor_c = c->ctb_bound_hor_c;
    v->aff_edg_ver = c->aff_edg_ver; v->aff_edg_hor = c->aff_edg_hor;
    v->bs2_ver = c->bs2_ver; v->bs2_hor = c->bs2_hor; v->bs2c_ver = c->bs2c_ver; v->bs2c_hor = c->bs2c_hor;
    v->bs1_ver = c->bs1_ver; v->bs1_hor = c->bs1_hor; v->bs1cb_ver = c->bs1cb_ver; v->bs1cb_hor = c->bs1cb_hor; v->bs1cr_ver = c->bs1cr_ver; v->bs1cr_hor = c->bs1cr_hor;
    v->affine_ver = c->affine_ver; v->affine_hor = c->affine_hor;
    v->qp_y = c->qp_y; v->qp_cb = c->qp_cb; v->qp_cr = c->qp_cr;
    v->beta_offset = c->beta_offset; v->tc_offset = c->tc_offset; v->disable_v = c->disable_v; v->disable_h = c->disable_h;
    v->log2_ctu_s = c->log2_ctu_s; v->last_x = c->last_x; v->last_y = c->last_y; v->ctu_lft = c->ctu_lft; v->ctu_abv = c->ctu_abv; v->pad = 0;
    v->ctu_w = c->ctu_w; v->ctu_h = c->ctu_h; v->ctb_x = c->ctb_x; v->ctb_y = c->ctb_y;
}

/* (the call log stores descriptors by value) */
static void
ctu_of(ovhip_dbf_ctu *c, const ovhip_dbf_view *v)
{
    memset(c, 0, sizeof(*c));
    memcpy(c->ctb_bound_ver, v->ctb_bound_ver, sizeof(c->ctb_bound_ver)); memcpy(c->ctb_bound_hor, v->ctb_bound_hor, sizeof(c->ctb_bound_hor));
    memcpy(c->ctb_bound_ver_c, v->ctb_bound_ver_c, sizeof(c->ctb_bound_ver_c)); memcpy(c->ctb_bound_hor_c, v->ctb_bound_hor_c, sizeof(c->ctb_bound_hor_c));
    memcpy(c->aff_edg_ver, v->aff_edg_ver, sizeof(c->aff_edg_ver)); memcpy(c->aff_edg_hor, v->aff_edg_hor, sizeof(c->aff_edg_hor));
    memcpy(c->bs2_ver, v->bs2_ver, sizeof(c->bs2_ver)); memcpy(c->bs2_hor, v->bs2_hor, sizeof(c->bs2_hor));
    memcpy(c->bs2c_ver, v->bs2c_ver, sizeof(c->bs2c_ver)); memcpy(c->bs2c_hor, v->bs2c_hor, sizeof(c->bs2c_hor));
    memcpy(c->bs1_ver, v->bs1_ver, sizeof(c->bs1_ver)); memcpy(c->bs1_hor, v->bs1_hor, sizeof(c->bs1_hor));
    memcpy(c->bs1cb_ver, v->bs1cb_ver, sizeof(c->bs1cb_ver)); memcpy(c->bs1cb_hor, v->bs1cb_hor, sizeof(c->bs1cb_hor));
    memcpy(c->bs1cr_ver, v->bs1cr_ver, sizeof(c->bs1cr_ver)); memcpy(c->bs1cr_hor, v->bs1cr_hor, sizeof(c->bs1cr_hor));
    memcpy(c->affine_ver, v->affine_ver, sizeof(c->affine_ver)); memcpy(c->affine_hor, v->affine_hor, sizeof(c->affine_hor));
    memcpy(c->qp_y, v->qp_y, sizeof(c->qp_y)); memcpy(c->qp_cb, v->qp_cb, sizeof(c->qp_cb)); memcpy(c->qp_cr, v->qp_cr, sizeof(c->qp_cr));
    c->beta_offset = v->beta_offset; c->tc_offset = v->tc_offset; c->disable_v = v->disable_v; c->disable_h = v->disable_h;
    c->log2_ctu_s = v->log2_ctu_s; c->last_x = v->last_x; c->last_y = v->last_y; c->ctu_lft = v->ctu_lft; c->ctu_abv = v->ctu_abv;
    c->ctu_w = v->ctu_w; c->ctu_h = v->ctu_h; c->ctb_x = v->ctb_x; c->ctb_y = v->ctb_y;
}

static int rec_dbf_view(ovhip_recorder *r, const ovhip_dbf_view *c);

int
ovhip_rec_dbf_ctu(ovhip_recorder *r, const ovhip_dbf_ctu *c)
{
    if (!r || !c) return OVHIP_EINVAL;
    if (c->log2_ctu_s < 5 || c->log2_ctu_s > 7) return OVHIP_EINVAL;
    if (r->log) ovhip_calllog_dbf_(r->log, c);
    ovhip_dbf_view v;
    view_of(&v, c);
    return rec_dbf_view(r, &v);
}

/* df.rcn_dbf_ctu for n consecutive CTUs (a CTU row, or n = 1), their maps read IN PLACE out of the caller's struct DBFInfo: no 9 KB
 * descriptor is filled and copied per CTU (510 of them per 4K picture).  The segments leave in the order ovhip_rec_dbf_ctu emits
 * them, CTU by CTU. */
int
ovhip_rec_dbf_row(ovhip_recorder *r, const ovhip_dbf_view *ctus, size_t n)
{
    if (!r || (!ctus && n)) return OVHIP_EINVAL;
    for (size_t i = 0; i < n; ++i) {
        const ovhip_dbf_view *c = &ctus[i];
        if (c->log2_ctu_s < 5 || c->log2_ctu_s > 7 || !c->ctb_bound_ver || !c->ctb_bound_hor || !c->ctb_bound_ver_c || !c->ctb_bound_hor_c || !c->aff_edg_ver
            || !c->aff_edg_hor || !c->bs2_ver || !c->bs2_hor || !c->bs2c_ver || !c->bs2c_hor || !c->bs1_ver || !c->bs1_hor || !c->bs1cb_ver || !c->bs1cb_hor
            || !c->bs1cr_ver || !c->bs1cr_hor || !c->affine_ver || !c->affine_hor || !c->qp_y || !c->qp_cb || !c->qp_cr)
            return OVHIP_EINVAL;
        if (r->log) { ovhip_dbf_ctu tmp; ctu_of(&tmp, c); ovhip_calllog_dbf_(r->log, &tmp); }
        const int q = rec_dbf_view(r, c);
        if (q != OVHIP_OK) return q;
    }
    return OVHIP_OK;
}

static int
rec_dbf_view(ovhip_recorder *r, const ovhip_dbf_view *c)
{
    if (dbf_alloc(r)) return OVHIP_ENOMEM;
    /* at most one luma segment per 4x4 unit and one per chroma plane and 8x4 / 4x8 unit, per direction */
    enum { CTU_EDGES = 32 * 32 + 2 * 8 * 32 };
    if ((r->n_edge_v + CTU_EDGES > r->cap_edge_v && ovhip_rec_grow_(r, (void **)&r->edge_v, &r->cap_edge_v, r->n_edge_v + CTU_EDGES, sizeof(ovhip_dbf_edge)))
        || (r->n_edge_h + CTU_EDGES > r->cap_edge_h && ovhip_rec_grow_(r, (void **)&r->edge_h, &r->cap_edge_h, r->n_edge_h + CTU_EDGES, sizeof(ovhip_dbf_edge))))
        return OVHIP_ENOMEM;
    const int oi = offset_index(r, c->beta_offset, c->tc_offset);
    if (oi < 0) return OVHIP_EUNSUP;         /* more than OVHIP_DBF_MAX_OFFSETS distinct slice offset pairs */
    /* the dense planes (legacy launch) carry ONE pair: only valid while the picture has a single one */
    r->dbf_beta_offset = c->beta_offset; r->dbf_tc_offset = c->tc_offset;
    const int dense = r->dense_planes;

    const int w4 = r->dbf_w4, h4 = r->dbf_h4, w4c = (w4 + 1) >> 1;
    const int nb_full = (1 << c->log2_ctu_s) >> 2;
    const int nb_w = c->ctu_w ? c->ctu_w >> 2 : nb_full, nb_h = c->ctu_h ? c->ctu_h >> 2 : nb_full;
    const int ux0 = c->ctb_x * nb_full, uy0 = c->ctb_y * nb_full;
    const int skip_v = !c->ctu_lft, skip_h = !c->ctu_abv;
    const uint64_t vmask = nb_h >= 64 ? ~0ull : ((1ull << nb_h) - 1);
    const int hbits = nb_w + (c->last_x ? 2 : 0);
    const uint64_t hmask = hbits >= 64 ? ~0ull : ((1ull << hbits) - 1);

    /* ---------------- luma, vertical edges (vvc_dbf_ctu_hor) ----------------
     * The masks are per unit column; the segments leave row by row (neighbours in the list = neighbours in a picture row: the
     * lanes of a wave of k_dbf_list<0> then share cache lines; column by column costs the device 30 % more time and traffic). */
    if (!c->disable_h) {
        const uint64_t *edg = &c->ctb_bound_ver[8], *sb = &c->aff_edg_ver[8];
        uint32_t row[32];                       /* the column masks transposed: row[j] bit i = column i has a segment in unit row j */
        struct edge_ctx ctx[32];
        memset(row, 0, sizeof(row));
        for (int i = skip_v; i < nb_w; ++i) {
            const uint64_t m = (edg[i] | sb[i]) & vmask & (c->bs2_ver[i] | c->bs1_ver[i]);
            row[i] = (uint32_t)m;               /* (a CTU has at most 32 unit rows) */
            if (m) edge_context(&ctx[i], edg, sb, i, 1);
        }
        transpose32(row);                       /* 160 word operations instead of one read-modify-write per segment (~770 per CTU) */
        for (int j = 0; j < nb_h && j < 32; ++j) {
            const uint64_t pos = 1ull << j;
            const int uy = uy0 + j;
            if (uy >= h4) break;
            for (uint32_t rm = row[j]; rm; rm &= rm - 1) {
                const int i = __builtin_ctz(rm), ux = ux0 + i;
                if (ux >= w4) break;
                const int bs = 1 + !!(c->bs2_ver[i] & pos);
                const uint8_t *q = &c->qp_y[36 + i + 34 * j];
                int qp = (q[-1] + q[0] + 1) >> 1, lp, lq;
                filter_length(&ctx[i], c->affine_ver[i], c->affine_ver[i + 1], pos, &lp, &lq);
                emit_edge(r, 0, 0, ux, uy, OVHIP_DBF_LUMA(bs, lp, lq, qp & 255), oi, dense ? &r->dbf_luma_v[uy * w4 + ux] : NULL);
            }
        }
    }
    /* ---------------- luma, horizontal edges (vvc_dbf_ctu_ver), shifted 2 units left ---------------- */
    if (!c->disable_v) {
        const uint64_t *edg = &c->ctb_bound_hor[8], *sb = &c->aff_edg_hor[8];
        for (int i = skip_h; i < nb_h; ++i) {
            uint64_t m = (edg[i] | sb[i]) & hmask & (c->bs2_hor[i] | c->bs1_hor[i]);
            if (!m) continue;
            struct edge_ctx e;
            edge_context(&e, edg, sb, i, i >= 7);
            while (m) {
                int k = __builtin_ctzll(m);
                m &= m - 1;
                uint64_t pos = 1ull << k;
                int bs = 1 + !!(c->bs2_hor[i] & pos);
                const uint8_t *q = &c->qp_y[34 * i + k];
                int qp = (q[0] + q[34] + 1) >> 1, lp, lq;
                filter_length(&e, c->affine_hor[i], c->affine_hor[i + 1], pos, &lp, &lq);
                int ux = ux0 + k - 2, uy = uy0 + i;
                if (ux >= 0 && ux < w4 && uy < h4) emit_edge(r, 1, 0, ux, uy, OVHIP_DBF_LUMA(bs, lp, lq, qp & 255), oi, dense ? &r->dbf_luma_h[uy * w4 + ux] : NULL);
            }
        }
    }
    /* ---------------- chroma, vertical edges on the 8-sample grid (vvc_dbf_chroma_hor), row by row as well ---------------- */
    if (!c->disable_h) {
        const uint64_t *tab = &c->ctb_bound_ver_c[8];
        const int nb_vedge = (nb_w + 3) >> 2;
        for (int comp = 0; comp < 2; ++comp) {
            const uint64_t *bs1v = comp ? c->bs1cr_ver : c->bs1cb_ver;
            const uint8_t *qpm = comp ? c->qp_cr : c->qp_cb;
            uint16_t *plane = comp ? r->dbf_cr_v : r->dbf_cb_v;
            uint64_t todo[8], large[8];
            uint8_t row[32];
            memset(row, 0, sizeof(row));
            for (int i = skip_v; i < nb_vedge; ++i) {
                const int idx = i << 2;
                const uint64_t bs2 = c->bs2c_ver[idx], bs1 = bs1v[idx];
                uint64_t m = tab[idx] & vmask & (bs2 | bs1);
                large[i] = m ? large_from_ngh(&tab[idx]) : 0;
                m &= bs2 | (bs1 & large[i]);
                todo[i] = m;
                while (m) { row[__builtin_ctzll(m)] |= (uint8_t)(1u << i); m &= m - 1; }
            }
            for (int j = 0; j < nb_h && j < 32; ++j) {
                const int uy = uy0 + j;
                if (uy >= h4) break;
                for (uint32_t rm = row[j]; rm; rm &= rm - 1) {
                    const int i = __builtin_ctz(rm), idx = i << 2, ux = ux0 + idx;
                    if (ux >= w4) break;
                    const uint8_t *q = &qpm[36 + idx + 34 * j];
                    const int qp = (q[-1] + q[0] + 1) >> 1;
                    const uint16_t word = (uint16_t)(OVHIP_DBF_C_ON | (((c->bs2c_ver[idx] >> j) & 1) ? OVHIP_DBF_C_BS2 : 0)
                                                     | (((large[i] >> j) & 1) ? OVHIP_DBF_C_LARGE : 0) | ((qp & 255) << 8));
                    emit_edge(r, 0, 1 + comp, ux, uy, word, oi, dense ? &plane[uy * w4c + (ux >> 1)] : NULL);
                }
            }
            (void)todo;
        }
    }
    /* ---------------- chroma, horizontal edges (vvc_dbf_chroma_ver), shifted 2 units left ---------------- */
    if (!c->disable_v) {
        const uint64_t *tab = &c->ctb_bound_hor_c[8];
        const int nb_hedge = (nb_h + 3) >> 2;
        for (int comp = 0; comp < 2; ++comp) {
            const uint64_t *bs1h = comp ? c->bs1cr_hor : c->bs1cb_hor;
            const uint8_t *qpm = comp ? c->qp_cr : c->qp_cb;
            uint16_t *plane = comp ? r->dbf_cr_h : r->dbf_cb_h;
            for (int i = skip_h; i < nb_hedge; ++i) {
                const int idx = i << 2;
                uint64_t bs2 = c->bs2c_hor[idx], bs1 = bs1h[idx];
                uint64_t m = tab[idx] & hmask & (bs2 | bs1);
                if (!m) continue;
                uint64_t large = large_from_ngh(&tab[idx]);
                m &= bs2 | (bs1 & large);
                while (m) {
                    int k = __builtin_ctzll(m);
                    m &= m - 1;
                    const uint8_t *q = &qpm[idx * 34 + k];
                    int qp = (q[0] + q[34] + 1) >> 1;
                    int ux = ux0 + k - 2, uy = uy0 + idx;
                    const uint16_t word = (uint16_t)(OVHIP_DBF_C_ON | (((bs2 >> k) & 1) ? OVHIP_DBF_C_BS2 : 0)
                                                     | (((large >> k) & 1) ? OVHIP_DBF_C_LARGE : 0)
                                                     | (i == 0 ? OVHIP_DBF_C_CTB_B : 0) | ((qp & 255) << 8));
                    if (ux >= 0 && ux < w4 && uy < h4) emit_edge(r, 1, 1 + comp, ux, uy, word, oi, dense ? &plane[(uy >> 1) * w4 + ux] : NULL);
                }
            }
        }
    }
    return OVHIP_OK;
}

const ovhip_dbf_edge *
ovhip_rec_dbf_edges(const ovhip_recorder *r, int dir, size_t *n, ovhip_dbf_offsets *offsets)
{
    if (!r || !n || (dir != 0 && dir != 1)) return NULL;
    *n = dir ? r->n_edge_h : r->n_edge_v;
    if (offsets) *offsets = r->dbf_off;
    return dir ? r->edge_h : r->edge_v;
}

/* ---------------------------------------------------------------- compact edge lists
 * The planes are dense (one word per 4-sample segment); about three quarters of the words are 0.  The device
 * kernel driven by the planes spends most lanes on "no edge here"; the lists keep only the segments to filter.
 * The skip rules are the ones k_dbf applies: bS 0 / chroma bit off, and never across the picture boundary. */
int64_t
ovhip_dbf_compact(const ovhip_dbf_planes *pl, int dir, ovhip_dbf_edge *out, size_t cap)
{
    if (!pl || (dir != 0 && dir != 1) || pl->w4 <= 0 || pl->h4 <= 0) return OVHIP_EINVAL;
    const int w4 = pl->w4, h4 = pl->h4;
    size_t n = 0;
    const uint16_t *luma = dir ? pl->luma_h : pl->luma_v;
    if (!luma) return OVHIP_EINVAL;
    for (int uy = 0; uy < h4; ++uy)
        for (int ux = 0; ux < w4; ++ux) {
            const uint16_t v = luma[uy * w4 + ux];
            if (!(v & 3) || (dir ? uy : ux) == 0) continue;
            if (out && n < cap) { ovhip_dbf_edge e = { (uint16_t)ux, (uint16_t)uy, v, 0, 0 }; out[n] = e; }
            ++n;
        }
    /* chroma planes: vertical [h4][ceil(w4/2)] (every second unit column), horizontal [ceil(h4/2)][w4] */
    const int cw = dir ? w4 : (w4 + 1) >> 1, chh = dir ? (h4 + 1) >> 1 : h4;
    for (int comp = 1; comp < 3; ++comp) {
        const uint16_t *plane = dir ? (comp == 1 ? pl->cb_h : pl->cr_h) : (comp == 1 ? pl->cb_v : pl->cr_v);
        if (!plane) return OVHIP_EINVAL;
        for (int cy = 0; cy < chh; ++cy)
            for (int cx = 0; cx < cw; ++cx) {
                const uint16_t v = plane[cy * cw + cx];
                const int ux = dir ? cx : cx * 2, uy = dir ? cy * 2 : cy;
                if (!(v & OVHIP_DBF_C_ON) || (dir ? uy : ux) == 0) continue;
                if (out && n < cap) { ovhip_dbf_edge e = { (uint16_t)ux, (uint16_t)uy, v, (uint8_t)comp, 0 }; out[n] = e; }
                ++n;
            }
    }
    return (int64_t)n;
}

/* ---------------------------------------------------------------- MV-based bS pre-pass (P / B slices)
 * dbf_ctu_preproc_h/_v (rcn_df.c:1821-1874): for every unit row / column, the CU and affine sub-block edges
 * that have neither bS 2 nor bS 1 yet are examined with the motion on their two sides. */
typedef struct { int32_t x, y; int ref; } mvq;

static mvq mv_at(const void *base, int bytes, int idx)
{
    const uint8_t *p = (const uint8_t *)base + (size_t)idx * bytes;
    mvq m;
    memcpy(&m.x, p, 4); memcpy(&m.y, p + 4, 4);
    m.ref = (int8_t)p[8];
    return m;
}

static int mv_far(mvq a, mvq b)          /* mv_threshold_check, rcn_df.c:1513-1522 */
{
    int64_t dx = (int64_t)a.x - b.x, dy = (int64_t)a.y - b.y;
    if (dx < 0) dx = -dx;
    if (dy < 0) dy = -dy;
    return dx >= 8 || dy >= 8;
}

/* check_dbf_enabled (rcn_df.c:1542-1576): both sides bi-predicted */
static int bs_bi(const ovhip_dbf_mv_ctx *c, mvq p0, mvq p1, mvq q0, mvq q1)
{
    const int r0p = c->dist_ref0[p0.ref], r1p = c->dist_ref1[p1.ref], r0q = c->dist_ref0[q0.ref], r1q = c->dist_ref1[q1.ref];
    const int paired = r0p == r0q && r1p == r1q, swapped = r0p == r1q && r1p == r0q;
    int bs = 1;
    if (r0p == r1p && paired) {
        bs  = mv_far(q0, p0) || mv_far(q1, p1);
        bs &= mv_far(q1, p0) || mv_far(q0, p1);
    } else if (paired) {
        bs = mv_far(q0, p0) | mv_far(q1, p1);
    } else if (swapped) {
        bs = mv_far(q1, p0) | mv_far(q0, p1);
    }
    return bs;
}

/* One row (dir = 1, horizontal edges above unit row y) or column (dir = 0, vertical edges left of unit column x):
 * dbf_mv_set_hedges / dbf_mv_set_vedges (rcn_df.c:1578-1819).  `todo` = edges without a strength yet. */
static uint64_t
mv_edges(const ovhip_dbf_mv_ctx *c, int dir, int pos, int n_units, uint64_t todo)
{
    const uint64_t *f0 = dir ? c->map0_h : c->map0_v, *f1 = dir ? c->map1_h : c->map1_v, *fi = dir ? c->ibc_h : c->ibc_v;
    const uint64_t unit_msk = ((uint64_t)1 << n_units) - 1;
    const int sh = dir ? 2 : 0;                       /* horizontal maps carry a 2-unit left margin */
    const uint64_t p0 = (f0[pos] >> 1) & unit_msk, p1 = (f1[pos] >> 1) & unit_msk, pi = (fi[pos] >> 1) & unit_msk;
    const uint64_t q0 = (f0[pos + 1] >> 1) & unit_msk, q1 = (f1[pos + 1] >> 1) & unit_msk, qi = (fi[pos + 1] >> 1) & unit_msk;

    uint64_t keep = ((~todo) >> sh) & unit_msk;       /* edges that already have a strength (or are no edge) */
    keep |= pi & (q1 | q0);                           /* IBC against inter: always bS 1 */
    keep |= qi & (p1 | p0);

    const uint64_t q_b = q0 & q1, q_l0 = q0 & ~q1, q_l1 = q1 & ~q0;
    const uint64_t p_b = p0 & p1, p_l0 = p0 & ~p1, p_l1 = p1 & ~p0;
    const uint64_t both_ibc = pi & qi;
    uint64_t chk_b = q_b & p_b, chk_l0 = q_l0 & (p_l0 | p_l1), chk_l1 = q_l1 & (p_l0 | p_l1);
    chk_b &= ~(keep | both_ibc) & unit_msk;
    chk_l0 &= ~(keep | both_ibc) & unit_msk;
    chk_l1 &= ~(keep | both_ibc) & unit_msk;

    /* everything that is not compared below gets bS 1, except IBC against IBC */
    uint64_t out = both_ibc ^ (~(chk_l0 | chk_l1 | chk_b) & unit_msk);

    /* motion of unit k on the P side (previous row / column) and the Q side: index 35 + x + 34 * y with a 1-unit border */
    const int step = dir ? 1 : 34;
    const int ip = dir ? 35 + 34 * (pos - 1) : 35 + (pos - 1), iq = dir ? 35 + 34 * pos : 35 + pos;
    for (int k = 0; k < n_units; ++k) {
        const uint64_t bit = (uint64_t)1 << k;
        if (chk_b & bit) {
            const int bs = bs_bi(c, mv_at(c->mvs0, c->mv_bytes, ip + k * step), mv_at(c->mvs1, c->mv_bytes, ip + k * step),
                                 mv_at(c->mvs0, c->mv_bytes, iq + k * step), mv_at(c->mvs1, c->mv_bytes, iq + k * step));
            out |= (uint64_t)bs << k;
        }
        if ((chk_l0 | chk_l1) & bit) {
            /* check_dbf_enabled_p (rcn_df.c:1526-1539): one list on each side */
            const int p_is_l0 = !!(p_l0 & bit), q_is_l0 = !!(chk_l0 & bit);
            const mvq mp = mv_at(p_is_l0 ? c->mvs0 : c->mvs1, c->mv_bytes, ip + k * step);
            const mvq mq = mv_at(q_is_l0 ? c->mvs0 : c->mvs1, c->mv_bytes, iq + k * step);
            const int rp = (p_is_l0 ? c->dist_ref0 : c->dist_ref1)[mp.ref], rq = (q_is_l0 ? c->dist_ref0 : c->dist_ref1)[mq.ref];
            const int bs = rp == rq ? mv_far(mq, mp) : 1;
            out |= (uint64_t)bs << k;
        }
    }
    return ((out | keep) << sh) & todo;
}

static int
mv_prepass(uint64_t *bs1_ver, uint64_t *bs1_hor, const uint64_t *bs2_ver, const uint64_t *bs2_hor, const uint64_t *aff_edg_ver, const uint64_t *aff_edg_hor,
           int log2_ctu_s, int ctu_w, int ctu_h, const ovhip_dbf_mv_ctx *mv)
{
    if (!mv || !mv->mvs0 || !mv->mvs1 || mv->mv_bytes < 9) return OVHIP_EINVAL;
    const int full = 1 << log2_ctu_s;
    const int nb_w = (ctu_w && ctu_w < full ? ctu_w : full) >> 2;
    const int nb_h = (ctu_h && ctu_h < full ? ctu_h : full) >> 2;
    for (int i = 0; i < nb_h; ++i) {                                   /* dbf_ctu_preproc_h */
        const uint64_t edges = mv->cu_edge_hor[i] | aff_edg_hor[8 + i];
        const uint64_t todo = edges ^ ((bs2_hor[i] | bs1_hor[i]) & edges);
        if (todo) bs1_hor[i] |= mv_edges(mv, 1, i, nb_w, todo);
    }
    for (int i = 0; i < nb_w; ++i) {                                   /* dbf_ctu_preproc_v */
        const uint64_t edges = mv->cu_edge_ver[i] | aff_edg_ver[8 + i];
        const uint64_t todo = edges ^ ((bs2_ver[i] | bs1_ver[i]) & edges);
        if (todo) bs1_ver[i] |= mv_edges(mv, 0, i, nb_h, todo);
    }
    return OVHIP_OK;
}

int
ovhip_rec_dbf_mv_prepass(ovhip_dbf_ctu *ctu, const ovhip_dbf_mv_ctx *mv)
{
    if (!ctu) return OVHIP_EINVAL;
    return mv_prepass(ctu->bs1_ver, ctu->bs1_hor, ctu->bs2_ver, ctu->bs2_hor, ctu->aff_edg_ver, ctu->aff_edg_hor, ctu->log2_ctu_s, ctu->ctu_w, ctu->ctu_h, mv);
}

/* the same on the caller's own maps (what the scalar slot does to dbf_info->bs1_map): bs1_ver / bs1_hor = the two writable arrays
 * the view's bs1_ver / bs1_hor point at */
int
ovhip_rec_dbf_mv_prepass_view(const ovhip_dbf_view *v, uint64_t *bs1_ver, uint64_t *bs1_hor, const ovhip_dbf_mv_ctx *mv)
{
    if (!v || !bs1_ver || !bs1_hor || !v->bs2_ver || !v->bs2_hor || !v->aff_edg_ver || !v->aff_edg_hor) return OVHIP_EINVAL;
    return mv_prepass(bs1_ver, bs1_hor, v->bs2_ver, v->bs2_hor, v->aff_edg_ver, v->aff_edg_hor, v->log2_ctu_s, v->ctu_w, v->ctu_h, mv);
}
